// mlp_f16_dev.h - device building blocks of the split-f16 (INERF_PREC_F16X3) kernels: packed-weight fetches,
// the pipelined wide GEMM over LDS-resident activations, the hi/lo-splitting epilogue, the skinny head GEMM.
// Shared by mlp_f16.hip (forward) and mlp_bwd.hip (input-gradient chain).  Arithmetic is described at the top of
// mlp_f16.hip.
#pragma once
#include <type_traits>

#include "mlp_common.h"

namespace inerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Weight / bias fetches go through ONE buffer descriptor over the packed blob: the per-lane part of the
// address (lane * 16 B) is a single VGPR shared by every layer and the layer / wave / k-block part is a
// wave-uniform SGPR offset.  With plain 64-bit global addresses the compiler materialised a VGPR address
// pair per layer, kept ~60 of them live across the tile loop and spilled them to scratch - whose 43 MB
// footprint then thrashed the L2 the weights are supposed to stay in.
// cache policy of the weight-fragment stream (bit 0 sc0, bit 1 nt, bit 4 sc1): every wave streams its own 650 KB per tile from L2.
// Round 5, same box: nt 315 vs 410 TFLOP/s (the blob no longer stays in L2), sc0 neutral in all three MLP kernels; default kept.
#ifndef INERF_WEIGHT_AUX
#define INERF_WEIGHT_AUX 0
#endif
struct WeightBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;                                     // lane * 16 bytes
    __device__ __forceinline__ f16x8 frag(int byte_off) const {          // byte_off: wave-uniform
#ifdef INERF_ABL_WLOAD_L1       // (timing ablation of a development build, scripts/build_variant.sh: every fragment from one 4 KiB window - the same
        byte_off &= 0xFFF;      // instructions, no L2 -> CU stream; results are wrong)
#endif
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, byte_off, INERF_WEIGHT_AUX));
    }
    __device__ __forceinline__ f32x4 vec4(int byte_off, int lane_bytes) const {   // small per-lane offset on top
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_bytes, byte_off, 0));
    }
    __device__ __forceinline__ float scalar(int byte_off) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, 0, byte_off, 0));
    }
};

// Cache policy of the fragment stores (the training kernels' 1 KB operand fragments: 4-5 GB per launch, written once, read once by
// another kernel): bit 0 sc0, bit 1 nt, bit 4 sc1.  Non-temporal: same-box A/B on the fine batch (profiles/r05_frag_store_policy.txt)
// training forward 1.79-1.85 -> 1.76-1.78 ms, chain 1.89-1.97 -> 1.86-1.93; sc1 (write-through) is slower, sc0 neutral.  (Without
// the stores at all: 1.48 / 1.61 ms - what the stores cost is not their bandwidth, 2.4 TB/s, but that every later load of the wave
// - the next GEMM's weight stream - is counted behind them in vmcnt.)
#ifndef INERF_FRAG_AUX
#define INERF_FRAG_AUX 2
#endif
constexpr int kRowH = kColB + kWidth + 8;      // 616 halfs per row
constexpr int kPlaneH = kTilePoints * kRowH;   // halfs per plane
constexpr int kLdsBytesH = 2 * kPlaneH * 2;    // 157,696
constexpr float kF16Safe = 6.0e4f;             // on the scaled value (kActScale * activation)

// `amax` is a per-thread running maximum of |v| over everything that was split into f16; it is compared
// with the representable range once, at the end of the kernel (a per-value compare-and-flag made the
// compiler keep every |v| alive and spill).
template <int PLANE = kPlaneH>
__device__ __forceinline__ void split_store(_Float16* hi_ptr, float v, float& amax) {
    const float t = v * kActScale;
    const _Float16 h = (_Float16)t;
    hi_ptr[0] = h;
    hi_ptr[PLANE] = (_Float16)(t - (float)h);
    amax = fmaxf(amax, fabsf(t));
}

// sin and cos of an encoding argument a = x * 2^f (|a| up to 6 * 2^9 on the chair scene, 2^9 on the room's x / 10).
// ocml's sincosf spends ~100 instructions (and diverges into its large-argument path) per call; the encodings are 24
// calls per sample point.  This is the classic three-constant Cody-Waite reduction, exact to ~6e-8 in the reduced
// argument for |a| < 2^15 thanks to the fused multiply-adds (pi/2 = c1 + c2 + c3 to 2^-72), followed by the cephes
// single-precision minimax polynomials on [-pi/4, pi/4]: max |error| 9.2e-8 against a double-precision evaluation for
// |a| <= 3e4 (libm's own float sin: 7e-8; measured over 1e7 arguments).  Larger arguments take ocml's path.
__device__ __forceinline__ void fast_sincosf(float a, float* sn, float* cs) {
    if (__builtin_expect(!(fabsf(a) < 32768.0f), 0)) { sincosf(a, sn, cs); return; }
    const float j = __builtin_rintf(a * 0.636619772367581343f);
    float r = __builtin_fmaf(-j, 1.5707963705062866f, a);
    r = __builtin_fmaf(-j, -4.371138828673793e-08f, r);
    r = __builtin_fmaf(-j, -1.7763568394002505e-15f, r);
    const float z = r * r;
    float sp = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
    const float s = __builtin_fmaf(sp * z, r, r);
    float cp = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
    const float c = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const int q = (int)j;
    const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

// ------------------------------------------------------------------------------------------------
// wide GEMM: RB blocks of 32 channels per wave x 2 blocks of 32 points, K = 16 * (KB0 + KB1)
// ------------------------------------------------------------------------------------------------
#ifndef INERF_EPI_PKFMA
#define INERF_EPI_PKFMA 0
#endif
#ifndef INERF_GEMM_PRIO
#define INERF_GEMM_PRIO 1      // s_setprio level inside the wide GEMM loops (0: none)
#endif

template <int RB>
struct WidePreH {
    f16x8 w[2][RB][2];      // k-block 0/1, row block, hi/lo
    f32x4 b[RB][4];         // bias (already in the scaled activation domain)
    float inv;              // accumulator -> scaled output factor (2^-kw)
};

template <int RB>
__device__ __forceinline__ void wide_prefetch_h(WidePreH<RB>& pre, const WeightBuf& wb, int frag_bytes /* wave's stream */,
                                                int bias_bytes /* wave's first channel */, int scale_bytes, int lane) {
    const int h16 = 16 * (lane >> 5);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int part = 0; part < 2; ++part) pre.w[kb][rb][part] = wb.frag(frag_bytes + ((kb * RB + rb) * 2 + part) * 1024);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) pre.b[rb][g] = wb.vec4(bias_bytes + (32 * rb + 8 * g) * 4, h16);
    pre.inv = wb.scalar(scale_bytes);
}

// KSTRIDE: bytes between a wave's consecutive k-blocks in the packed stream (default: its RB row blocks are contiguous; a
// wave that takes ONE of the two row blocks of the 4-wave packing passes 4096).
// PB: blocks of 32 points (2: the whole tile; 1: the 32 points `xl` points at).  RBSTRIDE: bytes between the wave's row blocks
// (default: consecutive inside a k-block; a wave that takes two 32-channel STREAMS of a 128-channel layer's 4-wave packing
// passes the size of one stream).
// PEEL: the first four k-blocks stand in front of the loop and, with ZERO, the first product of every accumulator takes a zero C operand (an
// inline constant) instead of 16 * RB * PB v_mov in front of the GEMM - exposed instructions: the matrix pipe is empty until they are through.
template <int RB, int KB0, int KB1, int ROW = kRowH, int PLANE = kPlaneH, bool ZERO = true, int KSTRIDE = RB * 2048, int PB = 2, int RBSTRIDE = 2048,
          bool PEEL = false>
__device__ __forceinline__ void wide_gemm_h(const WidePreH<RB>& pre, const WeightBuf& wb, int frag_bytes,
                                            const _Float16* xl,      // plane_hi + (lane&31)*kRowH + 8*(lane>>5)
                                            int col0, int col1, int lane, f32x16 (&am)[RB][PB]) {
    constexpr int KBT = KB0 + KB1;
    static_assert(KBT % 2 == 0 && KBT >= 4, "k-block count");
    static_assert(PB == 2 || (PB == 1 && RB == 2) || (PB == 4 && RB == 1), "point blocks");
    if constexpr (ZERO && !PEEL) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) am[rb][pb][4 * g + i] = 0.0f;
    }
    const f32x16 zero_c = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    auto xoff = [&](int kb) { return kb < KB0 ? col0 + 16 * kb : col1 + 16 * (kb - KB0); };
    // 4 rotating weight buffers (two k-blocks ahead; three measured slower), 2 activation buffers (one ahead);
    // all indices static
    f16x8 w[4][RB][2], x[2][PB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int part = 0; part < 2; ++part) { w[0][rb][part] = pre.w[0][rb][part]; w[1][rb][part] = pre.w[1][rb][part]; }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int part = 0; part < 2; ++part)
            x[0][pb][part] = *reinterpret_cast<const f16x8*>(xl + part * PLANE + xoff(0) + pb * 32 * ROW);
    // PEEL forms: the lo plane through a base register of its own (laundered: derived from `xl` by a constant the compiler re-adds it per read -
    // PLANE is beyond the 64 KB an LDS instruction's immediate offset reaches - 17 v_add per 48 MFMAs of the 128-point tile's loop, now 2)
    // (the OFFSET is laundered, not the pointer: a laundered pointer loses its LDS address space and the reads become flat loads)
    int lo_off = PLANE;
    if constexpr (PEEL) asm volatile("" : "+v"(lo_off));
    const _Float16* xl_lo = xl + lo_off;

// one k-block: request operands for k+2 (weights) / k+1 (activations) and run 3*RB*PB MFMAs on block k, with the
// 2*RB global loads and 2*PB LDS reads interleaved BETWEEN the MFMAs (an f16 MFMA occupies the pipe for only 32
// cycles, so a burst of 8 memory instructions ahead of them is not hidden; measured +x % vs the burst form).
// A macro, not a lambda: the buffer indices must stay compile-time constants for the arrays to live in registers.
#define INERF_F16_STEP(K, I) INERF_F16_STEP_(K, I, false)
#define INERF_F16_STEP_(K, I, FIRST)                                                                                 \
    {                                                                                                                \
        const int k1_ = (K) + 1 < KBT ? (K) + 1 : KBT - 1;                                                           \
        const int k2_ = (K) + 2 < KBT ? (K) + 2 : KBT - 1;                                                           \
        /* one column range (KB1 = 0): the NEXT k-block's columns without the clamp - affine in the loop counter, so the four steps of an   \
           iteration share one address per plane (+ immediates) instead of one v_add per read; the last step then reads 16 columns past    \
           the layer's input (inside the row: unused values) */                                                                            \
        const int xo_ = (KB1 == 0 && PEEL) ? col0 + 16 * ((K) + 1) : xoff(k1_);                                      \
        _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                                            \
            _Pragma("unroll") for (int part = 0; part < 2; ++part)                                                   \
                w[((I) + 2) & 3][rb][part] = wb.frag(frag_bytes + k2_ * KSTRIDE + rb * RBSTRIDE + part * 1024);      \
        _Pragma("unroll") for (int pb = 0; pb < PB; ++pb)                                                            \
            _Pragma("unroll") for (int part = 0; part < 2; ++part)                                                   \
                x[((I) + 1) & 1][pb][part] =                                                                         \
                    *reinterpret_cast<const f16x8*>((PEEL && part ? xl_lo : xl + part * PLANE) + xo_ + pb * 32 * ROW); \
        /* hi*hi, hi*lo, lo*hi into the same accumulator; product-major: an accumulator is touched every 4th MFMA */ \
        _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                                            \
            _Pragma("unroll") for (int pb = 0; pb < PB; ++pb)                                                        \
                am[rb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(I) & 3][rb][0], x[(I) & 1][pb][0], (FIRST) ? zero_c : am[rb][pb], 0, 0, 0); \
        _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                                            \
            _Pragma("unroll") for (int pb = 0; pb < PB; ++pb)                                                        \
                am[rb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(I) & 3][rb][0], x[(I) & 1][pb][1], am[rb][pb], 0, 0, 0); \
        _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                                            \
            _Pragma("unroll") for (int pb = 0; pb < PB; ++pb)                                                        \
                am[rb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(I) & 3][rb][1], x[(I) & 1][pb][0], am[rb][pb], 0, 0, 0); \
        /* issue order (masks: 0x8 MFMA, 0x20 VMEM read, 0x100 DS read).  PB = 2: MFMA, global load, MFMA, LDS read(s), MFMA - 2*RB  \
           times; PB = 1 (RB = 2: 6 MFMAs, 4 global loads, 2 LDS reads): MFMA, global load, MFMA, global load, MFMA, LDS read - twice; \
           PB = 4 (RB = 1, the 128-point tile: 12 MFMAs, 2 global loads, 8 LDS reads): the LDS reads (needed one step ahead) first */ \
        if constexpr (PB == 4) {                                                                                     \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                         \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
            }                                                                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                       \
        } else if constexpr (PB == 2) {                                                                              \
            _Pragma("unroll") for (int q = 0; q < 2 * RB; ++q) {                                                    \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x100, 4 / (2 * RB), 0);                                        \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
            }                                                                                                        \
        } else {                                                                                                     \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                         \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                   \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
    constexpr int KB4 = KBT & ~3;
#if INERF_GEMM_PRIO               // the wave inside a GEMM loop wins the issue arbitration against its SIMD's other wave (the other
    __builtin_amdgcn_s_setprio(INERF_GEMM_PRIO);      // workgroup's, in an epilogue): its MFMAs and weight requests are not queued behind
#endif                                                // the other's VALU / store bursts; +0.6 % inference, +1.6 % SSR, +1 % training kernels
    if constexpr (PEEL) {
        static_assert(KB4 >= 4, "peeled k-blocks");
        INERF_F16_STEP_(0, 0, ZERO)
        INERF_F16_STEP(1, 1)
        INERF_F16_STEP(2, 2)
        INERF_F16_STEP(3, 3)
    }
#pragma unroll 1
    for (int kb = PEEL ? 4 : 0; kb < KB4; kb += 4) {
        INERF_F16_STEP(kb + 0, 0)
        INERF_F16_STEP(kb + 1, 1)
        INERF_F16_STEP(kb + 2, 2)
        INERF_F16_STEP(kb + 3, 3)
    }
    if constexpr (KBT - KB4 == 2) {
        INERF_F16_STEP(KB4 + 0, 0)
        INERF_F16_STEP(KB4 + 1, 1)
    }
#if INERF_GEMM_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#undef INERF_F16_STEP
#undef INERF_F16_STEP_
}

// epilogue: t = acc * inv + bias' (= kActScale * layer output; optionally ReLU) -> hi/lo planes;
// optional fp32 copy of the unscaled value to global.
// Split by truncation: t_hi = t with the 13 low mantissa bits cleared is exactly representable in f16 (for
// |t| in f16's normal range), so hi = f16(t_hi) needs no rounding and lo = f16(t - t_hi) is the exact
// remainder rounded once - one AND and one SUB per element instead of convert / convert back / subtract.
// Range check: packed f16 max over the |hi| pairs (inf when |t| > 65504).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// The f16 range guard, once per tile: `amax` / `amax2` are the tile's running maxima (reset at the top of the tile).  A tile that left
// the range ORs INERF_STATUS_F16_RANGE into the status word of every ray it holds points of: ONE word per launch, or - with
// MlpParams.status_rays (inerf_encode_mlp_chunked) - one word per that many rays, so that a frame rendered as one launch still tells
// which of the caller's chunks has to be rendered again in exact fp32.  Returns the tile's maximum (scaled domain).
__device__ __forceinline__ float flag_f16_range(const MlpParams& p, int first_point, int tile_points, float amax, f16x2 amax2, int lane) {
    const float m = fmaxf(amax, fmaxf((float)amax2[0], (float)amax2[1]));
    if (p.status && __any(!(m <= kF16Safe)) && lane == 0) {          // (never taken on a network inside the range)
        int w0 = 0, w1 = 0;
        if (p.status_rays > 0) {
            int last = first_point + tile_points - 1;
            last = last < p.n_points ? last : p.n_points - 1;
            w0 = (first_point / p.n_samples) / p.status_rays;
            w1 = (last / p.n_samples) / p.status_rays;
        }
        for (int w = w0; w <= w1; ++w) atomicOr(p.status + w, INERF_STATUS_F16_RANGE);
    }
    return m;
}


// running packed maximum of three: amax = maximum(amax, a, b) per f16 half in ONE instruction (v_pk_maximum3_f16, new in gfx950; IEEE
// maximum: a NaN propagates into the maximum, where the range guard's `!(amax <= safe)` sees it).  Two v_pk_max_f16 before round 6.
__device__ __forceinline__ void pk_max3_into(f16x2& amax, f16x2 a, f16x2 b) {
#ifdef INERF_NO_MAX3             // (A/B build: the two-instruction form)
    amax = __builtin_elementwise_max(amax, __builtin_elementwise_max(a, b));
    return;
#endif
    unsigned m = __builtin_bit_cast(unsigned, amax);
    asm("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(m) : "v"(__builtin_bit_cast(unsigned, a)), "v"(__builtin_bit_cast(unsigned, b)));
    amax = __builtin_bit_cast(f16x2, m);
}

// hi = (t0, t1) rounded toward zero to f16 (= the 13 low mantissa bits cleared, for |t| in f16's normal range; the
// conversion saturates at 65504 instead of overflowing), lo = f16(t - hi): the difference is exact in fp32, so lo is the
// remainder rounded once.  Three instructions per pair: v_cvt_pkrtz_f16_f32 and two v_fma_mix{lo,hi}_f16, which read hi
// as an f16 source, subtract it from the fp32 value and write the f16 result into one half of the destination - the
// compiler does not form them from C (it emits two ANDs, a packed fp32 subtract and two packed conversions).
__device__ __forceinline__ void split_pair(float t0, float t1, f16x2& hi, f16x2& lo) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(t0, t1));
    unsigned l;
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(t0));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(t1));
    hi = __builtin_bit_cast(f16x2, h);
    lo = __builtin_bit_cast(f16x2, l);
}

// The optional training copy goes through a buffer descriptor (one per activation slot, wave-uniform) and ONE per-lane
// byte offset shared by all slots: with 64-bit global addresses the compiler kept an address pair per layer alive across
// the tile loop and spilled ~300 registers.  Points beyond the end of the slot are dropped by the descriptor's range check.
struct SaveDst {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;          // bytes: (pt0 * width + chan0 + 4 * (lane >> 5)) * 4
    int stride;        // floats per point
};

// ------------------------------------------------------------------------------------------------
// Fragment slots (layout.h SaveSlot): a layer's output, which sits in the hi / lo planes as X[point][channel], leaves the CU as
// operand fragments of the weight-gradient products (lane = channel, 8 k-values = 8 sample points).  The transposition is done
// by the matrix core: a [32 points x 16 channels] piece of a plane, read like any B operand of the layer GEMMs (lane = point,
// 8 consecutive channels), is the A operand of an MFMA whose B is a 0/1 selector - D[point][channel] = sum_k A[point][k] Sel[k][channel]
// comes back in accumulator layout: lane = CHANNEL, registers = points (r&3) + 8 (r>>2) + 4 (lane>>5) = frag_point order.
// The products are f16 x 1.0: exact, and any rounding mode converts them back to the f16 they were.  Per [32 x 32] block
// and plane: 2 ds_read_b128, 2 MFMAs, 8 conversions and two 1 KB stores (one instruction each, every 128-byte line complete).
struct Selector { f16x8 k[2]; };      // per lane (n = lane & 31, h = lane >> 5): Sel[8 h + i][n] of k-block kb, i = 0..7
// operands read from the planes: k-slot 8 h + i of k-block kb is channel 16 kb + 8 h + i of the block
__device__ __forceinline__ Selector plane_selector(int lane, float value = 1.0f) {
    Selector s;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 8; ++i) s.k[kb][i] = (16 * kb + 8 * (lane >> 5) + i) == (lane & 31) ? (_Float16)value : (_Float16)0.0f;
    return s;
}
struct FragDst {
    __amdgpu_buffer_rsrc_t rsrc;      // the slot: n_tiles * kFragTileBytes bytes
    unsigned voff;                    // bytes: tile * kFragTileBytes + (first channel block of this wave) * 2 * kFragBytes + lane * 16
};
// (CBS: 32-channel blocks of the slot - 8 for the 256-wide slots, 2 for the encoding's)
template <int CBS = 8>
__device__ __forceinline__ unsigned frag_off(int kb, int cb, int plane) { return (unsigned)((kb * CBS + cb) * 2 + plane) * kFragBytes; }

// NB: 32-channel blocks of this wave (consecutive, starting at the column `xa` points at).
template <int NB, int ROW, int PLANE, int CBS = 8>
__device__ __forceinline__ void planes_to_frag(const _Float16* xa /* plane_hi + (lane & 31) * ROW + 8 * (lane >> 5) + first column */,
                                               const Selector& sel, const FragDst& dst) {
#ifdef INERF_ABL_NO_FRAG        // (timing ablation of a development build, scripts/build_variant.sh: results are wrong)
    return;
#endif
    const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        // both point halves of a channel block together: eight operand reads, then eight MFMAs on four independent accumulators,
        // then the conversions and stores - one block at a time was a chain of exposed latencies (LDS, matrix core, LDS, ...)
        f16x8 ah[2][2], al[2][2];          // [pb][k-block]
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const _Float16* src = xa + pb * 32 * ROW + 32 * cb;
            ah[pb][0] = *reinterpret_cast<const f16x8*>(src);          ah[pb][1] = *reinterpret_cast<const f16x8*>(src + 16);
            al[pb][0] = *reinterpret_cast<const f16x8*>(src + PLANE);  al[pb][1] = *reinterpret_cast<const f16x8*>(src + PLANE + 16);
        }
        f32x16 th[2], tl[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            th[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[pb][0], sel.k[0], zero, 0, 0, 0);
            tl[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[pb][0], sel.k[0], zero, 0, 0, 0);
        }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            th[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[pb][1], sel.k[1], th[pb], 0, 0, 0);
            tl[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[pb][1], sel.k[1], tl[pb], 0, 0, 0);
        }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x4 oh, ol;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    oh[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(th[pb][8 * q + 2 * i], th[pb][8 * q + 2 * i + 1]));
                    ol[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(tl[pb][8 * q + 2 * i], tl[pb][8 * q + 2 * i + 1]));
                }
#ifdef INERF_ABL_NO_FRAG_STORE  // (timing ablation: the transposition without its stores; the asm keeps the values alive)
                asm volatile("" :: "v"(oh), "v"(ol));
#else
                // (whole offset in the VGPR operand: a 16-byte buffer store with a register SGPR offset gets no hazard wait state, tests/test_isa_audit_cpu.py)
                __builtin_amdgcn_raw_buffer_store_b128(oh, dst.rsrc, (int)(dst.voff + frag_off<CBS>(2 * pb + q, cb, 0)), 0, INERF_FRAG_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(ol, dst.rsrc, (int)(dst.voff + frag_off<CBS>(2 * pb + q, cb, 1)), 0, INERF_FRAG_AUX);
#endif
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One point half at a time (48 registers instead of 96), for the places where a layer's accumulators are still live: the two-workgroup
// chain writes every dZ slot out of the planes right behind the loop of the GEMM that consumes it (its input is intact until that
// GEMM's closing barrier).  (Tried for the ordering - a wave's later loads are counted behind its earlier stores in vmcnt, so stores
// issued in front of a GEMM could stall its first k-blocks: no measurable difference, forward or chain, profiles/r05_frag_store_order.txt;
// what the fragment stores cost - forward 1.79 -> 1.48 ms without them - is their share of the CU's vector-memory path, which the
// weight stream from L2 already loads to ~2/3.  Kept in the chain: one way out for all slots, no transposition in the epilogues.)
template <int NB, int ROW, int PLANE, int CBS = 8>
__device__ __forceinline__ void planes_to_frag_late(const _Float16* xa /* plane_hi + (lane & 31) * ROW + 8 * (lane >> 5) + first column */,
                                                    const Selector& sel, const FragDst& dst) {
#ifdef INERF_ABL_NO_FRAG
    return;
#endif
    const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const _Float16* src = xa + pb * 32 * ROW + 32 * cb;
            const f16x8 ah0 = *reinterpret_cast<const f16x8*>(src), ah1 = *reinterpret_cast<const f16x8*>(src + 16);
            const f16x8 al0 = *reinterpret_cast<const f16x8*>(src + PLANE), al1 = *reinterpret_cast<const f16x8*>(src + PLANE + 16);
            f32x16 th = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, sel.k[0], zero, 0, 0, 0);
            f32x16 tl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, sel.k[0], zero, 0, 0, 0);
            th = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, sel.k[1], th, 0, 0, 0);
            tl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, sel.k[1], tl, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x4 oh, ol;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    oh[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(th[8 * q + 2 * i], th[8 * q + 2 * i + 1]));
                    ol[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(tl[8 * q + 2 * i], tl[8 * q + 2 * i + 1]));
                }
#ifdef INERF_ABL_NO_FRAG_STORE
                asm volatile("" :: "v"(oh), "v"(ol));
#else
                __builtin_amdgcn_raw_buffer_store_b128(oh, dst.rsrc, (int)(dst.voff + frag_off<CBS>(2 * pb + q, cb, 0)), 0, INERF_FRAG_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(ol, dst.rsrc, (int)(dst.voff + frag_off<CBS>(2 * pb + q, cb, 1)), 0, INERF_FRAG_AUX);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// The same from REGISTER operands in the accumulator's k order (mlp_f16_heads.h to_operands: the hidden layers of the output
// heads never touch LDS): hi / lo [2 cb + q2][pb] hold, per lane (point lane & 31 of point block pb, half h = lane >> 5),
// channels 32 cb + 16 q2 + 8 (i >> 2) + 4 h + (i & 3) - the selector below picks that order apart.
__device__ __forceinline__ Selector accumulator_selector(int lane) {
    asm volatile("" : "+v"(lane));             // (rebuilt where it is used: as a loop invariant it would occupy eight registers for the whole kernel)
    Selector sel;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int m = 0; m < 8; ++m)
            sel.k[kb][m] = (16 * kb + 8 * (m >> 2) + 4 * (lane >> 5) + (m & 3)) == (lane & 31) ? (_Float16)1.0f : (_Float16)0.0f;
    return sel;
}
template <int NB, int CBS = 8>
__device__ __forceinline__ void operands_to_frag(const f16x8 (&hi)[2 * NB][2], const f16x8 (&lo)[2 * NB][2], const Selector& sel, const FragDst& dst) {
#ifdef INERF_ABL_NO_FRAG
    return;
#endif
    const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {           // one point half at a time: 32 accumulator registers beside the live operands
            f32x16 th = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi[2 * cb][pb], sel.k[0], zero, 0, 0, 0);
            f32x16 tl = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo[2 * cb][pb], sel.k[0], zero, 0, 0, 0);
            th = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi[2 * cb + 1][pb], sel.k[1], th, 0, 0, 0);
            tl = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo[2 * cb + 1][pb], sel.k[1], tl, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x4 oh, ol;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    oh[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(th[8 * q + 2 * i], th[8 * q + 2 * i + 1]));
                    ol[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(tl[8 * q + 2 * i], tl[8 * q + 2 * i + 1]));
                }
                __builtin_amdgcn_raw_buffer_store_b128(oh, dst.rsrc, (int)(dst.voff + frag_off<CBS>(2 * pb + q, cb, 0)), 0, INERF_FRAG_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(ol, dst.rsrc, (int)(dst.voff + frag_off<CBS>(2 * pb + q, cb, 1)), 0, INERF_FRAG_AUX);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// word = (word << 1) | (v > 0) in two instructions.  v > 0 <=> its bit pattern, read as a signed integer, is > 0 (-0.0 is
// INT_MIN); the median of (pattern, 0, 1) is that bit.  (From C the compiler builds compare + select + or.)
__device__ __forceinline__ unsigned push_positive_bit(unsigned word, float v) {
    int b;
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(b) : "v"(v));
    asm("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(word) : "v"(b));
    return word;
}

// ReLU masks of a trunk layer (layout.h relu_bits_offset): where this lane's two words of the layer go
struct BitsDst {
    __amdgpu_buffer_rsrc_t rsrc;      // the whole mask area
    int off;                          // bytes: (((tile * kReluBitLayers + layer) * 4 + wave) * 64 + lane) * 8
};

template <int RB, int ROW = kRowH, int PLANE = kPlaneH, bool SAVE = false, bool BITS = false, int PB = 2, bool MAX3 = true>
__device__ __forceinline__ void wide_store_h(const f32x16 (&am)[RB][PB], float inv, const f32x4 (&bias)[RB][4],
                                             _Float16* dl,   // plane_hi + (lane&31)*kRowH + 4*(lane>>5) + dcol + chan0
                                             bool relu, f16x2& amax2, float* gout /* or nullptr: + chan0 + 4*(lane>>5) */,
                                             int gstride, int valid0, int valid1, const SaveDst* sv = nullptr,
                                             const BitsDst* bd = nullptr) {
#ifdef INERF_ABL_NO_EPILOGUE     // (timing ablation of a development build: no bias / ReLU / split / LDS stores - compare CYCLES, the planes keep the encoding)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
#ifdef __HIP_DEVICE_COMPILE__
            asm volatile("" :: "v"(am[rb][pb]));      // (the GEMM stays alive)
#endif
        }
    return;
#endif
    // mask words (layout.h relu_bits_offset): per 64-point tile, layer, 64-channel group and lane TWO 32-bit words, one per 32-channel
    // row block, bits in (point block, register) order.  The 64-channel x 64-point wave tile (RB = 2, PB = 2) writes both words of its
    // tile; the 32-channel x 128-point wave tile (RB = 1, PB = 4: mlp_f16_t128.hip) writes ONE word (bd->off names it) of each of its two
    // 64-point tiles - the same bits at the same addresses.
    static_assert(!BITS || (RB == 2 && PB == 2) || (RB == 1 && PB == 4), "mask words are defined for the 64 x 64 and the 32 x 128 wave tile");
    static_assert(PB == 2 || !SAVE, "fp32 row copies are defined for the 64-point tile");
    constexpr int kWords = BITS ? RB * PB / 2 : 1;
    unsigned mask[kWords];
#pragma unroll
    for (int w = 0; w < kWords; ++w) mask[w] = 0u;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            unsigned& mword = mask[!BITS ? 0 : RB == 2 ? rb : pb >> 1];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float t[4];
#if INERF_EPI_PKFMA     // two values per v_pk_fma_f32 (the same IEEE fma per element): the epilogue is instruction issue, not arithmetic
                {
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    const f32x2_ i2 = {inv, inv};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2_ a2 = {am[rb][pb][4 * g + 2 * h], am[rb][pb][4 * g + 2 * h + 1]};
                        const f32x2_ b2 = {bias[rb][g][2 * h], bias[rb][g][2 * h + 1]};
                        const f32x2_ r2 = __builtin_elementwise_fma(a2, i2, b2);
                        t[2 * h] = r2[0];
                        t[2 * h + 1] = r2[1];
                    }
                }
#endif
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#if !INERF_EPI_PKFMA
                    t[i] = __builtin_fmaf(am[rb][pb][4 * g + i], inv, bias[rb][g][i]);
#endif
                    if (relu) t[i] = fmaxf(t[i], 0.0f);
                    if constexpr (BITS) mword = push_positive_bit(mword, t[i]);
                    if (gout && (pb == 0 ? valid0 : valid1))
                        gout[(size_t)pb * 32 * gstride + 32 * rb + 8 * g + i] = t[i] * (1.0f / kActScale);
                }
#ifndef INERF_ABL_NO_ROWS
                if constexpr (SAVE) {
                    const f32x4 v = f32x4{t[0], t[1], t[2], t[3]} * (1.0f / kActScale);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sv->rsrc,
                                                           sv->voff + (pb * 32 * sv->stride + 32 * rb + 8 * g) * 4, 0, 0);
                }
#endif
                f16x2 h01, h23, l01, l23;
                split_pair(t[0], t[1], h01, l01);
                split_pair(t[2], t[3], h23, l23);
                f16x2 a01 = h01, a23 = h23;
                if (!relu) {     // ReLU outputs are non-negative already
                    a01 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h01) & 0x7FFF7FFFu);
                    a23 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h23) & 0x7FFF7FFFu);
                }
                if constexpr (MAX3) pk_max3_into(amax2, a01, a23);
                else amax2 = __builtin_elementwise_max(amax2, __builtin_elementwise_max(a01, a23));      // (training forwards: the three-input form costs them 20 bytes of scratch)
                const f16x4 hi4 = {h01[0], h01[1], h23[0], h23[1]}, lo4 = {l01[0], l01[1], l23[0], l23[1]};
                _Float16* d = dl + pb * 32 * ROW + 32 * rb + 8 * g;
                *reinterpret_cast<f16x4*>(d) = hi4;
                *reinterpret_cast<f16x4*>(d + PLANE) = lo4;
            }
            {   // pinned per block: left free, the running maximum becomes a tree whose partial maxima stay alive (and were spilled in the
                // training forward) until a much later fold
                unsigned a = __builtin_bit_cast(unsigned, amax2);
                asm volatile("" : "+v"(a));
                amax2 = __builtin_bit_cast(f16x2, a);
            }
            // keep the scheduler from converting all 8 blocks at once (it would need >256 live VGPRs and spill)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (BITS) {
        if constexpr (RB == 2) {
            const u32x2 w = {mask[0], mask[1]};
            __builtin_amdgcn_raw_buffer_store_b64(w, bd->rsrc, bd->off, 0, 0);
        } else {        // this wave's row-block word of the tile's two 64-point halves
            __builtin_amdgcn_raw_buffer_store_b32(mask[0], bd->rsrc, bd->off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(mask[1], bd->rsrc, bd->off + kReluBitTileBytes, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM: 16 output rows x this wave's 16 points on v_mfma_f32_16x16x32_f16, K = 32*KB32
// ------------------------------------------------------------------------------------------------
template <int KB32, int PLANE = kPlaneH>
__device__ __forceinline__ f32x4 skinny_gemm_h(const WeightBuf& wb, int frag_bytes, int bias_bytes, int scale_bytes,
                                               const _Float16* xs /* plane_hi + (16*wave + (lane&15))*kRowH + col + 8*(lane>>4) */,
                                               int lane) {
    const f32x4 bias = wb.vec4(bias_bytes, 16 * (lane >> 4));
    const float inv = wb.scalar(scale_bytes);
    f32x4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kb = 0; kb < KB32; ++kb) {
        const f16x8 wh = wb.frag(frag_bytes + (2 * kb) * 1024), wl = wb.frag(frag_bytes + (2 * kb + 1) * 1024);
        const f16x8 xh = *reinterpret_cast<const f16x8*>(xs + 32 * kb);
        const f16x8 xl = *reinterpret_cast<const f16x8*>(xs + PLANE + 32 * kb);
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, a1, 0, 0, 0);
    }
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(a0[i] + a1[i], inv, bias[i]);
    return r;
}

__device__ __forceinline__ float sigmoid_ref_h(float x) { return __fdiv_rn(1.0f, 1.0f + expf(-x)); }


template <int RB, int KSTRIDE = RB * 2048, int RBSTRIDE = 2048>
__device__ __forceinline__ void prefetch_w(WidePreH<RB>& pre, const WeightBuf& wb, int frag_bytes) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int part = 0; part < 2; ++part) pre.w[kb][rb][part] = wb.frag(frag_bytes + kb * KSTRIDE + rb * RBSTRIDE + part * 1024);
}

template <int RB>
__device__ __forceinline__ void load_bias(f32x4 (&bias)[RB][4], float& inv, const WeightBuf& wb, int bias_bytes, int scale_bytes,
                                          int lane) {
    const int h16 = 16 * (lane >> 5);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[rb][g] = wb.vec4(bias_bytes + (32 * rb + 8 * g) * 4, h16);
    inv = wb.scalar(scale_bytes);
}

}  // namespace inerf
