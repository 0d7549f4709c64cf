// mlp_bwd.hip - input-gradient (dgrad) chain of the intrinsic-NeRF MLP, the first half of its backward pass
// (what autograd records for NeRF.forward / Semantic_NeRF.forward when the trainers call loss.backward():
// run_nerf.py:1018 through run_nerf_helpers.py:284-321; trainer.py:990 through semantic_nerf.py:123-181).
//
// Given d loss / d raw[P, CH] and the activations the training forward kept (k_encode_mlp_f16x3<.., kSave>,
// layout.h SaveSlot), one launch walks every tile of 64 sample points backwards through the network and writes the
// pre-activation gradient dZ of EVERY layer (same slot layout as the activations).  The weight gradients are then
// plain GEMMs over the sample points, dW_l = dZ_l^T X_l with K = P (mlp_wgrad.hip; train_api.hip drives both).
//
// Two kernels, same packed weights, same slots, same results (tests/test_train_masks_gpu.py):
//   k_mlp_dgrad_dual   (default)  4 waves x 64 channels x 64 points, ONE buffer updated in place, two workgroups per CU - further down;
//   k_mlp_dgrad<.., 8>            8 waves x 32 channels x 64 points, two buffers (A / B), one workgroup per CU (INERF_DGRAD_KERNEL=single).
//
// Same machinery as the forward kernel (mlp_f16.hip): activations - here gradients - live in LDS as f16 hi/lo planes
// X[point][channel], each layer is D[channel][point] = sum_k W^T[channel][k] X[point][k] on v_mfma_f32_32x32x16_f16
// with the transposed weights streamed from L2 in fragment order (layout.h BwdLayout), three products per fp32 MAC.
// Differences:
//   * gradients have no natural scale, so every point's chain is normalised by a power of two s_p >= its largest head
//     gradient (exact; columns of a GEMM scale independently) and scaled back when dZ is written;
//   * the ReLU masks of the trunk are the bit words the training forward left behind the activation slots (layout.h
//     relu_bits_offset: one 8-byte load per lane and layer instead of 64 floats; h0..h6 in the eight-wave kernel, h0..h7 in the
//     two-workgroup one); the stages that need the activations' VALUES anyway (h7 for the alpha_linear weight gradient, the
//     three head hidden layers) read their fragments;
//   * the heads with 1-4 outputs (sigma, albedo/shading outputs, residual) are outer products: VALU, not MFMA;
//   * the three matrices that feed d h7 (feature_linear^T, as1^T, sem1^T) share a weight scale and one accumulator.
#include <cstdlib>
#include <type_traits>

#include "mlp_f16_dev.h"
#include "mlp_f16_heads.h"

namespace inerf {

struct BwdParams {
    const float* wts;       // packed transposed weights (inerf_pack_weights_bwd)
    const float* raw;       // [P, channels] forward output (for the sigmoid derivatives)
    const float* d_raw;     // [P, channels]
    const float* save;      // activations kept by the training forward
    float* dz;              // out: pre-activation gradients, same slot layout
    float* dz_max;          // out: max |dz| over every slot (caller zeroes it): the weight-gradient kernels' operand scale
    float* head_partial;    // out, optional: [grid][kHeadFloats] weight / bias gradients of the 1-4-row heads, per workgroup
    int32_t* status;
    int64_t off[SAVE_SLOTS];
    int64_t bits_off;       // ReLU masks of h0..h6 inside `save` (layout.h relu_bits_offset)
    BwdLayout L;
    int n_points, n_tiles, channels, n_classes, endpoint;
    int stagger;            // start offset of the workgroups (mlp_common.h stagger_start), 0 = none
};

// Per-workgroup partial gradients of the heads with 1-4 output rows (their operands pass through this kernel's
// registers anyway): the caller sums the workgroups.
//   [0, 384)      residual head weight   [3][128]      (d_res_pre^T vh)
//   [384, 1408)   albedo|shading outputs [4][256]      row j < 3: d_albedo_pre[j]^T as1h, row 3: d_shading_pre^T as1h
//                 (the caller keeps columns 0..127 of rows 0..2 and columns 128..255 of row 3)
//   [1408, 1664)  alpha_linear weight    [256]         (d_sigma^T h7)
//   [1664, 1672)  biases: albedo 3, shading 1, residual 3, sigma 1  (sums of the head gradients)
constexpr int kHeadRes = 0, kHeadAs2 = 384, kHeadAlpha = 1408, kHeadBias = 1664, kHeadFloats = 1672;

// epilogue of one transposed layer: t = acc * inv (+ per-channel vector x per-point scalar), ReLU mask from the saved
// activation, hi/lo split into LDS (normalised, kActScale domain) and the true gradient to global memory
struct NoAlpha {};

// Where a layer's dZ goes: a FRAGMENT slot (layout.h SaveSlot) - the operand fragments of the weight-gradient product
// dW = dZ^T X, lane = channel, 8 k-values = 8 sample points.  The accumulator layout (lane = point, 16 registers = 4 + 4 + 4 + 4
// channels) is the transpose of that, and it stores badly anyway (a 16-byte piece per lane is 32 bytes per point and instruction:
// measured 1.2-1.5 x write amplification, the 512 partial-line transactions per wave and layer were most of an epilogue's
// 10 000 cycles).  So the block is transposed by the matrix core: with the block's f16 hi / lo halves - which the epilogue has
// anyway - as the A operand (row = point, k = the lane's channels) and a 0/1 selection matrix as B, D'[point][channel] comes back
// with lane = CHANNEL, registers = points - hi and lo separately, and as they are: the NORMALISED values (kActScale * dz / s_p),
// whose range is this kernel's guard, so that they fit f16 with their full 22 bits whatever the point's gradient scale.  The
// products are f16 x 1.0: exact, and any rounding mode converts them back.  They leave as four 1 KB fragments per block - one
// 16-byte store per lane and fragment, every 128-byte line complete.  The normalisers s_p (4 bytes per point, SAVE_ENC of
// the gradient buffer) travel beside them: the weight-gradient kernels, which know the batch's max |dz| by then, multiply them
// back in when they bring a fragment to their own scale.  4 MFMAs per 32 x 32 block, +8 % of a layer's matrix work.
// (Rounds 2-3: the transposed block left as fp32 rows, scaled back per value from LDS; a first fragment version split the true
// dz in the producer with a scale guessed from d_raw and lost 6 bits in the lower trunk layers - f16's 30 binades do not hold
// 13 binades between the points of a batch times ~16 between the layers of the chain.)
struct DzDst {
    __amdgpu_buffer_rsrc_t rsrc;      // the layer's dZ slot: n_tiles * kFragTileBytes (whole tiles: padding points carry zeros)
    unsigned voff;                    // bytes: tile * kFragTileBytes + (first channel block of this wave) * 2 * kFragBytes + lane * 16
};
// (the B operand of the transposing MFMAs: mlp_f16_dev.h accumulator_selector - the k order of the ACCUMULATOR registers;
// planes_to_frag's operands come from LDS in channel order)
// FRAG = false: planes only - the caller writes the fragments out later, from the planes (planes_to_frag_late: behind the GEMM that
// consumes the layer, so that the stores do not sit in front of that GEMM's weight stream).
template <int RB, bool BITS = false, typename AlphaAcc = NoAlpha, int ROW = kRowH, int PLANE = kPlaneH, bool FRAG = true>
__device__ __forceinline__ void bwd_store(const f32x16 (&am)[RB][2], float inv,
                                          const f32x4 (*acts)[2][4] /* [RB][2][4] saved activations of this lane's values (requested
                                                                       before the GEMM; zero for points beyond the end), or nullptr */,
                                          const f32x4 (*extra)[4] /* [RB][4] or nullptr */, float ex0, float ex1,
                                          _Float16* dl /* plane_hi + (lane&31)*kRowH + 4h + dcol + chan0 */, f16x2& amax2,
                                          const DzDst& dst, int lane, float s0, float s1,
                                          bool valid0, bool valid1, float& gmax,
                                          u32x2 mask_bits /* BITS: this lane's words of the layer (layout.h) */,
                                          AlphaAcc& alpha_acc /* f32x4[RB][4]: += (true d sigma of the point) * saved activation; by
                                                                 reference and selected at compile time - through a pointer-or-null
                                                                 argument the accumulators lived in scratch memory */) {
    constexpr bool kAlpha = !std::is_same<AlphaAcc, NoAlpha>::value;
    const Selector sel = accumulator_selector(lane);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const bool valid = pb == 0 ? valid0 : valid1;
            const int bits16 = valid ? (int)(mask_bits[rb] >> (16 * (1 - pb))) : 0;     // points beyond the end carry no gradient
            const float ex = pb == 0 ? ex0 : ex1;
            // |dz| bound of the block: all sixteen values belong to ONE point, so the running maximum is taken on the packed f16 halves
            // the range guard needs anyway and scaled back once per block (hi is t rounded toward zero: 1 + 2^-9 covers the cut) -
            // a compare-and-scale per value was 20 of the block's ~150 VALU instructions
            const float back = (pb == 0 ? s0 : s1) * ((1.0f + 0x1p-9f) / kActScale);
            f16x2 bmax = {(_Float16)0.0f, (_Float16)0.0f};
            f16x4 hi_g[4], lo_g[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float t[4];
                f32x4 m4 = {1.0f, 1.0f, 1.0f, 1.0f};
                if (acts) m4 = acts[rb][pb][g];
                if constexpr (kAlpha) alpha_acc[rb][g] += m4 * (ex * (pb == 0 ? s0 : s1));      // m4 = h7 itself (zero for invalid points)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    t[i] = am[rb][pb][4 * g + i] * inv;
                    if (extra) t[i] = __builtin_fmaf(extra[rb][g][i], ex, t[i]);
                    if constexpr (BITS)    // v_bfe_i32: the bit, sign-extended, is the AND mask
                        t[i] = __builtin_bit_cast(float, __builtin_bit_cast(int, t[i]) & __builtin_amdgcn_sbfe(bits16, 15 - (4 * g + i), 1));
                    else
                        t[i] = m4[i] > 0.0f ? t[i] : 0.0f;
                }
                f16x2 h01, h23, l01, l23;
                split_pair(t[0], t[1], h01, l01);
                split_pair(t[2], t[3], h23, l23);
                const f16x2 a01 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h01) & 0x7FFF7FFFu);
                const f16x2 a23 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h23) & 0x7FFF7FFFu);
                bmax = __builtin_elementwise_max(bmax, __builtin_elementwise_max(a01, a23));
                hi_g[g] = f16x4{h01[0], h01[1], h23[0], h23[1]};
                lo_g[g] = f16x4{l01[0], l01[1], l23[0], l23[1]};
                _Float16* d = dl + pb * 32 * ROW + 32 * rb + 8 * g;
                *reinterpret_cast<f16x4*>(d) = hi_g[g];
                *reinterpret_cast<f16x4*>(d + PLANE) = lo_g[g];
            }
            amax2 = __builtin_elementwise_max(amax2, bmax);
            // invalid points carry zeros: no need to exclude them from the running maximum
            gmax = fmaxf(gmax, fmaxf((float)bmax[0], (float)bmax[1]) * back);
#ifndef INERF_ABL_NO_FRAG
            if constexpr (FRAG) {
            // transpose (see DzDst): D'[point][channel] = sum_k A[point][k] Sel[k][channel], two k-blocks of 16 channels, hi and lo each
            const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            f32x16 trh = zero, trl = zero;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const f16x8 ah = {hi_g[2 * kb][0], hi_g[2 * kb][1], hi_g[2 * kb][2], hi_g[2 * kb][3],
                                  hi_g[2 * kb + 1][0], hi_g[2 * kb + 1][1], hi_g[2 * kb + 1][2], hi_g[2 * kb + 1][3]};
                const f16x8 al = {lo_g[2 * kb][0], lo_g[2 * kb][1], lo_g[2 * kb][2], lo_g[2 * kb][3],
                                  lo_g[2 * kb + 1][0], lo_g[2 * kb + 1][1], lo_g[2 * kb + 1][2], lo_g[2 * kb + 1][3]};
                trh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, sel.k[kb], trh, 0, 0, 0);
                trl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, sel.k[kb], trl, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x4 oh, ol;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    oh[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(trh[8 * q + 2 * i], trh[8 * q + 2 * i + 1]));
                    ol[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(trl[8 * q + 2 * i], trl[8 * q + 2 * i + 1]));
                }
#ifdef INERF_ABL_NO_FRAG_STORE
                asm volatile("" :: "v"(oh), "v"(ol));
#else
                // (whole offset in the VGPR operand: a 16-byte buffer store with a register SGPR offset gets no hazard wait state, tests/test_isa_audit_cpu.py)
                __builtin_amdgcn_raw_buffer_store_b128(oh, dst.rsrc, (int)(dst.voff + frag_off(2 * pb + q, rb, 0)), 0, INERF_FRAG_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(ol, dst.rsrc, (int)(dst.voff + frag_off(2 * pb + q, rb, 1)), 0, INERF_FRAG_AUX);
#endif
            }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// four consecutive channels of one point -> hi/lo planes (normalised value v, stored as kActScale * v)
__device__ __forceinline__ void split_store4(_Float16* hi_ptr, _Float16* lo_ptr, const float (&v)[4], f16x2& amax2) {
    f16x2 h01, h23, l01, l23;
    split_pair(v[0] * kActScale, v[1] * kActScale, h01, l01);
    split_pair(v[2] * kActScale, v[3] * kActScale, h23, l23);
    const f16x4 hi4 = {h01[0], h01[1], h23[0], h23[1]}, lo4 = {l01[0], l01[1], l23[0], l23[1]};
    const f16x2 a01 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h01) & 0x7FFF7FFFu);
    const f16x2 a23 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h23) & 0x7FFF7FFFu);
    amax2 = __builtin_elementwise_max(amax2, __builtin_elementwise_max(a01, a23));
    *reinterpret_cast<f16x4*>(hi_ptr) = hi4;
    *reinterpret_cast<f16x4*>(lo_ptr) = lo4;
}

// Pre-activation gradients of one point's output heads (albedo 3, shading 1, residual 3, sigma 1) from d loss / d raw and raw,
// and the point's normaliser: the power of two above its largest head gradient (1 for a point without gradient).
__device__ __forceinline__ float head_gradients(const BwdParams& p, int gp, bool sem, float (&dp)[8]) {
    const int ch = p.channels;
    const float* __restrict__ r = p.raw + (size_t)gp * ch;
    const float* __restrict__ g = p.d_raw + (size_t)gp * ch;
    const float sh = r[7];
    float dsh = g[7];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = r[4 + k], rs = r[8 + k];
        dp[k] = (g[k] * sh + g[4 + k]) * (a * (1.0f - a));            // rgb = albedo * shading + residual, sigmoid'
        dsh += g[k] * a;
        dp[4 + k] = (g[k] + g[8 + k]) * (rs * (1.0f - rs));
    }
    dp[3] = dsh * (sh * (1.0f - sh));
    dp[7] = g[3];                                                      // sigma has no activation inside the network
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(dp[k]));
    if (sem) for (int j = 0; j < p.n_classes; ++j) m = fmaxf(m, fabsf(g[INERF_BASE_CHANNELS + j]));
    if (p.endpoint) for (int c = 0; c < INERF_ENDPOINT_DIM; ++c) m = fmaxf(m, fabsf(g[ch - INERF_ENDPOINT_DIM + c]));
    float s = 1.0f;
    if (m > 0.0f && m < 3.0e38f) { int e; frexpf(m, &e); s = ldexpf(1.0f, e); }
    return s;
}

// The same in two halves (load, compute), for the two-workgroup chain.
struct HeadInputs { float r[11], g[11]; };
__device__ __forceinline__ void head_inputs(const BwdParams& p, int gp, HeadInputs& in) {
    const float* __restrict__ r = p.raw + (size_t)gp * p.channels;
    const float* __restrict__ g = p.d_raw + (size_t)gp * p.channels;
#pragma unroll
    for (int k = 0; k < 11; ++k) { in.r[k] = r[k]; in.g[k] = g[k]; }
}
__device__ __forceinline__ float head_gradients(const BwdParams& p, int gp, bool sem, const HeadInputs& in, float (&dp)[8]) {
    const float sh = in.r[7];
    float dsh = in.g[7];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = in.r[4 + k], rs = in.r[8 + k];
        dp[k] = (in.g[k] * sh + in.g[4 + k]) * (a * (1.0f - a));      // rgb = albedo * shading + residual, sigmoid'
        dsh += in.g[k] * a;
        dp[4 + k] = (in.g[k] + in.g[8 + k]) * (rs * (1.0f - rs));
    }
    dp[3] = dsh * (sh * (1.0f - sh));
    dp[7] = in.g[3];                                                   // sigma has no activation inside the network
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(dp[k]));
    const int ch = p.channels;
    const float* __restrict__ g = p.d_raw + (size_t)gp * ch;
    if (sem) for (int j = 0; j < p.n_classes; ++j) m = fmaxf(m, fabsf(g[INERF_BASE_CHANNELS + j]));
    if (p.endpoint) for (int c = 0; c < INERF_ENDPOINT_DIM; ++c) m = fmaxf(m, fabsf(g[ch - INERF_ENDPOINT_DIM + c]));
    float s = 1.0f;
    if (m > 0.0f && m < 3.0e38f) { int e; frexpf(m, &e); s = ldexpf(1.0f, e); }
    return s;
}

// NW waves per workgroup: 4 (each wave 64 channels = RB 2 row blocks; one wave per SIMD) or 8 (32 channels each; TWO waves per
// SIMD, so one wave's epilogue / VALU stage / memory wait runs under the other's MFMAs - with one wave per SIMD a tile was
// 63 k cycles of MFMA in 192 k).  Same tile, same LDS, same packed weights (an 8-wave wave takes one of the two row blocks of
// the 4-wave packing), same results.
template <bool kSsr, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k_mlp_dgrad(const BwdParams p) {
    constexpr int kPts = kTilePoints;
    constexpr int RB = 8 / NW;                 // 32-channel row blocks per wave
    constexpr int NT = 64 * NW;                // threads
    constexpr int WCH = 32 * RB;               // channels per wave
    constexpr int KS = 4096;                   // bytes between k-blocks of the 4-wave packing
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsb[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const BwdLayout& L = p.L;
    f16x2 amax2 = {(_Float16)0.0f, (_Float16)0.0f};
    float gmax = 0.0f;

    _Float16* const xw = ldsb + (lane & 31) * kRowH;
    const _Float16* const xr = xw + 8 * (lane >> 5);              // wide GEMM operand reads (+ column)
    _Float16* const xd = xw + 4 * (lane >> 5) + WCH * wave;       // wide stores: this wave's channels (+ column)
    auto ptf = [&](int pt) { return reinterpret_cast<float*>(ldsb + pt * kRowH); };   // per-point scratch in the enc columns:
                                                                                      // [0..7] head gradients / s, [8] s, [9] 1/s
    WeightBuf wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wts), 0, L.total_floats * 4, 0x00020000);
    wb.voff = lane * 16;
    const int wave4 = NW == 4 ? wave : wave >> 1, rbsel = NW == 4 ? 0 : wave & 1;      // position in the 4-wave packing
    auto frag = [&](const GemmSlot& s, int kbt) { return (s.w + wave4 * kbt * 2 * 2 * 256) * 4 + rbsel * 2048; };
    const bool sem = kSsr && L.has_sem;
    const int ch = p.channels;

    WidePreH<RB> preA, preB;
    prefetch_w<RB, KS>(preA, wb, frag(L.views_t, 8));
    // (Fragment and row stores carry their WHOLE offset in the VGPR operand, the SGPR offset stays the constant 0: a 16-byte buffer
    // store whose SGPR offset is a REGISTER gets no wait state before its data registers may be overwritten - the compiler's
    // hazard table says none is needed in that form - and on this chip the store then sometimes sends what the NEXT instruction
    // wrote: found in round 2 in 4 % of the launches of the row-wise stages this kernel had then; tests/test_isa_audit_cpu.py.)
    const __amdgpu_buffer_rsrc_t bits_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.save) + p.bits_off, 0, (int)((unsigned)p.n_tiles * (unsigned)kReluBitTileBytes), 0x00020000);

    // head weight gradients, accumulated over this workgroup's tiles (see kHead*)
    const bool heads = p.head_partial != nullptr;
    float hb[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};     // threads 0..63: sums of the head gradients
    // residual head / albedo|shading outputs: ONE channel per lane (the VALU stages work on the layers' fragments: lane = channel)
    float hres[3], has2[4];
    f32x4 halpha[RB][4];                                                   // alpha: [rb][g], four channels each
#pragma unroll
    for (int j = 0; j < 3; ++j) hres[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) has2[j] = 0.0f;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) halpha[rb][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

#ifdef INERF_DGRAD_STAMPS   // development build (scripts/build_variant.sh): dz_max points at 2 + 64 x uint64; cycle stamps of
    // workgroup 0 / thread 0 at the phase boundaries of its SECOND tile (steady state)
    unsigned long long* const dbg = (p.dz_max && blockIdx.x == 0 && tid == 0) ? reinterpret_cast<unsigned long long*>(p.dz_max) + 1 : nullptr;
    int dbg_n = 0;
#define STAMP() do { if (dbg && tile == (int)gridDim.x && dbg_n < 62) { dbg[1 + dbg_n] = __builtin_readcyclecounter(); ++dbg_n; dbg[0] = dbg_n; } } while (0)
#else
#define STAMP() do { } while (0)
#endif
    stagger_start(p.stagger);
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        STAMP();
        // The saved activations the two fragment-native VALU stages need are requested a stage ahead (a load issued inside the
        // stage that uses it is thousands of exposed cycles).  Both layers were kept as FRAGMENTS (lane = channel, 8 points per
        // 16-byte slot): the views hidden layer (128 channels: wave w takes channel block w & 3, k-blocks 2 (w >> 2) and + 1), the
        // albedo | shading hidden layer (256: channel block w, all four k-blocks).
        static_assert(NW == 8, "the fragment-native stages are written for eight waves");
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const int vh_cb = wave & 3, vh_kb0 = 2 * (wave >> 2);
        f16x8 act_vh[2][2];                  // [k-block][hi | lo]
        {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.save) + p.off[SAVE_VH], 0,
                                                                                (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 2)), 0x00020000);
            const unsigned voff = (unsigned)tile * (unsigned)(kFragTileBytes / 2) + (unsigned)lane_s * 16u;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int plane = 0; plane < 2; ++plane)
                    act_vh[k][plane] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(voff + frag_off<4>(vh_kb0 + k, vh_cb, plane)), 0, 0));
        }
        // ---------------- heads: pre-activation gradients of the output heads, per-point scale ----------------
        if (tid < kPts) {
            const int gp = tile * kPts + tid;
            float dp[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            float s = 1.0f;
            if (gp < p.n_points) {
                s = head_gradients(p, gp, sem, dp);
                float* __restrict__ o = p.dz + p.off[SAVE_DPRE] + (size_t)gp * 8;
                *reinterpret_cast<f32x4*>(o) = f32x4{dp[0], dp[1], dp[2], dp[3]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{dp[4], dp[5], dp[6], dp[7]};
#pragma unroll
                for (int k = 0; k < 8; ++k) hb[k] += dp[k];
            }
            float* f = ptf(tid);
            const float is = 1.0f / s;
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = dp[k] * is;
            f[8] = s;
            f[9] = is;
            p.dz[p.off[SAVE_ENC] + gp] = s;        // the point's normaliser, beside the fragments (padding points: 1, with all-zero fragments)
        }
        STAMP();
        __syncthreads();
        STAMP();

        // ---------------- dZ of the view-dependent layer: relu'(vh) * (W_res^T d_res [+ d endpoint feature]) -> A ----------------
        // lane = channel 32 (w & 3) + (lane & 31); per k-block the lane's eight points (layout.h frag_point).  The result IS the
        // layer's dZ fragment (stored as it is: no transposition) and goes into the planes for the views^T GEMM.
        {
            const int cch = 32 * vh_cb + (lane_s & 31), lh = lane_s >> 5;
            const f32x4 w4 = wb.vec4(L.res_w * 4, 16 * cch);                    // (W_res[0][c], W_res[1][c], W_res[2][c], 0)
            const __amdgpu_buffer_rsrc_t dz_vh = __builtin_amdgcn_make_buffer_rsrc(p.dz + p.off[SAVE_VH], 0,
                                                                                   (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 2)), 0x00020000);
            _Float16* const col = ldsb + kColA + cch;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int kb = vh_kb0 + k;
                f16x8 oh, ol;
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    float t[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int pt = 32 * (kb >> 1) + frag_point(kb & 1, lh, i + e);
                        const float* f = ptf(pt);
                        const f32x4 d = *reinterpret_cast<const f32x4*>(f + 4);          // normalised d_res (3), d_sigma
                        const float sp = f[8];
                        const float act = ((float)act_vh[k][0][i + e] + (float)act_vh[k][1][i + e]) * (1.0f / kActScale);
                        float v = w4[0] * d[0] + w4[1] * d[1] + w4[2] * d[2];
                        if (kSsr && p.endpoint) {       // raw[..., -128:] is this layer's output itself (semantic_nerf.py:163-164)
                            const int gp = tile * kPts + pt;
                            if (gp < p.n_points) v = __builtin_fmaf(p.d_raw[(size_t)gp * ch + ch - INERF_ENDPOINT_DIM + cch], f[9], v);
                        }
                        if (heads) {                   // d W_res[j][c] += (true d_res_pre[j] of the point) * vh[c]
                            hres[0] += act * (d[0] * sp);
                            hres[1] += act * (d[1] * sp);
                            hres[2] += act * (d[2] * sp);
                        }
                        v = act > 0.0f ? v : 0.0f;         // (a point beyond the end: its head gradients are zero -> v = 0)
                        gmax = fmaxf(gmax, fabsf(v) * sp);
                        t[e] = v * kActScale;
                    }
                    f16x2 h2, l2;
                    split_pair(t[0], t[1], h2, l2);
                    amax2 = __builtin_elementwise_max(amax2, __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h2) & 0x7FFF7FFFu));
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int pt = 32 * (kb >> 1) + frag_point(kb & 1, lh, i + e);
                        col[pt * kRowH] = h2[e];
                        col[pt * kRowH + kPlaneH] = l2[e];
                        oh[i + e] = h2[e]; ol[i + e] = l2[e];
                    }
                }
                const int voff = (int)((unsigned)tile * (unsigned)(kFragTileBytes / 2) + (unsigned)lane_s * 16u + frag_off<4>(kb, vh_cb, 0));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, oh), dz_vh, voff, 0, INERF_FRAG_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ol), dz_vh, voff + kFragBytes, 0, INERF_FRAG_AUX);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP();
        __syncthreads();
        STAMP();

        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        const int pt0 = tile * kPts + (lane & 31);
        const bool valid0 = pt0 < p.n_points, valid1 = pt0 + 32 < p.n_points;
        const float s0 = ptf(lane & 31)[8], s1 = ptf((lane & 31) + 32)[8];
        auto dz_dst = [&](int slot) {                 // 256-wide slots only (every layer this kernel runs on the matrix core)
            DzDst d;
            d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.dz + p.off[slot], 0, (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes), 0x00020000);
            d.voff = (unsigned)tile * (unsigned)kFragTileBytes + (unsigned)(WCH / 32 * wave) * (2u * kFragBytes) + (unsigned)lane * 16u;
            return d;
        };
        f32x16 am[RB][2];
        f16x8 act_as1[4][2];                 // [k-block][hi | lo] of channel block `wave`

        // ---------------- d feature = W_views^T[:256] dZ_vh -> B (feature_linear has no activation: this is its dZ) ----------------
        wide_gemm_h<RB, 8, 0, kRowH, kPlaneH, true, KS>(preA, wb, frag(L.views_t, 8), xr, kColA, 0, lane, am);
        {   // the next stage's activations: in flight during this layer's epilogue (requested before the GEMM they sat in front
            // of its weight stream - returns are in order - and cost it 9 k cycles)
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.save) + p.off[SAVE_AS1H], 0,
                                                                                (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes), 0x00020000);
            const unsigned voff = (unsigned)tile * (unsigned)kFragTileBytes + (unsigned)lane_s * 16u;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int plane = 0; plane < 2; ++plane)
                    act_as1[kb][plane] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(voff + frag_off(kb, wave, plane)), 0, 0));
        }
        {
            const float inv = wb.scalar(L.views_t.b * 4);
            prefetch_w<RB, KS>(preA, wb, frag(L.feat_t, 16));
            prefetch_w<RB, KS>(preB, wb, frag(L.as1_t, 16));
            NoAlpha none;
            bwd_store<RB>(am, inv, nullptr, nullptr, 0.0f, 0.0f, xd + kColB, amax2, dz_dst(SAVE_FEAT), lane, s0, s1, valid0, valid1, gmax,
                         u32x2{0u, 0u}, none);
        }
        STAMP();
        __syncthreads();                     // A (dZ_vh) has been read by every wave, B (d feature) is complete
        STAMP();

        // ---------------- dZ of the albedo | shading hidden layer: relu'(as1h) * (W_as2^T [d_albedo, d_shading]) -> A ----------------
        // lane = channel 32 wave + (lane & 31), four k-blocks: like the views stage - the result is the layer's dZ fragment
        {
            const int cch = 32 * wave + (lane_s & 31), lh = lane_s >> 5;
            const f32x4 w4 = wb.vec4(L.as2_w * 4, 16 * cch);                    // albedo_linear2[0..2][c] | shading output [c - 128]
            const __amdgpu_buffer_rsrc_t dz_as = __builtin_amdgcn_make_buffer_rsrc(p.dz + p.off[SAVE_AS1H], 0,
                                                                                   (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes), 0x00020000);
            _Float16* const col = ldsb + kColA + cch;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                f16x8 oh, ol;
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    float t[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int pt = 32 * (kb >> 1) + frag_point(kb & 1, lh, i + e);
                        const float* f = ptf(pt);
                        const f32x4 d = *reinterpret_cast<const f32x4*>(f);              // normalised d_albedo (3), d_shading
                        const float sp = f[8];
                        const float act = ((float)act_as1[kb][0][i + e] + (float)act_as1[kb][1][i + e]) * (1.0f / kActScale);
                        float v = w4[0] * d[0] + w4[1] * d[1] + w4[2] * d[2] + w4[3] * d[3];
                        if (heads) {
                            has2[0] += act * (d[0] * sp);
                            has2[1] += act * (d[1] * sp);
                            has2[2] += act * (d[2] * sp);
                            has2[3] += act * (d[3] * sp);
                        }
                        v = act > 0.0f ? v : 0.0f;
                        gmax = fmaxf(gmax, fabsf(v) * sp);
                        t[e] = v * kActScale;
                    }
                    f16x2 h2, l2;
                    split_pair(t[0], t[1], h2, l2);
                    amax2 = __builtin_elementwise_max(amax2, __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h2) & 0x7FFF7FFFu));
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int pt = 32 * (kb >> 1) + frag_point(kb & 1, lh, i + e);
                        col[pt * kRowH] = h2[e];
                        col[pt * kRowH + kPlaneH] = l2[e];
                        oh[i + e] = h2[e]; ol[i + e] = l2[e];
                    }
                }
                const int voff = (int)((unsigned)tile * (unsigned)kFragTileBytes + (unsigned)lane_s * 16u + frag_off(kb, wave, 0));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, oh), dz_as, voff, 0, INERF_FRAG_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ol), dz_as, voff + kFragBytes, 0, INERF_FRAG_AUX);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP();
        __syncthreads();
        STAMP();

        // ---------------- d h7 = W_feat^T d feature + W_as1^T dZ_as1 (+ W_sem1^T dZ_semh) + w_alpha d sigma ----------------
        // h7 of this lane's values - ReLU mask AND operand of the alpha_linear weight gradient - comes from the forward's FRAGMENTS
        // (lane = channel, 8 points per 16-byte operand slot): requested here, ahead of the two GEMMs, and transposed back into
        // the accumulator layout (lane = point, registers = channels) by the matrix core just before the epilogue that needs it.
        f16x8 h7f[RB][2][2][2];              // [row block][point half][k-block of the half][hi | lo]
        {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.save) + p.off[SAVE_H7], 0,
                                                                                (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes), 0x00020000);
            const unsigned voff = (unsigned)tile * (unsigned)kFragTileBytes + (unsigned)(WCH / 32 * wave) * (2u * kFragBytes) + (unsigned)lane * 16u;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int plane = 0; plane < 2; ++plane)
                            h7f[rb][pb][q][plane] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(voff + frag_off(2 * pb + q, rb, plane)), 0, 0));
        }
        wide_gemm_h<RB, 16, 0, kRowH, kPlaneH, true, KS>(preA, wb, frag(L.feat_t, 16), xr, kColB, 0, lane, am);
        if (sem) prefetch_w<RB, KS>(preA, wb, frag(L.sem1_t, 8));
        else     prefetch_w<RB, KS>(preA, wb, frag(L.trunk_t[7], 16));
        wide_gemm_h<RB, 16, 0, kRowH, kPlaneH, false, KS>(preB, wb, frag(L.as1_t, 16), xr, kColA, 0, lane, am);
        if (sem) {
            __syncthreads();                 // A and B are free
            constexpr int VH_STEP = NT / 32, VH_IT = kPts / VH_STEP;        // rows per pass of this row-wise stage, passes
            const int c4 = (tid & 31) * 4;
#pragma unroll 1
            for (int i = 0; i < VH_IT; ++i) {
                const int pt = (tid >> 5) + VH_STEP * i;
                const int gp = tile * kPts + pt;
                const bool valid = gp < p.n_points;
                const float* f = ptf(pt);
                float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                f32x4 act = {0.0f, 0.0f, 0.0f, 0.0f};
                if (valid) {
                    const float* gl = p.d_raw + (size_t)gp * ch + INERF_BASE_CHANNELS;       // logits: no activation
                    for (int j = 0; j < p.n_classes; ++j) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(p.wts + L.sem2_w + (size_t)j * kHalf + c4);
                        const float gj = gl[j] * f[9];
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) v[cc] = __builtin_fmaf(w[cc], gj, v[cc]);
                    }
                    act = *reinterpret_cast<const f32x4*>(p.save + p.off[SAVE_SEMH] + (size_t)gp * kHalf + c4);
                }
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) v[cc] = act[cc] > 0.0f ? v[cc] : 0.0f;
                split_store4(ldsb + pt * kRowH + kColA + c4, ldsb + pt * kRowH + kColA + c4 + kPlaneH, v, amax2);
                if (valid) {
                    const f32x4 o = f32x4{v[0], v[1], v[2], v[3]} * f[8];
                    gmax = fmaxf(gmax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                }
            }
            __syncthreads();
            if (wave < 4) {                  // dZ of the semantic hidden layer: fragments of a 128-channel slot, like dZ_vh
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.dz + p.off[SAVE_SEMH], 0, (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 2)), 0x00020000);
                d.voff = (unsigned)tile * (unsigned)(kFragTileBytes / 2) + (unsigned)wave * (2u * kFragBytes) + (unsigned)lane_t * 16u;
                planes_to_frag<1, kRowH, kPlaneH, 4>(xr + kColA + 32 * wave, plane_selector(lane_t), d);
            }
            wide_gemm_h<RB, 8, 0, kRowH, kPlaneH, false, KS>(preA, wb, frag(L.sem1_t, 8), xr, kColA, 0, lane, am);
            prefetch_w<RB, KS>(preA, wb, frag(L.trunk_t[7], 16));
        }
        {
            const float inv = wb.scalar(L.feat_t.b * 4);            // common scale of feat_t / as1_t / sem1_t
            f32x4 aw[RB][4];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    aw[rb][g] = wb.vec4((L.alpha_w + WCH * wave + 32 * rb + 8 * g) * 4, 16 * (lane >> 5)) * kActScale;
            const float e0 = ptf(lane & 31)[7], e1 = ptf((lane & 31) + 32)[7];
            // D[channel][point] = sum_k A[channel][k] Sel[k][point]: a fragment is the A operand as it is (row = channel, k-slot
            // 8 h + i = point frag_point(q, h, i) of the half), the selector of k-block q picks that point's column; hi + lo, both
            // k-blocks into one accumulator = h7 (x 1/kActScale, in the selector) in the layout of `am`
            f32x4 h7v[RB][2][4];
            {
                int ln = lane;
                asm volatile("" : "+v"(ln));              // (the selectors are rebuilt per tile, not kept in 16 registers)
                f16x8 selq[2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        selq[q][i] = frag_point(q, ln >> 5, i) == (ln & 31) ? (_Float16)(1.0f / kActScale) : (_Float16)0.0f;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) {
                        f32x16 hv = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            hv = __builtin_amdgcn_mfma_f32_32x32x16_f16(h7f[rb][pb][q][0], selq[q], hv, 0, 0, 0);
                            hv = __builtin_amdgcn_mfma_f32_32x32x16_f16(h7f[rb][pb][q][1], selq[q], hv, 0, 0, 0);
                        }
#pragma unroll
                        for (int g = 0; g < 4; ++g) h7v[rb][pb][g] = f32x4{hv[4 * g], hv[4 * g + 1], hv[4 * g + 2], hv[4 * g + 3]};
                    }
            }
            STAMP();
            __syncthreads();                 // every wave is done reading A and B
            STAMP();
            bwd_store<RB>(am, inv, h7v, aw, e0, e1, xd + kColA, amax2, dz_dst(SAVE_H7), lane, s0, s1, valid0, valid1, gmax,
                         u32x2{0u, 0u}, halpha);       // (accumulated whether or not `heads`)
        }
        STAMP();
        __syncthreads();
        STAMP();

        // ---------------- trunk, layers 7..1: dZ_{l-1} = relu'(h_{l-1}) * W_l^T dZ_l, ping-pong A <-> B ----------------
#pragma unroll 1
        for (int l = kDepth - 1; l >= 1; --l) {
            const int src = ((kDepth - 1 - l) & 1) ? kColB : kColA;
            const int dst = ((kDepth - 1 - l) & 1) ? kColA : kColB;
            u32x2 mbits;                        // this lane's mask words of the layer (layout.h relu_bits_offset)
            const int mbase = (((tile * kReluBitLayers + (l - 1)) * 4 + wave4) * 64) * 8;
            if constexpr (RB == 2) mbits = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(bits_rsrc, lane * 8, mbase, 0));
            else mbits = u32x2{__builtin_amdgcn_raw_buffer_load_b32(bits_rsrc, lane * 8, mbase + 4 * rbsel, 0), 0u};
            wide_gemm_h<RB, 16, 0, kRowH, kPlaneH, true, KS>(preA, wb, frag(L.trunk_t[l], 16), xr, src, 0, lane, am);
            const float inv = wb.scalar(L.trunk_t[l].b * 4);
            if (l > 1) prefetch_w<RB, KS>(preA, wb, frag(L.trunk_t[l - 1], 16));
            else       prefetch_w<RB, KS>(preA, wb, frag(L.views_t, 8));
            NoAlpha none;
            bwd_store<RB, true>(am, inv, nullptr, nullptr, 0.0f, 0.0f, xd + dst, amax2, dz_dst(SAVE_H0 + l - 1), lane, s0, s1, valid0, valid1,
                               gmax, mbits, none);
            STAMP();
            __syncthreads();
            STAMP();
        }
    }
    if (heads) {                          // reduce over the threads / lanes that shared a channel group, through LDS
        __syncthreads();
        constexpr int AST = 16 * RB + 1;                        // padded stride of a lane's alpha accumulators
        // residual head: lane = channel 32 (w & 3) + (lane & 31), partial over (lane >> 5, w >> 2); albedo|shading outputs: lane =
        // channel 32 w + (lane & 31), partial over lane >> 5.  The lane halves meet by a shuffle, the two wave groups through LDS.
        float* red = reinterpret_cast<float*>(ldsb);            // [2][3][128]: hres of the wave groups; behind it the alpha / bias areas
        float hres2[3], has22[4];
#pragma unroll
        for (int j = 0; j < 3; ++j) hres2[j] = hres[j] + __shfl_xor(hres[j], 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) has22[j] = has2[j] + __shfl_xor(has2[j], 32);
        if (lane < 32)
#pragma unroll
            for (int j = 0; j < 3; ++j) red[((wave >> 2) * 3 + j) * kHalf + 32 * (wave & 3) + lane] = hres2[j];
        float* alpha_red = red + 2 * 3 * kHalf;                 // [NW waves][64 lanes][AST]: halpha
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) alpha_red[(wave * 64 + lane) * AST + 16 * rb + 4 * g + i] = halpha[rb][g][i];
        float* bias_red = alpha_red + NT * AST;                 // [64][8]
        if (tid < kPts)
#pragma unroll
            for (int k = 0; k < 8; ++k) bias_red[tid * 8 + k] = hb[k];
        __syncthreads();
        float* out = p.head_partial + (size_t)blockIdx.x * kHeadFloats;
        if (tid < 3 * kHalf) out[kHeadRes + tid] = red[tid] + red[3 * kHalf + tid];                  // [j][c]
        if (lane < 32)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[kHeadAs2 + j * kWidth + 32 * wave + lane] = has22[j];
        // alpha: channel c of wave w = c / WCH: register slot (rb, g, i) with c % WCH = 32 rb + 8 g + 4 h + i; sum over the 32 lanes of half h
        if (tid < kWidth) {
            const int c = tid, w = c / WCH, cl = c % WCH, rb = cl >> 5, g = (cl >> 3) & 3, hh = (cl >> 2) & 1, i = cl & 3;
            float v = 0.0f;
            for (int l = 0; l < 32; ++l) v += alpha_red[(w * 64 + 32 * hh + l) * AST + 16 * rb + 4 * g + i];
            out[kHeadAlpha + c] = v;
        }
        if (tid < 8) {
            float v = 0.0f;
            for (int k = 0; k < kPts; ++k) v += bias_red[k * 8 + tid];
            out[kHeadBias + tid] = v;
        }
    }
    const float amax_all = fmaxf((float)amax2[0], (float)amax2[1]);
    if (p.status && __any(!(amax_all <= kF16Safe)) && lane == 0) atomicOr(p.status, INERF_STATUS_F16_RANGE);
    if (p.dz_max) {                       // non-negative floats order like their bit patterns
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
        if (lane == 0 && gmax == gmax) atomicMax(reinterpret_cast<unsigned int*>(p.dz_max), __builtin_bit_cast(unsigned int, gmax));
    }
}

// ================================================================================================
// Two workgroups per CU - the chain in the shape of the forward's k_encode_mlp_f16x3_dual (mlp_f16.hip).
//
// k_mlp_dgrad above keeps two 256-wide gradient buffers (A / B ping-pong) in LDS: 157 KB, one tile per CU, and its eight waves
// walk GEMM -> barrier -> epilogue -> barrier together - both waves of a SIMD are always in the same phase, each activation
// fragment is read from LDS by all eight waves (32 KB of LDS reads + 16 KB of L2 weight reads per k-block against 384 MFMA
// cycles per SIMD).  Here a workgroup is 4 waves x 64 channels x 64 points with ONE buffer updated in place - 75,776 B of LDS,
// <= 256 registers - so two tiles at different stages share a CU and fill each other's barriers and epilogues, and the
// 64-channel wave tile does 12 MFMAs per 4 LDS operand reads instead of 6.  What that took:
//   * in place: a layer's result waits in the accumulators across a barrier (every wave has read the layer's input) before
//     it overwrites the input; two barriers per layer instead of one;
//   * the head part re-sequenced around one buffer:  dZ_vh (128 columns) -> views^T -> d feature over the same columns ->
//     feat^T INTO the accumulators -> dZ_as1 over the same columns -> as1^T on top (-> dZ_semh, sem1^T on top) -> d h7;
//   * h7's ReLU mask comes from the forward's mask bits like h0..h6 (layout.h kReluBitLayers = 8) instead of from h7's
//     fragments transposed back by the matrix core (64 + 64 registers at a 64-channel wave tile);
//   * alpha_linear's weight gradient (d sigma^T h7) is accumulated where the other 1-4-row heads' are: in a fragment-native
//     VALU stage, lane = channel, straight from h7's fragments (2 accumulators per lane instead of 32).
// Same packed weights (the 4-wave packing), same slots, same head-partial layout; results differ from the eight-wave form only
// in the summation order of the head partials and in rays whose h7 sits below the fragments' 4e-9 floor (mask bit 1, decoded 0).
template <bool kSsr>
__global__ __launch_bounds__(256, 2) void k_mlp_dgrad_dual(const BwdParams p) {
    constexpr int kPts = kTilePoints;
    constexpr int RB = 2, NT = 256, WCH = 64;
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsb[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // (no `tid` kept: wave 0's lanes ARE threads 0..63, a wave-uniform test)
    const BwdLayout& L = p.L;
    f16x2 amax2 = {(_Float16)0.0f, (_Float16)0.0f};
    float gmax = 0.0f;

    _Float16* const xw = ldsb + (lane & 31) * kRowD;
    const _Float16* const xr = xw + 8 * (lane >> 5);              // wide GEMM operand reads: columns 0 ..
    _Float16* const xd = xw + 4 * (lane >> 5) + WCH * wave;       // wide stores: this wave's 64 channels
    // per-point scratch in the (unused) direction columns of the hi plane: [0..7] head gradients / s, [8] s, [9] 1/s
    auto ptf = [&](int pt) { return reinterpret_cast<float*>(ldsb + pt * kRowD + kColDirD); };

    WeightBuf wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wts), 0, L.total_floats * 4, 0x00020000);
    wb.voff = lane * 16;
    auto frag = [&](const GemmSlot& s, int kbt) { return (s.w + wave * kbt * 2 * 2 * 256) * 4; };
    const bool sem = kSsr && L.has_sem;
    const int ch = p.channels;

    WidePreH<RB> preA;
    const __amdgpu_buffer_rsrc_t bits_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.save) + p.bits_off, 0, (int)((unsigned)p.n_tiles * (unsigned)kReluBitTileBytes), 0x00020000);

    // weight gradients of the 1-4-row heads, accumulated over this workgroup's tiles (kHead*): ONE channel per lane and block
    const bool heads = p.head_partial != nullptr;
    // (sums of the head gradients = the heads' bias gradients: per thread 0..63 in the direction columns of the LO plane, row = thread -
    // eight accumulators that only one wave uses were eight registers of every wave, spilled)
    auto hbf = [&](int t) { return reinterpret_cast<float*>(ldsb + kPlaneD + t * kRowD + kColDirD); };
    if (wave == 0) {
        *reinterpret_cast<f32x4*>(hbf(lane)) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        *reinterpret_cast<f32x4*>(hbf(lane) + 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    float hres[3] = {0.0f, 0.0f, 0.0f};                                  // residual head: channel 32 wave + (lane & 31)
    float has2[2][4], halpha[2];                                          // albedo|shading outputs, alpha: channel 32 (2 wave + cbi) + (lane & 31)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        halpha[c] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) has2[c][j] = 0.0f;
    }

#ifdef INERF_DGRAD_STAMPS   // development build (scripts/build_variant.sh): dz_max points at 2 + 64 x uint64; cycle stamps of
    // workgroup 0 / thread 0 at the phase boundaries of its SECOND tile (steady state)
    unsigned long long* const dbg = (p.dz_max && blockIdx.x == 0 && threadIdx.x == 0) ? reinterpret_cast<unsigned long long*>(p.dz_max) + 1 : nullptr;
    int dbg_n = 0;
#define STAMP() do { if (dbg && tile == (int)gridDim.x && dbg_n < 62) { dbg[1 + dbg_n] = __builtin_readcyclecounter(); ++dbg_n; dbg[0] = dbg_n; } } while (0)
#else
#define STAMP() do { } while (0)
#endif
#ifndef INERF_DUAL_NO_STAGGER
    stagger_start(p.stagger);
#endif
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        STAMP();
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const int lh = lane_s >> 5;
        // wave 0: raw / d_raw of its 64 points, requested FIRST (head_inputs)
        const bool s0_valid = wave == 0 && tile * kPts + lane_s < p.n_points;
        // the views hidden layer's activations (fragments of a 128-channel slot: this wave's channel block, four k-blocks), requested a stage ahead
        f16x8 act_vh[4][2];                  // [k-block][hi | lo]
        {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.save) + p.off[SAVE_VH], 0,
                                                                                (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 2)), 0x00020000);
            const unsigned voff = (unsigned)tile * (unsigned)(kFragTileBytes / 2) + (unsigned)lane_s * 16u;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int plane = 0; plane < 2; ++plane)
                    act_vh[kb][plane] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(voff + frag_off<4>(kb, wave, plane)), 0, 0));
        }
        // ... and the albedo | shading hidden layer's and h7's (this wave's two channel blocks, four k-blocks each): operands of the
        // 1-4-row heads' weight gradients, and the ReLU mask of dZ_as1.  All of it is in flight while wave 0 computes the head
        // gradients below; consumed right behind that barrier, when no accumulator is live yet.  (Fetched where dZ_as1 is computed -
        // between two GEMMs whose accumulators occupy the registers, one k-block at a time - the stage waited for memory four times
        // per tile: 87 k of the tile's 315 k cycles, profiles/r05_dgrad_dual_timeline.txt.)
        f16x8 act_as1[2][4][2], act_h7[2][4][2];            // [channel block][k-block][hi | lo]
        {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.save) + p.off[SAVE_AS1H], 0,
                                                                                (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes), 0x00020000);
            const __amdgpu_buffer_rsrc_t r7 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.save) + p.off[SAVE_H7], 0,
                                                                                 (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes), 0x00020000);
            const unsigned voff = (unsigned)tile * (unsigned)kFragTileBytes + (unsigned)lane_s * 16u;
#pragma unroll
            for (int cbi = 0; cbi < 2; ++cbi)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int plane = 0; plane < 2; ++plane) {
                        const int o = (int)(voff + frag_off(kb, 2 * wave + cbi, plane));
                        act_as1[cbi][kb][plane] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0));
                        // (assigned on both paths: a register array that is only conditionally written is loop-carried state to the
                        // compiler - 64 registers kept alive, and spilled, across the whole tile)
                        if (heads) act_h7[cbi][kb][plane] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r7, o, 0, 0));
                        else act_h7[cbi][kb][plane] = f16x8{(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
                    }
        }
        // ---------------- heads: pre-activation gradients of the output heads, per-point scale ----------------
        if (wave == 0) {
            const int gp = tile * kPts + lane_s;
            float dp[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            float s = 1.0f;
            if (s0_valid) {
                // (Requested here, behind the tile's fragment operands.  Requesting them FIRST - so that they do not queue behind 160 KB
                // in the wave's vmcnt order - was no faster and made 2 % of the launches differ: k-block 3 of every slot of a
                // workgroup's non-first tiles, profiles/r05_chain_head_inputs_first.txt; the cause was not found, the form was dropped.)
                HeadInputs hin;
                head_inputs(p, gp, hin);
                s = head_gradients(p, gp, sem, hin, dp);
                float* __restrict__ o = p.dz + p.off[SAVE_DPRE] + (size_t)gp * 8;
                *reinterpret_cast<f32x4*>(o) = f32x4{dp[0], dp[1], dp[2], dp[3]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{dp[4], dp[5], dp[6], dp[7]};
                float* h = hbf(lane_s);       // (this thread's own words: no barrier needed)
                *reinterpret_cast<f32x4*>(h) += f32x4{dp[0], dp[1], dp[2], dp[3]};
                *reinterpret_cast<f32x4*>(h + 4) += f32x4{dp[4], dp[5], dp[6], dp[7]};
            }
            float* f = ptf(lane_s);
            const float is = 1.0f / s;
            *reinterpret_cast<f32x4*>(f) = f32x4{dp[0] * is, dp[1] * is, dp[2] * is, dp[3] * is};
            *reinterpret_cast<f32x4*>(f + 4) = f32x4{dp[4] * is, dp[5] * is, dp[6] * is, dp[7] * is};
            f[8] = s;
            f[9] = is;
            p.dz[p.off[SAVE_ENC] + gp] = s;        // the point's normaliser, beside the fragments (padding points: 1, with all-zero fragments)
        }
        STAMP();
        __syncthreads();
        STAMP();

        // ---------------- weight gradients of the albedo | shading outputs and of alpha_linear; ReLU mask of the hidden layer ----------------
        // lane = channel 32 (2 wave + cbi) + (lane & 31), the lane's eight points per k-block: d W_as2[j][c] += d_pre[j] as1h[c],
        // d w_alpha[c] += d sigma h7[c] (true gradients: normalised value x the point's scale).  What dZ_as1 needs of the hidden
        // layer later is only its sign: bit 8 kb + i of mask_as1[cbi].
        unsigned mask_as1[2] = {0u, 0u};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int ptb = 32 * (kb >> 1) + 16 * (kb & 1) + 4 * lh;         // frag_point(q, h, i) = (i & 3) + 8 (i >> 2) + 16 q + 4 h
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
                float sp = 0.0f, ds = 0.0f;
                if (heads) {                        // the point's scratch, once for both channel blocks
                    const float* f = ptf(ptb + (i & 3) + 8 * (i >> 2));
                    d = *reinterpret_cast<const f32x4*>(f);                                  // normalised d_albedo (3), d_shading
                    sp = f[8] * (1.0f / kActScale);
                    ds = f[7] * sp;                                                          // true d sigma / kActScale
                    d *= sp;
                }
#pragma unroll
                for (int cbi = 0; cbi < 2; ++cbi) {
                    const float a8 = (float)act_as1[cbi][kb][0][i] + (float)act_as1[cbi][kb][1][i];          // kActScale * activation
                    mask_as1[cbi] |= (a8 > 0.0f ? 1u : 0u) << (8 * kb + i);
                    if (heads) {
                        has2[cbi][0] += a8 * d[0];
                        has2[cbi][1] += a8 * d[1];
                        has2[cbi][2] += a8 * d[2];
                        has2[cbi][3] += a8 * d[3];
                        const float a7 = (float)act_h7[cbi][kb][0][i] + (float)act_h7[cbi][kb][1][i];
                        halpha[cbi] += a7 * ds;       // d alpha_linear.weight[c] += (true d sigma of the point) * h7[c]
                    }
                }
            }
            // (pinned: left free, the compiler postpones the comparisons to where the mask is used - behind two GEMMs - and keeps the 64 values instead)
            asm volatile("" : "+v"(mask_as1[0]), "+v"(mask_as1[1]));
            __builtin_amdgcn_sched_barrier(0);      // (one k-block at a time: left free, the scheduler reads every point's scratch first - 190 registers)
        }

        // ---------------- dZ of the view-dependent layer: relu'(vh) * (W_res^T d_res [+ d endpoint feature]) -> columns 0..127 ----------------
        // lane = channel 32 wave + (lane & 31); per k-block the lane's eight points (layout.h frag_point) -> the planes for the views^T GEMM.
        {
            const int cch = 32 * wave + (lane_s & 31);
            const f32x4 w4 = wb.vec4(L.res_w * 4, 16 * cch);                    // (W_res[0][c], W_res[1][c], W_res[2][c], 0)
            _Float16* const col = ldsb + cch;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    float t[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int pt = 32 * (kb >> 1) + frag_point(kb & 1, lh, i + e);
                        const float* f = ptf(pt);
                        const f32x4 d = *reinterpret_cast<const f32x4*>(f + 4);          // normalised d_res (3), d_sigma
                        const float sp = f[8];
                        const float act = ((float)act_vh[kb][0][i + e] + (float)act_vh[kb][1][i + e]) * (1.0f / kActScale);
                        float v = w4[0] * d[0] + w4[1] * d[1] + w4[2] * d[2];
                        if (kSsr && p.endpoint) {       // raw[..., -128:] is this layer's output itself (semantic_nerf.py:163-164)
                            const int gp = tile * kPts + pt;
                            if (gp < p.n_points) v = __builtin_fmaf(p.d_raw[(size_t)gp * ch + ch - INERF_ENDPOINT_DIM + cch], f[9], v);
                        }
                        if (heads) {                   // d W_res[j][c] += (true d_res_pre[j] of the point) * vh[c]
                            hres[0] += act * (d[0] * sp);
                            hres[1] += act * (d[1] * sp);
                            hres[2] += act * (d[2] * sp);
                        }
                        v = act > 0.0f ? v : 0.0f;         // (a point beyond the end: its head gradients are zero -> v = 0)
                        gmax = fmaxf(gmax, fabsf(v) * sp);
                        t[e] = v * kActScale;
                    }
                    f16x2 h2, l2;
                    split_pair(t[0], t[1], h2, l2);
                    amax2 = __builtin_elementwise_max(amax2, __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h2) & 0x7FFF7FFFu));
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int pt = 32 * (kb >> 1) + frag_point(kb & 1, lh, i + e);
                        col[pt * kRowD] = h2[e];
                        col[pt * kRowD + kPlaneD] = l2[e];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        prefetch_w<RB>(preA, wb, frag(L.views_t, 8));
        STAMP();
        __syncthreads();
        STAMP();

        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        const int pt0 = tile * kPts + (lane_t & 31);
        const bool valid0 = pt0 < p.n_points, valid1 = pt0 + 32 < p.n_points;
        const float s0 = ptf(lane_t & 31)[8], s1 = ptf((lane_t & 31) + 32)[8];
        auto mask_words = [&](int layer) {            // this lane's two mask words of trunk layer `layer` (layout.h relu_bits_offset)
            const int mbase = (((tile * kReluBitLayers + layer) * 4 + wave) * 64) * 8;
            return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(bits_rsrc, lane_t * 8, mbase, 0));
        };
        f32x16 am[RB][2];
        NoAlpha none;
        // Every dZ slot leaves the CU from the PLANES, right behind the loop of the GEMM that consumes it (its input stays intact until
        // that GEMM's closing barrier): the stores then have the whole following epilogue to complete.  Issued in front of a GEMM
        // - as the epilogue's last act, or by the VALU stages - they stand before that GEMM's weight loads in the wave's vmcnt order,
        // and its first k-blocks wait for them (mlp_f16_dev.h planes_to_frag_late).
        auto flush256 = [&](int slot) {                // this wave's 64 channels of a 256-wide slot
            int lane_o = lane_t;
            asm volatile("" : "+v"(lane_o));
            FragDst d;
            d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.dz + p.off[slot], 0, (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes), 0x00020000);
            d.voff = (unsigned)tile * (unsigned)kFragTileBytes + (unsigned)(2 * wave) * (2u * kFragBytes) + (unsigned)lane_o * 16u;
            planes_to_frag_late<2, kRowD, kPlaneD>(xr + 64 * wave, plane_selector(lane_o), d);
        };
        auto flush128 = [&](int slot) {                // this wave's 32 channels of a 128-wide slot (columns 0..127)
            int lane_o = lane_t;
            asm volatile("" : "+v"(lane_o));
            FragDst d;
            d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.dz + p.off[slot], 0, (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 2)), 0x00020000);
            d.voff = (unsigned)tile * (unsigned)(kFragTileBytes / 2) + (unsigned)wave * (2u * kFragBytes) + (unsigned)lane_o * 16u;
            planes_to_frag_late<1, kRowD, kPlaneD, 4>(xr + 32 * wave, plane_selector(lane_o), d);
        };

        // ---------------- d feature = W_views^T[:256] dZ_vh, in place (feature_linear has no activation: this is its dZ) ----------------
        {
            const float inv = wb.scalar(L.views_t.b * 4);        // (requested ahead of the GEMM: behind it, its L2 round trip is exposed)
            wide_gemm_h<RB, 8, 0, kRowD, kPlaneD, true>(preA, wb, frag(L.views_t, 8), xr, 0, 0, lane, am);
            prefetch_w<RB>(preA, wb, frag(L.feat_t, 16));
            flush128(SAVE_VH);
            STAMP();
            __syncthreads();                 // dZ_vh has been read by every wave
            STAMP();
            bwd_store<RB, false, NoAlpha, kRowD, kPlaneD, false>(am, inv, nullptr, nullptr, 0.0f, 0.0f, xd, amax2, DzDst{}, lane_t, s0, s1,
                                                                 valid0, valid1, gmax, u32x2{0u, 0u}, none);
        }
        STAMP();
        __syncthreads();
        STAMP();

        // ---------------- d h7 = W_feat^T d feature + W_as1^T dZ_as1 (+ W_sem1^T dZ_semh) + w_alpha d sigma ----------------
        wide_gemm_h<RB, 16, 0, kRowD, kPlaneD, true>(preA, wb, frag(L.feat_t, 16), xr, 0, 0, lane, am);
        flush256(SAVE_FEAT);
        STAMP();
        __syncthreads();                     // d feature has been read by every wave; its product waits in the accumulators
        STAMP();

        // dZ of the albedo | shading hidden layer: relu'(as1h) * (W_as2^T [d_albedo, d_shading]) over the same columns; lane = channel
        // 32 (2 wave + cbi) + (lane & 31), four k-blocks.  No memory operand: the head gradients from the scratch columns, the ReLU
        // mask from mask_as1.
        {
#pragma unroll
            for (int cbi = 0; cbi < 2; ++cbi) {
                const f32x4 w4 = wb.vec4(L.as2_w * 4, 16 * (32 * (2 * wave + cbi) + (lane_s & 31)));      // albedo_linear2[0..2][c] | shading output [c - 128]
                _Float16* const col = ldsb + 32 * (2 * wave + cbi) + (lane_s & 31);
#pragma unroll 1
                for (int kb = 0; kb < 4; ++kb) {
                    const int ptb = 32 * (kb >> 1) + 16 * (kb & 1) + 4 * lh;
                    const int bits8 = (int)(mask_as1[cbi] >> (8 * kb));
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        float t[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float* f = ptf(ptb + ((i + e) & 3) + 8 * ((i + e) >> 2));
                            const f32x4 d = *reinterpret_cast<const f32x4*>(f);              // normalised d_albedo (3), d_shading
                            float v = w4[0] * d[0] + w4[1] * d[1] + w4[2] * d[2] + w4[3] * d[3];
                            v = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & __builtin_amdgcn_sbfe(bits8, i + e, 1));      // relu'
                            gmax = fmaxf(gmax, fabsf(v) * f[8]);
                            t[e] = v * kActScale;
                        }
                        f16x2 h2, l2;
                        split_pair(t[0], t[1], h2, l2);
                        amax2 = __builtin_elementwise_max(amax2, __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h2) & 0x7FFF7FFFu));
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int pt = ptb + ((i + e) & 3) + 8 * ((i + e) >> 2);
                            col[pt * kRowD] = h2[e];
                            col[pt * kRowD + kPlaneD] = l2[e];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        prefetch_w<RB>(preA, wb, frag(L.as1_t, 16));          // (after the stage, not before: its 32 registers next to the live accumulators spilled)
        STAMP();
        __syncthreads();
        STAMP();
        wide_gemm_h<RB, 16, 0, kRowD, kPlaneD, false>(preA, wb, frag(L.as1_t, 16), xr, 0, 0, lane, am);
        if (sem) prefetch_w<RB>(preA, wb, frag(L.sem1_t, 8));
        else     prefetch_w<RB>(preA, wb, frag(L.trunk_t[7], 16));
        flush256(SAVE_AS1H);
        if (sem) {
            __syncthreads();                 // dZ_as1 has been read by every wave
            constexpr int VH_STEP = NT / 32, VH_IT = kPts / VH_STEP;        // rows per pass of this row-wise stage, passes
            const int c4 = (lane_t & 31) * 4;
#pragma unroll 1
            for (int i = 0; i < VH_IT; ++i) {
                const int pt = 2 * wave + (lane_t >> 5) + VH_STEP * i;
                const int gp = tile * kPts + pt;
                const bool valid = gp < p.n_points;
                const float* f = ptf(pt);
                float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                f32x4 act = {0.0f, 0.0f, 0.0f, 0.0f};
                if (valid) {
                    const float* gl = p.d_raw + (size_t)gp * ch + INERF_BASE_CHANNELS;       // logits: no activation
                    const float* wt = p.wts + L.sem2_w + c4;
                    const float is = f[9];
                    act = *reinterpret_cast<const f32x4*>(p.save + p.off[SAVE_SEMH] + (size_t)gp * kHalf + c4);
                    // four classes per step: their five loads (one unaligned 16-byte piece of the point's logit gradients, four weight
                    // rows) are in flight together - one class at a time every step waited for its own two loads (0.6 ms of the SSR step)
                    int j = 0;
                    for (; j + 4 <= p.n_classes; j += 4) {
                        f32x4 g4;
                        __builtin_memcpy(&g4, gl + j, 16);
                        f32x4 w[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const f32x4*>(wt + (size_t)(j + q) * kHalf);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float gj = g4[q] * is;
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) v[cc] = __builtin_fmaf(w[q][cc], gj, v[cc]);
                        }
                    }
                    for (; j < p.n_classes; ++j) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(wt + (size_t)j * kHalf);
                        const float gj = gl[j] * is;
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) v[cc] = __builtin_fmaf(w[cc], gj, v[cc]);
                    }
                }
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) v[cc] = act[cc] > 0.0f ? v[cc] : 0.0f;
                split_store4(ldsb + pt * kRowD + c4, ldsb + pt * kRowD + c4 + kPlaneD, v, amax2);
                if (valid) {
                    const f32x4 o = f32x4{v[0], v[1], v[2], v[3]} * f[8];
                    gmax = fmaxf(gmax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                }
            }
            STAMP();
        __syncthreads();
        STAMP();
            wide_gemm_h<RB, 8, 0, kRowD, kPlaneD, false>(preA, wb, frag(L.sem1_t, 8), xr, 0, 0, lane, am);
            prefetch_w<RB>(preA, wb, frag(L.trunk_t[7], 16));
            flush128(SAVE_SEMH);             // dZ of the semantic hidden layer: fragments of a 128-channel slot, one channel block per wave
        }
        {
            const float inv = wb.scalar(L.feat_t.b * 4);            // common scale of feat_t / as1_t / sem1_t
            const float e0 = ptf(lane_t & 31)[7], e1 = ptf((lane_t & 31) + 32)[7];
            const u32x2 mbits = mask_words(kDepth - 1);
            STAMP();
            __syncthreads();                 // every wave is done reading the buffer
            STAMP();
            // one 32-channel row block at a time: alpha_linear's weights of the block (16 registers) instead of the wave's (32) beside
            // the accumulators, the next layer's first fragments and the epilogue's own transposition
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                f32x4 aw[1][4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    aw[0][g] = wb.vec4((L.alpha_w + WCH * wave + 32 * rb + 8 * g) * 4, 16 * (lane_t >> 5)) * kActScale;
                bwd_store<1, true, NoAlpha, kRowD, kPlaneD, false>(*reinterpret_cast<const f32x16 (*)[1][2]>(&am[rb]), inv, nullptr, aw, e0, e1, xd + 32 * rb,
                                                                   amax2, DzDst{}, lane_t, s0, s1, valid0, valid1, gmax, u32x2{mbits[rb], 0u}, none);
            }
        }
        STAMP();
        __syncthreads();
        STAMP();

        // ---------------- trunk, layers 7..1: dZ_{l-1} = relu'(h_{l-1}) * W_l^T dZ_l, in place ----------------
#pragma unroll 1
        for (int l = kDepth - 1; l >= 1; --l) {
            const u32x2 mbits = mask_words(l - 1);
            const float inv = wb.scalar(L.trunk_t[l].b * 4);
            wide_gemm_h<RB, 16, 0, kRowD, kPlaneD, true>(preA, wb, frag(L.trunk_t[l], 16), xr, 0, 0, lane, am);
            if (l > 1) prefetch_w<RB>(preA, wb, frag(L.trunk_t[l - 1], 16));      // (the next tile's first GEMM: requested behind that tile's VALU stages, whose operands need the registers)
            flush256(SAVE_H0 + l);           // dZ_l, which this GEMM has just read
            STAMP();
            __syncthreads();                 // dZ_l has been read by every wave
            STAMP();
            bwd_store<RB, true, NoAlpha, kRowD, kPlaneD, false>(am, inv, nullptr, nullptr, 0.0f, 0.0f, xd, amax2, DzDst{}, lane_t, s0, s1,
                                                                valid0, valid1, gmax, mbits, none);
            STAMP();
            __syncthreads();
            STAMP();
        }
        flush256(SAVE_H0);                   // dZ_0 has no consumer in this kernel: before the next tile's stages overwrite the planes (behind their first barrier)
    }
    if (heads) {                          // lane halves meet by a shuffle (same channel, the other four points of every k-block)
        float hres2[3], has22[2][4], halpha2[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) hres2[j] = hres[j] + __shfl_xor(hres[j], 32);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            halpha2[c] = halpha[c] + __shfl_xor(halpha[c], 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) has22[c][j] = has2[c][j] + __shfl_xor(has2[c][j], 32);
        }
        __syncthreads();
        float* out = p.head_partial + (size_t)blockIdx.x * kHeadFloats;
        if (lane < 32) {
#pragma unroll
            for (int j = 0; j < 3; ++j) out[kHeadRes + j * kHalf + 32 * wave + lane] = hres2[j];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int j = 0; j < 4; ++j) out[kHeadAs2 + j * kWidth + 32 * (2 * wave + c) + lane] = has22[c][j];
                out[kHeadAlpha + 32 * (2 * wave + c) + lane] = halpha2[c];
            }
        }
        if (wave == 0 && lane < 8) {
            float v = 0.0f;
            for (int k = 0; k < kPts; ++k) v += hbf(k)[lane];
            out[kHeadBias + lane] = v;
        }
    }
    const float amax_all = fmaxf((float)amax2[0], (float)amax2[1]);
    if (p.status && __any(!(amax_all <= kF16Safe)) && lane == 0) atomicOr(p.status, INERF_STATUS_F16_RANGE);
    if (p.dz_max) {                       // non-negative floats order like their bit patterns
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
        if (lane == 0 && gmax == gmax) atomicMax(reinterpret_cast<unsigned int*>(p.dz_max), __builtin_bit_cast(unsigned int, gmax));
    }
}

// Which form runs: two workgroups per CU (default) or the eight-wave single-tile kernel (INERF_DGRAD_KERNEL=single: A/B runs and the
// kernel's own tests).  The environment is read per LAUNCH only: the head-partial buffer is always sized for the larger grid
// (inerf_mlp_backward_grid), so a variable that changes between the caller's sizing query and the launch cannot make the
// two-workgroup kernel write beyond a buffer sized for the other form; blocks a launch does not write are zeroed.
static bool dgrad_dual() {
    const char* form = getenv("INERF_DGRAD_KERNEL");
    return !(form && form[0] == 's');
}

}  // namespace inerf

// floats per workgroup of the head-gradient partials, and the number of workgroups inerf_mlp_backward_inputs launches
extern "C" int inerf_mlp_head_partial_floats(void) { return inerf::kHeadFloats; }
extern "C" int inerf_mlp_backward_grid(int64_t n_points) {
    using namespace inerf;
    if (n_points <= 0) return 0;
    const int64_t tiles = (n_points + kTilePoints - 1) / kTilePoints;
    const int64_t max_grid = 2 * device_cus();          // the two-workgroup form's: an upper bound for both forms
    return (int)(tiles < max_grid ? tiles : max_grid);
}

extern "C" int inerf_mlp_backward_inputs(const inerf_net_desc* net, const float* packed_bwd, const float* raw, const float* d_raw,
                                         const float* save, int64_t n_points, uint32_t flags, float* dz_out, float* dz_max,
                                         float* head_partial, int32_t* status, void* stream) {
    using namespace inerf;
    if (net && n_points == 0) return INERF_OK;
    if (!net || !packed_bwd || !raw || !d_raw || !save || !dz_out || n_points < 0) return INERF_E_INVALID;
    if (!net_supported(*net)) return INERF_E_UNSUPPORTED;
    if (n_points == 0) return INERF_OK;
    if (n_points > kMaxTrainPoints) return INERF_E_UNSUPPORTED;       // the slots are read through 32-bit buffer descriptors
    const bool ssr = net->variant == INERF_VARIANT_SSR;
    BwdParams p;
    p.wts = packed_bwd; p.raw = raw; p.d_raw = d_raw; p.save = save; p.dz = dz_out; p.dz_max = dz_max; p.head_partial = head_partial; p.status = status;
    for (int s = 0; s < SAVE_SLOTS; ++s) p.off[s] = save_offset(*net, s, n_points);
    p.bits_off = relu_bits_offset(*net, n_points);
    p.L = make_bwd_layout(*net);
    p.n_points = (int)n_points;
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    p.endpoint = (ssr && (flags & INERF_FLAG_ENDPOINT)) ? 1 : 0;
    p.n_classes = ssr ? net->n_classes : 0;
    p.channels = INERF_BASE_CHANNELS + p.n_classes + (p.endpoint ? INERF_ENDPOINT_DIM : 0);
    const bool dual = dgrad_dual();
    const int max_grid = (dual ? 2 : 1) * device_cus();
    const int grid = p.n_tiles < max_grid ? p.n_tiles : max_grid;
    p.stagger = stagger_units(p.n_tiles, grid);
    // two workgroups of four waves per CU (64 channels per wave), or eight waves per workgroup (two per SIMD, 32 channels each)
    void (*kern)(const BwdParams) = dual ? (ssr ? k_mlp_dgrad_dual<true> : k_mlp_dgrad_dual<false>) : (ssr ? k_mlp_dgrad<true, 8> : k_mlp_dgrad<false, 8>);
    const int lds = dual ? kLdsBytesD : kLdsBytesH;
    static PerDeviceOnce attr_set[4];
    const int variant = 2 * (int)dual + (int)ssr;
    if (attr_set[variant].first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return record(e);
        attr_set[variant].mark();
    }
    const int sized_for = inerf_mlp_backward_grid(n_points);          // what the caller's head-partial buffer holds
    if (head_partial && grid < sized_for) {                           // (the eight-wave form: one workgroup per CU)
        hipError_t e = hipMemsetAsync(head_partial + (size_t)grid * kHeadFloats, 0, (size_t)(sized_for - grid) * kHeadFloats * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return record(e);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(dual ? 256 : 512), lds, (hipStream_t)stream, p);
    return record(hipGetLastError());
}
