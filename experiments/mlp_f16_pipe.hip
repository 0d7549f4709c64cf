// mlp_f16_pipe.hip - the split-f16 encode+MLP kernel with register-resident weights and a software-pipelined epilogue.
// Arithmetic, packed blob, LDS row format and outputs are those of mlp_f16.hip; what differs is who waits for whom.
// Own translation unit: built with -mllvm -amdgpu-mfma-vgpr-form=1 (MFMA results in ordinary VGPRs, where the epilogue's
// VALU instructions can read them; the 256 resident weight registers go to the accumulation-register half of the file).
#include <stdlib.h>

#include "mlp_f16_heads.h"

#ifndef PIPE_ABL
#define PIPE_ABL 0          // timing experiments (scripts/build_ablations.sh): bit 0 no epilogue LDS writes, 1 no epilogue at all,
#endif                      // 2 operands read once per phase, 3 no barriers, 4 no weight refill.  Results are wrong when set.
#if PIPE_ABL & 8
#define PIPE_SYNC() ((void)0)
#else
#define PIPE_SYNC() __syncthreads()
#endif

namespace inerf {

// ================================================================================================
// "pipe": four waves (one per SIMD, up to 512 registers each), 128-point tiles worked as two 64-point halves A / B,
// a layer's weights RESIDENT in registers for both halves, the epilogue of one half hidden under the other half's MFMAs.
//
// The three kernels above leave the matrix pipe idle in two ways.  (1) Every weight fragment is fetched from L2 once per 64
// points (once per 128 in the quad form) - the vector-memory path, not the matrix pipe, paces the GEMM.  (2) Converting a
// layer's accumulators (bias, ReLU, hi/lo split, LDS stores: ~10 instructions per value pair and 131 KB of LDS writes per
// layer and CU) happens while that wave issues no MFMAs; only another workgroup in a different phase covers it.
// Here one wave owns 64 output channels (two row blocks) and works the tile's halves one after the other:
//     phase (L, A): acc_X += W_L . rows A        |  meanwhile the same wave retires acc_Y = layer L-1 of half B: 1/16 of it
//     phase (L, B): acc_Y += W_L . rows B        |  per k-block step (bias, ReLU, split, two ds_write_b64) into rows B / A
// W_L's 64 fragments per wave (256 registers) are loaded ONCE: during phase (L-1, B) each slot is refilled with layer L's
// fragment as soon as step s has used it - a whole phase (6 000 cycles) ahead of its first use, so no MFMA ever waits for
// L2 - and serve both halves: the weight stream per sample point is half the two-workgroup kernel's.  A half's rows are
// only ever rewritten a barrier after their last reader (in place, like the two-workgroup kernel), one barrier per phase.
// Layers with 64-wide inputs (layer 0, the encoding part of the skip layer) run both halves in one pass over their
// 16 fragments; their epilogue, the register-operand heads and the encodings are the parts that still run exposed.
// Register plan per lane: 256 resident weights + 2 x 64 accumulators + 32 activation fragments + 16 bias ring + temporaries.
// ================================================================================================
constexpr int kPtsP = 128;
constexpr int kPlaneP = kPtsP * kRowD;           // 128 rows of kRowD halfs
constexpr int kLdsBytesP = 2 * kPlaneP * 2 + kPtsP * 4 * 4 * 4;      // 151,552 + 8,192 (parked head partials)

typedef f16x8 ResidentA[16][2][2];               // [k-block][row block][hi | lo]

// The 128-row planes are 75,776 B each: the lo plane and the rows of half B lie beyond the 16-bit offset field of the DS
// instructions.  Left alone, the compiler forms one address register per distinct (base + large constant), hoists ~100 of
// them out of the layer loops and spills them.  Instead every plane / half gets ONE base offset that is opaque to
// constant folding; everything else (point block, k-block, channel) stays a small immediate.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// fragment (kb, rb, part) of a 2-row-block layer whose stream for this wave starts at frag_bytes
__device__ __forceinline__ int frag_at(int frag_bytes, int kb, int rb, int part, int rbs) { return frag_bytes + ((kb * rbs + rb) * 2 + part) * 1024; }

// One pipelined phase.  acc[rb][pb] (+)= A . X(rows at xl), KBT k-blocks, RB row blocks (RB = 1: the 128-channel view layer,
// whose k-blocks 16, 17 live in the rb = 1 slots 0, 1); meanwhile, one chunk per step:
//   EPI   : prev (the previous phase's accumulators, a 256-channel layer) -> t = prev * einv + bias (ReLU if erelu) -> hi/lo
//           halves -> LDS at dl (the rows of prev's half; dl includes this wave's channel offset and the lane's part);
//   REFILL: the slot(s) step s has just used are reloaded with the NEXT layer's fragments: RF = 2: a 2-row-block layer with
//           `next_kbt` k-blocks from k-block `next_first` on (slot s <- k-block s); RF = 1: the view layer (slot [s][0] <-
//           k-block s, [s][1] <- k-block 16 + s); RF = 3: as RF = 2 but slots 0..3 come from `alt_frag_bytes` (4 k-blocks).
template <int KBT, int RB, bool EPI, int RF, bool ZERO = true>
__device__ __forceinline__ void pipe_phase(ResidentA& A, const WeightBuf& wb, _Float16* lds, int x_hi, int x_lo /* element offsets of the
                                           operand rows: (row0 + (lane&31))*kRowD + 8*(lane>>5) [+ kPlaneP], opaque */, f32x16 (&acc)[2][2],
                                           const f32x16 (&prev)[2][2], int d_hi, int d_lo /* store offsets of prev's rows, or 0 */,
                                           f32x4 (&bring)[4] /* bias ring: holds vector pairs 0..2 of this phase's epilogue on entry, of the
                                           NEXT phase's (bias at nbias_bytes, 0 = none) on exit - no phase starts by waiting for L2 */,
                                           int ebias_bytes, int nbias_bytes, float einv, bool erelu,
                                           f16x2& amax2, int lane, int next_frag_bytes, int alt_frag_bytes = 0) {
    if constexpr (ZERO) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][pb][r] = 0.0f;
    }
    f16x8 x[2][2][2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int part = 0; part < 2; ++part) x[0][pb][part] = *reinterpret_cast<const f16x8*>(lds + (part ? x_lo : x_hi) + pb * 32 * kRowD);
    // bias ring: chunk c = 8*rb + 2*g + pb uses bias vector (rb, g); one vector serves two consecutive chunks and is
    // requested three chunk pairs (six steps) ahead
    const int h16 = 16 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < KBT; ++s) {
        const int s1 = s + 1 < KBT ? s + 1 : KBT - 1;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                if (!(PIPE_ABL & 4)) x[(s + 1) & 1][pb][part] = *reinterpret_cast<const f16x8*>(lds + (part ? x_lo : x_hi) + 16 * s1 + pb * 32 * kRowD);
                else x[(s + 1) & 1][pb][part] = x[s & 1][pb][part];
        // this step's weight fragments
        // product-major over (row block, point block): an accumulator is touched every 2*RB-th MFMA
#pragma unroll
        for (int combo = 0; combo < 3; ++combo)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const f16x8& wh = (RB == 1 && s >= 16) ? A[s - 16][1][0] : A[s][rb][0];
                const f16x8& wl = (RB == 1 && s >= 16) ? A[s - 16][1][1] : A[s][rb][1];
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
                    acc[rb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(combo == 2 ? wl : wh, x[s & 1][pb][combo == 1 ? 1 : 0], acc[rb][pb], 0, 0, 0);
            }
        if constexpr (EPI && !(PIPE_ABL & 2)) {
            if (s < 16) {
                const int c = s, rb = c >> 3, g = (c >> 1) & 3, pb = c & 1;
                if ((c & 1) == 0 && (c >> 1) + 3 < 8) {                       // request the bias vector of chunk pair c/2 + 3
                    const int v = (c >> 1) + 3;
                    bring[v & 3] = wb.vec4(ebias_bytes + (32 * (v >> 2) + 8 * (v & 3)) * 4, h16);
                }
                const f32x4 bias = bring[(c >> 1) & 3];
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    t[i] = __builtin_fmaf(prev[rb][pb][4 * g + i], einv, bias[i]);
                    if (erelu) t[i] = fmaxf(t[i], 0.0f);
                }
                f16x2 h01, h23, l01, l23;
                split_pair(t[0], t[1], h01, l01);
                split_pair(t[2], t[3], h23, l23);
                f16x2 a01 = h01, a23 = h23;
                if (!erelu) {
                    a01 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h01) & 0x7FFF7FFFu);
                    a23 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h23) & 0x7FFF7FFFu);
                }
                amax2 = __builtin_elementwise_max(amax2, __builtin_elementwise_max(a01, a23));
                const f16x4 hi4 = {h01[0], h01[1], h23[0], h23[1]}, lo4 = {l01[0], l01[1], l23[0], l23[1]};
                const int doff = pb * 32 * kRowD + 32 * rb + 8 * g;
                if (!(PIPE_ABL & 1)) {
                    *reinterpret_cast<f16x4*>(lds + d_hi + doff) = hi4;
                    *reinterpret_cast<f16x4*>(lds + d_lo + doff) = lo4;
                } else {
                    amax2 = __builtin_elementwise_max(amax2, __builtin_elementwise_max(f16x2{lo4[0], lo4[1]}, f16x2{lo4[2], lo4[3]}));
                }
            }
        }
        if (nbias_bytes != 0 && (s == 10 || s == 12 || s == 14)) {              // ring slots 0..2 are free again: next phase's pairs 0..2
            const int v = (s - 10) >> 1;
            bring[v] = wb.vec4(nbias_bytes + 8 * v * 4, h16);
        }
        if constexpr ((RF == 2 || RF == 3) && !(PIPE_ABL & 16)) {
            if (s < 16) {
                const int base = (RF == 3 && s < 4) ? alt_frag_bytes : next_frag_bytes;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int part = 0; part < 2; ++part) A[s][rb][part] = wb.frag(frag_at(base, s, rb, part, 2));
            }
        } else if constexpr (RF == 1 && !(PIPE_ABL & 16)) {
            if (s < 16) {
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    A[s][0][part] = wb.frag(frag_at(next_frag_bytes, s, 0, part, 1));
                    if (s < 2) A[s][1][part] = wb.frag(frag_at(next_frag_bytes, 16 + s, 0, part, 1));
                }
            }
        }
        // issue order: MFMA, (weight refill), MFMA, LDS read, epilogue VALU, MFMA ...  (0x8 MFMA, 0x20 VMEM read, 0x100 DS read, 0x2 VALU, 0x200 DS write)
#pragma unroll
        for (int q = 0; q < 2 * RB; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (RF != 0) __builtin_amdgcn_sched_group_barrier(0x020, RB == 2 ? 1 : 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RB == 2 ? 1 : 2, 0);
            if (EPI) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (EPI) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            if (EPI && q >= 2 * RB - 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// a 64-wide-input layer (layer 0 / the encoding part of the skip layer) for BOTH halves in one pass over its 16
// resident fragments (slots 0..3): accA (+)= W . rows A, accB (+)= W . rows B
template <bool ZERO>
__device__ __forceinline__ void pipe_both4(const ResidentA& A, const _Float16* lds, const int (&xo)[2][2] /* [half][hi|lo] operand offsets */,
                                           f32x16 (&accA)[2][2], f32x16 (&accB)[2][2]) {
    if constexpr (ZERO) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) { accA[rb][pb][r] = 0.0f; accB[rb][pb][r] = 0.0f; }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f16x8 x[4][2];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                x[pb][part] = *reinterpret_cast<const f16x8*>(lds + xo[pb >> 1][part] + 16 * s + (pb & 1) * 32 * kRowD);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                f32x16& a = pb < 2 ? accA[rb][pb] : accB[rb][pb - 2];
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][rb][0], x[pb][0], a, 0, 0, 0);
            }
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                f32x16& a = pb < 2 ? accA[rb][pb] : accB[rb][pb - 2];
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][rb][0], x[pb][1], a, 0, 0, 0);
            }
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                f32x16& a = pb < 2 ? accA[rb][pb] : accB[rb][pb - 2];
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][rb][1], x[pb][0], a, 0, 0, 0);
            }
        }
    }
}

// A 4-row output head fed straight from a hidden layer's accumulators: per 16-channel k-block of this wave's RB row
// blocks, (bias, ReLU, hi/lo split) of both point blocks -> B operands -> three MFMAs per point block against the head's
// register-operand fragments (layout.h: as2r / resr).  One k-block at a time, so that nothing but the 2 x 16 result
// registers stays live (converting the whole layer first needs 64 operand registers and spilled).
template <int RB>
__device__ __forceinline__ void regop_head(const WeightBuf& wb, int frag_bytes, const f32x16 (&acc)[2][2], float inv, int bias_bytes,
                                           int lane, f16x2& amax2, f32x4 (&part)[2]) {
    f32x16 r[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int i = 0; i < 16; ++i) r[pb][i] = 0.0f;
    const int h16 = 16 * (lane >> 5);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        f32x4 bias[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[g] = wb.vec4(bias_bytes + (32 * rb + 8 * g) * 4, h16);
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            const int q = 2 * rb + q2;
            const f16x8 wh = wb.frag(frag_bytes + (2 * q) * 1024), wl = wb.frag(frag_bytes + (2 * q + 1) * 1024);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                f16x8 fh, fl;
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const int j = 8 * q2 + i;
                    const float t0 = fmaxf(__builtin_fmaf(acc[rb][pb][j], inv, bias[j >> 2][j & 3]), 0.0f);
                    const float t1 = fmaxf(__builtin_fmaf(acc[rb][pb][j + 1], inv, bias[(j + 1) >> 2][(j + 1) & 3]), 0.0f);
                    f16x2 h2, l2;
                    split_pair(t0, t1, h2, l2);
                    amax2 = __builtin_elementwise_max(amax2, h2);
                    fh[i] = h2[0]; fh[i + 1] = h2[1];
                    fl[i] = l2[0]; fl[i + 1] = l2[1];
                }
                r[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fh, r[pb], 0, 0, 0);
                r[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fl, r[pb], 0, 0, 0);
                r[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fh, r[pb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) part[pb] = f32x4{r[pb][0], r[pb][1], r[pb][2], r[pb][3]};
}

__device__ __forceinline__ void pipe_load(ResidentA& A, const WeightBuf& wb, int frag_bytes, int first_kb, int n_kb, int slot0) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < n_kb) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int part = 0; part < 2; ++part) A[slot0 + k < 16 ? slot0 + k : 15][rb][part] = wb.frag(frag_at(frag_bytes, first_kb + k, rb, part, 2));
        }
}

template <bool kSsr>
__global__ __launch_bounds__(256, 1) void k_encode_mlp_f16x3_pipe(const MlpParams p) {
    constexpr int kPts = kPtsP;                 // 128
    constexpr int kParts = 256 / 64;            // encode: 64 rows x 4 frequency parts per pass, two passes (halves)
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsp[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NetLayout& L = p.L;
    float amax = 0.0f;
    f16x2 amax2 = {(_Float16)0.0f, (_Float16)0.0f};

    // element offsets into ldsp, one opaque base per (half, plane): [half][hi | lo]
    const int row = (lane & 31) * kRowD;
    const int xo[2][2] = {{opaque(row + 8 * (lane >> 5)), opaque(row + 8 * (lane >> 5) + kPlaneP)},
                          {opaque(row + 8 * (lane >> 5) + 64 * kRowD), opaque(row + 8 * (lane >> 5) + 64 * kRowD + kPlaneP)}};   // operand reads
    const int dw = row + 4 * (lane >> 5) + 64 * wave;                     // stores: this wave's 64 channels
    const int dO[2][2] = {{opaque(dw), opaque(dw + kPlaneP)}, {opaque(dw + 64 * kRowD), opaque(dw + 64 * kRowD + kPlaneP)}};

    WeightBuf wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wts), 0, L.total_floats * 4, 0x00020000);
    wb.voff = lane * 16;
    auto frag256 = [&](const GemmSlot& s, int kbt) { return (s.w + wave * kbt * 2 * 2 * 256) * 4; };
    auto frag128 = [&](const GemmSlot& s, int kbt) { return (s.w + wave * kbt * 1 * 2 * 256) * 4; };
    auto bias256 = [&](const GemmSlot& s) { return (s.b + 64 * wave) * 4; };
    auto scale256 = [&](const GemmSlot& s) { return wb.scalar((s.b + kWidth) * 4); };      // requested a phase before its first use
    f32x4 bring[4];
    auto prime = [&](int bias_bytes) {
#pragma unroll
        for (int v = 0; v < 3; ++v) bring[v] = wb.vec4(bias_bytes + 8 * v * 4, 16 * (lane >> 5));
    };

    // development aid (inerf_debug_encode_mlp): cycle stamps of workgroup 0, wave 0 at the phase boundaries of its first tile
    unsigned long long* const dbg = (p.act_max && blockIdx.x == 0 && tid == 0) ? reinterpret_cast<unsigned long long*>(p.act_max) : nullptr;
    int dbg_n = 0;
    auto stamp = [&]() { if (dbg && dbg_n < 63) { dbg[1 + dbg_n] = __builtin_readcyclecounter(); ++dbg_n; dbg[0] = dbg_n; } };
    // constants of the two output heads (tile-invariant)
    const f32x4 b_as = wb.vec4(L.as2.b * 4, 0), b_res = wb.vec4(L.res.b * 4, 0);
    const float inv_as = wb.scalar((L.as2.b + 16) * 4), inv_res = wb.scalar((L.res.b + 16) * 4);
    ResidentA A;
    f32x16 accX[2][2], accY[2][2];
    float nx[2][3], nv[2][3];              // next tile's positions / view directions of this thread's rows
    auto fetch_points = [&](int tile) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int gp = tile * kPtsP + 64 * h + (tid & 63);
            gp = gp < p.n_points ? gp : p.n_points - 1;
            gp = gp < 0 ? 0 : gp;
            const int ray = gp / p.n_samples;
            const float* __restrict__ r = p.rays + (size_t)ray * INERF_RAY_FLOATS;
            const float zz = __builtin_nontemporal_load(p.z + gp);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float x = __fadd_rn(r[c], __fmul_rn(r[3 + c], zz));                              // run_nerf.py:488
                if (kSsr && p.xyz_div != 1.0f) x = __fdiv_rn(x, p.xyz_div);                     // semantic_nerf.py:64
                nx[h][c] = x;
                nv[h][c] = r[8 + c];
            }
        }
    };
    fetch_points(blockIdx.x);

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        // ---------------- encode -> hi/lo planes (xyz: columns 0..63, dir: columns 256..287), 64 rows per pass ----------------
        // Sample positions / view directions of this thread's two rows (row tid & 63 of each half), fetched at the END of the previous
        // tile (or before the loop) so that no encode starts by waiting for HBM; the positions stay in registers for the skip layer's
        // second encoding pass.
        float px[2][3], pv[2][3];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 3; ++c) { px[h][c] = nx[h][c]; pv[h][c] = nv[h][c]; }
        auto encode = [&](int half, bool with_dir) {
            const int pt = 64 * half + (tid & 63), part = tid >> 6;
            _Float16* row = ldsp + pt * kRowD;
            const float* x = px[half];
            const float* r8 = pv[half];
            for (int f = part; f < p.l_xyz; f += kParts) {
                const float s = (float)(1 << f);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sn, cs;
                    fast_sincosf(x[c] * s, &sn, &cs);
                    split_store<kPlaneP>(row + 3 + 6 * f + c, sn, amax);
                    split_store<kPlaneP>(row + 6 + 6 * f + c, cs, amax);
                }
            }
            if (part == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) split_store<kPlaneP>(row + c, x[c], amax);
                for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) { row[c] = (_Float16)0.0f; row[kPlaneP + c] = (_Float16)0.0f; }
            }
            if (with_dir) {
                const int fd = kParts - 1 - part;
                if (fd < p.l_dir) {
                    const float s = (float)(1 << fd);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float sn, cs;
                        fast_sincosf(r8[c] * s, &sn, &cs);
                        split_store<kPlaneP>(row + kColDirD + 3 + 6 * fd + c, sn, amax);
                        split_store<kPlaneP>(row + kColDirD + 6 + 6 * fd + c, cs, amax);
                    }
                }
                if (part == 3) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) split_store<kPlaneP>(row + kColDirD + c, r8[c], amax);
                    for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) { row[kColDirD + c] = (_Float16)0.0f; row[kPlaneP + kColDirD + c] = (_Float16)0.0f; }
                }
            }
        };
        // this tile's weights: W(0) -> slots 0..3, W(1)[4..15] -> slots 4..15 (requested here: the encodings below cover the latency,
        // and nothing else is waiting behind these 64 loads)
        pipe_load(A, wb, frag256(L.trunk[0], 4), 0, 4, 0);
        pipe_load(A, wb, frag256(L.trunk[1], 16), 4, 12, 4);
        encode(0, true);
        encode(1, true);
        { stamp(); PIPE_SYNC(); stamp(); }

        // an exposed epilogue (layers whose two halves finish together): acc -> rows at dl, after a barrier
        f32x4 bias2[2][4];
        auto store_exposed = [&](const f32x16 (&acc)[2][2], const GemmSlot& s, int d_hi, int d_lo) {
            float inv;
            load_bias<2>(bias2, inv, wb, bias256(s), (s.b + kWidth) * 4, lane);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float t[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) t[i] = fmaxf(__builtin_fmaf(acc[rb][pb][4 * g + i], inv, bias2[rb][g][i]), 0.0f);
                        f16x2 h01, h23, l01, l23;
                        split_pair(t[0], t[1], h01, l01);
                        split_pair(t[2], t[3], h23, l23);
                        amax2 = __builtin_elementwise_max(amax2, __builtin_elementwise_max(h01, h23));
                        const int doff = pb * 32 * kRowD + 32 * rb + 8 * g;
                        *reinterpret_cast<f16x4*>(ldsp + d_hi + doff) = f16x4{h01[0], h01[1], h23[0], h23[1]};
                        *reinterpret_cast<f16x4*>(ldsp + d_lo + doff) = f16x4{l01[0], l01[1], l23[0], l23[1]};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };

        // ---------------- layer 0: both halves in one pass ----------------
        pipe_both4<true>(A, ldsp, xo, accX, accY);
        pipe_load(A, wb, frag256(L.trunk[1], 16), 0, 4, 0);               // W(1)[0..3] -> slots 0..3
        float inv_prev = scale256(L.trunk[0]);                             // output factor of the layer whose half B sits in accY
        prime(bias256(L.trunk[0]));
        { stamp(); PIPE_SYNC(); stamp(); }                                                   // every wave has read the encodings
        store_exposed(accX, L.trunk[0], dO[0][0], dO[0][1]);
        { stamp(); PIPE_SYNC(); stamp(); }
        // ---------------- trunk layers 1..4 ----------------
        const GemmSlot* prev_slot = &L.trunk[0];                           // layer whose half B sits in accY
#pragma unroll 1
        for (int layer = 1; layer < kSkipInput; ++layer) {
            const GemmSlot& s = L.trunk[layer];
            const float inv_cur = scale256(s);
            pipe_phase<16, 2, true, 0>(A, wb, ldsp, xo[0][0], xo[0][1], accX, accY, dO[1][0], dO[1][1], bring, bias256(*prev_slot), bias256(s),
                                       inv_prev, true, amax2, lane, 0);
            { stamp(); PIPE_SYNC(); stamp(); }
            const int next = layer + 1 < kSkipInput ? frag256(L.trunk[layer + 1], 16) : frag256(L.trunk[kSkipInput], 20) + 4 * 4096;
            pipe_phase<16, 2, true, 2>(A, wb, ldsp, xo[1][0], xo[1][1], accY, accX, dO[0][0], dO[0][1], bring, bias256(s), bias256(s),
                                       inv_cur, true, amax2, lane, next);
            { stamp(); PIPE_SYNC(); stamp(); }
            prev_slot = &s;
            inv_prev = inv_cur;
        }
        // ---------------- skip layer: pts_linears[5] over cat([pts, h]) ----------------
        {
            const GemmSlot& s = L.trunk[kSkipInput];
            const float inv_cur = scale256(s);
            pipe_phase<16, 2, true, 0>(A, wb, ldsp, xo[0][0], xo[0][1], accX, accY, dO[1][0], dO[1][1], bring, bias256(*prev_slot), 0,
                                       inv_prev, true, amax2, lane, 0);
            { stamp(); PIPE_SYNC(); stamp(); }
            // h-part of half B; slots 4..15 <- W(6)[4..15], slots 0..3 <- the encoding part W(5)[0..3]
            pipe_phase<16, 2, false, 3>(A, wb, ldsp, xo[1][0], xo[1][1], accY, accX, 0, 0, bring, 0, bias256(s), 0.0f, true, amax2, lane,
                                        frag256(L.trunk[6], 16), frag256(s, 20));
            { stamp(); PIPE_SYNC(); stamp(); }                                               // every wave has read h4
            encode(0, false);
            encode(1, false);
            { stamp(); PIPE_SYNC(); stamp(); }
            pipe_both4<false>(A, ldsp, xo, accX, accY);
            pipe_load(A, wb, frag256(L.trunk[6], 16), 0, 4, 0);            // W(6)[0..3] -> slots 0..3
            { stamp(); PIPE_SYNC(); stamp(); }
            store_exposed(accX, s, dO[0][0], dO[0][1]);
            { stamp(); PIPE_SYNC(); stamp(); }
            prev_slot = &s;
            inv_prev = inv_cur;
        }
        // ---------------- trunk layers 6, 7 ----------------
#pragma unroll 1
        for (int layer = kSkipInput + 1; layer < kDepth; ++layer) {
            const GemmSlot& s = L.trunk[layer];
            const float inv_cur = scale256(s);
            pipe_phase<16, 2, true, 0>(A, wb, ldsp, xo[0][0], xo[0][1], accX, accY, dO[1][0], dO[1][1], bring, bias256(*prev_slot), bias256(s),
                                       inv_prev, true, amax2, lane, 0);
            { stamp(); PIPE_SYNC(); stamp(); }
            const int next = layer + 1 < kDepth ? frag256(L.trunk[layer + 1], 16) : frag256(L.as1, 16);
            pipe_phase<16, 2, true, 2>(A, wb, ldsp, xo[1][0], xo[1][1], accY, accX, dO[0][0], dO[0][1], bring, bias256(s), bias256(s),
                                       inv_cur, true, amax2, lane, next);
            { stamp(); PIPE_SYNC(); stamp(); }
            prev_slot = &s;
            inv_prev = inv_cur;
        }
        // ---------------- heads ----------------
        // albedo | shading hidden layer of half A, while h7 of half B is retired
        const float inv_feat = scale256(L.feat);
        pipe_phase<16, 2, true, 0>(A, wb, ldsp, xo[0][0], xo[0][1], accX, accY, dO[1][0], dO[1][1], bring, bias256(*prev_slot), 0,
                                   inv_prev, true, amax2, lane, 0);
        { stamp(); PIPE_SYNC(); stamp(); }                                                   // h7 is complete for all 128 rows
        // Partial sums of the albedo|shading head (4 floats per point and wave) wait for the residual head in the 12 KiB of LDS
        // behind the two planes: park[point][wave][4]; the residual head's partials go into the rows of their half once those
        // are dead.  Nothing of the heads is held in registers across the remaining phases.
        float* const park = reinterpret_cast<float*>(ldsp + 2 * kPlaneP);
        float sig[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const _Float16* xs = ldsp + (64 * h + 16 * wave + (lane & 15)) * kRowD + 8 * (lane >> 4);
            sig[h] = skinny_gemm_h<8, kPlaneP>(wb, L.alpha.w * 4, L.alpha.b * 4, (L.alpha.b + 16) * 4, xs, lane)[0];
            if (kSsr && L.sem_rbs > 0) {
                const int my_pt = tile * kPts + 64 * h + 16 * wave + (lane & 15);
                const bool valid = my_pt < p.n_points;
                sem_head<false, kPlaneP>(wb, L, xs, lane, amax2, p.raw + (size_t)(valid ? my_pt : 0) * p.channels, valid, p.n_classes, nullptr);
            }
        }
        auto park_as = [&](int h, const f32x4 (&part)[2]) {
            if (lane < 32) {
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) *reinterpret_cast<f32x4*>(park + ((64 * h + 32 * pb + lane) * 4 + wave) * 4) = part[pb];
            }
        };
        auto exchange_res = [&](int h, const f32x4 (&part)[2]) {          // rows of half h are dead: columns 64.. of the hi plane, [wave][4]
            if (lane < 32) {
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ldsp + (64 * h + 32 * pb + lane) * kRowD + kColExD) + 4 * wave) = part[pb];
            }
        };
        {
            const float inv = scale256(L.as1);
            f32x4 part[2];
            regop_head<2>(wb, (L.as2r.w + wave * 4 * 2 * 256) * 4, accX, inv, bias256(L.as1), lane, amax2, part);
            park_as(0, part);
            pipe_phase<16, 2, false, 2>(A, wb, ldsp, xo[1][0], xo[1][1], accY, accX, 0, 0, bring, 0, 0, 0.0f, true, amax2, lane, frag256(L.feat, 16));
            regop_head<2>(wb, (L.as2r.w + wave * 4 * 2 * 256) * 4, accY, inv, bias256(L.as1), lane, amax2, part);
            park_as(1, part);
        }
        // feature (no activation) in place of h7, then the view-dependent layer over [feature | dir]
        pipe_phase<16, 2, false, 0>(A, wb, ldsp, xo[0][0], xo[0][1], accX, accY, 0, 0, bring, 0, bias256(L.feat), 0.0f, true, amax2, lane, 0);
        { stamp(); PIPE_SYNC(); stamp(); }                                 // every wave has read h7 of half A (GEMMs and heads)
        pipe_phase<16, 2, true, 1>(A, wb, ldsp, xo[1][0], xo[1][1], accY, accX, dO[0][0], dO[0][1], bring, bias256(L.feat), bias256(L.feat),
                                   inv_feat, false, amax2, lane, frag128(L.views, 18));
        { stamp(); PIPE_SYNC(); stamp(); }
        pipe_phase<18, 1, true, 0>(A, wb, ldsp, xo[0][0], xo[0][1], accX, accY, dO[1][0], dO[1][1], bring, bias256(L.feat), 0,
                                   inv_feat, false, amax2, lane, 0);
        { stamp(); PIPE_SYNC(); stamp(); }                                 // rows A are dead from here on
        {
            const float inv1 = wb.scalar((L.views.b + kHalf) * 4);
            f32x4 part[2];
            regop_head<1>(wb, (L.resr.w + wave * 2 * 2 * 256) * 4, accX, inv1, (L.views.b + 32 * wave) * 4, lane, amax2, part);
            exchange_res(0, part);
            pipe_phase<18, 1, false, 0>(A, wb, ldsp, xo[1][0], xo[1][1], accY, accX, 0, 0, bring, 0, 0, 0.0f, true, amax2, lane, 0);
            if (tile + (int)gridDim.x < p.n_tiles) fetch_points(tile + gridDim.x);     // the next tile's rays and depths
            regop_head<1>(wb, (L.resr.w + wave * 2 * 2 * 256) * 4, accY, inv1, (L.views.b + 32 * wave) * 4, lane, amax2, part);
            { stamp(); PIPE_SYNC(); stamp(); }                             // every wave has read rows B
            exchange_res(1, part);
        }
        { stamp(); PIPE_SYNC(); stamp(); }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int my_pt = tile * kPts + 64 * h + 16 * wave + (lane & 15);
            if (lane < 16 && my_pt < p.n_points) {
                const float* exa = park + (64 * h + 16 * wave + lane) * 16;
                const float* exr = reinterpret_cast<const float*>(ldsp + (64 * h + 16 * wave + lane) * kRowD + kColExD);
                f32x4 as4 = {0.0f, 0.0f, 0.0f, 0.0f}, res4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    as4 += *reinterpret_cast<const f32x4*>(exa + 4 * w);
                    res4 += *reinterpret_cast<const f32x4*>(exr + 4 * w);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    as4[i] = __builtin_fmaf(as4[i], inv_as, b_as[i]);
                    res4[i] = __builtin_fmaf(res4[i], inv_res, b_res[i]);
                }
                const float a0 = sigmoid_ref_h(as4[0]), a1 = sigmoid_ref_h(as4[1]), a2 = sigmoid_ref_h(as4[2]);
                const float sh = sigmoid_ref_h(as4[3]);
                const float r0 = sigmoid_ref_h(res4[0]), r1 = sigmoid_ref_h(res4[1]), r2 = sigmoid_ref_h(res4[2]);
                float* o = p.raw + (size_t)my_pt * p.channels;
                __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a0, sh), r0), o + 0);          // run_nerf_helpers.py:320
                __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a1, sh), r1), o + 1);
                __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a2, sh), r2), o + 2);
                __builtin_nontemporal_store(sig[h], o + 3);
                __builtin_nontemporal_store(a0, o + 4); __builtin_nontemporal_store(a1, o + 5); __builtin_nontemporal_store(a2, o + 6);
                __builtin_nontemporal_store(sh, o + 7);
                __builtin_nontemporal_store(r0, o + 8); __builtin_nontemporal_store(r1, o + 9); __builtin_nontemporal_store(r2, o + 10);
            }
        }
        stamp();
        dbg_n = 64;                                // first tile only
    }
    const float amax_all = fmaxf(amax, fmaxf((float)amax2[0], (float)amax2[1]));
    if (p.status && __any(!(amax_all <= kF16Safe)) && lane == 0) atomicOr(p.status, INERF_STATUS_F16_RANGE);
}

int launch_pipe(MlpParams& p, int64_t n_points, bool ssr, hipStream_t stream) {
    p.n_tiles = (int)((n_points + kPtsP - 1) / kPtsP);
    const int grid = p.n_tiles < device_cus() ? p.n_tiles : device_cus();
    void (*kern)(const MlpParams) = ssr ? k_encode_mlp_f16x3_pipe<true> : k_encode_mlp_f16x3_pipe<false>;
    static PerDeviceOnce attr_set[2];
    if (attr_set[ssr].first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesP);
        if (e != hipSuccess) return record(e);
        attr_set[ssr].mark();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLdsBytesP, stream, p);
    return record(hipGetLastError());
}

}  // namespace inerf
