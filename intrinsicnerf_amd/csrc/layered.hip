// layered.hip - the networks the fused kernels do not implement, layer by layer on the fp32 matrix core.
//
// The reference builds NeRF(D=args.netdepth, W=args.netwidth, skips=..., use_viewdirs=...) from command-line flags
// (object_level/run_nerf.py:286-296, 545-552; SSR/training/trainer.py:811-846) and trains it in fp32 throughout
// (run_nerf.py:942-1018).  The fused kernels (mlp*.hip) implement the shape every shipped config uses (D=8, W=256, skips=[4]); this
// file is the general form: every nn.Linear of NeRF.forward / Semantic_NeRF.forward (run_nerf_helpers.py:284-321,
// semantic_nerf.py:120-181) is one launch of k_linear_f32 - exact fp32 products and accumulation on v_mfma_f32_32x32x2_f32 -
// with the activations in HBM, and what loss.backward() records for those layers is the same kernel on other strides:
//
//     forward        Y[p, o]  = act(sum_k X[p, k] W[o, k] + b[o])          A = X (k contiguous), B = W  (k contiguous)
//     input gradient dX[p, k] = (sum_o dZ[p, o] W[o, k]) [+ add] [gated]   A = dZ (k contiguous), B = W  (row index contiguous)
//     weight gradient dW[o, k] = sum_p dZ[p, o] X[p, k];  db[o] = sum_p dZ[p, o]
//                                                                          A = dZ, B = X (both row-index contiguous), split over p
//
// torch.cat([input_pts, h]) (the skip) and torch.cat([feature, input_views]) never happen: the producing launches write into column
// ranges of one wider buffer (c + offset, c_ld).  It also evaluates a training batch whose activations leave the f16 range of the
// split-precision kernels (object_level.render_rays): no ATen GEMM is left on the path.
//
// Tiling: 256 threads = 4 waves, one workgroup = 128 x 128 outputs (each wave 64 x 64 = 2 x 2 MFMA blocks, 64 accumulator
// registers), 128 x 32 or 32 x 128 for the narrow heads; 16 k per step through LDS in [k][row] order (17 KB: four workgroups per CU
// overlap each other's loads; two LDS stages, one barrier per step), the next step's operands in flight in registers while this one is multiplied.  Bound: fp32 MFMA
// (157.3 TFLOP/s); arithmetic intensity of a 256 x 256 layer = 2*256*256 FLOP per 2 KB of activations = 64 FLOP/B.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/inerf.h"
#include "mlp_common.h"

namespace inerf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LinParams {
    const float* a; long long a_sm, a_sk;
    const float* b; long long b_sn, b_sk;
    const float* bias; const float* add; long long add_ld; const float* gate; long long gate_ld;
    float* c; long long c_ld;
    long long m, k, k_per;        // k_per: reduction range of one blockIdx.z (a multiple of 16); partial z goes to c + z * m * n
    int n, act, a_kc, b_kc;
};

__device__ __forceinline__ float sigmoid_exact(float x) { return __fdiv_rn(1.0f, 1.0f + expf(-x)); }   // torch.sigmoid (mlp.hip sigmoid_ref)

constexpr int kKS = 16;

template <int WM, int WN, int BM, int BN>
__global__ __launch_bounds__(256, 3) void k_linear_f32(const LinParams p) {
    constexpr int TM = WM * BM * 32, TN = WN * BN * 32;
    constexpr int LDA = TM + 4, LDB = TN + 4;          // [k][row] with 4 floats of padding: the transposing stores hit 64 banks
    constexpr int NA = TM * kKS / 256, NB = TN * kKS / 256;
    __shared__ float sA[2 * kKS * LDA];               // two stages: step s + 1 is written while step s is multiplied - ONE barrier per step
    __shared__ float sB[2 * kKS * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long long m0 = (long long)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    const long long kbeg = (long long)blockIdx.z * p.k_per;
    const long long kend = kbeg + p.k_per < p.k ? kbeg + p.k_per : p.k;

    // global -> registers -> LDS.  Element i of this thread's share of an operand tile: (row, k) = (e / 16, e % 16) of e = tid + 256 i when k is
    // the operand's unit stride, (e % T, e / T) when the row index is - either way its address is ONE per-thread base + i * a wave-uniform
    // step, and so is its LDS slot: no 64-bit multiply inside the k loop.
    constexpr int RA = 256 / kKS, RB = 256 / kKS;          // rows between consecutive elements, k-contiguous operand
    constexpr int KA = 256 / TM, KB = 256 / TN;             // k between consecutive elements, row-contiguous operand
    const int a_m = p.a_kc ? tid / kKS : tid % TM, a_k = p.a_kc ? tid % kKS : tid / TM;
    const int b_n = p.b_kc ? tid / kKS : tid % TN, b_k = p.b_kc ? tid % kKS : tid / TN;
    const float* pa = p.a + (m0 + a_m) * p.a_sm + (kbeg + a_k) * p.a_sk;
    const float* pb = p.b + (long long)(n0 + b_n) * p.b_sn + (kbeg + b_k) * p.b_sk;
    const long long a_step = p.a_kc ? RA * p.a_sm : KA * p.a_sk, b_step = p.b_kc ? RB * p.b_sn : KB * p.b_sk;
    const long long a_adv = kKS * p.a_sk, b_adv = kKS * p.b_sk;
    const int a_rows_left = (int)((p.m - m0 - a_m < (1 << 30)) ? p.m - m0 - a_m : (1 << 30));       // rows of the operand from this thread's first one
    const int b_rows_left = p.n - n0 - b_n;
    float* const sa = sA + (p.a_kc ? a_k * LDA + a_m : a_k * LDA + a_m);
    float* const sb = sB + (p.b_kc ? b_k * LDB + b_n : b_k * LDB + b_n);
    const int sa_step = p.a_kc ? RA : KA * LDA, sb_step = p.b_kc ? RB : KB * LDB;

    // Edge handling without branches (a conditional load per element is a basic block of its own: the sixteen requests of a step were issued one
    // after the other, each behind its own wait).  Interior steps - whole tile inside the matrix, sixteen k left - load unconditionally;
    // the others load from an address CLAMPED into the operand (this thread's first element, or its last valid k) and select 0 afterwards.
    const bool a_inside = m0 + TM <= p.m, b_inside = n0 + TN <= p.n;
    if (a_rows_left <= 0) pa = p.a + m0 * p.a_sm + (kbeg + a_k) * p.a_sk;                  // (row m0 exists: a valid address to clamp to)
    if (b_rows_left <= 0) pb = p.b + (long long)n0 * p.b_sn + (kbeg + b_k) * p.b_sk;
    const int a_rmul = p.a_kc ? RA : 0, a_kmul = p.a_kc ? 0 : KA;        // element i: row + i * rmul, k + i * kmul (a thread with no row: rows_left <= 0)
    const int b_rmul = p.b_kc ? RB : 0, b_kmul = p.b_kc ? 0 : KB;
    float ra[NA], rb[NB];
    auto fetch = [&](int k_left /* kend - k0 */) {
        if (a_inside && k_left >= kKS) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = pa[i * a_step];
        } else {
            const long long kfix = (long long)((k_left - 1 < a_k ? k_left - 1 : a_k) - a_k) * p.a_sk;      // back to the last valid k (<= 0)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                // (plain integer arithmetic: `a && b` and `mode ? x : y` inside this loop become a branch per element)
                const int ok = (int)(i * a_rmul < a_rows_left) & (int)(a_k + i * a_kmul < k_left);
                // (an AND with a mask, not `ok ? v : 0`: that form the compiler turns back into a load under a branch)
                ra[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, pa[ok ? i * a_step : kfix]) & (unsigned)(-ok));
            }
        }
        if (b_inside && k_left >= kKS) {
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = pb[i * b_step];
        } else {
            const long long kfix = (long long)((k_left - 1 < b_k ? k_left - 1 : b_k) - b_k) * p.b_sk;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int ok = (int)(i * b_rmul < b_rows_left) & (int)(b_k + i * b_kmul < k_left);
                rb[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, pb[ok ? i * b_step : kfix]) & (unsigned)(-ok));
            }
        }
        pa += a_adv;
        pb += b_adv;
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) sa[buf * kKS * LDA + i * sa_step] = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) sb[buf * kKS * LDB + i * sb_step] = rb[i];
    };

    f32x16 acc[BM][BN];
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int j = 0; j < BN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const float* la = sA + (lane >> 5) * LDA + wm * BM * 32 + (lane & 31);
    const float* lb = sB + (lane >> 5) * LDB + wn * BN * 32 + (lane & 31);
    auto left = [&](long long k0) { return (int)(kend - k0 < (1 << 30) ? kend - k0 : (1 << 30)); };
    if (kbeg < kend) {
        fetch(left(kbeg));
        stage(0);
    }
    __syncthreads();
    int buf = 0;
    for (long long k0 = kbeg; k0 < kend; k0 += kKS, buf ^= 1) {
        const bool more = k0 + kKS < kend;
        if (more) fetch(left(k0 + kKS));                     // in flight while this step is multiplied
        const float* xa = la + buf * kKS * LDA;
        const float* xb = lb + buf * kKS * LDB;
        float av[2][BM], bv[2][BN];                          // operands of k-pair kp + 1 are read while k-pair kp is multiplied
#pragma unroll
        for (int i = 0; i < BM; ++i) av[0][i] = xa[i * 32];
#pragma unroll
        for (int j = 0; j < BN; ++j) bv[0][j] = xb[j * 32];
#pragma unroll
        for (int kp = 0; kp < kKS / 2; ++kp) {
            if (kp + 1 < kKS / 2) {
#pragma unroll
                for (int i = 0; i < BM; ++i) av[(kp + 1) & 1][i] = xa[2 * (kp + 1) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < BN; ++j) bv[(kp + 1) & 1][j] = xb[2 * (kp + 1) * LDB + j * 32];
            }
#pragma unroll
            for (int i = 0; i < BM; ++i)
#pragma unroll
                for (int j = 0; j < BN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp & 1][i], bv[kp & 1][j], acc[i][j], 0, 0, 0);
            // keep the order written above: the next pair's LDS reads go out in front of this pair's MFMAs (left alone the scheduler sinks
            // them behind the MFMAs and every group of four waits for its own operands)
            __builtin_amdgcn_sched_group_barrier(0x100, BM + BN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, BM * BN, 0);
        }
        if (more) stage(buf ^ 1);                            // (its last readers passed the previous step's barrier)
        __syncthreads();
    }

    // epilogue: accumulator register r of lane l = row 8 (r / 4) + 4 (l >> 5) + r % 4, column l & 31.  Tiles inside the matrix (all but the last
    // row / column of tiles) take the branch-free form: a bounds check per element is a basic block per element, and the `add` / `gate`
    // loads of the backward launches were then issued one at a time, each behind its own wait.
    float* c = p.c + (long long)blockIdx.z * p.m * p.n;
    const bool inside = m0 + TM <= p.m && n0 + TN <= p.n;
#pragma unroll
    for (int j = 0; j < BN; ++j) {
        const int gn = n0 + (wn * BN + j) * 32 + (lane & 31);
        if (!inside && gn >= p.n) continue;
        const float bias = p.bias ? p.bias[gn] : 0.0f;
#pragma unroll
        for (int i = 0; i < BM; ++i) {
            const long long gm0 = m0 + (wm * BM + i) * 32 + 4 * (lane >> 5);
            auto row = [&](int r) { return gm0 + 8 * (r >> 2) + (r & 3); };
            auto finish = [&](float v, float ad, float gt) {
                if (p.bias) v = __fadd_rn(v, bias);
                if (p.add) v = __fadd_rn(v, ad);
                if (p.act == INERF_ACT_RELU) v = v < 0.0f ? 0.0f : v;             // NaN stays NaN, like F.relu
                else if (p.act == INERF_ACT_SIGMOID) v = sigmoid_exact(v);
                if (p.gate) v = gt > 0.0f ? v : 0.0f;                              // ReLU backward: grad * (output > 0)
                return v;
            };
            if (inside) {
                float ad[16], gt[16];
                if (p.add) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) ad[r] = p.add[row(r) * p.add_ld + gn];
                }
                if (p.gate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) gt[r] = p.gate[row(r) * p.gate_ld + gn];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) c[row(r) * p.c_ld + gn] = finish(acc[i][j][r], p.add ? ad[r] : 0.0f, p.gate ? gt[r] : 1.0f);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long gm = row(r);
                    if (gm >= p.m) continue;
                    c[gm * p.c_ld + gn] = finish(acc[i][j][r], p.add ? p.add[gm * p.add_ld + gn] : 0.0f, p.gate ? p.gate[gm * p.gate_ld + gn] : 1.0f);
                }
            }
        }
    }
}

template <int WM, int WN, int BM, int BN>
int launch_tile(const LinParams& p, int splits, hipStream_t s) {
    constexpr int TM = WM * BM * 32, TN = WN * BN * 32;
    const long long gx = (p.m + TM - 1) / TM;
    if (gx > 0x7fffffffLL) return INERF_E_INVALID;
    dim3 grid((unsigned)gx, (unsigned)((p.n + TN - 1) / TN), (unsigned)splits);
    hipLaunchKernelGGL((k_linear_f32<WM, WN, BM, BN>), grid, dim3(256), 0, s, p);
    return record(hipGetLastError());
}

// the tile of an m x n product and how many workgroups of it a CU holds (registers: launch bounds; LDS: 34 / 26 / 21 KB)
struct TileChoice { int tm, tn, per_cu; };
inline TileChoice tile_for(long long m, int n) {
    if (n <= 32) return {128, 32, 6};
    if (m <= 32) return {32, 128, 6};
    if (m <= 64 && n <= 64) return {64, 64, 6};
    if (n <= 64) return {128, 64, 4};
    return {128, 128, 3};
}
int launch_linear(const LinParams& p, int splits, hipStream_t s) {
    if (p.n <= 32) return launch_tile<4, 1, 1, 1>(p, splits, s);
    if (p.m <= 32) return launch_tile<1, 4, 1, 1>(p, splits, s);
    if (p.m <= 64 && p.n <= 64) return launch_tile<2, 2, 1, 1>(p, splits, s);
    if (p.n <= 64) return launch_tile<4, 1, 1, 2>(p, splits, s);
    return launch_tile<2, 2, 2, 2>(p, splits, s);
}

// ---- weight-gradient partials -> dW [rows, cols]: 32 elements per workgroup, eight threads per element; thread q adds the splits z = q, q + 8, ...
// in ascending order, the eight sums meet in LDS and are added in the order q = 0..7: a fixed order, bit-identical run to run ----
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int splits, long long total, float* __restrict__ dst, int accumulate) {
    __shared__ float sh[256];
    const int el = threadIdx.x & 31, q = threadIdx.x >> 5;
    const long long e = (long long)blockIdx.x * 32 + el;
    float s = 0.0f;
    if (e < total)
        for (int z = q; z < splits; z += 8) s = __fadd_rn(s, part[z * total + e]);
    sh[threadIdx.x] = s;
    __syncthreads();
    if (q == 0 && e < total) {
        float t = sh[el];
#pragma unroll
        for (int k = 1; k < 8; ++k) t = __fadd_rn(t, sh[32 * k + el]);
        dst[e] = accumulate ? __fadd_rn(dst[e], t) : t;
    }
}

// ---- bias gradient: column sums of g [n_points, rows] - blockIdx.x = a range of points, thread = column; partials [ranges][rows] go through
// k_wgrad_reduce like the weight partials.  (As a ones column of X inside the product it cost a whole extra column of 128-wide tiles whenever
// the layer's input width is a multiple of 128.)
constexpr int kColsumPoints = 512;
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ g, long long ldg, int rows, long long n_points, float* __restrict__ part) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= rows) return;
    const long long p0 = (long long)blockIdx.x * kColsumPoints;
    const long long p1 = p0 + kColsumPoints < n_points ? p0 + kColsumPoints : n_points;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    long long q = p0;
    for (; q + 4 <= p1; q += 4) {
        s0 = __fadd_rn(s0, g[q * ldg + c]);
        s1 = __fadd_rn(s1, g[(q + 1) * ldg + c]);
        s2 = __fadd_rn(s2, g[(q + 2) * ldg + c]);
        s3 = __fadd_rn(s3, g[(q + 3) * ldg + c]);
    }
    for (; q < p1; ++q) s0 = __fadd_rn(s0, g[q * ldg + c]);
    part[(long long)blockIdx.x * rows + c] = __fadd_rn(__fadd_rn(s0, s1), __fadd_rn(s2, s3));
}

// ---- frequency encoding (run_nerf_helpers.py:195-225; semantic_nerf.py:50-66 divides by scalar_factor first) ----
// one thread per (point, band): band 0 = the value itself, band f + 1 = sin / cos of value * 2^f
__global__ __launch_bounds__(256) void k_embed(const float* __restrict__ rays, const float* __restrict__ z, long long n_points, int n_samples,
                                               int n_freqs, float div, int dir, float* __restrict__ dst, long long ld) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int bands = n_freqs + 1;
    if (e >= n_points * bands) return;
    const long long pt = e / bands;
    const int band = (int)(e % bands);
    const float* r = rays + (pt / n_samples) * INERF_RAY_FLOATS;
    float x[3];
    if (dir) {
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = r[8 + c];
    } else {
        const float zz = z[pt];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = __fadd_rn(r[c], __fmul_rn(r[3 + c], zz));                    // run_nerf.py:488
            if (div != 1.0f) x[c] = __fdiv_rn(x[c], div);
        }
    }
    float* row = dst + pt * ld;
    if (band == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) row[c] = x[c];
        return;
    }
    const float s = (float)(1 << (band - 1));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sn, cs;
        sincosf(__fmul_rn(x[c], s), &sn, &cs);
        row[3 + 6 * (band - 1) + c] = sn;
        row[6 + 6 * (band - 1) + c] = cs;
    }
}

// ---- rgb = albedo * shading + residual (run_nerf_helpers.py:319) on the raw rows, and what autograd records for it + the three sigmoids
__global__ __launch_bounds__(256) void k_intrinsic_combine(float* __restrict__ raw, long long ld, long long n) {
    const long long pt = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pt >= n) return;
    float* r = raw + pt * ld;
    const float sh = r[7];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = __fadd_rn(__fmul_rn(r[4 + c], sh), r[8 + c]);
}

__global__ __launch_bounds__(256) void k_intrinsic_combine_bwd(const float* __restrict__ raw, const float* __restrict__ d_raw, long long ld,
                                                               long long n, float* __restrict__ dz /* [n, 8]: albedo3 shading residual3 0 */) {
    const long long pt = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pt >= n) return;
    const float* r = raw + pt * ld;
    const float* g = d_raw + pt * ld;
    const float sh = r[7];
    float d_sh = g[7];
    float* o = dz + pt * 8;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float alb = r[4 + c], res = r[8 + c];
        const float d_alb = __fadd_rn(g[4 + c], __fmul_rn(g[c], sh));
        d_sh = __fadd_rn(d_sh, __fmul_rn(g[c], alb));
        const float d_res = __fadd_rn(g[8 + c], g[c]);
        o[c] = __fmul_rn(__fmul_rn(d_alb, 1.0f - alb), alb);                   // sigmoid_backward: grad * (1 - y) * y
        o[4 + c] = __fmul_rn(__fmul_rn(d_res, 1.0f - res), res);
    }
    o[3] = __fmul_rn(__fmul_rn(d_sh, 1.0f - sh), sh);
    o[7] = 0.0f;
}

inline int64_t wgrad_splits(int64_t n_points, int rows, int cols) {
    // ONE round of workgroups: tiles x splits = the workgroups the chip holds at once (a second, partly filled round costs a whole round)
    const TileChoice t = tile_for(rows, cols);
    const int64_t tiles = (int64_t)((rows + t.tm - 1) / t.tm) * ((cols + t.tn - 1) / t.tn);
    int64_t want = (int64_t)t.per_cu * device_cus() / tiles;
    const int64_t most = (n_points + 255) / 256;                              // at least 256 points per split
    if (want > most) want = most;
    if (want < 1) want = 1;
    return want;
}
inline int64_t colsum_ranges(int64_t n_points) { return (n_points + kColsumPoints - 1) / kColsumPoints; }
inline int64_t align64(int64_t floats) { return (floats + 63) / 64 * 64; }
inline int64_t wgrad_k_per(int64_t n_points, int64_t splits) { return ((n_points + splits - 1) / splits + kKS - 1) / kKS * kKS; }

}  // namespace
}  // namespace inerf

using namespace inerf;

extern "C" int inerf_linear(const inerf_linear_args* a, void* stream) {
    if (!a || a->m < 0 || a->n < 1 || a->k < 0) return INERF_E_INVALID;
    if (a->m == 0) return INERF_OK;
    if (!a->a || !a->b || !a->c) return INERF_E_INVALID;
    if (a->act < INERF_ACT_NONE || a->act > INERF_ACT_SIGMOID) return INERF_E_INVALID;
    if ((a->a_sk != 1 && a->a_sm != 1) || (a->b_sk != 1 && a->b_sn != 1)) return INERF_E_UNSUPPORTED;   // one unit stride per operand
    LinParams p{};
    p.a = a->a; p.a_sm = a->a_sm; p.a_sk = a->a_sk;
    p.b = a->b; p.b_sn = a->b_sn; p.b_sk = a->b_sk;
    p.bias = a->bias; p.add = a->add; p.add_ld = a->add_ld; p.gate = a->gate; p.gate_ld = a->gate_ld;
    p.c = a->c; p.c_ld = a->c_ld;
    p.m = a->m; p.n = a->n; p.k = a->k;
    p.k_per = (a->k + kKS - 1) / kKS * kKS;
    if (p.k_per == 0) p.k_per = kKS;
    p.act = a->act;
    p.a_kc = a->a_sk == 1; p.b_kc = a->b_sk == 1;
    return launch_linear(p, 1, static_cast<hipStream_t>(stream));
}

extern "C" int64_t inerf_linear_wgrad_workspace_bytes(int64_t n_points, int rows, int cols) {
    if (n_points < 0 || rows < 1 || cols < 1) return INERF_E_INVALID;
    return (align64(wgrad_splits(n_points, rows, cols) * rows * (int64_t)cols) + colsum_ranges(n_points) * rows) * 4;
}

extern "C" int inerf_linear_wgrad(const float* g, int64_t ldg, int rows, const float* x, int64_t ldx, int cols, int64_t n_points,
                                  float* d_weight, float* d_bias, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!g || !x || !d_weight || rows < 1 || cols < 1 || n_points < 0 || ldg < rows || ldx < cols) return INERF_E_INVALID;
    const int64_t splits = wgrad_splits(n_points, rows, cols);
    const int64_t part_floats = align64(splits * rows * (int64_t)cols), ranges = colsum_ranges(n_points);
    if (!workspace || workspace_bytes < (part_floats + ranges * rows) * 4) return INERF_E_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    LinParams p{};
    p.a = g; p.a_sm = 1; p.a_sk = ldg;            // A(row o, k = point) = dZ[point, o]
    p.b = x; p.b_sn = 1; p.b_sk = ldx;            // B(column j, k = point) = X[point, j]
    p.c = part; p.c_ld = cols;
    p.m = rows; p.n = cols; p.k = n_points;
    p.k_per = wgrad_k_per(n_points, splits);
    if (p.k_per == 0) p.k_per = kKS;
    p.a_kc = 0; p.b_kc = 0;
    int rc = launch_linear(p, (int)splits, s);
    if (rc) return rc;
    const long long total = (long long)rows * cols;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, part, (int)splits, total, d_weight, accumulate);
    if (d_bias) {
        if (ranges > 0)
            hipLaunchKernelGGL(k_colsum, dim3((unsigned)ranges, (unsigned)((rows + 255) / 256)), dim3(256), 0, s, g, (long long)ldg, rows,
                               (long long)n_points, part + part_floats);
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, s, part + part_floats, (int)ranges, (long long)rows, d_bias,
                           accumulate);
    }
    return record(hipGetLastError());
}

extern "C" int inerf_embed(const float* rays, const float* z_vals, int64_t n_rays, int n_samples, int n_freqs, float divisor, int directions,
                           float* out, int64_t ld, void* stream) {
    if (n_rays < 0 || n_samples < 1 || n_freqs < 0 || n_freqs > 30 || ld < 3 + 6 * n_freqs) return INERF_E_INVALID;
    if (n_rays == 0) return INERF_OK;
    if (!rays || !out || (!directions && !z_vals) || !(divisor != 0.0f)) return INERF_E_INVALID;
    const long long n = n_rays * (long long)n_samples, total = n * (n_freqs + 1);
    if ((total + 255) / 256 > 0x7fffffffLL) return INERF_E_INVALID;
    hipLaunchKernelGGL(k_embed, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rays, z_vals, n, n_samples,
                       n_freqs, divisor, directions, out, (long long)ld);
    return record(hipGetLastError());
}

extern "C" int inerf_intrinsic_combine(float* raw, int64_t ld, int64_t n_points, void* stream) {
    if (n_points < 0 || ld < INERF_BASE_CHANNELS) return INERF_E_INVALID;
    if (n_points == 0) return INERF_OK;
    if (!raw) return INERF_E_INVALID;
    hipLaunchKernelGGL(k_intrinsic_combine, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), raw, (long long)ld,
                       (long long)n_points);
    return record(hipGetLastError());
}

extern "C" int inerf_intrinsic_combine_backward(const float* raw, const float* d_raw, int64_t ld, int64_t n_points, float* dz, void* stream) {
    if (n_points < 0 || ld < INERF_BASE_CHANNELS) return INERF_E_INVALID;
    if (n_points == 0) return INERF_OK;
    if (!raw || !d_raw || !dz) return INERF_E_INVALID;
    hipLaunchKernelGGL(k_intrinsic_combine_bwd, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), raw, d_raw,
                       (long long)ld, (long long)n_points, dz);
    return record(hipGetLastError());
}
