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
// overlap each other's loads), the next step's operands in flight in registers while this one is multiplied.  Bound: fp32 MFMA
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
    int n, act, b_ones, a_kc, b_kc;
};

__device__ __forceinline__ float sigmoid_exact(float x) { return __fdiv_rn(1.0f, 1.0f + expf(-x)); }   // torch.sigmoid (mlp.hip sigmoid_ref)

constexpr int kKS = 16;

template <int WM, int WN, int BM, int BN>
__global__ __launch_bounds__(256) void k_linear_f32(const LinParams p) {
    constexpr int TM = WM * BM * 32, TN = WN * BN * 32;
    constexpr int LDA = TM + 4, LDB = TN + 4;          // [k][row] with 4 floats of padding: the transposing stores hit 64 banks
    constexpr int NA = TM * kKS / 256, NB = TN * kKS / 256;
    __shared__ float sA[kKS * LDA];
    __shared__ float sB[kKS * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long long m0 = (long long)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    const long long kbeg = (long long)blockIdx.z * p.k_per;
    const long long kend = kbeg + p.k_per < p.k ? kbeg + p.k_per : p.k;

    float ra[NA], rb[NB];
    auto fetch = [&](long long k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + i * 256;
            const int kk = p.a_kc ? e % kKS : e / TM, mm = p.a_kc ? e / kKS : e % TM;
            const long long gm = m0 + mm, gk = k0 + kk;
            ra[i] = (gm < p.m && gk < kend) ? p.a[gm * p.a_sm + gk * p.a_sk] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + i * 256;
            const int kk = p.b_kc ? e % kKS : e / TN, nn = p.b_kc ? e / kKS : e % TN;
            const long long gk = k0 + kk;
            const int gn = n0 + nn;
            float v = 0.0f;
            if (gn < p.n && gk < kend) v = (p.b_ones && gn == p.n - 1) ? 1.0f : p.b[(long long)gn * p.b_sn + gk * p.b_sk];
            rb[i] = v;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + i * 256;
            const int kk = p.a_kc ? e % kKS : e / TM, mm = p.a_kc ? e / kKS : e % TM;
            sA[kk * LDA + mm] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + i * 256;
            const int kk = p.b_kc ? e % kKS : e / TN, nn = p.b_kc ? e / kKS : e % TN;
            sB[kk * LDB + nn] = rb[i];
        }
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
    if (kbeg < kend) fetch(kbeg);
    for (long long k0 = kbeg; k0 < kend; k0 += kKS) {
        stage();
        __syncthreads();
        if (k0 + kKS < kend) fetch(k0 + kKS);
#pragma unroll
        for (int kp = 0; kp < kKS / 2; ++kp) {
            float av[BM], bv[BN];
#pragma unroll
            for (int i = 0; i < BM; ++i) av[i] = la[2 * kp * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < BN; ++j) bv[j] = lb[2 * kp * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < BM; ++i)
#pragma unroll
                for (int j = 0; j < BN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: accumulator register r of lane l = row 8 (r / 4) + 4 (l >> 5) + r % 4, column l & 31
    float* c = p.c + (long long)blockIdx.z * p.m * p.n;
#pragma unroll
    for (int j = 0; j < BN; ++j) {
        const int gn = n0 + (wn * BN + j) * 32 + (lane & 31);
        if (gn >= p.n) continue;
        const float bias = p.bias ? p.bias[gn] : 0.0f;
#pragma unroll
        for (int i = 0; i < BM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long gm = m0 + (wm * BM + i) * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (gm >= p.m) continue;
                float v = acc[i][j][r];
                if (p.bias) v = __fadd_rn(v, bias);
                if (p.add) v = __fadd_rn(v, p.add[gm * p.add_ld + gn]);
                if (p.act == INERF_ACT_RELU) v = v < 0.0f ? 0.0f : v;             // NaN stays NaN, like F.relu
                else if (p.act == INERF_ACT_SIGMOID) v = sigmoid_exact(v);
                if (p.gate) v = p.gate[gm * p.gate_ld + gn] > 0.0f ? v : 0.0f;     // ReLU backward: grad * (output > 0)
                c[gm * p.c_ld + gn] = v;
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

int launch_linear(const LinParams& p, int splits, hipStream_t s) {
    if (p.n <= 32) return launch_tile<4, 1, 1, 1>(p, splits, s);
    if (p.m <= 32) return launch_tile<1, 4, 1, 1>(p, splits, s);
    return launch_tile<2, 2, 2, 2>(p, splits, s);
}

// ---- weight-gradient partials -> dW [rows, cols], db [rows]; one thread per element, splits added in ascending order ----
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int splits, int rows, int ncol /* cols + 1 */,
                                                      float* __restrict__ dw, float* __restrict__ db, int accumulate) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)rows * ncol;
    if (e >= total) return;
    float s = 0.0f;
    for (int z = 0; z < splits; ++z) s = __fadd_rn(s, part[z * total + e]);
    const int r = (int)(e / ncol), cix = (int)(e % ncol);
    float* dst = cix == ncol - 1 ? (db ? db + r : nullptr) : dw + (long long)r * (ncol - 1) + cix;
    if (dst) *dst = accumulate ? __fadd_rn(*dst, s) : s;
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

inline int64_t wgrad_splits(int64_t n_points, int rows, int ncol) {
    // the tile launch_linear picks for a rows x ncol product
    const int tm = ncol <= 32 ? 128 : rows <= 32 ? 32 : 128, tn = ncol <= 32 ? 32 : 128;
    const int64_t tiles = (int64_t)((rows + tm - 1) / tm) * ((ncol + tn - 1) / tn);
    int64_t want = (4 * (int64_t)device_cus() + tiles - 1) / tiles;          // ~four workgroups per CU in flight
    const int64_t most = (n_points + 255) / 256;                              // at least 256 points per split
    if (want > most) want = most;
    if (want < 1) want = 1;
    return want;
}
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
    p.act = a->act; p.b_ones = 0;
    p.a_kc = a->a_sk == 1; p.b_kc = a->b_sk == 1;
    return launch_linear(p, 1, static_cast<hipStream_t>(stream));
}

extern "C" int64_t inerf_linear_wgrad_workspace_bytes(int64_t n_points, int rows, int cols) {
    if (n_points < 0 || rows < 1 || cols < 1) return INERF_E_INVALID;
    return wgrad_splits(n_points, rows, cols + 1) * rows * (int64_t)(cols + 1) * 4;
}

extern "C" int inerf_linear_wgrad(const float* g, int64_t ldg, int rows, const float* x, int64_t ldx, int cols, int64_t n_points,
                                  float* d_weight, float* d_bias, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!g || !x || !d_weight || rows < 1 || cols < 1 || n_points < 0 || ldg < rows || ldx < cols) return INERF_E_INVALID;
    const int ncol = cols + 1;
    const int64_t splits = wgrad_splits(n_points, rows, ncol);
    if (!workspace || workspace_bytes < splits * rows * (int64_t)ncol * 4) return INERF_E_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    LinParams p{};
    p.a = g; p.a_sm = 1; p.a_sk = ldg;            // A(row o, k = point) = dZ[point, o]
    p.b = x; p.b_sn = 1; p.b_sk = ldx;            // B(column j, k = point) = X[point, j]; column `cols` reads 1: the bias gradient
    p.c = static_cast<float*>(workspace); p.c_ld = ncol;
    p.m = rows; p.n = ncol; p.k = n_points;
    p.k_per = wgrad_k_per(n_points, splits);
    if (p.k_per == 0) p.k_per = kKS;
    p.b_ones = 1; p.a_kc = 0; p.b_kc = 0;
    int rc = launch_linear(p, (int)splits, s);
    if (rc) return rc;
    const long long total = (long long)rows * ncol;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, static_cast<const float*>(workspace), (int)splits,
                       rows, ncol, d_weight, d_bias, accumulate);
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
