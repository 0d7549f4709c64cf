// mlp.hip - fused frequency encoding + intrinsic-NeRF MLP for gfx950 (MI355X).
//
// Replaces, per sample point, the whole ATen sequence of
//   run_network  (object_level/run_nerf.py:42-56   | SSR/models/model_utils.py:19-35)
//   Embedder     (run_nerf_helpers.py:195-243      | SSR/models/semantic_nerf.py:14-65)
//   NeRF.forward (run_nerf_helpers.py:284-321)     | Semantic_NeRF.forward (semantic_nerf.py:123-181)
// without ever materialising the 90-wide embedding or any 256-wide activation in HBM.
//
// Design (DESIGN.md has the long form):
//   * one workgroup (4 waves, one per SIMD) owns a tile of 64 consecutive sample points and walks
//     the tile through all layers; activations X[point][channel] stay in LDS (layout.h);
//   * every layer is computed transposed, D[channel][point] = sum_k W[channel][k] * X[point][k], on
//     the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32): A = weights, B = activations.  Each wave
//     owns 64 (or 32) output channels for all 64 points, so a lane ends up holding 4 consecutive
//     channels of one point and writes them back with one ds_write_b128;
//   * weights are NOT staged in LDS: the host packer stores them in MFMA-fragment order, so each
//     wave streams its own fragments from L2 with fully coalesced 1 KiB global_load_dwordx4, one
//     k-block (16 MFMAs = 1024 cycles) ahead of use.  No barrier inside a layer;
//   * the 1..4-row output heads (sigma, albedo/shading out, residual, semantic logits) run on
//     v_mfma_f32_16x16x4_f32 with the four waves splitting the points, so no wave idles.
// All arithmetic is fp32; compiled with -ffp-contract=off so that o + d*z and albedo*shading +
// residual round exactly like the reference's separate mul/add.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mlp_common.h"

namespace inerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// wide GEMM: RB blocks of 32 output channels per wave x PB blocks of 32 points, K = 8 * (kb0 + kb1)
// ------------------------------------------------------------------------------------------------
// Operands of the first two k-blocks and the bias are fetched by wide_prefetch(), which the caller
// issues BEFORE the previous layer's epilogue + barrier, so a layer never starts with an exposed L2
// round trip.  Inside the loop the weight fragments run two k-blocks ahead and the LDS activation
// fragments one k-block ahead of the MFMAs that consume them.
template <int RB>
struct WidePre {
    f32x4 w0[RB], w1[RB];   // weight fragments of k-block 0 and 1
    f32x4 b[RB][4];         // bias for this lane's 4-channel groups
};

template <int RB>
__device__ __forceinline__ void wide_prefetch(WidePre<RB>& pre, const float* __restrict__ wfrag,
                                              const float* __restrict__ bias, int lane) {
    const f32x4* wv = reinterpret_cast<const f32x4*>(wfrag) + lane;
    const int h4 = 4 * (lane >> 5);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        pre.w0[rb] = wv[rb * 64];
        pre.w1[rb] = wv[(RB + rb) * 64];
#pragma unroll
        for (int g = 0; g < 4; ++g) pre.b[rb][g] = *reinterpret_cast<const f32x4*>(bias + 32 * rb + 8 * g + h4);
    }
}

template <int RB, int PB>
__device__ __forceinline__ void wide_gemm(const WidePre<RB>& pre, const float* __restrict__ wfrag,
                                          const float* xl,                  // lds + (lane&31)*stride + 4*(lane>>5)
                                          int col0, int kb0, int col1, int kb1, int lane,
                                          f32x16 (&acc)[RB][PB]) {
    // accumulators start at the bias: channel of register r is 32*rb + (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc[rb][pb][4 * g + i] = pre.b[rb][g][i];
    const f32x4* wv = reinterpret_cast<const f32x4*>(wfrag) + lane;
    const int kbt = kb0 + kb1;
    auto xoff = [&](int kb) { return kb < kb0 ? col0 + 8 * kb : col1 + 8 * (kb - kb0); };
    // 4 rotating weight buffers (distance 2) and 2 activation buffers (distance 1), indexed by compile-time
    // constants inside a 4x unrolled body, so no in-flight load is ever copied (a copy would force a wait).
    // Every K here is a multiple of 32, i.e. kbt is a multiple of 4.
    f32x4 w[4][RB], x[2][PB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { w[0][rb] = pre.w0[rb]; w[1][rb] = pre.w1[rb]; }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) x[0][pb] = *reinterpret_cast<const f32x4*>(xl + xoff(0) + pb * 32 * kLdsStride);
#pragma unroll 1
    for (int kb = 0; kb < kbt; kb += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // issue the loads for later k-blocks first ...
            const int k1 = kb + i + 1 < kbt ? kb + i + 1 : kbt - 1;
            const int k2 = kb + i + 2 < kbt ? kb + i + 2 : kbt - 1;
            const int xo = xoff(k1);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) w[(i + 2) & 3][rb] = wv[(k2 * RB + rb) * 64];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
                x[(i + 1) & 1][pb] = *reinterpret_cast<const f32x4*>(xl + xo + pb * 32 * kLdsStride);
            __builtin_amdgcn_sched_barrier(0);
            // ... then 4*RB*PB MFMAs on operands requested two (weights) / one (activations) steps ago
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        acc[rb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i & 3][rb][c], x[i & 1][pb][c], acc[rb][pb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// write the wave's accumulators to LDS activation columns [dcol + 32*RB*wave, ...) (optionally ReLU)
template <int RB, int PB>
__device__ __forceinline__ void wide_store(const f32x16 (&acc)[RB][PB],
                                           float* dl /* lds + (lane&31)*stride + 4*(lane>>5) + dcol + chan0 */, bool relu) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float a = acc[rb][pb][4 * g + i];
                    v[i] = relu ? fmaxf(a, 0.0f) : a;
                }
                *reinterpret_cast<f32x4*>(dl + pb * 32 * kLdsStride + 32 * rb + 8 * g) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM: 16 output rows x this wave's 16 points, K = 16*KB16; result row = 4*(lane>>4) + i
// ------------------------------------------------------------------------------------------------
template <int KB16>
__device__ __forceinline__ f32x4 skinny_gemm(const float* __restrict__ wfrag, const float* __restrict__ bias16,
                                             const float* xs /* lds + (16*wave + (lane&15))*stride + col + 4*(lane>>4) */,
                                             int lane) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(bias16 + 4 * (lane >> 4));
    f32x4 a1 = {0.0f, 0.0f, 0.0f, 0.0f};
    const f32x4* wv = reinterpret_cast<const f32x4*>(wfrag) + lane;
#pragma unroll 2
    for (int kb = 0; kb < KB16; kb += 2) {
        const f32x4 w0 = wv[kb * 64], w1 = wv[(kb + 1) * 64];
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + 16 * kb);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs + 16 * kb + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[c], x0[c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[c], x1[c], a1, 0, 0, 0);
        }
    }
    return a0 + a1;
}

__device__ __forceinline__ float sigmoid_ref(float x) {
    // torch.sigmoid (F.sigmoid at run_nerf_helpers.py:299,305,316): 1 / (1 + exp(-x)), IEEE division
    return __fdiv_rn(1.0f, 1.0f + expf(-x));
}

// ------------------------------------------------------------------------------------------------
// the kernel.  PB = 32-point blocks per tile:
//   PB = 2: 64-point tile, 153 KiB of LDS, one workgroup per CU (1 wave per SIMD);
//   PB = 1: 32-point tile,  76.5 KiB of LDS, TWO workgroups per CU (2 waves per SIMD): while one
//           workgroup sits in a barrier / epilogue / encode phase the other one keeps the matrix
//           pipe busy.  Costs 2x the L2->CU weight stream (8.6 TB/s chip-wide, 25 % of L2 bandwidth).
// ------------------------------------------------------------------------------------------------
template <bool kSsr, int PB>
__global__ __launch_bounds__(256, 3 - PB) void k_encode_mlp(const MlpParams p) {
    constexpr int kPts = 32 * PB;
    constexpr int kParts = 256 / kPts;            // encode: thread = (point, part)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ wts = p.wts;
    const NetLayout& L = p.L;

    // per-lane LDS bases
    float* const xw = lds + (lane & 31) * kLdsStride + 4 * (lane >> 5);               // wide operand / result rows
    const bool skinny_wave = 16 * wave < kPts;                                         // PB = 1: waves 0,1 only
    const float* const xs = lds + (16 * (skinny_wave ? wave : 0) + (lane & 15)) * kLdsStride + 4 * (lane >> 4);

    auto frag256 = [&](const GemmSlot& s, int kbt) { return wts + s.w + (size_t)wave * kbt * 2 * 256; };
    auto frag128 = [&](const GemmSlot& s, int kbt) { return wts + s.w + (size_t)wave * kbt * 256; };

    WidePre<2> pre2;
    WidePre<1> pre1;
    wide_prefetch<2>(pre2, frag256(L.trunk[0], 8), wts + L.trunk[0].b + 64 * wave, lane);

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        // ---------------- encode: X[:, enc | dir] ----------------
        {
            const int pt = tid % kPts, part = tid / kPts;
            int gp = tile * kPts + pt;
            gp = gp < p.n_points ? gp : p.n_points - 1;
            const int ray = gp / p.n_samples;
            const float* __restrict__ r = p.rays + (size_t)ray * INERF_RAY_FLOATS;
            const float zz = __builtin_nontemporal_load(p.z + gp);     // streamed once: keep it out of the L2 the weights live in
            float* row = lds + pt * kLdsStride;
            float x[3], v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // pts = rays_o + rays_d * z  (run_nerf.py:488): separate multiply and add
                x[c] = __fadd_rn(r[c], __fmul_rn(r[3 + c], zz));
                if (p.xyz_div != 1.0f) x[c] = __fdiv_rn(x[c], p.xyz_div);   // semantic_nerf.py:64
                v[c] = r[8 + c];
            }
            // frequency bands are spread over the parts; 2^f scaling is exact (run_nerf_helpers.py:212)
            for (int f = part; f < p.l_xyz; f += kParts) {
                const float s = (float)(1 << f);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sn, cs;
                    sincosf(x[c] * s, &sn, &cs);
                    row[kColEnc + 3 + 6 * f + c] = sn;
                    row[kColEnc + 6 + 6 * f + c] = cs;
                }
            }
            const int fd = kParts - 1 - part;                                   // direction bands: one per part, from the top
            if (fd < p.l_dir) {
                const float s = (float)(1 << fd);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sn, cs;
                    sincosf(v[c] * s, &sn, &cs);
                    row[kColDir + 3 + 6 * fd + c] = sn;
                    row[kColDir + 6 + 6 * fd + c] = cs;
                }
            }
            if (part == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) row[kColEnc + c] = x[c];
                for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) row[kColEnc + c] = 0.0f;
            }
            if (part == 3) {
#pragma unroll
                for (int c = 0; c < 3; ++c) row[kColDir + c] = v[c];
                for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) row[kColDir + c] = 0.0f;
            }
        }
        __syncthreads();

        // ---------------- trunk: 8 x (Linear + ReLU), ping-pong A/B ----------------
        // every step: GEMM on prefetched operands -> prefetch the NEXT step's first operands -> epilogue -> barrier
        auto step256 = [&](const GemmSlot& s, int c0, int kb0, int c1, int kb1, int dcol, bool relu, auto&& prefetch_next) {
            f32x16 acc[2][PB];
            wide_gemm<2, PB>(pre2, frag256(s, kb0 + kb1), xw, c0, kb0, c1, kb1, lane, acc);
            prefetch_next();
            wide_store<2, PB>(acc, xw + dcol + 64 * wave, relu);
            __syncthreads();
        };
        auto step128 = [&](const GemmSlot& s, int c0, int kb0, int c1, int kb1, int dcol, bool relu, auto&& prefetch_next) {
            f32x16 acc[1][PB];
            wide_gemm<1, PB>(pre1, frag128(s, kb0 + kb1), xw, c0, kb0, c1, kb1, lane, acc);
            prefetch_next();
            wide_store<1, PB>(acc, xw + dcol + 32 * wave, relu);
            __syncthreads();
        };
        auto pf256 = [&](const GemmSlot& s, int kbt) {
            return [&, kbt]() { wide_prefetch<2>(pre2, frag256(s, kbt), wts + s.b + 64 * wave, lane); };
        };
        auto pf128 = [&](const GemmSlot& s, int kbt) {
            return [&, kbt]() { wide_prefetch<1>(pre1, frag128(s, kbt), wts + s.b + 32 * wave, lane); };
        };
        const bool sem = kSsr && L.sem_rbs > 0;
        step256(L.trunk[0], kColEnc, 8, 0, 0, kColA, true, pf256(L.trunk[1], 32));
        step256(L.trunk[1], kColA, 32, 0, 0, kColB, true, pf256(L.trunk[2], 32));
        step256(L.trunk[2], kColB, 32, 0, 0, kColA, true, pf256(L.trunk[3], 32));
        step256(L.trunk[3], kColA, 32, 0, 0, kColB, true, pf256(L.trunk[4], 32));
        step256(L.trunk[4], kColB, 32, 0, 0, kColA, true, pf256(L.trunk[5], 40));
        step256(L.trunk[5], kColEnc, 8, kColA, 32, kColB, true, pf256(L.trunk[6], 32));   // cat([pts, h]) (run_nerf_helpers.py:290-291)
        step256(L.trunk[6], kColB, 32, 0, 0, kColA, true, pf256(L.trunk[7], 32));
        if (sem) step256(L.trunk[7], kColA, 32, 0, 0, kColB, true, pf128(L.sem1, 32));      // h7 in B
        else     step256(L.trunk[7], kColA, 32, 0, 0, kColB, true, pf256(L.as1, 32));

        // ---------------- heads ----------------
        const int my_pt = tile * kPts + 16 * wave + (lane & 15);     // the point this lane reports in skinny results
        const bool my_valid = skinny_wave && my_pt < p.n_points;
        float* const out_row = p.raw + (size_t)(my_valid ? my_pt : 0) * p.channels;

        // sigma = alpha_linear(h7)  (run_nerf_helpers.py:294) - no activation here, ReLU happens in raw2outputs
        f32x4 sig4 = {0.f, 0.f, 0.f, 0.f};
        if (skinny_wave) sig4 = skinny_gemm<16>(wts + L.alpha.w, wts + L.alpha.b, xs + kColB, lane);

        if (sem) {
            // semantic head: Linear(256,128)+ReLU then Linear(128,C), logits raw (semantic_nerf.py:110,142)
            step128(L.sem1, kColB, 32, 0, 0, kColA, true, pf256(L.as1, 32));
            if (skinny_wave) {
                for (int rb = 0; rb < L.sem_rbs; ++rb) {
                    const f32x4 lg = skinny_gemm<8>(wts + L.sem2.w + rb * 8 * 256, wts + L.sem2.b + 16 * rb, xs + kColA, lane);
                    const int ch0 = 16 * rb + 4 * (lane >> 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (my_valid && ch0 + i < p.n_classes) __builtin_nontemporal_store(lg[i], out_row + INERF_BASE_CHANNELS + ch0 + i);
                }
            }
            __syncthreads();                                  // A is about to be overwritten
        }

        // albedo / shading hidden layers (one 256-row GEMM), then their 3+1 outputs (one skinny GEMM)
        step256(L.as1, kColB, 32, 0, 0, kColA, true, pf256(L.feat, 32));
        f32x4 as4 = {0.f, 0.f, 0.f, 0.f};
        if (skinny_wave) as4 = skinny_gemm<16>(wts + L.as2.w, wts + L.as2.b, xs + kColA, lane);
        __syncthreads();                                      // A is about to be overwritten

        // feature = feature_linear(h7) (no activation), views layer over cat([feature, dirs]), residual head
        step256(L.feat, kColB, 32, 0, 0, kColA, false, pf128(L.views, 36));
        step128(L.views, kColA, 32, kColDir, 4, kColB, true, pf256(L.trunk[0], 8));          // next tile's first layer
        f32x4 res4 = {0.f, 0.f, 0.f, 0.f};
        if (skinny_wave) res4 = skinny_gemm<8>(wts + L.res.w, wts + L.res.b, xs + kColB, lane);

        if (lane < 16 && my_valid) {
            const float a0 = sigmoid_ref(as4[0]), a1 = sigmoid_ref(as4[1]), a2 = sigmoid_ref(as4[2]);
            const float sh = sigmoid_ref(as4[3]);
            const float r0 = sigmoid_ref(res4[0]), r1 = sigmoid_ref(res4[1]), r2 = sigmoid_ref(res4[2]);
            // rgb = albedo * shading + residual (run_nerf_helpers.py:320): multiply, then add
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a0, sh), r0), out_row + 0);
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a1, sh), r1), out_row + 1);
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a2, sh), r2), out_row + 2);
            __builtin_nontemporal_store(sig4[0], out_row + 3);
            __builtin_nontemporal_store(a0, out_row + 4); __builtin_nontemporal_store(a1, out_row + 5); __builtin_nontemporal_store(a2, out_row + 6);
            __builtin_nontemporal_store(sh, out_row + 7);
            __builtin_nontemporal_store(r0, out_row + 8); __builtin_nontemporal_store(r1, out_row + 9); __builtin_nontemporal_store(r2, out_row + 10);
        }
        if (kSsr && p.endpoint) {
            // show_endpoint: append the post-ReLU views activation (semantic_nerf.py:163-164,181)
            const int col = tid & 127;
            const int base = INERF_BASE_CHANNELS + p.n_classes;
            for (int pt = tid >> 7; pt < kPts; pt += 2) {
                const int gp = tile * kPts + pt;
                if (gp < p.n_points) p.raw[(size_t)gp * p.channels + base + col] = lds[pt * kLdsStride + kColB + col];
            }
        }
        // No barrier needed here: the next tile's encode writes only the enc/dir columns, whose last
        // readers (trunk[5], views) finished before barriers every wave has already passed; buffer B
        // (still being read by slower waves' residual GEMM) is first rewritten two barriers later.
    }
}

static int g_num_cus[256] = {0};          // per device id; 0 = not queried yet
static int g_last_hip_error = 0;

int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 ? dev : 0;
}

int device_cus() {
    const int dev = current_device() & 255;
    int n = __atomic_load_n(&g_num_cus[dev], __ATOMIC_RELAXED);
    if (n == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
        n = cus;
        __atomic_store_n(&g_num_cus[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

// Staggered start of the chain's workgroups (mlp_common.h stagger_start): four units of 2 048 cycles per step of the
// eight-step pattern, none for launches with fewer than four tiles per workgroup.
int stagger_units(int n_tiles, int grid) { return n_tiles < 4 * grid ? 0 : 4; }

// Tile shape of the exact-fp32 kernel: 64-point tiles, one workgroup per CU (129.2 TFLOP/s).  (A 32-point-tile form with two
// workgroups per CU measured 124.8 - the second workgroup hides barriers and epilogues but doubles the weight stream and halves
// the skinny-GEMM parallelism; its template instance is no longer selectable at run time.)
int tile_blocks() { return 2; }

int record(hipError_t e) {
    if (e == hipSuccess) return INERF_OK;
    g_last_hip_error = (int)e;
    return INERF_E_HIP;
}

}  // namespace inerf

extern "C" int inerf_last_hip_error(void) { return inerf::g_last_hip_error; }

namespace inerf {

int launch_mlp_f32(MlpParams& p, int64_t n_points, bool ssr, hipStream_t stream) {
    const int pb = tile_blocks();
    const int tile_pts = 32 * pb;
    p.n_tiles = (int)((n_points + tile_pts - 1) / tile_pts);
    const int lds_bytes = tile_pts * kLdsStride * 4;
    const int max_grid = device_cus() * (3 - pb);                 // PB=1: two workgroups per CU
    const int grid = p.n_tiles < max_grid ? p.n_tiles : max_grid;
    void (*kern)(const MlpParams) = pb == 2 ? (ssr ? k_encode_mlp<true, 2> : k_encode_mlp<false, 2>)
                                            : (ssr ? k_encode_mlp<true, 1> : k_encode_mlp<false, 1>);
    static PerDeviceOnce attr_set[2][2];
    if (attr_set[ssr][pb - 1].first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           lds_bytes);
        if (e != hipSuccess) return record(e);
        attr_set[ssr][pb - 1].mark();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, stream, p);
    return record(hipGetLastError());
}

}  // namespace inerf

static int encode_mlp_impl(const inerf_net_desc* net, const float* packed, const float* rays, const float* z, int64_t n_rays,
                           int n_samples, uint32_t flags, float* raw_out, float* save, float* act_max, int32_t* status, void* stream,
                           float* sem_scratch = nullptr, int64_t status_rays = 0);

extern "C" int64_t inerf_encode_mlp_workspace_bytes(const inerf_net_desc* net, int64_t n_rays, int n_samples, uint32_t flags) {
    if (!net || !inerf::net_supported(*net) || n_rays < 0 || n_samples < 1) return INERF_E_INVALID;
    return inerf::sem_scratch_bytes(*net, n_rays * (int64_t)n_samples, (flags & INERF_FLAG_ENDPOINT) != 0);
}

extern "C" int inerf_encode_mlp_ws(const inerf_net_desc* net, const float* packed, const float* rays, const float* z, int64_t n_rays,
                                   int n_samples, uint32_t flags, float* raw_out, int32_t* status, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    if (!net || !inerf::net_supported(*net) || n_rays < 0 || n_samples < 1) return INERF_E_INVALID;
    const int64_t need = inerf::sem_scratch_bytes(*net, n_rays * (int64_t)n_samples, (flags & INERF_FLAG_ENDPOINT) != 0);
    if (need > 0 && (!workspace || workspace_bytes < need)) return INERF_E_WORKSPACE;
    return encode_mlp_impl(net, packed, rays, z, n_rays, n_samples, flags, raw_out, nullptr, nullptr, status, stream,
                           need > 0 ? static_cast<float*>(workspace) : nullptr);
}

extern "C" int inerf_encode_mlp_chunked(const inerf_net_desc* net, const float* packed, const float* rays, const float* z, int64_t n_rays,
                                        int n_samples, uint32_t flags, float* raw_out, int32_t* status, int64_t status_rays, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
    if (!net || !inerf::net_supported(*net) || n_rays < 0 || n_samples < 1) return INERF_E_INVALID;
    const int64_t need = inerf::sem_scratch_bytes(*net, n_rays * (int64_t)n_samples, (flags & INERF_FLAG_ENDPOINT) != 0);
    if (need > 0 && (!workspace || workspace_bytes < need)) return INERF_E_WORKSPACE;
    return encode_mlp_impl(net, packed, rays, z, n_rays, n_samples, flags, raw_out, nullptr, nullptr, status, stream,
                           need > 0 ? static_cast<float*>(workspace) : nullptr, status_rays);
}

extern "C" int inerf_encode_mlp(const inerf_net_desc* net, const float* packed, const float* rays, const float* z,
                                int64_t n_rays, int n_samples, uint32_t flags, float* raw_out, int32_t* status,
                                void* stream) {
    return encode_mlp_impl(net, packed, rays, z, n_rays, n_samples, flags, raw_out, nullptr, nullptr, status, stream);
}

extern "C" int64_t inerf_mlp_save_floats(const inerf_net_desc* net, int64_t n_points) {
    if (!net || !inerf::net_supported(*net) || n_points < 0) return INERF_E_INVALID;
    return inerf::save_total_floats(*net, n_points);
}

extern "C" int inerf_mlp_save_slot(const inerf_net_desc* net, int slot, int64_t n_points, int64_t* offset_floats, int* width) {
    if (!net || !inerf::net_supported(*net) || slot < 0 || slot >= inerf::SAVE_SLOTS || n_points < 0) return INERF_E_INVALID;
    if (offset_floats) *offset_floats = inerf::save_offset(*net, slot, n_points);
    if (width) *width = inerf::save_width(*net, slot);
    return INERF_OK;
}

extern "C" int inerf_mlp_save_slot_is_fragment(int slot, int gradient) {
    if (slot < 0 || slot >= inerf::SAVE_SLOTS) return INERF_E_INVALID;
    return inerf::save_is_frag(slot, gradient != 0);
}

extern "C" int inerf_encode_mlp_train(const inerf_net_desc* net, const float* packed, const float* rays, const float* z,
                                      int64_t n_rays, int n_samples, uint32_t flags, float* raw_out, float* save_out,
                                      float* act_max, int32_t* status, void* stream) {
    if (net && n_rays == 0) return INERF_OK;
    if (!save_out || !net) return INERF_E_INVALID;
    if (net->precision != INERF_PREC_F16X3) return INERF_E_UNSUPPORTED;
    if (n_rays * (int64_t)n_samples > inerf::kMaxTrainPoints) return INERF_E_UNSUPPORTED;     // one activation slot stays below 4 GiB (buffer descriptors)
    return encode_mlp_impl(net, packed, rays, z, n_rays, n_samples, flags, raw_out, save_out, act_max, status, stream);
}

static int encode_mlp_impl(const inerf_net_desc* net, const float* packed, const float* rays, const float* z, int64_t n_rays,
                           int n_samples, uint32_t flags, float* raw_out, float* save, float* act_max, int32_t* status, void* stream,
                           float* sem_scratch, int64_t status_rays) {
    using namespace inerf;
    if (net && n_rays == 0) return net_supported(*net) ? INERF_OK : INERF_E_UNSUPPORTED;      // empty batch: pointers may be null
    if (!net || !packed || !rays || !z || !raw_out || n_rays < 0 || n_samples < 1) return INERF_E_INVALID;
    if (!net_supported(*net)) return INERF_E_UNSUPPORTED;
    if (n_rays == 0) return INERF_OK;
    const int64_t n_points = n_rays * (int64_t)n_samples;
    if (n_points >= (int64_t)1 << 31) return INERF_E_UNSUPPORTED;     // caller chunks (the front-ends do)
    const bool ssr = net->variant == INERF_VARIANT_SSR;
    MlpParams p;
    p.wts = packed; p.rays = rays; p.z = z; p.raw = raw_out; p.status = status;
    p.status_rays = (status && status_rays > 0 && status_rays < n_rays) ? (int)status_rays : 0;      // (>= n_rays: everything in word 0)
    p.save = save;
    p.act_max = act_max;
    p.sem_scratch = sem_scratch;
    for (int s = 0; s < SAVE_SLOTS; ++s) p.save_off[s] = save_offset(*net, s, n_points);
    p.bits_off = relu_bits_offset(*net, n_points);
    p.L = make_layout(*net);
    p.n_points = (int)n_points;
    p.n_samples = n_samples;
    p.n_tiles = 0;
    p.endpoint = (ssr && (flags & INERF_FLAG_ENDPOINT)) ? 1 : 0;
    p.n_classes = ssr ? net->n_classes : 0;
    p.channels = INERF_BASE_CHANNELS + p.n_classes + (p.endpoint ? INERF_ENDPOINT_DIM : 0);
    p.l_xyz = net->l_xyz; p.l_dir = net->l_dir; p.xyz_div = net->xyz_div;
    return net->precision == INERF_PREC_F16X3 ? launch_mlp_f16x3(p, n_points, ssr, (hipStream_t)stream)
                                              : launch_mlp_f32(p, n_points, ssr, (hipStream_t)stream);
}
