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

#include "layout.h"

namespace inerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MlpParams {
    const float* wts;       // packed blob
    const float* rays;      // [N,11]
    const float* z;         // [N,S]
    float* raw;             // [N*S, channels]
    NetLayout L;
    int n_points;           // N*S  (< 2^31, checked on the host)
    int n_samples;
    int n_tiles;
    int channels;
    int n_classes;
    int endpoint;
    int l_xyz, l_dir;
    float xyz_div;
};

// ------------------------------------------------------------------------------------------------
// wide GEMM: RB blocks of 32 output channels per wave, 64 points, K = 8 * (kb0 + kb1)
// ------------------------------------------------------------------------------------------------
template <int RB>
__device__ __forceinline__ void wide_gemm(const float* __restrict__ wfrag,  // wave's fragment stream
                                          const float* __restrict__ bias,   // + first channel of this wave
                                          const float* xl,                  // lds + (lane&31)*stride + 4*(lane>>5)
                                          int col0, int kb0, int col1, int kb1, int lane,
                                          f32x16 (&acc)[RB][2]) {
    const int h4 = 4 * (lane >> 5);
    // accumulators start at the bias: channel of register r is 32*rb + (r&3) + 8*(r>>2) + h4
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * rb + 8 * g + h4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[rb][0][4 * g + i] = b[i];
                acc[rb][1][4 * g + i] = b[i];
            }
        }
    }
    const f32x4* wv = reinterpret_cast<const f32x4*>(wfrag) + lane;
    const int kbt = kb0 + kb1;
    f32x4 wn[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) wn[rb] = wv[rb * 64];
#pragma unroll 2
    for (int kb = 0; kb < kbt; ++kb) {
        f32x4 wc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) wc[rb] = wn[rb];
        const int kn = kb + 1 < kbt ? kb + 1 : kb;            // prefetch the next k-block's fragments
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) wn[rb] = wv[(kn * RB + rb) * 64];
        const int xo = kb < kb0 ? col0 + 8 * kb : col1 + 8 * (kb - kb0);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xl + xo);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(xl + xo + 32 * kLdsStride);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[rb][c], x0[c], acc[rb][0], 0, 0, 0);
                acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[rb][c], x1[c], acc[rb][1], 0, 0, 0);
            }
        }
    }
}

// write the wave's accumulators to LDS activation columns [dcol + 32*RB*wave, ...) (optionally ReLU)
template <int RB>
__device__ __forceinline__ void wide_store(const f32x16 (&acc)[RB][2], float* dl /* lds + (lane&31)*stride + 4*(lane>>5) + dcol + chan0 */,
                                           bool relu) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
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
// the kernel
// ------------------------------------------------------------------------------------------------
template <bool kSsr>
__global__ __launch_bounds__(256) void k_encode_mlp(const MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ wts = p.wts;
    const NetLayout& L = p.L;

    // per-lane LDS bases
    float* const xw = lds + (lane & 31) * kLdsStride + 4 * (lane >> 5);               // wide operand / result rows
    const float* const xs = lds + (16 * wave + (lane & 15)) * kLdsStride + 4 * (lane >> 4);   // skinny operand rows

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        // ---------------- encode: X[:, enc | dir] ----------------
        {
            const int pt = tid & 63;
            int gp = tile * kTilePoints + pt;
            gp = gp < p.n_points ? gp : p.n_points - 1;
            const int ray = gp / p.n_samples;
            const float* __restrict__ r = p.rays + (size_t)ray * INERF_RAY_FLOATS;
            const float zz = p.z[gp];
            float* row = lds + pt * kLdsStride;
            float x[3], v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // pts = rays_o + rays_d * z  (run_nerf.py:488): separate multiply and add
                x[c] = __fadd_rn(r[c], __fmul_rn(r[3 + c], zz));
                if (p.xyz_div != 1.0f) x[c] = __fdiv_rn(x[c], p.xyz_div);   // semantic_nerf.py:64
                v[c] = r[8 + c];
            }
            // frequencies are spread over the 4 waves; 2^f scaling is exact (run_nerf_helpers.py:212)
            for (int f = wave; f < p.l_xyz; f += kWaves) {
                const float s = (float)(1 << f);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sn, cs;
                    sincosf(x[c] * s, &sn, &cs);
                    row[kColEnc + 3 + 6 * f + c] = sn;
                    row[kColEnc + 6 + 6 * f + c] = cs;
                }
            }
            if (wave < p.l_dir) {
                const float s = (float)(1 << wave);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sn, cs;
                    sincosf(v[c] * s, &sn, &cs);
                    row[kColDir + 3 + 6 * wave + c] = sn;
                    row[kColDir + 6 + 6 * wave + c] = cs;
                }
            }
            if (wave == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) row[kColEnc + c] = x[c];
                for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) row[kColEnc + c] = 0.0f;
            }
            if (wave == 3) {
#pragma unroll
                for (int c = 0; c < 3; ++c) row[kColDir + c] = v[c];
                for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) row[kColDir + c] = 0.0f;
            }
        }
        __syncthreads();

        // ---------------- trunk: 8 x (Linear + ReLU), ping-pong A/B ----------------
        auto wide256 = [&](const GemmSlot& s, int c0, int kb0, int c1, int kb1, int dcol, bool relu) {
            f32x16 acc[2][2];
            wide_gemm<2>(wts + s.w + (size_t)wave * (kb0 + kb1) * 2 * 256, wts + s.b + 64 * wave, xw, c0, kb0, c1, kb1,
                         lane, acc);
            wide_store<2>(acc, xw + dcol + 64 * wave, relu);
            __syncthreads();
        };
        auto wide128 = [&](const GemmSlot& s, int c0, int kb0, int c1, int kb1, int dcol, bool relu) {
            f32x16 acc[1][2];
            wide_gemm<1>(wts + s.w + (size_t)wave * (kb0 + kb1) * 256, wts + s.b + 32 * wave, xw, c0, kb0, c1, kb1, lane,
                         acc);
            wide_store<1>(acc, xw + dcol + 32 * wave, relu);
            __syncthreads();
        };
        wide256(L.trunk[0], kColEnc, 8, 0, 0, kColA, true);
        wide256(L.trunk[1], kColA, 32, 0, 0, kColB, true);
        wide256(L.trunk[2], kColB, 32, 0, 0, kColA, true);
        wide256(L.trunk[3], kColA, 32, 0, 0, kColB, true);
        wide256(L.trunk[4], kColB, 32, 0, 0, kColA, true);
        wide256(L.trunk[5], kColEnc, 8, kColA, 32, kColB, true);     // cat([pts, h]) (run_nerf_helpers.py:290-291)
        wide256(L.trunk[6], kColB, 32, 0, 0, kColA, true);
        wide256(L.trunk[7], kColA, 32, 0, 0, kColB, true);           // h7 in B

        // ---------------- heads ----------------
        const int my_pt = tile * kTilePoints + 16 * wave + (lane & 15);     // the point this lane reports in skinny results
        const bool my_valid = my_pt < p.n_points;
        float* const out_row = p.raw + (size_t)(my_valid ? my_pt : 0) * p.channels;

        // sigma = alpha_linear(h7)  (run_nerf_helpers.py:294) - no activation here, ReLU happens in raw2outputs
        const f32x4 sig4 = skinny_gemm<16>(wts + L.alpha.w, wts + L.alpha.b, xs + kColB, lane);

        if (kSsr && L.sem_rbs > 0) {
            // semantic head: Linear(256,128)+ReLU then Linear(128,C), logits raw (semantic_nerf.py:110,142)
            wide128(L.sem1, kColB, 32, 0, 0, kColA, true);
            for (int rb = 0; rb < L.sem_rbs; ++rb) {
                const f32x4 lg = skinny_gemm<8>(wts + L.sem2.w + rb * 8 * 256, wts + L.sem2.b + 16 * rb, xs + kColA, lane);
                const int ch0 = 16 * rb + 4 * (lane >> 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (my_valid && ch0 + i < p.n_classes) out_row[INERF_BASE_CHANNELS + ch0 + i] = lg[i];
            }
            __syncthreads();                                  // A is about to be overwritten
        }

        // albedo / shading hidden layers (one 256-row GEMM), then their 3+1 outputs (one skinny GEMM)
        wide256(L.as1, kColB, 32, 0, 0, kColA, true);
        const f32x4 as4 = skinny_gemm<16>(wts + L.as2.w, wts + L.as2.b, xs + kColA, lane);
        __syncthreads();                                      // A is about to be overwritten

        // feature = feature_linear(h7) (no activation), views layer over cat([feature, dirs]), residual head
        wide256(L.feat, kColB, 32, 0, 0, kColA, false);
        wide128(L.views, kColA, 32, kColDir, 4, kColB, true);
        const f32x4 res4 = skinny_gemm<8>(wts + L.res.w, wts + L.res.b, xs + kColB, lane);

        if (lane < 16 && my_valid) {
            const float a0 = sigmoid_ref(as4[0]), a1 = sigmoid_ref(as4[1]), a2 = sigmoid_ref(as4[2]);
            const float sh = sigmoid_ref(as4[3]);
            const float r0 = sigmoid_ref(res4[0]), r1 = sigmoid_ref(res4[1]), r2 = sigmoid_ref(res4[2]);
            // rgb = albedo * shading + residual (run_nerf_helpers.py:320): multiply, then add
            out_row[0] = __fadd_rn(__fmul_rn(a0, sh), r0);
            out_row[1] = __fadd_rn(__fmul_rn(a1, sh), r1);
            out_row[2] = __fadd_rn(__fmul_rn(a2, sh), r2);
            out_row[3] = sig4[0];
            out_row[4] = a0; out_row[5] = a1; out_row[6] = a2;
            out_row[7] = sh;
            out_row[8] = r0; out_row[9] = r1; out_row[10] = r2;
        }
        if (kSsr && p.endpoint) {
            // show_endpoint: append the post-ReLU views activation (semantic_nerf.py:163-164,181)
            const int col = tid & 127;
            const int base = INERF_BASE_CHANNELS + p.n_classes;
            for (int pt = tid >> 7; pt < kTilePoints; pt += 2) {
                const int gp = tile * kTilePoints + pt;
                if (gp < p.n_points) p.raw[(size_t)gp * p.channels + base + col] = lds[pt * kLdsStride + kColB + col];
            }
        }
        // No barrier needed here: the next tile's encode writes only the enc/dir columns, whose last
        // readers (trunk[5], views) finished before barriers every wave has already passed; buffer B
        // (still being read by slower waves' residual GEMM) is first rewritten two barriers later.
    }
}

static int g_num_cus = 0;
static int g_last_hip_error = 0;

int device_cus() {
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_num_cus;
}

int record(hipError_t e) {
    if (e == hipSuccess) return INERF_OK;
    g_last_hip_error = (int)e;
    return INERF_E_HIP;
}

}  // namespace inerf

extern "C" int inerf_last_hip_error(void) { return inerf::g_last_hip_error; }

extern "C" int inerf_encode_mlp(const inerf_net_desc* net, const float* packed, const float* rays, const float* z,
                                int64_t n_rays, int n_samples, uint32_t flags, float* raw_out, void* stream) {
    using namespace inerf;
    if (!net || !packed || !rays || !z || !raw_out || n_rays < 0 || n_samples < 1) return INERF_E_INVALID;
    if (!net_supported(*net)) return INERF_E_UNSUPPORTED;
    if (n_rays == 0) return INERF_OK;
    const int64_t n_points = n_rays * (int64_t)n_samples;
    if (n_points >= (int64_t)1 << 31) return INERF_E_UNSUPPORTED;     // caller chunks (the front-ends do)
    const bool ssr = net->variant == INERF_VARIANT_SSR;
    MlpParams p;
    p.wts = packed; p.rays = rays; p.z = z; p.raw = raw_out;
    p.L = make_layout(*net);
    p.n_points = (int)n_points;
    p.n_samples = n_samples;
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    p.endpoint = (ssr && (flags & INERF_FLAG_ENDPOINT)) ? 1 : 0;
    p.n_classes = ssr ? net->n_classes : 0;
    p.channels = INERF_BASE_CHANNELS + p.n_classes + (p.endpoint ? INERF_ENDPOINT_DIM : 0);
    p.l_xyz = net->l_xyz; p.l_dir = net->l_dir; p.xyz_div = net->xyz_div;
    const int grid = p.n_tiles < device_cus() ? p.n_tiles : device_cus();
    auto kern = ssr ? k_encode_mlp<true> : k_encode_mlp<false>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[ssr]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kLdsBytes);
        if (e != hipSuccess) return record(e);
        attr_set[ssr] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLdsBytes, (hipStream_t)stream, p);
    return record(hipGetLastError());
}
