// frame_ops.hip - image-level producers / consumers either side of render_rays:
//   k_gen_rays     : camera pose -> the [N, 11] ray batch render() / create_rays assemble
//                    (run_nerf_helpers.py:359-368 + run_nerf.py:99-128 | rays.py:27-67, 223-256)
//   k_frame_to_u8  : to8b of rendered maps on the device (run_nerf_helpers.py:13 | trainer.py:1242), so that an image
//                    loop (run_nerf.py:142-212 | trainer.py:1221-1389) moves a quarter of the bytes to the host
// Both are a few MB of traffic per frame - microseconds; they exist for EXACTNESS (one rounding order, the same bits
// on every device) and to take ~15 small launches and six host synchronisations out of the per-image loop.
//
// Rounding order of the ray generator.  The 2^9-frequency encoding turns a one-ulp difference of a direction into a
// 2e-4 phase error, so the reference's fp32 evaluation order is reproduced operation by operation.  What ATen does on the
// reference's CPU path was established against tests/golden/rays_generators.npz and at full frame sizes
// (800x800, 320x240; 1 and 8 threads):
//   dirs   x = (i - cx) / fx,  y = (j - cy) / fy      one fp32 subtraction, one true division
//   rays_d d_k = (x R[k][0] + y' R[k][1]) + z' R[k][2]  three products, two additions, left to right, NO fma - both for
//          torch.sum(dirs[..., None, :] * c2w[:3,:3], -1) (object-level) and for torch.matmul(R, dirs) (SSR)
//   |d|    = sqrt(fma(d2, d2, fma(d1, d1, d0 * d0)))   torch.norm's vectorised kernel contracts; then a true division
// Built with -ffp-contract=off; every operation below is an explicit __f*_rn.
#include <hip/hip_runtime.h>

#include "layout.h"

namespace inerf {

int record(hipError_t e);

struct GenRaysParams {
    const float* poses;        // [B] camera-to-world matrices, rows of 4 floats, `pose_stride` floats apart
    const float* static_poses; // optional: origin / direction come from these, view directions from `poses`
    float* out;                // [B * H * W, 11]
    long long n_rays;          // B * H * W
    int pose_stride;
    int H, W;
    float fx, fy, cx, cy, near, far;
    int opengl;                // 1: dirs = (x, -y, -1) (object-level, SSR convention "opengl"); 0: (x, y, 1) ("opencv")
};

__device__ __forceinline__ void rotate(const float* __restrict__ p, float x, float y, float z, float (&d)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
        d[k] = __fadd_rn(__fadd_rn(__fmul_rn(x, p[4 * k + 0]), __fmul_rn(y, p[4 * k + 1])), __fmul_rn(z, p[4 * k + 2]));
}

__global__ __launch_bounds__(256) void k_gen_rays(const GenRaysParams p) {
    __shared__ float stage[256 * INERF_RAY_FLOATS];
    const long long base = (long long)blockIdx.x * 256;
    const long long idx = base + threadIdx.x;
    if (idx < p.n_rays) {
        const long long hw = (long long)p.H * p.W;
        const int b = (int)(idx / hw);
        const int pix = (int)(idx - b * hw);
        const int j = pix / p.W, i = pix - j * p.W;
        const float x = __fdiv_rn(__fsub_rn((float)i, p.cx), p.fx);
        float y = __fdiv_rn(__fsub_rn((float)j, p.cy), p.fy);
        float z = 1.0f;
        if (p.opengl) { y = -y; z = -1.0f; }
        const float* __restrict__ cam = p.poses + (size_t)b * p.pose_stride;
        float d[3], v[3];
        rotate(cam, x, y, z, v);                                   // view direction: always the moving camera's
        const float* __restrict__ src = cam;
        if (p.static_poses) {                                      // c2w_staticcam (run_nerf.py:103-105 | rays.py:240-243)
            src = p.static_poses + (size_t)b * p.pose_stride;
            rotate(src, x, y, z, d);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) d[k] = v[k];
        }
        // sqrtf, not __fsqrt_rn: the latter lowers to a bare v_sqrt_f32 (1 ulp); ocml's sqrtf carries the correctly-rounding fix-up
        const float nrm = sqrtf(__fmaf_rn(v[2], v[2], __fmaf_rn(v[1], v[1], __fmul_rn(v[0], v[0]))));
        float* s = stage + threadIdx.x * INERF_RAY_FLOATS;
        s[0] = src[3]; s[1] = src[7]; s[2] = src[11];
        s[3] = d[0]; s[4] = d[1]; s[5] = d[2];
        s[6] = p.near; s[7] = p.far;
        s[8] = __fdiv_rn(v[0], nrm); s[9] = __fdiv_rn(v[1], nrm); s[10] = __fdiv_rn(v[2], nrm);
    }
    __syncthreads();
    // the block's 256 rays are 2816 consecutive floats of the output: write them coalesced
    const long long left = p.n_rays - base;
    const int n_floats = (int)(left < 256 ? left : 256) * INERF_RAY_FLOATS;
    float* __restrict__ o = p.out + base * INERF_RAY_FLOATS;
    for (int t = threadIdx.x; t < n_floats; t += 256) o[t] = stage[t];
}

// out[i] = (uint8)(255 * clip(in[i], 0, 1)) - numpy's float32 product, truncation toward zero.  NaN -> 0.
__global__ __launch_bounds__(256) void k_frame_to_u8(const float* __restrict__ in, unsigned char* __restrict__ out, long long n) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = in[i];
        const float c = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);          // NaN fails both comparisons and stays NaN
        out[i] = c == c ? (unsigned char)(int)__fmul_rn(255.0f, c) : (unsigned char)0;
    }
}

}  // namespace inerf

extern "C" int inerf_gen_rays(const float* poses, int pose_stride, const float* static_poses, int n_poses, int height, int width,
                              float fx, float fy, float cx, float cy, float near, float far, uint32_t flags, float* rays_out,
                              void* stream) {
    using namespace inerf;
    if (n_poses == 0 || height == 0 || width == 0) return INERF_OK;
    if (!poses || !rays_out || n_poses < 0 || height < 0 || width < 0 || pose_stride < 12) return INERF_E_INVALID;
    GenRaysParams p;
    p.poses = poses; p.static_poses = static_poses; p.out = rays_out;
    p.n_rays = (long long)n_poses * height * width;
    p.pose_stride = pose_stride; p.H = height; p.W = width;
    p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.near = near; p.far = far;
    p.opengl = (flags & INERF_CAM_OPENGL) ? 1 : 0;
    const long long blocks = (p.n_rays + 255) / 256;
    if (blocks > 0x7fffffffLL) return INERF_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_gen_rays, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return record(hipGetLastError());
}

extern "C" int inerf_frame_to_u8(const float* values, int64_t n, unsigned char* out, void* stream) {
    using namespace inerf;
    if (n == 0) return INERF_OK;
    if (!values || !out || n < 0) return INERF_E_INVALID;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_frame_to_u8, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, values, out, (long long)n);
    return record(hipGetLastError());
}
