// ray_ops.hip - the per-ray stages around the MLP, one ray per 64-lane wavefront:
//   k_sample_coarse : stratified depths                 (run_nerf.py:464-486 | trainer.py:730-746)
//   k_composite     : raw2outputs alpha compositing     (run_nerf.py:359-412 | model_utils.py:39-116)
//   k_sample_fine   : z_mid + sample_pdf + sort + std   (run_nerf.py:499-503,519 + run_nerf_helpers.py:402-445
//                                                        | trainer.py:758-766,799 + rays.py:176-220)
// All three are HBM-bound streaming kernels over O(100 B .. 10 KB) per ray; together they are <1 % of
// the path's time (the MLP is the rest), so they are written for exactness and coalescing, not for
// instruction count.  Built with -ffp-contract=off: every mul/add rounds like the reference's.
#include <hip/hip_runtime.h>

#include "layout.h"

namespace inerf {

int record(hipError_t e);

constexpr int kRaysPerBlock = 4;      // 4 waves per workgroup, one ray each

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------
// coarse depths
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float depth_at(float near, float far, float t, bool lindisp) {
    if (!lindisp)   // near * (1.-t_vals) + far * (t_vals)                       (run_nerf.py:466)
        return __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));
    // 1./(1./near * (1.-t_vals) + 1./far * (t_vals))                            (run_nerf.py:468)
    const float a = __fmul_rn(__fdiv_rn(1.0f, near), __fsub_rn(1.0f, t));
    const float b = __fmul_rn(__fdiv_rn(1.0f, far), t);
    return __fdiv_rn(1.0f, __fadd_rn(a, b));
}

__global__ __launch_bounds__(256) void k_sample_coarse(const float* __restrict__ rays, const float* __restrict__ t_vals,
                                                       const float* __restrict__ t_rand, long long n_rays, int s_count,
                                                       int lindisp, float* __restrict__ z_out) {
    const long long total = n_rays * s_count;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long ray = i / s_count;
        const int s = (int)(i - ray * s_count);
        const float near = rays[ray * INERF_RAY_FLOATS + 6], far = rays[ray * INERF_RAY_FLOATS + 7];
        const float zc = depth_at(near, far, t_vals[s], lindisp != 0);
        float z = zc;
        if (t_rand) {   // stratified jitter inside [lower, upper] (run_nerf.py:472-486)
            const float zp = s > 0 ? depth_at(near, far, t_vals[s - 1], lindisp != 0) : zc;
            const float zn = s + 1 < s_count ? depth_at(near, far, t_vals[s + 1], lindisp != 0) : zc;
            const float lower = s > 0 ? __fmul_rn(0.5f, __fadd_rn(zc, zp)) : zc;
            const float upper = s + 1 < s_count ? __fmul_rn(0.5f, __fadd_rn(zn, zc)) : zc;
            z = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[i]));
        }
        z_out[i] = z;
    }
}

// ------------------------------------------------------------------------------------------------
// compositing: one ray per wave; sample s of chunk c sits in lane s - 64*c
// ------------------------------------------------------------------------------------------------
constexpr int kMaxChunks = 16;     // up to 1024 samples per ray

__global__ __launch_bounds__(256) void k_composite(const float* __restrict__ raw, const float* __restrict__ z,
                                                   const float* __restrict__ rays_d, int d_stride,
                                                   const float* __restrict__ noise, long long n_rays, int s_count, int ch,
                                                   int n_classes, int feat_dim, int white_bkgd, inerf_composite_out out) {
    const int lane = threadIdx.x & 63;
    const long long ray = blockIdx.x * (long long)kRaysPerBlock + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float* __restrict__ rr = raw + ray * (long long)s_count * ch;
    const float* __restrict__ zr = z + ray * (long long)s_count;
    const float* d = rays_d + ray * (long long)d_stride;
    // torch.norm(rays_d[..., None, :], dim=-1)
    const float dnorm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const int chunks = (s_count + 63) >> 6;
    __shared__ float w_smem[kRaysPerBlock][64 * kMaxChunks];   // each lane re-reads only what it wrote
    float* const w = w_smem[threadIdx.x >> 6] + lane;
    float carry = 1.0f;            // running transmittance entering this chunk
    float acc = 0.0f, depth = 0.0f;
    float m[INERF_BASE_CHANNELS];  // per-lane partial sums of w * raw[s, k] (k = 3, sigma, unused)
#pragma unroll
    for (int k = 0; k < INERF_BASE_CHANNELS; ++k) m[k] = 0.0f;
#pragma unroll 1
    for (int c = 0; c < chunks; ++c) {
        const int s = c * 64 + lane;
        const bool in = s < s_count;
        float alpha = 0.0f, zz = 0.0f;
        float v[INERF_BASE_CHANNELS];      // this sample's 11 base channels, read once (44 contiguous bytes per lane)
#pragma unroll
        for (int k = 0; k < INERF_BASE_CHANNELS; ++k) v[k] = 0.0f;
        if (in) {
            zz = zr[s];
            const float* __restrict__ rs = rr + (long long)s * ch;
#pragma unroll
            for (int k = 0; k < INERF_BASE_CHANNELS; ++k) v[k] = rs[k];
            // dists = z[s+1]-z[s], last = 1e10, times |d|           (run_nerf.py:374-377)
            const float gap = s + 1 < s_count ? __fsub_rn(zr[s + 1], zz) : 1e10f;
            const float dist = __fmul_rn(gap, dnorm);
            float sigma = v[3];
            if (noise) sigma = __fadd_rn(sigma, noise[ray * (long long)s_count + s]);
            // alpha = 1 - exp(-relu(sigma) * dist)                   (run_nerf.py:372,395)
            alpha = __fsub_rn(1.0f, expf(-__fmul_rn(fmaxf(sigma, 0.0f), dist)));
            if (sigma != sigma) alpha = sigma;                         // relu(NaN) stays NaN in torch
        }
        // exclusive product of (1 - alpha + 1e-10) in sample order  (run_nerf.py:397)
        const float f = in ? __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f) : 1.0f;
        float incl = f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(incl, o);
            if (lane >= o) incl = __fmul_rn(incl, up);
        }
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0f;
        const float trans = __fmul_rn(carry, excl);
        const float wt = in ? __fmul_rn(alpha, trans) : 0.0f;
        w[c * 64] = wt;
        carry = __fmul_rn(carry, __shfl(incl, 63));
        if (in && out.weights) out.weights[ray * (long long)s_count + s] = wt;
        acc += wt;
        depth += __fmul_rn(wt, zz);
        if (in) {
#pragma unroll
            for (int k = 0; k < INERF_BASE_CHANNELS; ++k) m[k] += __fmul_rn(wt, v[k]);
        }
    }
    acc = wave_sum(acc);
    depth = wave_sum(depth);
    const float bg = white_bkgd ? __fsub_rn(1.0f, acc) : 0.0f;
#pragma unroll
    for (int k = 0; k < INERF_BASE_CHANNELS; ++k) m[k] = k == 3 ? 0.0f : wave_sum(m[k]);

    // optional channels (semantic logits, endpoint feature): lanes = channels.  Every sample's channels are contiguous in
    // raw, so 64 lanes read 256 contiguous bytes per sample and each lane accumulates its own channel over the samples in
    // order; the weight of sample s is an LDS broadcast.  No cross-lane reduction.
    const float* const w_ray = w_smem[threadIdx.x >> 6];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the lanes now read weights other lanes of this wave wrote
    __builtin_amdgcn_wave_barrier();
    auto channel_block = [&](int first, int count) -> float {        // channel first + lane, lanes >= count idle
        float v = 0.0f;
        if (lane < count) {
            const float* __restrict__ col = rr + first + lane;
#pragma unroll 4
            for (int s = 0; s < s_count; ++s) v += __fmul_rn(w_ray[s], col[(long long)s * ch]);
        }
        return v;
    };
    if (lane == 0) {
        // white background is added to rgb, albedo and shading but NOT to residual (run_nerf.py:407-410)
        if (out.rgb) { for (int k = 0; k < 3; ++k) out.rgb[ray * 3 + k] = white_bkgd ? __fadd_rn(m[k], bg) : m[k]; }
        if (out.albedo) { for (int k = 0; k < 3; ++k) out.albedo[ray * 3 + k] = white_bkgd ? __fadd_rn(m[4 + k], bg) : m[4 + k]; }
        if (out.shading) out.shading[ray] = white_bkgd ? __fadd_rn(m[7], bg) : m[7];
        if (out.residual) { for (int k = 0; k < 3; ++k) out.residual[ray * 3 + k] = m[8 + k]; }
        if (out.acc) out.acc[ray] = acc;
        if (out.depth) out.depth[ray] = depth;
        if (out.disp) {
            // 1 / max(1e-10, depth / acc) with torch.max's NaN propagation (0/0 when acc == 0)  (run_nerf.py:404)
            const float q = __fdiv_rn(depth, acc);
            const float mx = (q != q) ? q : fmaxf(1e-10f, q);
            out.disp[ray] = __fdiv_rn(1.0f, mx);
        }
    }
    if (out.sem && n_classes > 0) {            // model_utils.py:90-94,113-114
        for (int k0 = 0; k0 < n_classes; k0 += 64) {
            const int cnt = n_classes - k0 < 64 ? n_classes - k0 : 64;
            const float v = channel_block(INERF_BASE_CHANNELS + k0, cnt);
            if (lane < cnt) out.sem[ray * (long long)n_classes + k0 + lane] = white_bkgd ? __fadd_rn(v, bg) : v;
        }
    }
    if (out.feat && feat_dim > 0) {            // the LAST feat_dim channels (model_utils.py:99-103)
        for (int k0 = 0; k0 < feat_dim; k0 += 64) {
            const int cnt = feat_dim - k0 < 64 ? feat_dim - k0 : 64;
            const float v = channel_block(ch - feat_dim + k0, cnt);
            if (lane < cnt) out.feat[ray * (long long)feat_dim + k0 + lane] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// compositing, backward: one ray per wave; recomputes alpha / transmittance / weights, then
//   G_s      = dL/dw_s      = <g_maps, raw[s, map channels]> + g_depth * z_s + g_acc + g_weights[s]
//   dL/da_s  = G_s * T_s - (sum_{k>s} G_k w_k) / (1 - a_s + 1e-10)        (w_k = a_k * prod_{j<k} (1 - a_j + 1e-10))
//   dL/dsigma_s = dL/da_s * dist_s * exp(-sigma_s * dist_s), through relu
// with the white-background and disparity terms folded into g_acc / g_depth first.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_suffix_excl(float v, int lane, float& total) {
    // exclusive suffix sum over the wave: result[lane] = sum_{l > lane} v[l]; total = sum over all lanes
    float incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float dn = __shfl_down(incl, o);
        if (lane + o < 64) incl += dn;
    }
    total = __shfl(incl, 0);
    return incl - v;
}

__global__ __launch_bounds__(256) void k_composite_bwd(const float* __restrict__ raw, const float* __restrict__ z,
                                                       const float* __restrict__ rays_d, int d_stride,
                                                       const float* __restrict__ noise, long long n_rays, int s_count, int ch,
                                                       int n_classes, int feat_dim, int white_bkgd, inerf_composite_out g,
                                                       float* __restrict__ d_raw) {
    const int lane = threadIdx.x & 63;
    const long long ray = blockIdx.x * (long long)kRaysPerBlock + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float* __restrict__ rr = raw + ray * (long long)s_count * ch;
    const float* __restrict__ zr = z + ray * (long long)s_count;
    float* __restrict__ dr = d_raw + ray * (long long)s_count * ch;
    const float* d = rays_d + ray * (long long)d_stride;
    const float dnorm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const int chunks = (s_count + 63) >> 6;
    __shared__ float smem[kRaysPerBlock][3][64 * kMaxChunks];   // alpha-complement, transmittance, weight per sample
    float* const f_s = smem[threadIdx.x >> 6][0] + lane;
    float* const t_s = smem[threadIdx.x >> 6][1] + lane;
    float* const w_s = smem[threadIdx.x >> 6][2] + lane;

    // ---- forward recompute (same arithmetic as k_composite) ----
    float carry = 1.0f, acc = 0.0f, depth = 0.0f;
#pragma unroll 1
    for (int c = 0; c < chunks; ++c) {
        const int s = c * 64 + lane;
        const bool in = s < s_count;
        float alpha = 0.0f, zz = 0.0f;
        if (in) {
            zz = zr[s];
            const float gap = s + 1 < s_count ? __fsub_rn(zr[s + 1], zz) : 1e10f;
            const float dist = __fmul_rn(gap, dnorm);
            float sigma = rr[(long long)s * ch + 3];
            if (noise) sigma = __fadd_rn(sigma, noise[ray * (long long)s_count + s]);
            alpha = __fsub_rn(1.0f, expf(-__fmul_rn(fmaxf(sigma, 0.0f), dist)));
            if (sigma != sigma) alpha = sigma;
        }
        const float f = in ? __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f) : 1.0f;
        float incl = f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(incl, o);
            if (lane >= o) incl = __fmul_rn(incl, up);
        }
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0f;
        const float trans = __fmul_rn(carry, excl);
        const float wt = in ? __fmul_rn(alpha, trans) : 0.0f;
        f_s[c * 64] = f;
        t_s[c * 64] = trans;
        w_s[c * 64] = wt;
        carry = __fmul_rn(carry, __shfl(incl, 63));
        acc += wt;
        depth += __fmul_rn(wt, zz);
    }
    acc = wave_sum(acc);
    depth = wave_sum(depth);

    // ---- per-ray output gradients (wave-uniform) ----
    float gm[INERF_BASE_CHANNELS];
#pragma unroll
    for (int k = 0; k < INERF_BASE_CHANNELS; ++k) gm[k] = 0.0f;
    if (g.rgb) { for (int k = 0; k < 3; ++k) gm[k] = g.rgb[ray * 3 + k]; }
    if (g.albedo) { for (int k = 0; k < 3; ++k) gm[4 + k] = g.albedo[ray * 3 + k]; }
    if (g.shading) gm[7] = g.shading[ray];
    if (g.residual) { for (int k = 0; k < 3; ++k) gm[8 + k] = g.residual[ray * 3 + k]; }
    const float* gsem = (g.sem && n_classes > 0) ? g.sem + ray * (long long)n_classes : nullptr;
    const float* gfeat = (g.feat && feat_dim > 0) ? g.feat + ray * (long long)feat_dim : nullptr;
    float g_acc = g.acc ? g.acc[ray] : 0.0f;
    float g_depth = g.depth ? g.depth[ray] : 0.0f;
    if (white_bkgd) {       // rgb, albedo, shading (and sem) += 1 - acc   (run_nerf.py:407-410, model_utils.py:113-114)
        g_acc -= gm[0] + gm[1] + gm[2] + gm[4] + gm[5] + gm[6] + gm[7];
        if (gsem) {
            float t = 0.0f;
            for (int k = lane; k < n_classes; k += 64) t += gsem[k];
            g_acc -= wave_sum(t);
        }
    }
    if (g.disp) {           // disp = 1 / max(1e-10, depth / acc)            (run_nerf.py:404)
        const float q = __fdiv_rn(depth, acc);
        if (!(q <= 1e-10f)) {                  // q > 1e-10, or NaN (0/0): NaN reaches depth and acc like in autograd
            const float dq = -g.disp[ray] / (q * q);
            g_depth += dq / acc;
            g_acc -= dq * q / acc;
        }
    }

    // ---- samples, last chunk first (suffix sums run towards the camera) ----
    float behind = 0.0f;            // sum of G_k w_k over the chunks already done (all k beyond this chunk)
#pragma unroll 1
    for (int c = chunks - 1; c >= 0; --c) {
        const int s = c * 64 + lane;
        const bool in = s < s_count;
        float gw = 0.0f, wt = 0.0f;
        if (in) {
            const float* __restrict__ rs = rr + (long long)s * ch;
            wt = w_s[c * 64];
#pragma unroll
            for (int k = 0; k < INERF_BASE_CHANNELS; ++k)
                if (k != 3) gw += gm[k] * rs[k];
            if (gsem) for (int k = 0; k < n_classes; ++k) gw += gsem[k] * rs[INERF_BASE_CHANNELS + k];
            if (gfeat) for (int k = 0; k < feat_dim; ++k) gw += gfeat[k] * rs[ch - feat_dim + k];
            gw += g_depth * zr[s] + g_acc;
            if (g.weights) gw += g.weights[ray * (long long)s_count + s];
        }
        float total;
        const float after = wave_suffix_excl(in ? gw * wt : 0.0f, lane, total) + behind;
        behind += total;
        if (in) {
            const float* __restrict__ rs = rr + (long long)s * ch;
            float* __restrict__ ds = dr + (long long)s * ch;
            const float d_alpha = gw * t_s[c * 64] - after / f_s[c * 64];
            const float gap = s + 1 < s_count ? __fsub_rn(zr[s + 1], zr[s]) : 1e10f;
            const float dist = __fmul_rn(gap, dnorm);
            float sigma = rs[3];
            if (noise) sigma = __fadd_rn(sigma, noise[ray * (long long)s_count + s]);
            // d alpha / d sigma = dist * exp(-sigma * dist) for sigma > 0 (relu: zero at and below 0); NaN stays NaN
            float d_sigma = sigma > 0.0f ? d_alpha * (dist * expf(-__fmul_rn(sigma, dist))) : 0.0f;
            if (sigma != sigma) d_sigma = sigma;
#pragma unroll
            for (int k = 0; k < INERF_BASE_CHANNELS; ++k) ds[k] = k == 3 ? d_sigma : wt * gm[k];
            for (int k = INERF_BASE_CHANNELS; k < ch; ++k) ds[k] = 0.0f;
            if (gsem) for (int k = 0; k < n_classes; ++k) ds[INERF_BASE_CHANNELS + k] = wt * gsem[k];
            if (gfeat) for (int k = 0; k < feat_dim; ++k) ds[ch - feat_dim + k] = wt * gfeat[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// hierarchical resampling + merge: one ray per wave, everything staged in LDS
// ------------------------------------------------------------------------------------------------
constexpr int kMaxCoarse = 256;
constexpr int kMaxImportance = 512;

struct FineSmem {
    float cdf[kMaxCoarse];
    float bins[kMaxCoarse];
    float vals[kMaxCoarse + kMaxImportance];
};

__device__ __forceinline__ bool less_nan_last(float a, float b) {
    // torch.sort order: ascending, NaN after everything
    const bool an = a != a, bn = b != b;
    if (an || bn) return !an && bn;
    return a < b;
}

// kDirect = false: z_coarse[N,sc] + full coarse weights[N,sc] (render_rays usage: bins = mid-points,
//                   pdf over weights[1:-1], merged + sorted output).
// kDirect = true : the stand-alone sample_pdf(bins, weights, N) signature: z_coarse is bins[N,sc],
//                   weights is [N,sc-1]; no merge.
template <bool kDirect>
__global__ __launch_bounds__(256) void k_sample_fine(const float* __restrict__ z_coarse, const float* __restrict__ weights,
                                                     const float* __restrict__ u, int u_per_ray, long long n_rays, int sc,
                                                     int ni, float* __restrict__ z_samples, float* __restrict__ z_merged,
                                                     float* __restrict__ z_std) {
    __shared__ FineSmem smem[kRaysPerBlock];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long ray = blockIdx.x * (long long)kRaysPerBlock + wv;
    if (ray >= n_rays) return;             // whole wave exits together; no block-level barrier is used below
    FineSmem& sm = smem[wv];
    const float* __restrict__ zr = z_coarse + ray * (long long)sc;
    const int nb = kDirect ? sc : sc - 1;  // bins = z mid-points              (run_nerf.py:499)
    const int nw = nb - 1;                 // weights[..., 1:-1]               (run_nerf.py:500)
    // wr[i + 1] below is pdf weight i: skip the first coarse weight unless the caller already sliced
    const float* __restrict__ wr = weights + ray * (long long)(kDirect ? nw : sc) - (kDirect ? 1 : 0);

    // pdf = (w + 1e-5) / sum(w + 1e-5)                                        (run_nerf_helpers.py:404-405)
    float part = 0.0f;
    for (int i = lane; i < nb; i += 64) sm.bins[i] = kDirect ? zr[i] : __fmul_rn(0.5f, __fadd_rn(zr[i + 1], zr[i]));
    for (int i = lane; i < nw; i += 64) part += __fadd_rn(wr[i + 1], 1e-5f);
    const float total = wave_sum(part);
    // cdf = [0, cumsum(pdf)]                                                   (run_nerf_helpers.py:406-407)
    float carry = 0.0f;
    if (lane == 0) sm.cdf[0] = 0.0f;
    for (int base = 0; base < nw; base += 64) {
        const int i = base + lane;
        float v = i < nw ? __fdiv_rn(__fadd_rn(wr[i + 1], 1e-5f), total) : 0.0f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(v, o);
            if (lane >= o) v = __fadd_rn(v, up);
        }
        v = __fadd_rn(v, carry);
        if (i < nw) sm.cdf[i + 1] = v;
        carry = __shfl(v, 63);
    }
    if (!kDirect) for (int i = lane; i < sc; i += 64) sm.vals[i] = zr[i];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    // inverse CDF                                                              (run_nerf_helpers.py:427-443)
    float s1 = 0.0f;
    for (int j = lane; j < ni; j += 64) {
        const float uu = u_per_ray ? u[ray * (long long)ni + j] : u[j];
        // searchsorted(cdf, u, right=True): number of cdf entries <= u
        int lo = 0, hi = nb;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sm.cdf[mid] <= uu) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nb - 1 ? lo : nb - 1;
        const float cb = sm.cdf[below], ca = sm.cdf[above];
        const float bb = sm.bins[below], ba = sm.bins[above];
        float denom = __fsub_rn(ca, cb);
        if (denom < 1e-5f) denom = 1.0f;
        const float t = __fdiv_rn(__fsub_rn(uu, cb), denom);
        const float smp = __fadd_rn(bb, __fmul_rn(t, __fsub_rn(ba, bb)));
        sm.vals[sc + j] = smp;
        if (z_samples) z_samples[ray * (long long)ni + j] = smp;
        s1 += smp;
    }
    if (z_std) {   // torch.std(z_samples, -1, unbiased=False)                  (run_nerf.py:519)
        const float mean = wave_sum(s1) / (float)ni;
        float s2 = 0.0f;
        for (int j = lane; j < ni; j += 64) {
            const float dlt = sm.vals[sc + j] - mean;
            s2 += dlt * dlt;
        }
        s2 = wave_sum(s2);
        if (lane == 0) z_std[ray] = sqrtf(s2 / (float)ni);
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    // z_vals, _ = sort(cat([z_vals, z_samples]))                               (run_nerf.py:503)
    if (!kDirect && z_merged) {
        const int n = sc + ni;
        float* __restrict__ zo = z_merged + ray * (long long)n;
        // Both lists are normally already ascending (coarse depths by construction, new samples because the
        // inverse CDF is monotone in an ascending u): then the sort is a MERGE and every element finds its slot
        // with one binary search in the other list.  Any NaN or inversion (random u) fails the check below and
        // takes the general stable rank sort instead.  Either way the output is the ascending multiset.
        bool ascending = true;
        for (int i = lane; i < sc - 1; i += 64) ascending &= sm.vals[i] <= sm.vals[i + 1];
        for (int j = lane; j < ni - 1; j += 64) ascending &= sm.vals[sc + j] <= sm.vals[sc + j + 1];
        if (lane == 0) ascending &= sm.vals[0] == sm.vals[0] && sm.vals[sc] == sm.vals[sc];      // a lone NaN (n == 1)
        if (__all(ascending)) {
            for (int i = lane; i < sc; i += 64) {           // coarse element: after the samples strictly below it
                const float v = sm.vals[i];
                int lo = 0, hi = ni;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sm.vals[sc + mid] < v) lo = mid + 1; else hi = mid;
                }
                zo[i + lo] = v;
            }
            for (int j = lane; j < ni; j += 64) {           // new sample: after the coarse depths not above it
                const float v = sm.vals[sc + j];
                int lo = 0, hi = sc;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sm.vals[mid] <= v) lo = mid + 1; else hi = mid;
                }
                zo[j + lo] = v;
            }
        } else {
            // rank sort: position = #(elements ordered before this one); ties broken by index (stable)
            for (int e = lane; e < n; e += 64) {
                const float ve = sm.vals[e];
                int rank = 0;
                for (int f = 0; f < n; ++f) {
                    const float vf = sm.vals[f];      // same address in every lane: LDS broadcast
                    rank += (less_nan_last(vf, ve) || (!less_nan_last(ve, vf) && f < e)) ? 1 : 0;
                }
                zo[rank] = ve;
            }
        }
    }
}

}  // namespace inerf

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int inerf_sample_coarse(const float* rays, const float* t_vals, const float* t_rand, int64_t n_rays,
                                   int n_samples, uint32_t flags, float* z_out, void* stream) {
    using namespace inerf;
    if (n_rays == 0) return INERF_OK;              // an empty batch: its (possibly null) pointers are never touched
    if (!rays || !t_vals || !z_out || n_rays < 0 || n_samples < 1) return INERF_E_INVALID;
    const long long total = (long long)n_rays * n_samples;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_sample_coarse, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rays, t_vals, t_rand,
                       (long long)n_rays, n_samples, (flags & INERF_FLAG_LINDISP) ? 1 : 0, z_out);
    return record(hipGetLastError());
}

extern "C" int inerf_composite(const float* raw, const float* z_vals, const float* rays_d, int rays_d_stride,
                               const float* noise, int64_t n_rays, int n_samples, int channels, int n_classes, int feat_dim,
                               uint32_t flags, const inerf_composite_out* out, void* stream) {
    using namespace inerf;
    if (n_rays == 0 && out) return INERF_OK;
    if (!raw || !z_vals || !rays_d || !out || n_rays < 0 || n_samples < 1 || rays_d_stride < 3) return INERF_E_INVALID;
    if (channels < INERF_BASE_CHANNELS || n_classes < 0 || feat_dim < 0 ||
        INERF_BASE_CHANNELS + n_classes + feat_dim > channels)
        return INERF_E_INVALID;
    if (n_samples > 64 * kMaxChunks) return INERF_E_UNSUPPORTED;
    if (n_rays == 0) return INERF_OK;
    const long long blocks = (n_rays + kRaysPerBlock - 1) / kRaysPerBlock;
    if (blocks > 0x7fffffffLL) return INERF_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_composite, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, z_vals, rays_d,
                       rays_d_stride, noise, (long long)n_rays, n_samples, channels, n_classes, feat_dim,
                       (flags & INERF_FLAG_WHITE_BKGD) ? 1 : 0, *out);
    return record(hipGetLastError());
}

extern "C" int inerf_composite_backward(const float* raw, const float* z_vals, const float* rays_d, int rays_d_stride,
                                        const float* noise, int64_t n_rays, int n_samples, int channels, int n_classes,
                                        int feat_dim, uint32_t flags, const inerf_composite_out* grads, float* d_raw,
                                        void* stream) {
    using namespace inerf;
    if (n_rays == 0 && grads) return INERF_OK;
    if (!raw || !z_vals || !rays_d || !grads || !d_raw || n_rays < 0 || n_samples < 1 || rays_d_stride < 3) return INERF_E_INVALID;
    if (channels < INERF_BASE_CHANNELS + n_classes + feat_dim || n_classes < 0 || feat_dim < 0) return INERF_E_INVALID;
    if (n_samples > 64 * kMaxChunks) return INERF_E_UNSUPPORTED;
    if (n_rays == 0) return INERF_OK;
    const long long blocks = (n_rays + kRaysPerBlock - 1) / kRaysPerBlock;
    if (blocks > 0x7fffffffLL) return INERF_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_composite_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, z_vals, rays_d,
                       rays_d_stride, noise, (long long)n_rays, n_samples, channels, n_classes, feat_dim,
                       (flags & INERF_FLAG_WHITE_BKGD) ? 1 : 0, *grads, d_raw);
    return record(hipGetLastError());
}

extern "C" int inerf_sample_fine(const float* z_coarse, const float* weights, const float* u, int64_t n_rays, int n_coarse,
                                 int n_importance, uint32_t flags, float* z_samples, float* z_merged, float* z_std,
                                 void* stream) {
    using namespace inerf;
    if (n_rays == 0) return INERF_OK;
    if (!z_coarse || !weights || !u || n_rays < 0) return INERF_E_INVALID;
    if (n_coarse < 3 || n_coarse > kMaxCoarse || n_importance < 1 || n_importance > kMaxImportance) return INERF_E_UNSUPPORTED;
    if (n_rays == 0) return INERF_OK;
    const long long blocks = (n_rays + kRaysPerBlock - 1) / kRaysPerBlock;
    if (blocks > 0x7fffffffLL) return INERF_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_sample_fine<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z_coarse, weights, u,
                       (flags & INERF_FLAG_U_PER_RAY) ? 1 : 0, (long long)n_rays, n_coarse, n_importance, z_samples,
                       z_merged, z_std);
    return record(hipGetLastError());
}

extern "C" int inerf_sample_pdf(const float* bins, const float* weights, const float* u, int64_t n_rays, int n_bins,
                                int n_samples, uint32_t flags, float* samples, void* stream) {
    using namespace inerf;
    if (n_rays == 0) return INERF_OK;
    if (!bins || !weights || !u || !samples || n_rays < 0) return INERF_E_INVALID;
    if (n_bins < 2 || n_bins > kMaxCoarse || n_samples < 1 || n_samples > kMaxImportance) return INERF_E_UNSUPPORTED;
    if (n_rays == 0) return INERF_OK;
    const long long blocks = (n_rays + kRaysPerBlock - 1) / kRaysPerBlock;
    if (blocks > 0x7fffffffLL) return INERF_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_sample_fine<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, bins, weights, u,
                       (flags & INERF_FLAG_U_PER_RAY) ? 1 : 0, (long long)n_rays, n_bins, n_samples, samples,
                       (float*)nullptr, (float*)nullptr);
    return record(hipGetLastError());
}
