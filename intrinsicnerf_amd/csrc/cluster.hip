// cluster.hip - nearest-anchor lookup of the albedo clustering:
//   Cluster_Manager.dest_color / dest_class  SSR/training/cluster.py:73-98
//   Cluster.dest_color / dest_class          cluster.py:275-297   (batches of 10 240 pixels)
//   compute_dist + nearest_anchor            cluster.py:299-310   ([anchors, batch] distance matrix + argmin)
//   mapping_color                            cluster.py:324-330
// The reference loops over the semantic classes on the host (a boolean-mask gather, a distance matrix of
// anchors x 10 240 floats, an argmin and a scatter per class: ~10 launches and one host sync each).  Here ONE launch
// covers every class: the anchors of all classes sit in one table ({a0, a1, a2, |a|^2} per anchor, class c owning
// rows anchor_begin[c] .. anchor_begin[c+1]) and nothing but the 12-byte pixel and its label is read from HBM -
// the distance matrix never exists.
//
// Mapping: a wave owns a tile of 8 consecutive pixels, its 64 lanes stride over the anchors of the tile's class
// (16-byte coalesced loads out of L2; every anchor is used for all 8 pixels), each lane keeps its best
// (distance, index) per pixel and a 6-step butterfly picks the winner.  Tiles whose pixels belong to several
// classes (class boundaries in a frame) take one pass per distinct class.  Batches of up to 16 384 pixels (a
// training step: 1 024 random pixels) get one wave per pixel instead, so that the chip is filled and no tile mixes classes.
//
// Arithmetic: d_rgb = (sum/3*f, g/sum, b/sum) and dist = (|a|^2 + |b|^2) - 2 a.b exactly as written in the
// reference (fp32, true divisions, no contraction); a.b, which the reference takes from a library GEMM with an
// unspecified accumulation order, is the FMA chain fma(a2,b2, fma(a1,b1, a0*b0)).  argmin follows torch's rule:
// the first minimal index wins, a NaN distance counts as smaller than everything.
#include <hip/hip_runtime.h>

#include <climits>

#include "layout.h"

namespace inerf {

int record(hipError_t e);

constexpr int kFrameTile = 8;      // pixels per wave when there are enough pixels to fill the chip that way
constexpr int kBatchTile = 1;      // ... and for small batches (a training step): one pixel per wave

struct ClusterTables {
    const float4* anchors;        // [A] {a0, a1, a2, a0^2+a1^2+a2^2}
    const int* links;             // [A] centre of each anchor, relative to its class
    const int* anchor_begin;      // [K+1]
    const float* factor;          // [K] intensity_factor of each class's cluster
    const float* centers;         // [Ctot,3]
    const int* center_begin;      // [K+1]
    int n_classes;
};

constexpr long long kSmallBatch = 16384;

// true when (d1, i1) loses against (d2, i2) under torch.argmin's ordering
__device__ __forceinline__ bool loses(float d1, int i1, float d2, int i2) {
    if (i2 == INT_MAX) return false;
    if (i1 == INT_MAX) return true;
    const bool n1 = d1 != d1, n2 = d2 != d2;
    if (n1 || n2) return n1 && n2 ? i2 < i1 : n2;
    return d2 < d1 || (d2 == d1 && i2 < i1);
}

// every lane strides over the class's anchors and keeps, per pixel of the tile, the best (distance, index) it has seen
template <int kPixTile, bool kMasked>
__device__ __forceinline__ void scan_anchors(const float4* __restrict__ rows, int count, int lane, unsigned long long members,
                                             const float (&q0)[kPixTile], const float (&q1)[kPixTile], const float (&q2)[kPixTile],
                                             const float (&qs)[kPixTile], float (&best)[kPixTile], int (&idx)[kPixTile]) {
    constexpr int kAhead = kPixTile == 1 ? 8 : 4;                               // independent 16-byte loads in flight per lane (L2 latency)
    for (int a0 = lane; a0 < count; a0 += 64 * kAhead) {
        float4 an[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) an[u] = a0 + 64 * u < count ? rows[a0 + 64 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int a = a0 + 64 * u;
            if (a >= count) break;
#pragma unroll
            for (int j = 0; j < kPixTile; ++j) {
                if (kMasked && !((members >> j) & 1ull)) continue;      // wave-uniform
                const float dot = __fmaf_rn(an[u].z, q2[j], __fmaf_rn(an[u].y, q1[j], __fmul_rn(an[u].x, q0[j])));
                const float dist = __fsub_rn(__fadd_rn(an[u].w, qs[j]), __fmul_rn(2.0f, dot));
                // ascending a inside the lane: strict < keeps the first minimum, a NaN sticks once taken
                const bool take = idx[j] == INT_MAX || dist < best[j] || (dist != dist && best[j] == best[j]);
                best[j] = take ? dist : best[j];
                idx[j] = take ? a : idx[j];
            }
        }
    }
}

template <int kPixTile>
__global__ __launch_bounds__(256) void k_cluster_lookup(const float* __restrict__ rgb, const long long* __restrict__ label,
                                                        long long n, ClusterTables t, int ignore_label,
                                                        float* __restrict__ out_color, long long* __restrict__ out_class) {
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long p = tile * kPixTile + lane;
    if (tile * kPixTile >= n) return;                       // whole wave
    const bool pixel_lane = lane < kPixTile && p < n;

    float r = 0.f, g = 0.f, b = 0.f;
    int cls = -1;
    if (pixel_lane) {
        r = rgb[p * 3 + 0]; g = rgb[p * 3 + 1]; b = rgb[p * 3 + 2];
        long long lab = ignore_label ? 0 : label[p];
        if (lab >= 0 && lab < t.n_classes && t.anchor_begin[lab + 1] > t.anchor_begin[lab]) cls = (int)lab;
    }
    // mapping_color (cluster.py:324-330): intensity = r+g+b; (intensity/3.0*factor, g/intensity, b/intensity)
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, sb = 0.f;
    if (cls >= 0) {
        const float intensity = __fadd_rn(__fadd_rn(r, g), b);
        d0 = __fmul_rn(__fdiv_rn(intensity, 3.0f), t.factor[cls]);
        d1 = __fdiv_rn(g, intensity);
        d2 = __fdiv_rn(b, intensity);
        sb = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2));
    }
    float q0[kPixTile], q1[kPixTile], q2[kPixTile], qs[kPixTile];
#pragma unroll
    for (int j = 0; j < kPixTile; ++j) {
        q0[j] = __shfl(d0, j); q1[j] = __shfl(d1, j); q2[j] = __shfl(d2, j); qs[j] = __shfl(sb, j);
    }

    int winner = -1;                                        // pixel lanes: index of the nearest anchor inside the class
    unsigned long long todo = __ballot(cls >= 0);
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const int c = __shfl(cls, first);
        const unsigned long long members = __ballot(cls == c) & todo;
        todo &= ~members;
        const int begin = t.anchor_begin[c], count = t.anchor_begin[c + 1] - begin;
        float best[kPixTile];
        int idx[kPixTile];
#pragma unroll
        for (int j = 0; j < kPixTile; ++j) { best[j] = 0.f; idx[j] = INT_MAX; }
        const float4* __restrict__ rows = t.anchors + begin;
        if (__popcll(members) == kPixTile)                  // the usual tile of a frame: one class, no tests inside the loop
            scan_anchors<kPixTile, false>(rows, count, lane, members, q0, q1, q2, qs, best, idx);
        else                                                // mixed tile (training batch): only this class's pixels
            scan_anchors<kPixTile, true>(rows, count, lane, members, q0, q1, q2, qs, best, idx);
#pragma unroll
        for (int j = 0; j < kPixTile; ++j) {
            if (!((members >> j) & 1ull)) continue;         // wave-uniform
            float bd = best[j];
            int bi = idx[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float od = __shfl_xor(bd, o);
                const int oi = __shfl_xor(bi, o);
                if (loses(bd, bi, od, oi)) { bd = od; bi = oi; }
            }
            if (lane == j) winner = bi;
        }
    }

    if (!pixel_lane) return;
    if (cls >= 0) {
        const int link = t.links[t.anchor_begin[cls] + winner];
        if (out_color) {
            const float* ctr = t.centers + 3ll * (t.center_begin[cls] + link);
            out_color[p * 3 + 0] = ctr[0]; out_color[p * 3 + 1] = ctr[1]; out_color[p * 3 + 2] = ctr[2];
        }
        if (out_class) out_class[p] = link;
    } else {                                                // no cluster for this label: colour unchanged, class 0
        if (out_color) { out_color[p * 3 + 0] = r; out_color[p * 3 + 1] = g; out_color[p * 3 + 2] = b; }
        if (out_class) out_class[p] = 0;
    }
}

}  // namespace inerf

extern "C" int inerf_cluster_lookup(const float* rgb, const int64_t* label, int64_t n_pixels, const float* anchors,
                                    const int32_t* links, const int32_t* anchor_begin, const float* factor,
                                    const float* centers, const int32_t* center_begin, int n_classes, uint32_t flags,
                                    float* out_color, int64_t* out_class, void* stream) {
    using namespace inerf;
    if (n_pixels == 0) return INERF_OK;
    const bool ignore_label = (flags & INERF_CLUSTER_IGNORE_LABEL) != 0;
    if (n_pixels < 0 || !rgb || (!label && !ignore_label) || !anchors || !links || !anchor_begin || !factor || !centers ||
        !center_begin || n_classes < 1 || (!out_color && !out_class))
        return INERF_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(anchors) & 15) != 0) return INERF_E_INVALID;      // float4 loads
    const bool small = n_pixels <= kSmallBatch;
    const int tile = small ? kBatchTile : kFrameTile;
    const long long tiles = (n_pixels + tile - 1) / tile;
    const long long blocks = (tiles + 3) / 4;
    if (blocks > 0x7fffffffLL) return INERF_E_UNSUPPORTED;
    ClusterTables t{reinterpret_cast<const float4*>(anchors), links, anchor_begin, factor, centers, center_begin, n_classes};
    auto kernel = small ? k_cluster_lookup<kBatchTile> : k_cluster_lookup<kFrameTile>;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rgb,
                       reinterpret_cast<const long long*>(label), (long long)n_pixels, t, ignore_label ? 1 : 0, out_color,
                       reinterpret_cast<long long*>(out_class));
    return record(hipGetLastError());
}
