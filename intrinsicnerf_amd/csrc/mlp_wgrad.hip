// mlp_wgrad.hip - weight gradients of the MLP, dW[M, N] = sum over sample points p of G[p, m] * X[p, n]
// (what autograd computes for every nn.Linear when the trainers call loss.backward(): run_nerf.py:1018,
// trainer.py:990).  G is a pre-activation gradient slot written by k_mlp_dgrad, X the matching activation slot
// kept by the training forward: both plain row-major [P, width] fp32 matrices (layout.h SaveSlot).
//
// A GEMM with a tiny output (<= 256 x 256) and K = P in the hundreds of thousands: split over K across the
// chip, one fp32 accumulator tile per workgroup - 256 x 256 floats are exactly the 512-entry register files of
// four waves - partial tiles summed afterwards (deterministic, no atomics).  Arithmetic as everywhere on this
// path: fp32 operands split into f16 hi/lo, three v_mfma_f32_32x32x16_f16 per fp32 MAC.
//
// The operands of that MFMA must hold, per lane, 8 consecutive k (= points) of one channel, while G and X are
// point-major.  The transpose is done by the matrix core itself: an MFMA of a [32 points x 16 channels]
// fragment (lane = point, 8 consecutive channels: a natural 32-byte read of a point's row) with an identity
// matrix returns that block in accumulator layout - lane = channel, registers = points - which, converted back
// to f16 (exact: the inputs were f16), IS an operand of the next MFMA.  Two such MFMAs (identity placed in columns
// 0..15 / 16..31) fill a [32 x 32] block; the resulting k order is the same permutation for G and X, so the
// contraction is unaffected.  Cost: 2 extra MFMAs per 32 x 32 block and plane (+17 %), no LDS transposes.
#include "mlp_f16_dev.h"

namespace inerf {

struct WgradParams {
    const float* G;          // [P, ldg] (pointer to the first used column)
    const float* X;          // [P, ldx]
    const float* ranges;     // device: {gmax, xmax}: upper bounds of |G| and |X|
    float* partial;          // workgroup g writes its M x N tile at partial + g * partial_stride
    float* bias_partial;     // optional: workgroup g writes its column sums of G at bias_partial + g * partial_stride
    int64_t partial_stride;  // floats
    int ldg, ldx, n_points, n_tiles, M, N;
};

constexpr int kWgFragBytes = 1024;      // one operand fragment: 64 lanes x 8 halfs

// one [32 points x 16 channels] piece of a point-major matrix as transposer operand: lane = point, 8 channels
__device__ __forceinline__ void load_piece(const float* base, int ld, int pt, int n_points, int col, float (&v)[8]) {
    if (pt < n_points) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + (size_t)pt * ld + col);
        const f32x4 b = *reinterpret_cast<const f32x4*>(base + (size_t)pt * ld + col + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.0f;
    }
}

__device__ __forceinline__ void split8(const float (&v)[8], float scale, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float t = v[i] * scale;
        const float th = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, t) & 0xFFFFE000u);
        hi[i] = (_Float16)th;
        lo[i] = (_Float16)(t - th);
    }
}

// [32 points x 32 channels] block (two 16-channel pieces g0, g1, already split) -> operands with lane = channel:
// out[q] holds, for k-block q of this 32-point half, 8 points of the lane's channel
// returns the sum of the lane's 16 transposed values (= this lane's channel over 16 of the block's 32 points)
__device__ __forceinline__ float transpose_block(f16x8 g0, f16x8 g1, f16x8 id0, f16x8 id1, f16x8 (&out)[2]) {
    f32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.0f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(g0, id0, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(g1, id1, d, 0, 0, 0);
    float sum = 0.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            out[q][i] = (_Float16)d[8 * q + i];
            sum += d[8 * q + i];
        }
    return sum;
}

// NW waves, each owning one 32-row block of the output (M = 32 * NW); CB: 32-column blocks of the output (N = 32 * CB).
// M = 256 runs as 8 waves of <= 256 registers (128 of them accumulators): two waves per SIMD, so that one wave's loads and
// conversions overlap the other's MFMAs.  (With 4 waves x 64 rows - 256 accumulator registers per lane - prefetching the
// next tile spilled and was 20 % slower than not prefetching at all.)
template <int NW, int CB>
__global__ __launch_bounds__(64 * NW, 1) void k_mlp_wgrad(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char ldsw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 31, lh = lane >> 5;
    // powers of two that bring the operands' bounds into [2^13, 2^14) (f16 hi/lo split range)
    auto pow2_for = [](float m) { int e; if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f; frexpf(m, &e); return ldexpf(1.0f, 14 - e); };
    const float sg = pow2_for(p.ranges[0]), sx = pow2_for(p.ranges[1]);

    // identity operands of the transposer: B[k][n] = (n == k) resp. (n == k + 16); lane n holds k = 8 * lh + i
    f16x8 id0, id1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        id0[i] = (_Float16)((lp == 8 * lh + i) ? 1.0f : 0.0f);
        id1[i] = (_Float16)((lp == 16 + 8 * lh + i) ? 1.0f : 0.0f);
    }

    float bias_sum = 0.0f;             // sum over this workgroup's points of G[p][channel of this lane], in units of 1 / sg
    f32x16 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;

    // LDS: X operands of two tiles (double buffer): [buf][cb][ph][q][plane] fragments
    constexpr int kBufBytes = CB * 8 * kWgFragBytes;
    auto xfrag = [&](int buf, int cb, int ph, int q, int plane) {
        return ldsw + buf * kBufBytes + ((((cb * 2 + ph) * 2 + q) * 2 + plane) * kWgFragBytes) + lane * 16;
    };
    constexpr int XS = (CB + NW - 1) / NW;        // column blocks of X this wave converts (cb = wave + NW * i)

    // The next tile's X rows (this wave's share) are requested before the contraction and converted after it; G's rows
    // are requested at the top of a tile and arrive while the X share is converted.  One barrier per tile.
    float xraw[XS][2][2][8];
    auto load_x = [&](int tile) {
        const int pt_base = tile * kTilePoints;
#pragma unroll
        for (int i = 0; i < XS; ++i) {
            const int cb = wave + NW * i;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    if (cb < CB) load_piece(p.X, p.ldx, pt_base + 32 * ph + lp, p.n_points, 32 * cb + 16 * g + 8 * lh, xraw[i][ph][g]);
        }
    };

    int buf = 0;
    if ((int)blockIdx.x < p.n_tiles) load_x(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int pt_base = tile * kTilePoints;
        float graw[2][2][8];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                load_piece(p.G, p.ldg, pt_base + 32 * ph + lp, p.n_points, 32 * wave + 16 * g + 8 * lh, graw[ph][g]);
        // ---- X: every wave transposes its share of the column blocks and parks the operands in LDS[buf] ----
#pragma unroll
        for (int i = 0; i < XS; ++i) {
            const int cb = wave + NW * i;
            if (cb < CB) {
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    f16x8 h0, l0, h1, l1, th[2], tl[2];
                    split8(xraw[i][ph][0], sx, h0, l0);
                    split8(xraw[i][ph][1], sx, h1, l1);
                    transpose_block(h0, h1, id0, id1, th);
                    transpose_block(l0, l1, id0, id1, tl);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        *reinterpret_cast<f16x8*>(xfrag(buf, cb, ph, q, 0)) = th[q];
                        *reinterpret_cast<f16x8*>(xfrag(buf, cb, ph, q, 1)) = tl[q];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- G: this wave's row block, kept in registers ----
        f16x8 gh[2][2], gl[2][2];          // [ph][q]
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            f16x8 h0, l0, h1, l1;
            split8(graw[ph][0], sg, h0, l0);
            split8(graw[ph][1], sg, h1, l1);
            bias_sum += transpose_block(h0, h1, id0, id1, gh[ph]);
            bias_sum += transpose_block(l0, l1, id0, id1, gl[ph]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tile + (int)gridDim.x < p.n_tiles) load_x(tile + gridDim.x);
        __syncthreads();                   // LDS[buf] complete; LDS[buf ^ 1] (last read before the previous barrier) is free
        // ---- contraction over the 64 points of the tile ----
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const f16x8 xh = *reinterpret_cast<const f16x8*>(xfrag(buf, cb, ph, q, 0));
                    const f16x8 xl = *reinterpret_cast<const f16x8*>(xfrag(buf, cb, ph, q, 1));
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[ph][q], xh, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[ph][q], xl, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[ph][q], xh, acc[cb], 0, 0, 0);
                }
        buf ^= 1;
    }

    // ---- this workgroup's partial tile: row m = channel of G, column n = channel of X ----
    const float back = 1.0f / (sg * sx);
    float* out = p.partial + (size_t)blockIdx.x * p.partial_stride;
    if (p.bias_partial) {              // the two lane halves hold complementary points of the same channel
        const float both = bias_sum + __shfl_xor(bias_sum, 32);
        if (lh == 0) p.bias_partial[(size_t)blockIdx.x * p.partial_stride + 32 * wave + lp] = both / sg;
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = 32 * wave + (j & 3) + 8 * (j >> 2) + 4 * lh;
            out[(size_t)m * p.N + 32 * cb + lp] = acc[cb][j] * back;
        }
}

}  // namespace inerf

// G[P, ldg], X[P, ldx] row-major device matrices (pointers to the first used column; M and N columns are read);
// ranges: device {gmax, xmax}; workgroup g of inerf_wgrad_grid() writes its tile (row-major [M, N]) at
// partial + g * partial_stride and, if bias_partial is given, its column sums of G ([M]) at bias_partial + g * partial_stride:
// several gradients can share one [grid, stride] buffer and one final sum.  Supported shapes: M in {128, 256},
// N in {32, 64, 128, 256}.
extern "C" int inerf_wgrad_grid(int64_t n_points) {
    using namespace inerf;
    if (n_points <= 0) return 0;
    const int64_t tiles = (n_points + kTilePoints - 1) / kTilePoints;
    return (int)(tiles < device_cus() ? tiles : device_cus());
}

extern "C" int inerf_mlp_weight_gradient(const float* G, int ldg, const float* X, int ldx, int64_t n_points, int M, int N,
                                         const float* ranges, float* partial, float* bias_partial, int64_t partial_stride,
                                         void* stream) {
    using namespace inerf;
    if (!G || !X || !ranges || !partial || n_points <= 0 || ldg < M || ldx < N) return INERF_E_INVALID;
    if (n_points >= (int64_t)1 << 31) return INERF_E_UNSUPPORTED;
    if ((ldg & 3) || (ldx & 3) || (((uintptr_t)G | (uintptr_t)X) & 15)) return INERF_E_INVALID;      // 16-byte row pieces
    WgradParams p;
    if (partial_stride < (int64_t)M * N) return INERF_E_INVALID;
    p.G = G; p.X = X; p.ranges = ranges; p.partial = partial; p.bias_partial = bias_partial; p.partial_stride = partial_stride;
    p.ldg = ldg; p.ldx = ldx; p.n_points = (int)n_points; p.M = M; p.N = N;
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    const int grid = inerf_wgrad_grid(n_points);
    const int cb = N / 32;
    if ((M != 128 && M != 256) || N % 32 || cb < 1 || cb > 8) return INERF_E_UNSUPPORTED;
    const int lds = 2 * cb * 8 * kWgFragBytes;          // double-buffered X operands
    void (*kern)(const WgradParams) = nullptr;
#define INERF_WG_CASE(NWV, CBV) if (M == 32 * NWV && cb == CBV) kern = k_mlp_wgrad<NWV, CBV>;
    INERF_WG_CASE(8, 8) INERF_WG_CASE(8, 2) INERF_WG_CASE(4, 8) INERF_WG_CASE(4, 1) INERF_WG_CASE(8, 1) INERF_WG_CASE(4, 2)
    INERF_WG_CASE(8, 4) INERF_WG_CASE(4, 4)
#undef INERF_WG_CASE
    if (!kern) return INERF_E_UNSUPPORTED;
    if (lds > 64 * 1024) {          // only <8, 8> and <4, 8>; set on every launch (a few microseconds): the attribute is per device
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return record(e);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(M * 2), lds, (hipStream_t)stream, p);       // 64 threads per 32-row block
    return record(hipGetLastError());
}
