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

// one [32 points x 16 channels] piece of a point-major matrix as transposer operand: lane = point, 8 channels.  Through a
// buffer descriptor over the matrix's n_points rows and ONE 32-bit offset per (tile, half) that carries the WHOLE offset (the
// range check sees all of it; the pieces of a half differ by immediates) - with 64-bit per-lane pointers the addresses of a
// tile's 16 pieces cost more registers than prefetching G leaves room for.  Rows beyond the end read as zeros: no branches.
__device__ __forceinline__ void load_piece(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, float (&v)[8]) {
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0));
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff + 16u), 0, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}

// v * scale -> f16 hi (towards zero) + f16 lo (the remainder, rounded once): mlp_f16_dev.h split_pair, 3 instructions per pair
__device__ __forceinline__ void split8(const float (&v)[8], float scale, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        f16x2 h2, l2;
        split_pair(v[i] * scale, v[i + 1] * scale, h2, l2);
        hi[i] = h2[0]; hi[i + 1] = h2[1];
        lo[i] = l2[0]; lo[i + 1] = l2[1];
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
        for (int i = 0; i < 8; i += 2) {       // the values ARE f16 numbers (f16 x 1.0): any rounding mode converts them exactly
            const f16x2 pk = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(d[8 * q + i], d[8 * q + i + 1]));
            out[q][i] = pk[0]; out[q][i + 1] = pk[1];
            sum += d[8 * q + i] + d[8 * q + i + 1];
        }
    return sum;
}

// NW waves, each owning one 32-row block of the output (M = 32 * NW); CB: 32-column blocks of the output (N = 32 * CB).
// M = 256 runs as 8 waves of <= 256 registers (128 of them accumulators): two waves per SIMD.
//
// The points are walked in steps of 32 (one k-block pair of the MFMA).  Per step a wave (a) CONVERTS the next step's rows -
// its share of X (split, transposed, parked in LDS for everybody) and its own 32 channels of G (kept in registers) - and
// (b) CONTRACTS the current step: 6 x CB MFMAs against the X operands in LDS.  One barrier per step separates "everybody has
// converted step k" from "anybody contracts step k"; between two barriers (a) and (b) are independent (double-buffered LDS),
// so the two waves that share a SIMD do them in OPPOSITE order: waves 0..3 contract first, waves 4..7 convert first - one
// wave's splits, conversions and LDS writes run under the other's MFMAs.  (Until round 3 every wave converted, then every
// wave contracted, a 64-point tile at a time: the cycle stamps of scripts/wgrad_timeline.py showed a wave waiting at the
// barrier for half of a tile's 13 600 cycles while its SIMD partner was still converting and the matrix core idled.)
// Rows are requested two steps ahead, right after the conversion that frees their registers.
template <int NW, int CB>
__global__ __launch_bounds__(64 * NW, 1) void k_mlp_wgrad(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char ldsw[];
    constexpr int kStep = 32;                                     // points per step
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 31, lh = lane >> 5;
    const bool convert_first = NW == 8 && wave >= 4;             // the SIMD partner of wave - 4
    // powers of two that bring the operands' bounds into [2^13, 2^14) (f16 hi/lo split range)
    auto pow2_for = [](float m) { int e; if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f; frexpf(m, &e); return ldexpf(1.0f, 14 - e); };
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float sg = uniform(pow2_for(p.ranges[0])), sx = uniform(pow2_for(p.ranges[1]));       // wave-uniform: scalar registers

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

    // LDS: X operands of two steps (double buffer): [buf][cb][q][plane] fragments
    constexpr int kBufBytes = CB * 4 * kWgFragBytes;
    auto xfrag = [&](int buf, int cb, int q, int plane) {
        return ldsw + buf * kBufBytes + (((cb * 2 + q) * 2 + plane) * kWgFragBytes) + lane * 16;
    };
    constexpr int XS = (CB + NW - 1) / NW;        // column blocks of X this wave converts (cb = wave + NW * i)

    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, (int)((unsigned)p.n_points * (unsigned)p.ldg * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)((unsigned)p.n_points * (unsigned)p.ldx * 4u), 0x00020000);
    const unsigned g_voff0 = (unsigned)(lp * p.ldg + 32 * wave + 8 * lh) * 4u, x_voff0 = (unsigned)(lp * p.ldx + 8 * lh) * 4u;
    float xraw[XS][2][8], graw[2][8];
    auto opaque = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };      // keeps `voff + constant` an immediate, not a hoisted register
    // Requests are unconditional: behind the last step the rows lie beyond the descriptors' range, read as zeros and cost
    // nothing - and the loop body has no branches around loads (with them the compiler parked the loaded rows in scratch).
    auto load_rows = [&](int step) {
#pragma unroll
        for (int i = 0; i < XS; ++i) {
            const int cb = wave + NW * i;
            const unsigned voff = opaque(x_voff0 + ((unsigned)(step * kStep) * (unsigned)p.ldx + 32u * cb) * 4u);
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if (CB % NW == 0 || cb < CB) load_piece(x_rsrc, voff + 64u * g, xraw[i][g]);
        }
        const unsigned voff = opaque(g_voff0 + (unsigned)(step * kStep) * (unsigned)p.ldg * 4u);
#pragma unroll
        for (int g = 0; g < 2; ++g) load_piece(g_rsrc, voff + 64u * g, graw[g]);
    };
    // (a): the rows in xraw / graw -> X operands in LDS[buf], G operands in gh / gl
    auto convert = [&](int buf, f16x8 (&gh)[2], f16x8 (&gl)[2]) {
#pragma unroll
        for (int i = 0; i < XS; ++i) {
            const int cb = wave + NW * i;
            if (CB % NW == 0 || cb < CB) {
                f16x8 h0, l0, h1, l1, th[2], tl[2];
                split8(xraw[i][0], sx, h0, l0);
                split8(xraw[i][1], sx, h1, l1);
                transpose_block(h0, h1, id0, id1, th);
                transpose_block(l0, l1, id0, id1, tl);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    *reinterpret_cast<f16x8*>(xfrag(buf, cb, q, 0)) = th[q];
                    *reinterpret_cast<f16x8*>(xfrag(buf, cb, q, 1)) = tl[q];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f16x8 h0, l0, h1, l1;
        split8(graw[0], sg, h0, l0);
        split8(graw[1], sg, h1, l1);
        bias_sum += transpose_block(h0, h1, id0, id1, gh);
        bias_sum += transpose_block(l0, l1, id0, id1, gl);
        __builtin_amdgcn_sched_barrier(0);
    };
    // (b): X operands one read ahead of their MFMAs, fenced: unfenced, the scheduler hoists ~20 operand reads (80 registers) to
    // the top and spills the prefetched rows to make room; reads first, else they share registers with the running MFMAs'
    // operands and slip behind them
    auto contract = [&](int buf, const f16x8 (&gh)[2], const f16x8 (&gl)[2]) {
        f16x8 xh[2], xl[2];
        xh[0] = *reinterpret_cast<const f16x8*>(xfrag(buf, 0, 0, 0));
        xl[0] = *reinterpret_cast<const f16x8*>(xfrag(buf, 0, 0, 1));
#pragma unroll
        for (int st = 0; st < 2 * CB; ++st) {
            const int q = st / CB, cb = st % CB;
            if (st + 1 < 2 * CB) {
                xh[(st + 1) & 1] = *reinterpret_cast<const f16x8*>(xfrag(buf, (st + 1) % CB, (st + 1) / CB, 0));
                xl[(st + 1) & 1] = *reinterpret_cast<const f16x8*>(xfrag(buf, (st + 1) % CB, (st + 1) / CB, 1));
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[q], xh[st & 1], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[q], xl[st & 1], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[q], xh[st & 1], acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#ifdef INERF_WGRAD_STAMPS   // development build (scripts/build_variant.sh, scripts/wgrad_timeline.py): cycle stamps of workgroup 0's
    // waves 0 and 4 at the phase boundaries of one steady-state step, kept in scalar registers (selects, no branches: branches
    // around the stamps changed the register allocation of the whole loop), written over the start of the partial tile
    unsigned long long wg_st[5] = {0, 0, 0, 0, 0};
#define WG_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); wg_st[k] = (u == 0 && step == (int)blockIdx.x + 8 * (int)gridDim.x) ? now_ : wg_st[k]; } while (0)
#else
#define WG_STAMP(k) do { } while (0)
#endif
    // this workgroup's steps: blockIdx.x, + gridDim.x, ... of ceil(n_points / 32)
    const int n_steps = (p.n_points + kStep - 1) / kStep;
    f16x8 gh[2][2], gl[2][2];          // [step parity][q]
    load_rows(blockIdx.x);
    convert(0, gh[0], gl[0]);
    load_rows(blockIdx.x + gridDim.x);
    for (int step = blockIdx.x; step < n_steps; step += 2 * gridDim.x) {
        // two steps per iteration: the parities of the LDS buffers and of the G operand registers are compile-time constants
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            WG_STAMP(0);
            __syncthreads();           // every wave has converted step `step + u g`, every wave has contracted the one before
            WG_STAMP(1);
            if (convert_first) {
                convert(u ^ 1, gh[u ^ 1], gl[u ^ 1]);
                load_rows(step + (u + 2) * gridDim.x);
            }
            WG_STAMP(2);
            contract(u, gh[u], gl[u]);
            WG_STAMP(3);
            if (!convert_first) {
                convert(u ^ 1, gh[u ^ 1], gl[u ^ 1]);
                load_rows(step + (u + 2) * gridDim.x);
            }
            WG_STAMP(4);
        }
    }

    // ---- this workgroup's partial tile: row m = channel of G, column n = channel of X ----
    const float back = 1.0f / (sg * sx);
    float* out = p.partial + (size_t)blockIdx.x * p.partial_stride;
#ifdef INERF_WGRAD_STAMPS
    if (blockIdx.x == 0) {
        if (tid == 0 || tid == 256) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(p.bias_partial ? p.bias_partial : p.partial) + (tid ? 8 : 0);
            d[0] = 5;
            for (int i = 0; i < 5; ++i) d[1 + i] = wg_st[i];
        }
        return;
    }
#endif
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
    // rows are addressed through 32-bit buffer descriptors, the prefetch reaches three grid strides of 32-point steps beyond the end
    if ((n_points + (int64_t)32 * (3 * device_cus() + 2)) * (ldg > ldx ? ldg : ldx) * 4 >= (int64_t)1 << 32) return INERF_E_UNSUPPORTED;
    if ((ldg & 3) || (ldx & 3) || (((uintptr_t)G | (uintptr_t)X) & 15)) return INERF_E_INVALID;      // 16-byte row pieces
    WgradParams p;
    if (partial_stride < (int64_t)M * N) return INERF_E_INVALID;
    p.G = G; p.X = X; p.ranges = ranges; p.partial = partial; p.bias_partial = bias_partial; p.partial_stride = partial_stride;
    p.ldg = ldg; p.ldx = ldx; p.n_points = (int)n_points; p.M = M; p.N = N;
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    const int grid = inerf_wgrad_grid(n_points);
    const int cb = N / 32;
    if ((M != 128 && M != 256) || N % 32 || cb < 1 || cb > 8) return INERF_E_UNSUPPORTED;
    const int lds = 2 * cb * 4 * kWgFragBytes;          // X operands of two 32-point steps
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
