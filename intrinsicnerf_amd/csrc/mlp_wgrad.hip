// mlp_wgrad.hip - weight gradients of the MLP, dW[M, N] = sum over sample points p of G[p, m] * X[p, n]
// (what autograd computes for every nn.Linear when the trainers call loss.backward(): run_nerf.py:1018,
// trainer.py:990).  G is a pre-activation gradient slot written by k_mlp_dgrad, X the matching activation slot
// kept by the training forward (layout.h SaveSlot).
//
// A GEMM with a tiny output (<= 256 x 256) and K = P in the hundreds of thousands: split over K across the
// chip, one fp32 accumulator tile per workgroup - 256 x 256 floats are exactly the 512-entry register files of
// four waves - partial tiles summed afterwards (deterministic, no atomics).  Arithmetic as everywhere on this
// path: fp32 operands split into f16 hi/lo, three v_mfma_f32_32x32x16_f16 per fp32 MAC.
//
// The operands of that MFMA must hold, per lane, 8 consecutive k (= points) of one channel, while the network
// kernels work point-major.  Two kernels:
//   * k_mlp_wgrad_frag - every product whose operands are both FRAGMENT slots (all thirteen of the object-level network, in ONE
//     launch: nine 256 x 256, two 256 x 64, 128 x 256, 128 x 32): both operands arrive i.e. split into f16 hi / lo and in operand order (the producers transpose with the matrix core and
//     store whole 1 KB fragments) - the activations at the forward's fixed scale, the gradients NORMALISED per point, with
//     the points' normalisers beside them (their common scale is only known now).  The kernel is a ring of LDS stages filled
//     by LDS-DMA (buffer_load ... lds: no row registers, no transposition), a re-scaling of the wave's own G operands to the
//     batch's max |dz|, and 24 MFMAs per wave and 16-point k-block; bound by HBM.
//   * k_mlp_wgrad - products with a row-format operand (the SSR semantic output layer's; the public single-product entry
//     points): row-format X (and G, unless it is a fragment
//     slot) transposed inside the kernel by the matrix core: an MFMA of a [32 points x 16 channels] fragment (lane = point,
//     8 consecutive channels: a natural 32-byte read of a point's row) with an identity matrix returns that block in
//     accumulator layout - lane = channel, registers = points - which, converted back to f16 (exact: the inputs were f16), IS
//     an operand of the next MFMA.  The resulting k order is layout.h's frag_point order, the fragment slots' own.
#include <cstdio>
#include <cstdlib>

#include "mlp_f16_dev.h"

namespace inerf {

struct WgradParams {
    const float* G;          // [P, ldg] (pointer to the first used column)
    const float* X;          // [P, ldx]
    const float* ranges;     // device: {gmax, xmax}: upper bounds of |G| and |X|
    const float* g_scale;    // GFRAG: the points' normalisers s_p (64 * n_tiles floats): G's fragments hold kActScale * dz / s_p
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
// M = 256 runs as 8 waves of <= 256 registers (128 of them accumulators): two waves per SIMD, so that one wave's loads and
// conversions overlap the other's MFMAs.  (With 4 waves x 64 rows - 256 accumulator registers per lane - prefetching the
// next tile spilled and was 20 % slower than not prefetching at all.)
// GFRAG: G is a fragment slot of the gradient buffer (this wave's 32 channels = channel block `wave`): its operands are loaded
// in operand order - two 16-byte requests per lane and k-block, plus the eight points' normalisers - and brought to this
// product's scale (hi + lo, times s_p, split again).
// XFRAG: X is a fragment slot of the activation buffer (256 channels at the forward's fixed scale): a tile's 64 fragments ARE the
// LDS image the contraction reads - moved there by LDS-DMA a tile ahead (wave w: k-block w), nothing to convert.
template <int NW, int CB, bool GFRAG = false, bool XFRAG = false>
__global__ __launch_bounds__(64 * NW, 1) void k_mlp_wgrad(const WgradParams p) {
    static_assert(!GFRAG || NW == 8, "fragment slots are 256 channels wide");
    static_assert(!XFRAG || (NW == 4 && CB == 8 && !GFRAG), "fragment X: 256 channels, one k-block per wave");
    extern __shared__ __attribute__((aligned(16))) char ldsw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 31, lh = lane >> 5;
    // powers of two that bring the operands' bounds into [2^13, 2^14) (f16 hi/lo split range)
    auto pow2_for = [](float m) { int e; if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f; frexpf(m, &e); return ldexpf(1.0f, 14 - e); };
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float sg = uniform(pow2_for(p.ranges[0])), sx = XFRAG ? kActScale : uniform(pow2_for(p.ranges[1]));       // wave-uniform: scalar registers

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
        if constexpr (XFRAG) return ldsw + buf * kBufBytes + frag_off(2 * ph + q, cb, plane) + lane * 16;       // the slot's own order
        else return ldsw + buf * kBufBytes + ((((cb * 2 + ph) * 2 + q) * 2 + plane) * kWgFragBytes) + lane * 16;
    };
    constexpr int XS = XFRAG ? 1 : (CB + NW - 1) / NW;        // column blocks of X this wave converts (cb = wave + NW * i)

    // The next tile's rows - this wave's share of X and its own 32 channels of G - are requested before the contraction and
    // converted after it: a tile's loads have a whole contraction (~4 000 cycles) to arrive.  (Until round 3 G was requested
    // at the top of its own tile and waited for - ~3 000 exposed cycles of a tile's 20 000.)  One barrier per tile.
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0,
        GFRAG ? (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes) : (int)((unsigned)p.n_points * (unsigned)p.ldg * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0,
        XFRAG ? (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes) : (int)((unsigned)p.n_points * (unsigned)p.ldx * 4u), 0x00020000);
    const unsigned g_voff0 = (unsigned)(lp * p.ldg + 32 * wave + 8 * lh) * 4u, x_voff0 = (unsigned)(lp * p.ldx + 8 * lh) * 4u;
    float xraw[XS][2][2][8], graw[GFRAG ? 1 : 2][2][8];
    f16x8 gfr[GFRAG ? 2 : 1][2][2];           // GFRAG: [point half][k-block][hi | lo]
    f32x4 gsc[GFRAG ? 2 : 1][2][2];           // ... and the normalisers of the lane's points 0..3 | 4..7
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GFRAG ? p.g_scale : p.G), 0,
                                                                             GFRAG ? p.n_tiles * kTilePoints * 4 : 0, 0x00020000);
    auto opaque = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };      // keeps `voff + constant` an immediate, not a hoisted register
    // XFRAG: k-block `wave` of a tile (16 fragments, 16 KB) straight into LDS[buf]: lane l's 16 bytes of fragment j land at
    // 1024 j + 16 l of the destination (the instruction offset advances source and destination alike); tiles beyond the end
    // lie outside the descriptor and write zeros
    auto dma_x = [&](int tile, int buf) {
        if constexpr (XFRAG) {
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                __attribute__((address_space(3))) char* dst = (__attribute__((address_space(3))) char*)(ldsw) + buf * kBufBytes + wave * kFragKbBytes + grp * 4 * kFragBytes;
                const int soff = (int)((unsigned)tile * (unsigned)kFragTileBytes) + wave * kFragKbBytes + grp * 4 * kFragBytes;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, dst, 16, lane * 16, soff, 0 * kFragBytes, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, dst, 16, lane * 16, soff, 1 * kFragBytes, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, dst, 16, lane * 16, soff, 2 * kFragBytes, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, dst, 16, lane * 16, soff, 3 * kFragBytes, 0);
            }
        }
    };
    auto load_x = [&](int tile, int ph) {
        if constexpr (XFRAG) return;
#pragma unroll
            for (int i = 0; i < XS; ++i) {
                const int cb = wave + NW * i;
                const unsigned voff = opaque(x_voff0 + ((unsigned)(tile * kTilePoints + 32 * ph) * (unsigned)p.ldx + 32u * cb) * 4u);
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    if (CB % NW == 0 || cb < CB) load_piece(x_rsrc, voff + 64u * g, xraw[i][ph][g]);
            }
    };
    auto load_g = [&](int tile, int ph) {
        if constexpr (GFRAG) {                 // tiles beyond the end lie outside the descriptor: zeros
            const unsigned voff = opaque((unsigned)tile * (unsigned)kFragTileBytes + (unsigned)lane * 16u);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    gfr[ph][q][j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, (int)(voff + frag_off(2 * ph + q, wave, j)), 0, 0));
                    // points 16 (4 tile + 2 ph + q) + 4 (lane >> 5) + 8 j + 0..3 (layout.h frag_point)
                    gsc[ph][q][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        s_rsrc, (int)opaque((unsigned)(tile * kTilePoints + 32 * ph + 16 * q + 4 * lh + 8 * j) * 4u), 0, 0));
                }
        } else {
            const unsigned voff = opaque(g_voff0 + (unsigned)(tile * kTilePoints + 32 * ph) * (unsigned)p.ldg * 4u);
#pragma unroll
            for (int g = 0; g < 2; ++g) load_piece(g_rsrc, voff + 64u * g, graw[ph][g]);
        }
    };

#ifdef INERF_WGRAD_STAMPS   // development build (scripts/build_variant.sh, scripts/wgrad_timeline.py): cycle stamps of workgroup 0's
    // wave 0 at the phase boundaries of its THIRD tile, kept in scalar registers (selects, no branches: branches around the
    // stamps changed the register allocation of the whole loop) and written over the start of the partial tile at the end
    unsigned long long wg_st[7] = {0, 0, 0, 0, 0, 0, 0};
#define WG_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); wg_st[k] = (tile == 2 * (int)gridDim.x) ? now_ : wg_st[k]; } while (0)
#else
#define WG_STAMP(k) do { } while (0)
#endif
    int buf = 0;
    load_x(blockIdx.x, 0); load_x(blockIdx.x, 1); dma_x(blockIdx.x, 0); load_g(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        WG_STAMP(0);
        // ---- X: every wave transposes its share of the column blocks and parks the operands in LDS[buf] ----
#pragma unroll
        for (int i = 0; i < (XFRAG ? 0 : XS); ++i) {
            const int cb = wave + NW * i;
            if (CB % NW == 0 || cb < CB) {
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
        // Requests are unconditional: behind the last tile the rows lie beyond the descriptors' range and cost nothing, and the
        // loop body stays one basic block (with branches around the loads the compiler parked the loaded rows in scratch).
        WG_STAMP(1);
        load_g(tile, 1);                           // this tile's second half of G: needed after the first half's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        // ---- G: this wave's row block, one 32-point half at a time: converted into registers, contracted against that half's X
        // operands.  The first half was requested a tile ago (its registers are refilled as soon as it is converted), the
        // second half's conversion sits behind the barrier between the two halves' MFMAs.
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            f16x8 gh[2], gl[2];                // [q]
            if constexpr (GFRAG) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                    {
                        const float spc = gsc[ph][q][i >> 2][i & 3] * (sg * (1.0f / kActScale));          // a power of two: both products exact
                        v[i] = __builtin_fmaf((float)gfr[ph][q][0][i], spc, (float)gfr[ph][q][1][i] * spc);
                    }
                    split8(v, 1.0f, gh[q], gl[q]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) bias_sum += v[i];            // (in units of 1 / sg, like the other form's)
                }
            } else {
                f16x8 h0, l0, h1, l1;
                split8(graw[ph][0], sg, h0, l0);
                split8(graw[ph][1], sg, h1, l1);
                bias_sum += transpose_block(h0, h1, id0, id1, gh);
                bias_sum += transpose_block(l0, l1, id0, id1, gl);
            }
            // The next tile's rows, in the order the next tile converts them: X's first half a whole contraction ahead, X's second
            // half and G's first half from the middle of this one (they are needed 500 / 1 000 cycles into the next tile).
            WG_STAMP(ph == 0 ? 2 : 5);
            if (ph == 0) {
                load_x(tile + gridDim.x, 0);
                if constexpr (XFRAG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's k-block of the tile has landed
                __syncthreads();               // LDS[buf] complete; LDS[buf ^ 1] (last read before the previous barrier) is free
                if constexpr (XFRAG) { asm volatile("" ::: "memory"); dma_x(tile + gridDim.x, buf ^ 1); }     // a whole contraction ahead
                WG_STAMP(3);
            } else {
                load_x(tile + gridDim.x, 1);
                load_g(tile + gridDim.x, 0);
            }
            // X operands one step ahead of their MFMAs, fenced: unfenced, the scheduler hoists ~20 operand reads (80 registers)
            // to the top of the half and spills the prefetched rows to make room
            f16x8 xh[2], xl[2];
            xh[0] = *reinterpret_cast<const f16x8*>(xfrag(buf, 0, ph, 0, 0));
            xl[0] = *reinterpret_cast<const f16x8*>(xfrag(buf, 0, ph, 0, 1));
#pragma unroll
            for (int st = 0; st < 2 * CB; ++st) {
                const int q = st / CB, cb = st % CB;
                if (st + 1 < 2 * CB) {
                    xh[(st + 1) & 1] = *reinterpret_cast<const f16x8*>(xfrag(buf, (st + 1) % CB, ph, (st + 1) / CB, 0));
                    xl[(st + 1) & 1] = *reinterpret_cast<const f16x8*>(xfrag(buf, (st + 1) % CB, ph, (st + 1) / CB, 1));
                }
                __builtin_amdgcn_sched_barrier(0);       // reads first (else they share registers with this step's operands and slip behind its MFMAs)
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[q], xh[st & 1], acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[q], xl[st & 1], acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[q], xh[st & 1], acc[cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            WG_STAMP(ph == 0 ? 4 : 6);
        }
        buf ^= 1;
    }

    // ---- this workgroup's partial tile: row m = channel of G, column n = channel of X ----
    const float back = 1.0f / (sg * sx);
    float* out = p.partial + (size_t)blockIdx.x * p.partial_stride;
#ifdef INERF_WGRAD_STAMPS
    if (blockIdx.x == 0) {
        if (lane == 0) {                       // every wave (waves w and w + 4 share a SIMD)
            unsigned long long* d = reinterpret_cast<unsigned long long*>(p.bias_partial ? p.bias_partial : p.partial) + 8 * wave;
            d[0] = 7;
            for (int i = 0; i < 7; ++i) d[1 + i] = wg_st[i];
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


// ---------------------------------------------------------------------------------------------------------------------
// Products of two FRAGMENT slots (described for the 256 x 256 shape; the narrower ones - wgrad_frag_body<GB, XB> - move fewer
// fragments per stage).
//
// Per 16-point k-block a 256-channel slot holds 16 fragments of 1 KB ([32-channel block][hi | lo], layout.h), contiguous: one LDS stage =
// 16 KB of G + 16 KB of X (+ the k-block's 16 normalisers).  A ring of kFragStages stages is filled by LDS-DMA -
// `buffer_load_dwordx4 ... lds` moves a fragment from HBM to LDS in one instruction, lane l's 16 bytes to byte 16 l of the
// destination, which is exactly where the lane that contracts reads its operand slot (conflict-free by construction) - so the
// loads need no registers and nothing is transposed.  X is consumed as it lands; G's fragments hold NORMALISED gradients
// (kActScale * dz / s_p: the chain's own f16 halves, 22 bits whatever the point's gradient scale), so a wave brings its own
// two G blocks to the product's scale first: hi + lo, times s_p and the power of two that puts the batch's max |dz| at 2^13,
// split again - ~50 VALU instructions per block and k-block against 24 MFMAs.  Up to kFragStages - 1 stages (96 KB per CU) are
// in flight while one is contracted.  Eight waves: wave w
// requests fragments 4 (w & 3) .. + 3 of G (w < 4) or X (w >= 4) of every stage and contracts rows 32 w .. + 31 x all 256 columns
// (1 x 8 accumulator blocks: 18 operand reads and 24 MFMAs per k-block) - every G block is rescaled by exactly one wave.
// Synchronisation per stage: every wave waits for ITS OWN requests of the stage (counted vmcnt: the younger stages stay in
// flight), then one raw s_barrier - behind it every wave's fragments of the stage have landed, and every wave has finished
// reading the stage before, whose buffer is the one re-filled next.
// The column sums of G (the bias gradient) come from the operands the waves 0..3 read anyway.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef INERF_WGRAD_MAX_BATCH
#define INERF_WGRAD_MAX_BATCH 16
#endif
#ifndef INERF_WGRAD_FRAG_STAGES
#define INERF_WGRAD_FRAG_STAGES 4
#endif
constexpr int kFragStages = INERF_WGRAD_FRAG_STAGES;
#ifndef INERF_WGRAD_DMA_AUX
#define INERF_WGRAD_DMA_AUX 2
#endif
constexpr int kDmaAux = INERF_WGRAD_DMA_AUX;                    // cache policy of the fragment stream: 2 = nt (every byte is read once, by one CU;
                                                                // same-box A/B against the default policy: 1.686 vs 1.705 ms per training step)
constexpr int kFragStageBytes = 2 * kFragKbBytes;               // G | X of one k-block
constexpr int kFragScaleBytes = 8 * 256;                        // per stage: every wave's own copy of the k-block's normalisers (64 lanes x 4 bytes)
static_assert(kFragStages >= 3 && kFragStages <= 5, "ring depth");

// SEVERAL products over the same sample points share one launch (all of a network's whose operands are fragment slots): every
// workgroup works on ONE product, as one of its K-slices - a product is split over its share of the grid (in proportion to
// what a stage of it costs) instead of the whole grid, so the chip writes (and the final sum reads) ~n_jobs
// times fewer partial tiles - 7 MB per 256 x 256 product instead of 64 MB at eleven jobs, which was a tenth of the operand bytes
// themselves - fills its ring once instead of n_jobs times, and one launch boundary replaces n_jobs.  Every workgroup still
// streams whole stages of two contiguous matrices.  Workgroup b -> (job, slice): round-robin over the jobs that still have
// slices left (neighbouring workgroups stream different matrices; contiguous shares measured slower in round 2).
// X of a job is 256 channels wide (8 channel blocks: 16 KB per k-block) or 64 (the encoding: 2 blocks, 4 KB).
constexpr int kMaxFragJobs = INERF_WGRAD_MAX_BATCH;
struct WgradFragParams {
    const void* G[kMaxFragJobs];        // fragment slot of dZ: f16 hi / lo of kActScale * dz / s_p
    const void* X[kMaxFragJobs];        // fragment slot of activations: f16 hi / lo of kActScale * h
    float* partial[kMaxFragJobs];       // K-slice s of the job writes its 256 x N tile at partial + s * partial_stride
    float* bias_partial[kMaxFragJobs];  // optional: ... its column sums of G
    const float* g_scale;    // the points' normalisers s_p: 64 * n_tiles (+ 64 readable) floats
    const float* ranges;     // device: {gmax, ...}: upper bound of |dz|
    int64_t partial_stride;  // floats
    int n_kb;                // 16-point k-blocks: 4 per tile
    int n_jobs;
    uint16_t slices[kMaxFragJobs];      // K-slices (= workgroups) of every job; their sum is the grid
    uint8_t g_blocks[kMaxFragJobs];     // 32-channel blocks of G: 8 | 4
    uint8_t x_blocks[kMaxFragJobs];     // ... of X: 8 | 2 | 1; supported pairs: (8, 8) (8, 2) (4, 8) (4, 1)
};

// GB / XB: 32-channel blocks of G (8 | 4) and X (8 | 2 | 1).  Roles of the eight waves:
//   DMA        waves 0..3 request G's fragments of the stage (GB / 2 each), waves 4..7 X's (XB / 2 each; XB = 1: one each for
//              waves 4 and 5), every wave its own copy of the k-block's normalisers;
//   contract   GB = 8: wave w = row block w x all column blocks; GB = 4: wave w = row block (w & 3) x column half (w >> 2).
template <int GB, int XB>
__device__ __forceinline__ void wgrad_frag_body(const WgradFragParams& p, char* ldsw, const int job, const int b /* slice */, const int g /* slices */) {
    static_assert((GB == 8 || GB == 4) && (XB == 8 || XB == 2 || XB == 1), "shapes");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 31, lh = lane >> 5;
    const bool dma_g = wave < 4;
    constexpr int kGKbBytes = GB * 2 * kFragBytes, kXKbBytes = XB * 2 * kFragBytes;        // bytes per k-block
    constexpr int kXPer = XB >= 2 ? XB / 2 : 1;
    const int n_frag = dma_g ? GB / 2 : (XB >= 2 ? XB / 2 : ((wave & 3) < 2 ? 1 : 0));    // fragments this wave requests per stage
    const int n_req = n_frag + 1;                                                          // ... + the normalisers
    float* const bias_partial = (GB == 8 || wave < 4) ? p.bias_partial[job] : nullptr;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dma_g ? p.G[job] : p.X[job]), 0,
        (int)((unsigned)p.n_kb * (unsigned)(dma_g ? kGKbBytes : kXKbBytes)), 0x00020000);
    const int src_bytes = (wave & 3) * (dma_g ? GB / 2 : kXPer) * kFragBytes;            // this wave's fragments inside its matrix's k-block
    const int my_bytes = (dma_g ? 0 : kFragKbBytes) + src_bytes;                         // ... inside a stage (G | X at 16 KB)
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g_scale), 0, (p.n_kb * 16 + 64) * 4, 0x00020000);
    char* const scale_lds = ldsw + kFragStages * kFragStageBytes + wave * 256;          // + buf * kFragScaleBytes
    // one stage: this wave's fragment requests - lane l's 16 bytes of fragment j land at stage + my_bytes + 1024 j + 16 l (the
    // instruction offset advances the source and the destination alike) - and one for the k-block's normalisers (lane l's float
    // = point 16 kb + l; 16 of the 64 are used): every wave fetches its OWN copy, so that nothing but its own counter orders them
    auto request = [&](int kb, int buf) {
        __attribute__((address_space(3))) char* dst = (__attribute__((address_space(3))) char*)(ldsw) + buf * kFragStageBytes + my_bytes;
        const int soff = (int)((unsigned)kb * (unsigned)(dma_g ? kGKbBytes : kXKbBytes)) + src_bytes;
        if (n_frag > 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, lane * 16, soff, 0 * kFragBytes, kDmaAux);
        if (n_frag > 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, lane * 16, soff, 1 * kFragBytes, kDmaAux);
        if (n_frag > 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, lane * 16, soff, 2 * kFragBytes, kDmaAux);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, lane * 16, soff, 3 * kFragBytes, kDmaAux);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (__attribute__((address_space(3))) char*)(scale_lds) + buf * kFragScaleBytes, 4, lane * 4, kb * 64, 0, 0);
    };
    // wait until at most `stages` of this wave's younger stages are still in flight (n_req requests each, returned in order)
    auto wait_for = [&](int stages) {
        switch (stages * n_req) {
            case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 9:  asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 6:  asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 5:  asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 4:  asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 3:  asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 2:  asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 1:  asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    constexpr int NC = GB == 8 ? XB : (XB >= 2 ? XB / 2 : 1);         // column blocks this wave contracts
    const int rbk = GB == 8 ? wave : (wave & 3);                      // its row block
    const int cb0 = GB == 8 ? 0 : (wave >> 2) * NC;                   // its first column block
    const bool contracts = GB == 8 || XB >= 2 || wave < 4;
    f32x16 acc[NC];
#pragma unroll
    for (int cb = 0; cb < NC; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
    float bias_sum = 0.0f;                     // this lane's channel of the wave's row block, its k-half's points (units of 1 / sg)
    auto pow2_for = [](float m) { int e; if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f; frexpf(m, &e); return ldexpf(1.0f, 14 - e); };
    const float sg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pow2_for(p.ranges[0]))));
    const float sgc = sg * (1.0f / kActScale);
    auto contract = [&](int buf) {
        const char* gset = ldsw + buf * kFragStageBytes + lane * 16;
        const char* xset = gset + kFragKbBytes + cb0 * 2 * kFragBytes;
        auto frag = [&](const char* set, int block, int plane) { return *reinterpret_cast<const f16x8*>(set + (block * 2 + plane) * kFragBytes); };
        f16x8 gh, gl, xh[2], xl[2];
        const f16x8 g16h = frag(gset, rbk, 0), g16l = frag(gset, rbk, 1);
        // the normalisers of this lane's eight points: 16 kb + 4 (lane >> 5) + 0..3 and + 8..11 (layout.h frag_point)
        const float* sc = reinterpret_cast<const float*>(scale_lds + buf * kFragScaleBytes) + 4 * lh;
        const f32x4 s03 = *reinterpret_cast<const f32x4*>(sc), s47 = *reinterpret_cast<const f32x4*>(sc + 8);
        xh[0] = frag(xset, 0, 0); xl[0] = frag(xset, 0, 1);
        {   // this wave's G block at the product's scale: (hi + lo) x s_p x 2^k - both products exact (powers of two), one rounding
            // that does nothing (22 bits) - then the usual split
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float spc = (i < 4 ? s03[i & 3] : s47[i & 3]) * sgc;
                v[i] = __builtin_fmaf((float)g16h[i], spc, (float)g16l[i] * spc);
            }
            split8(v, 1.0f, gh, gl);
            if (bias_partial) {
#pragma unroll
                for (int i = 0; i < 8; ++i) bias_sum += v[i];
            }
        }
#pragma unroll
        for (int cb = 0; cb < NC; ++cb) {    // X operands one read ahead of their MFMAs, fenced (unfenced, the scheduler hoists every read to the top)
            if (cb + 1 < NC) { xh[(cb + 1) & 1] = frag(xset, cb + 1, 0); xl[(cb + 1) & 1] = frag(xset, cb + 1, 1); }
            __builtin_amdgcn_sched_barrier(0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xh[cb & 1], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xl[cb & 1], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, xh[cb & 1], acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // this workgroup's k-blocks: its slice number, + the job's slice count, ... (the workgroups of a job stream neighbouring
    // stages: the chip walks every matrix front to back)
    const int n_mine = p.n_kb > b ? (p.n_kb - b + g - 1) / g : 0;
#pragma unroll
    for (int j = 0; j < kFragStages - 1; ++j)
        if (j < n_mine) request(b + j * g, j);
    int buf = 0;
    for (int i = 0; i < n_mine; ++i) {
        // requests of stages i + 1 .. may stay in flight: at most kFragStages - 2 stages, fewer at the end
        const int ahead = n_mine - 1 - i;
        wait_for(ahead < kFragStages - 2 ? ahead : kFragStages - 2);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + kFragStages - 1 < n_mine) request(b + (i + kFragStages - 1) * g, buf == 0 ? kFragStages - 1 : buf - 1);
        if (contracts) contract(buf);
        buf = buf + 1 == kFragStages ? 0 : buf + 1;
    }

    // ---- this slice's partial tile and its column sums of G ----
    if (bias_partial) {                        // the two lane halves hold complementary points of the same channel
        const float both = bias_sum + __shfl_xor(bias_sum, 32);
        if (lh == 0) bias_partial[(size_t)b * p.partial_stride + 32 * rbk + lp] = both / sg;
    }
    if (!contracts) return;
    const float back = 1.0f / (sg * kActScale);
    float* out = p.partial[job] + (size_t)b * p.partial_stride;
#pragma unroll
    for (int cb = 0; cb < NC; ++cb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = 32 * rbk + (j & 3) + 8 * (j >> 2) + 4 * lh;
            out[(size_t)m * (32 * XB) + 32 * (cb0 + cb) + lp] = acc[cb][j] * back;
        }
}

__global__ __launch_bounds__(512, 1) void k_mlp_wgrad_frag(const WgradFragParams p) {
    extern __shared__ __attribute__((aligned(16))) char ldsw[];
    // workgroup -> (job, slice): round r hands one slice to every job that has more than r
    int rem = blockIdx.x, round = 0, job = 0;
    for (;;) {
        int live = 0;
        for (int j = 0; j < p.n_jobs; ++j) live += p.slices[j] > round ? 1 : 0;
        if (rem < live || live == 0) break;
        rem -= live;
        ++round;
    }
    for (int j = 0; j < p.n_jobs; ++j)
        if (p.slices[j] > round) {
            if (rem == 0) { job = j; break; }
            --rem;
        }
    job = __builtin_amdgcn_readfirstlane(job);
    round = __builtin_amdgcn_readfirstlane(round);
    const int n_slices = p.slices[job];
    const int shape = p.g_blocks[job] * 16 + p.x_blocks[job];
    if (shape == 8 * 16 + 8)      wgrad_frag_body<8, 8>(p, ldsw, job, round, n_slices);
    else if (shape == 8 * 16 + 2) wgrad_frag_body<8, 2>(p, ldsw, job, round, n_slices);
    else if (shape == 4 * 16 + 8) wgrad_frag_body<4, 8>(p, ldsw, job, round, n_slices);
    else                          wgrad_frag_body<4, 1>(p, ldsw, job, round, n_slices);
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

namespace inerf {
namespace {

int launch_rows(WgradParams& p, bool gfrag, void* stream, bool xfrag = false) {
    const int grid = inerf_wgrad_grid(p.n_points);
    const int M = p.M, cb = p.N / 32;
    if ((M != 128 && M != 256) || p.N % 32 || cb < 1 || cb > 8) return INERF_E_UNSUPPORTED;
    // rows are addressed through 32-bit buffer descriptors, the prefetch reaches one grid stride of tiles beyond the end
    const int64_t ld = gfrag ? p.ldx : xfrag ? p.ldg : (p.ldg > p.ldx ? p.ldg : p.ldx);
    if (((int64_t)p.n_points + (int64_t)kTilePoints * (grid + 1)) * ld * 4 >= (int64_t)1 << 32) return INERF_E_UNSUPPORTED;
    const int lds = 2 * cb * 8 * kWgFragBytes;          // double-buffered X operands
    void (*kern)(const WgradParams) = nullptr;
    int variant = -1;
    if (gfrag) {
        if (M == 256 && cb == 2) { kern = k_mlp_wgrad<8, 2, true>; variant = 8; }
    } else if (xfrag) {
        if (M == 128 && cb == 8) { kern = k_mlp_wgrad<4, 8, false, true>; variant = 9; }
    } else {
#define INERF_WG_CASE(V, NWV, CBV) if (M == 32 * NWV && cb == CBV) { kern = k_mlp_wgrad<NWV, CBV>; variant = V; }
        INERF_WG_CASE(0, 8, 8) INERF_WG_CASE(1, 8, 2) INERF_WG_CASE(2, 4, 8) INERF_WG_CASE(3, 4, 1) INERF_WG_CASE(4, 8, 1) INERF_WG_CASE(5, 4, 2)
        INERF_WG_CASE(6, 8, 4) INERF_WG_CASE(7, 4, 4)
#undef INERF_WG_CASE
    }
    if (!kern) return INERF_E_UNSUPPORTED;
    static PerDeviceOnce attr_set[10];
    if (lds > 64 * 1024 && attr_set[variant].first()) {          // only <8, 8> and the two <4, 8>
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return record(e);
        attr_set[variant].mark();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(M * 2), lds, (hipStream_t)stream, p);       // 64 threads per 32-row block
    return record(hipGetLastError());
}

}  // namespace
}  // namespace inerf

extern "C" int inerf_mlp_weight_gradient(const float* G, int ldg, const float* X, int ldx, int64_t n_points, int M, int N,
                                         const float* ranges, float* partial, float* bias_partial, int64_t partial_stride,
                                         void* stream) {
    using namespace inerf;
    if (!G || !X || !ranges || !partial || n_points <= 0 || ldg < M || ldx < N) return INERF_E_INVALID;
    if (n_points > kMaxTrainPoints) return INERF_E_UNSUPPORTED;
    if ((ldg & 3) || (ldx & 3) || (((uintptr_t)G | (uintptr_t)X) & 15)) return INERF_E_INVALID;      // 16-byte row pieces
    if (partial_stride < (int64_t)M * N) return INERF_E_INVALID;
    WgradParams p;
    p.G = G; p.g_scale = nullptr; p.X = X; p.ranges = ranges; p.partial = partial; p.bias_partial = bias_partial; p.partial_stride = partial_stride;
    p.ldg = ldg; p.ldx = ldx; p.n_points = (int)n_points; p.M = M; p.N = N;
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    return launch_rows(p, false, stream);
}

// G: a FRAGMENT slot of the gradient buffer (256 channels) and the points' normalisers (slot 0 of that buffer); X: rows.
extern "C" int inerf_mlp_weight_gradient_gfrag(const void* G_frag, const float* g_scale, const float* X, int ldx, int64_t n_points, int N,
                                               const float* ranges, float* partial, float* bias_partial, int64_t partial_stride,
                                               void* stream) {
    using namespace inerf;
    if (!G_frag || !g_scale || !X || !ranges || !partial || n_points <= 0 || ldx < N) return INERF_E_INVALID;
    if (n_points > kMaxTrainPoints) return INERF_E_UNSUPPORTED;
    if ((ldx & 3) || (((uintptr_t)G_frag | (uintptr_t)X | (uintptr_t)g_scale) & 15)) return INERF_E_INVALID;
    if (partial_stride < (int64_t)kWidth * N) return INERF_E_INVALID;
    WgradParams p;
    p.G = static_cast<const float*>(G_frag); p.g_scale = g_scale; p.X = X; p.ranges = ranges; p.partial = partial; p.bias_partial = bias_partial;
    p.partial_stride = partial_stride;
    p.ldg = kWidth; p.ldx = ldx; p.n_points = (int)n_points; p.M = kWidth; p.N = N;
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    return launch_rows(p, true, stream);
}

// G: rows (128 channels); X: a FRAGMENT slot of the activation buffer (256 channels).  ranges: device {gmax, ...}.
extern "C" int inerf_mlp_weight_gradient_xfrag(const float* G, int ldg, const void* X_frag, int64_t n_points, int M,
                                               const float* ranges, float* partial, float* bias_partial, int64_t partial_stride,
                                               void* stream) {
    using namespace inerf;
    if (!G || !X_frag || !ranges || !partial || n_points <= 0 || ldg < M) return INERF_E_INVALID;
    if (n_points > kMaxTrainPoints) return INERF_E_UNSUPPORTED;
    if ((ldg & 3) || (((uintptr_t)G | (uintptr_t)X_frag) & 15)) return INERF_E_INVALID;
    if (partial_stride < (int64_t)M * kWidth) return INERF_E_INVALID;
    WgradParams p;
    p.G = G; p.g_scale = nullptr; p.X = static_cast<const float*>(X_frag); p.ranges = ranges; p.partial = partial; p.bias_partial = bias_partial;
    p.partial_stride = partial_stride;
    p.ldg = ldg; p.ldx = kWidth; p.n_points = (int)n_points; p.M = M; p.N = kWidth;
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    return launch_rows(p, false, stream, true);
}

// Both operands FRAGMENT slots: G of the gradient buffer (g_rows[j] = 256, or 128 for the views hidden layer's slot) with the
// points' normalisers, X of the activation buffer (x_cols[j] = 256, 64 for the position encoding's slot, 32 for the view
// encoding's; NULL: all 256; supported pairs 256 x 256, 256 x 64, 128 x 256, 128 x 32), on the same n_points.  ranges: device {gmax, ...}: an upper
// bound of |dz| (the dz_max of inerf_mlp_backward_inputs).  n_jobs products in one launch of
// inerf_wgrad_frag_grid(n_points, n_jobs) workgroups: job j is split over inerf_wgrad_frag_rows(n_points, n_jobs, g_rows, x_cols, j)
// K-slices, slice s writing its [g_rows[j], x_cols[j]] tile at partial[j] + s * partial_stride.
extern "C" int inerf_wgrad_frag_grid(int64_t n_points, int n_jobs) {
    const int g = inerf_wgrad_grid(n_points);
    return g < n_jobs ? n_jobs : g;
}

namespace inerf {
namespace {
bool frag_shape_ok(int rows, int cols) {
    return (rows == kWidth && (cols == kWidth || cols == kEncCols)) || (rows == kHalf && (cols == kWidth || cols == kDirCols));
}

// the grid divided among the jobs in proportion to what a k-block of each costs - the KB it moves (2 per 32-channel block of G
// and of X) plus a fixed 8 for the stage's barrier, waits and re-scaling (a 256 x 64 product's stage takes ~0.7 of a 256 x 256
// one's, not 0.625): largest remainders, at least one slice each
bool frag_slices(int64_t n_points, int n_jobs, const int* g_rows, const int* x_cols, uint16_t* slices) {
    if (n_jobs < 1 || n_jobs > kMaxFragJobs || n_points <= 0) return false;
    const int grid = inerf_wgrad_frag_grid(n_points, n_jobs);
    int64_t w[kMaxFragJobs], total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int r = g_rows ? g_rows[j] : kWidth, c = x_cols ? x_cols[j] : kWidth;
        if (!frag_shape_ok(r, c)) return false;
        w[j] = 8 + 2 * (r / 32) + 2 * (c / 32);
#ifdef INERF_WGRAD_TUNE_SHARES          // development build: weights of the four shapes from $INERF_WGRAD_WEIGHTS = "w88,w82,w48,w41"
        if (const char* e = getenv("INERF_WGRAD_WEIGHTS")) {
            int w88, w82, w48, w41;
            if (sscanf(e, "%d,%d,%d,%d", &w88, &w82, &w48, &w41) == 4) w[j] = r == kWidth ? (c == kWidth ? w88 : w82) : (c == kWidth ? w48 : w41);
        }
#endif
        total += w[j];
    }
    int given = 0;
    int64_t frac[kMaxFragJobs];
    const int spare = grid - n_jobs;                       // one slice each up front
    for (int j = 0; j < n_jobs; ++j) {
        const int64_t num = (int64_t)spare * w[j];
        slices[j] = (uint16_t)(1 + num / total);
        frac[j] = num % total;
        given += slices[j];
    }
    while (given < grid) {                                 // (first job wins ties: deterministic)
        int best = 0;
        for (int j = 1; j < n_jobs; ++j)
            if (frac[j] > frac[best]) best = j;
        ++slices[best]; frac[best] = -1; ++given;
    }
    return true;
}
}  // namespace
}  // namespace inerf

extern "C" int inerf_wgrad_frag_rows(int64_t n_points, int n_jobs, const int* g_rows, const int* x_cols, int job) {
    uint16_t slices[inerf::kMaxFragJobs];
    if (job < 0 || job >= n_jobs || !inerf::frag_slices(n_points, n_jobs, g_rows, x_cols, slices)) return 0;
    return slices[job];
}

extern "C" int inerf_mlp_weight_gradient_frag_batch(int n_jobs, const void* const* G_frag, const float* g_scale, const void* const* X_frag,
                                                    const int* g_rows, const int* x_cols, const float* ranges, int64_t n_points,
                                                    float* const* partial, float* const* bias_partial, int64_t partial_stride, void* stream) {
    using namespace inerf;
    if (n_jobs < 1 || n_jobs > kMaxFragJobs) return INERF_E_INVALID;
    if (!G_frag || !g_scale || !X_frag || !ranges || !partial || n_points <= 0) return INERF_E_INVALID;
    if (n_points > kMaxTrainPoints) return INERF_E_UNSUPPORTED;
    if (((uintptr_t)g_scale & 3)) return INERF_E_INVALID;
    WgradFragParams p{};
    if (!frag_slices(n_points, n_jobs, g_rows, x_cols, p.slices)) return INERF_E_INVALID;
    for (int j = 0; j < n_jobs; ++j) {
        const int rows = g_rows ? g_rows[j] : kWidth, cols = x_cols ? x_cols[j] : kWidth;
        if (!G_frag[j] || !X_frag[j] || !partial[j] || (((uintptr_t)G_frag[j] | (uintptr_t)X_frag[j]) & 15)) return INERF_E_INVALID;
        if (partial_stride < (int64_t)rows * cols) return INERF_E_INVALID;
        p.G[j] = G_frag[j]; p.X[j] = X_frag[j]; p.partial[j] = partial[j]; p.bias_partial[j] = bias_partial ? bias_partial[j] : nullptr;
        p.g_blocks[j] = (uint8_t)(rows / 32); p.x_blocks[j] = (uint8_t)(cols / 32);
    }
    p.g_scale = g_scale; p.ranges = ranges; p.partial_stride = partial_stride; p.n_jobs = n_jobs;
    p.n_kb = (int)((n_points + kTilePoints - 1) / kTilePoints) * 4;
    const int grid = inerf_wgrad_frag_grid(n_points, n_jobs);
    constexpr int lds = kFragStages * (kFragStageBytes + kFragScaleBytes);
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_wgrad_frag), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return record(e);
        attr_set.mark();
    }
    hipLaunchKernelGGL(k_mlp_wgrad_frag, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
    return record(hipGetLastError());
}

// one product: inerf_wgrad_grid(n_points) K-slices
extern "C" int inerf_mlp_weight_gradient_frag(const void* G_frag, const float* g_scale, const void* X_frag, const float* ranges,
                                              int64_t n_points, float* partial, float* bias_partial, int64_t partial_stride, void* stream) {
    return inerf_mlp_weight_gradient_frag_batch(1, &G_frag, g_scale, &X_frag, nullptr, nullptr, ranges, n_points, &partial, &bias_partial, partial_stride, stream);
}
