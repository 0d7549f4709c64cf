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
#include <cstdlib>

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

    // LDS: X operands of two tiles (double buffer): [buf][cb][ph][q][plane] fragments
    constexpr int kBufBytes = CB * 8 * kWgFragBytes;
    auto xfrag = [&](int buf, int cb, int ph, int q, int plane) {
        return ldsw + buf * kBufBytes + ((((cb * 2 + ph) * 2 + q) * 2 + plane) * kWgFragBytes) + lane * 16;
    };
    constexpr int XS = (CB + NW - 1) / NW;        // column blocks of X this wave converts (cb = wave + NW * i)

    // The next tile's rows - this wave's share of X and its own 32 channels of G - are requested before the contraction and
    // converted after it: a tile's loads have a whole contraction (~4 000 cycles) to arrive.  (Until round 3 G was requested
    // at the top of its own tile and waited for - ~3 000 exposed cycles of a tile's 20 000.)  One barrier per tile.
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, (int)((unsigned)p.n_points * (unsigned)p.ldg * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)((unsigned)p.n_points * (unsigned)p.ldx * 4u), 0x00020000);
    const unsigned g_voff0 = (unsigned)(lp * p.ldg + 32 * wave + 8 * lh) * 4u, x_voff0 = (unsigned)(lp * p.ldx + 8 * lh) * 4u;
    float xraw[XS][2][2][8], graw[2][2][8];
    auto opaque = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };      // keeps `voff + constant` an immediate, not a hoisted register
    auto load_x = [&](int tile, int ph) {
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
        const unsigned voff = opaque(g_voff0 + (unsigned)(tile * kTilePoints + 32 * ph) * (unsigned)p.ldg * 4u);
#pragma unroll
        for (int g = 0; g < 2; ++g) load_piece(g_rsrc, voff + 64u * g, graw[ph][g]);
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
    load_x(blockIdx.x, 0); load_x(blockIdx.x, 1); load_g(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        WG_STAMP(0);
        // ---- X: every wave transposes its share of the column blocks and parks the operands in LDS[buf] ----
#pragma unroll
        for (int i = 0; i < XS; ++i) {
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
            {
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
                __syncthreads();               // LDS[buf] complete; LDS[buf ^ 1] (last read before the previous barrier) is free
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
// 256 x 256 products (nine of a network's thirteen, 90 % of the weight-gradient time): ROW-COALESCED form.
//
// k_mlp_wgrad above reads its rows with lane = point (the layout the transposing MFMA wants): every request fetches 16 bytes from
// each of 32 rows, and the cycle stamps (profiles/r03_wgrad_timeline.txt) show the vector-memory front end saturated by them.
// Here a request is one whole row - 64 lanes x 16 bytes = the 1 KB of a point's 256 channels - and no MFMA transposes anything:
// a lane that has fetched the same four channels of EIGHT consecutive points holds, per channel, exactly the 8 k-values of one
// operand slot (lane = channel, 8 consecutive points of a 16-point k-block).  Both matrices go through LDS as operand fragments:
// waves 0..3 stage G (8 of a step's 32 points each), waves 4..7 stage X; every wave then contracts a [64 rows x 128 columns]
// part of the tile (2 x 4 accumulator blocks, 24 KB of operand reads per step instead of 36).  Per 32-point step and wave:
// 8 requests, 16 three-instruction splits, 8 conflict-free ds_write_b128, 48 MFMAs.
// The point -> (k-block, k-half, element) assignment is the same for G and X, so the contraction does not see it.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kRowStep = 32;                                    // points per step
constexpr int kRowSetBytes = 8 * 2 * 2 * kWgFragBytes;          // one matrix, one step: [32-channel block][k-block][hi | lo] fragments
constexpr int kRowBufBytes = 2 * kRowSetBytes;                  // G | X

__global__ __launch_bounds__(512, 1) void k_mlp_wgrad_rows(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char ldsw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 31, lh = lane >> 5;
    const bool is_g = wave < 4;                 // the matrix this wave stages
    const int w4 = wave & 3;                    // ... and which 8 of a step's 32 points
    auto pow2_for = [](float m) { int e; if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f; frexpf(m, &e); return ldexpf(1.0f, 14 - e); };
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float sg = uniform(pow2_for(p.ranges[0])), sx = uniform(pow2_for(p.ranges[1]));
    const float s_mine = is_g ? sg : sx;
    const int ld = is_g ? p.ldg : p.ldx;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(is_g ? p.G : p.X), 0,
                                                                           (int)((unsigned)p.n_points * (unsigned)ld * 4u), 0x00020000);
    const unsigned row_bytes = (unsigned)ld * 4u;
    const unsigned voff0 = (unsigned)(8 * w4) * row_bytes + 16u * lane;       // the whole offset goes through the VGPR (range check)
    auto opaque = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };
    // rows 32 step + 8 w4 + j, j = 0..7; beyond the end of the matrix they read as zeros (and cost nothing)
    auto load_rows = [&](int step, f32x4 (&raw)[8]) {
        const unsigned v = opaque(voff0 + (unsigned)(step * kRowStep) * row_bytes);
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(v + j * row_bytes), 0, 0));
    };
    // Operand slot (16 bytes = 8 points of one channel) of channel n (0..31 of its block) and k-half kh inside a 1 KB fragment:
    // (n >> 2) + 8 (n & 3) + 32 kh.  A staging lane owns channels 4 (lane & 7) .. + 3 of block lane >> 3: for a given one of its
    // four channels the 8 lanes of a block write 128 consecutive bytes and the 8 blocks 8 different fragments - every ds_write_b128
    // is conflict-free.  A contracting lane (n = lane & 31, kh = lane >> 5) reads its own slot: 64 different slots of one fragment.
    float bias4[4] = {0.0f, 0.0f, 0.0f, 0.0f};     // G waves: sums over this workgroup's points of the lane's four channels
    auto convert = [&](int buf, const f32x4 (&raw)[8]) {
        char* set = ldsw + buf * kRowBufBytes + (is_g ? 0 : kRowSetBytes);
        const int q = w4 >> 1, kh = w4 & 1;        // the 8 points 8 w4 .. + 7 of the step: k-block q, k-half kh, elements 0..7
        char* dst0 = set + (((lane >> 3) * 2 + q) * 2) * kWgFragBytes + ((lane & 7) + 32 * kh) * 16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f16x8 hi, lo;
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                f16x2 h2, l2;
                split_pair(raw[j][e] * s_mine, raw[j + 1][e] * s_mine, h2, l2);
                hi[j] = h2[0]; hi[j + 1] = h2[1];
                lo[j] = l2[0]; lo[j + 1] = l2[1];
                sum += raw[j][e] + raw[j + 1][e];
            }
            bias4[e] += sum;
            *reinterpret_cast<f16x8*>(dst0 + 8 * e * 16) = hi;
            *reinterpret_cast<f16x8*>(dst0 + 8 * e * 16 + kWgFragBytes) = lo;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // this wave's part of the tile: rows 64 (wave & 3) .. + 63, columns 128 (wave >> 2) .. + 127
    const int rb0 = 2 * (wave & 3), cb0 = 4 * (wave >> 2);
    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.0f;
    const int my_slot = ((lp >> 2) + 8 * (lp & 3) + 32 * lh) * 16;
    auto frag = [&](const char* set, int block, int q, int plane) {
        return *reinterpret_cast<const f16x8*>(set + ((block * 2 + q) * 2 + plane) * kWgFragBytes + my_slot);
    };
    auto contract = [&](int buf) {
        const char* gset = ldsw + buf * kRowBufBytes;
        const char* xset = gset + kRowSetBytes;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f16x8 gh[2], gl[2], xh[2], xl[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) { gh[rb] = frag(gset, rb0 + rb, q, 0); gl[rb] = frag(gset, rb0 + rb, q, 1); }
            xh[0] = frag(xset, cb0, q, 0); xl[0] = frag(xset, cb0, q, 1);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {     // X operands one read ahead of their MFMAs, fenced (see k_mlp_wgrad)
                if (cb + 1 < 4) { xh[(cb + 1) & 1] = frag(xset, cb0 + cb + 1, q, 0); xl[(cb + 1) & 1] = frag(xset, cb0 + cb + 1, q, 1); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[rb], xh[cb & 1], acc[rb][cb], 0, 0, 0);
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[rb], xl[cb & 1], acc[rb][cb], 0, 0, 0);
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[rb], xh[cb & 1], acc[rb][cb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // this workgroup's steps: blockIdx.x, + gridDim.x, ... of ceil(n_points / 32); rows are converted one step before they are
    // contracted (two LDS buffers, one barrier per step) and requested one step before they are converted - two steps with
    // INERF_WGRAD_ROWS_DEPTH2 (a second register set: 32 more registers, spills)
    const int n_steps = (p.n_points + kRowStep - 1) / kRowStep;
    const int g = gridDim.x;
#ifdef INERF_WGRAD_ROWS_DEPTH2
    f32x4 raw_a[8], raw_b[8];
    load_rows(blockIdx.x, raw_a);
    load_rows(blockIdx.x + g, raw_b);
    convert(0, raw_a);
    load_rows(blockIdx.x + 2 * g, raw_a);
    for (int s = blockIdx.x; s < n_steps; s += 2 * g) {
        __syncthreads();               // LDS[0] holds step s; nobody reads LDS[1] any more
        contract(0);
        convert(1, raw_b);             // step s + g
        load_rows(s + 3 * g, raw_b);
        __syncthreads();               // LDS[1] holds step s + g (zeros beyond the end); nobody reads LDS[0] any more
        contract(1);
        convert(0, raw_a);             // step s + 2 g
        load_rows(s + 4 * g, raw_a);
    }
#else
    f32x4 raw[8];
    load_rows(blockIdx.x, raw);
    convert(0, raw);
    load_rows(blockIdx.x + g, raw);
    for (int s = blockIdx.x; s < n_steps; s += 2 * g) {
        __syncthreads();               // LDS[0] holds step s; nobody reads LDS[1] any more
        contract(0);
        convert(1, raw);               // step s + g
        load_rows(s + 2 * g, raw);
        __syncthreads();               // LDS[1] holds step s + g (zeros beyond the end); nobody reads LDS[0] any more
        contract(1);
        convert(0, raw);               // step s + 2 g
        load_rows(s + 3 * g, raw);
    }
#endif

    // ---- this workgroup's partial tile and its column sums of G ----
    if (p.bias_partial) {              // the four G waves hold different points of the same 256 channels
        __syncthreads();
        float* sums = reinterpret_cast<float*>(ldsw);
        if (is_g) *reinterpret_cast<f32x4*>(sums + w4 * 256 + 4 * lane) = f32x4{bias4[0], bias4[1], bias4[2], bias4[3]};
        __syncthreads();
        if (tid < 256) p.bias_partial[(size_t)blockIdx.x * p.partial_stride + tid] = (sums[tid] + sums[256 + tid]) + (sums[512 + tid] + sums[768 + tid]);
    }
    const float back = 1.0f / (sg * sx);
    float* out = p.partial + (size_t)blockIdx.x * p.partial_stride;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = 32 * (rb0 + rb) + (j & 3) + 8 * (j >> 2) + 4 * lh;
                out[(size_t)m * 256 + 32 * (cb0 + cb) + lp] = acc[rb][cb][j] * back;
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
    // rows are addressed through 32-bit buffer descriptors, the prefetch reaches one grid stride of tiles beyond the end
    if ((n_points + (int64_t)kTilePoints * (device_cus() + 1)) * (ldg > ldx ? ldg : ldx) * 4 >= (int64_t)1 << 32) return INERF_E_UNSUPPORTED;
    if ((ldg & 3) || (ldx & 3) || (((uintptr_t)G | (uintptr_t)X) & 15)) return INERF_E_INVALID;      // 16-byte row pieces
    WgradParams p;
    if (partial_stride < (int64_t)M * N) return INERF_E_INVALID;
    p.G = G; p.X = X; p.ranges = ranges; p.partial = partial; p.bias_partial = bias_partial; p.partial_stride = partial_stride;
    p.ldg = ldg; p.ldx = ldx; p.n_points = (int)n_points; p.M = M; p.N = N;
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    const int grid = inerf_wgrad_grid(n_points);
    const int cb = N / 32;
    if ((M != 128 && M != 256) || N % 32 || cb < 1 || cb > 8) return INERF_E_UNSUPPORTED;
    // 256 x 256: the row-coalesced form (INERF_WGRAD_FORM=points keeps the lane = point form for A/B runs)
    const char* form = getenv("INERF_WGRAD_FORM");
    if (M == 256 && N == 256 && !(form && form[0] == 'p')) {
        if ((n_points + (int64_t)kRowStep * (4 * grid + 2)) * (ldg > ldx ? ldg : ldx) * 4 >= (int64_t)1 << 32) return INERF_E_UNSUPPORTED;
        const int lds_rows = 2 * kRowBufBytes;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_wgrad_rows), hipFuncAttributeMaxDynamicSharedMemorySize, lds_rows);
        if (e != hipSuccess) return record(e);
        hipLaunchKernelGGL(k_mlp_wgrad_rows, dim3(grid), dim3(512), lds_rows, (hipStream_t)stream, p);
        return record(hipGetLastError());
    }
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
