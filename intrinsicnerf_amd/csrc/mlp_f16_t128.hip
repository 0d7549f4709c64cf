// mlp_f16_t128.hip - the inference form of the split-f16 encode+MLP kernel on a 128-POINT tile (round 6).
//
// Why: in k_encode_mlp_f16x3_dual (mlp_f16.hip) every wave pulls 4 KiB of weight fragments from L2 per 12 MFMAs - 2.65 MB per
// 64-point tile, 43 B/clk/CU at the full matrix rate against an L2 that delivers ~56 B/clk/CU with every CU asking - and the
// matrix pipe waits for them a third of the time (profiles/r01_mfma_mix_microbench.txt, r06_weight_stream_*.txt).  Here ONE workgroup of
// EIGHT waves walks 128 points through the same 14 GEMMs: a wave owns 32 output channels x all 128 points of a 256-wide layer
// (1 row block x 4 point blocks), so a weight fragment feeds four point blocks instead of two - 2 KiB per 12 MFMAs, half the
// L2 -> CU stream - and the operand reads that double instead (8 ds_read_b128 per 12 MFMAs) go to LDS, whose 256 B/clk for
// that instruction is a quarter used.  151,552 B of LDS, one workgroup per CU, two waves per SIMD (<= 256 registers each).
//
// Object-level network and the SSR network with <= 32 classes (no endpoint feature).
// Same arithmetic, same summation order per output element as the 64-point kernels (the packed blob is the same one: wave
// w8 reads row block w8 & 1 of packing wave w8 >> 1):  the results are bit-identical to k_encode_mlp_f16x3_dual's.  To keep
// that true for the two heads whose hidden layers stay in registers (albedo | shading hidden, view-dependent hidden: their
// output sums are formed per 64- / 32-channel group, in group order), those two layers keep the 64-point kernel's wave tile -
// channel group w8 & 3 x point half w8 >> 2 - and stream their weights twice per tile (15 % of the FLOPs).
#include <stdlib.h>

#include <type_traits>

#include "mlp_f16_dev.h"
#include "mlp_f16_heads.h"
#include "mlp_f16_pp.h"

#ifndef INERF_T128_PEEL
#define INERF_T128_PEEL 1       // (0: development builds for A/B runs)
#endif

namespace inerf {

constexpr int kPtsT = 128;                       // points per tile
constexpr int kPlaneT = kPtsT * kRowD;           // halfs per plane (rows of 296 halfs = 592 B: conflict-free ds_read_b128)
constexpr int kLdsBytesT = 2 * kPlaneT * 2;      // 151,552
constexpr int kSemScratchBytesT = 8 * 2 * 16 * 64 * 4;      // SSR: per workgroup 64 KiB - per wave [2 point blocks][16 registers][64 lanes] floats

// Object-level inference: the tile's position encoding (columns 0..63 of both planes: 32 KiB) parked in an L2-resident slot per workgroup
// between the first layer and the skip layer, which needs it again after the layers in between overwrote it in place - 4 LDS reads + 4
// stores + 4 loads + 4 LDS writes of 16 bytes per thread instead of a second evaluation of the encoder (~540 VALU instructions and 42
// two-byte LDS writes per thread; the kernel is bound by instruction issue: profiles/r06_pp_pipeline.txt).  The same bits.
constexpr int kEncCacheBytesT = 2 * kPtsT * kEncCols * 2;      // 32,768 per workgroup
int64_t enc_cache_bytes_t128(int64_t n_points) {
    const int64_t tiles = (n_points + kPtsT - 1) / kPtsT;
    return (tiles < device_cus() ? tiles : device_cus()) * (int64_t)kEncCacheBytesT;
}

int64_t sem_scratch_bytes_t128(int64_t n_points) {
    const int64_t tiles = (n_points + kPtsT - 1) / kPtsT;
    return (tiles < device_cus() ? tiles : device_cus()) * (int64_t)kSemScratchBytesT;
}

// kSsr: the SSR network (Semantic_NeRF, semantic_nerf.py:74-181) with at most 32 classes and no endpoint feature: xyz / 10 in the
// encoder, and between the albedo|shading head and the feature layer the semantic head - hidden layer (128 channels) split over the
// waves like the view-dependent one (32-channel group cg x 64-point half ph: semantic_linear.0.0 streamed twice per 128 points; the
// 64-point kernel streams it twice per 64), the wave's hidden channels stay in registers as B operands of the logits product
// (layout.h sem2q), and its PARTIAL logits (32 classes x 64 points over 32 hidden channels: 32 accumulator registers) are parked in
// the wave's own L2-resident scratch slot (MlpParams.sem_scratch: 8 KiB per wave - an explicit spill placed where nothing waits for
// it; held in registers across the feature and view layers they cost 48 spilled registers inside those loops) until the tile's planes
// are dead, where the four partials of a point meet in LDS and are added in group order.
// kSave (object-level network): the training forward - every layer's output also leaves as operand fragments of the weight-gradient
// products, the ReLU masks of h0..h7 as bits (mlp_f16.hip, k_encode_mlp_f16x3_dual<true, ..>): the SAME bytes at the same addresses
// as the 64-point kernel writes (a 128-point tile is two of the slots' 64-point tiles; this wave's 32 channels are one channel block
// of a fragment, one of the two mask words of a lane), so the chain and the weight-gradient kernels read either forward's buffer.
// kPipe (object-level inference, INERF_F16_KERNEL=pp): the trunk as a software pipeline over the tile's two 64-point halves - each half's
// epilogue is issued between the MFMAs of the other half's GEMM (mlp_f16_pp.h).  Same values, same summation order: bit-identical results.
template <bool kSsr, bool kSave = false, bool kPipe = false>
__global__ __launch_bounds__(512, 2) void k_encode_mlp_f16x3_t128(const MlpParams p) {
    static_assert(!(kSsr && kSave), "the saving form is the object-level network's");
    static_assert(!kPipe || (!kSsr && !kSave), "the pipelined trunk is the object-level inference form");
    constexpr int kParts = 512 / kPtsT;
    // the wide GEMMs' first products take a zero C operand instead of 64 v_mov in front of every GEMM (wide_gemm_h PEEL): +1.3 % same-box,
    // same bits (profiles/r06_peel_ab.txt); the saving form keeps the explicit zeroing like the two-workgroup one
    constexpr bool kPeel = INERF_T128_PEEL && !kSave;
    extern __shared__ __attribute__((aligned(16))) _Float16 ldst[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..7
    const int cg = wave & 3, ph = wave >> 2;                        // heads' hidden layers: channel group x point half
    const NetLayout& L = p.L;

    _Float16* const xw = ldst + (lane & 31) * kRowD;
    const _Float16* const xr = xw + 8 * (lane >> 5);                       // wide GEMM operand reads (+ column)
    _Float16* const xd = xw + 4 * (lane >> 5) + 32 * wave;                 // wide stores: this wave's 32 channels

    WeightBuf wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wts), 0, L.total_floats * 4, 0x00020000);
    wb.voff = lane * 16;
    // this wave's 32-channel stream of a 256-wide layer: row block (wave & 1) of packing wave (wave >> 1), k-blocks 4 KiB apart
    auto frag32 = [&](const GemmSlot& s, int kbt) { return (s.w + (wave >> 1) * kbt * 2 * 2 * 256) * 4 + (wave & 1) * 2048; };
    auto frag256 = [&](const GemmSlot& s, int kbt) { return (s.w + cg * kbt * 2 * 2 * 256) * 4; };       // 64-channel group cg
    auto frag128 = [&](const GemmSlot& s, int kbt) { return (s.w + cg * kbt * 1 * 2 * 256) * 4; };       // 32-channel group cg of a 128-wide layer

    WidePreH<1> pre1;
    WidePreH<2> pre2;
    prefetch_w<1, 4096>(pre1, wb, frag32(L.trunk[0], 4));
    float amax = 0.0f;                            // running maxima of |scaled value| (encoder inputs / layer outputs): the f16 range guard
    f16x2 amax2 = {(_Float16)0.0f, (_Float16)0.0f};
    const int n_tiles64 = (p.n_points + kTilePoints - 1) / kTilePoints;      // the save slots' tiles

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        int lane_t = tid;                         // (laundered per tile: keeps the lane_t parts of per-tile offsets out of the loop-invariant set,
        asm volatile("" : "+v"(lane_t));          // and `lane_t` itself out of the registers that live across the tile loop)
        lane_t &= 63;
        if constexpr (!kSave) {                   // inference: the f16 range guard is per tile (flag_f16_range); training forwards keep the
            amax = 0.0f;                          // launch's maximum (act_max; their status is one word)
            amax2 = f16x2{(_Float16)0.0f, (_Float16)0.0f};
        }
        // ---------------- encode -> hi/lo planes (xyz: columns 0..63, dir: columns 256..287) ----------------
        auto encode = [&](bool with_dir) {
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));
            const int pt = tid_o % kPtsT, part = tid_o / kPtsT;
            int gp = tile * kPtsT + pt;
            gp = gp < p.n_points ? gp : p.n_points - 1;
            int n_samples = p.n_samples;           // (laundered: the division's reciprocal, as a loop invariant, is one more register held - and
            asm volatile("" : "+s"(n_samples));    // at this kernel's 256 spilled - across the whole tile)
            const int ray = gp / n_samples;
            const float* __restrict__ r = p.rays + (size_t)ray * INERF_RAY_FLOATS;
            const float zz = __builtin_nontemporal_load(p.z + gp);
            _Float16* row = ldst + pt * kRowD;
            float x[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                x[c] = __fadd_rn(r[c], __fmul_rn(r[3 + c], zz));                                // run_nerf.py:488
                if (kSsr && p.xyz_div != 1.0f) x[c] = __fdiv_rn(x[c], p.xyz_div);              // semantic_nerf.py:64
            }
            for (int f = part; f < p.l_xyz; f += kParts) {
                const float s = (float)(1 << f);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sn, cs;
                    fast_sincosf(x[c] * s, &sn, &cs);
                    split_store<kPlaneT>(row + 3 + 6 * f + c, sn, amax);
                    split_store<kPlaneT>(row + 6 + 6 * f + c, cs, amax);
                }
            }
            if (part == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) split_store<kPlaneT>(row + c, x[c], amax);
                for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) { row[c] = (_Float16)0.0f; row[kPlaneT + c] = (_Float16)0.0f; }
            }
            if (with_dir) {
                const int fd = kParts - 1 - part;
                if (fd < p.l_dir) {
                    const float s = (float)(1 << fd);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float sn, cs;
                        fast_sincosf(r[8 + c] * s, &sn, &cs);
                        split_store<kPlaneT>(row + kColDirD + 3 + 6 * fd + c, sn, amax);
                        split_store<kPlaneT>(row + kColDirD + 6 + 6 * fd + c, cs, amax);
                    }
                }
                if (part == 3) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) split_store<kPlaneT>(row + kColDirD + c, r[8 + c], amax);
                    for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) { row[kColDirD + c] = (_Float16)0.0f; row[kPlaneT + kColDirD + c] = (_Float16)0.0f; }
                }
            }
        };
#ifdef INERF_ABL_NO_ENCODE      // (timing ablation of a development build: the first tile's encoding stays in LDS; results are wrong)
        if (tile == (int)blockIdx.x)
#endif
        encode(true);
        __syncthreads();
        // park the encoding for the skip layer (piece q = tid + 512 i of the tile's 2 048 sixteen-byte pieces: plane q / 1024, row (q % 1024) / 8)
        const bool enc_cached = !kSsr && !kSave && p.sem_scratch != nullptr;
        const __amdgpu_buffer_rsrc_t enc_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.sem_scratch, 0, enc_cached ? (int)((unsigned)gridDim.x * (unsigned)kEncCacheBytesT) : 0, 0x00020000);
        auto enc_piece = [&](int i) {
            const int q = (tid & 511) + 512 * i;
            return ldst + (q >> 10) * kPlaneT + ((q & 1023) >> 3) * kRowD + (q & 7) * 8;
        };
        if constexpr (!kSsr && !kSave) {
            if (enc_cached) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(enc_piece(i)), enc_rsrc,
                                                           (int)blockIdx.x * kEncCacheBytesT + (tid + 512 * i) * 16, 0, 0);
            }
        }
        auto encode_again = [&]() {         // the encoding back into columns 0..63 (sc0: past the vector L1, whose lines of an earlier tile may be stale)
            if constexpr (!kSsr && !kSave) {
                if (enc_cached) {
                    u32x4 v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        v[i] = __builtin_amdgcn_raw_buffer_load_b128(enc_rsrc, (int)blockIdx.x * kEncCacheBytesT + (tid + 512 * i) * 16, 0, 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(enc_piece(i)) = v[i];
                    return;
                }
            }
            encode(false);
        };
        // training forward: the encoding as operand fragments of dW = dZ^T enc (pts_linears.0 and .5's encoding columns), straight from
        // the planes before the first layer's output lands there: per 64-point half a 64-channel fragment slot, one channel block per
        // wave (waves 0..3 = half x block); waves 4, 5: the view encoding of half 0 / 1 (32 channels, one block)
        if constexpr (kSave) {
            int lane_e = lane_t;
            asm volatile("" : "+v"(lane_e));
            if (wave < 4) {
                const int half = wave >> 1, blk = wave & 1;
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + p.save_off[SAVE_ENC], 0, (int)((unsigned)n_tiles64 * (unsigned)(kFragTileBytes / 4)), 0x00020000);
                d.voff = (unsigned)(2 * tile + half) * (unsigned)(kFragTileBytes / 4) + (unsigned)blk * (2u * kFragBytes) + (unsigned)lane_e * 16u;
                planes_to_frag<1, kRowD, kPlaneT, 2>(xr + half * 64 * kRowD + 32 * blk, plane_selector(lane_e), d);
            } else if (wave < 6) {
                const int half = wave - 4;
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + p.save_off[SAVE_DIR], 0, (int)((unsigned)n_tiles64 * (unsigned)(kFragTileBytes / 8)), 0x00020000);
                d.voff = (unsigned)(2 * tile + half) * (unsigned)(kFragTileBytes / 8) + (unsigned)lane_e * 16u;
                planes_to_frag<1, kRowD, kPlaneT, 1>(xr + half * 64 * kRowD + kColDirD, plane_selector(lane_e), d);
            }
        }

        // a 256-wide layer in place: GEMM over columns [0, 16*KBT) | barrier | store to columns [0, 256) | barrier
        f32x16 am1[1][4];
        f32x4 bias1[1][4];
        float inv1;
        // ReLU masks of h0..h7 for the input-gradient chain (layout.h relu_bits_offset) and the layers' fragment slots (layout.h SaveSlot)
        const __amdgpu_buffer_rsrc_t bits_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.save + (kSave ? p.bits_off : 0), 0, kSave ? (int)((unsigned)n_tiles64 * (unsigned)kReluBitTileBytes) : 0, 0x00020000);
        auto frag_dst = [&](int slot, int half, int first_block) {
            FragDst d;
            d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + (kSave ? p.save_off[slot] : 0), 0,
                                                       kSave ? (int)((unsigned)n_tiles64 * (unsigned)kFragTileBytes) : 0, 0x00020000);
            d.voff = (unsigned)(2 * tile + half) * (unsigned)kFragTileBytes + (unsigned)first_block * (2u * kFragBytes) + (unsigned)lane_t * 16u;
            return d;
        };
        auto store256 = [&](const GemmSlot& s, bool relu, int frag_slot, auto&& prefetch_next, auto bits_tag) {
            load_bias<1>(bias1, inv1, wb, (s.b + 32 * wave) * 4, (s.b + kWidth) * 4, lane_t);
            prefetch_next();
            constexpr bool kBits = kSave && decltype(bits_tag)::value;
            // this wave's 32 channels = row block (wave & 1) of the 64-channel group (wave >> 1): word (wave & 1) of that group's lane pair
            const BitsDst bd = {bits_rsrc, kBits ? ((((2 * tile) * kReluBitLayers + (frag_slot - SAVE_H0)) * 4 + (wave >> 1)) * 64 + lane_t) * 8 + 4 * (wave & 1) : 0};
#ifndef INERF_ABL_NO_BARRIER    // (timing ablation of a development build: the layer barriers gone - racy, results are wrong)
            __syncthreads();                       // every wave has read the layer's input
#endif
            wide_store_h<1, kRowD, kPlaneT, false, kBits, 4, !kSave>(am1, inv1, bias1, xd, relu, amax2, nullptr, 0, 0, 0, nullptr, &bd);
#ifndef INERF_ABL_NO_BARRIER
            __syncthreads();
#endif
            if constexpr (kSave)                   // this wave's 32 channels of all 128 points (the layer is complete behind the barrier)
                if (frag_slot >= 0) {
                    int lane_o = lane_t;           // (laundered: the selector is rebuilt per layer instead of living in 8 registers)
                    asm volatile("" : "+v"(lane_o));
                    const Selector sel = plane_selector(lane_o);
                    planes_to_frag<1, kRowD, kPlaneT>(xr + 32 * wave, sel, frag_dst(frag_slot, 0, wave));
                    planes_to_frag<1, kRowD, kPlaneT>(xr + 64 * kRowD + 32 * wave, sel, frag_dst(frag_slot, 1, wave));
                }
        };
        constexpr std::true_type kWithBits{};
        constexpr std::false_type kNoBits{};
        auto pf32 = [&](const GemmSlot& s, int kbt) { return [&, kbt]() { prefetch_w<1, 4096>(pre1, wb, frag32(s, kbt)); }; };
        auto pf32_at = [&](const GemmSlot& s, int kbt, int kb_first) {
            return [&, kbt, kb_first]() { prefetch_w<1, 4096>(pre1, wb, frag32(s, kbt) + kb_first * 4096); };
        };
        auto pf256 = [&](const GemmSlot& s, int kbt) { return [&, kbt]() { prefetch_w<2>(pre2, wb, frag256(s, kbt)); }; };

        // ---------------- trunk ----------------
        if constexpr (kPipe) {
            // halves A (rows 0..63) and B (rows 64..127); accX: the half's accumulators of its latest GEMM, epilogued during the other half's next one
            f32x16 accA[2], accB[2];
            f32x4 bias[1][4];                      // of the layer whose epilogue the next step carries (requested a barrier + a k-block ahead)
            float inv = 0.0f;
            const _Float16* const xrB = xr + 64 * kRowD;
            _Float16* const xdB = xd + 64 * kRowD;
            auto bias_of = [&](const GemmSlot& s, f32x4 (&b)[1][4], float& iv) { load_bias<1>(b, iv, wb, (s.b + 32 * wave) * 4, (s.b + kWidth) * 4, lane_t); };
            // layer 0 (K = the 64 encoding columns): G(0,A) | G(0,B) || E(0,A)
            pp_step<4, 0, true, kRowD, kPlaneT>(pre1, wb, frag32(L.trunk[0], 4), xr, 0, accA, accB, inv, bias[0], xdB, amax2);
            bias_of(L.trunk[0], bias, inv);
            prefetch_w<1, 4096>(pre1, wb, frag32(L.trunk[0], 4));
            __syncthreads();                       // every wave has read the encoding of half A
            pp_step<4, 8, true, kRowD, kPlaneT>(pre1, wb, frag32(L.trunk[0], 4), xrB, 0, accB, accA, inv, bias[0], xd, amax2);
            prefetch_w<1, 4096>(pre1, wb, frag32(L.trunk[1], 16));
            __syncthreads();
#pragma unroll 1
            for (int layer = 1; layer < kDepth; ++layer) {
                const GemmSlot& s = L.trunk[layer];
                if (layer != kSkipInput) {
                    // A(l): G(l,A) || E(l-1,B);  B(l): G(l,B) || E(l,A)
                    pp_step<16, 8, true, kRowD, kPlaneT>(pre1, wb, frag32(s, 16), xr, 0, accA, accB, inv, bias[0], xdB, amax2);
                    bias_of(s, bias, inv);
                    prefetch_w<1, 4096>(pre1, wb, frag32(s, 16));
                    __syncthreads();
                    pp_step<16, 8, true, kRowD, kPlaneT>(pre1, wb, frag32(s, 16), xrB, 0, accB, accA, inv, bias[0], xd, amax2);
                    if (layer + 1 == kSkipInput) prefetch_w<1, 4096>(pre1, wb, frag32(L.trunk[kSkipInput], 20) + 4 * 4096);
                    else if (layer + 1 < kDepth) prefetch_w<1, 4096>(pre1, wb, frag32(L.trunk[layer + 1], 16));
                    __syncthreads();
                } else {
                    // pts_linears[5] over cat([pts, h]): the h part (k-blocks 4..19 of the stream) of both halves, then the encoding again in
                    // columns 0..63 (both halves' h4 are consumed), then the encoding part on top of the same accumulators
                    pp_step<16, 8, true, kRowD, kPlaneT>(pre1, wb, frag32(s, 20) + 4 * 4096, xr, 0, accA, accB, inv, bias[0], xdB, amax2);      // || E(4,B)
                    bias_of(s, bias, inv);
                    prefetch_w<1, 4096>(pre1, wb, frag32(s, 20) + 4 * 4096);
                    __syncthreads();
                    pp_step<16, 0, true, kRowD, kPlaneT>(pre1, wb, frag32(s, 20) + 4 * 4096, xrB, 0, accB, accA, inv, bias[0], xd, amax2);
                    prefetch_w<1, 4096>(pre1, wb, frag32(s, 20));
                    __syncthreads();
                    encode_again();
                    __syncthreads();
                    pp_step<4, 0, false, kRowD, kPlaneT>(pre1, wb, frag32(s, 20), xr, 0, accA, accB, inv, bias[0], xdB, amax2);
                    prefetch_w<1, 4096>(pre1, wb, frag32(s, 20));
                    __syncthreads();               // every wave has read the encoding of half A: E(5,A) may overwrite it
                    pp_step<4, 8, false, kRowD, kPlaneT>(pre1, wb, frag32(s, 20), xrB, 0, accB, accA, inv, bias[0], xd, amax2);                 // || E(5,A)
                    prefetch_w<1, 4096>(pre1, wb, frag32(L.trunk[kSkipInput + 1], 16));
                    __syncthreads();
                }
            }
            prefetch_w<2>(pre2, wb, frag256(L.as1, 16));                      // (outside the layer loop: assigned inside it the 32 registers are loop-carried)
            pp_epilogue<kRowD, kPlaneT>(accB, inv, bias[0], xdB, amax2);      // E(7,B): nothing left to run it under
            __syncthreads();
        } else {
            wide_gemm_h<1, 4, 0, kRowD, kPlaneT, true, 4096, 4, 2048, kPeel>(pre1, wb, frag32(L.trunk[0], 4), xr, 0, 0, lane_t, am1);
            store256(L.trunk[0], true, SAVE_H0, pf32(L.trunk[1], 16), kWithBits);
    #pragma unroll 1
            for (int layer = 1; layer < kSkipInput; ++layer) {
                wide_gemm_h<1, 16, 0, kRowD, kPlaneT, true, 4096, 4, 2048, kPeel>(pre1, wb, frag32(L.trunk[layer], 16), xr, 0, 0, lane_t, am1);
                if (layer + 1 < kSkipInput) store256(L.trunk[layer], true, SAVE_H0 + layer, pf32(L.trunk[layer + 1], 16), kWithBits);
                else                        store256(L.trunk[layer], true, SAVE_H0 + layer, pf32_at(L.trunk[kSkipInput], 20, 4), kWithBits);
            }
            {   // pts_linears[5] over cat([pts, h]): h-part (k-blocks 4..19 of the stream), then the encoding again
                const GemmSlot& s = L.trunk[kSkipInput];
                wide_gemm_h<1, 16, 0, kRowD, kPlaneT, true, 4096, 4, 2048, kPeel>(pre1, wb, frag32(s, 20) + 4 * 4096, xr, 0, 0, lane_t, am1);
                prefetch_w<1, 4096>(pre1, wb, frag32(s, 20));
                __syncthreads();
                encode_again();
                __syncthreads();
                wide_gemm_h<1, 4, 0, kRowD, kPlaneT, false, 4096, 4>(pre1, wb, frag32(s, 20), xr, 0, 0, lane_t, am1);
                store256(s, true, SAVE_H0 + kSkipInput, pf32(L.trunk[6], 16), kWithBits);
            }
            wide_gemm_h<1, 16, 0, kRowD, kPlaneT, true, 4096, 4, 2048, kPeel>(pre1, wb, frag32(L.trunk[6], 16), xr, 0, 0, lane_t, am1);
            store256(L.trunk[6], true, SAVE_H0 + 6, pf32(L.trunk[7], 16), kWithBits);
            wide_gemm_h<1, 16, 0, kRowD, kPlaneT, true, 4096, 4, 2048, kPeel>(pre1, wb, frag32(L.trunk[7], 16), xr, 0, 0, lane_t, am1);
            store256(L.trunk[7], true, SAVE_H7, pf256(L.as1, 16), kWithBits);

        }

        // ---------------- heads ----------------
        const bool sem = kSsr && L.sem_rbs > 0;
        const int my_pt = tile * kPtsT + 16 * wave + (lane_t & 15);
        const bool my_valid = my_pt < p.n_points;
        float* const out_row = p.raw + (size_t)(my_valid ? my_pt : 0) * p.channels;
        // (operand addresses of the heads from the per-tile laundered lane_t index: as loop invariants they are two more spilled registers)
        const _Float16* const xs = ldst + (16 * wave + (lane_t & 15)) * kRowD + 8 * (lane_t >> 4);   // skinny operand reads: this wave's 16 points

        // albedo + shading: hidden layer (this wave: channel group cg of point half ph) -> registers -> partial output sums
        f32x4 part_as[2], part_res[2];
        const _Float16* const xr_h = ldst + ((lane_t & 31) + 64 * ph) * kRowD + 8 * (lane_t >> 5);
        {
            f32x16 am2[2][2];
            f32x4 bias2[2][4];
            float inv2;
            wide_gemm_h<2, 16, 0, kRowD, kPlaneT, true, 4096, 2, 2048, kPeel>(pre2, wb, frag256(L.as1, 16), xr_h, 0, 0, lane_t, am2);
            load_bias<2>(bias2, inv2, wb, (L.as1.b + 64 * cg) * 4, (L.as1.b + kWidth) * 4, lane_t);
            if (sem) prefetch_w<1>(pre1, wb, frag128(L.sem1, 16));
            else     prefetch_w<1, 4096>(pre1, wb, frag32(L.feat, 16));
            f16x8 hi[4][2], lo[4][2];
            to_operands<2, false, 2, !kSave>(am2, inv2, bias2, amax2, hi, lo);
            if constexpr (kSave) {                // the hidden layer as operand fragments, transposed out of the registers (it never touches LDS):
                int lane_o = lane_t;              // channel group cg of point half ph = the 64-point kernel's wave cg of tile 2 * tile + ph
                asm volatile("" : "+v"(lane_o));
                operands_to_frag<2>(hi, lo, accumulator_selector(lane_o), frag_dst(SAVE_AS1H, ph, 2 * cg));
            }
            regop_gemm<4>(wb, (L.as2r.w + cg * 4 * 2 * 256) * 4, hi, lo, part_as);
        }
        // semantic head (semantic_nerf.py:150-152): hidden = relu(semantic_linear.0.0 h7), this wave's 32 channels x 64 points -> registers
        // -> partial logits of the (one) 32-class block, kept until the exchange at the end of the tile
        const __amdgpu_buffer_rsrc_t sem_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.sem_scratch, 0, kSsr ? (int)((unsigned)gridDim.x * (unsigned)kSemScratchBytesT) : 0, 0x00020000);
        const int sem_slot = ((int)blockIdx.x * 8 + wave) * (kSemScratchBytesT / 8) + lane_t * 16;      // + (pb * 4 + g) * 1024
        if constexpr (kSsr) {
            if (sem) {
                f32x16 sem_part[2];
                f32x16 ams[1][2];
                f32x4 biass[1][4];
                float invs;
                wide_gemm_h<1, 16, 0, kRowD, kPlaneT, true, 2048, 2, 2048, kPeel>(pre1, wb, frag128(L.sem1, 16), xr_h, 0, 0, lane_t, ams);
                load_bias<1>(biass, invs, wb, (L.sem1.b + 32 * cg) * 4, (L.sem1.b + kHalf) * 4, lane_t);
                prefetch_w<1, 4096>(pre1, wb, frag32(L.feat, 16));
                f16x8 hi[2][2], lo[2][2];
                to_operands<1>(ams, invs, biass, amax2, hi, lo);
                regop_gemm_full<2, 2>(wb, (L.sem2q.w + cg * 2 * 2 * 256) * 4, hi, lo, sem_part);
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {sem_part[pb][4 * g], sem_part[pb][4 * g + 1], sem_part[pb][4 * g + 2], sem_part[pb][4 * g + 3]};
                        // (whole offset in the VGPR operand: a 16-byte buffer store with a register SGPR offset gets no hazard wait state, tests/test_isa_audit_cpu.py)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sem_rsrc, sem_slot + (pb * 4 + g) * 1024, 0, 0);
                    }
            }
        }
        // feature (no activation) in place of h7, then the view-dependent layer over [feature | dir] -> registers
        wide_gemm_h<1, 16, 0, kRowD, kPlaneT, true, 4096, 4, 2048, kPeel>(pre1, wb, frag32(L.feat, 16), xr, 0, 0, lane_t, am1);
        // sigma, last of h7's readers (here, not in front of the heads: four registers fewer across their GEMM loops)
        const f32x4 sig4 = skinny_gemm_h<8, kPlaneT>(wb, L.alpha.w * 4, L.alpha.b * 4, (L.alpha.b + 16) * 4, xs, lane_t);
        {
            WidePreH<1> prev;
            store256(L.feat, false, SAVE_FEAT, [&]() { prefetch_w<1>(prev, wb, frag128(L.views, 18)); }, kNoBits);
            f32x16 amv[1][2];
            f32x4 biasv[1][4];
            float invv;
            wide_gemm_h<1, 18, 0, kRowD, kPlaneT, true, 2048, 2, 2048, kPeel>(prev, wb, frag128(L.views, 18), xr_h, 0, 0, lane_t, amv);
            load_bias<1>(biasv, invv, wb, (L.views.b + 32 * cg) * 4, (L.views.b + kHalf) * 4, lane_t);
            prefetch_w<1, 4096>(pre1, wb, frag32(L.trunk[0], 4));
            f16x8 hi[2][2], lo[2][2];
            to_operands<1, false, 2, !kSave>(amv, invv, biasv, amax2, hi, lo);
            if constexpr (kSave) {                // 128 channels: a four-block fragment slot, block cg of point half ph
                int lane_o = lane_t;
                asm volatile("" : "+v"(lane_o));
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + p.save_off[SAVE_VH], 0, (int)((unsigned)n_tiles64 * (unsigned)(kFragTileBytes / 2)), 0x00020000);
                d.voff = (unsigned)(2 * tile + ph) * (unsigned)(kFragTileBytes / 2) + (unsigned)cg * (2u * kFragBytes) + (unsigned)lane_o * 16u;
                operands_to_frag<1, 4>(hi, lo, accumulator_selector(lane_o), d);
            }
            regop_gemm<2>(wb, (L.resr.w + cg * 2 * 2 * 256) * 4, hi, lo, part_res);
        }
        __syncthreads();                           // feature / dir columns are dead: the exchange area may be written
        if (lane_t < 32) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                float* ex = reinterpret_cast<float*>(ldst + (lane_t + 32 * pb + 64 * ph) * kRowD + kColExD) + 8 * cg;
                *reinterpret_cast<f32x4*>(ex) = part_as[pb];
                *reinterpret_cast<f32x4*>(ex + 4) = part_res[pb];
            }
        }
        // the four partial logit blocks of a point (one per 32-channel group of the hidden layer), 32 floats each, in dead columns that
        // neither the heads' exchange area (hi plane, bytes 128..255) nor the next tile's encode (bytes 0..127 and 512..575) touch:
        // groups 0 / 1 at bytes 256..383 / 384..511 of the row in the hi plane, groups 2 / 3 at bytes 128..255 / 256..383 in the lo plane
        auto sem_ex = [&](int g, int r) {
            return reinterpret_cast<float*>(ldst) + (g < 2 ? 64 + 32 * g : kPlaneT / 2 + 32 * (g - 1)) + r * (kRowD / 2);
        };
        if constexpr (kSsr) {
            if (sem) {      // this wave's partial logits back from its slot (sc0: past the vector L1, whose lines of an earlier tile may be stale;
                            // fetched here, not ahead of the barrier: held across it they were spilled to scratch) and into the exchange area:
                            // accumulator register 4 g + i = class 8 g + 4 (lane >> 5) + i of point (lane & 31) of point block pb of this wave's half
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    u32x4 sem_v[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) sem_v[g] = __builtin_amdgcn_raw_buffer_load_b128(sem_rsrc, sem_slot + (pb * 4 + g) * 1024, 0, 1);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<u32x4*>(sem_ex(cg, 64 * ph + 32 * pb + (lane_t & 31)) + 8 * g + 4 * (lane_t >> 5)) = sem_v[g];
                }
            }
        }
        __syncthreads();
        if constexpr (kSsr) {
            if (sem) {      // this wave's 16 points x 32 classes: lane = (point, 8 classes); the four partials in group order (deterministic)
                const int r = 16 * wave + (lane_t & 15), c8 = 8 * (lane_t >> 4);
                f32x4 s0 = *reinterpret_cast<const f32x4*>(sem_ex(0, r) + c8), s1 = *reinterpret_cast<const f32x4*>(sem_ex(0, r) + c8 + 4);
#pragma unroll
                for (int g = 1; g < 4; ++g) {
                    s0 += *reinterpret_cast<const f32x4*>(sem_ex(g, r) + c8);
                    s1 += *reinterpret_cast<const f32x4*>(sem_ex(g, r) + c8 + 4);
                }
                const float sem_inv2 = wb.scalar((L.sem2.b + 16 * L.sem_rbs) * 4);
                const f32x4 b0 = wb.vec4((L.sem2.b + c8) * 4, 0), b1 = wb.vec4((L.sem2.b + c8 + 4) * 4, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (my_valid && c8 + i < p.n_classes) __builtin_nontemporal_store(__builtin_fmaf(s0[i], sem_inv2, b0[i]), out_row + INERF_BASE_CHANNELS + c8 + i);
                    if (my_valid && c8 + 4 + i < p.n_classes) __builtin_nontemporal_store(__builtin_fmaf(s1[i], sem_inv2, b1[i]), out_row + INERF_BASE_CHANNELS + c8 + 4 + i);
                }
            }
        }
        // a wave's 16 points x 11 floats are 704 CONTIGUOUS bytes of raw: they leave as 44 sixteen-byte pieces (lanes 0..43, staged in
        // dead columns of the lo plane), every 64-byte sector written once by one instruction
        const bool whole_rows = p.channels == INERF_BASE_CHANNELS && tile * kPtsT + kPtsT <= p.n_points &&
                                (reinterpret_cast<uintptr_t>(p.raw) & 15) == 0;
        auto stage_row = [&](int r) { return reinterpret_cast<float*>(ldst + kPlaneT + r * kRowD + 128); };      // lo plane, bytes 256..299 of row r
        if (lane_t < 16 && my_valid) {
            const float* ex = reinterpret_cast<const float*>(ldst + (16 * wave + lane_t) * kRowD + kColExD);
            f32x4 as4 = {0.0f, 0.0f, 0.0f, 0.0f}, res4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                as4 += *reinterpret_cast<const f32x4*>(ex + 8 * w);
                res4 += *reinterpret_cast<const f32x4*>(ex + 8 * w + 4);
            }
            const f32x4 b_as = wb.vec4(L.as2.b * 4, 0), b_res = wb.vec4(L.res.b * 4, 0);
            const float inv_as = wb.scalar((L.as2.b + 16) * 4), inv_res = wb.scalar((L.res.b + 16) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                as4[i] = __builtin_fmaf(as4[i], inv_as, b_as[i]);
                res4[i] = __builtin_fmaf(res4[i], inv_res, b_res[i]);
            }
            const float a0 = sigmoid_ref_h(as4[0]), a1 = sigmoid_ref_h(as4[1]), a2 = sigmoid_ref_h(as4[2]);
            const float sh = sigmoid_ref_h(as4[3]);
            const float r0 = sigmoid_ref_h(res4[0]), r1 = sigmoid_ref_h(res4[1]), r2 = sigmoid_ref_h(res4[2]);
            const float c0 = __fadd_rn(__fmul_rn(a0, sh), r0), c1 = __fadd_rn(__fmul_rn(a1, sh), r1), c2 = __fadd_rn(__fmul_rn(a2, sh), r2);   // run_nerf_helpers.py:320
            if (whole_rows) {
                float* st = stage_row(16 * wave + lane_t);
                *reinterpret_cast<f32x4*>(st) = f32x4{c0, c1, c2, sig4[0]};
                *reinterpret_cast<f32x4*>(st + 4) = f32x4{a0, a1, a2, sh};
                st[8] = r0; st[9] = r1; st[10] = r2;
            } else {
                __builtin_nontemporal_store(c0, out_row + 0);
                __builtin_nontemporal_store(c1, out_row + 1);
                __builtin_nontemporal_store(c2, out_row + 2);
                __builtin_nontemporal_store(sig4[0], out_row + 3);
                __builtin_nontemporal_store(a0, out_row + 4); __builtin_nontemporal_store(a1, out_row + 5); __builtin_nontemporal_store(a2, out_row + 6);
                __builtin_nontemporal_store(sh, out_row + 7);
                __builtin_nontemporal_store(r0, out_row + 8); __builtin_nontemporal_store(r1, out_row + 9); __builtin_nontemporal_store(r2, out_row + 10);
            }
        }
        if (whole_rows && lane_t < 44) {      // (LDS is in order within a wave: the sixteen lanes' rows are there)
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * lane_t + i, pt = e / INERF_BASE_CHANNELS;
                v[i] = stage_row(16 * wave + pt)[e - INERF_BASE_CHANNELS * pt];
            }
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.raw + (size_t)(tile * kPtsT + 16 * wave) * INERF_BASE_CHANNELS) + lane_t);
        }
        flag_f16_range(p, tile * kPtsT, kPtsT, amax, amax2, lane_t);
        // (the next tile's encode writes bytes 0..127 and 512..575 of the rows, both planes: clear of the exchange area (hi plane, bytes
        // 128..255) and of the staging rows (lo plane, bytes 256..299) this tile's last readers may still be in)
    }
    if (kSave && p.act_max) {
        float m = fmaxf(amax, fmaxf((float)amax2[0], (float)amax2[1])) * (1.0f / kActScale);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0 && m == m) atomicMax(reinterpret_cast<unsigned int*>(p.act_max), __builtin_bit_cast(unsigned int, m));
    }
}

int launch_mlp_f16x3_t128(MlpParams& p, int64_t n_points, bool ssr, hipStream_t stream) {
    p.n_tiles = (int)((n_points + kPtsT - 1) / kPtsT);
    const int grid = p.n_tiles < device_cus() ? p.n_tiles : device_cus();
    const bool save = p.save != nullptr;
    if (ssr && save) return INERF_E_UNSUPPORTED;
    const char* form = getenv("INERF_F16_KERNEL");
    const bool pipe = !ssr && !save && form && form[0] == 'p';        // the pipelined trunk (opt-in)
    void (*kern)(const MlpParams) = ssr ? k_encode_mlp_f16x3_t128<true> : save ? k_encode_mlp_f16x3_t128<false, true>
                                  : pipe ? k_encode_mlp_f16x3_t128<false, false, true> : k_encode_mlp_f16x3_t128<false>;
    static PerDeviceOnce attr_sets[4];
    PerDeviceOnce& attr_set = attr_sets[ssr ? 1 : save ? 2 : pipe ? 3 : 0];
    if (attr_set.first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesT);
        if (e != hipSuccess) return record(e);
        attr_set.mark();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kLdsBytesT, stream, p);
    return record(hipGetLastError());
}

}  // namespace inerf
