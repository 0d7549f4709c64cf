// mlp_f16.hip - the encode+MLP kernel on the f16 matrix pipe with split fp32 operands
// (INERF_PREC_F16X3).  Same tile walk, same layer list, same outputs as mlp.hip; only the GEMM
// arithmetic differs:
//
//     every fp32 operand v (weight or activation) is first multiplied by a power of two (weights: per GEMM,
//     max|W'| in (2^13, 2^14]; activations: 8) and then carried as  hi = f16(v'),  lo = f16(v' - hi):
//     v' = hi + lo to 22 bits (relative) / 2^-25 (absolute - far below fp32 round-off of the unscaled
//     value).  Each product  w * x  is evaluated as  hi_w*hi_x + hi_w*lo_x + lo_w*hi_x  on the f16 matrix
//     cores, all three accumulated in ONE fp32 accumulator (the lo*lo term is 2^-22 relative and dropped).
//     f16 x f16 products are exact in fp32, so the only roundings are the fp32 accumulations: the error
//     against an fp64 evaluation is the same as the all-fp32 kernel's (5e-7 of the channel scale for both,
//     measured on the oracle), and the parity tests hold both kernels to the same 1e-4.
//
// Three v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate each) replace eight v_mfma_f32_32x32x2_f32:
// 5.3x fewer matrix-pipe cycles per point.  |activation| must stay below 6e4 / 8: beyond that the
// kernel raises INERF_STATUS_F16_RANGE in the caller's status word and the caller re-runs in fp32.
//
// LDS: two planes (hi, lo) of X[64 points][616 halfs]; 616*2 B = 77*16 B, so the 16 rows a
// ds_read_b128 lane group touches fall on 16 distinct bank slots.  Columns as in layout.h
// (enc 64 | dir 32 | A 256 | B 256).  157,696 B per workgroup, one workgroup per CU.
//
// Two kernels: k_encode_mlp_f16x3 (the layout above; SSR network with the endpoint feature, A/B runs) and, further
// down, k_encode_mlp_f16x3_dual (both networks: half the LDS and registers, two workgroups per CU).
#include <stdlib.h>

#include "mlp_f16_dev.h"
#include "mlp_f16_heads.h"

#ifndef INERF_DUAL_PEEL
#define INERF_DUAL_PEEL 1       // (0: development builds for A/B runs)
#endif

namespace inerf {

// ------------------------------------------------------------------------------------------------
// kSave: training forward - every layer's output (and the encodings) is also written to p.save as fp32
// (layout.h SaveSlot) for the backward pass (mlp_bwd.hip)
template <bool kSsr, bool kSave>
__global__ __launch_bounds__(256, 1) void k_encode_mlp_f16x3(const MlpParams p) {
    constexpr int kPts = kTilePoints;
    constexpr int kParts = 256 / kPts;
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ wts = p.wts;
    const NetLayout& L = p.L;
    float amax = 0.0f;                          // encoder inputs (fp32 running max of |scaled value|)
    f16x2 amax2 = {(_Float16)0.0f, (_Float16)0.0f};   // layer outputs (packed f16 running max of |hi|)

    _Float16* const xw = ldsh + (lane & 31) * kRowH;                                   // + 8*(lane>>5) for reads, 4*(lane>>5) for writes
    const _Float16* const xr = xw + 8 * (lane >> 5);
    _Float16* const xd = xw + 4 * (lane >> 5);
    const _Float16* const xs = ldsh + (16 * wave + (lane & 15)) * kRowH + 8 * (lane >> 4);

    WeightBuf wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, L.total_floats * 4, 0x00020000);
    wb.voff = lane * 16;
    // fragment streams (byte offsets, wave-uniform): a wide GEMM with KBT k-blocks stores per wave KBT*RB*2 KiB
    auto frag256 = [&](const GemmSlot& s, int kbt) { return (s.w + wave * kbt * 2 * 2 * 256) * 4; };
    auto frag128 = [&](const GemmSlot& s, int kbt) { return (s.w + wave * kbt * 1 * 2 * 256) * 4; };
    auto bias256 = [&](const GemmSlot& s) { return (s.b + 64 * wave) * 4; };
    auto bias128 = [&](const GemmSlot& s) { return (s.b + 32 * wave) * 4; };
    auto scale256 = [&](const GemmSlot& s) { return (s.b + kWidth) * 4; };     // the constant after the bias vector
    auto scale128 = [&](const GemmSlot& s) { return (s.b + kHalf) * 4; };

    WidePreH<2> pre2;
    WidePreH<1> pre1;
    wide_prefetch_h<2>(pre2, wb, frag256(L.trunk[0], 4), bias256(L.trunk[0]), scale256(L.trunk[0]), lane);

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        if constexpr (!kSave) {                     // inference: the f16 range guard is per tile (flag_f16_range); training forwards keep the
            amax = 0.0f;                            // launch's maximum (act_max; their status is one word)
            amax2 = f16x2{(_Float16)0.0f, (_Float16)0.0f};
        }
        // ---------------- encode -> hi/lo planes ----------------
        {
            const int pt = tid % kPts, part = tid / kPts;
            int gp = tile * kPts + pt;
            gp = gp < p.n_points ? gp : p.n_points - 1;
            const int ray = gp / p.n_samples;
            const float* __restrict__ r = p.rays + (size_t)ray * INERF_RAY_FLOATS;
            const float zz = __builtin_nontemporal_load(p.z + gp);     // streamed once: keep it out of the L2 the weights live in
            _Float16* row = ldsh + pt * kRowH;
            float x[3], v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                x[c] = __fadd_rn(r[c], __fmul_rn(r[3 + c], zz));                 // run_nerf.py:488
                if (p.xyz_div != 1.0f) x[c] = __fdiv_rn(x[c], p.xyz_div);        // semantic_nerf.py:64
                v[c] = r[8 + c];
            }
            for (int f = part; f < p.l_xyz; f += kParts) {
                const float s = (float)(1 << f);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sn, cs;
                    fast_sincosf(x[c] * s, &sn, &cs);
                    split_store(row + kColEnc + 3 + 6 * f + c, sn, amax);
                    split_store(row + kColEnc + 6 + 6 * f + c, cs, amax);
                }
            }
            const int fd = kParts - 1 - part;
            if (fd < p.l_dir) {
                const float s = (float)(1 << fd);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sn, cs;
                    fast_sincosf(v[c] * s, &sn, &cs);
                    split_store(row + kColDir + 3 + 6 * fd + c, sn, amax);
                    split_store(row + kColDir + 6 + 6 * fd + c, cs, amax);
                }
            }
            if (part == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) split_store(row + kColEnc + c, x[c], amax);
                for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) { row[kColEnc + c] = (_Float16)0.0f; row[kPlaneH + kColEnc + c] = (_Float16)0.0f; }
            }
            if (part == 3) {
#pragma unroll
                for (int c = 0; c < 3; ++c) split_store(row + kColDir + c, v[c], amax);
                for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) { row[kColDir + c] = (_Float16)0.0f; row[kPlaneH + kColDir + c] = (_Float16)0.0f; }
            }
        }
        __syncthreads();
        // training forward: the encoding leaves as operand fragments of the products dW = dZ^T enc (pts_linears.0, and .5's
        // encoding columns), straight from the planes: a 64-channel fragment slot, one channel block per wave 0 / 1 (wave 2: the view encoding)
        if constexpr (kSave) {
            if (wave < 2) {
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + p.save_off[SAVE_ENC], 0, (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 4)), 0x00020000);
                d.voff = (unsigned)tile * (unsigned)(kFragTileBytes / 4) + (unsigned)wave * (2u * kFragBytes) + (unsigned)lane * 16u;
                planes_to_frag<1, kRowH, kPlaneH, 2>(xr + kColEnc + 32 * wave, plane_selector(lane), d);
            } else if (wave == 2) {                 // the view encoding: a 32-channel fragment slot (one block per k-block)
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + p.save_off[SAVE_DIR], 0, (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 8)), 0x00020000);
                d.voff = (unsigned)tile * (unsigned)(kFragTileBytes / 8) + (unsigned)lane * 16u;
                planes_to_frag<1, kRowH, kPlaneH, 1>(xr + kColDir, plane_selector(lane), d);
            }
        }

        const int pt0 = tile * kPts + (lane & 31);              // this lane's points in wide results
        // training forward: fp32 copy of a layer's output, this lane's first point / first channel of its wave
        auto save_dst = [&](int slot, int width, int chan0) {
            SaveDst d;
            // sizes and offsets in unsigned 32-bit arithmetic: a 256-wide slot of 2^21+ points is beyond INT_MAX bytes
            // (the host keeps n_points * 256 * 4 < 2^32, see inerf_encode_mlp_train)
            d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + (kSave ? p.save_off[slot] : 0), 0,
                                                       kSave ? (int)((unsigned)p.n_points * (unsigned)width * 4u) : 0, 0x00020000);
            d.voff = (int)(((unsigned)pt0 * (unsigned)width + (unsigned)(chan0 + 4 * (lane >> 5))) * 4u);
            d.stride = width;
            return d;
        };
        // ReLU masks of h0..h7 for the input-gradient chain (layout.h relu_bits_offset): one descriptor for the area
        const __amdgpu_buffer_rsrc_t bits_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.save + (kSave ? p.bits_off : 0), 0, kSave ? (int)((unsigned)p.n_tiles * (unsigned)kReluBitTileBytes) : 0, 0x00020000);
        // training forward (layout.h SaveSlot): h0..h7 and the feature layer leave as operand fragments of the weight-gradient
        // products (planes_to_frag, after the layer's barrier), h0..h7 also as ReLU mask bits; the albedo|shading hidden layer as
        // fp32 rows from the epilogue's registers
        const Selector fsel = plane_selector(lane);
        auto step256 = [&](const GemmSlot& s, auto kb0c, auto kb1c, int c0, int c1, int dcol, bool relu, auto slot_c,
                           auto&& prefetch_next) {
            constexpr int KB0 = decltype(kb0c)::value, KB1 = decltype(kb1c)::value;
            constexpr int slot = decltype(slot_c)::value;
            constexpr bool kTrunk = slot >= SAVE_H0 && slot <= SAVE_H7;
            constexpr bool kFrag = kTrunk || slot == SAVE_FEAT || slot == SAVE_AS1H;
            constexpr bool kRows = false;
            constexpr bool kBits = kSave && kTrunk;
            f32x16 am[2][2];
            f32x4 bias[2][4];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int g = 0; g < 4; ++g) bias[rb][g] = pre2.b[rb][g];
            const float inv = pre2.inv;
            wide_gemm_h<2, KB0, KB1>(pre2, wb, frag256(s, KB0 + KB1), xr, c0, c1, lane, am);
            prefetch_next();
            const SaveDst sv = save_dst(slot, kWidth, 64 * wave);
            const BitsDst bd = {bits_rsrc, kBits ? (((tile * kReluBitLayers + (slot - SAVE_H0)) * 4 + wave) * 64 + lane) * 8 : 0};
            wide_store_h<2, kRowH, kPlaneH, kRows, kBits, 2, !kSave>(am, inv, bias, xd + dcol + 64 * wave, relu, amax2, nullptr, 0, 0, 0, &sv, &bd);
            __syncthreads();
            if constexpr (kSave && kFrag) {
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + p.save_off[slot], 0, (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes), 0x00020000);
                d.voff = (unsigned)tile * (unsigned)kFragTileBytes + (unsigned)(2 * wave) * (2u * kFragBytes) + (unsigned)lane * 16u;
                planes_to_frag<2, kRowH, kPlaneH>(xr + dcol + 64 * wave, fsel, d);
            }
        };
        auto step128 = [&](const GemmSlot& s, auto kb0c, auto kb1c, int c0, int c1, int dcol, bool relu, float* gout, auto slot_c,
                           auto&& prefetch_next) {
            constexpr int KB0 = decltype(kb0c)::value, KB1 = decltype(kb1c)::value;
            constexpr int slot = decltype(slot_c)::value;
            constexpr bool kFrag128 = kSave && slot == SAVE_VH;           // views hidden layer: a four-block fragment slot; semantic hidden: rows
            f32x16 am[1][2];
            f32x4 bias[1][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[0][g] = pre1.b[0][g];
            const float inv = pre1.inv;
            wide_gemm_h<1, KB0, KB1>(pre1, wb, frag128(s, KB0 + KB1), xr, c0, c1, lane, am);
            prefetch_next();
            const SaveDst sv = save_dst(slot, kHalf, 32 * wave);
            wide_store_h<1, kRowH, kPlaneH, kSave && !kFrag128, false, 2, !kSave>(am, inv, bias, xd + dcol + 32 * wave, relu, amax2, gout, p.channels, pt0 < p.n_points,
                                                               pt0 + 32 < p.n_points, &sv);
            __syncthreads();
            if constexpr (kFrag128) {
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + p.save_off[slot], 0, (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 2)), 0x00020000);
                d.voff = (unsigned)tile * (unsigned)(kFragTileBytes / 2) + (unsigned)wave * (2u * kFragBytes) + (unsigned)lane * 16u;
                planes_to_frag<1, kRowH, kPlaneH, 4>(xr + dcol + 32 * wave, fsel, d);
            }
        };
        auto pf256 = [&](const GemmSlot& s, int kbt) {
            return [&, kbt]() { wide_prefetch_h<2>(pre2, wb, frag256(s, kbt), bias256(s), scale256(s), lane); };
        };
        auto pf128 = [&](const GemmSlot& s, int kbt) {
            return [&, kbt]() { wide_prefetch_h<1>(pre1, wb, frag128(s, kbt), bias128(s), scale128(s), lane); };
        };
        using std::integral_constant;
        constexpr integral_constant<int, 0> K0{};
        constexpr integral_constant<int, 2> K2{};
        constexpr integral_constant<int, 4> K4{};
        constexpr integral_constant<int, 16> K16{};
        const bool sem = kSsr && L.sem_rbs > 0;
        // ---------------- trunk ----------------
        step256(L.trunk[0], K4, K0, kColEnc, 0, kColA, true, integral_constant<int, SAVE_H0 + 0>{}, pf256(L.trunk[1], 16));
        step256(L.trunk[1], K16, K0, kColA, 0, kColB, true, integral_constant<int, SAVE_H0 + 1>{}, pf256(L.trunk[2], 16));
        step256(L.trunk[2], K16, K0, kColB, 0, kColA, true, integral_constant<int, SAVE_H0 + 2>{}, pf256(L.trunk[3], 16));
        step256(L.trunk[3], K16, K0, kColA, 0, kColB, true, integral_constant<int, SAVE_H0 + 3>{}, pf256(L.trunk[4], 16));
        step256(L.trunk[4], K16, K0, kColB, 0, kColA, true, integral_constant<int, SAVE_H0 + 4>{}, pf256(L.trunk[5], 20));
        step256(L.trunk[5], K4, K16, kColEnc, kColA, kColB, true, integral_constant<int, SAVE_H0 + 5>{}, pf256(L.trunk[6], 16));
        step256(L.trunk[6], K16, K0, kColB, 0, kColA, true, integral_constant<int, SAVE_H0 + 6>{}, pf256(L.trunk[7], 16));
        if (sem) step256(L.trunk[7], K16, K0, kColA, 0, kColB, true, integral_constant<int, SAVE_H7>{}, pf128(L.sem1, 16));
        else     step256(L.trunk[7], K16, K0, kColA, 0, kColB, true, integral_constant<int, SAVE_H7>{}, pf256(L.as1, 16));

        // ---------------- heads ----------------
        const int my_pt = tile * kPts + 16 * wave + (lane & 15);
        const bool my_valid = my_pt < p.n_points;
        float* const out_row = p.raw + (size_t)(my_valid ? my_pt : 0) * p.channels;

        const f32x4 sig4 = skinny_gemm_h<8>(wb, L.alpha.w * 4, L.alpha.b * 4, (L.alpha.b + 16) * 4, xs + kColB, lane);

        if (sem) {
            step128(L.sem1, K16, K0, kColB, 0, kColA, true, nullptr, integral_constant<int, SAVE_SEMH>{}, pf256(L.as1, 16));
            for (int rb = 0; rb < L.sem_rbs; ++rb) {
                const f32x4 lg = skinny_gemm_h<4>(wb, (L.sem2.w + rb * 4 * 2 * 256) * 4, (L.sem2.b + 16 * rb) * 4,
                                                   (L.sem2.b + 16 * L.sem_rbs) * 4, xs + kColA, lane);
                const int ch0 = 16 * rb + 4 * (lane >> 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (my_valid && ch0 + i < p.n_classes) __builtin_nontemporal_store(lg[i], out_row + INERF_BASE_CHANNELS + ch0 + i);
            }
            __syncthreads();
        }

        step256(L.as1, K16, K0, kColB, 0, kColA, true, integral_constant<int, SAVE_AS1H>{}, pf256(L.feat, 16));
        const f32x4 as4 = skinny_gemm_h<8>(wb, L.as2.w * 4, L.as2.b * 4, (L.as2.b + 16) * 4, xs + kColA, lane);
        __syncthreads();

        step256(L.feat, K16, K0, kColB, 0, kColA, false, integral_constant<int, SAVE_FEAT>{}, pf128(L.views, 18));
        // endpoint feature (semantic_nerf.py:163-164): the fp32 views activation goes straight to raw
        float* ep = nullptr;
        if (kSsr && p.endpoint)
            ep = p.raw + (size_t)pt0 * p.channels + INERF_BASE_CHANNELS + p.n_classes + 32 * wave + 4 * (lane >> 5);
        step128(L.views, K16, K2, kColA, kColDir, kColB, true, ep, integral_constant<int, SAVE_VH>{}, pf256(L.trunk[0], 4));
        const f32x4 res4 = skinny_gemm_h<4>(wb, L.res.w * 4, L.res.b * 4, (L.res.b + 16) * 4, xs + kColB, lane);

        if (lane < 16 && my_valid) {
            const float a0 = sigmoid_ref_h(as4[0]), a1 = sigmoid_ref_h(as4[1]), a2 = sigmoid_ref_h(as4[2]);
            const float sh = sigmoid_ref_h(as4[3]);
            const float r0 = sigmoid_ref_h(res4[0]), r1 = sigmoid_ref_h(res4[1]), r2 = sigmoid_ref_h(res4[2]);
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a0, sh), r0), out_row + 0);          // run_nerf_helpers.py:320
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a1, sh), r1), out_row + 1);
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a2, sh), r2), out_row + 2);
            __builtin_nontemporal_store(sig4[0], out_row + 3);
            __builtin_nontemporal_store(a0, out_row + 4); __builtin_nontemporal_store(a1, out_row + 5); __builtin_nontemporal_store(a2, out_row + 6);
            __builtin_nontemporal_store(sh, out_row + 7);
            __builtin_nontemporal_store(r0, out_row + 8); __builtin_nontemporal_store(r1, out_row + 9); __builtin_nontemporal_store(r2, out_row + 10);
        }
        flag_f16_range(p, tile * kPts, kPts, amax, amax2, lane);
    }
    if (kSave && p.act_max) {             // bound of every saved activation (they were split as kActScale * value)
        float m = fmaxf(amax, fmaxf((float)amax2[0], (float)amax2[1])) * (1.0f / kActScale);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0 && m == m) atomicMax(reinterpret_cast<unsigned int*>(p.act_max), __builtin_bit_cast(unsigned int, m));
    }
}

// ================================================================================================
// Two workgroups per CU (object-level network; SSR network with its semantic head per wave, see sem_head).
//
// One wave per SIMD cannot hide anything: while it converts accumulators, waits on the weight stream or sits
// at a barrier, the matrix pipe idles (measured: 57 % MFMA-busy in the kernel above).  This variant halves
// the footprint of a workgroup - 75,776 B of LDS, <= 256 registers - so that two tiles at different stages
// share a CU and fill each other's gaps.  What had to go:
//   * the second activation buffer: a layer's output overwrites its input IN PLACE, which costs a barrier
//     between the GEMM and the store (accumulators wait in registers) besides the one after the store;
//   * the resident position encoding: layer 0 reads it from columns 0..63 of the activation buffer; the skip
//     layer runs its h-part first, then the encoding is recomputed into the (now dead) columns 0..63 and the
//     enc-part accumulates on top;
//   * LDS copies of the heads' hidden layers: the albedo/shading hidden layer and the view-dependent layer stay
//     in registers - a 32x32 accumulator, converted in place, IS the B operand of the next MFMA when the next
//     weights are stored in accumulator k order (layout.h: as2r / resr) - and each wave's partial output sums
//     (its 64 / 32 hidden channels) meet in a 128-byte per-point exchange area.
// LDS per workgroup: two planes (hi, lo) of X[64 points][296 halfs] = [h 256 | dir 32 | pad 8]; 592-byte rows
// put the 16 rows of a ds_read_b128 lane group on 16 distinct bank slots.
// ================================================================================================
// kSplit (SSR inference with a scratch buffer): the semantic hidden layer (128 channels) is split over the waves by channel
// HALF and point HALF - wave w computes channels 64 (w & 1) .. + 63 of points 32 (w >> 1) .. + 31: the 64-channel wave tile of
// the 256-wide layers (6 MFMAs per 2 LDS operand reads; the 32-channel x 64-point form of round 3 had 6 per 4 and ran at 34 %
// matrix-pipe busy), semantic_linear.0.0 streamed twice per tile instead of four times (in the per-wave form below the head's
// weight stream, 512 KB per tile through the CU's 64 B/clk vector memory path, takes 2.7x the cycles of its own MFMAs).  The
// price is that the logits then exist as TWO partial sums per point (one per channel half) while LDS is still full of h7: each
// wave parks ITS partials (16 accumulator registers per 32 classes) in a private, L2-resident scratch slot - an explicit
// spill, placed where nothing waits for it - and fetches them back at the end of the tile, when the activation planes are
// dead and the two partials of a point can meet in LDS.
template <bool kSave, bool kSsr, bool kSplit = false>
__global__ __launch_bounds__(256, 2) void k_encode_mlp_f16x3_dual(const MlpParams p) {
    static_assert(!kSplit || (kSsr && !kSave), "the channel-split semantic head is the SSR inference form");
    constexpr int kPts = kTilePoints;
    constexpr int kParts = 256 / kPts;
    // the wide GEMMs' first products take a zero C operand (wide_gemm_h PEEL): SSR frame +0.6-0.8 % same-box; the saving forms are
    // indifferent (1.85-1.87 vs 1.82-1.86 ms) and keep the explicit zeroing (profiles/r06_peel_ab.txt)
    constexpr bool kPeelD = INERF_DUAL_PEEL && !kSave;
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsd[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NetLayout& L = p.L;
    float amax = 0.0f;
    f16x2 amax2 = {(_Float16)0.0f, (_Float16)0.0f};

    _Float16* const xw = ldsd + (lane & 31) * kRowD;
    const _Float16* const xr = xw + 8 * (lane >> 5);                       // wide GEMM operand reads (+ column)
    _Float16* const xd = xw + 4 * (lane >> 5) + 64 * wave;                 // wide stores: this wave's 64 channels
    const _Float16* const xs = ldsd + (16 * wave + (lane & 15)) * kRowD + 8 * (lane >> 4);   // skinny operand reads

    WeightBuf wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wts), 0, L.total_floats * 4, 0x00020000);
    wb.voff = lane * 16;
    auto frag256 = [&](const GemmSlot& s, int kbt) { return (s.w + wave * kbt * 2 * 2 * 256) * 4; };
    auto frag128 = [&](const GemmSlot& s, int kbt) { return (s.w + wave * kbt * 1 * 2 * 256) * 4; };

    WidePreH<2> pre2;
    WidePreH<1> pre1;
    prefetch_w<2>(pre2, wb, frag256(L.trunk[0], 4));
#ifdef INERF_DEPHASE            // (experiment of a development build: one of a CU's two workgroups starts INERF_DEPHASE_UNITS x 2048 cycles late)
    if (INERF_DEPHASE == 1 ? blockIdx.x >= gridDim.x / 2 : (blockIdx.x & 1))
        for (int d = 0; d < INERF_DEPHASE_UNITS; ++d) __builtin_amdgcn_s_sleep(32);
#endif

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        // The per-tile slot / output offsets below are sums of a tile part and a lane part; derived from `lane` itself the lane
        // parts are loop invariants, hoisted out of the tile loop (~15 registers, spilled at this kernel's 256).  Laundered per tile.
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        if constexpr (!kSave) {                   // inference: the f16 range guard is per tile (flag_f16_range); training forwards keep the
            amax = 0.0f;                          // launch's maximum (act_max; their status is one word)
            amax2 = f16x2{(_Float16)0.0f, (_Float16)0.0f};
        }
        // ---------------- encode -> hi/lo planes (xyz: columns 0..63, dir: columns 256..287) ----------------
        auto encode = [&](bool with_dir) {
            // (the thread index is laundered per call: derived from `tid` itself, this stage's ~20 LDS / slot addresses are
            // loop invariants that the compiler hoists out of the tile loop and, at this kernel's 256 registers, spills)
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));
            const int pt = tid_o % kPts, part = tid_o / kPts;
            int gp = tile * kPts + pt;
            gp = gp < p.n_points ? gp : p.n_points - 1;
            const int ray = gp / p.n_samples;
            const float* __restrict__ r = p.rays + (size_t)ray * INERF_RAY_FLOATS;
            const float zz = __builtin_nontemporal_load(p.z + gp);
            _Float16* row = ldsd + pt * kRowD;
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
                    split_store<kPlaneD>(row + 3 + 6 * f + c, sn, amax);
                    split_store<kPlaneD>(row + 6 + 6 * f + c, cs, amax);
                }
            }
            if (part == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) split_store<kPlaneD>(row + c, x[c], amax);
                for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) { row[c] = (_Float16)0.0f; row[kPlaneD + c] = (_Float16)0.0f; }
            }
            if (with_dir) {
                const int fd = kParts - 1 - part;
                if (fd < p.l_dir) {
                    const float s = (float)(1 << fd);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float sn, cs;
                        fast_sincosf(r[8 + c] * s, &sn, &cs);
                        split_store<kPlaneD>(row + kColDirD + 3 + 6 * fd + c, sn, amax);
                        split_store<kPlaneD>(row + kColDirD + 6 + 6 * fd + c, cs, amax);
                    }
                }
                if (part == 3) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) split_store<kPlaneD>(row + kColDirD + c, r[8 + c], amax);
                    for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) { row[kColDirD + c] = (_Float16)0.0f; row[kPlaneD + kColDirD + c] = (_Float16)0.0f; }
                }
            }
        };
#ifdef INERF_ABL_NO_ENCODE      // (timing ablation of a development build: the first tile's encoding stays in LDS; results are wrong)
        if (tile == (int)blockIdx.x)
#endif
        encode(true);
        __syncthreads();
        // training forward: the encoding leaves as operand fragments of the products dW = dZ^T enc (pts_linears.0, and .5's
        // encoding columns), straight from the planes (columns 0..63, before the first layer's output lands there): a 64-channel
        // fragment slot, one channel block per wave 0 / 1; wave 2: the view encoding (32 channels, one block).  (Until round 4:
        // 96 four-byte stores per point at strides of 256 / 128 bytes.)
        if constexpr (kSave) {
            int lane_e = lane_t;
            asm volatile("" : "+v"(lane_e));
            if (wave < 2) {
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + p.save_off[SAVE_ENC], 0, (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 4)), 0x00020000);
                d.voff = (unsigned)tile * (unsigned)(kFragTileBytes / 4) + (unsigned)wave * (2u * kFragBytes) + (unsigned)lane_e * 16u;
                planes_to_frag<1, kRowD, kPlaneD, 2>(xr + 32 * wave, plane_selector(lane_e), d);
            } else if (wave == 2) {                 // the view encoding: a 32-channel fragment slot (one block per k-block)
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + p.save_off[SAVE_DIR], 0, (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 8)), 0x00020000);
                d.voff = (unsigned)tile * (unsigned)(kFragTileBytes / 8) + (unsigned)lane_e * 16u;
                planes_to_frag<1, kRowD, kPlaneD, 1>(xr + kColDirD, plane_selector(lane_e), d);
            }
        }

        // a 256-wide layer in place: GEMM over columns [0, 16*KBT) | barrier | store to columns [0, 256) | barrier
        f32x16 am2[2][2];
        f32x4 bias2[2][4];
        float inv2;
        const int pt0 = tile * kPts + (lane_t & 31);
        // ReLU masks of h0..h7 for the input-gradient chain (layout.h relu_bits_offset)
        const __amdgpu_buffer_rsrc_t bits_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.save + (kSave ? p.bits_off : 0), 0, kSave ? (int)((unsigned)p.n_tiles * (unsigned)kReluBitTileBytes) : 0, 0x00020000);
        // training forward: where a 256-wide layer goes besides the planes (layout.h SaveSlot) - fp32 rows from the epilogue's
        // registers cost 2.3 x as much per byte as the fragments' whole-line stores (16-byte pieces): no layer leaves that way any
        // more except the SSR semantic hidden layer;
        // `frag_slot`: operand fragments of the weight-gradient products, transposed out of the finished planes by the matrix
        // core (planes_to_frag: whole 1 KB stores); the ReLU masks of h0..h7 as bits.
        auto frag_dst = [&](int slot) {
            FragDst d;
            d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + (kSave ? p.save_off[slot] : 0), 0,
                                                       kSave ? (int)((unsigned)p.n_tiles * (unsigned)kFragTileBytes) : 0, 0x00020000);
            d.voff = (unsigned)tile * (unsigned)kFragTileBytes + (unsigned)(2 * wave) * (2u * kFragBytes) + (unsigned)lane_t * 16u;
            return d;
        };
        auto store256 = [&](const GemmSlot& s, bool relu, int frag_slot, auto&& prefetch_next, auto bits_tag) {
            load_bias<2>(bias2, inv2, wb, (s.b + 64 * wave) * 4, (s.b + kWidth) * 4, lane);
            prefetch_next();
            constexpr bool kBits = kSave && decltype(bits_tag)::value;
            const BitsDst bd = {bits_rsrc, kBits ? (((tile * kReluBitLayers + (frag_slot - SAVE_H0)) * 4 + wave) * 64 + lane_t) * 8 : 0};
#ifndef INERF_ABL_NO_BARRIER    // (timing ablation of a development build: the layer barriers gone - racy, results are wrong)
            __syncthreads();                       // every wave has read the layer's input
#endif
            wide_store_h<2, kRowD, kPlaneD, false, kBits, 2, !kSave>(am2, inv2, bias2, xd, relu, amax2, nullptr, 0, 0, 0, nullptr, &bd);
#ifndef INERF_ABL_NO_BARRIER
            __syncthreads();
#endif
            if constexpr (kSave)                   // this wave's 64 channels of all 64 points (the layer is complete behind the barrier)
                if (frag_slot >= 0) {
                    int lane_o = lane_t;           // (laundered: the selector is rebuilt per layer instead of living in 8 registers)
                    asm volatile("" : "+v"(lane_o));
                    planes_to_frag<2, kRowD, kPlaneD>(xr + 64 * wave, plane_selector(lane_o), frag_dst(frag_slot));
                }
        };
        constexpr std::true_type kWithBits{};
        constexpr std::false_type kNoBits{};
        auto pf256 = [&](const GemmSlot& s, int kbt) { return [&, kbt]() { prefetch_w<2>(pre2, wb, frag256(s, kbt)); }; };
        auto pf256_at = [&](const GemmSlot& s, int kbt, int kb_first) {
            return [&, kbt, kb_first]() { prefetch_w<2>(pre2, wb, frag256(s, kbt) + kb_first * 2 * 2 * 1024); };
        };
        auto pf128 = [&](const GemmSlot& s, int kbt) { return [&, kbt]() { prefetch_w<1>(pre1, wb, frag128(s, kbt)); }; };

        // ---------------- trunk ----------------
        wide_gemm_h<2, 4, 0, kRowD, kPlaneD, true, 4096, 2, 2048, kPeelD>(pre2, wb, frag256(L.trunk[0], 4), xr, 0, 0, lane, am2);
        store256(L.trunk[0], true, SAVE_H0, pf256(L.trunk[1], 16), kWithBits);
#pragma unroll 1
        for (int layer = 1; layer < kSkipInput; ++layer) {
            wide_gemm_h<2, 16, 0, kRowD, kPlaneD, true, 4096, 2, 2048, kPeelD>(pre2, wb, frag256(L.trunk[layer], 16), xr, 0, 0, lane, am2);
            if (layer + 1 < kSkipInput) store256(L.trunk[layer], true, SAVE_H0 + layer, pf256(L.trunk[layer + 1], 16), kWithBits);
            else                        store256(L.trunk[layer], true, SAVE_H0 + layer, pf256_at(L.trunk[kSkipInput], 20, 4), kWithBits);
        }
        {   // pts_linears[5] over cat([pts, h]): h-part (k-blocks 4..19 of the stream), then the encoding again
            const GemmSlot& s = L.trunk[kSkipInput];
            wide_gemm_h<2, 16, 0, kRowD, kPlaneD, true, 4096, 2, 2048, kPeelD>(pre2, wb, frag256(s, 20) + 4 * 2 * 2 * 1024, xr, 0, 0, lane, am2);
            prefetch_w<2>(pre2, wb, frag256(s, 20));
            __syncthreads();
            encode(false);
            __syncthreads();
            wide_gemm_h<2, 4, 0, kRowD, kPlaneD, false>(pre2, wb, frag256(s, 20), xr, 0, 0, lane, am2);
            store256(s, true, SAVE_H0 + kSkipInput, pf256(L.trunk[6], 16), kWithBits);
        }
        wide_gemm_h<2, 16, 0, kRowD, kPlaneD, true, 4096, 2, 2048, kPeelD>(pre2, wb, frag256(L.trunk[6], 16), xr, 0, 0, lane, am2);
        store256(L.trunk[6], true, SAVE_H0 + 6, pf256(L.trunk[7], 16), kWithBits);
        wide_gemm_h<2, 16, 0, kRowD, kPlaneD, true, 4096, 2, 2048, kPeelD>(pre2, wb, frag256(L.trunk[7], 16), xr, 0, 0, lane, am2);
        store256(L.trunk[7], true, SAVE_H7, pf256(L.as1, 16), kWithBits);

        // ---------------- heads ----------------
        const int my_pt = tile * kPts + 16 * wave + (lane_t & 15);
        const bool my_valid = my_pt < p.n_points;
        float* const out_row = p.raw + (size_t)(my_valid ? my_pt : 0) * p.channels;
        const f32x4 sig4 = skinny_gemm_h<8, kPlaneD>(wb, L.alpha.w * 4, L.alpha.b * 4, (L.alpha.b + 16) * 4, xs, lane);

        // albedo + shading: hidden layer (this wave: 64 of its 256 channels) -> registers -> partial output sums
        f32x4 part_as[2], part_res[2];
        WidePreH<2> pre2s;                           // kSplit: first fragments of the semantic hidden layer, requested under the albedo|shading head
        {
            wide_gemm_h<2, 16, 0, kRowD, kPlaneD, true, 4096, 2, 2048, kPeelD>(pre2, wb, frag256(L.as1, 16), xr, 0, 0, lane, am2);
            load_bias<2>(bias2, inv2, wb, (L.as1.b + 64 * wave) * 4, (L.as1.b + kWidth) * 4, lane);
            prefetch_w<2>(pre2, wb, frag256(L.feat, 16));
            if constexpr (kSplit) prefetch_w<2, 2048, 16 * 2048>(pre2s, wb, (L.sem1.w + 2 * (wave & 1) * 16 * 2 * 256) * 4);
            f16x8 hi[4][2], lo[4][2];
            to_operands<2, false, 2, !kSave>(am2, inv2, bias2, amax2, hi, lo);
            if constexpr (kSave) {                // the hidden layer as operand fragments, transposed out of the registers (it never touches LDS)
                int lane_o = lane_t;
                asm volatile("" : "+v"(lane_o));
                operands_to_frag<2>(hi, lo, accumulator_selector(lane_o), frag_dst(SAVE_AS1H));
            }
            regop_gemm<4>(wb, (L.as2r.w + wave * 4 * 2 * 256) * 4, hi, lo, part_as);
        }
        const __amdgpu_buffer_rsrc_t sem_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.sem_scratch, 0, kSplit ? (int)((unsigned)gridDim.x * (unsigned)L.sem_rb32 * (unsigned)kSemScratchBytes) : 0, 0x00020000);
        // kSplit: this wave's partial logits of classes 0..31 stay in REGISTERS until the exchange at the end of the tile (round 4 parked every
        // block in the scratch slot: the inference kernels had no 16 registers to spare before the epilogues' running maximum was pinned -
        // 252 -> 235; further blocks of a head with more than 32 classes still go through the slot)
        f32x16 sem_keep = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (kSplit) {                    // semantic_nerf.py:150-152, hidden layer split over the waves by channel half x point half
            const int ch = wave & 1, ph = wave >> 1;
            f32x16 am1[2][1];
            f32x4 bias1[2][4];
            float inv1;
            // this wave's two 32-channel streams of the layer's 4-wave packing (16 k-blocks x 2 KiB each)
            // (operand rows from the per-tile laundered lane index: as a loop invariant this address is one more spilled register)
            const _Float16* xr_half = ldsd + ((lane_t & 31) + 32 * ph) * kRowD + 8 * (lane_t >> 5);
            wide_gemm_h<2, 16, 0, kRowD, kPlaneD, true, 2048, 1, 16 * 2048, kPeelD>(pre2s, wb, (L.sem1.w + 2 * ch * 16 * 2 * 256) * 4, xr_half, 0, 0, lane, am1);
            load_bias<2>(bias1, inv1, wb, (L.sem1.b + 64 * ch) * 4, (L.sem1.b + kHalf) * 4, lane);
            f16x8 hi[4][1], lo[4][1];
            to_operands<2, false, 1>(am1, inv1, bias1, amax2, hi, lo, nullptr);
#pragma unroll 1
            for (int rb = 0; rb < L.sem_rb32; ++rb) {      // this wave's partial logits of classes 32 rb .. +31, its 32 points -> its scratch slot
                f32x16 acc[1];
                // sem2q fragments of hidden channels 64 ch .. + 63 = packing waves 2 ch, 2 ch + 1 (two k-blocks each, contiguous)
                regop_gemm_full<4, 1>(wb, (L.sem2q.w + (rb * 4 + 2 * ch) * 2 * 2 * 256) * 4, hi, lo, acc);
                if (rb == 0) { sem_keep = acc[0]; continue; }
                const int slot = ((int)blockIdx.x * L.sem_rb32 + rb) * kSemScratchBytes + wave * (kSemScratchBytes / 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[0][4 * g], acc[0][4 * g + 1], acc[0][4 * g + 2], acc[0][4 * g + 3]};
                    // (whole offset in the VGPR operand: a 16-byte buffer store with a register SGPR offset gets no hazard wait state, tests/test_isa_audit_cpu.py)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sem_rsrc, lane_t * 16 + slot + g * 1024, 0, 0);
                }
            }
        } else if (kSsr && L.sem_rbs > 0) {        // semantic logits straight to raw[11 .. 11+C), every wave the whole head for its 16 points
            SaveDst sv;
            sv.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + (kSave ? p.save_off[SAVE_SEMH] : 0), 0,
                                                        kSave ? (int)((unsigned)p.n_points * (unsigned)kHalf * 4u) : 0, 0x00020000);
            sv.voff = (int)(((unsigned)(tile * kPts + 16 * wave + (lane_t & 15)) * (unsigned)kHalf + (unsigned)(4 * (lane_t >> 4))) * 4u);
            sv.stride = kHalf;
            sem_head<kSave>(wb, L, xs, lane, amax2, out_row, my_valid, p.n_classes, &sv);
        }
        // feature (no activation) in place of h7, then the view-dependent layer over [feature | dir] -> registers
        wide_gemm_h<2, 16, 0, kRowD, kPlaneD, true, 4096, 2, 2048, kPeelD>(pre2, wb, frag256(L.feat, 16), xr, 0, 0, lane, am2);
        store256(L.feat, false, SAVE_FEAT, pf128(L.views, 18), kNoBits);
        {
            f32x16 am1[1][2];
            f32x4 bias1[1][4];
            float inv1;
            wide_gemm_h<1, 18, 0, kRowD, kPlaneD, true, 2048, 2, 2048, kPeelD>(pre1, wb, frag128(L.views, 18), xr, 0, 0, lane, am1);
            load_bias<1>(bias1, inv1, wb, (L.views.b + 32 * wave) * 4, (L.views.b + kHalf) * 4, lane);
            prefetch_w<2>(pre2, wb, frag256(L.trunk[0], 4));
            f16x8 hi[2][2], lo[2][2];
            to_operands<1, false, 2, !kSave>(am1, inv1, bias1, amax2, hi, lo);
            if constexpr (kSave) {                // 128 channels: a four-block fragment slot, this wave's block
                int lane_o = lane_t;
                asm volatile("" : "+v"(lane_o));
                FragDst d;
                d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + p.save_off[SAVE_VH], 0, (int)((unsigned)p.n_tiles * (unsigned)(kFragTileBytes / 2)), 0x00020000);
                d.voff = (unsigned)tile * (unsigned)(kFragTileBytes / 2) + (unsigned)wave * (2u * kFragBytes) + (unsigned)lane_o * 16u;
                operands_to_frag<1, 4>(hi, lo, accumulator_selector(lane_o), d);
            }
            regop_gemm<2>(wb, (L.resr.w + wave * 2 * 2 * 256) * 4, hi, lo, part_res);
        }
        // kSplit: the partial logits meet in dead columns too - per point 2 x 32 floats (one per channel half): half 0 at bytes
        // 256..383 of the row in the hi plane, half 1 at bytes 128..255 in the lo plane (clear of the heads' exchange area, bytes
        // 128..255 of the hi plane, and of what the next tile's encode writes before its first barriers: bytes 0..127 and 512..575).
        // Block 0 comes from the wave's registers (sem_keep); of further blocks each wave fetches its OWN partials back from its
        // scratch slot (sc0: past the vector L1, whose lines of an earlier tile may be stale).
        constexpr int kRowF = kRowD / 2;                                      // floats per LDS row
        auto sem_ex = [&](int half, int r) { return reinterpret_cast<float*>(ldsd) + (half == 0 ? 64 : kPlaneD / 2 + 32) + r * kRowF; };
        auto sem_fetch = [&](int rb, u32x4 (&v)[4]) {
            const int slot = ((int)blockIdx.x * L.sem_rb32 + rb) * kSemScratchBytes + wave * (kSemScratchBytes / 4);
#pragma unroll
            for (int g = 0; g < 4; ++g) v[g] = __builtin_amdgcn_raw_buffer_load_b128(sem_rsrc, lane_t * 16 + slot + g * 1024, 0, 1);
        };
        auto sem_post = [&](const u32x4 (&v)[4]) {                            // accumulator register 4g + i = class 8g + 4 (lane >> 5) + i of point (lane & 31) of this wave's half
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x4*>(sem_ex(wave & 1, 32 * (wave >> 1) + (lane_t & 31)) + 8 * g + 4 * (lane_t >> 5)) = v[g];
        };
        auto sem_sum = [&](int rb, float inv2) {                              // this wave's 16 points x 32 classes: lane = (point, 8 classes)
            const int r = 16 * wave + (lane_t & 15), c8 = 8 * (lane_t >> 4);
            // fixed order: deterministic
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sem_ex(0, r) + c8) + *reinterpret_cast<const f32x4*>(sem_ex(1, r) + c8);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(sem_ex(0, r) + c8 + 4) + *reinterpret_cast<const f32x4*>(sem_ex(1, r) + c8 + 4);
            const int c0 = 32 * rb + c8;
            const f32x4 b0 = wb.vec4((L.sem2.b + c0) * 4, 0), b1 = wb.vec4((L.sem2.b + c0 + 4) * 4, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (my_valid && c0 + i < p.n_classes) __builtin_nontemporal_store(__builtin_fmaf(s0[i], inv2, b0[i]), out_row + INERF_BASE_CHANNELS + c0 + i);
                if (my_valid && c0 + 4 + i < p.n_classes) __builtin_nontemporal_store(__builtin_fmaf(s1[i], inv2, b1[i]), out_row + INERF_BASE_CHANNELS + c0 + 4 + i);
            }
        };
        u32x4 sem_v[4];
        float sem_inv2 = 0.0f;
        if constexpr (kSplit) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                sem_v[g] = __builtin_bit_cast(u32x4, f32x4{sem_keep[4 * g], sem_keep[4 * g + 1], sem_keep[4 * g + 2], sem_keep[4 * g + 3]});
            sem_inv2 = wb.scalar((L.sem2.b + 16 * L.sem_rbs) * 4);
        }
        __syncthreads();                           // feature / dir columns are dead: exchange areas may be written
        if (lane_t < 32) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                float* ex = reinterpret_cast<float*>(ldsd + (lane_t + 32 * pb) * kRowD + kColExD) + 8 * wave;
                *reinterpret_cast<f32x4*>(ex) = part_as[pb];
                *reinterpret_cast<f32x4*>(ex + 4) = part_res[pb];
            }
        }
        if constexpr (kSplit) sem_post(sem_v);
        __syncthreads();
        if constexpr (kSplit) {
            sem_sum(0, sem_inv2);
#pragma unroll 1
            for (int rb = 1; rb < L.sem_rb32; ++rb) {      // more than 32 classes: one block at a time through the same area
                sem_fetch(rb, sem_v);
                __syncthreads();                           // the previous block's sums have been read
                sem_post(sem_v);
                __syncthreads();
                sem_sum(rb, sem_inv2);
            }
            if (L.sem_rb32 > 1) __syncthreads();           // (keeps the area's last readers ahead of the next tile's fetch-and-post)
        }
        // Object-level network, whole tile: a wave's 16 points x 11 floats are 704 CONTIGUOUS bytes of raw.  They leave as 44 sixteen-byte
        // pieces (lanes 0..43, via 44 bytes per point in dead columns of the lo plane) instead of 11 four-byte pieces per point: every
        // 64-byte sector is written once, by one instruction.  (As 4-byte non-temporal pieces the rows cost 1.05-1.20 x their bytes in
        // HBM writes, depending on how far the waves that share a 128-byte line had drifted apart: profiles/r05_pmc_digest.txt.)
        const bool whole_rows = !kSsr && p.channels == INERF_BASE_CHANNELS && tile * kPts + kPts <= p.n_points &&
                                (reinterpret_cast<uintptr_t>(p.raw) & 15) == 0;
        auto stage_row = [&](int r) { return reinterpret_cast<float*>(ldsd + kPlaneD + r * kRowD + 128); };      // lo plane, bytes 256..299 of row r
        if (lane_t < 16 && my_valid) {
            const float* ex = reinterpret_cast<const float*>(ldsd + (16 * wave + lane_t) * kRowD + kColExD);
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
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.raw + (size_t)(tile * kPts + 16 * wave) * INERF_BASE_CHANNELS) + lane_t);
        }
        flag_f16_range(p, tile * kPts, kPts, amax, amax2, lane_t);
    }
    if (kSave && p.act_max) {
        float m = fmaxf(amax, fmaxf((float)amax2[0], (float)amax2[1])) * (1.0f / kActScale);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0 && m == m) atomicMax(reinterpret_cast<unsigned int*>(p.act_max), __builtin_bit_cast(unsigned int, m));
    }
}

// Which inference launches take the 128-point tile: the object-level network unless a 64-point form is asked for (A/B runs, the forms' own
// tests); the SSR network (at most 32 classes, no endpoint feature: one 32-class block of partial logits per wave) only on request
// (INERF_F16_KERNEL=t128) - same box, C = 28: 410.5 / 411.5 / 410.1 TFLOP/s against 414.4 / 414.9 / 414.6 for the two-workgroup kernel with
// the channel-split head, frames 69.0 vs 68.5 ms (profiles/r06_ssr_t128_ab.txt): there the second workgroup's overlap is worth more than the
// halved weight stream.
bool mlp_f16x3_takes_t128(bool ssr, bool save, bool endpoint, int n_classes) {
    const char* form = getenv("INERF_F16_KERNEL");
    if (endpoint || (form && (form[0] == 'd' || form[0] == 's' || form[0] == 'w' || form[0] == 'c'))) return false;
    if (save) {      // training forward: the object-level network's saving form of the 128-point tile (same save buffer, bit for bit)
        const char* tf = getenv("INERF_TRAIN_FWD");
        return !ssr && tf && tf[0] == 't';
    }
    return !ssr || (form && form[0] == 't' && n_classes <= 32);
}

int64_t sem_scratch_bytes(const inerf_net_desc& net, int64_t n_points, bool endpoint) {
    const char* form = getenv("INERF_F16_KERNEL");
    if (net.precision == INERF_PREC_F16X3 && net.variant == INERF_VARIANT_OBJECT && n_points > 0 && mlp_f16x3_takes_t128(false, false, false, 0)) {
        // INERF_ENC_CACHE=1 parks the position encoding in the workspace instead of evaluating the encoder a second time per tile: +0.7 % -
        // and 6.6 x the kernel's HBM traffic (32 KB written and read back per 128 points: the slots do not survive in an L2 that also holds
        // the 2.7 MB of weights; profiles/r06_enc_cache_traffic.txt).  Off by default: the kernel's traffic stays 1.00 x its algorithmic bytes.
        const char* ec = getenv("INERF_ENC_CACHE");
        return ec && ec[0] == '1' ? enc_cache_bytes_t128(n_points) : 0;
    }          // the 128-point tile parks its position encoding for the skip layer (mlp_f16_t128.hip)
    if (net.precision != INERF_PREC_F16X3 || net.variant != INERF_VARIANT_SSR || net.n_classes <= 0 || endpoint || n_points <= 0) return 0;
    if (mlp_f16x3_takes_t128(true, false, endpoint, net.n_classes)) return sem_scratch_bytes_t128(n_points);     // the 128-point tile: 8 KiB per wave
    if (form && (form[0] == 's' || form[0] == 'w')) return 0;          // single: one workgroup per CU; wave: the per-wave head (A/B runs)
    // More than 32 classes go block by block through the exchange area (two barriers and an exposed scratch fetch per extra block):
    // measured at C = 101 the channel-split head is 8 % SLOWER than the per-wave one (27.8 vs 25.6 ms per 32768 x 192 launch), at
    // C = 28 it is 2.5 % faster (22.3 vs 22.9 ms) - profiles/r03_mlp_sem_head_forms.txt.  INERF_F16_KERNEL=csplit forces it (tests).
    if (net.n_classes > 32 && !(form && form[0] == 'c')) return 0;
    const int64_t tiles = (n_points + kTilePoints - 1) / kTilePoints;
    const int64_t grid = tiles < 2 * device_cus() ? tiles : 2 * device_cus();
    return grid * ((net.n_classes + 31) / 32) * kSemScratchBytes;
}

static int launch_dual(MlpParams& p, int64_t n_points, bool ssr, hipStream_t stream) {
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    const int max_grid = 2 * device_cus();
    const int grid = p.n_tiles < max_grid ? p.n_tiles : max_grid;
    const bool save = p.save != nullptr;
    const bool split = ssr && !save && p.sem_scratch && p.L.sem_rbs > 0;
    void (*kern)(const MlpParams) = split ? k_encode_mlp_f16x3_dual<false, true, true>
                                  : ssr ? (save ? k_encode_mlp_f16x3_dual<true, true> : k_encode_mlp_f16x3_dual<false, true>)
                                        : (save ? k_encode_mlp_f16x3_dual<true, false> : k_encode_mlp_f16x3_dual<false, false>);
    static PerDeviceOnce attr_set[5];
    const int variant = split ? 4 : 2 * (int)ssr + (int)save;
    if (attr_set[variant].first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesD);
        if (e != hipSuccess) return record(e);
        attr_set[variant].mark();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLdsBytesD, stream, p);
    return record(hipGetLastError());
}

int launch_mlp_f16x3(MlpParams& p, int64_t n_points, bool ssr, hipStream_t stream) {
    // Two workgroups per CU, except for the SSR network with its endpoint feature (the view layer's activation never
    // reaches memory in that form).  INERF_F16_KERNEL=single keeps the one-workgroup kernel everywhere (A/B runs).
    // (The SSR training forward takes 10.1-10.3 ms per step in either form.)
    const char* form = getenv("INERF_F16_KERNEL");
    // inference: the 128-point tile (mlp_f16_t128.hip; object-level: bit-identical to the two-workgroup kernel, 0.58 x its L2 -> CU
    // weight stream, +1.3-1.7 % same-box: profiles/r06_weight_stream_ab.txt).  INERF_F16_KERNEL=dual|single select the 64-point forms.
    // (the SSR form parks its partial logits in the caller's scratch: inerf_encode_mlp, which has none, takes the 64-point per-wave head)
    if (mlp_f16x3_takes_t128(ssr, p.save != nullptr, p.endpoint != 0, p.n_classes) && (!ssr || p.L.sem_rbs == 0 || p.sem_scratch))
        return launch_mlp_f16x3_t128(p, n_points, ssr, stream);
    if (!(ssr && p.endpoint) && !(form && form[0] == 's')) return launch_dual(p, n_points, ssr, stream);
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    const int grid = p.n_tiles < device_cus() ? p.n_tiles : device_cus();
    const bool save = p.save != nullptr;
    void (*kern)(const MlpParams) = ssr ? (save ? k_encode_mlp_f16x3<true, true> : k_encode_mlp_f16x3<true, false>)
                                        : (save ? k_encode_mlp_f16x3<false, true> : k_encode_mlp_f16x3<false, false>);
    static PerDeviceOnce attr_set[4];
    const int variant = 2 * (int)ssr + (int)save;
    if (attr_set[variant].first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kLdsBytesH);
        if (e != hipSuccess) return record(e);
        attr_set[variant].mark();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLdsBytesH, stream, p);
    return record(hipGetLastError());
}

}  // namespace inerf
