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
    auto enc_save = [&](int slot, int cols, int gp, bool ok) {          // training copy of a point's encoding row (EncSave)
        EncSave e;
        e.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + (kSave ? p.save_off[slot] : 0), 0,
                                                   kSave ? (int)((unsigned)p.n_points * (unsigned)cols * 4u) : 0, 0x00020000);
        e.voff = ok ? gp * cols * 4 : EncSave::kDropOffset;
        return e;
    };

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        // ---------------- encode -> hi/lo planes ----------------
        {
            const int pt = tid % kPts, part = tid / kPts;
            int gp = tile * kPts + pt;
            gp = gp < p.n_points ? gp : p.n_points - 1;
            const int ray = gp / p.n_samples;
            const float* __restrict__ r = p.rays + (size_t)ray * INERF_RAY_FLOATS;
            const float zz = __builtin_nontemporal_load(p.z + gp);     // streamed once: keep it out of the L2 the weights live in
            _Float16* row = ldsh + pt * kRowH;
            const bool sv_ok = kSave && tile * kPts + pt < p.n_points;
            const EncSave sv_enc = enc_save(SAVE_ENC, kEncCols, gp, sv_ok), sv_dir = enc_save(SAVE_DIR, kDirCols, gp, sv_ok);
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
                    if (kSave) { sv_enc.put(3 + 6 * f + c, sn); sv_enc.put(6 + 6 * f + c, cs); }
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
                    if (kSave) { sv_dir.put(3 + 6 * fd + c, sn); sv_dir.put(6 + 6 * fd + c, cs); }
                }
            }
            if (part == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) split_store(row + kColEnc + c, x[c], amax);
                for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) { row[kColEnc + c] = (_Float16)0.0f; row[kPlaneH + kColEnc + c] = (_Float16)0.0f; }
                if (kSave) {
                    for (int c = 0; c < 3; ++c) sv_enc.put(c, x[c]);
                    for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) sv_enc.put(c, 0.0f);
                }
            }
            if (part == 3) {
#pragma unroll
                for (int c = 0; c < 3; ++c) split_store(row + kColDir + c, v[c], amax);
                for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) { row[kColDir + c] = (_Float16)0.0f; row[kPlaneH + kColDir + c] = (_Float16)0.0f; }
                if (kSave) {
                    for (int c = 0; c < 3; ++c) sv_dir.put(c, v[c]);
                    for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) sv_dir.put(c, 0.0f);
                }
            }
        }
        __syncthreads();

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
        // ReLU masks of h0..h6 for the input-gradient chain (layout.h relu_bits_offset): one descriptor for the area
        const __amdgpu_buffer_rsrc_t bits_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.save + (kSave ? p.bits_off : 0), 0, kSave ? (int)((unsigned)p.n_tiles * (unsigned)kReluBitTileBytes) : 0, 0x00020000);
        auto step256 = [&](const GemmSlot& s, auto kb0c, auto kb1c, int c0, int c1, int dcol, bool relu, int slot,
                           auto&& prefetch_next) {
            constexpr int KB0 = decltype(kb0c)::value, KB1 = decltype(kb1c)::value;
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
            if (kSave && slot < SAVE_H7) {          // (a compile-time constant at every call site)
                const BitsDst bd = {bits_rsrc, (((tile * kReluBitLayers + (slot - SAVE_H0)) * 4 + wave) * 64 + lane) * 8};
                wide_store_h<2, kRowH, kPlaneH, kSave, kSave>(am, inv, bias, xd + dcol + 64 * wave, relu, amax2, nullptr, 0, 0, 0, &sv, &bd);
            } else {
                wide_store_h<2, kRowH, kPlaneH, kSave>(am, inv, bias, xd + dcol + 64 * wave, relu, amax2, nullptr, 0, 0, 0, &sv);
            }
            __syncthreads();
        };
        auto step128 = [&](const GemmSlot& s, auto kb0c, auto kb1c, int c0, int c1, int dcol, bool relu, float* gout, int slot,
                           auto&& prefetch_next) {
            constexpr int KB0 = decltype(kb0c)::value, KB1 = decltype(kb1c)::value;
            f32x16 am[1][2];
            f32x4 bias[1][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[0][g] = pre1.b[0][g];
            const float inv = pre1.inv;
            wide_gemm_h<1, KB0, KB1>(pre1, wb, frag128(s, KB0 + KB1), xr, c0, c1, lane, am);
            prefetch_next();
            const SaveDst sv = save_dst(slot, kHalf, 32 * wave);
            wide_store_h<1, kRowH, kPlaneH, kSave>(am, inv, bias, xd + dcol + 32 * wave, relu, amax2, gout, p.channels, pt0 < p.n_points,
                                                  pt0 + 32 < p.n_points, &sv);
            __syncthreads();
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
        step256(L.trunk[0], K4, K0, kColEnc, 0, kColA, true, SAVE_H0 + 0, pf256(L.trunk[1], 16));
        step256(L.trunk[1], K16, K0, kColA, 0, kColB, true, SAVE_H0 + 1, pf256(L.trunk[2], 16));
        step256(L.trunk[2], K16, K0, kColB, 0, kColA, true, SAVE_H0 + 2, pf256(L.trunk[3], 16));
        step256(L.trunk[3], K16, K0, kColA, 0, kColB, true, SAVE_H0 + 3, pf256(L.trunk[4], 16));
        step256(L.trunk[4], K16, K0, kColB, 0, kColA, true, SAVE_H0 + 4, pf256(L.trunk[5], 20));
        step256(L.trunk[5], K4, K16, kColEnc, kColA, kColB, true, SAVE_H0 + 5, pf256(L.trunk[6], 16));
        step256(L.trunk[6], K16, K0, kColB, 0, kColA, true, SAVE_H0 + 6, pf256(L.trunk[7], 16));
        if (sem) step256(L.trunk[7], K16, K0, kColA, 0, kColB, true, SAVE_H7, pf128(L.sem1, 16));
        else     step256(L.trunk[7], K16, K0, kColA, 0, kColB, true, SAVE_H7, pf256(L.as1, 16));

        // ---------------- heads ----------------
        const int my_pt = tile * kPts + 16 * wave + (lane & 15);
        const bool my_valid = my_pt < p.n_points;
        float* const out_row = p.raw + (size_t)(my_valid ? my_pt : 0) * p.channels;

        const f32x4 sig4 = skinny_gemm_h<8>(wb, L.alpha.w * 4, L.alpha.b * 4, (L.alpha.b + 16) * 4, xs + kColB, lane);

        if (sem) {
            step128(L.sem1, K16, K0, kColB, 0, kColA, true, nullptr, SAVE_SEMH, pf256(L.as1, 16));
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

        step256(L.as1, K16, K0, kColB, 0, kColA, true, SAVE_AS1H, pf256(L.feat, 16));
        const f32x4 as4 = skinny_gemm_h<8>(wb, L.as2.w * 4, L.as2.b * 4, (L.as2.b + 16) * 4, xs + kColA, lane);
        __syncthreads();

        step256(L.feat, K16, K0, kColB, 0, kColA, false, SAVE_FEAT, pf128(L.views, 18));
        // endpoint feature (semantic_nerf.py:163-164): the fp32 views activation goes straight to raw
        float* ep = nullptr;
        if (kSsr && p.endpoint)
            ep = p.raw + (size_t)pt0 * p.channels + INERF_BASE_CHANNELS + p.n_classes + 32 * wave + 4 * (lane >> 5);
        step128(L.views, K16, K2, kColA, kColDir, kColB, true, ep, SAVE_VH, pf256(L.trunk[0], 4));
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
    }
    const float amax_all = fmaxf(amax, fmaxf((float)amax2[0], (float)amax2[1]));
    if (p.status && __any(!(amax_all <= kF16Safe)) && lane == 0) atomicOr(p.status, INERF_STATUS_F16_RANGE);
    if (kSave && p.act_max) {             // bound of every saved activation (they were split as kActScale * value)
        float m = amax_all * (1.0f / kActScale);
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
template <bool kSave, bool kSsr>
__global__ __launch_bounds__(256, 2) void k_encode_mlp_f16x3_dual(const MlpParams p) {
    constexpr int kPts = kTilePoints;
    constexpr int kParts = 256 / kPts;
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
    auto enc_save = [&](int slot, int cols, int gp, bool ok) {          // training copy of a point's encoding row (EncSave)
        EncSave e;
        e.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + (kSave ? p.save_off[slot] : 0), 0,
                                                   kSave ? (int)((unsigned)p.n_points * (unsigned)cols * 4u) : 0, 0x00020000);
        e.voff = ok ? gp * cols * 4 : EncSave::kDropOffset;
        return e;
    };

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        // ---------------- encode -> hi/lo planes (xyz: columns 0..63, dir: columns 256..287) ----------------
        auto encode = [&](bool with_dir) {
            const int pt = tid % kPts, part = tid / kPts;
            int gp = tile * kPts + pt;
            gp = gp < p.n_points ? gp : p.n_points - 1;
            const int ray = gp / p.n_samples;
            const float* __restrict__ r = p.rays + (size_t)ray * INERF_RAY_FLOATS;
            const float zz = __builtin_nontemporal_load(p.z + gp);
            _Float16* row = ldsd + pt * kRowD;
            const bool sv_ok = kSave && with_dir && tile * kPts + pt < p.n_points;       // the skip layer's second pass re-computes only
            const EncSave sv_enc = enc_save(SAVE_ENC, kEncCols, gp, sv_ok), sv_dir = enc_save(SAVE_DIR, kDirCols, gp, sv_ok);
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
                    if (kSave) { sv_enc.put(3 + 6 * f + c, sn); sv_enc.put(6 + 6 * f + c, cs); }
                }
            }
            if (part == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) split_store<kPlaneD>(row + c, x[c], amax);
                for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) { row[c] = (_Float16)0.0f; row[kPlaneD + c] = (_Float16)0.0f; }
                if (kSave) {
                    for (int c = 0; c < 3; ++c) sv_enc.put(c, x[c]);
                    for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) sv_enc.put(c, 0.0f);
                }
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
                        if (kSave) { sv_dir.put(3 + 6 * fd + c, sn); sv_dir.put(6 + 6 * fd + c, cs); }
                    }
                }
                if (part == 3) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) split_store<kPlaneD>(row + kColDirD + c, r[8 + c], amax);
                    for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) { row[kColDirD + c] = (_Float16)0.0f; row[kPlaneD + kColDirD + c] = (_Float16)0.0f; }
                    if (kSave) {
                        for (int c = 0; c < 3; ++c) sv_dir.put(c, r[8 + c]);
                        for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) sv_dir.put(c, 0.0f);
                    }
                }
            }
        };
        encode(true);
        __syncthreads();

        // a 256-wide layer in place: GEMM over columns [0, 16*KBT) | barrier | store to columns [0, 256) | barrier
        f32x16 am2[2][2];
        f32x4 bias2[2][4];
        float inv2;
        const int pt0 = tile * kPts + (lane & 31);
        auto save_dst = [&](int slot, int width, int chan0) {
            SaveDst d;
            d.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + (kSave ? p.save_off[slot] : 0), 0,
                                                       kSave ? (int)((unsigned)p.n_points * (unsigned)width * 4u) : 0, 0x00020000);
            d.voff = (int)(((unsigned)pt0 * (unsigned)width + (unsigned)(chan0 + 4 * (lane >> 5))) * 4u);
            d.stride = width;
            return d;
        };
        // ReLU masks of h0..h6 for the input-gradient chain (layout.h relu_bits_offset)
        const __amdgpu_buffer_rsrc_t bits_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.save + (kSave ? p.bits_off : 0), 0, kSave ? (int)((unsigned)p.n_tiles * (unsigned)kReluBitTileBytes) : 0, 0x00020000);
        auto store256 = [&](const GemmSlot& s, bool relu, int slot, auto&& prefetch_next, auto bits_tag) {
            load_bias<2>(bias2, inv2, wb, (s.b + 64 * wave) * 4, (s.b + kWidth) * 4, lane);
            prefetch_next();
            const SaveDst sv = save_dst(slot, kWidth, 64 * wave);
            constexpr bool kBits = kSave && decltype(bits_tag)::value;
            const BitsDst bd = {bits_rsrc, kBits ? (((tile * kReluBitLayers + (slot - SAVE_H0)) * 4 + wave) * 64 + lane) * 8 : 0};
            __syncthreads();                       // every wave has read the layer's input
            wide_store_h<2, kRowD, kPlaneD, kSave, kBits>(am2, inv2, bias2, xd, relu, amax2, nullptr, 0, 0, 0, &sv, &bd);
            __syncthreads();
        };
        constexpr std::true_type kWithBits{};
        constexpr std::false_type kNoBits{};
        auto pf256 = [&](const GemmSlot& s, int kbt) { return [&, kbt]() { prefetch_w<2>(pre2, wb, frag256(s, kbt)); }; };
        auto pf256_at = [&](const GemmSlot& s, int kbt, int kb_first) {
            return [&, kbt, kb_first]() { prefetch_w<2>(pre2, wb, frag256(s, kbt) + kb_first * 2 * 2 * 1024); };
        };
        auto pf128 = [&](const GemmSlot& s, int kbt) { return [&, kbt]() { prefetch_w<1>(pre1, wb, frag128(s, kbt)); }; };

        // ---------------- trunk ----------------
        wide_gemm_h<2, 4, 0, kRowD, kPlaneD>(pre2, wb, frag256(L.trunk[0], 4), xr, 0, 0, lane, am2);
        store256(L.trunk[0], true, SAVE_H0, pf256(L.trunk[1], 16), kWithBits);
#pragma unroll 1
        for (int layer = 1; layer < kSkipInput; ++layer) {
            wide_gemm_h<2, 16, 0, kRowD, kPlaneD>(pre2, wb, frag256(L.trunk[layer], 16), xr, 0, 0, lane, am2);
            if (layer + 1 < kSkipInput) store256(L.trunk[layer], true, SAVE_H0 + layer, pf256(L.trunk[layer + 1], 16), kWithBits);
            else                        store256(L.trunk[layer], true, SAVE_H0 + layer, pf256_at(L.trunk[kSkipInput], 20, 4), kWithBits);
        }
        {   // pts_linears[5] over cat([pts, h]): h-part (k-blocks 4..19 of the stream), then the encoding again
            const GemmSlot& s = L.trunk[kSkipInput];
            wide_gemm_h<2, 16, 0, kRowD, kPlaneD>(pre2, wb, frag256(s, 20) + 4 * 2 * 2 * 1024, xr, 0, 0, lane, am2);
            prefetch_w<2>(pre2, wb, frag256(s, 20));
            __syncthreads();
            encode(false);
            __syncthreads();
            wide_gemm_h<2, 4, 0, kRowD, kPlaneD, false>(pre2, wb, frag256(s, 20), xr, 0, 0, lane, am2);
            store256(s, true, SAVE_H0 + kSkipInput, pf256(L.trunk[6], 16), kWithBits);
        }
        wide_gemm_h<2, 16, 0, kRowD, kPlaneD>(pre2, wb, frag256(L.trunk[6], 16), xr, 0, 0, lane, am2);
        store256(L.trunk[6], true, SAVE_H0 + 6, pf256(L.trunk[7], 16), kWithBits);
        wide_gemm_h<2, 16, 0, kRowD, kPlaneD>(pre2, wb, frag256(L.trunk[7], 16), xr, 0, 0, lane, am2);
        store256(L.trunk[7], true, SAVE_H7, pf256(L.as1, 16), kNoBits);

        // ---------------- heads ----------------
        const int my_pt = tile * kPts + 16 * wave + (lane & 15);
        const bool my_valid = my_pt < p.n_points;
        float* const out_row = p.raw + (size_t)(my_valid ? my_pt : 0) * p.channels;
        const f32x4 sig4 = skinny_gemm_h<8, kPlaneD>(wb, L.alpha.w * 4, L.alpha.b * 4, (L.alpha.b + 16) * 4, xs, lane);

        // albedo + shading: hidden layer (this wave: 64 of its 256 channels) -> registers -> partial output sums
        f32x4 part_as[2], part_res[2];
        {
            wide_gemm_h<2, 16, 0, kRowD, kPlaneD>(pre2, wb, frag256(L.as1, 16), xr, 0, 0, lane, am2);
            load_bias<2>(bias2, inv2, wb, (L.as1.b + 64 * wave) * 4, (L.as1.b + kWidth) * 4, lane);
            prefetch_w<2>(pre2, wb, frag256(L.feat, 16));
            f16x8 hi[4][2], lo[4][2];
            const SaveDst sv = save_dst(SAVE_AS1H, kWidth, 64 * wave);
            to_operands<2, kSave>(am2, inv2, bias2, amax2, hi, lo, &sv);
            regop_gemm<4>(wb, (L.as2r.w + wave * 4 * 2 * 256) * 4, hi, lo, part_as);
        }
        if (kSsr && L.sem_rbs > 0) {               // semantic logits straight to raw[11 .. 11+C) (semantic_nerf.py:150-152)
            SaveDst sv;
            sv.rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save + (kSave ? p.save_off[SAVE_SEMH] : 0), 0,
                                                        kSave ? (int)((unsigned)p.n_points * (unsigned)kHalf * 4u) : 0, 0x00020000);
            sv.voff = (int)(((unsigned)(tile * kPts + 16 * wave + (lane & 15)) * (unsigned)kHalf + (unsigned)(4 * (lane >> 4))) * 4u);
            sv.stride = kHalf;
            sem_head<kSave>(wb, L, xs, lane, amax2, out_row, my_valid, p.n_classes, &sv);
        }
        // feature (no activation) in place of h7, then the view-dependent layer over [feature | dir] -> registers
        wide_gemm_h<2, 16, 0, kRowD, kPlaneD>(pre2, wb, frag256(L.feat, 16), xr, 0, 0, lane, am2);
        store256(L.feat, false, SAVE_FEAT, pf128(L.views, 18), kNoBits);
        {
            f32x16 am1[1][2];
            f32x4 bias1[1][4];
            float inv1;
            wide_gemm_h<1, 18, 0, kRowD, kPlaneD>(pre1, wb, frag128(L.views, 18), xr, 0, 0, lane, am1);
            load_bias<1>(bias1, inv1, wb, (L.views.b + 32 * wave) * 4, (L.views.b + kHalf) * 4, lane);
            prefetch_w<2>(pre2, wb, frag256(L.trunk[0], 4));
            f16x8 hi[2][2], lo[2][2];
            const SaveDst sv = save_dst(SAVE_VH, kHalf, 32 * wave);
            to_operands<1, kSave>(am1, inv1, bias1, amax2, hi, lo, &sv);
            regop_gemm<2>(wb, (L.resr.w + wave * 2 * 2 * 256) * 4, hi, lo, part_res);
        }
        __syncthreads();                           // feature / dir columns are dead: exchange area may be written
        if (lane < 32) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                float* ex = reinterpret_cast<float*>(ldsd + (lane + 32 * pb) * kRowD + kColExD) + 8 * wave;
                *reinterpret_cast<f32x4*>(ex) = part_as[pb];
                *reinterpret_cast<f32x4*>(ex + 4) = part_res[pb];
            }
        }
        __syncthreads();
        if (lane < 16 && my_valid) {
            const float* ex = reinterpret_cast<const float*>(ldsd + (16 * wave + lane) * kRowD + kColExD);
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
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a0, sh), r0), out_row + 0);          // run_nerf_helpers.py:320
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a1, sh), r1), out_row + 1);
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a2, sh), r2), out_row + 2);
            __builtin_nontemporal_store(sig4[0], out_row + 3);
            __builtin_nontemporal_store(a0, out_row + 4); __builtin_nontemporal_store(a1, out_row + 5); __builtin_nontemporal_store(a2, out_row + 6);
            __builtin_nontemporal_store(sh, out_row + 7);
            __builtin_nontemporal_store(r0, out_row + 8); __builtin_nontemporal_store(r1, out_row + 9); __builtin_nontemporal_store(r2, out_row + 10);
        }
    }
    const float amax_all = fmaxf(amax, fmaxf((float)amax2[0], (float)amax2[1]));
    if (p.status && __any(!(amax_all <= kF16Safe)) && lane == 0) atomicOr(p.status, INERF_STATUS_F16_RANGE);
    if (kSave && p.act_max) {
        float m = amax_all * (1.0f / kActScale);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0 && m == m) atomicMax(reinterpret_cast<unsigned int*>(p.act_max), __builtin_bit_cast(unsigned int, m));
    }
}

// ================================================================================================
// One workgroup of EIGHT waves per CU, 128-point tiles ("quad": four 32-point blocks per wave).
//
// Both kernels above fetch every weight fragment once per 64 sample points: the whole 2.7 MB blob streams from L2
// through the CU's vector-memory path (64 B/clk) for each tile - 42 KB per point, two thirds of that path's bandwidth
// at full matrix rate.  Here a fetched fragment serves 128 points: wave w (of 8) owns 32 output channels x ALL 128 points
// of the tile (1 row block x 4 point blocks, still 64 accumulator registers), so the A stream per MFMA halves; the B
// operands (activations, from LDS) double instead - 85 B/clk of the LDS's 256.  Two waves per SIMD as in the
// two-workgroup kernel, but in ONE workgroup: 2 x 75,776 B of LDS hold the 128 rows, layers still update in place
// (GEMM | barrier | store | barrier).  What is lost is the second workgroup's independent phase: both waves of a SIMD
// convert accumulators at the same time.  Same packed blob: wave w's fragments are row block (w & 1) of the 4-wave
// layout's wave (w >> 1), i.e. the same bytes at a different stride.
// The view-dependent layer has 128 outputs = four 32-channel groups: waves (2g, 2g+1) share group g's weights and take
// the tile's first / second 64 points.
// Inference only (the training forward keeps using the two-workgroup kernel); SSR semantic head per wave as there.
// ================================================================================================
constexpr int kPtsQ = 128;
constexpr int kPlaneQ = kPtsQ * kRowD;
constexpr int kLdsBytesQ = 2 * kPlaneQ * 2;      // 151,552
constexpr int kColExQ = 64;                       // exchange area (floats) in columns 64.. of the hi plane: [8 waves][4] as | [4 groups][4] res

// A fragments of this wave: k-blocks are `kb_stride` bytes apart (2 KiB hi+lo inside)
template <int PB, int KBT, bool ZERO = true>
__device__ __forceinline__ void gemm_q(const f16x8 (&pre)[2][2], const WeightBuf& wb, int frag_bytes, int kb_stride,
                                       const _Float16* xl /* plane_hi + (lane&31)*kRowD + 8*(lane>>5) + column */,
                                       f32x16 (&am)[PB]) {
    static_assert(KBT % 2 == 0 && KBT >= 2, "k-block count");
    if constexpr (ZERO) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) am[pb][r] = 0.0f;
    }
    f16x8 w[4][2], x[2][PB][2];
#pragma unroll
    for (int part = 0; part < 2; ++part) { w[0][part] = pre[0][part]; w[1][part] = pre[1][part]; }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int part = 0; part < 2; ++part) x[0][pb][part] = *reinterpret_cast<const f16x8*>(xl + part * kPlaneQ + pb * 32 * kRowD);

#define INERF_Q_STEP(K, I)                                                                                            \
    {                                                                                                                 \
        const int k1_ = (K) + 1 < KBT ? (K) + 1 : KBT - 1;                                                            \
        const int k2_ = (K) + 2 < KBT ? (K) + 2 : KBT - 1;                                                            \
        _Pragma("unroll") for (int part = 0; part < 2; ++part)                                                        \
            w[((I) + 2) & 3][part] = wb.frag(frag_bytes + k2_ * kb_stride + part * 1024);                             \
        _Pragma("unroll") for (int pb = 0; pb < PB; ++pb)                                                             \
            _Pragma("unroll") for (int part = 0; part < 2; ++part)                                                    \
                x[((I) + 1) & 1][pb][part] =                                                                          \
                    *reinterpret_cast<const f16x8*>(xl + part * kPlaneQ + 16 * k1_ + pb * 32 * kRowD);                \
        _Pragma("unroll") for (int pb = 0; pb < PB; ++pb)                                                             \
            am[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(I) & 3][0], x[(I) & 1][pb][0], am[pb], 0, 0, 0);       \
        _Pragma("unroll") for (int pb = 0; pb < PB; ++pb)                                                             \
            am[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(I) & 3][0], x[(I) & 1][pb][1], am[pb], 0, 0, 0);       \
        _Pragma("unroll") for (int pb = 0; pb < PB; ++pb)                                                             \
            am[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(I) & 3][1], x[(I) & 1][pb][0], am[pb], 0, 0, 0);       \
        /* issue order per third of the step: MFMA, (weight load), MFMA, LDS reads, MFMA ... (masks: 0x8 MFMA, 0x20 VMEM read, 0x100 DS read) */ \
        _Pragma("unroll") for (int q = 0; q < PB; ++q) {                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                        \
            if (q < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                        \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                        \
        }                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
    constexpr int KB4 = KBT & ~3;
#pragma unroll 1
    for (int kb = 0; kb < KB4; kb += 4) {
        INERF_Q_STEP(kb + 0, 0)
        INERF_Q_STEP(kb + 1, 1)
        INERF_Q_STEP(kb + 2, 2)
        INERF_Q_STEP(kb + 3, 3)
    }
    if constexpr (KBT - KB4 == 2) {
        INERF_Q_STEP(KB4 + 0, 0)
        INERF_Q_STEP(KB4 + 1, 1)
    }
#undef INERF_Q_STEP
}

__device__ __forceinline__ void prefetch_q(f16x8 (&pre)[2][2], const WeightBuf& wb, int frag_bytes, int kb_stride) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int part = 0; part < 2; ++part) pre[kb][part] = wb.frag(frag_bytes + kb * kb_stride + part * 1024);
}

// epilogue of a 32-channel x (32*PB)-point block: t = acc * inv + bias' (ReLU) -> hi/lo planes (truncation split, see
// wide_store_h); dl = plane_hi + (lane&31)*kRowD + 4*(lane>>5) + first channel
template <int PB>
__device__ __forceinline__ void store_q(const f32x16 (&am)[PB], float inv, const f32x4 (&bias)[4], _Float16* dl, bool relu, f16x2& amax2) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                t[i] = __builtin_fmaf(am[pb][4 * g + i], inv, bias[g][i]);
                if (relu) t[i] = fmaxf(t[i], 0.0f);
            }
            f16x2 h01, h23, l01, l23;
            split_pair(t[0], t[1], h01, l01);
            split_pair(t[2], t[3], h23, l23);
            f16x2 a01 = h01, a23 = h23;
            if (!relu) {
                a01 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h01) & 0x7FFF7FFFu);
                a23 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, h23) & 0x7FFF7FFFu);
            }
            amax2 = __builtin_elementwise_max(amax2, __builtin_elementwise_max(a01, a23));
            const f16x4 hi4 = {h01[0], h01[1], h23[0], h23[1]}, lo4 = {l01[0], l01[1], l23[0], l23[1]};
            _Float16* d = dl + pb * 32 * kRowD + 8 * g;
            *reinterpret_cast<f16x4*>(d) = hi4;
            *reinterpret_cast<f16x4*>(d + kPlaneQ) = lo4;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// accumulators of this wave's 32 channels x (32*PB) points -> (bias, ReLU, hi/lo split) -> B operands of the next MFMA,
// one per 16-channel k-block q2 (accumulator registers 8*q2 .. +7) and point block
template <int PB>
__device__ __forceinline__ void to_operands_q(const f32x16 (&am)[PB], float inv, const f32x4 (&bias)[4], f16x2& amax2,
                                              f16x8 (&hi)[2][PB], f16x8 (&lo)[2][PB]) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            f16x8 fh, fl;
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const int j = 8 * q2 + i;
                const float t0 = fmaxf(__builtin_fmaf(am[pb][j], inv, bias[j >> 2][j & 3]), 0.0f);
                const float t1 = fmaxf(__builtin_fmaf(am[pb][j + 1], inv, bias[(j + 1) >> 2][(j + 1) & 3]), 0.0f);
                f16x2 h2, l2;
                split_pair(t0, t1, h2, l2);
                fh[i] = h2[0]; fh[i + 1] = h2[1];
                fl[i] = l2[0]; fl[i + 1] = l2[1];
            }
            const f16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(f16x2{fh[0], fh[1]}, f16x2{fh[2], fh[3]}),
                                                      __builtin_elementwise_max(f16x2{fh[4], fh[5]}, f16x2{fh[6], fh[7]}));
            amax2 = __builtin_elementwise_max(amax2, m);
            hi[q2][pb] = fh;
            lo[q2][pb] = fl;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// output head from register operands: rows 0..3 of the 32-row result, summed over this wave's two k-blocks (partial sums)
template <int PB>
__device__ __forceinline__ void regop_gemm_q(const WeightBuf& wb, int frag_bytes, const f16x8 (&hi)[2][PB], const f16x8 (&lo)[2][PB],
                                             f32x4 (&part)[PB]) {
    f32x16 acc[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pb][r] = 0.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f16x8 wh = wb.frag(frag_bytes + (2 * q) * 1024), wl = wb.frag(frag_bytes + (2 * q + 1) * 1024);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hi[q][pb], acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, lo[q][pb], acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hi[q][pb], acc[pb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) part[pb] = f32x4{acc[pb][0], acc[pb][1], acc[pb][2], acc[pb][3]};
}

__device__ __forceinline__ void load_bias_q(f32x4 (&bias)[4], float& inv, const WeightBuf& wb, int bias_bytes, int scale_bytes, int lane) {
    const int h16 = 16 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = wb.vec4(bias_bytes + 8 * g * 4, h16);
    inv = wb.scalar(scale_bytes);
}

// The quad layout worked as two 64-point halves with a software-pipelined epilogue (kernel form "halves"): one phase is the
// GEMM of ONE half - this wave's 32 channels x 64 points, am[2H], am[2H+1], weights streamed two k-blocks ahead exactly as in
// gemm_q - while, one chunk every second step, the OTHER half's accumulators (the previous phase's result) are retired:
// bias, ReLU, hi/lo split, two ds_write_b64 into that half's rows.  A half's rows are rewritten only a barrier after their
// last reader, so a 256-wide layer costs two barriers as before, but no wave ever converts accumulators without issuing
// MFMAs - and the second wave of its SIMD fills the gaps its VALU instructions leave (measured on the one-wave-per-SIMD
// form of the same idea, mlp_f16_pipe.hip: there every filler instruction delays the wave's own next MFMA by ~3 cycles).
// x_hi / x_lo, d_hi / d_lo: opaque element offsets (the planes exceed the DS instructions' 16-bit offset field).
template <int H, int KBT, bool EPI, bool ZERO = true>
__device__ __forceinline__ void gemm_half_epi(const f16x8 (&pre)[4][2], const WeightBuf& wb, int frag_bytes, int kb_stride, _Float16* lds,
                                              int x_hi, int x_lo, f32x16 (&am)[4], int d_hi, int d_lo, const f32x4 (&ebias)[4], float einv,
                                              f16x2& amax2) {
    static_assert(KBT == 16, "pipelined phases are the 256-wide layers");
    constexpr int O = 2 * (1 - H);                 // the half being retired
    if constexpr (ZERO) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) am[2 * H + pb][r] = 0.0f;
    }
    // a step is only 6 MFMAs (~200 cycles) here, so operands are requested further ahead than in gemm_q: weights FOUR k-blocks
    // (ring of 6; the first four arrive in `pre`, requested by the previous phase), activations two (ring of 3)
    f16x8 w[6][2], x[3][2][2];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int part = 0; part < 2; ++part) w[k][part] = pre[k][part];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int part = 0; part < 2; ++part) x[k][pb][part] = *reinterpret_cast<const f16x8*>(lds + (part ? x_lo : x_hi) + 16 * k + pb * 32 * kRowD);
#pragma unroll
    for (int s = 0; s < KBT; ++s) {
        const int s2 = s + 2 < KBT ? s + 2 : KBT - 1, s4 = s + 4 < KBT ? s + 4 : KBT - 1;
#pragma unroll
        for (int part = 0; part < 2; ++part) w[(s + 4) % 6][part] = wb.frag(frag_bytes + s4 * kb_stride + part * 1024);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                x[(s + 2) % 3][pb][part] = *reinterpret_cast<const f16x8*>(lds + (part ? x_lo : x_hi) + 16 * s2 + pb * 32 * kRowD);
#pragma unroll
        for (int combo = 0; combo < 3; ++combo)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
                am[2 * H + pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s % 6][combo == 2 ? 1 : 0], x[s % 3][pb][combo == 1 ? 1 : 0],
                                                                          am[2 * H + pb], 0, 0, 0);
        if constexpr (EPI) {
            if ((s & 1) == 0) {
                const int c = s >> 1, pb = c >> 2, g = c & 3;              // 8 chunks: (point block, 8-channel group)
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = fmaxf(__builtin_fmaf(am[O + pb][4 * g + i], einv, ebias[g][i]), 0.0f);
                f16x2 h01, h23, l01, l23;
                split_pair(t[0], t[1], h01, l01);
                split_pair(t[2], t[3], h23, l23);
                amax2 = __builtin_elementwise_max(amax2, __builtin_elementwise_max(h01, h23));
                const int doff = pb * 32 * kRowD + 8 * g;
                *reinterpret_cast<f16x4*>(lds + d_hi + doff) = f16x4{h01[0], h01[1], h23[0], h23[1]};
                *reinterpret_cast<f16x4*>(lds + d_lo + doff) = f16x4{l01[0], l01[1], l23[0], l23[1]};
            }
        }
        // issue order: MFMA, weight load, MFMA, LDS reads, (epilogue VALU), MFMA ...
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if (EPI) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (EPI) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            if (EPI && q == 1) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ int opaque_off(int v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ void prefetch_q4(f16x8 (&pre)[4][2], const WeightBuf& wb, int frag_bytes, int kb_stride) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int part = 0; part < 2; ++part) pre[kb][part] = wb.frag(frag_bytes + kb * kb_stride + part * 1024);
}

template <bool kSsr, bool kHalves>
__global__ __launch_bounds__(512, 2) void k_encode_mlp_f16x3_quad(const MlpParams p) {
    constexpr int kPts = kPtsQ;
    constexpr int kParts = 512 / kPts;          // 4
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsq[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..7
    const NetLayout& L = p.L;
    float amax = 0.0f;
    f16x2 amax2 = {(_Float16)0.0f, (_Float16)0.0f};

    _Float16* const xw = ldsq + (lane & 31) * kRowD;
    const _Float16* const xr = xw + 8 * (lane >> 5);                       // wide GEMM operand reads (+ column)
    _Float16* const xd = xw + 4 * (lane >> 5) + 32 * wave;                 // wide stores: this wave's 32 channels
    const _Float16* const xs = ldsq + (16 * wave + (lane & 15)) * kRowD + 8 * (lane >> 4);   // skinny operand reads: this wave's 16 points
    // halves form: one opaque element offset per (half, plane) for operand reads and for this wave's stores
    const int xbase = (lane & 31) * kRowD + 8 * (lane >> 5), dbase = (lane & 31) * kRowD + 4 * (lane >> 5) + 32 * wave;
    const int xoff[2][2] = {{kHalves ? opaque_off(xbase) : 0, kHalves ? opaque_off(xbase + kPlaneQ) : 0},
                            {kHalves ? opaque_off(xbase + 64 * kRowD) : 0, kHalves ? opaque_off(xbase + 64 * kRowD + kPlaneQ) : 0}};
    const int doff[2][2] = {{kHalves ? opaque_off(dbase) : 0, kHalves ? opaque_off(dbase + kPlaneQ) : 0},
                            {kHalves ? opaque_off(dbase + 64 * kRowD) : 0, kHalves ? opaque_off(dbase + 64 * kRowD + kPlaneQ) : 0}};

    WeightBuf wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wts), 0, L.total_floats * 4, 0x00020000);
    wb.voff = lane * 16;
    // 256-channel layers in the 4-wave blob: wave4 = wave >> 1 owns KBT*2 KiB-pairs [kb][rb][hi|lo]; this wave is rb = wave & 1
    auto frag256 = [&](const GemmSlot& s, int kbt) { return (s.w + (wave >> 1) * kbt * 2 * 2 * 256) * 4 + (wave & 1) * 2048; };
    constexpr int kStride256 = 4096;             // bytes between this wave's consecutive k-blocks
    // 128-channel view layer: channel group g = wave >> 1 (the 4-wave blob's wave g), k-blocks 2 KiB apart
    auto frag128 = [&](const GemmSlot& s, int kbt) { return (s.w + (wave >> 1) * kbt * 1 * 2 * 256) * 4; };
    constexpr int kStride128 = 2048;

    f16x8 pre[2][2];
    prefetch_q(pre, wb, frag256(L.trunk[0], 4), kStride256);

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        // ---------------- encode -> hi/lo planes (xyz: columns 0..63, dir: columns 256..287) ----------------
        auto encode = [&](bool with_dir) {
            const int pt = tid % kPts, part = tid / kPts;
            int gp = tile * kPts + pt;
            gp = gp < p.n_points ? gp : p.n_points - 1;
            const int ray = gp / p.n_samples;
            const float* __restrict__ r = p.rays + (size_t)ray * INERF_RAY_FLOATS;
            const float zz = __builtin_nontemporal_load(p.z + gp);
            _Float16* row = ldsq + pt * kRowD;
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
                    split_store<kPlaneQ>(row + 3 + 6 * f + c, sn, amax);
                    split_store<kPlaneQ>(row + 6 + 6 * f + c, cs, amax);
                }
            }
            if (part == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) split_store<kPlaneQ>(row + c, x[c], amax);
                for (int c = 3 + 6 * p.l_xyz; c < kEncCols; ++c) { row[c] = (_Float16)0.0f; row[kPlaneQ + c] = (_Float16)0.0f; }
            }
            if (with_dir) {
                const int fd = kParts - 1 - part;
                if (fd < p.l_dir) {
                    const float s = (float)(1 << fd);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float sn, cs;
                        fast_sincosf(r[8 + c] * s, &sn, &cs);
                        split_store<kPlaneQ>(row + kColDirD + 3 + 6 * fd + c, sn, amax);
                        split_store<kPlaneQ>(row + kColDirD + 6 + 6 * fd + c, cs, amax);
                    }
                }
                if (part == 3) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) split_store<kPlaneQ>(row + kColDirD + c, r[8 + c], amax);
                    for (int c = 3 + 6 * p.l_dir; c < kDirCols; ++c) { row[kColDirD + c] = (_Float16)0.0f; row[kPlaneQ + kColDirD + c] = (_Float16)0.0f; }
                }
            }
        };
        encode(true);
        __syncthreads();

        f32x16 am[4];
        f32x4 bias[4];
        float inv;
        // a 256-wide layer in place: GEMM | barrier (every wave has read the input) | store | barrier
        auto store256 = [&](const GemmSlot& s, bool relu, auto&& prefetch_next) {
            load_bias_q(bias, inv, wb, (s.b + 32 * wave) * 4, (s.b + kWidth) * 4, lane);
            prefetch_next();
            __syncthreads();
            store_q<4>(am, inv, bias, xd, relu, amax2);
            __syncthreads();
        };
        auto pf256 = [&](const GemmSlot& s, int kbt, int kb_first = 0) {
            return [&, kbt, kb_first]() { prefetch_q(pre, wb, frag256(s, kbt) + kb_first * kStride256, kStride256); };
        };

        // ---------------- trunk ----------------
        gemm_q<4, 4>(pre, wb, frag256(L.trunk[0], 4), kStride256, xr, am);
        if constexpr (!kHalves) store256(L.trunk[0], true, pf256(L.trunk[1], 16));
        if constexpr (!kHalves) {
#pragma unroll 1
            for (int layer = 1; layer < kSkipInput; ++layer) {
                gemm_q<4, 16>(pre, wb, frag256(L.trunk[layer], 16), kStride256, xr, am);
                if (layer + 1 < kSkipInput) store256(L.trunk[layer], true, pf256(L.trunk[layer + 1], 16));
                else                        store256(L.trunk[layer], true, pf256(L.trunk[kSkipInput], 20, 4));
            }
            {   // pts_linears[5] over cat([pts, h]): h-part (k-blocks 4..19 of the stream), then the encoding again
                const GemmSlot& s = L.trunk[kSkipInput];
                gemm_q<4, 16>(pre, wb, frag256(s, 20) + 4 * kStride256, kStride256, xr, am);
                prefetch_q(pre, wb, frag256(s, 20), kStride256);
                __syncthreads();
                encode(false);
                __syncthreads();
                gemm_q<4, 4, false>(pre, wb, frag256(s, 20), kStride256, xr, am);
                store256(s, true, pf256(L.trunk[6], 16));
            }
            gemm_q<4, 16>(pre, wb, frag256(L.trunk[6], 16), kStride256, xr, am);
            store256(L.trunk[6], true, pf256(L.trunk[7], 16));
            gemm_q<4, 16>(pre, wb, frag256(L.trunk[7], 16), kStride256, xr, am);
            store256(L.trunk[7], true, pf256(L.as1, 16));
        } else {
            // Two 64-point halves, epilogues pipelined (gemm_half_epi).  am[0..1] = half A, am[2..3] = half B.  Phase (L, A) retires
            // layer L-1 of half B, phase (L, B) layer L of half A; one barrier per phase.  Bias / output factor of the layer being
            // retired are requested a phase before their first use (two register sets, alternating per layer).
            f32x4 bset[2][4];
            float iset[2];
            f16x8 pre4[4][2];
            auto want_bias = [&](int set, const GemmSlot& s) { load_bias_q(bset[set], iset[set], wb, (s.b + 32 * wave) * 4, (s.b + kWidth) * 4, lane); };
            auto phase = [&](auto hc, auto epic, int frag, int eset, int next_frag) {
                constexpr int Hh = decltype(hc)::value;
                constexpr bool kEpi = decltype(epic)::value;
                gemm_half_epi<Hh, 16, kEpi>(pre4, wb, frag, kStride256, ldsq, xoff[Hh][0], xoff[Hh][1], am,
                                            doff[1 - Hh][0], doff[1 - Hh][1], bset[eset], iset[eset], amax2);
                prefetch_q4(pre4, wb, next_frag, kStride256);
                __syncthreads();
            };
            using std::integral_constant;
            constexpr integral_constant<int, 0> HA{};
            constexpr integral_constant<int, 1> HB{};
            constexpr integral_constant<bool, true> EPI{};
            // layer 0: half A stored now (exposed), half B left pending with its bias in set 1 - the pipeline's entry state
            want_bias(1, L.trunk[0]);
            prefetch_q4(pre4, wb, frag256(L.trunk[1], 16), kStride256);
            __syncthreads();                                                              // every wave has read the encodings
            store_q<2>(reinterpret_cast<const f32x16(&)[2]>(am[0]), iset[1], bset[1], xd, true, amax2);
            __syncthreads();
            // a pair of layers (l, l + 1): on entry half B of layer l - 1 is pending with its bias in set 1 and pre4 holds W(l)[0..3];
            // on exit half B of layer l + 1 is pending with its bias in set 1 and pre4 holds the first fragments at next_frag
            auto layer_pair = [&](int l, int next_frag) {
                const GemmSlot& s0 = L.trunk[l];
                const GemmSlot& s1 = L.trunk[l + 1];
                const int f0 = frag256(s0, 16), f1 = frag256(s1, 16);
                want_bias(0, s0);                                                         // first used in phase (l, B)
                phase(HA, EPI, f0, 1, f0);                                                // retires (l - 1, B) with set 1
                phase(HB, EPI, f0, 0, f1);                                                // retires (l, A)
                phase(HA, EPI, f1, 0, f1);                                                // retires (l, B); set 1 is free from here on ...
                want_bias(1, s1);                                                         // ... but requested only now: first used in the next phase
                phase(HB, EPI, f1, 1, next_frag);                                         // retires (l + 1, A)
            };
#pragma unroll 1
            for (int l = 1; l < kSkipInput; l += 2)
                layer_pair(l, l + 2 < kSkipInput ? frag256(L.trunk[l + 2], 16) : frag256(L.trunk[kSkipInput], 20) + 4 * kStride256);
            {   // pts_linears[5] over cat([pts, h]): h-parts of both halves, the encoding again, then its part for all 128 rows at once
                const GemmSlot& s = L.trunk[kSkipInput];
                const int fh = frag256(s, 20) + 4 * kStride256;
                want_bias(0, s);
                phase(HA, EPI, fh, 1, fh);                                                // retires (4, B)
                gemm_half_epi<1, 16, false>(pre4, wb, fh, kStride256, ldsq, xoff[1][0], xoff[1][1], am, 0, 0, bset[0], iset[0], amax2);
                prefetch_q(pre, wb, frag256(s, 20), kStride256);
                __syncthreads();                                                          // every wave has read h4
                encode(false);
                __syncthreads();
                gemm_q<4, 4, false>(pre, wb, frag256(s, 20), kStride256, xr, am);
                prefetch_q4(pre4, wb, frag256(L.trunk[6], 16), kStride256);
                __syncthreads();                                                          // every wave has read the encodings
                store_q<2>(reinterpret_cast<const f32x16(&)[2]>(am[0]), iset[0], bset[0], xd, true, amax2);   // half A, exposed
                __syncthreads();
                // set 1 must hold layer 5's bias for the pending half B: copy (a few moves, once per tile)
#pragma unroll
                for (int g = 0; g < 4; ++g) bset[1][g] = bset[0][g];
                iset[1] = iset[0];
            }
            layer_pair(6, frag256(L.as1, 16));                                            // leaves W(as1)[0..3] in pre4 ...
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int part = 0; part < 2; ++part) pre[k][part] = pre4[k][part];        // ... of which the heads' gemm_q wants two
            // drain: half B of layer 7
            store_q<2>(reinterpret_cast<const f32x16(&)[2]>(am[2]), iset[1], bset[1], xd + 64 * kRowD, true, amax2);
            __syncthreads();
        }

        // ---------------- heads ----------------
        const int my_pt = tile * kPts + 16 * wave + (lane & 15);
        const bool my_valid = my_pt < p.n_points;
        float* const out_row = p.raw + (size_t)(my_valid ? my_pt : 0) * p.channels;
        const f32x4 sig4 = skinny_gemm_h<8, kPlaneQ>(wb, L.alpha.w * 4, L.alpha.b * 4, (L.alpha.b + 16) * 4, xs, lane);

        // albedo + shading: hidden layer (this wave: 32 of its 256 channels) -> registers -> partial output sums
        f32x4 part_as[4], part_res[2];
        {
            gemm_q<4, 16>(pre, wb, frag256(L.as1, 16), kStride256, xr, am);
            load_bias_q(bias, inv, wb, (L.as1.b + 32 * wave) * 4, (L.as1.b + kWidth) * 4, lane);
            prefetch_q(pre, wb, frag256(L.feat, 16), kStride256);
            // as2r: [wave4][q 0..3][hi|lo] KiB; this wave's two k-blocks are q = 2*(wave&1), +1 of wave4 = wave>>1.
            // Two point blocks at a time: converting all four first would hold 64 operand + 64 accumulator registers
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f16x8 hi[2][2], lo[2][2];
                to_operands_q<2>(reinterpret_cast<const f32x16(&)[2]>(am[2 * hf]), inv, bias, amax2, hi, lo);
                regop_gemm_q<2>(wb, (L.as2r.w + (wave >> 1) * 4 * 2 * 256) * 4 + (wave & 1) * 4096, hi, lo,
                                reinterpret_cast<f32x4(&)[2]>(part_as[2 * hf]));
            }
        }
        if (kSsr && L.sem_rbs > 0) {               // semantic logits straight to raw[11 .. 11+C) (semantic_nerf.py:150-152)
            sem_head<false, kPlaneQ>(wb, L, xs, lane, amax2, out_row, my_valid, p.n_classes, nullptr);
        }
        // feature (no activation) in place of h7, then the view-dependent layer over [feature | dir] -> registers
        gemm_q<4, 16>(pre, wb, frag256(L.feat, 16), kStride256, xr, am);
        store256(L.feat, false, [&]() { prefetch_q(pre, wb, frag128(L.views, 18), kStride128); });
        {
            const int half = wave & 1;                                  // this wave's 64 points of the tile
            f32x16 am2[2];
            gemm_q<2, 18>(pre, wb, frag128(L.views, 18), kStride128, xr + half * 64 * kRowD, am2);
            load_bias_q(bias, inv, wb, (L.views.b + 32 * (wave >> 1)) * 4, (L.views.b + kHalf) * 4, lane);
            prefetch_q(pre, wb, frag256(L.trunk[0], 4), kStride256);
            f16x8 hi[2][2], lo[2][2];
            to_operands_q<2>(am2, inv, bias, amax2, hi, lo);
            regop_gemm_q<2>(wb, (L.resr.w + (wave >> 1) * 2 * 2 * 256) * 4, hi, lo, part_res);
        }
        __syncthreads();                           // feature / dir columns are dead: exchange area may be written
        if (lane < 32) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                float* ex = reinterpret_cast<float*>(ldsq + (lane + 32 * pb) * kRowD + kColExQ) + 4 * wave;
                *reinterpret_cast<f32x4*>(ex) = part_as[pb];
            }
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                float* ex = reinterpret_cast<float*>(ldsq + (lane + 32 * pb + 64 * (wave & 1)) * kRowD + kColExQ) + 32 + 4 * (wave >> 1);
                *reinterpret_cast<f32x4*>(ex) = part_res[pb];
            }
        }
        __syncthreads();
        if (lane < 16 && my_valid) {
            const float* ex = reinterpret_cast<const float*>(ldsq + (16 * wave + lane) * kRowD + kColExQ);
            f32x4 as4 = {0.0f, 0.0f, 0.0f, 0.0f}, res4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int w = 0; w < 8; ++w) as4 += *reinterpret_cast<const f32x4*>(ex + 4 * w);
#pragma unroll
            for (int g = 0; g < 4; ++g) res4 += *reinterpret_cast<const f32x4*>(ex + 32 + 4 * g);
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
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a0, sh), r0), out_row + 0);          // run_nerf_helpers.py:320
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a1, sh), r1), out_row + 1);
            __builtin_nontemporal_store(__fadd_rn(__fmul_rn(a2, sh), r2), out_row + 2);
            __builtin_nontemporal_store(sig4[0], out_row + 3);
            __builtin_nontemporal_store(a0, out_row + 4); __builtin_nontemporal_store(a1, out_row + 5); __builtin_nontemporal_store(a2, out_row + 6);
            __builtin_nontemporal_store(sh, out_row + 7);
            __builtin_nontemporal_store(r0, out_row + 8); __builtin_nontemporal_store(r1, out_row + 9); __builtin_nontemporal_store(r2, out_row + 10);
        }
        // no barrier: the next tile's encode writes columns 0..63 and 256..287 only, the exchange area (columns 64..159) is
        // first overwritten by layer 0's store, two barriers later
    }
    const float amax_all = fmaxf(amax, fmaxf((float)amax2[0], (float)amax2[1]));
    if (p.status && __any(!(amax_all <= kF16Safe)) && lane == 0) atomicOr(p.status, INERF_STATUS_F16_RANGE);
}

static int launch_quad(MlpParams& p, int64_t n_points, bool ssr, bool halves, hipStream_t stream) {
    p.n_tiles = (int)((n_points + kPtsQ - 1) / kPtsQ);
    const int grid = p.n_tiles < device_cus() ? p.n_tiles : device_cus();
    void (*kern)(const MlpParams) = halves ? (ssr ? k_encode_mlp_f16x3_quad<true, true> : k_encode_mlp_f16x3_quad<false, true>)
                                           : (ssr ? k_encode_mlp_f16x3_quad<true, false> : k_encode_mlp_f16x3_quad<false, false>);
    static PerDeviceOnce attr_set4[4];
    PerDeviceOnce* attr_set = attr_set4 + 2 * (int)halves;
    if (attr_set[ssr].first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesQ);
        if (e != hipSuccess) return record(e);
        attr_set[ssr].mark();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kLdsBytesQ, stream, p);
    return record(hipGetLastError());
}

static int launch_dual(MlpParams& p, int64_t n_points, bool ssr, hipStream_t stream) {
    p.n_tiles = (int)((n_points + kTilePoints - 1) / kTilePoints);
    const int max_grid = 2 * device_cus();
    const int grid = p.n_tiles < max_grid ? p.n_tiles : max_grid;
    const bool save = p.save != nullptr;
    void (*kern)(const MlpParams) = ssr ? (save ? k_encode_mlp_f16x3_dual<true, true> : k_encode_mlp_f16x3_dual<false, true>)
                                        : (save ? k_encode_mlp_f16x3_dual<true, false> : k_encode_mlp_f16x3_dual<false, false>);
    static PerDeviceOnce attr_set[4];
    const int variant = 2 * (int)ssr + (int)save;
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
    if (form && form[0] == 'q' && !p.save && !(ssr && p.endpoint)) return launch_quad(p, n_points, ssr, false, stream);   // 128-point tiles, 8 waves
    if (form && form[0] == 'h' && !p.save && !(ssr && p.endpoint)) return launch_quad(p, n_points, ssr, true, stream);    // ... as two pipelined halves
    if (form && form[0] == 'p' && !p.save && !(ssr && p.endpoint)) return launch_pipe(p, n_points, ssr, stream);   // resident weights, pipelined epilogue
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
