// mlp_f16_heads.h - pieces of the in-place (296-half rows) two-workgroup split-f16 kernel of mlp_f16.hip: row geometry,
// accumulator -> register-operand conversion, the register-operand output heads and the per-wave semantic head.
#pragma once
#include "mlp_f16_dev.h"

namespace inerf {

constexpr int kRowD = kWidth + kDirCols + 8;      // 296
constexpr int kPlaneD = kTilePoints * kRowD;
constexpr int kLdsBytesD = 2 * kPlaneD * 2;       // 75,776
constexpr int kColDirD = kWidth;                  // direction encoding right behind h: views reads [feature | dir] in one sweep
constexpr int kColExD = 64;                       // exchange area: floats [wave][8] per point in columns 64..127 of the hi plane

// accumulators of this wave's 32*RB channels x 64 points -> (bias, ReLU, hi/lo split) -> B operands, one per
// 16-channel k-block q (accumulator registers 8*(q&1) .. +7 of row block q>>1) and point block
template <int RB, bool SAVE = false, int PB = 2, bool MAX3 = true>
__device__ __forceinline__ void to_operands(const f32x16 (&am)[RB][PB], float inv, const f32x4 (&bias)[RB][4], f16x2& amax2,
                                            f16x8 (&hi)[2 * RB][PB], f16x8 (&lo)[2 * RB][PB], const SaveDst* sv = nullptr) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                f16x8 fh, fl;
                float tv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int j = 8 * q2 + i;
                    tv[i] = fmaxf(__builtin_fmaf(am[rb][pb][j], inv, bias[rb][j >> 2][j & 3]), 0.0f);
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    f16x2 h2, l2;
                    split_pair(tv[i], tv[i + 1], h2, l2);
                    fh[i] = h2[0]; fh[i + 1] = h2[1];
                    fl[i] = l2[0]; fl[i + 1] = l2[1];
                }
#ifndef INERF_ABL_NO_ROWS
                if constexpr (SAVE) {          // registers 8*q2 .. +7 are channels 32*rb + 8*(2*q2) + 4h .. +3 and + 8*(2*q2 + 1) + 4h .. +3
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const f32x4 v = f32x4{tv[4 * gg], tv[4 * gg + 1], tv[4 * gg + 2], tv[4 * gg + 3]} * (1.0f / kActScale);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sv->rsrc,
                                                               sv->voff + (pb * 32 * sv->stride + 32 * rb + 8 * (2 * q2 + gg)) * 4, 0, 0);
                    }
                }
#endif
                if constexpr (MAX3) {       // (inference; in the training forward the three-input form costs 20 bytes of scratch)
                    pk_max3_into(amax2, f16x2{fh[0], fh[1]}, f16x2{fh[2], fh[3]});
                    pk_max3_into(amax2, f16x2{fh[4], fh[5]}, f16x2{fh[6], fh[7]});
                } else {
                    const f16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(f16x2{fh[0], fh[1]}, f16x2{fh[2], fh[3]}),
                                                              __builtin_elementwise_max(f16x2{fh[4], fh[5]}, f16x2{fh[6], fh[7]}));
                    amax2 = __builtin_elementwise_max(amax2, m);
                }
                {   // pinned: left free, the compiler keeps the eight partial maxima of a call alive (spilled) and folds them in much later
                    unsigned a = __builtin_bit_cast(unsigned, amax2);
                    asm volatile("" : "+v"(a));
                    amax2 = __builtin_bit_cast(f16x2, a);
                }
                hi[2 * rb + q2][pb] = fh;
                lo[2 * rb + q2][pb] = fl;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// an output head straight from register operands: rows 0..3 of the 32-row result are the head's outputs, summed over
// this wave's Q k-blocks only (partial sums; the caller adds the four waves')
template <int Q>
__device__ __forceinline__ void regop_gemm(const WeightBuf& wb, int frag_bytes, const f16x8 (&hi)[Q][2], const f16x8 (&lo)[Q][2],
                                           f32x4 (&part)[2]) {
    f32x16 acc[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pb][r] = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const f16x8 wh = wb.frag(frag_bytes + (2 * q) * 1024), wl = wb.frag(frag_bytes + (2 * q + 1) * 1024);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hi[q][pb], acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, lo[q][pb], acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hi[q][pb], acc[pb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) part[pb] = f32x4{acc[pb][0], acc[pb][1], acc[pb][2], acc[pb][3]};   // rows 0..3: lanes 0..31
}

// all 32 rows of a register-operand product (the semantic logits of one 32-class block, summed over this wave's Q k-blocks)
template <int Q, int PB = 2>
__device__ __forceinline__ void regop_gemm_full(const WeightBuf& wb, int frag_bytes, const f16x8 (&hi)[Q][PB], const f16x8 (&lo)[Q][PB],
                                                f32x16 (&acc)[PB]) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pb][r] = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const f16x8 wh = wb.frag(frag_bytes + (2 * q) * 1024), wl = wb.frag(frag_bytes + (2 * q + 1) * 1024);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hi[q][pb], acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, lo[q][pb], acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hi[q][pb], acc[pb], 0, 0, 0);
        }
    }
}

constexpr int kSemScratchBytes = 4 * 2 * 16 * 64 * 4;      // per workgroup and 32-class block: 32 KiB, 8 KiB per wave (the channel-split head parks
                                                           // [16 registers][64 lanes] floats = 4 KiB there: its partial logits of 32 classes x 32 points)

// semantic head of the two-workgroup kernel (SSR): there is no room in LDS for the 128-channel hidden layer and no
// registers to hold partial logits across the feature / view layers, so every wave does the whole head for ITS 16
// points on v_mfma_f32_16x16x32_f16: hidden = relu(sem1 . h7) from the skinny-format copy of sem1 (streamed by all four
// waves), converted in registers into the B operands of the logits GEMM (layout.h: sem2r) - no LDS, no exchange.
template <bool kSave, int PLANE = kPlaneD>
__device__ __forceinline__ void sem_head(const WeightBuf& wb, const NetLayout& L, const _Float16* xs /* h7, this wave's points */,
                                         int lane, f16x2& amax2, float* out_row, bool valid, int n_classes, const SaveDst* sv) {
    constexpr int kRb = kHalf / 16, kKb = kWidth / 32;
    f32x4 acc[kRb];
#pragma unroll
    for (int rb = 0; rb < kRb; ++rb) acc[rb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // One k-block's sixteen weight fragments are requested together, a k-block ahead of their MFMAs (two register sets): fetched
    // one at a time where they are used, every MFMA waited for its own L2 round trip (the SSR training forward: + 0.4 ms per step).
    auto fetch = [&](int kb, f16x8 (&w)[kRb][2]) {
#pragma unroll
        for (int rb = 0; rb < kRb; ++rb) {
            const int frag = (L.sem1s.w * 4) + ((rb * kKb + kb) * 2) * 1024;
            w[rb][0] = wb.frag(frag);
            w[rb][1] = wb.frag(frag + 1024);
        }
    };
    f16x8 wa[kRb][2], wbuf[kRb][2];
    fetch(0, wa);
#pragma unroll
    for (int kb = 0; kb < kKb; ++kb) {
        f16x8 (&cur)[kRb][2] = (kb & 1) ? wbuf : wa;
        f16x8 (&nxt)[kRb][2] = (kb & 1) ? wa : wbuf;
        if (kb + 1 < kKb) fetch(kb + 1, nxt);
        const f16x8 xh = *reinterpret_cast<const f16x8*>(xs + 32 * kb);
        const f16x8 xl = *reinterpret_cast<const f16x8*>(xs + PLANE + 32 * kb);
#pragma unroll
        for (int rb = 0; rb < kRb; ++rb) {
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[rb][0], xh, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[rb][0], xl, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[rb][1], xh, acc[rb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // bias, ReLU, hi/lo split: accumulators of row blocks 2m, 2m+1 -> the 32-deep B operand m
    const float inv = wb.scalar((L.sem1.b + kHalf) * 4);
    f16x8 bh[kRb / 2], bl[kRb / 2];
#pragma unroll
    for (int rb = 0; rb < kRb; ++rb) {
        const f32x4 bias = wb.vec4((L.sem1.b + 16 * rb) * 4, 16 * (lane >> 4));
        f32x4 tv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = fmaxf(__builtin_fmaf(acc[rb][i], inv, bias[i]), 0.0f);
            const float th = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, t) & 0xFFFFE000u);
            bh[rb >> 1][4 * (rb & 1) + i] = (_Float16)th;
            bl[rb >> 1][4 * (rb & 1) + i] = (_Float16)(t - th);
            tv[i] = t;
        }
        if constexpr (kSave)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tv * (1.0f / kActScale)), sv->rsrc, sv->voff + 16 * rb * 4, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < kRb / 2; ++m) {
        const f16x2 mx = __builtin_elementwise_max(__builtin_elementwise_max(f16x2{bh[m][0], bh[m][1]}, f16x2{bh[m][2], bh[m][3]}),
                                                   __builtin_elementwise_max(f16x2{bh[m][4], bh[m][5]}, f16x2{bh[m][6], bh[m][7]}));
        amax2 = __builtin_elementwise_max(amax2, mx);
    }
    const float inv2 = wb.scalar((L.sem2.b + 16 * L.sem_rbs) * 4);
#pragma unroll 1
    for (int rb = 0; rb < L.sem_rbs; ++rb) {
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int m = 0; m < kRb / 2; ++m) {
            const int frag = (L.sem2r.w * 4) + ((rb * (kRb / 2) + m) * 2) * 1024;
            const f16x8 wh = wb.frag(frag), wl = wb.frag(frag + 1024);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[m], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[m], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[m], a, 0, 0, 0);
        }
        const f32x4 bias = wb.vec4((L.sem2.b + 16 * rb) * 4, 16 * (lane >> 4));
        const int ch0 = 16 * rb + 4 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (valid && ch0 + i < n_classes)
                __builtin_nontemporal_store(__builtin_fmaf(a[i], inv2, bias[i]), out_row + INERF_BASE_CHANNELS + ch0 + i);
    }
}


}  // namespace inerf
