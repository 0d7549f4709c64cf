// mlp_f16_pp.h - one step of the PIPELINED trunk of the 128-point tile (mlp_f16_t128.hip, kPipe): a 32-channel x 64-point GEMM of one half of the
// tile with the epilogue of the OTHER half's previous GEMM issued between its MFMAs.
//
// Why: in the in-place layer scheme (GEMM | barrier | epilogue | barrier) the epilogue - bias, ReLU, hi/lo split, LDS stores: ~3.9 VALU per value,
// ~250 instructions per wave and layer - is 16 % of the 128-point kernel's cycles (profiles/r06_ablations.txt), and neither the SIMD's second wave
// (same workgroup: in the same phase) nor a second workgroup (VALU issued beside a partner's MFMAs still costs it issue slots) hides it.  Inside ONE
// wave an MFMA occupies the matrix pipe for 32 cycles = 8 issue slots and the wave itself has only ~1.5 of them in use (operand reads, weight
// requests): the epilogue of an INDEPENDENT set of accumulators fits between the MFMAs (~1.3 VALU per gap).  Independent work of the same wave =
// the other 64-point half of the tile, one stage behind:
//
//     step A(l): GEMM(layer l, half A) || epilogue(layer l-1, half B) -> rows of half B      barrier
//     step B(l): GEMM(layer l, half B) || epilogue(layer l,   half A) -> rows of half A      barrier
//
// One barrier per step (two per layer, as before): the epilogue of a half writes rows nobody reads in that step (every wave finished that half's GEMM
// before the previous barrier), and the GEMM of a half reads rows that were complete at the previous barrier.  Price: the weights are streamed once
// per half (the 64-point kernels' L2 -> CU stream: 3 % -> 5 % of the cycles), and the k-blocks of a step are unrolled (an epilogue slice names its
// accumulator registers statically).  Same arithmetic and summation order per output element as the other forms: bit-identical results.
#pragma once
#include "mlp_f16_dev.h"

namespace inerf {

// epilogue slice U (0..7) of a half's 32 channels x 64 points held in prev[2] (point block U >> 2, register group U & 3): four values per lane,
// in two halves of two values (H = 0: values 0, 1 -> the packed pair stays in ph_/pl_; H = 1: values 2, 3, the running maximum and the two
// 8-byte LDS stores) so that a k-block of the GEMM carries HALF a slice: ~8 VALU per 6 MFMAs - a whole slice on every other k-block made
// those k-blocks issue-bound (the kernel took as many cycles as without the pipeline).
// (fma as volatile asm: as plain arithmetic on values that are ready at the top of the step, instruction selection emits all 32 fmas of the
// eight slices in front of the first MFMA - 32 more live registers and a VALU burst nothing covers)
#define INERF_PP_HALF(U, H)                                                                                            \
    {                                                                                                                  \
        constexpr int pb_ = (U) >> 2, g_ = (U) & 3;                                                                    \
        float t0_, t1_;                                                                                                \
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t0_) : "v"(prev[pb_][4 * g_ + 2 * (H)]), "v"(inv), "v"(bias[g_][2 * (H)]));          \
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t1_) : "v"(prev[pb_][4 * g_ + 2 * (H) + 1]), "v"(inv), "v"(bias[g_][2 * (H) + 1]));  \
        t0_ = fmaxf(t0_, 0.0f);                                                                                        \
        t1_ = fmaxf(t1_, 0.0f);                                                                                        \
        if constexpr ((H) == 0) {                                                                                      \
            split_pair(t0_, t1_, ph_, pl_);                                                                            \
        } else {                                                                                                       \
            f16x2 h23_, l23_;                                                                                          \
            split_pair(t0_, t1_, h23_, l23_);                                                                          \
            pk_max3_into(amax2, ph_, h23_);                                                                            \
            const f16x4 hi4_ = {ph_[0], ph_[1], h23_[0], h23_[1]}, lo4_ = {pl_[0], pl_[1], l23_[0], l23_[1]};          \
            _Float16* d_ = dl + pb_ * 32 * ROW + 8 * g_;                                                               \
            *reinterpret_cast<f16x4*>(d_) = hi4_;                                                                      \
            *reinterpret_cast<f16x4*>(d_ + PLANE) = lo4_;                                                              \
        }                                                                                                              \
    }
#define INERF_PP_UNIT(U) INERF_PP_HALF(U, 0) INERF_PP_HALF(U, 1)

// KBT k-blocks (16 channels of the layer's input each, from column col0 of the rows `xl` points at); UNITS = 8: the eight epilogue slices of
// `prev` spread over them (one per odd k-block when KBT = 16, two per k-block when KBT = 4), 0: none.  ZERO: start `cur` from zero.
// `pre`: the first two k-blocks' weight fragments (requested by the caller in front of the barrier that precedes this step).
template <int KBT, int UNITS, bool ZERO, int ROW, int PLANE>
__device__ __forceinline__ void pp_step(const WidePreH<1>& pre, const WeightBuf& wb, int frag_bytes /* this wave's 32-channel stream, k-blocks 4 KiB apart */,
                                        const _Float16* xl /* plane_hi + (first row of the half + (lane & 31)) * ROW + 8 * (lane >> 5) */, int col0,
                                        f32x16 (&cur)[2], const f32x16 (&prev)[2], float inv, const f32x4 (&bias)[4],
                                        _Float16* dl /* plane_hi + (first row of the OTHER half + (lane & 31)) * ROW + 4 * (lane >> 5) + this wave's first channel */,
                                        f16x2& amax2) {
    static_assert((KBT == 16 || KBT == 4) && (UNITS == 0 || UNITS == 8), "step shapes of the trunk");
    if constexpr (ZERO) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) cur[pb][r] = 0.0f;
    }
    f16x8 w[4][2], x[2][2][2];              // weights: four rotating k-blocks (two ahead); activations: two (one ahead)
#pragma unroll
    for (int part = 0; part < 2; ++part) { w[0][part] = pre.w[0][0][part]; w[1][part] = pre.w[1][0][part]; }
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int part = 0; part < 2; ++part) x[0][pb][part] = *reinterpret_cast<const f16x8*>(xl + part * PLANE + col0 + pb * 32 * ROW);

// one k-block: weights of k + 2 and activations of k + 1 requested, 6 MFMAs on block k (hi*hi, hi*lo, lo*hi per point block), and NU epilogue
// slices; issue order pinned (masks: 0x2 VALU, 0x8 MFMA, 0x20 VMEM read, 0x100 DS read, 0x200 DS write)
#define INERF_PP_STEP(K, BODY, NV /* VALU instructions of BODY */, NW /* LDS stores of BODY */)                        \
    {                                                                                                                  \
        constexpr int i_ = (K) & 3;                                                                                    \
        constexpr int k1_ = (K) + 1 < KBT ? (K) + 1 : KBT - 1;                                                         \
        constexpr int k2_ = (K) + 2 < KBT ? (K) + 2 : KBT - 1;                                                         \
        _Pragma("unroll") for (int part = 0; part < 2; ++part) w[(i_ + 2) & 3][part] = wb.frag(frag_bytes + k2_ * 4096 + part * 1024); \
        _Pragma("unroll") for (int pb = 0; pb < 2; ++pb)                                                               \
            _Pragma("unroll") for (int part = 0; part < 2; ++part)                                                     \
                x[(i_ + 1) & 1][pb][part] = *reinterpret_cast<const f16x8*>(xl + part * PLANE + col0 + 16 * k1_ + pb * 32 * ROW); \
        _Pragma("unroll") for (int pb = 0; pb < 2; ++pb) cur[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i_][0], x[i_ & 1][pb][0], cur[pb], 0, 0, 0); \
        _Pragma("unroll") for (int pb = 0; pb < 2; ++pb) cur[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i_][0], x[i_ & 1][pb][1], cur[pb], 0, 0, 0); \
        _Pragma("unroll") for (int pb = 0; pb < 2; ++pb) cur[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i_][1], x[i_ & 1][pb][0], cur[pb], 0, 0, 0); \
        BODY                                                                                                           \
        constexpr int v_ = ((NV) + 4) / 5;      /* VALU per gap: BODY's spread over five of the six gaps */              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                             \
        if constexpr ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, v_, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
        if constexpr ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, v_, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                             \
        if constexpr ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, v_, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
        if constexpr ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, v_, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
        if constexpr ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, v_ + 2, 0);                                \
        if constexpr ((NW) > 0) __builtin_amdgcn_sched_group_barrier(0x200, (NW), 0);                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
#if INERF_GEMM_PRIO
    __builtin_amdgcn_s_setprio(INERF_GEMM_PRIO);
#endif
    f16x2 ph_ = {(_Float16)0.0f, (_Float16)0.0f}, pl_ = ph_;      // first pair of the slice in flight
#define INERF_PP_NONE
    if constexpr (KBT == 16 && UNITS == 8) {        // k-block K: half H = K & 1 of slice K >> 1 (7 / 9 VALU, 0 / 2 stores)
        INERF_PP_STEP(0, INERF_PP_HALF(0, 0), 7, 0)   INERF_PP_STEP(1, INERF_PP_HALF(0, 1), 9, 2)
        INERF_PP_STEP(2, INERF_PP_HALF(1, 0), 7, 0)   INERF_PP_STEP(3, INERF_PP_HALF(1, 1), 9, 2)
        INERF_PP_STEP(4, INERF_PP_HALF(2, 0), 7, 0)   INERF_PP_STEP(5, INERF_PP_HALF(2, 1), 9, 2)
        INERF_PP_STEP(6, INERF_PP_HALF(3, 0), 7, 0)   INERF_PP_STEP(7, INERF_PP_HALF(3, 1), 9, 2)
        INERF_PP_STEP(8, INERF_PP_HALF(4, 0), 7, 0)   INERF_PP_STEP(9, INERF_PP_HALF(4, 1), 9, 2)
        INERF_PP_STEP(10, INERF_PP_HALF(5, 0), 7, 0)  INERF_PP_STEP(11, INERF_PP_HALF(5, 1), 9, 2)
        INERF_PP_STEP(12, INERF_PP_HALF(6, 0), 7, 0)  INERF_PP_STEP(13, INERF_PP_HALF(6, 1), 9, 2)
        INERF_PP_STEP(14, INERF_PP_HALF(7, 0), 7, 0)  INERF_PP_STEP(15, INERF_PP_HALF(7, 1), 9, 2)
    } else if constexpr (KBT == 16) {
        INERF_PP_STEP(0, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(1, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(2, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(3, INERF_PP_NONE, 0, 0)
        INERF_PP_STEP(4, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(5, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(6, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(7, INERF_PP_NONE, 0, 0)
        INERF_PP_STEP(8, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(9, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(10, INERF_PP_NONE, 0, 0)  INERF_PP_STEP(11, INERF_PP_NONE, 0, 0)
        INERF_PP_STEP(12, INERF_PP_NONE, 0, 0)  INERF_PP_STEP(13, INERF_PP_NONE, 0, 0)  INERF_PP_STEP(14, INERF_PP_NONE, 0, 0)  INERF_PP_STEP(15, INERF_PP_NONE, 0, 0)
    } else if constexpr (UNITS == 8) {              // four k-blocks: two slices each
        INERF_PP_STEP(0, INERF_PP_UNIT(0) INERF_PP_UNIT(1), 32, 4)   INERF_PP_STEP(1, INERF_PP_UNIT(2) INERF_PP_UNIT(3), 32, 4)
        INERF_PP_STEP(2, INERF_PP_UNIT(4) INERF_PP_UNIT(5), 32, 4)   INERF_PP_STEP(3, INERF_PP_UNIT(6) INERF_PP_UNIT(7), 32, 4)
    } else {
        INERF_PP_STEP(0, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(1, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(2, INERF_PP_NONE, 0, 0)   INERF_PP_STEP(3, INERF_PP_NONE, 0, 0)
    }
#undef INERF_PP_NONE
#if INERF_GEMM_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (UNITS > 0) {      // pinned: left free, the running maximum becomes a tree whose partial maxima stay alive
        unsigned a = __builtin_bit_cast(unsigned, amax2);
        asm volatile("" : "+v"(a));
        amax2 = __builtin_bit_cast(f16x2, a);
    }
#undef INERF_PP_STEP
}

// the epilogue alone (the last half of the trunk's last layer: nothing left to run it under)
template <int ROW, int PLANE>
__device__ __forceinline__ void pp_epilogue(const f32x16 (&prev)[2], float inv, const f32x4 (&bias)[4], _Float16* dl, f16x2& amax2) {
    f16x2 ph_, pl_;
    INERF_PP_UNIT(0) INERF_PP_UNIT(1) INERF_PP_UNIT(2) INERF_PP_UNIT(3)
    __builtin_amdgcn_sched_barrier(0);
    INERF_PP_UNIT(4) INERF_PP_UNIT(5) INERF_PP_UNIT(6) INERF_PP_UNIT(7)
    unsigned a = __builtin_bit_cast(unsigned, amax2);
    asm volatile("" : "+v"(a));
    amax2 = __builtin_bit_cast(f16x2, a);
}

}  // namespace inerf
