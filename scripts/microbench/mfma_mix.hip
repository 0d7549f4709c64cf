// mfma_mix.hip - what the companions of the MFMAs cost under the board's power budget.  Baseline: every wave issues
// v_mfma_f32_32x32x16_f16 back to back on random register operands (mfma_peak.hip).  Variants add, per 12 MFMAs (one k-block
// step of the encode+MLP kernel): 4 x 1 KiB weight-fragment loads from an L2-resident 2.6 MB buffer, 4 ds_read_b128 of
// activation fragments, or 48 VALU operations (the epilogue's share) - one at a time and all together.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWeightBytes = 2600 * 1024;

template <int kLoads, bool kLds, bool kValu>
__global__ __launch_bounds__(256, 1) void k_mix(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[64 * 616];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 616; i += 256) lds[i] = src[i];
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(src), 0, kWeightBytes, 0x00020000);
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(src + (size_t)(tid * 8 + i) * 8);
        b[i] = *reinterpret_cast<const f16x8*>(src + (size_t)(tid * 8 + 4 + i) * 8);
    }
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    float v[8] = {1.0f, 1.1f, 1.2f, 1.3f, 1.4f, 1.5f, 1.6f, 1.7f};
    int woff = wave * 4096;
    // operands for step it + 1 are requested before the MFMAs of step it (two register sets, unrolled by two): latencies are
    // covered as in the real kernel, what remains is issue slots and power
    f16x8 a2[4], b2[4], a3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a2[i] = a[i]; b2[i] = b[i]; a3[i] = a[i]; }
#define MIX_STEP(ACUR, BCUR, ANXT, BNXT, IT)                                                                                   \
    {                                                                                                                          \
        if (kLoads) {                                                                                                          \
            _Pragma("unroll") for (int i = 0; i < kLoads; ++i)                                                                 \
                ANXT[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, woff + i * 1024, 0)); \
            woff += 16384;                                                                                                     \
            if (woff >= kWeightBytes - 16384) woff = wave * 4096;                                                              \
        }                                                                                                                      \
        if (kLds) {                                                                                                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
                BNXT[i] = *reinterpret_cast<const f16x8*>(lds + (lane & 31) * 616 + 8 * (lane >> 5) + 16 * (((IT) + i) & 31));  \
        }                                                                                                                      \
        _Pragma("unroll") for (int k = 0; k < 12; ++k)                                                                         \
            acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACUR[k & 3], BCUR[(k >> 2) & 3], acc[k & 3], 0, 0, 0);          \
        if (kValu) {                                                                                                           \
            _Pragma("unroll") for (int r = 0; r < 6; ++r)                                                                      \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 0.999f, 0.001f * (float)r);          \
        }                                                                                                                      \
    }
    // weights two steps ahead (three register sets), activations one step ahead - as in the real kernel
    for (int it = 0; it < iters; it += 6) {
        MIX_STEP(a, b, a3, b2, it)
        MIX_STEP(a2, b2, a, b, it + 1)
        MIX_STEP(a3, b, a2, b2, it + 2)
        MIX_STEP(a, b2, a3, b, it + 3)
        MIX_STEP(a2, b, a, b2, it + 4)
        MIX_STEP(a3, b2, a2, b, it + 5)
    }
#undef MIX_STEP
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[k][r];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + tid] = s + (float)a[0][0] + (float)b[0][0];
}

template <int L, bool D, bool V>
void run(const char* name, const _Float16* d_src, float* d_out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<L, D, V>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<L, D, V>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    const double tflops = (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("%-44s %7.1f ms  %5.0f TFLOP/s dense f16  (%4.0f algorithmic at 3 products per MAC)\n", name, ms, tflops, tflops / 3);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount;
    std::vector<_Float16> h(kWeightBytes / 2);
    unsigned lcg = 12345u;
    for (auto& x : h) { lcg = lcg * 1664525u + 1013904223u; x = (_Float16)(((int)(lcg >> 9) % 4096 - 2048) / 512.0f); }
    _Float16* d_src;
    float* d_out;
    hipMalloc(&d_src, kWeightBytes);
    hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(float));
    hipMemcpy(d_src, h.data(), kWeightBytes, hipMemcpyHostToDevice);
    run<0, false, false>("MFMAs only", d_src, d_out, blocks, iters);
    run<4, false, false>("+ 4 KiB of weights from L2 per 12 MFMAs", d_src, d_out, blocks, iters);
    run<2, false, false>("+ 2 KiB of weights from L2 per 12 MFMAs", d_src, d_out, blocks, iters);
    run<1, false, false>("+ 1 KiB of weights from L2 per 12 MFMAs", d_src, d_out, blocks, iters);
    run<0, true, false>("+ 4 ds_read_b128 per 12 MFMAs", d_src, d_out, blocks, iters);
    run<0, false, true>("+ 48 VALU per 12 MFMAs", d_src, d_out, blocks, iters);
    run<4, true, true>("+ 4 KiB weights + LDS reads + VALU", d_src, d_out, blocks, iters);
    run<2, true, true>("+ 2 KiB weights + LDS reads + VALU", d_src, d_out, blocks, iters);
    return 0;
}
