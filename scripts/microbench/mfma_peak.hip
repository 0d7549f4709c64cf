// mfma_peak.hip - what the f16 matrix pipe of this board sustains when it does nothing else: every wave issues
// v_mfma_f32_32x32x16_f16 back to back on register operands (8 independent accumulators), no memory traffic in the loop.
// The result is the practical ceiling the encode+MLP kernels are measured against in DESIGN.md: the name-plate 2.5 PFLOP/s
// assumes 2.4 GHz, the board clocks to its power budget.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_mfma(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)(tid * 8 + i) * 8) % (1 << 20));
        b[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)(tid * 8 + 4 + i) * 8) % (1 << 20));
    }
    f32x16 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k & 3], b[(k >> 1) & 3], acc[k], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[tid] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    std::vector<_Float16> h(1 << 20);
    _Float16* d_src;
    float* d_out;
    hipMalloc(&d_src, h.size() * sizeof(_Float16));
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int blocks = cus * waves_per_simd;           // 256 threads = 4 waves = one wave per SIMD and block
        hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(float));
        for (int zero = 0; zero < 2; ++zero) {
            unsigned lcg = 12345u;
            for (auto& v : h) {
                lcg = lcg * 1664525u + 1013904223u;
                v = zero ? (_Float16)0.0f : (_Float16)(((int)(lcg >> 9) % 4096 - 2048) / 512.0f);
            }
            hipMemcpy(d_src, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d_src, d_out, iters / 10);    // warm-up
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d_src, d_out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.0f;
            hipEventElapsedTime(&ms, e0, e1);
            const double mfmas = (double)blocks * 4 * iters * 8;
            const double tflops = mfmas * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
            // a SIMD issues one such MFMA per 32 cycles: per-SIMD MFMA count * 32 / time = clock the matrix pipe ran at
            const double ghz = (double)iters * 8 * waves_per_simd * 32 / (ms * 1e-3) / 1e9;
            printf("%d wave(s)/SIMD, %s operands: %.1f ms, %.0f TFLOP/s dense f16 (%.0f %% of 2500), matrix-pipe clock if never idle %.2f GHz\n",
                   waves_per_simd, zero ? "zero  " : "random", ms, tflops, tflops / 25.0, ghz);
        }
        hipFree(d_out);
    }
    return 0;
}
