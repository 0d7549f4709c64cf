// mfma_shapes.hip - which f16 MFMA shape does the most work per joule?  The encode+MLP kernels are held by the board's power budget
// (profiles/r06_weight_stream_ab.txt), so the question is energy per MAC: v_mfma_f32_32x32x16_f16 (16 384 MAC, 1024 accumulator
// values read and written, 1024 operand halfs) against v_mfma_f32_16x16x32_f16 (8 192 MAC, 256 accumulator values, 1024 operand
// halfs), both back to back on register operands, random / half-zero ("post-ReLU") / zero data, one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes 600000
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>        // 0: 32x32x16 (8 accumulators of 16 registers), 1: 16x16x32 (16 accumulators of 4 registers)
__global__ __launch_bounds__(256) void k_mfma(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)(tid * 8 + i) * 8) % (1 << 20));
        b[i] = *reinterpret_cast<const f16x8*>(src + (1 << 20) + ((size_t)(tid * 8 + 4 + i) * 8) % (1 << 20));
    }
    float s = 0.0f;
    if constexpr (SHAPE == 0) {
        f32x16 acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k & 3], b[(k >> 1) & 3], acc[k], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[k][r];
    } else {
        f32x4 acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = 0; it < iters; ++it) {        // the same MACs per iteration: 16 x 8192
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k & 3], b[(k >> 2) & 3], acc[k], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[k][r];
    }
    out[tid] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    std::vector<_Float16> h(2 << 20);
    _Float16* d_src;
    float* d_out;
    hipMalloc(&d_src, h.size() * sizeof(_Float16));
    hipMalloc(&d_out, (size_t)cus * 2 * 256 * sizeof(float));
    const char* names[] = {"random", "B half zeros (post-ReLU)", "A small (lo halves: |v| < 2^-9 of B's), B random", "zero"};
    for (int data = 0; data < 4; ++data) {
        unsigned lcg = 12345u;
        for (size_t i = 0; i < h.size(); ++i) {
            lcg = lcg * 1664525u + 1013904223u;
            float v = ((int)(lcg >> 9) % 4096 - 2048) / 512.0f;
            const bool is_b = i >= (1u << 20);
            if (data == 1 && is_b && ((lcg >> 5) & 1)) v = 0.0f;
            if (data == 2 && !is_b) v *= 1.0f / 512.0f;
            if (data == 3) v = 0.0f;
            h[i] = (_Float16)v;
        }
        hipMemcpy(d_src, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
        for (int shape = 0; shape < 2; ++shape)
            for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
                const int blocks = cus * waves_per_simd;
                auto launch = [&](int n) {
                    if (shape == 0) hipLaunchKernelGGL(k_mfma<0>, dim3(blocks), dim3(256), 0, 0, d_src, d_out, n);
                    else hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, d_src, d_out, n);
                };
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                launch(iters / 10);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                launch(iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0.0f;
                hipEventElapsedTime(&ms, e0, e1);
                const double macs = (double)blocks * 4 * iters * 8 * 16384.0;
                printf("%-52s %s, %d wave(s)/SIMD: %7.1f ms  %5.0f TFLOP/s dense f16\n", names[data], shape ? "16x16x32" : "32x32x16", waves_per_simd, ms,
                       macs * 2.0 / (ms * 1e-3) / 1e12);
            }
    }
    return 0;
}
