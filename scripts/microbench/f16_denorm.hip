// f16_denorm.hip - do v_mfma_f32_32x32x16_f16 and v_cvt_pkrtz_f16_f32 keep f16 subnormals on gfx950?
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/f16_denorm.hip -o /tmp/f16_denorm && /tmp/f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float tiny) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
    // A[row = lane & 31][k = 8 (lane >> 5) + i]; B[k][col = lane & 31]: A[r][0] = tiny (f16 subnormal), B[0][c] = 1
    if (lane < 32) { a[0] = (_Float16)tiny; b[0] = (_Float16)1.0f; }
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) {
        out[0] = acc[0];                                   // tiny * 1 through the matrix core
        out[1] = (float)(_Float16)tiny;                    // the f16 value itself
        auto pk = __builtin_amdgcn_cvt_pkrtz(tiny, tiny * 0.5f);
        out[2] = (float)pk[0]; out[3] = (float)pk[1];      // fp32 -> f16 (round toward zero) in the subnormal range
        // B subnormal, A one
        f16x8 a2 = a, b2 = b; a2[0] = (_Float16)1.0f; b2[0] = (_Float16)tiny;
        f32x16 acc2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, acc2, 0, 0, 0);
        out[4] = acc2[0];
    }
}
int main() {
    float* d; hipMalloc(&d, 64);
    for (float tiny : {3.0e-5f, 1.0e-6f, 6.2e-5f}) {
        hipMemset(d, 0, 64);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, tiny);
        float h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("tiny %.3e: mfma(A=tiny) %.6e  f16(tiny) %.6e  cvt_pkrtz %.6e %.6e  mfma(B=tiny) %.6e\n", tiny, h[0], h[1], h[2], h[3], h[4]);
    }
    return 0;
}
