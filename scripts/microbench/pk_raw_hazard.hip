// pk_raw_hazard.hip - is the result of a packed-fp32 VALU instruction (v_pk_mul_f32 / v_pk_mov_b32: two passes per wave on gfx9x0) visible
// to an LDS / global store issued right behind it, in every lane?  (Suspected in round 5: lanes 48..63 of a per-point scratch row.)
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/pk_raw_hazard.hip -o /tmp/pk_raw && /tmp/pk_raw
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIters = 3000;
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* __restrict__ out, unsigned* __restrict__ bad, unsigned* __restrict__ bad_hi) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 + 8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave != 0) {
        float acc = 0;
        for (int it = 0; it < kIters * 4; ++it) { lds[256 + ((tid * 32 + it) & 8191)] = acc; acc += lds[256 + ((tid * 64 + it * 7) & 8191)]; }
        if (acc == 12345.0f) out[0] = 1;
        return;
    }
    unsigned errors = 0, errors_hi = 0;
    const unsigned laddr = lane * 16;
    for (int it = 0; it < kIters; ++it) {
        const float a = 1.0f + (float)((it * 64 + lane) & 1023), b = 3.0f;
        // v[100:103] = old garbage; then v[100:101] = (a, a) * b ; v[102:103] = (a, a) * b by packed multiplies, store right behind
        if (MODE == 0)
            asm volatile("v_mov_b32 v100, -1.0\n v_mov_b32 v101, -1.0\n v_mov_b32 v102, -1.0\n v_mov_b32 v103, -1.0\n v_mov_b32 v104, %1\n v_mov_b32 v105, %1\n v_mov_b32 v106, %2\n v_mov_b32 v107, %2\n s_nop 7\n"
                         "v_pk_mul_f32 v[100:101], v[104:105], v[106:107] op_sel_hi:[1,0]\n v_pk_mul_f32 v[102:103], v[104:105], v[106:107] op_sel_hi:[1,0]\n"
                         "ds_write_b128 %0, v[100:103]\n s_waitcnt lgkmcnt(0)\n"
                         :: "v"(laddr), "v"(a), "v"(b) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory");
        if (MODE == 1)
            asm volatile("v_mov_b32 v100, -1.0\n v_mov_b32 v101, -1.0\n v_mov_b32 v102, -1.0\n v_mov_b32 v103, -1.0\n v_mov_b32 v104, %1\n v_mov_b32 v105, %1\n v_mov_b32 v106, %2\n v_mov_b32 v107, %2\n s_nop 7\n"
                         "v_pk_mul_f32 v[100:101], v[104:105], v[106:107] op_sel_hi:[1,0]\n v_pk_mul_f32 v[102:103], v[104:105], v[106:107] op_sel_hi:[1,0]\n s_nop 0\n"
                         "ds_write_b128 %0, v[100:103]\n s_waitcnt lgkmcnt(0)\n"
                         :: "v"(laddr), "v"(a), "v"(b) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory");
        if (MODE == 2)       // the chain's sequence: two packed multiplies, the write, then the registers are overwritten by a packed move
            asm volatile("v_mov_b32 v100, -1.0\n v_mov_b32 v101, -1.0\n v_mov_b32 v102, -1.0\n v_mov_b32 v103, -1.0\n v_mov_b32 v104, %1\n v_mov_b32 v105, %1\n v_mov_b32 v106, %2\n v_mov_b32 v107, %2\n v_mov_b32 v108, -2.0\n v_mov_b32 v109, -2.0\n s_nop 7\n"
                         "v_pk_mul_f32 v[100:101], v[104:105], v[106:107] op_sel_hi:[1,0]\n v_pk_mul_f32 v[102:103], v[104:105], v[106:107] op_sel_hi:[1,0]\n"
                         "ds_write_b128 %0, v[100:103]\n v_pk_mov_b32 v[100:101], v[108:109], v[108:109] op_sel:[1,0]\n v_pk_mul_f32 v[100:101], v[108:109], v[106:107] op_sel_hi:[1,0]\n v_pk_mul_f32 v[102:103], v[108:109], v[106:107] op_sel_hi:[1,0]\n s_waitcnt lgkmcnt(0)\n"
                         :: "v"(laddr), "v"(a), "v"(b) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "memory");
        if (MODE == 3)       // the divide's tail in front: v_div_fixup_f32 -> packed multiply by its result -> write
            asm volatile("v_mov_b32 v100, -1.0\n v_mov_b32 v101, -1.0\n v_mov_b32 v102, -1.0\n v_mov_b32 v103, -1.0\n v_mov_b32 v104, %1\n v_mov_b32 v105, %1\n v_mov_b32 v106, -5.0\n s_nop 7\n"
                         "v_div_fixup_f32 v106, %2, %2, %2\n"      /* = b (quotient operand passed through for finite inputs? no: fixes up; use as a VALU producer) */
                         "v_mov_b32 v106, %2\n"
                         "v_pk_mul_f32 v[100:101], v[104:105], v[106:107] op_sel_hi:[1,0]\n v_pk_mul_f32 v[102:103], v[104:105], v[106:107] op_sel_hi:[1,0]\n"
                         "ds_write_b128 %0, v[100:103]\n s_waitcnt lgkmcnt(0)\n"
                         :: "v"(laddr), "v"(a), "v"(b) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory");
        const float* rd = lds + lane * 4;
        unsigned e = 0;
        for (int c = 0; c < 4; ++c) e += rd[c] != a * b;
        errors += e;
        if (lane >= 48) errors_hi += e;
    }
    atomicAdd(bad, errors);
    atomicAdd(bad_hi, errors_hi);
}
template <int MODE> void run(unsigned* out, unsigned* bad, const char* what) {
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(512), 0, 0, out, bad, bad + 1);
    hipDeviceSynchronize();
    unsigned h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-90s wrong floats %9u of %.0f (lanes 48..63: %u)\n", what, h[0], 1024.0 * kIters * 256, h[1]);
}
int main() {
    unsigned *out, *bad; hipMalloc(&out, 4096); hipMalloc(&bad, 64);
    run<0>(out, bad, "2 x v_pk_mul_f32 -> ds_write_b128 of the results, back to back");
    run<1>(out, bad, "2 x v_pk_mul_f32, s_nop 0, ds_write_b128");
    run<2>(out, bad, "2 x v_pk_mul_f32 -> ds_write_b128 -> v_pk_mov_b32 / v_pk_mul_f32 over the same registers");
    run<3>(out, bad, "v_mov -> 2 x v_pk_mul_f32 (scalar operand just written) -> ds_write_b128");
    return 0;
}
