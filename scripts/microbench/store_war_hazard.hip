// store_war_hazard.hip - does gfx950 read the DATA / ADDRESS registers of an LDS or global store late enough that the very next
// VALU instruction of the wave can overwrite them first?  (Round 5: the two-workgroup chain came out different in 2 % of its launches -
// lanes 48..63 of two floats of a per-point scratch row - when the compiler scheduled `v_pk_mov_b32 v[162:163]` right behind
// `ds_write_b128 ..., v[162:165]`.)  Wave 0 of every workgroup issues the suspect pairs back to back; the other waves keep the LDS
// and the vector-memory queues busy.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/store_war_hazard.hip -o /tmp/store_war && /tmp/store_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kIters = 2000;
template <int MODE>      // 0: ds_write_b128 + v_mov; 1: ds_write_b128 + v_pk_mov_b32; 2: global_store_dwordx4 + v_mov (1 instr between); 3: global_store_dword + address overwrite;
                         // 4: ds_write_b128 + s_nop 0 + v_pk_mov_b32; 5: ds_write_b64 + v_pk_mov_b32
__global__ __launch_bounds__(512) void k(unsigned* __restrict__ out, unsigned* __restrict__ scratch, unsigned* __restrict__ bad) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 + 8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned errors = 0;
    if (wave != 0) {                       // pressure: conflicting LDS traffic and a stream of global stores
        unsigned acc = 0;
        for (int it = 0; it < kIters * 4; ++it) {
            lds[256 + ((tid * 32 + it) & 8191)] = acc;
            acc += lds[256 + ((tid * 64 + it * 7) & 8191)];
            if ((it & 3) == 0) scratch[(size_t)blockIdx.x * 512 * 64 + (size_t)tid * 64 + (it & 63)] = acc;
        }
        if (acc == 0x12345) out[0] = acc;
        return;
    }
    unsigned* gdst = out + 1 + ((size_t)blockIdx.x * 64 + lane) * 4;
    for (int it = 0; it < kIters; ++it) {
        const unsigned val = 0x1000000u + (unsigned)it * 64u + (unsigned)lane;
        const unsigned laddr = (unsigned)(lane * 16);
        if (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5) {
            if (MODE == 0)
                asm volatile("v_mov_b32 v100, %1\n v_mov_b32 v101, %1\n v_mov_b32 v102, %1\n v_mov_b32 v103, %1\n s_nop 4\n"
                             "ds_write_b128 %0, v[100:103]\n"
                             "v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n"
                             :: "v"(laddr), "v"(val) : "v100", "v101", "v102", "v103", "memory");
            if (MODE == 1)
                asm volatile("v_mov_b32 v100, %1\n v_mov_b32 v101, %1\n v_mov_b32 v102, %1\n v_mov_b32 v103, %1\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n s_nop 4\n"
                             "ds_write_b128 %0, v[100:103]\n"
                             "v_pk_mov_b32 v[100:101], v[104:105], v[104:105] op_sel:[1,0]\n v_pk_mov_b32 v[102:103], v[104:105], v[104:105] op_sel:[1,0]\n"
                             :: "v"(laddr), "v"(val) : "v100", "v101", "v102", "v103", "v104", "v105", "memory");
            if (MODE == 4)
                asm volatile("v_mov_b32 v100, %1\n v_mov_b32 v101, %1\n v_mov_b32 v102, %1\n v_mov_b32 v103, %1\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n s_nop 4\n"
                             "ds_write_b128 %0, v[100:103]\n s_nop 0\n"
                             "v_pk_mov_b32 v[100:101], v[104:105], v[104:105] op_sel:[1,0]\n v_pk_mov_b32 v[102:103], v[104:105], v[104:105] op_sel:[1,0]\n"
                             :: "v"(laddr), "v"(val) : "v100", "v101", "v102", "v103", "v104", "v105", "memory");
            if (MODE == 5)
                asm volatile("v_mov_b32 v100, %1\n v_mov_b32 v101, %1\n v_mov_b32 v102, %1\n v_mov_b32 v103, %1\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n s_nop 4\n"
                             "ds_write_b64 %0, v[100:101]\n ds_write_b64 %0, v[102:103] offset:8\n"
                             "v_pk_mov_b32 v[100:101], v[104:105], v[104:105] op_sel:[1,0]\n v_pk_mov_b32 v[102:103], v[104:105], v[104:105] op_sel:[1,0]\n"
                             :: "v"(laddr), "v"(val) : "v100", "v101", "v102", "v103", "v104", "v105", "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned* rd = lds + lane * 4;
            for (int c = 0; c < 4; ++c) errors += rd[c] != val;
        } else if (MODE == 2) {
            asm volatile("v_mov_b32 v100, %1\n v_mov_b32 v101, %1\n v_mov_b32 v102, %1\n v_mov_b32 v103, %1\n s_nop 4\n"
                         "global_store_dwordx4 %0, v[100:103], off\n s_nop 0\n"
                         "v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n"
                         "s_waitcnt vmcnt(0)\n"
                         :: "v"(gdst), "v"(val) : "v100", "v101", "v102", "v103", "memory");
            for (int c = 0; c < 4; ++c) errors += __builtin_nontemporal_load(gdst + c) != val;
        } else {
            asm volatile("v_mov_b32 v100, %1\n v_mov_b32 v106, %0\n v_mov_b32 v107, %2\n s_nop 4\n"
                         "global_store_dword v[106:107], v100, off\n"
                         "v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n"
                         "s_waitcnt vmcnt(0)\n"
                         :: "v"((unsigned)(size_t)gdst), "v"(val), "v"((unsigned)((size_t)gdst >> 32)) : "v100", "v106", "v107", "memory");
            errors += __builtin_nontemporal_load(gdst) != val;
        }
    }
    atomicAdd(bad + MODE, errors);
}
int main() {
    unsigned *out, *scratch, *bad;
    const int grid = 1024;
    hipMalloc(&out, (1 + (size_t)grid * 64 * 4) * 4);
    hipMalloc(&scratch, (size_t)grid * 512 * 64 * 4);
    hipMalloc(&bad, 64);
    hipMemset(bad, 0, 64);
    hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, out, scratch, bad);
    hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, out, scratch, bad);
    hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, out, scratch, bad);
    hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, out, scratch, bad);
    hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, out, scratch, bad);
    hipLaunchKernelGGL(k<5>, dim3(grid), dim3(512), 0, 0, out, scratch, bad);
    hipDeviceSynchronize();
    unsigned h[8];
    hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
    const double n = (double)grid * kIters * 64;
    const char* names[6] = {"ds_write_b128 then v_mov_b32 of its data registers", "ds_write_b128 then v_pk_mov_b32 of its data registers",
                            "global_store_dwordx4, s_nop 0, then v_mov_b32 of its data registers", "global_store_dword then v_mov_b32 of its ADDRESS registers",
                            "ds_write_b128, s_nop 0, v_pk_mov_b32", "2 x ds_write_b64 then v_pk_mov_b32"};
    for (int m = 0; m < 6; ++m) printf("%-75s wrong values: %u of %.0f lane-stores x4\n", names[m], h[m], n);
    return 0;
}
