// store_war_hazard2.hip - how many wait states does a 16-byte store need on gfx950 before a VALU instruction may overwrite its data
// registers?  (The compiler inserts ONE - `s_nop 0`; store_war_hazard.hip found wrong data with one.)  Wave 0 of every workgroup:
//   fill v[100:103]; <store>; s_nop N (or nothing); v_mov_b32 v100..103, 0; s_waitcnt vmcnt(0); read back.   The other seven waves keep
// the vector-memory path busy.  FORM 0: global_store_dwordx4 ... off; 1: buffer_store_dwordx4 ... offen (constant 0 soffset); 2: ... nt.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/store_war_hazard2.hip -o /tmp/store_war2 && /tmp/store_war2
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIters = 1500;
#define CASE(N, NOPS)                                                                                                   \
    if (GAP == N) {                                                                                                     \
        if (FORM == 0)                                                                                                  \
            asm volatile("v_mov_b32 v100, %1\n v_mov_b32 v101, %1\n v_mov_b32 v102, %1\n v_mov_b32 v103, %1\n s_nop 7\n" \
                         "global_store_dwordx4 %0, v[100:103], off\n" NOPS                                             \
                         "v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n s_waitcnt vmcnt(0)\n" \
                         :: "v"(gdst), "v"(val) : "v100", "v101", "v102", "v103", "memory");                            \
        else if (FORM == 1)                                                                                             \
            asm volatile("v_mov_b32 v100, %1\n v_mov_b32 v101, %1\n v_mov_b32 v102, %1\n v_mov_b32 v103, %1\n s_nop 7\n" \
                         "buffer_store_dwordx4 v[100:103], %0, %2, 0 offen\n" NOPS                                      \
                         "v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n s_waitcnt vmcnt(0)\n" \
                         :: "v"(voff), "v"(val), "s"(rsrc) : "v100", "v101", "v102", "v103", "memory");                  \
        else                                                                                                            \
            asm volatile("v_mov_b32 v100, %1\n v_mov_b32 v101, %1\n v_mov_b32 v102, %1\n v_mov_b32 v103, %1\n s_nop 7\n" \
                         "buffer_store_dwordx4 v[100:103], %0, %2, 0 offen nt\n" NOPS                                   \
                         "v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n s_waitcnt vmcnt(0)\n" \
                         :: "v"(voff), "v"(val), "s"(rsrc) : "v100", "v101", "v102", "v103", "memory");                  \
    }
template <int FORM, int GAP>
__global__ __launch_bounds__(512) void k(unsigned* __restrict__ out, unsigned* __restrict__ scratch, unsigned* __restrict__ bad, unsigned* __restrict__ bad_hi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave != 0) {                       // pressure: a stream of 16-byte global stores and loads
        uint4 acc = {1, 2, 3, 4};
        uint4* s4 = reinterpret_cast<uint4*>(scratch) + (size_t)blockIdx.x * 512 * 16 + (size_t)tid * 16;
        for (int it = 0; it < kIters * 2; ++it) {
            s4[it & 15] = acc;
            const uint4 v = s4[(it * 5 + 3) & 15];
            acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
        }
        if (acc.x == 0x12345) out[0] = acc.y;
        return;
    }
    unsigned* gdst = out + 4 + ((size_t)blockIdx.x * 64 + lane) * 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out + 4 + (size_t)blockIdx.x * 256, 0, 64 * 16, 0x00020000);
    const unsigned voff = lane * 16;
    unsigned errors = 0, errors_hi = 0;
    for (int it = 0; it < kIters; ++it) {
        const unsigned val = 0x1000000u + (unsigned)it * 64u + (unsigned)lane;
        CASE(0, "")
        CASE(1, "s_nop 0\n")
        CASE(2, "s_nop 1\n")
        CASE(3, "s_nop 2\n")
        CASE(4, "s_nop 3\n")
        CASE(6, "s_nop 5\n")
        CASE(8, "s_nop 7\n")
        CASE(16, "s_nop 7\n s_nop 7\n")
        unsigned e = 0;
        for (int c = 0; c < 4; ++c) e += __builtin_nontemporal_load(gdst + c) != val;
        errors += e;
        if (lane >= 48) errors_hi += e;
    }
    atomicAdd(bad, errors);
    atomicAdd(bad_hi, errors_hi);
}
template <int FORM, int GAP>
void run(unsigned* out, unsigned* scratch, unsigned* bad) {
    const int grid = 1024;
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL((k<FORM, GAP>), dim3(grid), dim3(512), 0, 0, out, scratch, bad, bad + 1);
    hipDeviceSynchronize();
    unsigned h[2];
    hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    const char* forms[3] = {"global_store_dwordx4 off        ", "buffer_store_dwordx4 offen      ", "buffer_store_dwordx4 offen nt   "};
    printf("%s wait states before the overwrite: %2d   wrong dwords %9u of %.0f (lanes 48..63: %u)\n", forms[FORM], GAP, h[0], 1024.0 * kIters * 256, h[1]);
}
int main() {
    unsigned *out, *scratch, *bad;
    hipMalloc(&out, (4 + (size_t)1024 * 64 * 4) * 4 + 4096);
    hipMalloc(&scratch, (size_t)1024 * 512 * 16 * 16);
    hipMalloc(&bad, 64);
    run<0, 0>(out, scratch, bad); run<0, 1>(out, scratch, bad); run<0, 2>(out, scratch, bad); run<0, 3>(out, scratch, bad); run<0, 4>(out, scratch, bad);
    run<0, 6>(out, scratch, bad); run<0, 8>(out, scratch, bad); run<0, 16>(out, scratch, bad);
    run<1, 0>(out, scratch, bad); run<1, 1>(out, scratch, bad); run<1, 2>(out, scratch, bad); run<1, 3>(out, scratch, bad); run<1, 4>(out, scratch, bad);
    run<1, 6>(out, scratch, bad); run<1, 8>(out, scratch, bad); run<1, 16>(out, scratch, bad);
    run<2, 1>(out, scratch, bad); run<2, 2>(out, scratch, bad); run<2, 4>(out, scratch, bad); run<2, 8>(out, scratch, bad);
    return 0;
}
