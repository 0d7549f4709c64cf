// mlp_common.h - shared between the two encode+MLP kernels (mlp.hip: exact fp32 MFMA; mlp_f16.hip:
// f16 hi/lo split operands) and their launcher.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace inerf {

struct MlpParams {
    const float* wts;       // packed blob
    const float* rays;      // [N,11]
    const float* z;         // [N,S]
    float* raw;             // [N*S, channels]
    int32_t* status;        // optional device word(s) for INERF_STATUS_* bits
    int status_rays;        // 0: one word; > 0: one word per that many rays (inerf_encode_mlp_chunked)
    float* save;            // training forward: activation buffer (layout.h SaveSlot), else nullptr
    float* act_max;         // training forward, optional: device float that receives max |activation| (caller zeroes it)
    float* sem_scratch;     // inference, optional scratch (sem_scratch_bytes): SSR - per-workgroup slots of the channel-split semantic head; object-level 128-point tile - the parked position encoding (sem_scratch_bytes)
    int64_t save_off[SAVE_SLOTS];   // float offsets of the slots for this launch's n_points
    int64_t bits_off;       // float offset of the ReLU-mask area (layout.h relu_bits_offset)
    NetLayout L;
    int n_points;           // N*S  (< 2^31, checked on the host)
    int n_samples;
    int n_tiles;
    int channels;
    int n_classes;
    int endpoint;
    int l_xyz, l_dir;
    float xyz_div;
};

// Staggered start of the input-gradient chain.  Every workgroup walks the same stages at the same pace, so the whole chip asks HBM for the same kind of rows at the same moment - 64 KB per CU x 256 CUs is
// 6 000 cycles of HBM whatever the kernel does meanwhile.  A start offset of ((b ^ (b >> 3)) & 7) x units x 2 048 cycles spreads
// those bursts over an eighth of a tile each: chain 2.20 -> 2.04 ms on 393 216 points.
// (No effect on the training forward, whose two workgroups per CU drift apart by themselves: 2.13 vs 2.12 ms.)
#ifdef __HIPCC__
__device__ __forceinline__ void stagger_start(int units) {
    const int b = blockIdx.x;
    for (int d = ((b ^ (b >> 3)) & 7) * units; d > 0; --d) __builtin_amdgcn_s_sleep(32);
}
#endif
int stagger_units(int n_tiles, int grid);       // 0 when a workgroup has fewer than four tiles

// Training entry points: an activation / gradient slot ([n_points, 256] fp32) is addressed through one buffer descriptor
// (32-bit range), so one evaluation kept for a backward pass is limited to this many sample points; callers split larger
// batches into several evaluations (kernels.mlp_train does: parameter gradients add up).
constexpr int64_t kMaxTrainPoints = 4000000;

int device_cus();                 // CU count of the CURRENT device (cached per device id)
int current_device();             // hipGetDevice, 0 on error
int tile_blocks();

// "Has this been done for the current device?" - function attributes (dynamic LDS size) belong to the device a kernel
// is loaded on, and a process may drive several (the Python launchers switch devices per call), so every launcher keeps
// one of these per kernel variant instead of a process-wide bool.  Lock-free; a race only repeats an idempotent call.
struct PerDeviceOnce {
    unsigned long long done[4] = {0, 0, 0, 0};          // bit d of word d/64: devices 0..255
    bool first() {
        const int d = current_device() & 255;
        const unsigned long long bit = 1ull << (d & 63);
        if (__atomic_load_n(&done[d >> 6], __ATOMIC_ACQUIRE) & bit) return false;
        return true;
    }
    void mark() {
        const int d = current_device() & 255;
        __atomic_fetch_or(&done[d >> 6], 1ull << (d & 63), __ATOMIC_RELEASE);
    }
};
int record(hipError_t e);
int launch_mlp_f32(MlpParams& p, int64_t n_points, bool ssr, hipStream_t stream);
int launch_mlp_f16x3(MlpParams& p, int64_t n_points, bool ssr, hipStream_t stream);   // p.save != nullptr: saving variant
int launch_mlp_f16x3_t128(MlpParams& p, int64_t n_points, bool ssr, hipStream_t stream);      // inference on the 128-point tile (mlp_f16_t128.hip)
bool mlp_f16x3_takes_t128(bool ssr, bool save, bool endpoint, int n_classes);                   // which launches take it
int64_t enc_cache_bytes_t128(int64_t n_points);                                                 // object-level inference on the 128-point tile: the parked encoding
int64_t sem_scratch_bytes_t128(int64_t n_points);                                               // ... and the SSR form's scratch (classes > 0)
int64_t sem_scratch_bytes(const inerf_net_desc& net, int64_t n_points, bool endpoint);   // 0 when the launch would not use one

}  // namespace inerf
