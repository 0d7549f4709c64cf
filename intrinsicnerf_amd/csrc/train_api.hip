// train_api.hip - the network's whole backward pass behind ONE entry point of the C ABI.
//
// inerf_mlp_backward enqueues, for one network evaluation kept by inerf_encode_mlp_train, everything loss.backward()
// records for run_network + NeRF.forward / Semantic_NeRF.forward (run_nerf.py:1018 through run_nerf_helpers.py:284-321;
// trainer.py:990 through semantic_nerf.py:123-181):
//   1. the input-gradient chain (k_mlp_dgrad) with the 1-4-row heads' gradients accumulated on the way,
//   2. every 128/256-row weight-gradient product dW = dZ^T X as a split-K launch - the nine 256 x 256 products from FRAGMENT
//      slots by LDS-DMA (k_mlp_wgrad_frag), the narrow ones from rows (k_mlp_wgrad) -, the per-workgroup partial
//      tiles of ALL products side by side in one [grid, total] buffer,
//   3. one reduction over the workgroups that writes each sum straight into its place in the caller's gradient blob - the
//      reference's parameter tensors in inerf_tensor_info() order (the cat([pts, h]) and cat([feature, views]) layers are
//      assembled by column offset, padded columns are dropped).
// No host synchronisation, no allocation (caller's workspace), deterministic (fixed summation order, no float atomics).
// Until round 3 steps 2-3 were driven from Python: 26 ctypes calls + torch reductions per network and step.
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "mlp_common.h"

namespace inerf {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int64_t kAlign = 256;
inline int64_t up(int64_t bytes) { return (bytes + kAlign - 1) / kAlign * kAlign; }
constexpr int kHeadRes = 0, kHeadAs2 = 384, kHeadAlpha = 1408, kHeadBias = 1664, kHeadFloats = 1672;     // mlp_bwd.hip

// a rectangular piece of a job's [m, n] result (or of its [m] bias) and where it goes in the gradient blob.  32-bit fields:
// the table travels as a kernel argument (< 4 KB) and every offset is far below 2^31 floats.
struct Piece {
    int32_t row0, row1;    // rows [row0, row1) of the job's result (row1 == 0: unused)
    int32_t dst;           // float offset of the destination tensor's first element
    int32_t dst_ld;        // its row length
    int32_t dst_col0;      // column offset inside the destination rows
    int32_t cols;          // valid columns (the rest of the job's n is padding)
};
struct BiasPiece { int32_t row0, row1, dst; };
struct Job {
    int32_t src;           // float offset of the [m, n] tile inside a partial row
    int32_t bias_src;      // ... of its [m] column sums of G, or -1
    int32_t m, n;
    int32_t rows;          // partial rows of this job (the batched 256 x 256 products are split over fewer workgroups than the grid)
    Piece w[2];
    BiasPiece b[2];
};
constexpr int kMaxJobs = 16;
struct ReduceTable {       // device side (kernel argument)
    Job job[kMaxJobs];
    int32_t n_jobs;
    int32_t total;         // floats per partial row
    int32_t grid;          // partial rows
};
struct Launch { int g_slot, x_slot, sem; };          // host side: operands of job k
struct Table {
    ReduceTable dev;
    Launch launch[kMaxJobs];
};
static_assert(sizeof(ReduceTable) <= 3072, "kernel-argument budget");

struct ParamTable {        // float offsets of the reference's tensors inside the gradient blob
    int64_t total = 0;
    int64_t off(const inerf_net_desc& net, const std::string& name, int64_t* cols = nullptr) const {
        const int n = inerf_num_tensors(&net);
        int64_t o = 0;
        for (int i = 0; i < n; ++i) {
            const char* nm; int64_t r, c;
            inerf_tensor_info(&net, i, &nm, &r, &c);
            if (name == nm) { if (cols) *cols = c; return o; }
            o += r * (c ? c : 1);
        }
        return -1;
    }
};

int64_t param_floats(const inerf_net_desc& net) {
    const int n = inerf_num_tensors(&net);
    int64_t o = 0;
    for (int i = 0; i < n; ++i) {
        const char* nm; int64_t r, c;
        inerf_tensor_info(&net, i, &nm, &r, &c);
        o += r * (c ? c : 1);
    }
    return o;
}

// the products of kernels.mlp_weight_gradients (round 2), as data
Table build_table(const inerf_net_desc& net, int sem_rows) {
    Table tb{};
    ReduceTable& t = tb.dev;
    const ParamTable P;
    const bool obj = net.variant == INERF_VARIANT_OBJECT;
    const bool sem = net.variant == INERF_VARIANT_SSR && net.n_classes > 0;
    const int e = 3 + 6 * net.l_xyz, dv = 3 + 6 * net.l_dir;
    const std::string sh1 = obj ? "test_linear1" : "shading_linear1";
    int32_t off = 0;
    bool have_bias[SAVE_SLOTS] = {};
    int n_w[kMaxJobs] = {};
    auto add = [&](int g_slot, int x_slot, int m, int n) -> int {
        const int k = t.n_jobs++;
        Job& j = t.job[k];
        j = Job{};
        tb.launch[k] = Launch{g_slot, x_slot, 0};
        j.m = m; j.n = n;
        j.src = off; off += m * n;
        j.bias_src = -1;
        if (g_slot >= 0 && !have_bias[g_slot]) { j.bias_src = off; off += m; have_bias[g_slot] = true; }
        return k;
    };
    auto wpiece = [&](int k, int r0, int r1, const std::string& name, int col0, int cols) {
        int64_t ld;
        const int64_t o = P.off(net, name + ".weight", &ld);
        t.job[k].w[n_w[k]++] = Piece{r0, r1, (int32_t)o, (int32_t)ld, col0, cols};
    };
    auto bpiece = [&](int k, int q, int r0, int r1, const std::string& name) {
        t.job[k].b[q] = BiasPiece{r0, r1, (int32_t)P.off(net, name + ".bias")};
    };
    {   const int k = add(SAVE_H0, SAVE_ENC, kWidth, kEncCols);                  // pts_linears.0
        wpiece(k, 0, kWidth, "pts_linears.0", 0, e); bpiece(k, 0, 0, kWidth, "pts_linears.0"); }
    for (int i = 1; i < kDepth; ++i) {                                           // pts_linears.1-7 (layer 5: its h columns)
        const int k = add(SAVE_H0 + i, SAVE_H0 + i - 1, kWidth, kWidth);
        const std::string name = "pts_linears." + std::to_string(i);
        wpiece(k, 0, kWidth, name, i == kSkipInput ? e : 0, kWidth); bpiece(k, 0, 0, kWidth, name);
    }
    {   const int k = add(SAVE_H0 + kSkipInput, SAVE_ENC, kWidth, kEncCols);    // ... and its encoding columns: cat([pts, h]), helpers:290-291
        wpiece(k, 0, kWidth, "pts_linears." + std::to_string(kSkipInput), 0, e); }
    {   const int k = add(SAVE_AS1H, SAVE_H7, kWidth, kWidth);                   // albedo_linear1 | shading hidden (fragments x fragments)
        wpiece(k, 0, kHalf, "albedo_linear1", 0, kWidth); wpiece(k, kHalf, kWidth, sh1, 0, kWidth);
        bpiece(k, 0, 0, kHalf, "albedo_linear1"); bpiece(k, 1, kHalf, kWidth, sh1); }
    {   const int k = add(SAVE_FEAT, SAVE_H7, kWidth, kWidth);
        wpiece(k, 0, kWidth, "feature_linear", 0, kWidth); bpiece(k, 0, 0, kWidth, "feature_linear"); }
    {   const int k = add(SAVE_VH, SAVE_FEAT, kHalf, kWidth);                     // views_linears.0 over cat([feature, views]), helpers:308 (rows x fragments)
        wpiece(k, 0, kHalf, "views_linears.0", 0, kWidth); bpiece(k, 0, 0, kHalf, "views_linears.0"); }
    {   const int k = add(SAVE_VH, SAVE_DIR, kHalf, kDirCols);
        wpiece(k, 0, kHalf, "views_linears.0", kWidth, dv); }
    if (sem) {
        {   const int k = add(SAVE_SEMH, SAVE_H7, kHalf, kWidth);                 // (rows x fragments)
            wpiece(k, 0, kHalf, "semantic_linear.0.0", 0, kWidth); bpiece(k, 0, 0, kHalf, "semantic_linear.0.0"); }
        {   const int k = add(-1, SAVE_SEMH, sem_rows, kHalf);                   // semantic_linear.1: G = padded d_logits
            tb.launch[k].sem = 1;
            t.job[k].bias_src = off; off += sem_rows;
            wpiece(k, 0, net.n_classes, "semantic_linear.1", 0, kHalf); bpiece(k, 0, 0, net.n_classes, "semantic_linear.1"); }
    }
    t.total = off;
    return tb;
}

struct HeadDst { int64_t res_w, as2_w, sh2_w, alpha_w, as2_b, sh2_b, res_b, alpha_b; };

// Everything above is a function of the network description alone and costs ~10^5 small string operations to derive
// (inerf_tensor_info rebuilds its table per call): derived once per description, then looked up - a training step calls
// inerf_mlp_backward twice and must not spend milliseconds of host time there (first version: 13.6 ms per step instead of 11.2).
struct Cached {
    inerf_net_desc key;
    Table table;
    HeadDst heads;
    int64_t n_params;
    int sem_rows;
};

const Cached& cached(const inerf_net_desc& net) {
    static std::mutex mu;
    static std::vector<Cached*> all;                     // entries are never freed or moved: references stay valid
    std::lock_guard<std::mutex> lock(mu);
    for (const Cached* c : all)
        if (c->key.variant == net.variant && c->key.n_classes == net.n_classes && c->key.l_xyz == net.l_xyz && c->key.l_dir == net.l_dir) return *c;
    Cached* c = new Cached{};
    c->key = net;
    const bool sem = net.variant == INERF_VARIANT_SSR && net.n_classes > 0;
    c->sem_rows = sem ? (net.n_classes <= kHalf ? kHalf : kWidth) : 0;
    c->table = build_table(net, c->sem_rows);
    c->n_params = param_floats(net);
    const ParamTable P;
    const bool obj = net.variant == INERF_VARIANT_OBJECT;
    const std::string sh2 = obj ? "test_linear2" : "shading_linear2", res = obj ? "shading_linear" : "residual_linear";
    HeadDst& d = c->heads;
    d.res_w = P.off(net, res + ".weight");         d.res_b = P.off(net, res + ".bias");
    d.as2_w = P.off(net, "albedo_linear2.weight"); d.as2_b = P.off(net, "albedo_linear2.bias");
    d.sh2_w = P.off(net, sh2 + ".weight");         d.sh2_b = P.off(net, sh2 + ".bias");
    d.alpha_w = P.off(net, "alpha_linear.weight"); d.alpha_b = P.off(net, "alpha_linear.bias");
    all.push_back(c);
    return *c;
}

struct Plan {
    int64_t dz, scalars, heads, partial, gsem, total;
    int bwd_grid, wg_grid, sem_rows;
};

Plan make_plan(const inerf_net_desc& net, int64_t n_points) {
    Plan p{};
    const Cached& c = cached(net);
    p.sem_rows = c.sem_rows;
    p.bwd_grid = inerf_mlp_backward_grid(n_points);
    p.wg_grid = inerf_wgrad_grid(n_points);
    const ReduceTable& t = c.table.dev;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off += up(bytes); return o; };
    p.dz = take(save_total_floats(net, n_points) * 4);
    p.scalars = take(16 * 4);
    p.heads = take((int64_t)p.bwd_grid * kHeadFloats * 4);
    p.partial = take((int64_t)p.wg_grid * t.total * 4);
    p.gsem = take(n_points * (int64_t)p.sem_rows * 4);
    p.total = off;
    return p;
}

// scalars: [0] dz_max  [1] gsem_max  [4..5] ranges {dz_max, act_max}  [8..9] sem ranges {gsem_max, act_max}
// (One atomic per WAVE on one address made this copy of 200 MB take 2.1 ms of the SSR step's 13: 786 000 atomics on a single cache line
// at ~2.7 ns each - the lesson of round 4's k_repack_gmax, missed here until round 5.  Now: a grid-stride loop over float4 pieces, one
// maximum per workgroup through LDS, and an atomic only if that maximum beats what the word already holds.)
__global__ __launch_bounds__(256) void k_sem_pad(const float* __restrict__ d_raw, int channels, int n_classes, int rows, int64_t n_points,
                                                 float* __restrict__ g, float* __restrict__ gmax) {
    __shared__ float wave_max[4];
    const int64_t n4 = n_points * rows / 4;                  // rows is a multiple of 4: a piece never straddles a point
    float m = 0.0f;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = 4 * q, p = i / rows;
        const int c = (int)(i - p * rows);
        const float* src = d_raw + p * channels + INERF_BASE_CHANNELS;
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = c + k < n_classes ? src[c + k] : 0.0f;
            m = fmaxf(m, fabsf(v[k]));
        }
        *reinterpret_cast<f32x4*>(g + i) = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        // non-negative floats order like their bit patterns; NaN (m != m) is left out as before
        if (m > 0.0f && m == m && m > *reinterpret_cast<volatile float*>(gmax)) atomicMax(reinterpret_cast<unsigned int*>(gmax), __builtin_bit_cast(unsigned int, m));
    }
}

__global__ void k_ranges(float* __restrict__ s, const float* __restrict__ act_max) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float a = act_max[0];
        s[4] = s[0]; s[5] = a;
        s[8] = fmaxf(s[1], 1e-30f); s[9] = a;
    }
}

// one thread per FOUR consecutive elements of a partial row (every tile and bias vector starts and ends on a multiple of 4, and
// a tile's rows are multiples of 32 long, so a quad never straddles a row or a job): sum over the workgroups' rows with 16-byte
// loads, eight rows in flight, then scatter.  The [grid, total] buffer (1 GB for the reference's fine batch) was just written
// by the products and does not fit any cache: this kernel runs at HBM speed or not at all.
__global__ void k_reduce_scatter(const ReduceTable t, const float* __restrict__ partial, float* __restrict__ grads) {
    const int64_t i = 4 * (blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
    if (i >= t.total) return;
    int rows = t.grid;
    for (int k = 0; k < t.n_jobs; ++k) {
        const Job& j = t.job[k];
        if ((i >= j.src && i < j.src + j.m * j.n) || (j.bias_src >= 0 && i >= j.bias_src && i < j.bias_src + j.m)) rows = j.rows;
    }
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    int g = 0;
    for (; g + 8 <= rows; g += 8) {
        f32x4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(partial + (int64_t)(g + k) * t.total + i));
#pragma unroll
        for (int k = 0; k < 8; ++k) v += r[k];           // fixed order: deterministic
    }
    for (; g < rows; ++g) v += *reinterpret_cast<const f32x4*>(partial + (int64_t)g * t.total + i);
    for (int k = 0; k < t.n_jobs; ++k) {
        const Job& j = t.job[k];
        if (i >= j.src && i < j.src + j.m * j.n) {
            const int r = (int)(i - j.src) / j.n, c = (int)(i - j.src) % j.n;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const Piece& w = j.w[q];
                if (r >= w.row0 && r < w.row1)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < w.cols) grads[w.dst + (r - w.row0) * w.dst_ld + w.dst_col0 + c + e] = v[e];
            }
            return;
        }
        if (j.bias_src >= 0 && i >= j.bias_src && i < j.bias_src + j.m) {
            const int r = (int)(i - j.bias_src);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const BiasPiece& b = j.b[q];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (r + e >= b.row0 && r + e < b.row1) grads[b.dst + (r + e - b.row0)] = v[e];
            }
            return;
        }
    }
}

// the 1-4-row heads' partials [grid][kHeadFloats]: 64 columns per workgroup, its sixteen waves take every sixteenth row
__global__ __launch_bounds__(1024) void k_reduce_heads(const float* __restrict__ heads, int grid, HeadDst d, float* __restrict__ grads) {
    __shared__ float part[16][64];
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    float v = 0.0f;
    if (i < kHeadFloats) {
        int g = rg;
        for (; g + 48 < grid; g += 64) {           // four rows in flight per thread (64 dependent reads in a row took 29 us)
            const float a = heads[(size_t)g * kHeadFloats + i], b = heads[(size_t)(g + 16) * kHeadFloats + i];
            const float c = heads[(size_t)(g + 32) * kHeadFloats + i], e = heads[(size_t)(g + 48) * kHeadFloats + i];
            v += a; v += b; v += c; v += e;         // fixed order: deterministic
        }
        for (; g < grid; g += 16) v += heads[(size_t)g * kHeadFloats + i];
    }
    part[rg][col] = v;
    __syncthreads();
    if (rg != 0 || i >= kHeadFloats) return;
    v = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][col];
    if (i < kHeadAs2) grads[d.res_w + i] = v;                                                 // residual head [3][128]
    else if (i < kHeadAlpha) {
        const int j = (i - kHeadAs2) / kWidth, c = (i - kHeadAs2) % kWidth;
        if (j < 3 && c < kHalf) grads[d.as2_w + j * kHalf + c] = v;                          // albedo_linear2 [3][128]
        else if (j == 3 && c >= kHalf) grads[d.sh2_w + (c - kHalf)] = v;                     // shading output [1][128]
    } else if (i < kHeadBias) grads[d.alpha_w + (i - kHeadAlpha)] = v;                        // alpha_linear [1][256]
    else {
        const int b = i - kHeadBias;
        if (b < 3) grads[d.as2_b + b] = v;
        else if (b == 3) grads[d.sh2_b] = v;
        else if (b < 7) grads[d.res_b + (b - 4)] = v;
        else grads[d.alpha_b] = v;
    }
}

}  // namespace
}  // namespace inerf

extern "C" int64_t inerf_param_floats(const inerf_net_desc* net) {
    if (!net || !inerf::net_supported(*net)) return INERF_E_INVALID;
    return inerf::cached(*net).n_params;
}

extern "C" int64_t inerf_mlp_backward_workspace_bytes(const inerf_net_desc* net, int64_t n_points) {
    if (!net || !inerf::net_supported(*net) || n_points < 0) return INERF_E_INVALID;
    if (n_points > inerf::kMaxTrainPoints) return INERF_E_UNSUPPORTED;
    if (n_points == 0) return 0;
    return inerf::make_plan(*net, n_points).total;
}

extern "C" int inerf_mlp_backward(const inerf_net_desc* net, const float* packed_bwd, const float* raw, const float* d_raw,
                                  const float* save, const float* act_max, int64_t n_points, uint32_t flags, float* grads_out,
                                  void* workspace, int64_t workspace_bytes, int32_t* status, void* stream_) {
    using namespace inerf;
    if (!net || !grads_out || n_points < 0) return INERF_E_INVALID;
    if (!net_supported(*net)) return INERF_E_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    const Cached& cache = cached(*net);
    const int64_t n_params = cache.n_params;
    if (n_points == 0) return record(hipMemsetAsync(grads_out, 0, n_params * 4, stream));        // no sample points: every gradient is zero
    if (!packed_bwd || !raw || !d_raw || !save || !act_max) return INERF_E_INVALID;
    if (n_points > kMaxTrainPoints) return INERF_E_UNSUPPORTED;
    const Plan plan = make_plan(*net, n_points);
    if (!workspace || workspace_bytes < plan.total) return INERF_E_WORKSPACE;
    char* ws = static_cast<char*>(workspace);
    float* dz = reinterpret_cast<float*>(ws + plan.dz);
    float* sc = reinterpret_cast<float*>(ws + plan.scalars);
    float* heads = reinterpret_cast<float*>(ws + plan.heads);
    float* partial = reinterpret_cast<float*>(ws + plan.partial);
    float* gsem = reinterpret_cast<float*>(ws + plan.gsem);
    const bool ssr = net->variant == INERF_VARIANT_SSR;
    const bool sem = ssr && net->n_classes > 0;
    const int channels = INERF_BASE_CHANNELS + (ssr ? net->n_classes : 0) + ((ssr && (flags & INERF_FLAG_ENDPOINT)) ? INERF_ENDPOINT_DIM : 0);

    hipError_t e = hipMemsetAsync(sc, 0, 16 * 4, stream);
    if (e != hipSuccess) return record(e);
    int rc = inerf_mlp_backward_inputs(net, packed_bwd, raw, d_raw, save, n_points, flags, dz, sc + 0, heads, status, stream);
    if (rc) return rc;
    if (sem) {
        const int64_t n4 = n_points * plan.sem_rows / 4;
        const int64_t blocks = (n4 + 255) / 256;
        hipLaunchKernelGGL(k_sem_pad, dim3((unsigned)(blocks < 8 * device_cus() ? blocks : 8 * device_cus())), dim3(256), 0, stream, d_raw, channels,
                           net->n_classes, plan.sem_rows, n_points, gsem, sc + 1);
    }
    hipLaunchKernelGGL(k_ranges, dim3(1), dim3(64), 0, stream, sc, act_max);
    const Table& tb = cache.table;
    ReduceTable t = tb.dev;
    t.grid = plan.wg_grid;
    const float* g_scale = dz + save_offset(*net, SAVE_ENC, n_points);        // the points' normalisers, written by the chain
    // the products whose operands are both fragment slots (the nine 256 x 256 ones, the two against the position encoding, the views
    // hidden layer's two): ONE launch, each product split over its share of the grid
    const void* fg[kMaxJobs]; const void* fx[kMaxJobs]; float* ft[kMaxJobs]; float* fb[kMaxJobs];
    int frows[kMaxJobs], fcols[kMaxJobs];
    int n_frag = 0;
    for (int k = 0; k < t.n_jobs; ++k) {
        const Launch& l = tb.launch[k];
        t.job[k].rows = t.grid;
        if (!l.sem && save_is_frag(l.g_slot, true) && save_is_frag(l.x_slot, false)) { frows[n_frag] = t.job[k].m; fcols[n_frag++] = t.job[k].n; }
    }
    if (n_frag > INERF_WGRAD_MAX_BATCH) return INERF_E_UNSUPPORTED;
    for (int k = 0, f = 0; k < t.n_jobs; ++k) {
        Job& j = t.job[k];
        const Launch& l = tb.launch[k];
        const float* G = l.sem ? gsem : dz + save_offset(*net, l.g_slot, n_points);
        const int ldg = l.sem ? plan.sem_rows : save_width(*net, l.g_slot);
        const float* X = save + save_offset(*net, l.x_slot, n_points);
        float* tile = partial + j.src;
        float* bias = j.bias_src >= 0 ? partial + j.bias_src : nullptr;
        const bool g_frag = !l.sem && save_is_frag(l.g_slot, true), x_frag = save_is_frag(l.x_slot, false);
        if (g_frag && x_frag) {
            fg[f] = G; fx[f] = X; ft[f] = tile; fb[f] = bias;
            j.rows = inerf_wgrad_frag_rows(n_points, n_frag, frows, fcols, f);
            if (j.rows > t.grid) return INERF_E_WORKSPACE;
            ++f;
            continue;
        }
        if (g_frag)      rc = inerf_mlp_weight_gradient_gfrag(G, g_scale, X, save_width(*net, l.x_slot), n_points, j.n, sc + 4, tile, bias, t.total, stream);
        else if (x_frag) rc = inerf_mlp_weight_gradient_xfrag(G, ldg, X, n_points, j.m, sc + 4, tile, bias, t.total, stream);
        else rc = inerf_mlp_weight_gradient(G, ldg, X, save_width(*net, l.x_slot), n_points, j.m, j.n, sc + (l.sem ? 8 : 4), tile, bias, t.total, stream);
        if (rc) return rc;
    }
    if (n_frag) {
        rc = inerf_mlp_weight_gradient_frag_batch(n_frag, fg, g_scale, fx, frows, fcols, sc + 4, n_points, ft, fb, t.total, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_reduce_scatter, dim3((unsigned)((t.total / 4 + 255) / 256)), dim3(256), 0, stream, t, partial, grads_out);
    const HeadDst& d = cache.heads;
    hipLaunchKernelGGL(k_reduce_heads, dim3((kHeadFloats + 63) / 64), dim3(1024), 0, stream, heads, plan.bwd_grid, d, grads_out);
    return record(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------
// inerf_repack: a parameter set -> packed blob ON THE DEVICE, driven by the map of inerf_pack_map (uploaded once per network
// description).  After every optimiser step both blobs of both networks are rebuilt; as framework operations that was ~16
// launches per blob (two concatenations, gathers, abs, max, frexp, ldexp, where, casts, an index_put) on a step whose tail is
// bound by the host's launch rate - here it is a memset and three kernels that read the parameters where they live.
// Bit-identical to inerf_pack_weights / inerf_pack_weights_bwd (power-of-two scales, round-to-nearest-even f16 casts).
// ---------------------------------------------------------------------------------------------------------------------
namespace inerf {
namespace {

constexpr int kMaxParamTensors = 40;
struct ParamPtrs {
    const float* p[kMaxParamTensors];
    int32_t first[kMaxParamTensors + 1];      // flat index (1-based: 0 = zero padding) of every tensor's first element
    int32_t n;
};

// The tensor table in LDS: the look-up below is a chain of ~6 dependent reads per element, and read from the kernel-argument
// segment (per-lane indices: global loads) that chain was most of the three kernels' run time (33 + 13 + 5 us per blob).
struct ParamTableLds {
    const float* p[kMaxParamTensors];
    int32_t first[kMaxParamTensors + 1];
    int32_t n;
};

__device__ __forceinline__ void stage_params(const ParamPtrs& t, ParamTableLds& lds) {
    for (int i = threadIdx.x; i < t.n; i += blockDim.x) { lds.p[i] = t.p[i]; lds.first[i] = t.first[i]; }
    if (threadIdx.x == 0) lds.n = t.n;
    __syncthreads();
}

__device__ __forceinline__ const float* param_ptr(const ParamTableLds& t, int src) {   // src: 1 + index into the flat concatenation, 0 = padding
    if (src <= 0) return nullptr;
    int lo = 0, hi = t.n - 1;
    while (lo < hi) {                                                          // last tensor whose first element is <= src
        const int mid = (lo + hi + 1) >> 1;
        if (t.first[mid] <= src) lo = mid; else hi = mid - 1;
    }
    return t.p[lo] + (src - t.first[lo]);
}
__device__ __forceinline__ float param_at(const ParamTableLds& t, int src) {
    const float* at = param_ptr(t, src);
    return at ? *at : 0.0f;
}

__device__ __forceinline__ float group_scale(float gmax) {                    // 2^k with gmax * 2^k in [2^13, 2^14); 1 for 0 / inf / nan
    if (!(gmax > 0.0f) || !(gmax < __builtin_inff())) return 1.0f;
    int e;
    frexpf(gmax, &e);
    return ldexpf(1.0f, 14 - e);
}

// Eight elements per thread, their look-ups batched, and ONE atomic per block that saw a parameter (a group's row of the map is
// padded to the longest group).  The first form walked a row with 64 blocks - twelve dependent map -> table -> parameter
// look-ups per thread, 39 us per blob, four blobs per training step; one element per thread was no better (32 us): then every
// block's atomic met the others in the one cache line that holds all maxima.
__global__ __launch_bounds__(256) void k_repack_gmax(const ParamPtrs t, const int32_t* __restrict__ group_src, int longest, float* __restrict__ gmax) {
    __shared__ float wave_max[4];
    __shared__ ParamTableLds tl;
    stage_params(t, tl);
    const int g = blockIdx.y;
    float m = 0.0f;
    constexpr int U = 8;                       // elements per thread and round: the map reads, then the table look-ups, then the
    for (int j0 = blockIdx.x * blockDim.x * U + threadIdx.x; j0 < longest; j0 += gridDim.x * blockDim.x * U) {      // parameter reads, each batch in flight together
        int src[U];
#pragma unroll
        for (int u = 0; u < U; ++u) src[u] = j0 + u * 256 < longest ? group_src[(size_t)g * longest + j0 + u * 256] : 0;
        const float* at[U];
#pragma unroll
        for (int u = 0; u < U; ++u) at[u] = param_ptr(tl, src[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) m = fmaxf(m, at[u] ? fabsf(*at[u]) : 0.0f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    // fmaxf drops a NaN parameter exactly as the host packer's std::fmax does (weight_scale, pack.cpp): same scale, and the NaN
    // itself survives as the NaN halves of that element
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        // (a block whose maximum is not above what the line already holds has nothing to add; a stale read only costs a
        // redundant atomic)
        unsigned int* dst = reinterpret_cast<unsigned int*>(gmax + g);
        const unsigned int bits = __builtin_bit_cast(unsigned int, m);
        if (m > 0.0f && bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
    }
}

__global__ void k_repack_halves(const ParamPtrs t, const int32_t* __restrict__ half_src, const int32_t* __restrict__ half_grp,
                                const float* __restrict__ gmax, int64_t n_halves, _Float16* __restrict__ out) {
    __shared__ ParamTableLds tl;
    stage_params(t, tl);
    const int64_t h = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (h >= n_halves) return;
    const int grp = half_grp[h];
    float vs = param_at(tl, half_src[h]) * group_scale(gmax[grp >= 0 ? grp : -grp - 1]);
    // keep the product and its conversion apart: fused, the compiler emits v_fma_mixlo_f16(a, b, +0), and (-0 * s) + 0 is +0 -
    // the sign of a zero weight would differ from the host packer's (found by tests/test_repack_gpu.py on an all-zero layer)
    asm volatile("" : "+v"(vs));
    const _Float16 hi = (_Float16)vs;
    out[h] = grp >= 0 ? hi : (_Float16)(vs - (float)hi);
}

__global__ void k_repack_consts(const ParamPtrs t, const int32_t* __restrict__ c_dst, const int32_t* __restrict__ c_src,
                                const int32_t* __restrict__ c_grp, const int32_t* __restrict__ c_code, const float* __restrict__ c_mult,
                                const float* __restrict__ gmax, int n, float* __restrict__ out) {
    __shared__ ParamTableLds tl;
    stage_params(t, tl);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int code = c_code[i];
    float v;
    if (code == 0) v = param_at(tl, c_src[i]) * c_mult[i];
    else {
        const float inv = 1.0f / group_scale(gmax[c_grp[i]]);
        v = code == 1 ? inv : inv * 0.125f;
    }
    out[c_dst[i]] = v;
}

}  // namespace
}  // namespace inerf

extern "C" int inerf_repack(const float* const* params /*[host] device pointers, canonical order*/, const int64_t* counts /*[host]*/,
                            int n_tensors, const int32_t* half_src, const int32_t* half_grp, int64_t packed_floats,
                            const int32_t* group_src, int n_groups, int longest, const int32_t* c_dst, const int32_t* c_src,
                            const int32_t* c_grp, const int32_t* c_code, const float* c_mult, int n_consts, float* gmax_scratch,
                            float* packed_out, void* stream_) {
    using namespace inerf;
    if (!params || !counts || n_tensors < 1 || n_tensors > kMaxParamTensors || !half_src || !half_grp || packed_floats <= 0 || !group_src ||
        n_groups < 1 || longest < 1 || !gmax_scratch || !packed_out || (n_consts > 0 && (!c_dst || !c_src || !c_grp || !c_code || !c_mult)))
        return INERF_E_INVALID;
    ParamPtrs t{};
    int64_t first = 1;
    for (int i = 0; i < n_tensors; ++i) {
        if (!params[i] || counts[i] < 0 || first + counts[i] >= ((int64_t)1 << 31)) return INERF_E_INVALID;
        t.p[i] = params[i];
        t.first[i] = (int32_t)first;
        first += counts[i];
    }
    t.first[n_tensors] = (int32_t)first;
    t.n = n_tensors;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(gmax_scratch, 0, sizeof(float) * (size_t)n_groups, stream);
    if (e != hipSuccess) return record(e);
    const int chunks = longest < 2048 * 1024 ? (longest + 2047) / 2048 : 1024;             // 8 elements per thread (k_repack_gmax)
    hipLaunchKernelGGL(k_repack_gmax, dim3(chunks, n_groups), dim3(256), 0, stream, t, group_src, longest, gmax_scratch);
    const int64_t n_halves = 2 * packed_floats;
    hipLaunchKernelGGL(k_repack_halves, dim3((unsigned)((n_halves + 255) / 256)), dim3(256), 0, stream, t, half_src, half_grp, gmax_scratch,
                       n_halves, reinterpret_cast<_Float16*>(packed_out));
    if (n_consts > 0)
        hipLaunchKernelGGL(k_repack_consts, dim3((n_consts + 255) / 256), dim3(256), 0, stream, t, c_dst, c_src, c_grp, c_code, c_mult,
                           gmax_scratch, n_consts, packed_out);
    return record(hipGetLastError());
}
