// pack.cpp - host side of the C ABI: network description, canonical tensor order, weight packer.
// Replaces (as seen by forward()) the nn.Module parameter storage of
//   object_level/run_nerf_helpers.py:259-279 (NeRF)  and  SSR/models/semantic_nerf.py:98-118 (Semantic_NeRF).
#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "layout.h"

namespace {

struct TensorSpec {
    std::string name;
    int64_t rows, cols;   // cols == 0 for a bias
};

struct Net {
    inerf_net_desc d;
    int e, dv;            // real encoded widths
    std::vector<TensorSpec> spec;
};

void add_linear(std::vector<TensorSpec>& s, const std::string& n, int64_t out, int64_t in) {
    s.push_back({n + ".weight", out, in});
    s.push_back({n + ".bias", out, 0});
}

Net describe(const inerf_net_desc& d) {
    using namespace inerf;
    Net n{d, 3 + 6 * d.l_xyz, 3 + 6 * d.l_dir, {}};
    for (int i = 0; i < kDepth; ++i) {
        int in = i == 0 ? n.e : (i == kSkipInput ? n.e + kWidth : kWidth);
        add_linear(n.spec, "pts_linears." + std::to_string(i), kWidth, in);
    }
    add_linear(n.spec, "views_linears.0", kHalf, kWidth + n.dv);
    add_linear(n.spec, "feature_linear", kWidth, kWidth);
    add_linear(n.spec, "alpha_linear", 1, kWidth);
    if (d.variant == INERF_VARIANT_OBJECT) {
        add_linear(n.spec, "shading_linear", 3, kHalf);      // residual head (run_nerf_helpers.py:314)
        add_linear(n.spec, "albedo_linear1", kHalf, kWidth);
        add_linear(n.spec, "albedo_linear2", 3, kHalf);
        add_linear(n.spec, "test_linear1", kHalf, kWidth);   // shading head (run_nerf_helpers.py:302)
        add_linear(n.spec, "test_linear2", 1, kHalf);
    } else {
        if (d.n_classes > 0) {
            add_linear(n.spec, "semantic_linear.0.0", kHalf, kWidth);
            add_linear(n.spec, "semantic_linear.1", d.n_classes, kHalf);
        }
        add_linear(n.spec, "residual_linear", 3, kHalf);
        add_linear(n.spec, "albedo_linear1", kHalf, kWidth);
        add_linear(n.spec, "albedo_linear2", 3, kHalf);
        add_linear(n.spec, "shading_linear1", kHalf, kWidth);
        add_linear(n.spec, "shading_linear2", 1, kHalf);
    }
    return n;
}

using Elem = std::function<float(int /*row*/, int /*virtual k*/)>;

// ---- map mode ------------------------------------------------------------------------------------------------------
// The same packing code can, instead of packing values, record WHERE every packed element comes from: it is then run on
// "index tensors" (element j of tensor i holds 1 + its position in the flat concatenation of all parameters; 0 = padding)
// and writes, per half-precision position of the blob, the source index and the scale group (one group per per-GEMM
// power-of-two scale), and per fp32 constant its recipe.  The Python side turns that map into a device-side packer
// (a gather, a per-group max, a split) so that a training step never moves weights through the host
// (intrinsicnerf_amd/packing.py: DevicePacker).  Only the f16 formats are mapped.
struct FloatEntry { int32_t dst, src, group, code; float mult; };   // code 0: flat[src] * mult; 1: 1/scale[group]; 2: 1/(scale[group] * kActScale)
struct MapSink {
    float* base;                       // scratch blob the packing code writes into (index values in the float regions)
    int32_t* half_src;                 // [2 * total_floats]
    int32_t* half_grp;                 // [2 * total_floats]: g >= 0 hi of group g; g < 0 lo of group (-g - 1)
    std::vector<char> is_weight;       // per blob float
    std::vector<FloatEntry> consts;
    std::vector<std::pair<int32_t, int32_t>> times8;   // float ranges [first, last) whose biases live in the kActScale domain
    int32_t group = -1;                // group of the most recent pack_* call
    int32_t next_group = 0;
    int32_t pinned = -1;               // >= 0: the following pack_* calls share this group (a common forced scale)
};
thread_local MapSink* g_map = nullptr;

inline void map_pair(_Float16* dst_hi, _Float16* dst_lo, float index_value) {
    MapSink& m = *g_map;
    const int64_t hi = dst_hi - reinterpret_cast<_Float16*>(m.base), lo = dst_lo - reinterpret_cast<_Float16*>(m.base);
    m.half_src[hi] = m.half_src[lo] = (int32_t)index_value;
    m.half_grp[hi] = m.group;
    m.half_grp[lo] = -m.group - 1;
    m.is_weight[hi / 2] = m.is_weight[lo / 2] = 1;
}
inline void map_new_group() {
    MapSink& m = *g_map;
    m.group = m.pinned >= 0 ? m.pinned : m.next_group++;
}

// wide GEMM: fragments of v_mfma_f32_32x32x2_f32's A operand, four k-steps per float4 (layout.h)
float pack_wide_f32(float* dst, int n_out, int k_total, const Elem& w) {
    const int rb_per_wave = n_out / (32 * inerf::kWaves), kb_count = k_total / 8;
    for (int wave = 0; wave < inerf::kWaves; ++wave)
        for (int kb = 0; kb < kb_count; ++kb)
            for (int rb = 0; rb < rb_per_wave; ++rb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int c = 0; c < 4; ++c) {
                        int row = wave * 32 * rb_per_wave + 32 * rb + (lane & 31);
                        int kv = 8 * kb + 4 * (lane >> 5) + c;
                        dst[((((int64_t)wave * kb_count + kb) * rb_per_wave + rb) * 64 + lane) * 4 + c] = w(row, kv);
                    }
    return 1.0f;
}

// skinny GEMM: fragments of v_mfma_f32_16x16x4_f32's A operand
float pack_skinny_f32(float* dst, int rbs, int k_total, const Elem& w) {
    const int kb_count = k_total / 16;
    for (int rb = 0; rb < rbs; ++rb)
        for (int kb = 0; kb < kb_count; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int c = 0; c < 4; ++c) {
                    int row = 16 * rb + (lane & 15);
                    int kv = 16 * kb + 4 * (lane >> 4) + c;
                    dst[(((int64_t)rb * kb_count + kb) * 64 + lane) * 4 + c] = w(row, kv);
                }
    return 1.0f;
}

// ---- INERF_PREC_F16X3: W' = W * 2^kw, hi = f16(W'), lo = f16(W' - hi)  (layout.h) ----
struct HalfPair { _Float16 hi, lo; };

inline HalfPair split_f16(float w) {
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)(w - (float)hi);
    return {hi, lo};
}

// power of two that brings the largest |w| of a GEMM into (2^13, 2^14]
float weight_scale(int rows, int k_total, const Elem& w) {
    float m = 0.0f;
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < k_total; ++k) m = std::fmax(m, std::fabs(w(r, k)));
    if (!(m > 0.0f) || !std::isfinite(m)) return 1.0f;
    int e;
    std::frexp(m, &e);                      // m = f * 2^e, f in [0.5, 1)
    return std::ldexp(1.0f, 14 - e);        // m * scale in [2^13, 2^14)
}

// wide GEMM, A-operand fragments of v_mfma_f32_32x32x16_f16: [wave][kb16][rb][hi|lo][lane][8 halfs]
float pack_wide_f16(float* dst_f, int n_out, int k_total, const Elem& w, float forced_scale = 0.0f) {
    _Float16* dst = reinterpret_cast<_Float16*>(dst_f);
    if (g_map) map_new_group();
    const float sc = g_map ? 1.0f : (forced_scale > 0.0f ? forced_scale : weight_scale(n_out, k_total, w));
    const int rb_per_wave = n_out / (32 * inerf::kWaves), kb_count = k_total / 16;
    for (int wave = 0; wave < inerf::kWaves; ++wave)
        for (int kb = 0; kb < kb_count; ++kb)
            for (int rb = 0; rb < rb_per_wave; ++rb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int c = 0; c < 8; ++c) {
                        const int row = wave * 32 * rb_per_wave + 32 * rb + (lane & 31);
                        const int kv = 16 * kb + 8 * (lane >> 5) + c;
                        const int64_t frag = (((int64_t)wave * kb_count + kb) * rb_per_wave + rb) * 2;
                        if (g_map) { map_pair(dst + (frag * 64 + lane) * 8 + c, dst + ((frag + 1) * 64 + lane) * 8 + c, w(row, kv)); continue; }
                        const HalfPair h = split_f16(w(row, kv) * sc);
                        dst[(frag * 64 + lane) * 8 + c] = h.hi;
                        dst[((frag + 1) * 64 + lane) * 8 + c] = h.lo;
                    }
    return sc;
}

// skinny GEMM, A-operand fragments of v_mfma_f32_16x16x32_f16: [rb][kb32][hi|lo][lane][8 halfs]
// forced_scale > 0: a second copy of a matrix packed just before - its scale and, in map mode, its scale group
float pack_skinny_f16(float* dst_f, int rbs, int k_total, const Elem& w, float forced_scale = 0.0f) {
    _Float16* dst = reinterpret_cast<_Float16*>(dst_f);
    if (g_map && !(forced_scale > 0.0f)) map_new_group();
    const float sc = g_map ? 1.0f : (forced_scale > 0.0f ? forced_scale : weight_scale(16 * rbs, k_total, w));
    const int kb_count = k_total / 32;
    for (int rb = 0; rb < rbs; ++rb)
        for (int kb = 0; kb < kb_count; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int c = 0; c < 8; ++c) {
                    const int row = 16 * rb + (lane & 15);
                    const int kv = 32 * kb + 8 * (lane >> 4) + c;
                    const int64_t frag = ((int64_t)rb * kb_count + kb) * 2;
                    if (g_map) { map_pair(dst + (frag * 64 + lane) * 8 + c, dst + ((frag + 1) * 64 + lane) * 8 + c, w(row, kv)); continue; }
                    const HalfPair h = split_f16(w(row, kv) * sc);
                    dst[(frag * 64 + lane) * 8 + c] = h.hi;
                    dst[((frag + 1) * 64 + lane) * 8 + c] = h.lo;
                }
    return sc;
}

// register-operand copy of an output head (layout.h: as2r / resr): 32-row A fragments, k in accumulator order;
// same scale as the skinny copy of the same weights
// row0 / forced_scale: block `row0 / 32` of a head with more than 32 rows (sem2q), scale of the skinny copy given
void pack_regop_f16(float* dst_f, int q_per_wave, int rows, int k_total, const Elem& w, int row0 = 0, float forced_scale = 0.0f) {
    _Float16* dst = reinterpret_cast<_Float16*>(dst_f);
    // same matrix as the skinny copy packed just before: same scale, and in map mode the same scale group
    const float sc = g_map ? 1.0f : (forced_scale > 0.0f ? forced_scale : weight_scale(rows, k_total, w));
    for (int wave = 0; wave < inerf::kWaves; ++wave)
        for (int q = 0; q < q_per_wave; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int c = 0; c < 8; ++c) {
                    const int row = row0 + (lane & 31);
                    const int kv = wave * 16 * q_per_wave + inerf::regop_chan(q, lane >> 5, c);
                    const int64_t frag = ((int64_t)wave * q_per_wave + q) * 2;
                    if (g_map) {
                        map_pair(dst + (frag * 64 + lane) * 8 + c, dst + ((frag + 1) * 64 + lane) * 8 + c, row < rows ? w(row, kv) : 0.0f);
                        continue;
                    }
                    const HalfPair h = split_f16(row < rows ? w(row, kv) * sc : 0.0f);
                    dst[(frag * 64 + lane) * 8 + c] = h.hi;
                    dst[((frag + 1) * 64 + lane) * 8 + c] = h.lo;
                }
}

// register-operand copy of a skinny head whose input is a 16x16x32 accumulator set (layout.h: sem2r); scale (and scale
// group) of the skinny copy packed just before
void pack_regop16_f16(float* dst_f, int rbs, int k_total, const Elem& w, float scale) {
    _Float16* dst = reinterpret_cast<_Float16*>(dst_f);
    const float sc = g_map ? 1.0f : scale;
    const int kb_count = k_total / 32;
    for (int rb = 0; rb < rbs; ++rb)
        for (int kb = 0; kb < kb_count; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int c = 0; c < 8; ++c) {
                    const int row = 16 * rb + (lane & 15);
                    const int kv = 32 * kb + inerf::regop16_chan(lane >> 4, c);
                    const int64_t frag = ((int64_t)rb * kb_count + kb) * 2;
                    if (g_map) { map_pair(dst + (frag * 64 + lane) * 8 + c, dst + ((frag + 1) * 64 + lane) * 8 + c, w(row, kv)); continue; }
                    const HalfPair h = split_f16(w(row, kv) * sc);
                    dst[(frag * 64 + lane) * 8 + c] = h.hi;
                    dst[((frag + 1) * 64 + lane) * 8 + c] = h.lo;
                }
}

}  // namespace

extern "C" {

const char* inerf_version(void) { return "inerf 0.2 (gfx950)"; }
int inerf_abi_version(void) { return INERF_ABI_VERSION; }

int inerf_num_tensors(const inerf_net_desc* net) {
    if (!net || !inerf::net_supported(*net)) return INERF_E_INVALID;
    return (int)describe(*net).spec.size();
}

int inerf_tensor_info(const inerf_net_desc* net, int index, const char** name, int64_t* rows, int64_t* cols) {
    if (!net || !inerf::net_supported(*net)) return INERF_E_INVALID;
    // names are kept alive for the life of the process (one small table per distinct description)
    static thread_local std::vector<TensorSpec> table;
    table = describe(*net).spec;
    if (index < 0 || index >= (int)table.size()) return INERF_E_INVALID;
    if (name) *name = table[index].name.c_str();
    if (rows) *rows = table[index].rows;
    if (cols) *cols = table[index].cols;
    return INERF_OK;
}

int64_t inerf_packed_floats(const inerf_net_desc* net) {
    if (!net || !inerf::net_supported(*net)) return INERF_E_INVALID;
    return inerf::make_layout(*net).total_floats;
}

int inerf_raw_channels(const inerf_net_desc* net, uint32_t flags, int fine) {
    if (!net || !inerf::net_supported(*net)) return INERF_E_INVALID;
    int ch = INERF_BASE_CHANNELS + (net->variant == INERF_VARIANT_SSR ? net->n_classes : 0);
    if (fine && net->variant == INERF_VARIANT_SSR && (flags & INERF_FLAG_ENDPOINT)) ch += INERF_ENDPOINT_DIM;
    return ch;
}

int64_t inerf_bwd_packed_floats(const inerf_net_desc* net) {
    if (!net || !inerf::net_supported(*net)) return INERF_E_INVALID;
    return inerf::make_bwd_layout(*net).total_floats;
}

// Transposed layers for the input-gradient chain (layout.h BwdLayout); always the f16 hi/lo fragment format.
int inerf_pack_weights_bwd(const inerf_net_desc* net, const float* const* tensors, int n_tensors, float* out,
                           int64_t capacity) {
    using namespace inerf;
    if (!net || !tensors || !out || !net_supported(*net)) return INERF_E_INVALID;
    const Net n = describe(*net);
    if (n_tensors != (int)n.spec.size()) return INERF_E_INVALID;
    for (int i = 0; i < n_tensors; ++i)
        if (!tensors[i]) return INERF_E_INVALID;
    const BwdLayout L = make_bwd_layout(*net);
    if (capacity < L.total_floats) return INERF_E_INVALID;
    std::memset(out, 0, sizeof(float) * (size_t)L.total_floats);
    auto find = [&](const std::string& key) -> int {
        for (int i = 0; i < n_tensors; ++i)
            if (n.spec[i].name == key) return i;
        return -1;
    };
    auto W = [&](const std::string& lin) { return tensors[find(lin + ".weight")]; };
    auto put = [&](const GemmSlot& s, int n_out, int k, const Elem& el, float forced = 0.0f) {
        const float sc = pack_wide_f16(out + s.w, n_out, k, el, forced);
        if (g_map) g_map->consts.push_back({s.b, 0, g_map->group, 1, 1.0f});
        else out[s.b] = 1.0f / sc;
    };
    const bool obj = net->variant == INERF_VARIANT_OBJECT;
    const std::string sh1 = obj ? "test_linear1" : "shading_linear1";
    const std::string sh2 = obj ? "test_linear2" : "shading_linear2";
    const std::string rs = obj ? "shading_linear" : "residual_linear";
    const int e = n.e, dv = n.dv;
    // views^T (feature columns only): row = feature channel, k = views output
    const float* wv = W("views_linears.0");
    const int vin = kWidth + dv;
    put(L.views_t, kWidth, kHalf, [=](int r, int k) { return wv[(int64_t)k * vin + r]; });
    // the three matrices summed into d h7: one common scale
    const float* wf = W("feature_linear");
    const float* wa = W("albedo_linear1");
    const float* ws = W(sh1);
    const float* w1 = L.has_sem ? W("semantic_linear.0.0") : nullptr;
    const Elem feat_t = [=](int r, int k) { return wf[(int64_t)k * kWidth + r]; };
    const Elem as1_t = [=](int r, int k) { return k < kHalf ? wa[(int64_t)k * kWidth + r] : ws[(int64_t)(k - kHalf) * kWidth + r]; };
    const Elem sem1_t = [=](int r, int k) { return w1[(int64_t)k * kWidth + r]; };
    float common = std::fmin(weight_scale(kWidth, kWidth, feat_t), weight_scale(kWidth, kWidth, as1_t));
    if (L.has_sem) common = std::fmin(common, weight_scale(kWidth, kHalf, sem1_t));
    if (g_map) g_map->pinned = g_map->next_group++;
    put(L.feat_t, kWidth, kWidth, feat_t, common);
    put(L.as1_t, kWidth, kWidth, as1_t, common);
    if (L.has_sem) put(L.sem1_t, kWidth, kHalf, sem1_t, common);
    if (g_map) g_map->pinned = -1;
    for (int i = 1; i < kDepth; ++i) {
        const std::string lin = "pts_linears." + std::to_string(i);
        const float* w = W(lin);
        const int in = (int)n.spec[find(lin + ".weight")].cols;
        const int skip = i == kSkipInput ? e : 0;          // cat([pts, h]): only the h columns carry a gradient onward
        put(L.trunk_t[i], kWidth, kWidth, [=](int r, int k) { return w[(int64_t)k * in + skip + r]; });
    }
    const float* wr = W(rs);
    for (int c = 0; c < kHalf; ++c)
        for (int j = 0; j < 3; ++j) out[L.res_w + 4 * c + j] = wr[(int64_t)j * kHalf + c];
    const float* wa2 = W("albedo_linear2");
    const float* ws2 = W(sh2);
    for (int c = 0; c < kHalf; ++c) {
        for (int j = 0; j < 3; ++j) out[L.as2_w + 4 * c + j] = wa2[(int64_t)j * kHalf + c];
        out[L.as2_w + 4 * (kHalf + c) + 3] = ws2[c];
    }
    std::memcpy(out + L.alpha_w, W("alpha_linear"), sizeof(float) * kWidth);
    if (L.has_sem) std::memcpy(out + L.sem2_w, W("semantic_linear.1"), sizeof(float) * (size_t)net->n_classes * kHalf);
    return INERF_OK;
}

int inerf_pack_weights(const inerf_net_desc* net, const float* const* tensors, int n_tensors, float* out,
                       int64_t capacity) {
    using namespace inerf;
    if (!net || !tensors || !out || !net_supported(*net)) return INERF_E_INVALID;
    const Net n = describe(*net);
    if (n_tensors != (int)n.spec.size()) return INERF_E_INVALID;
    for (int i = 0; i < n_tensors; ++i)
        if (!tensors[i]) return INERF_E_INVALID;
    const NetLayout L = make_layout(*net);
    if (capacity < L.total_floats) return INERF_E_INVALID;
    std::memset(out, 0, sizeof(float) * (size_t)L.total_floats);

    const bool f16 = net->precision == INERF_PREC_F16X3;
    float last_scale = 1.0f;            // weight scale of the most recent pack_* call
    auto pack_wide = [&](float* dst, int n_out, int k, const Elem& el) {
        last_scale = f16 ? pack_wide_f16(dst, n_out, k, el) : pack_wide_f32(dst, n_out, k, el);
    };
    auto pack_skinny = [&](float* dst, int rbs, int k, const Elem& el) {
        last_scale = f16 ? pack_skinny_f16(dst, rbs, k, el) : pack_skinny_f32(dst, rbs, k, el);
    };
    // after the biases of a slot are in place: store the accumulator->output factor behind them and, for the
    // wide (hidden) layers of the f16 format, move the biases into the scaled activation domain (layout.h)
    auto finish_wide = [&](const GemmSlot& s, int n_out) {
        if (g_map) {
            g_map->consts.push_back({s.b + n_out, 0, g_map->group, 1, 1.0f});
            g_map->times8.push_back({s.b, s.b + n_out});
            return;
        }
        out[s.b + n_out] = 1.0f / last_scale;
        if (f16) for (int i = 0; i < n_out; ++i) out[s.b + i] *= kActScale;
    };
    auto finish_skinny = [&](const GemmSlot& s, int rbs) {
        if (g_map) { g_map->consts.push_back({s.b + 16 * rbs, 0, g_map->group, 2, 1.0f}); return; }
        out[s.b + 16 * rbs] = f16 ? 1.0f / (last_scale * kActScale) : 1.0f;
    };
    auto find = [&](const char* key) -> int {
        for (int i = 0; i < n_tensors; ++i)
            if (n.spec[i].name == key) return i;
        return -1;
    };
    auto W = [&](const std::string& lin) { return tensors[find((lin + ".weight").c_str())]; };
    auto B = [&](const std::string& lin) { return tensors[find((lin + ".bias").c_str())]; };
    auto copy_bias = [&](int32_t off, const float* b, int count) { std::memcpy(out + off, b, sizeof(float) * count); };
    const int e = n.e, dv = n.dv;

    // ---- trunk ----
    for (int i = 0; i < kDepth; ++i) {
        const std::string lin = "pts_linears." + std::to_string(i);
        const float* w = W(lin);
        const int in = (int)n.spec[find((lin + ".weight").c_str())].cols;
        Elem el;
        if (i == 0)                       // virtual k = encoded xyz column (64 wide, zero padded)
            el = [=](int r, int kv) { return kv < e ? w[(int64_t)r * in + kv] : 0.0f; };
        else if (i == kSkipInput)         // virtual k = [enc64 | h256]; reference order is cat([pts, h])
            el = [=](int r, int kv) {
                if (kv < kEncCols) return kv < e ? w[(int64_t)r * in + kv] : 0.0f;
                return w[(int64_t)r * in + e + (kv - kEncCols)];
            };
        else
            el = [=](int r, int kv) { return w[(int64_t)r * in + kv]; };
        pack_wide(out + L.trunk[i].w, kWidth, trunk_k(i), el);
        copy_bias(L.trunk[i].b, B(lin), kWidth);
        finish_wide(L.trunk[i], kWidth);
    }
    // ---- sigma ----
    {
        const float* w = W("alpha_linear");
        pack_skinny(out + L.alpha.w, 1, kWidth, [=](int r, int kv) { return r == 0 ? w[kv] : 0.0f; });
        out[L.alpha.b] = B("alpha_linear")[0];
        finish_skinny(L.alpha, 1);
    }
    // ---- semantic head (ssr) ----
    if (L.sem_rbs > 0) {
        const float* w1 = W("semantic_linear.0.0");
        const Elem sem1 = [=](int r, int kv) { return w1[(int64_t)r * kWidth + kv]; };
        pack_wide(out + L.sem1.w, kHalf, kWidth, sem1);
        if (f16) pack_skinny_f16(out + L.sem1s.w, kHalf / 16, kWidth, sem1, g_map ? 1.0f : last_scale);
        copy_bias(L.sem1.b, B("semantic_linear.0.0"), kHalf);
        finish_wide(L.sem1, kHalf);
        const float* w2 = W("semantic_linear.1");
        const int c = net->n_classes;
        const Elem sem2 = [=](int r, int kv) { return r < c ? w2[(int64_t)r * kHalf + kv] : 0.0f; };
        pack_skinny(out + L.sem2.w, L.sem_rbs, kHalf, sem2);
        if (f16) pack_regop16_f16(out + L.sem2r.w, L.sem_rbs, kHalf, sem2, last_scale);
        if (f16)
            for (int rb = 0; rb < L.sem_rb32; ++rb) pack_regop_f16(out + L.sem2q.w + rb * 32 * kHalf, 2, c, kHalf, sem2, 32 * rb, last_scale);
        copy_bias(L.sem2.b, B("semantic_linear.1"), c);
        finish_skinny(L.sem2, L.sem_rbs);
    }
    // ---- albedo + shading hidden layers fused into one 256-row GEMM, then one skinny output GEMM ----
    const bool obj = net->variant == INERF_VARIANT_OBJECT;
    const std::string sh1 = obj ? "test_linear1" : "shading_linear1";
    const std::string sh2 = obj ? "test_linear2" : "shading_linear2";
    const std::string rs = obj ? "shading_linear" : "residual_linear";
    {
        const float* wa = W("albedo_linear1");
        const float* ws = W(sh1);
        pack_wide(out + L.as1.w, kWidth, kWidth, [=](int r, int kv) {
            return r < kHalf ? wa[(int64_t)r * kWidth + kv] : ws[(int64_t)(r - kHalf) * kWidth + kv];
        });
        copy_bias(L.as1.b, B("albedo_linear1"), kHalf);
        copy_bias(L.as1.b + kHalf, B(sh1), kHalf);
        finish_wide(L.as1, kWidth);
        const float* wa2 = W("albedo_linear2");
        const float* ws2 = W(sh2);
        const Elem as2 = [=](int r, int kv) {
            if (r < 3) return kv < kHalf ? wa2[(int64_t)r * kHalf + kv] : 0.0f;
            if (r == 3) return kv >= kHalf ? ws2[kv - kHalf] : 0.0f;
            return 0.0f;
        };
        pack_skinny(out + L.as2.w, 1, kWidth, as2);
        if (f16) pack_regop_f16(out + L.as2r.w, 4, 16, kWidth, as2);
        copy_bias(L.as2.b, B("albedo_linear2"), 3);
        out[L.as2.b + 3] = B(sh2)[0];
        finish_skinny(L.as2, 1);
    }
    // ---- feature, views, residual ----
    {
        const float* wf = W("feature_linear");
        pack_wide(out + L.feat.w, kWidth, kWidth, [=](int r, int kv) { return wf[(int64_t)r * kWidth + kv]; });
        copy_bias(L.feat.b, B("feature_linear"), kWidth);
        finish_wide(L.feat, kWidth);
        const float* wv = W("views_linears.0");
        const int in = kWidth + dv;
        pack_wide(out + L.views.w, kHalf, kWidth + kDirCols, [=](int r, int kv) {
            if (kv < kWidth) return wv[(int64_t)r * in + kv];
            const int dc = kv - kWidth;
            return dc < dv ? wv[(int64_t)r * in + kWidth + dc] : 0.0f;
        });
        copy_bias(L.views.b, B("views_linears.0"), kHalf);
        finish_wide(L.views, kHalf);
        const float* wr = W(rs);
        const Elem res = [=](int r, int kv) { return r < 3 ? wr[(int64_t)r * kHalf + kv] : 0.0f; };
        pack_skinny(out + L.res.w, 1, kHalf, res);
        if (f16) pack_regop_f16(out + L.resr.w, 2, 16, kHalf, res);
        copy_bias(L.res.b, B(rs), 3);
        finish_skinny(L.res, 1);
    }
    return INERF_OK;
}

// Where every element of a packed blob comes from (see "map mode" above).  backward = 0: inerf_pack_weights blob in the
// INERF_PREC_F16X3 format; 1: inerf_pack_weights_bwd blob.  half_src / half_grp: 2 * packed floats entries each.
// The constants (biases, plain fp32 weights, scale factors) come back as up to const_capacity entries; returns their
// number, or a negative INERF_E_*.
int64_t inerf_pack_map(const inerf_net_desc* net, int backward, int32_t* half_src, int32_t* half_grp, int64_t half_capacity,
                       int32_t* c_dst, int32_t* c_src, int32_t* c_group, int32_t* c_code, float* c_mult, int64_t const_capacity,
                       int32_t* n_groups) {
    using namespace inerf;
    if (!net || !half_src || !half_grp || !c_dst || !c_src || !c_group || !c_code || !c_mult || !n_groups) return INERF_E_INVALID;
    inerf_net_desc d = *net;
    d.precision = INERF_PREC_F16X3;
    if (!net_supported(d)) return INERF_E_INVALID;
    const int64_t total = backward ? make_bwd_layout(d).total_floats : make_layout(d).total_floats;
    if (half_capacity < 2 * total) return INERF_E_INVALID;
    const Net n = describe(d);
    std::vector<std::vector<float>> index_tensors;
    std::vector<const float*> ptrs;
    int64_t flat = 0;
    for (const TensorSpec& t : n.spec) {
        const int64_t count = t.rows * (t.cols ? t.cols : 1);
        std::vector<float> v((size_t)count);
        for (int64_t j = 0; j < count; ++j) v[(size_t)j] = (float)(flat + j + 1);       // < 2^24: exact
        flat += count;
        index_tensors.push_back(std::move(v));
    }
    for (const auto& v : index_tensors) ptrs.push_back(v.data());
    std::vector<float> scratch((size_t)total, 0.0f);
    std::memset(half_src, 0, sizeof(int32_t) * 2 * (size_t)total);
    std::memset(half_grp, 0, sizeof(int32_t) * 2 * (size_t)total);
    MapSink sink;
    sink.base = scratch.data();
    sink.half_src = half_src;
    sink.half_grp = half_grp;
    sink.is_weight.assign((size_t)total, 0);
    g_map = &sink;
    const int rc = backward ? inerf_pack_weights_bwd(&d, ptrs.data(), (int)ptrs.size(), scratch.data(), total)
                            : inerf_pack_weights(&d, ptrs.data(), (int)ptrs.size(), scratch.data(), total);
    g_map = nullptr;
    if (rc != INERF_OK) return rc;
    std::vector<char> is_const((size_t)total, 0);
    for (const FloatEntry& e : sink.consts) is_const[(size_t)e.dst] = 1;
    std::vector<FloatEntry> all = sink.consts;
    for (int64_t f = 0; f < total; ++f) {
        if (sink.is_weight[(size_t)f] || is_const[(size_t)f] || scratch[(size_t)f] == 0.0f) continue;
        float mult = 1.0f;
        for (const auto& r : sink.times8)
            if (f >= r.first && f < r.second) mult = kActScale;
        all.push_back({(int32_t)f, (int32_t)scratch[(size_t)f], 0, 0, mult});
    }
    if ((int64_t)all.size() > const_capacity) return INERF_E_INVALID;
    for (size_t i = 0; i < all.size(); ++i) {
        c_dst[i] = all[i].dst; c_src[i] = all[i].src; c_group[i] = all[i].group; c_code[i] = all[i].code; c_mult[i] = all[i].mult;
    }
    *n_groups = sink.next_group;
    return (int64_t)all.size();
}

}  // extern "C"
