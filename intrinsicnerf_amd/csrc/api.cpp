// api.cpp - whole-path entry point of the C ABI: enqueues the stage kernels for one ray batch.
// Replaces render_rays (object_level/run_nerf.py:415-528) / SSRTrainer.volumetric_rendering
// (SSR/training/trainer.py:717-808).  No host synchronisation, no allocation: all scratch lives in
// the caller's workspace.
#include <cstdint>

#include "layout.h"

namespace {

constexpr int64_t kAlign = 256;
inline int64_t up(int64_t b) { return (b + kAlign - 1) / kAlign * kAlign; }

struct Plan {
    int64_t z_c, w_c, raw_c, z_s, z_f, raw_f, mlp_ws, mlp_ws_bytes, total;
    int ch_c, ch_f, s_f;
};

// which intermediate tensors the caller supplies as outputs (inerf_render_args): no workspace is reserved for those.
// raw is by far the largest (N x S x CH floats: ~6 GB for 32768 rays x 192 samples x 240 channels), and the SSR
// front-end asks for raw_coarse / raw_fine by default.
enum : unsigned { kHaveZc = 1, kHaveWc = 2, kHaveRawC = 4, kHaveZs = 8, kHaveZf = 16, kHaveRawF = 32 };

unsigned provided(const inerf_render_args& a) {
    return (a.z_coarse ? kHaveZc : 0u) | (a.coarse.weights ? kHaveWc : 0u) | (a.raw_coarse ? kHaveRawC : 0u) |
           (a.z_samples ? kHaveZs : 0u) | (a.z_fine ? kHaveZf : 0u) | (a.raw_fine ? kHaveRawF : 0u);
}

Plan plan(const inerf_net_desc& net, int64_t n, int sc, int ni, uint32_t flags, unsigned have = 0) {
    Plan p{};
    p.ch_c = inerf_raw_channels(&net, flags, 0);
    p.ch_f = inerf_raw_channels(&net, flags, 1);
    p.s_f = sc + ni;
    int64_t off = 0;
    auto take = [&](int64_t floats, unsigned bit) {
        if (have & bit) return (int64_t)-1;
        int64_t o = off;
        off += up(floats * 4);
        return o;
    };
    p.z_c = take(n * sc, kHaveZc);
    p.w_c = ni > 0 ? take(n * sc, kHaveWc) : (int64_t)-1;   // the coarse weights are only needed for resampling
    p.raw_c = take(n * sc * p.ch_c, kHaveRawC);
    if (ni > 0) {
        p.z_s = take(n * ni, kHaveZs);
        p.z_f = take(n * p.s_f, kHaveZf);
        p.raw_f = take(n * p.s_f * p.ch_f, kHaveRawF);
    }
    // scratch of the encode+MLP launches (the SSR network's channel-split semantic head); one region serves both passes
    const int64_t ws_c = inerf_encode_mlp_workspace_bytes(&net, n, sc, flags & ~INERF_FLAG_ENDPOINT);
    const int64_t ws_f = ni > 0 ? inerf_encode_mlp_workspace_bytes(&net, n, p.s_f, flags) : 0;
    p.mlp_ws_bytes = ws_c > ws_f ? ws_c : ws_f;
    p.mlp_ws = off;
    off += up(p.mlp_ws_bytes);
    p.total = off;
    return p;
}

}  // namespace

extern "C" int64_t inerf_workspace_bytes(const inerf_net_desc* net, int64_t n_rays, int n_samples, int n_importance,
                                         uint32_t flags) {
    if (!net || !inerf::net_supported(*net) || n_rays < 0 || n_samples < 1 || n_importance < 0) return INERF_E_INVALID;
    return plan(*net, n_rays, n_samples, n_importance, flags).total;
}

extern "C" int64_t inerf_render_workspace_bytes(const inerf_render_args* a) {
    if (!a || !inerf::net_supported(a->net) || a->n_rays < 0 || a->n_samples < 1 || a->n_importance < 0) return INERF_E_INVALID;
    return plan(a->net, a->n_rays, a->n_samples, a->n_importance, a->flags, provided(*a)).total;
}

extern "C" int inerf_render_rays(const inerf_render_args* a, void* stream) {
    if (a && a->n_rays == 0) return inerf::net_supported(a->net) ? INERF_OK : INERF_E_UNSUPPORTED;   // empty batch: null pointers allowed
    if (!a || !a->packed_coarse || !a->rays || !a->t_vals || a->n_rays < 0 || a->n_samples < 1 || a->n_importance < 0)
        return INERF_E_INVALID;
    if (!inerf::net_supported(a->net)) return INERF_E_UNSUPPORTED;
    if (a->n_importance > 0 && !a->u) return INERF_E_INVALID;
    const Plan p = plan(a->net, a->n_rays, a->n_samples, a->n_importance, a->flags, provided(*a));
    if (p.total > 0 && (!a->workspace || a->workspace_bytes < p.total)) return INERF_E_WORKSPACE;
    char* ws = static_cast<char*>(a->workspace);
    auto f = [&](int64_t off) { return reinterpret_cast<float*>(ws + off); };
    const bool ssr = a->net.variant == INERF_VARIANT_SSR;
    const int n_cls = ssr ? a->net.n_classes : 0;
    const int64_t n = a->n_rays;
    const int sc = a->n_samples, ni = a->n_importance;
    int rc;

    // ---- coarse pass ----
    float* z_c = a->z_coarse ? a->z_coarse : f(p.z_c);
    rc = inerf_sample_coarse(a->rays, a->t_vals, a->t_rand, n, sc, a->flags, z_c, stream);
    if (rc) return rc;
    float* raw_c = a->raw_coarse ? a->raw_coarse : f(p.raw_c);
    // the coarse net never emits the endpoint feature (trainer.py:751-755: endpoint_feat=False)
    rc = inerf_encode_mlp_chunked(&a->net, a->packed_coarse, a->rays, z_c, n, sc, a->flags & ~INERF_FLAG_ENDPOINT, raw_c, a->status, a->status_rays,
                             ws + p.mlp_ws, p.mlp_ws_bytes, stream);
    if (rc) return rc;
    inerf_composite_out oc = a->coarse;
    oc.feat = nullptr;
    if (ni > 0 && !oc.weights) oc.weights = f(p.w_c);
    rc = inerf_composite(raw_c, z_c, a->rays + 3, INERF_RAY_FLOATS, a->noise_coarse, n, sc, p.ch_c, oc.sem ? n_cls : 0, 0,
                         a->flags, &oc, stream);
    if (rc || ni == 0) return rc;

    // ---- resample, fine pass ----
    float* z_s = a->z_samples ? a->z_samples : f(p.z_s);
    float* z_f = a->z_fine ? a->z_fine : f(p.z_f);
    rc = inerf_sample_fine(z_c, oc.weights, a->u, n, sc, ni, a->flags, z_s, z_f, a->z_std, stream);
    if (rc) return rc;
    float* raw_f = a->raw_fine ? a->raw_fine : f(p.raw_f);
    const float* w_fine = a->packed_fine ? a->packed_fine : a->packed_coarse;   // run_nerf.py:506
    rc = inerf_encode_mlp_chunked(&a->net, w_fine, a->rays, z_f, n, p.s_f, a->flags, raw_f, a->status, a->status_rays, ws + p.mlp_ws, p.mlp_ws_bytes, stream);
    if (rc) return rc;
    const bool ep = ssr && (a->flags & INERF_FLAG_ENDPOINT);
    inerf_composite_out of = a->fine;
    if (!ep) of.feat = nullptr;
    return inerf_composite(raw_f, z_f, a->rays + 3, INERF_RAY_FLOATS, a->noise_fine, n, p.s_f, p.ch_f, of.sem ? n_cls : 0,
                           (ep && of.feat) ? INERF_ENDPOINT_DIM : 0, a->flags, &of, stream);
}
