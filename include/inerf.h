/*
 * inerf.h - C ABI of libinerf.so: the MI355X (gfx950) implementation of IntrinsicNeRF's volumetric
 * render_rays hot path.
 *
 * The reference (zju3dv/IntrinsicNeRF) is pure Python on PyTorch and has no FFI of its own: the
 * "operator interface" of this path is the set of Python functions listed below.  Each entry point
 * of this header replaces the ATen op sequence of one of them and is what a ctypes / cffi / pybind
 * binding on the reference side would bind (INTEGRATION.md shows the ctypes stub).  Citations are
 * relative to the reference repository root.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless the parameter is marked [host];
 *   - the library never allocates device memory: outputs and scratch are caller-owned;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: 0 = success, negative = INERF_E_* (never throws, never aborts);
 *   - thread-safe as long as concurrent calls use distinct streams and distinct output/workspace
 *     buffers; packed weights are read-only and may be shared.
 */
#ifndef INERF_H
#define INERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INERF_VERSION_MAJOR 0
#define INERF_VERSION_MINOR 2
/* Bumped whenever a struct layout, an argument list or the packed-weight format of this header changes; bindings
 * compare it with inerf_abi_version() of the library they loaded (a stale .so then fails loudly, not silently). */
#define INERF_ABI_VERSION 40008

/* error codes */
#define INERF_OK              0
#define INERF_E_INVALID      -1   /* bad argument (null pointer, non-positive size, unknown variant ...) */
#define INERF_E_UNSUPPORTED  -2   /* configuration outside what the kernels implement (see each call)    */
#define INERF_E_WORKSPACE    -3   /* workspace pointer null or too small                                  */
#define INERF_E_HIP          -4   /* a HIP runtime call failed; inerf_last_hip_error() has the code       */

/* network variants */
#define INERF_VARIANT_OBJECT  0   /* object_level/run_nerf_helpers.py:247 NeRF (11 raw channels)          */
#define INERF_VARIANT_SSR     1   /* SSR/models/semantic_nerf.py:74 Semantic_NeRF (11 + C [+128] channels) */

/* render flags */
#define INERF_FLAG_WHITE_BKGD   1u   /* run_nerf.py:407-410 / model_utils.py:109-114                      */
#define INERF_FLAG_LINDISP      2u   /* run_nerf.py:467-468 (object-level only)                           */
#define INERF_FLAG_ENDPOINT     4u   /* SSR endpoint_feat: fine raw carries the 128-d views activation    */
#define INERF_FLAG_U_PER_RAY    8u   /* `u` is [N, n_importance] (random) instead of a shared [n_importance] */

/* arithmetic of the MLP GEMMs (inerf_net_desc.precision) */
#define INERF_PREC_F32        0   /* v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation             */
#define INERF_PREC_F16X3      1   /* every fp32 operand split into f16 hi + f16 (lo * 2^11); three f16 MFMA
                                     products hi*hi + 2^-11 (hi*lo + lo*hi), fp32 accumulation: 22-bit operands,
                                     fp32-level accuracy at the 16x faster f16 matrix pipe.  Activations above
                                     6e4 in magnitude cannot be represented: the kernel then raises
                                     INERF_STATUS_F16_RANGE in the caller's status word (re-run with F32).     */

/* bits of the device status word (inerf_encode_mlp / inerf_render_rays `status` argument) */
#define INERF_STATUS_F16_RANGE  1

#define INERF_BASE_CHANNELS   11   /* rgb3 sigma albedo3 shading residual3 (run_nerf_helpers.py:321)       */
#define INERF_ENDPOINT_DIM    128
#define INERF_RAY_FLOATS      11   /* o3 d3 near far viewdir3 (run_nerf.py:122-128, rays.py:251-255)       */
#define INERF_MAX_CLASSES     240  /* semantic classes supported by the packed layout                     */

const char* inerf_version(void);
int inerf_abi_version(void);          /* INERF_ABI_VERSION the library was built against */
const char* inerf_build_digest(void); /* sha256 (64 hex digits) of the sources, headers and compiler flags this library was built from */
int inerf_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------
 * Network description and weight packing.
 * Replaces: the nn.Module parameter storage of NeRF (run_nerf_helpers.py:259-279) and Semantic_NeRF
 * (semantic_nerf.py:98-118) as seen by forward().  Fixed architecture D=8, W=256, skips=[4],
 * use_viewdirs=True - every shipped config (run_nerf.py:545-552, every SSR/configs yaml).
 * ------------------------------------------------------------------------------------------- */
typedef struct inerf_net_desc {
    int32_t variant;      /* INERF_VARIANT_*                                                       */
    int32_t n_classes;    /* SSR: semantic classes C (0 = semantic head absent); object: must be 0 */
    int32_t l_xyz;        /* multires       (0..10)  -> 3+6*l_xyz encoded position channels        */
    int32_t l_dir;        /* multires_views (0..4)   -> 3+6*l_dir encoded direction channels       */
    float   xyz_div;      /* encoder input divisor: 1 (object) / 10 (SSR, semantic_nerf.py:64)     */
    int32_t precision;    /* INERF_PREC_*: selects the packed format AND the MLP kernel            */
} inerf_net_desc;

/* Number of state-dict tensors the packer expects, and the canonical order/shape of tensor i:
 * pts_linears.{0..7}.{weight,bias}, views_linears.0, feature_linear, alpha_linear, then
 *   object: shading_linear (=residual head), albedo_linear1, albedo_linear2, test_linear1, test_linear2
 *   ssr   : [semantic_linear.0.0, semantic_linear.1,] residual_linear, albedo_linear1, albedo_linear2,
 *           shading_linear1, shading_linear2
 * (weight then bias for each).  inerf_tensor_info lets a binding verify a state dict; the returned
 * name pointer stays valid until the next inerf_tensor_info call on the same thread. */
int inerf_num_tensors(const inerf_net_desc* net);
int inerf_tensor_info(const inerf_net_desc* net, int index, const char** name /*[host] out*/,
                      int64_t* rows /*out*/, int64_t* cols /*out; 0 for a bias*/);

/* Size in floats of the packed blob for this network. */
int64_t inerf_packed_floats(const inerf_net_desc* net);

/* Pack the [host] fp32 state-dict tensors (row-major [out,in], canonical order above) into the
 * MFMA-fragment-ordered blob the kernels stream ([host] packed_out, inerf_packed_floats() floats).
 * The caller uploads the blob to the device once per weight update. */
int inerf_pack_weights(const inerf_net_desc* net, const float* const* tensors /*[host]*/, int n_tensors,
                       float* packed_out /*[host]*/, int64_t packed_capacity_floats);

/* ---------------------------------------------------------------------------------------------
 * Stage kernels (each is also a parity-test boundary).
 * ------------------------------------------------------------------------------------------- */

/* z_vals[N,S]: stratified depths.  Replaces run_nerf.py:464-486 / trainer.py:730-746.
 * rays[N,11]; t_vals[S] = linspace(0,1,S) supplied by the caller (so its rounding is the host
 * framework's); t_rand[N,S] or NULL (perturb == 0); lindisp via flags. */
int inerf_sample_coarse(const float* rays, const float* t_vals, const float* t_rand, int64_t n_rays, int n_samples,
                        uint32_t flags, float* z_out, void* stream);

/* raw[N,S,CH] = MLP(encode(o + d*z), encode(viewdir)).  Replaces run_network + NeRF.forward:
 * run_nerf.py:42-56 + run_nerf_helpers.py:195-243,284-321 / model_utils.py:19-35 +
 * semantic_nerf.py:14-65,123-181.  CH = 11 + n_classes (+128 if INERF_FLAG_ENDPOINT).
 * packed_weights: device copy of the inerf_pack_weights() blob.
 * status: optional device int32 the kernel ORs INERF_STATUS_* bits into (caller zeroes it). */
int inerf_encode_mlp(const inerf_net_desc* net, const float* packed_weights, const float* rays, const float* z_vals,
                     int64_t n_rays, int n_samples, uint32_t flags, float* raw_out, int32_t* status, void* stream);

/* The same with a caller-owned scratch buffer of inerf_encode_mlp_workspace_bytes() bytes (0 for most configurations).  With
 * it the INERF_PREC_F16X3 kernel of the SSR network splits the semantic hidden layer (semantic_nerf.py:110,150-152) over the
 * waves by channel and parks the per-wave partial logits in the scratch until the tile's activations are dead - the weights of
 * semantic_linear.0.0 are then streamed once per 64-point tile instead of four times.  Same results as inerf_encode_mlp up to
 * the summation order of the logits.  inerf_render_rays uses this form (its workspace includes the scratch). */
int64_t inerf_encode_mlp_workspace_bytes(const inerf_net_desc* net, int64_t n_rays, int n_samples, uint32_t flags);
int inerf_encode_mlp_ws(const inerf_net_desc* net, const float* packed_weights, const float* rays, const float* z_vals,
                        int64_t n_rays, int n_samples, uint32_t flags, float* raw_out, int32_t* status, void* workspace,
                        int64_t workspace_bytes, void* stream);
/* The same with ONE STATUS WORD PER `status_rays` RAYS: `status` then points at ceil(n_rays / status_rays) device words (caller
 * zeroes them) and word w collects the INERF_STATUS_* bits of rays [w * status_rays, (w + 1) * status_rays).  The front-ends render
 * an eval-mode frame as one launch whatever `chunk` the caller of batchify_rays (run_nerf.py:59-71, training_utils.py:5-17) asked
 * for and still learn WHICH of the caller's chunks left the f16 range - only that one is rendered again in exact fp32.
 * status_rays <= 0: one word, as inerf_encode_mlp_ws. */
int inerf_encode_mlp_chunked(const inerf_net_desc* net, const float* packed_weights, const float* rays, const float* z_vals,
                             int64_t n_rays, int n_samples, uint32_t flags, float* raw_out, int32_t* status, int64_t status_rays,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training: what autograd records for the network when the trainers call loss.backward()
 * (run_nerf.py:1018 through run_nerf_helpers.py:284-321; trainer.py:990 through
 * semantic_nerf.py:123-181).  Split-precision (INERF_PREC_F16X3) path only.
 *   forward : inerf_encode_mlp_train  = inerf_encode_mlp + every layer's output kept in `save`
 *   backward: inerf_mlp_backward_inputs: d raw -> pre-activation gradient dZ of every layer
 *             (the input-gradient chain, MFMA); the weight gradients are then plain GEMMs over the
 *             sample points, dW_l = dZ_l^T X_l, db_l = column sums of dZ_l, left to the caller's
 *             GEMM library.
 * `save` and `dz` are buffers of inerf_mlp_save_floats() 4-byte elements laid out [slot][point][width], every slot sized for
 * WHOLE 64-point tiles (64 * ceil(n_points / 64) points; inerf_mlp_save_slot() gives offset, width and format):
 *   slot 0 enc 64 | 1 dir 32 | 2..9 h0..h7 256 | 10 albedo|shading hidden 256 | 11 feature 256 |
 *   12 views hidden 128 | 13 semantic hidden 128 (SSR with classes, else width 0) |
 *   14 (dz only) head pre-activation gradients 8: albedo 3, shading 1, residual 3, sigma 1 | 15 unused (width 0).
 * Two slot formats:
 *   ROWS      a row-major fp32 [n_points, width] matrix.  save: 13; dz: 14.
 *   FRAGMENTS the operands of the 256-wide weight-gradient products dW = dZ^T X exactly as the matrix core consumes
 *             them: every value v is stored as f16 hi = f16(v') (towards zero) and f16 lo = f16(v' - hi) in 1 KB fragments
 *             [32 channels x 16 points]; with kb = 16-point block of the tile (0..3), cb = 32-channel block:
 *               byte offset = (((tile * 4 + kb) * 8 + cb) * 2 + (0: hi, 1: lo)) * 1024 + lane * 16 + 2 * i      (i = 0..7)
 *               channel = 32 cb + (lane & 31),  point = 64 tile + 32 (kb >> 1) + (i & 3) + 8 ((i >> 2) + 2 (kb & 1)) + 4 (lane >> 5)
 *             (lane = channel, 8 k-values = 8 points: one 16-byte operand slot of v_mfma_f32_32x32x16_f16; the point order
 *             inside a block is the accumulator's register order).
 *             save, slots 2..11: v' = 8 v (the forward kernel's own operand halves); slots 12 (views hidden, 128 channels), 0 and 1
 *             (the encodings, 64 / 32 channels): the same with width / 32 = FOUR / TWO / ONE channel blocks per k-block instead of eight -
 *             byte offset = (((tile * 4 + kb) * (width / 32) + cb) * 2 + plane) * 1024 + ...
 *             dz, slots 2..11 and 12, 13 (the views / semantic hidden layers: 128 channels = FOUR blocks per k-block): v' = 8 v / s_p, s_p = the point's NORMALISER (the power of two above its largest head
 *             gradient; the chain works on normalised gradients, so these halves keep 22 bits whatever a point's gradient
 *             scale) - the normalisers are the first 64 * ceil(n_points / 64) floats of slot 0 of dz, and the weight-gradient
 *             kernels multiply them back in when they bring a fragment to the batch's max |dz|.
 *             Padding points of the last tile hold a copy of the last point (save) / zeros (dz).  Same 4 bytes per element as
 *             fp32; the producers write whole fragments and inerf_mlp_weight_gradient_frag moves them HBM -> LDS by DMA.
 * Behind the slots `save` carries the ReLU masks of h0..h7 as bits (16 384 bytes per 64-point tile, written by
 * inerf_encode_mlp_train and read by inerf_mlp_backward_inputs in place of the activations; layout private to the
 * two kernels) and 64 scalars: always pass a buffer that inerf_encode_mlp_train itself filled, of inerf_mlp_save_floats() floats.
 * The gradient w.r.t. the semantic logits is d_raw[..., 11:11+C] itself (no activation).
 * One kept evaluation is limited to 4 000 000 sample points (a slot is addressed through a 32-bit buffer descriptor):
 * inerf_encode_mlp_train, inerf_mlp_backward_inputs and inerf_mlp_backward return INERF_E_UNSUPPORTED beyond it; split a
 * larger batch into several evaluations and add the parameter gradients (the Python mirror does).
 * ------------------------------------------------------------------------------------------- */
#define INERF_SAVE_SLOTS 16
int64_t inerf_mlp_save_floats(const inerf_net_desc* net, int64_t n_points);
int inerf_mlp_save_slot(const inerf_net_desc* net, int slot, int64_t n_points, int64_t* offset_floats, int* width);
/* 1 when `slot` of the gradient buffer (gradient != 0) / the activation buffer is in FRAGMENT format, 0 for rows, negative: error */
int inerf_mlp_save_slot_is_fragment(int slot, int gradient);
int inerf_encode_mlp_train(const inerf_net_desc* net, const float* packed_weights, const float* rays, const float* z_vals,
                           int64_t n_rays, int n_samples, uint32_t flags, float* raw_out, float* save_out,
                           float* act_max /* optional device float the kernel max-es |activation| into (caller zeroes it) */,
                           int32_t* status, void* stream);

/* Transposed weights for the input-gradient chain, [host] -> [host] like inerf_pack_weights. */
int64_t inerf_bwd_packed_floats(const inerf_net_desc* net);
int inerf_pack_weights_bwd(const inerf_net_desc* net, const float* const* tensors /*[host]*/, int n_tensors,
                           float* packed_out /*[host]*/, int64_t packed_capacity_floats);

/* raw / d_raw: [n_points, CH] (CH as for inerf_encode_mlp with the same flags); save: from
 * inerf_encode_mlp_train on the same points; dz_out: every slot is written for every point. */
int inerf_mlp_backward_inputs(const inerf_net_desc* net, const float* packed_bwd, const float* raw, const float* d_raw,
                              const float* save, int64_t n_points, uint32_t flags, float* dz_out,
                              float* dz_max /* optional device float the kernel max-es |dz| into (caller zeroes it) */,
                              float* head_partial /* optional [inerf_mlp_backward_grid()][inerf_mlp_head_partial_floats()]:
                                 per-workgroup weight / bias gradients of the 1-4-row heads, to be summed by the caller:
                                 residual W [3][128] | albedo,shading outputs [4][256] (rows 0-2: columns 0..127 are
                                 albedo_linear2, row 3: columns 128..255 the shading output) | alpha W [256] |
                                 biases albedo 3, shading 1, residual 3, sigma 1 */,
                              int32_t* status, void* stream);
int inerf_mlp_head_partial_floats(void);
int inerf_mlp_backward_grid(int64_t n_points);

/* One weight gradient: workgroup g of inerf_wgrad_grid(n_points) writes sum over its sample points of G[p, m] * X[p, n]
 * (row-major [M, N]) at partial + g * partial_stride and, if bias_partial is given, its sums of G[p, m] ([M]) at
 * bias_partial + g * partial_stride; the caller adds the grid's tiles (deterministic, no atomics; several gradients can
 * share one [grid, partial_stride] buffer and one final sum).  G[P, ldg] / X[P, ldx]: row-major device
 * matrices (a slot of the gradient / activation buffers, or a 32-column-aligned part of one; pointers 16-byte
 * aligned, ld a multiple of 4); M in {128, 256}, N in {32, 64, 128, 256}.  ranges: device floats {gmax, xmax}, upper
 * bounds of |G| and |X| (e.g. the dz_max of inerf_mlp_backward_inputs; 7.5e3 for activations that passed the forward's
 * range check): the kernel scales the operands by the powers of two that bring those bounds into [2^13, 2^14).
 * The rows are read through 32-bit buffer descriptors: n_points * max(ldg, ldx) * 4 bytes (plus a 25 MB prefetch margin) must
 * stay below 4 GiB - INERF_E_UNSUPPORTED otherwise (4 000 000 points of a 256-wide slot fit). */
int inerf_wgrad_grid(int64_t n_points);
int inerf_mlp_weight_gradient(const float* G, int ldg, const float* X, int ldx, int64_t n_points, int M, int N,
                              const float* ranges, float* partial, float* bias_partial, int64_t partial_stride, void* stream);
/* The same with G a FRAGMENT slot of the gradient buffer (M = 256) - g_scale: the points' normalisers, slot 0 of that buffer,
 * see above - and X row-format: N in {64} (the encoding columns of pts_linears.0 / .5).  ranges as above (of the TRUE |dz|). */
int inerf_mlp_weight_gradient_gfrag(const void* G_frag, const float* g_scale, const float* X, int ldx, int64_t n_points, int N,
                                    const float* ranges, float* partial, float* bias_partial, int64_t partial_stride, void* stream);
/* G row-format (M = 128), X a FRAGMENT slot of the activation buffer (N = 256): views_linears.0's feature columns and the
 * semantic hidden layer.  ranges[0]: an upper bound of |G|. */
int inerf_mlp_weight_gradient_xfrag(const float* G, int ldg, const void* X_frag, int64_t n_points, int M,
                                    const float* ranges, float* partial, float* bias_partial, int64_t partial_stride, void* stream);
/* ... and with BOTH operands FRAGMENT slots (256 x 256: G of the gradient buffer, X of the activation buffer, same points):
 * a ring of LDS stages filled by LDS-DMA - bound by HBM bandwidth.  ranges[0]: an upper bound of the true |dz|.
 * _batch: n_jobs (<= INERF_WGRAD_MAX_BATCH) such products over the SAME points and normalisers in one launch of
 * inerf_wgrad_frag_grid(n_points, n_jobs) workgroups; g_rows[j] x x_cols[j] = 256 x 256, 256 x 64 (X the position encoding's
 * slot), 128 x 256 or 128 x 32 (G the views / semantic hidden layer's slot; X the view encoding's) - NULL: all 256.  Job j is split over
 * inerf_wgrad_frag_rows(n_points, n_jobs, g_rows, x_cols, j) of them (its share of the work), K-slice s
 * writing its [g_rows[j], x_cols[j]] tile at partial[j] + s * partial_stride (and its column sums of G at bias_partial[j] + s *
 * partial_stride where that entry is not NULL): ~n_jobs times fewer partial tiles to write and to sum than n_jobs single
 * launches.  The single form is the batch of one 256 x 256 product (inerf_wgrad_grid(n_points) slices). */
#define INERF_WGRAD_MAX_BATCH 16
int inerf_wgrad_frag_grid(int64_t n_points, int n_jobs);
int inerf_wgrad_frag_rows(int64_t n_points, int n_jobs, const int* g_rows /*[host] or NULL*/, const int* x_cols /*[host] or NULL*/, int job);
int inerf_mlp_weight_gradient_frag_batch(int n_jobs, const void* const* G_frag /*[host]*/, const float* g_scale, const void* const* X_frag /*[host]*/,
                                         const int* g_rows /*[host] or NULL*/, const int* x_cols /*[host] or NULL*/, const float* ranges,
                                         int64_t n_points, float* const* partial /*[host]*/,
                                         float* const* bias_partial /*[host] or NULL*/, int64_t partial_stride, void* stream);
int inerf_mlp_weight_gradient_frag(const void* G_frag, const float* g_scale, const void* X_frag, const float* ranges, int64_t n_points,
                                   float* partial, float* bias_partial, int64_t partial_stride, void* stream);

/* The network's whole backward pass in ONE call: the input-gradient chain, every weight-gradient product (split-K launches,
 * the workgroups' partial tiles side by side in the workspace) and one reduction that writes every sum into its place in
 * grads_out - the flat concatenation (inerf_param_floats() floats) of the reference's parameter tensors in
 * inerf_tensor_info() order, i.e. what autograd leaves in .grad of every nn.Linear after loss.backward()
 * (run_nerf.py:1018, trainer.py:990).  raw / d_raw [n_points, CH]; save: the buffer inerf_encode_mlp_train filled for these
 * points; act_max: the device float that call max-ed |activation| into.  workspace: inerf_mlp_backward_workspace_bytes()
 * bytes (the pre-activation gradients of every layer, 11 KB per point, live only there).  Asynchronous on `stream`,
 * deterministic (fixed summation order).  n_points == 0 zeroes grads_out. */
int64_t inerf_param_floats(const inerf_net_desc* net);
int64_t inerf_mlp_backward_workspace_bytes(const inerf_net_desc* net, int64_t n_points);
int inerf_mlp_backward(const inerf_net_desc* net, const float* packed_bwd, const float* raw, const float* d_raw, const float* save,
                       const float* act_max, int64_t n_points, uint32_t flags, float* grads_out, void* workspace,
                       int64_t workspace_bytes, int32_t* status, void* stream);

/* Where every element of a packed blob comes from, so that a caller can re-pack ON THE DEVICE after each
 * optimiser step (a gather, a per-group max for the power-of-two scales, an f16 hi/lo split) instead of moving the
 * weights through the host.  backward = 0: the INERF_PREC_F16X3 blob of inerf_pack_weights; 1: the blob of
 * inerf_pack_weights_bwd.  Per half-precision position of the blob (2 per float): half_src = 1 + index into the
 * flat concatenation of the tensors in canonical order (0 = zero padding), half_grp = scale group g (>= 0: the
 * "hi" half, f16(v * 2^k_g); < 0: the "lo" half of group -g-1, f16(v * 2^k_g - hi)), 2^k_g bringing the group's
 * largest |v| into [2^13, 2^14).  Per fp32 constant: c_code 0: flat[c_src - 1] * c_mult; 1: 2^-k_g; 2: 2^-k_g / 8
 * (g = c_group).  [host] arrays; returns the number of constants or a negative INERF_E_*. */
int64_t inerf_pack_map(const inerf_net_desc* net, int backward, int32_t* half_src, int32_t* half_grp, int64_t half_capacity,
                       int32_t* c_dst, int32_t* c_src, int32_t* c_group, int32_t* c_code, float* c_mult, int64_t const_capacity,
                       int32_t* n_groups);

/* The re-packing that inerf_pack_map describes, done by the library: params = n_tensors DEVICE pointers to the parameter
 * tensors in canonical order ([host] array; counts = their element counts), the map arrays uploaded to the device as int32
 * (half_src / half_grp: 2 * packed_floats entries; group_src: [n_groups, longest] flat indices (1-based, 0 = padding) of every
 * scale group's sources; the constants as returned), gmax_scratch: n_groups device floats.  Writes the packed blob
 * (packed_floats floats), bit-identical to inerf_pack_weights / inerf_pack_weights_bwd.  A memset and three launches. */
int inerf_repack(const float* const* params, const int64_t* counts, int n_tensors, const int32_t* half_src, const int32_t* half_grp,
                 int64_t packed_floats, const int32_t* group_src, int n_groups, int longest, const int32_t* c_dst,
                 const int32_t* c_src, const int32_t* c_grp, const int32_t* c_code, const float* c_mult, int n_consts,
                 float* gmax_scratch, float* packed_out, void* stream);

/* Alpha compositing.  Replaces raw2outputs: run_nerf.py:359-412 / model_utils.py:39-116.
 * raw[N,S,CH]; z[N,S]; rays_d: pointer to the first direction, consecutive rays `rays_d_stride`
 * floats apart (3 for a [N,3] tensor, 11 for the d-part of a packed ray batch);
 * noise[N,S] (already scaled by raw_noise_std) or NULL.  n_classes / feat_dim select the optional
 * semantic (raw[...,11:11+C]) and endpoint-feature (the LAST feat_dim channels) sums.
 * Any output pointer may be NULL to skip it.  disp is NaN where acc == 0, as in the reference. */
typedef struct inerf_composite_out {
    float* rgb;       /* [N,3] */
    float* disp;      /* [N]   */
    float* acc;       /* [N]   */
    float* depth;     /* [N]   */
    float* albedo;    /* [N,3] */
    float* shading;   /* [N]   */
    float* residual;  /* [N,3] */
    float* sem;       /* [N,C]        or NULL */
    float* feat;      /* [N,feat_dim] or NULL */
    float* weights;   /* [N,S]        or NULL */
} inerf_composite_out;

int inerf_composite(const float* raw, const float* z_vals, const float* rays_d, int rays_d_stride, const float* noise,
                    int64_t n_rays, int n_samples, int channels, int n_classes, int feat_dim, uint32_t flags,
                    const inerf_composite_out* out, void* stream);

/* Backward of the alpha compositing: d_raw[N,S,CH] = sum over the given output gradients of
 * d(output)/d(raw).  Replaces what autograd records for raw2outputs when the trainers call
 * loss.backward() (run_nerf.py:1018 through :359-412; trainer.py:990 through model_utils.py:39-116).
 * Same inputs as inerf_composite (the forward pass is recomputed, nothing has to be saved); `grads`
 * holds dL/d(output) for every output the loss used, NULL for the others (const pointers, the
 * struct is inerf_composite_out read as inputs; `weights` is dL/d weights[N,S]).  z_vals, rays_d
 * and noise get no gradient (the reference detaches the resampled depths, run_nerf.py:501, and the
 * rays are data).  Every element of d_raw is written (zero where nothing depends on it). */
int inerf_composite_backward(const float* raw, const float* z_vals, const float* rays_d, int rays_d_stride,
                             const float* noise, int64_t n_rays, int n_samples, int channels, int n_classes,
                             int feat_dim, uint32_t flags, const inerf_composite_out* grads, float* d_raw, void* stream);

/* Hierarchical resampling + merge.  Replaces z_vals_mid + sample_pdf + sort(cat(...)) + std:
 * run_nerf.py:499-503,519 + run_nerf_helpers.py:402-445 / trainer.py:758-766,799 + rays.py:176-220.
 * z_coarse[N,Sc], weights[N,Sc] (the full coarse weights; the kernel takes [1:-1] itself);
 * u: [n_importance] shared, or [N,n_importance] with INERF_FLAG_U_PER_RAY.
 * Outputs (any may be NULL): z_samples[N,n_importance], z_merged[N,Sc+n_importance] ascending,
 * z_std[N] (population std of the new samples).  Limits: 3 <= Sc <= 256, 1 <= n_importance <= 512. */
int inerf_sample_fine(const float* z_coarse, const float* weights, const float* u, int64_t n_rays, int n_coarse,
                      int n_importance, uint32_t flags, float* z_samples, float* z_merged, float* z_std, void* stream);

/* Stand-alone inverse-CDF sampler with the reference's own signature sample_pdf(bins, weights, N):
 * run_nerf_helpers.py:402-445 / rays.py:176-220.  bins[N,n_bins], weights[N,n_bins-1],
 * u as above -> samples[N,n_samples].  Limits: 2 <= n_bins <= 256, 1 <= n_samples <= 512. */
int inerf_sample_pdf(const float* bins, const float* weights, const float* u, int64_t n_rays, int n_bins, int n_samples,
                     uint32_t flags, float* samples, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole path.  Replaces render_rays (run_nerf.py:415-528) / SSRTrainer.volumetric_rendering
 * (trainer.py:717-808) for one ray batch: coarse sampling -> coarse MLP -> compositing ->
 * resampling -> fine MLP -> compositing, all enqueued on `stream` with no host synchronisation.
 * ------------------------------------------------------------------------------------------- */
typedef struct inerf_render_args {
    /* networks */
    inerf_net_desc net;
    const float* packed_coarse;   /* device blob                                                   */
    const float* packed_fine;     /* device blob; NULL = reuse coarse (network_fine is None)        */
    /* inputs */
    const float* rays;            /* [N,11]                                                        */
    int64_t      n_rays;
    int32_t      n_samples;       /* coarse samples per ray                                        */
    int32_t      n_importance;    /* 0 = coarse pass only                                          */
    uint32_t     flags;           /* INERF_FLAG_*                                                  */
    const float* t_vals;          /* [n_samples] linspace(0,1,n_samples)                           */
    const float* t_rand;          /* [N,n_samples] or NULL                                         */
    const float* u;               /* [n_importance] or [N,n_importance] (INERF_FLAG_U_PER_RAY)     */
    const float* noise_coarse;    /* [N,n_samples] or NULL                                         */
    const float* noise_fine;      /* [N,n_samples+n_importance] or NULL                            */
    /* outputs (any pointer may be NULL) */
    inerf_composite_out coarse;
    inerf_composite_out fine;
    float* z_std;                 /* [N]                                                           */
    float* raw_coarse;            /* [N,n_samples,CHc]; NULL = keep in workspace                   */
    float* raw_fine;              /* [N,n_samples+n_importance,CHf]; NULL = keep in workspace      */
    float* z_coarse;              /* [N,n_samples]          optional copy-out of stage tensors     */
    float* z_samples;             /* [N,n_importance]                                               */
    float* z_fine;                /* [N,n_samples+n_importance]                                     */
    int32_t* status;              /* optional device word for INERF_STATUS_* bits (caller zeroes it) */
    int64_t  status_rays;         /* <= 0: one status word; > 0: ceil(n_rays / status_rays) words, one per that many rays
                                     (inerf_encode_mlp_chunked)                                     */
    /* scratch */
    void*   workspace;
    int64_t workspace_bytes;
} inerf_render_args;

/* Bytes of workspace inerf_render_rays needs for these sizes when the caller supplies none of the optional stage
 * outputs (an upper bound for every call with these sizes). */
int64_t inerf_workspace_bytes(const inerf_net_desc* net, int64_t n_rays, int n_samples, int n_importance, uint32_t flags);

/* Bytes of workspace THIS call needs: stage tensors the caller supplies as outputs (raw_coarse, raw_fine, z_coarse,
 * z_samples, z_fine, coarse.weights) are written in place and get no workspace region - raw alone is N x S x CH
 * floats.  `workspace` / `workspace_bytes` of the argument are ignored.  May return 0 (workspace may then be NULL). */
int64_t inerf_render_workspace_bytes(const inerf_render_args* args);

int inerf_render_rays(const inerf_render_args* args, void* stream);

/* Raw channel counts for this network. */
int inerf_raw_channels(const inerf_net_desc* net, uint32_t flags, int fine);

/* ---------------------------------------------------------------------------------------------
 * Image-level neighbours of the path.
 * ------------------------------------------------------------------------------------------- */

/* Ray generation: the [n_poses * height * width, 11] ray batch [o3, d3, near, far, viewdir3] of a pinhole camera.
 * Replaces get_rays (run_nerf_helpers.py:359-368) + the view-direction normalisation and concatenation inside
 * render() (run_nerf.py:99-128), and get_rays_camera / get_rays_world / create_rays (SSR/models/rays.py:27-67,
 * 223-256, depth_type "z").  poses: device, camera-to-world matrices as rows of 4 floats ([3,4] or [4,4] row-major),
 * consecutive poses `pose_stride` (>= 12) floats apart; static_poses: NULL, or the c2w_staticcam of the reference
 * (origins and directions from it, view directions from `poses`).  Pixel (i = column, j = row) of pose b is ray
 * b*H*W + j*W + i.  INERF_CAM_OPENGL: dirs = ((i-cx)/fx, -(j-cy)/fy, -1) (object-level, SSR convention "opengl");
 * otherwise ((i-cx)/fx, (j-cy)/fy, 1) (SSR "opencv").  Bit-identical to the reference's CPU evaluation
 * (tests/golden/rays_generators.npz): every fp32 operation is issued in ATen's order (see csrc/frame_ops.hip). */
#define INERF_CAM_OPENGL 1u
int inerf_gen_rays(const float* poses, int pose_stride, const float* static_poses, int n_poses, int height, int width,
                   float fx, float fy, float cx, float cy, float near, float far, uint32_t flags, float* rays_out,
                   void* stream);

/* to8b on the device: out[i] = (uint8)(255 * clip(values[i], 0, 1)) (run_nerf_helpers.py:13 / trainer.py:1242), so an
 * image loop (run_nerf.py:142-212, trainer.py:1221-1389) copies a quarter of the bytes to the host.  NaN -> 0. */
int inerf_frame_to_u8(const float* values, int64_t n, unsigned char* out, void* stream);

/* Nearest-anchor lookup of the albedo clustering, every semantic class in one launch.  Replaces
 * Cluster_Manager.dest_color / dest_class (SSR/training/cluster.py:73-98) and, underneath, Cluster.dest_color /
 * dest_class (:275-297), mapping_color (:324-330), compute_dist (:299-305) and nearest_anchor (:307-310); called
 * once per training step (trainer.py:913-920) and once per rendered frame (:1427-1430).
 *   rgb[n,3]; label[n] (int64; ignored - may be NULL - with INERF_CLUSTER_IGNORE_LABEL, the reference's
 *   class_num == 1 shortcut of dest_color, cluster.py:75-77);
 *   anchors[A,4] = {a0, a1, a2, a0^2+a1^2+a2^2} in the mapped colour space, 16-byte aligned, the classes' anchors
 *   back to back: class c owns rows anchor_begin[c] .. anchor_begin[c+1] (an empty range = the reference's
 *   `clusters[c] is None`); links[A] = centre of each anchor, relative to its class (int32);
 *   factor[K] = intensity_factor of each class's cluster; centers[Ctot,3] = rgb_centers back to back, class c
 *   starting at row center_begin[c].  All device pointers.
 * Outputs (either may be NULL): out_color[n,3] = rgb_centers[links[argmin]] of the pixel's class - the pixel itself
 * where its label has no cluster or lies outside [0, K); out_class[n] (int64) = links[argmin], 0 for those pixels.
 * argmin follows torch: the first minimal index wins, NaN distances (zero-intensity pixels) count as minimal. */
#define INERF_CLUSTER_IGNORE_LABEL 1u
int inerf_cluster_lookup(const float* rgb, const int64_t* label, int64_t n_pixels, const float* anchors,
                         const int32_t* links, const int32_t* anchor_begin, const float* factor, const float* centers,
                         const int32_t* center_begin, int n_classes, uint32_t flags, float* out_color,
                         int64_t* out_class, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Networks outside the fused architecture, and fp32 training batches: one launch per nn.Linear on the fp32 matrix core.
 *
 * The reference builds NeRF(D=args.netdepth, W=args.netwidth, skips, use_viewdirs) / Semantic_NeRF(...) from its flags
 * (object_level/run_nerf.py:286-296, SSR/training/trainer.py:811-846); the fused kernels above implement D=8, W=256,
 * skips=[4].  These entry points evaluate ANY such network layer by layer (object_level/run_nerf_helpers.py:284-321,
 * SSR/models/semantic_nerf.py:120-181: every F.linear / F.relu / F.sigmoid / torch.cat of the two forwards) and what
 * loss.backward() records for those layers (run_nerf.py:1018, trainer.py:990), in exact fp32 (v_mfma_f32_32x32x2_f32,
 * fp32 accumulate), activations in caller-owned HBM buffers.  The Python mirrors also use them for a training batch whose
 * activations leave the f16 range of the split-precision kernels, so no ATen GEMM remains on the path.
 * --------------------------------------------------------------------------------------------------------------- */
#define INERF_ACT_NONE     0
#define INERF_ACT_RELU     1   /* F.relu: negative -> 0, NaN stays NaN */
#define INERF_ACT_SIGMOID  2   /* torch.sigmoid: 1 / (1 + exp(-x)), IEEE division */

/* C[i, j] = epilogue(sum_k A(i, k) B(j, k)),  i < m, j < n, k < k:
 *   A(i, k) = a[i * a_sm + k * a_sk], B(j, k) = b[j * b_sn + k * b_sk]; each operand needs ONE unit stride (either one);
 *   epilogue, in this order: + bias[j] (if not NULL), + add[i * add_ld + j] (if not NULL; add == c accumulates), act,
 *   then zeroed where gate[i * gate_ld + j] <= 0 (if gate is not NULL: the ReLU backward of the layer that produced `gate`);
 *   c[i * c_ld + j] may be a column range of a wider buffer (that is how torch.cat([input_pts, h]) and
 *   torch.cat([feature, input_views]) are formed, run_nerf_helpers.py:291,308).
 * nn.Linear forward: a = X (a_sm = ld, a_sk = 1), b = weight (b_sn = in_features, b_sk = 1), m = points, n = out, k = in.
 * Its input gradient:  a = dZ (a_sm = ld, a_sk = 1), b = weight (+ first input column), b_sn = 1, b_sk = in_features,
 *                      n = input columns, k = out_features. */
typedef struct inerf_linear_args {
    const float* a; int64_t a_sm, a_sk;
    const float* b; int64_t b_sn, b_sk;
    const float* bias;
    const float* add; int64_t add_ld;
    const float* gate; int64_t gate_ld;
    float* c; int64_t c_ld;
    int64_t m; int32_t n; int32_t act; int64_t k;
} inerf_linear_args;
int inerf_linear(const inerf_linear_args* args, void* stream);

/* d_weight[rows, cols] (=|+=) sum_p g[p, r] x[p, c];  d_bias[rows] (=|+=) sum_p g[p, r]  (NULL: not wanted) - what autograd
 * accumulates into nn.Linear.weight.grad / .bias.grad for g = the gradient of the layer's output (already multiplied by the
 * activation's derivative), x = its input.  g[p * ldg + r], x[p * ldx + c].  Split over the points into a fixed number of
 * partial products (workspace: inerf_linear_wgrad_workspace_bytes), added in ascending order: bit-identical run to run. */
int64_t inerf_linear_wgrad_workspace_bytes(int64_t n_points, int rows, int cols);
int inerf_linear_wgrad(const float* g, int64_t ldg, int rows, const float* x, int64_t ldx, int cols, int64_t n_points,
                       float* d_weight, float* d_bias, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* Embedder.embed (run_nerf_helpers.py:195-225; semantic_nerf.py:50-66 with divisor = scalar_factor) into
 * out[p * ld + 0 .. 3 + 6 n_freqs): [x, sin(x), cos(x), sin(2x), cos(2x), ...] of x = (o + d * z[p]) / divisor
 * (run_nerf.py:488), or of the ray's view direction when `directions` != 0 (z_vals may then be NULL); p = ray * n_samples + s. */
int inerf_embed(const float* rays, const float* z_vals, int64_t n_rays, int n_samples, int n_freqs, float divisor, int directions,
                float* out, int64_t ld, void* stream);

/* raw[p, 0:3] = raw[p, 4:7] * raw[p, 7] + raw[p, 8:11] (rgb = albedo * shading + residual, run_nerf_helpers.py:319) and its
 * backward together with the three sigmoids': dz[p, 0:3] / [3] / [4:7] = gradients of the PRE-sigmoid albedo / shading /
 * residual given d_raw (the gradient of all raw channels) and raw (the forward's values); dz is [n_points, 8], dz[p, 7] = 0. */
int inerf_intrinsic_combine(float* raw, int64_t ld, int64_t n_points, void* stream);
int inerf_intrinsic_combine_backward(const float* raw, const float* d_raw, int64_t ld, int64_t n_points, float* dz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INERF_H */
