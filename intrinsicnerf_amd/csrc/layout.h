// layout.h - geometry of a work tile in LDS and of the packed weight blob.  Shared by the host
// packer (pack.cpp) and the encode+MLP kernel (mlp.hip); plain C++, no HIP types.
#pragma once
#include <stdint.h>

#include "../../include/inerf.h"

namespace inerf {

// ---- architecture constants (run_nerf.py:545-552: netdepth 8, netwidth 256; skips=[4], :285) ----
constexpr int kWidth = 256;        // W
constexpr int kDepth = 8;          // D
constexpr int kSkipInput = 5;      // pts_linears[5] consumes cat([pts, h]) (run_nerf_helpers.py:290-291)
constexpr int kHalf = 128;         // W/2
constexpr int kMaxLxyz = 10;
constexpr int kMaxLdir = 4;

// ---- tile geometry ----
// One workgroup = 4 waves = one tile of 64 sample points.  Activations live in LDS as
// X[point][column] (column = channel, contiguous), so both MFMA operands are read along k with one
// ds_read_b128 per 4 k-values.  The row stride is 4*odd floats: rows 0..15 then start on 16
// distinct 16-byte bank slots, which makes every ds_read_b128 / ds_write_b128 used here
// conflict-free (bank = (36*row + const) mod 64 dwords).
constexpr int kTilePoints = 64;
constexpr int kWaves = 4;
constexpr int kEncCols = 64;       // 3 + 6*10 = 63 encoded xyz channels + 1 zero pad
constexpr int kDirCols = 32;       // 3 + 6*4  = 27 encoded dir channels + 5 zero pad
constexpr int kColEnc = 0;
constexpr int kColDir = kColEnc + kEncCols;     // 64
constexpr int kColA = kColDir + kDirCols;       // 96   activation buffer A (256 wide)
constexpr int kColB = kColA + kWidth;           // 352  activation buffer B (256 wide)
constexpr int kLdsStride = kColB + kWidth + 4;  // 612 = 4 * 153
constexpr int kLdsBytes = kTilePoints * kLdsStride * 4;   // 156,672 B of the CU's 160 KiB

// ---- packed blob ----
// "Wide" GEMMs (32x32x2 MFMA): the output channels are split evenly over the 4 waves; wave w owns
// RB = n_out/128 blocks of 32 channels.  For k-block kb (8 consecutive virtual k) and row block rb,
// the 64 lanes of the wave load one float4 each, contiguous:
//     blob[off + (((w*KB + kb)*RB + rb)*64 + lane)*4 + c] = W[w*32*RB + 32*rb + (lane&31)][8*kb + 4*(lane>>5) + c]
// i.e. exactly the A-operand fragments of four consecutive v_mfma_f32_32x32x2_f32 (c = 0..3).
// "Skinny" GEMMs (16x16x4 MFMA, <= 16*RBS output rows, every wave reads the same weights):
//     blob[off + ((rb*KB16 + kb)*64 + lane)*4 + c] = W[16*rb + (lane&15)][16*kb + 4*(lane>>4) + c]
// With precision INERF_PREC_F16X3 the same float-sized regions hold f16 fragments instead (see
// pack.cpp): per wave and 16-wide k-block, for each row block a 1 KiB "hi" fragment followed by a
// 1 KiB "lo" fragment, lane l holding 8 halfs W'[..+(l&31)][16*kb + 8*(l>>5) .. +7]; skinny GEMMs per
// 32-wide k-block, lane l holding W'[16*rb + (l&15)][32*kb + 8*(l>>4) .. +7].  W' = W * 2^kw with a
// per-GEMM power of two kw that brings max|W'| into (2^13, 2^14]; hi = f16(W'), lo = f16(W' - hi).
// Region sizes are identical (2 halfs per weight = 4 bytes), so NetLayout serves both formats.
// Activations travel scaled by kActScale, so the float after a bias vector holds the factor that maps
// the f16 kernel's accumulator back: 2^-kw for wide GEMMs (whose biases are stored pre-multiplied by
// kActScale, the output staying in the scaled domain) and 2^-kw / kActScale for the skinny output heads.
// "Register-operand" copies of the two output heads that follow a hidden layer (as2r, resr; f16 format only,
// zero otherwise): the two-workgroups-per-CU kernel keeps that hidden layer in registers - the accumulator
// layout of v_mfma_f32_32x32x16_f16 read back as the B operand of the next MFMA - so the head's weights are
// stored as 32x16 A fragments whose k order follows the accumulator registers:
//     [wave][q][hi|lo][lane][8 halfs],  row = lane & 31 (zero beyond the head's rows),
//     hidden channel = wave * 16*Q + regop_chan(q, lane >> 5, c),  Q = k-blocks per wave (4: as2r, 2: resr).
// sem2q: the same per 32-class block rb: [rb][wave][q][hi|lo][lane][8 halfs], row = 32*rb + (lane & 31), Q = 2.
// The semantic head of that kernel's TRAINING form works per wave on v_mfma_f32_16x16x32_f16 (16 points x 16 rows): sem1s is sem1 in the
// skinny format with sem1's scale; its four-row accumulators (lane l: rows 4*(l>>4) .. +3 of a 16-row block) of two
// neighbouring row blocks form one 32-deep B operand, so sem2r stores semantic_linear.1 as skinny fragments whose k order is
//     hidden channel = 32*kb + regop16_chan(lane >> 4, c),   regop16_chan(g, c) = 16*(c >> 2) + 4*g + (c & 3).
// Virtual k runs over the concatenation of the layer's LDS source segments (e.g. [enc64 | h256] for
// pts_linears[5]); padded columns/rows hold zeros.  Biases are stored unpermuted (padded with zeros).
constexpr float kActScale = 8.0f;   // INERF_PREC_F16X3: activations are split as f16 hi/lo of (8 * value)

struct GemmSlot {
    int32_t w;      // float offset of the packed weights
    int32_t b;      // float offset of the bias vector
};

struct NetLayout {
    GemmSlot trunk[kDepth];   // pts_linears.0-7                       wide, 256 out
    GemmSlot alpha;           // alpha_linear                          skinny, 1 row block, K=256
    GemmSlot sem1;            // semantic_linear.0.0 (ssr, C>0)        wide, 128 out, K=256
    GemmSlot sem2;            // semantic_linear.1                     skinny, sem_rbs row blocks, K=128
    GemmSlot as1;             // albedo_linear1 (rows 0-127) + shading hidden (rows 128-255)   wide, K=256
    GemmSlot as2;             // rows 0-2 albedo_linear2 (k<128), row 3 shading out (k>=128)   skinny, K=256
    GemmSlot feat;            // feature_linear                        wide, 256 out, K=256
    GemmSlot views;           // views_linears.0 over [feature256 | dir32]   wide, 128 out, K=288
    GemmSlot res;             // residual head                         skinny, K=128
    GemmSlot as2r;            // as2 again, as register-operand fragments (w only; bias/scale are as2's)
    GemmSlot resr;            // res again, as register-operand fragments (w only; bias/scale are res's)
    GemmSlot sem1s;           // sem1 again, in the skinny format (8 row blocks, K=256; w only): every wave of the two-workgroup
                              //   kernel computes the whole semantic hidden layer for its own 16 points
    GemmSlot sem2r;           // sem2 again, as 16-row register-operand fragments (w only)
    GemmSlot sem2q;           // sem2 once more, as 32-row register-operand fragments per 32-class block (w only): the inference form
                              //   of the two-workgroup kernel splits the semantic hidden layer over the waves BY CHANNEL (32 each,
                              //   like the view layer), so semantic_linear.0.0 is streamed once per tile instead of four times
    int32_t sem_rbs;          // ceil(C/16), 0 when the semantic head is absent
    int32_t sem_rb32;         // ceil(C/32)
    int32_t total_floats;
};

// hidden channel (within a wave's block of channels) that slot s of lane-half h of k-block q carries when a
// 32x32 accumulator (register j: row (j&3) + 8*(j>>2) + 4*h) is reused as a B operand (registers 8*(q&1) .. +7
// of row block q>>1)
inline int regop_chan(int q, int h, int s) { return 32 * (q >> 1) + 8 * (2 * (q & 1) + (s >> 2)) + 4 * h + (s & 3); }

inline int regop16_chan(int g, int c) { return 16 * (c >> 2) + 4 * g + (c & 3); }

inline int trunk_k(int layer) { return layer == 0 ? kEncCols : (layer == kSkipInput ? kEncCols + kWidth : kWidth); }

inline NetLayout make_layout(const inerf_net_desc& net) {
    NetLayout L{};
    int32_t off = 0;
    auto take = [&](int32_t n) { int32_t o = off; off += (n + 3) & ~3; return o; };
    // each bias vector is followed by 4 floats of per-GEMM constants ([0] = output scale, see kScaleSlot)
    auto wide = [&](GemmSlot& s, int n_out, int k) { s.w = take(n_out * k); s.b = take(n_out + 4); };
    auto skinny = [&](GemmSlot& s, int rbs, int k) { s.w = take(rbs * 16 * k); s.b = take(rbs * 16 + 4); };
    for (int i = 0; i < kDepth; ++i) wide(L.trunk[i], kWidth, trunk_k(i));
    skinny(L.alpha, 1, kWidth);
    L.sem_rbs = 0;
    if (net.variant == INERF_VARIANT_SSR && net.n_classes > 0) {
        L.sem_rbs = (net.n_classes + 15) / 16;
        wide(L.sem1, kHalf, kWidth);
        skinny(L.sem2, L.sem_rbs, kHalf);
    }
    wide(L.as1, kWidth, kWidth);
    skinny(L.as2, 1, kWidth);
    wide(L.feat, kWidth, kWidth);
    wide(L.views, kHalf, kWidth + kDirCols);
    skinny(L.res, 1, kHalf);
    L.as2r.w = take(32 * kWidth);   L.as2r.b = L.as2.b;      // 4 waves x 4 k-blocks x (hi + lo) KiB
    L.resr.w = take(32 * kHalf);    L.resr.b = L.res.b;      // 4 waves x 2 k-blocks x (hi + lo) KiB
    if (L.sem_rbs > 0) {
        L.sem1s.w = take(kHalf * kWidth);           L.sem1s.b = L.sem1.b;
        L.sem2r.w = take(L.sem_rbs * 16 * kHalf);   L.sem2r.b = L.sem2.b;
        L.sem_rb32 = (net.n_classes + 31) / 32;
        L.sem2q.w = take(L.sem_rb32 * 32 * kHalf);  L.sem2q.b = L.sem2.b;      // per block: 4 waves x 2 k-blocks x (hi + lo) KiB
    }
    L.total_floats = off;
    return L;
}

// ---- training: activations kept by the forward pass / pre-activation gradients produced by the backward pass ----
// One buffer of [slot][point][width] 4-byte elements per evaluation; the same slot list serves both buffers.  Slots are sized
// for WHOLE 64-point tiles (padded_points): the last tile's padding rows belong to the slot.  Two slot formats:
//   * ROWS: a plain row-major fp32 [n_points, width] matrix - what the VALU stages of the chain and the weight-gradient
//     products with narrow / 128-row operands read;
//   * FRAGMENTS (256-wide slots that feed the nine 256 x 256 weight-gradient products, dW = dZ^T X with K = sample points):
//     the values ALREADY SPLIT into f16 hi / lo and stored as the operand fragments of v_mfma_f32_32x32x16_f16 - per 16-point
//     k-block kb (four per tile) and 32-channel block cb one 1 KB fragment per plane,
//         byte offset = (((tile * 4 + kb) * 8 + cb) * 2 + plane) * 1024 + lane * 16 + 2 * i,
//         channel     = 32 * cb + (lane & 31),
//         point       = 64 * tile + 32 * (kb >> 1) + frag_point(kb & 1, lane >> 5, i)        (i = 0..7),
//     i.e. exactly what a lane of the weight-gradient kernel feeds the matrix core (lane = channel, 8 k-values = points): the
//     consumer moves the fragments HBM -> LDS by LDS-DMA and never converts, transposes or even touches them with the VALU.
//     The producers hold these values as hi / lo planes in LDS anyway (X[point][channel]); the transposition is one pass of
//     the matrix core over the planes (a 0/1 selector as B operand: D[point][channel] comes back with lane = channel,
//     registers = points, mlp_f16_dev.h planes_to_frag) and a fragment leaves the CU as ONE contiguous 1 KB store.  Same
//     4 bytes per element as fp32.  The point order inside a k-block is the accumulator's register order - the same for the
//     activations and the gradients, and the one the row-format kernel's transposing MFMAs produce - so products may mix
//     formats (G fragments x row-format X).
//     Scales: activations are stored as kActScale * h (the forward's own LDS planes: their range is the kernel's guard).
//     Gradients have no scale that is known before the whole batch has been walked, and a sum over points cannot be re-scaled
//     per point afterwards; f16's 30 binades do not hold 13 binades between the points of a batch times ~16 between the layers
//     of the chain at any fixed scale (a first version that split the true dz in the producer lost 6 bits in the lower trunk
//     layers).  So the gradient slots hold the chain's own NORMALISED halves - kActScale * dZ / s_p, s_p the point's normaliser
//     (a power of two): 22 bits whatever the point's scale, straight from the chain's LDS planes - and the normalisers travel
//     beside them (SAVE_ENC of the gradient buffer: one float per point); the consumer multiplies them back in when it brings
//     its own operands to the batch's max |dz| (~50 VALU instructions per fragment pair, beside 24 MFMAs per k-block).
//   activation buffer: SEMH rows; H0..H7, FEAT, AS1H fragments, VH fragments of a 128-channel slot; ENC / DIR fragments of a 64- /
//                      32-channel slot (the same format with four / two / one channel blocks per k-block instead of eight);
//                      the chain reads h7 - ReLU mask and operand of the alpha_linear weight gradient - from its fragments: the
//                      matrix core transposes them back; H7R unused;
//   gradient buffer:   DPRE rows (8 floats per point: albedo 3, shading 1, residual 3, sigma 1); H0..H7, AS1H, FEAT fragments,
//                      VH and SEMH fragments of 128-channel slots (four channel blocks per k-block); ENC: the normalisers;
//                      DIR, H7R unused.
enum SaveSlot {
    SAVE_ENC = 0,      // 64  encoded position (63 + zero pad)
    SAVE_DIR,          // 32  encoded view direction (27 + zero pad)
    SAVE_H0,           // 256 x 8: trunk layer outputs h0..h7 (post-ReLU)
    SAVE_H7 = SAVE_H0 + 7,
    SAVE_AS1H,         // 256 albedo hidden (0..127) | shading hidden (128..255), post-ReLU
    SAVE_FEAT,         // 256 feature_linear output (no activation)
    SAVE_VH,           // 128 views_linears.0 output, post-ReLU
    SAVE_SEMH,         // 128 semantic hidden, post-ReLU (SSR with classes only; width 0 otherwise)
    SAVE_DPRE,         // 8   (gradient buffer only)
    SAVE_H7R,          // 0   (unused: early in round 4, h7 once more as rows)
    SAVE_SLOTS
};

constexpr int kFragBytes = 1024;                       // one operand fragment: 64 lanes x 8 halfs
constexpr int kFragKbBytes = 8 * 2 * kFragBytes;       // one 16-point k-block of a 256-wide slot: [cb 8][hi | lo]
constexpr int kFragTileBytes = 4 * kFragKbBytes;       // = 64 points x 256 channels x 4 bytes
constexpr int frag_point(int q, int h, int i) { return (i & 3) + 8 * ((i >> 2) + 2 * q) + 4 * h; }
inline int64_t padded_points(int64_t n_points) { return (n_points + kTilePoints - 1) / kTilePoints * kTilePoints; }

inline int save_width(const inerf_net_desc& net, int slot) {
    if (slot == SAVE_ENC) return kEncCols;
    if (slot == SAVE_DIR) return kDirCols;
    if (slot >= SAVE_H0 && slot <= SAVE_H7) return kWidth;
    if (slot == SAVE_AS1H || slot == SAVE_FEAT) return kWidth;
    if (slot == SAVE_VH) return kHalf;
    if (slot == SAVE_SEMH) return (net.variant == INERF_VARIANT_SSR && net.n_classes > 0) ? kHalf : 0;
    if (slot == SAVE_DPRE) return 8;
    return 0;
}

// format of a slot: 1 = fragments, 0 = rows (gradient: the buffer of pre-activation gradients, else the activation buffer)
inline int save_is_frag(int slot, bool gradient) {
    if ((slot >= SAVE_H0 && slot <= SAVE_H7) || slot == SAVE_FEAT || slot == SAVE_AS1H || slot == SAVE_VH) return 1;   // (VH: 128 channels = 4 blocks)
    if (slot == SAVE_ENC || slot == SAVE_DIR) return gradient ? 0 : 1;      // activation buffer: NARROW fragment slots (64 / 32 channels = 2 / 1 blocks per k-block)
    return (gradient && slot == SAVE_SEMH) ? 1 : 0;                         // (the semantic hidden layer's activations stay rows: a scalar-loop stage reads them)
}

inline int64_t save_offset(const inerf_net_desc& net, int slot, int64_t n_points) {     // in floats
    int64_t w = 0;
    for (int s = 0; s < slot; ++s) w += save_width(net, s);
    return w * padded_points(n_points);
}

// Behind the slots of the ACTIVATION buffer: the ReLU masks of the trunk outputs h0..h7, one bit per activation, which the
// input-gradient chain reads instead of the activations themselves (1 KB instead of 8 KB per sample point).  The chain's
// accumulator layout is the forward's, so a mask word is simply what one lane owns: per 64-point tile, per layer, per wave
// (64 channels), per lane two 32-bit words - word rb = row block (32 channels), bit 31 - (16*pb + 4*g + i) = the lane's
// register 4*g + i of point block pb (the forward shifts the bits in in that order), i.e. channel
// 64*wave + 32*rb + 8*g + 4*(lane >> 5) + i of point 32*pb + (lane & 31).
// Tiles are whole (the last one is padded), so every word of the area is written by the forward.
constexpr int kReluBitLayers = 8;                                   // h0..h7 (h7's: the two-workgroup chain masks d h7 with them)
constexpr int kReluBitTileBytes = kReluBitLayers * 4 * 64 * 8;      // 16,384 B per 64-point tile
inline int64_t relu_bits_offset(const inerf_net_desc& net, int64_t n_points) { return save_offset(net, SAVE_SLOTS, n_points); }
inline int64_t relu_bits_floats(int64_t n_points) { return (n_points + kTilePoints - 1) / kTilePoints * (kReluBitTileBytes / 4); }
// Behind the mask area: 64 floats of per-evaluation scalars (reserved).
constexpr int kSaveScalars = 64;
inline int64_t save_scalars_offset(const inerf_net_desc& net, int64_t n_points) { return relu_bits_offset(net, n_points) + relu_bits_floats(n_points); }
inline int64_t save_total_floats(const inerf_net_desc& net, int64_t n_points) {
    return save_scalars_offset(net, n_points) + kSaveScalars;
}

// ---- packed blob of the input-gradient (dgrad) chain, mlp_bwd.hip ----
// The transposed layers in the wide f16 fragment format above (rows = the forward layer's INPUT channels, k = its
// output channels), preceded by nothing and followed by 4 constants ([0] = accumulator -> output factor).  The three
// matrices whose products are summed into d h7 (feature_linear^T, as1^T, sem1^T) share ONE weight scale so that they
// can share one accumulator.  The small heads are plain fp32, laid out for per-channel VALU use:
//   res_w[128][4]   residual head: (W[0][c], W[1][c], W[2][c], 0)
//   as2_w[256][4]   c < 128: (albedo_linear2[0..2][c], 0); c >= 128: (0, 0, 0, shading out[c-128])
//   alpha_w[256]    alpha_linear
//   sem2_w[C][128]  semantic_linear.1, transposed: [class][hidden]
struct BwdLayout {
    GemmSlot views_t;         // 256 x 128  (feature part of views_linears.0)
    GemmSlot feat_t;          // 256 x 256
    GemmSlot as1_t;           // 256 x 256
    GemmSlot sem1_t;          // 256 x 128  (ssr with classes)
    GemmSlot trunk_t[kDepth]; // [1..7]: 256 x 256 (layer 5: its h part); [0] unused
    int32_t res_w, as2_w, alpha_w, sem2_w;
    int32_t has_sem;
    int32_t total_floats;
};

inline BwdLayout make_bwd_layout(const inerf_net_desc& net) {
    BwdLayout L{};
    int32_t off = 0;
    auto take = [&](int32_t n) { int32_t o = off; off += (n + 3) & ~3; return o; };
    auto wide = [&](GemmSlot& s, int n_out, int k) { s.w = take(n_out * k); s.b = take(4); };
    wide(L.views_t, kWidth, kHalf);
    wide(L.feat_t, kWidth, kWidth);
    wide(L.as1_t, kWidth, kWidth);
    L.has_sem = (net.variant == INERF_VARIANT_SSR && net.n_classes > 0) ? 1 : 0;
    if (L.has_sem) wide(L.sem1_t, kWidth, kHalf);
    for (int i = 1; i < kDepth; ++i) wide(L.trunk_t[i], kWidth, kWidth);
    L.res_w = take(kHalf * 4);
    L.as2_w = take(kWidth * 4);
    L.alpha_w = take(kWidth);
    L.sem2_w = L.has_sem ? take(net.n_classes * kHalf) : 0;
    L.total_floats = off;
    return L;
}

inline bool net_supported(const inerf_net_desc& n) {
    if (n.variant != INERF_VARIANT_OBJECT && n.variant != INERF_VARIANT_SSR) return false;
    if (n.l_xyz < 0 || n.l_xyz > kMaxLxyz || n.l_dir < 0 || n.l_dir > kMaxLdir) return false;
    if (n.n_classes < 0 || n.n_classes > INERF_MAX_CLASSES) return false;
    if (n.variant == INERF_VARIANT_OBJECT && n.n_classes != 0) return false;
    if (!(n.xyz_div > 0.0f)) return false;
    if (n.precision != INERF_PREC_F32 && n.precision != INERF_PREC_F16X3) return false;
    return true;
}

}  // namespace inerf
