"""CPU: the C-ABI library loads, exports every symbol include/inerf.h declares, validates its arguments,
and the host-side weight packer produces exactly the fragment layout the kernel documents.
No compute entry point is launched here (there is no GPU in this suite)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import oracle
from conftest import REPO


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__
    __graft_entry__.build()
    from intrinsicnerf_amd import _capi
    return _capi


def test_header_and_binding_agree(capi):
    """Every function declared in include/inerf.h is exported by libinerf.so and bound in _capi.SYMBOLS."""
    text = open(os.path.join(REPO, "include", "inerf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(inerf_[a-z_0-9]+)\s*\(", text))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    lib = capi.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.inerf_version().startswith(b"inerf")


def test_struct_layouts_match_header(capi):
    # field order of the ctypes mirrors = field order of the C structs (checked by name against the header)
    text = open(os.path.join(REPO, "include", "inerf.h")).read()
    body = text[text.index("typedef struct inerf_render_args"):text.index("} inerf_render_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"([a-z_]+)\s*;", body)
    assert names == [f[0] for f in capi.RenderArgs._fields_]
    body = text[text.index("typedef struct inerf_composite_out"):text.index("} inerf_composite_out;")]
    names = re.findall(r"float\*\s*([a-z]+);", body)
    assert names == [f[0] for f in capi.CompositeOut._fields_]
    assert C.sizeof(capi.NetDesc) == 24


def test_tensor_table_matches_reference_state_dicts(capi):
    from intrinsicnerf_amd import packing
    for variant, c in (("object", 0), ("ssr", 28), ("ssr", 0)):
        desc = capi.net_desc(capi.VARIANT_SSR if variant == "ssr" else capi.VARIANT_OBJECT, c, 10, 4, 1.0)
        table = packing.tensor_table(desc)
        spec = oracle.state_dict_spec(variant, c)
        assert [n for n, _ in table] == [n for n, _ in spec]
        for (_, (r, cc)), (_, shape) in zip(table, spec):
            assert (r, cc) == (shape[0], shape[1] if len(shape) == 2 else 0)


def test_argument_validation(capi):
    lib = capi.lib()
    bad = capi.net_desc(7)
    assert lib.inerf_num_tensors(bad) == capi.E_INVALID
    assert lib.inerf_packed_floats(capi.net_desc(capi.VARIANT_OBJECT, 3)) == capi.E_INVALID      # object net has no classes
    assert lib.inerf_packed_floats(capi.net_desc(capi.VARIANT_SSR, 0, 11, 4, 10.0)) == capi.E_INVALID   # multires > 10
    good = capi.net_desc(capi.VARIANT_OBJECT)
    assert lib.inerf_encode_mlp(good, None, None, None, 4, 64, 0, None, None, None) == capi.E_INVALID
    assert lib.inerf_sample_fine(None, None, None, 4, 64, 128, 0, None, None, None, None) == capi.E_INVALID
    assert lib.inerf_workspace_bytes(good, 1024, 64, 128, 0) > 1024 * 192 * 11 * 4
    assert lib.inerf_raw_channels(capi.net_desc(capi.VARIANT_SSR, 28, 10, 4, 10.0), capi.FLAG_ENDPOINT, 1) == 11 + 28 + 128
    assert lib.inerf_raw_channels(capi.net_desc(capi.VARIANT_SSR, 28, 10, 4, 10.0), capi.FLAG_ENDPOINT, 0) == 11 + 28
    assert lib.inerf_packed_floats(capi.net_desc(capi.VARIANT_OBJECT, precision=7)) == capi.E_INVALID     # unknown precision
    with pytest.raises(RuntimeError):
        capi.check(capi.E_UNSUPPORTED, "x")
    # every entry point rejects null pointers / impossible sizes before it touches the device, and accepts empty batches
    import ctypes as C
    ssr28 = capi.net_desc(capi.VARIANT_SSR, 28, 10, 4, 10.0)
    assert lib.inerf_sample_coarse(None, None, None, 4, 64, 0, None, None) == capi.E_INVALID
    assert lib.inerf_sample_pdf(None, None, None, 4, 63, 128, 0, None, None) == capi.E_INVALID
    assert lib.inerf_composite(None, None, None, 3, None, 4, 64, 11, 0, 0, 0, C.byref(capi.CompositeOut()), None) == capi.E_INVALID
    assert lib.inerf_composite_backward(None, None, None, 3, None, 4, 64, 11, 0, 0, 0, C.byref(capi.CompositeOut()), None, None) == capi.E_INVALID
    assert lib.inerf_encode_mlp_train(good, None, None, None, 4, 64, 0, None, None, None, None, None) == capi.E_INVALID
    assert lib.inerf_mlp_backward_inputs(good, None, None, None, None, 256, 0, None, None, None, None, None) == capi.E_INVALID
    assert lib.inerf_mlp_weight_gradient(None, 256, None, 256, 1000, 256, 256, None, None, None, 65536, None) == capi.E_INVALID
    assert lib.inerf_cluster_lookup(None, None, 10, None, None, None, None, None, None, 1, 0, None, None, None) == capi.E_INVALID
    assert lib.inerf_gen_rays(None, 12, None, 1, 4, 4, 1., 1., 2., 2., 0., 1., 0, None, None) == capi.E_INVALID
    assert lib.inerf_gen_rays(None, 12, None, 0, 4, 4, 1., 1., 2., 2., 0., 1., 0, None, None) == capi.OK            # no poses
    assert lib.inerf_frame_to_u8(None, 16, None, None) == capi.E_INVALID and lib.inerf_frame_to_u8(None, 0, None, None) == capi.OK
    for rc in (lib.inerf_sample_coarse(None, None, None, 0, 64, 0, None, None),
               lib.inerf_sample_pdf(None, None, None, 0, 63, 128, 0, None, None),
               lib.inerf_encode_mlp(good, None, None, None, 0, 64, 0, None, None, None),
               lib.inerf_encode_mlp_train(good, None, None, None, 0, 64, 0, None, None, None, None, None),
               lib.inerf_mlp_backward_inputs(good, None, None, None, None, 0, 0, None, None, None, None, None),
               lib.inerf_cluster_lookup(None, None, 0, None, None, None, None, None, None, 1, 0, None, None, None)):
        assert rc == capi.OK
    bits = 8 * 4 * 64 * 2                          # ReLU-mask words per 64-point tile (h0..h7, 4 waves, 64 lanes, 2 words)
    sc = 64                                        # per-evaluation scalars behind the mask area
    per_point = 64 + 32 + 8 * 256 + 256 + 256 + 128 + 8                # (slot 15: width 0)
    assert lib.inerf_mlp_save_floats(good, 64) == 64 * per_point + bits + sc
    assert lib.inerf_mlp_save_floats(ssr28, 64) == 64 * (per_point + 128) + bits + sc
    assert lib.inerf_mlp_save_floats(good, 65) == 128 * per_point + 2 * bits + sc          # whole tiles: slots AND masks
    off, width = C.c_int64(), C.c_int()
    assert lib.inerf_mlp_save_slot(ssr28, 13, 100, C.byref(off), C.byref(width)) == capi.OK and width.value == 128
    assert lib.inerf_mlp_save_slot(good, 13, 100, C.byref(off), C.byref(width)) == capi.OK and width.value == 0
    assert lib.inerf_mlp_save_slot(good, 15, 100, C.byref(off), C.byref(width)) == capi.OK and width.value == 0
    assert off.value == 128 * per_point                                          # slots are sized for whole tiles
    assert lib.inerf_mlp_save_slot(good, 16, 100, C.byref(off), C.byref(width)) == capi.E_INVALID
    # formats: h0..h7, the feature layer, the albedo|shading and the views hidden layers are fragment slots in both buffers, the
    # encodings (64 / 32 channels) only as activations, the semantic hidden layer (128) only as gradients; job shares of a batched
    # launch add up to its grid
    assert [lib.inerf_mlp_save_slot_is_fragment(s, 0) for s in range(16)] == [1, 1] + [1] * 8 + [1, 1, 1] + [0] * 3
    assert [lib.inerf_mlp_save_slot_is_fragment(s, 1) for s in range(16)] == [0, 0] + [1] * 12 + [0] * 2
    assert lib.inerf_mlp_save_slot_is_fragment(16, 0) == capi.E_INVALID
    assert lib.inerf_mlp_weight_gradient_frag(None, None, None, None, 1000, None, None, 65536, None) == capi.E_INVALID
    cols = (C.c_int * 13)(*([256] * 9 + [64] * 2 + [256, 32]))
    rws = (C.c_int * 13)(*([256] * 11 + [128, 128]))
    shares = [lib.inerf_wgrad_frag_rows(393216, 13, rws, cols, j) for j in range(13)]
    assert sum(shares) == lib.inerf_wgrad_frag_grid(393216, 13) and min(shares[:9]) > max(shares[9:]) and shares[11] > shares[9] > shares[12] >= 1
    assert [lib.inerf_wgrad_frag_rows(64, 13, rws, cols, j) for j in range(13)] == [1] * 13 and lib.inerf_wgrad_frag_grid(64, 13) == 13
    assert lib.inerf_wgrad_frag_rows(393216, 1, (C.c_int * 1)(128), (C.c_int * 1)(64), 0) == 0          # not a supported pair
    assert lib.inerf_mlp_weight_gradient_xfrag(None, 128, None, 1000, 128, None, None, None, 32768, None) == capi.E_INVALID
    assert lib.inerf_mlp_weight_gradient_gfrag(None, None, None, 64, 1000, 64, None, None, None, 16384, None) == capi.E_INVALID
    assert lib.inerf_wgrad_grid(0) == 0 and lib.inerf_mlp_backward_grid(64 * 7) == 7
    assert lib.inerf_mlp_head_partial_floats() == 1672


# ---- packer: rebuild each layer's (virtual-k) weight matrix from the blob with the documented formula ----
def _unpack_wide(blob, off, n_out, k_total):
    rb_per_wave, kb_count = n_out // 128, k_total // 8
    w = np.zeros((n_out, k_total), np.float32)
    frag = blob[off: off + n_out * k_total].reshape(4, kb_count, rb_per_wave, 64, 4)
    for wave in range(4):
        for rb in range(rb_per_wave):
            for lane in range(64):
                row = wave * 32 * rb_per_wave + 32 * rb + (lane & 31)
                for kb in range(kb_count):
                    k0 = 8 * kb + 4 * (lane >> 5)
                    w[row, k0:k0 + 4] = frag[wave, kb, rb, lane]
    return w


def _unpack_skinny(blob, off, rbs, k_total):
    kb_count = k_total // 16
    w = np.zeros((16 * rbs, k_total), np.float32)
    frag = blob[off: off + 16 * rbs * k_total].reshape(rbs, kb_count, 64, 4)
    for rb in range(rbs):
        for lane in range(64):
            for kb in range(kb_count):
                k0 = 16 * kb + 4 * (lane >> 4)
                w[16 * rb + (lane & 15), k0:k0 + 4] = frag[rb, kb, lane]
    return w


def _layout(variant, c):
    """Python twin of csrc/layout.h make_layout (float offsets)."""
    off = 0
    slots = {}

    def take(n):
        nonlocal off
        o = off
        off += (n + 3) & ~3
        return o
    # each bias vector is followed by 4 floats of per-GEMM constants ([0] = accumulator -> output factor)
    def wide(name, n_out, k): slots[name] = ("wide", take(n_out * k), take(n_out + 4), n_out, k)
    def skinny(name, rbs, k): slots[name] = ("skinny", take(rbs * 16 * k), take(rbs * 16 + 4), rbs, k)
    for i in range(8):
        wide(f"trunk{i}", 256, 64 if i == 0 else (320 if i == 5 else 256))
    skinny("alpha", 1, 256)
    if variant == "ssr" and c > 0:
        wide("sem1", 128, 256)
        skinny("sem2", (c + 15) // 16, 128)
    wide("as1", 256, 256); skinny("as2", 1, 256); wide("feat", 256, 256); wide("views", 128, 288); skinny("res", 1, 128)
    # register-operand copies of the two output heads (f16 format only; zeros otherwise): weights only
    slots["as2r"] = ("regop", take(32 * 256), slots["as2"][2], 4, 256)
    slots["resr"] = ("regop", take(32 * 128), slots["res"][2], 2, 128)
    if variant == "ssr" and c > 0:       # the two-workgroup kernel's semantic head: sem1 in the skinny format, sem2 as 16-row register operands
        slots["sem1s"] = ("skinny", take(128 * 256), slots["sem1"][2], 8, 256)
        slots["sem2r"] = ("regop16", take(((c + 15) // 16) * 16 * 128), slots["sem2"][2], (c + 15) // 16, 128)
        # ... and its inference form: sem2 as 32-row register operands per 32-class block (the hidden layer split over the waves by channel)
        slots["sem2q"] = ("regop32", take(((c + 31) // 32) * 32 * 128), slots["sem2"][2], (c + 31) // 32, 128)
    return slots, off


@pytest.mark.parametrize("variant,c", [("object", 0), ("ssr", 28), ("ssr", 101), ("ssr", 0)])
def test_packer_layout(capi, variant, c):
    from intrinsicnerf_amd import packing
    desc = capi.net_desc(capi.VARIANT_SSR if variant == "ssr" else capi.VARIANT_OBJECT, c, 10, 4, 10.0 if variant == "ssr" else 1.0,
                         precision=capi.PREC_F32)
    sd = oracle.make_state_dict(variant, c, seed=11)
    blob = packing.pack_state_dict(desc, sd).numpy()
    slots, total = _layout(variant, c)
    assert total == capi.lib().inerf_packed_floats(desc) == blob.shape[0]
    g = lambda k: sd[k].numpy()
    # trunk
    for i in range(8):
        _, ow, ob, n_out, k = slots[f"trunk{i}"]
        w = _unpack_wide(blob, ow, n_out, k)
        ref = g(f"pts_linears.{i}.weight")
        if i == 0:
            assert np.array_equal(w[:, :63], ref) and not w[:, 63].any()
        elif i == 5:      # virtual k = [enc64 | h256], reference columns = [pts63 | h256]
            assert np.array_equal(w[:, :63], ref[:, :63]) and not w[:, 63].any() and np.array_equal(w[:, 64:], ref[:, 63:])
        else:
            assert np.array_equal(w, ref)
        assert np.array_equal(blob[ob:ob + 256], g(f"pts_linears.{i}.bias")) and blob[ob + 256] == 1.0
    # sigma
    _, ow, ob, rbs, k = slots["alpha"]
    w = _unpack_skinny(blob, ow, rbs, k)
    assert np.array_equal(w[0], g("alpha_linear.weight")[0]) and not w[1:].any()
    assert blob[ob] == g("alpha_linear.bias")[0] and not blob[ob + 1: ob + 16].any()
    # albedo/shading fused
    sh1, sh2, rs = (("test_linear1", "test_linear2", "shading_linear") if variant == "object"
                    else ("shading_linear1", "shading_linear2", "residual_linear"))
    _, ow, ob, n_out, k = slots["as1"]
    w = _unpack_wide(blob, ow, n_out, k)
    assert np.array_equal(w[:128], g("albedo_linear1.weight")) and np.array_equal(w[128:], g(sh1 + ".weight"))
    assert np.array_equal(blob[ob:ob + 128], g("albedo_linear1.bias")) and np.array_equal(blob[ob + 128:ob + 256], g(sh1 + ".bias"))
    _, ow, ob, rbs, k = slots["as2"]
    w = _unpack_skinny(blob, ow, rbs, k)
    assert np.array_equal(w[:3, :128], g("albedo_linear2.weight")) and not w[:3, 128:].any()
    assert np.array_equal(w[3, 128:], g(sh2 + ".weight")[0]) and not w[3, :128].any() and not w[4:].any()
    assert np.array_equal(blob[ob:ob + 3], g("albedo_linear2.bias")) and blob[ob + 3] == g(sh2 + ".bias")[0]
    # feature / views / residual
    _, ow, ob, n_out, k = slots["feat"]
    assert np.array_equal(_unpack_wide(blob, ow, n_out, k), g("feature_linear.weight"))
    _, ow, ob, n_out, k = slots["views"]
    w = _unpack_wide(blob, ow, n_out, k)
    ref = g("views_linears.0.weight")
    assert np.array_equal(w[:, :256], ref[:, :256]) and np.array_equal(w[:, 256:283], ref[:, 256:]) and not w[:, 283:].any()
    _, ow, ob, rbs, k = slots["res"]
    w = _unpack_skinny(blob, ow, rbs, k)
    assert np.array_equal(w[:3], g(rs + ".weight")) and not w[3:].any()
    if variant == "ssr" and c > 0:
        _, ow, ob, n_out, k = slots["sem1"]
        assert np.array_equal(_unpack_wide(blob, ow, n_out, k), g("semantic_linear.0.0.weight"))
        _, ow, ob, rbs, k = slots["sem2"]
        w = _unpack_skinny(blob, ow, rbs, k)
        assert np.array_equal(w[:c], g("semantic_linear.1.weight")) and not w[c:].any()
        assert np.array_equal(blob[ob:ob + c], g("semantic_linear.1.bias"))


def _unpack_wide_f16(blob, off, n_out, k_total):
    """Rebuild (hi, lo) matrices from the INERF_PREC_F16X3 fragment layout documented in csrc/layout.h."""
    rb_per_wave, kb_count = n_out // 128, k_total // 16
    halfs = blob[off: off + n_out * k_total].view(np.float16).reshape(4, kb_count, rb_per_wave, 2, 64, 8)
    hi = np.zeros((n_out, k_total), np.float32); lo = np.zeros_like(hi)
    for wave in range(4):
        for rb in range(rb_per_wave):
            for lane in range(64):
                row = wave * 32 * rb_per_wave + 32 * rb + (lane & 31)
                for kb in range(kb_count):
                    k0 = 16 * kb + 8 * (lane >> 5)
                    hi[row, k0:k0 + 8] = halfs[wave, kb, rb, 0, lane]
                    lo[row, k0:k0 + 8] = halfs[wave, kb, rb, 1, lane]
    return hi, lo


def _unpack_skinny_f16(blob, off, rbs, k_total):
    kb_count = k_total // 32
    halfs = blob[off: off + 16 * rbs * k_total].view(np.float16).reshape(rbs, kb_count, 2, 64, 8)
    hi = np.zeros((16 * rbs, k_total), np.float32); lo = np.zeros_like(hi)
    for rb in range(rbs):
        for lane in range(64):
            for kb in range(kb_count):
                k0 = 32 * kb + 8 * (lane >> 4)
                hi[16 * rb + (lane & 15), k0:k0 + 8] = halfs[rb, kb, 0, lane]
                lo[16 * rb + (lane & 15), k0:k0 + 8] = halfs[rb, kb, 1, lane]
    return hi, lo


def _unpack_regop_f16(blob, off, q_per_wave, k_total):
    """(hi, lo)[32 rows, k_total] from the register-operand fragments (csrc/layout.h: as2r / resr): k follows the
    accumulator registers of v_mfma_f32_32x32x16_f16 (regop_chan)."""
    halfs = blob[off: off + 32 * k_total].view(np.float16).reshape(4, q_per_wave, 2, 64, 8)
    hi = np.zeros((32, k_total), np.float32); lo = np.zeros_like(hi)
    for wave in range(4):
        for q in range(q_per_wave):
            for lane in range(64):
                for c in range(8):
                    chan = wave * 16 * q_per_wave + 32 * (q >> 1) + 8 * (2 * (q & 1) + (c >> 2)) + 4 * (lane >> 5) + (c & 3)
                    hi[lane & 31, chan] = halfs[wave, q, 0, lane, c]
                    lo[lane & 31, chan] = halfs[wave, q, 1, lane, c]
    return hi, lo


def test_packer_regop_heads(capi):
    """The register-operand copies of the output heads hold the same split weights as the skinny copies, permuted
    into accumulator order, and cover every hidden channel exactly once; the fp32 format leaves them zero."""
    from intrinsicnerf_amd import packing
    for variant, c in (("object", 0), ("ssr", 5), ("ssr", 70)):
        desc = capi.net_desc(capi.VARIANT_SSR if variant == "ssr" else capi.VARIANT_OBJECT, c, 10, 4, 1.0, precision=capi.PREC_F16X3)
        sd = oracle.make_state_dict(variant, c, seed=13)
        blob = packing.pack_state_dict(desc, sd).numpy()
        slots, total = _layout(variant, c)
        assert blob.shape[0] == total
        for reg, skinny, q, k in (("as2r", "as2", 4, 256), ("resr", "res", 2, 128)):
            hi_r, lo_r = _unpack_regop_f16(blob, slots[reg][1], q, k)
            hi_s, lo_s = _unpack_skinny_f16(blob, slots[skinny][1], 1, k)
            assert np.array_equal(hi_r[:16], hi_s) and np.array_equal(lo_r[:16], lo_s)
            assert not hi_r[16:].any() and not lo_r[16:].any() and hi_r[:3].any()
        if variant == "ssr":
            # sem1s: the semantic hidden layer again, skinny fragments, SAME split halves as the wide copy
            hi_w, lo_w = _unpack_wide_f16(blob, slots["sem1"][1], 128, 256)
            hi_s, lo_s = _unpack_skinny_f16(blob, slots["sem1s"][1], 8, 256)
            assert np.array_equal(hi_w, hi_s) and np.array_equal(lo_w, lo_s) and hi_w.any()
            # sem2r: semantic_linear.1 with k in the order of two stacked 16x16x32 accumulators (layout.h regop16_chan)
            rbs = slots["sem2r"][3]
            hi_s, lo_s = _unpack_skinny_f16(blob, slots["sem2"][1], rbs, 128)
            halfs = blob[slots["sem2r"][1]: slots["sem2r"][1] + 16 * rbs * 128].view(np.float16).reshape(rbs, 4, 2, 64, 8)
            hi_r = np.zeros((16 * rbs, 128), np.float32); lo_r = np.zeros_like(hi_r)
            seen = np.zeros((16 * rbs, 128), np.int32)
            for rb in range(rbs):
                for kb in range(4):
                    for lane in range(64):
                        for cc in range(8):
                            chan = 32 * kb + 16 * (cc >> 2) + 4 * (lane >> 4) + (cc & 3)
                            hi_r[16 * rb + (lane & 15), chan] = halfs[rb, kb, 0, lane, cc]
                            lo_r[16 * rb + (lane & 15), chan] = halfs[rb, kb, 1, lane, cc]
                            seen[16 * rb + (lane & 15), chan] += 1
            assert (seen == 1).all() and np.array_equal(hi_r, hi_s) and np.array_equal(lo_r, lo_s) and hi_r[:c].any()
            # sem2q: the same weights as 32-row register operands, one block per 32 classes, k in accumulator order (regop_chan, Q = 2)
            for rb in range(slots["sem2q"][3]):
                hi_q, lo_q = _unpack_regop_f16(blob, slots["sem2q"][1] + rb * 32 * 128, 2, 128)
                rows = min(32, 16 * rbs - 32 * rb)
                assert np.array_equal(hi_q[:rows], hi_s[32 * rb: 32 * rb + rows]) and np.array_equal(lo_q[:rows], lo_s[32 * rb: 32 * rb + rows])
                assert not hi_q[rows:].any() and not lo_q[max(0, c - 32 * rb):].any()
        desc32 = capi.net_desc(capi.VARIANT_SSR if variant == "ssr" else capi.VARIANT_OBJECT, c, 10, 4, 1.0, precision=capi.PREC_F32)
        blob32 = packing.pack_state_dict(desc32, sd).numpy()
        assert not blob32[slots["as2r"][1]:].any()


def test_packer_f16x3_split(capi):
    """(hi + lo) / 2^kw reproduces every fp32 weight to 22 bits, in the documented fragment order; the constants
    behind each bias vector undo the operand scaling (csrc/layout.h)."""
    from intrinsicnerf_amd import packing
    desc = capi.net_desc(capi.VARIANT_SSR, 28, 10, 4, 10.0, precision=capi.PREC_F16X3)
    sd = oracle.make_state_dict("ssr", 28, seed=12)
    blob = packing.pack_state_dict(desc, sd).numpy()
    slots, total = _layout("ssr", 28)
    assert blob.shape[0] == total
    ACT = 8.0

    def weight_scale(*mats):
        m = max(float(np.abs(x).max()) for x in mats)
        return 2.0 ** (14 - (np.frexp(m)[1]))                              # max|W| * scale in [2^13, 2^14)

    def check(hi_lo, ref, sc):
        hi, lo = hi_lo
        assert np.array_equal(hi, (ref * sc).astype(np.float16).astype(np.float32))      # hi = round-to-nearest f16 of W * 2^kw
        rec = (hi.astype(np.float64) + lo.astype(np.float64)) / sc
        assert np.max(np.abs(rec - ref)) <= 2.0 ** -21 * np.max(np.abs(ref))

    ref = sd["pts_linears.3.weight"].numpy()
    _, ow, ob, n_out, k = slots["trunk3"]
    sc = weight_scale(ref)
    assert 2 ** 13 <= np.abs(ref).max() * sc < 2 ** 14
    check(_unpack_wide_f16(blob, ow, n_out, k), ref, sc)
    assert np.array_equal(blob[ob:ob + 256], sd["pts_linears.3.bias"].numpy() * ACT)     # hidden-layer biases live in the scaled domain
    assert blob[ob + 256] == np.float32(1.0 / sc)
    _, ow, ob, n_out, k = slots["trunk5"]
    hi, lo = _unpack_wide_f16(blob, ow, n_out, k)
    ref = sd["pts_linears.5.weight"].numpy()
    sc = weight_scale(ref)
    check((hi[:, 64:], lo[:, 64:]), ref[:, 63:], sc)
    check((hi[:, :63], lo[:, :63]), ref[:, :63], sc)
    assert not hi[:, 63].any() and not lo[:, 63].any()
    _, ow, ob, n_out, k = slots["views"]
    hi, lo = _unpack_wide_f16(blob, ow, n_out, k)
    ref = sd["views_linears.0.weight"].numpy()
    check((hi[:, :283], lo[:, :283]), ref, weight_scale(ref))
    assert not hi[:, 283:].any()
    _, ow, ob, rbs, k = slots["sem2"]
    hi, lo = _unpack_skinny_f16(blob, ow, rbs, k)
    ref = sd["semantic_linear.1.weight"].numpy()
    sc = weight_scale(ref)
    check((hi[:28], lo[:28]), ref, sc)
    assert not hi[28:].any() and not lo[28:].any()
    assert np.array_equal(blob[ob:ob + 28], sd["semantic_linear.1.bias"].numpy())        # output heads: unscaled bias ...
    assert blob[ob + 16 * rbs] == np.float32(1.0 / (sc * ACT))                           # ... and a factor that also undoes the activation scale
    _, ow, ob, rbs, k = slots["as2"]
    hi, lo = _unpack_skinny_f16(blob, ow, rbs, k)
    wa, ws = sd["albedo_linear2.weight"].numpy(), sd["shading_linear2.weight"].numpy()
    sc = weight_scale(wa, ws)
    check((hi[:3, :128], lo[:3, :128]), wa, sc)
    check((hi[3:4, 128:], lo[3:4, 128:]), ws, sc)


def test_packer_rejects_bad_state_dicts(capi):
    from intrinsicnerf_amd import packing
    desc = capi.net_desc(capi.VARIANT_OBJECT)
    sd = oracle.make_state_dict("object", 0, seed=1)
    bad = dict(sd); bad.pop("alpha_linear.bias")
    with pytest.raises(KeyError):
        packing.pack_state_dict(desc, bad)
    bad = dict(sd); bad["pts_linears.5.weight"] = torch.zeros(256, 256)
    with pytest.raises(ValueError):
        packing.pack_state_dict(desc, bad)
    with pytest.raises(KeyError):       # an SSR checkpoint is not an object-level one
        packing.pack_state_dict(desc, oracle.make_state_dict("ssr", 5, seed=1))


def test_reduced_encoding_widths_pack(capi):
    """multires < 10 / multires_views < 4: narrower first-layer inputs land in the same padded columns."""
    from intrinsicnerf_amd import packing
    desc = capi.net_desc(capi.VARIANT_OBJECT, 0, 6, 2, 1.0, precision=capi.PREC_F32)
    table = dict(packing.tensor_table(desc))
    assert table["pts_linears.0.weight"] == (256, 39) and table["pts_linears.5.weight"] == (256, 295)
    assert table["views_linears.0.weight"] == (128, 271)
    sd = {k: torch.randn(r, c) if c else torch.randn(r) for k, (r, c) in table.items()}
    blob = packing.pack_state_dict(desc, sd).numpy()
    slots, _ = _layout("object", 0)
    w = _unpack_wide(blob, slots["trunk5"][1], 256, 320)
    assert np.array_equal(w[:, :39], sd["pts_linears.5.weight"].numpy()[:, :39]) and not w[:, 39:64].any()
    assert np.array_equal(w[:, 64:], sd["pts_linears.5.weight"].numpy()[:, 39:])


@pytest.mark.parametrize("variant,c", [("object", 0), ("ssr", 28), ("ssr", 0)])
def test_device_packer_is_bit_identical_to_host_packer(capi, variant, c):
    """packing.DevicePacker (torch ops driven by inerf_pack_map; used after every optimiser step of a training run)
    reproduces inerf_pack_weights and inerf_pack_weights_bwd bit for bit - here on CPU tensors."""
    from intrinsicnerf_amd import packing
    desc = capi.net_desc(capi.VARIANT_SSR if variant == "ssr" else capi.VARIANT_OBJECT, c, 10, 4, 10.0 if variant == "ssr" else 1.0,
                         precision=capi.PREC_F16X3)
    for seed in (3, 4):
        sd = oracle.make_state_dict(variant, c, seed=seed)
        if seed == 4:
            sd["pts_linears.2.weight"] = sd["pts_linears.2.weight"] * 0.0          # an all-zero GEMM: scale 1
            sd["feature_linear.weight"] = sd["feature_linear.weight"] * 1000.0     # moves the common scale of the d h7 group
        for backward in (False, True):
            host = packing.pack_state_dict_bwd(desc, sd) if backward else packing.pack_state_dict(desc, sd)
            dev = packing.device_packer(desc, backward, "cpu")(sd)
            assert torch.equal(host.view(torch.int32), dev.view(torch.int32)), (variant, c, seed, backward)


@pytest.mark.parametrize("variant,c", [("object", 0), ("ssr", 28), ("ssr", 101)])
def test_device_packer_f32_is_bit_identical_to_the_host_packer(capi, variant, c):
    """packing.DevicePackerF32 (the exact-fp32 blob as one gather; used by the INERF_PRECISION=f32 training forward, which
    re-packs after every optimiser step) against inerf_pack_weights on real weights - the fp32 format must be a pure
    permutation + zero padding, with no derived constants."""
    from intrinsicnerf_amd import packing
    desc = capi.net_desc(capi.VARIANT_SSR if variant == "ssr" else capi.VARIANT_OBJECT, c, 10, 4, 10.0 if variant == "ssr" else 1.0,
                         precision=capi.PREC_F32)
    packer = packing.DevicePackerF32(desc, "cpu")
    for seed in (3, 4):
        sd = oracle.make_state_dict(variant, c, seed=seed)
        host = packing.pack_state_dict(desc, sd)
        assert torch.equal(host.view(torch.int32), packer(sd).view(torch.int32)), (variant, c, seed)


def test_abi_version_and_stale_library_guard(capi, monkeypatch, tmp_path):
    """The binding refuses a library whose ABI number differs from the one its ctypes mirrors were written for, and a
    library built from other sources / headers / compiler flags than the tree holds is rebuilt (or refused when there is
    no hipcc) instead of being loaded silently.  Staleness is a CONTENT digest linked into the library
    (inerf_build_digest), not a comparison of mtimes."""
    text = open(os.path.join(REPO, "include", "inerf.h")).read()
    assert int(re.search(r"#define INERF_ABI_VERSION (\d+)", text).group(1)) == capi.ABI_VERSION
    assert capi.lib().inerf_abi_version() == capi.ABI_VERSION
    from intrinsicnerf_amd import _build
    # the digest inside the library is the tree's, and reading it from the file's bytes agrees with calling it
    assert capi.lib().inerf_build_digest().decode() == _build.source_digest() == _build.built_digest()
    assert re.fullmatch(r"[0-9a-f]{64}", _build.built_digest()) and not _build._stale()
    # reordered mtimes do not matter ...
    src = os.path.join(_build.CSRC, "ray_ops.hip")
    st = os.stat(src)
    try:
        os.utime(src, (st.st_atime, st.st_mtime + 1e6))
        assert not _build._stale()
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    # ... a changed compiler flag or source list does (editing EXTRA_FLAGS reused the old library)
    monkeypatch.setattr(_build, "EXTRA_FLAGS", {**_build.EXTRA_FLAGS, "mlp.hip": ["-O1"]})
    assert _build._stale()
    monkeypatch.undo()
    assert not _build._stale()
    # a file without the digest (or no file) is stale
    junk = tmp_path / "libjunk.so"
    junk.write_bytes(b"\x7fELF no digest here")
    assert _build.built_digest(str(junk)) == "" and _build._stale(str(junk)) and _build._stale(str(tmp_path / "missing.so"))

    # a tree that holds only SOME of the library's sources is broken, not a binary deployment: no digest can be computed for it and
    # whatever library lies there must not be accepted in its name; with none of them an existing library is taken as is
    real_exists = os.path.exists
    gone = os.path.join(_build.CSRC, "mlp_bwd.hip")
    monkeypatch.setattr(_build.os.path, "exists", lambda q: False if q == gone else real_exists(q))
    with pytest.raises(RuntimeError, match="incomplete.*mlp_bwd.hip"):
        _build._stale()
    every = set(_build._source_paths())
    monkeypatch.setattr(_build.os.path, "exists", lambda q: False if q in every else real_exists(q))
    assert not _build.sources_present() and not _build._stale()
    # a binary deployment naturally keeps the public C-ABI header next to the library: csrc/ alone decides
    csrc_only = {q for q in every if os.path.dirname(q) == _build.CSRC}
    assert len(every - csrc_only) == 1
    monkeypatch.setattr(_build.os.path, "exists", lambda q: False if q in csrc_only else real_exists(q))
    assert not _build.sources_present() and not _build._stale()
    monkeypatch.undo()

    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "ABI_VERSION", capi.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI"):
        capi.lib()
    monkeypatch.setattr(capi, "ABI_VERSION", capi.ABI_VERSION - 1)
    monkeypatch.setattr(_build, "_stale", lambda *a: True)
    monkeypatch.setattr(_build, "have_hipcc", lambda: False)
    with pytest.raises(RuntimeError, match="built from different sources"):
        capi.lib()
    monkeypatch.undo()
    assert capi.lib().inerf_abi_version() == capi.ABI_VERSION


def test_workspace_skips_caller_provided_stage_tensors(capi):
    """inerf_render_workspace_bytes: stage tensors the caller supplies as outputs get no workspace region (raw was
    held twice - ~6 GB per chunk at C = 101 with the endpoint feature)."""
    lib = capi.lib()
    c, n, sc, ni = 101, 32768, 64, 128
    desc = capi.net_desc(capi.VARIANT_SSR, c, 10, 4, 10.0)
    full = lib.inerf_workspace_bytes(desc, n, sc, ni, capi.FLAG_ENDPOINT)
    a = capi.RenderArgs()
    a.net, a.n_rays, a.n_samples, a.n_importance, a.flags = desc, n, sc, ni, capi.FLAG_ENDPOINT
    assert lib.inerf_render_workspace_bytes(C.byref(a)) == full
    up = lambda floats: (floats * 4 + 255) // 256 * 256
    raw_c, raw_f = up(n * sc * (11 + c)), up(n * (sc + ni) * (11 + c + 128))
    # the encode+MLP launches' own scratch (the SSR network's channel-split semantic head; the fine pass with the endpoint
    # feature runs the one-workgroup kernel and needs none) stays in the workspace whatever the caller provides
    mlp_ws = max(lib.inerf_encode_mlp_workspace_bytes(desc, n, sc, 0), lib.inerf_encode_mlp_workspace_bytes(desc, n, sc + ni, capi.FLAG_ENDPOINT))
    assert lib.inerf_encode_mlp_workspace_bytes(desc, n, sc + ni, capi.FLAG_ENDPOINT) == 0 and mlp_ws % 256 == 0
    assert mlp_ws == 0                                               # C = 101 > 32: the per-wave head is the faster one, no scratch
    d28 = capi.net_desc(capi.VARIANT_SSR, 28, 10, 4, 10.0)
    ws28 = lib.inerf_encode_mlp_workspace_bytes(d28, n, sc, 0)
    assert ws28 > 0 and ws28 % 32768 == 0                            # C <= 32: one 32 KiB block per workgroup
    a28 = capi.RenderArgs()
    a28.net, a28.n_rays, a28.n_samples, a28.n_importance = d28, n, sc, 0
    assert lib.inerf_render_workspace_bytes(C.byref(a28)) == up(n * sc) + up(n * sc * (11 + 28)) + ws28
    a.raw_coarse, a.raw_fine = 0x1000, 0x2000                       # "provided" (never dereferenced here)
    assert lib.inerf_render_workspace_bytes(C.byref(a)) == full - raw_c - raw_f
    a.z_coarse, a.z_samples, a.z_fine = 0x10, 0x20, 0x30
    a.coarse.weights = 0x40
    assert lib.inerf_render_workspace_bytes(C.byref(a)) == mlp_ws
    # coarse-only: no resampling, no weights region
    b = capi.RenderArgs()
    b.net, b.n_rays, b.n_samples, b.n_importance = desc, n, sc, 0
    assert lib.inerf_render_workspace_bytes(C.byref(b)) == up(n * sc) + up(n * sc * (11 + c)) + mlp_ws
    obj = capi.net_desc(capi.VARIANT_OBJECT, 0, 10, 4, 1.0)
    # the object-level network needs no scratch by default; INERF_ENC_CACHE=1 asks for the slot of the parked position encoding (32 KiB per
    # workgroup of the 128-point tile, f16x3 only)
    assert lib.inerf_encode_mlp_workspace_bytes(obj, n, sc + ni, 0) == 0
    os.environ["INERF_ENC_CACHE"] = "1"
    try:
        ws_obj = lib.inerf_encode_mlp_workspace_bytes(obj, n, sc + ni, 0)
        assert (ws_obj > 0 and ws_obj % 32768 == 0) if obj.precision == capi.PREC_F16X3 else ws_obj == 0
        obj32 = capi.net_desc(capi.VARIANT_OBJECT, 0, 10, 4, 1.0, capi.PREC_F32)
        assert lib.inerf_encode_mlp_workspace_bytes(obj32, n, sc + ni, 0) == 0
    finally:
        del os.environ["INERF_ENC_CACHE"]
    assert lib.inerf_render_workspace_bytes(None) == capi.E_INVALID


@pytest.mark.parametrize("variant,c", [("object", 0), ("ssr", 0), ("ssr", 28), ("ssr", 150)])
def test_one_call_backward_sizes(capi, variant, c):
    """Host-side contract of inerf_mlp_backward (csrc/train_api.hip): the gradient blob is the reference's parameter tensors in
    inerf_tensor_info order; the workspace holds the pre-activation gradients of every layer (11 KB per point), the partial
    tiles of every product and - SSR with classes - the padded logit gradients; invalid arguments are refused without a GPU."""
    from intrinsicnerf_amd import packing
    lib = capi.lib()
    desc = capi.net_desc(capi.VARIANT_SSR if variant == "ssr" else capi.VARIANT_OBJECT, c, 10, 4, 1.0, precision=capi.PREC_F16X3)
    table = packing.tensor_table(desc)
    n_params = sum(r * (cc if cc else 1) for _, (r, cc) in table)
    assert lib.inerf_param_floats(desc) == n_params == sum(v.numel() for v in oracle.make_state_dict(variant, c, seed=0).values())
    assert lib.inerf_mlp_backward_workspace_bytes(desc, 0) == 0
    p = 64 * 300
    ws = lib.inerf_mlp_backward_workspace_bytes(desc, p)
    save = lib.inerf_mlp_save_floats(desc, p) * 4
    assert ws >= save + (p * 128 * 4 if c > 0 else 0) and ws % 256 == 0
    assert lib.inerf_mlp_backward_workspace_bytes(desc, 2 * p) > ws
    assert lib.inerf_mlp_backward_workspace_bytes(None, p) == capi.E_INVALID
    # one kept evaluation is limited to 4 000 000 points (32-bit buffer descriptors; include/inerf.h): refused, not wrapped around
    assert lib.inerf_mlp_backward_workspace_bytes(desc, 4_000_000) > 0
    assert lib.inerf_mlp_backward_workspace_bytes(desc, 4_000_001) == capi.E_UNSUPPORTED
    assert lib.inerf_mlp_backward(desc, C.cast(one4 := (C.c_float * 4)(), C.c_void_p), C.cast(one4, C.c_void_p), C.cast(one4, C.c_void_p),
                                  C.cast(one4, C.c_void_p), C.cast(one4, C.c_void_p), 4_000_001, 0, C.cast(one4, C.c_void_p), None, 0,
                                  None, None) == capi.E_UNSUPPORTED
    assert lib.inerf_mlp_backward(None, None, None, None, None, None, p, 0, None, None, 0, None, None) == capi.E_INVALID
    one = (C.c_float * 4)()
    # everything but the gradient blob missing: refused before anything is launched
    assert lib.inerf_mlp_backward(desc, None, None, None, None, None, p, 0, C.cast(one, C.c_void_p), None, 0, None, None) == capi.E_INVALID
    # the weight-gradient kernel reads its rows through 32-bit descriptors: a matrix beyond 4 GiB is refused, not wrapped around
    assert lib.inerf_mlp_weight_gradient(C.cast(one4, C.c_void_p), 256, C.cast(one4, C.c_void_p), 256, 4_200_000, 256, 256,
                                         C.cast(one4, C.c_void_p), C.cast(one4, C.c_void_p), None, 256 * 256, None) == capi.E_UNSUPPORTED
    # param_views cuts the blob into the reference's shapes
    from intrinsicnerf_amd import kernels
    views = kernels.param_views(desc, torch.arange(n_params, dtype=torch.float32))
    assert [tuple(v.shape) for v in views.values()] == [((r, cc) if cc else (r,)) for _, (r, cc) in table]
    assert float(views[table[-1][0]].flatten()[-1]) == n_params - 1
