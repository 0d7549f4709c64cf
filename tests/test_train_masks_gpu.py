"""ReLU-mask words of the training forward (layout.h relu_bits_offset) and what the input-gradient chain does with them."""
import numpy as np
import pytest
import torch

import oracle
from intrinsicnerf_amd import _capi, kernels, packing

pytestmark = pytest.mark.gpu


def _slot_range(desc, p, first_slot, last_slot):
    """Element range of the slots first_slot .. last_slot (inclusive) of an activation / gradient buffer."""
    import ctypes as C
    off, width = C.c_int64(), C.c_int()
    lib = _capi.lib()
    lib.inerf_mlp_save_slot(desc, first_slot, p, C.byref(off), C.byref(width))
    first = off.value
    lib.inerf_mlp_save_slot(desc, last_slot, p, C.byref(off), C.byref(width))
    return first, off.value + (p + 63) // 64 * 64 * width.value


def _rays(n, s, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
    d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
    rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
    z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0].to(dev)
    return rays, z


@pytest.mark.parametrize("variant,classes,endpoint,n,s", [("object", 0, False, 37, 19),      # 703 points: a ragged last tile
                                                           ("ssr", 28, False, 64, 5),
                                                           ("ssr", 5, True, 21, 7),           # one-workgroup training forward
                                                           ("object", 0, False, 700, 37)])    # more tiles than workgroups
@pytest.mark.parametrize("form", ["dual", "single"])
def test_mask_words_equal_saved_activations_and_gate_the_chain(variant, classes, endpoint, n, s, form, monkeypatch):
    monkeypatch.setenv("INERF_DGRAD_KERNEL", form)     # two workgroups per CU (default) | the eight-wave single-tile chain
    dev = torch.device("cuda:0")
    ssr = variant == "ssr"
    desc = _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, classes, 10, 4, 10.0 if ssr else 1.0, _capi.PREC_F16X3)
    sd = {k: v.to(dev) for k, v in oracle.make_state_dict(variant, classes, seed=3).items()}
    pf, pb = packing.device_packer(desc, False, dev)(sd), packing.device_packer(desc, True, dev)(sd)
    rays, z = _rays(n, s, dev)
    raw, save = kernels.encode_mlp_train(desc, pf, rays, z, endpoint=endpoint)
    p = n * s
    X = kernels.save_slot_views(desc, save, p)
    tiles = (p + 63) // 64
    sc = kernels.SAVE_SCALARS                          # (the mask area sits between the slots and the buffer's scalars)
    words = save[-(tiles * 4096 + sc):-sc].view(torch.int32).view(tiles, 8, 4, 64, 2).cpu().numpy().astype(np.uint32)
    t, w, l, rb, pb_, g, i = np.meshgrid(np.arange(tiles), np.arange(4), np.arange(64), np.arange(2), np.arange(2), np.arange(4), np.arange(4), indexing="ij")
    chan = 64 * w + 32 * rb + 8 * g + 4 * (l >> 5) + i
    point = 64 * t + 32 * pb_ + (l & 31)
    ok = point < p
    for layer in range(8):
        h = X[kernels.SAVE_H0 + layer].cpu().numpy()
        bit = (words[t, layer, w, l, rb] >> (31 - (16 * pb_ + 4 * g + i))) & 1
        hv = h[np.minimum(point, p - 1), chan]
        want = hv > 0
        # (the fragments keep 8 h to 2^-25 absolute: an activation below 4e-9 decodes to 0 while its mask bit is set)
        flushed = (bit == 1) & (hv == 0)
        assert not ((bit != want) & ok & ~flushed).any(), f"layer {layer}: mask bits differ from (h > 0)"
        assert int((flushed & ok).sum()) <= 3
        assert 0.05 < want[ok].mean() < 0.95
    ch = raw.shape[-1]
    d_raw = torch.randn(p, ch, device=dev)
    dz = kernels.mlp_backward_inputs(desc, pb, raw.view(p, ch), d_raw, save, endpoint=endpoint)
    G = kernels.save_slot_views(desc, dz, p, gradient=True)
    for layer in range(8):
        gz = G[kernels.SAVE_H0 + layer]
        if layer < 7 or form == "dual":       # closed = the forward's mask bit (a decoded h of 0 may be an activation below the fragments' 4e-9 floor)
            closed = np.ones((p, 256), dtype=bool)
            bit = (words[t, layer, w, l, rb] >> (31 - (16 * pb_ + 4 * g + i))) & 1
            closed[np.minimum(point, p - 1)[ok], chan[ok]] = bit[ok] == 0
            closed = torch.from_numpy(closed).to(dev)
        else:
            closed = X[kernels.SAVE_H0 + 7] <= 0           # the eight-wave chain masks d h7 with h7's fragments themselves
        assert not ((gz != 0) & closed).any(), f"dZ of layer {layer} leaks through a closed ReLU"
        assert torch.isfinite(gz).all() and float(gz.abs().max()) > 0


@pytest.mark.parametrize("n,s,launches", [(700, 48, 400), (2048, 192, 300)])
def test_chain_repeats_bit_for_bit_over_many_launches(n, s, launches):
    """Hundreds of launches of the chain on the same inputs: every slot identical every time - on 525 tiles (two per workgroup) and on
    the training step's own fine batch, 2048 x 192 = 6144 tiles, twelve per workgroup: the most sensitive detector of the round-5
    corruption (a development form of k_mlp_dgrad_dual whose non-first tiles got rows 48..63 of the per-point scratch wrong differed
    in 149 of 149 launches there, in 2 % of them on the small shape).  (A 16-byte buffer store with its SGPR
    offset in a register gets no wait state before its data registers are overwritten; the last row of the two VALU stages
    came out wrong in 4 % of the launches of the eight-wave form until the offset moved into the VGPR operand.)  The fragment
    slots are compared as the bytes they are."""
    dev = torch.device("cuda:0")
    desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, 1.0, _capi.PREC_F16X3)
    sd = {k: v.to(dev) for k, v in oracle.lcg_state_dict("object", 0, seed=23, sigma_gain_log2=3, freq_decay=True).items()}
    pf, pb = packing.device_packer(desc, False, dev)(sd), packing.device_packer(desc, True, dev)(sd)
    rays, z = _rays(n, s, dev, seed=5)
    p = n * s
    cot = torch.randn(p, 11, device=dev)
    raw, save = kernels.encode_mlp_train(desc, pf, rays, z)
    ref = None
    for it in range(launches):           # (round 5: a form of the two-workgroup chain that differed in 2 % of its launches passed 120 more than once)
        dz, heads = kernels.mlp_backward_inputs(desc, pb, raw.view(p, 11), cot, save, want_heads=True)
        views = kernels.save_slot_views(desc, dz, p, gradient=True)
        first, last = _slot_range(desc, p, kernels.SAVE_H0, kernels.SAVE_FEAT)
        cur = [dz[first:last].view(torch.int32).clone(), views[kernels.SAVE_VH].clone(), views[kernels.SAVE_DPRE].clone(), heads.clone()]
        if ref is None:
            ref = cur
            continue
        for slot, (a, b) in enumerate(zip(cur, ref)):
            assert torch.equal(a, b), f"launch {it}: part {slot} differs at {int((a != b).sum())} elements"


@pytest.mark.parametrize("variant,classes,endpoint,n,s", [("object", 0, False, 700, 37), ("ssr", 28, False, 300, 23), ("ssr", 5, True, 64, 9),
                                                          ("object", 0, False, 2048, 192)])
def test_two_workgroup_chain_equals_the_eight_wave_chain(variant, classes, endpoint, n, s, monkeypatch):
    """k_mlp_dgrad_dual (4 waves x 64 channels, in place, two tiles per CU) against k_mlp_dgrad (8 waves, A/B buffers): the same
    GEMMs in the same k order and the same epilogue arithmetic, so every dZ slot holds the same f16 VALUES - except where h7 sits below the
    fragments' 4e-9 floor (the dual form masks d h7 with the forward's bit, the eight-wave form with the decoded value) - and the
    1-4-row heads' sums agree to summation order."""
    dev = torch.device("cuda:0")
    ssr = variant == "ssr"
    desc = _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, classes, 10, 4, 10.0 if ssr else 1.0, _capi.PREC_F16X3)
    sd = {k: v.to(dev) for k, v in oracle.make_state_dict(variant, classes, seed=7).items()}
    pf, pb = packing.device_packer(desc, False, dev)(sd), packing.device_packer(desc, True, dev)(sd)
    rays, z = _rays(n, s, dev, seed=2)
    raw, save = kernels.encode_mlp_train(desc, pf, rays, z, endpoint=endpoint)
    p = n * s
    ch = raw.shape[-1]
    d_raw = torch.randn(p, ch, device=dev)
    out = {}
    for form in ("single", "dual"):
        monkeypatch.setenv("INERF_DGRAD_KERNEL", form)
        dz_max = torch.zeros(1, device=dev)
        dz, heads = kernels.mlp_backward_inputs(desc, pb, raw.view(p, ch), d_raw, save, endpoint=endpoint, dz_max=dz_max, want_heads=True)
        torch.cuda.synchronize()
        # the slots the chain writes (the buffer's unused slots - DIR, most of ENC - are whatever the allocator left there)
        first, last = _slot_range(desc, p, kernels.SAVE_H0, kernels.SAVE_SEMH if ssr else kernels.SAVE_VH)
        dpre = _slot_range(desc, p, kernels.SAVE_DPRE, kernels.SAVE_DPRE)
        norm = _slot_range(desc, p, kernels.SAVE_ENC, kernels.SAVE_ENC)[0]
        words = torch.cat([dz[first:last], dz[dpre[0]:dpre[0] + 8 * p], dz[norm:norm + (p + 63) // 64 * 64]]).view(torch.int32).clone()
        out[form] = (words, heads.sum(0).clone(), float(dz_max), heads.shape[0])
    assert out["dual"][3] in (out["single"][3], 2 * out["single"][3], (p + 63) // 64)
    # as f16 halves, -0 taken as +0: a remainder lo that underflows keeps its sign in a register (the eight-wave chain stores the two
    # VALU stages' fragments straight from registers), not through the matrix core (0 x 1 + 0 = +0), where every slot of the
    # two-workgroup chain passes on its way out of the planes
    def halves(w):
        h = w.view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(h == 0x8000, torch.zeros_like(h), h)
    a, b = halves(out["single"][0]), halves(out["dual"][0])
    differ = int((a != b).sum())
    if differ > 64:
        # A floor case flips one mask bit of d h7 at ONE point and the dense layers below carry it into every channel of that point's
        # dZ_h6 .. dZ_h0: thousands of words, but of ONE tile, and only in the trunk slots.  So on a large batch the bound is on TILES:
        # (a corrupted LDS row - the round-5 development form - hit every non-first tile of every workgroup.)
        where = (a != b).nonzero().flatten() // 2                                 # word index within `words`
        f256, l256 = _slot_range(desc, p, kernels.SAVE_H0, kernels.SAVE_H0 + 7)        # the eight trunk slots open `words`
        n_tiles = (p + 63) // 64
        words_per_slot = n_tiles * 16384                                           # 64 points x 256 channels x 4 bytes per tile
        assert l256 - f256 == 8 * words_per_slot
        outside = int((where >= l256 - f256).sum())
        tiles = ((where[where < l256 - f256] % words_per_slot) // 16384).unique()
        assert outside == 0 and tiles.numel() <= max(2, n_tiles // 1000), \
            f"{differ} gradient words differ between the two chains: {tiles.numel()} of {n_tiles} tiles, {outside} words outside the trunk slots"
    assert out["single"][2] == pytest.approx(out["dual"][2], rel=1e-6)
    hs, hd = out["single"][1], out["dual"][1]
    scale = float(hs.abs().max())
    assert float((hs - hd).abs().max()) <= 2e-5 * scale, float((hs - hd).abs().max()) / scale


@pytest.mark.parametrize("n,s", [(37, 19), (700, 37), (2048, 24), (1, 1), (2, 64), (3, 43)])
def test_training_forward_on_the_128_point_tile_writes_the_same_buffer(n, s, monkeypatch):
    """INERF_TRAIN_FWD=t128 (k_encode_mlp_f16x3_t128<false, true>): raw, every fragment slot, the ReLU mask words and act_max are the
    64-point kernel's bit for bit - on a zeroed buffer, so that what either form leaves unwritten compares too (703 points = eleven
    64-point tiles: the sixth 128-point tile has no second half)."""
    import ctypes as C
    dev = torch.device("cuda:0")
    desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, 1.0, _capi.PREC_F16X3)
    sd = {k: v.to(dev) for k, v in oracle.make_state_dict("object", 0, seed=5).items()}
    pf = packing.device_packer(desc, False, dev)(sd)
    rays, z = _rays(n, s, dev, seed=n)
    lib = _capi.lib()
    out = {}
    for form in ("dual", "t128"):
        if form == "t128":
            monkeypatch.setenv("INERF_TRAIN_FWD", "t128")
        else:
            monkeypatch.delenv("INERF_TRAIN_FWD", raising=False)
        raw = torch.zeros(n, s, 11, device=dev)
        save = torch.zeros(lib.inerf_mlp_save_floats(desc, n * s), device=dev)
        act_max = torch.zeros(1, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = lib.inerf_encode_mlp_train(desc, C.c_void_p(pf.data_ptr()), C.c_void_p(rays.data_ptr()), C.c_void_p(z.data_ptr()), n, s, 0,
                                        C.c_void_p(raw.data_ptr()), C.c_void_p(save.data_ptr()), C.c_void_p(act_max.data_ptr()),
                                        C.c_void_p(status.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _capi.check(rc, "inerf_encode_mlp_train")
        torch.cuda.synchronize()
        out[form] = (raw, save.view(torch.int32), act_max, status)
    assert torch.equal(out["dual"][0], out["t128"][0]), "raw differs"
    diff = (out["dual"][1] != out["t128"][1]).nonzero().flatten()
    assert diff.numel() == 0, f"{diff.numel()} words of the save buffer differ, first at {int(diff[0])} of {out['dual'][1].numel()}"
    assert torch.equal(out["dual"][2], out["t128"][2]) and float(out["dual"][2]) > 0
    assert torch.equal(out["dual"][3], out["t128"][3])
