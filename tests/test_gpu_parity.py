"""GPU parity: the HIP path (through the C ABI) against the golden fixtures (= the real reference's
outputs) and against the CPU oracle on seeded inputs.

Tolerance (BASELINE.json north_star: "within 1e-4 rel fp32"): ``|got - want| <= 1e-5 + 1e-4 |want|``
for every map, raw tensor and stage tensor.  ``disp`` is the one exception: it is 1/(depth/acc), which
amplifies fp32 round-off - the reference's own fp32-vs-fp64 noise floor on it is 8.9e-5 (SURVEY.md
section 6) - so it is held to 5e-4 relative, NaNs required at identical rays.
"""
import numpy as np
import pytest
import torch

import oracle
from _cases import (CDF_NOISE, assert_maps_close, case_config, case_random_inputs, case_weights, injected_np_rand,
                    sample_pdf_sensitivity)
from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["f32", "f16x3", "f16x3-1wg"])
def precision(request, monkeypatch):
    """Every GPU parity test runs against every MLP kernel: exact-fp32 MFMA, the f16 hi/lo split one in its default
    form (object-level network: two workgroups per CU) and in its one-workgroup form (the SSR network always uses it)."""
    monkeypatch.setenv("INERF_PRECISION", request.param.split("-")[0])
    if request.param.endswith("-1wg"):
        monkeypatch.setenv("INERF_F16_KERNEL", "single")
    else:
        monkeypatch.delenv("INERF_F16_KERNEL", raising=False)
    return request.param

RTOL, ATOL, RTOL_DISP = 1e-4, 1e-5, 5e-4


def _dev():
    return torch.device("cuda:0")


def _desc(cfg):
    from intrinsicnerf_amd import _capi
    ssr = cfg.variant == "ssr"
    return _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, cfg.n_classes if ssr else 0,
                          cfg.l_xyz, cfg.l_dir, cfg.xyz_div)


def _packed(cfg, sd):
    from intrinsicnerf_amd import packing
    return packing.pack_state_dict(_desc(cfg), sd).to(_dev())


def _rtol(key):
    return RTOL_DISP if key.startswith("disp") else RTOL


# ------------------------------------------------------------------------------------------------
# whole path vs the reference's outputs
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names("object_") + golden_names("ssr_"))
def test_fused_path_matches_reference(name):
    from intrinsicnerf_amd import kernels
    fx = load_golden(name)
    cfg = case_config(fx)
    sd_c, sd_f = case_weights(fx)
    dev = _dev()
    rnd = case_random_inputs(fx, dev)
    u = rnd.pop("u", None)
    if u is None and cfg.n_importance > 0:
        u = torch.linspace(0.0, 1.0, cfg.n_importance).to(dev)
    out = kernels.render_rays_fused(
        _desc(cfg), _packed(cfg, sd_c), _packed(cfg, sd_f), torch.from_numpy(fx["rays"]).to(dev), 64, cfg.n_importance,
        torch.from_numpy(fx["t_vals"]).to(dev), u, rnd.get("t_rand"), rnd.get("noise_coarse"), rnd.get("noise_fine"),
        white_bkgd=cfg.white_bkgd, lindisp=cfg.lindisp, endpoint=cfg.endpoint_feat,
        want_raw_coarse=True, want_raw_fine=True, want_stages=True)
    torch.cuda.synchronize()
    # new-sample depths may move by (cdf round-off) x (their bin's 1/denom amplification) - see _cases.py
    z_extra = {}
    if cfg.n_importance > 0:
        u_np = fx["in_u"] if "in_u" in fx else np.linspace(0.0, 1.0, cfg.n_importance, dtype=np.float32)
        sens = sample_pdf_sensitivity(fx["stage_z_coarse"], fx["stage_weights_coarse"], u_np)
        z_extra = {"z_samples": CDF_NOISE * sens, "z_fine": CDF_NOISE * sens.max(-1, keepdims=True)}
    checked = 0
    for key, want in fx.items():
        if key.startswith("ref_"):
            k = key[4:]
            got = out[k].cpu().numpy()
            if k.startswith("raw"):
                got = got[: want.shape[0]]
            if k == "raw_fine" and cfg.n_importance > 0:
                continue          # per-sample tensor at resampled depths: checked stage-wise (test_fine_stage_...)
            assert_maps_close(got, want, _rtol(k), ATOL, f"{name}:{k}")
            checked += 1
        elif key.startswith("stage_") and key != "stage_raw_coarse":
            k = key[6:]
            if k == "weights_fine":
                continue          # per-sample tensor at resampled depths: checked stage-wise (test_fine_stage_...)
            assert_maps_close(out[k].cpu().numpy(), want, RTOL, ATOL, f"{name}:stage {k}", extra=z_extra.get(k))
            checked += 1
    assert checked >= 8


@pytest.mark.parametrize("name", [n for n in golden_names("object_") + golden_names("ssr_") if "coarse_only" not in n])
def test_fine_stage_on_reference_depths(name):
    """Stage-wise parity of the fine pass: the HIP MLP and compositing kernels are fed the REFERENCE's merged depths
    (fixture ``z_fine``), so their per-sample outputs (raw_fine, weights_fine) can be held to the plain tolerance -
    in an end-to-end run those tensors sit downstream of sample_pdf's 1/denom amplification."""
    from intrinsicnerf_amd import kernels
    fx = load_golden(name)
    cfg = case_config(fx)
    _, sd_f = case_weights(fx)
    dev = _dev()
    k = fx["ref_raw_fine"].shape[0]                       # rays whose reference raw_fine is stored
    rays = torch.from_numpy(fx["rays"][:k]).to(dev)
    z_fine = torch.from_numpy(fx["stage_z_fine"][:k]).to(dev)
    raw = kernels.encode_mlp(_desc(cfg), _packed(cfg, sd_f), rays, z_fine, endpoint=cfg.endpoint_feat)
    scale = np.abs(fx["ref_raw_fine"]).max(axis=(0, 1), keepdims=True)       # per-channel magnitude (sigma has cancellation)
    assert_maps_close(raw.cpu().numpy(), fx["ref_raw_fine"], RTOL, ATOL, f"{name}:raw_fine", extra=RTOL * scale)
    noise = torch.from_numpy(fx["in_noise_fine"][:k]).to(dev) if "in_noise_fine" in fx else None
    ssr = cfg.variant == "ssr"
    comp = kernels.composite(torch.from_numpy(fx["ref_raw_fine"]).to(dev), z_fine, rays[:, 3:6].contiguous(), noise,
                             cfg.white_bkgd, n_classes=cfg.n_classes if ssr else 0,
                             feat_dim=128 if (ssr and cfg.endpoint_feat) else 0)
    assert_maps_close(comp["weights"].cpu().numpy(), fx["stage_weights_fine"][:k], RTOL, ATOL, f"{name}:weights_fine")
    for key in ("rgb", "acc", "depth", "albedo", "shading", "residual", "sem", "feat"):
        if "ref_" + key + "_fine" in fx and key in comp:
            assert_maps_close(comp[key].cpu().numpy(), fx["ref_" + key + "_fine"][:k], RTOL, ATOL, f"{name}:{key}_fine (stage)")


# ------------------------------------------------------------------------------------------------
# the reference-signature front-ends
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["object_chair_det", "object_chair_train_rng", "object_coarse_only_lindisp"])
def test_object_level_render_rays_frontend(name):
    from intrinsicnerf_amd import object_level as ol
    fx = load_golden(name)
    cfg = case_config(fx)
    sd_c, sd_f = case_weights(fx)
    dev = _dev()
    embed, ch = ol.get_embedder(10, 0)
    embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True)
    net_c, net_f = mk().to(dev), mk().to(dev)
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    q = ol.NetworkQuery(embed, embed_d, 65536)
    train = "in_t_rand" in fx
    rnd = case_random_inputs(fx)
    feed = [rnd[k] for k in ("t_rand", "noise_coarse", "u", "noise_fine") if k in rnd]     # the reference's draw order
    with torch.no_grad(), injected_np_rand(feed):
        ret = ol.render_rays(torch.from_numpy(fx["rays"]).to(dev), net_c, q, 64, retraw=True, lindisp=cfg.lindisp,
                             perturb=1.0 if train else 0.0, N_importance=cfg.n_importance, network_fine=net_f,
                             white_bkgd=cfg.white_bkgd, raw_noise_std=1.0 if train else 0.0, pytest=train)
    lvl = "fine" if cfg.n_importance > 0 else "coarse"
    pairs = [("rgb_map", "rgb_" + lvl), ("disp_map", "disp_" + lvl), ("acc_map", "acc_" + lvl),
             ("albedo_map", "albedo_" + lvl), ("shading_map", "shading_" + lvl), ("residual_map", "residual_" + lvl)]
    expected_keys = {"rgb_map", "disp_map", "acc_map", "albedo_map", "shading_map", "residual_map", "raw"}
    if cfg.n_importance > 0:
        pairs += [("rgb0", "rgb_coarse"), ("disp0", "disp_coarse"), ("acc0", "acc_coarse"), ("albedo0", "albedo_coarse"),
                  ("shading0", "shading_coarse"), ("residual0", "residual_coarse"), ("z_std", "z_std")]
        expected_keys |= {"rgb0", "disp0", "acc0", "albedo0", "shading0", "residual0", "z_std"}
    assert set(ret.keys()) == expected_keys            # run_nerf.py:512-522
    for rk, ok in pairs:
        assert_maps_close(ret[rk].cpu().numpy(), fx["ref_" + ok], _rtol(ok), ATOL, f"{name}:{rk}")
    assert tuple(ret["raw"].shape) == (fx["rays"].shape[0], 64 + cfg.n_importance, 11)


def test_object_level_render_image_api(monkeypatch):
    """render(H, W, K, c2w=...) returns the reference's 7-element list with image-shaped maps; chunking is invisible."""
    monkeypatch.setenv("INERF_COALESCE_BYTES", "0")           # the caller's chunks as given (coalescing: tests/test_coalesce_gpu.py)
    from intrinsicnerf_amd import object_level as ol
    dev = _dev()
    H = W = 24
    focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    c2w = torch.tensor([[-0.7660, 0.3214, -0.5567, -2.2270], [-0.6428, -0.3830, 0.6634, 2.6537],
                        [0.0, 0.8660, 0.5, 2.0]], device=dev)
    ro0, rd0 = ol.get_rays(H, W, K, c2w)
    probe = torch.cat([ro0, rd0, 2 * torch.ones_like(rd0[..., :1]), 6 * torch.ones_like(rd0[..., :1]),
                       rd0 / rd0.norm(dim=-1, keepdim=True)], -1).reshape(-1, 11).cpu()
    sd_c, _ = oracle.calibrated_lcg_weights("object", 0, 20, probe)
    sd_f, _ = oracle.calibrated_lcg_weights("object", 0, 21, probe)
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net_c, net_f = mk(), mk()
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    kw = dict(network_fn=net_c, network_fine=net_f, network_query_fn=ol.NetworkQuery(embed, embed_d), N_samples=64,
              N_importance=128, white_bkgd=True, perturb=False, raw_noise_std=0., use_viewdirs=True, ndc=False, lindisp=False)
    with torch.no_grad():
        full = ol.render(H, W, K, chunk=1 << 15, c2w=c2w, near=2., far=6., **kw)
        small = ol.render(H, W, K, chunk=100, c2w=c2w, near=2., far=6., **kw)
    assert len(full) == 7 and isinstance(full[6], dict)
    assert tuple(full[0].shape) == (H, W, 3) and tuple(full[1].shape) == (H, W) and tuple(full[3].shape) == (H, W, 3)
    for a, b in zip(full[:6], small[:6]):
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))        # chunk-size invariance: bit-exact
    # against the oracle on the same rays
    ro, rd = ol.get_rays(H, W, K, c2w)
    vd = rd / rd.norm(dim=-1, keepdim=True)
    rays = torch.cat([ro, rd, 2 * torch.ones_like(rd[..., :1]), 6 * torch.ones_like(rd[..., :1]), vd], -1).reshape(-1, 11).cpu()
    cfg = oracle.RenderConfig(variant="object", white_bkgd=True)
    t_vals = torch.linspace(0., 1., 64)
    with torch.no_grad():
        want = oracle.render_rays(rays, sd_c, sd_f, cfg, t_vals=t_vals)
    # compare on the rays where the reference arithmetic itself is reproducible (oracle/conditioning.py)
    ok = (oracle.conditioning_scores(rays, sd_c, sd_f, cfg, t_vals) <= 0.2).numpy()
    assert ok.mean() > 0.5
    for idx, key in ((0, "rgb_fine"), (2, "acc_fine"), (3, "albedo_fine"), (4, "shading_fine"), (5, "residual_fine")):
        got = full[idx].reshape(H * W, -1).cpu().numpy()[ok]
        assert_maps_close(got, want[key].reshape(H * W, -1).numpy()[ok], RTOL, ATOL, f"render {key}")


@pytest.mark.parametrize("name", ["ssr_room_det_c28", "ssr_endpoint_c5_wb", "ssr_c101"])
def test_ssr_trainer_render_rays_frontend(name):
    from intrinsicnerf_amd import ssr
    fx = load_golden(name)
    cfg = case_config(fx)
    sd_c, sd_f = case_weights(fx)
    r = ssr.SSRRenderer(cfg.n_classes, white_bkgd=cfg.white_bkgd, endpoint_feat=cfg.endpoint_feat, chunk=7, device=_dev())
    r.ssr_net_coarse.load_state_dict(sd_c); r.ssr_net_fine.load_state_dict(sd_f)
    r.check_numerics = False
    with torch.no_grad():
        ret = r.render_rays(torch.from_numpy(fx["rays"]).to(_dev()))
    keys = {f"{k}_{l}" for l in ("coarse", "fine") for k in ("rgb", "disp", "acc", "depth", "albedo", "shading", "residual")}
    keys |= {"raw_coarse", "raw_fine", "z_std", "sem_logits_coarse", "sem_logits_fine"}
    if cfg.endpoint_feat:
        keys.add("feat_map_fine")
    assert set(ret.keys()) == keys                       # trainer.py:777-802
    ren = {"sem_logits_coarse": "sem_coarse", "sem_logits_fine": "sem_fine", "feat_map_fine": "feat_fine"}
    for k in keys:
        want = fx["ref_" + ren.get(k, k)]
        got = ret[k].cpu().numpy()
        if k.startswith("raw"):
            got = got[: want.shape[0]]
        assert_maps_close(got, want, _rtol(k), ATOL, f"{name}:{k}")


# ------------------------------------------------------------------------------------------------
# stage kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names("stage_composite_"))
def test_composite_edge_cases(name):
    from intrinsicnerf_amd import kernels
    fx = load_golden(name)
    ssr = "ssr" in name
    dev = _dev()
    out = kernels.composite(torch.from_numpy(fx["raw"]).to(dev), torch.from_numpy(fx["z"]).to(dev),
                            torch.from_numpy(fx["rays_d"]).to(dev), None, bool(fx["white_bkgd"]),
                            n_classes=int(fx["n_classes"]) if ssr else 0, feat_dim=128 if ssr else 0)
    for key, want in fx.items():
        if key.startswith("ref_"):
            assert_maps_close(out[key[4:]].cpu().numpy(), want, _rtol(key[4:]), ATOL, f"{name}:{key}")
    if not ssr:
        assert torch.isnan(out["disp"][0]).item() and out["acc"][0].item() == 0.0       # empty ray
        assert out["weights"][1, -1].item() == 1.0                                         # only the 1e10 interval
        assert out["weights"][3, 0].item() == 1.0 and out["weights"][3, 1:].abs().max().item() < 1e-9


def test_sample_pdf_edge_cases():
    from intrinsicnerf_amd import kernels
    fx = load_golden("stage_sample_pdf")
    dev = _dev()
    bins, w = torch.from_numpy(fx["bins"]).to(dev), torch.from_numpy(fx["weights"]).to(dev)
    det = kernels.sample_pdf(bins, w, torch.linspace(0., 1., 128).to(dev), 128)
    assert_maps_close(det.cpu().numpy(), fx["ref_det"], RTOL, ATOL, "sample_pdf det")
    rnd = kernels.sample_pdf(bins, w, torch.from_numpy(fx["u_rnd"]).to(dev), 128)
    # ray 4 has u values sitting exactly on cdf entries: a 1-ulp cdf difference legitimately moves a sample
    # to the neighbouring bin edge, so that ray is compared on the sorted sample set with a bin-width tolerance
    mask = fx["ref_rnd_mask"].astype(bool)
    assert_maps_close(rnd.cpu().numpy()[mask], fx["ref_rnd"][mask], RTOL, ATOL, "sample_pdf rnd")
    assert np.all(np.abs(np.sort(rnd.cpu().numpy()[4]) - np.sort(fx["ref_rnd"][4])) < 0.08)


def test_sample_fine_merge_sorted_and_std():
    from intrinsicnerf_amd import kernels
    fx = load_golden("stage_sample_pdf")
    dev = _dev()
    z = torch.from_numpy(fx["z_coarse"]).to(dev)
    n = z.shape[0]
    g = torch.Generator().manual_seed(4)
    wfull = torch.rand(n, 64, generator=g)
    wfull[3, 10:50] = 0.0                                   # empty bins: many samples collapse onto bin edges (ties)
    bins = 0.5 * (z[:, 1:] + z[:, :-1]).cpu()
    u_ties = torch.linspace(0., 1., 128).clone(); u_ties[40:60] = u_ties[40]
    cases = {"random u (general rank sort)": torch.rand(n, 128, generator=g),
             "ascending u (merge path)": torch.linspace(0., 1., 128),
             "ascending u with ties": u_ties}
    for tag, u in cases.items():
        zs, zm, zstd = kernels.sample_fine(z, wfull.to(dev), u.to(dev), 128)
        want = oracle.inverse_cdf_sample(bins, wfull[:, 1:-1], u if u.dim() == 2 else u.expand(n, 128))
        assert_maps_close(zs.cpu().numpy(), want.numpy(), RTOL, ATOL, f"z_samples, {tag}")
        merged = torch.sort(torch.cat([z.cpu(), zs.cpu()], -1), -1)[0]
        assert torch.equal(zm.cpu(), merged), tag           # a permutation of its own inputs, ascending: bit-exact
        assert_maps_close(zstd.cpu().numpy(), torch.std(want, -1, unbiased=False).numpy(), RTOL, ATOL, f"z_std, {tag}")
    # a NaN among the inputs must not hang or corrupt the other rays (torch.sort puts NaN last)
    zbad = z.clone(); zbad[5, 7] = float("nan")
    zs, zm, _ = kernels.sample_fine(zbad, wfull.to(dev), torch.linspace(0., 1., 128).to(dev), 128)
    ref = torch.sort(torch.cat([zbad.cpu(), zs.cpu()], -1), -1)[0]
    assert torch.equal(torch.nan_to_num(zm.cpu(), nan=-1.0), torch.nan_to_num(ref, nan=-1.0))


@pytest.mark.parametrize("variant,c,s", [("object", 0, 64), ("object", 0, 192), ("ssr", 28, 64), ("ssr", 3, 50)])
def test_encode_mlp_raw(variant, c, s):
    """Stage parity of the dominant kernel on default-init-like weights and awkward sizes (ragged last tile)."""
    from intrinsicnerf_amd import kernels
    dev = _dev()
    cfg = oracle.RenderConfig(variant=variant, n_samples=s, n_importance=0, n_classes=c)
    sd = oracle.make_state_dict(variant, c, seed=5)
    g = torch.Generator().manual_seed(6)
    n = 37                                                    # 37 * s is not a multiple of the 64-point tile
    o = torch.randn(n, 3, generator=g)
    d = torch.randn(n, 3, generator=g)
    near, far = (2.0, 6.0) if variant == "object" else (0.1, 10.0)
    rays = torch.cat([o, d, near * torch.ones(n, 1), far * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1)
    z = torch.sort(torch.rand(n, s, generator=g) * (far - near) + near, -1)[0]
    with torch.no_grad():
        pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
        want = oracle.query_network(sd, pts, rays[:, 8:11], cfg)
    raw = kernels.encode_mlp(_desc(cfg), _packed(cfg, sd), rays.to(dev), z.to(dev))
    assert_maps_close(raw.cpu().numpy(), want.numpy(), RTOL, ATOL, f"raw {variant} C={c} S={s}")


@pytest.mark.parametrize("c", [5, 28, 33, 101, 240])
def test_ssr_semantic_head_forms_agree(c, precision, monkeypatch):
    """The SSR network's two semantic-head forms of the two-workgroup kernel - per wave (every wave the whole head for its 16
    points) and channel-split (hidden layer split over the waves, partial logits through an L2-resident scratch; the default
    for C <= 32, forced here for every C: more than 32 classes go block by block through the exchange area) - against the
    oracle and against each other: the 11 base channels bit for bit, the logits up to their summation order."""
    from intrinsicnerf_amd import kernels
    if precision != "f16x3":
        pytest.skip("forms of the default f16x3 kernel")
    dev = _dev()
    cfg = oracle.RenderConfig(variant="ssr", n_samples=48, n_importance=0, n_classes=c)
    sd = oracle.make_state_dict("ssr", c, seed=7)
    g = torch.Generator().manual_seed(8)
    n = 1501                                                  # 1501 * 48 points = 1125 full tiles + a ragged one: 2-3 tiles per workgroup
    d = torch.randn(n, 3, generator=g)
    rays = torch.cat([torch.randn(n, 3, generator=g), d, 0.1 * torch.ones(n, 1), 10 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1)
    z = torch.sort(torch.rand(n, 48, generator=g) * 9.9 + 0.1, -1)[0]
    with torch.no_grad():
        want = oracle.query_network(sd, rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None], rays[:, 8:11], cfg)
    out = {}
    for form in ("wave", "csplit", "t128"):     # t128: the 128-point tile (the default for C <= 32; more classes take the per-wave head)
        monkeypatch.setenv("INERF_F16_KERNEL", form)
        raw = kernels.encode_mlp(_desc(cfg), _packed(cfg, sd), rays.to(dev), z.to(dev))
        assert_maps_close(raw.cpu().numpy(), want.numpy(), RTOL, ATOL, f"raw C={c} {form}")
        out[form] = raw
    assert torch.equal(out["wave"][..., :11], out["csplit"][..., :11]) and torch.equal(out["wave"][..., :11], out["t128"][..., :11])
    assert_maps_close(out["csplit"][..., 11:].cpu().numpy(), out["wave"][..., 11:].cpu().numpy(), 1e-5, 1e-6, "logits, split vs per-wave head")
    assert_maps_close(out["t128"][..., 11:].cpu().numpy(), out["wave"][..., 11:].cpu().numpy(), 1e-5, 1e-6, "logits, 128-point tile vs per-wave head")


@pytest.mark.parametrize("n,s", [(1, 64), (3, 1), (1, 191), (33, 64), (1000, 192), (4099, 192), (32768, 64)])
def test_object_kernel_tile_forms_are_bit_identical(n, s, precision, monkeypatch):
    """The object-level inference kernel's forms - 128-point tile (k_encode_mlp_f16x3_t128, the default since round 6) and
    64-point tile with two workgroups per CU (k_encode_mlp_f16x3_dual, ``INERF_F16_KERNEL=dual``), and the 128-point tile's
    pipelined trunk (``INERF_F16_KERNEL=pp``) - sum every output element in
    the same order: raw must agree bit for bit, whole tiles, ragged tiles and launches smaller than one tile alike - and both meet
    the oracle (reproducible arithmetic: curated weights)."""
    from intrinsicnerf_amd import kernels
    if precision != "f16x3":
        pytest.skip("forms of the default f16x3 kernel")
    dev = _dev()
    cfg = oracle.RenderConfig(variant="object", n_samples=s, n_importance=0)
    g = torch.Generator().manual_seed(1000 * n + s)
    o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
    d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
    rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).contiguous()
    z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0]
    sd, _ = oracle.calibrated_lcg_weights("object", 0, 40, rays[:256])
    out = {}
    for form in ("t128", "dual", "pp"):      # pp: the 128-point tile with the pipelined trunk (mlp_f16_pp.h; opt-in, measured slower)
        monkeypatch.setenv("INERF_F16_KERNEL", form)
        out[form] = kernels.encode_mlp(_desc(cfg), _packed(cfg, sd), rays.to(dev), z.to(dev))
    assert torch.equal(out["t128"], out["dual"]), f"max |diff| {float((out['t128'] - out['dual']).abs().max()):.3e}"
    assert torch.equal(out["t128"], out["pp"]), f"pipelined trunk: max |diff| {float((out['t128'] - out['pp']).abs().max()):.3e}"
    k = min(n, 64)
    with torch.no_grad():
        want = oracle.query_network(sd, rays[:k, None, 0:3] + rays[:k, None, 3:6] * z[:k, :, None], rays[:k, 8:11], cfg)
    assert_maps_close(out["t128"][:k].cpu().numpy(), want.numpy(), RTOL, ATOL, f"raw {n}x{s}")


def test_run_network_arbitrary_points():
    """run_network(pts, viewdirs, net, ...) on a point grid (what extract_colour_mesh.py:158-162 does)."""
    from intrinsicnerf_amd import ssr
    dev = _dev()
    c = 6
    sd = oracle.make_state_dict("ssr", c, seed=8)
    embed, ch = ssr.get_embedder(10, 0, scalar_factor=10); embed_d, ch_d = ssr.get_embedder(4, 0, scalar_factor=1)
    net = ssr.Semantic_NeRF(True, c, D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d,
                            use_viewdirs=True).to(dev)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    pts = torch.randn(50, 7, 3, generator=g) * 3
    vd = torch.zeros(50, 3)
    with torch.no_grad():
        got = ssr.run_network(pts.to(dev), vd.to(dev), net, embed, embed_d, netchunk=1024)
        cfg = oracle.RenderConfig(variant="ssr", n_classes=c)
        want = oracle.query_network(sd, pts, vd, cfg)
    assert_maps_close(got.cpu().numpy(), want.numpy(), RTOL, ATOL, "run_network grid")


def test_sample_coarse_bit_exact():
    from intrinsicnerf_amd import kernels
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    n = 33
    rays = torch.randn(n, 11, generator=g)
    rays[:, 6] = torch.rand(n, generator=g) + 0.5
    rays[:, 7] = rays[:, 6] + torch.rand(n, generator=g) * 5 + 1
    t = torch.linspace(0., 1., 64)
    tr = torch.rand(n, 64, generator=g)
    for lindisp in (False, True):
        for t_rand in (None, tr):
            want = oracle.coarse_depths(rays[:, 6:7], rays[:, 7:8], t, lindisp, t_rand)
            got = kernels.sample_coarse(rays.to(dev), t.to(dev), None if t_rand is None else t_rand.to(dev), lindisp)
            assert torch.equal(got.cpu(), want), f"lindisp={lindisp} perturb={t_rand is not None}"


def test_f16_range_guard(precision):
    """Activations beyond f16's range: the split-precision kernel must say so (status word -> FloatingPointError),
    the fp32 kernel must simply compute them."""
    from intrinsicnerf_amd import _capi, kernels, packing
    dev = _dev()
    sd = oracle.make_state_dict("object", 0, seed=5)
    sd["pts_linears.0.bias"] = sd["pts_linears.0.bias"] + 1.0e5          # 8 * h1 ~ 8e5 > 65504
    cfg = oracle.RenderConfig(variant="object")
    desc = _desc(cfg)
    rays = torch.rand(8, 11)
    z = torch.rand(8, 64) + 2
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    raw = kernels.encode_mlp(desc, packing.pack_state_dict(desc, sd).to(dev), rays.to(dev), z.to(dev), status=status)
    if precision.startswith("f16x3"):
        assert int(status.item()) & _capi.STATUS_F16_RANGE
        with pytest.raises(FloatingPointError):
            kernels.check_f16_range(status, "test")
    else:
        assert int(status.item()) == 0
        pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
        want = oracle.query_network(sd, pts, rays[:, 8:11], cfg)
        assert_maps_close(raw.cpu().numpy(), want.numpy(), RTOL, 1e-4 * float(want.abs().max()), "raw with huge activations")


def test_frontend_falls_back_to_f32_on_range(precision):
    """A network whose activations leave f16's range still renders correctly through the front-end: the
    split-precision attempt is detected (status word) and the batch re-run on the exact fp32 kernel."""
    import warnings
    from intrinsicnerf_amd import object_level as ol
    dev = _dev()
    sd = oracle.make_state_dict("object", 0, seed=5)
    sd["pts_linears.0.bias"] = sd["pts_linears.0.bias"] + 2.0e4                # 8 * h1 > 6e4
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    net = ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    pts = torch.randn(40, 5, 3, generator=g)
    vd = torch.randn(40, 3, generator=g)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = ol.run_network(pts.to(dev), vd.to(dev), net, embed, embed_d)
        want = oracle.query_network(sd, pts, vd, oracle.RenderConfig(variant="object"))
    assert_maps_close(got.cpu().numpy(), want.numpy(), RTOL, 1e-4 * float(want.abs().max()), "fallback raw")


def test_training_step_takes_the_staged_path_loudly():
    """Under autograd with trainable networks the front-end says (once per process) that it switches to the staged
    path, returns maps with a grad_fn, and the same call under no_grad goes back to the fused kernel with equal maps."""
    import warnings
    from intrinsicnerf_amd import object_level as ol
    dev = _dev()
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    net = ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    fx = load_golden("object_chair_det")
    net.load_state_dict(case_weights(fx)[0])
    rays = torch.from_numpy(fx["rays"][:6]).to(dev)
    ol._told_training_path = False
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        ret = ol.render_rays(rays, net, ol.NetworkQuery(embed, embed_d), 64, N_importance=32, white_bkgd=True)
    assert any("staged training path" in str(w.message) for w in rec)
    assert ret["rgb_map"].grad_fn is not None and ret["rgb0"].grad_fn is not None
    ret["rgb_map"].sum().backward()
    assert net.alpha_linear.weight.grad is not None and float(net.alpha_linear.weight.grad.abs().sum()) > 0
    with torch.no_grad():
        fused = ol.render_rays(rays, net, ol.NetworkQuery(embed, embed_d), 64, N_importance=32, white_bkgd=True)
    for k in ("rgb_map", "acc_map", "albedo_map", "shading_map", "residual_map", "rgb0", "acc0"):
        assert fused[k].grad_fn is None
        assert_maps_close(ret[k].detach().cpu().numpy(), fused[k].cpu().numpy(), 2e-4, 2e-5, k)


# ------------------------------------------------------------------------------------------------
# awkward shapes and reduced encoders
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_randomized_shapes_and_flags(seed):
    """Whole path vs the oracle over odd ray counts, sample counts that do not divide the 64-point tile, 1-sample
    importance passes, reduced multires / multires_views, lindisp, shared coarse/fine network, random C."""
    from intrinsicnerf_amd import _capi, kernels, packing
    rng = np.random.RandomState(100 + seed)
    variant = "ssr" if seed % 2 else "object"
    n = int(rng.randint(1, 90))
    s_c = int(rng.choice([5, 17, 33, 64, 100]))
    n_imp = int(rng.choice([0, 1, 7, 64, 128, 150]))
    l_xyz, l_dir = int(rng.randint(0, 11)), int(rng.randint(0, 5))
    c = int(rng.randint(1, 40)) if variant == "ssr" else 0
    cfg = oracle.RenderConfig(variant=variant, n_samples=s_c, n_importance=n_imp, l_xyz=l_xyz, l_dir=l_dir,
                              white_bkgd=bool(rng.randint(2)), lindisp=bool(rng.randint(2)) and variant == "object",
                              n_classes=c, endpoint_feat=bool(rng.randint(2)) and variant == "ssr" and n_imp > 0)
    g = torch.Generator().manual_seed(200 + seed)
    o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3) * (10.0 if variant == "ssr" else 1.0) * 0.3
    d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
    near, far = (2.0, 6.0) if variant == "object" else (0.1, 10.0)
    rays = torch.cat([o, d, near * torch.ones(n, 1), far * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1)
    kw = dict(sigma_gain_log2=5, freq_decay=True, l_xyz=l_xyz, l_dir=l_dir)
    sd_c = oracle.lcg_state_dict(variant, c, seed=300 + seed, sigma_bias=0.25, **kw)
    sd_f = sd_c if seed % 3 == 0 else oracle.lcg_state_dict(variant, c, seed=400 + seed, sigma_bias=0.25, **kw)
    t_vals = torch.linspace(0., 1., s_c)
    u = torch.linspace(0., 1., n_imp) if n_imp > 0 else None
    with torch.no_grad():
        want = oracle.render_rays(rays, sd_c, sd_f, cfg, t_vals=t_vals, u=u)
    ok = (oracle.conditioning_scores(rays, sd_c, sd_f, cfg, t_vals, stage_keys=()) <= 0.2).numpy()
    if ok.sum() == 0:
        pytest.skip("no well-conditioned ray in this draw")
    dev = _dev()
    desc = _capi.net_desc(_capi.VARIANT_SSR if variant == "ssr" else _capi.VARIANT_OBJECT, c, l_xyz, l_dir, cfg.xyz_div)
    pc = packing.pack_state_dict(desc, sd_c).to(dev)
    pf = None if sd_f is sd_c else packing.pack_state_dict(desc, sd_f).to(dev)
    got = kernels.render_rays_fused(desc, pc, pf, rays.to(dev), s_c, n_imp, t_vals.to(dev), None if u is None else u.to(dev),
                                    white_bkgd=cfg.white_bkgd, lindisp=cfg.lindisp, endpoint=cfg.endpoint_feat)
    kernels.check_f16_range(got.pop("status", None), "test")
    for k, w in want.items():
        if k in got:
            assert_maps_close(got[k].cpu().numpy()[ok], w.numpy()[ok], _rtol(k), ATOL,
                              f"seed {seed} {variant} n={n} S={s_c}+{n_imp} L={l_xyz}/{l_dir} C={c}: {k}")
    assert {k for k in want if not k.startswith(("raw", "z_", "weights"))} - {"z_std"} <= set(got) | {"z_std"}


def test_empty_batches():
    """Zero rays: every stage, the fused path and the front-end return correctly shaped empty tensors (the reference's
    torch ops do the same on an empty ray batch)."""
    from intrinsicnerf_amd import _capi, kernels, object_level as ol, packing
    dev = _dev()
    desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, 1.0)
    sd = oracle.make_state_dict("object", 0, seed=1)
    packed = packing.pack_state_dict(desc, sd).to(dev)
    rays = torch.zeros(0, 11, device=dev)
    t = torch.linspace(0., 1., 64, device=dev)
    z = kernels.sample_coarse(rays, t)
    assert tuple(z.shape) == (0, 64)
    raw = kernels.encode_mlp(desc, packed, rays, z)
    assert tuple(raw.shape) == (0, 64, 11)
    c = kernels.composite(raw, z, rays[:, 3:6].contiguous(), None, True)
    assert tuple(c["rgb"].shape) == (0, 3) and tuple(c["weights"].shape) == (0, 64)
    zs, zm, zstd = kernels.sample_fine(z, c["weights"], torch.linspace(0., 1., 128, device=dev), 128)
    assert tuple(zs.shape) == (0, 128) and tuple(zm.shape) == (0, 192) and tuple(zstd.shape) == (0,)
    out = kernels.render_rays_fused(desc, packed, packed, rays, 64, 128, t, torch.linspace(0., 1., 128, device=dev), white_bkgd=True)
    assert tuple(out["rgb_fine"].shape) == (0, 3) and tuple(out["z_std"].shape) == (0,)
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    net = ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    with torch.no_grad():
        ret = ol.render_rays(rays, net, ol.NetworkQuery(embed, embed_d), 64, N_importance=128, white_bkgd=True, retraw=True)
    assert tuple(ret["rgb_map"].shape) == (0, 3) and tuple(ret["raw"].shape) == (0, 192, 11)


@pytest.mark.gpu
@pytest.mark.parametrize("n,s", [(3, 1), (33, 64), (700, 192)])
def test_parked_encoding_equals_the_second_encoder_pass(n, s, precision, monkeypatch):
    """INERF_ENC_CACHE=1: the 128-point tile parks the tile's position encoding in the caller's workspace for the skip layer
    (inerf_encode_mlp_ws / _chunked / inerf_render_rays); by default, and without a workspace (inerf_encode_mlp), it evaluates the encoder a
    second time (run_nerf_helpers.py:290-291: cat([input_pts, h])).  Same bits."""
    import ctypes as C
    from intrinsicnerf_amd import _capi, kernels
    if precision != "f16x3":
        pytest.skip("a form of the default f16x3 kernel")
    dev = _dev()
    cfg = oracle.RenderConfig(variant="object", n_samples=s, n_importance=0)
    g = torch.Generator().manual_seed(7 * n + s)
    o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
    d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
    rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).contiguous().to(dev)
    z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0].to(dev)
    sd, _ = oracle.calibrated_lcg_weights("object", 0, 40, rays[:256].cpu())
    desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, 1.0, _capi.PREC_F16X3)
    packed = _packed(cfg, sd)
    lib = _capi.lib()
    monkeypatch.delenv("INERF_F16_KERNEL", raising=False)
    monkeypatch.setenv("INERF_ENC_CACHE", "1")
    assert lib.inerf_encode_mlp_workspace_bytes(desc, n, s, 0) > 0, "INERF_ENC_CACHE=1: the object-level launch asks for the parking slot"
    with_ws = kernels.encode_mlp(desc, packed, rays, z)
    plain = torch.empty_like(with_ws)
    rc = lib.inerf_encode_mlp(desc, C.c_void_p(packed.data_ptr()), C.c_void_p(rays.data_ptr()), C.c_void_p(z.data_ptr()), n, s, 0,
                              C.c_void_p(plain.data_ptr()), None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _capi.check(rc, "inerf_encode_mlp")
    monkeypatch.delenv("INERF_ENC_CACHE")
    assert lib.inerf_encode_mlp_workspace_bytes(desc, n, s, 0) == 0          # the default: the encoder runs a second time
    twice = kernels.encode_mlp(desc, packed, rays, z)
    assert torch.equal(with_ws, plain) and torch.equal(with_ws, twice)
