"""Backward of the path, first part: compositing.

CPU: autograd through the oracle reproduces the gradients the REFERENCE's autograd produced
(tests/golden/make_golden_grad.py), so the oracle's backward is pinned like its forward.
GPU (-m gpu): the HIP backward kernel (inerf_composite_backward) against the same reference gradients, through
the C ABI and through the autograd wiring of the front-ends; then a whole training-step gradient (both networks)
through the front-end's staged path against the reference's parameter gradients."""
import numpy as np
import pytest
import torch

import oracle
from _cases import assert_maps_close, case_config, case_weights
from conftest import golden_names, load_golden

OBJ_KEYS = ["rgb", "disp", "acc", "weights", "depth", "albedo", "shading", "residual"]
SSR_KEYS = OBJ_KEYS + ["sem", "feat"]
# gradients are sums of O(100) products of O(1) numbers; the reference's own fp32 autograd differs from an fp64
# evaluation by ~1e-6 relative to the largest entry of a ray.  Bound: 1e-4 relative + 1e-5 of the tensor's scale.
RTOL = 1e-4


def _atol(want):
    w = np.asarray(want, np.float64)
    return 1e-5 * float(np.nanmax(np.abs(w))) if np.isfinite(w).any() else 1e-5


def _cfg(fx):
    ssr = "n_classes" in fx
    return oracle.RenderConfig(variant="ssr" if ssr else "object", white_bkgd=bool(fx["white_bkgd"]),
                               n_classes=int(fx["n_classes"]) if ssr else 0, endpoint_feat=ssr), ssr


@pytest.mark.parametrize("name", golden_names("grad_composite_"))
def test_oracle_autograd_matches_reference(name):
    fx = load_golden(name)
    cfg, ssr = _cfg(fx)
    keys = SSR_KEYS if ssr else OBJ_KEYS
    for noise_key, want_key in ((None, "d_raw"), ("noise", "d_raw_noise")):
        if want_key not in fx:
            continue
        raw = torch.from_numpy(fx["raw"]).clone().requires_grad_(True)
        noise = None if noise_key is None else torch.from_numpy(fx[noise_key])
        out = oracle.composite(raw, torch.from_numpy(fx["z"]), torch.from_numpy(fx["rays_d"]), cfg, noise, feat=ssr)
        (d,) = torch.autograd.grad(sum((torch.from_numpy(fx["cot_" + k]) * out[k]).sum() for k in keys), raw)
        assert_maps_close(d.numpy(), fx[want_key], 1e-6, 1e-6 * _atol(fx[want_key]) / 1e-5, f"{name}/{want_key}")


def test_oracle_parameter_gradients_match_reference():
    """Whole path, both networks: digests (norm, projection, leading entries) of every parameter gradient."""
    fx = load_golden("grad_render_object")
    src = load_golden(str(fx["source_fixture"]))
    cfg = case_config(src)
    sd_c, sd_f = case_weights(src)
    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    out = oracle.render_rays(torch.from_numpy(fx["rays"]), pc, pf, cfg, t_vals=torch.from_numpy(src["t_vals"]))
    loss = sum((torch.from_numpy(fx[k]) * out[k[4:]]).sum() for k in fx if k.startswith("cot_"))
    loss.backward()
    checked = 0
    for tag, params in (("coarse", pc), ("fine", pf)):
        for i, (name, p) in enumerate(params.items()):
            want = fx[f"grad_{tag}/{name}"]
            got = _digest(p.grad, 1000 + i)
            assert abs(got[0] - want[0]) <= 1e-6 * want[0], (tag, name)
            assert abs(got[1] - want[1]) <= 1e-5 * want[0] * np.sqrt(p.numel()), (tag, name)
            np.testing.assert_allclose(got[2:], want[2:], rtol=1e-5, atol=1e-6 * want[0])
            checked += 1
    assert checked == 2 * len(sd_c)


def _digest(t, seed, head=16):
    t = t.detach().double().flatten().cpu()
    g = torch.Generator().manual_seed(seed)
    proj = torch.randn(t.numel(), generator=g, dtype=torch.float64)
    return np.concatenate([[float(t.norm())], [float((t * proj).sum())], t[:head].numpy()])


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names("grad_composite_"))
def test_hip_composite_backward_vs_reference(name):
    from intrinsicnerf_amd import kernels
    dev = torch.device("cuda:0")
    fx = load_golden(name)
    cfg, ssr = _cfg(fx)
    keys = SSR_KEYS if ssr else OBJ_KEYS
    c = cfg.n_classes if ssr else 0
    feat = 128 if ssr else 0
    t = lambda k: torch.from_numpy(fx[k]).to(dev)
    grads = {k: t("cot_" + k) for k in keys}
    for noise_key, want_key in ((None, "d_raw"), ("noise", "d_raw_noise")):
        if want_key not in fx:
            continue
        noise = None if noise_key is None else t(noise_key)
        d = kernels.composite_backward(t("raw"), t("z"), t("rays_d"), grads, noise, cfg.white_bkgd, c, feat)
        assert_maps_close(d.cpu().numpy(), fx[want_key], RTOL, _atol(fx[want_key]), f"{name}/{want_key}")
        # the same through autograd: a leaf raw, the front-end's differentiable composite, loss.backward()
        raw = t("raw").clone().requires_grad_(True)
        out = kernels.composite(raw, t("z"), t("rays_d"), noise, cfg.white_bkgd, c, feat)
        assert all(out[k].grad_fn is not None for k in keys)
        sum((grads[k] * out[k]).sum() for k in keys).backward()
        assert_maps_close(raw.grad.cpu().numpy(), fx[want_key], RTOL, _atol(fx[want_key]), f"{name}/{want_key} (autograd)")


@pytest.mark.gpu
def test_hip_composite_backward_subsets_and_linearity():
    """Only some outputs carry gradient (NULL pointers for the rest); the backward is linear in the cotangents."""
    from intrinsicnerf_amd import kernels
    dev = torch.device("cuda:0")
    fx = load_golden("grad_composite_object_wb1")
    t = lambda k: torch.from_numpy(fx[k]).to(dev)
    raw, z, d = t("raw"), t("z"), t("rays_d")
    full = {k: t("cot_" + k) for k in OBJ_KEYS}
    total = kernels.composite_backward(raw, z, d, full, None, True)
    parts = sum(kernels.composite_backward(raw, z, d, {k: full[k]}, None, True) for k in OBJ_KEYS)
    assert torch.equal(torch.isnan(total), torch.isnan(parts))       # rays with acc == 0: NaN through disp, in both
    assert bool(torch.isnan(total).any()) and not bool(torch.isnan(total[3:]).all())
    scale = float(torch.nan_to_num(total).abs().max())
    assert float(torch.nan_to_num(total - parts).abs().max()) <= 1e-5 * scale
    no_disp = kernels.composite_backward(raw, z, d, {k: v for k, v in full.items() if k != "disp"}, None, True)
    assert not bool(torch.isnan(no_disp).any())                      # without a disp gradient nothing is NaN
    assert float(kernels.composite_backward(raw, z, d, {}, None, True).abs().max()) == 0.0
    with pytest.raises(KeyError):
        kernels.composite_backward(raw, z, d, {"colour": full["rgb"]}, None, True)


@pytest.mark.gpu
def test_training_step_gradients_vs_reference():
    """loss.backward() through the object-level front-end (render_rays with trainable networks): every parameter
    gradient of both networks against the reference's autograd (digests in grad_render_object.npz)."""
    from intrinsicnerf_amd import object_level as ol
    dev = torch.device("cuda:0")
    fx = load_golden("grad_render_object")
    src = load_golden(str(fx["source_fixture"]))
    sd_c, sd_f = case_weights(src)
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net_c, net_f = mk(), mk()
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    rays = torch.from_numpy(fx["rays"]).to(dev)
    ret = ol.render_rays(rays, net_c, ol.NetworkQuery(embed, embed_d), 64, retraw=True, N_importance=128, network_fine=net_f,
                         white_bkgd=True)
    name_of = {"rgb_fine": "rgb_map", "albedo_fine": "albedo_map", "shading_fine": "shading_map", "residual_fine": "residual_map",
               "disp_fine": "disp_map", "acc_fine": "acc_map", "rgb_coarse": "rgb0", "albedo_coarse": "albedo0",
               "shading_coarse": "shading0", "residual_coarse": "residual0", "acc_coarse": "acc0"}
    loss = sum((torch.from_numpy(fx["cot_" + k]).to(dev) * ret[name_of[k]]).sum() for k in name_of)
    loss.backward()
    for tag, net in (("coarse", net_c), ("fine", net_f)):
        for i, (name, p) in enumerate(net.named_parameters()):
            want = fx[f"grad_{tag}/{name}"]
            assert p.grad is not None, (tag, name)
            got = _digest(p.grad, 1000 + i)
            # per tensor: norm to 1e-4, projection onto a random direction to 1e-4 of norm * sqrt(numel), leading entries
            assert abs(got[0] - want[0]) <= 1e-4 * want[0], (tag, name, got[0], want[0])
            assert abs(got[1] - want[1]) <= 1e-4 * want[0] * np.sqrt(p.numel()), (tag, name)
            np.testing.assert_allclose(got[2:], want[2:], rtol=1e-3, atol=1e-4 * want[0], err_msg=f"{tag}/{name}")


@pytest.mark.gpu
def test_ssr_training_step_gradients_vs_oracle():
    """SSRTrainer.step's backward (trainer.py:882-990) through the SSR front-end: semantic logits, depth, endpoint feature
    and the intrinsic maps all carry gradient; every parameter gradient against autograd through the CPU oracle
    (whose backward is pinned to the reference's by the tests above)."""
    import warnings
    from intrinsicnerf_amd import ssr
    dev = torch.device("cuda:0")
    src = load_golden("ssr_endpoint_c5_wb")
    cfg = case_config(src)
    sd_c, sd_f = case_weights(src)
    rays = torch.from_numpy(src["rays"])[:6].contiguous()
    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    want = oracle.render_rays(rays, pc, pf, cfg, t_vals=torch.from_numpy(src["t_vals"]))
    keys = {"rgb_fine": "rgb_fine", "albedo_fine": "albedo_fine", "shading_fine": "shading_fine", "residual_fine": "residual_fine",
            "depth_fine": "depth_fine", "sem_fine": "sem_logits_fine", "feat_fine": "feat_map_fine", "rgb_coarse": "rgb_coarse",
            "depth_coarse": "depth_coarse", "sem_coarse": "sem_logits_coarse", "acc_coarse": "acc_coarse"}
    g = torch.Generator().manual_seed(11)
    cot = {k: torch.randn(want[k].shape, generator=g) for k in keys}
    sum((cot[k] * want[k]).sum() for k in keys).backward()

    r = ssr.SSRRenderer(cfg.n_classes, white_bkgd=cfg.white_bkgd, endpoint_feat=cfg.endpoint_feat, device=dev)
    r.ssr_net_coarse.load_state_dict(sd_c); r.ssr_net_fine.load_state_dict(sd_f)
    r.check_numerics = False
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ret = r.render_rays(rays.to(dev))
    for ok, fk in keys.items():       # forward values of the staged path first
        assert_maps_close(ret[fk].detach().cpu().numpy(), want[ok].detach().numpy(), 2e-4, 2e-5, fk)
    sum((cot[ok].to(dev) * ret[fk]).sum() for ok, fk in keys.items()).backward()
    for tag, net, params in (("coarse", r.ssr_net_coarse, pc), ("fine", r.ssr_net_fine, pf)):
        for name, p in net.named_parameters():
            w = params[name].grad.double()
            assert p.grad is not None, (tag, name)
            err = float((p.grad.double().cpu() - w).norm())
            assert err <= 2e-4 * float(w.norm()) + 1e-12, (tag, name, err, float(w.norm()))


# ---------------------------------------------------------------------------------------------------------------------
# network backward: fused training forward + MFMA input-gradient chain + weight-gradient GEMMs (kernels.mlp_train)
# ---------------------------------------------------------------------------------------------------------------------
def _relu_tie_points(module, emb, endpoint, rel=1e-6):
    """Sample points at which some pre-activation of ``module`` (every nn.Linear output) is within ``rel`` of its layer's
    rms of zero.  There an fp32 evaluation lands on either side of the ReLU depending on its rounding, and the whole
    downstream gradient of that channel switches on or off: with tens of thousands of points a few such ties always
    exist, and one at a point with a large cotangent moves a tensor's gradient by 1e-3 (scripts/diag_train_grads.py:
    pre-activation 2.4e-9 against an rms of 0.039 at the dominant point of the 44 800-point SSR case)."""
    ties = torch.zeros(emb.shape[0], dtype=torch.bool, device=emb.device)
    hooks = []

    def watch(mod, inputs, out):
        if out.dim() == 2 and out.shape[0] == emb.shape[0]:
            ties.logical_or_((out.detach().abs() < rel * out.detach().pow(2).mean().sqrt()).any(1))

    for m in module.modules():
        if isinstance(m, torch.nn.Linear):
            hooks.append(m.register_forward_hook(watch))
    with torch.no_grad():
        module(emb, True) if endpoint else module(emb)
    for h in hooks:
        h.remove()
    return ties


def _torch_reference_grads(module, embed, embed_d, rays, z, cot, endpoint=False, ties_out=None):
    """raw and parameter gradients of the same network through torch autograd (the module's own forward).  The sample
    positions are always the fp32 ones (o + d z rounded like the reference and the kernels do): one ulp of position is 1e-4
    rad in the 2^9 frequency band, so an fp64 run on fp64 positions would be a different function."""
    dtype = cot.dtype
    rays, z = rays.float(), z.float()
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
    sf = float(getattr(embed, "scalar_factor", 1.0))
    if dtype != torch.float32 and sf != 1.0:
        # the SSR encoding's x / 10 (semantic_nerf.py:64) belongs to the positions too: an fp32 true division (done on the
        # CPU: torch's GPU kernel multiplies by the reciprocal), then an embedder without a divisor
        from intrinsicnerf_amd import ssr
        embed = ssr.get_embedder(10, 0, scalar_factor=1)[0]
        pts = (pts.cpu() / sf).to(pts.device)
    pts, rays = pts.to(dtype), rays.to(dtype)
    emb = torch.cat([embed(pts.reshape(-1, 3)), embed_d(rays[:, None, 8:11].expand(pts.shape).reshape(-1, 3))], -1)
    if ties_out is not None:
        ties_out.append(_relu_tie_points(module, emb, endpoint).reshape(z.shape))
    raw = module(emb, True) if endpoint else module(emb)
    raw = raw.reshape(z.shape[0], z.shape[1], -1)
    module.zero_grad()
    (raw * cot).sum().backward()
    return raw.detach(), {k: p.grad.clone() for k, p in module.named_parameters()}


@pytest.mark.gpu
@pytest.mark.parametrize("variant,c,endpoint,n,s", [("object", 0, False, 37, 5), ("object", 0, False, 64, 64),
                                                    ("ssr", 5, True, 23, 11), ("ssr", 28, False, 16, 192), ("ssr", 0, False, 9, 7),
                                                    # more 64-point tiles than workgroups: the kernels' tile loops go round
                                                    ("ssr", 5, True, 350, 64), ("object", 0, False, 700, 64), ("ssr", 3, False, 700, 64),
                                                    # C > 128: semantic_linear.1's weight gradient takes the 256-row tile of the split-K kernel
                                                    ("ssr", 150, False, 40, 16)])
@pytest.mark.parametrize("form", ["default", "single"])
def test_network_backward_vs_torch_autograd(variant, c, endpoint, n, s, form, monkeypatch):
    """One network, arbitrary cotangent on raw (every channel: sigma, the sigmoid heads, logits, endpoint feature), ragged
    point counts: raw and every parameter gradient of kernels.mlp_train against torch autograd through the module's own
    forward on the same GPU.  The default training forward is the two-workgroup kernel; `single` forces the one-workgroup
    one, which SSR renders with the endpoint feature always use."""
    from intrinsicnerf_amd import kernels, object_level as ol, ssr
    if form == "single":
        if endpoint:
            pytest.skip("same kernel as the default")
        monkeypatch.setenv("INERF_F16_KERNEL", "single")
    else:
        monkeypatch.delenv("INERF_F16_KERNEL", raising=False)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7 + n)
    sd = oracle.lcg_state_dict(variant, c, seed=21, sigma_gain_log2=3, freq_decay=True)
    if variant == "object":
        embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
        net = ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    else:
        embed, ch = ssr.get_embedder(10, 0, scalar_factor=10); embed_d, ch_d = ssr.get_embedder(4, 0, scalar_factor=1)
        net = ssr.Semantic_NeRF(c > 0, c, D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net.load_state_dict(sd)
    o = torch.rand(n, 3, generator=g) * 2 - 1
    d = torch.randn(n, 3, generator=g)
    rays = torch.cat([o, d, torch.zeros(n, 2), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
    z = torch.sort(torch.rand(n, s, generator=g) * 3 + 0.5, -1)[0].to(dev)
    chn = 11 + c + (128 if endpoint else 0)
    cot = (torch.randn(n, s, chn, generator=g) * torch.logspace(-3, 1, n, base=10.0)[:, None, None]).to(dev)   # 4 decades of scale
    if n * s > 8192:      # a few ReLU ties always exist among tens of thousands of points: they get no cotangent
        import copy
        net64 = copy.deepcopy(net).double()
        ties = []
        _torch_reference_grads(net64, embed, embed_d, rays.double(), z.double(), cot.double(), endpoint, ties_out=ties)
        assert 0 < int(ties[0].sum()) < n * s // 100, int(ties[0].sum())
        cot = cot * (~ties[0])[:, :, None].to(cot.dtype)
    want_raw, want = _torch_reference_grads(net, embed, embed_d, rays, z, cot, endpoint)
    if n * s > 8192:      # tens of thousands of cancelling terms per gradient element: torch's fp32 autograd is 1e-3 off in
                          # the lower trunk layers at 44 800 points (scripts/diag_train_grads.py) - fp64 autograd is the judge
        _, want64 = _torch_reference_grads(net64, embed, embed_d, rays.double(), z.double(), cot.double(), endpoint)
        fp32_dev = max(float((want[k].double() - want64[k]).norm() / want64[k].norm().clamp_min(1e-30)) for k in want)
        print(f"torch fp32 autograd vs fp64: {fp32_dev:.2e} of a tensor's norm at worst")
        want = want64
    net.zero_grad()
    desc = net.fused_desc()
    desc.xyz_div = embed.scalar_factor
    raw = kernels.mlp_train(desc, net, rays, z, endpoint)
    assert type(raw.grad_fn).__name__ == "_FusedMlpFnBackward"
    assert_maps_close(raw.detach().cpu().numpy(), want_raw.cpu().numpy(), 1e-4, 1e-5 * float(want_raw.abs().max()), "raw")
    (raw * cot).sum().backward()
    errs = {name: (float((p.grad.double() - want[name].double()).norm()), float(want[name].double().norm()))
            for name, p in net.named_parameters()}
    bad = {k: f"{e / max(w, 1e-30):.1e}" for k, (e, w) in errs.items() if e > 2e-4 * w + 1e-10}
    if n * s > 8192:
        print("relative errors:", {k: f"{e / max(w, 1e-30):.1e}" for k, (e, w) in errs.items()})
    assert not bad, bad


@pytest.mark.gpu
def test_training_step_uses_the_hip_network_backward(monkeypatch):
    """The front-end's training step wires kernels.mlp_train in by default and torch's layers with INERF_TRAIN_MLP=torch."""
    import warnings
    from intrinsicnerf_amd import object_level as ol
    dev = torch.device("cuda:0")
    fx = load_golden("object_chair_det")
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    net = ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net.load_state_dict(case_weights(fx)[0])
    rays = torch.from_numpy(fx["rays"][:5]).to(dev)
    grads = {}
    for mode in ("hip", "torch"):
        monkeypatch.setenv("INERF_TRAIN_MLP", mode)
        net.zero_grad()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ret = ol.render_rays(rays, net, ol.NetworkQuery(embed, embed_d), 64, retraw=True, N_importance=16, white_bkgd=True)
        assert (type(ret["raw"].grad_fn).__name__ == "_FusedMlpFnBackward") == (mode == "hip")
        (ret["rgb_map"].sum() + ret["albedo_map"].sum() + ret["rgb0"].sum()).backward()
        grads[mode] = {k: p.grad.clone() for k, p in net.named_parameters()}
    for k in grads["hip"]:
        w = grads["torch"][k].double()
        assert float((grads["hip"][k].double() - w).norm()) <= 2e-4 * float(w.norm()) + 1e-10, k


@pytest.mark.gpu
def test_exact_fp32_precision_trains_on_the_hip_kernels(monkeypatch):
    """INERF_PRECISION=f32 in a training step (VERDICT r02 missing #4: it used to fall back to torch layers): the values that
    leave the network node are the exact-fp32 MFMA kernel's - bit for bit what the no_grad render of the same rays returns -
    while the backward runs on the HIP chain / weight-gradient kernels from the activations a split-precision forward saved
    (22-bit operands, fp32 accumulation).  Gradients agree with the f16x3 training path and with torch's layers."""
    import warnings
    from intrinsicnerf_amd import _capi, kernels, object_level as ol, packing
    dev = torch.device("cuda:0")
    fx = load_golden("object_chair_det")
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    net = ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net.load_state_dict(case_weights(fx)[0])
    rays = torch.from_numpy(fx["rays"][:6]).to(dev)
    grads, raws = {}, {}
    for prec, mlp in (("f16x3", "hip"), ("f32", "hip"), ("f32", "torch")):
        monkeypatch.setenv("INERF_PRECISION", prec)
        monkeypatch.setenv("INERF_TRAIN_MLP", mlp)
        net.zero_grad()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ret = ol.render_rays(rays, net, ol.NetworkQuery(embed, embed_d), 64, retraw=True, N_importance=16, white_bkgd=True)
        assert (type(ret["raw"].grad_fn).__name__ == "_FusedMlpFnBackward") == (mlp == "hip")
        assert type(ret["rgb_map"].grad_fn).__name__ == "_CompositeFnBackward"          # compositing: HIP forward + HIP backward in all three
        (ret["rgb_map"].square().sum() + ret["albedo_map"].sum() + ret["rgb0"].sum()).backward()
        grads[(prec, mlp)] = {k: p.grad.clone() for k, p in net.named_parameters()}
        raws[(prec, mlp)] = ret["raw"].detach().clone()
    monkeypatch.setenv("INERF_PRECISION", "f32")
    with torch.no_grad():
        eval_raw = ol.render_rays(rays, net, ol.NetworkQuery(embed, embed_d), 64, retraw=True, N_importance=16, white_bkgd=True)["raw"]
    assert torch.equal(raws[("f32", "hip")], eval_raw), "the training forward under f32 must return the exact-fp32 kernel's values"
    assert not torch.equal(raws[("f16x3", "hip")], eval_raw)
    for other in (("f16x3", "hip"), ("f32", "torch")):
        for k in grads[("f32", "hip")]:
            w = grads[other][k].double()
            assert float((grads[("f32", "hip")][k].double() - w).norm()) <= 2e-4 * float(w.norm()) + 1e-10, (other, k)


@pytest.mark.gpu
def test_network_backward_splits_large_batches(monkeypatch):
    """More sample points than one autograd node keeps: the batch is split over rays, gradients add up to the same."""
    from intrinsicnerf_amd import kernels, object_level as ol
    dev = torch.device("cuda:0")
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    net = ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net.load_state_dict(oracle.lcg_state_dict("object", 0, seed=22, sigma_gain_log2=3, freq_decay=True))
    g = torch.Generator().manual_seed(1)
    n, s = 50, 9
    d = torch.randn(n, 3, generator=g)
    rays = torch.cat([torch.rand(n, 3, generator=g), d, torch.zeros(n, 2), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
    z = torch.sort(torch.rand(n, s, generator=g) * 3 + 0.5, -1)[0].to(dev)
    cot = torch.randn(n, s, 11, generator=g).to(dev)
    out = {}
    for limit in (kernels.TRAIN_POINTS_PER_NODE, 7 * s):            # one node / eight nodes
        monkeypatch.setattr(kernels, "TRAIN_POINTS_PER_NODE", limit)
        net.zero_grad()
        raw = kernels.mlp_train(net.fused_desc(), net, rays, z)
        (raw * cot).sum().backward()
        out[limit] = (raw.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()})
    (raw_a, g_a), (raw_b, g_b) = out.values()
    assert torch.equal(raw_a, raw_b)
    for k in g_a:
        assert float((g_a[k] - g_b[k]).norm()) <= 1e-5 * float(g_a[k].norm()) + 1e-12, k


@pytest.mark.gpu
@pytest.mark.parametrize("p,m,n", [(64, 256, 256), (1000, 256, 256), (12345, 256, 64), (777, 128, 256), (4097, 128, 32), (70001, 256, 256),
                                   (44800, 256, 64), (40001, 128, 32), (30000, 128, 256), (50000, 256, 128)])    # several tiles per workgroup, every shape
def test_weight_gradient_kernel_vs_fp64(p, m, n):
    """G^T X and the column sums of G from the split-K MFMA kernel against an fp64 product: ragged point counts, gradients
    spanning four decades, every supported tile shape; also with operands that are column-aligned views of wider buffers."""
    from intrinsicnerf_amd import kernels
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(p)
    G = (torch.randn(p, 256, generator=g) * torch.logspace(-4, 0, p)[:, None]).to(dev)
    X = torch.relu(torch.randn(p, 320, generator=g)).to(dev)
    gv, xv = G[:, 256 - m:], X[:, 64:64 + n]                       # views: row strides 256 / 320, 16-byte aligned starts
    w, b = kernels.weight_gradient(gv, xv, m, n, want_bias=True)
    want_w = gv.double().t() @ xv.double()
    want_b = gv.double().sum(0)
    assert float((w.double() - want_w).norm()) <= 2e-6 * float(want_w.norm())
    assert float((b.double() - want_b).norm()) <= 2e-6 * float(want_b.norm()) + 1e-12
    # explicit ranges (upper bounds, as the training kernels deliver them) instead of the measured maxima
    ranges = torch.tensor([float(gv.abs().max()) * 3.0, 7.5e3], device=dev)
    w2 = kernels.weight_gradient(gv, xv, m, n, ranges)
    assert float((w2.double() - want_w).norm()) <= 2e-5 * float(want_w.norm())


@pytest.mark.gpu
@pytest.mark.parametrize("p", [1, 15, 17, 63, 65, 1000, 5000, 70001, 131072])
def test_weight_gradient_from_fragment_slots(p):
    """The LDS-DMA kernel of the nine 256 x 256 products (both operands FRAGMENT slots, include/inerf.h: per-point normalised
    gradients with their normalisers x activations) and the mixed forms (G fragments x row-format X, 64 columns; row-format G, 128 channels, x X fragments) against fp64 products of
    what the fragments encode - down to a single sample point (k-blocks and tiles that are mostly padding), ragged counts,
    more k-blocks than the ring is deep and than the grid is wide; gradients spanning four decades.  The row-format kernel
    on the same matrices agrees."""
    from intrinsicnerf_amd import kernels
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(100 + p)
    G = (torch.randn(p, 256, generator=g) * torch.logspace(0, -4, p)[:, None] * 2.0 ** -7 * 0.9).to(dev)
    X = torch.relu(torch.randn(p, 256, generator=g) * 3).to(dev)
    gf, gs = kernels.grad_frag_encode(G)
    xf = kernels.frag_encode(X)
    Gq, Xq = kernels.grad_frag_decode(gf, gs, p).double(), kernels.frag_decode(xf, p).double()
    assert float((Gq - G.double()).norm()) <= 1e-6 * float(G.double().norm()) and float((Xq - X.double()).norm()) <= 1e-6 * float(X.double().norm())
    want_w, want_b = Gq.t() @ Xq, Gq.sum(0)
    ranges = torch.stack([G.abs().max() * 1.7, X.abs().max()]).float()          # upper bounds, as the training kernels deliver them
    w, b = kernels.weight_gradient_frag(gf, gs, xf, ranges, p, want_bias=True)
    assert float((w.double() - want_w).norm()) <= 2e-6 * float(want_w.norm())
    assert float((b.double() - want_b).norm()) <= 2e-6 * float(want_b.norm()) + 1e-12
    w0 = kernels.weight_gradient_frag(gf, gs, xf, ranges, p)                                    # without the bias sums
    assert torch.equal(w0, w)
    # G fragments x 64 columns of row-format X (pts_linears.0 / .5 against the encoding)
    xr = X[:, 64:128]
    w64, b64 = kernels.weight_gradient_frag(gf, gs, None, ranges, p, want_bias=True, x_rows=xr, n=64)
    want64 = Gq.t() @ xr.double()
    assert float((w64.double() - want64).norm()) <= 2e-6 * float(want64.norm())
    assert float((b64.double() - want_b).norm()) <= 2e-6 * float(want_b.norm()) + 1e-12
    # several products in one launch, each split over its share of the grid: (G, X), (G, X'), (G', X) with G' = -G, X' = X rolled
    gf2, gs2 = kernels.grad_frag_encode(-G)
    assert torch.equal(gs2, gs)                                   # (one set of normalisers per gradient buffer)
    x2 = torch.roll(X, 1, dims=1).contiguous()
    xf2 = kernels.frag_encode(x2)
    # ... and one against a 64-channel fragment slot (the encoding's format), which gets a smaller share of the grid
    x64 = torch.sin(torch.arange(p * 64, dtype=torch.float32).view(p, 64) * 0.37).to(dev)
    xf64 = kernels.frag_encode(x64)
    want64f = Gq.t() @ kernels.frag_decode(xf64, p, width=64).double()
    # ... and with a 128-channel G slot (the views hidden layer's format) against 256 and 32 channels (the view encoding's)
    g128 = (G[:, 64:192] * 0.5).contiguous()
    gf128 = kernels.frag_encode(g128 / gs[:p, None])             # (one set of normalisers per gradient buffer: the 256-wide slot's)
    Gq128 = kernels.grad_frag_decode(gf128, gs, p, 128).double()
    x32 = torch.cos(torch.arange(p * 32, dtype=torch.float32).view(p, 32) * 0.73).to(dev)
    xf32 = kernels.frag_encode(x32)
    res = kernels.weight_gradient_frag_batch([gf, gf, gf2, gf, gf128, gf128], gs, [xf, xf2, xf, xf64, xf, xf32], ranges, p,
                                             x_cols=[256, 256, 256, 64, 256, 32], g_rows=[256, 256, 256, 256, 128, 128])
    wants = (want_w, Gq.t() @ kernels.frag_decode(xf2, p).double(), -want_w, want64f, Gq128.t() @ Xq, Gq128.t() @ kernels.frag_decode(xf32, p, width=32).double())
    for (wj, bj), ww, wb in zip(res, wants, (want_b, want_b, -want_b, want_b, Gq128.sum(0), Gq128.sum(0))):
        assert float((wj.double() - ww).norm()) <= 2e-6 * float(ww.norm())
        assert float((bj.double() - wb).norm()) <= 2e-6 * float(wb.norm()) + 1e-12
    # row-format G (128 channels) x X fragments (views_linears.0 against the feature layer, the semantic hidden layer against h7)
    gr = G[:, 128:256].contiguous()
    wx, bx = kernels.weight_gradient_xfrag(gr, xf, ranges, p, want_bias=True)
    wantx = gr.double().t() @ Xq
    assert float((wx.double() - wantx).norm()) <= 2e-6 * float(wantx.norm())
    assert float((bx.double() - gr.double().sum(0)).norm()) <= 2e-6 * float(gr.double().sum(0).norm()) + 1e-12
    # the row-format kernel (lane = point form) on the same 256 x 256 product
    wr, br = kernels.weight_gradient(G, X, 256, 256, want_bias=True)
    assert float((wr.double() - G.double().t() @ X.double()).norm()) <= 2e-6 * float(want_w.norm())
    assert float((br.double() - want_b).norm()) <= 2e-6 * float(want_b.norm()) + 1e-12


@pytest.mark.gpu
def test_training_batch_outside_f16_range_is_reevaluated_on_the_fp32_layer_kernels(monkeypatch):
    """The training forward reads the f16 range words of both networks ONCE, after the batch is enqueued.  A batch that trips it is
    evaluated again layer by layer in exact fp32 on the HIP kernels (layered.py; VERDICT r05 item 5: NO ATen GEMM on the path) -
    same random draws (the RNG state is put back), hence the same maps and gradients, to fp32 summation order, as a run that used
    torch's layers from the start."""
    import warnings
    from conftest import assert_same_within, aten_gemm_watch
    from intrinsicnerf_amd import object_level as ol
    dev = torch.device("cuda:0")
    fx = load_golden("object_chair_det")
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net_c, net_f = mk(), mk()
    sd_c, sd_f = case_weights(fx)
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    with torch.no_grad():
        net_f.pts_linears[2].weight.mul_(1.0e6)                  # hidden activations of the fine network far beyond 7.5e3
    rays = torch.from_numpy(fx["rays"][:9]).to(dev)
    out = {}
    for mode in ("hip", "torch"):
        monkeypatch.setenv("INERF_TRAIN_MLP", mode)
        net_c.zero_grad(); net_f.zero_grad()
        torch.manual_seed(11)
        with warnings.catch_warnings(record=True) as w, aten_gemm_watch() as watch:
            warnings.simplefilter("always")
            ret = ol.render_rays(rays, net_c, ol.NetworkQuery(embed, embed_d), 64, retraw=True, N_importance=32, network_fine=net_f,
                                 white_bkgd=True, perturb=1.0, raw_noise_std=1.0)
            (ret["rgb_map"].square().sum() + ret["acc0"].sum()).backward()
        told = any("fp32 layer kernels instead" in str(x.message) for x in w)
        assert told == (mode == "hip")
        assert (watch.gemms == []) == (mode == "hip"), f"ATen GEMMs in mode {mode}: {sorted(set(watch.gemms))}"
        out[mode] = ({k: v.detach().clone() for k, v in ret.items()},
                     {k: p.grad.clone() for k, p in list(net_c.named_parameters()) + [("f." + k, p) for k, p in net_f.named_parameters()]})
    for k in out["torch"][0]:          # (NaN disparities of empty rays included)
        assert_same_within(out["hip"][0][k], out["torch"][0][k], k)
    for k in out["torch"][1]:
        assert_same_within(out["hip"][1][k], out["torch"][1][k], "d " + k, rel=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_composite_backward_randomized_shapes(seed):
    """Compositing backward against autograd through the oracle over ragged sample counts (chunks of 64 with a partial last
    chunk, a single sample, more than four chunks), optional noise / white background / semantic and feature channels and
    cotangents on a random subset of the outputs."""
    from intrinsicnerf_amd import kernels
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(300 + seed)
    n = int(rng.randint(1, 40))
    s = int(rng.choice([1, 2, 5, 63, 64, 65, 100, 192, 257, 300]))
    c = int(rng.choice([0, 0, 1, 7, 33]))
    feat = bool(rng.rand() < 0.3) and c > 0
    wb = bool(rng.rand() < 0.5)
    with_noise = bool(rng.rand() < 0.5)
    g = torch.Generator().manual_seed(seed)
    ch = 11 + c + (128 if feat else 0)
    raw = torch.rand(n, s, ch, generator=g)
    raw[..., 3] = torch.randn(n, s, generator=g) * 2 + 0.3
    z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0]
    d = torch.randn(n, 3, generator=g)
    noise = torch.randn(n, s, generator=g) * 0.3 if with_noise else None
    cfg = oracle.RenderConfig(variant="ssr" if c > 0 else "object", white_bkgd=wb, n_classes=c, endpoint_feat=feat)
    r = raw.clone().requires_grad_(True)
    out = oracle.composite(r, z, d, cfg, noise, feat=feat)
    keys = [k for k in (SSR_KEYS if c > 0 else OBJ_KEYS) if out.get(k) is not None and k != "disp"]
    if bool((out["acc"] > 1e-3).all()):
        keys.append("disp")                      # disp is 1 / (depth / acc): only pinned away from acc ~ 0
    used = [k for k in keys if rng.rand() < 0.7] or ["rgb"]
    cot = {k: torch.randn(out[k].shape, generator=g) for k in used}
    (want,) = torch.autograd.grad(sum((cot[k] * out[k]).sum() for k in used), r)
    got = kernels.composite_backward(raw.to(dev), z.to(dev), d.to(dev), {k: v.to(dev) for k, v in cot.items()},
                                     None if noise is None else noise.to(dev), wb, c, 128 if feat else 0)
    assert_maps_close(got.cpu().numpy(), want.numpy(), 2e-4, 2e-5 * float(want.abs().max()) + 1e-12, f"seed {seed}: n={n} s={s} c={c} feat={feat}")


@pytest.mark.gpu
def test_training_step_is_deterministic():
    """No floating-point atomics anywhere in the backward (partial tiles are summed in a fixed order; the only atomics are
    integer maxima): the same step twice gives bit-identical raw outputs and parameter gradients."""
    from intrinsicnerf_amd import kernels, object_level as ol
    dev = torch.device("cuda:0")
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    net = ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net.load_state_dict(oracle.lcg_state_dict("object", 0, seed=23, sigma_gain_log2=3, freq_decay=True))
    g = torch.Generator().manual_seed(5)
    n, s = 700, 48
    d = torch.randn(n, 3, generator=g)
    rays = torch.cat([torch.rand(n, 3, generator=g), d, torch.zeros(n, 2), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
    z = torch.sort(torch.rand(n, s, generator=g) * 3 + 0.5, -1)[0].to(dev)
    cot = torch.randn(n, s, 11, generator=g).to(dev)
    runs = []
    for _ in range(2):
        net.zero_grad()
        raw = kernels.mlp_train(net.fused_desc(), net, rays, z)
        (raw * cot).sum().backward()
        runs.append((raw.detach().clone(), [p.grad.clone() for p in net.parameters()]))
    assert torch.equal(runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_short_training_run_matches_torch_layers(monkeypatch):
    """End to end: fit a student network to maps rendered by a teacher, a few Adam steps through the front-end - once with the
    HIP network backward, once with torch's layers (INERF_TRAIN_MLP=torch).  Same initial weights, same rays: the loss curves
    must coincide to 1e-3 and go down."""
    import warnings
    from intrinsicnerf_amd import object_level as ol
    dev = torch.device("cuda:0")
    fx = load_golden("object_chair_det")
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    teacher_c, teacher_f = mk(), mk()
    sd_c, sd_f = case_weights(fx)
    teacher_c.load_state_dict(sd_c); teacher_f.load_state_dict(sd_f)
    rays = torch.from_numpy(fx["rays"]).to(dev)
    query = ol.NetworkQuery(embed, embed_d)
    with torch.no_grad():
        want = ol.render_rays(rays, teacher_c, query, 64, N_importance=64, network_fine=teacher_f, white_bkgd=True)
    start_c = {k: v + 0.02 * torch.randn(v.shape, generator=torch.Generator().manual_seed(1)) for k, v in sd_c.items()}
    start_f = {k: v + 0.02 * torch.randn(v.shape, generator=torch.Generator().manual_seed(2)) for k, v in sd_f.items()}
    curves = {}
    for mode in ("hip", "torch"):
        monkeypatch.setenv("INERF_TRAIN_MLP", mode)
        net_c, net_f = mk(), mk()
        net_c.load_state_dict(start_c); net_f.load_state_dict(start_f)
        opt = torch.optim.Adam(list(net_c.parameters()) + list(net_f.parameters()), lr=2e-4)
        losses = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(12):
                ret = ol.render_rays(rays, net_c, query, 64, N_importance=64, network_fine=net_f, white_bkgd=True)
                loss = sum(((ret[k] - want[k]) ** 2).mean() for k in ("rgb_map", "albedo_map", "shading_map", "residual_map", "rgb0"))
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss))
        curves[mode] = np.array(losses)
    assert curves["hip"][-1] < 0.5 * curves["hip"][0], curves["hip"]
    np.testing.assert_allclose(curves["hip"], curves["torch"], rtol=1e-3, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("variant,c,endpoint", [("object", 0, False), ("ssr", 28, False), ("ssr", 5, True), ("ssr", 150, False)])
def test_one_call_backward_equals_library_products_of_its_own_buffers(variant, c, endpoint, monkeypatch):
    """inerf_mlp_backward (chain + every weight-gradient product + reduction + scatter in ONE C call) against the same step
    taken apart: the chain alone (inerf_mlp_backward_inputs), its fragment / row slots decoded, every product as a library GEMM
    in fp64 (kernels.mlp_weight_gradients).  Pins the fragment formats of both producers, the LDS-DMA kernel, the mixed-format
    products, the bias sums and the scatter into the reference's parameter layout; plus the entry's own contract (workspace
    check, empty batch)."""
    import ctypes as C
    from intrinsicnerf_amd import _capi, kernels, object_level as ol, packing, ssr
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    sd = oracle.lcg_state_dict(variant, c, seed=29, sigma_gain_log2=3, freq_decay=True)
    if variant == "object":
        embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
        net = ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    else:
        embed, ch = ssr.get_embedder(10, 0, scalar_factor=10); embed_d, ch_d = ssr.get_embedder(4, 0, scalar_factor=1)
        net = ssr.Semantic_NeRF(c > 0, c, D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net.load_state_dict(sd)
    n, s = 300, 47                      # 14 100 points: a ragged last tile
    d = torch.randn(n, 3, generator=g)
    rays = torch.cat([torch.rand(n, 3, generator=g) * 2 - 1, d, torch.zeros(n, 2), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
    z = torch.sort(torch.rand(n, s, generator=g) * 3 + 0.5, -1)[0].to(dev)
    chn = 11 + c + (128 if endpoint else 0)
    cot = (torch.randn(n, s, chn, generator=g) * torch.logspace(-2, 1, n)[:, None, None] * 2.0 ** -9).to(dev)
    desc = net.fused_desc()
    desc.xyz_div = embed.scalar_factor
    net.zero_grad()
    raw = kernels.mlp_train(desc, net, rays, z, endpoint)
    (raw * cot).sum().backward()
    got = {k: p.grad.clone() for k, p in net.named_parameters()}
    # the whole backward is deterministic: fixed K-slices, partial tiles summed in a fixed order, no atomics on the data path
    net.zero_grad()
    (kernels.mlp_train(desc, net, rays, z, endpoint) * cot).sum().backward()
    for k, p in net.named_parameters():
        assert torch.equal(p.grad, got[k]), f"{k}: a second identical step gave other gradients"
    # the same step taken apart
    d16 = _capi.NetDesc(desc.variant, desc.n_classes, desc.l_xyz, desc.l_dir, desc.xyz_div, _capi.PREC_F16X3)
    named = dict(net.named_parameters())
    names = tuple(name for name, _ in packing.tensor_table(d16))
    pf, pb = packing.device_packer(d16, False, dev)(named), packing.device_packer(d16, True, dev)(named)
    with torch.no_grad():
        raw2, save = kernels.encode_mlp_train(d16, pf, rays, z, endpoint)
        d2 = cot.reshape(n * s, chn).contiguous()
        dz, heads = kernels.mlp_backward_inputs(d16, pb, raw2.view(n * s, chn), d2, save, endpoint, want_heads=True)
        X = kernels.save_slot_views(d16, save, n * s)
        G = kernels.save_slot_views(d16, dz, n * s, gradient=True)
        want = kernels.mlp_weight_gradients(d16, names, save, dz, d2, n * s, endpoint, heads)
        # fp64 products of the decoded slots for the square layers (the library's fp32 GEMMs are 1e-6 themselves)
        for i in range(1, 8):
            w64 = (G[kernels.SAVE_H0 + i].double().t() @ X[kernels.SAVE_H0 + i - 1].double())
            name = f"pts_linears.{i}.weight"
            ref = w64 if i != 5 else torch.cat([want[name][:, :63].double(), w64], 1)
            assert float((got[name].double() - ref).norm()) <= 3e-6 * float(ref.norm()), name
    assert torch.equal(raw.detach(), raw2)
    for k in got:
        a, b = got[k].double(), want[k].double()
        assert a.shape == b.shape and float((a - b).norm()) <= 2e-5 * float(b.norm()) + 1e-12, (k, float((a - b).norm()), float(b.norm()))
    lib = _capi.lib()
    d16 = _capi.NetDesc(desc.variant, desc.n_classes, desc.l_xyz, desc.l_dir, desc.xyz_div, _capi.PREC_F16X3)
    n_params = lib.inerf_param_floats(d16)
    assert n_params == sum(p.numel() for p in net.parameters())
    grads = torch.full((n_params,), 7.0, device=dev)
    assert lib.inerf_mlp_backward(d16, None, None, None, None, None, 0, 0, C.c_void_p(grads.data_ptr()), None, 0, None, None) == _capi.OK
    torch.cuda.synchronize()
    assert float(grads.abs().max()) == 0.0                                   # empty batch: every gradient is zero
    need = lib.inerf_mlp_backward_workspace_bytes(d16, n * s)
    assert need > 4 * n * s * 2784                                            # holds the pre-activation gradients of every layer
    one = torch.zeros(1, device=dev)
    rc = lib.inerf_mlp_backward(d16, C.c_void_p(one.data_ptr()), C.c_void_p(one.data_ptr()), C.c_void_p(one.data_ptr()), C.c_void_p(one.data_ptr()),
                                C.c_void_p(one.data_ptr()), n * s, 0, C.c_void_p(grads.data_ptr()), C.c_void_p(one.data_ptr()), need - 1, None, None)
    assert rc == _capi.E_WORKSPACE


@pytest.mark.gpu
def test_unused_outputs_of_an_empty_ray_do_not_poison_the_gradients(monkeypatch):
    """A ray on which every density is <= 0 has acc == 0 and disp == NaN (run_nerf.py:404, as in the reference).  A loss that
    does not use disp must still give finite gradients - the reference's autograd never visits disp's branch then.  (Autograd
    Functions materialise unused outputs' cotangents as ZEROS by default, and 0 x d disp / d acc is NaN at acc == 0: found by
    scripts/fit_synthetic.py, whose trained network has empty rays in every batch.)"""
    from intrinsicnerf_amd import kernels
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    n, s = 6, 64
    raw = torch.randn(n, s, 11, generator=g)
    raw[0, :, 3] = -raw[0, :, 3].abs() - 0.1                       # ray 0: empty
    raw[1, :, 3] = 0.0                                             # ray 1: sigma == 0 exactly
    z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0]
    d = torch.randn(n, 3, generator=g)
    for wb in (False, True):
        r = raw.clone().to(dev).requires_grad_(True)
        out = kernels.composite(r, z.to(dev), d.to(dev), None, wb)
        assert torch.isnan(out["disp"][0]) and float(out["acc"][0]) == 0.0
        (out["rgb"].square().sum() + out["albedo"].sum()).backward()
        assert torch.isfinite(r.grad).all()
        rc = raw.clone().requires_grad_(True)
        want = oracle.composite(rc, z, d, oracle.RenderConfig(variant="object", white_bkgd=wb))
        (want["rgb"].square().sum() + want["albedo"].sum()).backward()
        assert_maps_close(r.grad.cpu().numpy(), rc.grad.numpy(), 1e-4, 1e-6, "d_raw with unused disp")
        # ... and a loss that DOES use disp is NaN on that ray in the reference's autograd too
        r2 = raw.clone().to(dev).requires_grad_(True)
        kernels.composite(r2, z.to(dev), d.to(dev), None, wb)["disp"][2:].sum().backward()
        assert torch.isfinite(r2.grad[2:]).all()
