"""Networks outside the fused architecture (any netdepth / netwidth / skips, use_viewdirs=False) and the exact-fp32 evaluation of a
training batch: ``csrc/layered.hip`` through ``intrinsicnerf_amd.layered`` and the front-ends, against

  * vectors from the REAL reference (tests/golden/layered_*.npz, make_golden_layered.py: the reference's own ``run_network`` +
    ``NeRF.forward`` / ``Semantic_NeRF.forward`` and its autograd, object_level/run_nerf_helpers.py:284-321,
    SSR/models/semantic_nerf.py:120-181),
  * a plain fp64 torch evaluation of the same op for the kernel-level cases.

Tolerance: north_star's ``1e-5 + 1e-4 |want|`` for values; gradients per tensor, relative to the tensor's norm (1e-4).
"""
import warnings

import numpy as np
import pytest
import torch

from conftest import aten_gemm_watch, golden_names, load_golden

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-5, 1e-4


def close(got, want, what, atol=ATOL, rtol=RTOL):
    got, want = got.detach().double().cpu(), torch.as_tensor(want).double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.equal(torch.isnan(got), torch.isnan(want)), f"{what}: NaN pattern"
    err = (got - want).abs() - (atol + rtol * want.abs())
    worst = float(torch.nan_to_num(err, nan=-1.0).max())
    assert worst <= 0, f"{what}: {worst:.3e} over the tolerance (max |diff| {float(torch.nan_to_num(got - want).abs().max()):.3e})"


def grad_close(got, want, what, rel=1e-4):
    got, want = got.detach().double().cpu(), torch.as_tensor(want).double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    dev = float((got - want).norm()) / max(float(want.norm()), 1e-30)
    assert dev <= rel, f"{what}: relative deviation {dev:.3e}"


# ----------------------------------------------------------------------------------------------
# kernel level: inerf_linear, inerf_linear_wgrad, inerf_embed
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (77, 3, 128), (1000, 28, 63), (513, 128, 283), (4099, 256, 319), (300, 80, 119), (31, 257, 16)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_forward_matches_fp64(m, n, k, act):
    from intrinsicnerf_amd import layered
    g = torch.Generator().manual_seed(m * 31 + n * 7 + k)
    x = torch.randn(m, k + 5, generator=g).cuda()              # the operand is a column range of a wider buffer
    w = (torch.randn(n, k, generator=g) / np.sqrt(k)).cuda()
    b = torch.randn(n, generator=g).cuda()
    out = torch.full((m, n + 3), 7.0, device="cuda")
    layered.linear(layered.Cols(x, 2, k), w, b, layered.Cols(out, 1, n), act)
    want = x[:, 2:2 + k].double() @ w.double().t() + b.double()
    want = torch.relu(want) if act == 1 else torch.sigmoid(want) if act == 2 else want
    close(out[:, 1:1 + n], want.cpu(), f"linear {m}x{n}x{k} act {act}", atol=1e-5, rtol=1e-5)
    assert float(out[:, 0].min()) == 7.0 and float(out[:, n + 1:].min()) == 7.0, "wrote outside its columns"


@pytest.mark.parametrize("m,o,k,col0", [(500, 128, 283, 0), (1031, 256, 319, 63), (64, 3, 128, 0), (200, 1, 256, 0), (300, 80, 119, 39)])
def test_linear_input_gradient_with_add_and_gate(m, o, k, col0):
    from intrinsicnerf_amd import layered
    g = torch.Generator().manual_seed(o + k)
    dz = torch.randn(m, o, generator=g).cuda()
    w = (torch.randn(o, k, generator=g) / np.sqrt(o)).cuda()
    width = k - col0
    add = torch.randn(m, width, generator=g).cuda()
    gate = torch.randn(m, width, generator=g).cuda()
    out = torch.empty(m, width, device="cuda")
    layered.linear_dgrad(layered.Cols(dz), w, layered.Cols(out), col0=col0, add=layered.Cols(add), gate=layered.Cols(gate))
    want = (dz.double() @ w.double()[:, col0:] + add.double()) * (gate > 0)
    close(out, want.cpu(), "input gradient", atol=1e-5, rtol=1e-5)
    # accumulate in place (add == out), no gate
    out2 = add.clone()
    layered.linear_dgrad(layered.Cols(dz), w, layered.Cols(out2), col0=col0, add=layered.Cols(out2))
    close(out2, (dz.double() @ w.double()[:, col0:] + add.double()).cpu(), "accumulated input gradient", atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("n,rows,cols", [(1, 1, 1), (257, 3, 128), (5000, 1, 256), (40000, 256, 319), (12345, 80, 119), (3000, 28, 64)])
def test_linear_weight_gradient_matches_fp64_and_repeats(n, rows, cols):
    from intrinsicnerf_amd import layered
    g = torch.Generator().manual_seed(n + rows)
    dz = torch.randn(n, rows + 2, generator=g).cuda()
    x = torch.randn(n, cols + 3, generator=g).cuda()
    dw, db = layered.linear_wgrad(layered.Cols(dz, 1, rows), layered.Cols(x, 3, cols))
    want_w = dz[:, 1:1 + rows].double().t() @ x[:, 3:].double()
    want_b = dz[:, 1:1 + rows].double().sum(0)
    scale = float(np.sqrt(n))
    close(dw, want_w.cpu(), "d_weight", atol=3e-6 * scale, rtol=3e-6)
    close(db, want_b.cpu(), "d_bias", atol=3e-6 * scale, rtol=3e-6)
    dw2, db2 = layered.linear_wgrad(layered.Cols(dz, 1, rows), layered.Cols(x, 3, cols))
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "weight gradient is not bit-identical run to run"


def test_embed_matches_the_encoder_definition():
    from intrinsicnerf_amd import layered, object_level as ol
    g = torch.Generator().manual_seed(3)
    rays = torch.randn(50, 11, generator=g).cuda()
    z = (torch.rand(50, 7, generator=g) * 4 + 2).cuda()
    for l, div in ((10, 1.0), (0, 1.0), (8, 10.0), (3, 1.0)):
        e = ol.Embedder(l, scalar_factor=div)
        out = torch.zeros(50 * 7, e.out_dim + 2, device="cuda")
        spec = type("S", (), dict(l_xyz=l, xyz_div=div, l_dir=l))()
        src = layered.RaySource(spec, rays, z)
        src.xyz_into(layered.Cols(out, 1, e.out_dim))
        pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3)
        # the encoder's argument reaches 2^9 * |x|: compare against an fp64 evaluation of the SAME fp32 arguments
        x = (pts.cpu() / div) if div != 1.0 else pts.cpu()     # (on the CPU: torch's GPU division by a scalar multiplies by 1 / div)
        bands = [x.double()] + [f(x.double() * float(2 ** k)) for k in range(l) for f in (torch.sin, torch.cos)]
        close(out[:, 1:1 + e.out_dim], torch.cat(bands, -1).cpu(), f"embed L={l} div={div}", atol=2e-7, rtol=0)
        if div == 1.0:
            outd = torch.zeros(50 * 7, e.out_dim, device="cuda")
            src.dir_into(layered.Cols(outd))
            d = rays[:, None, 8:11].expand(50, 7, 3).reshape(-1, 3)
            bands = [d.double()] + [f(d.double() * float(2 ** k)) for k in range(l) for f in (torch.sin, torch.cos)]
            close(outd, torch.cat(bands, -1).cpu(), f"embed dirs L={l}", atol=2e-7, rtol=0)


# ----------------------------------------------------------------------------------------------
# network level, against the real reference's vectors
# ----------------------------------------------------------------------------------------------
def _build(fx):
    """This package's mirror of the fixture's network + encoders, the reference's weights loaded through load_state_dict."""
    from intrinsicnerf_amd import object_level as ol, ssr
    D, W, skips = int(fx["D"]), int(fx["W"]), [int(i) for i in fx["skips"]]
    l_xyz, l_dir, views = int(fx["l_xyz"]), int(fx["l_dir"]), bool(fx["use_viewdirs"])
    if str(fx["variant"]) == "ssr":
        embed, ch = ssr.get_embedder(l_xyz, 0, scalar_factor=float(fx["xyz_div"]))
        embed_d, ch_d = ssr.get_embedder(l_dir, 0, scalar_factor=1)
        net = ssr.Semantic_NeRF(True, int(fx["n_classes"]), D=D, W=W, input_ch=ch, output_ch=5, skips=skips, input_ch_views=ch_d,
                                use_viewdirs=True)
    else:
        embed, ch = ol.get_embedder(l_xyz, 0)
        embed_d, ch_d = ol.get_embedder(l_dir, 0) if views else (None, 0)
        net = ol.NeRF(D=D, W=W, input_ch=ch, output_ch=5, skips=skips, input_ch_views=ch_d, use_viewdirs=views)
    sd = {k[len("param/"):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("param/")}
    net.load_state_dict(sd)
    return net.cuda(), embed, embed_d


@pytest.mark.parametrize("name", golden_names("layered_"))
def test_network_forward_and_gradients_match_the_reference(name):
    from intrinsicnerf_amd import layered
    fx = load_golden(name)
    net, embed, embed_d = _build(fx)
    spec = layered.spec_for(net, embed, embed_d)
    assert spec is not None and net.fused_desc() is None, "the fixture's network must be outside the fused architecture"
    rays, z = torch.from_numpy(fx["rays"]).cuda(), torch.from_numpy(fx["z"]).cuda()
    endpoint = bool(fx.get("endpoint", False))
    with torch.no_grad():
        raw = layered.evaluate(spec, net, rays, z, endpoint)
    close(raw, fx["raw"], f"{name}: raw")
    # gradients of sum(cot * raw) w.r.t. every parameter the forward reads: the reference's autograd
    raw = layered.evaluate(spec, net, rays, z, endpoint)
    assert raw.requires_grad
    (torch.from_numpy(fx["cot"]).cuda() * raw).sum().backward()
    seen = 0
    for k, p in net.named_parameters():
        want = fx.get("grad/" + k)
        if want is None:
            assert p.grad is None, f"{k}: the reference's forward does not read it"
            continue
        grad_close(p.grad, want, f"{name}: d {k}")
        seen += 1
    assert seen == sum(k.startswith("grad/") for k in fx)
    # ... and the same through torch's own layers on the GPU (same forward values => same ReLU masks): tighter
    net.zero_grad()
    from intrinsicnerf_amd.object_level import _run_network_torch
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., :, None]
    call = (lambda x: net(x, True)) if endpoint else net
    raw_t = _run_network_torch(pts, rays[:, 8:11] if embed_d is not None else None, call, embed, embed_d, 65536)
    close(raw.detach(), raw_t.detach().cpu(), f"{name}: raw vs torch layers")


def test_front_ends_route_foreign_shapes_to_the_layer_kernels(monkeypatch):
    """object_level.render_rays / run_network with NeRF(D=4, W=128, skips=[2]): no ATen GEMM runs (the dispatcher is watched), the
    maps agree with the same networks through torch's layers (INERF_LAYERED=0), forward and parameter gradients."""
    from intrinsicnerf_amd import object_level as ol
    Watch = aten_gemm_watch

    fx = load_golden("layered_object_d4_w128")
    net_c, embed, embed_d = _build(fx)
    net_f, _, _ = _build(fx)
    with torch.no_grad():
        for p in net_f.parameters():
            p.mul_(1.1)
    q = ol.NetworkQuery(embed, embed_d)
    rays = torch.from_numpy(fx["rays"]).cuda()
    kw = dict(network_fn=net_c, network_query_fn=q, N_samples=32, N_importance=32, network_fine=net_f, white_bkgd=True, perturb=0.,
              raw_noise_std=0., retraw=True)
    keys = ("rgb_map", "albedo_map", "shading_map", "residual_map", "acc_map", "rgb0", "raw")

    def step():
        for n in (net_c, net_f):
            n.zero_grad()
        out = ol.render_rays(rays, **kw)
        sum((out[k] ** 2).sum() for k in keys if k != "raw").backward()
        return {k: out[k].detach().clone() for k in keys}, {f"{t}/{k}": p.grad.clone() for t, n in (("c", net_c), ("f", net_f))
                                                            for k, p in n.named_parameters() if p.grad is not None}

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with Watch() as w:
            maps, grads = step()
            with torch.no_grad():
                ev = ol.render_rays(rays, **kw)
                pts = torch.rand(5, 9, 3, device="cuda")
                rn = ol.run_network(pts, rays[:5, 8:11], net_c, embed, embed_d)
        assert w.gemms == [], f"ATen GEMMs on the path: {sorted(set(w.gemms))}"
        monkeypatch.setenv("INERF_LAYERED", "0")
        with Watch() as w0:
            maps_t, grads_t = step()
            with torch.no_grad():
                rn_t = ol.run_network(pts, rays[:5, 8:11], net_c, embed, embed_d)
        assert w0.gemms, "the comparison run was meant to go through torch's layers"
    for k in keys:
        # the coarse pass at the plain tolerance; behind sample_pdf (which amplifies the last bit of the coarse weights on a
        # freshly initialised network) two fp32 evaluations of the layers agree to ~1e-3 on the fine maps
        fine = k != "rgb0"
        if k != "raw":          # (a resampled depth that lands in another bin moves that sample's raw row altogether)
            close(maps[k], maps_t[k].cpu(), f"render_rays {k} (layer kernels vs torch layers)", atol=2e-3 if fine else 2e-5,
                  rtol=2e-3 if fine else 1e-4)
        close(ev[k], maps[k].cpu(), f"eval-mode {k} vs training-mode forward", atol=0, rtol=0)
    close(rn, rn_t.cpu(), "run_network on arbitrary points")
    assert set(grads) == set(grads_t) and len(grads) > 40
    for k in grads:
        grad_close(grads[k], grads_t[k].cpu(), f"d {k}", rel=5e-3)      # (network-level gradients are held to 1e-4 by the reference's vectors above)


def test_ssr_front_end_with_another_netwidth_uses_the_layer_kernels():
    from intrinsicnerf_amd import layered, ssr
    fx = load_golden("layered_ssr_d5_w64_c5")
    net, embed, embed_d = _build(fx)
    rays, z = torch.from_numpy(fx["rays"]).cuda(), torch.from_numpy(fx["z"]).cuda()
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., :, None]
    with torch.no_grad():
        raw = ssr.run_network(pts, rays[:, 8:11], net, embed, embed_d, show_endpoint=True)
        raw_plain = ssr.run_network(pts, rays[:, 8:11], net, embed, embed_d)
    # the points were formed in torch here (o + d z, rounded once) exactly as the fixture's were
    close(raw, fx["raw"], "ssr.run_network(show_endpoint=True)")
    close(raw_plain, fx["raw"][..., :16], "ssr.run_network")
    assert layered.spec_for(net, embed, embed_d).channels(True) == 48


def test_exact_fp32_training_of_the_default_architecture_on_the_layer_kernels(monkeypatch):
    """INERF_TRAIN_MLP=layered: the D=8, W=256 networks trained in fp32 throughout, like the reference (run_nerf.py:942-1018) - no ATen
    GEMM, maps and parameter gradients equal to torch's layers (INERF_TRAIN_MLP=torch) to fp32 summation order."""
    from _cases import case_weights
    from conftest import assert_same_within
    from intrinsicnerf_amd import object_level as ol
    fx = load_golden("object_chair_det")
    embed, ch = ol.get_embedder(10, 0)
    embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).cuda()
    net_c, net_f = mk(), mk()
    sd_c, sd_f = case_weights(fx)
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    rays = torch.from_numpy(fx["rays"][:16]).cuda()
    out = {}
    for mode in ("layered", "torch"):
        monkeypatch.setenv("INERF_TRAIN_MLP", mode)
        net_c.zero_grad(); net_f.zero_grad()
        with warnings.catch_warnings(), aten_gemm_watch() as watch:
            warnings.simplefilter("ignore")
            ret = ol.render_rays(rays, net_c, ol.NetworkQuery(embed, embed_d), 64, N_importance=128, network_fine=net_f, white_bkgd=True,
                                 perturb=0., raw_noise_std=0.)
            (ret["rgb_map"].square().sum() + ret["rgb0"].square().sum() + ret["acc_map"].sum()).backward()
        assert (watch.gemms == []) == (mode == "layered"), sorted(set(watch.gemms))
        out[mode] = ({k: v.detach().clone() for k, v in ret.items()},
                     {k: p.grad.clone() for k, p in list(net_c.named_parameters()) + [("f." + k, p) for k, p in net_f.named_parameters()]})
    for k in ("rgb0", "acc0", "albedo0"):                      # the coarse pass (in front of sample_pdf): plain tolerance
        close(out["layered"][0][k], out["torch"][0][k].cpu(), k)
    for k in ("rgb_map", "acc_map"):
        assert_same_within(out["layered"][0][k], out["torch"][0][k], k, rel=2e-3)
    for k in out["torch"][1]:
        grad_close(out["layered"][1][k], out["torch"][1][k].cpu(), "d " + k, rel=5e-3)


def test_empty_and_single_point_batches():
    """0 rays (render() of an empty ray set, run_nerf.py:59-71 with nothing to do) and one point: shapes as the reference's, zero gradients for
    the empty batch."""
    from intrinsicnerf_amd import layered
    fx = load_golden("layered_object_d4_w128")
    net, embed, embed_d = _build(fx)
    spec = layered.spec_for(net, embed, embed_d)
    rays, z = torch.from_numpy(fx["rays"]).cuda(), torch.from_numpy(fx["z"]).cuda()
    raw0 = layered.evaluate(spec, net, rays[:0], z[:0])
    assert raw0.shape == (0, z.shape[1], 11) and raw0.requires_grad
    raw0.sum().backward()
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in net.parameters())
    with torch.no_grad():
        raw1 = layered.evaluate(spec, net, rays[:1], z[:1, :1])
    close(raw1, fx["raw"][:1, :1], "one point")
