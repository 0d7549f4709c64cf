"""GPU: the f16 range guard of the split-precision MLP kernel through the front-ends' chunk loops.

* A frame in which ONE chunk leaves f16's range costs one extra exact-fp32 render of THAT chunk (the
  fallback used to re-render the whole frame): the other chunks keep their f16x3 results bit for bit, the tripped chunk
  equals a pure fp32 render bit for bit, and the chunk loop issues exactly one extra ``render_rays`` call.
* A training step that trips the guard INSIDE a frame loop (``render()`` -> ``batchify_rays`` -> ``render_rays`` under
  autograd) is re-evaluated by its own handler - RNG restored, same draws, torch layers - not by the frame loop's
  (the nested block used to hand its words outwards, so the frame loop re-ran the call and drew again).
"""
import warnings

import numpy as np
import pytest
import torch

from _cases import case_weights
from conftest import assert_same_within, load_golden

pytestmark = pytest.mark.gpu


def _scene(dev, H=40, W=40):
    import bench
    from intrinsicnerf_amd import object_level as ol
    focal = 0.5 * W / np.tan(0.5 * bench.CAMERA_ANGLE_X)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    pose = bench.chair_pose()
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net_c, net_f = mk(), mk()
    sd_c, sd_f = case_weights(load_golden("object_chair_det"))
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    return K, pose, net_c, net_f, ol.NetworkQuery(embed, embed_d), focal


@pytest.mark.parametrize("coalesce", [False, True])
def test_only_the_tripped_chunk_is_rendered_again_in_fp32(coalesce, monkeypatch):
    """``coalesce``: eval-mode frames go through in one launch sequence where the workspace cap allows (kernels.coalesced_chunk), with
    one f16 range word per CALLER's chunk (inerf_encode_mlp_chunked): the frame is ONE f16x3 call, the chunk that left the range - and
    only that one - one more call in exact fp32, the final frame the same bits as with the caller's chunks."""
    import bench
    from intrinsicnerf_amd import _capi, object_level as ol
    monkeypatch.setenv("INERF_PRECISION", "f16x3")
    if not coalesce:
        monkeypatch.setenv("INERF_COALESCE_BYTES", "0")
    dev = torch.device("cuda:0")
    H = W = 40
    K, pose, net_c, net_f, query, focal = _scene(dev, H, W)
    rows_per_chunk = 4
    chunk = rows_per_chunk * W                                                   # 10 chunks of 4 image rows
    # the fine network overflows where a point lies more than y_min above the optical axis: only rays of the top 4 rows get there
    slope = lambda row: (0.5 * H - row) / focal
    y_min = 6.0 * 0.5 * (slope(rows_per_chunk - 1) + slope(rows_per_chunk))
    big = bench.overflow_above(net_f, pose, y_min)
    kw = dict(network_fn=net_c, network_fine=big, network_query_fn=query, N_samples=64, N_importance=128, white_bkgd=True,
              perturb=False, raw_noise_std=0., use_viewdirs=True, ndc=False, lindisp=False, near=2.0, far=6.0)
    calls = []
    real = ol.render_rays
    monkeypatch.setattr(ol, "render_rays", lambda rb, **k: (calls.append((rb.shape[0], _capi.default_precision())), real(rb, **k))[1])
    ro, rd = ol.get_rays(H, W, K, pose.to(dev))
    flat = lambda t, first=0: t[first:].reshape(-1, 3)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = ol.render(H, W, K, chunk=chunk, rays=(flat(ro), flat(rd)), **kw)
        whole = [(H * W, _capi.PREC_F16X3)] if coalesce else [(chunk, _capi.PREC_F16X3)] * 10
        assert calls == whole + [(chunk, _capi.PREC_F32)], calls                               # one extra call, in exact fp32
        calls.clear()
        monkeypatch.setenv("INERF_PRECISION", "f32")
        exact = ol.render(H, W, K, chunk=chunk, rays=(flat(ro), flat(rd)), **kw)
        monkeypatch.setenv("INERF_PRECISION", "f16x3")
        rest = ol.render(H, W, K, chunk=chunk, rays=(flat(ro, rows_per_chunk), flat(rd, rows_per_chunk)), **kw)
    n_exact = 1 if coalesce else 10                                               # (the pure-fp32 frame's calls come first)
    assert all(p == _capi.PREC_F16X3 for _, p in calls[n_exact:]) and len(calls) == n_exact + (1 if coalesce else 9), "the other rows alone must not trip"
    for i, name in enumerate(("rgb", "disp", "acc", "albedo", "shading", "residual")):
        g, e, r = got[i], exact[i], rest[i]
        assert torch.isfinite(g[~torch.isnan(e)]).all()
        torch.testing.assert_close(g[:chunk], e[:chunk], rtol=0, atol=0, equal_nan=True, msg=name + " (tripped chunk = pure fp32)")
        torch.testing.assert_close(g[chunk:], r, rtol=0, atol=0, equal_nan=True, msg=name + " (other chunks keep f16x3)")
    # and the same through the SSR chunk loop (ssr.batchify_rays): tags = chunk indices, only chunk 1 of 3 trips
    from intrinsicnerf_amd import kernels, ssr
    words = {0: 0, 1: _capi.STATUS_F16_RANGE, 2: 0}
    seen = []

    def fake_render(rays):
        j = int(rays[0, 0].item())
        seen.append((j, _capi.default_precision()))
        if _capi.default_precision() == _capi.PREC_F16X3:
            kernels.check_f16_range(torch.full((1,), words[j], dtype=torch.int32, device=dev), "chunk", deferrable=True)
        return {"x": rays[:, :1] + (100.0 if _capi.default_precision() == _capi.PREC_F32 else 0.0)}

    rays = torch.arange(3, device=dev, dtype=torch.float32).repeat_interleave(5)[:, None].expand(15, 11).contiguous()
    out = ssr.batchify_rays(fake_render, rays, chunk=5)
    assert seen == [(0, _capi.PREC_F16X3), (1, _capi.PREC_F16X3), (2, _capi.PREC_F16X3), (1, _capi.PREC_F32)]
    assert out["x"].flatten().tolist() == [0.0] * 5 + [101.0] * 5 + [2.0] * 5


@pytest.mark.parametrize("through", ["render_rays", "render"])
def test_training_step_that_trips_inside_a_frame_loop_keeps_the_rng_stream(through, monkeypatch):
    """hip (which falls back to the fp32 layer kernels) vs torch layers from the start: same maps, same gradients (to fp32
    summation order), same RNG position afterwards - called directly and
    through render() (whose chunk loop has its own deferred block around the training step's)."""
    from intrinsicnerf_amd import object_level as ol
    monkeypatch.setenv("INERF_PRECISION", "f16x3")
    dev = torch.device("cuda:0")
    K, pose, net_c, net_f, query, _ = _scene(dev)
    with torch.no_grad():
        net_f.pts_linears[2].weight.mul_(1.0e6)                  # hidden activations of the fine network far beyond 7.5e3
    fx = load_golden("object_chair_det")
    rays = torch.from_numpy(fx["rays"][:9]).to(dev)
    out = {}
    for mode in ("hip", "torch"):
        monkeypatch.setenv("INERF_TRAIN_MLP", mode)
        net_c.zero_grad(); net_f.zero_grad()
        torch.manual_seed(11)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            kw = dict(network_fn=net_c, network_query_fn=query, N_samples=64, retraw=True, N_importance=32, network_fine=net_f,
                      white_bkgd=True, perturb=1.0, raw_noise_std=1.0)
            if through == "render_rays":
                ret = ol.render_rays(rays, **kw)
            else:        # three chunks of three rays, each a training step inside batchify_rays' block
                r = ol.render(40, 40, K, chunk=3, rays=(rays[:, 0:3], rays[:, 3:6]), near=2.0, far=6.0, use_viewdirs=True, ndc=False, **kw)
                ret = {"rgb_map": r[0], "acc0": r[6]["acc0"], "raw": r[6]["raw"]}
        told = sum("fp32 layer kernels instead" in str(x.message) for x in w)
        assert (told > 0) == (mode == "hip")
        assert not any("exact fp32 MFMA kernel" in str(x.message) for x in w), "the frame loop's fallback must not run for a training step"
        (ret["rgb_map"].square().sum() + ret["acc0"].sum()).backward()
        out[mode] = ({k: v.detach().clone() for k, v in ret.items()},
                     {k: p.grad.clone() for k, p in list(net_c.named_parameters()) + [("f." + k, p) for k, p in net_f.named_parameters()]},
                     torch.rand(4, device=dev))                          # where the RNG stream stands afterwards
    for k in out["torch"][0]:          # the fallback = the fp32 layer kernels (layered.py): torch's layers to fp32 summation order
        assert_same_within(out["hip"][0][k], out["torch"][0][k], k)
    for k in out["torch"][1]:
        assert_same_within(out["hip"][1][k], out["torch"][1][k], "d " + k, rel=1e-3)
    assert torch.equal(out["hip"][2], out["torch"][2]), "the fallback consumed the RNG differently"
