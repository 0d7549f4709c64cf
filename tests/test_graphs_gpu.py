"""GPU: the training step as HIP graphs (intrinsicnerf_amd/graphs.py) against the same step issued launch by launch.

The library's kernels are launched through hipLaunchKernelGGL on the stream torch hands over, so stream capture turns them
into graph nodes; these tests pin that the replayed graphs compute exactly what the eager step computes (same kernels, same
inputs, deterministic reductions: bit for bit), that a new batch per step flows through the static inputs, and that a batch
which trips the f16 range guard never reaches the optimizer through the graph."""
import warnings

import numpy as np
import pytest
import torch

from _cases import case_weights
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _setup(dev, seed_shift=0.0):
    from intrinsicnerf_amd import object_level as ol
    fx = load_golden("object_chair_det")
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net_c, net_f = mk(), mk()
    sd_c, sd_f = case_weights(fx)
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    rays = torch.from_numpy(fx["rays"]).to(dev)
    return ol, net_c, net_f, ol.NetworkQuery(embed, embed_d), rays


@pytest.mark.parametrize("perturb", [0.0, 1.0])
def test_graphed_step_equals_the_eager_step(perturb, monkeypatch):
    from intrinsicnerf_amd import graphs
    monkeypatch.setenv("INERF_PRECISION", "f16x3")
    dev = torch.device("cuda:0")
    results = {}
    for mode in ("eager", "graph"):
        ol, net_c, net_f, query, rays = _setup(dev)
        n = 12
        batches = [rays[i:i + n] for i in (0, 7, 3, 11)]
        targets = [torch.rand(n, 3, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(4)]
        opt = torch.optim.Adam(list(net_c.parameters()) + list(net_f.parameters()), lr=1e-4, capturable=True)

        def loss_fn(r, t):
            ret = ol.render_rays(r, net_c, query, 64, retraw=True, perturb=perturb, N_importance=64, network_fine=net_f, white_bkgd=True)
            return ((ret["rgb_map"] - t) ** 2).mean() + ((ret["rgb0"] - t) ** 2).mean() + 0.01 * ret["albedo_map"].abs().mean()

        losses = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if mode == "graph":
                step = graphs.GraphedTrainStep(loss_fn, (batches[0], targets[0]), opt)      # its warm-up steps must leave no trace
                assert step.status is not None, "the step's f16 range words must be part of the graph"
            torch.manual_seed(5)
            for it, (r, t) in enumerate(zip(batches, targets)):
                # the reference's schedule: a new rate written into the group every iteration (run_nerf.py:1023-1027, trainer.py:1005-1009).
                # A steep one, so that a rate baked into the captured optimizer at its capture-time value would show in the parameters
                for group in opt.param_groups:      # (eager: a device scalar too, so that both runs feed Adam the same fp32 rate)
                    lr = 1e-4 * (0.1 ** (it / 2.0))
                    group["lr"] = torch.tensor(lr, dtype=torch.float32, device=dev) if mode == "eager" else lr
                if mode == "eager":
                    opt.zero_grad(set_to_none=True)
                    loss = loss_fn(r, t)
                    loss.backward()
                    opt.step()
                else:
                    loss = step(r, t)
                losses.append(float(loss))
        if mode == "graph":
            assert step.fallbacks == 0
            assert all(float(g["lr"]) == pytest.approx(1e-4 * 0.1 ** 1.5, rel=1e-6) for g in opt.param_groups)
        results[mode] = (losses, [p.detach().clone() for p in list(net_c.parameters()) + list(net_f.parameters())])
    assert np.isfinite(results["graph"][0]).all()
    if perturb == 0.0:          # no random draws: the replayed graphs ARE the eager step
        assert results["eager"][0] == results["graph"][0], (results["eager"][0], results["graph"][0])
        for a, b in zip(results["eager"][1], results["graph"][1]):
            assert torch.equal(a, b)
    else:                       # jitter comes from the graph-safe generator: another sample of the same distribution
        np.testing.assert_allclose(results["eager"][0], results["graph"][0], rtol=0.2)


def test_checkpoint_of_a_graphed_optimizer_loads_into_an_eager_one(monkeypatch, tmp_path):
    """run_nerf.py:1035-1043 / trainer.py:1042-1047 save optimizer.state_dict() next to the networks.  The graphed step keeps every
    group's learning rate in a device tensor; ``optimizer_state_dict()`` writes what an eager optimizer would have written (float
    rates, host step counts), ``close()`` hands the live optimizer back the same way, and a fresh eager Adam that loads the file
    continues with the same update an eager Adam that had done those steps itself would make."""
    from intrinsicnerf_amd import graphs
    monkeypatch.setenv("INERF_PRECISION", "f16x3")
    dev = torch.device("cuda:0")
    ol, net_c, net_f, query, rays = _setup(dev)
    params = list(net_c.parameters()) + list(net_f.parameters())
    opt = torch.optim.Adam(params, lr=2e-4, capturable=True)
    r, t = rays[:12], torch.rand(12, 3, generator=torch.Generator().manual_seed(0)).to(dev)

    def loss_fn(r, t):
        ret = ol.render_rays(r, net_c, query, 64, retraw=True, perturb=0.0, N_importance=64, network_fine=net_f, white_bkgd=True)
        return ((ret["rgb_map"] - t) ** 2).mean() + ((ret["rgb0"] - t) ** 2).mean()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        step = graphs.GraphedTrainStep(loss_fn, (r, t), opt)
        for _ in range(3):
            step(r, t)
    sd = step.optimizer_state_dict()
    assert all(isinstance(g["lr"], float) and g["lr"] == pytest.approx(2e-4) and g["capturable"] is False for g in sd["param_groups"])
    assert all(st["step"].device.type == "cpu" and float(st["step"]) == 3.0 for st in sd["state"].values())
    assert isinstance(opt.param_groups[0]["lr"], torch.Tensor)            # the live optimizer is untouched: the graphs still replay
    path = tmp_path / "ckpt.tar"
    torch.save({"optimizer_state_dict": sd, "network_fn_state_dict": net_c.state_dict()}, path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    clones = [p.detach().clone().requires_grad_(True) for p in params]
    fresh = torch.optim.Adam(clones, lr=5e-4)                              # run_nerf.py:307 + :322: created, then load_state_dict
    fresh.load_state_dict(ck["optimizer_state_dict"])
    assert fresh.param_groups[0]["lr"] == pytest.approx(2e-4) and isinstance(fresh.param_groups[0]["lr"], float)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        step(r, t)                                                         # step 4 through the graphs ...
    for c, p in zip(clones, params):                                       # ... and through the reloaded eager optimizer, same gradients
        c.grad = p.grad.detach().clone()
    fresh.step()
    worst = max(float((c - p).abs().max()) for c, p in zip(clones, params))
    assert worst <= 1e-7, worst
    step.close()
    assert all(isinstance(g["lr"], float) and not g["capturable"] for g in opt.param_groups)
    assert all(st["step"].device.type == "cpu" for st in opt.state.values())
    opt.step()                                                             # eager use works again
    with pytest.raises(RuntimeError, match="closed"):
        step(r, t)


def test_a_batch_outside_the_f16_range_does_not_reach_the_optimizer_through_the_graph(monkeypatch):
    from intrinsicnerf_amd import graphs
    monkeypatch.setenv("INERF_PRECISION", "f16x3")
    dev = torch.device("cuda:0")
    ol, net_c, net_f, query, rays = _setup(dev)
    r, t = rays[:9], torch.rand(9, 3, device=dev)
    opt = torch.optim.Adam(list(net_c.parameters()) + list(net_f.parameters()), lr=1e-4, capturable=True)

    def loss_fn(r, t):
        ret = ol.render_rays(r, net_c, query, 64, retraw=True, perturb=0.0, N_importance=32, network_fine=net_f, white_bkgd=True)
        return ((ret["rgb_map"] - t) ** 2).mean()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        step = graphs.GraphedTrainStep(loss_fn, (r, t), opt)
        assert float(step(r, t)) > 0 and step.fallbacks == 0
        with torch.no_grad():
            net_f.pts_linears[2].weight.mul_(1.0e6)              # hidden activations of the fine network far beyond 7.5e3
        loss = step(r, t)
    assert step.fallbacks == 1, "the graph's range word must have sent this batch to the eager path"
    assert torch.isfinite(loss).all()
    # the eager re-run re-binds p.grad; afterwards p.grad must again be the tensors the graph writes, holding that step's gradients
    params = list(net_c.parameters()) + list(net_f.parameters())
    assert all(p.grad is g for p, g in zip(step.params, step._grads)) and all(torch.isfinite(p.grad).all() for p in params if p.grad is not None)
    assert sum(p.grad is not None for p in params) >= len(params) // 2
    assert all(torch.isfinite(p).all() for p in list(net_c.parameters()) + list(net_f.parameters()))


def test_graphed_ssr_trainer_step_equals_the_eager_step(monkeypatch):
    """The SSR trainer's step (trainer.py:876-991: render_rays -> photometric + semantic cross-entropy -> backward -> Adam) through
    SSRRenderMixin.render_rays as HIP graphs: bit for bit the eager step (no jitter / noise, so no random draws)."""
    from intrinsicnerf_amd import graphs, ssr
    from oracle import calibration as cal
    monkeypatch.setenv("INERF_PRECISION", "f16x3")
    dev = torch.device("cuda:0")
    C, n = 5, 40
    g = torch.Generator().manual_seed(2)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    rays = torch.cat([torch.tensor([[0.5, 0.2, 0.1]]).expand(n, 3), d, 0.1 * torch.ones(n, 1), 10 * torch.ones(n, 1), d], -1)
    sd_c, sd_f = cal.calibrated_default_init("ssr", C, 0, rays), cal.calibrated_default_init("ssr", C, 1, rays)
    target = torch.rand(n, 3, generator=g).to(dev)
    labels = torch.randint(0, C, (n,), generator=g).to(dev)
    rays = rays.to(dev)
    out = {}
    for mode in ("eager", "graph"):
        r = ssr.SSRRenderer(C, white_bkgd=False, endpoint_feat=False, chunk=16, device=dev, perturb=0., raw_noise_std=0.)
        r.ssr_net_coarse.load_state_dict(sd_c); r.ssr_net_fine.load_state_dict(sd_f)
        r.training = True                                                    # check_numerics stays on: skipped while a graph records
        opt = torch.optim.Adam(list(r.ssr_net_coarse.parameters()) + list(r.ssr_net_fine.parameters()), lr=1e-4, capturable=True)

        def loss_fn(rb, tg):
            ret = r.render_rays(rb)                                          # three chunks of <= 16 rays
            ce = torch.nn.functional.cross_entropy
            return ((ret["rgb_fine"] - tg) ** 2).mean() + ((ret["rgb_coarse"] - tg) ** 2).mean() \
                + 0.04 * ce(ret["sem_logits_fine"], labels) + 0.04 * ce(ret["sem_logits_coarse"], labels)

        losses = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            step = graphs.GraphedTrainStep(loss_fn, (rays, target), opt) if mode == "graph" else None
            for _ in range(3):
                if step is None:
                    opt.zero_grad(set_to_none=True)
                    loss = loss_fn(rays, target)
                    loss.backward()
                    opt.step()
                else:
                    loss = step(rays, target)
                losses.append(float(loss))
        out[mode] = (losses, [p.detach().clone() for p in list(r.ssr_net_coarse.parameters()) + list(r.ssr_net_fine.parameters())])
    assert out["eager"][0] == out["graph"][0] and out["graph"][0][-1] < out["graph"][0][0], out
    for a, b in zip(out["eager"][1], out["graph"][1]):
        assert torch.equal(a, b)
