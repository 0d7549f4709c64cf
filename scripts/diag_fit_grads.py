#!/usr/bin/env python3
"""Diagnostic: the training step of scripts/fit_synthetic.py at a few points of a fit - parameter gradients of the HIP network
backward against torch's layers on the SAME batch and random draws, the fragments' scale S against the exact largest normaliser,
range words.   python scripts/diag_fit_grads.py [--steps 60]"""
import argparse
import importlib.util
import os
import sys
import warnings

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__  # noqa: E402

__graft_entry__.build()
spec = importlib.util.spec_from_file_location("fit_synthetic", os.path.join(REPO, "scripts", "fit_synthetic.py"))
fs = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fs)
from intrinsicnerf_amd import kernels, object_level as ol  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
a = ap.parse_args()
dev = torch.device("cuda:0")
rays, target = fs.training_set(dev, n_poses=8, side=48)
net_c, net_f, query = fs.make_nets(dev)
opt = torch.optim.Adam(list(net_c.parameters()) + list(net_f.parameters()), lr=5e-4)
g = torch.Generator(device=dev).manual_seed(1)
torch.manual_seed(1)

# record what the chain is given: wrap kernels.mlp_backward
seen = []
real = kernels.mlp_backward


JUDGE = {"on": False}


def spy(desc, packed_bwd, raw, d_raw, save, act_max, endpoint=False, status=None):
    out = real(desc, packed_bwd, raw, d_raw, save, act_max, endpoint, status)
    if JUDGE["on"]:
        # the same network evaluation judged by fp64 autograd: the embedded inputs are in the activation buffer (fp32 rows),
        # the cotangent is d_raw; torch's own fp32 autograd on the same inputs next to it
        import copy
        p = raw.shape[0]
        X = kernels.save_slot_views(desc, save, p)
        emb = torch.cat([X[kernels.SAVE_ENC][:, :63], X[kernels.SAVE_DIR][:, :27]], -1)
        net = net_f if p > 200000 else net_c
        got = kernels.param_views(desc, out)
        res = {}
        for tag, dt in (("fp64", torch.float64), ("torch32", torch.float32)):
            n2 = copy.deepcopy(net).to(dt)
            n2.zero_grad()
            with torch.enable_grad():
                (n2(emb.to(dt)) * d_raw.to(dt)).sum().backward()
            res[tag] = {k: q.grad.double() for k, q in n2.named_parameters()}
        e_hip = max((float((got[k].double() - res["fp64"][k]).norm() / res["fp64"][k].norm().clamp_min(1e-300)), k) for k in res["fp64"])
        e_t32 = max((float((res["torch32"][k] - res["fp64"][k]).norm() / res["fp64"][k].norm().clamp_min(1e-300)), k) for k in res["fp64"])
        per = {k: f"{float((got[k].double() - res['fp64'][k]).norm() / res['fp64'][k].norm().clamp_min(1e-300)):.1e}" for k in res["fp64"] if "pts_linears" in k and "weight" in k}
        print(f"   vs fp64 autograd ({p} points): HIP worst {e_hip[0]:.2e} ({e_hip[1]}), torch fp32 autograd worst {e_t32[0]:.2e} ({e_t32[1]}); HIP trunk: {per}")
    nz = d_raw.abs().amax(1)
    seen.append(dict(points=raw.shape[0], max_entry=float(d_raw.abs().max()), zero_points=int((nz == 0).sum()),
                     q=[float(x) for x in torch.quantile(nz[nz > 0][:2000000].float(), torch.tensor([0.01, 0.5, 0.99, 1.0], device=dev))] if (nz > 0).any() else None,
                     status=None if status is None else int(status.item()), act_max=float(act_max)))
    return out


kernels.mlp_backward = spy


def grads(mode, sel, rng):
    os.environ["INERF_TRAIN_MLP"] = mode
    torch.cuda.set_rng_state(rng)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ret = ol.render_rays(rays[sel], net_c, query, 64, retraw=True, perturb=1.0, N_importance=128, network_fine=net_f, white_bkgd=True)
    loss = ((ret["rgb_map"] - target[sel]) ** 2).mean() + ((ret["rgb0"] - target[sel]) ** 2).mean()
    opt.zero_grad()
    loss.backward()
    return float(loss), {k: p.grad.clone() for k, p in list(net_c.named_parameters(prefix="c")) + list(net_f.named_parameters(prefix="f"))}


for it in range(a.steps):
    sel = torch.randint(0, rays.shape[0], (2048,), device=dev, generator=g)
    rng = torch.cuda.get_rng_state(dev)
    if it in (0, 5, 20, a.steps - 1):
        seen.clear()
        lt, gt = grads("torch", sel, rng)
        JUDGE["on"] = True
        lh, gh = grads("hip", sel, rng)
        JUDGE["on"] = False
        worst = sorted(((float((gh[k].double() - gt[k].double()).norm() / gt[k].double().norm().clamp_min(1e-30)), k) for k in gt), reverse=True)
        print(f"step {it}: loss hip {lh:.6f} torch {lt:.6f}; worst relative gradient errors: " + ", ".join(f"{k} {e:.1e}" for e, k in worst[:4]))
        for s in seen:
            print("   chain input:", s)
    os.environ["INERF_TRAIN_MLP"] = "hip"
    torch.cuda.set_rng_state(rng)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ret = ol.render_rays(rays[sel], net_c, query, 64, retraw=True, perturb=1.0, N_importance=128, network_fine=net_f, white_bkgd=True)
    loss = ((ret["rgb_map"] - target[sel]) ** 2).mean() + ((ret["rgb0"] - target[sel]) ** 2).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
    if it % 10 == 0:
        print(f"step {it}: loss {float(loss):.5f}", flush=True)
