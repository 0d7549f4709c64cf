#!/usr/bin/env python3
"""Training-step timing through the object-level front-end (staged path: HIP sampling / compositing with HIP backward,
network layers through torch autograd) and the compositing kernels' HBM rates.

    python scripts/bench_train_step.py [--rays 2048] [--iters 10]
The batch is the reference's: N_rand = 1024 rays plus one neighbour each (run_nerf.py:918-929), 64 + 128 samples.
"""
import argparse
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.build()
from intrinsicnerf_amd import kernels, object_level as ol  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=2048)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--ssr", type=int, default=-1, help="C >= 0: the SSR network with C classes through ssr.SSRRenderer instead")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
if a.ssr >= 0:          # trainer.py:876-991: 1024 rays (512 + neighbours), depth range [0.1, 10], semantic cross-entropy + photometric loss
    from intrinsicnerf_amd import ssr
    n = a.rays
    r = ssr.SSRRenderer(a.ssr, white_bkgd=False, endpoint_feat=False, device=dev)
    r.training, r.check_numerics = True, False
    opt = torch.optim.Adam(list(r.ssr_net_coarse.parameters()) + list(r.ssr_net_fine.parameters()), lr=5e-4)
    o = torch.tensor([[0.5, 0.2, 0.1]]).expand(n, 3)
    d = torch.randn(n, 3); d = d / d.norm(dim=-1, keepdim=True)
    rays = torch.cat([o, d, 0.1 * torch.ones(n, 1), 10 * torch.ones(n, 1), d], -1).to(dev)
    target = torch.rand(n, 3, device=dev)
    labels = torch.randint(0, max(a.ssr, 1), (n,), device=dev)

    def sstep():
        ret = r.render_rays(rays)
        loss = ((ret["rgb_fine"] - target) ** 2).mean() + ((ret["rgb_coarse"] - target) ** 2).mean()
        if a.ssr > 0:
            loss = loss + 0.04 * torch.nn.functional.cross_entropy(ret["sem_logits_fine"], labels) \
                + 0.04 * torch.nn.functional.cross_entropy(ret["sem_logits_coarse"], labels)
        opt.zero_grad(); loss.backward(); opt.step()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(10):
            sstep()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.iters):
            sstep()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
    print(f"SSR training step (C = {a.ssr}): {n} rays x (64+128) samples: {dt * 1e3:.1f} ms -> {n / dt:.0f} rays/s "
          f"[{os.environ.get('INERF_TRAIN_MLP', 'hip')} network backward]")
    sys.exit(0)
embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
net_c, net_f = mk(), mk()
opt = torch.optim.Adam(list(net_c.parameters()) + list(net_f.parameters()), lr=5e-4)
n = a.rays
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3)
rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
target = torch.rand(n, 3, device=dev)
q = ol.NetworkQuery(embed, embed_d)


def step():
    ret = ol.render_rays(rays, net_c, q, 64, retraw=True, perturb=1.0, N_importance=128, network_fine=net_f, white_bkgd=True,
                         raw_noise_std=0.0)
    loss = ((ret["rgb_map"] - target) ** 2).mean() + ((ret["rgb0"] - target) ** 2).mean() \
        + 0.01 * ret["albedo_map"].abs().mean() + 0.01 * (ret["shading_map"] - 0.5).pow(2).mean() + 0.01 * ret["residual_map"].abs().mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
    return float(loss.detach()) if False else loss


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(10):          # the caching allocator needs a few steps to settle on the step's multi-GB blocks (3 warm-ups measured
        step()                   # 13.1 ms where the steady state is 11.4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
print(f"training step (staged path): {n} rays x (64+128) samples: {dt * 1e3:.1f} ms -> {n / dt:.0f} rays/s")

# compositing kernels alone, fine-pass shape
s, chn = 192, 11
raw = torch.rand(n * 16, s, chn, device=dev)
z = torch.sort(torch.rand(n * 16, s, device=dev) * 4 + 2, -1)[0]
dd = torch.randn(n * 16, 3, device=dev)
grads = {k: torch.randn(n * 16, 3, device=dev) for k in ("rgb", "albedo", "residual")}
grads.update({k: torch.randn(n * 16, device=dev) for k in ("acc", "depth", "shading")})
for name, fn, nbytes in (("k_composite", lambda: kernels.composite(raw, z, dd, None, True), raw.numel() * 4 + 2 * z.numel() * 4),
                         ("k_composite_bwd", lambda: kernels.composite_backward(raw, z, dd, grads, None, True), 2 * raw.numel() * 4 + z.numel() * 4)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name}: {n * 16} rays x {s} samples x {chn} ch: {ms:.3f} ms -> {nbytes / ms / 1e6:.0f} GB/s algorithmic")
