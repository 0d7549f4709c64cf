#!/usr/bin/env python3
"""Timings of the fp32 layer kernels (csrc/layered.hip) on the training batch's point count: single launches against the fp32 MFMA peak
(157.3 TFLOP/s) and against torch's library GEMM of the same shape, then whole networks (forward, forward + backward).

    python scripts/bench_layered.py [--rays 2048] [--samples 192]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicnerf_amd import layered, object_level as ol  # noqa: E402

PEAK = 157.3


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--only", type=int, nargs=2, metavar=("OUT", "IN"), help="one forward layer of this shape, --iters times (for counter passes)")
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    n = a.rays * a.samples
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    print(f"{n} points")
    if a.only:
        o, k = a.only
        x = torch.randn(n, k, device=dev, generator=g)
        w = torch.randn(o, k, device=dev, generator=g) / k ** 0.5
        b = torch.randn(o, device=dev, generator=g)
        y = torch.empty(n, o, device=dev)
        t = timed(lambda: layered.linear(layered.Cols(x), w, b, layered.Cols(y), 1), iters=a.iters, warm=1)
        print(f"  {o} x {k} forward {t:7.3f} ms {2.0 * n * o * k / t / 1e9:6.1f} TFLOP/s")
        return
    for label, o, k in (("256 x 256", 256, 256), ("256 x 319 (skip)", 256, 319), ("128 x 283 (views)", 128, 283), ("3 x 128 (head)", 3, 128),
                        ("128 x 128", 128, 128), ("64 x 64", 64, 64)):
        x = torch.randn(n, k, device=dev, generator=g)
        w = torch.randn(o, k, device=dev, generator=g) / k ** 0.5
        b = torch.randn(o, device=dev, generator=g)
        y = torch.empty(n, o, device=dev)
        dz = torch.randn(n, o, device=dev, generator=g)
        dx = torch.empty(n, k, device=dev)
        flop = 2.0 * n * o * k
        t_f = timed(lambda: layered.linear(layered.Cols(x), w, b, layered.Cols(y), 1))
        t_d = timed(lambda: layered.linear_dgrad(layered.Cols(dz), w, layered.Cols(dx), gate=layered.Cols(x)))
        t_w = timed(lambda: layered.linear_wgrad(layered.Cols(dz), layered.Cols(x)))
        t_tf = timed(lambda: torch.relu_(F.linear(x, w, b)))
        t_td = timed(lambda: dz @ w)
        t_tw = timed(lambda: dz.t() @ x)
        print(f"  {label:18s} forward {t_f:7.3f} ms {flop / t_f / 1e9:6.1f} TFLOP/s ({flop / t_f / 1e9 / PEAK:.2f} of fp32 MFMA peak; torch {t_tf:7.3f} ms) | "
              f"input grad {t_d:7.3f} ms {flop / t_d / 1e9:6.1f} (torch {t_td:7.3f}) | weight grad {t_w:7.3f} ms {flop / t_w / 1e9:6.1f} (torch {t_tw:7.3f})")
    # whole networks
    o3 = torch.tensor([[2.5, 1.5, 2.0]]).expand(a.rays, 3)
    d = -o3 / o3.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(a.rays, 3)
    rays = torch.cat([o3, d, 2 * torch.ones(a.rays, 1), 6 * torch.ones(a.rays, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
    z = torch.sort(torch.rand(a.rays, a.samples) * 4 + 2, -1)[0].to(dev)
    embed, ch = ol.get_embedder(10, 0)
    embed_d, ch_d = ol.get_embedder(4, 0)
    for D, W, skips in ((8, 256, [4]), (8, 128, [4]), (4, 64, [2])):
        net = ol.NeRF(D=D, W=W, input_ch=ch, output_ch=5, skips=skips, input_ch_views=ch_d, use_viewdirs=True).to(dev)
        spec = layered.spec_for(net, embed, embed_d)
        macs = sum(p.numel() for k, p in net.named_parameters() if k.endswith("weight"))
        flop = 2.0 * n * macs
        with torch.no_grad():
            t_f = timed(lambda: layered.evaluate(spec, net, rays, z), iters=5, warm=2)

        def step():
            net.zero_grad()
            raw = layered.evaluate(spec, net, rays, z)
            raw.backward(torch.ones_like(raw))

        t_s = timed(step, iters=5, warm=2)

        def step_torch():
            net.zero_grad()
            pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., :, None]
            raw = ol._run_network_torch(pts, rays[:, 8:11], net, embed, embed_d, 1 << 30)
            raw.backward(torch.ones_like(raw))

        t_t = timed(step_torch, iters=5, warm=2)
        print(f"  NeRF(D={D}, W={W}): forward {t_f:7.3f} ms {flop / t_f / 1e9:6.1f} TFLOP/s ({flop / t_f / 1e9 / PEAK:.2f}) | forward + backward {t_s:7.3f} ms "
              f"{3 * flop / t_s / 1e9:6.1f} TFLOP/s ({3 * flop / t_s / 1e9 / PEAK:.2f}) | the same through torch's layers {t_t:7.3f} ms")


if __name__ == "__main__":
    main()
