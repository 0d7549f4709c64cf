#!/usr/bin/env python3
"""Micro-benchmark of the dominant kernel (k_encode_mlp) alone - used for kernel tuning and PMC passes.

    python scripts/bench_mlp.py [--rays 65536] [--samples 192] [--iters 5] [--ssr C]
Prints achieved fp32 TFLOP/s (algorithmic: 2*659456 FLOP per point for the object-level net) and a
checksum of the output, so that two builds can be compared for speed and for identical results.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.build()
import oracle  # noqa: E402
from intrinsicnerf_amd import _capi, kernels, packing  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=65536)
ap.add_argument("--samples", type=int, default=192)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--ssr", type=int, default=-1)
ap.add_argument("--precision", default=None, choices=[None, "f32", "f16x3"])
a = ap.parse_args()
dev = torch.device("cuda:0")
ssr = a.ssr >= 0
c = max(a.ssr, 0)
prec = None if a.precision is None else (_capi.PREC_F16X3 if a.precision == "f16x3" else _capi.PREC_F32)
desc = _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, c, 10, 4, 10.0 if ssr else 1.0, prec)
sd = oracle.make_state_dict("ssr" if ssr else "object", c, seed=0)
packed = packing.pack_state_dict(desc, sd).to(dev)
g = torch.Generator().manual_seed(0)
n = a.rays
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
z = torch.sort(torch.rand(n, a.samples, generator=g) * 4 + 2, -1)[0].to(dev)
raw = kernels.encode_mlp(desc, packed, rays, z)           # warm-up
torch.cuda.synchronize()
ts = []
for _ in range(a.iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    raw = kernels.encode_mlp(desc, packed, rays, z)
    e1.record()
    e1.synchronize()
    ts.append(e0.elapsed_time(e1))
flop_pt = 2 * (659456 + (32768 + 128 * c if (ssr and c > 0) else 0))
best, med = min(ts), sorted(ts)[len(ts) // 2]
print(f"k_encode_mlp[{'f16x3' if desc.precision else 'f32'}]: {n} rays x {a.samples} samples, median {med:.2f} ms (best {best:.2f}) -> "
      f"{flop_pt * n * a.samples / med / 1e9:.1f} TFLOP/s median, {flop_pt * n * a.samples / best / 1e9:.1f} best; "
      f"checksum {float(raw.double().sum()):.10e} absmax {float(raw.abs().max()):.6f}")
