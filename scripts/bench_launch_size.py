#!/usr/bin/env python3
"""Does the encode+MLP kernel's rate depend on the launch size?  For each size: launches timed ONE BY ONE (HIP events, a host
synchronisation between them - what scripts/bench_mlp.py and bench.py's roofline legs do) and the same number of points as launches
BACK TO BACK on the stream (what a frame's chunks are).   python scripts/bench_launch_size.py [--ssr C]"""
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
ap.add_argument("--ssr", type=int, default=-1)
ap.add_argument("--samples", type=int, default=192)
a = ap.parse_args()
dev = torch.device("cuda:0")
ssr, c = a.ssr >= 0, max(a.ssr, 0)
desc = _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, c, 10, 4, 10.0 if ssr else 1.0, _capi.PREC_F16X3)
packed = packing.pack_state_dict(desc, oracle.make_state_dict("ssr" if ssr else "object", c, seed=0)).to(dev)
flop_pt = 2 * (659456 + (32768 + 128 * c if (ssr and c > 0) else 0))
total = 262144
g = torch.Generator().manual_seed(0)
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(total, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(total, 3, generator=g)
rays = torch.cat([o, d, 2 * torch.ones(total, 1), 6 * torch.ones(total, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
z = torch.sort(torch.rand(total, a.samples, generator=g) * 4 + 2, -1)[0].to(dev)
kernels.encode_mlp(desc, packed, rays[:65536], z[:65536])
torch.cuda.synchronize()
print(f"# {'SSR C=%d' % c if ssr else 'object'} network, {a.samples} samples per ray, INERF_F16_KERNEL={os.environ.get('INERF_F16_KERNEL', 'default')}; TFLOP/s algorithmic")
for n in (4096, 8192, 16384, 32768, 65536, 131072, 262144):
    reps = total // n
    single = []
    for i in range(min(reps, 6)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        kernels.encode_mlp(desc, packed, rays[i * n:(i + 1) * n], z[i * n:(i + 1) * n])
        e1.record()
        e1.synchronize()
        single.append(e0.elapsed_time(e1))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        kernels.encode_mlp(desc, packed, rays[i * n:(i + 1) * n], z[i * n:(i + 1) * n])
    e1.record()
    e1.synchronize()
    b2b = e0.elapsed_time(e1) / reps
    med = sorted(single)[len(single) // 2]
    rate = lambda ms: flop_pt * n * a.samples / ms / 1e9
    print(f"{n:7d} rays per launch: one by one {med:8.3f} ms = {rate(med):6.1f} | {reps:3d} back to back {b2b:8.3f} ms each = {rate(b2b):6.1f}")
