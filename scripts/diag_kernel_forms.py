#!/usr/bin/env python3
"""Are two forms of the f16x3 encode+MLP kernel bit-identical?  (INERF_F16_KERNEL is read per launch.)

    python scripts/diag_kernel_forms.py [--forms dual,t128] [--sizes 1x64,1000x192,...]
Compares raw outputs with torch.equal over whole and ragged point counts and prints per-form timings of the last size."""
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
ap.add_argument("--forms", default="dual,t128")
ap.add_argument("--sizes", default="1x64,3x1,1x191,7x192,33x64,1000x192,4099x192,65536x192")
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, 1.0, _capi.PREC_F16X3)
packed = packing.pack_state_dict(desc, oracle.make_state_dict("object", 0, seed=0)).to(dev)
forms = a.forms.split(",")


def run(form, rays, z):
    if form == "dual":
        os.environ.pop("INERF_F16_KERNEL", None)
    else:
        os.environ["INERF_F16_KERNEL"] = form
    return kernels.encode_mlp(desc, packed, rays, z)


bad = 0
sizes = a.sizes.split(",")
for size in sizes:
    n, s = (int(v) for v in size.split("x"))
    g = torch.Generator().manual_seed(n * 1000 + s)
    o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
    d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
    rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
    z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0].to(dev)
    outs = [run(f, rays, z) for f in forms]
    torch.cuda.synchronize()
    for f, out in zip(forms[1:], outs[1:]):
        same = torch.equal(outs[0], out)
        diff = float((outs[0] - out).abs().max())
        nan = int(torch.isnan(out).sum())
        print(f"{size:>12s} ({n * s} points): {forms[0]} vs {f}: {'bit-identical' if same else 'DIFFERENT'} (max |diff| {diff:.3e}, NaNs {nan}), checksum {float(out.double().sum()):.10e}")
        bad += 0 if same else 1
    if size == sizes[-1]:
        for f in forms:
            ts = []
            for _ in range(a.iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(f, rays, z)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            med = sorted(ts)[len(ts) // 2]
            print(f"{f:>8s}: {size} median {med:.2f} ms (best {min(ts):.2f}) -> {2 * 659456 * n * s / med / 1e9:.1f} TFLOP/s algorithmic")
print("forms differ" if bad else "all forms bit-identical")
sys.exit(1 if bad else 0)
