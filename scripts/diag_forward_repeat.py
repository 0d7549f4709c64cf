#!/usr/bin/env python3
"""Development aid: N launches of the training forward (and of the inference forward) on the same inputs; raw and every written word of the
activation buffer compared bit for bit with the first launch.   python scripts/diag_forward_repeat.py [--launches 300] [--ssr C] [--endpoint]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from intrinsicnerf_amd import _capi, kernels, packing  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--launches", type=int, default=300); ap.add_argument("--rays", type=int, default=700); ap.add_argument("--samples", type=int, default=48)
ap.add_argument("--ssr", type=int, default=-1); ap.add_argument("--endpoint", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
ssr = a.ssr >= 0
desc = _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, max(a.ssr, 0), 10, 4, 10.0 if ssr else 1.0, _capi.PREC_F16X3)
sd = {k: v.to(dev) for k, v in oracle.lcg_state_dict("ssr" if ssr else "object", max(a.ssr, 0), seed=23, sigma_gain_log2=3, freq_decay=True).items()}
pf = packing.device_packer(desc, False, dev)(sd)
n, s = a.rays, a.samples
g = torch.Generator().manual_seed(5)
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0].to(dev)
p = n * s
X_ref = raw_ref = inf_ref = None
bad = 0
for it in range(a.launches):
    raw, save = kernels.encode_mlp_train(desc, pf, rays, z, endpoint=a.endpoint)
    views = kernels.save_slot_views(desc, save, p)          # decoded slots (fragments -> values): every written element
    cur = [v.clone() for v in views if v is not None and v.numel()]
    bits = save[-(((p + 63) // 64) * 4096 + kernels.SAVE_SCALARS):-kernels.SAVE_SCALARS].view(torch.int32).clone()
    inf = kernels.encode_mlp(desc, pf, rays, z, endpoint=a.endpoint)
    if X_ref is None:
        X_ref, raw_ref, bits_ref, inf_ref = cur, raw.clone(), bits, inf.clone()
        continue
    msgs = []
    if not torch.equal(raw.view(torch.int32), raw_ref.view(torch.int32)): msgs.append("raw(train)")
    if not torch.equal(inf.view(torch.int32), inf_ref.view(torch.int32)): msgs.append("raw(inference)")
    if not torch.equal(bits, bits_ref): msgs.append("mask bits")
    for k, (x, y) in enumerate(zip(cur, X_ref)):
        if not torch.equal(x.view(torch.int32), y.view(torch.int32)): msgs.append(f"slot#{k}: {int((x != y).sum())}")
    if msgs:
        bad += 1
        if bad <= 3: print(f"launch {it}: " + ", ".join(msgs))
print(f"{bad} of {a.launches - 1} launches differ from the first; training raw == inference raw: {torch.equal(raw_ref.view(torch.int32), inf_ref.view(torch.int32))}")
