#!/usr/bin/env python3
"""Development aid: N launches of the input-gradient chain on the same inputs; which launches / slots / tiles are not bit-identical to the first.
    python scripts/diag_chain_repeat.py [--launches 400] [--rays 700] [--samples 48]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from intrinsicnerf_amd import _capi, kernels, packing  # noqa: E402
import ctypes as C
ap = argparse.ArgumentParser(); ap.add_argument("--launches", type=int, default=400); ap.add_argument("--rays", type=int, default=700); ap.add_argument("--samples", type=int, default=48)
ap.add_argument("--ssr", type=int, default=-1); ap.add_argument("--endpoint", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
ssr = a.ssr >= 0
desc = _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, max(a.ssr, 0), 10, 4, 10.0 if ssr else 1.0, _capi.PREC_F16X3)
sd = {k: v.to(dev) for k, v in oracle.lcg_state_dict("ssr" if ssr else "object", max(a.ssr, 0), seed=23, sigma_gain_log2=3, freq_decay=True).items()}
pf, pb = packing.device_packer(desc, False, dev)(sd), packing.device_packer(desc, True, dev)(sd)
n, s = a.rays, a.samples
g = torch.Generator().manual_seed(5)
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0].to(dev)
p = n * s
raw, save = kernels.encode_mlp_train(desc, pf, rays, z, endpoint=a.endpoint)
ch = raw.shape[-1]
cot = torch.randn(p, ch, device=dev)
names = {2 + i: f"H{i}" for i in range(8)} | {10: "AS1H", 11: "FEAT", 12: "VH", 14: "DPRE", 0: "norm"} | ({13: "SEMH"} if ssr and a.ssr > 0 else {})
tiles = (p + 63) // 64
def slot(buf, k):
    off, width = C.c_int64(), C.c_int()
    _capi.lib().inerf_mlp_save_slot(desc, k, p, C.byref(off), C.byref(width))
    n_el = tiles * 64 * width.value if k != 0 else tiles * 64
    return buf[off.value: off.value + n_el].view(torch.int32)
ref, bad = None, 0
for it in range(a.launches):
    dz, heads = kernels.mlp_backward_inputs(desc, pb, raw.view(p, ch), cot, save, endpoint=a.endpoint, want_heads=True)
    cur = {k: slot(dz, k).clone() for k in names}
    cur["heads"] = heads.view(torch.int32).clone()
    if ref is None:
        ref = cur; continue
    msgs = []
    for k, v in cur.items():
        diff = (v != ref[k]).nonzero().flatten()
        if len(diff):
            nm = names.get(k, k)
            per_tile = v.numel() // tiles if k != "heads" else 1672
            t = torch.unique(diff // per_tile).cpu().tolist()
            kb = torch.unique((diff % per_tile) * 4 // 1024).cpu().tolist() if k != "heads" else []
            msgs.append(f"{nm}: {len(diff)} words, tiles/rows {t[:10]}{'...' if len(t) > 10 else ''}, KB offsets in tile {kb[:20]}")
    if msgs:
        bad += 1
        print(f"launch {it}: " + (" | ".join(msgs) if bad <= 2 else f"{len(msgs)} slots"))
print(f"{bad} of {a.launches - 1} launches differ from the first ({os.environ.get('INERF_DGRAD_KERNEL', 'dual')}, {os.environ.get('INERF_LIB_OVERRIDE', 'tree library')})")
