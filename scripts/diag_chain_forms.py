#!/usr/bin/env python3
"""Development aid: where the two forms of the input-gradient chain (INERF_DGRAD_KERNEL=single | dual) differ, slot by slot, and
whether each form repeats itself bit for bit.   python scripts/diag_chain_forms.py [--ssr 28]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from intrinsicnerf_amd import _capi, kernels, packing  # noqa: E402
import ctypes as C
ap = argparse.ArgumentParser(); ap.add_argument("--ssr", type=int, default=-1); a = ap.parse_args()
dev = torch.device("cuda:0")
ssr = a.ssr >= 0
desc = _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, max(a.ssr, 0), 10, 4, 10.0 if ssr else 1.0, _capi.PREC_F16X3)
sd = {k: v.to(dev) for k, v in oracle.make_state_dict("ssr" if ssr else "object", max(a.ssr, 0), seed=7).items()}
pf, pb = packing.device_packer(desc, False, dev)(sd), packing.device_packer(desc, True, dev)(sd)
n, s = 700, 37
g = torch.Generator().manual_seed(2)
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0].to(dev)
raw, save = kernels.encode_mlp_train(desc, pf, rays, z)
p = n * s; ch = raw.shape[-1]
d_raw = torch.randn(p, ch, device=dev)
names = {2 + i: f"H{i}" for i in range(8)} | {10: "AS1H", 11: "FEAT", 12: "VH", 13: "SEMH", 14: "DPRE", 0: "ENC(norm)"}
def slot(buf, k):
    off, width = C.c_int64(), C.c_int()
    _capi.lib().inerf_mlp_save_slot(desc, k, p, C.byref(off), C.byref(width))
    n_el = (p + 63) // 64 * 64 * width.value if k != 0 else (p + 63) // 64 * 64
    return buf[off.value: off.value + n_el].view(torch.int32)
def run(form):
    os.environ["INERF_DGRAD_KERNEL"] = form
    dz, heads = kernels.mlp_backward_inputs(desc, pb, raw.view(p, ch), d_raw, save, want_heads=True)
    torch.cuda.synchronize()
    return {k: slot(dz, k).clone() for k in names if (k != 13 or ssr)}
ref = {f: run(f) for f in ("single", "dual")}
for k, nm in sorted(names.items()):
    if k not in ref["single"]: continue
    a_, b_ = ref["single"][k], ref["dual"][k]
    bad = (a_ != b_).nonzero().flatten().cpu().numpy()
    msg = f"{nm:10s} {len(bad):8d} of {a_.numel()} words differ"
    if len(bad):
        per_tile = a_.numel() // ((p + 63) // 64)
        tiles = np.unique(bad // per_tile)
        within = bad % per_tile
        msg += f"; tiles {tiles[:12].tolist()}{'...' if len(tiles) > 12 else ''} ({len(tiles)} tiles); byte offsets in tile {np.unique(within * 4 // 1024)[:16].tolist()} (KB)"
    print(msg)
    if len(bad) and nm in ("AS1H", "VH"):
        av, bv = a_.cpu().numpy().view(np.uint16), b_.cpu().numpy().view(np.uint16)
        h = np.nonzero(av != bv)[0]
        cbs = 8 if nm == "AS1H" else 4
        for w in h[:24]:
            byte = int(w) * 2
            t_, r = divmod(byte, per_tile * 4)
            kb, r = divmod(r, cbs * 2048); cb, r = divmod(r, 2048); plane, r = divmod(r, 1024); lane, i = divmod(r, 16); i //= 2
            print(f"    tile {t_} kb {kb} cb {cb} plane {plane} lane {lane} i {i}: single {av[w]:#06x} = {av[w:w+1].view(np.float16)[0]!r}  dual {bv[w]:#06x} = {bv[w:w+1].view(np.float16)[0]!r}   hi(single) {av.reshape(-1)[w - 512]:#06x}")
for form in ("single", "dual"):
    worst = {}
    for it in range(20):
        cur = run(form)
        for k in cur:
            nb = int((cur[k] != ref[form][k]).sum())
            worst[names[k]] = max(worst.get(names[k], 0), nb)
    print(form, "repeat differences over 20 launches:", {k: v for k, v in worst.items() if v})
