#!/usr/bin/env python3
"""Timing of the two training kernels alone (training forward with activation saving, input-gradient chain) and of the
weight-gradient GEMMs, on the reference's fine-pass batch shape.   python scripts/bench_train_kernels.py [--rays 2048]"""
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
ap.add_argument("--rays", type=int, default=2048)
ap.add_argument("--samples", type=int, default=192)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, 1.0, _capi.PREC_F16X3)
sd = {k: v.to(dev) for k, v in oracle.make_state_dict("object", 0, seed=0).items()}
pf, pb = packing.device_packer(desc, False, dev)(sd), packing.device_packer(desc, True, dev)(sd)
names = tuple(n for n, _ in packing.tensor_table(desc))
n, s = a.rays, a.samples
g = torch.Generator().manual_seed(0)
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0].to(dev)
d_raw = torch.randn(n * s, 11, device=dev)


def timed(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2], out


flop = 2 * 659456 * n * s
p = n * s
t_inf, _ = timed(lambda: kernels.encode_mlp(desc, pf, rays, z))
act_max, dz_max = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
t_fwd, (raw, save) = timed(lambda: kernels.encode_mlp_train(desc, pf, rays, z, act_max=act_max))
t_bwd, (dz, heads) = timed(lambda: kernels.mlp_backward_inputs(desc, pb, raw.view(p, 11), d_raw, save, dz_max=dz_max, want_heads=True))
t_all, _ = timed(lambda: kernels.mlp_backward(desc, pb, raw.view(p, 11), d_raw, save, act_max))
t_wl, _ = timed(lambda: kernels.mlp_weight_gradients(desc, names, save, dz, d_raw, p, heads=heads))


def frag_slot(buf, slot):
    import ctypes as C
    off, width = C.c_int64(), C.c_int()
    _capi.lib().inerf_mlp_save_slot(desc, slot, p, C.byref(off), C.byref(width))
    return buf[off.value: off.value + (p + 63) // 64 * 64 * 256]


ranges = torch.cat([dz_max, act_max])
t_one, _ = timed(lambda: kernels.weight_gradient_frag(frag_slot(dz, kernels.SAVE_H0 + 3), frag_slot(dz, kernels.SAVE_ENC), frag_slot(save, kernels.SAVE_H0 + 2), ranges, p, want_bias=True))
# the nine 256 x 256 products of the network as mlp_backward launches them: one launch, each product on its share of the grid
pairs = [(kernels.SAVE_H0 + i, kernels.SAVE_H0 + i - 1) for i in range(1, 8)] + [(kernels.SAVE_AS1H, kernels.SAVE_H0 + 7), (kernels.SAVE_FEAT, kernels.SAVE_H0 + 7)]
t_nine, _ = timed(lambda: kernels.weight_gradient_frag_batch([frag_slot(dz, g) for g, _ in pairs], frag_slot(dz, kernels.SAVE_ENC),
                                                              [frag_slot(save, x) for _, x in pairs], ranges, p))
gb = save.numel() * 4 / 1e9
saved = (8 * 256 + 256 + 256 + 128 + 96) * 4 * p / 1e9          # what the training forward writes (h0..h7 and feat fragments, as1h, vh, enc + dir rows)
print(f"{n} rays x {s} samples = {p} points; activation buffer {gb:.2f} GB")
print(f"inference forward        {t_inf:7.3f} ms  {flop / t_inf / 1e9:6.1f} TFLOP/s")
print(f"training forward (save)  {t_fwd:7.3f} ms  {flop / t_fwd / 1e9:6.1f} TFLOP/s   writes {saved / t_fwd:5.2f} TB/s")
print(f"input-gradient chain (+ head gradients, pre-pass) {t_bwd:7.3f} ms  {0.89 * flop / t_bwd / 1e9:6.1f} TFLOP/s")
print(f"whole backward, one C call (chain + 13 products + reduction) {t_all:7.3f} ms; weight gradients = {t_all - t_bwd:7.3f} ms  {flop / (t_all - t_bwd) / 1e9:6.1f} TFLOP/s")
print(f"one 256 x 256 product from fragment slots (incl. the sum over {_capi.lib().inerf_wgrad_grid(p)} partial tiles) {t_one * 1e3:7.1f} us: operands {2 * p * 1024 / t_one / 1e9:5.2f} TB/s")
print(f"the nine 256 x 256 products in one launch (incl. the sums over ~{_capi.lib().inerf_wgrad_frag_rows(p, 9, None, None, 0)} partial tiles each) {t_nine:7.3f} ms: "
      f"operands {9 * 2 * p * 1024 / t_nine / 1e9:5.2f} TB/s")
print(f"weight gradients, library GEMMs on decoded slots (reference) {t_wl:7.3f} ms")
