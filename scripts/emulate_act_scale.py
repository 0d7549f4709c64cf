#!/usr/bin/env python3
"""CPU gate for re-running an f16x3 launch that left f16's range at a SMALLER activation scale instead of on the exact-fp32 kernel /
torch layers: emulates the split arithmetic (operands v * 2^k -> f16 hi (round toward zero) + f16 lo, products hi*hi + hi*lo + lo*hi
exact, fp32 accumulation) for the object-level network at activation scales 8 * 2^-shift and compares raw against an fp64 evaluation,
next to the reference arithmetic (fp32) itself.   python scripts/emulate_act_scale.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle

def split(v32, scale):
    """v * scale -> (hi, lo) as float32 arrays holding f16 values; hi by truncation, lo rounded to nearest (f16, subnormals kept)."""
    t = (v32.astype(np.float32) * np.float32(scale)).astype(np.float32)
    bits = t.view(np.uint32) & np.uint32(0xFFFFE000)
    hi = bits.view(np.float32).copy()
    small = np.abs(t) < 2.0 ** -14                      # f16 subnormal range: truncate to a multiple of 2^-24
    hi[small] = (np.trunc(t[small].astype(np.float64) * 2.0 ** 24) / 2.0 ** 24).astype(np.float32)
    hi = np.clip(hi, -65504, 65504)
    lo = (t - hi).astype(np.float16).astype(np.float32)
    return hi, lo, float(np.abs(t).max())

def gemm3(xh, xl, wh, wl):
    # fp32 accumulation of exact f16 x f16 products (numpy float32 matmul: another summation order than the MFMA's, same grade)
    return (xh @ wh.T + xh @ wl.T + xl @ wh.T).astype(np.float32)

def layer(x, w, b, act_scale, relu=True):
    if act_scale is None:       # PER-LAYER scale (round 6): the largest power of two <= 8 that keeps THIS operand inside f16's range
        m = float(np.abs(x).max())
        act_scale = 8.0 * 2.0 ** -max(0, int(np.ceil(np.log2(max(m, 1e-30) * 8.0 / 6.0e4))))
    kw = int(np.floor(np.log2(16384.0 / np.abs(w).max())))           # max |W'| in (2^13, 2^14]
    wh, wl, _ = split(w, 2.0 ** kw)
    xh, xl, amax = split(x, act_scale)
    acc = gemm3(xh, xl, wh, wl)
    out = acc * np.float32(2.0 ** -kw / act_scale) + b               # (the kernel keeps the scaled domain; same roundings up to exact powers of two)
    return (np.maximum(out, 0) if relu else out).astype(np.float32), amax

def forward_emul(sd, emb, act_scale):
    e = 63
    pts, dirs = emb[:, :e], emb[:, e:]
    g = lambda k: sd[k].numpy()
    h, worst = pts, 0.0
    for i in range(8):
        h, a = layer(h, g(f"pts_linears.{i}.weight"), g(f"pts_linears.{i}.bias"), act_scale); worst = max(worst, a)
        if i == 4:
            h = np.concatenate([pts, h], -1)
    sigma, a = layer(h, g("alpha_linear.weight"), g("alpha_linear.bias"), act_scale, False); worst = max(worst, a)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float32)))
    al, _ = layer(h, g("albedo_linear1.weight"), g("albedo_linear1.bias"), act_scale)
    al, _ = layer(al, g("albedo_linear2.weight"), g("albedo_linear2.bias"), act_scale, False)
    sh, _ = layer(h, g("test_linear1.weight"), g("test_linear1.bias"), act_scale)
    sh, _ = layer(sh, g("test_linear2.weight"), g("test_linear2.bias"), act_scale, False)
    ft, _ = layer(h, g("feature_linear.weight"), g("feature_linear.bias"), act_scale, False)
    v, a = layer(np.concatenate([ft, dirs], -1), g("views_linears.0.weight"), g("views_linears.0.bias"), act_scale); worst = max(worst, a)
    rs, _ = layer(v, g("shading_linear.weight"), g("shading_linear.bias"), act_scale, False)
    al, sh, rs = sig(al), sig(sh), sig(rs)
    return np.concatenate([al * sh + rs, sigma, al, sh, rs], -1), worst

torch.manual_seed(0)
cfg = oracle.RenderConfig(variant="object")
n = 4096
pts = (torch.rand(n, 3) * 4 - 2)
dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
emb = torch.cat([oracle.freq_encode(pts, 10), oracle.freq_encode(dirs, 4)], -1)
print("activation gain | largest hidden activation | shift | raw error vs fp64: max |err| / (1e-5 + 1e-4 |ref|) over rgb/albedo/shading/residual, sigma relative | fp32 reference's own | one scale per GEMM operand")
for gain in (1.0, 2.0 ** 14, 2.0 ** 18, 2.0 ** 24):
    sd = oracle.make_state_dict("object", 0, seed=3)
    # a network whose first trunk layer is `gain` times larger (and whose second undoes most of it): hidden activations of layer 0 ~ gain
    sd["pts_linears.0.weight"] *= gain; sd["pts_linears.0.bias"] *= gain
    sd["pts_linears.1.weight"] /= gain
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        r64 = oracle.mlp_forward(sd64, emb.double(), cfg).numpy()
        r32 = oracle.mlp_forward(sd, emb, cfg).numpy()
        h0 = torch.relu(torch.nn.functional.linear(emb[:, :63], sd["pts_linears.0.weight"], sd["pts_linears.0.bias"]))
    def score(r):
        e = np.abs(r - r64) / (1e-5 + 1e-4 * np.abs(r64))
        e[:, 3] = np.abs(r[:, 3] - r64[:, 3]) / (1e-5 * max(1.0, np.sqrt((r64[:, 3] ** 2).mean())) + 1e-4 * np.abs(r64[:, 3]))
        return float(e.max())
    line = f"gain 2^{int(np.log2(gain)):2d} | max h0 {float(h0.max()):9.3g} | fp32 reference {score(r32):7.3f} |"
    for shift in (0, 6, 12, 18, 24):
        raw, worst = forward_emul(sd, emb.numpy(), 8.0 * 2.0 ** -shift)
        ok = worst <= 6.0e4
        line += f" shift {shift:2d}: {'%7.3f' % score(raw) if ok else '  range'}"
    raw, worst = forward_emul(sd, emb.numpy(), None)
    line += f" | per-layer scale: {'%7.3f' % score(raw) if worst <= 6.0e4 else '  range'}"
    print(line)
