#!/usr/bin/env python3
"""PSNR delta of the HIP path vs the reference arithmetic (BASELINE.json: "<= 1e-4 dB PSNR delta vs reference").

No dataset images exist here, so the "ground truth" is synthetic: the fp64 evaluation of the same scene plus a fixed
pseudo-random perturbation sized to give the ~30 dB a trained model reaches.  Against that target T,
    delta = PSNR(HIP frame, T) - PSNR(CPU oracle frame, T)          (oracle == reference, tests/golden)
is what switching implementations does to a reported PSNR.  Also printed: the same delta for the oracle's fp32 vs fp64
evaluation (the reference's own numerical noise), and both restricted to well-conditioned rays.

    python scripts/psnr_delta.py [--side 40]         (needs a GPU; the CPU oracle takes ~10 s per 1600 rays)
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.build()
import oracle  # noqa: E402
from intrinsicnerf_amd import _capi, kernels, packing  # noqa: E402


def psnr(a, t):
    return -10.0 * np.log10(np.mean((np.asarray(a, np.float64) - t) ** 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=40)
    ap.add_argument("--precision", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    side = a.side
    # a side x side crop around the principal point of the 800x800 chair camera
    H = W = 800
    focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    j, i = torch.meshgrid(torch.arange(side, dtype=torch.float32), torch.arange(side, dtype=torch.float32), indexing="ij")
    i, j = i * (200.0 / side) + 300.0, j * (200.0 / side) + 300.0
    dirs = torch.stack([(i - W * 0.5) / focal, -(j - H * 0.5) / focal, -torch.ones_like(i)], -1).reshape(-1, 3)
    th, ph, rad = np.deg2rad(40.0), np.deg2rad(-30.0), 4.0
    rot_phi = torch.tensor([[1, 0, 0], [0, np.cos(ph), -np.sin(ph)], [0, np.sin(ph), np.cos(ph)]], dtype=torch.float32)
    rot_th = torch.tensor([[np.cos(th), 0, -np.sin(th)], [0, 1, 0], [np.sin(th), 0, np.cos(th)]], dtype=torch.float32)
    flip = torch.tensor([[-1, 0, 0], [0, 0, 1], [0, 1, 0]], dtype=torch.float32)
    R = flip @ rot_th @ rot_phi
    o = (R @ torch.tensor([0.0, 0.0, rad])).expand(dirs.shape[0], 3)
    d = dirs @ R.T
    n = d.shape[0]
    rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).contiguous()
    cfg = oracle.RenderConfig(variant="object", n_samples=64, n_importance=128, white_bkgd=True)
    sd_c, _ = oracle.calibrated_lcg_weights("object", 0, 40, rays[:256])
    sd_f, _ = oracle.calibrated_lcg_weights("object", 0, 41, rays[:256])
    t_vals = torch.linspace(0., 1., 64)
    with torch.no_grad():
        ref32 = oracle.render_rays(rays, sd_c, sd_f, cfg, t_vals=t_vals)
        to64 = lambda sd: {k: v.double() for k, v in sd.items()}
        ref64 = oracle.render_rays(rays.double(), to64(sd_c), to64(sd_f), cfg, t_vals=t_vals.double())
        score = oracle.conditioning_scores(rays, sd_c, sd_f, cfg, t_vals)
    good = (score <= 0.2).numpy()
    prec = _capi.default_precision() if a.precision is None else {"f32": _capi.PREC_F32, "f16x3": _capi.PREC_F16X3}[a.precision]
    desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, 1.0, prec)
    pc, pf = packing.pack_state_dict(desc, sd_c).to(dev), packing.pack_state_dict(desc, sd_f).to(dev)
    hip = kernels.render_rays_fused(desc, pc, pf, rays.to(dev), 64, 128, t_vals.to(dev), torch.linspace(0., 1., 128, device=dev),
                                    white_bkgd=True)
    rng = np.random.RandomState(0)
    print(f"# {n} rays ({side}x{side} crop of the 800x800 chair view), 64+128 samples, precision "
          f"{['f32', 'f16x3'][prec]}; {int(good.sum())} well-conditioned rays; acc range "
          f"[{float(ref32['acc_fine'].min()):.3f}, {float(ref32['acc_fine'].max()):.3f}]")
    print("# map       PSNR(oracle,T)   delta HIP-oracle [dB]   delta oracle fp32-fp64 [dB]   (well-conditioned rays only: same two deltas)")
    worst = 0.0
    for k in ("rgb", "albedo", "shading", "residual"):
        t64 = ref64[k + "_fine"].numpy().reshape(n, -1)
        T = t64 + 0.03 * rng.standard_normal(t64.shape)
        h = hip[k + "_fine"].cpu().numpy().reshape(n, -1)
        r = ref32[k + "_fine"].numpy().reshape(n, -1)
        d_all, n_all = psnr(h, T) - psnr(r, T), psnr(r, T) - psnr(t64, T)
        d_good, n_good = psnr(h[good], T[good]) - psnr(r[good], T[good]), psnr(r[good], T[good]) - psnr(t64[good], T[good])
        worst = max(worst, abs(d_good))
        print(f"{k:9s}   {psnr(r, T):9.4f}       {d_all:+.2e}              {n_all:+.2e}                    {d_good:+.2e}  {n_good:+.2e}")
    print(f"# worst |delta| on well-conditioned rays: {worst:.2e} dB (requirement 1e-4 dB)")
    return worst


if __name__ == "__main__":
    main()
