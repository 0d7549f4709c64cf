#!/usr/bin/env python3
"""Full-frame PSNR delta of the HIP path against the reference arithmetic on the HEADLINE workload (BASELINE.json: "<= 1e-4
dB PSNR delta vs reference" on Blender chair 800x800, 64+128): ALL 640,000 rays of bench.py's frame, bench.py's networks.

Two stages, because the CPU oracle (== reference, tests/golden) needs ~1 h for the frame in fp32 + fp64:

    python scripts/psnr_full_frame.py --oracle [--threads 6]     build container (CPU): oracle fp32 and fp64 maps of the whole
                                                                 frame -> gpurun_in/psnr_full_frame_oracle.npz (resumable)
    python scripts/psnr_full_frame.py --hip                      GPU box: render the frame through object_level.render (the timed
                                                                 path of bench.py), compare -> profiles/r03_psnr_full_frame.txt

PSNR(x, T) = -10 log10 mean (x - T)^2 (run_nerf_helpers.py:11-12).  No dataset image exists here, so T = the fp64 evaluation
of the same scene + a fixed N(0, sigma) perturbation with sigma = 10^(-30/20) (a trained IntrinsicNeRF reaches ~30 dB);
delta = PSNR(HIP, T) - PSNR(oracle fp32, T).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

STORE = os.path.join(REPO, "gpurun_in", "psnr_full_frame_oracle.npz")
PART = os.path.join(REPO, "gpurun_in", "psnr_full_frame_part_{:03d}.npz")
KEYS = (("rgb_map", "rgb_fine", 3), ("albedo_map", "albedo_fine", 3), ("shading_map", "shading_fine", 1), ("residual_map", "residual_fine", 3))
CHUNK = 16384


def frame_rays_cpu():
    """The frame's [640000, 11] ray batch on the CPU - the same bits bench.py's GPU rays have (tests/test_frames_gpu.py)."""
    from intrinsicnerf_amd import object_level as ol
    ro, rd = ol.get_rays(bench.H, bench.W, bench.chair_intrinsics(), bench.chair_pose())
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    vd = rd / rd.norm(dim=-1, keepdim=True)
    return torch.cat([ro, rd, bench.NEAR * torch.ones_like(vd[:, :1]), bench.FAR * torch.ones_like(vd[:, :1]), vd], -1).contiguous()


def bench_weights(rays):
    """bench.py's networks: default init (seeds 0 / 1), density head calibrated on the frame's strided 4077-ray sample."""
    from oracle import calibration as cal
    n_total = rays.shape[0]
    sel = torch.arange(0, n_total, n_total // bench.PARITY_RAYS + 1)[:bench.PARITY_RAYS]
    return cal.calibrated_default_init("object", 0, 0, rays[sel]), cal.calibrated_default_init("object", 0, 1, rays[sel])


def stage_oracle(threads):
    import oracle
    torch.set_num_threads(threads)
    os.makedirs(os.path.dirname(STORE), exist_ok=True)
    rays = frame_rays_cpu()
    sd_c, sd_f = bench_weights(rays)
    to64 = lambda sd: {k: v.double() for k, v in sd.items()}
    cfg = oracle.RenderConfig(variant="object", n_samples=bench.N_SAMPLES, n_importance=bench.N_IMPORTANCE, white_bkgd=True)
    n = rays.shape[0]
    t0 = time.time()
    for j, b in enumerate(range(0, n, CHUNK)):
        if os.path.exists(PART.format(j)):
            continue
        r = rays[b:b + CHUNK]
        with torch.no_grad():
            o32 = oracle.render_rays(r, sd_c, sd_f, cfg)
            o64 = oracle.render_rays(r.double(), to64(sd_c), to64(sd_f), cfg)
        cat = lambda o: np.concatenate([o[ok].numpy().reshape(len(r), w) for _, ok, w in KEYS], 1)
        np.savez(PART.format(j) + ".tmp.npz", o32=cat(o32).astype(np.float32), o64=cat(o64).astype(np.float64))
        os.replace(PART.format(j) + ".tmp.npz", PART.format(j))
        print(f"chunk {j + 1}/{(n + CHUNK - 1) // CHUNK} done, {time.time() - t0:.0f} s", flush=True)
    parts = [np.load(PART.format(j)) for j in range((n + CHUNK - 1) // CHUNK)]
    o64 = np.concatenate([p["o64"] for p in parts], 0)
    # the fp64 maps travel as (fp32 maps, float32 difference): o64 = o32 + d to ~1e-13, a third fewer bytes than float64 + float32
    o32 = np.concatenate([p["o32"] for p in parts], 0)
    np.savez(STORE, o32=o32, o64_minus_o32=(o64 - o32.astype(np.float64)).astype(np.float32))
    for j in range(len(parts)):
        os.remove(PART.format(j))
    print("wrote", STORE, os.path.getsize(STORE) >> 20, "MiB")


def stage_hip():
    import __graft_entry__
    __graft_entry__.build()
    from intrinsicnerf_amd import _capi, object_level as ol
    from oracle import stagewise
    dev = torch.device("cuda:0")
    z = np.load(STORE)
    o32 = z["o32"].astype(np.float64)
    o64 = o32 + z["o64_minus_o32"].astype(np.float64)
    rays = frame_rays_cpu()
    sd_c, sd_f = bench_weights(rays)
    embed, ch = ol.get_embedder(10, 0)
    embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net_c, net_f = mk(), mk()
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    kw = dict(network_fn=net_c, network_fine=net_f, network_query_fn=ol.NetworkQuery(embed, embed_d), N_samples=bench.N_SAMPLES,
              N_importance=bench.N_IMPORTANCE, white_bkgd=True, perturb=False, raw_noise_std=0., use_viewdirs=True, ndc=False, lindisp=False)
    lines = [f"# full-frame PSNR delta, Blender chair 800x800 (640000 rays), 64+128, bench.py's calibrated default-init networks",
             f"# T = oracle fp64 + N(0, 10^-1.5) (PSNR(fp64, T) = 30 dB); delta = PSNR(HIP, T) - PSNR(oracle fp32, T); {torch.cuda.get_device_name(0)}",
             "# precision   map           PSNR(HIP,T) dB   PSNR(oracle32,T) dB   delta dB      | oracle fp32 vs fp64 delta dB"]
    worst = {}
    for name, prec in (("f16x3", _capi.PREC_F16X3), ("f32", _capi.PREC_F32)):
        with torch.no_grad(), _capi.forced_precision(prec):
            r = ol.render(bench.H, bench.W, bench.chair_intrinsics(), chunk=rays.shape[0],
                          rays=(rays[:, 0:3].to(dev).contiguous(), rays[:, 3:6].to(dev).contiguous()), near=bench.NEAR, far=bench.FAR, **kw)
        maps = dict(zip(("rgb_map", "disp_map", "acc_map", "albedo_map", "shading_map", "residual_map"), r[:6]))
        c = 0
        for fk, _, w in KEYS:
            hip = maps[fk].reshape(rays.shape[0], w).cpu().numpy().astype(np.float64)
            a, b = o32[:, c:c + w], o64[:, c:c + w]
            rng = np.random.RandomState(0)
            target = b + rng.randn(*b.shape) * 10.0 ** (-30.0 / 20.0)
            p_h, p_o, p_64 = stagewise.psnr(hip, target), stagewise.psnr(a, target), stagewise.psnr(b, target)
            lines.append(f"{name:10s}  {fk:12s}  {p_h:.9f}    {p_o:.9f}         {p_h - p_o:+.3e}    | {p_o - p_64:+.3e}")
            worst[name] = max(worst.get(name, 0.0), abs(p_h - p_o))
            c += w
    for name, v in worst.items():
        lines.append(f"# worst |delta| over the four maps, {name}: {v:.3e} dB  (budget 1e-4 dB)")
    out = os.path.join(REPO, "gpurun_out", "r03_psnr_full_frame.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--hip", action="store_true")
    ap.add_argument("--threads", type=int, default=6)
    a = ap.parse_args()
    if a.oracle:
        stage_oracle(a.threads)
    if a.hip:
        stage_hip()
