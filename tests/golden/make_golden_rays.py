#!/usr/bin/env python3
"""Golden vectors for the ray generators and for render()'s ray-batch assembly, from the reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_rays.py

Stores, for a small camera: run_nerf_helpers.get_rays / get_rays_np / ndc_rays, SSR rays.create_rays (both conventions, both
depth types, with a static camera) and the [N, 11] ray batch that run_nerf.render builds from a pose (captured at its call of
batchify_rays: run_nerf.py:122-131) for ndc on / off and c2w_staticcam.  The 2^9-frequency encoding amplifies a one-ulp
difference in a ray direction to 2e-4, so these have to be reproduced bit for bit (tests/test_frontends_cpu.py).
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    run_nerf, H_ref, SSRTrainer, ssr_rays, ssr_mu = mg.import_reference()
    H, W = 6, 8
    focal = 7.25
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    g = torch.Generator().manual_seed(3)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    c2w = torch.cat([q, torch.tensor([[0.3], [-1.2], [2.5]])], 1)
    q2, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    c2w_static = torch.cat([q2, torch.tensor([[-0.4], [0.7], [1.5]])], 1)
    out = dict(H=H, W=W, K=K, focal=focal, c2w=c2w, c2w_static=c2w_static)
    ro, rd = H_ref.get_rays(H, W, K, c2w)
    out["get_rays_o"], out["get_rays_d"] = ro, rd
    ro_np, rd_np = H_ref.get_rays_np(H, W, K, c2w.numpy())
    out["get_rays_np_o"], out["get_rays_np_d"] = ro_np, rd_np
    o2, d2 = H_ref.ndc_rays(H, W, focal, 1.0, ro + torch.tensor([0.0, 0.0, 4.0]), rd)
    out["ndc_o"], out["ndc_d"] = o2, d2
    # render(): capture the assembled ray batch
    captured = {}
    orig = run_nerf.batchify_rays

    def spy(rays_flat, chunk=1024 * 32, **kw):
        captured["rays"] = rays_flat.clone()
        n = rays_flat.shape[0]
        z3, z1 = torch.zeros(n, 3), torch.zeros(n)
        return {"rgb_map": z3, "disp_map": z1, "acc_map": z1, "albedo_map": z3, "shading_map": z1, "residual_map": z3}

    run_nerf.batchify_rays = spy
    try:
        for tag, kw in (("plain", dict(ndc=False)), ("ndc", dict(ndc=True)), ("static", dict(ndc=False, c2w_staticcam=c2w_static))):
            run_nerf.render(H, W, K, chunk=64, c2w=c2w, near=2.0, far=6.0, use_viewdirs=True, **kw)
            out["render_rays_" + tag] = captured["rays"]
        rays_in = (ro.reshape(-1, 3)[:10] * 1.5, rd.reshape(-1, 3)[:10] * 0.7)
        run_nerf.render(H, W, K, chunk=64, rays=rays_in, ndc=False, near=0.5, far=3.0, use_viewdirs=True)
        out["render_rays_given"] = captured["rays"]
        out["given_o"], out["given_d"] = rays_in
    finally:
        run_nerf.batchify_rays = orig
    # SSR (the reference's create_rays prints rays_cam[0, 1, 11] and dirs_C[0, 331]: it needs >= 12 columns and >= 332 pixels)
    H, W = 20, 20
    out["H_ssr"], out["W_ssr"] = H, W
    T = torch.eye(4)[None].repeat(2, 1, 1)
    T[0, :3, :3], T[0, :3, 3] = q, torch.tensor([0.1, 0.2, 0.3])
    T[1, :3, :3], T[1, :3, 3] = q2, torch.tensor([-1.0, 0.5, 2.0])
    out["ssr_T"] = T
    for conv in ("opencv", "opengl"):
        for dt in ("z", "euclidean"):
            out[f"ssr_rays_{conv}_{dt}"] = ssr_rays.create_rays(2, T, H, W, 5.5, 6.5, 9.5, 9.5, 0.1, 10.0, depth_type=dt, convention=conv)
    out["ssr_rays_static"] = ssr_rays.create_rays(2, T, H, W, 5.5, 6.5, 9.5, 9.5, 0.1, 10.0, c2w_staticcam=T.flip(0))
    mg.save("rays_generators", **out)


if __name__ == "__main__":
    main()
