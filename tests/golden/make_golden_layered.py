#!/usr/bin/env python3
"""Golden vectors for networks OUTSIDE the fused architecture, produced by the real reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_layered.py

The reference builds its networks from flags (object_level/run_nerf.py:286-296: ``NeRF(D=args.netdepth, W=args.netwidth,
...)``; SSR/training/trainer.py:811-846).  Each fixture holds one such non-default network (its state dict - these are small),
rays, depths, and what the REFERENCE's own ``run_network`` + ``NeRF.forward`` / ``Semantic_NeRF.forward`` return for them on CPU,
together with the parameter gradients the reference's autograd gives for a seeded cotangent on ``raw``:

  layered_object_d4_w128.npz     NeRF(D=4, W=128, skips=[2]), multires 10 / 4
  layered_object_d5_w80_2skips   NeRF(D=5, W=80, skips=[1, 3]), multires 6 / 2 (a width that is no multiple of 32, two skips)
  layered_object_noviews         NeRF(D=3, W=64, skips=[1], use_viewdirs=False, output_ch=5) (run_nerf_helpers.py:281-282, 323)
  layered_ssr_d5_w64_c5          Semantic_NeRF(D=5, W=64, skips=[2], 5 classes), x / 10 encoder, with the endpoint feature

Same rules as make_golden.py: the reference is imported from /root/reference, never copied; only inputs and outputs are stored.
"""
import os
import sys

sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import make_golden as mg  # noqa: E402


def inputs(H_ref, n_rays, n_samples, seed, near, far):
    rays = mg.chair_rays(H_ref, n_rays, seed, near, far)
    g = torch.Generator().manual_seed(seed)
    z = torch.sort(torch.rand(n_rays, n_samples, generator=g) * (far - near) + near, dim=-1)[0]
    return rays, z


def init(net, seed, gain=1.0):
    """torch's default nn.Linear init under a fixed seed (what a freshly created network holds), biases widened a little so that
    every ReLU mask has both states."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in net.parameters():
            bound = gain / np.sqrt(p.shape[-1] if p.dim() > 1 else max(p.shape[0], 1))
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound * (1.0 if p.dim() > 1 else 0.5))


def run_case(name, run_network, net, embed, embed_d, rays, z, call=None, **meta):
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., :, None]              # run_nerf.py:488
    viewdirs = rays[:, 8:11] if embed_d is not None else None
    raw = run_network(pts, viewdirs, call or net, embed, embed_d, 1024 * 64)
    g = torch.Generator().manual_seed(77)
    cot = torch.randn(raw.shape, generator=g)
    (cot * raw).sum().backward()
    grads = {f"grad/{k}": p.grad for k, p in net.named_parameters() if p.grad is not None}
    unused = [k for k, p in net.named_parameters() if p.grad is None]
    print(f"{name}: raw {tuple(raw.shape)}, |raw| max {float(raw.abs().max()):.3f}, {len(grads)} gradients"
          + (f", unused by the forward: {unused}" if unused else ""))
    mg.save(name, rays=rays, z=z, raw=raw, cot=cot, **{f"param/{k}": v for k, v in net.state_dict().items()}, **grads, **meta)


def main():
    run_nerf, H_ref, SSRTrainer, ssr_rays, ssr_mu = mg.import_reference()
    from SSR.models import semantic_nerf as ssr_models

    # ---- object level ----
    for name, D, W, skips, l_xyz, l_dir, seed in (("layered_object_d4_w128", 4, 128, [2], 10, 4, 11),
                                                  ("layered_object_d5_w80_2skips", 5, 80, [1, 3], 6, 2, 12)):
        embed, ch = H_ref.get_embedder(l_xyz, 0)
        embed_d, ch_d = H_ref.get_embedder(l_dir, 0)
        net = H_ref.NeRF(D=D, W=W, input_ch=ch, output_ch=5, skips=skips, input_ch_views=ch_d, use_viewdirs=True)
        init(net, seed, gain=2.0)
        rays, z = inputs(H_ref, 24, 40, seed, 2.0, 6.0)
        run_case(name, run_nerf.run_network, net, embed, embed_d, rays, z, variant="object", D=D, W=W, skips=np.array(skips),
                 l_xyz=l_xyz, l_dir=l_dir, use_viewdirs=True, xyz_div=1.0)

    embed, ch = H_ref.get_embedder(5, 0)
    net = H_ref.NeRF(D=3, W=64, input_ch=ch, output_ch=5, skips=[1], input_ch_views=0, use_viewdirs=False)
    init(net, 13, gain=2.0)
    rays, z = inputs(H_ref, 16, 24, 13, 2.0, 6.0)
    run_case("layered_object_noviews", run_nerf.run_network, net, embed, None, rays, z, variant="object", D=3, W=64,
             skips=np.array([1]), l_xyz=5, l_dir=0, use_viewdirs=False, xyz_div=1.0)

    # ---- SSR: x / 10 encoder (semantic_nerf.py:50-66), semantic head, endpoint feature (trainer.py:770) ----
    embed, ch = ssr_models.get_embedder(8, 0, scalar_factor=10)
    embed_d, ch_d = ssr_models.get_embedder(3, 0, scalar_factor=1)
    net = ssr_models.Semantic_NeRF(enable_semantic=True, num_semantic_classes=5, D=5, W=64, input_ch=ch, output_ch=5, skips=[2],
                                   input_ch_views=ch_d, use_viewdirs=True)
    init(net, 14, gain=2.0)
    rays, z = inputs(H_ref, 20, 48, 14, 0.1, 10.0)
    run_case("layered_ssr_d5_w64_c5", ssr_mu.run_network, net, embed, embed_d, rays, z, call=lambda x: net(x, True), variant="ssr",
             D=5, W=64, skips=np.array([2]), l_xyz=8, l_dir=3, use_viewdirs=True, xyz_div=10.0, n_classes=5, endpoint=True)


if __name__ == "__main__":
    main()
