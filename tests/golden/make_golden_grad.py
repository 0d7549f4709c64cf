#!/usr/bin/env python3
"""Golden GRADIENT vectors, produced by the reference's own autograd (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grad.py

Companion of make_golden.py (same import recipe, same rules: the reference is imported from /root/reference,
never copied; only inputs and reference outputs are stored).  What the trainers do with the path is
``loss.backward()`` (run_nerf.py:1018, trainer.py:990); a scalar loss is a weighted sum of the returned maps,
so its gradient is fixed by one cotangent tensor per map.  Each fixture stores seeded cotangents and what the
REFERENCE's autograd returns for them:

  grad_composite_{object,ssr}_wb{0,1}.npz   raw2outputs alone: d loss / d raw                (stage boundary)
  grad_render_object.npz                    render_rays, coarse + fine: d loss / d every network parameter,
                                            stored as digests (norm, leading entries, a seeded projection)

and asserts that autograd through the CPU oracle (oracle/intrinsic_render.py) gives the same numbers, so the
oracle's backward is pinned like its forward.
"""
import os
import sys

sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))

import make_golden as mg  # noqa: E402
import oracle  # noqa: E402
from _cases import case_config, case_weights  # noqa: E402
from conftest import load_golden  # noqa: E402

DIGEST_HEAD = 16


def cotangents(outs, seed, scale=None):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(v.shape, generator=g) * (1.0 if scale is None else scale.get(k, 1.0)) for k, v in outs.items()}


def digest(t, seed):
    """norm, first entries and a seeded random projection of a gradient tensor."""
    t = t.detach().double().flatten()
    g = torch.Generator().manual_seed(seed)
    proj = torch.randn(t.numel(), generator=g, dtype=torch.float64)
    return np.concatenate([[float(t.norm())], [float((t * proj).sum())], t[:DIGEST_HEAD].numpy()])


def composite_cases(run_nerf, ssr_mu):
    g = torch.Generator().manual_seed(5)
    n, s = 12, 64
    z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, dim=-1)[0]
    rays_d = torch.randn(n, 3, generator=g)
    raw = torch.rand(n, s, 11, generator=g)
    raw[..., 3] = torch.randn(n, s, generator=g) * 4
    raw[0, :, 3] = -1.0                      # acc == 0: disp (and its gradient) NaN
    raw[1, :, 3] = -1.0; raw[1, -1, 3] = 2.0  # only the 1e10 interval is opaque
    raw[2, :, 3] = 0.0                       # relu'(0) = 0
    raw[3, :, 3] = 1e4                       # saturated first sample: 1 - alpha + 1e-10 = 1e-10 in the denominators
    raw[4, :, 3] = 1e-6
    z[5] = z[5, 0]                           # zero-length intervals
    noise = torch.rand(n, s, generator=g) * 0.5
    names = ["rgb", "disp", "acc", "weights", "depth", "albedo", "shading", "residual"]
    for wb in (False, True):
        for with_noise in (False, True):
            r = raw.clone().requires_grad_(True)
            if with_noise:      # the reference adds randn * std inside (run_nerf.py:386-387): feed the stored tensor (std = 1)
                with mg.injected_rng(torch_randn=[noise]):
                    ref = run_nerf.raw2outputs(r, z, rays_d, 1.0, wb)
            else:
                ref = run_nerf.raw2outputs(r, z, rays_d, 0, wb)
            outs = dict(zip(names, ref))
            cot = cotangents(outs, 100 + int(wb))
            loss = sum((cot[k] * v).sum() for k, v in outs.items())      # NaN where disp is NaN; the gradient is what is kept
            (d_ref,) = torch.autograd.grad(loss, r)
            r2 = raw.clone().requires_grad_(True)
            mine = oracle.composite(r2, z, rays_d, oracle.RenderConfig(variant="object", white_bkgd=wb), noise if with_noise else None)
            loss2 = sum((cot[k] * mine[k]).sum() for k in names)
            (d_mine,) = torch.autograd.grad(loss2, r2)
            mg.check_same(f"grad_composite_object wb={wb} noise={with_noise}", d_ref, d_mine, tol=1e-6)
            if with_noise:
                mg.save(f"grad_composite_object_wb{int(wb)}", raw=raw, z=z, rays_d=rays_d, noise=noise, white_bkgd=wb,
                        d_raw_noise=d_ref, **{"cot_" + k: v for k, v in cot.items()}, d_raw=d_plain)
            else:
                d_plain = d_ref
    # SSR flavour: semantic logits + endpoint feature, S = 192
    n, s, c = 6, 192, 7
    z = torch.sort(torch.rand(n, s, generator=g) * 9.9 + 0.1, dim=-1)[0]
    rays_d = torch.randn(n, 3, generator=g)
    raw = torch.randn(n, s, 11 + c + 128, generator=g)
    raw[0, :, 3] = -3.0
    names = ["rgb", "disp", "acc", "weights", "depth", "sem", "feat", "albedo", "shading", "residual"]
    for wb in (False, True):
        r = raw.clone().requires_grad_(True)
        ref = ssr_mu.raw2outputs(r, z, rays_d, 0, wb, enable_semantic=True, num_sem_class=c, endpoint_feat=True)
        outs = dict(zip(names, ref))
        cot = cotangents(outs, 200 + int(wb))
        loss = sum((cot[k] * v).sum() for k, v in outs.items())
        (d_ref,) = torch.autograd.grad(loss, r)
        r2 = raw.clone().requires_grad_(True)
        cfg = oracle.RenderConfig(variant="ssr", white_bkgd=wb, n_classes=c, endpoint_feat=True)
        mine = oracle.composite(r2, z, rays_d, cfg, feat=True)
        (d_mine,) = torch.autograd.grad(sum((cot[k] * mine[k]).sum() for k in names), r2)
        mg.check_same(f"grad_composite_ssr wb={wb}", d_ref, d_mine, tol=1e-6)
        mg.save(f"grad_composite_ssr_wb{int(wb)}", raw=raw, z=z, rays_d=rays_d, white_bkgd=wb, n_classes=c,
                d_raw=d_ref, **{"cot_" + k: v for k, v in cot.items()})
    print("grad_composite_*: oracle autograd == reference autograd")


def render_case(run_nerf, H_ref):
    """Parameter gradients of the whole object-level path (training-step shape: coarse + fine nets, white background),
    on the rays of an existing forward fixture."""
    fx = load_golden("object_chair_det")
    cfg = case_config(fx)
    sd_c, sd_f = case_weights(fx)
    rays = torch.from_numpy(fx["rays"])[:8].contiguous()
    embed, ch = H_ref.get_embedder(10, 0)
    embed_d, ch_d = H_ref.get_embedder(4, 0)
    mk = lambda: H_ref.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True)
    net_c, net_f = mk(), mk()
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    q = lambda x, v, fn: run_nerf.run_network(x, v, fn, embed_fn=embed, embeddirs_fn=embed_d, netchunk=65536)
    ref = run_nerf.render_rays(rays, net_c, q, 64, retraw=True, lindisp=False, perturb=0.0, N_importance=128,
                               network_fine=net_f, white_bkgd=True, raw_noise_std=0.0)
    # the maps the training loss reads (run_nerf.py:976-1008): fine and coarse rgb / albedo / shading / residual, disp, acc
    keys = ["rgb_map", "albedo_map", "shading_map", "residual_map", "disp_map", "acc_map",
            "rgb0", "albedo0", "shading0", "residual0", "acc0"]
    cot = cotangents({k: ref[k] for k in keys}, 300, scale={"disp_map": 0.1})
    loss = sum((cot[k] * ref[k]).sum() for k in keys)
    loss.backward()
    # oracle autograd on the same thing
    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    mine = oracle.render_rays(rays, pc, pf, cfg, t_vals=torch.from_numpy(fx["t_vals"]))
    name_of = {"rgb_map": "rgb_fine", "albedo_map": "albedo_fine", "shading_map": "shading_fine", "residual_map": "residual_fine",
               "disp_map": "disp_fine", "acc_map": "acc_fine", "rgb0": "rgb_coarse", "albedo0": "albedo_coarse",
               "shading0": "shading_coarse", "residual0": "residual_coarse", "acc0": "acc_coarse"}
    sum((cot[k] * mine[name_of[k]]).sum() for k in keys).backward()
    out = dict(rays=rays, n_rays=8, **{"cot_" + name_of[k]: v for k, v in cot.items()})
    worst = 0.0
    for tag, net, params in (("coarse", net_c, pc), ("fine", net_f, pf)):
        for i, (name, p) in enumerate(net.named_parameters()):
            assert p.grad is not None, name
            gn = float(p.grad.double().norm())
            dev = float((p.grad.double() - params[name].grad.double()).norm()) / max(gn, 1e-30)
            worst = max(worst, dev)
            out[f"grad_{tag}/{name}"] = digest(p.grad, 1000 + i)
    assert worst <= 1e-5, f"oracle autograd deviates from the reference's by {worst:.2e} (relative, per tensor)"
    print(f"grad_render_object: oracle autograd == reference autograd (worst per-tensor relative deviation {worst:.1e})")
    mg.save("grad_render_object", source_fixture="object_chair_det", **out)


def main():
    run_nerf, H_ref, SSRTrainer, ssr_rays, ssr_mu = mg.import_reference()
    composite_cases(run_nerf, ssr_mu)
    render_case(run_nerf, H_ref)


if __name__ == "__main__":
    main()
