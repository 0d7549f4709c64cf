#!/usr/bin/env python3
"""Golden fixture on a TRAINED network: the REAL reference on weights that a training run produced (VERDICT r02 missing #2:
"every weight set is default-init or closed-form").

    python scripts/fit_synthetic.py --save-weights gpurun_out/trained.pt        (on the GPU box: 3 000 steps on the analytic scene)
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trained.py gpurun_out/trained.pt      (build container only)
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trained.py                           (no checkpoint: the weights
                                                                       stored in the existing fixture, other rays / more rays)

The weights come from scripts/fit_synthetic.py - both networks fitted through the product's own training path (the
reference's step, run_nerf.py:868-1027) to an analytic 5-blob scene, held-out view at 48.6 dB - and are stored IN the fixture
(2 x 662 152 fp32 parameters).  Everything else is make_golden_uncurated.py's recipe: ALL 4 096 rays of the held-out 64x64
view (round 6; until then every 8th), nothing filtered, raw rows of every 16th; the reference's ``render_rays`` runs on them with its ``raw2outputs`` / ``sample_pdf`` wrapped so that
its own stage tensors are recorded; the oracle is asserted to reproduce every output and stage tensor bit for bit; the same
arithmetic in fp64 is stored next to it.
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_uncurated as mu  # noqa: E402

import oracle  # noqa: E402
from oracle import calibration as cal  # noqa: E402


def held_out_rays(H_ref, side=64, theta=40.0, every=1):
    focal = 0.5 * side / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 0.5 * side], [0, focal, 0.5 * side], [0, 0, 1]])
    ro, rd = H_ref.get_rays(side, side, K, mg.pose_spherical(theta, -30.0, 4.0)[:3, :4])
    ro, rd = ro.reshape(-1, 3)[::every].float(), rd.reshape(-1, 3)[::every].float()
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    return torch.cat([ro, rd, 2.0 * torch.ones_like(rd[:, :1]), 6.0 * torch.ones_like(rd[:, :1]), vd], -1).contiguous()


def main(path):
    run_nerf, H_ref, _, _, _ = mg.import_reference()
    if path and os.path.exists(path):
        ck = torch.load(path, map_location="cpu")
        sd_c = {k: v.float().contiguous() for k, v in ck["coarse"].items()}
        sd_f = {k: v.float().contiguous() for k, v in ck["fine"].items()}
    else:           # the trained weights travel inside the fixture: regenerate it (other rays, more rays) from its own copy
        old = np.load(os.path.join(HERE, "trained_object_chair.npz"))
        sd_c, sd_f = ({k.split("/", 1)[1]: torch.from_numpy(np.array(old[k])) for k in old.files if k.startswith(f"w_{lvl}/")} for lvl in ("coarse", "fine"))
        print(f"weights: the {len(sd_c)} + {len(sd_f)} tensors stored in trained_object_chair.npz")
    rays = held_out_rays(H_ref)
    n = rays.shape[0]
    cfg = oracle.RenderConfig(variant="object", n_samples=64, n_importance=128, white_bkgd=True)
    embed, ch = H_ref.get_embedder(10, 0)
    embed_d, ch_d = H_ref.get_embedder(4, 0)
    mk = lambda: H_ref.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True)
    net_c, net_f = mk(), mk()
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    q = lambda x, v, fn: run_nerf.run_network(x, v, fn, embed_fn=embed, embeddirs_fn=embed_d, netchunk=65536)
    with torch.no_grad():
        with mu.captured_stages(run_nerf) as cap:
            ref = run_nerf.render_rays(rays, net_c, q, 64, retraw=False, perturb=0.0, N_importance=128, network_fine=net_f,
                                       white_bkgd=True, raw_noise_std=0.0)
        mine = oracle.render_rays(rays, sd_c, sd_f, cfg, stages=True)
        m64 = oracle.render_rays(rays.double(), mu.to64(sd_c), mu.to64(sd_f), cfg, stages=True)
    pairs = [(f"{k}_map", f"{k}_fine") for k in ("rgb", "disp", "acc", "albedo", "shading", "residual")]
    pairs += [(f"{k}0", f"{k}_coarse") for k in ("rgb", "disp", "acc", "albedo", "shading", "residual")] + [("z_std", "z_std")]
    worst = max(mg.check_same(f"trained/{rk}", ref[rk], mine[ok], tol=0.0) for rk, ok in pairs)
    acc = mine["acc_fine"]
    print(f"trained_object_chair: oracle == reference on {n} rays of the held-out view (max dev {worst:.1e}); acc quantiles "
          f"{[round(float(torch.quantile(acc, x)), 3) for x in (0., .1, .5, .9, 1.)]}; rays with acc == 0: {int((acc == 0).sum())}")
    fx = dict(variant="object", n_classes=0, n_importance=128, white_bkgd=True, rays=rays)
    for name in sd_c:
        fx["w_coarse/" + name] = sd_c[name]
        fx["w_fine/" + name] = sd_f[name]
    for rk, ok in pairs:
        fx["ref_" + ok] = ref[rk]
        fx["f64_" + ok] = m64[ok]
    for k in ("z_samples", "weights_coarse", "weights_fine", "z_fine"):
        fx["stage_score_" + k] = cal.scaled_errors(mine[k].numpy(), m64[k].numpy())
    fx["stage_score_fine_pass_hazard"] = mu.fine_hazard(fx, rays, sd_f, cfg, mine, m64, pairs)
    fx.update(cap.fixture_entries(mine, torch.arange(0, n, 16)))
    mg.save("trained_object_chair", **fx)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
