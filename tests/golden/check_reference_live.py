#!/usr/bin/env python3
"""Randomised cross-check of the oracle against the REAL reference, run live (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_reference_live.py [--cases 12] [--seed 0]

The committed fixtures pin the oracle on eleven hand-picked configurations; this script sweeps what they do not: random
ray counts (1 .. 40), sample counts, importance counts (0 included), white background / lindisp / perturb / noise flags,
class counts, the endpoint feature, default `nn.Linear` initialisation instead of the closed-form weights, unconditioned
rays - and asserts oracle == reference (<= 2e-6, NaN patterns included) on every returned tensor, for
`object_level/run_nerf.render_rays`, `SSRTrainer.render_rays` and `Cluster_Manager.dest_color / dest_class`.
tests/test_oracle_golden.py runs it in a subprocess when /root/reference is mounted (the import recipe patches
`torch.Tensor.cuda` globally) and skips otherwise.  Nothing is written.
"""
import argparse
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
import oracle  # noqa: E402


def object_case(run_nerf, H_ref, rng, idx):
    n = int(rng.integers(1, 41))
    n_imp = int(rng.choice([0, 1, 16, 128]))
    s = int(rng.choice([4, 7, 64] if n_imp > 0 else [2, 7, 64]))      # the reference itself needs >= 3 coarse samples to resample
    white, lindisp, train = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
    cfg = oracle.RenderConfig(variant="object", n_samples=s, n_importance=n_imp, white_bkgd=white, lindisp=lindisp)
    rays = mg.chair_rays(H_ref, n, 1000 + idx)
    embed, ch = H_ref.get_embedder(10, 0)
    embed_d, ch_d = H_ref.get_embedder(4, 0)
    torch.manual_seed(idx)
    mk = lambda: H_ref.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True)
    net_c, net_f = mk(), mk()
    with torch.no_grad():                      # default init gives sigma ~ 0: lift it so that weights / cdf are non-trivial
        net_c.alpha_linear.bias += float(rng.uniform(0.0, 3.0))
        net_f.alpha_linear.bias += float(rng.uniform(0.0, 3.0))
    sd_c, sd_f = net_c.state_dict(), net_f.state_dict()
    q = lambda x, v, fn: run_nerf.run_network(x, v, fn, embed_fn=embed, embeddirs_fn=embed_d, netchunk=65536)
    extra = {}
    if train:
        g = torch.Generator().manual_seed(77 + idx)
        extra = dict(t_rand=torch.rand(n, s, generator=g), noise_coarse=torch.rand(n, s, generator=g))
        if n_imp > 0:
            extra.update(u=torch.rand(n, n_imp, generator=g), noise_fine=torch.rand(n, s + n_imp, generator=g))
    feed = [extra[k] for k in ("t_rand", "noise_coarse", "u", "noise_fine") if k in extra]
    with torch.no_grad(), mg.injected_rng(np_rand=feed):
        ref = run_nerf.render_rays(rays, net_c, q, s, retraw=True, lindisp=lindisp, perturb=1.0 if train else 0.0,
                                   N_importance=n_imp, network_fine=net_f, white_bkgd=white,
                                   raw_noise_std=1.0 if train else 0.0, pytest=train)
        mine = oracle.render_rays(rays, sd_c, sd_f, cfg, t_vals=torch.linspace(0.0, 1.0, s), stages=True, **extra)
    suffix = "fine" if n_imp > 0 else "coarse"
    pairs = [(k + "_map", f"{k}_{suffix}") for k in ("rgb", "disp", "acc", "albedo", "shading", "residual")] + [("raw", "raw_" + suffix)]
    if n_imp > 0:
        pairs += [(k + "0", k + "_coarse") for k in ("rgb", "disp", "acc", "albedo", "shading", "residual")] + [("z_std", "z_std")]
    worst = max(mg.check_same(f"object#{idx}/{rk}", ref[rk], mine[ok]) for rk, ok in pairs)
    return f"object n={n} S={s}+{n_imp} white={white:d} lindisp={lindisp:d} train={train:d}: {worst:.1e}"


def ssr_case(SSRTrainer, ssr_rays, rng, idx):
    n = int(rng.integers(1, 41))
    c = int(rng.choice([0, 1, 5, 28, 101]))
    endpoint, white, train = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
    n_imp = int(rng.choice([16, 128]))
    cfg = oracle.RenderConfig(variant="ssr", n_samples=64, n_importance=n_imp, white_bkgd=white, n_classes=c,
                              endpoint_feat=endpoint, netchunk=32768)
    rays = mg.room_rays(ssr_rays, n, 2000 + idx)
    torch.manual_seed(100 + idx)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = mg.ssr_trainer(SSRTrainer, c, endpoint, white, train, n_importance=n_imp)
    with torch.no_grad():
        tr.ssr_net_coarse.alpha_linear.bias += float(rng.uniform(0.0, 3.0))
        tr.ssr_net_fine.alpha_linear.bias += float(rng.uniform(0.0, 3.0))
    sd_c, sd_f = tr.ssr_net_coarse.state_dict(), tr.ssr_net_fine.state_dict()
    extra = {}
    if train:
        g = torch.Generator().manual_seed(88 + idx)
        extra = dict(t_rand=torch.rand(n, 64, generator=g), noise_coarse=torch.randn(n, 64, generator=g),
                     u=torch.rand(n, n_imp, generator=g), noise_fine=torch.randn(n, 64 + n_imp, generator=g))
    feed = mg.injected_rng(torch_rand=[extra[k] for k in ("t_rand", "u") if k in extra],
                           torch_randn=[extra[k] for k in ("noise_coarse", "noise_fine") if k in extra])
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()), feed:
        ref = tr.render_rays(rays)
    with torch.no_grad():
        mine = oracle.render_rays(rays, sd_c, sd_f, cfg, t_vals=torch.linspace(0.0, 1.0, 64), stages=True, **extra)
    keys = ["rgb", "disp", "acc", "depth", "albedo", "shading", "residual"]
    pairs = [(f"{k}_{lvl}", f"{k}_{lvl}") for lvl in ("coarse", "fine") for k in keys]
    pairs += [("raw_coarse", "raw_coarse"), ("raw_fine", "raw_fine"), ("z_std", "z_std")]
    if c > 0:
        pairs += [("sem_logits_coarse", "sem_coarse"), ("sem_logits_fine", "sem_fine")]
    if endpoint:
        pairs += [("feat_map_fine", "feat_fine")]
    worst = max(mg.check_same(f"ssr#{idx}/{rk}", ref[rk], mine[ok]) for rk, ok in pairs)
    return f"ssr n={n} C={c} imp={n_imp} endpoint={endpoint:d} white={white:d} train={train:d}: {worst:.1e}"


def cluster_case(ref_cluster, rng, idx):
    k = int(rng.integers(1, 7))
    cpu = torch.device("cpu")
    mgr = ref_cluster.Cluster_Manager(class_num=k)
    for i in range(k):
        if k > 1 and rng.integers(4) == 0:
            mgr.clusters.append(None)
            continue
        c = ref_cluster.Cluster(device=cpu, intensity_factor=float(rng.uniform(0.2, 0.9)))
        a, m = int(rng.integers(1, 400)), int(rng.integers(1, 6))
        c.anchors = torch.from_numpy(rng.uniform(0, 1, size=(a, 3)).astype(np.float32))
        c.links = torch.from_numpy(rng.integers(0, m, size=(a, 1)))
        c.rgb_centers = torch.from_numpy(rng.uniform(0, 1, size=(m, 3)).astype(np.float32))
        c.batch_size = int(rng.choice([7, 100, 10240]))
        mgr.clusters.append(c)
    n = int(rng.integers(1, 300))
    rgb = torch.from_numpy(rng.uniform(0.01, 1, size=(n, 3)).astype(np.float32))
    label = torch.from_numpy(rng.integers(-1, k + 1, size=(n, 1)))
    clusters = [None if c is None else {"anchors": c.anchors, "links": c.links, "rgb_centers": c.rgb_centers,
                                        "intensity_factor": c.intensity_factor, "batch_size": c.batch_size} for c in mgr.clusters]
    assert torch.equal(mgr.dest_color(rgb, label), oracle.cluster.dest_color(clusters, rgb, label)), f"cluster#{idx}: dest_color"
    assert torch.equal(mgr.dest_class(rgb, label), oracle.cluster.dest_class(clusters, rgb, label)), f"cluster#{idx}: dest_class"
    return f"cluster K={k} n={n}: identical"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    run_nerf, H_ref, SSRTrainer, ssr_rays, _ = mg.import_reference()
    from SSR.training import cluster as ref_cluster
    rng = np.random.default_rng(a.seed)
    for i in range(a.cases):
        print(object_case(run_nerf, H_ref, rng, i))
        print(ssr_case(SSRTrainer, ssr_rays, rng, i))
        print(cluster_case(ref_cluster, rng, i))
    print(f"oracle == reference on {3 * a.cases} random configurations")


if __name__ == "__main__":
    main()
