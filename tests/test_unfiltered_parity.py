"""GPU parity on UN-CURATED inputs: unfiltered rays of the benchmark frames, default-``nn.Linear``-init networks whose
density head is calibrated so that acc spans (0, 1] (oracle/calibration.py), every ray judged.

With white-spectrum random weights many rays are ill-conditioned for any fp32 evaluation of the two-pass path (the
reference's own fp32 result is far from the same arithmetic in fp64 on them), so a per-ray 1e-4 assertion cannot hold
for ANY implementation.  Instead of discarding those rays the test compares distributions, per output map:

  e_hip[r] = |HIP - oracle_fp32| in units of the tolerance (1e-5 + 1e-4 |want|; disp: 5e-4)
  e_ref[r] = |oracle_fp32 - oracle_fp64| in the same units

and requires (oracle.calibration.rank_report) every quantile (50 ... 99 %) of e_hip <= max(0.5, 3 x the same quantile
of e_ref) and, for the tails, #(e_hip > T) <= 3 x #(e_ref > T) + 3 for T in {1, 10, 100} tolerances.  On top of that the
plain tolerance must hold on EVERY ray whose fp32-vs-fp64 score (max over all maps and stage tensors) is <= 0.2 -
and on every ray for the coarse maps, which do not sit behind sample_pdf.
"""
import functools

import numpy as np
import pytest
import torch

import oracle
from _cases import uncurated_config, uncurated_judge, uncurated_weights
from conftest import golden_names, load_golden
from oracle import calibration as cal

pytestmark = pytest.mark.gpu

N_RAYS = 1024
RTOL, ATOL, RTOL_DISP = 1e-4, 1e-5, 5e-4


@pytest.fixture(autouse=True, params=["f32", "f16x3", "f16x3-1wg"])
def precision(request, monkeypatch):
    monkeypatch.setenv("INERF_PRECISION", request.param.split("-")[0])
    if request.param.endswith("-1wg"):
        monkeypatch.setenv("INERF_F16_KERNEL", "single")
    else:
        monkeypatch.delenv("INERF_F16_KERNEL", raising=False)
    return request.param


def chair_rays(n=N_RAYS, side=800):
    """Every (side*side // n)-th ray of the benchmark's 800x800 chair frame (bench.py), unfiltered."""
    import bench
    from intrinsicnerf_amd import object_level as ol
    focal = 0.5 * side / np.tan(0.5 * bench.CAMERA_ANGLE_X)
    K = np.array([[focal, 0, 0.5 * side], [0, focal, 0.5 * side], [0, 0, 1]])
    ro, rd = ol.get_rays(side, side, K, bench.chair_pose())
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    sel = torch.arange(0, side * side, side * side // n + 1)[:n]           # odd stride: walks over rows and columns
    ro, rd = ro[sel], rd[sel]
    return torch.cat([ro, rd, 2.0 * torch.ones_like(rd[:, :1]), 6.0 * torch.ones_like(rd[:, :1]),
                      rd / rd.norm(dim=-1, keepdim=True)], -1).contiguous()


def room_rays(n=N_RAYS):
    """Every k-th ray of the 320x240 Replica-like frame of scripts/bench_ssr_frame.py, unfiltered."""
    from intrinsicnerf_amd import ssr
    H, W = 240, 320
    fx = fy = W / 2.0 / np.tan(np.deg2rad(45.0))
    rays = ssr.create_rays(1, torch.eye(4)[None], H, W, fx, fy, (W - 1) / 2.0, (H - 1) / 2.0, 0.1, 10.0).reshape(-1, 11)
    return rays[torch.arange(0, H * W, H * W // n + 1)[:n]].contiguous()


@functools.lru_cache(maxsize=None)
def workload(variant):
    """(rays, cfg, sd_c, sd_f, oracle fp32 outputs, oracle fp64 outputs) - computed once per session."""
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    if variant == "object":
        rays, c = chair_rays(), 0
        cfg = oracle.RenderConfig(variant="object", white_bkgd=True)
    else:
        rays, c = room_rays(), 28
        cfg = oracle.RenderConfig(variant="ssr", white_bkgd=False, n_classes=c, netchunk=32768)
    sd_c = cal.calibrated_default_init(variant, c, 0, rays)
    sd_f = cal.calibrated_default_init(variant, c, 1, rays)
    to64 = lambda sd: {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        o32 = oracle.render_rays(rays, sd_c, sd_f, cfg, stages=True)
        o64 = oracle.render_rays(rays.double(), to64(sd_c), to64(sd_f), cfg, stages=True)
    return rays, cfg, sd_c, sd_f, o32, o64


@functools.lru_cache(maxsize=None)
def _hazard(variant, subset_bytes):
    rays, cfg, sd_c, sd_f, o32, o64 = workload(variant)
    return cal.fine_pass_hazard(rays, sd_f, cfg, o32, o64, subset=np.frombuffer(subset_bytes, bool))


def hazard(variant, subset):
    """oracle.calibration.fine_pass_hazard on the rays that passed the fp32-vs-fp64 score (cached per session)."""
    return _hazard(variant, np.ascontiguousarray(subset, bool).tobytes())


def _keys(out):
    maps = [f"{k}_{lvl}" for lvl in ("coarse", "fine") for k in cal.MAP_KEYS if f"{k}_{lvl}" in out]
    return maps + ["z_std"]


def _tol(key):
    return RTOL_DISP if key.startswith("disp") else RTOL


@pytest.mark.parametrize("variant", ["object", "ssr"])
def test_unfiltered_rays_rank_statistics(variant, precision):
    from intrinsicnerf_amd import _capi, kernels, packing
    rays, cfg, sd_c, sd_f, o32, o64 = workload(variant)
    acc = o32["acc_fine"].numpy()
    assert acc.min() < 0.9 and (acc > 0.999).mean() > 0.02 and (acc > 0.999).mean() < 0.98, "workload is degenerate"
    dev = torch.device("cuda:0")
    ssr = variant == "ssr"
    desc = _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, cfg.n_classes if ssr else 0, 10, 4, cfg.xyz_div)
    got = kernels.render_rays_fused(desc, packing.pack_state_dict(desc, sd_c).to(dev), packing.pack_state_dict(desc, sd_f).to(dev),
                                    rays.to(dev), 64, 128, torch.linspace(0., 1., 64).to(dev), torch.linspace(0., 1., 128).to(dev),
                                    white_bkgd=cfg.white_bkgd, want_stages=True)
    kernels.check_f16_range(got.pop("status", None), "test")
    keys = _keys(o32)
    e_ref = {k: cal.scaled_errors(o32[k].numpy(), o64[k].numpy(), _tol(k), ATOL) for k in keys}
    e_hip = {k: cal.scaled_errors(got[k].cpu().numpy(), o32[k].numpy(), _tol(k), ATOL) for k in keys}
    # the reference arithmetic's own reproducibility per ray: max over maps AND the stage tensors behind them
    stage = ("z_samples", "weights_coarse", "weights_fine", "z_fine")
    score = np.maximum.reduce([e_ref[k] for k in keys] + [cal.scaled_errors(o32[k].numpy(), o64[k].numpy(), RTOL, ATOL) for k in stage])
    score = np.maximum(score, hazard(variant, score <= 0.2))
    well = score <= 0.2
    problems = []
    for k in keys:
        for v in cal.rank_report(e_hip[k], e_ref[k]):
            problems.append(f"{k}: {v}   [hip: {cal.summarize(e_hip[k])} | ref: {cal.summarize(e_ref[k])}]")
        strict = np.ones_like(well) if k.endswith("_coarse") else well
        worst = float(np.max(e_hip[k][strict], initial=0.0))
        if worst > 1.0:
            problems.append(f"{k}: {int((e_hip[k][strict] > 1).sum())} of {int(strict.sum())} reproducible rays beyond the "
                            f"plain tolerance (worst {worst:.3g})")
    assert not problems, f"{variant}/{precision}: {len(problems)} violations\n" + "\n".join(problems)
    # the run must have had something to say.  (On the chair frame almost no ray of a white-spectrum network is
    # reproducible once the 128 resampled depths count - the rank statistics and the coarse maps carry that case;
    # about half of the room frame's rays are.)
    if variant == "ssr":
        assert well.sum() >= 50, f"only {int(well.sum())} reproducible rays"
    print(f"\n[{variant}/{precision}] {len(rays)} unfiltered rays, {int(well.sum())} reproducible (score <= 0.2)")
    for k in keys:
        print(f"  {k:16s} hip-vs-fp32: {cal.summarize(e_hip[k])}\n  {'':16s} fp32-vs-fp64: {cal.summarize(e_ref[k])}")


@pytest.mark.parametrize("name", golden_names("uncurated_") + golden_names("trained_"))
def test_uncurated_reference_fixtures(name, precision):
    """The same judgement against outputs of the REAL reference (tests/golden/make_golden_uncurated.py): default-init
    networks, every k-th ray of the benchmark frames, nothing selected.  ``uncurated_object_coarse_only_wb`` is BASELINE
    configs[1] in miniature (64 coarse samples, coarse network only, white background): every ray at the plain tolerance."""
    from intrinsicnerf_amd import _capi, kernels, packing
    fx = load_golden(name)
    cfg = uncurated_config(fx)
    sd_c, sd_f = uncurated_weights(fx)
    dev = torch.device("cuda:0")
    ssr = cfg.variant == "ssr"
    desc = _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, cfg.n_classes if ssr else 0, 10, 4, cfg.xyz_div)
    ni = cfg.n_importance
    got = kernels.render_rays_fused(desc, packing.pack_state_dict(desc, sd_c).to(dev),
                                    packing.pack_state_dict(desc, sd_f).to(dev) if ni > 0 else None,
                                    torch.from_numpy(fx["rays"]).to(dev), 64, ni, torch.linspace(0., 1., 64).to(dev),
                                    torch.linspace(0., 1., ni).to(dev) if ni > 0 else None, white_bkgd=cfg.white_bkgd)
    kernels.check_f16_range(got.pop("status", None), "test")
    problems, summary = uncurated_judge(fx, {k: v.cpu().numpy() for k, v in got.items()}, f"{name}/{precision}")
    print("\n" + summary)
    assert not problems, "\n".join(problems) + "\n" + summary


@pytest.mark.parametrize("name", golden_names("trained_"))
def test_trained_network_most_rays_at_the_plain_tolerance_end_to_end(name, precision):
    """on a TRAINED network the two-pass path is well-conditioned on most rays, so the plain tolerance
    (1e-5 + 1e-4 |want|, disp 5e-4) can bite END TO END, per ray, on the final maps - all 4 096 rays of the held-out view, the
    REAL reference's fp32 outputs as `want`, nothing filtered by a conditioning score: >= 90 % of the rays must meet it on EVERY
    fine and coarse map at once (the reference's own fp32-vs-fp64 distance on the same rays is printed next to it; the rays
    beyond are judged by the rank statistics of test_uncurated_reference_fixtures)."""
    from intrinsicnerf_amd import _capi, kernels, packing
    fx = load_golden(name)
    cfg = uncurated_config(fx)
    sd_c, sd_f = uncurated_weights(fx)
    dev = torch.device("cuda:0")
    desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, cfg.xyz_div)
    ni = cfg.n_importance
    got = kernels.render_rays_fused(desc, packing.pack_state_dict(desc, sd_c).to(dev), packing.pack_state_dict(desc, sd_f).to(dev),
                                    torch.from_numpy(fx["rays"]).to(dev), 64, ni, torch.linspace(0., 1., 64).to(dev),
                                    torch.linspace(0., 1., ni).to(dev), white_bkgd=cfg.white_bkgd)
    kernels.check_f16_range(got.pop("status", None), "test")
    keys = [k[4:] for k in fx if k.startswith("ref_")]
    assert len(fx["rays"]) >= 4096 and {"rgb_fine", "albedo_fine", "shading_fine", "residual_fine", "acc_fine", "disp_fine"} <= set(keys)
    tol = lambda k: RTOL_DISP if k.startswith("disp") else RTOL
    e_hip = np.maximum.reduce([cal.scaled_errors(got[k].cpu().numpy(), fx["ref_" + k], tol(k), ATOL) for k in keys])
    e_ref = np.maximum.reduce([cal.scaled_errors(fx["ref_" + k], fx["f64_" + k], tol(k), ATOL) for k in keys])
    frac_hip, frac_ref = float((e_hip <= 1.0).mean()), float((e_ref <= 1.0).mean())
    print(f"\n[{name}/{precision}] {len(e_hip)} rays end to end, every map: HIP vs reference within the plain tolerance on {100 * frac_hip:.2f} % "
          f"(median {np.median(e_hip):.3g}, q90 {np.quantile(e_hip, 0.9):.3g} tolerances); the reference's fp32 vs its fp64: {100 * frac_ref:.2f} % "
          f"(median {np.median(e_ref):.3g}, q90 {np.quantile(e_ref, 0.9):.3g})")
    assert frac_hip >= 0.90, f"only {100 * frac_hip:.1f} % of the trained network's rays meet the plain tolerance end to end"


def _stage_reference(fx):
    """The reference's own stage tensors and maps of an ``uncurated_*`` fixture under the oracle's key names."""
    ref = {k[len("stage_"):]: fx[k] for k in fx if k.startswith("stage_") and not k.startswith("stage_score_")}
    ref.update({k[len("ref_"):]: fx[k] for k in fx if k.startswith("ref_")})
    return ref


@pytest.mark.parametrize("name", golden_names("uncurated_") + golden_names("trained_"))
def test_uncurated_stagewise_strict(name, precision):
    """the PLAIN 1e-4 tolerance on EVERY ray of the un-curated fixtures (default-init networks, unfiltered
    rays of the benchmark frames), stage by stage against tensors recorded from the REAL reference while it ran
    (make_golden_uncurated.py: captured_stages): each HIP stage gets the reference's input for that stage
    (oracle/stagewise.py) - the reference's z_coarse / z_fine go through inerf_encode_mlp + inerf_composite and must
    reproduce raw, the compositing weights and every map; the reference's weights_coarse go through inerf_sample_fine and
    must reproduce z_samples / z_fine / z_std up to what two correct fp32 sample_pdf evaluations can differ by."""
    from intrinsicnerf_amd import _capi, packing
    from oracle import stagewise
    fx = load_golden(name)
    cfg = uncurated_config(fx)
    sd_c, sd_f = uncurated_weights(fx)
    dev = torch.device("cuda:0")
    ssr = cfg.variant == "ssr"
    desc = _capi.net_desc(_capi.VARIANT_SSR if ssr else _capi.VARIANT_OBJECT, cfg.n_classes if ssr else 0, 10, 4, cfg.xyz_div)
    ref = _stage_reference(fx)
    rows = fx["stage_raw_rows"]
    got = stagewise.hip_stages(desc, packing.pack_state_dict(desc, sd_c).to(dev),
                               packing.pack_state_dict(desc, sd_f).to(dev) if cfg.n_importance > 0 else None,
                               torch.from_numpy(fx["rays"]).to(dev), ref, cfg.white_bkgd, cfg.n_classes if ssr else 0, raw_rows=rows)
    per, problems = stagewise.strict_report(got, ref, raw_rows=rows)
    print(f"\n[{name}/{precision}] " + "  ".join(f"{k}:{v['worst']:.2g}" for k, v in per.items()))
    expected = {"z_coarse", "raw_coarse", "weights_coarse", "rgb_coarse", "acc_coarse", "albedo_coarse"}
    if cfg.n_importance > 0:
        expected |= {"z_samples", "z_fine", "z_std", "raw_fine", "weights_fine", "rgb_fine", "disp_fine", "acc_fine", "albedo_fine",
                     "shading_fine", "residual_fine"}
    if ssr:
        expected |= {"sem_coarse", "sem_fine", "depth_fine"}
    assert expected <= set(per), f"stages not judged: {sorted(expected - set(per))}"
    assert all(v["rays"] == (len(rows) if k.startswith("raw_") else len(fx["rays"])) for k, v in per.items())
    assert not problems, f"{name}/{precision}:\n" + "\n".join(problems)
