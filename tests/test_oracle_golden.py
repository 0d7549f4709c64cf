"""CPU: the oracle restatement replays every golden fixture (= outputs of the real reference).

This is BASELINE config[0] ("PyTorch-CPU, 1 ray-batch, plumbing") in test form: the reference's own
outputs for small ray batches, reproduced by the oracle on whatever CPU runs the suite.  On the CPU
that generated the fixtures the match is bit-exact; another CPU may take different GEMM blocking,
hence the small tolerance (rtol 2e-5, atol 2e-6; disp 2e-4 - see tests/_cases.py).
"""
import os

import numpy as np
import pytest
import torch

import oracle
from _cases import assert_maps_close, case_config, case_random_inputs, case_weights
from conftest import golden_names, load_golden

RTOL, ATOL = 2e-5, 2e-6


@pytest.mark.parametrize("name", golden_names("object_") + golden_names("ssr_"))
def test_render_rays_matches_reference(name, torch_threads):
    fx = load_golden(name)
    cfg = case_config(fx)
    sd_c, sd_f = case_weights(fx)
    rays = torch.from_numpy(fx["rays"])
    with torch.no_grad():
        out = oracle.render_rays(rays, sd_c, sd_f, cfg, t_vals=torch.from_numpy(fx["t_vals"]), stages=True,
                                 **case_random_inputs(fx))
    checked = 0
    for key, want in fx.items():
        if not key.startswith("ref_"):
            continue
        k = key[4:]
        got = out[k].numpy()
        if k.startswith("raw"):
            got = got[: want.shape[0]]
        rtol = 2e-4 if k.startswith("disp") else RTOL
        assert_maps_close(got, want, rtol, ATOL, f"{name}:{k}")
        checked += 1
    assert checked >= 7
    for k in ("z_coarse", "weights_coarse", "z_samples", "z_fine", "weights_fine"):
        if "stage_" + k in fx:
            assert_maps_close(out[k].numpy(), fx["stage_" + k], RTOL, ATOL, f"{name}:stage {k}")


@pytest.mark.parametrize("name", golden_names("stage_composite_"))
def test_composite_edge_cases(name):
    fx = load_golden(name)
    ssr = "ssr" in name
    cfg = oracle.RenderConfig(variant="ssr" if ssr else "object", white_bkgd=bool(fx["white_bkgd"]),
                              n_classes=int(fx["n_classes"]) if ssr else 0, endpoint_feat=ssr)
    out = oracle.composite(torch.from_numpy(fx["raw"]), torch.from_numpy(fx["z"]), torch.from_numpy(fx["rays_d"]),
                           cfg, feat=ssr)
    for key, want in fx.items():
        if key.startswith("ref_"):
            assert_maps_close(out[key[4:]].numpy(), want, 2e-6, 1e-7, f"{name}:{key}")
    if not ssr:   # crafted rays: 0 = empty (NaN disp), 1 = only the 1e10 interval opaque, 3 = opaque at sample 0
        assert np.isnan(out["disp"][0].item()) and out["acc"][0].item() == 0.0
        assert out["weights"][1, -1].item() == 1.0 and out["acc"][1].item() == 1.0
        assert out["weights"][3, 0].item() == 1.0


def test_sample_pdf_edge_cases():
    fx = load_golden("stage_sample_pdf")
    bins, w = torch.from_numpy(fx["bins"]), torch.from_numpy(fx["weights"])
    n = bins.shape[0]
    det = oracle.inverse_cdf_sample(bins, w, torch.linspace(0.0, 1.0, 128).expand(n, 128))
    assert_maps_close(det.numpy(), fx["ref_det"], 2e-6, 1e-7, "sample_pdf det")
    rnd = oracle.inverse_cdf_sample(bins, w, torch.from_numpy(fx["u_rnd"]))
    assert_maps_close(rnd.numpy(), fx["ref_rnd"], 2e-6, 1e-7, "sample_pdf rnd")
    # samples stay inside the bin range
    assert torch.all(det >= bins[:, :1] - 1e-6) and torch.all(det <= bins[:, -1:] + 1e-6)


def test_lcg_weights_are_exact_dyadics():
    sd = oracle.lcg_state_dict("ssr", 28, seed=3, sigma_gain_log2=3, sigma_bias=-2.5, weight_gain_log2=1)
    spec = oracle.state_dict_spec("ssr", 28)
    assert [k for k, _ in spec] == list(sd.keys())
    for (k, shape) in spec:
        assert tuple(sd[k].shape) == shape and sd[k].dtype == torch.float32
        v = sd[k].double() * 2.0 ** 24
        assert torch.equal(v, torch.round(v)), k       # integer multiples of 2**-24: exact in fp32
    n_params = sum(int(np.prod(s)) for _, s in spec)
    assert n_params == 698660                           # SURVEY.md 8a M2 @ C=28
    assert sum(int(np.prod(s)) for _, s in oracle.state_dict_spec("object")) == 662152   # M1


def test_oracle_equals_the_live_reference_on_random_configurations():
    """Where the reference is mounted (the build container), sweep random configurations the fixtures do not hold - ray /
    sample / importance counts, flags, class counts, default-initialised networks, the cluster lookup - and require
    oracle == reference on every returned tensor (tests/golden/check_reference_live.py; a subprocess, because the import
    recipe patches torch.Tensor.cuda).  The GPU box has no reference: skipped there."""
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/object_level"):
        pytest.skip("reference not mounted")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "check_reference_live.py")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, script, "--cases", "6", "--seed", "1"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "oracle == reference on 18 random configurations" in out.stdout


@pytest.mark.parametrize("name", golden_names("uncurated_"))
def test_uncurated_fixtures_replay(name, torch_threads):
    """Un-curated fixtures (default-init networks, unfiltered rays of the benchmark frames, outputs of the REAL reference -
    tests/golden/make_golden_uncurated.py): the oracle reproduces the reference (bit for bit on the CPU that wrote them;
    on another CPU within what the reference's own fp32-vs-fp64 distance allows, ray ranks compared)."""
    from _cases import uncurated_config, uncurated_judge, uncurated_weights
    fx = load_golden(name)
    sd_c, sd_f = uncurated_weights(fx)
    cfg = uncurated_config(fx)
    with torch.no_grad():
        out = oracle.render_rays(torch.from_numpy(fx["rays"]), sd_c, sd_f if cfg.n_importance > 0 else None, cfg)
    problems, summary = uncurated_judge(fx, {k: v.numpy() for k, v in out.items()}, name)
    assert not problems, "\n".join(problems) + "\n" + summary
    assert "ref_rgb_coarse" in fx and (cfg.n_importance == 0 or "ref_z_std" in fx)
    acc = fx["ref_acc_fine" if cfg.n_importance > 0 else "ref_acc_coarse"]
    assert acc.min() < 0.9 and acc.max() > 0.999              # not the all-background frame
