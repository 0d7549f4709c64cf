"""Shared helpers that rebuild a golden case's inputs (weights from seed, config, random tensors)."""
import numpy as np
import torch

import oracle


def case_config(fx):
    variant = str(fx["variant"])
    return oracle.RenderConfig(
        variant=variant, n_samples=64, n_importance=int(fx["n_importance"]),
        white_bkgd=bool(fx["white_bkgd"]), lindisp=bool(fx["lindisp"]),
        n_classes=int(fx["n_classes"]), endpoint_feat=bool(fx["endpoint_feat"]),
        netchunk=32768 if variant == "ssr" else 65536)


def case_weights(fx):
    variant, c, seed = str(fx["variant"]), int(fx["n_classes"]), int(fx["seed"])
    kw = dict(sigma_gain_log2=int(fx["sigma_gain_log2"]), weight_gain_log2=int(fx["weight_gain_log2"]), freq_decay=True)
    sd_c = oracle.lcg_state_dict(variant, c, seed=2 * seed, sigma_bias=float(fx["sigma_bias_coarse"]), **kw)
    sd_f = oracle.lcg_state_dict(variant, c, seed=2 * seed + 1, sigma_bias=float(fx["sigma_bias_fine"]), **kw)
    return sd_c, sd_f


def case_random_inputs(fx, device="cpu"):
    out = {}
    for k in ("t_rand", "noise_coarse", "noise_fine", "u"):
        if "in_" + k in fx:
            out[k] = torch.from_numpy(fx["in_" + k]).to(device)
    return out


# oracle key -> tolerance class.  ``disp`` is ill-conditioned (1/(depth/acc)); the reference's own
# fp32-vs-fp64 noise floor for it is 8.9e-5, so it gets a looser bound.
def sample_pdf_sensitivity(z_coarse, weights_coarse, u):
    """|d z_sample / d cdf| of every importance sample: (bin width) / (cdf difference of its bin).

    sample_pdf interpolates ``bins_lo + (u - cdf_lo) / denom * (bins_hi - bins_lo)`` with ``denom`` as small as
    1e-5 (run_nerf_helpers.py:440-443), so a cdf that differs by one fp32 rounding (~1e-7; it is a 62-term
    cumulative sum of normalised weights) legitimately moves a sample in a nearly-empty bin by up to
    ``0.06 / 1e-5 * 1e-7``.  Returns the factor per sample, [N, n_importance]."""
    z = np.asarray(z_coarse, np.float64)
    w = np.asarray(weights_coarse, np.float64)[:, 1:-1] + 1e-5
    u = np.broadcast_to(np.asarray(u, np.float64), (z.shape[0], np.asarray(u).shape[-1]))
    bins = 0.5 * (z[:, 1:] + z[:, :-1])
    cdf = np.concatenate([np.zeros((z.shape[0], 1)), np.cumsum(w / w.sum(-1, keepdims=True), -1)], -1)
    sens = np.zeros_like(u)
    for r in range(z.shape[0]):
        idx = np.searchsorted(cdf[r], u[r], side="right")
        lo, hi = np.clip(idx - 1, 0, None), np.clip(idx, None, cdf.shape[1] - 1)
        denom = cdf[r, hi] - cdf[r, lo]
        denom = np.where(denom < 1e-5, 1.0, denom)
        sens[r] = np.abs(bins[r, hi] - bins[r, lo]) / denom
    return sens


# Allowed difference of a cdf entry between two fp32 evaluations: 1e-5.  The compositing weights behind the cdf are
# themselves pinned to 1e-4 relative (the parity tolerance), which would permit cdf differences of ~1e-4; the
# tests grant a tenth of that.  Observed: up to 4e-6 (round-off of the 62-term cumulative sum is ~3e-7, the rest is
# the ~1e-6 noise of the densities behind the weights).
CDF_NOISE = 1e-5


def assert_maps_close(got, want, rtol, atol, tag="", extra=None):
    """``|got - want| <= atol + rtol * |want| (+ extra)`` with NaNs required at identical positions."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{tag}: shape {got.shape} vs {want.shape}"
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), f"{tag}: NaN pattern differs ({nan_g.sum()} vs {nan_w.sum()})"
    ok = ~nan_w
    err = np.abs(got[ok] - want[ok])
    bound = atol + rtol * np.abs(want[ok])
    if extra is not None:
        bound = bound + np.broadcast_to(np.asarray(extra, np.float64), want.shape)[ok]
    if err.size and not np.all(err <= bound):
        i = int(np.argmax(err - bound))
        raise AssertionError(f"{tag}: max violation err={err[i]:.3e} bound={bound[i]:.3e} "
                             f"(want {want[ok][i]:.6g}, got {got[ok][i]:.6g}); max err {err.max():.3e}")


class injected_np_rand:
    """Feed the fixture's random tensors to ``np.random.rand`` calls, in call order.

    The object-level front-end keeps the reference's ``pytest=True`` hooks (np.random.seed(0) followed by
    np.random.rand, run_nerf.py:389-393,480-484; run_nerf_helpers.py:416-425).  The golden generator
    ran the reference with these same tensors injected the same way (tests/golden/make_golden.py).
    """

    def __init__(self, tensors):
        self.q = list(tensors)

    def __enter__(self):
        self.saved = np.random.rand
        def fake(*shape):
            t = self.q.pop(0)
            assert tuple(t.shape) == tuple(shape), (tuple(t.shape), shape)
            return t.double().cpu().numpy()
        np.random.rand = fake
        return self

    def __exit__(self, *exc):
        np.random.rand = self.saved
        assert exc[0] is not None or not self.q, "front-end made fewer RNG draws than the reference"


# ------------------------------------------------------------------------------------------------
# un-curated fixtures (tests/golden/make_golden_uncurated.py): default-init networks, unfiltered rays
# ------------------------------------------------------------------------------------------------
def uncurated_weights(fx):
    """The two default-init state dicts of an ``uncurated_*`` fixture: seeded ``make_state_dict`` plus the stored
    calibration of the density head (a power-of-two gain on alpha_linear.weight, the bias as stored)."""
    variant, c = str(fx["variant"]), int(fx["n_classes"])
    if any(k.startswith("w_coarse/") for k in fx):       # trained weights, stored in the fixture (make_golden_trained.py)
        return [{k.split("/", 1)[1]: torch.from_numpy(np.array(fx[k])) for k in fx if k.startswith(f"w_{lvl}/")} for lvl in ("coarse", "fine")]
    out = []
    for lvl in ("coarse", "fine"):
        sd = oracle.make_state_dict(variant, c, seed=int(fx["seed_" + lvl]))
        sd["alpha_linear.weight"] = sd["alpha_linear.weight"] * float(fx["alpha_gain_" + lvl])
        sd["alpha_linear.bias"] = torch.full_like(sd["alpha_linear.bias"], float(fx["alpha_bias_" + lvl]))
        out.append(sd)
    return out


def uncurated_config(fx):
    variant = str(fx["variant"])
    return oracle.RenderConfig(variant=variant, n_samples=64, n_importance=int(fx["n_importance"]), white_bkgd=bool(fx["white_bkgd"]),
                               n_classes=int(fx["n_classes"]), netchunk=32768 if variant == "ssr" else 65536)


def uncurated_judge(fx, got, tag, rtol=1e-4, atol=1e-5, rtol_disp=5e-4):
    """Judge ``got`` (dict of arrays under the oracle's key names) against an un-curated fixture: every output map on
    EVERY ray by rank statistics against the reference's own fp32-vs-fp64 distance, and the plain tolerance on every ray
    the reference arithmetic reproduces (score over maps and stage tensors <= 0.2; for coarse maps: every ray).
    Returns (list of violations, one-line-per-map summary)."""
    from oracle import calibration as cal
    keys = [k[4:] for k in fx if k.startswith("ref_")]
    tol = lambda k: rtol_disp if k.startswith("disp") else rtol
    e_ref = {k: cal.scaled_errors(fx["ref_" + k], fx["f64_" + k], tol(k), atol) for k in keys}
    score = np.maximum.reduce(list(e_ref.values()) + [fx[k] for k in fx if k.startswith("stage_score_")])
    well = score <= 0.2
    problems, lines = [], [f"{tag}: {len(well)} unfiltered rays, {int(well.sum())} reproducible"]
    for k in keys:
        e = cal.scaled_errors(np.asarray(got[k]), fx["ref_" + k], tol(k), atol)
        problems += [f"{tag}:{k}: {v}" for v in cal.rank_report(e, e_ref[k])]
        strict = np.ones_like(well) if k.endswith("_coarse") else well
        worst = float(np.max(e[strict], initial=0.0))
        if worst > 1.0:
            problems.append(f"{tag}:{k}: {int((e[strict] > 1).sum())} of {int(strict.sum())} reproducible rays beyond the plain "
                            f"tolerance (worst {worst:.3g})")
        lines.append(f"  {k:16s} vs reference: {cal.summarize(e)} | reference vs fp64: {cal.summarize(e_ref[k])}")
    return problems, "\n".join(lines)
