"""Numerical conditioning of the render path on given rays / weights (ORACLE - test infrastructure).

The two-pass path is not uniformly well-conditioned: sample_pdf divides by cdf differences as small
as 1e-5 (near-empty bins), the 2^9 frequency band turns a 1e-6 depth change into a visible phase
change and the 1e10 last interval makes alpha a step function of sigma.  On an ill-conditioned ray
the reference's own fp32 result differs from the same arithmetic in fp64 by far more than 1e-4, so
that ray cannot pin ANY fp32 implementation to 1e-4.  ``conditioning_scores`` measures this per
ray, as the fp32-vs-fp64 distance of the oracle in units of the parity tolerance; tests compare the
HIP path against the oracle on rays whose score is well below 1.
"""
import torch

from .intrinsic_render import RenderConfig, lcg_state_dict, render_rays

MAP_KEYS = ("rgb", "acc", "depth", "albedo", "shading", "residual", "sem", "feat")
STAGE_KEYS = ("z_samples", "z_std", "weights_coarse", "weights_fine", "z_fine")


def conditioning_scores(rays, sd_c, sd_f, cfg, t_vals, extra=None, rtol=1e-4, atol=1e-5, stage_keys=STAGE_KEYS):
    """Per-ray max over maps / stage tensors of ``|fp32 - fp64| / (atol + rtol |fp64|)``; inf on a NaN mismatch."""
    extra = extra or {}
    to64 = lambda sd: None if sd is None else {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        a = render_rays(rays, sd_c, sd_f, cfg, t_vals=t_vals, stages=True, **extra)
        b = render_rays(rays.double(), to64(sd_c), to64(sd_f), cfg, t_vals=t_vals.double(), stages=True,
                        **{k: v.double() for k, v in extra.items()})
    n = rays.shape[0]
    score = torch.zeros(n, dtype=torch.float64)
    keys = [f"{k}_{lvl}" for k in MAP_KEYS for lvl in ("coarse", "fine")] + list(stage_keys)
    for k in keys:
        if k not in a:
            continue
        x, y = a[k].double().reshape(n, -1), b[k].reshape(n, -1)
        mismatch = (torch.isnan(x) != torch.isnan(y)).any(dim=1)
        e = (x - y).abs() / (atol + rtol * y.abs())
        e = torch.where(torch.isnan(e), torch.zeros_like(e), e)
        score = torch.maximum(score, e.max(dim=1)[0])
        score = torch.where(mismatch, torch.full_like(score, float("inf")), score)
    return score


def calibrated_lcg_weights(variant, n_classes, seed, rays, sigma_gain_log2=5, quantile=0.7, weight_gain_log2=0,
                           netchunk=1 << 16):
    """Closed-form 1/f-spectrum weights whose density straddles zero on ``rays``; returns ``(state_dict, bias)``.

    ``sigma_bias`` := -(the ``quantile`` of the un-biased coarse-sample densities on these rays), rounded
    to 1/16 so that it stays a dyadic rational - seed + bias are all a fixture has to store.
    ``quantile=None`` keeps the bias at 0; ``quantile > 1`` pushes every density below zero (empty space).
    """
    wp = dict(sigma_gain_log2=sigma_gain_log2, weight_gain_log2=weight_gain_log2, freq_decay=True)
    sd0 = lcg_state_dict(variant, n_classes, seed=seed, sigma_bias=0.0, **wp)
    if quantile is None:
        return sd0, 0.0
    with torch.no_grad():
        probe = render_rays(rays, sd0, None, RenderConfig(variant=variant, n_samples=64, n_importance=0,
                                                          n_classes=n_classes, netchunk=netchunk), stages=True)
    b = -float(torch.quantile(probe["raw_coarse"][..., 3].flatten(), min(quantile, 1.0)))
    b = round(b * 16.0) / 16.0 - (1.0 if quantile > 1.0 else 0.0)
    return lcg_state_dict(variant, n_classes, seed=seed, sigma_bias=b, **wp), b
