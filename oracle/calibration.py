"""Non-degenerate default-init networks and un-curated parity statistics (ORACLE - test infrastructure).

``nn.Linear``'s default init (``make_state_dict``) gives a density head whose output is almost constant in
space and negative on most of the chair volume: with seeds 0/1 every sample has sigma < 0, acc == 0 and every
map is exactly the background - the MLP work is the same, but neither ``sample_pdf`` nor compositing see a
non-trivial input.  ``calibrated_default_init`` keeps every default-init tensor and only rescales / shifts
``alpha_linear`` (a power-of-two gain, a dyadic bias) so that the density straddles zero on the given rays and
acc spans (0, 1].  Nothing else is conditioned: the network keeps the white spectrum of a random one, so many rays
are ill-conditioned for ANY fp32 evaluation (the reference's included).  ``rank_report`` is how tests judge an
implementation on such rays without discarding any: its error distribution against the fp32 oracle is compared,
quantile by quantile and tail count by tail count, with the fp32 oracle's own distance from the same arithmetic
in fp64.
"""
import numpy as np
import torch

from .intrinsic_render import RenderConfig, make_state_dict, render_rays

MAP_KEYS = ("rgb", "acc", "depth", "disp", "albedo", "shading", "residual", "sem", "feat")


def calibrated_default_init(variant, n_classes, seed, rays, gain=4.0, quantile=0.7, n_probe=128):
    """Default-``nn.Linear``-init state dict (``make_state_dict(variant, n_classes, seed)``) whose ``alpha_linear`` is
    rescaled so that sigma = g * (s - q): ``s`` the un-calibrated density, ``q`` its ``quantile`` over the coarse samples
    of (a strided subset of) ``rays`` and ``g`` the power of two nearest to ``gain / std(s)``.  The bias is rounded to
    2^-10, so seed + rays determine the result up to razor-edge rounding of the probe."""
    sd = make_state_dict(variant, n_classes, seed=seed)
    step = max(1, rays.shape[0] // n_probe)
    probe_cfg = RenderConfig(variant=variant, n_samples=64, n_importance=0, n_classes=n_classes,
                             netchunk=32768 if variant == "ssr" else 65536)
    with torch.no_grad():
        s = render_rays(rays[::step][:n_probe].float(), sd, None, probe_cfg, stages=True)["raw_coarse"][..., 3].flatten()
    q = float(torch.quantile(s, quantile))
    g = 2.0 ** round(float(np.log2(gain / max(float(s.std()), 1e-12))))
    sd["alpha_linear.weight"] = sd["alpha_linear.weight"] * g
    sd["alpha_linear.bias"] = torch.round((sd["alpha_linear.bias"] - q) * g * 1024.0) / 1024.0
    return sd


def scaled_errors(got, want, rtol=1e-4, atol=1e-5):
    """Per-ray max over the trailing axes of ``|got - want| / (atol + rtol |want|)``; inf where the NaN patterns differ."""
    g = np.asarray(got, np.float64).reshape(len(got), -1)
    w = np.asarray(want, np.float64).reshape(len(want), -1)
    bad = (np.isnan(g) != np.isnan(w)).any(1)
    e = np.abs(g - w) / (atol + rtol * np.abs(w))
    e = np.where(np.isnan(e), 0.0, e).max(1) if e.shape[1] else np.zeros(len(g))
    return np.where(bad, np.inf, e)


QUANTILES = (0.5, 0.75, 0.9, 0.95, 0.99)
TAILS = (1.0, 10.0, 100.0)


def rank_report(e_impl, e_ref, factor=3.0, floor=0.5, slack=3):
    """Compare two per-ray error samples by rank statistics.  Returns a list of violation strings (empty = pass).

    * every listed quantile of ``e_impl`` must be <= max(floor, factor * the same quantile of ``e_ref``);
    * for every tail threshold T, #(e_impl > T) <= factor * #(e_ref > T) + slack.
    ``floor`` (in units of the tolerance) is the implementation's own allowance where the reference arithmetic is
    essentially exact; ``slack`` absorbs Poisson noise of rare flips (a density crossing zero at the 1e10 interval)."""
    out = []
    fin_i = np.where(np.isfinite(e_impl), e_impl, 1e30)
    fin_r = np.where(np.isfinite(e_ref), e_ref, 1e30)
    for q in QUANTILES:
        a, b = float(np.quantile(fin_i, q)), float(np.quantile(fin_r, q))
        if a > max(floor, factor * b):
            out.append(f"q{int(q * 100)}: {a:.3g} > max({floor}, {factor} x {b:.3g})")
    for t in TAILS:
        a, b = int((fin_i > t).sum()), int((fin_r > t).sum())
        if a > factor * b + slack:
            out.append(f"#(err > {t:g} tol): {a} > {factor} x {b} + {slack}")
    return out


def summarize(e):
    fin = np.where(np.isfinite(e), e, 1e30)
    qs = " ".join(f"q{int(q * 100)} {float(np.quantile(fin, q)):.3g}" for q in QUANTILES)
    return f"{qs} max {float(fin.max()):.3g} #>1 {int((fin > 1).sum())}/{len(fin)}"


def resampling_hazard(z_coarse, weights64, weights32, u=None, n_importance=128, trials=4, rtol=1e-4, atol=1e-5, seed=0):
    """Per-ray stability of ``sample_pdf`` under weight noise of the size fp32 arithmetic actually produces on that ray.

    ``sample_pdf`` is discontinuous: a bin whose cdf difference falls below 1e-5 switches from ``t = (u - cdf_lo) / denom``
    to ``t = u - cdf_lo`` (run_nerf_helpers.py:440-443) - and on an opaque ray the empty bins' pdf, 1e-5 / (sum(w) + 62e-5),
    sits right AT that threshold - so a 1e-7 change of the coarse weights can move an importance sample by a whole bin.
    One fp32-vs-fp64 comparison samples that hazard once; this probes it several more times: the fp64 weights are
    perturbed by +-2 x |fp32 - fp64| (uniformly up, uniformly down, and ``trials`` random sign patterns) and the largest
    move of any importance sample, in tolerances, is returned per ray.  Rays with a large value cannot pin ANY fp32
    implementation's fine pass to the plain tolerance; tests leave them to the rank statistics."""
    from .intrinsic_render import inverse_cdf_sample
    z = torch.as_tensor(z_coarse).double()
    w64 = torch.as_tensor(weights64).double()[:, 1:-1]
    delta = 2.0 * (torch.as_tensor(weights32).double()[:, 1:-1] - w64).abs() + 1e-9
    n = z.shape[0]
    if u is None:
        u = torch.linspace(0.0, 1.0, n_importance, dtype=torch.float64)
    u = torch.as_tensor(u).double()
    if u.dim() == 1:
        u = u.expand(n, u.shape[0])
    bins = 0.5 * (z[:, 1:] + z[:, :-1])
    base = inverse_cdf_sample(bins, w64, u)
    g = torch.Generator().manual_seed(seed)
    signs = [torch.ones_like(w64), -torch.ones_like(w64)]
    signs += [torch.randint(0, 2, w64.shape, generator=g).double() * 2 - 1 for _ in range(trials)]
    worst = torch.zeros(n, dtype=torch.float64)
    for s in signs:
        moved = inverse_cdf_sample(bins, (w64 + s * delta).clamp_min(0.0), u)
        worst = torch.maximum(worst, ((moved - base).abs() / (atol + rtol * base.abs())).amax(1))
    return worst.numpy()
