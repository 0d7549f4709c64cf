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


def _perturbed_samples(z_coarse, weights64, weights32, u, trials, seed, cdf_noise):
    """Importance samples (fp64) of ``sample_pdf`` on the fp64 coarse weights, unperturbed (first entry) and under
    ``trials + 2`` perturbations of the size fp32 arithmetic produces on each ray: weights moved by
    +-2 x max(|fp32 - fp64| of the element, rms of the ray's differences) (uniformly up, uniformly down, then random
    sign patterns), cdf entries by uniform noise of +-``cdf_noise`` (a 62-term fp32 cumulative sum rounds differently
    in every summation order; a few ulps of a value <= 1)."""
    z = torch.as_tensor(z_coarse).double()
    w64 = torch.as_tensor(weights64).double()[:, 1:-1]
    d = (torch.as_tensor(weights32).double()[:, 1:-1] - w64).abs()
    delta = 2.0 * torch.maximum(d, d.square().mean(1, keepdim=True).sqrt()) + 1e-9
    n = z.shape[0]
    u = torch.as_tensor(u).double()
    if u.dim() == 1:
        u = u.expand(n, u.shape[0])
    u = u.contiguous()
    bins = 0.5 * (z[:, 1:] + z[:, :-1])

    def sample(w, noise):
        w = w + 1e-5                                                   # run_nerf_helpers.py:404-409
        cdf = torch.cumsum(w / w.sum(-1, keepdim=True), -1)
        cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1) + noise
        idx = torch.searchsorted(cdf, u, right=True)
        lo, hi = (idx - 1).clamp_min(0), idx.clamp_max(cdf.shape[-1] - 1)
        denom = torch.gather(cdf, 1, hi) - torch.gather(cdf, 1, lo)
        denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
        t = (u - torch.gather(cdf, 1, lo)) / denom
        return torch.gather(bins, 1, lo) + t * (torch.gather(bins, 1, hi) - torch.gather(bins, 1, lo))

    zero = torch.zeros(n, w64.shape[1] + 1, dtype=torch.float64)
    out = [sample(w64, zero)]
    g = torch.Generator().manual_seed(seed)
    for trial in range(trials + 2):
        if trial < 2:
            s, noise = torch.full_like(w64, 1.0 - 2.0 * trial), zero
        else:
            s = torch.randint(0, 2, w64.shape, generator=g).double() * 2 - 1
            noise = (torch.rand(zero.shape, generator=g, dtype=torch.float64) * 2 - 1) * cdf_noise
        out.append(sample((w64 + s * delta).clamp_min(0.0), noise))
    return out


def fine_pass_hazard(rays, sd_fine, cfg, o32, o64, subset=None, trials=6, seed=0, cdf_noise=2e-7, rtol=1e-4, atol=1e-5,
                     rtol_disp=5e-4):
    """Per-ray conditioning of the FINE pass behind ``sample_pdf``: how far (in tolerances) the fine maps and ``z_std`` move
    when the importance samples are drawn from coarse weights / cdf entries perturbed by what fp32 arithmetic produces on
    that ray (``_perturbed_samples``).

    ``sample_pdf`` is discontinuous: a bin whose cdf difference falls below 1e-5 switches from ``t = (u - cdf_lo) / denom``
    to ``t = u - cdf_lo`` (run_nerf_helpers.py:440-443) - and on an opaque ray the empty bins' pdf, 1e-5 / (sum(w) + 62e-5),
    sits right AT that threshold - so a 1e-7 change of a coarse weight or of a cdf entry can move an importance sample
    by a whole bin, and ``1 / denom`` amplifies cdf round-off by up to 1e5.  Whether that matters depends on what the
    fine network returns where the sample lands, so the probe re-runs the fine pass (fp32, on the fp32 oracle's own
    coarse depths) for every perturbation and compares its maps with the unperturbed run's.  One fp32-vs-fp64 comparison
    samples this hazard once; the probe samples it ``trials + 2`` more times.  Rays with a large value cannot pin ANY
    fp32 implementation to the plain tolerance; tests leave those to the rank statistics.  ``subset`` (bool mask) limits
    the work to the rays of interest; the others get +inf."""
    from .intrinsic_render import composite, query_network
    n = rays.shape[0]
    mask = np.ones(n, bool) if subset is None else np.asarray(subset, bool)
    out = np.full(n, np.inf)
    if not mask.any():
        return out
    sel = torch.from_numpy(np.nonzero(mask)[0])
    r = rays[sel].float()
    z_c = torch.as_tensor(o32["z_coarse"])[sel].float()
    u = torch.linspace(0.0, 1.0, cfg.n_importance, dtype=torch.float64)
    draws = _perturbed_samples(torch.as_tensor(o64["z_coarse"])[sel], torch.as_tensor(o64["weights_coarse"])[sel],
                               torch.as_tensor(o32["weights_coarse"])[sel], u, trials, seed, cdf_noise)

    def fine(z_new):
        z_new = z_new.float()
        z_all, _ = torch.sort(torch.cat([z_c, z_new], -1), -1)
        with torch.no_grad():
            pts = r[:, None, 0:3] + r[:, None, 3:6] * z_all[:, :, None]
            c = composite(query_network(sd_fine, pts, r[:, 8:11], cfg), z_all, r[:, 3:6], cfg)
        c["z_std"] = torch.std(z_new, dim=-1, unbiased=False)
        return {k: v.numpy() for k, v in c.items() if v is not None and k != "weights"}

    base = fine(draws[0])
    worst = np.zeros(len(sel))
    for z_new in draws[1:]:
        got = fine(z_new)
        for k in base:
            worst = np.maximum(worst, scaled_errors(got[k], base[k], rtol_disp if k == "disp" else rtol, atol))
    out[mask] = worst
    return out
