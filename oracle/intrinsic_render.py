"""PyTorch-CPU restatement of IntrinsicNeRF's ``render_rays`` hot path (ORACLE - test infrastructure).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
file.  It is never on the product path.

One implementation covers both reference code bases; ``RenderConfig.variant`` selects which one is
being restated (all citations are relative to ``/root/reference``):

=============================  ==========================================  ===========================================
stage                          ``variant="object"``                        ``variant="ssr"``
=============================  ==========================================  ===========================================
frequency encoding             object_level/run_nerf_helpers.py:195-243    SSR/models/semantic_nerf.py:14-65 (x/10)
MLP forward                    object_level/run_nerf_helpers.py:284-325    SSR/models/semantic_nerf.py:123-181
encode + chunked MLP           object_level/run_nerf.py:32-56              SSR/models/model_utils.py:19-35
alpha compositing              object_level/run_nerf.py:359-412            SSR/models/model_utils.py:39-116
inverse-CDF resampling         object_level/run_nerf_helpers.py:402-445    SSR/models/rays.py:176-220
render_rays orchestration      object_level/run_nerf.py:415-528            SSR/training/trainer.py:717-808
=============================  ==========================================  ===========================================

The network is described by a plain ``dict[str, Tensor]`` with the reference's state-dict key names
(``state_dict_spec``), not by an ``nn.Module`` - the oracle has no module classes of its own.

Random inputs are *arguments* (``t_rand``, ``noise_coarse``, ``noise_fine``, ``u``): the reference
draws them from the global torch RNG (run_nerf.py:478,387; run_nerf_helpers.py:414); the caller of
the oracle draws them once and hands the same tensors to the oracle and to the HIP path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

NET_WIDTH = 256          # run_nerf.py:545-552 / SSR_room0_config.yaml:17-20 (every shipped config)
NET_DEPTH = 8
SKIP_AFTER = 4           # run_nerf.py:285 ``skips = [4]``
BASE_CHANNELS = 11       # rgb3, sigma, albedo3, shading1, residual3 (run_nerf_helpers.py:321)
ENDPOINT_DIM = 128       # semantic_nerf.py:163-164 / model_utils.py:99-103


@dataclass
class RenderConfig:
    variant: str = "object"          # "object" | "ssr"
    n_samples: int = 64
    n_importance: int = 128
    l_xyz: int = 10                  # multires
    l_dir: int = 4                   # multires_views
    white_bkgd: bool = False
    lindisp: bool = False            # object-level only (run_nerf.py:465-468)
    n_classes: int = 0               # SSR semantic head width C (0 = semantic head disabled)
    endpoint_feat: bool = False      # SSR: append 128-d views activation to the fine raw
    netchunk: int = 1 << 16

    @property
    def xyz_div(self) -> float:
        # trainer.py:817 ``scalar_factor=10`` for xyz, 1 for dirs (trainer.py:822-824)
        return 10.0 if self.variant == "ssr" else 1.0

    def raw_channels(self, fine: bool) -> int:
        ch = BASE_CHANNELS + (self.n_classes if self.variant == "ssr" else 0)
        if fine and self.variant == "ssr" and self.endpoint_feat:
            ch += ENDPOINT_DIM
        return ch


# --------------------------------------------------------------------------------------------
# network description
# --------------------------------------------------------------------------------------------
def state_dict_spec(variant: str, n_classes: int = 0, l_xyz: int = 10, l_dir: int = 4) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (key, shape) list of the reference module's parameters.

    object: run_nerf_helpers.py:259-279.  ssr: semantic_nerf.py:98-118.  ``Linear`` weights are
    ``[out, in]``.  This order is also the tensor order of the C-ABI packer (include/inerf.h).
    """
    w, e, d = NET_WIDTH, 3 + 6 * l_xyz, 3 + 6 * l_dir
    spec: List[Tuple[str, Tuple[int, ...]]] = []

    def lin(name, n_out, n_in):
        spec.append((name + ".weight", (n_out, n_in)))
        spec.append((name + ".bias", (n_out,)))

    for i in range(NET_DEPTH):
        n_in = e if i == 0 else (w + e if i == SKIP_AFTER + 1 else w)
        lin(f"pts_linears.{i}", w, n_in)
    lin("views_linears.0", w // 2, w + d)
    lin("feature_linear", w, w)
    lin("alpha_linear", 1, w)
    if variant == "object":
        lin("shading_linear", 3, w // 2)          # the RESIDUAL head (run_nerf_helpers.py:314-316)
        lin("albedo_linear1", w // 2, w)
        lin("albedo_linear2", 3, w // 2)
        lin("test_linear1", w // 2, w)            # the SHADING head (run_nerf_helpers.py:302-305)
        lin("test_linear2", 1, w // 2)
    elif variant == "ssr":
        if n_classes > 0:
            lin("semantic_linear.0.0", w // 2, w)
            lin("semantic_linear.1", n_classes, w // 2)
        lin("residual_linear", 3, w // 2)
        lin("albedo_linear1", w // 2, w)
        lin("albedo_linear2", 3, w // 2)
        lin("shading_linear1", w // 2, w)
        lin("shading_linear2", 1, w // 2)
    else:
        raise ValueError(variant)
    return spec


def make_state_dict(variant: str, n_classes: int = 0, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Default-``nn.Linear``-style init, U(-1/sqrt(fan_in), 1/sqrt(fan_in)), from a torch generator."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in state_dict_spec(variant, n_classes):
        if key.endswith(".weight"):
            fan_in = shape[1]
        bound = 1.0 / np.sqrt(fan_in)
        sd[key] = ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
    return sd


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic; identical on every platform)."""
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def lcg_state_dict(variant: str, n_classes: int = 0, seed: int = 0, sigma_gain_log2: int = 0,
                   sigma_bias: float = 0.0, weight_gain_log2: int = 0, freq_decay: bool = False,
                   l_xyz: int = 10, l_dir: int = 4) -> Dict[str, Tensor]:
    """Closed-form weights that are exactly representable in fp32 on every platform.

    Every value is ``k * 2**-15 * 2**-e`` with integer ``|k| <= 2**15`` and ``2**-e`` the power of two
    nearest below ``1/sqrt(fan_in)``, so a golden fixture only has to store ``seed``.
    ``weight_gain_log2`` scales every weight matrix by a power of two (1 => roughly variance-
    preserving through the ReLU trunk, so the output depends visibly on position like a trained
    net's does); ``sigma_gain_log2`` / ``sigma_bias`` (an exact dyadic value, please) further scale /
    shift ``alpha_linear`` so that densities, compositing weights and the resampling pdf are far
    from uniform.  ``freq_decay`` multiplies the input columns that carry frequency band ``f`` of the
    positional / directional encoding by ``2**-f`` (pts_linears.0, pts_linears.5, views_linears.0): a
    1/f spectrum like a trained network's, instead of the white spectrum of a random one whose
    output changes by O(1) when a sample moves by 1e-3 (which makes any fp32 evaluation of the
    two-pass path irreproducible, the reference's included - see tests/golden/make_golden.py).
    """
    def band_scale(n_cols_enc, n_freqs, offset, total):
        sc = np.ones(total)
        for f in range(n_freqs):
            sc[offset + 3 + 6 * f: offset + 9 + 6 * f] = 2.0 ** -f
        return sc

    e_cols, d_cols = 3 + 6 * l_xyz, 3 + 6 * l_dir
    sd = {}
    for t_idx, (key, shape) in enumerate(state_dict_spec(variant, n_classes, l_xyz, l_dir)):
        if key.endswith(".weight"):
            fan_in = shape[1]
        n = int(np.prod(shape))
        with np.errstate(over="ignore"):
            base = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
                    + np.uint64(t_idx + 1) * np.uint64(0xD1B54A32D192ED03))
            h = _mix64(np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + base)
        k = (h >> np.uint64(48)).astype(np.int64) - 32768          # [-32768, 32767]
        e = int(np.ceil(0.5 * np.log2(fan_in)))                     # 2**-e <= 1/sqrt(fan_in)
        vals = k.astype(np.float64) * 2.0 ** (-15 - e)
        if key.endswith(".weight"):
            vals = vals * 2.0 ** weight_gain_log2
            if freq_decay and key in ("pts_linears.0.weight", f"pts_linears.{SKIP_AFTER + 1}.weight"):
                vals = (vals.reshape(shape) * band_scale(e_cols, l_xyz, 0, shape[1])[None, :]).reshape(-1)
            if freq_decay and key == "views_linears.0.weight":
                vals = (vals.reshape(shape) * band_scale(d_cols, l_dir, NET_WIDTH, shape[1])[None, :]).reshape(-1)
        if key.startswith("alpha_linear"):
            vals = vals * 2.0 ** sigma_gain_log2
            if key.endswith(".bias"):
                vals = vals + sigma_bias
        sd[key] = torch.from_numpy(vals.reshape(shape)).to(torch.float32)
    return sd


# --------------------------------------------------------------------------------------------
# stages
# --------------------------------------------------------------------------------------------
def freq_encode(x: Tensor, n_freqs: int, div: float = 1.0) -> Tensor:
    """``[x, sin(x 2^0), cos(x 2^0), ..., sin(x 2^(L-1)), cos(x 2^(L-1))]`` on the last axis.

    run_nerf_helpers.py:216-225 (bands = ``2**linspace(0, L-1, L)`` are exact powers of two, :212);
    the SSR encoder first divides the input by ``scalar_factor`` (semantic_nerf.py:64) - a true
    division, kept as one here.
    """
    if div != 1.0:
        x = x / div
    parts = [x]
    for k in range(n_freqs):
        f = float(2 ** k)
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, dim=-1)


def mlp_forward(sd: Dict[str, Tensor], emb: Tensor, cfg: RenderConfig, endpoint: bool = False) -> Tensor:
    """Intrinsic NeRF MLP on embedded points ``emb[P, E+Dv]`` -> ``raw[P, 11(+C)(+128)]``.

    object: run_nerf_helpers.py:284-321.  ssr: semantic_nerf.py:123-181.  Trunk: 8 x (Linear, ReLU)
    with ``cat([pts, h])`` after layer 4; heads: sigma, albedo (sigmoid), shading (sigmoid),
    feature -> cat dirs -> 128 ReLU -> residual (sigmoid); ``rgb = albedo * shading + residual``.
    """
    e = 3 + 6 * cfg.l_xyz
    pts, dirs = emb[..., :e], emb[..., e:]
    h = pts
    for i in range(NET_DEPTH):
        h = F.relu(F.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"]))
        if i == SKIP_AFTER:
            h = torch.cat([pts, h], dim=-1)
    sigma = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
    if cfg.variant == "object":
        sh1, sh2, res = "test_linear1", "test_linear2", "shading_linear"
    else:
        sh1, sh2, res = "shading_linear1", "shading_linear2", "residual_linear"
    sem = None
    if cfg.variant == "ssr" and cfg.n_classes > 0:
        sem = F.relu(F.linear(h, sd["semantic_linear.0.0.weight"], sd["semantic_linear.0.0.bias"]))
        sem = F.linear(sem, sd["semantic_linear.1.weight"], sd["semantic_linear.1.bias"])
    albedo = F.relu(F.linear(h, sd["albedo_linear1.weight"], sd["albedo_linear1.bias"]))
    albedo = torch.sigmoid(F.linear(albedo, sd["albedo_linear2.weight"], sd["albedo_linear2.bias"]))
    shading = F.relu(F.linear(h, sd[sh1 + ".weight"], sd[sh1 + ".bias"]))
    shading = torch.sigmoid(F.linear(shading, sd[sh2 + ".weight"], sd[sh2 + ".bias"]))
    feature = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
    v = torch.cat([feature, dirs], dim=-1)
    v = F.relu(F.linear(v, sd["views_linears.0.weight"], sd["views_linears.0.bias"]))
    residual = torch.sigmoid(F.linear(v, sd[res + ".weight"], sd[res + ".bias"]))
    rgb = albedo * shading + residual
    outs = [rgb, sigma, albedo, shading, residual]
    if sem is not None:
        outs.append(sem)
    if endpoint:
        outs.append(v)
    return torch.cat(outs, dim=-1)


def query_network(sd, pts: Tensor, viewdirs: Tensor, cfg: RenderConfig, endpoint: bool = False) -> Tensor:
    """Encode ``pts[N,S,3]`` and ``viewdirs[N,3]`` and run the MLP in ``netchunk`` row blocks.

    run_nerf.py:42-56 / model_utils.py:19-35.
    """
    n, s, _ = pts.shape
    flat = pts.reshape(-1, 3)
    dflat = viewdirs[:, None, :].expand(n, s, 3).reshape(-1, 3)
    outs = []
    for i in range(0, flat.shape[0], cfg.netchunk):
        emb = torch.cat([freq_encode(flat[i:i + cfg.netchunk], cfg.l_xyz, cfg.xyz_div),
                         freq_encode(dflat[i:i + cfg.netchunk], cfg.l_dir, 1.0)], dim=-1)
        outs.append(mlp_forward(sd, emb, cfg, endpoint))
    out = torch.cat(outs, dim=0)
    return out.reshape(n, s, out.shape[-1])


def composite(raw: Tensor, z: Tensor, rays_d: Tensor, cfg: RenderConfig, noise: Optional[Tensor] = None,
              feat: bool = False) -> Dict[str, Tensor]:
    """Alpha-composite ``raw[N,S,CH]`` along each ray (run_nerf.py:359-412 / model_utils.py:39-116).

    Returns every map either reference variant produces; ``sem``/``feat`` are ``None`` when absent.
    ``disp`` is NaN where ``acc == 0`` exactly as in the reference (0/0 inside ``torch.max``).
    """
    far_gap = torch.full_like(z[..., :1], 1e10)
    dists = torch.cat([z[..., 1:] - z[..., :-1], far_gap], dim=-1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    sigma = raw[..., 3] if noise is None else raw[..., 3] + noise
    alpha = 1.0 - torch.exp(-F.relu(sigma) * dists)
    ones = torch.ones_like(alpha[:, :1])
    trans = torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    w = alpha * trans
    out = {
        "weights": w,
        "rgb": torch.sum(w[..., None] * raw[..., 0:3], dim=-2),
        "albedo": torch.sum(w[..., None] * raw[..., 4:7], dim=-2),
        "shading": torch.sum(w * raw[..., 7], dim=-1),
        "residual": torch.sum(w[..., None] * raw[..., 8:11], dim=-2),
        "sem": None,
        "feat": None,
    }
    n_cls = cfg.n_classes if cfg.variant == "ssr" else 0
    if n_cls > 0:
        out["sem"] = torch.sum(w[..., None] * raw[..., 11:11 + n_cls], dim=-2)
    if feat:
        out["feat"] = torch.sum(w[..., None] * raw[..., -ENDPOINT_DIM:], dim=-2)
    depth = torch.sum(w * z, dim=-1)
    acc = torch.sum(w, dim=-1)
    out["depth"] = depth
    out["disp"] = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / torch.sum(w, dim=-1))
    out["acc"] = acc
    if cfg.white_bkgd:
        out["rgb"] = out["rgb"] + (1.0 - acc[..., None])
        out["albedo"] = out["albedo"] + (1.0 - acc[..., None])
        out["shading"] = out["shading"] + (1.0 - acc)
        if out["sem"] is not None:                      # model_utils.py:113-114 (SSR only)
            out["sem"] = out["sem"] + (1.0 - acc[..., None])
    return out


def inverse_cdf_sample(bins: Tensor, weights: Tensor, u: Tensor) -> Tensor:
    """Draw ``u.shape[-1]`` depths per ray from the piecewise-constant pdf ``weights`` over ``bins``.

    run_nerf_helpers.py:402-445 / rays.py:176-220.  ``u`` is ``[N, n]`` (already expanded when it is
    the deterministic ``linspace(0, 1, n)``).
    """
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    u = u.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(idx - 1, min=0)
    hi = torch.clamp(idx, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    bin_lo, bin_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return bin_lo + t * (bin_hi - bin_lo)


def coarse_depths(near: Tensor, far: Tensor, t_vals: Tensor, lindisp: bool = False,
                  t_rand: Optional[Tensor] = None) -> Tensor:
    """Stratified depths ``z[N,S]`` (run_nerf.py:464-486 / trainer.py:730-746).

    ``near``/``far`` are ``[N,1]``; ``t_vals`` is the caller's ``linspace(0,1,S)``.
    """
    if not lindisp:
        z = near * (1.0 - t_vals) + far * t_vals
    else:
        z = 1.0 / (1.0 / near * (1.0 - t_vals) + 1.0 / far * t_vals)
    z = z.expand(near.shape[0], t_vals.shape[0])
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], dim=-1)
        lower = torch.cat([z[..., :1], mids], dim=-1)
        z = lower + (upper - lower) * t_rand
    return z


def render_rays(rays: Tensor, sd_coarse, sd_fine, cfg: RenderConfig, t_vals: Optional[Tensor] = None,
                u: Optional[Tensor] = None, t_rand: Optional[Tensor] = None,
                noise_coarse: Optional[Tensor] = None, noise_fine: Optional[Tensor] = None,
                stages: bool = False) -> Dict[str, Tensor]:
    """Whole hot path for a ray batch ``rays[N,11] = [o3, d3, near, far, viewdir3]``.

    Returns a dict with stage-neutral keys: ``{rgb,disp,acc,depth,albedo,shading,residual,sem}_
    {coarse,fine}``, ``z_std`` and (with ``stages=True``) ``z_coarse, raw_coarse, weights_coarse,
    z_samples, z_fine, raw_fine, weights_fine, feat_fine``.  ``u=None`` means the deterministic
    ``linspace(0, 1, N_importance)`` (perturb == 0 / eval mode).
    """
    dt = rays.dtype
    n = rays.shape[0]
    rays_o, rays_d, viewdirs = rays[:, 0:3], rays[:, 3:6], rays[:, 8:11]
    near, far = rays[:, 6:7], rays[:, 7:8]
    if t_vals is None:
        t_vals = torch.linspace(0.0, 1.0, cfg.n_samples, dtype=dt)
    z = coarse_depths(near, far, t_vals, cfg.lindisp, t_rand)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
    raw_c = query_network(sd_coarse, pts, viewdirs, cfg)
    comp_c = composite(raw_c, z, rays_d, cfg, noise_coarse)
    out: Dict[str, Tensor] = {}
    for k in ("rgb", "disp", "acc", "depth", "albedo", "shading", "residual", "sem"):
        if comp_c[k] is not None:
            out[k + "_coarse"] = comp_c[k]
    if stages:
        out.update(z_coarse=z, raw_coarse=raw_c, weights_coarse=comp_c["weights"])
    if cfg.n_importance > 0:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        if u is None:
            u = torch.linspace(0.0, 1.0, cfg.n_importance, dtype=dt)
        if u.dim() == 1:
            u = u.expand(n, cfg.n_importance)
        z_new = inverse_cdf_sample(mids, comp_c["weights"][..., 1:-1], u).detach()
        z_all, _ = torch.sort(torch.cat([z, z_new], dim=-1), dim=-1)
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_all[:, :, None]
        sd_f = sd_fine if sd_fine is not None else sd_coarse
        ep = cfg.variant == "ssr" and cfg.endpoint_feat
        raw_f = query_network(sd_f, pts, viewdirs, cfg, endpoint=ep)
        comp_f = composite(raw_f, z_all, rays_d, cfg, noise_fine, feat=ep)
        for k in ("rgb", "disp", "acc", "depth", "albedo", "shading", "residual", "sem", "feat"):
            if comp_f[k] is not None:
                out[k + "_fine"] = comp_f[k]
        out["z_std"] = torch.std(z_new, dim=-1, unbiased=False)
        if stages:
            out.update(z_samples=z_new, z_fine=z_all, raw_fine=raw_f, weights_fine=comp_f["weights"])
    return out
