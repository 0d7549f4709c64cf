"""Tensor-level launchers for the C ABI (``include/inerf.h``): validate, allocate, pass raw pointers.

torch is plumbing here - it owns device memory and the HIP stream; all arithmetic happens in
``libinerf.so``.  Every function requires fp32 tensors on a HIP device and raises otherwise: there is
no CPU or eager fallback.
"""
import ctypes as C
import os

import torch

from . import _capi
from ._capi import (BASE_CHANNELS, ENDPOINT_DIM, FLAG_ENDPOINT, FLAG_LINDISP, FLAG_U_PER_RAY, FLAG_WHITE_BKGD,
                    RAY_FLOATS, CompositeOut, RenderArgs)

MAX_POINTS_PER_LAUNCH = (1 << 31) - 1


def _dev(t, name, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} lives on {t.device}: intrinsicnerf_amd runs only on a HIP device "
                           "(no CPU / eager fallback exists)")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    if shape is not None:
        if t.dim() != len(shape) or any(s is not None and int(t.shape[i]) != s for i, s in enumerate(shape)):
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple('*' if s is None else s for s in shape)}")
    return t if t.is_contiguous() else t.contiguous()


def _opt(t, name, shape, like):
    if t is None:
        return None
    t = _dev(t, name, shape)
    if t.device != like.device:
        raise ValueError(f"{name} is on {t.device}, expected {like.device}")
    return t


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _new(like, *shape):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def sample_coarse(rays, t_vals, t_rand=None, lindisp=False):
    """z_vals[N,S] (run_nerf.py:464-486 / trainer.py:730-746)."""
    rays = _dev(rays, "rays", (None, RAY_FLOATS))
    t_vals = _dev(t_vals, "t_vals", (None,))
    n, s = rays.shape[0], t_vals.shape[0]
    t_rand = _opt(t_rand, "t_rand", (n, s), rays)
    z = _new(rays, n, s)
    with torch.cuda.device(rays.device):
        rc = _capi.lib().inerf_sample_coarse(_ptr(rays), _ptr(t_vals), _ptr(t_rand), n, s,
                                             FLAG_LINDISP if lindisp else 0, _ptr(z), _stream(rays))
    _capi.check(rc, "inerf_sample_coarse")
    return z


def gen_rays(poses, height, width, fx, fy, cx, cy, near, far, opengl, static_poses=None):
    """[B * H * W, 11] ray batch ``[o3, d3, near, far, viewdir3]`` of B pinhole cameras through ``inerf_gen_rays``
    (get_rays + render()'s assembly, run_nerf_helpers.py:359-368 + run_nerf.py:99-128 | create_rays, rays.py:223-256).
    ``poses``: [B, 3, 4] / [B, 4, 4] / [3, 4] / [4, 4] float32 device tensor; bit-identical to the reference's CPU rays."""
    def prep(t, name):
        t = _dev(t, name)
        if t.dim() == 2:
            t = t[None]
        if t.dim() != 3 or t.shape[1] not in (3, 4) or t.shape[2] != 4:
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected [B, 3|4, 4]")
        return t.contiguous()
    poses = prep(poses, "poses")
    b = poses.shape[0]
    if static_poses is not None:
        static_poses = prep(static_poses, "static_poses")
        if static_poses.shape != poses.shape:
            raise ValueError("static_poses must have the shape of poses")
    out = _new(poses, b * int(height) * int(width), RAY_FLOATS)
    with torch.cuda.device(poses.device):
        rc = _capi.lib().inerf_gen_rays(_ptr(poses), poses.shape[1] * 4, _ptr(static_poses), b, int(height), int(width),
                                        float(fx), float(fy), float(cx), float(cy), float(near), float(far),
                                        _capi.CAM_OPENGL if opengl else 0, _ptr(out), _stream(poses))
    _capi.check(rc, "inerf_gen_rays")
    return out


def frame_to_u8(values):
    """``(255 * clip(x, 0, 1)).astype(uint8)`` (to8b, run_nerf_helpers.py:13) on the device; same shape, dtype uint8."""
    values = _dev(values, "values")
    out = torch.empty(values.shape, dtype=torch.uint8, device=values.device)
    with torch.cuda.device(values.device):
        rc = _capi.lib().inerf_frame_to_u8(_ptr(values), values.numel(), C.c_void_p(out.data_ptr()), _stream(values))
    _capi.check(rc, "inerf_frame_to_u8")
    return out


def _new_status(like):
    return torch.zeros(1, dtype=torch.int32, device=like.device)


_deferred = None          # the innermost open deferred_range_checks() block, else None
_status_rays = 0          # rays per f16 range word of the launches inside a chunked_status() block (0: one word per launch)


class chunked_status:
    """Launches of ``render_rays_fused`` inside this block keep ONE f16 range word per ``rays_per_word`` rays
    (inerf_encode_mlp_chunked): the chunk loops of the front-ends render an eval-mode frame as one launch sequence and still learn
    which of the CALLER's chunks left the range.  In a ``deferred_range_checks`` block such a launch's words are tagged
    ``(block.tag, word index)``."""

    def __init__(self, rays_per_word):
        self.rays = int(rays_per_word)

    def __enter__(self):
        global _status_rays
        self.outer, _status_rays = _status_rays, self.rays
        return self

    def __exit__(self, *exc):
        global _status_rays
        _status_rays = self.outer
        return False
captured_status = None    # list collecting the status words of launches recorded into a HIP graph (graphs.GraphedTrainStep)


def _capturing():
    """True while the current stream records a HIP graph: nothing may be read back to the host then."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _defer_to_graph_owner(words):
    """Status words of launches that are being CAPTURED cannot be read here; they are handed to whoever owns the graph
    (graphs.GraphedTrainStep reads them after every replay).  Capturing without an owner would silently drop the f16 range
    guard, so that is an error."""
    if captured_status is None:
        raise RuntimeError("split-precision MLP launches are being captured into a HIP graph outside graphs.GraphedTrainStep: "
                           "their f16 range words would never be checked")
    captured_status.extend(words)


class deferred_range_checks:
    """Context manager for launches whose f16 range words can be read later: ``check_f16_range(..., deferrable=True)`` calls
    inside only remember their status word (together with the block's current ``tag``), and ONE device->host read at the
    end of the block checks them all - a multi-chunk frame costs one host synchronisation instead of one per chunk.

    At exit ``tripped`` holds the tags (in order, without repeats) whose launches left f16's range.  With
    ``raise_on_trip`` (default) a non-empty set raises FloatingPointError; the chunk loops of the front-ends pass False,
    tag every chunk with its index and re-render ONLY the tripped chunks in exact fp32.

    Blocks do not delegate: a block opened inside another one (a training step inside a frame loop) reads its OWN words at
    its own exit, so its handler - which restores the RNG and re-evaluates the same draws - is the one that runs."""

    def __init__(self, what, raise_on_trip=True):
        self.what, self.raise_on_trip = what, raise_on_trip
        self.tag, self.tripped = None, []
        self.words, self.tags = [], []

    def __len__(self):
        return len(self.words)

    def __enter__(self):
        global _deferred
        self.outer, _deferred = _deferred, self
        return self

    def __exit__(self, exc_type, exc, tb):
        global _deferred
        _deferred = self.outer
        if exc_type is not None or not self.words:
            return False
        if _capturing():
            _defer_to_graph_owner(self.words)
            return False
        flags = (self.words[0] if len(self.words) == 1 else torch.cat(self.words)).cpu().tolist()      # the block's one sync
        assert len(flags) == len(self.tags)
        for flag, tag in zip(flags, self.tags):
            if int(flag) & _capi.STATUS_F16_RANGE and tag not in self.tripped:
                self.tripped.append(tag)
        if self.tripped and self.raise_on_trip:
            raise FloatingPointError(_RANGE_MESSAGE.format(what=self.what))
        return False


_RANGE_MESSAGE = ("{what}: an activation exceeded the f16 range (|v| > 6e4) in the split-precision MLP kernel; "
                  "results are invalid - re-run with precision f32 (INERF_PRECISION=f32)")


def check_f16_range(status, what, deferrable=False):
    """Raise if a PREC_F16X3 launch met an activation outside f16's range (one device sync) - or, inside a
    ``deferred_range_checks`` block and with ``deferrable``, leave the word for the block's single check."""
    if status is None:
        return
    status = status.reshape(-1)                # (one word per launch, or one per chunk of its rays: chunked_status)
    if _capturing():
        _defer_to_graph_owner([status[i:i + 1] for i in range(status.numel())])
        return
    if deferrable and _deferred is not None:
        _deferred.words.append(status)
        _deferred.tags.extend([_deferred.tag] if status.numel() == 1 else [(_deferred.tag, i) for i in range(status.numel())])
        return
    if int(status.max().item()) & _capi.STATUS_F16_RANGE:
        raise FloatingPointError(_RANGE_MESSAGE.format(what=what))


def coalesced_chunk(n_rays, chunk, n_samples, n_importance, channels, device):
    """How many rays ONE launch sequence of an eval-mode frame may take instead of the caller's ``chunk``.

    The reference's ``chunk`` (run_nerf.py:59-71, training_utils.py:5-17: 32 768 rays) exists for 11 GB cards and "does not affect
    final results"; here results are bit-identical for any chunking (tests/test_full_size_properties.py, test_gpu_parity.py), a
    launch of 32 768 rays runs 7 % below a launch of 65 536+ (it ends on a partly filled wave of tiles per CU:
    profiles/r05_gemm_priority_ab.txt) and a frame's tail chunk is worse.  So when nothing depends on the chunk boundaries - no random
    draw per chunk, no ``raw`` returned, no autograd - the front-ends merge chunks up to a workspace cap: ``INERF_COALESCE_BYTES``
    (default 16 GiB, at most half the device's free memory; 0 = keep the caller's chunks).  Returns a multiple of ``chunk``."""
    import os
    cap = int(float(os.environ.get("INERF_COALESCE_BYTES", 16 * 2 ** 30)))
    if cap <= 0 or n_rays <= chunk or chunk <= 0:
        return chunk
    if device.type == "cuda":
        cap = min(cap, torch.cuda.mem_get_info(device)[0] // 2)
    s, f = int(n_samples), int(n_samples) + int(n_importance)
    # per ray: z coarse / new / merged, coarse weights, raw of both levels, maps (include/inerf.h: inerf_render_workspace_bytes), + 10 %
    per_ray = int(1.1 * 4 * (s + n_importance + f + s + (s + f) * channels + 2 * (16 + channels)))
    rays = min(n_rays, cap // max(per_ray, 1))
    return max(chunk, rays // chunk * chunk if rays < n_rays else -(-n_rays // chunk) * chunk)


_warned_fallback = False


def warn_f32_fallback(e):
    global _warned_fallback
    if not _warned_fallback:
        import warnings
        warnings.warn(f"{e}  Re-running this batch with the exact fp32 MFMA kernel (set INERF_PRECISION=f32 to "
                      "skip the attempt).")
        _warned_fallback = True


def with_f32_fallback(desc, run):
    """``run(desc)`` and, if the split-precision kernel reports an out-of-range activation, once more in exact fp32.

    ``run`` must call ``check_f16_range`` on its status word (that is the one host sync of the f16x3 mode)."""
    global _warned_fallback
    try:
        return run(desc)
    except FloatingPointError as e:
        if desc.precision != _capi.PREC_F16X3:
            raise
        warn_f32_fallback(e)
        d32 = _capi.NetDesc(desc.variant, desc.n_classes, desc.l_xyz, desc.l_dir, desc.xyz_div, _capi.PREC_F32)
        return run(d32)


def encode_mlp(desc, packed, rays, z_vals, endpoint=False, status=None):
    """raw[N,S,CH]: fused encoding + MLP (run_network + NeRF.forward).

    ``status``: optional int32[1] device tensor that collects INERF_STATUS_* bits (PREC_F16X3 range check)."""
    rays = _dev(rays, "rays", (None, RAY_FLOATS))
    z_vals = _dev(z_vals, "z_vals", (rays.shape[0], None))
    packed = _dev(packed, "packed weights", (None,))
    n, s = z_vals.shape
    flags = FLAG_ENDPOINT if endpoint else 0
    ch = _capi.lib().inerf_raw_channels(desc, flags, 1)
    raw = _new(rays, n, s, ch)
    with torch.cuda.device(rays.device):
        ws_bytes = int(_capi.lib().inerf_encode_mlp_workspace_bytes(desc, n, s, flags))      # > 0: the SSR network's semantic-head scratch
        if ws_bytes < 0:
            _capi.check(ws_bytes, "inerf_encode_mlp_workspace_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=rays.device) if ws_bytes > 0 else None
        rc = _capi.lib().inerf_encode_mlp_ws(desc, _ptr(packed), _ptr(rays), _ptr(z_vals), n, s, flags, _ptr(raw),
                                             None if status is None else C.c_void_p(status.data_ptr()),
                                             None if ws is None else C.c_void_p(ws.data_ptr()), ws_bytes, _stream(rays))
    _capi.check(rc, "inerf_encode_mlp_ws")
    return raw


_COMPOSITE_KEYS = ("rgb", "disp", "acc", "depth", "albedo", "shading", "residual", "sem", "feat", "weights")


def composite(raw, z_vals, rays_d, noise=None, white_bkgd=False, n_classes=0, feat_dim=0, want_weights=True):
    """raw2outputs (run_nerf.py:359-412 / model_utils.py:39-116) -> dict of maps.

    When ``raw`` requires grad (a training step through the staged path) the maps carry a grad_fn whose backward
    is the HIP kernel behind ``inerf_composite_backward``; ``z_vals`` / ``rays_d`` / ``noise`` get no gradient,
    as in the reference (resampled depths are detached, run_nerf.py:501)."""
    if torch.is_grad_enabled() and isinstance(raw, torch.Tensor) and raw.requires_grad:
        keys, maps = _CompositeFn.run(raw, z_vals, rays_d, noise, white_bkgd, n_classes, feat_dim, want_weights)
        return dict(zip(keys, maps))
    return _composite_forward(raw, z_vals, rays_d, noise, white_bkgd, n_classes, feat_dim, want_weights)


def composite_backward(raw, z_vals, rays_d, grads, noise=None, white_bkgd=False, n_classes=0, feat_dim=0):
    """d_raw[N,S,CH] for the output gradients in ``grads`` (dict: subset of rgb, disp, acc, depth, albedo, shading,
    residual, sem, feat, weights) - what autograd computes for raw2outputs in the reference's training step."""
    raw = _dev(raw, "raw", (None, None, None))
    n, s, ch = raw.shape
    z_vals = _dev(z_vals, "z_vals", (n, s))
    rays_d = _dev(rays_d, "rays_d", (n, 3))
    noise = _opt(noise, "noise", (n, s), raw)
    shapes = {"rgb": (n, 3), "albedo": (n, 3), "residual": (n, 3), "disp": (n,), "acc": (n,), "depth": (n,), "shading": (n,),
              "sem": (n, n_classes), "feat": (n, feat_dim), "weights": (n, s)}
    held = {}
    for k, t in grads.items():
        if k not in shapes:
            raise KeyError(f"unknown output {k!r}")
        if t is not None:
            held[k] = _dev(t, "grad of " + k, shapes[k])
    d_raw = _new(raw, n, s, ch)
    co = CompositeOut(**{k: t.data_ptr() for k, t in held.items()})
    with torch.cuda.device(raw.device):
        rc = _capi.lib().inerf_composite_backward(_ptr(raw), _ptr(z_vals), _ptr(rays_d), 3, _ptr(noise), n, s, ch, n_classes,
                                                  feat_dim, FLAG_WHITE_BKGD if white_bkgd else 0, C.byref(co), _ptr(d_raw),
                                                  _stream(raw))
    _capi.check(rc, "inerf_composite_backward")
    return d_raw


class _CompositeFn(torch.autograd.Function):
    """Differentiable raw2outputs: HIP forward, HIP backward (the forward is recomputed there: only inputs are saved)."""

    @staticmethod
    def run(raw, z_vals, rays_d, noise, white_bkgd, n_classes, feat_dim, want_weights):
        keys = [k for k in _COMPOSITE_KEYS if not (k == "sem" and n_classes == 0) and not (k == "feat" and feat_dim == 0)
                and not (k == "weights" and not want_weights)]
        maps = _CompositeFn.apply(raw, z_vals, rays_d, noise, white_bkgd, n_classes, feat_dim, tuple(keys))
        return keys, maps

    @staticmethod
    def forward(ctx, raw, z_vals, rays_d, noise, white_bkgd, n_classes, feat_dim, keys):
        raw_c, z_c, d_c = raw.detach().float().contiguous(), z_vals.detach().float().contiguous(), rays_d.detach().float().contiguous()
        noise_c = None if noise is None else noise.detach().float().contiguous()
        out = _composite_forward(raw_c, z_c, d_c, noise_c, white_bkgd, n_classes, feat_dim, "weights" in keys)
        # Outputs the loss does not use must arrive as None, not as zero tensors: the reference's autograd never visits the
        # branch of an unused output, while a ZERO cotangent on disp = 1 / (depth / acc) turns into NaN on every ray with
        # acc == 0 (0 x d(1/(0/0))) - and a trained network has such rays (sigma <= 0 along a ray through empty space) in every
        # batch.  Found by scripts/fit_synthetic.py: with materialised zeros the fit went NaN after a few hundred steps.
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(raw_c, z_c, d_c, *([] if noise_c is None else [noise_c]))
        ctx.cfg = (white_bkgd, n_classes, feat_dim, keys, noise_c is not None)
        return tuple(out[k] for k in keys)

    @staticmethod
    def backward(ctx, *gouts):
        white_bkgd, n_classes, feat_dim, keys, has_noise = ctx.cfg
        saved = ctx.saved_tensors
        raw, z_vals, rays_d = saved[:3]
        noise = saved[3] if has_noise else None
        grads = {k: g for k, g in zip(keys, gouts) if g is not None}
        d_raw = composite_backward(raw, z_vals, rays_d, grads, noise, white_bkgd, n_classes, feat_dim)
        return d_raw, None, None, None, None, None, None, None


def _composite_forward(raw, z_vals, rays_d, noise=None, white_bkgd=False, n_classes=0, feat_dim=0, want_weights=True):
    raw = _dev(raw, "raw", (None, None, None))
    n, s, ch = raw.shape
    z_vals = _dev(z_vals, "z_vals", (n, s))
    rays_d = _dev(rays_d, "rays_d", (n, 3))
    noise = _opt(noise, "noise", (n, s), raw)
    out = {k: _new(raw, n, 3) for k in ("rgb", "albedo", "residual")}
    out.update({k: _new(raw, n) for k in ("disp", "acc", "depth", "shading")})
    if n_classes > 0:
        out["sem"] = _new(raw, n, n_classes)
    if feat_dim > 0:
        out["feat"] = _new(raw, n, feat_dim)
    if want_weights:
        out["weights"] = _new(raw, n, s)
    co = CompositeOut(**{k: t.data_ptr() for k, t in out.items()})
    with torch.cuda.device(raw.device):
        rc = _capi.lib().inerf_composite(_ptr(raw), _ptr(z_vals), _ptr(rays_d), 3, _ptr(noise), n, s, ch, n_classes,
                                         feat_dim, FLAG_WHITE_BKGD if white_bkgd else 0, C.byref(co), _stream(raw))
    _capi.check(rc, "inerf_composite")
    return out


def _u_arg(u, n, n_imp, like):
    u = _dev(u, "u")
    if u.dim() == 1 and u.shape[0] == n_imp:
        return u, 0
    if u.dim() == 2 and tuple(u.shape) == (n, n_imp):
        return u, FLAG_U_PER_RAY
    raise ValueError(f"u has shape {tuple(u.shape)}, expected ({n_imp},) or ({n}, {n_imp})")


def sample_fine(z_coarse, weights, u, n_importance):
    """z_mid + sample_pdf + sort(cat) + std (run_nerf.py:499-503,519) -> (z_samples, z_merged, z_std)."""
    z_coarse = _dev(z_coarse, "z_coarse", (None, None))
    n, sc = z_coarse.shape
    weights = _dev(weights, "weights", (n, sc))
    u, flags = _u_arg(u, n, n_importance, z_coarse)
    z_s, z_m, z_std = _new(z_coarse, n, n_importance), _new(z_coarse, n, sc + n_importance), _new(z_coarse, n)
    with torch.cuda.device(z_coarse.device):
        rc = _capi.lib().inerf_sample_fine(_ptr(z_coarse), _ptr(weights), _ptr(u), n, sc, n_importance, flags,
                                           _ptr(z_s), _ptr(z_m), _ptr(z_std), _stream(z_coarse))
    _capi.check(rc, "inerf_sample_fine")
    return z_s, z_m, z_std


def sample_pdf(bins, weights, u, n_samples):
    """Stand-alone sample_pdf(bins, weights, N) (run_nerf_helpers.py:402-445)."""
    bins = _dev(bins, "bins", (None, None))
    n, nb = bins.shape
    weights = _dev(weights, "weights", (n, nb - 1))
    u, flags = _u_arg(u, n, n_samples, bins)
    out = _new(bins, n, n_samples)
    with torch.cuda.device(bins.device):
        rc = _capi.lib().inerf_sample_pdf(_ptr(bins), _ptr(weights), _ptr(u), n, nb, n_samples, flags, _ptr(out),
                                          _stream(bins))
    _capi.check(rc, "inerf_sample_pdf")
    return out


_MAP_KEYS = ("rgb", "disp", "acc", "depth", "albedo", "shading", "residual")


def render_rays_fused(desc, packed_coarse, packed_fine, rays, n_samples, n_importance, t_vals, u=None, t_rand=None,
                      noise_coarse=None, noise_fine=None, white_bkgd=False, lindisp=False, endpoint=False,
                      want_raw_coarse=False, want_raw_fine=False, want_stages=False, want_sem=True):
    """Whole render_rays path for one ray batch through ``inerf_render_rays``.

    Returns a dict with ``{rgb,disp,acc,depth,albedo,shading,residual[,sem]}_{coarse,fine}``, ``z_std``
    and, on request, ``raw_*`` / stage tensors (``z_coarse, weights_coarse, z_samples, z_fine,
    weights_fine``).  Everything is enqueued on the current stream; nothing synchronises.
    """
    L = _capi.lib()
    rays = _dev(rays, "rays", (None, RAY_FLOATS))
    n = rays.shape[0]
    if n * (n_samples + n_importance) > MAX_POINTS_PER_LAUNCH:
        raise ValueError("ray batch too large for one launch; chunk it (the render() front-ends do)")
    packed_coarse = _dev(packed_coarse, "packed_coarse", (None,))
    packed_fine = _opt(packed_fine, "packed_fine", (None,), rays)
    t_vals = _dev(t_vals, "t_vals", (n_samples,))
    t_rand = _opt(t_rand, "t_rand", (n, n_samples), rays)
    noise_coarse = _opt(noise_coarse, "noise_coarse", (n, n_samples), rays)
    s_f = n_samples + n_importance
    noise_fine = _opt(noise_fine, "noise_fine", (n, s_f), rays) if n_importance > 0 else None
    flags = (FLAG_WHITE_BKGD if white_bkgd else 0) | (FLAG_LINDISP if lindisp else 0) | (FLAG_ENDPOINT if endpoint else 0)
    if n_importance > 0:
        if u is None:
            raise ValueError("u is required when n_importance > 0")
        u, uf = _u_arg(u, n, n_importance, rays)
        flags |= uf
    n_cls = desc.n_classes if (desc.variant == _capi.VARIANT_SSR and want_sem) else 0
    ch_c, ch_f = L.inerf_raw_channels(desc, flags, 0), L.inerf_raw_channels(desc, flags, 1)

    out = {}

    def maps(level, s_count, with_feat):
        d = {k: _new(rays, n, 3) for k in ("rgb", "albedo", "residual")}
        d.update({k: _new(rays, n) for k in ("disp", "acc", "depth", "shading")})
        if n_cls > 0:
            d["sem"] = _new(rays, n, n_cls)
        if with_feat:
            d["feat"] = _new(rays, n, ENDPOINT_DIM)
        if want_stages:
            d["weights"] = _new(rays, n, s_count)
        for k, t in d.items():
            out[f"{k}_{level}"] = t
        return CompositeOut(**{k: t.data_ptr() for k, t in d.items()})

    args = RenderArgs()
    args.net = desc
    args.packed_coarse, args.packed_fine = packed_coarse.data_ptr(), (packed_fine.data_ptr() if packed_fine is not None else None)
    args.rays, args.n_rays, args.n_samples, args.n_importance, args.flags = rays.data_ptr(), n, n_samples, n_importance, flags
    args.t_vals = t_vals.data_ptr()
    args.t_rand = t_rand.data_ptr() if t_rand is not None else None
    args.u = u.data_ptr() if n_importance > 0 else None
    args.noise_coarse = noise_coarse.data_ptr() if noise_coarse is not None else None
    args.noise_fine = noise_fine.data_ptr() if noise_fine is not None else None
    args.coarse = maps("coarse", n_samples, False)
    if n_importance > 0:
        args.fine = maps("fine", s_f, bool(endpoint and desc.variant == _capi.VARIANT_SSR))
        out["z_std"] = _new(rays, n)
        args.z_std = out["z_std"].data_ptr()
    if want_raw_coarse:
        out["raw_coarse"] = _new(rays, n, n_samples, ch_c)
        args.raw_coarse = out["raw_coarse"].data_ptr()
    if want_raw_fine and n_importance > 0:
        out["raw_fine"] = _new(rays, n, s_f, ch_f)
        args.raw_fine = out["raw_fine"].data_ptr()
    if want_stages:
        out["z_coarse"] = _new(rays, n, n_samples)
        args.z_coarse = out["z_coarse"].data_ptr()
        if n_importance > 0:
            out["z_samples"], out["z_fine"] = _new(rays, n, n_importance), _new(rays, n, s_f)
            args.z_samples, args.z_fine = out["z_samples"].data_ptr(), out["z_fine"].data_ptr()
    ws_bytes = L.inerf_render_workspace_bytes(C.byref(args))       # nothing is reserved for stage tensors requested as outputs
    if ws_bytes < 0:
        _capi.check(int(ws_bytes), "inerf_render_workspace_bytes")
    ws = torch.empty(max(int(ws_bytes), 1), dtype=torch.uint8, device=rays.device)
    args.workspace, args.workspace_bytes = ws.data_ptr(), int(ws_bytes)
    if desc.precision == _capi.PREC_F16X3:
        # caller checks it (check_f16_range) when it next synchronises; inside a chunked_status() block: one word per chunk of rays
        words = -(-n // _status_rays) if (_status_rays > 0 and n > _status_rays) else 1
        out["status"] = torch.zeros(words, dtype=torch.int32, device=rays.device)
        args.status = out["status"].data_ptr()
        args.status_rays = _status_rays if words > 1 else 0
    with torch.cuda.device(rays.device):
        rc = L.inerf_render_rays(C.byref(args), _stream(rays))
    _capi.check(rc, "inerf_render_rays")
    # `ws` and the input tensors are kept alive by the caching allocator's stream semantics: they are
    # released on the same stream the kernels were enqueued on.
    return out


# ---------------------------------------------------------------------------------------------------------------------
# training: fused forward that keeps the activations, MFMA input-gradient chain, split-K MFMA weight gradients
# ---------------------------------------------------------------------------------------------------------------------
(SAVE_ENC, SAVE_DIR, SAVE_H0, SAVE_AS1H, SAVE_FEAT, SAVE_VH, SAVE_SEMH, SAVE_DPRE, SAVE_H7R, SAVE_SLOTS) = (0, 1, 2, 10, 11, 12, 13, 14, 15, 16)
ACT_SCALE = 8.0                 # activations travel as f16 hi/lo of 8 * value (csrc/layout.h kActScale)
SAVE_SCALARS = 64               # floats behind the slots and the mask area (reserved; include/inerf.h)


def frag_decode(frag, n_points, scale=ACT_SCALE, width=256):
    """A FRAGMENT slot of an ACTIVATION buffer (include/inerf.h: f16 hi/lo operand fragments of the weight-gradient products;
    ``frag``: its elements as a float32 or float16 tensor; ``width``: 256, or 64 for the encoding's slot) -> the fp32
    [n_points, width] matrix it encodes: (hi + lo) / scale."""
    h = frag.view(torch.float16) if frag.dtype != torch.float16 else frag
    cbs = width // 32
    tiles = h.numel() // (64 * width * 2)
    # [tile, pb, q, cb, plane, h, c, i_hi, i_lo]: point = 32 pb + 16 q + 8 i_hi + 4 h + i_lo, channel = 32 cb + c
    v = h.view(tiles, 2, 2, cbs, 2, 2, 32, 2, 4).float()
    v = v[:, :, :, :, 0] + v[:, :, :, :, 1]                                  # hi + lo: [tile, pb, q, cb, h, c, i_hi, i_lo]
    v = v.permute(0, 1, 2, 6, 4, 7, 3, 5).reshape(tiles * 64, width)        # -> [tile, pb, q, i_hi, h, i_lo, cb, c]
    return v[:n_points] / scale


def frag_encode(rows, scale=ACT_SCALE):
    """fp32 [n_points, W] (W = 256 or 64) -> the FRAGMENT slot of ``frag_decode`` (float16 tensor of 64 * ceil(n / 64) * 2 W
    halfs): hi = f16 of scale * value rounded towards zero, lo = f16(scale * value - hi); padding points are zero.  What the
    training forward's epilogue emits, restated with torch for the tests of the weight-gradient kernel."""
    n, width = rows.shape
    tiles = (n + 63) // 64
    v = torch.zeros(tiles * 64, width, dtype=torch.float32, device=rows.device)
    v[:n] = rows.float() * scale
    hi = (v.view(torch.int32) & ~0x1FFF).view(torch.float32)                 # 13 low mantissa bits cleared: exact in f16's normal range
    hi16 = hi.clamp(-65504.0, 65504.0).half()
    sub = hi16.float().abs() < 2.0 ** -14                                   # subnormal results were rounded, not truncated: fix up
    hi16 = torch.where(sub & (hi16.float().abs() > v.abs()), torch.nextafter(hi16.float(), torch.zeros_like(v)).half(), hi16)
    lo16 = (v - hi16.float()).half()
    both = torch.stack([hi16, lo16], 0)                                      # [plane, point, channel]
    both = both.view(2, tiles, 2, 2, 2, 2, 4, width // 32, 32)               # [plane, tile, pb, q, i_hi, h, i_lo, cb, c]
    return both.permute(1, 2, 3, 7, 0, 5, 8, 4, 6).contiguous().view(-1)     # [tile, pb, q, cb, plane, h, c, i_hi, i_lo]


def grad_frag_encode(rows):
    """fp32 [n_points, W] gradients (W = 256, or 128) -> (FRAGMENT slot, normalisers) as the input-gradient chain writes them into a gradient
    buffer: every point p is normalised by s_p = the power of two above its largest |value| (the chain uses its largest HEAD
    gradient; any power of two keeps the encoding exact) and stored as the split f16 halves of 8 * value / s_p;
    normalisers: float32[64 * ceil(n / 64) + 64] (1 for padding points; 64 readable floats behind the end)."""
    n = rows.shape[0]
    tiles = (n + 63) // 64
    m = rows.float().abs().amax(1)
    _, e = torch.frexp(m)
    s = torch.where(m > 0, torch.ldexp(torch.ones_like(m), e), torch.ones_like(m))
    scales = torch.ones(tiles * 64 + 64, dtype=torch.float32, device=rows.device)
    scales[:n] = s
    return frag_encode(rows.float() / s[:, None], ACT_SCALE), scales


def grad_frag_decode(frag, scales, n_points, width=256):
    """The inverse: (FRAGMENT slot of a gradient buffer, the points' normalisers) -> fp32 [n_points, width]."""
    return frag_decode(frag, n_points, ACT_SCALE, width) * scales[:n_points, None]


def save_slot_views(desc, buf, n_points, gradient=False):
    """The [n_points, width] matrices of an activation buffer (``gradient``: of a buffer of pre-activation gradients;
    include/inerf.h: slot list).  Row-format slots are views; FRAGMENT slots are decoded into new fp32 tensors."""
    views = []
    lib = _capi.lib()
    off, width = C.c_int64(), C.c_int()
    padded = (n_points + 63) // 64 * 64
    for slot in range(SAVE_SLOTS):
        _capi.check(lib.inerf_mlp_save_slot(desc, slot, n_points, C.byref(off), C.byref(width)), "inerf_mlp_save_slot")
        if width.value == 0:                        # (a slot this network does not have)
            views.append(buf[0:0].view(n_points, 0))
        elif lib.inerf_mlp_save_slot_is_fragment(slot, 1 if gradient else 0) == 1:
            frag = buf[off.value: off.value + padded * width.value]
            views.append(grad_frag_decode(frag, views[SAVE_ENC], n_points, width.value) if gradient else frag_decode(frag, n_points, width=width.value))
        elif gradient and slot == SAVE_ENC:         # the points' normalisers (include/inerf.h), not an [n, 64] matrix
            views.append(buf[off.value: off.value + padded])
        else:
            views.append(buf[off.value: off.value + n_points * width.value].view(n_points, width.value))
    return views


def encode_mlp_train(desc, packed, rays, z_vals, endpoint=False, status=None, act_max=None):
    """``encode_mlp`` that also returns the activation buffer the backward pass needs (PREC_F16X3 only).  ``act_max``:
    optional zeroed float32[1] device tensor that receives max |activation|."""
    rays = _dev(rays, "rays", (None, RAY_FLOATS))
    z_vals = _dev(z_vals, "z_vals", (rays.shape[0], None))
    packed = _dev(packed, "packed weights", (None,))
    n, s = z_vals.shape
    flags = FLAG_ENDPOINT if endpoint else 0
    ch = _capi.lib().inerf_raw_channels(desc, flags, 1)
    raw = _new(rays, n, s, ch)
    save = _new(rays, _capi.lib().inerf_mlp_save_floats(desc, n * s))
    with torch.cuda.device(rays.device):
        rc = _capi.lib().inerf_encode_mlp_train(desc, _ptr(packed), _ptr(rays), _ptr(z_vals), n, s, flags, _ptr(raw), _ptr(save),
                                                _ptr(act_max), None if status is None else C.c_void_p(status.data_ptr()),
                                                _stream(rays))
    _capi.check(rc, "inerf_encode_mlp_train")
    return raw, save


def mlp_backward_inputs(desc, packed_bwd, raw, d_raw, save, endpoint=False, status=None, dz_max=None, want_heads=False):
    """Pre-activation gradients of every layer (same slot layout as ``save``) from d loss / d raw.  ``dz_max``: optional
    zeroed float32[1] device tensor that receives max |dz| (the weight-gradient kernel's operand range).  ``want_heads``:
    also return the summed weight / bias gradients of the 1-4-row heads (see include/inerf.h) as a second value."""
    raw = _dev(raw, "raw", (None, None))
    p, ch = raw.shape
    d_raw = _dev(d_raw, "d_raw", (p, ch))
    save = _dev(save, "save", (None,))
    packed_bwd = _dev(packed_bwd, "packed transposed weights", (None,))
    dz = _new(raw, save.shape[0])
    heads = _new(raw, _capi.lib().inerf_mlp_backward_grid(p), _capi.lib().inerf_mlp_head_partial_floats()) if want_heads else None
    with torch.cuda.device(raw.device):
        rc = _capi.lib().inerf_mlp_backward_inputs(desc, _ptr(packed_bwd), _ptr(raw), _ptr(d_raw), _ptr(save), p,
                                                   FLAG_ENDPOINT if endpoint else 0, _ptr(dz), _ptr(dz_max), _ptr(heads),
                                                   None if status is None else C.c_void_p(status.data_ptr()), _stream(raw))
    _capi.check(rc, "inerf_mlp_backward_inputs")
    return (dz, heads.sum(0)) if want_heads else dz


def mlp_backward(desc, packed_bwd, raw, d_raw, save, act_max, endpoint=False, status=None):
    """The network's whole backward pass through ``inerf_mlp_backward`` (ONE call: chain, every weight-gradient product,
    reduction, scatter into the reference's parameter layout).  Returns the flat gradient blob - the parameter tensors of
    ``packing.tensor_table(desc)`` in order; ``param_views`` cuts it into named views."""
    L = _capi.lib()
    raw = _dev(raw, "raw", (None, None))
    p, ch = raw.shape
    d_raw = _dev(d_raw, "d_raw", (p, ch))
    save = _dev(save, "save", (None,))
    packed_bwd = _dev(packed_bwd, "packed transposed weights", (None,))
    act_max = _dev(act_max, "act_max", (1,))
    grads = _new(raw, L.inerf_param_floats(desc))
    ws_bytes = L.inerf_mlp_backward_workspace_bytes(desc, p)
    if ws_bytes < 0:
        _capi.check(int(ws_bytes), "inerf_mlp_backward_workspace_bytes")
    ws = torch.empty(max(int(ws_bytes), 1), dtype=torch.uint8, device=raw.device)
    with torch.cuda.device(raw.device):
        rc = L.inerf_mlp_backward(desc, _ptr(packed_bwd), _ptr(raw), _ptr(d_raw), _ptr(save), _ptr(act_max), p,
                                  FLAG_ENDPOINT if endpoint else 0, _ptr(grads), C.c_void_p(ws.data_ptr()), int(ws_bytes),
                                  None if status is None else C.c_void_p(status.data_ptr()), _stream(raw))
    _capi.check(rc, "inerf_mlp_backward")
    return grads


def param_views(desc, flat):
    """name -> view of the flat gradient / parameter blob with the reference's shapes (packing.tensor_table order)."""
    from . import packing
    out, off = {}, 0
    for name, (rows, cols) in packing.tensor_table(desc):
        n = rows * (cols if cols else 1)
        out[name] = flat[off:off + n].view((rows, cols) if cols else (rows,))
        off += n
    return out


def _head_names(desc):
    if desc.variant == _capi.VARIANT_OBJECT:      # run_nerf_helpers.py:259-279: 'shading_linear' is the residual head
        return "test_linear1", "test_linear2", "shading_linear"
    return "shading_linear1", "shading_linear2", "residual_linear"


def _split_k(n_points, target=64):
    """Largest divisor of n_points not above ``target``: dW = dZ^T X has a tiny output (<= 256 x 320) and K = n_points in
    the hundreds of thousands, which a GEMM library runs on a handful of workgroups; as a batched GEMM over K-chunks plus
    a sum it fills the chip (measured on 393 216 points: 11.0 ms with 8 chunks, 5.8 ms with 64, no gain beyond)."""
    for nc in range(min(target, n_points), 0, -1):
        if n_points % nc == 0:
            return nc
    return 1


def _tn(g, x, nc):
    """g^T @ x for g[P, M], x[P, N], split over P into nc chunks."""
    if nc == 1:
        return g.t() @ x
    p = g.shape[0]
    return torch.bmm(g.view(nc, p // nc, g.shape[1]).transpose(1, 2), x.view(nc, p // nc, x.shape[1])).sum(0)


def _pow2_scale(t):
    """Device scalar 2^k with max|t| * 2^k in [2^13, 2^14) (1 for an all-zero tensor) - no host sync."""
    m = t.abs().amax()
    _, e = torch.frexp(m)
    return torch.where((m > 0) & torch.isfinite(m), torch.ldexp(torch.ones_like(m), 14 - e), torch.ones_like(m))


ACTIVATION_RANGE = 6.0e4 / 8.0          # what the forward's f16 range check admits


class _WgradBatch:
    """Several ``G^T X`` products (and the column sums of their G) through inerf_mlp_weight_gradient, every workgroup's
    partial tiles side by side in ONE [grid, total] buffer: one launch per product, one sum at the end."""

    def __init__(self, n_points, ranges, like):
        self.p, self.ranges, self.like = n_points, ranges, like
        self.jobs, self.total = [], 0

    def add(self, g, x, m, n, want_bias=False, ranges=None):
        """Schedules g[:, :m]^T @ x[:, :n]; returns a key for ``result`` (and ``bias`` if want_bias).  ``ranges``: this
        product's own operand bounds (default: the batch's)."""
        job = dict(g=g, x=x, m=m, n=n, off=self.total, boff=None, ranges=ranges)
        self.total += m * n
        if want_bias:
            job["boff"] = self.total
            self.total += m
        self.jobs.append(job)
        return len(self.jobs) - 1

    def run(self):
        lib = _capi.lib()
        if self.p == 0:                        # no sample points: every gradient is zero
            self.sums = torch.zeros(self.total, dtype=torch.float32, device=self.like.device)
            return
        grid = lib.inerf_wgrad_grid(self.p)
        buf = _new(self.like, grid, self.total)
        with torch.cuda.device(self.like.device):
            for j in self.jobs:
                g, x = j["g"], j["x"]
                base = buf.data_ptr()
                rc = lib.inerf_mlp_weight_gradient(C.c_void_p(g.data_ptr()), g.stride(0), C.c_void_p(x.data_ptr()), x.stride(0), self.p,
                                                   j["m"], j["n"], _ptr(self.ranges if j["ranges"] is None else j["ranges"]),
                                                   C.c_void_p(base + 4 * j["off"]),
                                                   None if j["boff"] is None else C.c_void_p(base + 4 * j["boff"]), self.total,
                                                   _stream(self.like))
                _capi.check(rc, "inerf_mlp_weight_gradient")
        self.sums = buf.sum(0)

    def result(self, key):
        j = self.jobs[key]
        return self.sums[j["off"]: j["off"] + j["m"] * j["n"]].view(j["m"], j["n"])

    def bias(self, key):
        j = self.jobs[key]
        return self.sums[j["boff"]: j["boff"] + j["m"]]


def weight_gradient(g, x, m, n, ranges=None, want_bias=False):
    """g[:, :m]^T @ x[:, :n] (and optionally the column sums of g[:, :m]) through the HIP split-K kernel.  g / x: row-major
    fp32 matrices (or 32-column-aligned views of one); m in {128, 256}, n in {32, 64, 128, 256}; ``ranges``: float32[2] device
    tensor with upper bounds of |g| and |x| (computed here when absent - two extra passes over the data)."""
    if ranges is None:
        ranges = torch.stack([g[:, :m].abs().amax(), x[:, :n].abs().amax()]).float()
    b = _WgradBatch(g.shape[0], ranges, g)
    k = b.add(g, x, m, n, want_bias)
    b.run()
    return (b.result(k), b.bias(k)) if want_bias else b.result(k)


def weight_gradient_frag(g_frag, g_scale, x_frag, ranges, n_points, want_bias=False, x_rows=None, n=256):
    """G^T X with G a FRAGMENT slot of a gradient buffer (256 channels; ``g_scale``: the points' normalisers, see
    ``grad_frag_encode``) through the HIP split-K kernels: against ``x_frag``, a FRAGMENT slot of activations (the LDS-DMA
    kernel, 256 x 256), or, with ``x_rows`` ([n_points, >= n] fp32 rows), against row-format activations.  ``ranges``:
    float32[2] device tensor, upper bounds of the true |G| and of |X|."""
    lib = _capi.lib()
    grid = lib.inerf_wgrad_grid(n_points)
    total = 256 * n + (256 if want_bias else 0)
    buf = _new(ranges, grid, total)
    base = buf.data_ptr()
    bias = C.c_void_p(base + 4 * 256 * n) if want_bias else None
    with torch.cuda.device(ranges.device):
        if x_rows is None:
            rc = lib.inerf_mlp_weight_gradient_frag(_ptr(g_frag), _ptr(g_scale), _ptr(x_frag), _ptr(ranges), n_points, C.c_void_p(base), bias,
                                                    total, _stream(ranges))
        else:
            rc = lib.inerf_mlp_weight_gradient_gfrag(_ptr(g_frag), _ptr(g_scale), C.c_void_p(x_rows.data_ptr()), x_rows.stride(0), n_points, n,
                                                     _ptr(ranges), C.c_void_p(base), bias, total, _stream(ranges))
    _capi.check(rc, "inerf_mlp_weight_gradient_frag")
    sums = buf.sum(0)
    w = sums[:256 * n].view(256, n)
    return (w, sums[256 * n:]) if want_bias else w


def weight_gradient_frag_batch(g_frags, g_scale, x_frags, ranges, n_points, x_cols=None, g_rows=None):
    """Several products G_j^T X_j over the same points (fragment slots, shared normalisers; shapes ``g_rows[j]`` x ``x_cols[j]`` =
    256 x 256 (default), 256 x 64, 128 x 256, 128 x 32) in ONE launch of the LDS-DMA kernel, each split over its share of the
    grid (include/inerf.h inerf_mlp_weight_gradient_frag_batch).  Returns [(dW_j [rows, cols], db_j [rows])]."""
    lib = _capi.lib()
    n = len(g_frags)
    cols = [256] * n if x_cols is None else [int(c) for c in x_cols]
    grows = [256] * n if g_rows is None else [int(r) for r in g_rows]
    ccols, crows = (C.c_int * n)(*cols), (C.c_int * n)(*grows)
    rows = [lib.inerf_wgrad_frag_rows(n_points, n, crows, ccols, j) for j in range(n)]
    per = 256 * 256 + 256
    total = n * per
    buf = _new(ranges, max(rows), total)
    base = buf.data_ptr()
    arr = lambda vals: (C.c_void_p * n)(*vals)
    with torch.cuda.device(ranges.device):
        rc = lib.inerf_mlp_weight_gradient_frag_batch(n, arr([g.data_ptr() for g in g_frags]), _ptr(g_scale), arr([x.data_ptr() for x in x_frags]),
                                                      crows, ccols, _ptr(ranges), n_points, arr([base + 4 * j * per for j in range(n)]),
                                                      arr([base + 4 * (j * per + 65536) for j in range(n)]), total, _stream(ranges))
    _capi.check(rc, "inerf_mlp_weight_gradient_frag_batch")
    out = []
    for j in range(n):
        sums = buf[:rows[j], j * per:(j + 1) * per].sum(0)
        out.append((sums[:grows[j] * cols[j]].view(grows[j], cols[j]), sums[65536:65536 + grows[j]]))
    return out


def weight_gradient_xfrag(g_rows, x_frag, ranges, n_points, want_bias=False):
    """G^T X with G row-format ([n_points, >= 128] fp32: a 128-channel gradient slot) and X a FRAGMENT slot of activations
    (256 channels): the products of views_linears.0's feature columns and of the semantic hidden layer.  ``ranges[0]``: an
    upper bound of |G|."""
    lib = _capi.lib()
    grid = lib.inerf_wgrad_grid(n_points)
    m = 128
    total = m * 256 + (m if want_bias else 0)
    buf = _new(ranges, grid, total)
    base = buf.data_ptr()
    bias = C.c_void_p(base + 4 * m * 256) if want_bias else None
    with torch.cuda.device(ranges.device):
        rc = lib.inerf_mlp_weight_gradient_xfrag(C.c_void_p(g_rows.data_ptr()), g_rows.stride(0), _ptr(x_frag), n_points, m, _ptr(ranges),
                                                 C.c_void_p(base), bias, total, _stream(ranges))
    _capi.check(rc, "inerf_mlp_weight_gradient_xfrag")
    sums = buf.sum(0)
    w = sums[:m * 256].view(m, 256)
    return (w, sums[m * 256:]) if want_bias else w


def _colsum(g, nc):
    return g.sum(0) if nc == 1 else g.view(nc, g.shape[0] // nc, g.shape[1]).sum(1).sum(0)


def mlp_weight_gradients(desc, names, save, dz, d_raw, n_points, endpoint=False, heads=None):
    """dW = dZ^T X and db = column sums of dZ for every layer, from the buffers of ``encode_mlp_train`` / ``mlp_backward_inputs``
    through LIBRARY GEMMs split over K (fragment slots are decoded first): the independent restatement that the tests and
    scripts/bench_train_kernels.py hold ``mlp_backward`` - the product path: one C call, HIP kernels only - against.  The 1-4-row
    heads' gradients are taken from ``heads`` (as accumulated by the chain kernel) when given.  Returns a dict name -> gradient
    with the reference's parameter names and shapes."""
    X = save_slot_views(desc, save, n_points)
    G = save_slot_views(desc, dz, n_points, gradient=True)
    e, dv = 3 + 6 * desc.l_xyz, 3 + 6 * desc.l_dir
    sh1, sh2, res = _head_names(desc)
    sem = desc.variant == _capi.VARIANT_SSR and desc.n_classes > 0
    nc = _split_k(n_points)
    h = [X[SAVE_H0 + i] for i in range(8)]        # enc / dir: zero-padded columns, cut from the products
    # (G slot, X slot): the products with 128 / 256 rows
    big = {"t0": (SAVE_H0, SAVE_ENC), "t5e": (SAVE_H0 + 5, SAVE_ENC), "as1": (SAVE_AS1H, SAVE_H0 + 7), "feat": (SAVE_FEAT, SAVE_H0 + 7),
           "vf": (SAVE_VH, SAVE_FEAT), "vd": (SAVE_VH, SAVE_DIR)}
    for i in range(1, 8):
        big[f"t{i}"] = (SAVE_H0 + i, SAVE_H0 + i - 1)
    if sem:
        big["sem1"] = (SAVE_SEMH, SAVE_H0 + 7)
    W, B = {}, {}
    for k, (gs, xs) in big.items():
        W[k] = _tn(G[gs], X[xs], nc)
        if gs not in B:
            B[gs] = _colsum(G[gs], nc)
    out = {}

    def lin(name, g, x):
        out[name + ".weight"] = _tn(g, x, nc)
        out[name + ".bias"] = _colsum(g, nc)

    out["pts_linears.0.weight"] = W["t0"][:, :e]
    for i in range(1, 8):
        out[f"pts_linears.{i}.weight"] = torch.cat([W["t5e"][:, :e], W["t5"]], 1) if i == 5 else W[f"t{i}"]   # cat([pts, h]), helpers:290-291
    for i in range(8):
        out[f"pts_linears.{i}.bias"] = B[SAVE_H0 + i]
    out["albedo_linear1.weight"], out["albedo_linear1.bias"] = W["as1"][0:128], B[SAVE_AS1H][0:128]
    out[sh1 + ".weight"], out[sh1 + ".bias"] = W["as1"][128:256], B[SAVE_AS1H][128:256]
    out["feature_linear.weight"], out["feature_linear.bias"] = W["feat"], B[SAVE_FEAT]
    out["views_linears.0.weight"], out["views_linears.0.bias"] = torch.cat([W["vf"], W["vd"][:, :dv]], 1), B[SAVE_VH]
    dpre = G[SAVE_DPRE]
    if heads is None:
        lin("alpha_linear", dpre[:, 7:8], h[7])
    if sem:
        out["semantic_linear.0.0.weight"], out["semantic_linear.0.0.bias"] = W["sem1"], B[SAVE_SEMH]
        c = desc.n_classes
        lin("semantic_linear.1", d_raw[:, BASE_CHANNELS:BASE_CHANNELS + c], X[SAVE_SEMH])
    if heads is None:
        as1h = X[SAVE_AS1H]
        lin("albedo_linear2", dpre[:, 0:3], as1h[:, :128])
        lin(sh2, dpre[:, 3:4], as1h[:, 128:])
        lin(res, dpre[:, 4:7], X[SAVE_VH])
    else:                      # accumulated by the chain kernel itself (layout: include/inerf.h)
        as2 = heads[384:1408].view(4, 256)
        out[res + ".weight"], out[res + ".bias"] = heads[0:384].view(3, 128), heads[1668:1671]
        out["albedo_linear2.weight"], out["albedo_linear2.bias"] = as2[0:3, 0:128], heads[1664:1667]
        out[sh2 + ".weight"], out[sh2 + ".bias"] = as2[3:4, 128:256], heads[1667:1668]
        out["alpha_linear.weight"], out["alpha_linear.bias"] = heads[1408:1664].view(1, 256), heads[1671:1672]
    return {k: out[k] for k in names}


class _FusedMlpFn(torch.autograd.Function):
    """raw = MLP(encode(o + d z), encode(viewdir)) with a HIP forward AND backward: fused split-f16 forward that keeps the
    activations, MFMA input-gradient chain, split-K MFMA weight gradients.  Parameters are re-packed on the device
    (packing.DevicePacker).  rays / z get no gradient (as in the reference: rays are data, resampled depths are detached)."""

    @staticmethod
    def forward(ctx, rays, z_vals, desc, endpoint, names, exact, *params):
        from . import packing
        named = dict(zip(names, params))
        status = _new_status(rays)
        act_max = torch.zeros(1, dtype=torch.float32, device=rays.device)
        packed = packing.device_packer(desc, False, rays.device)(named)
        raw, save = encode_mlp_train(desc, packed, rays.detach(), z_vals.detach(), endpoint, status, act_max)
        if exact:
            # INERF_PRECISION=f32: the values that leave the node come from the exact-fp32 MFMA kernel (bit for bit what a
            # no_grad render returns); the split-precision forward above is kept for what it SAVES - activations, ReLU masks,
            # operand ranges - which is what the backward kernels consume (22-bit operands, fp32 accumulation: within 1e-5 of
            # fp64 autograd).  The chain takes its sigmoid' from the exact outputs.
            d32 = _capi.NetDesc(desc.variant, desc.n_classes, desc.l_xyz, desc.l_dir, desc.xyz_div, _capi.PREC_F32)
            raw = encode_mlp(d32, packing.device_packer_f32(desc, rays.device)(named), rays.detach(), z_vals.detach(), endpoint)
        # the transposed blob of the backward pass is packed NOW, behind the forward kernel in the queue: at the start of the
        # backward the queue is empty, and its ~25 small launches would each cost a host round trip of GPU idle time
        packed_bwd = packing.device_packer(desc, True, rays.device)(named)
        check_f16_range(status, "training forward", deferrable=True)      # the front-ends read both networks' words once per batch
        ctx.packed_bwd = packed_bwd
        ctx.save_for_backward(raw, save, act_max, *params)
        ctx.cfg = (desc, endpoint, names)
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        desc, endpoint, names = ctx.cfg
        raw, save, act_max = ctx.saved_tensors[:3]
        n, s, ch = raw.shape
        status = _new_status(raw)
        d2 = d_raw.contiguous().view(n * s, ch).float()
        # one C call: chain + weight gradients + reduction + scatter (inerf_mlp_backward)
        flat = mlp_backward(desc, ctx.packed_bwd, raw.view(n * s, ch), d2, save, act_max, endpoint, status)
        grads = param_views(desc, flat)
        check_f16_range(status, "training backward")      # read AFTER everything is enqueued: the GPU works through it meanwhile
        return (None, None, None, None, None, None) + tuple(grads[k] for k in names)


TRAIN_POINTS_PER_NODE = 2 * 1024 * 1024


def mlp_train(desc, module, rays, z_vals, endpoint=False):
    """Differentiable fused network evaluation for a training step: ``module``'s parameters receive gradients.
    ``desc.precision`` = INERF_PREC_F32: the forward values come from the exact-fp32 kernel (see _FusedMlpFn.forward)."""
    from . import packing
    exact = desc.precision == _capi.PREC_F32
    d = _capi.NetDesc(desc.variant, desc.n_classes, desc.l_xyz, desc.l_dir, desc.xyz_div, _capi.PREC_F16X3)
    names = tuple(name for name, _ in packing.tensor_table(d))
    named = dict(module.named_parameters())
    params = [named[k] for k in names]
    n, s = z_vals.shape
    per = max(1, TRAIN_POINTS_PER_NODE // s)        # the library keeps 11 KB per point; one node stays below its 4 M-point limit
    if n <= per:
        return _FusedMlpFn.apply(rays, z_vals, d, bool(endpoint), names, exact, *params)
    return torch.cat([_FusedMlpFn.apply(rays[i:i + per], z_vals[i:i + per], d, bool(endpoint), names, exact, *params)
                      for i in range(0, n, per)], 0)
