"""Layer-by-layer evaluation of NeRF / Semantic_NeRF on the fp32 matrix core (``csrc/layered.hip``).

The fused kernels implement the architecture every shipped config uses (D=8, W=256, skips=[4]).  The reference builds its
networks from flags (``NeRF(D=args.netdepth, W=args.netwidth, ...)``, object_level/run_nerf.py:286-296;
SSR/training/trainer.py:811-846); any other depth / width / skip list - and ``use_viewdirs=False`` - goes through here: each
``nn.Linear`` of ``NeRF.forward`` / ``Semantic_NeRF.forward`` (run_nerf_helpers.py:284-321, semantic_nerf.py:120-181) is one
``inerf_linear`` launch (exact fp32 MFMA), the concatenations are column ranges of one buffer, and the backward pass
(what ``loss.backward()`` records for the network, run_nerf.py:1018) is the same kernel on other strides plus
``inerf_linear_wgrad``.  A training batch whose activations leave the f16 range of the split-precision kernels is evaluated
here too, so no ATen GEMM remains on the render / training path.

This file is plumbing: which buffer feeds which launch.  No arithmetic happens in torch.
"""
import ctypes as C
import os

import torch

from . import _capi
from .kernels import _dev, _stream

__all__ = ["spec_for", "evaluate", "evaluate_points", "POINTS_PER_PASS"]

POINTS_PER_PASS = 1 << 19          # inference: sample points per pass (activations are ~14 buffers of W floats per point)


class Cols:
    """Columns ``off .. off + width`` of a contiguous fp32 matrix ``buf[n, ld]`` (a pointer + a leading dimension)."""
    __slots__ = ("buf", "off", "width")

    def __init__(self, buf, off=0, width=None):
        self.buf, self.off = buf, off
        self.width = buf.shape[1] - off if width is None else width

    @property
    def ptr(self):
        return self.buf.data_ptr() + 4 * self.off

    @property
    def ld(self):
        return self.buf.shape[1]


class Spec:
    """What the launches need to know about one network: layer names and shapes, read off the module."""

    def __init__(self, module, l_xyz, xyz_div, l_dir):
        self.l_xyz, self.xyz_div, self.l_dir = l_xyz, xyz_div, l_dir
        self.depth = len(module.pts_linears)
        self.width = module.pts_linears[0].out_features
        self.in_xyz = module.pts_linears[0].in_features
        self.skips = tuple(int(i) for i in module.skips)
        self.use_viewdirs = bool(module.use_viewdirs)
        self.n_views = len(module.views_linears)
        self.in_dir = module.views_linears[0].in_features - self.width if self.use_viewdirs else 0
        ssr = hasattr(module, "residual_linear")
        self.residual = "residual_linear" if ssr else "shading_linear"            # run_nerf_helpers.py:268: 'shading_linear' is the residual head
        self.shading = ("shading_linear1", "shading_linear2") if ssr else ("test_linear1", "test_linear2")
        self.n_classes = int(module.semantic_linear[1].out_features) if hasattr(module, "semantic_linear") else 0
        self.out_ch = 0 if self.use_viewdirs else module.output_linear.out_features
        self.endpoint_dim = module.views_linears[-1].out_features if self.use_viewdirs else 0

    def channels(self, endpoint):
        if not self.use_viewdirs:
            return self.out_ch
        return _capi.BASE_CHANNELS + self.n_classes + (self.endpoint_dim if endpoint else 0)

    def names(self):
        """Parameters the forward reads, in launch order (``name.weight`` / ``name.bias`` of each)."""
        out = [f"pts_linears.{i}" for i in range(self.depth)]
        if not self.use_viewdirs:
            return out + ["output_linear"]
        out += ["alpha_linear", "albedo_linear1", "albedo_linear2", self.shading[0], self.shading[1]]
        if self.n_classes:
            out += ["semantic_linear.0.0", "semantic_linear.1"]
        return out + ["feature_linear"] + [f"views_linears.{j}" for j in range(self.n_views)] + [self.residual]


def enabled():
    """``INERF_LAYERED=0`` sends networks outside the fused architecture through their torch ``forward`` (debugging)."""
    return os.environ.get("INERF_LAYERED", "1") != "0"


def spec_for(module, embed_fn, embeddirs_fn):
    """Spec if ``module`` is a NeRF / Semantic_NeRF (this package's mirrors or classes with the same attributes) fed by
    this package's Embedders, else None (a foreign callable is called as the reference calls it)."""
    from .object_level import Embedder
    if not enabled() or not isinstance(module, torch.nn.Module):
        return None
    for attr in ("pts_linears", "views_linears", "skips", "use_viewdirs"):
        if not hasattr(module, attr):
            return None
    if not isinstance(embed_fn, Embedder) or embed_fn.input_dims != 3:
        return None
    depth = len(module.pts_linears)
    if depth < 1 or any(not isinstance(m, torch.nn.Linear) for m in module.pts_linears):
        return None
    if (depth - 1) in module.skips or any(not 0 <= int(i) < depth for i in module.skips):
        return None                      # the heads are Linear(W, .): the reference itself cannot run a skip after the last layer
    if module.pts_linears[0].in_features != embed_fn.out_dim:
        return None
    l_dir = 0
    if module.use_viewdirs:
        if not isinstance(embeddirs_fn, Embedder) or embeddirs_fn.scalar_factor != 1.0 or embeddirs_fn.input_dims != 3:
            return None
        if module.views_linears[0].in_features != module.pts_linears[0].out_features + embeddirs_fn.out_dim:
            return None
        l_dir = embeddirs_fn.n_freqs
        for attr in ("feature_linear", "alpha_linear", "albedo_linear1", "albedo_linear2"):
            if not hasattr(module, attr):
                return None
    elif not hasattr(module, "output_linear"):
        return None
    return Spec(module, embed_fn.n_freqs, float(embed_fn.scalar_factor), l_dir)


# ----------------------------------------------------------------------------------------------
# launches
# ----------------------------------------------------------------------------------------------
def _p(v):
    return None if v is None else C.c_void_p(v)


def _linear(a_ptr, a_sm, a_sk, b_ptr, b_sn, b_sk, out, m, n, k, like, bias=None, add=None, gate=None, act=_capi.ACT_NONE):
    args = _capi.LinearArgs(_p(a_ptr), a_sm, a_sk, _p(b_ptr), b_sn, b_sk, _p(None if bias is None else bias.data_ptr()),
                            _p(None if add is None else add.ptr), 0 if add is None else add.ld,
                            _p(None if gate is None else gate.ptr), 0 if gate is None else gate.ld,
                            _p(out.ptr), out.ld, m, n, act, k)
    _capi.check(_capi.lib().inerf_linear(C.byref(args), _stream(like)), "inerf_linear")


def linear(x, weight, bias, out, act=_capi.ACT_NONE):
    """``out = act(x @ weight.T + bias)`` - one nn.Linear (+ F.relu / torch.sigmoid) of the reference's forward."""
    o, k = weight.shape
    assert x.width == k and out.width == o, (x.width, k, out.width, o)
    _linear(x.ptr, x.ld, 1, weight.data_ptr(), k, 1, out, x.buf.shape[0], o, k, x.buf, bias=bias, act=act)


def linear_dgrad(dz, weight, out, col0=0, add=None, gate=None):
    """``out = (dz @ weight[:, col0 : col0 + out.width]) [+ add]``, zeroed where ``gate <= 0`` (the producing layer's ReLU)."""
    o, k = weight.shape
    assert dz.width == o and col0 + out.width <= k
    _linear(dz.ptr, dz.ld, 1, weight.data_ptr() + 4 * col0, 1, k, out, dz.buf.shape[0], out.width, o, dz.buf, add=add, gate=gate)


def linear_wgrad(dz, x):
    """(d_weight [dz.width, x.width], d_bias [dz.width]) of a layer with output gradient ``dz`` and input ``x``."""
    n, rows, cols = dz.buf.shape[0], dz.width, x.width
    dev = dz.buf.device
    if n == 0:                                   # an empty batch: the sums are empty
        return torch.zeros(rows, cols, dtype=torch.float32, device=dev), torch.zeros(rows, dtype=torch.float32, device=dev)
    dw = torch.empty(rows, cols, dtype=torch.float32, device=dev)
    db = torch.empty(rows, dtype=torch.float32, device=dev)
    lib = _capi.lib()
    nbytes = int(lib.inerf_linear_wgrad_workspace_bytes(n, rows, cols))
    if nbytes < 0:
        _capi.check(nbytes, "inerf_linear_wgrad_workspace_bytes")
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
    rc = lib.inerf_linear_wgrad(_p(dz.ptr), dz.ld, rows, _p(x.ptr), x.ld, cols, n, _p(dw.data_ptr()), _p(db.data_ptr()), 0,
                                _p(ws.data_ptr()), nbytes, _stream(dz.buf))
    _capi.check(rc, "inerf_linear_wgrad")
    return dw, db


class RaySource:
    """Embeddings computed by ``inerf_embed`` from rays + depths (run_nerf.py:488, run_nerf_helpers.py:195-225)."""

    def __init__(self, spec, rays, z_vals):
        self.spec, self.rays, self.z = spec, rays, z_vals
        self.n_rays, self.n_samples = z_vals.shape
        self.n = self.n_rays * self.n_samples

    def _embed(self, out, n_freqs, div, directions):
        assert out.width == 3 + 6 * n_freqs
        rc = _capi.lib().inerf_embed(_p(self.rays.data_ptr()), _p(self.z.data_ptr()), self.n_rays, self.n_samples, n_freqs, div,
                                     directions, _p(out.ptr), out.ld, _stream(self.rays))
        _capi.check(rc, "inerf_embed")

    def xyz_into(self, out):
        self._embed(out, self.spec.l_xyz, self.spec.xyz_div, 0)

    def dir_into(self, out):
        self._embed(out, self.spec.l_dir, 1.0, 1)


def _combine(raw):
    rc = _capi.lib().inerf_intrinsic_combine(_p(raw.data_ptr()), raw.shape[1], raw.shape[0], _stream(raw))
    _capi.check(rc, "inerf_intrinsic_combine")


def _combine_backward(raw, d_raw):
    dz = torch.empty(raw.shape[0], 8, dtype=torch.float32, device=raw.device)
    rc = _capi.lib().inerf_intrinsic_combine_backward(_p(raw.data_ptr()), _p(d_raw.data_ptr()), raw.shape[1], raw.shape[0],
                                                      _p(dz.data_ptr()), _stream(raw))
    _capi.check(rc, "inerf_intrinsic_combine_backward")
    return dz


# ----------------------------------------------------------------------------------------------
# the two forwards (run_nerf_helpers.py:284-321, semantic_nerf.py:120-181) and their backward
# ----------------------------------------------------------------------------------------------
def _forward(spec, P, src, endpoint, keep, raw_out=None):
    """raw [n, CH] (written into ``raw_out`` if given) and, with ``keep``, the buffers the backward reads."""
    n, dev = src.n, src.rays.device
    W, kx, kd, R, S = spec.width, spec.in_xyz, spec.in_dir, _capi.ACT_RELU, _capi.ACT_SIGMOID

    def new(cols):
        return torch.empty(n, cols, dtype=torch.float32, device=dev)

    def lin(name, x, out, act=_capi.ACT_NONE):
        linear(x, P[name + ".weight"], P[name + ".bias"], out, act)

    saved = {"trunk": [], "views": []}
    cur = Cols(new(kx))
    src.xyz_into(cur)
    out = None
    for i in range(spec.depth):
        if i in spec.skips:                                   # h = torch.cat([input_pts, h], -1): one buffer, two writers
            buf = new(kx + W)
            src.xyz_into(Cols(buf, 0, kx))
            out, nxt = Cols(buf, kx, W), Cols(buf, 0, kx + W)
        else:
            out = nxt = Cols(new(W))
        lin(f"pts_linears.{i}", cur, out, R)
        if keep:
            saved["trunk"].append((cur, out))
        cur = nxt
    h = out
    ch = spec.channels(endpoint)
    raw = new(ch) if raw_out is None else raw_out
    assert raw.shape == (n, ch) and raw.is_contiguous()
    if not spec.use_viewdirs:
        lin("output_linear", h, Cols(raw))
        saved["raw"] = raw
        return raw, saved
    C0 = _capi.BASE_CHANNELS
    lin("alpha_linear", h, Cols(raw, 3, 1))
    a1 = Cols(new(P["albedo_linear1.weight"].shape[0]))
    lin("albedo_linear1", h, a1, R)
    lin("albedo_linear2", a1, Cols(raw, 4, 3), S)
    s1 = Cols(new(P[spec.shading[0] + ".weight"].shape[0]))
    lin(spec.shading[0], h, s1, R)
    lin(spec.shading[1], s1, Cols(raw, 7, 1), S)
    m1 = None
    if spec.n_classes:
        m1 = Cols(new(P["semantic_linear.0.0.weight"].shape[0]))
        lin("semantic_linear.0.0", h, m1, R)
        lin("semantic_linear.1", m1, Cols(raw, C0, spec.n_classes))
    vbuf = new(W + kd)                                          # h = torch.cat([feature, input_views], -1)
    lin("feature_linear", h, Cols(vbuf, 0, W))
    src.dir_into(Cols(vbuf, W, kd))
    v = Cols(vbuf)
    for j in range(spec.n_views):
        last = j == spec.n_views - 1
        wv = P[f"views_linears.{j}.weight"].shape[0]
        vo = Cols(raw, C0 + spec.n_classes, wv) if (last and endpoint) else Cols(new(wv))      # endpoint_feat = h (semantic_nerf.py:163)
        lin(f"views_linears.{j}", v, vo, R)
        if keep:
            saved["views"].append((v, vo))
        v = vo
    lin(spec.residual, v, Cols(raw, 8, 3), S)
    _combine(raw)
    if keep:
        saved.update(h=h, a1=a1, s1=s1, m1=m1, raw=raw)
    return raw, saved


def _backward(spec, P, saved, d_raw, endpoint):
    """name -> gradient for every parameter of ``spec.names()``, given d_raw [n, CH]."""
    raw = saved["raw"]
    n, dev = raw.shape[0], raw.device
    W, G = spec.width, {}

    def new(cols):
        return Cols(torch.empty(n, cols, dtype=torch.float32, device=dev))

    def wg(name, g, x):
        G[name + ".weight"], G[name + ".bias"] = linear_wgrad(g, x)

    if not spec.use_viewdirs:
        x_in, h = saved["trunk"][-1]
        g = Cols(d_raw)
        wg("output_linear", g, h)
        dz = new(W)
        linear_dgrad(g, P["output_linear.weight"], dz, gate=h)
    else:
        C0, nc = _capi.BASE_CHANNELS, spec.n_classes
        h, a1, s1, m1 = saved["h"], saved["a1"], saved["s1"], saved["m1"]
        dzh = _combine_backward(raw, d_raw)                     # pre-sigmoid gradients: albedo 0:3, shading 3, residual 4:7
        g_alb, g_sh, g_res = Cols(dzh, 0, 3), Cols(dzh, 3, 1), Cols(dzh, 4, 3)
        # view branch: residual head <- views_linears <- [feature | dirs]
        v_last = saved["views"][-1][1]
        wg(spec.residual, g_res, v_last)
        dv = new(v_last.width)
        linear_dgrad(g_res, P[spec.residual + ".weight"], dv, gate=v_last,
                     add=Cols(d_raw, C0 + nc, v_last.width) if endpoint else None)
        d_feat = None
        for j in reversed(range(spec.n_views)):
            v_in, _ = saved["views"][j]
            name = f"views_linears.{j}"
            wg(name, dv, v_in)
            if j > 0:
                nxt = new(v_in.width)
                linear_dgrad(dv, P[name + ".weight"], nxt, gate=v_in)
                dv = nxt
            else:
                d_feat = new(W)
                linear_dgrad(dv, P[name + ".weight"], d_feat)          # columns 0..W of the cat: the feature (no activation)
        wg("feature_linear", d_feat, h)
        # dH = sum of the heads' input gradients; the LAST launch applies h's ReLU mask
        dz = new(W)
        linear_dgrad(d_feat, P["feature_linear.weight"], dz)
        g_alpha = Cols(d_raw, 3, 1)
        wg("alpha_linear", g_alpha, h)
        linear_dgrad(g_alpha, P["alpha_linear.weight"], dz, add=dz)
        branches = [("albedo_linear1", "albedo_linear2", g_alb, a1), (spec.shading[0], spec.shading[1], g_sh, s1)]
        if nc:
            branches.append(("semantic_linear.0.0", "semantic_linear.1", Cols(d_raw, C0, nc), m1))
        for bi, (first, second, g_out, hidden) in enumerate(branches):
            wg(second, g_out, hidden)
            dhid = new(hidden.width)
            linear_dgrad(g_out, P[second + ".weight"], dhid, gate=hidden)
            wg(first, dhid, h)
            linear_dgrad(dhid, P[first + ".weight"], dz, add=dz, gate=h if bi == len(branches) - 1 else None)
    for i in reversed(range(spec.depth)):
        x_in, _ = saved["trunk"][i]
        name = f"pts_linears.{i}"
        wg(name, dz, x_in)
        if i > 0:
            prev_out = saved["trunk"][i - 1][1]
            nxt = new(W)
            linear_dgrad(dz, P[name + ".weight"], nxt, col0=x_in.width - prev_out.width, gate=prev_out)
            dz = nxt
    return G


def _params(spec, module):
    named = dict(module.named_parameters())
    names = [f"{k}.{s}" for k in spec.names() for s in ("weight", "bias")]
    return names, [named[k] for k in names]


class _LayeredFn(torch.autograd.Function):
    """raw = network(embed(o + d z), embed(viewdir)) with HIP forward and backward, any depth / width / skips."""

    @staticmethod
    def forward(ctx, rays, z_vals, spec, endpoint, names, *params):
        P = {k: _dev(p.detach(), k) for k, p in zip(names, params)}
        raw, saved = _forward(spec, P, RaySource(spec, rays, z_vals), endpoint, keep=True)
        ctx.cfg, ctx.P, ctx.kept = (spec, endpoint, names, tuple(z_vals.shape)), P, saved
        return raw.view(z_vals.shape[0], z_vals.shape[1], raw.shape[1])

    @staticmethod
    def backward(ctx, d_raw):
        spec, endpoint, names, (n, s) = ctx.cfg
        d2 = _dev(d_raw, "d_raw").view(n * s, d_raw.shape[-1])
        G = _backward(spec, ctx.P, ctx.kept, d2, endpoint)
        ctx.kept = ctx.P = None
        return (None, None, None, None, None) + tuple(G[k] for k in names)


def evaluate(spec, module, rays, z_vals, endpoint=False):
    """raw[N, S, CH] of ``module`` on the sample points ``o + d z`` of ``rays`` - differentiable w.r.t. the module's
    parameters when autograd is recording (rays / depths get no gradient, as in the reference)."""
    rays = _dev(rays, "rays", (None, _capi.RAY_FLOATS))
    z_vals = _dev(z_vals, "z_vals", (rays.shape[0], None))
    names, params = _params(spec, module)
    n, s = z_vals.shape
    with torch.cuda.device(rays.device):
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _LayeredFn.apply(rays, z_vals, spec, bool(endpoint), tuple(names), *params)
        P = {k: _dev(p.detach(), k) for k, p in zip(names, params)}
        per = max(1, POINTS_PER_PASS // s)
        raw = torch.empty(n * s, spec.channels(endpoint), dtype=torch.float32, device=rays.device)
        for i in range(0, n, per):             # activations of one pass at a time; raw rows land in place
            _forward(spec, P, RaySource(spec, rays[i:i + per], z_vals[i:i + per]), endpoint, keep=False, raw_out=raw[i * s:(i + per) * s])
    return raw.view(n, s, raw.shape[1])


def evaluate_points(spec, module, inputs, viewdirs, endpoint=False):
    """``run_network`` on arbitrary points (run_nerf.py:42-56): ``inputs[..., 3]`` with one view direction per leading row."""
    pts = torch.reshape(inputs, [-1, 3]).float()
    rays = torch.zeros(pts.shape[0], _capi.RAY_FLOATS, dtype=torch.float32, device=pts.device)
    rays[:, 0:3] = pts                                           # one "ray" per point: o = point, d = 0, depth 0 -> o + 0 * 0 = o
    if viewdirs is not None:
        rays[:, 8:11] = torch.reshape(viewdirs[:, None].expand(inputs.shape), [-1, 3]).float()
    z = torch.zeros(pts.shape[0], 1, dtype=torch.float32, device=pts.device)
    raw = evaluate(spec, module, rays, z, endpoint)
    return torch.reshape(raw, list(inputs.shape[:-1]) + [raw.shape[-1]])
