"""Drop-in front-end for the scene-level code base (``SSR/`` + ``train_SSR_main.py``).

Mirrors, with identical names / argument meaning / returned keys (citations relative to
``/root/reference/SSR``):
  Semantic_NeRF            models/semantic_nerf.py:74-181     get_embedder   models/semantic_nerf.py:50-65
  run_network              models/model_utils.py:19-35        raw2outputs    models/model_utils.py:39-116
  sample_pdf               models/rays.py:176-220             create_rays    models/rays.py:223-256
  batchify_rays            training/training_utils.py:5-17
  SSRRenderMixin.render_rays / volumetric_rendering / create_ssr
                           training/trainer.py:693-715 / 717-808 / 811-846
``SSRTrainer`` keeps its data loading, losses and logging; it only has to inherit the mixin (or
assign the three methods) - see INTEGRATION.md.  All arithmetic runs in ``libinerf.so``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _capi, kernels, layered, packing
from .object_level import (FP32_LAYERS_NOTE, Embedder, _layered_spec, _run_network_torch, _train_desc, _train_query, _training_path_notice,
                           _wants_grad)

__all__ = ["get_embedder", "Semantic_NeRF", "run_network", "raw2outputs", "sample_pdf", "create_rays",
           "get_rays_camera", "get_rays_world", "batchify_rays", "SSRRenderMixin", "SSRRenderer"]


def get_embedder(multires, i=0, scalar_factor=1):
    """(embed_fn, out_dim); the input is divided by ``scalar_factor`` first - semantic_nerf.py:50-65."""
    if i == -1:
        return nn.Identity(), 3
    e = Embedder(multires, scalar_factor=scalar_factor)
    return e, e.out_dim


def fc_block(in_f, out_f):
    return nn.Sequential(nn.Linear(in_f, out_f), nn.ReLU(out_f))


class Semantic_NeRF(nn.Module):
    """Intrinsic NeRF + position-only semantic head, reference parameter names (semantic_nerf.py:98-118).

    ``semantic_linear.0.0`` / ``semantic_linear.1``, ``residual_linear``, ``albedo_linear1/2``,
    ``shading_linear1/2``.  As for the object-level module, ``forward`` defines the network in torch
    ops for holders of an embedded tensor; the render path uses the fused kernel instead.
    """

    def __init__(self, enable_semantic, num_semantic_classes, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4,
                 skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.skips = list(skips)
        self.use_viewdirs = use_viewdirs
        self.enable_semantic = enable_semantic
        self.num_semantic_classes = num_semantic_classes if enable_semantic else 0
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] +
            [nn.Linear(W + input_ch, W) if i in self.skips else nn.Linear(W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            if enable_semantic:
                self.semantic_linear = nn.Sequential(fc_block(W, W // 2), nn.Linear(W // 2, num_semantic_classes))
            self.residual_linear = nn.Linear(W // 2, 3)
            self.albedo_linear1 = nn.Linear(W, W // 2)
            self.albedo_linear2 = nn.Linear(W // 2, 3)
            self.shading_linear1 = nn.Linear(W, W // 2)
            self.shading_linear2 = nn.Linear(W // 2, 1)
        else:
            self.output_linear = nn.Linear(W, output_ch)

    def fused_desc(self):
        if not (self.use_viewdirs and self.D == 8 and self.W == 256 and self.skips == [4]):
            return None
        l_xyz, rx = divmod(self.input_ch - 3, 6)
        l_dir, rd = divmod(self.input_ch_views - 3, 6)
        if rx or rd or not (0 <= l_xyz <= 10 and 0 <= l_dir <= 4) or self.num_semantic_classes > _capi.MAX_CLASSES:
            return None
        return _capi.net_desc(_capi.VARIANT_SSR, self.num_semantic_classes, l_xyz, l_dir, 1.0)

    def forward(self, x, show_endpoint=False):
        pts, views = torch.split(x, [self.input_ch, self.input_ch_views], dim=-1)
        h = pts
        for i, layer in enumerate(self.pts_linears):
            h = F.relu(layer(h))
            if i in self.skips:
                h = torch.cat([pts, h], -1)
        if not self.use_viewdirs:
            out = self.output_linear(h)
            return out
        sigma = self.alpha_linear(h)
        sem = self.semantic_linear(h) if self.enable_semantic else None
        albedo = torch.sigmoid(self.albedo_linear2(F.relu(self.albedo_linear1(h))))
        shading = torch.sigmoid(self.shading_linear2(F.relu(self.shading_linear1(h))))
        v = torch.cat([self.feature_linear(h), views], -1)
        for layer in self.views_linears:
            v = F.relu(layer(v))
        residual = torch.sigmoid(self.residual_linear(v))
        rgb = albedo * shading + residual
        parts = [rgb, sigma, albedo, shading, residual] + ([sem] if sem is not None else [])
        if show_endpoint:
            parts.append(v)
        return torch.cat(parts, -1)


_told_unfused = False


def _unfused_notice(what):
    """Said once per process (the object-level front-end's staged branch does the same for foreign networks)."""
    global _told_unfused
    if not _told_unfused:
        import warnings
        warnings.warn(f"{what}: network / encoders outside the fused kernels' architecture (D=8, W=256, skips=[4], multires<=10, "
                      "multires_views<=4): HIP sampling and compositing, the networks layer by layer on the fp32 MFMA kernels "
                      "(a foreign callable: through its own forward).")
        _told_unfused = True


def _fusable(fn, embed_fn, embeddirs_fn):
    if not hasattr(fn, "fused_desc"):
        return None
    desc = fn.fused_desc()
    if desc is None or not isinstance(embed_fn, Embedder) or not isinstance(embeddirs_fn, Embedder):
        return None
    if embed_fn.n_freqs != desc.l_xyz or embeddirs_fn.n_freqs != desc.l_dir or embeddirs_fn.scalar_factor != 1.0:
        return None
    desc.xyz_div = embed_fn.scalar_factor
    return desc


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64, show_endpoint=False):
    """Encode + apply the network - model_utils.py:19-35.  ``fn`` is a Semantic_NeRF (fused HIP launch) or
    any callable on the embedded tensor (called as the reference does).  ``show_endpoint`` is what the
    reference expresses as ``lambda x: net(x, self.endpoint_feat)`` (trainer.py:770)."""
    desc = _fusable(fn, embed_fn, embeddirs_fn) if viewdirs is not None else None
    if desc is None or _wants_grad(fn):
        # another netdepth / netwidth (trainer.py:811-846 builds what the YAML says), or gradients for arbitrary points: layer by
        # layer on the fp32 MFMA kernels (layered.py).  Any other callable is called as the reference calls it (model_utils.py:19-35)
        if desc is not None:
            _training_path_notice("run_network")
        spec = _layered_spec(fn, embed_fn, embeddirs_fn, viewdirs is not None) if inputs.is_cuda else None
        if spec is not None:
            return layered.evaluate_points(spec, fn, inputs, viewdirs, endpoint=show_endpoint)
        call = (lambda x: fn(x, True)) if show_endpoint else fn          # trainer.py:770
        return _run_network_torch(inputs, viewdirs, call, embed_fn, embeddirs_fn, netchunk)
    pts = torch.reshape(inputs, [-1, 3]).float()
    dirs = torch.reshape(viewdirs[:, None].expand(inputs.shape), [-1, 3]).float()
    rays = torch.zeros(pts.shape[0], _capi.RAY_FLOATS, dtype=torch.float32, device=pts.device)
    rays[:, 0:3], rays[:, 8:11] = pts, dirs
    z = torch.zeros(pts.shape[0], 1, dtype=torch.float32, device=pts.device)

    def run(d):
        status = kernels._new_status(pts) if d.precision == _capi.PREC_F16X3 else None
        out = kernels.encode_mlp(d, packing.packed_for_module(fn, d, pts.device), rays, z, endpoint=show_endpoint,
                                 status=status)
        kernels.check_f16_range(status, "run_network")
        return out

    raw = kernels.with_f32_fallback(desc, run)
    return torch.reshape(raw, list(inputs.shape[:-1]) + [raw.shape[-1]])


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, enable_semantic=True, num_sem_class=0,
                endpoint_feat=False):
    """Compositing incl. semantic logits / endpoint feature - model_utils.py:39-116.

    Returns ``(rgb, disp, acc, weights, depth, sem, feat, albedo, shading, residual)``; ``sem`` / ``feat``
    are ``torch.tensor(0)`` when disabled, as in the reference.
    """
    if enable_semantic:
        assert num_sem_class > 0
    noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std if raw_noise_std > 0. else None
    o = kernels.composite(raw.float(), z_vals.float(), rays_d.float(), noise, white_bkgd,
                          n_classes=num_sem_class if enable_semantic else 0, feat_dim=128 if endpoint_feat else 0)
    sem = o["sem"] if enable_semantic else torch.tensor(0)
    feat = o["feat"] if endpoint_feat else torch.tensor(0)
    return o["rgb"], o["disp"], o["acc"], o["weights"], o["depth"], sem, feat, o["albedo"], o["shading"], o["residual"]


def sample_pdf(bins, weights, N_samples, det=False):
    """Inverse-CDF sampling - rays.py:176-220."""
    b2 = torch.reshape(bins, [-1, bins.shape[-1]]).float()
    w2 = torch.reshape(weights, [-1, weights.shape[-1]]).float()
    if det:
        u = torch.linspace(0., 1., steps=N_samples, device=bins.device)
    else:
        u = torch.rand(b2.shape[0], N_samples, device=bins.device)
    return torch.reshape(kernels.sample_pdf(b2, w2, u, N_samples), list(bins.shape[:-1]) + [N_samples])


# ----------------------------------------------------------------------------------------------
# ray generation (input producer; rays.py:27-67,223-256)
# ----------------------------------------------------------------------------------------------
def get_rays_camera(B, H, W, fx, fy, cx, cy, depth_type="z", convention="opencv"):
    assert depth_type in ("z", "euclidean")
    j, i = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    i = i.float()[None].expand(B, H, W)
    j = j.float()[None].expand(B, H, W)
    if convention == "opencv":
        dirs = torch.stack(((i - cx) / fx, (j - cy) / fy, torch.ones(B, H, W)), dim=3)
    elif convention == "opengl":
        dirs = torch.stack(((i - cx) / fx, -(j - cy) / fy, -torch.ones(B, H, W)), dim=3)
    else:
        raise AssertionError(convention)
    if depth_type == "euclidean":
        dirs = dirs * (1. / torch.norm(dirs, dim=3, keepdim=True))
    return dirs


def get_rays_world(T_WC, dirs_C):
    R_WC = T_WC[:, :3, :3]
    dirs_W = torch.matmul(R_WC[:, None, ...], dirs_C[..., None]).squeeze(-1)
    origins = torch.broadcast_tensors(T_WC[:, :3, -1][:, None, :], dirs_W)[0]
    return origins, dirs_W


def create_rays(num_rays, Ts_c2w, height, width, fx, fy, cx, cy, near, far, c2w_staticcam=None, depth_type="z",
                use_viewdirs=True, convention="opencv"):
    """[num_images, H*W, 11] ray batch ``[o3, d3, near, far, viewdir3]`` - rays.py:223-256.

    Poses on a HIP device (and depth_type "z", view directions on - every shipped config): one ``inerf_gen_rays`` launch
    that reproduces the reference's CPU bits; CPU poses (what trainer.py:608-624 passes) take the torch expressions below,
    which ARE the reference's."""
    if (isinstance(Ts_c2w, torch.Tensor) and Ts_c2w.is_cuda and depth_type == "z" and use_viewdirs
            and (c2w_staticcam is None or (isinstance(c2w_staticcam, torch.Tensor) and c2w_staticcam.is_cuda))):
        if Ts_c2w.shape[0] != num_rays:
            raise ValueError(f"create_rays: {num_rays} images but {Ts_c2w.shape[0]} poses")
        rays = kernels.gen_rays(Ts_c2w.float(), height, width, fx, fy, cx, cy, near, far, convention == "opengl",
                                None if c2w_staticcam is None else c2w_staticcam.float())
        return rays.reshape(num_rays, height * width, -1)
    dirs_C = get_rays_camera(num_rays, height, width, fx, fy, cx, cy, depth_type=depth_type,
                             convention=convention).view(num_rays, -1, 3).to(Ts_c2w.device)
    rays_o, rays_d = get_rays_world(Ts_c2w, dirs_C)
    if use_viewdirs:
        viewdirs = rays_d
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays_world(c2w_staticcam, dirs_C)
        viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True).float()
    near, far = near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])
    rays = torch.cat([rays_o, rays_d, near, far], -1)
    if use_viewdirs:
        rays = torch.cat([rays, viewdirs], -1)
    return rays


def batchify_rays(render_fn, rays_flat, chunk=1024 * 32, coalesce_to=None):
    """training_utils.py:5-17.  As in object_level.batchify_rays the chunks' f16 range words are read together after the
    last chunk (one host synchronisation per frame) and only the chunks that left the range are rendered again in exact
    fp32.  ``coalesce_to`` (SSRRenderMixin.render_rays, eval-mode frames whose result cannot depend on the chunk boundaries):
    rays per call instead of ``chunk``; the launches keep one f16 range word per ``chunk`` rays, so a chunk that leaves the range is still the
    only one rendered again."""
    if coalesce_to is not None and coalesce_to > chunk:
        rets = []
        with kernels.deferred_range_checks("render_rays", raise_on_trip=False) as block, kernels.chunked_status(chunk):
            for j, i in enumerate(range(0, rays_flat.shape[0], coalesce_to)):
                block.tag = j
                rets.append(render_fn(rays_flat[i:i + coalesce_to]))
        out = {k: torch.cat([r[k] for r in rets], 0) for k in rets[0]}
        if block.tripped:     # one range word per caller's chunk (kernels.chunked_status): only the chunks that left the range, in exact fp32
            spans = sorted({(t[0] * coalesce_to + t[1] * chunk, chunk) if isinstance(t, tuple) else (t * coalesce_to, coalesce_to)
                            for t in block.tripped})
            kernels.warn_f32_fallback(f"render_rays: {len(spans)} of {-(-rays_flat.shape[0] // chunk)} chunks left the f16 range of the "
                                      "split-precision MLP kernel.")
            with _capi.forced_precision(_capi.PREC_F32):
                for i, n_i in spans:
                    again = render_fn(rays_flat[i:i + n_i])
                    for k in out:
                        out[k][i:i + n_i] = again[k]
        return out
    starts = list(range(0, rays_flat.shape[0], chunk))
    rets = []
    with kernels.deferred_range_checks("render_rays", raise_on_trip=False) as block:
        for j, i in enumerate(starts):
            block.tag = j
            rets.append(render_fn(rays_flat[i:i + chunk]))
    if block.tripped:
        kernels.warn_f32_fallback(f"render_rays: {len(block.tripped)} of {len(starts)} chunks left the f16 range of the "
                                  "split-precision MLP kernel.")
        with _capi.forced_precision(_capi.PREC_F32):
            for j in block.tripped:
                rets[j] = render_fn(rays_flat[starts[j]:starts[j] + chunk])
    if not rets:
        return {}
    return {k: torch.cat([r[k] for r in rets], 0) for k in rets[0]}


# ----------------------------------------------------------------------------------------------
# the trainer's render methods
# ----------------------------------------------------------------------------------------------
class SSRRenderMixin:
    """``render_rays`` / ``volumetric_rendering`` / ``create_ssr`` of SSRTrainer (trainer.py:693-846).

    Reads the same attributes the reference methods read: ``N_samples, N_importance, perturb,
    training, raw_noise_std, white_bkgd, enable_semantic, num_valid_semantic_class, endpoint_feat,
    netchunk, chunk, ssr_net_coarse, ssr_net_fine, embed_fn, embeddirs_fn``.  Two optional extras:
    ``return_raw`` (default True, as the reference always returns raw_coarse / raw_fine - ~3 GB per
    320x240 frame; set False to skip materialising them for the caller) and ``check_numerics``
    (default True: the reference's ungated nan/inf print, trainer.py:803-806, one device sync per key).
    """

    return_raw = True
    check_numerics = True

    def render_rays(self, flat_rays):
        ray_shape = flat_rays.shape
        all_ret = batchify_rays(self.volumetric_rendering, flat_rays, self.chunk, self._coalesced(flat_rays))      # one host synchronisation per frame
        for k in all_ret:
            all_ret[k] = torch.reshape(all_ret[k], list(ray_shape[:-1]) + list(all_ret[k].shape[1:]))
        return all_ret

    def render_path(self, rays, save_dir=None, update_cluster=False, b_f=0.5):
        """Render the frames ``rays[i]`` ([n_images, H*W, 11]) - trainer.py:1221-1456; returns the reference's 12-tuple
        ``(rgbs, disps, deps, vis_deps, sems, vis_sems, entropys, vis_entropys, albedos, shadings, residuals,
        cluster_manager)``.

        Per frame the label map (argmax of the softmax, :1244) and the entropy map (:1245) are computed on the device, all
        maps travel to the host as one pinned asynchronous copy that overlaps the next frame's kernels
        (frames.FrameStreamer) - the reference issues ~10 blocking ``.cpu()`` calls per frame.  The visualisations are
        the trainer's own business: ``vis_deps`` / ``vis_entropys`` need ``imgviz.depth2rgb`` (None when imgviz is not
        importable), ``vis_sems`` reads ``self.valid_colour_map`` (None without it).  Reads ``H_scaled, W_scaled, near,
        far`` like the reference.  ``update_cluster`` needs ``self.cluster_manager_factory`` (the reference's
        ``Cluster_Manager``: its mean-shift fitting is training control plane, not rebuilt here)."""
        import os
        import numpy as np
        from . import frames
        H, W = int(self.H_scaled), int(self.W_scaled)
        try:
            from imgviz import depth2rgb
        except ImportError:
            depth2rgb = None
        lvl = "fine" if self.N_importance > 0 else "coarse"
        keys = [f"{k}_{lvl}" for k in ("rgb", "disp", "depth", "albedo", "shading", "residual")]
        if self.enable_semantic:
            keys += ["sem_label", "sem_entropy"]
        cmap = getattr(self, "valid_colour_map", None)
        if cmap is not None:      # the trainer keeps it on the GPU (trainer.py:262,449,588: valid_colour_map.cuda()); indexed on the host here
            cmap = cmap.detach().cpu().numpy() if isinstance(cmap, torch.Tensor) else np.asarray(cmap)
        if save_dir is not None:
            assert os.path.exists(save_dir)
        acc = {k: [] for k in ("rgb", "disp", "dep", "vis_dep", "albedo", "shading", "residual", "sem", "vis_sem", "ent", "vis_ent")}
        sample_pixels, sample_labels = [], []
        widths = None

        def finish(i, frame):
            m = frames.unpack_frame(frame, widths, keys, (H, W))
            acc["rgb"].append(m[f"rgb_{lvl}"]); acc["disp"].append(m[f"disp_{lvl}"]); acc["albedo"].append(m[f"albedo_{lvl}"])
            acc["shading"].append(m[f"shading_{lvl}"]); acc["residual"].append(m[f"residual_{lvl}"]); acc["dep"].append(m[f"depth_{lvl}"])
            if depth2rgb is not None:
                acc["vis_dep"].append(depth2rgb(acc["dep"][-1], min_value=self.near, max_value=self.far))
            if self.enable_semantic:
                label = m["sem_label"].astype(np.uint8)
                acc["sem"].append(label); acc["ent"].append(m["sem_entropy"])
                if cmap is not None:
                    acc["vis_sem"].append(cmap[label.astype(np.int64)].astype(np.uint8))
                if depth2rgb is not None:
                    acc["vis_ent"].append(depth2rgb(acc["ent"][-1]))
            if update_cluster:
                sample_pixels.append(acc["albedo"][-1][::2, ::2, :].reshape(-1, 3))
                sample_labels.append(acc["sem"][-1][::2, ::2].reshape(-1, 1))
            if save_dir is not None:
                w = lambda name, img: frames.write_png(os.path.join(save_dir, name.format(i)), img)
                w("rgb_{:03d}.png", frames.to8b(acc["rgb"][-1])); w("albedo_{:03d}.png", frames.to8b(acc["albedo"][-1]))
                w("shading_{:03d}.png", frames.to8b(acc["shading"][-1])); w("residual_{:03d}.png", frames.to8b(acc["residual"][-1]))
                w("disp_{:03d}.png", acc["disp"][-1].astype(np.uint16)); w("depth_{:03d}.png", (acc["dep"][-1] * 1000).astype(np.uint16))
                if acc["vis_dep"]:
                    w("vis_depth_{:03d}.png", acc["vis_dep"][-1])
                if self.enable_semantic:
                    w("label_{:03d}.png", acc["sem"][-1]); w("entropy_{:03d}.png", frames.to8b(acc["ent"][-1]))
                    if acc["vis_sem"]:
                        w("vis_label_{:03d}.png", acc["vis_sem"][-1])
                    if acc["vis_ent"]:
                        w("vis_entropy_{:03d}.png", acc["vis_ent"][-1])

        streamer, in_flight = None, []
        for i in range(len(rays)):
            out = self.render_rays(rays[i])
            maps = {k: out[k].detach() for k in keys if k in out}
            if self.enable_semantic:
                logits = out[f"sem_logits_{lvl}"].detach()
                logp = F.log_softmax(logits, dim=-1)
                maps["sem_label"] = torch.argmax(F.softmax(logits, dim=-1), dim=-1).float()         # < 2^24: exact in fp32
                maps["sem_entropy"] = torch.sum(-logp * F.softmax(logits, dim=-1), dim=-1)
            pack, widths = frames.pack_maps(maps, keys)
            if pack.is_cuda:
                if streamer is None:
                    streamer = frames.FrameStreamer(pack.device)
                done = streamer.push(pack)
                in_flight.append(i)
                if done is not None:
                    finish(in_flight.pop(0), done)
            else:
                finish(i, pack.numpy())
        if streamer is not None:
            for frame in streamer.drain():
                finish(in_flight.pop(0), frame)
        st = lambda name: np.stack(acc[name], 0) if acc[name] else None
        cluster_manager = None
        if update_cluster:
            factory = getattr(self, "cluster_manager_factory", None)
            if factory is None:
                raise NotImplementedError("render_path(update_cluster=True) fits mean-shift clusters (Cluster_Manager.update_center, "
                                          "SSR/training/cluster.py:101-182); set self.cluster_manager_factory to that class")
            cluster_manager = factory(class_num=1 if getattr(self, "no_semantic_tree", False) else self.num_valid_semantic_class)
            cluster_manager.update_center(np.stack(sample_labels, 0), np.stack(sample_pixels, 0), band_factor=b_f)
            # trainer.py:1425-1440: every rendered albedo snapped to its cluster centres (c*.png) and the image re-composed from
            # the clustered albedo (edit*.png); manager.dest_color is the reference's method (cluster.dest_color: the HIP lookup)
            dev = rays[0].device if isinstance(rays[0], torch.Tensor) else torch.device("cpu")
            for i, albedo in enumerate(acc["albedo"]):
                pixel = torch.from_numpy(albedo).reshape(-1, 3).to(dev)
                label = torch.from_numpy(acc["sem"][i]).reshape(-1, 1).to(dev)
                result = cluster_manager.dest_color(pixel, label).reshape(albedo.shape).cpu().numpy()
                if save_dir is not None:
                    frames.write_png(os.path.join(save_dir, "c{:03d}.png".format(i)), frames.to8b(result))
                    edit = (result.reshape(-1, 3) * acc["shading"][i].reshape(-1, 1) + acc["residual"][i].reshape(-1, 3)).reshape(result.shape)
                    frames.write_png(os.path.join(save_dir, "edit{:03d}.png".format(i)), frames.to8b(edit))
        return (st("rgb"), st("disp"), st("dep"), st("vis_dep"), st("sem"), st("vis_sem"), st("ent"), st("vis_ent"),
                st("albedo"), st("shading"), st("residual"), cluster_manager)

    def _coalesced(self, flat_rays):
        """``self.chunk``, or the larger number of rays per ``volumetric_rendering`` call that gives the same frame
        (kernels.coalesced_chunk; results are bit-identical for any chunking): only when no random number is drawn per chunk (eval
        mode, or ``perturb == 0`` and ``raw_noise_std == 0``), ``raw_*`` is not returned, nothing records autograd and both networks
        take the fused kernels."""
        training = bool(self.training)
        if (self.return_raw or (training and (self.perturb > 0. or self.raw_noise_std > 0.)) or not flat_rays.is_cuda
                or flat_rays.shape[-1] <= 8 or flat_rays.shape[0] <= self.chunk or _wants_grad(self.ssr_net_coarse, self.ssr_net_fine)):
            return None
        if _fusable(self.ssr_net_coarse, self.embed_fn, self.embeddirs_fn) is None or (
                self.N_importance > 0 and _fusable(self.ssr_net_fine, self.embed_fn, self.embeddirs_fn) is None):
            return None
        c = int(self.num_valid_semantic_class) if self.enable_semantic else 0
        channels = _capi.BASE_CHANNELS + c + (_capi.ENDPOINT_DIM if (self.endpoint_feat and self.N_importance > 0) else 0)
        return kernels.coalesced_chunk(flat_rays.shape[0], self.chunk, self.N_samples, self.N_importance, channels, flat_rays.device)

    def volumetric_rendering(self, ray_batch):
        ray_batch = ray_batch.float()
        n, dev = ray_batch.shape[0], ray_batch.device
        if ray_batch.shape[-1] <= 8:
            raise NotImplementedError("volumetric_rendering needs view directions (use_viewdirs: true in every config)")
        desc = _fusable(self.ssr_net_coarse, self.embed_fn, self.embeddirs_fn)
        if desc is not None and self.N_importance > 0 and _fusable(self.ssr_net_fine, self.embed_fn, self.embeddirs_fn) is None:
            desc = None
        if desc is None:
            # The reference builds whatever netdepth / netwidth the YAML says (trainer.py:811-846) and calls it
            # (model_utils.py:19-35).  Outside the fused kernels' architecture (D=8, W=256, skips=[4], multires<=10,
            # multires_views<=4) the path runs STAGED, like object_level.render_rays does for a user-supplied network:
            # sampling and compositing on the HIP kernels (with their HIP backward), the networks through their own forward.
            _unfused_notice("volumetric_rendering")
        training = bool(self.training)
        t_vals = torch.linspace(0., 1., steps=self.N_samples, device=dev)
        # RNG draws in the reference's order: t_rand (:744), coarse noise (model_utils.py:70), u (rays.py:197), fine noise
        t_rand = torch.rand(n, self.N_samples, device=dev) if (self.perturb > 0. and training) else None
        std = self.raw_noise_std if training else 0
        noise_c = torch.randn(n, self.N_samples, device=dev) * std if std > 0. else None
        u = noise_f = None
        if self.N_importance > 0:
            det = (self.perturb == 0.) or (not training)
            u = (torch.linspace(0., 1., steps=self.N_importance, device=dev) if det
                 else torch.rand(n, self.N_importance, device=dev))
            noise_f = torch.randn(n, self.N_samples + self.N_importance, device=dev) * std if std > 0. else None
        ep = bool(self.endpoint_feat) and self.N_importance > 0

        def run(d):
            res = kernels.render_rays_fused(
                d, packing.packed_for_module(self.ssr_net_coarse, d, dev),
                packing.packed_for_module(self.ssr_net_fine, d, dev) if self.N_importance > 0 else None,
                ray_batch, self.N_samples, self.N_importance, t_vals, u, t_rand, noise_c, noise_f,
                white_bkgd=self.white_bkgd, endpoint=ep, want_raw_coarse=self.return_raw, want_raw_fine=self.return_raw,
                want_sem=bool(self.enable_semantic))
            kernels.check_f16_range(res.pop("status", None), "volumetric_rendering", deferrable=t_rand is None and noise_c is None)
            return res

        if desc is None:
            o = self._staged(ray_batch, t_vals, t_rand, noise_c, u, noise_f, ep, None)
        elif _wants_grad(self.ssr_net_coarse, self.ssr_net_fine):
            _training_path_notice("volumetric_rendering")
            td = _train_desc(desc)
            if td is None:
                o = self._staged(ray_batch, t_vals, t_rand, noise_c, u, noise_f, ep, None)
            else:       # one read of both networks' f16 range words, after the whole forward has been enqueued (object_level.render_rays)
                try:
                    with kernels.deferred_range_checks("volumetric_rendering (training step)"):
                        o = self._staged(ray_batch, t_vals, t_rand, noise_c, u, noise_f, ep, td)
                except FloatingPointError as e:
                    import warnings
                    warnings.warn(f"{e}  {FP32_LAYERS_NOTE}")
                    o = self._staged(ray_batch, t_vals, t_rand, noise_c, u, noise_f, ep, None)
        else:
            o = kernels.with_f32_fallback(desc, run)
        ret = {}
        if self.return_raw:
            ret["raw_coarse"] = o["raw_coarse"]
        for k in ("rgb", "disp", "acc", "depth", "albedo", "shading", "residual"):
            ret[k + "_coarse"] = o[k + "_coarse"]
        if self.enable_semantic:
            ret["sem_logits_coarse"] = o["sem_coarse"]
        if self.N_importance > 0:
            for k in ("rgb", "disp", "acc", "depth", "albedo", "shading", "residual"):
                ret[k + "_fine"] = o[k + "_fine"]
            if self.enable_semantic:
                ret["sem_logits_fine"] = o["sem_fine"]
            ret["z_std"] = o["z_std"]
            if self.return_raw:
                ret["raw_fine"] = o["raw_fine"]
            if ep:
                ret["feat_map_fine"] = o["feat_fine"]
        if self.check_numerics and not kernels._capturing():      # (a host read per key: not possible while a HIP graph records)
            for k in ret:
                if torch.isnan(ret[k]).any() or torch.isinf(ret[k]).any():
                    print(f"! [Numerical Error] {k} contains nan or inf.")
        return ret

    def _staged(self, ray_batch, t_vals, t_rand, noise_c, u, noise_f, ep, train_desc=None):
        """trainer.py:717-808 stage by stage, for training steps: HIP sampling, HIP compositing with its HIP backward, each
        network one autograd node with HIP forward and backward (kernels.mlp_train; torch ``forward`` when ``train_desc`` is
        None).  Same keys as kernels.render_rays_fused."""
        rays_o, rays_d, viewdirs = ray_batch[:, 0:3], ray_batch[:, 3:6].contiguous(), ray_batch[:, -3:]
        c_sem = self.num_valid_semantic_class if self.enable_semantic else 0

        def query(z, fn, endpoint=False):
            raw = _train_query(train_desc, fn, ray_batch, z, endpoint) if train_desc is not None else None
            if raw is None:
                spec = _layered_spec(fn, self.embed_fn, self.embeddirs_fn)
                if spec is not None:            # another netdepth / netwidth, or a batch outside the f16 range: fp32 layer kernels
                    raw = layered.evaluate(spec, fn, ray_batch, z, endpoint)
                else:
                    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
                    raw = run_network(pts, viewdirs, fn, self.embed_fn, self.embeddirs_fn, self.netchunk, show_endpoint=endpoint)
            return raw

        z_vals = kernels.sample_coarse(ray_batch, t_vals, t_rand, False)
        raw = query(z_vals, self.ssr_net_coarse)
        c = kernels.composite(raw, z_vals, rays_d, noise_c, self.white_bkgd, n_classes=c_sem)
        o = {k + "_coarse": v for k, v in c.items()}
        o["raw_coarse"] = raw
        if self.N_importance > 0:
            # the resampled depths carry no gradient (z_samples.detach(), trainer.py:762)
            z_samples, z_fine, z_std = kernels.sample_fine(z_vals, c["weights"].detach(), u, self.N_importance)
            raw = query(z_fine, self.ssr_net_fine, ep)
            # the endpoint feature is the LAST 128 channels of raw (model_utils.py:99-103, a literal 128 there).  A foreign netwidth
            # gives raw another width (W // 2 feature channels): the kernel's feature lanes only take the 256-wide network's layout,
            # the reference's literal slice is evaluated as written for any other
            native_feat = ep and raw.shape[-1] == 11 + c_sem + 128
            f = kernels.composite(raw, z_fine, rays_d, noise_f, self.white_bkgd, n_classes=c_sem, feat_dim=128 if native_feat else 0)
            if ep and not native_feat:
                f["feat"] = torch.sum(f["weights"][..., None] * raw[..., -128:], -2)
            o.update({k + "_fine": v for k, v in f.items()})
            o["raw_fine"], o["z_std"] = raw, z_std
        return o

    def create_ssr(self):
        """Build coarse + fine Semantic_NeRF and the encoders - trainer.py:811-846 (optimiser included)."""
        r, m = self.config["render"], self.config["model"]
        embed_fn, input_ch = get_embedder(r["multires"], r["i_embed"], scalar_factor=10)
        embeddirs_fn, input_ch_views = (get_embedder(r["multires_views"], r["i_embed"], scalar_factor=1)
                                        if r["use_viewdirs"] else (None, 0))
        output_ch = 5 if self.N_importance > 0 else 4
        mk = lambda d, w: Semantic_NeRF(enable_semantic=self.enable_semantic,
                                        num_semantic_classes=self.num_valid_semantic_class, D=d, W=w, input_ch=input_ch,
                                        output_ch=output_ch, skips=[4], input_ch_views=input_ch_views,
                                        use_viewdirs=r["use_viewdirs"]).cuda()
        model = mk(m["netdepth"], m["netwidth"])
        grad_vars = list(model.parameters())
        model_fine = None
        if self.N_importance > 0:
            model_fine = mk(m["netdepth_fine"], m["netwidth_fine"])
            grad_vars += list(model_fine.parameters())
        self.ssr_net_coarse, self.ssr_net_fine = model, model_fine
        self.embed_fn, self.embeddirs_fn = embed_fn, embeddirs_fn
        self.optimizer = torch.optim.Adam(params=grad_vars, lr=getattr(self, "lrate", 5e-4))


class SSRRenderer(SSRRenderMixin):
    """Stand-alone holder of the attributes the mixin reads (for rendering without the trainer)."""

    def __init__(self, num_classes, N_samples=64, N_importance=128, multires=10, multires_views=4, white_bkgd=False,
                 enable_semantic=True, endpoint_feat=False, chunk=1024 * 32, netchunk=1024 * 32, perturb=1.,
                 raw_noise_std=1., device="cuda"):
        self.N_samples, self.N_importance = N_samples, N_importance
        self.perturb, self.raw_noise_std, self.white_bkgd = perturb, raw_noise_std, white_bkgd
        self.enable_semantic, self.num_valid_semantic_class = enable_semantic, num_classes
        self.endpoint_feat, self.chunk, self.netchunk = endpoint_feat, chunk, netchunk
        self.training = False
        self.embed_fn, ch = get_embedder(multires, 0, scalar_factor=10)
        self.embeddirs_fn, chv = get_embedder(multires_views, 0, scalar_factor=1)
        mk = lambda: Semantic_NeRF(enable_semantic, num_classes, D=8, W=256, input_ch=ch, output_ch=5, skips=[4],
                                   input_ch_views=chv, use_viewdirs=True).to(device)
        self.ssr_net_coarse = mk()
        self.ssr_net_fine = mk() if N_importance > 0 else None
