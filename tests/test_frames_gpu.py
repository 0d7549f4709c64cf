"""GPU: the image-level neighbours of the path - ray generation in one launch, bit for bit the
reference's CPU rays; to8b on the device; the render_path loops with one asynchronous device->host copy per frame."""
import os
import time

import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def test_gen_rays_bit_exact_against_the_reference():
    """inerf_gen_rays against tests/golden/rays_generators.npz (the REFERENCE's get_rays / render() assembly / create_rays,
    evaluated on its CPU path): every float identical - a one-ulp difference in a direction is a 2e-4 phase error after the
    2^9-frequency encoding."""
    from intrinsicnerf_amd import kernels, object_level as ol, ssr
    fx = load_golden("rays_generators")
    dev = _dev()
    H, W, K = int(fx["H"]), int(fx["W"]), fx["K"]
    c2w, c2s = torch.from_numpy(fx["c2w"]).to(dev), torch.from_numpy(fx["c2w_static"]).to(dev)
    got = kernels.gen_rays(c2w, H, W, K[0][0], K[1][1], K[0][2], K[1][2], 2.0, 6.0, True)
    assert torch.equal(got.cpu(), torch.from_numpy(fx["render_rays_plain"]))
    got = kernels.gen_rays(c2w, H, W, K[0][0], K[1][1], K[0][2], K[1][2], 2.0, 6.0, True, static_poses=c2s)
    assert torch.equal(got.cpu(), torch.from_numpy(fx["render_rays_static"]))
    ro, rd = ol.get_rays(H, W, K, c2w)                                     # device pose -> the kernel
    assert torch.equal(ro.cpu(), torch.from_numpy(fx["get_rays_o"])) and torch.equal(rd.cpu(), torch.from_numpy(fx["get_rays_d"]))
    # render() itself: capture what it hands to batchify_rays
    seen = {}
    orig = ol.batchify_rays

    def spy(rays_flat, chunk=1024 * 32, **kw):
        seen["rays"] = rays_flat.clone()
        n = rays_flat.shape[0]
        z3, z1 = torch.zeros(n, 3, device=dev), torch.zeros(n, device=dev)
        return {"rgb_map": z3, "disp_map": z1, "acc_map": z1, "albedo_map": z3, "shading_map": z1, "residual_map": z3}

    ol.batchify_rays = spy
    try:
        out = ol.render(H, W, K, chunk=64, c2w=c2w, near=2.0, far=6.0, use_viewdirs=True, ndc=False)
        assert torch.equal(seen["rays"].cpu(), torch.from_numpy(fx["render_rays_plain"])) and tuple(out[0].shape) == (H, W, 3)
        ol.render(H, W, K, chunk=64, c2w=c2w, near=2.0, far=6.0, use_viewdirs=True, ndc=False, c2w_staticcam=c2s)
        assert torch.equal(seen["rays"].cpu(), torch.from_numpy(fx["render_rays_static"]))
        ol.render(H, W, K, chunk=64, c2w=c2w, near=2.0, far=6.0, use_viewdirs=True, ndc=True)       # ndc stays on the torch expressions
        assert np.allclose(seen["rays"].cpu().numpy(), fx["render_rays_ndc"], rtol=1e-6, atol=1e-7)
    finally:
        ol.batchify_rays = orig
    # SSR create_rays, both conventions, static camera
    Hs, Ws, T = int(fx["H_ssr"]), int(fx["W_ssr"]), torch.from_numpy(fx["ssr_T"]).to(dev)
    for conv in ("opencv", "opengl"):
        got = ssr.create_rays(2, T, Hs, Ws, 5.5, 6.5, 9.5, 9.5, 0.1, 10.0, convention=conv)
        assert torch.equal(got.cpu(), torch.from_numpy(fx[f"ssr_rays_{conv}_z"])), conv
    got = ssr.create_rays(2, T, Hs, Ws, 5.5, 6.5, 9.5, 9.5, 0.1, 10.0, c2w_staticcam=T.flip(0))
    assert torch.equal(got.cpu(), torch.from_numpy(fx["ssr_rays_static"]))
    got = ssr.create_rays(2, T, Hs, Ws, 5.5, 6.5, 9.5, 9.5, 0.1, 10.0, depth_type="euclidean")           # torch expressions
    assert np.allclose(got.cpu().numpy(), fx["ssr_rays_opencv_euclidean"], rtol=1e-6, atol=1e-7)


def test_gen_rays_full_frames_equal_the_cpu_mirrors():
    """At BASELINE.json's frame sizes: the kernel's rays == the torch mirrors evaluated on the CPU (which reproduce the
    reference bit for bit, tests/test_frontends_cpu.py) - all 640 000 and 76 800 rays."""
    import bench
    from intrinsicnerf_amd import object_level as ol, ssr
    dev = _dev()
    K, pose = bench.chair_intrinsics(), bench.chair_pose()
    ro, rd = ol.get_rays(800, 800, K, pose)                                # CPU: the reference's expressions
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    want = torch.cat([ro, rd, 2.0 * torch.ones_like(rd[..., :1]), 6.0 * torch.ones_like(rd[..., :1]), vd], -1).reshape(-1, 11)
    from intrinsicnerf_amd import kernels
    got = kernels.gen_rays(pose.to(dev), 800, 800, K[0][0], K[1][1], K[0][2], K[1][2], 2.0, 6.0, True)
    assert torch.equal(got.cpu(), want)
    g = torch.Generator().manual_seed(1)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    T = torch.eye(4)[None].repeat(3, 1, 1)
    T[:, :3, :3], T[:, :3, 3] = q, torch.tensor([0.1, 0.2, 0.3])
    T[2, :3, 3] = torch.tensor([-1.0, 0.4, 2.2])
    fxx = 320 / 2.0 / np.tan(np.deg2rad(45.0))
    for conv in ("opencv", "opengl"):
        want = ssr.create_rays(3, T, 240, 320, fxx, fxx, 159.5, 119.5, 0.1, 10.0, convention=conv)
        got = ssr.create_rays(3, T.to(dev), 240, 320, fxx, fxx, 159.5, 119.5, 0.1, 10.0, convention=conv)
        assert torch.equal(got.cpu(), want), conv


def test_frame_to_u8_is_numpy_to8b():
    from intrinsicnerf_amd import frames
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.rand(100000, generator=g) * 1.4 - 0.2, torch.arange(256, dtype=torch.float32) / 255.0,
                   torch.tensor([0.0, 1.0, -0.0, 1e-9, 0.999999, 1.0000001, float("inf"), -float("inf")])])
    want = (255 * np.clip(x.numpy(), 0, 1)).astype(np.uint8)
    got = frames.to8b(x.to(_dev()).reshape(-1, 4)).reshape(-1).cpu().numpy()
    assert np.array_equal(got, want)
    assert frames.to8b(torch.tensor([float("nan")]).to(_dev())).item() == 0


def _chair_nets(dev, side):
    import bench
    from intrinsicnerf_amd import object_level as ol
    focal = 0.5 * side / np.tan(0.5 * bench.CAMERA_ANGLE_X)
    K = np.array([[focal, 0, 0.5 * side], [0, focal, 0.5 * side], [0, 0, 1]])
    ro, rd = ol.get_rays(side, side, K, bench.chair_pose())
    vd = rd / rd.norm(dim=-1, keepdim=True)
    rays = torch.cat([ro, rd, 2 * torch.ones_like(rd[..., :1]), 6 * torch.ones_like(rd[..., :1]), vd], -1).reshape(-1, 11)
    probe = rays[::max(1, rays.shape[0] // 384)]                    # a few hundred rays for the CPU-side calibration probe
    sd_c, _ = oracle.calibrated_lcg_weights("object", 0, 50, probe)
    sd_f, _ = oracle.calibrated_lcg_weights("object", 0, 51, probe)
    embed, ch = ol.get_embedder(10, 0); embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net_c, net_f = mk(), mk()
    net_c.load_state_dict(sd_c); net_f.load_state_dict(sd_f)
    kw = dict(network_fn=net_c, network_fine=net_f, network_query_fn=ol.NetworkQuery(embed, embed_d), N_samples=64, N_importance=128,
              white_bkgd=True, perturb=False, raw_noise_std=0., use_viewdirs=True, ndc=False, lindisp=False, near=2., far=6.)
    return K, focal, kw


def test_object_render_path_pipelined_equals_the_blocking_loop(tmp_path):
    """render_path == the reference's loop (render, then one blocking .cpu() per map) frame for frame, bit for bit; the
    files the reference writes are written; and the per-frame saving of the pipelined loop is reported."""
    import bench
    from intrinsicnerf_amd import object_level as ol
    dev = _dev()
    side = 48
    K, focal, kw = _chair_nets(dev, side)
    poses = torch.stack([torch.cat([bench.chair_pose(theta_deg=t), torch.tensor([[0., 0., 0., 1.]])], 0) for t in (20., 40., 75., 130., 200.)]).to(dev)
    with torch.no_grad():
        rgbs, disps, mgr = ol.render_path(poses, (side, side, focal), K, 1 << 15, kw, savedir=str(tmp_path))
        want = [ol.render(side, side, K, chunk=1 << 15, c2w=p[:3, :4], **kw) for p in poses]
    assert mgr is None and rgbs.shape == (5, side, side, 3) and disps.shape == (5, side, side) and rgbs.dtype == np.float32
    for i, w in enumerate(want):
        assert np.array_equal(rgbs[i], w[0].cpu().numpy()) and np.array_equal(disps[i], w[1].cpu().numpy(), equal_nan=True)
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(f"{p}{i:03d}.png" for i in range(5) for p in ("", "a", "s", "res", "acc"))
    assert open(tmp_path / "000.png", "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    with pytest.raises(NotImplementedError):
        with torch.no_grad():
            ol.render_path(poses[:1], (side, side, focal), K, 1 << 15, kw, update_cluster=True)
    # what the pipelining is worth: the reference's loop shape vs render_path on full 800x800 frames
    side = 800
    K, focal, kw = _chair_nets(dev, side)
    poses = poses[:3]
    with torch.no_grad():
        ol.render_path(poses[:1], (side, side, focal), K, side * side, kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in poses:                                                      # run_nerf.py:163-174
            ro, rd = (t.reshape(-1, 3) for t in _torch_get_rays(side, side, K, p[:3, :4]))
            out = ol.render(side, side, K, chunk=side * side, rays=(ro, rd), **kw)
            _ = [out[j].cpu().numpy() for j in range(6)]
        t_ref = (time.perf_counter() - t0) / len(poses)
        t0 = time.perf_counter()
        ol.render_path(poses, (side, side, focal), K, side * side, kw)
        t_new = (time.perf_counter() - t0) / len(poses)
    print(f"\nrender_path, 800x800, per frame: blocking loop {t_ref * 1e3:.2f} ms, pipelined {t_new * 1e3:.2f} ms "
          f"(saved {(t_ref - t_new) * 1e3:.2f} ms = {(t_ref - t_new) / t_ref * 100:.2f} %)")


def _torch_get_rays(H, W, K, c2w):
    """The reference's get_rays evaluated with torch ops on the device (what the front-end did before inerf_gen_rays)."""
    j, i = torch.meshgrid(torch.linspace(0, H - 1, H, device=c2w.device), torch.linspace(0, W - 1, W, device=c2w.device), indexing="ij")
    dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    return c2w[:3, -1].expand(rays_d.shape), rays_d


def test_ssr_render_path_tuple():
    from intrinsicnerf_amd import ssr
    from oracle import calibration as cal
    dev = _dev()
    H, W, C = 12, 16, 5
    r = ssr.SSRRenderer(C, white_bkgd=False, chunk=100, device=dev)
    r.H_scaled, r.W_scaled, r.near, r.far = H, W, 0.1, 10.0
    r.check_numerics = False
    T = torch.eye(4)[None].repeat(3, 1, 1)
    T[1, :3, 3], T[2, :3, 3] = torch.tensor([0.2, 0.0, 0.1]), torch.tensor([-0.3, 0.1, 0.0])
    rays = ssr.create_rays(3, T.to(dev), H, W, 8.0, 8.0, (W - 1) / 2.0, (H - 1) / 2.0, 0.1, 10.0)
    sd = cal.calibrated_default_init("ssr", C, 0, rays[0].cpu())
    r.ssr_net_coarse.load_state_dict(sd); r.ssr_net_fine.load_state_dict(cal.calibrated_default_init("ssr", C, 1, rays[0].cpu()))
    r.valid_colour_map = torch.arange(C * 3, dtype=torch.uint8).reshape(C, 3)
    with torch.no_grad():
        out = r.render_path(rays)
        direct = [r.render_rays(rays[i]) for i in range(3)]
    assert len(out) == 12 and out[11] is None
    rgbs, disps, deps, vis_deps, sems, vis_sems, ents, vis_ents, albedos, shadings, residuals, _ = out
    assert rgbs.shape == (3, H, W, 3) and deps.shape == (3, H, W) and sems.shape == (3, H, W) and sems.dtype == np.uint8
    assert vis_sems.shape == (3, H, W, 3) and ents.shape == (3, H, W) and albedos.shape == (3, H, W, 3)
    for i, d in enumerate(direct):
        assert np.array_equal(rgbs[i].reshape(-1, 3), d["rgb_fine"].cpu().numpy())
        assert np.array_equal(deps[i].reshape(-1), d["depth_fine"].cpu().numpy())
        assert np.array_equal(sems[i].reshape(-1), torch.argmax(d["sem_logits_fine"], -1).cpu().numpy().astype(np.uint8))
        assert np.array_equal(vis_sems[i].reshape(-1, 3), r.valid_colour_map.numpy()[sems[i].reshape(-1).astype(np.int64)])


def test_ssr_render_path_with_a_cuda_colour_map_and_the_cluster_post_pass(tmp_path):
    """the trainer keeps ``valid_colour_map`` on the GPU (trainer.py:262,449,588) - render_path must index it on
    the host without raising; and ``update_cluster=True`` runs the reference's post-pass (trainer.py:1425-1440): every albedo
    frame through ``cluster_manager.dest_color`` -> ``c{:03d}.png`` and the re-composed ``edit{:03d}.png``."""
    from intrinsicnerf_amd import cluster, ssr
    from oracle import calibration as cal
    dev = _dev()
    H, W, C = 12, 16, 5
    r = ssr.SSRRenderer(C, white_bkgd=False, chunk=100, device=dev)
    r.H_scaled, r.W_scaled, r.near, r.far = H, W, 0.1, 10.0
    r.check_numerics = False
    T = torch.eye(4)[None].repeat(2, 1, 1)
    T[1, :3, 3] = torch.tensor([0.2, 0.0, 0.1])
    rays = ssr.create_rays(2, T.to(dev), H, W, 8.0, 8.0, (W - 1) / 2.0, (H - 1) / 2.0, 0.1, 10.0)
    r.ssr_net_coarse.load_state_dict(cal.calibrated_default_init("ssr", C, 0, rays[0].cpu()))
    r.ssr_net_fine.load_state_dict(cal.calibrated_default_init("ssr", C, 1, rays[0].cpu()))
    colours = torch.arange(C * 3, dtype=torch.uint8).reshape(C, 3)
    r.valid_colour_map = colours.to(dev)                               # as the reference trainer holds it

    seen = {}

    class Manager:      # stands in for the reference's Cluster_Manager (its mean-shift fitting is control plane): records the calls
        def __init__(self, class_num):
            seen["class_num"] = class_num

        def update_center(self, labels, pixels, band_factor):
            seen["fit"] = (labels.shape, pixels.shape, band_factor)

        def dest_color(self, pixel, label):
            seen.setdefault("lookups", []).append((tuple(pixel.shape), tuple(label.shape), pixel.device.type))
            return 0.5 * pixel + 0.25

    r.cluster_manager_factory = Manager
    with torch.no_grad():
        out = r.render_path(rays, save_dir=str(tmp_path), update_cluster=True, b_f=0.4)
    rgbs, _, _, _, sems, vis_sems, _, _, albedos, shadings, residuals, manager = out
    assert isinstance(manager, Manager) and seen["class_num"] == C and seen["fit"][2] == 0.4
    assert seen["fit"][0] == (2, (H // 2) * (W // 2), 1) and seen["fit"][1] == (2, (H // 2) * (W // 2), 3)
    assert seen["lookups"] == [((H * W, 3), (H * W, 1), "cuda")] * 2
    assert np.array_equal(vis_sems[1].reshape(-1, 3), colours.numpy()[sems[1].reshape(-1).astype(np.int64)])
    import zlib
    import struct

    def read_png(path):
        data = open(path, "rb").read()
        pos, idat, hdr = 8, b"", None
        while pos < len(data):
            n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
            if tag == b"IHDR":
                hdr = struct.unpack(">IIBBBBB", data[pos + 8:pos + 8 + n])
            if tag == b"IDAT":
                idat += data[pos + 8:pos + 8 + n]
            pos += 12 + n
        w, h, depth, colour = hdr[:4]
        raw = zlib.decompress(idat)
        bpp = (3 if colour == 2 else 1) * depth // 8
        rows = [raw[r * (1 + w * bpp) + 1:(r + 1) * (1 + w * bpp)] for r in range(h)]
        assert all(raw[r * (1 + w * bpp)] == 0 for r in range(h))       # filter type 0 (what both encoders here emit for tiny images is checked by decoding)
        return np.frombuffer(b"".join(rows), dtype=np.uint8 if depth == 8 else ">u2").reshape((h, w, 3) if colour == 2 else (h, w))

    to8 = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)
    for i in range(2):
        clustered = 0.5 * albedos[i] + 0.25
        try:
            c_img = read_png(tmp_path / f"c{i:03d}.png")
            e_img = read_png(tmp_path / f"edit{i:03d}.png")
        except AssertionError:      # imageio present and chose another PNG filter: existence is all that can be checked cheaply
            assert (tmp_path / f"c{i:03d}.png").stat().st_size > 0 and (tmp_path / f"edit{i:03d}.png").stat().st_size > 0
            continue
        assert np.array_equal(c_img, to8(clustered))
        edit = (clustered.reshape(-1, 3) * shadings[i].reshape(-1, 1) + residuals[i].reshape(-1, 3)).reshape(clustered.shape)
        assert np.array_equal(e_img, to8(edit))
    # 16-bit maps always go through the built-in encoder: disp_000.png decodes to the uint16 image the reference writes
    d16 = read_png(tmp_path / "disp_000.png")
    assert d16.dtype.itemsize == 2 and d16.shape == (H, W)
