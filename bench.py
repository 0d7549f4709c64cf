#!/usr/bin/env python3
"""Headline benchmark: rays/s of the IntrinsicNeRF render path at 64+128 samples per ray.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" renders one synthetic 800x800 Blender-chair frame (BASELINE.json configs[2]: 640,000 rays,
64 coarse + 128 importance samples, separate coarse/fine networks, white background, eval mode)
through the product front-end ``intrinsicnerf_amd.object_level.render``.  With N GPUs the frame's rays
are split into N row bands (one process per GPU) and the rendered maps are all-gathered over RCCL, so
the total work is fixed: "scaling": "strong".  Inputs are resident in HBM before the timed region.

The MLP GEMMs run in the package's default arithmetic (INERF_PRECISION, default "f16x3": fp32 operands
split into f16 hi/lo pairs, 3 f16 MFMA products per MAC, fp32 accumulation - same error against fp64
as the exact-fp32 MFMA kernel); "dtype" says which one ran.

Prints ONE JSON line (rank 0) with the bench contract's fields plus
  "roofline"     : algorithmic TFLOP/s (2 x 659,456 MAC per sample point) of the dominant kernel from
                   HIP-event timings of its launches at this workload's sizes.  For the f16x3 kernel the
                   peak is the dense f16 MFMA peak divided by the 3 products it issues per fp32 MAC
                   (2500 / 3 = 833 TFLOP/s); for the exact-fp32 kernel it is the 157.3 TFLOP/s fp32 MFMA
                   peak.  "roofline_f32_kernel" always carries the exact-fp32 kernel's figures too;
  "cpu_baseline" : the CPU oracle (PyTorch-CPU restatement == reference, see oracle/) timed on this
                   box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

H = W = 800
N_SAMPLES, N_IMPORTANCE = 64, 128
CAMERA_ANGLE_X = 0.6911112070083618          # NeRF-synthetic transforms_*.json
NEAR, FAR = 2.0, 6.0                         # run_nerf.py:705-706
FLOP_PER_POINT = 2 * 659456                  # BASELINE.md section 2 (GEMM MACs of one NeRF evaluation)
PEAK_F32_MFMA_TFLOPS = 157.3                 # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_F16_MFMA_TFLOPS = 2500.0                # MI355X_MICROARCH.md: dense f16/bf16 matrix peak
# HBM bytes per sample point of the encode+MLP kernels from the PMC passes (FETCH_SIZE + WRITE_SIZE, separate
# passes) in profiles/r01_mlp_pmc_traffic.txt: 311.6 MB per 6,291,456-point launch (algorithmic: 306 MB).
MEASURED_F16_MFMA_ONLY_TFLOPS = 1590.0      # dense f16, random operands, 170-340 ms runs (zero operands: 2470)
PMC_HBM_BYTES_PER_POINT = {"f16x3": 55.9, "f32": 49.5}    # rocprofv3 PMC, profiles/r01_mlp_pmc_traffic.txt


def chair_pose(theta_deg=40.0, phi_deg=-30.0, radius=4.0):
    """pose_spherical of the NeRF-synthetic orbit (load_blender.py:29-34 convention)."""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    t = np.eye(4); t[2, 3] = radius
    rp = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1.0]])
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1.0]])
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    return torch.tensor((flip @ rt @ rp @ t)[:3, :4], dtype=torch.float32)


def cpu_baseline(budget_s=10.0):
    """Time the CPU oracle on rays of the same frame with the same networks.

    torch's intra-op pool is not monotone in thread count on many-core hosts (256 threads on a 512-ray
    chunk is ~60x slower than 32), so the thread count is probed first and the remaining budget is
    spent at the best one; "cores" reports the threads actually used.
    """
    import oracle
    from intrinsicnerf_amd import object_level as ol
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    focal = 0.5 * W / np.tan(0.5 * CAMERA_ANGLE_X)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    ro, rd = ol.get_rays(H, W, K, chair_pose())
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    sel = torch.arange(0, H * W, 39)[:16384]                   # rays spread over the frame
    ro, rd = ro[sel], rd[sel]
    rays = torch.cat([ro, rd, NEAR * torch.ones_like(rd[:, :1]), FAR * torch.ones_like(rd[:, :1]),
                      rd / rd.norm(dim=-1, keepdim=True)], -1)
    sd_c, sd_f = oracle.make_state_dict("object", seed=0), oracle.make_state_dict("object", seed=1)
    cfg = oracle.RenderConfig(variant="object", n_samples=N_SAMPLES, n_importance=N_IMPORTANCE, white_bkgd=True)
    chunk = 1024

    def run(lo, m=chunk):
        with torch.no_grad():
            oracle.render_rays(rays[lo:lo + m], sd_c, sd_f, cfg)

    best_t, best_rate = 1, 0.0
    for t in sorted({min(avail, c) for c in (8, 16, 32, 64, 128, 256)}):    # short probe: 256 rays per width
        torch.set_num_threads(t)
        run(0, 256)                                             # warm the pool at this width
        t0 = time.perf_counter()
        run(256, 256)
        rate = 256 / (time.perf_counter() - t0)
        if rate > best_rate:
            best_t, best_rate = t, rate
        elif rate < 0.5 * best_rate:
            break                                               # past the knee: wider only gets slower
    torch.set_num_threads(best_t)
    run(0)
    done, t0 = 0, time.perf_counter()
    while True:
        run((done + 2 * chunk) % (rays.shape[0] - chunk))
        done += chunk
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {"value": done / dt, "unit": "rays/s", "cores": int(best_t), "kind": "port",
            "sample": f"{done} rays of the same 800x800 frame, 64+128 samples, PyTorch-CPU oracle "
                      f"(oracle/intrinsic_render.py == reference, see tests/golden) in {dt:.1f} s at {best_t} threads "
                      f"(best of a thread-count probe; host exposes {avail} logical CPUs)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    # one process per GPU.  INERF_BENCH_SHARE_GPU=1 (debug only) lets several ranks share device 0 over gloo, to
    # exercise the sharding + gather logic on a single-GPU box; the numbers it prints mean nothing.
    share = os.environ.get("INERF_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world)

    import __graft_entry__
    __graft_entry__.build()
    import oracle                                      # only for make_state_dict (seeded default init) + cpu_baseline
    from intrinsicnerf_amd import _capi, distributed as idist, kernels, object_level as ol, packing

    # ---- synthetic workload: configs[2], resident in HBM ----
    focal = 0.5 * W / np.tan(0.5 * CAMERA_ANGLE_X)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    ro, rd = ol.get_rays(H, W, K, chair_pose().to(dev))
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    n_total = H * W
    b, e = idist.shard_bounds(n_total, rank, world)
    ro_l, rd_l = ro[b:e].contiguous(), rd[b:e].contiguous()
    n_local = e - b
    embed, ch = ol.get_embedder(10, 0)
    embed_d, ch_d = ol.get_embedder(4, 0)
    mk = lambda: ol.NeRF(D=8, W=256, input_ch=ch, output_ch=5, skips=[4], input_ch_views=ch_d, use_viewdirs=True).to(dev)
    net_c, net_f = mk(), mk()
    net_c.load_state_dict(oracle.make_state_dict("object", seed=0))      # random-init weights, seeds 0 / 1
    net_f.load_state_dict(oracle.make_state_dict("object", seed=1))
    kw = dict(network_fn=net_c, network_fine=net_f, network_query_fn=ol.NetworkQuery(embed, embed_d), N_samples=N_SAMPLES,
              N_importance=N_IMPORTANCE, white_bkgd=True, perturb=False, raw_noise_std=0., use_viewdirs=True, ndc=False,
              lindisp=False)

    def step():
        with torch.no_grad():
            r = ol.render(H, W, K, chunk=n_local, rays=(ro_l, rd_l), near=NEAR, far=FAR, **kw)
            maps = dict(zip(("rgb_map", "disp_map", "acc_map", "albedo_map", "shading_map", "residual_map"), r[:6]))
            return idist.gather_maps(maps, n_total) if world > 1 else maps

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert frame["rgb_map"].shape[0] == n_total and torch.isfinite(frame["rgb_map"]).all()
    if world > 1:   # every rank must hold the same full frame, and its own band must be what it rendered
        chk = torch.tensor([float(frame["rgb_map"].double().sum()), float(frame["acc_map"].double().sum())],
                           device=dev, dtype=torch.float64)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "ranks disagree on the gathered frame"
    rays_per_s = n_total * args.steps / dt

    # ---- roofline of the dominant kernel: HIP events around its launches, same sizes as the timed region ----
    vd = rd_l / rd_l.norm(dim=-1, keepdim=True)
    rays_l = torch.cat([ro_l, rd_l, NEAR * torch.ones_like(rd_l[:, :1]), FAR * torch.ones_like(rd_l[:, :1]), vd], -1)
    t_vals = torch.linspace(0., 1., N_SAMPLES, device=dev)
    u = torch.linspace(0., 1., N_IMPORTANCE, device=dev)
    flop_per_launch = FLOP_PER_POINT * n_local * (N_SAMPLES + (N_SAMPLES + N_IMPORTANCE)) / 2.0   # mean of the two launches

    def kernel_roofline(prec, reps):
        desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, 1.0, prec)
        pc, pf = packing.packed_for_module(net_c, desc, dev), packing.packed_for_module(net_f, desc, dev)
        st = kernels.render_rays_fused(desc, pc, pf, rays_l, N_SAMPLES, N_IMPORTANCE, t_vals, u, white_bkgd=True, want_stages=True)
        z_c, z_f = st["z_coarse"], st["z_fine"]
        del st
        torch.cuda.synchronize()
        durs = []                                              # one entry per launch [ms]
        for _ in range(reps):
            for packed, z in ((pc, z_c), (pf, z_f)):
                a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                kernels.encode_mlp(desc, packed, rays_l, z)
                bb.record()
                bb.synchronize()
                durs.append(a.elapsed_time(bb))
        avg_ms = sum(durs) / len(durs)
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12
        f16 = prec == _capi.PREC_F16X3
        peak = PEAK_F16_MFMA_TFLOPS / 3.0 if f16 else PEAK_F32_MFMA_TFLOPS
        return {"bound": "mfma", "kernel": "k_encode_mlp_f16x3_dual<false, false>" if f16 else "k_encode_mlp<false, 2>", "achieved": achieved,
                "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": PMC_HBM_BYTES_PER_POINT["f16x3" if f16 else "f32"] * n_local * (N_SAMPLES + (N_SAMPLES + N_IMPORTANCE)) / 2.0,
                "traffic_note": f"HBM bytes per launch = {PMC_HBM_BYTES_PER_POINT['f16x3' if f16 else 'f32']} B/point measured by "
                                "rocprofv3 PMC (FETCH_SIZE, WRITE_SIZE in separate passes, profiles/r01_mlp_pmc_traffic.txt) x this "
                                f"launch's points; {'1.15' if f16 else '1.02'}x algorithmic",
                "avg_launch_ms": avg_ms,
                "launches_timed": len(durs), "flop_per_launch": flop_per_launch,
                "peak_basis": ("dense f16 MFMA 2500 TFLOP/s / 3 products per fp32 MAC" if f16
                               else "dense fp32 MFMA 157.3 TFLOP/s"),
                "achieved_vs_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                # what the matrix pipe sustains on this board when it does nothing but MFMAs on random operands
                # (scripts/microbench/mfma_peak.hip, profiles/r01_mfma_peak_microbench.txt): the power budget, not the name-plate
                "measured_mfma_only_ceiling": (MEASURED_F16_MFMA_ONLY_TFLOPS / 3.0) if f16 else None,
                "frac_of_measured_ceiling": (achieved / (MEASURED_F16_MFMA_ONLY_TFLOPS / 3.0)) if f16 else None}

    prec = _capi.default_precision()
    roofline = kernel_roofline(prec, max(1, args.steps))
    roofline_f32 = roofline if prec == _capi.PREC_F32 else kernel_roofline(_capi.PREC_F32, 1)

    # the same frame through the product front-end with the exact-fp32 MFMA kernel, for reference (one timed step)
    exact = None
    if prec != _capi.PREC_F32:
        os.environ["INERF_PRECISION"] = "f32"
        step(); fence()
        t1 = time.perf_counter()
        step(); fence()
        dt32 = time.perf_counter() - t1
        os.environ["INERF_PRECISION"] = "f16x3"
        if world > 1:
            t = torch.tensor([dt32], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt32 = float(t.item())
        exact = {"value": n_total / dt32, "unit": "rays/s", "ms_per_step": dt32 * 1e3, "steps": 1,
                 "note": "whole path with INERF_PRECISION=f32 (v_mfma_f32_32x32x2_f32 everywhere)"}

    # the reference's training step through the same front-end (rank 0, N = 1 only; untimed
    # relative to `value`): 1024 rays + one neighbour each (run_nerf.py:918-929), 64 + 128 samples, forward + backward + Adam
    train = None
    if rank == 0 and world == 1 and prec == _capi.PREC_F16X3:
        import warnings
        tnet_c, tnet_f = mk(), mk()
        query = ol.NetworkQuery(embed, embed_d)
        opt = torch.optim.Adam(list(tnet_c.parameters()) + list(tnet_f.parameters()), lr=5e-4)
        tr = rays_l[torch.randperm(rays_l.shape[0], device=dev)[:2048]].contiguous()
        target = torch.rand(tr.shape[0], 3, device=dev)

        def train_step():
            ret = ol.render_rays(tr, tnet_c, query, N_SAMPLES, retraw=True, perturb=1.0, N_importance=N_IMPORTANCE,
                                 network_fine=tnet_f, white_bkgd=True)
            loss = ((ret["rgb_map"] - target) ** 2).mean() + ((ret["rgb0"] - target) ** 2).mean() + 0.01 * ret["albedo_map"].abs().mean()
            opt.zero_grad()
            loss.backward()
            opt.step()

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(2):
                train_step()
            fence()
            t1 = time.perf_counter()
            for _ in range(5):
                train_step()
            fence()
        t_train = (time.perf_counter() - t1) / 5
        train = {"ms_per_step": t_train * 1e3, "rays": int(tr.shape[0]), "rays_per_s": tr.shape[0] / t_train,
                 "note": "the reference's training batch (2048 rays x (64+128) samples) through object_level.render_rays under "
                         "autograd: HIP forward + backward (networks, compositing) + torch Adam; not part of `value`"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        print(json.dumps({
            "metric": "rays/sec (64+128 samples/ray)", "value": rays_per_s, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (operands split into f16 hi+lo, 3 f16 MFMA products per MAC, fp32 accumulate)"
                     if prec == _capi.PREC_F16X3 else "f32", "data": "synthetic",
            "config": {"workload": "Blender chair 800x800 frame (640000 rays), 64 coarse + 128 importance samples, "
                                   "coarse+fine intrinsic NeRF (D=8, W=256), white_bkgd, eval mode, random-init weights "
                                   "(seeds 0/1)", "rays_per_step": n_total, "parallelism": f"ray-sharded x{world}",
                       "gather": "all_gather of 12 floats/ray" if world > 1 else "none"},
            "roofline": roofline, "roofline_f32_kernel": roofline_f32, "exact_f32_path": exact, "train_step": train,
            "cpu_baseline": cpu}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
