#!/usr/bin/env python3
"""BASELINE.json configs[3]: a Replica room_0-like 320x240 frame through the SSR front-end (Semantic_NeRF with C = 28
classes, reflectance + shading + residual + semantic heads, 64 + 128 samples, depth range [0.1, 10], xyz / 10 encoding,
eval mode).  Synthetic camera and random-init weights (the dataset is not available here).

    python scripts/bench_ssr_frame.py [--frames 3] [--classes 28]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/bench_ssr_frame.py
        (configs[4]: the frame's rays in 8 row bands, one process per GPU, one RCCL all-gather of the 82 floats per ray)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.build()
from intrinsicnerf_amd import ssr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--classes", type=int, default=28)
ap.add_argument("--chunk", type=int, default=32768, help="rays per render_rays chunk (SSR_room0_config.yaml: 32768; results do not depend on it)")
a = ap.parse_args()
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
share = os.environ.get("INERF_BENCH_SHARE_GPU") == "1"      # debug: every rank on device 0 over gloo (1-GPU box)
if share:
    local_rank = 0
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world)
from intrinsicnerf_amd import distributed as idist  # noqa: E402
torch.manual_seed(0)                                       # the same networks on every rank
H, W = 240, 320
fx = fy = W / 2.0 / np.tan(np.deg2rad(45.0))          # hfov 90 deg (trainer.py:68-74)
cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
T = torch.eye(4)[None]                                   # rays are built on the host and moved, as trainer.py:608-624 does
rays = ssr.create_rays(1, T, H, W, fx, fy, cx, cy, 0.1, 10.0).reshape(-1, 11).contiguous().to(dev)
r = ssr.SSRRenderer(a.classes, white_bkgd=False, endpoint_feat=False, device=dev)
r.return_raw = False
r.chunk = a.chunk
r.check_numerics = False
layout = idist.ssr_map_layout(a.classes)


def frame():
    return idist.render_sharded(r.render_rays, rays, layout) if world > 1 else r.render_rays(rays)


def fence():
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


with torch.no_grad():
    ret = frame()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.frames):
        ret = frame()
    fence()
    dt = (time.perf_counter() - t0) / a.frames
if world > 1:
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
assert ret["rgb_fine"].shape[0] == rays.shape[0]
if rank != 0:
    dist.destroy_process_group()
    sys.exit(0)
flop = 2 * (659456 + 32768 + 128 * a.classes) * rays.shape[0] * 256
print(f"SSR frame {W}x{H} = {rays.shape[0]} rays on {world} GPU(s), C = {a.classes}, 64+128 samples: {dt * 1e3:.1f} ms per frame -> {rays.shape[0] / dt:.0f} rays/s "
      f"({flop / dt / 1e12:.0f} TFLOP/s algorithmic over the whole frame); keys: {sorted(ret.keys())[:4]}...")
if world > 1:
    dist.destroy_process_group()
