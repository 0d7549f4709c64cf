#!/usr/bin/env python3
"""Albedo-cluster lookup: one HIP launch vs the reference's per-class torch expressions on the same GPU.

    python scripts/bench_cluster.py [--classes 28] [--anchors 2000]

Two workloads: an SSR frame (320x240 pixels, labels in runs) and a training batch (1024 random pixels).  The torch side is
the oracle's restatement of Cluster_Manager.dest_color (cluster.py:73-86) run on the GPU - the same ATen calls the reference
makes: per class a boolean-mask gather, the [anchors, 10240] distance matrix, argmin, masked scatter.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (the torch restatement is the thing compared against, not part of the product path)
from intrinsicnerf_amd import cluster as ic  # noqa: E402


class Manager:
    def __init__(self, clusters):
        self.class_num, self.clusters = len(clusters), clusters


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", type=int, default=28)
    ap.add_argument("--anchors", type=int, default=2000)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    clusters = []
    for _ in range(args.classes):
        a = args.anchors
        anchors = ((torch.randint(0, 100, (a, 3), generator=g) + torch.rand(a, 3, generator=g)) * 0.01).float().to(dev)
        clusters.append({"anchors": anchors, "links": torch.randint(0, 6, (a, 1), generator=g).to(dev),
                         "rgb_centers": torch.rand(6, 3, generator=g).to(dev), "intensity_factor": 0.5, "batch_size": 10240})
    mgr = Manager(clusters)
    out = {"classes": args.classes, "anchors_per_class": args.anchors}
    for name, n, coherent in (("frame_320x240", 76800, True), ("train_batch_1024", 1024, False)):
        rgb = (torch.rand(n, 3, generator=g) * 0.95 + 0.02).to(dev)
        if coherent:
            label = torch.randint(0, args.classes, (n // 640 + 1,), generator=g).repeat_interleave(640)[:n, None].to(dev)
        else:
            label = torch.randint(0, args.classes, (n, 1), generator=g).to(dev)
        hip = timed(lambda: ic.dest_color(mgr, rgb, label), args.iters)
        ref = timed(lambda: oracle.cluster.dest_color(clusters, rgb, label), max(2, args.iters // 4))
        same = float((ic.dest_color(mgr, rgb, label) == oracle.cluster.dest_color(clusters, rgb, label)).all(1).float().mean())
        out[name] = {"hip_ms": hip, "torch_same_gpu_ms": ref, "speedup": ref / hip, "pixels_per_s": n / hip * 1e3,
                     "anchor_tests_per_s": n * args.anchors / hip * 1e3, "identical_pixels": same}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
