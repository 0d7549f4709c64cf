#!/usr/bin/env python3
"""Golden vectors for the albedo-cluster lookup, from the reference's own classes (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cluster.py

Builds clusters with the reference's ``Cluster.update_center`` (mean-shift + voxel-filtered anchors, SSR/training/cluster.py:
138-182) on synthetic albedo samples, hangs them on a ``Cluster_Manager`` (one class left without a cluster, as
``update_center`` leaves classes without pixels, :61-65) and records ``dest_color`` / ``dest_class`` (:73-98) for query pixels
that include labels without a cluster, labels outside the class range and zero-intensity pixels (NaN distances).  A second,
single-class manager covers the ``class_num == 1`` shortcut (:75-77).  Asserts that oracle/cluster.py reproduces every output
exactly, then writes tests/golden/cluster_lookup.npz.
"""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402


def albedo_samples(rng, n, n_modes):
    base = rng.uniform(0.08, 0.9, size=(n_modes, 3))
    pick = rng.integers(0, n_modes, size=n)
    shade = rng.uniform(0.6, 1.2, size=(n, 1))
    return np.clip(base[pick] * shade + rng.normal(0, 0.02, size=(n, 3)), 0.01, 1.0).astype(np.float32)


def as_dict(c):
    return {"anchors": c.anchors.float(), "links": c.links.long(), "rgb_centers": c.rgb_centers.float(),
            "intensity_factor": float(c.intensity_factor), "batch_size": int(c.batch_size)}


def main():
    mg.import_reference()
    from SSR.training import cluster as ref
    import oracle
    cpu = torch.device("cpu")
    rng = np.random.default_rng(20220414)
    K = 5
    mgr = ref.Cluster_Manager(class_num=K)
    factors = [0.5, 0.5, 0.35, 0.5, 0.8]
    for i in range(K):
        if i == 3:
            mgr.clusters.append(None)
            continue
        c = ref.Cluster(device=cpu, intensity_factor=factors[i])
        c.batch_size = 1000                                   # several batches per class
        with contextlib.redirect_stdout(io.StringIO()):
            c.update_center(albedo_samples(rng, 1500, 2 + i), n_samples=400)
        c.anchors, c.rgb_centers = c.anchors.float(), c.rgb_centers.float()     # what load() makes of a saved cluster (:117-119)
        mgr.clusters.append(c)
    n = 4000
    rgb = torch.from_numpy(albedo_samples(rng, n, 9))
    label = torch.from_numpy(rng.integers(0, K, size=(n, 1)))
    label[::97] = K + 2                                         # outside the class range: left alone
    label[5::131] = -1
    rgb[7] = 0.0                                                # zero intensity: 0/0 -> NaN distances -> anchor 0
    rgb[8] = torch.tensor([0.0, 0.3, -0.3])                     # zero intensity, non-zero channels -> inf/NaN
    label[7], label[8] = 0, 2
    out = {"K": K, "rgb": rgb.numpy(), "label": label.numpy()}
    for i, c in enumerate(mgr.clusters):
        if c is None:
            continue
        out[f"c{i}_anchors"], out[f"c{i}_links"] = c.anchors.numpy(), c.links.numpy()
        out[f"c{i}_centers"], out[f"c{i}_factor"], out[f"c{i}_batch"] = c.rgb_centers.numpy(), c.intensity_factor, c.batch_size
    out["dest_color"] = mgr.dest_color(rgb, label).numpy()
    out["dest_class"] = mgr.dest_class(rgb, label).numpy()
    clusters = [None if c is None else as_dict(c) for c in mgr.clusters]
    assert np.array_equal(oracle.cluster.dest_color(clusters, rgb, label).numpy(), out["dest_color"], equal_nan=True)
    assert np.array_equal(oracle.cluster.dest_class(clusters, rgb, label).numpy(), out["dest_class"])
    # class_num == 1
    one = ref.Cluster_Manager(class_num=1)
    one.clusters = [mgr.clusters[4]]
    out["single_color"] = one.dest_color(rgb, label).numpy()
    out["single_class"] = one.dest_class(rgb, label).numpy()
    assert np.array_equal(oracle.cluster.dest_color([clusters[4]], rgb, label).numpy(), out["single_color"], equal_nan=True)
    assert np.array_equal(oracle.cluster.dest_class([clusters[4]], rgb, label).numpy(), out["single_class"])
    out["single_color_one_pixel"] = one.dest_color(rgb[:1], label[:1]).numpy()          # squeeze() quirk: shape [3]
    np.savez_compressed(os.path.join(HERE, "cluster_lookup.npz"), **out)
    sizes = {i: tuple(c.anchors.shape) for i, c in enumerate(mgr.clusters) if c is not None}
    print("anchors per class:", sizes, "centres:", {i: c.rgb_centers.shape[0] for i, c in enumerate(mgr.clusters) if c is not None})
    print("wrote cluster_lookup.npz", os.path.getsize(os.path.join(HERE, "cluster_lookup.npz")), "bytes")


if __name__ == "__main__":
    main()
