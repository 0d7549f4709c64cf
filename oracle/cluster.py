"""CPU oracle of the albedo-cluster lookup.  TEST INFRASTRUCTURE ONLY.

Plain torch restatement of ``SSR/training/cluster.py``: ``Cluster_Manager.dest_color`` (:73-86) / ``dest_class``
(:88-98) over ``Cluster.dest_color`` (:275-285) / ``dest_class`` (:287-297), ``compute_dist`` (:299-305),
``nearest_anchor`` (:307-310) and ``mapping_color`` (:324-330).  A cluster is a dict with the fields the reference's
``config.json`` holds (``anchors`` [A,3] float32 in the mapped colour space, ``links`` [A,1] int64, ``rgb_centers``
[M,3] float32, ``intensity_factor``, ``batch_size``); a manager is a list of such dicts with ``None`` for classes
without pixels.  Pinned by ``tests/golden/cluster_lookup.npz``, written by ``tests/golden/make_golden_cluster.py``
from the reference's own classes (index for index, bit for bit on CPU).
"""
import torch


def mapping_color(rgb, intensity_factor):
    """cluster.py:324-330."""
    intensity = torch.sum(rgb, axis=-1)
    d_rgb = torch.zeros_like(rgb)
    d_rgb[..., 0] = intensity / 3.0 * intensity_factor
    d_rgb[..., 1] = rgb[..., 1] / intensity
    d_rgb[..., 2] = rgb[..., 2] / intensity
    return d_rgb


def squared_distances(a, b):
    """cluster.py:299-305: |a|^2 + |b|^2 - 2 a.b, [m, n]."""
    sum_sq_a = torch.sum(a ** 2, dim=1).unsqueeze(1)
    sum_sq_b = torch.sum(b ** 2, dim=1).unsqueeze(0)
    return sum_sq_a + sum_sq_b - 2 * a.mm(b.t())


def nearest_anchor(cluster, rgb):
    """Index of the nearest anchor for every pixel (cluster.py:275-283,307-310), in the reference's batches."""
    d_rgb = mapping_color(rgb, cluster["intensity_factor"])
    idxs, start, step = [], 0, int(cluster.get("batch_size", 10240))
    while start < d_rgb.shape[0]:
        end = min(d_rgb.shape[0], start + step)
        idxs.append(torch.argmin(squared_distances(cluster["anchors"], d_rgb[start:end]), dim=0).long())
        start = end
    return torch.cat(idxs, 0)


def cluster_dest_color(cluster, rgb):
    """cluster.py:275-285."""
    return torch.squeeze(cluster["rgb_centers"][cluster["links"][nearest_anchor(cluster, rgb)]])


def cluster_dest_class(cluster, rgb):
    """cluster.py:287-297."""
    return cluster["links"][nearest_anchor(cluster, rgb)]


def dest_color(clusters, rgb, label):
    """Cluster_Manager.dest_color, cluster.py:73-86."""
    result = rgb.clone()
    if len(clusters) == 1:
        return cluster_dest_color(clusters[0], rgb)
    for i, cluster in enumerate(clusters):
        if cluster is None:
            continue
        class_idx = torch.squeeze(label == i)
        class_rgb = rgb[class_idx]
        if class_rgb.shape[0] == 0:
            continue
        result[class_idx] = cluster_dest_color(cluster, class_rgb).to(result.dtype).reshape(-1, 3)
    return result


def dest_class(clusters, rgb, label):
    """Cluster_Manager.dest_class, cluster.py:88-98."""
    result = torch.zeros([rgb.shape[0], 1], dtype=torch.long)
    for i, cluster in enumerate(clusters):
        if cluster is None:
            continue
        class_idx = torch.squeeze(label == i)
        class_rgb = rgb[class_idx]
        if class_rgb.shape[0] == 0:
            continue
        result[class_idx] = cluster_dest_class(cluster, class_rgb)
    return result


def gap_to_runner_up(cluster, rgb, chosen):
    """fp64 distance of the ``chosen`` anchors minus the fp64 minimum, and the scale fp32 rounding of the reference's
    expression is measured against (|a|^2 + |b|^2): the tolerance of an index result under fp32 arithmetic."""
    d = mapping_color(rgb, cluster["intensity_factor"]).double()
    a = cluster["anchors"].double()
    dist = ((a[:, None, :] - d[None, :, :]) ** 2).sum(-1)                     # [A, n]
    scale = (a ** 2).sum(1).max() + (d ** 2).sum(1)
    return dist.gather(0, chosen.reshape(1, -1))[0] - dist.min(0).values, scale
