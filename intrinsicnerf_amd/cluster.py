"""Albedo-cluster lookup behind the reference's ``Cluster`` / ``Cluster_Manager`` interface.

``Cluster_Manager.dest_color(rgb, label)`` / ``dest_class(rgb, label)`` (SSR/training/cluster.py:73-98) run once per
training step (trainer.py:913-920) and once per rendered frame (:1427-1430).  The reference loops over the semantic
classes on the host - per class a boolean-mask gather (a host sync), the ``[anchors, 10240]`` distance matrix, an argmin
and a masked scatter.  Here every class goes through ONE launch of ``inerf_cluster_lookup`` (csrc/cluster.hip): the
anchors of all classes live in one device table and the distance matrix is never materialised.

Two ways in:

* ``dest_color(manager, rgb, label)`` / ``dest_class(manager, rgb, label)`` take ANY object with the reference's
  attributes (``class_num``, ``clusters[i].anchors / .links / .rgb_centers / .intensity_factor``), so the reference's own
  ``Cluster_Manager`` - including one that just ran its mean-shift ``update_center`` - can be handed over as it is;
* ``Cluster`` / ``Cluster_Manager`` below read and write the reference's ``clusters.json`` / ``c<i>/config.json`` files
  (cluster.py:20-50,112-129) and expose the same two lookups.  Building clusters (mean-shift on the CPU with sklearn,
  cluster.py:138-182) is training control plane and stays with the reference.

There is no CPU path: pixels that are not on a HIP device raise.
"""
import json
import os

import torch

from . import _capi
from .kernels import _dev, _ptr, _stream


class ClusterTables:
    """The device tables ``inerf_cluster_lookup`` reads, built from a manager's clusters."""

    def __init__(self, clusters, device):
        anchors, links, centers, factors = [], [], [], []
        a_begin, c_begin = [0], [0]
        for c in clusters:
            if _has(c):
                a = torch.as_tensor(_field(c, "anchors")).to(device=device, dtype=torch.float32).reshape(-1, 3)
                # |a|^2 with the reference's own expression (cluster.py:300-301), so that it rounds like the reference's
                sq = torch.sum(a ** 2, dim=1)
                lk = torch.as_tensor(_field(c, "links")).to(device=device).reshape(-1).to(torch.int32)
                ctr = torch.as_tensor(_field(c, "rgb_centers")).to(device=device, dtype=torch.float32).reshape(-1, 3)
                if lk.shape[0] != a.shape[0]:
                    raise ValueError("a cluster's links and anchors differ in length")
                if int(lk.max()) >= ctr.shape[0] or int(lk.min()) < 0:
                    raise ValueError("a cluster's links point outside its rgb_centers")
                anchors.append(torch.cat([a, sq[:, None]], 1))
                links.append(lk)
                centers.append(ctr)
                factors.append(float(_field(c, "intensity_factor")))
                a_begin.append(a_begin[-1] + a.shape[0])
                c_begin.append(c_begin[-1] + ctr.shape[0])
            else:                                   # `clusters[i] is None`: an empty anchor range
                factors.append(0.0)
                a_begin.append(a_begin[-1])
                c_begin.append(c_begin[-1])
        self.n_classes = len(clusters)
        self.device = torch.device(device)
        i32 = dict(dtype=torch.int32, device=device)
        self.anchors = torch.cat(anchors, 0).contiguous() if anchors else torch.zeros(1, 4, device=device)
        self.links = torch.cat(links, 0).contiguous() if links else torch.zeros(1, **i32)
        self.centers = torch.cat(centers, 0).contiguous() if centers else torch.zeros(1, 3, device=device)
        self.anchor_begin = torch.tensor(a_begin, **i32)
        self.center_begin = torch.tensor(c_begin, **i32)
        self.factor = torch.tensor(factors, dtype=torch.float32, device=device)


def _field(c, name):
    return c[name] if isinstance(c, dict) else getattr(c, name, None)


def _has(c):
    return c is not None and _field(c, "anchors") is not None and _field(c, "anchors").shape[0] > 0


def _signature(clusters, device):
    sig = [str(device)]
    for c in clusters:
        if not _has(c):
            sig.append(None)
            continue
        parts = []
        for name in ("anchors", "links", "rgb_centers"):
            t = _field(c, name)
            parts.append((t.data_ptr(), t._version, tuple(t.shape)) if isinstance(t, torch.Tensor) else id(t))
        sig.append((id(c), tuple(parts), float(_field(c, "intensity_factor"))))
    return tuple(sig)


_tables = {}        # id(owner) -> (signature, ClusterTables); rebuilt when the owner's clusters change


def tables_for(owner, clusters, device):
    """Device tables for ``clusters`` (a list with None for classes without a cluster), cached per owning manager."""
    device = torch.device(device)
    sig = _signature(clusters, device)
    hit = _tables.get(id(owner))
    if hit is None or hit[0] != sig:
        if len(_tables) > 64:
            _tables.clear()
        hit = (sig, ClusterTables(clusters, device))
        _tables[id(owner)] = hit
    return hit[1]


def lookup(tables, rgb, label=None, want_color=True, want_class=False, ignore_label=False):
    """One launch of ``inerf_cluster_lookup``: (colour [n,3] float32 or None, class [n] int64 or None)."""
    rgb = _dev(rgb, "rgb", (None, 3))
    n = rgb.shape[0]
    if rgb.device != tables.device:
        raise ValueError(f"rgb is on {rgb.device}, the cluster tables on {tables.device}")
    if not ignore_label:
        if label is None:
            raise ValueError("label is required unless ignore_label is set")
        label = label.reshape(-1)
        if label.shape[0] != n or not label.is_cuda:
            raise ValueError(f"label must hold {n} entries on {rgb.device}")
        label = label.to(torch.int64).contiguous()
    color = torch.empty(n, 3, dtype=torch.float32, device=rgb.device) if want_color else None
    cls = torch.empty(n, dtype=torch.int64, device=rgb.device) if want_class else None
    with torch.cuda.device(rgb.device):
        rc = _capi.lib().inerf_cluster_lookup(
            _ptr(rgb), None if ignore_label else _ptr(label), n, _ptr(tables.anchors), _ptr(tables.links),
            _ptr(tables.anchor_begin), _ptr(tables.factor), _ptr(tables.centers), _ptr(tables.center_begin),
            tables.n_classes, _capi.CLUSTER_IGNORE_LABEL if ignore_label else 0, _ptr(color), _ptr(cls), _stream(rgb))
    _capi.check(rc, "inerf_cluster_lookup")
    return color, cls


def dest_color(manager, rgb, label):
    """``Cluster_Manager.dest_color`` (cluster.py:73-86): every pixel replaced by the centre colour of the cluster its
    mapped colour falls into, per semantic class; pixels of classes without a cluster come back unchanged."""
    single = manager.class_num == 1
    clusters = list(manager.clusters)[:1] if single else list(manager.clusters)[:manager.class_num]
    color, _ = lookup(tables_for(manager, clusters, rgb.device), rgb, label, ignore_label=single)
    # the single-class shortcut returns Cluster.dest_color's squeezed tensor (cluster.py:75-77,285)
    return torch.squeeze(color) if single else color


def dest_class(manager, rgb, label):
    """``Cluster_Manager.dest_class`` (cluster.py:88-98): [n,1] int64 cluster index inside the pixel's class."""
    clusters = list(manager.clusters)[:manager.class_num]
    _, cls = lookup(tables_for(manager, clusters, rgb.device), rgb, label, want_color=False, want_class=True)
    return cls[:, None]


class Cluster:
    """Data holder with the reference's ``Cluster`` fields and file format (cluster.py:101-129)."""

    def __init__(self, device=None, intensity_factor=0.5, cluster_dir=None):
        self.batch_size = 10240          # kept for the file format; the HIP lookup has no batches
        self.anchors = None
        self.links = None
        self.rgb_centers = None
        self.device = torch.device("cuda") if device is None else torch.device(device)
        self.intensity_factor = intensity_factor
        if cluster_dir is not None:
            self.load(cluster_dir)

    def load(self, cluster_dir):
        with open(os.path.join(cluster_dir, "config.json"), "r") as f:
            data = json.load(f)
        self.batch_size = data["batch_size"]
        self.intensity_factor = data["intensity_factor"]
        self.anchors = torch.tensor(data["anchors"], dtype=torch.float32).reshape(-1, 3).to(self.device)
        self.rgb_centers = torch.tensor(data["rgb_centers"], dtype=torch.float32).reshape(-1, 3).to(self.device)
        self.links = torch.tensor(data["links"]).long().reshape(-1, 1).to(self.device)

    def save(self, cluster_dir):
        """Writes config.json like cluster.py:122-129 (the reference also drops a 50x50 PNG swatch per centre next to it)."""
        os.makedirs(cluster_dir, exist_ok=True)
        data = {"batch_size": self.batch_size, "intensity_factor": self.intensity_factor,
                "rgb_centers": self.rgb_centers.cpu().numpy().tolist(), "anchors": self.anchors.cpu().numpy().tolist(),
                "links": self.links.cpu().numpy().tolist()}
        with open(os.path.join(cluster_dir, "config.json"), "w") as f:
            json.dump(data, f)

    def dest_color(self, rgb):
        return torch.squeeze(lookup(tables_for(self, [self], rgb.device), rgb, ignore_label=True)[0])

    def dest_class(self, rgb):
        return lookup(tables_for(self, [self], rgb.device), rgb, want_color=False, want_class=True, ignore_label=True)[1][:, None]


class Cluster_Manager:
    """The reference's ``Cluster_Manager`` minus the mean-shift fitting (cluster.py:12-98)."""

    def __init__(self, class_num=0, cluster_config_file=None, device=None):
        self.class_num = class_num
        self.clusters = []
        self.device = device
        if cluster_config_file is not None:
            self.load(cluster_config_file)

    def load(self, cluster_config_file):
        with open(os.path.join(cluster_config_file, "clusters.json"), "r") as f:
            data = json.load(f)
        self.class_num = data["class_num"]
        configs = data["cluster_dirs"]
        assert self.class_num == len(configs)
        self.clusters = [None if cfg is None else Cluster(device=self.device, cluster_dir=os.path.join(cluster_config_file, "c" + str(i)))
                         for i, cfg in enumerate(configs)]

    def save(self, cluster_manager_dir):
        os.makedirs(cluster_manager_dir, exist_ok=True)
        dirs = []
        for i, cluster in enumerate(self.clusters):
            if cluster is None:
                dirs.append(None)
                continue
            d = os.path.join(cluster_manager_dir, "c" + str(i))
            cluster.save(d)
            dirs.append(d)
        with open(os.path.join(cluster_manager_dir, "clusters.json"), "w") as f:
            json.dump({"class_num": self.class_num, "cluster_dirs": dirs}, f)

    def dest_color(self, rgb, label):
        return dest_color(self, rgb, label)

    def dest_class(self, rgb, label):
        return dest_class(self, rgb, label)
