"""The zero-edit launcher (intrinsicnerf_amd/launch.py) against the REAL reference scripts, where they are mounted (the build
container; the GPU box has no reference and skips): the reference's entry scripts are loaded unmodified, the render path's
names in THEIR namespaces are this package's afterwards, and everything else - create_nerf, train, the main block - is still
the reference's own.  Runs in a subprocess: importing the reference needs stubs for packages this image lacks and a no-op
torch.cuda.set_device."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

PRELUDE = r'''
import importlib.machinery, os, sys, types
sys.dont_write_bytecode = True
import torch
def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); m.__spec__ = importlib.machinery.ModuleSpec(name, None); sys.modules[name] = m
for m in ("cv2", "imageio", "configargparse", "open3d"):
    stub(m)
stub("imgviz", label_colormap=lambda *a, **k: None, depth2rgb=lambda *a, **k: None, draw=types.ModuleType("draw"))
stub("torch.utils.tensorboard", SummaryWriter=object)
stub("skimage"); stub("skimage.io", imread=lambda *a, **k: None)          # the data loaders' image readers (never called here)
torch.cuda.set_device = lambda *a, **k: None          # run_nerf.py:10 in a container without a GPU
sys.path.insert(0, %(repo)r)
from intrinsicnerf_amd import launch, object_level as ol, ssr
'''

OBJECT = PRELUDE + r'''
import tempfile
mod, main = launch.prepare(%(ref)r + "/object_level/run_nerf.py")
assert mod.__name__ == "run_nerf" and mod.__file__.endswith("object_level/run_nerf.py")
for name in launch.OBJECT_SYMBOLS:
    assert getattr(mod, name) is getattr(ol, name), name
# ... and nothing else: the training loop, the network factory and the image loop are the reference's own functions
for name in ("train", "create_nerf", "render_path", "config_parser", "batchify"):
    assert getattr(mod, name).__code__.co_filename.endswith("object_level/run_nerf.py"), name
assert set(main.co_names) >= {"train", "torch", "np"}            # the script's own main block, compiled, not yet run
with tempfile.TemporaryDirectory() as base:
    os.makedirs(os.path.join(base, "exp"))
    args = types.SimpleNamespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, netdepth=8, netwidth=256,
                                 netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4, basedir=base, expname="exp", ft_path=None,
                                 no_reload=True, perturb=1.0, N_samples=64, white_bkgd=True, raw_noise_std=0.0, dataset_type="blender",
                                 no_ndc=False, lindisp=False)
    train_kw, test_kw, start, grad_vars, optimizer = mod.create_nerf(args)            # run_nerf.py:275-356, untouched
assert isinstance(train_kw["network_fn"], ol.NeRF) and isinstance(train_kw["network_fine"], ol.NeRF)
q = train_kw["network_query_fn"]
assert q.__code__.co_filename.endswith("object_level/run_nerf.py")                    # the reference's lambda (:298-301) ...
nq = ol._as_network_query(q)
assert nq is not None and isinstance(nq.embed_fn, ol.Embedder)                        # ... recognised: the fused path is taken
assert ol._fusable(train_kw["network_fn"], nq.embed_fn, nq.embeddirs_fn) is not None
print("object-level launcher ok")
'''

SSR = PRELUDE + r'''
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
mod, main = launch.prepare(%(ref)r + "/train_SSR_main.py")
assert mod.__name__ == "train_SSR_main" and mod.train.__code__.co_filename.endswith("train_SSR_main.py")
trainer = sys.modules["SSR.training.trainer"]
for name in launch.SSR_METHODS:
    assert getattr(trainer.SSRTrainer, name) is getattr(ssr.SSRRenderMixin, name), name
for name in ("step", "render_path", "init_rays", "set_params", "prepare_data_replica"):       # the rest of the trainer is the reference's
    assert getattr(trainer.SSRTrainer, name).__code__.co_filename.endswith("SSR/training/trainer.py"), name
mu, rays, sn = sys.modules["SSR.models.model_utils"], sys.modules["SSR.models.rays"], sys.modules["SSR.models.semantic_nerf"]
assert mu.run_network is ssr.run_network and mu.raw2outputs is ssr.raw2outputs
assert rays.sample_pdf is ssr.sample_pdf and rays.create_rays is ssr.create_rays
assert sn.Semantic_NeRF is ssr.Semantic_NeRF and sn.get_embedder is ssr.get_embedder
for name in ("Semantic_NeRF", "get_embedder", "run_network", "raw2outputs", "sample_pdf", "create_rays"):
    if hasattr(trainer, name):
        assert getattr(trainer, name) is getattr(ssr, name), name                     # the names trainer.py imported by value
from intrinsicnerf_amd import cluster as ic
cm = sys.modules["SSR.training.cluster"].Cluster_Manager
assert cm.dest_color is ic.dest_color and cm.dest_class is ic.dest_class and cm.update_center.__code__.co_filename.endswith("cluster.py")
assert "train" in main.co_names
# a trainer object built the way train_SSR_main.py builds it reaches the mixin through the reference's class
t = trainer.SSRTrainer.__new__(trainer.SSRTrainer)
assert t.render_rays.__func__ is ssr.SSRRenderMixin.render_rays and t.return_raw is True
print("ssr launcher ok")
'''

OBJECT_RP = PRELUDE + r"""
import numpy as np
mod, main = launch.prepare(%(ref)r + "/object_level/run_nerf.py", with_render_path=True)
rp = mod.render_path
assert rp.func is ol.render_path and rp.keywords["cluster_manager_factory"] is mod.Cluster_Manager      # run_nerf.py:24's class
assert mod.Cluster_Manager.__module__ == "cluster" and mod.to8b is ol.to8b
# run_nerf.py:818 / :1071 call render_path(..., update_cluster=True): a stub renderer and a recording stand-in for the mean-shift fit
seen = {}
mod.Cluster_Manager.update_center = lambda self, labels, pixels, band_factor=0.5: seen.update(labels=labels.shape, pixels=pixels.shape, b_f=band_factor)
mod.Cluster_Manager.dest_color = lambda self, rgb, label: rgb * 0.5
H = W = 8
def fake_render(H_, W_, K, chunk=0, c2w=None, **kw):
    g = torch.Generator().manual_seed(int(c2w[0, 3]))
    m = lambda c: torch.rand(H_, W_, c, generator=g) if c > 1 else torch.rand(H_, W_, generator=g)
    return [m(3), m(1), m(1), m(3), m(1), m(3), {}]
ol.render = fake_render
poses = [torch.cat([torch.eye(4)[:, :3], torch.full((4, 1), float(i))], 1) for i in range(3)]
rgbs, disps, cm = rp(poses, (H, W, 10.0), np.eye(3), 64, {}, update_cluster=True, b_f=0.25)
assert isinstance(cm, mod.Cluster_Manager) and cm.class_num == 1
assert rgbs.shape == (3, H, W, 3) and disps.shape == (3, H, W)
assert seen == {"labels": (3 * 16, 1), "pixels": (3 * 16, 3), "b_f": 0.25}, seen
print("object-level render_path ok")
"""

SSR_RP = PRELUDE + r"""
import numpy as np
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
mod, main = launch.prepare(%(ref)r + "/train_SSR_main.py", with_render_path=True)
trainer = sys.modules["SSR.training.trainer"]
assert trainer.SSRTrainer.render_path is ssr.SSRRenderMixin.render_path
t = trainer.SSRTrainer.__new__(trainer.SSRTrainer)
assert t.cluster_manager_factory is trainer.Cluster_Manager                                # trainer.py:16's class, not bound as a method
seen = {}
trainer.Cluster_Manager.update_center = lambda self, labels, pixels, band_factor=0.5: seen.update(labels=labels.shape, pixels=pixels.shape, n=self.class_num)
trainer.Cluster_Manager.dest_color = lambda self, rgb, label: rgb
H, W, C = 6, 8, 5
t.H_scaled, t.W_scaled, t.near, t.far, t.N_importance, t.enable_semantic, t.num_valid_semantic_class, t.no_semantic_tree = H, W, 0.1, 10.0, 128, True, C, False
def fake_render_rays(rays):
    n = rays.shape[0]
    g = torch.Generator().manual_seed(n)
    out = {k + "_fine": torch.rand(n, w, generator=g).squeeze(-1) for k, w in (("rgb", 3), ("disp", 1), ("depth", 1), ("albedo", 3), ("shading", 1), ("residual", 3))}
    out["sem_logits_fine"] = torch.randn(n, C, generator=g)
    return out
t.render_rays = fake_render_rays
ret = t.render_path(torch.zeros(2, H * W, 11), update_cluster=True)            # trainer.py:1065: update_cluster = not self.no_cluster
assert len(ret) == 12 and isinstance(ret[-1], trainer.Cluster_Manager) and seen["n"] == C, seen
assert ret[0].shape == (2, H, W, 3) and ret[4].shape == (2, H, W)
print("ssr render_path ok")
"""


def _run(code):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code % {"repo": REPO, "ref": REF}], capture_output=True, text=True, env=env, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    return out.stdout


@pytest.mark.skipif(not os.path.isdir(REF + "/object_level"), reason="reference not mounted")
def test_launcher_rebinds_run_nerf_and_nothing_else():
    assert "object-level launcher ok" in _run(OBJECT)


@pytest.mark.skipif(not os.path.isdir(REF + "/SSR"), reason="reference not mounted")
def test_launcher_rebinds_the_ssr_trainer_and_nothing_else():
    assert "ssr launcher ok" in _run(SSR)


def test_launcher_rejects_other_scripts(tmp_path):
    from intrinsicnerf_amd import launch
    other = tmp_path / "something.py"
    other.write_text("print('hi')\n")
    with pytest.raises(SystemExit):
        launch.prepare(str(other))
    body, main = launch._split_main("x = 1\nif __name__ == '__main__':\n    y = x + 1\n", "t.py")
    ns = {}
    exec(body, ns)
    assert ns["x"] == 1 and "y" not in ns
    exec(main, ns)
    assert ns["y"] == 2


@pytest.mark.skipif(not os.path.isdir(REF + "/object_level"), reason="reference not mounted")
def test_launcher_render_path_hands_the_reference_cluster_manager_to_the_mirror():
    """--inerf-render-path: the reference's loop calls render_path(update_cluster=True) (run_nerf.py:818,1071) - the mirror must
    have the reference's Cluster_Manager as its factory."""
    assert "object-level render_path ok" in _run(OBJECT_RP)


@pytest.mark.skipif(not os.path.isdir(REF + "/SSR"), reason="reference not mounted")
def test_launcher_render_path_gives_the_ssr_trainer_its_cluster_manager_factory():
    assert "ssr render_path ok" in _run(SSR_RP)
