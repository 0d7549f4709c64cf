"""pytest configuration: registers the ``gpu`` marker and shared fixture helpers."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def torch_threads():
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def aten_gemm_watch():
    """Context manager that records every ATen matrix product dispatched inside it (``.gemms``): the render / training path must
    not contain any - its GEMMs are the library's HIP kernels."""
    from torch.utils._python_dispatch import TorchDispatchMode

    class Watch(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.gemms = []

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            if any(s in str(func) for s in ("aten.mm", "aten.addmm", "aten.bmm", "aten.linear", "aten.matmul", "aten.baddbmm", "aten.mv")):
                self.gemms.append(str(func))
            return func(*args, **(kwargs or {}))

    return Watch()


def assert_same_within(got, want, what, rel=2e-4):
    """|got - want| <= rel * max(1, max |want|) elementwise, NaNs coinciding: two fp32 evaluations of the same layers in another
    summation order (the tensors' own scale is the yardstick: a test that scales weights by 1e6 scales raw with them)."""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.equal(torch.isnan(got), torch.isnan(want)), f"{what}: NaN pattern differs"
    fin = torch.isfinite(want)
    assert torch.equal(got[~fin & ~torch.isnan(want)], want[~fin & ~torch.isnan(want)]), f"{what}: infinities differ"
    scale = max(1.0, float(want[fin].abs().max())) if fin.any() else 1.0
    worst = float((got[fin] - want[fin]).abs().max()) if fin.any() else 0.0
    assert worst <= rel * scale, f"{what}: max |diff| {worst:.3e} against a scale of {scale:.3e}"
