"""GPU: the four stand-alone reference-signature wrappers, tuple by tuple (VERDICT r02 missing #6).

``object_level.raw2outputs`` / ``sample_pdf`` (run_nerf.py:359-412, run_nerf_helpers.py:402-445) and ``ssr.raw2outputs`` /
``sample_pdf`` (model_utils.py:39-116, rays.py:176-220): position of every element of the returned tuples against the
reference-generated stage fixtures (tests/golden/stage_*.npz - outputs of the reference's own functions), the ``pytest=``
hooks, ``det`` handling, the ``torch.tensor(0)`` placeholders, leading batch dimensions and the order of RNG draws."""
import numpy as np
import pytest
import torch

import oracle
from _cases import assert_maps_close
from conftest import load_golden

pytestmark = pytest.mark.gpu
RTOL, ATOL, RTOL_DISP = 1e-4, 1e-5, 5e-4
DEV = "cuda:0"


def _rtol(k):
    return RTOL_DISP if k == "disp" else RTOL


@pytest.mark.parametrize("wb", [0, 1])
def test_object_raw2outputs_tuple(wb):
    from intrinsicnerf_amd import object_level as ol
    fx = load_golden(f"stage_composite_object_wb{wb}")
    raw, z, d = (torch.from_numpy(fx[k]).to(DEV) for k in ("raw", "z", "rays_d"))
    out = ol.raw2outputs(raw, z, d, 0, bool(fx["white_bkgd"]))
    order = ("rgb", "disp", "acc", "weights", "depth", "albedo", "shading", "residual")           # run_nerf.py:412
    assert isinstance(out, tuple) and len(out) == 8
    for got, k in zip(out, order):
        assert got.shape == fx["ref_" + k].shape, k
        assert_maps_close(got.cpu().numpy(), fx["ref_" + k], _rtol(k), ATOL, f"raw2outputs[{k}]")
    # raw_noise_std > 0: one torch.randn draw of raw[..., 3]'s shape, scaled (run_nerf.py:386-387) ...
    cfg = oracle.RenderConfig(variant="object", white_bkgd=bool(fx["white_bkgd"]))
    torch.manual_seed(3)
    out_n = ol.raw2outputs(raw, z, d, 0.5, bool(fx["white_bkgd"]))
    torch.manual_seed(3)
    noise = torch.randn(raw[..., 3].shape, device=DEV) * 0.5
    want = oracle.composite(raw.cpu(), z.cpu(), d.cpu(), cfg, noise=noise.cpu())
    for got, k in zip(out_n, order):
        assert_maps_close(got.cpu().numpy(), want[k].numpy(), _rtol(k), ATOL, f"noisy raw2outputs[{k}]")
    # ... and the pytest hook replaces it by np.random.seed(0); np.random.rand(...) * std (:389-393)
    out_p = ol.raw2outputs(raw, z, d, 0.5, bool(fx["white_bkgd"]), pytest=True)
    np.random.seed(0)
    noise_p = torch.Tensor(np.random.rand(*raw[..., 3].shape) * 0.5)
    want = oracle.composite(raw.cpu(), z.cpu(), d.cpu(), cfg, noise=noise_p)
    for got, k in zip(out_p, order):
        assert_maps_close(got.cpu().numpy(), want[k].numpy(), _rtol(k), ATOL, f"pytest raw2outputs[{k}]")


@pytest.mark.parametrize("wb", [0, 1])
def test_ssr_raw2outputs_tuple(wb):
    from intrinsicnerf_amd import ssr
    fx = load_golden(f"stage_composite_ssr_wb{wb}")
    raw, z, d = (torch.from_numpy(fx[k]).to(DEV) for k in ("raw", "z", "rays_d"))
    c = int(fx["n_classes"])
    out = ssr.raw2outputs(raw, z, d, 0, bool(fx["white_bkgd"]), enable_semantic=True, num_sem_class=c, endpoint_feat=True)
    order = ("rgb", "disp", "acc", "weights", "depth", "sem", "feat", "albedo", "shading", "residual")      # model_utils.py:116
    assert isinstance(out, tuple) and len(out) == 10
    for got, k in zip(out, order):
        assert got.shape == fx["ref_" + k].shape, k
        assert_maps_close(got.cpu().numpy(), fx["ref_" + k], _rtol(k), ATOL, f"raw2outputs[{k}]")
    # disabled heads come back as torch.tensor(0) (model_utils.py:95-96,103), the other eight elements are unchanged
    off = ssr.raw2outputs(raw, z, d, 0, bool(fx["white_bkgd"]), enable_semantic=False, num_sem_class=0, endpoint_feat=False)
    assert off[5].dim() == 0 and int(off[5]) == 0 and off[6].dim() == 0 and int(off[6]) == 0
    for i, k in enumerate(order):       # (the kernel sums depth / acc in another lane order when the extra heads ride along: ulps)
        if k not in ("sem", "feat"):
            assert_maps_close(off[i].cpu().numpy(), out[i].cpu().numpy(), 1e-5, 1e-7, f"heads off vs on: {k}")
    with pytest.raises(AssertionError):
        ssr.raw2outputs(raw, z, d, 0, False, enable_semantic=True, num_sem_class=0)              # model_utils.py:53-54
    # training noise: one torch.randn draw (model_utils.py:70-72)
    torch.manual_seed(9)
    out_n = ssr.raw2outputs(raw, z, d, 1.0, bool(fx["white_bkgd"]), enable_semantic=True, num_sem_class=c, endpoint_feat=True)
    torch.manual_seed(9)
    noise = torch.randn(raw[..., 3].shape, device=DEV)
    cfg = oracle.RenderConfig(variant="ssr", white_bkgd=bool(fx["white_bkgd"]), n_classes=c)
    want = oracle.composite(raw.cpu(), z.cpu(), d.cpu(), cfg, noise=noise.cpu(), feat=True)
    for got, k in zip(out_n, order):
        assert_maps_close(got.cpu().numpy(), want[k].numpy(), _rtol(k), ATOL, f"noisy raw2outputs[{k}]")


def test_object_sample_pdf_det_pytest_and_batch_dims():
    from intrinsicnerf_amd import object_level as ol
    fx = load_golden("stage_sample_pdf")
    bins, w = torch.from_numpy(fx["bins"]).to(DEV), torch.from_numpy(fx["weights"]).to(DEV)
    det = ol.sample_pdf(bins, w, 128, det=True)
    assert det.shape == (16, 128)
    assert_maps_close(det.cpu().numpy(), fx["ref_det"], RTOL, ATOL, "sample_pdf det")
    # det + pytest: u = np.linspace (run_nerf_helpers.py:416-420) - the same samples
    det_p = ol.sample_pdf(bins, w, 128, det=True, pytest=True)
    assert_maps_close(det_p.cpu().numpy(), fx["ref_det"], RTOL, ATOL, "sample_pdf det pytest")
    # not det + pytest: np.random.seed(0); u = np.random.rand(N, n) (:421-425)
    rnd_p = ol.sample_pdf(bins, w, 128, det=False, pytest=True)
    np.random.seed(0)
    u = torch.Tensor(np.random.rand(16, 128))
    want = oracle.inverse_cdf_sample(bins.cpu(), w.cpu(), u)
    ok = np.arange(16) != 4                                # ray 4 of this fixture has cdf entries that u can hit exactly (see test_sample_pdf_edge_cases)
    assert_maps_close(rnd_p.cpu().numpy()[ok], want.numpy()[ok], RTOL, ATOL, "sample_pdf pytest rnd")
    # not det: one torch.rand(N, n) draw on the bins' device (:414)
    torch.manual_seed(21)
    rnd = ol.sample_pdf(bins, w, 64, det=False)
    torch.manual_seed(21)
    u = torch.rand(16, 64, device=DEV)
    want = oracle.inverse_cdf_sample(bins.cpu(), w.cpu(), u.cpu())
    assert rnd.shape == (16, 64)
    assert_maps_close(rnd.cpu().numpy()[ok], want.numpy()[ok], RTOL, ATOL, "sample_pdf rnd")
    # leading batch dimensions are flattened and restored (the reference expands u to bins.shape[:-1], :410,413)
    lead = ol.sample_pdf(bins.reshape(2, 8, 63), w.reshape(2, 8, 62), 128, det=True)
    assert lead.shape == (2, 8, 128) and torch.equal(lead.reshape(16, 128), det)


def test_ssr_sample_pdf_det_and_random():
    from intrinsicnerf_amd import ssr
    fx = load_golden("stage_sample_pdf")
    bins, w = torch.from_numpy(fx["bins"]).to(DEV), torch.from_numpy(fx["weights"]).to(DEV)
    det = ssr.sample_pdf(bins, w, 128, det=True)
    assert_maps_close(det.cpu().numpy(), fx["ref_det"], RTOL, ATOL, "ssr sample_pdf det")
    torch.manual_seed(5)
    rnd = ssr.sample_pdf(bins, w, 128, det=False)
    torch.manual_seed(5)
    u = torch.rand(16, 128, device=DEV)                     # rays.py:197: on bins.device
    want = oracle.inverse_cdf_sample(bins.cpu(), w.cpu(), u.cpu())
    ok = np.arange(16) != 4
    assert_maps_close(rnd.cpu().numpy()[ok], want.numpy()[ok], RTOL, ATOL, "ssr sample_pdf rnd")
    assert np.all(np.abs(np.sort(rnd.cpu().numpy()[4]) - np.sort(want.numpy()[4])) < 0.08)
