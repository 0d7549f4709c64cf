"""Shared helpers that rebuild a golden case's inputs (weights from seed, config, random tensors)."""
import numpy as np
import torch

import oracle


def case_config(fx):
    variant = str(fx["variant"])
    return oracle.RenderConfig(
        variant=variant, n_samples=64, n_importance=int(fx["n_importance"]),
        white_bkgd=bool(fx["white_bkgd"]), lindisp=bool(fx["lindisp"]),
        n_classes=int(fx["n_classes"]), endpoint_feat=bool(fx["endpoint_feat"]),
        netchunk=32768 if variant == "ssr" else 65536)


def case_weights(fx):
    variant, c, seed = str(fx["variant"]), int(fx["n_classes"]), int(fx["seed"])
    kw = dict(sigma_gain_log2=int(fx["sigma_gain_log2"]), weight_gain_log2=int(fx["weight_gain_log2"]), freq_decay=True)
    sd_c = oracle.lcg_state_dict(variant, c, seed=2 * seed, sigma_bias=float(fx["sigma_bias_coarse"]), **kw)
    sd_f = oracle.lcg_state_dict(variant, c, seed=2 * seed + 1, sigma_bias=float(fx["sigma_bias_fine"]), **kw)
    return sd_c, sd_f


def case_random_inputs(fx, device="cpu"):
    out = {}
    for k in ("t_rand", "noise_coarse", "noise_fine", "u"):
        if "in_" + k in fx:
            out[k] = torch.from_numpy(fx["in_" + k]).to(device)
    return out


# oracle key -> tolerance class.  ``disp`` is ill-conditioned (1/(depth/acc)); the reference's own
# fp32-vs-fp64 noise floor for it is 8.9e-5, so it gets a looser bound.
def assert_maps_close(got, want, rtol, atol, tag=""):
    """``|got - want| <= atol + rtol * |want|`` with NaNs required at identical positions."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{tag}: shape {got.shape} vs {want.shape}"
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), f"{tag}: NaN pattern differs ({nan_g.sum()} vs {nan_w.sum()})"
    ok = ~nan_w
    err = np.abs(got[ok] - want[ok])
    bound = atol + rtol * np.abs(want[ok])
    if err.size and not np.all(err <= bound):
        i = int(np.argmax(err - bound))
        raise AssertionError(f"{tag}: max violation err={err[i]:.3e} bound={bound[i]:.3e} "
                             f"(want {want[ok][i]:.6g}, got {got[ok][i]:.6g}); max err {err.max():.3e}")


class injected_np_rand:
    """Feed the fixture's random tensors to ``np.random.rand`` calls, in call order.

    The object-level front-end keeps the reference's ``pytest=True`` hooks (np.random.seed(0) followed by
    np.random.rand, run_nerf.py:389-393,480-484; run_nerf_helpers.py:416-425).  The golden generator
    ran the reference with these same tensors injected the same way (tests/golden/make_golden.py).
    """

    def __init__(self, tensors):
        self.q = list(tensors)

    def __enter__(self):
        self.saved = np.random.rand
        def fake(*shape):
            t = self.q.pop(0)
            assert tuple(t.shape) == tuple(shape), (tuple(t.shape), shape)
            return t.double().cpu().numpy()
        np.random.rand = fake
        return self

    def __exit__(self, *exc):
        np.random.rand = self.saved
        assert exc[0] is not None or not self.q, "front-end made fewer RNG draws than the reference"
