"""CPU: the un-curated-parity tooling (oracle/calibration.py) - the calibrated default-init network is deterministic and
non-degenerate, and the rank-statistics comparison accepts like-distributed errors and rejects worse ones."""
import numpy as np
import torch

import oracle
from oracle import calibration as cal


def _rays(n=96):
    g = torch.Generator().manual_seed(3)
    o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
    d = -o / o.norm(dim=-1, keepdim=True) + 0.15 * torch.randn(n, 3, generator=g)
    return torch.cat([o, d, 2.0 * torch.ones(n, 1), 6.0 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1)


def test_calibrated_default_init_is_default_init_plus_density_head(torch_threads):
    rays = _rays()
    sd = cal.calibrated_default_init("object", 0, 0, rays)
    base = oracle.make_state_dict("object", 0, seed=0)
    for k in base:
        if not k.startswith("alpha_linear"):
            assert torch.equal(sd[k], base[k]), k
    ratio = sd["alpha_linear.weight"] / base["alpha_linear.weight"]
    g = float(ratio.flatten()[0])
    assert torch.all(ratio == g) and np.log2(g) == round(np.log2(g))             # one power-of-two gain
    assert float(sd["alpha_linear.bias"] * 1024) == round(float(sd["alpha_linear.bias"] * 1024))
    again = cal.calibrated_default_init("object", 0, 0, rays)
    assert all(torch.equal(sd[k], again[k]) for k in sd)
    with torch.no_grad():
        out = oracle.render_rays(rays, sd, sd, oracle.RenderConfig(variant="object", white_bkgd=True))
    acc = out["acc_fine"]
    assert float(acc.min()) < 0.95 and float(acc.max()) > 0.999      # spans (0, 1]: not the all-background frame of seeds 0/1
    with torch.no_grad():
        plain = oracle.render_rays(rays, base, base, oracle.RenderConfig(variant="object", white_bkgd=True))
    assert float(plain["acc_fine"].max()) == 0.0                          # what VERDICT r01 pointed out


def test_rank_report_accepts_same_distribution_and_rejects_worse():
    rng = np.random.RandomState(0)
    heavy = lambda n, s: np.abs(rng.standard_cauchy(n)) * s                # heavy-tailed, like ill-conditioned rays
    ref = heavy(2000, 0.05)
    assert cal.rank_report(heavy(2000, 0.05), ref) == []
    assert cal.rank_report(heavy(2000, 0.05) * 1.4, ref) == []            # a difference of two fp32 evaluations
    assert cal.rank_report(heavy(2000, 0.05) * 10, ref) != []
    # an implementation that is exact where the reference is exact may use the floor, not more
    assert cal.rank_report(np.full(100, 0.4), np.zeros(100)) == []
    assert cal.rank_report(np.full(100, 0.6), np.zeros(100)) != []
    # NaN-pattern mismatches (inf) count as tail events
    bad = np.zeros(100); bad[:10] = np.inf
    assert cal.rank_report(bad, np.zeros(100)) != []
    e = cal.scaled_errors(np.array([[1.0, np.nan], [1.0, 2.0]]), np.array([[1.0, 3.0], [1.0001, 2.0]]))
    assert np.isinf(e[0]) and abs(e[1] - 1e-4 / (1e-5 + 1e-4 * 1.0001)) < 1e-9
