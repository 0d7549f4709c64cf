"""CPU oracle for the IntrinsicNeRF volumetric render path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``intrinsicnerf_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

The oracle is a plain PyTorch-CPU restatement (fp32 by default, fp64 on request)
of the reference algorithm.  The reference is itself pure Python on ATen, so a
torch restatement shares its arithmetic kernels (``F.linear``, ``cumprod``,
``searchsorted`` ...) - there is no C restatement because there is no C in the
reference path to restate.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4).  The oracle is pinned instead against outputs of the
reference itself, imported in the build container by
``tests/golden/make_golden.py`` (which asserts oracle == reference and writes
the ``tests/golden/*.npz`` fixtures the CPU test-suite replays).
"""
from .intrinsic_render import (  # noqa: F401
    RenderConfig,
    freq_encode,
    mlp_forward,
    query_network,
    composite,
    inverse_cdf_sample,
    coarse_depths,
    render_rays,
    make_state_dict,
    state_dict_spec,
    lcg_state_dict,
)
from .conditioning import calibrated_lcg_weights, conditioning_scores  # noqa: F401
from . import cluster  # noqa: F401,E402
