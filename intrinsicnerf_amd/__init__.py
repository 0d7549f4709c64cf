"""intrinsicnerf_amd - MI355X (gfx950) implementation of IntrinsicNeRF's volumetric render_rays hot path.

  object_level   drop-in for object_level/run_nerf.py + run_nerf_helpers.py (render, render_rays, ...)
  ssr            drop-in for SSR/ + train_SSR_main.py's render path (SSRRenderMixin, Semantic_NeRF, ...)
  kernels        tensor-level launchers of the C ABI (include/inerf.h, libinerf.so)
  packing        state dict -> MFMA-fragment-ordered weight blob
  distributed    ray sharding across the GPUs of a node + RCCL gather of the rendered maps

The arithmetic lives in ``libinerf.so`` (hand-written HIP, built by ``intrinsicnerf_amd._build`` /
``__graft_entry__.build()``).  There is no CPU, eager or oracle fallback: without the library and a HIP
device the render entry points raise.
"""
from . import _capi, kernels, packing  # noqa: F401
from . import object_level, ssr  # noqa: F401

__version__ = "0.1.0"
