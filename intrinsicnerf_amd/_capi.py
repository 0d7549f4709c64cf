"""ctypes binding of the C ABI declared in ``include/inerf.h``.

There is deliberately NO fallback: if ``libinerf.so`` is missing or does not load, importing the
render front-ends raises.  Tensors cross the boundary as raw device pointers (``Tensor.data_ptr()``)
and the current HIP stream handle; torch only owns memory and streams.
"""
import ctypes as C
import os

from . import _build
from ._build import LIB_PATH

ABI_VERSION = 40008          # INERF_ABI_VERSION of include/inerf.h these ctypes declarations mirror

OK, E_INVALID, E_UNSUPPORTED, E_WORKSPACE, E_HIP = 0, -1, -2, -3, -4
VARIANT_OBJECT, VARIANT_SSR = 0, 1
PREC_F32, PREC_F16X3 = 0, 1
STATUS_F16_RANGE = 1
FLAG_WHITE_BKGD, FLAG_LINDISP, FLAG_ENDPOINT, FLAG_U_PER_RAY, FLAG_BINS_DIRECT = 1, 2, 4, 8, 16
CLUSTER_IGNORE_LABEL = 1
CAM_OPENGL = 1
BASE_CHANNELS, ENDPOINT_DIM, RAY_FLOATS, MAX_CLASSES = 11, 128, 11, 240

_ERR = {E_INVALID: "invalid argument", E_UNSUPPORTED: "unsupported configuration",
        E_WORKSPACE: "workspace missing or too small", E_HIP: "HIP runtime error"}


class NetDesc(C.Structure):
    _fields_ = [("variant", C.c_int32), ("n_classes", C.c_int32), ("l_xyz", C.c_int32), ("l_dir", C.c_int32),
                ("xyz_div", C.c_float), ("precision", C.c_int32)]


class CompositeOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in
                ("rgb", "disp", "acc", "depth", "albedo", "shading", "residual", "sem", "feat", "weights")]


class RenderArgs(C.Structure):
    _fields_ = [("net", NetDesc), ("packed_coarse", C.c_void_p), ("packed_fine", C.c_void_p),
                ("rays", C.c_void_p), ("n_rays", C.c_int64), ("n_samples", C.c_int32), ("n_importance", C.c_int32),
                ("flags", C.c_uint32), ("t_vals", C.c_void_p), ("t_rand", C.c_void_p), ("u", C.c_void_p),
                ("noise_coarse", C.c_void_p), ("noise_fine", C.c_void_p),
                ("coarse", CompositeOut), ("fine", CompositeOut), ("z_std", C.c_void_p),
                ("raw_coarse", C.c_void_p), ("raw_fine", C.c_void_p), ("z_coarse", C.c_void_p),
                ("z_samples", C.c_void_p), ("z_fine", C.c_void_p), ("status", C.c_void_p), ("status_rays", C.c_int64),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


class LinearArgs(C.Structure):
    _fields_ = [("a", C.c_void_p), ("a_sm", C.c_int64), ("a_sk", C.c_int64), ("b", C.c_void_p), ("b_sn", C.c_int64), ("b_sk", C.c_int64),
                ("bias", C.c_void_p), ("add", C.c_void_p), ("add_ld", C.c_int64), ("gate", C.c_void_p), ("gate_ld", C.c_int64),
                ("c", C.c_void_p), ("c_ld", C.c_int64), ("m", C.c_int64), ("n", C.c_int32), ("act", C.c_int32), ("k", C.c_int64)]


ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2

# every symbol include/inerf.h declares: (restype, argtypes)
_P, _I, _L, _U = C.c_void_p, C.c_int, C.c_int64, C.c_uint32
SYMBOLS = {
    "inerf_version": (C.c_char_p, []),
    "inerf_abi_version": (_I, []),
    "inerf_build_digest": (C.c_char_p, []),
    "inerf_last_hip_error": (_I, []),
    "inerf_num_tensors": (_I, [C.POINTER(NetDesc)]),
    "inerf_tensor_info": (_I, [C.POINTER(NetDesc), _I, C.POINTER(C.c_char_p), C.POINTER(_L), C.POINTER(_L)]),
    "inerf_packed_floats": (_L, [C.POINTER(NetDesc)]),
    "inerf_raw_channels": (_I, [C.POINTER(NetDesc), _U, _I]),
    "inerf_pack_weights": (_I, [C.POINTER(NetDesc), C.POINTER(_P), _I, _P, _L]),
    "inerf_sample_coarse": (_I, [_P, _P, _P, _L, _I, _U, _P, _P]),
    "inerf_encode_mlp": (_I, [C.POINTER(NetDesc), _P, _P, _P, _L, _I, _U, _P, _P, _P]),
    "inerf_encode_mlp_workspace_bytes": (_L, [C.POINTER(NetDesc), _L, _I, _U]),
    "inerf_encode_mlp_ws": (_I, [C.POINTER(NetDesc), _P, _P, _P, _L, _I, _U, _P, _P, _P, _L, _P]),
    "inerf_encode_mlp_chunked": (_I, [C.POINTER(NetDesc), _P, _P, _P, _L, _I, _U, _P, _P, _L, _P, _L, _P]),
    "inerf_mlp_save_floats": (_L, [C.POINTER(NetDesc), _L]),
    "inerf_mlp_save_slot": (_I, [C.POINTER(NetDesc), _I, _L, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "inerf_encode_mlp_train": (_I, [C.POINTER(NetDesc), _P, _P, _P, _L, _I, _U, _P, _P, _P, _P, _P]),
    "inerf_bwd_packed_floats": (_L, [C.POINTER(NetDesc)]),
    "inerf_pack_weights_bwd": (_I, [C.POINTER(NetDesc), C.POINTER(_P), _I, _P, _L]),
    "inerf_mlp_backward_inputs": (_I, [C.POINTER(NetDesc), _P, _P, _P, _P, _L, _U, _P, _P, _P, _P, _P]),
    "inerf_mlp_head_partial_floats": (_I, []),
    "inerf_mlp_backward_grid": (_I, [_L]),
    "inerf_wgrad_grid": (_I, [_L]),
    "inerf_mlp_weight_gradient": (_I, [_P, _I, _P, _I, _L, _I, _I, _P, _P, _P, _L, _P]),
    "inerf_mlp_weight_gradient_gfrag": (_I, [_P, _P, _P, _I, _L, _I, _P, _P, _P, _L, _P]),
    "inerf_mlp_weight_gradient_xfrag": (_I, [_P, _I, _P, _L, _I, _P, _P, _P, _L, _P]),
    "inerf_wgrad_frag_grid": (_I, [_L, _I]),
    "inerf_wgrad_frag_rows": (_I, [_L, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), _I]),
    "inerf_mlp_weight_gradient_frag_batch": (_I, [_I, C.POINTER(_P), _P, C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int), _P, _L, C.POINTER(_P),
                                                  C.POINTER(_P), _L, _P]),
    "inerf_mlp_weight_gradient_frag": (_I, [_P, _P, _P, _P, _L, _P, _P, _L, _P]),
    "inerf_mlp_save_slot_is_fragment": (_I, [_I, _I]),
    "inerf_param_floats": (_L, [C.POINTER(NetDesc)]),
    "inerf_mlp_backward_workspace_bytes": (_L, [C.POINTER(NetDesc), _L]),
    "inerf_mlp_backward": (_I, [C.POINTER(NetDesc), _P, _P, _P, _P, _P, _L, _U, _P, _P, _L, _P, _P]),
    "inerf_pack_map": (_L, [C.POINTER(NetDesc), _I, _P, _P, _L, _P, _P, _P, _P, _P, _L, C.POINTER(C.c_int32)]),
    "inerf_repack": (_I, [C.POINTER(_P), C.POINTER(_L), _I, _P, _P, _L, _P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "inerf_composite": (_I, [_P, _P, _P, _I, _P, _L, _I, _I, _I, _I, _U, C.POINTER(CompositeOut), _P]),
    "inerf_composite_backward": (_I, [_P, _P, _P, _I, _P, _L, _I, _I, _I, _I, _U, C.POINTER(CompositeOut), _P, _P]),
    "inerf_sample_fine": (_I, [_P, _P, _P, _L, _I, _I, _U, _P, _P, _P, _P]),
    "inerf_sample_pdf": (_I, [_P, _P, _P, _L, _I, _I, _U, _P, _P]),
    "inerf_workspace_bytes": (_L, [C.POINTER(NetDesc), _L, _I, _I, _U]),
    "inerf_render_workspace_bytes": (_L, [C.POINTER(RenderArgs)]),
    "inerf_render_rays": (_I, [C.POINTER(RenderArgs), _P]),
    "inerf_gen_rays": (_I, [_P, _I, _P, _I, _I, _I, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _U, _P, _P]),
    "inerf_frame_to_u8": (_I, [_P, _L, _P, _P]),
    "inerf_cluster_lookup": (_I, [_P, _P, _L, _P, _P, _P, _P, _P, _P, _I, _U, _P, _P, _P]),
    "inerf_linear": (_I, [C.POINTER(LinearArgs), _P]),
    "inerf_linear_wgrad_workspace_bytes": (_L, [_L, _I, _I]),
    "inerf_linear_wgrad": (_I, [_P, _L, _I, _P, _L, _I, _L, _P, _P, _I, _P, _L, _P]),
    "inerf_embed": (_I, [_P, _P, _L, _I, _I, C.c_float, _I, _P, _L, _P]),
    "inerf_intrinsic_combine": (_I, [_P, _L, _L, _P]),
    "inerf_intrinsic_combine_backward": (_I, [_P, _P, _L, _L, _P, _P]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and "INERF_LIB_OVERRIDE" not in os.environ and not _build.have_hipcc():
            raise RuntimeError(
                f"{LIB_PATH} is not built.  Run `python -c 'import __graft_entry__ as g; g.build()'` (or "
                "`python -m intrinsicnerf_amd._build`).  intrinsicnerf_amd has no CPU or eager fallback.")
        path = os.environ.get("INERF_LIB_OVERRIDE", LIB_PATH)               # kernel-tuning experiments load variant builds
        if path == LIB_PATH and _build._stale():
            # the library was built from other sources, headers or flags than the tree holds (content digest linked into
            # it, _build.source_digest): the hand-mirrored structs below may no longer match it.  Rebuild when the
            # toolchain is here, refuse to load otherwise - never run a stale library.
            if _build.have_hipcc():
                _build.build_library()
            else:
                raise RuntimeError(f"{LIB_PATH} was built from different sources (digest {_build.built_digest()[:12]}, tree "
                                   f"{_build.source_digest()[:12]}) and hipcc is not available to rebuild it")
        if path != LIB_PATH and _build.sources_present() and _build.built_digest(path) != _build.source_digest():
            # an override library is loaded as asked for (the ABI number below is its only guard) - but never silently
            import warnings
            warnings.warn(f"INERF_LIB_OVERRIDE={path}: built from other sources than this tree (digest "
                          f"{_build.built_digest(path)[:12] or 'none'}, tree {_build.source_digest()[:12]})")
        handle = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        got = handle.inerf_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{path} reports ABI {got}, this binding was written for {ABI_VERSION}: rebuild the "
                               "library (python -m intrinsicnerf_amd._build) or update _capi.py")
        _lib = handle
    return _lib


def check(rc, what):
    if rc == OK:
        return
    msg = _ERR.get(rc, f"error {rc}")
    if rc == E_HIP:
        msg += f" (hipError_t {lib().inerf_last_hip_error()})"
    raise RuntimeError(f"{what}: {msg}")


_forced_precision = None


class forced_precision:
    """Context manager: every ``default_precision()`` inside answers ``prec`` (used to re-render a whole frame with the
    exact fp32 kernel after the split-precision one reported an out-of-range activation)."""

    def __init__(self, prec):
        self.prec = prec

    def __enter__(self):
        global _forced_precision
        self.saved, _forced_precision = _forced_precision, self.prec

    def __exit__(self, *exc):
        global _forced_precision
        _forced_precision = self.saved


def default_precision():
    """MLP arithmetic used when a caller does not choose: $INERF_PRECISION = f16x3 (default) | f32.

    f16x3 = fp32 operands split into f16 hi/lo pairs, three f16 MFMA products per MAC, fp32 accumulation:
    same accuracy against fp64 as the all-fp32 kernel, ~2.9x its speed.  The front-ends
    re-run a batch in f32 automatically if an activation leaves f16's range (never seen on real networks)."""
    if _forced_precision is not None:
        return _forced_precision
    name = os.environ.get("INERF_PRECISION", "f16x3").lower()
    if name not in ("f32", "f16x3"):
        raise ValueError(f"INERF_PRECISION={name!r}: expected 'f32' or 'f16x3'")
    return PREC_F16X3 if name == "f16x3" else PREC_F32


def net_desc(variant, n_classes=0, l_xyz=10, l_dir=4, xyz_div=1.0, precision=None):
    prec = default_precision() if precision is None else int(precision)
    return NetDesc(int(variant), int(n_classes), int(l_xyz), int(l_dir), float(xyz_div), prec)
