"""State dict -> MFMA-fragment-ordered weight blob (host side of ``inerf_pack_weights``).

Consumes the reference's checkpoint key names unchanged (``network_fn_state_dict`` /
``network_fine_state_dict`` of run_nerf.py:1037-1042, ``network_coarse_state_dict`` /
``network_fine_state_dict`` of trainer.py:1042-1047).  A packed blob is cached per module and
rebuilt when any parameter's version counter moves (i.e. after an optimiser step or a
``load_state_dict``).
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _capi


def tensor_table(desc):
    """[(name, (rows, cols))] in the C packer's canonical order; cols == 0 marks a bias."""
    L = _capi.lib()
    n = L.inerf_num_tensors(desc)
    if n < 0:
        raise ValueError("unsupported network description")
    out = []
    name, rows, cols = C.c_char_p(), C.c_int64(), C.c_int64()
    for i in range(n):
        _capi.check(L.inerf_tensor_info(desc, i, C.byref(name), C.byref(rows), C.byref(cols)), "inerf_tensor_info")
        out.append((name.value.decode(), (rows.value, cols.value)))
    return out


def pack_state_dict(desc, state_dict):
    """Pack ``state_dict`` (any device / dtype convertible to fp32) into a CPU float32 blob."""
    L = _capi.lib()
    table = tensor_table(desc)
    arrays = []
    for name, (rows, cols) in table:
        if name not in state_dict:
            raise KeyError(f"state dict has no '{name}' (variant {desc.variant}, C={desc.n_classes})")
        t = state_dict[name].detach().to(device="cpu", dtype=torch.float32).contiguous()
        want = (rows, cols) if cols else (rows,)
        if tuple(t.shape) != want:
            raise ValueError(f"'{name}' has shape {tuple(t.shape)}, expected {want}")
        arrays.append(t.numpy())
    ptrs = (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
    n_floats = L.inerf_packed_floats(desc)
    blob = np.empty(n_floats, dtype=np.float32)
    _capi.check(L.inerf_pack_weights(desc, ptrs, len(arrays), blob.ctypes.data, n_floats), "inerf_pack_weights")
    return torch.from_numpy(blob)


class _Entry:
    __slots__ = ("versions", "blob")


_cache = weakref.WeakKeyDictionary()      # module -> {precision: _Entry}


def packed_for_module(module, desc, device):
    """Device blob for ``module`` (an nn.Module with the reference's parameter names), cached."""
    params = list(module.parameters())
    versions = tuple((p._version, p.data_ptr()) for p in params) + (str(device), desc.n_classes, desc.l_xyz, desc.l_dir, desc.precision)
    per_module = _cache.setdefault(module, {})
    ent = per_module.get(desc.precision)
    if ent is None or ent.versions != versions:
        ent = _Entry()
        ent.versions = versions
        ent.blob = pack_state_dict(desc, module.state_dict()).to(device)
        per_module[desc.precision] = ent
    return ent.blob
