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


_tables = {}


def tensor_table(desc):
    """[(name, (rows, cols))] in the C packer's canonical order; cols == 0 marks a bias.  (Cached per description: the training
    step asks for it several times per network and step.)"""
    key = (desc.variant, desc.n_classes, desc.l_xyz, desc.l_dir)
    if key not in _tables:
        _tables[key] = tuple(_tensor_table(desc))
    return list(_tables[key])


def _tensor_table(desc):
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
    """Device blob for ``module`` (an nn.Module with the reference's parameter names), cached.

    The cache key is every parameter's ``(_version, data_ptr())``: optimiser steps, ``load_state_dict`` and any other
    in-place update through the autograd-visible tensor bump ``_version`` and re-pack.  Writes through ``.data``
    (``p.data.copy_(ema)``, a common EMA / weight-swap idiom) do NOT bump it and keep the pointer - call
    ``packing.invalidate(module)`` after such a write, or the next render uses the previously packed weights.

    Modules whose parameters already live on ``device`` are re-packed there (``DevicePacker``: a gather, a per-group
    max and an f16 split, ~0.3 ms, bit-identical to the host packer; the exact-fp32 format: ``DevicePackerF32``, one gather)
    instead of through a device->host copy and the C packer (~25 ms); CPU-resident modules go through the host packer."""
    params = list(module.parameters())
    versions = tuple((p._version, p.data_ptr()) for p in params) + (str(device), desc.n_classes, desc.l_xyz, desc.l_dir, desc.precision)
    per_module = _cache.setdefault(module, {})
    ent = per_module.get(desc.precision)
    if ent is None or ent.versions != versions:
        ent = _Entry()
        ent.versions = versions
        dev = torch.device(device)
        on_device = dev.type == "cuda" and all(p.device == dev or (p.device.type == "cuda" and dev.index is None) for p in params)
        if desc.precision == _capi.PREC_F16X3 and on_device and params:
            with torch.no_grad():
                ent.blob = device_packer(desc, False, params[0].device)(dict(module.named_parameters()))
        elif desc.precision == _capi.PREC_F32 and on_device and params:
            with torch.no_grad():
                ent.blob = device_packer_f32(desc, params[0].device)(dict(module.named_parameters()))
        else:
            ent.blob = pack_state_dict(desc, module.state_dict()).to(device)
        per_module[desc.precision] = ent
    return ent.blob


def invalidate(module=None):
    """Forget the packed blobs of ``module`` (of every module when None): the next render packs again.  Needed only
    after parameter writes that bypass autograd's version counter (``p.data.copy_(...)``, ``p.data = ...``)."""
    if module is None:
        _cache.clear()
    else:
        _cache.pop(module, None)


# ---------------------------------------------------------------------------------------------------------------------
# device-side packing (training): after every optimiser step the blobs have to be rebuilt; doing that through the host
# packer would cost a device->host copy of the weights and ~25 ms per blob.  The C library tells, once per network
# description, where every packed element comes from (inerf_pack_map); re-packing is then a gather, a per-group max
# (the per-GEMM power-of-two scales) and an f16 hi/lo split on whatever device the parameters live on.  Bit-identical
# to inerf_pack_weights / inerf_pack_weights_bwd (tests/test_capi_cpu.py).
# ---------------------------------------------------------------------------------------------------------------------
def pack_state_dict_bwd(desc, state_dict):
    """Host packer of the transposed (input-gradient) blob, ``inerf_pack_weights_bwd``."""
    L = _capi.lib()
    arrays = []
    for name, (rows, cols) in tensor_table(desc):
        arrays.append(state_dict[name].detach().to(device="cpu", dtype=torch.float32).contiguous().numpy())
    ptrs = (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
    n_floats = L.inerf_bwd_packed_floats(desc)
    blob = np.empty(n_floats, dtype=np.float32)
    _capi.check(L.inerf_pack_weights_bwd(desc, ptrs, len(arrays), blob.ctypes.data, n_floats), "inerf_pack_weights_bwd")
    return torch.from_numpy(blob)


class DevicePacker:
    """Re-packs a module's parameters into the forward (``backward=False``) or transposed blob with torch ops only."""

    def __init__(self, desc, backward, device):
        L = _capi.lib()
        d = _capi.NetDesc(desc.variant, desc.n_classes, desc.l_xyz, desc.l_dir, desc.xyz_div, _capi.PREC_F16X3)
        total = L.inerf_bwd_packed_floats(d) if backward else L.inerf_packed_floats(d)
        half_src = np.zeros(2 * total, np.int32)
        half_grp = np.zeros(2 * total, np.int32)
        cap = total
        c_dst, c_src, c_grp, c_code = (np.zeros(cap, np.int32) for _ in range(4))
        c_mult = np.zeros(cap, np.float32)
        n_groups = C.c_int32()
        n = L.inerf_pack_map(d, 1 if backward else 0, half_src.ctypes.data, half_grp.ctypes.data, 2 * total,
                             c_dst.ctypes.data, c_src.ctypes.data, c_grp.ctypes.data, c_code.ctypes.data, c_mult.ctypes.data, cap,
                             C.byref(n_groups))
        if n < 0:
            _capi.check(int(n), "inerf_pack_map")
        t = lambda a, dt=torch.int64: torch.from_numpy(a.astype(np.int64 if dt == torch.int64 else a.dtype)).to(device)
        self.names = [name for name, _ in tensor_table(d)]
        self.total = total
        self.n_groups = max(int(n_groups.value), 1)
        self.half_src = t(half_src)
        self.is_lo = t((half_grp < 0).astype(np.uint8), torch.uint8).bool()
        self.half_grp = t(np.where(half_grp < 0, -half_grp - 1, half_grp))
        # sources of every scale group as one zero-padded [groups, longest] index matrix: the per-group max is then a
        # gather + a row reduction (a scatter-max over 20 addresses serialises on atomics: 46 ms instead of 0.1)
        grp_abs = np.where(half_grp < 0, -half_grp - 1, half_grp)
        hi_mask = (half_grp >= 0) & (half_src > 0)
        per_group = [np.unique(half_src[hi_mask & (grp_abs == g)]) for g in range(self.n_groups)]
        longest = max(1, max(len(u) for u in per_group))
        gsrc = np.zeros((self.n_groups, longest), np.int32)
        for g, u in enumerate(per_group):
            gsrc[g, :len(u)] = u
        self.group_src = t(gsrc)
        self.c_dst, self.c_src, self.c_grp = t(c_dst[:n]), t(c_src[:n]), t(c_grp[:n])
        self.c_code = t(c_code[:n])
        self.c_mult = torch.from_numpy(c_mult[:n].copy()).to(device)
        # the same map as int32 device arrays for the library's own re-packing kernels (inerf_repack; HIP devices only)
        self.hip = None
        if torch.device(device).type == "cuda":
            i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)
            self.hip = dict(half_src=i32(half_src), half_grp=i32(half_grp), group_src=i32(gsrc), longest=int(longest),
                            c_dst=i32(c_dst[:n]), c_src=i32(c_src[:n]), c_grp=i32(c_grp[:n]), c_code=i32(c_code[:n]), n_consts=int(n))

    def __call__(self, named_params):
        """``named_params``: dict name -> tensor (parameters or a state dict) on this packer's device."""
        if self.hip is not None:
            return self._repack_hip(named_params)
        return self._repack_torch(named_params)        # (CPU tensors: the tests of the map itself)

    def _repack_hip(self, named_params):
        """One memset + three kernels of the library (inerf_repack) that read the parameters where they live."""
        ts = [named_params[k].detach() for k in self.names]
        ts = [t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous() for t in ts]
        h = self.hip
        blob = torch.empty(self.total, dtype=torch.float32, device=ts[0].device)
        gmax = torch.empty(self.n_groups, dtype=torch.float32, device=ts[0].device)        # zeroed by the library on the stream
        ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        counts = (C.c_int64 * len(ts))(*[t.numel() for t in ts])
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(blob.device):
            rc = _capi.lib().inerf_repack(ptrs, counts, len(ts), p(h["half_src"]), p(h["half_grp"]), self.total, p(h["group_src"]),
                                          self.n_groups, h["longest"], p(h["c_dst"]), p(h["c_src"]), p(h["c_grp"]), p(h["c_code"]),
                                          p(self.c_mult), h["n_consts"], p(gmax), p(blob),
                                          C.c_void_p(torch.cuda.current_stream(blob.device).cuda_stream))
        _capi.check(rc, "inerf_repack")
        return blob

    def _repack_torch(self, named_params):
        """The same with framework operations (any device; the CPU tests pin it to the host packer bit for bit)."""
        flat = torch.cat([named_params[k].detach().reshape(-1).float() for k in self.names])
        flat1 = torch.cat([flat.new_zeros(1), flat])
        v = flat1[self.half_src]
        gmax = flat1[self.group_src].abs().amax(dim=1)
        _, e = torch.frexp(gmax)                                   # gmax = f * 2^e, f in [0.5, 1)
        ok = (gmax > 0) & torch.isfinite(gmax)
        scale = torch.where(ok, torch.ldexp(torch.ones_like(gmax), 14 - e), torch.ones_like(gmax))
        vs = v * scale[self.half_grp]
        hi = vs.half()
        lo = (vs - hi.float()).half()
        blob = torch.where(self.is_lo, lo, hi).view(torch.float32)
        inv = 1.0 / scale[self.c_grp]
        consts = torch.where(self.c_code == 0, flat1[self.c_src] * self.c_mult, torch.where(self.c_code == 1, inv, inv * 0.125))
        blob[self.c_dst] = consts
        return blob


_device_packers = {}


def device_packer(desc, backward, device):
    key = (desc.variant, desc.n_classes, desc.l_xyz, desc.l_dir, bool(backward), str(device))
    if key not in _device_packers:
        _device_packers[key] = DevicePacker(desc, backward, device)
    return _device_packers[key]


class DevicePackerF32:
    """The exact-fp32 blob (INERF_PREC_F32) re-packed on the device: that format is a pure permutation of the parameters
    into MFMA-fragment order plus zero padding and a few constant 1.0 entries (the per-GEMM scale slots the split format
    fills), so the map is what the host packer makes of index-valued tensors (element j of the flattened parameters carries
    the value j + 2 < 2^24, exact in fp32; 0 and 1 stay themselves) and re-packing is one gather.
    ``verify`` (tests/test_capi_cpu.py) pins "pure permutation" against the host packer on real weights."""

    def __init__(self, desc, device):
        d = _capi.NetDesc(desc.variant, desc.n_classes, desc.l_xyz, desc.l_dir, desc.xyz_div, _capi.PREC_F32)
        table = tensor_table(d)
        self.names = [name for name, _ in table]
        index_sd, flat = {}, 0
        for name, (rows, cols) in table:
            count = rows * (cols if cols else 1)
            if flat + count >= 1 << 24:
                raise ValueError("network too large for the index-valued packing map")
            index_sd[name] = torch.arange(flat + 2, flat + count + 2, dtype=torch.float32).reshape((rows, cols) if cols else (rows,))
            flat += count
        src = pack_state_dict(d, index_sd)
        if not torch.equal(src, src.round()) or float(src.min()) < 0 or float(src.max()) > flat + 1:
            raise RuntimeError("the fp32 blob is not a permutation of the parameters: DevicePackerF32 does not apply")
        self.src = src.to(torch.int64).to(device)
        # the two constants the index map refers to besides the parameters, resident on the device: built per call they were a
        # pageable host-to-device copy - a host synchronisation per INERF_PRECISION=f32 training forward, and illegal under stream capture
        self.prefix = torch.tensor([0.0, 1.0], dtype=torch.float32, device=device)

    def __call__(self, named_params):
        flat = torch.cat([self.prefix] + [named_params[k].detach().reshape(-1).float() for k in self.names])
        return flat[self.src]


def device_packer_f32(desc, device):
    key = (desc.variant, desc.n_classes, desc.l_xyz, desc.l_dir, "f32", str(device))
    if key not in _device_packers:
        _device_packers[key] = DevicePackerF32(desc, device)
    return _device_packers[key]
