#!/usr/bin/env python3
"""Development aid: cycle stamps of one workgroup of the pipelined MLP kernel at its phase boundaries (first tile)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["INERF_F16_KERNEL"] = "pipe"
import oracle  # noqa: E402
from intrinsicnerf_amd import _capi, packing  # noqa: E402

dev = torch.device("cuda:0")
desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, 1.0, _capi.PREC_F16X3)
packed = packing.pack_state_dict(desc, oracle.make_state_dict("object", 0, seed=0)).to(dev)
n, s = 65536, 192
g = torch.Generator().manual_seed(0)
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0].to(dev)
raw = torch.empty(n, s, 11, device=dev)
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
L = _capi.lib()
for _ in range(2):
    stamps.zero_()
    rc = L.inerf_debug_encode_mlp(desc, C.c_void_p(packed.data_ptr()), C.c_void_p(rays.data_ptr()), C.c_void_p(z.data_ptr()), n, s, 0,
                                  C.c_void_p(raw.data_ptr()), C.c_void_p(stamps.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _capi.check(rc, "inerf_debug_encode_mlp")
    torch.cuda.synchronize()
t = stamps.cpu().tolist()
k = t[0]
v = t[1:1 + k]
print(f"{k} stamps; total {v[-1] - v[0]} cycles for the first tile")
for i in range(1, k):
    kind = "barrier wait" if i % 2 == 1 else "work        "
    print(f"  {i:2d} {kind} {v[i] - v[i - 1]:8d}")
