#!/usr/bin/env python3
"""Development aid: cycle stamps of workgroup 0 of the input-gradient chain at its phase boundaries (second tile).  Needs the
stamped build:  scripts/build_variant.sh stamps mlp_bwd.hip "-DINERF_DGRAD_STAMPS=1 -mllvm -amdgpu-mfma-vgpr-form=1"
               INERF_LIB_OVERRIDE=$PWD/intrinsicnerf_amd/libinerf_stamps.so python scripts/dgrad_timeline.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from intrinsicnerf_amd import _capi, kernels, packing  # noqa: E402

dev = torch.device("cuda:0")
desc = _capi.net_desc(_capi.VARIANT_OBJECT, 0, 10, 4, 1.0, _capi.PREC_F16X3)
sd = {k: v.to(dev) for k, v in oracle.make_state_dict("object", 0, seed=0).items()}
pf, pb = packing.device_packer(desc, False, dev)(sd), packing.device_packer(desc, True, dev)(sd)
n, s = 2048, 192
g = torch.Generator().manual_seed(0)
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
z = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, -1)[0].to(dev)
d_raw = torch.randn(n * s, 11, device=dev)
raw, save = kernels.encode_mlp_train(desc, pf, rays, z)
dual = os.environ.get("INERF_DGRAD_KERNEL", "dual")[0] != "s"
if dual:       # k_mlp_dgrad_dual: every phase ends at a barrier (odd stamps: work, even: waiting in the barrier)
    names = ["heads (VALU)", "dZ_vh (VALU)", "views^T GEMM", "d feature store", "feat^T GEMM", "dZ_as1 (VALU, k-block loads)", "as1^T GEMM", "d h7 store"]
    for l in range(7, 0, -1):
        names += [f"trunk layer {l}: GEMM", f"trunk layer {l}: store"]
else:
    names = ["heads (VALU)", "dZ_vh (VALU)", "views^T GEMM + store", "dZ_as1 (VALU)", "feat^T + as1^T GEMMs", "d h7 store"] + \
            [f"trunk layer {l}: GEMM + store" for l in range(7, 0, -1)]
for _ in range(2):
    stamps = torch.zeros(2 * 66, dtype=torch.float32, device=dev)
    kernels.mlp_backward_inputs(desc, pb, raw.view(n * s, 11), d_raw, save, dz_max=stamps, want_heads=True)
    torch.cuda.synchronize()
t = stamps.view(torch.int64).cpu().tolist()
k = t[1]
v = t[2:2 + k]
print(f"{k} stamps; {v[-1] - v[0]} cycles for the tile ({'two workgroups per CU' if dual else 'eight waves, one tile per CU'})")
work = wait = 0
for i in range(1, k):
    what = "  barrier wait" if i % 2 == 0 else names[(i - 1) // 2] if (i - 1) // 2 < len(names) else "?"
    print(f"  {i:2d} {what:32s} {v[i] - v[i - 1]:8d}")
    if i % 2 == 0: wait += v[i] - v[i - 1]
    else: work += v[i] - v[i - 1]
print(f"  work {work}, waiting in barriers {wait}")
