#!/usr/bin/env python3
"""Development aid: cycle stamps of workgroup 0 of one 256 x 256 weight-gradient product at the phase boundaries of its third
tile.  Needs the stamped build:  scripts/build_variant.sh wgstamps mlp_wgrad.hip "-DINERF_WGRAD_STAMPS=1"
               INERF_LIB_OVERRIDE=$PWD/intrinsicnerf_amd/libinerf_wgstamps.so python scripts/wgrad_timeline.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicnerf_amd import _capi  # noqa: E402

dev = torch.device("cuda:0")
lib = _capi.lib()
p = 2048 * 192
# G and X as two slots of ONE buffer, like the library's [slot][point][width] activation / gradient buffers: X starts
# p * 256 floats (+ an optional skew, bytes, multiple of 16: argv[1]) behind G
skew = int(sys.argv[1]) // 4 if len(sys.argv) > 1 else 0
both = torch.empty(2 * p * 256 + skew + 1024, device=dev)
g = both[:p * 256].view(p, 256).normal_()
x = both[p * 256 + skew:2 * p * 256 + skew].view(p, 256).normal_().abs_()
print(f"G at {g.data_ptr():#x}, X at {x.data_ptr():#x}: X - G = {x.data_ptr() - g.data_ptr():#x}")
ranges = torch.tensor([float(g.abs().max()), float(x.max())], device=dev)
grid = lib.inerf_wgrad_grid(p)
stride = 256 * 256 + 256
partial = torch.zeros(grid, stride, device=dev)
names = ["X share: split, transpose, LDS write", "G first half: split, transpose", "barrier wait", "contraction, first half",
         "G second half: wait, split, transpose", "contraction, second half"]
for _ in range(3):
    rc = lib.inerf_mlp_weight_gradient(C.c_void_p(g.data_ptr()), 256, C.c_void_p(x.data_ptr()), 256, p, 256, 256, C.c_void_p(ranges.data_ptr()),
                                       C.c_void_p(partial.data_ptr()), C.c_void_p(partial[:, 256 * 256:].data_ptr()), stride,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _capi.check(rc, "inerf_mlp_weight_gradient")
    torch.cuda.synchronize()
t = partial[0, 256 * 256:256 * 256 + 128].contiguous().view(torch.int64).cpu().tolist()
if all(t[8 * w] == 7 for w in range(8)):
    base = t[1]
    print("wave   top   X share  G half 1  at barrier  released  contraction 1  G half 2  contraction 2 = end of tile   (cycles; waves w and w + 4 share a SIMD)")
    for w in range(8):
        v = [x - base for x in t[8 * w + 1:8 * w + 8]]
        print(f"  {w}  {v[0]:6d}  {v[1] - v[0]:7d}  {v[2] - v[1]:8d}  {v[2]:10d}  {v[3]:8d}  {v[4] - v[3]:13d}  {v[5] - v[4]:8d}  {v[6] - v[5]:13d} = {v[6]:6d}")
else:
    print("(not a stamped build: timing only)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.inerf_mlp_weight_gradient(C.c_void_p(g.data_ptr()), 256, C.c_void_p(x.data_ptr()), 256, p, 256, 256, C.c_void_p(ranges.data_ptr()),
                                  C.c_void_p(partial.data_ptr()), C.c_void_p(partial[:, 256 * 256:].data_ptr()), stride,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
e1.record(); e1.synchronize()
print(f"one 256 x 256 product over {p} points: {e0.elapsed_time(e1) / 10 * 1000:.1f} us")
