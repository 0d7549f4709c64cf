#!/usr/bin/env python3
"""Where a kernel's SGPR spills (v_writelane / v_readlane pairs) sit relative to its loops: compiles one source the way _build.py does
and lists every loop of one kernel (matched by a substring of its mangled name) with its MFMA / readlane / writelane counts.
usage: python scripts/sgpr_spill_loops.py mlp_f16.hip dualILb0ELb0ELb0E"""
import os, re, subprocess, sys, tempfile
src, pat = sys.argv[1], sys.argv[2]
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "intrinsicnerf_amd", "csrc")
extra = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if src == "mlp_bwd.hip" else []
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-comment",
                    "-Wno-unused-result"] + extra + ["-c", os.path.join(csrc, src), "-o", os.path.join(d, "o.o"), "-save-temps=obj"], check=True)
    s = open(os.path.join(d, [f for f in os.listdir(d) if f.endswith(".s") and "gfx950" in f][0])).read()
for name in [m for m in re.findall(r"^(_Z\w+):", s, re.M) if pat in m]:
    i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
    body = s[i:j].split("\n")
    labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            loops.append((labels[m.group(1)], k))
    cnt = lambda seg, w: sum(w in l for l in seg)
    print(f"{name}: {len(body)} lines, v_writelane {cnt(body, 'v_writelane_b32')}, v_readlane {cnt(body, 'v_readlane_b32')}")
    for a, b in sorted(loops, key=lambda t: t[1] - t[0]):
        seg = body[a:b + 1]
        if cnt(seg, "v_mfma") == 0:
            continue
        inner = not any(a < a2 and b2 < b for a2, b2 in loops)
        print(f"  loop of {b - a:5d} lines ({'innermost' if inner else 'outer'}): {cnt(seg, 'v_mfma'):4d} MFMAs, {cnt(seg, 'v_readlane_b32'):4d} v_readlane, "
              f"{cnt(seg, 'v_writelane_b32'):4d} v_writelane")
