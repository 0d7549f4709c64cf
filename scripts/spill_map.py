"""Where a kernel's register spills sit: compiles one source the way _build.py does, then lists every scratch store / load of
one kernel (matched by a substring of its mangled name) with the first instruction that uses a reloaded register.
usage: python scripts/spill_map.py mlp_f16.hip 'dualILb1ELb0ELb0E'"""
import os, re, subprocess, sys, tempfile
src, pat = sys.argv[1], sys.argv[2]
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "intrinsicnerf_amd", "csrc")
extra = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if src == "mlp_bwd.hip" else []
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-comment",
                    "-Wno-unused-result"] + extra + sys.argv[3:] + ["-c", os.path.join(csrc, src), "-o", os.path.join(d, "o.o"), "-save-temps=obj"], check=True)
    asm = [f for f in os.listdir(d) if f.endswith(".s") and "gfx950" in f][0]
    s = open(os.path.join(d, asm)).read()
names = [m for m in re.findall(r"^(_Z\w+):", s, re.M) if pat in m]
for name in names:
    i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
    body = s[i:j].split("\n")
    print(name, len(body), "lines")
    def off(l):
        m = re.search(r"offset:(\d+)", l); return m.group(1) if m else "0"
    nb = 0
    for k, l in enumerate(body):
        if "s_barrier" in l: nb += 1
        if "scratch_store" in l: print(f"  line {k:6d} barrier#{nb:3d} STORE off {off(l)}")
        if "scratch_load" in l:
            reg = re.search(r"scratch_load_dword\w* (v\[?\d+)", l).group(1).replace("[", "")
            use = next((body[m].strip() for m in range(k + 1, min(k + 120, len(body))) if re.search(r"\b" + reg + r"\b", body[m])), "?")
            print(f"  line {k:6d} barrier#{nb:3d} LOAD  off {off(l)} -> {use}")
