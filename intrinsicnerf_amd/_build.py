"""Build recipe for ``libinerf.so`` (hand-written HIP for gfx950 + the C ABI of include/inerf.h).

Plain ``hipcc`` - no cmake, no torch extension machinery: the library has no torch types in its
interface.  ``hipcc`` cross-compiles for gfx950 without a GPU, so this runs in the build container;
the resulting ``.so`` is git-ignored but travels to the GPU box with the working tree.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libinerf.so")
SOURCES = ["pack.cpp", "api.cpp", "mlp.hip", "mlp_f16.hip", "mlp_bwd.hip", "mlp_wgrad.hip", "ray_ops.hip", "cluster.hip"]
HEADERS = [os.path.join(CSRC, "layout.h"), os.path.join(CSRC, "mlp_common.h"), os.path.join(CSRC, "mlp_f16_dev.h"), os.path.join(os.path.dirname(PKG_DIR), "include", "inerf.h")]
# -ffp-contract=off: the reference rounds o + d*z, albedo*shading + residual, near*(1-t) + far*t ... as
# separate multiplies and adds; fused multiply-adds would move sample positions by an ulp, which the
# 2^9 frequency encoding amplifies.  MFMA accumulation is unaffected (it is an explicit builtin).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-comment", "-Wno-unused-result"]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile ``libinerf.so`` in-tree if it is missing or older than its sources.  Returns its path.

    Safe under ``torch.distributed.run``: the ranks of one node serialise on a lock file, the first one builds (into a
    private temporary name, then an atomic rename) and the others find the library up to date when they get the lock."""
    if not force and not _stale():
        return LIB_PATH
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB_PATH
            hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
            if not os.path.exists(hipcc):
                raise RuntimeError("hipcc not found: cannot build libinerf.so (ROCm toolchain required)")
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
            os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
