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
OBJ_DIR = os.path.join(CSRC, "_obj")          # git-ignored and gpurun-ignored: only the linked library travels
SOURCES = ["pack.cpp", "api.cpp", "mlp.hip", "mlp_f16.hip", "mlp_f16_pipe.hip", "mlp_bwd.hip", "mlp_wgrad.hip", "ray_ops.hip", "frame_ops.hip", "cluster.hip"]
HEADERS = [os.path.join(CSRC, "layout.h"), os.path.join(CSRC, "mlp_common.h"), os.path.join(CSRC, "mlp_f16_dev.h"), os.path.join(CSRC, "mlp_f16_heads.h"), os.path.join(os.path.dirname(PKG_DIR), "include", "inerf.h")]
# -ffp-contract=off: the reference rounds o + d*z, albedo*shading + residual, near*(1-t) + far*t ... as
# separate multiplies and adds; fused multiply-adds would move sample positions by an ulp, which the
# 2^9 frequency encoding amplifies.  MFMA accumulation is unaffected (it is an explicit builtin).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-comment", "-Wno-unused-result"]


# mlp_f16_pipe.hip: MFMA results in ordinary VGPRs (its epilogue reads them with VALU instructions while the next MFMAs run);
# the 256 registers of resident weights then take the AGPR half of the unified file.  Without the option the allocator puts
# the accumulators there and copies 64-128 registers back per phase.
# mlp_bwd.hip: the chain kernel also runs one wave per SIMD; the option cuts its accumulator<->VGPR copies from ~580 to ~200.
EXTRA_FLAGS = {"mlp_f16_pipe.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "mlp_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def have_hipcc():
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


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
            # one object per source, compiled in parallel and only when that source (or a header) changed: editing one
            # kernel file costs one compile + the link instead of the whole 40 s
            os.makedirs(OBJ_DIR, exist_ok=True)
            newest_header = max(os.path.getmtime(h) for h in HEADERS)

            def compile_one(src):
                obj = os.path.join(OBJ_DIR, src + ".o")
                path = os.path.join(CSRC, src)
                if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), newest_header):
                    return obj, None
                cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + EXTRA_FLAGS.get(src, []) + ["-c", path, "-o", obj + f".{os.getpid()}.tmp"]
                if verbose:
                    print(" ".join(cmd))
                proc = subprocess.run(cmd, capture_output=True, text=True)
                if proc.returncode != 0:
                    return obj, proc.stdout + proc.stderr
                os.replace(obj + f".{os.getpid()}.tmp", obj)
                return obj, None

            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
                results = list(pool.map(compile_one, SOURCES))
            errors = [e for _, e in results if e]
            if errors:
                raise RuntimeError("hipcc failed:\n" + "\n".join(errors))
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _ in results] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError("hipcc (link) failed:\n" + proc.stdout + proc.stderr)
            os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
