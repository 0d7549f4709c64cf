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
SOURCES = ["pack.cpp", "api.cpp", "mlp.hip", "mlp_f16.hip", "mlp_f16_t128.hip", "mlp_bwd.hip", "mlp_wgrad.hip", "train_api.hip", "ray_ops.hip", "frame_ops.hip", "cluster.hip", "layered.hip"]
HEADERS = [os.path.join(CSRC, "layout.h"), os.path.join(CSRC, "mlp_common.h"), os.path.join(CSRC, "mlp_f16_dev.h"), os.path.join(CSRC, "mlp_f16_heads.h"), os.path.join(os.path.dirname(PKG_DIR), "include", "inerf.h")]
# -ffp-contract=off: the reference rounds o + d*z, albedo*shading + residual, near*(1-t) + far*t ... as
# separate multiplies and adds; fused multiply-adds would move sample positions by an ulp, which the
# 2^9 frequency encoding amplifies.  MFMA accumulation is unaffected (it is an explicit builtin).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-comment", "-Wno-unused-result"]


# mlp_bwd.hip: the chain kernel's epilogues read the MFMA results with VALU instructions; with the accumulators in ordinary
# VGPRs (one or two waves per SIMD, 512 unified registers) its accumulator<->VGPR copies drop from ~580 to ~200.
EXTRA_FLAGS = {"mlp_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
DIGEST_MARK = b"INERF_BUILD_DIGEST="          # followed by 64 hex digits inside the library (inerf_build_digest())


def source_digest():
    """sha256 over everything the library is a function of: every source and header (bytes, not mtimes - a pushed or
    checked-out tree reorders mtimes), the compiler flags and this recipe's source list."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((SOURCES, FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def built_digest(lib_path=None):
    """The source digest linked into ``lib_path`` (what ``inerf_build_digest()`` returns), read from the file's bytes so
    that a stale library is never loaded; '' if the file carries none."""
    try:
        with open(lib_path or LIB_PATH, "rb") as fh:
            data = fh.read()
    except OSError:
        return ""
    i = data.find(DIGEST_MARK)
    return data[i + len(DIGEST_MARK): i + len(DIGEST_MARK) + 64].decode("ascii", "replace") if i >= 0 else ""


def _source_paths():
    return [os.path.join(CSRC, s) for s in SOURCES] + HEADERS


def sources_present():
    """False in a deployment that ships the built library without ``csrc/`` and the headers: nothing to compare with."""
    return all(os.path.exists(p) for p in _source_paths())


def _stale(lib_path=None):
    """True when ``lib_path`` (default: the in-tree library) is missing or was built from other sources / flags than the
    tree holds now.  Content-based: kernel edits, flag edits in this file and reordered mtimes are all caught.  A binary
    deployment (no file of csrc/ in the tree; the public header include/inerf.h may be there) cannot judge: an existing library is then taken as it is (the
    binding still checks its ABI number).  A tree that holds SOME of them is broken, not a deployment: the digest cannot be
    computed and a library of unknown origin must not be loaded in its name."""
    lib_path = lib_path or LIB_PATH
    if not os.path.exists(lib_path):
        return True
    # (decided from csrc/ alone: a binary deployment naturally ships the public C-ABI header include/inerf.h next to the library)
    csrc_paths = [p for p in _source_paths() if os.path.dirname(p) == CSRC]
    if not any(os.path.exists(p) for p in csrc_paths):
        return False
    present = [os.path.exists(p) for p in _source_paths()]
    if not all(present):
        missing = [os.path.relpath(p, os.path.dirname(CSRC)) for p, ok in zip(_source_paths(), present) if not ok]
        raise RuntimeError(f"intrinsicnerf_amd: the source tree is incomplete ({', '.join(missing)} missing): cannot tell whether "
                           f"{os.path.basename(lib_path)} was built from it.  Restore the files, or remove csrc/ and the headers entirely "
                           "for a binary deployment")
    return built_digest(lib_path) != source_digest()


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
            import hashlib
            header_bytes = b"".join(open(h, "rb").read() for h in HEADERS)

            def compile_one(src):
                obj = os.path.join(OBJ_DIR, src + ".o")
                path = os.path.join(CSRC, src)
                flags = [f for f in FLAGS if f != "-shared"] + EXTRA_FLAGS.get(src, [])
                key = hashlib.sha256(repr(flags).encode() + open(path, "rb").read() + header_bytes).hexdigest()
                try:
                    cached = open(obj + ".key").read().strip()
                except OSError:
                    cached = ""
                if not force and os.path.exists(obj) and cached == key:
                    return obj, None
                cmd = [hipcc] + flags + ["-c", path, "-o", obj + f".{os.getpid()}.tmp"]
                if verbose:
                    print(" ".join(cmd))
                proc = subprocess.run(cmd, capture_output=True, text=True)
                if proc.returncode != 0:
                    return obj, proc.stdout + proc.stderr
                os.replace(obj + f".{os.getpid()}.tmp", obj)
                with open(obj + ".key", "w") as fh:
                    fh.write(key)
                return obj, None

            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
                results = list(pool.map(compile_one, SOURCES))
            errors = [e for _, e in results if e]
            if errors:
                raise RuntimeError("hipcc failed:\n" + "\n".join(errors))
            # the digest of what was just compiled is linked INTO the library (inerf_build_digest): _stale() and the binding
            # compare it with the tree, so neither mtimes nor a side file can make a stale library look current
            stamp_src = os.path.join(OBJ_DIR, f"build_digest.{os.getpid()}.cpp")
            with open(stamp_src, "w") as fh:
                fh.write('extern "C" const char* inerf_build_digest(void) { return "%s%s" + %d; }\n'
                         % (DIGEST_MARK.decode(), source_digest(), len(DIGEST_MARK)))
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-x", "c++", stamp_src, "-x", "none"] + [o for o, _ in results] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            proc = subprocess.run(cmd, capture_output=True, text=True)
            os.remove(stamp_src)
            if proc.returncode != 0:
                raise RuntimeError("hipcc (link) failed:\n" + proc.stdout + proc.stderr)
            os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
