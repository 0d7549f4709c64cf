"""Audit of the machine code inside libinerf.so (no GPU needed: the gfx950 code objects are pulled out of the library's
.hip_fatbin section and disassembled with ROCm's llvm-objdump / llvm-readelf).

* A 16-byte (or 12-byte) buffer store whose SGPR offset operand is a REGISTER gets no wait state from the compiler before its data
  registers may be overwritten, and on gfx950 the store then sometimes sends what the next instruction wrote (found in round 2 in
  the input-gradient chain).  With the constant 0 in that operand the compiler inserts the s_nop.  No such
  store may appear anywhere in the library.
* The default inference kernel must stay (nearly) free of scratch: its 2 workgroups per CU sit at the 256-register limit, and a
  change that makes it spill is a performance regression before it is anything else.  The same for the training kernels
  (the training forward had 29 spilled VGPRs / 120 bytes, the SSR chain 31 / 112): bounds on their scratch, and
  none at all in the kernels that stream (weight gradients) or whose every tile phase is on the critical path (object chain)."""
import os
import re
import shutil
import struct
import subprocess

import pytest

from intrinsicnerf_amd import _build, _capi

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(tmp_path):
    objcopy = shutil.which("objcopy")
    if objcopy is None or not os.path.exists(f"{LLVM}/llvm-objdump"):
        pytest.skip("objcopy / llvm-objdump not available")
    _capi.lib()                                            # builds the library if it is missing or stale
    fat = tmp_path / "fat.bin"
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", _build.LIB_PATH, str(fat)], check=True)
    data = fat.read_bytes()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        (num,) = struct.unpack_from("<Q", data, i + 24)
        p = i + 32
        for _ in range(num):
            off, size, idl = struct.unpack_from("<QQQ", data, p)
            p += 24
            ident = data[p:p + idl].decode()
            p += idl
            if "gfx950" in ident and size > 0:
                path = tmp_path / f"co_{len(out)}.elf"
                path.write_bytes(data[i + off:i + off + size])
                out.append(str(path))
        pos = i + 24
    assert len(out) >= len([s for s in _build.SOURCES if s.endswith(".hip")]), "one gfx950 code object per HIP source expected"
    return out


def test_no_wide_buffer_store_with_a_register_soffset(tmp_path):
    wide = bad = 0
    for co in _code_objects(tmp_path):
        asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
        for line in asm.split("\n"):
            if re.search(r"\bbuffer_store_dwordx[34]\b", line):
                wide += 1
                if re.search(r"\], s\d+\b", line):             # ... v[a:b], vN, s[rsrc], sN  <- SGPR offset in a register
                    bad += 1
    assert wide > 500, "the training forward and the chain store 16 bytes per lane in hundreds of places: audit found none?"
    assert bad == 0, f"{bad} of {wide} 16-byte buffer stores carry their SGPR offset in a register (store-data hazard, DESIGN.md 3.1b)"


def _scratch_notes(tmp_path):
    scratch = {}
    for co in _code_objects(tmp_path):
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        name = None
        for line in notes.split("\n"):
            m = re.search(r"\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name:
                scratch[name] = int(m.group(1))
    return scratch


def test_default_inference_kernels_do_not_live_in_scratch(tmp_path):
    scratch = _scratch_notes(tmp_path)
    dual = {k: v for k, v in scratch.items() if "k_encode_mlp_f16x3_dual" in k}
    assert len(dual) == 5, sorted(scratch)
    # <kSave = false>: object-level and SSR inference (mangled: ...dualILb0ELb0ELb0EE / ...dualILb0ELb1ELb0EE, and the SSR form with the
    # channel-split semantic head ...dualILb0ELb1ELb1EE)
    for k, v in dual.items():
        if "ILb0E" in k:
            assert v == 0, f"{k}: {v} bytes of scratch per lane"
    assert scratch["_ZN5inerf23k_encode_mlp_f16x3_dualILb0ELb0ELb0EEEvNS_9MlpParamsE"] == 0          # the 64-point form of the headline kernel: none
    # the headline kernels (128-point tile, round 6: object-level and SSR): none
    t128 = {k: v for k, v in scratch.items() if "k_encode_mlp_f16x3_t128" in k}
    # <kSsr, kSave, kPipe>: the two default inference forms hold no scratch; the opt-in saving form (INERF_TRAIN_FWD=t128) and the opt-in
    # pipelined trunk (INERF_F16_KERNEL=pp) may keep a few dwords (at most 128 bytes per lane)
    assert len(t128) == 4, t128
    for k, v in t128.items():
        assert v == 0 if k.endswith(("ILb0ELb0ELb0EEEvNS_9MlpParamsE", "ILb1ELb0ELb0EEEvNS_9MlpParamsE")) else v <= 128, (k, v)
    assert scratch.get("_ZN5inerf12k_encode_mlpILb0ELi2EEEvNS_9MlpParamsE", 0) == 0


def test_training_kernels_stay_out_of_scratch(tmp_path):
    scratch = _scratch_notes(tmp_path)
    bytes_of = lambda part: {k: v for k, v in scratch.items() if part in k}
    fwd = bytes_of("k_encode_mlp_f16x3_dualILb1E")                     # <kSave = true>: object-level, SSR
    assert len(fwd) == 2, sorted(scratch)
    assert fwd["_ZN5inerf23k_encode_mlp_f16x3_dualILb1ELb0ELb0EEEvNS_9MlpParamsE"] == 0            # object-level (round 3: 120 bytes, round 4: 44)
    assert fwd["_ZN5inerf23k_encode_mlp_f16x3_dualILb1ELb1ELb0EEEvNS_9MlpParamsE"] <= 8            # SSR: one lane-derived invariant, re-read once per tile (132 / 64)
    chain = bytes_of("k_mlp_dgrad")
    assert len(chain) == 4, sorted(chain)
    assert chain["_ZN5inerf11k_mlp_dgradILb0ELi8EEEvNS_9BwdParamsE"] == 0
    assert chain["_ZN5inerf11k_mlp_dgradILb1ELi8EEEvNS_9BwdParamsE"] <= 80          # (round 3: 112; the 1-4-row heads' accumulators of one VALU stage)
    # the two-workgroup chains (the default): none
    assert chain["_ZN5inerf16k_mlp_dgrad_dualILb0EEEvNS_9BwdParamsE"] == 0
    assert chain["_ZN5inerf16k_mlp_dgrad_dualILb1EEEvNS_9BwdParamsE"] == 0
    wgrad = bytes_of("k_mlp_wgrad")
    assert len(wgrad) >= 9 and all(v == 0 for v in wgrad.values()), wgrad


def test_no_wide_store_is_overwritten_within_two_wait_states(tmp_path):
    """gfx950 reads the data registers of a 16-byte (or 12-byte) store up to one issue slot later than the compiler assumes: with ONE
    wait state between a buffer/global_store_dwordx4 and a VALU write of its data registers 0.07 % of the stored dwords are the
    overwritten ones under vector-memory pressure, with two none (scripts/microbench/store_war_hazard2.hip, profiles/
    r05_store_war_hazard.txt).  The compiler inserts one `s_nop 0`; the library's stores come in pairs or are followed by other
    instructions, which makes it two everywhere today - this test keeps it that way (scripts/store_hazard_audit.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("store_hazard_audit", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "store_hazard_audit.py"))
    audit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(audit)
    sites, stores = [], 0
    for co in _code_objects(tmp_path):
        sites += audit.audit(co, need=2)
        stores += len(audit.audit(co, need=64))           # (every wide store whose registers are reused at all: the audit sees them)
    assert stores > 300, "the audit should see hundreds of wide stores whose data registers are reused"
    assert not audit.unparsed, f"{len(audit.unparsed)} wide stores whose data registers the audit could not read (AGPR forms?): {audit.unparsed[:3]}"
    assert not sites, f"{len(sites)} wide stores are overwritten less than two wait states later: {sites[:3]}"
