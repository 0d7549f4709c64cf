#!/usr/bin/env python3
"""Audit of libinerf.so's machine code for the 16-byte store hazard of gfx950 (scripts/microbench/store_war_hazard2.hip: a
buffer/global_store_dwordx3/x4 needs TWO wait states before a VALU instruction may overwrite its data registers; the compiler
inserts one): lists every such store whose data registers are written by a VALU instruction less than two wait states later.
usage: python scripts/store_hazard_audit.py [path to a gfx950 code object or to libinerf.so]   (exit code 1 if any site is found)"""
import os, re, shutil, struct, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([shutil.which("objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data = open(fat, "rb").read()
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
                path = os.path.join(tmp, f"co_{len(out)}.elf")
                open(path, "wb").write(data[i + off:i + off + size])
                out.append(path)
        pos = i + 24
    return out


def regs(tok):
    """Register numbers named by an operand like v12, v[12:15], a3 or a[0:3] (accumulation registers: + 1000)."""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        base = 1000 if m.group(1) == "a" else 0
        return set(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(1000 if m.group(1) == "a" else 0) + int(m.group(2))} if m else set()


unparsed = []          # wide stores whose data operand the audit could not read (must stay empty: tests/test_isa_audit_cpu.py)


def audit(path, need=2):
    asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", path], capture_output=True, text=True, check=True).stdout.split("\n")
    sites, func = [], "?"
    ins = []
    for l in asm:
        m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
        if m:
            ins.append(("func", m.group(1)))
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", l)
        if m:
            ins.append((m.group(1), m.group(2)))
    for k, (op, args) in enumerate(ins):
        if op == "func":
            func = args
            continue
        if not re.fullmatch(r"(buffer|global|flat|scratch)_store_dwordx[34]", op):
            continue
        toks = [t.strip() for t in args.split(",")]
        data = regs(toks[0]) if op.startswith("buffer") else regs(toks[1])
        if not data:
            unparsed.append((func, op + " " + args))
            continue
        waited, j = 0, k + 1
        while waited < need and j < len(ins):
            o2, a2 = ins[j]
            if o2 == "func":
                break
            if o2 == "s_nop":
                waited += int(a2.split()[0], 0) + 1
                j += 1
                continue
            # overwriters: every VALU instruction with a vector destination - v_accvgpr_write (destination a..) and the MFMAs (destination
            # v[..] or a[..]) included; the first operand is the destination in all of them
            if o2.startswith("v_") and not o2.startswith("v_cmp") and not o2.startswith("v_readlane") and not o2.startswith("v_readfirstlane"):
                dst = regs(a2.split(",")[0].strip())
                if dst & data:
                    sites.append((func, op + " " + args, o2 + " " + a2, waited))
                    break
            if o2.startswith("s_cbranch") or o2.startswith("s_branch") or o2 == "s_endpgm":
                break                      # (a taken branch costs more than a wait state; the fall-through is checked by the next instructions anyway)
            waited += 1
            j += 1
    return sites


if __name__ == "__main__":
    target = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "intrinsicnerf_amd", "libinerf.so")
    with tempfile.TemporaryDirectory() as tmp:
        cos = [target] if open(target, "rb").read(4) == b"\x7fELF" and b".hip_fatbin" not in open(target, "rb").read() else code_objects(target, tmp)
        total = 0
        for co in cos:
            for func, st, wr, waited in audit(co):
                total += 1
                print(f"{func[:70]}: {st}  ->  {wr}   ({waited} wait state(s) between)")
    print(f"{total} store(s) whose data registers are overwritten too early")
    sys.exit(1 if total else 0)
