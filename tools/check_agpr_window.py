#!/usr/bin/env python3
"""k_deep keeps the row it gathers ahead in a FIXED window of accumulation registers, a[0:42], loaded by an asm block and taken
out behind a hand-written s_waitcnt (csrc/kernels_deep.h: deep_row_issue / deep_row_take).  The compiler is told that the asm
blocks clobber the window, not that it is reserved: this script disassembles the device code of the built objects and reports any
instruction that touches the window other than the asm blocks' loads (buffer_load_dwordx4 / global_load_dword INTO it) and the
compiler's copies out of it (v_accvgpr_read_b32 FROM it).  Exit status 1 on a stray.
Works on object files and on the linked library alike (every code object of the .hip_fatbin section is disassembled), so the check
also runs where only the .so exists (the GPU box).  build.py runs it on every product build and fails the build on a stray.
Usage: tools/check_agpr_window.py [object | library ...]      default: 2d-lb_amd/LB_D2Q9/liblbhip.so"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LO, HI = 0, 42


MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(obj):
    """the gfx950 code objects inside an object file or a shared library: its .hip_fatbin section is one offload bundle per
    translation unit (magic, number of entries, then offset / size / triple per entry), laid end to end"""
    with tempfile.TemporaryDirectory(prefix="agpr_") as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        blob = open(fat, "rb").read()
    out, at = [], blob.find(MAGIC)
    while at >= 0:
        n = struct.unpack_from("<Q", blob, at + len(MAGIC))[0]
        q = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(MAGIC, at + len(MAGIC))
    return out


def device_disassembly(obj):
    text = []
    for co in code_objects(obj):
        if b"k_deep" not in co:                          # (a library: only the code objects of the k_deep translation units)
            continue
        with tempfile.NamedTemporaryFile(prefix="agpr_", suffix=".co") as f:
            f.write(co)
            f.flush()
            text.append(subprocess.run([LLVM + "/llvm-objdump", "-d", f.name], check=True, capture_output=True, text=True).stdout)
    return "\n".join(text)


def window_regs(operand):
    """accumulation registers of the window an operand names: 'a200' or 'a[196:199]'"""
    m = re.fullmatch(r"a(\d+)", operand)
    if m:
        lo = hi = int(m.group(1))
    else:
        m = re.fullmatch(r"a\[(\d+):(\d+)\]", operand)
        if not m:
            return False
        lo, hi = int(m.group(1)), int(m.group(2))
    return hi >= LO and lo <= HI


def check(obj, text=None):
    loads = reads = 0
    strays = []
    func = "?"
    for line in (text if text is not None else device_disassembly(obj)).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            func = m.group(1)
            continue
        m = re.match(r"^\s+(\S+)\s+([^/]*)//", line)
        if not m:
            continue
        op, ops = m.group(1), [o.strip() for o in m.group(2).split(",")]
        hit = [k for k, o in enumerate(ops) if window_regs(o.split()[0] if o else "")]
        if not hit:
            continue
        if op in ("buffer_load_dwordx4", "global_load_dword") and hit == [0]:
            loads += 1
        elif op == "v_accvgpr_read_b32" and hit == [1]:
            reads += 1
        else:
            strays.append("%s: %s %s" % (func, op, ", ".join(ops)))
    return loads, reads, strays


def check_waits(obj, text=None):
    """The hand-written waits.  deep_row_take waits `s_waitcnt vmcnt(N)`, N = 9 (12 where the launch also stores rho, u, v): "all but
    my N youngest vector-memory operations are done" means "the row gathered ahead has arrived" only if the wave has issued EXACTLY N
    such operations since that row's last load -- the stores of one steady iteration.  Too few only waits longer; too many would read
    registers still in flight, silently.  The asm blocks are volatile and clobber memory, so the ORDER of a kernel's vector-memory
    instructions is the source's; what a change of the source (or of the compiler) could break is their NUMBER.  Per kernel that holds
    W such waits (one per steady iteration the compiler laid out: two marching directions x the loop's iterations + the lone one
    behind a loop of pairs) the device code must hold exactly 9 W non-temporal stores (the store block's one alternative), N W plain
    ones (the other alternative + the three of rho, u, v in either) and NO other vector-memory instruction than the window loads.
    Returns (waits checked, problems)."""
    text = text if text is not None else device_disassembly(obj)
    checked, problems = 0, []
    func, fs = None, {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            func = m.group(1)
            fs[func] = {"w9": 0, "w12": 0, "nt": 0, "plain": 0, "win": 0, "other": []}
            continue
        m = re.match(r"^\s+(\S+)\s*([^/]*)//", line)
        if not m or func is None:
            continue
        op, ops = m.group(1), m.group(2).strip()
        c = fs[func]
        if op == "s_waitcnt" and ops in ("vmcnt(9)", "vmcnt(12)"):
            c["w9" if ops == "vmcnt(9)" else "w12"] += 1
        elif classify_vmem(op) is not None:
            if op in ("buffer_load_dwordx4", "global_load_dword") and window_regs(ops.split(",")[0].strip()):
                c["win"] += 1
            elif op in ("buffer_store_dwordx4", "global_store_dwordx4"):
                c["nt" if ops.endswith(" nt") else "plain"] += 1
            else:
                c["other"].append("%s %s" % (op, ops))
    for name, c in fs.items():
        if not c["win"]:
            continue
        w = c["w9"] + c["w12"]
        checked += w
        if "k_deep2" in name:
            # a front wave of k_deep2 stores nothing: its wait is vmcnt(0), whatever else (the back waves' stores, spills) is the compiler's
            if w:
                problems.append("%s: %d waits vmcnt(9 / 12) in a kernel whose front waves store nothing" % (name, w))
            continue
        if (c["w9"] and c["w12"]) or not w:
            problems.append("%s: %d waits vmcnt(9), %d waits vmcnt(12) beside %d window loads" % (name, c["w9"], c["w12"], c["win"]))
            continue
        n = 12 if c["w12"] else 9
        if c["nt"] != 9 * w or c["plain"] != n * w or c["other"]:
            problems.append("%s: %d waits vmcnt(%d): %d non-temporal stores (want %d), %d plain (want %d), other vector-memory instructions: %s"
                            % (name, w, n, c["nt"], 9 * w, c["plain"], n * w, c["other"][:3] or "none"))
    return checked, problems


def classify_vmem(op):
    return op if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else None


def check_all(obj):
    """both checks on one disassembly: (loads, reads, strays, waits checked, wait problems)"""
    text = device_disassembly(obj)
    loads, reads, strays = check(obj, text)
    checked, problems = check_waits(obj, text)
    return loads, reads, strays, checked, problems


def main():
    objs = sys.argv[1:] or [os.path.join(ROOT, "2d-lb_amd", "LB_D2Q9", "liblbhip.so")]
    bad = 0
    for obj in objs:
        loads, reads, strays, checked, problems = check_all(obj)
        print("%s: %d loads into a[%d:%d], %d copies out of it, %d other instructions touch it; %d hand-written waits, %d miscounted" % (
            os.path.relpath(obj, ROOT), loads, LO, HI, reads, len(strays), checked, len(problems)))
        for s in strays[:20]:
            print("   STRAY", s)
        for s in problems[:20]:
            print("   WAIT", s)
        bad += len(strays) + len(problems)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
