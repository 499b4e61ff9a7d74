#!/usr/bin/env python3
"""k_deep keeps the row it gathers ahead in a FIXED window of accumulation registers, a[192:234], loaded by an asm block and taken
out behind a hand-written s_waitcnt (csrc/kernels_deep.h: deep_row_issue / deep_row_take).  The compiler is told that the asm
blocks clobber the window, not that it is reserved: this script disassembles the device code of the built objects and reports any
instruction that touches the window other than the asm blocks' loads (buffer_load_dwordx4 / global_load_dword INTO it) and the
compiler's copies out of it (v_accvgpr_read_b32 FROM it).  Exit status 1 on a stray.
Usage: tools/check_agpr_window.py [object ...]      default: 2d-lb_amd/build/deep6.o deep7.o"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LO, HI = 192, 234


def device_disassembly(obj):
    with tempfile.TemporaryDirectory(prefix="agpr_") as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        return subprocess.run([LLVM + "/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


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


def check(obj):
    loads = reads = 0
    strays = []
    func = "?"
    for line in device_disassembly(obj).splitlines():
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


def main():
    objs = sys.argv[1:] or [os.path.join(ROOT, "2d-lb_amd", "build", u) for u in ("deep6.o", "deep7.o")]
    bad = 0
    for obj in objs:
        loads, reads, strays = check(obj)
        print("%s: %d loads into a[%d:%d], %d copies out of it, %d other instructions touch it" % (
            os.path.relpath(obj, ROOT), loads, LO, HI, reads, len(strays)))
        for s in strays[:20]:
            print("   STRAY", s)
        bad += len(strays)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
