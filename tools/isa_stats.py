#!/usr/bin/env python3
"""Static instruction mix of one kernel instantiation, per loop (hipcc --save-temps assembly).

    tools/isa_stats.py "k_step4<LB_BC_PERIODIC, false, false, true, true>" [-D...] [--dump]

Compiles a translation unit holding only that instantiation (seconds instead of the library's minutes), finds the
innermost loops (backward branches) of every function in the assembly and prints, per loop: VALU instructions (packed,
v_mov, v_cndmask, dpp, other), SALU, LDS, VMEM, scalar memory.  The marching kernels execute their steady-state loop once
per wave and row, so the VALU count of that loop is what SQ_INSTS_VALU / (waves x rows) measures on the device.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "2d-lb_amd", "csrc")

TU = r"""
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "%(root)s/include/lb_hip.h"
#include "%(csrc)s/d2q9_cell.h"
#include "%(csrc)s/kernels_fused.h"
#include "%(csrc)s/kernels_step4.h"
#include "%(csrc)s/kernels_step5.h"
#include "%(csrc)s/kernels_deep.h"
#include "%(csrc)s/kernels_deep2.h"
#include "%(csrc)s/kernels_tile.h"
#include "%(csrc)s/kernels_phases.h"
void isa_stats_force(hipStream_t st) { void *p = (void *)(&%(kernel)s); hipLaunchKernel(p, dim3(1), dim3(1), nullptr, 0, st); }
"""


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer") or op.startswith("s_store"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    flags = []
    for a in sys.argv[1:]:
        if a.startswith("-D"):
            flags.append(a)
        elif a.startswith("-mllvm="):                    # -mllvm=<option>  ->  -mllvm <option>
            flags += ["-mllvm", a[len("-mllvm="):]]
    dump = "--dump" in sys.argv
    kernel = args[0]
    tmp = tempfile.mkdtemp(prefix="isa_")
    src = os.path.join(tmp, "tu.cpp")
    open(src, "w").write(TU % {"root": ROOT, "csrc": CSRC, "kernel": kernel})
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-c", src, "-o",
           os.path.join(tmp, "tu.o"), "--save-temps", "-Rpass-analysis=kernel-resource-usage"] + flags
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
    if r.returncode:
        sys.stderr.write(r.stderr)
        sys.exit(1)
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (.*?)\s*\[-Rpass", line)
        if m and re.search(r"VGPRs:|AGPRs|Scratch|Occupancy|LDS Size|SGPRs|Spill", m.group(1)):
            print("  ", m.group(1).strip())
    asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    lines = open(os.path.join(tmp, asm)).read().splitlines()
    # instructions with their block labels
    insts, labels = [], {}
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")):
            m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", s)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        op = s.split()[0]
        insts.append((op, s))
    # loops = backward branches
    loops = []
    for i, (op, s) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    print("%d instructions, %d loops" % (len(insts), len(loops)))
    for a, b, tgt in loops:
        body = insts[a:b + 1]
        c = {}
        for op, s in body:
            k = classify(op)
            c[k] = c.get(k, 0) + 1
        valu = [x for x in body if classify(x[0]) == "valu"]
        pk = sum(1 for op, s in valu if op.startswith("v_pk_"))
        mov = sum(1 for op, s in valu if op.startswith("v_mov") and "dpp" not in s and "row_" not in s and "wave_" not in s)
        dpp = sum(1 for op, s in valu if "dpp" in op or "row_" in s or "wave_" in s or "quad_perm" in s)
        cnd = sum(1 for op, s in valu if op.startswith("v_cndmask"))
        acc = sum(1 for op, s in valu if op.startswith("v_accvgpr"))
        print("loop %-12s %5d insts: VALU %4d (pk %d, mov %d, cndmask %d, dpp %d, accvgpr %d, other %d)  SALU %d  LDS %d  VMEM %d  SMEM %d  scratch %d" % (
            tgt, len(body), len(valu), pk, mov, cnd, dpp, acc, len(valu) - pk - mov - cnd - dpp - acc, c.get("salu", 0), c.get("lds", 0),
            c.get("vmem", 0), c.get("smem", 0), c.get("scratch", 0)))
        if dump:
            hist = {}
            for op, s in valu:
                hist[op] = hist.get(op, 0) + 1
            for op, n in sorted(hist.items(), key=lambda t: -t[1]):
                print("      %-28s %d" % (op, n))
    if dump:
        print("assembly:", os.path.join(tmp, asm))
        if loops:                                        # the largest loop, instruction by instruction, next to the assembly
            a, b, tgt = max(loops, key=lambda t: t[1] - t[0])
            out = os.path.join(tmp, "loop.s")
            open(out, "w").write("\n".join(s for op, s in insts[a:b + 1]) + "\n")
            hist = {}
            for op, s in insts[a:b + 1]:
                if classify(op) != "valu":
                    hist[op] = hist.get(op, 0) + 1
            print("largest loop %s -> %s; its non-VALU instructions:" % (tgt, out))
            for op, n in sorted(hist.items(), key=lambda t: -t[1]):
                print("      %-28s %d" % (op, n))


if __name__ == "__main__":
    main()
