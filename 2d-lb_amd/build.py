#!/usr/bin/env python3
"""Build liblbhip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python 2d-lb_amd/build.py            # -> 2d-lb_amd/LB_D2Q9/liblbhip.so

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "lb_hip.cpp")            # one translation unit; includes the csrc/*.h kernels
HDR = os.path.join(os.path.dirname(HERE), "include", "lb_hip.h")
OUT = os.path.join(HERE, "LB_D2Q9", "liblbhip.so")


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def sources():
    csrc = os.path.join(HERE, "csrc")
    return [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".cpp", ".h"))] + [HDR, __file__]


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(p) for p in sources())


def build_diag():
    """Diagnostic build with the ablation switches compiled in (tools/ablate.py): liblbhip_diag.so.
    Never used by the product; select it with LB_LIB=<path>."""
    out = OUT.replace("liblbhip.so", "liblbhip_diag.so")
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-DLB_DIAG",
                           "-fPIC", "-shared", SRC, "-o", out, "-ldl"])
    return out


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    # -ffp-contract=on: a*b+c fuses to an FMA only inside one source expression, so every kernel
    # instantiation (single step, two-step, slab edge rows) rounds identically: results are bitwise
    # independent of the kernel variant and of the slab partition.
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", SRC, "-o", OUT, "-ldl"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--diag" in sys.argv:
        print(build_diag())
    else:
        print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
