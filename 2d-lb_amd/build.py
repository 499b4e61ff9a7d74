#!/usr/bin/env python3
"""Build liblbhip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python 2d-lb_amd/build.py            # -> 2d-lb_amd/LB_D2Q9/liblbhip.so
    python 2d-lb_amd/build.py --diag     # -> 2d-lb_amd/LB_D2Q9/liblbhip_diag.so (ablation switches, tools/ablate.py)

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.  One translation unit per kernel
family (csrc/launchers.h), compiled in parallel; objects under 2d-lb_amd/build/ (git- and gpurun-ignored) are reused while
neither their source nor any header is newer.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HDR = os.path.join(os.path.dirname(HERE), "include", "lb_hip.h")
OUT = os.path.join(HERE, "LB_D2Q9", "liblbhip.so")
OBJ = os.path.join(HERE, "build")
# (largest first: the pool starts them in this order)
UNITS = ["deep7.cpp", "deep6.cpp", "march5.cpp", "march4.cpp", "lb_hip.cpp", "march23.cpp", "tile.cpp", "step1.cpp"]
# -ffp-contract=on: a*b+c fuses to an FMA only inside one source expression, so every kernel instantiation (single step,
# multi-step, slab edge rows) -- in whichever translation unit -- rounds identically: results are bitwise independent of the
# kernel variant and of the slab partition.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def headers():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [HDR, __file__]


def sources():
    return [os.path.join(CSRC, u) for u in UNITS] + headers()


def up_to_date(out=OUT):
    return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(p) for p in sources())


def _compile(unit, tag, extra, verbose):
    src = os.path.join(CSRC, unit)
    obj = os.path.join(OBJ, "%s%s.o" % (os.path.splitext(unit)[0], tag))
    deps = [src] + headers()
    if os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(p) for p in deps):
        return obj
    cmd = [hipcc()] + FLAGS + extra + ["-c", src, "-o", obj]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def _build(out, tag, extra, force, verbose, jobs):
    if not force and up_to_date(out):
        return out
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            if f.endswith(tag + ".o"):
                os.remove(os.path.join(OBJ, f))
    jobs = jobs or int(os.environ.get("LB_BUILD_JOBS", "0")) or min(len(UNITS), os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(lambda u: _compile(u, tag, extra, verbose), UNITS))
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", out, "-ldl"])
    return out


def build_diag(force=False, verbose=False, jobs=0):
    """Diagnostic build with the ablation switches compiled in (tools/ablate.py): liblbhip_diag.so.
    Never used by the product; select it with LB_LIB=<path>."""
    return _build(OUT.replace("liblbhip.so", "liblbhip_diag.so"), "_diag", ["-DLB_DIAG"], force, verbose, jobs)


def build(force=False, verbose=False, jobs=0):
    return _build(OUT, "", [], force, verbose, jobs)


def build_variant(tag, flags, force=False, verbose=False, jobs=0):
    """An experimental build with extra compiler flags: liblbhip_<tag>.so (A/B timing through LB_LIB; never the product)."""
    return _build(OUT.replace("liblbhip.so", "liblbhip_%s.so" % tag), "_" + tag, list(flags), force, verbose, jobs)


if __name__ == "__main__":
    kw = dict(force="--force" in sys.argv, verbose="-v" in sys.argv)
    if "--variant" in sys.argv:          # --variant <tag> <flag> [<flag> ...]
        i = sys.argv.index("--variant")
        extra = []
        for f in sys.argv[i + 2:]:
            if f.startswith("-D"):
                extra.append(f)
            elif f.startswith("-mllvm="):                # -mllvm=<option>  ->  -mllvm <option>
                extra += ["-mllvm", f[len("-mllvm="):]]
        print(build_variant(sys.argv[i + 1], extra, **kw))
    else:
        print(build_diag(**kw) if "--diag" in sys.argv else build(**kw))
