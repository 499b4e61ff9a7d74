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
UNITS = ["deep7.cpp", "deep2.cpp", "deep6.cpp", "march5.cpp", "march4.cpp", "lb_hip.cpp", "march23.cpp", "tile.cpp", "step1.cpp"]
# Units whose DEVICE code is compiled through LLVM IR so that a function attribute clang cannot spell can be added to their kernels
# (_compile_patched): k_deep2 runs two waves per SIMD and keeps a row in flight in accumulation registers a[0:42]; the compiler's default
# for such a kernel is 128 vector + 128 accumulation registers, "amdgpu-agpr-alloc"="44" makes it 212 + 44 (kernels_deep2.h).
IR_ATTRS = {}      # (round 6 built k_deep2 that way until its gather ahead moved into LDS: kernels_deep.h, deep_row_issue_lds)
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
# -ffp-contract=on: a*b+c fuses to an FMA only inside one source expression, so every kernel instantiation (single step,
# multi-step, slab edge rows) -- in whichever translation unit -- rounds identically: results are bitwise independent of the
# kernel variant and of the slab partition.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-constant-logical-operand"]


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
    # (an object is reused only if it was compiled with these very flags: build_variant under one tag with other flags, or an
    #  LB_* define that changed, must not relink stale code)
    stamp, flags = obj + ".flags", " ".join(FLAGS + extra)
    if (os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(p) for p in deps) and
            os.path.exists(stamp) and open(stamp).read() == flags):
        return obj
    tmp = obj + ".tmp%d" % os.getpid()
    if unit in IR_ATTRS:
        _compile_patched(src, tmp, FLAGS + extra, IR_ATTRS[unit], verbose)
    else:
        cmd = [hipcc()] + FLAGS + extra + ["-c", src, "-o", tmp]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    os.replace(tmp, obj)
    open(stamp, "w").write(flags)
    return obj


def _compile_patched(src, out, flags, attr, verbose):
    """hipcc's own steps for one translation unit, with a stop in the middle: device code to optimised LLVM IR; `attr` added to the
    attribute group of every kernel (amdgpu_kernel functions); IR to a code object (llc, lld); code object bundled
    (clang-offload-bundler) and embedded into the host object (-fcuda-include-gpubinary), as `hipcc -c` does (`hipcc -###`)."""
    import re
    base = out + ".ir"
    ll, ll2, dev, hsaco, fb = base + ".ll", base + ".patched.ll", base + ".dev.o", base + ".hsaco", base + ".hipfb"
    run = (lambda c: (print(" ".join(c), flush=True), subprocess.check_call(c))) if verbose else subprocess.check_call
    run([hipcc()] + flags + ["--cuda-device-only", "-emit-llvm", "-S", src, "-o", ll])
    text = open(ll).read()
    groups = set(re.findall(r"^define [^\n]*\bamdgpu_kernel\b[^\n]*#(\d+)", text, re.M))
    if not groups:
        raise RuntimeError("%s: no amdgpu_kernel function found in the device IR" % src)
    for g in groups:
        text, n = re.subn(r"^(attributes #%s = \{ )" % g, r"\1%s " % attr.replace("\\", "\\\\"), text, flags=re.M)
        if n != 1:
            raise RuntimeError("%s: attribute group #%s not found" % (src, g))
    open(ll2, "w").write(text)
    llc = [os.path.join(LLVM_BIN, "llc"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O3", "-filetype=obj", "-relocation-model=pic", ll2, "-o", dev]
    if verbose:
        llc.insert(1, "-pass-remarks-analysis=kernel-resource-usage")
    run(llc)
    run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev])
    run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + hsaco, "-output=" + fb])
    run([hipcc()] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", src, "-o", out])
    for f in (ll, ll2, dev, hsaco, fb):
        os.remove(f)


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
    if "-DLB_DIAG" not in extra and "-DLB_DEEP_MANUAL=0" not in extra:       # (those builds leave the row in flight to the compiler)
        _check_hand_waited_gather([o for o in objs if os.path.basename(o).startswith("deep")])
    tmp = out + ".tmp%d" % os.getpid()                     # (linked beside the target, then moved into place: a reader never sees half a library)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", tmp, "-ldl"])
    os.replace(tmp, out)
    return out


def _check_hand_waited_gather(objs):
    """k_deep's row in flight sits in accumulation registers the compiler only knows as clobbered, behind wait counts written by hand
    (csrc/kernels_deep.h, LB_DEEP_MANUAL): every build that uses them is disassembled and refused if anything else touches the window
    or if the stores a wait count stands for are not all there (tools/check_agpr_window.py)."""
    import importlib.util
    tool = os.path.join(os.path.dirname(HERE), "tools", "check_agpr_window.py")
    spec = importlib.util.spec_from_file_location("check_agpr_window", tool)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(os.path.join(mod.LLVM, "llvm-objdump")):
        raise RuntimeError("llvm-objdump not found under %s: the hand-waited gather of k_deep cannot be checked" % mod.LLVM)
    for obj in objs:
        loads, reads, strays, checked, problems = mod.check_all(obj)
        if not loads:
            continue                                     # (a variant built with -DLB_DEEP_MANUAL=0: nothing to check)
        if strays or problems:
            raise RuntimeError("%s: k_deep's accumulation-register window / hand-written waits are not as written:\n  %s"
                               % (obj, "\n  ".join((strays + problems)[:10])))


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
