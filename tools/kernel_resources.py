#!/usr/bin/env python3
"""Register / scratch / LDS table of the kernels of one translation unit of liblbhip (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: tools/kernel_resources.py <unit.cpp> [name-filter] [-D...]      e.g.  tools/kernel_resources.py deep6.cpp k_deep"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-D")]
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    unit = args[0] if args else "lb_hip.cpp"
    flt = args[1] if len(args) > 1 else ""
    src = os.path.join(ROOT, "2d-lb_amd", "csrc", unit)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC", "-c", src,
           "-o", "/tmp/lb_resources.o", "-Rpass-analysis=kernel-resource-usage"] + defs
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (.*?)\s*\[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        t = re.sub(r"^\S+:\d+:\d+:\s*", "", t)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    print("%-58s %5s %5s %7s %5s %9s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "SGPRspill", "LDS"))
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", name)
        if flt and flt not in name:
            continue
        print("%-58s %5s %5s %7s %5s %9s %7s" % (name[:58], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
                                               r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"), r.get("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()
