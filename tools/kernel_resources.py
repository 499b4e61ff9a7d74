#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in liblbhip (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: tools/kernel_resources.py [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    src = os.path.join(ROOT, "2d-lb_amd", "csrc", "lb_hip.cpp")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC", "-c", src,
           "-o", "/tmp/lb_resources.o", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (.*?)\s*\[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    print("%-46s %5s %5s %7s %5s %9s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "SGPRspill"))
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", name)
        if flt and flt not in name:
            continue
        print("%-46s %5s %5s %7s %5s %9s" % (name[:46], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
                                           r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill")))


if __name__ == "__main__":
    main()
