#!/usr/bin/env python3
"""Print the tail of a rocprofv3 --kernel-trace CSV as a timeline (us relative to the first printed
kernel): which queue, start, duration, gap to the previous kernel on the same queue.
Usage: tools/timeline.py <dir or kernel_trace.csv> [count] [name]      (name: the window is centred on the middle launch of that kernel)"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    if os.path.isdir(path):
        files = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
        path = files[-1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if len(sys.argv) > 3:
        hits = [i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"]]
        if not hits:
            print("no kernel named *%s* among %d launches" % (sys.argv[3], len(rows)))
            return
        mid = hits[len(hits) // 2]
        rows = rows[max(0, mid - count // 2):mid + count // 2]
    else:
        rows = rows[-count:]
    t0 = int(rows[0]["Start_Timestamp"])
    last_end = {}
    for r in rows:
        q = r.get("Queue_Id", "?")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = e
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        wg = int(r.get("Workgroup_Size_X") or 0) or 1
        shape = "%5d x %-4d lds %6s vgpr %3s+%-3s" % (int(r.get("Grid_Size_X") or 0) // wg, wg, r.get("LDS_Block_Size", "?"),
                                                     r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"))
        print("q%-3s start %9.1f  dur %8.1f  gap_same_q %7.1f  wgs %s %s"
              % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, shape, name))


if __name__ == "__main__":
    main()
