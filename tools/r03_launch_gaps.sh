#!/bin/bash
# Round 3: what a launch of k_step4 costs beyond its waves' work: kernel-trace of back-to-back launches on grids of
# 8192 x {512, 1024, 2048, 8192}; prints per kernel the average duration and the average gap to the next launch.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ny in 512 1024 2048 8192; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/ny$ny -- python3 $REPO/tools/run_case.py --n 8192 --ny $ny --steps 96 --repeat 3 > $OUT/ny$ny.log 2>&1
  python3 - $OUT/ny$ny $ny <<'PY'
import csv, glob, sys, statistics as st
rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "k_step4" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gap = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
gap = [g for g in gap if g < 200]          # (launches of one timed run; the pauses between runs are host time)
print("8192 x %s: %d launches of k_step4, duration avg %.1f us (min %.1f), gap to the next launch avg %.1f us (median %.1f, min %.1f)"
      % (sys.argv[2], len(dur), st.mean(dur), min(dur), st.mean(gap), st.median(gap), min(gap)))
PY
  tail -1 $OUT/ny$ny.log
done
