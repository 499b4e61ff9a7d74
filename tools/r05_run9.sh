#!/bin/bash
# round 5, GPU run 9: k_deep with eight-slot LDS windows (D = 7: RW = 1): bitwise, timing, per-wave timeline of the pipe family
set -u
cd $GRAFT_REPO_ROOT
{
python3 tools/step5_check.py --six --no-time 2>&1 | grep -v "^checked"
python3 tools/step5_check.py --seven --sizes 8192,4096 2>&1 | grep -v "^checked"
} > gpurun_out/r05_deep_families2.txt 2>&1
{
for bc in periodic pipe; do for d in 6 7; do
  echo "=== $bc depth $d"; LB_TIMELINE_BC=$bc LB_TIMELINE_DEPTH=$d python3 tools/wave_timeline.py 8192
done; done
} > gpurun_out/r05_wave_timeline_deep.txt 2>&1
