#!/bin/bash
# round 6, GPU run 57: per-wave timeline of one k_deep2<7> launch (diagnostic build, LB_DIAG bit 12), 8192^2 periodic; k_deep<7> beside it
set -u
cd $GRAFT_REPO_ROOT
LB_TIMELINE_DEEP2=1 LB_TIMELINE_DEPTH=7 timeout 200 python3 tools/wave_timeline.py 8192 > gpurun_out/r06y_wave_timeline_deep2.txt 2>&1
LB_TIMELINE_DEPTH=7 timeout 200 python3 tools/wave_timeline.py 8192 > gpurun_out/r06y_wave_timeline_deep7.txt 2>&1
cat gpurun_out/r06y_wave_timeline_deep2.txt | cut -c1-400
head -12 gpurun_out/r06y_wave_timeline_deep7.txt | cut -c1-300
