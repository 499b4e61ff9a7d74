#!/bin/bash
# round 5, GPU run 34: the GPU suite with the lowered k_deep thresholds; rocprofv3 kernel trace + FETCH/WRITE passes of the four configurations
set -u
cd $GRAFT_REPO_ROOT
timeout 3000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_d.txt 2>&1
for c in 4 3 5 2; do bash tools/gpu_profile.sh r05c$c --config $c > gpurun_out/r05_profile_c$c.log 2>&1; done
