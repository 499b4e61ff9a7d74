#!/bin/bash
# round 5, GPU run 43: rocprofv3 kernel trace + FETCH/WRITE passes of the four configurations on the final library (k_step5 strips 240 apart)
set -u
cd $GRAFT_REPO_ROOT
for c in 4 3 5 2; do bash tools/gpu_profile.sh r05c$c --config $c > gpurun_out/r05_profile_c$c.log 2>&1; done
exit 0
