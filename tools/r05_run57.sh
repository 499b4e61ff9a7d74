#!/bin/bash
# round 5, GPU run 57: rocprofv3 passes of configuration 5 on the final library (wall strips' cost 2.1)
set -u
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh r05c5 --config 5 > gpurun_out/r05_profile_c5.log 2>&1
exit 0
