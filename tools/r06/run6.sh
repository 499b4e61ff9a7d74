#!/bin/bash
# round 6, GPU run 6: k_deep2 (two waves per strip and direction, two waves per SIMD, the row gathered ahead loaded straight into LDS):
# bitwise against the single-step kernel in every family, then timed against k_deep<6> / k_deep<7>
set -u
cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/step5_check.py --deep2 --sizes 8192,4096 > gpurun_out/r06_deep2_check.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_deep2_check.txt
cat gpurun_out/r06_deep2_check.txt | tail -45
