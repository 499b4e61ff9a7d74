#!/bin/bash
# round 5, GPU run 6: issue rate of an all-paired CU (k_deep<5,2,0> at eight waves per CU) with and without global memory traffic
set -u
cd $GRAFT_REPO_ROOT
export LB_LIB=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9/liblbhip_diag.so
{
for cfg in "520 8" "620 4" "611 4" "511 4"; do set -- $cfg
  for d in 0 1 12582912 12582913; do
    echo -n "LB_DEEP=$1 wpc=$2 diag=$d  "
    LB_DIAG=$d LB_DEEP=$1 LB_STEP2_WAVES_PER_CU=$2 python3 tools/run_case.py --bc periodic --n 8192 --steps 60 --repeat 3 | sed 's/.*\]: //'
  done
done
} > gpurun_out/r05_deep_ablate.txt 2>&1
