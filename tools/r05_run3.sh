#!/bin/bash
# round 5, GPU run 3: where k_step6's millisecond goes -- timing-only ablations on the diagnostic build (8192^2 periodic)
set -u
cd $GRAFT_REPO_ROOT
export LB_LIB=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9/liblbhip_diag.so
N=${1:-8192}
{
for cfg in "0 6" "1 4"; do
  set -- $cfg; pfd=$1; w=$2
  for d in 0 1 2097152 2097153 4194304 8388608 12582912 12582913 14680064 14680065; do
    echo -n "pfd=$pfd wpc=$w diag=$d  "
    LB_DIAG=$d LB_STEP6_PFD=$pfd LB_STEP2_WAVES_PER_CU=$w python3 tools/run_case.py --bc periodic --n $N --steps 60 --repeat 3 | sed 's/.*\]: //'
  done
done
} > gpurun_out/r05_step6_ablate_$N.txt 2>&1
