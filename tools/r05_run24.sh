#!/bin/bash
# round 5, GPU run 24: shorter segments for the two seam strips of a periodic box under k_deep (LB_SEAM_COST scan)
set -u
cd $GRAFT_REPO_ROOT
{
LB_SEAM_COST=1.06 python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
for c in 1.0 1.03 1.06 1.1 1.2 1.0 1.06; do for cfg in "periodic 8192 53601" "periodic 8192 20833" "periodic 4096 53601" "periodic 6144 53601" "periodic 3072 53601"; do set -- $cfg
  echo -n "seam cost $c $1 $2 variant $3: "; LB_SEAM_COST=$c python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_seam_scan.txt 2>&1
