#!/bin/bash
# round 5, GPU run 50: LB_EDGE_COST scan for k_deep<7> / k_deep<6> in the walled families after the interior strips got faster (hand-waited gather, no pairs)
set -u
cd $GRAFT_REPO_ROOT
{
for c in 1.8 2.0 2.2 2.5 2.8 3.2 1.8; do
  for cfg in "pipe 8192 53601" "cavity 8192 53601" "pipe 6144 53601" "pipe 4096 53601" "cavity 4096 53601" "pipe 8192 20833"; do set -- $cfg
    echo -n "LB_EDGE_COST=$c $1 $2 variant $3: "; LB_EDGE_COST=$c python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
  echo -n "LB_EDGE_COST=$c pipe+mask 8192 variant 53601: "; LB_EDGE_COST=$c python3 tools/run_case.py --bc pipe --mask --n 8192 --steps 84 --repeat 3 --variant 53601 | sed 's/.*\]: //'
  echo -n "LB_EDGE_COST=$c pipe --tiff 4096 variant 53601: "; LB_EDGE_COST=$c python3 tools/run_case.py --bc pipe --tiff --n 4096 --steps 84 --repeat 3 --variant 53601 | sed 's/.*\]: //'
done
} > gpurun_out/r05_edge_cost_scan2.txt 2>&1
exit 0
