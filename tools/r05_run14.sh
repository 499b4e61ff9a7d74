#!/bin/bash
# round 5, GPU run 14: edge-cost scan, upper range; both depths
set -u
cd $GRAFT_REPO_ROOT
{
for ec in 1.8 2.0 2.2 2.5 3.0; do for cfg in "pipe 8192" "cavity 8192" "pipe 4096" "cavity 4096" "pipe 6144" "pipe 3072"; do set -- $cfg
  for v in 53601 20833; do
  echo -n "edge cost $ec $1 $2 variant $v: "; LB_EDGE_COST=$ec python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
  done
done; done
} > gpurun_out/r05_edge_cost_scan2.txt 2>&1
