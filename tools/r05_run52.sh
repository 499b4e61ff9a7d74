#!/bin/bash
# round 5, GPU run 52: LB_EDGE_COST scan for k_step5 in the walled families (strips 240 apart, the split searched from the closed form)
set -u
cd $GRAFT_REPO_ROOT
{
for c in 0 1.0 1.2 1.5 1.8 2.1 2.5 0; do
  for cfg in "pipe 2048 4449" "pipe 3072 4449" "cavity 3072 4449" "pipe 3584 4449" "cavity 2048 4449"; do set -- $cfg
    echo -n "LB_EDGE_COST=$c $1 $2 variant $3: "; LB_EDGE_COST=$c python3 tools/run_case.py --bc $1 --n $2 --steps 80 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
  echo -n "LB_EDGE_COST=$c pipe --cyl 3751x1251 variant 4449: "; LB_EDGE_COST=$c python3 tools/run_case.py --bc pipe --cyl --n 3751 --ny 1251 --steps 100 --repeat 3 --variant 4449 | sed 's/.*\]: //'
done
} > gpurun_out/r05_edge_cost_step5.txt 2>&1
exit 0
