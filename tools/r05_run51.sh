#!/bin/bash
# round 5, GPU run 51: the walled families on k_deep with the wall strips' cost at 2.1 (default) against 1.8 and 2.0 / 2.2
set -u
cd $GRAFT_REPO_ROOT
{
for c in 0 1.8 2.0 2.2 0; do
  for cfg in "pipe 8192 53601" "cavity 8192 53601" "pipe 6144 53601" "pipe 4096 53601" "cavity 4096 53601" "cavity 6144 53601"; do set -- $cfg
    echo -n "LB_EDGE_COST=$c $1 $2 variant $3: "; LB_EDGE_COST=$c python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
  echo -n "LB_EDGE_COST=$c pipe --tiff 4096 variant 53601: "; LB_EDGE_COST=$c python3 tools/run_case.py --bc pipe --tiff --n 4096 --steps 84 --repeat 3 --variant 53601 | sed 's/.*\]: //'
done
python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
} > gpurun_out/r05_edge_cost_21.txt 2>&1
exit 0
