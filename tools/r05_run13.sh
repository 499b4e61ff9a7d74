#!/bin/bash
# round 5, GPU run 13: wall strips' true cost at one wave per SIMD (per-wave timelines), edge-cost scan
set -u
cd $GRAFT_REPO_ROOT
{
for cfg in "pipe 8192 7" "cavity 8192 7" "pipe 4096 7" "pipe 4096 6"; do set -- $cfg
  echo "=== $1 $2 depth $3"; LB_TIMELINE_BC=$1 LB_TIMELINE_DEPTH=$3 python3 tools/wave_timeline.py $2 | grep -v "^XCD\|^SIMD\|wave slot\|slowest"
done
} > gpurun_out/r05_wave_timeline_walls.txt 2>&1
{
for ec in 1.0 1.2 1.35 1.5 1.8; do for cfg in "pipe 8192" "cavity 8192" "pipe 4096" "cavity 4096"; do set -- $cfg
  echo -n "edge cost $ec $1 $2 k_deep<7>: "; LB_EDGE_COST=$ec python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 2 --variant 53601 | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_edge_cost_scan.txt 2>&1
