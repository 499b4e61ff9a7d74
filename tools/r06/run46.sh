#!/bin/bash
# round 6, GPU run 46: periodic boxes without a mask around the static thresholds (tiles below 1200^2, k_step5 to 1500^2, k_deep above):
# tiles | k_step5 | k_deep<6> | k_deep<7>; 1680 steps, best of 3 (the clocks have ramped by then)
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06q_periodic_small_sweep.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 1680 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for n in 1024 1152 1280 1408 1536 1792 2048; do
  for v in 625 4464 20848 53616; do
    run --bc periodic --n $n --variant $v
  done
done
cat $P
