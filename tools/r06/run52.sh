#!/bin/bash
# round 6, GPU run 52: k_deep<7> against k_deep2<7> on the headline grid (8192^2 periodic) and on 4096^2, alternating, three rounds
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06u_deep2_headline.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for rep in 1 2 3; do
  for v in 53601 119137; do
    run --bc periodic --n 8192 --variant $v
    run --bc periodic --n 4096 --variant $v
  done
done
sort $P
