#!/bin/bash
# round 6, GPU run 48: k_deep2<7> against k_deep<7> on the plain grids a slab of 8 | 4 | 2 ranks holds (short segments), periodic and pipe
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06s_deep2_slab_shapes.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for rep in 1 2; do
for ny in 1024 2048 4096; do
  for v in 53601 119137 20833; do
    run --bc periodic --n 8192 --ny $ny --variant $v
  done
done
done
cat $P
