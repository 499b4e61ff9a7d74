#!/bin/bash
# round 6, GPU run 59: obstacle masks at full size: k_deep<7> | k_deep2<7>, periodic 8192^2 with 1 % random solid cells, pipe + mask 8192^2,
# config 5's image 4096^2, unmasked beside them
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06z_mask_deep2.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 420 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for v in 53601 119137; do
  run --bc periodic --n 8192 --variant $v
  run --bc periodic --mask --n 8192 --variant $v
  run --bc pipe --n 8192 --variant $v
  run --bc pipe --mask --n 8192 --variant $v
  run --bc pipe --tiff --n 4096 --variant $v
  run --bc cavity --cyl --n 6144 --variant $v
done
sort $P
