#!/bin/bash
# round 6, GPU run 44: the LDS tiles (k_tile4) against the marching kernels in walled boxes around the tiles' static threshold (1850^2)
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06o_walled_tile_sweep.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for n in 1280 1536 1664 1792 1920 2048; do
  for fam in "--bc pipe" "--bc cavity" "--bc pipe --mask"; do
    for v in 625 4464 119152; do
      run $fam --n $n --variant $v
    done
  done
done
cat $P
