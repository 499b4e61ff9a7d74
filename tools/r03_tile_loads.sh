#!/bin/bash
# Round 3: k_tile4 -- one band of tile rows per XCD (default; 864 = 352 + 512 = four steps per launch through LDS tiles) against
# tile = blockIdx (variant bit 13: 9056) and against the paired marching kernel (353); then the Cython-path GPU mode (k1_tile4, same
# mapping) on the reference's 3751 x 1251 case.  (The numbers of profiles/r03_experiments.txt section 15 also list a build with
# 16-byte region loads, which was not kept.)  Usage (GPU box): tools/r03_tile_loads.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for rep in 1 2; do
  for bc in cavity periodic pipe; do
    for n in 512 1024 1280 1536 1792 1920 2048; do
      for v in 864 9056 353; do
        timeout 300 python tools/run_case.py --bc $bc --n $n --steps 400 --repeat 5 --variant $v
      done
    done
  done
  for v in 864 9056; do
    timeout 300 python tools/run_case.py --bc pipe --mask --n 1024 --steps 400 --repeat 5 --variant $v
  done
done
timeout 600 python tools/reference_grid_bench.py
