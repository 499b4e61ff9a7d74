#!/bin/bash
# Round 3: k_tile4 / k1_tile4 -- one band of tile rows per XCD (default) against tile = blockIdx (variant bit 14), the region load in
# 16-byte groups (default) against per-cell 4-byte loads (bit 13); 864 = 352 + 512 = four steps per launch through LDS tiles.
# Usage (GPU box): tools/r03_tile_loads.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for rep in 1 2; do
  for bc in cavity periodic pipe; do
    for n in 512 1024 1280 1536 2048; do
      for v in 864 17248 9056 353; do
        timeout 300 python tools/run_case.py --bc $bc --n $n --steps 400 --repeat 5 --variant $v
      done
    done
  done
  for v in 864 17248 9056; do
    timeout 300 python tools/run_case.py --bc pipe --mask --n 1024 --steps 400 --repeat 5 --variant $v
  done
done
timeout 600 python tools/reference_grid_bench.py
