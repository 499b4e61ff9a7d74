#!/bin/bash
# Round 3: grids of 1024^2 .. 2048^2 (config 2's regime): LDS tiles (variant 512+...) against the paired four-step marching kernel
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for bc in cavity periodic; do
  for n in 1024 1280 1536 2048; do
    for v in 864 352 353; do        # 864 = 352 + 512: k_tile4; 352: k_step4, plain stores; 353: + non-temporal stores
      python tools/run_case.py --bc $bc --n $n --steps 400 --repeat 5 --variant $v
    done
    for wpc in 4 6; do
      LB_STEP2_WAVES_PER_CU=$wpc python tools/run_case.py --bc $bc --n $n --steps 400 --repeat 5 --variant 352 | sed "s/^/waves_per_cu=$wpc /"
    done
  done
done
