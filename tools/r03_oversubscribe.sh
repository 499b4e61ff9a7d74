#!/bin/bash
# Round 3: more, shorter segment pairs than resident wave slots (k_step4 holds 8 waves per CU; LB_STEP2_WAVES_PER_CU > 8 cuts
# the grid for more): does handing out work in smaller pieces shorten the launch's tail (a wave slot is busy ~92 % of it)?
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for shape in "8192 8192" "4096 4096" "8192 1024"; do
  set -- $shape
  for wpc in 8 10 12 16 24; do
    LB_STEP2_WAVES_PER_CU=$wpc python tools/run_case.py --n $1 --ny $2 --steps 96 --repeat 5 | sed "s/(marching[^)]*)//; s/^/waves_per_cu=$wpc /"
  done
done
for bc in pipe cavity; do
  for wpc in 8 12 16; do
    LB_STEP2_WAVES_PER_CU=$wpc python tools/run_case.py --bc $bc --n 8192 --steps 96 --repeat 5 | sed "s/(marching[^)]*)//; s/^/waves_per_cu=$wpc /"
  done
done
