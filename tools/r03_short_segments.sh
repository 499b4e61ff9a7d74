#!/bin/bash
# Round 3: what bounds k_step4 on the short segments of an 8-GPU slab (8192 x 1024 per GPU).  Usage (GPU box): tools/r03_short_segments.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for bit in 0 10 11 12 13; do
  LB_PRIO_TURN_BIT=$bit python tools/run_case.py --n 8192 --ny 1024 --steps 96 --repeat 5 | sed "s/^/prio_turn_bit=$bit  /"
done
for wpc in 4 6 8; do
  LB_STEP2_WAVES_PER_CU=$wpc python tools/run_case.py --n 8192 --ny 1024 --steps 96 --repeat 5 | sed "s/^/waves_per_cu=$wpc  /"
done
for ny in 512 1024 2048 4096 8192; do
  python tools/run_case.py --n 8192 --ny $ny --steps 96 --repeat 5
done
