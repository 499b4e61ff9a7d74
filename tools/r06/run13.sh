#!/bin/bash
# round 6, GPU run 13: what k_deep2's two barriers per row cost -- diagnostic build, timing only (races): steady state without barriers
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_deep2_nobarrier.txt
: > $out
L=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_diag.so
for rep in 1 2; do
  for diag in 0 16777216 12582912 29360128 29360129; do
    r=$(LB_LIB=$L LB_DIAG=$diag python3 tools/run_case.py --bc periodic --n 8192 --steps 70 --variant 119137 --repeat 3 2>&1 | tail -1)
    us=$(echo "$r" | sed -n 's/.* \([0-9.]*\) us per step.*/\1/p')
    echo "k_deep2<7> LB_DIAG=$diag: launch $(python3 -c "print('%.1f' % (7*float('${us:-0}')))") us" >> $out
  done
done
cat $out
