#!/bin/bash
# round 6, GPU run 8: where k_deep2<7>'s launch goes -- diagnostic build, timing only: everything / no stores / no loads / no global memory /
# neither memory nor arithmetic; k_deep<7> beside it.  Microseconds per launch, 8192^2 periodic.
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_deep2_ablate.txt
: > $out
L=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_diag.so
for rep in 1 2; do
for v in 53601 119137; do
  for diag in 0 4194304 8388608 12582912 12582913 1; do
    r=$(LB_LIB=$L LB_DIAG=$diag python3 tools/run_case.py --bc periodic --n 8192 --steps 70 --variant $v --repeat 3 2>&1 | tail -1)
    us=$(echo "$r" | sed -n 's/.* \([0-9.]*\) us per step.*/\1/p')
    echo "variant $v LB_DIAG=$diag: launch $(python3 -c "print('%.1f' % (7*float('${us:-0}')))") us  [$(echo "$r" | cut -c1-70)]" >> $out
  done
done
done
cat $out
