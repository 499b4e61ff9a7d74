#!/bin/bash
# round 6: is a SIMD that holds TWO marching waves faster than one that holds one?  ("price it first with a timing-only
# probe".)  The same kernel -- k_deep<4, RW 1, gather ahead>, 213 registers, 16 KB of LDS per wave -- built twice behind the six-step
# launcher (LB_DEEP6_DEPTH=4; the host counts six steps per launch: wrong results, timing only): __launch_bounds__(128, 1) at four waves
# per CU against (128, 2) at eight (each wave then marches half the rows).  Diagnostic builds: everything / no global memory / neither
# memory nor arithmetic.  Prints microseconds per LAUNCH (8192^2 periodic).
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/r06_occ2_probe.txt
: > $out
for rep in 1 2; do
for cfg in "p4a 4" "p4b 8"; do
  set -- $cfg
  for diag in 0 12582912 12582913; do
    r=$(LB_LIB=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_$1.so LB_DIAG=$diag LB_STEP2_WAVES_PER_CU=$2 python3 tools/run_case.py --bc periodic --n 8192 --steps 60 --variant 20833 --repeat 3 2>&1 | tail -1)
    us=$(echo "$r" | sed -n 's/.* \([0-9.]*\) us per step.*/\1/p')
    echo "lib $1 waves/CU $2 LB_DIAG=$diag: launch $(python3 -c "print('%.1f' % (6*float('${us:-0}')))") us   [$r]" | cut -c1-200 >> $out
  done
done
done
cat $out
