#!/bin/bash
# A/B on ONE box (boxes of the pool differ by +-5 %): library A against library B over a list of cases, alternating, ROUNDS times.
#   tools/gpu_ab.sh <out.txt> <libA.so> <libB.so> "<run_case.py args>" ["<run_case.py args>" ...]
# e.g. tools/gpu_ab.sh gpurun_out/ab.txt 2d-lb_amd/LB_D2Q9/liblbhip_r05.so 2d-lb_amd/LB_D2Q9/liblbhip.so "--bc periodic --n 8192 --steps 84"
# Environment: ROUNDS (default 3), AB_ENV_A / AB_ENV_B = extra "VAR=value ..." for either side (e.g. the same library under two
# settings of a tuning knob).
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=$1; A=$(realpath $2); B=$(realpath $3); shift 3
: > $out
for i in $(seq 1 ${ROUNDS:-3}); do
  for c in "$@"; do
    echo -n "A  " >> $out; env ${AB_ENV_A:-} LB_LIB=$A python3 tools/run_case.py $c --repeat 2 >> $out 2>&1
    echo -n "B  " >> $out; env ${AB_ENV_B:-} LB_LIB=$B python3 tools/run_case.py $c --repeat 2 >> $out 2>&1
  done
done
sort -k2 -s $out > $out.sorted
cat $out.sorted
