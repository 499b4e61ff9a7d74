#!/bin/bash
# round 6, GPU run 43: walled boxes below k_deep's static threshold (2300^2): k_step5 | k_deep<7> | k_deep2<7>, 840 steps, best of 3
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06o_walled_small_sweep.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for n in 1280 1536 1792 2048 2304 2560; do
  if [ $n -ge 2110 ]; then nt=1; else nt=16; fi
  for fam in "--bc pipe" "--bc cavity" "--bc pipe --mask" "--bc periodic --mask"; do
    for v in $((4448 + nt)) $((53600 + nt)) $((53600 + 65536 + nt)); do
      run $fam --n $n --variant $v
    done
  done
done
cat $P
