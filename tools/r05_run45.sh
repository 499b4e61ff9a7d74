#!/bin/bash
# round 5, GPU run 45: round 4's library (tree at a12bcc1, built into tools/_build/r04tree) against this round's on one box, automatic
# kernel choice, at the sizes that still run the kernels round 4 had: is anything slower than it was?
set -u
cd $GRAFT_REPO_ROOT
R4=$GRAFT_REPO_ROOT/tools/_build/r04tree
{
for rep in 1 2; do
for cfg in "velocity_inlet 4096" "velocity_inlet 8192" "velocity_inlet 2048" "pipe 3072" "pipe 2048" "cavity 2048" "cavity 1024" "periodic 1280" "periodic 1024" "pipe 1536"; do set -- $cfg
  echo -n "r04 $1 $2: "; (cd $R4 && python3 tools/run_case.py --bc $1 --n $2 --steps 80 --repeat 3) | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
  echo -n "r05 $1 $2: "; python3 tools/run_case.py --bc $1 --n $2 --steps 80 --repeat 3 | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
done
echo -n "r04 pipe+mask 3751x1251: "; (cd $R4 && python3 tools/run_case.py --bc pipe --mask --n 3751 --ny 1251 --steps 100 --repeat 3) | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
echo -n "r05 pipe+mask 3751x1251: "; python3 tools/run_case.py --bc pipe --mask --n 3751 --ny 1251 --steps 100 --repeat 3 | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
done
} > gpurun_out/r05_vs_r04_one_box.txt 2>&1
exit 0
