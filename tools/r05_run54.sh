#!/bin/bash
# round 5, GPU run 54: the reference case's geometry (3751 x 1251, pipe + disc) per kernel after the split repair; the automatic choice around the new thresholds
set -u
cd $GRAFT_REPO_ROOT
{
for rep in 1 2; do for v in 4449 20833 53601 -1; do echo -n "pipe --cyl 3751x1251 variant $v: "; python3 tools/run_case.py --bc pipe --cyl --n 3751 --ny 1251 --steps 100 --repeat 3 --variant $v | sed 's/.*\]: //'; done; done
for cfg in "pipe 2304" "pipe 2560" "cavity 2560" "pipe 3072"; do set -- $cfg
  echo -n "$1 $2 automatic: "; python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
  echo -n "$1 $2 mask automatic: "; python3 tools/run_case.py --bc $1 --mask --n $2 --steps 84 --repeat 3 | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
done
} > gpurun_out/r05_refcase_kernels.txt 2>&1
exit 0
