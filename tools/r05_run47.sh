#!/bin/bash
# round 5, GPU run 47: round 4's library against this round's on one box, small grids (graph replay, tiles) and masks
set -u
cd $GRAFT_REPO_ROOT
R4=$GRAFT_REPO_ROOT/tools/_build/r04tree
{
for rep in 1 2; do
for cfg in "periodic 256" "periodic 512" "periodic 768" "pipe 512" "cavity 768" "pipe 1024" "periodic 2048" "periodic 3072"; do set -- $cfg
  echo -n "r04 $1 $2: "; (cd $R4 && python3 tools/run_case.py --bc $1 --n $2 --steps 400 --repeat 3) | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
  echo -n "r05 $1 $2: "; python3 tools/run_case.py --bc $1 --n $2 --steps 400 --repeat 3 | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
done
for cfg in "pipe 2048" "cavity 3072" "periodic 4096"; do set -- $cfg
  echo -n "r04 $1 $2 mask: "; (cd $R4 && python3 tools/run_case.py --bc $1 --mask --n $2 --steps 80 --repeat 3) | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
  echo -n "r05 $1 $2 mask: "; python3 tools/run_case.py --bc $1 --mask --n $2 --steps 80 --repeat 3 | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
done; done
} > gpurun_out/r05_vs_r04_small.txt 2>&1
exit 0
