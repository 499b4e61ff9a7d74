#!/bin/bash
# round 5, GPU run 46: the wall-strip split searched from the closed form downwards: the walled families, automatic choice, against round 4's library
set -u
cd $GRAFT_REPO_ROOT
R4=$GRAFT_REPO_ROOT/tools/_build/r04tree
{
for rep in 1 2; do
for cfg in "velocity_inlet 4096" "velocity_inlet 3072" "velocity_inlet 6144" "velocity_inlet 8192" "pipe 3072" "pipe 4096" "cavity 4096" "pipe 8192" "cavity 2048"; do set -- $cfg
  echo -n "r04 $1 $2: "; (cd $R4 && python3 tools/run_case.py --bc $1 --n $2 --steps 80 --repeat 3) | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
  echo -n "r05 $1 $2: "; python3 tools/run_case.py --bc $1 --n $2 --steps 80 --repeat 3 | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
done; done
python3 tools/step5_check.py --no-time 2>&1 | grep -v "^checked"
python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
} > gpurun_out/r05_vs_r04_walls.txt 2>&1
exit 0
