#!/bin/bash
# round 5, GPU run 48: pipe + mask 2048^2 (the one case slower than round 4): strips 240 apart (liblbhip.so) / 248 apart (liblbhip_s5u.so) / round 4
set -u
cd $GRAFT_REPO_ROOT
R4=$GRAFT_REPO_ROOT/tools/_build/r04tree
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
for rep in 1 2 3; do
for cfg in "pipe 2048" "pipe 1536" "pipe 2560" "cavity 2048"; do set -- $cfg
  echo -n "r04 $1 $2 mask: "; (cd $R4 && python3 tools/run_case.py --bc $1 --mask --n $2 --steps 80 --repeat 3 --variant 4449) | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
  echo -n "r05-248 $1 $2 mask: "; LB_LIB=$L/liblbhip_s5u.so python3 tools/run_case.py --bc $1 --mask --n $2 --steps 80 --repeat 3 --variant 4449 | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
  echo -n "r05-240 $1 $2 mask: "; python3 tools/run_case.py --bc $1 --mask --n $2 --steps 80 --repeat 3 --variant 4449 | sed 's/.*\[\(k[^ ]*\).*\]: /\1 /'
done; done
} > gpurun_out/r05_pipe_mask_2048.txt 2>&1
exit 0
