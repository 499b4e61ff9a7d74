#!/bin/bash
# round 5, GPU run 19: A/B on one box: shifted LDS window reads (liblbhip.so) against register shifts (liblbhip_noshift.so)
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
for rep in 1 2; do for lib in liblbhip.so liblbhip_noshift.so; do
  for cfg in "periodic 8192" "periodic 4096" "pipe 8192"; do set -- $cfg
    for v in 20833 53601; do
      echo -n "$lib $1 $2 variant $v: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $v | sed 's/.*\]: //'
    done
  done
done; done
} > gpurun_out/r05_shifted_ab.txt 2>&1
