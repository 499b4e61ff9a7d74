#!/bin/bash
# round 5, GPU run 12: k_deep's code footprint -- boundary rule out of line; steady iterations in pairs or not; interior strips' own code path or not
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
for lib in liblbhip.so liblbhip_u0.so liblbhip_u0s1.so liblbhip_u1s1.so; do
  echo "=== $lib"
  LB_LIB=$L/$lib python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
  for cfg in "periodic 8192" "pipe 8192" "cavity 8192" "pipe 4096" "periodic 4096"; do set -- $cfg
    for v in 20833 53601; do
      echo -n "$1 $2 variant $v: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
    done
  done
  for v in 20833 53601; do
    echo -n "pipe+mask 4096 variant $v: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc pipe --mask --n 4096 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
    echo -n "periodic+mask 8192 variant $v: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc periodic --mask --n 8192 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
  done
done
} > gpurun_out/r05_footprint.txt 2>&1
