#!/bin/bash
# round 5, GPU run 30: with the gathered row waited for by hand: LDS stage windows read one stage ahead (liblbhip_ahead.so) against
# where they are used (liblbhip.so), one box
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
LB_LIB=$L/liblbhip_ahead.so python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
LB_LIB=$L/liblbhip_ahead.so python3 tools/step5_check.py --six --no-time 2>&1 | grep -v "^checked"
for rep in 1 2; do for lib in liblbhip.so liblbhip_ahead.so; do
  for cfg in "periodic 8192 53601" "periodic 8192 20833" "periodic 4096 53601" "pipe 8192 53601" "cavity 8192 53601"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
  echo -n "$lib periodic+mask 8192 variant 53601: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc periodic --mask --n 8192 --steps 84 --repeat 3 --variant 53601 | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_window_ahead2_ab.txt 2>&1
