#!/bin/bash
# round 5, GPU run 26: A/B on one box: the next row's gather issued after the wave has waited for the current one
# (liblbhip_touch.so: the compiler's wait for the row in hand can then only cover the stores of the previous iteration, not the
# nine loads just issued) against issued first (liblbhip.so)
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
LB_LIB=$L/liblbhip_touch.so python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
for rep in 1 2; do for lib in liblbhip.so liblbhip_touch.so; do
  for cfg in "periodic 8192 53601" "periodic 8192 20833" "periodic 4096 53601" "pipe 8192 53601"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
done; done
} > gpurun_out/r05_touch_first_ab.txt 2>&1
