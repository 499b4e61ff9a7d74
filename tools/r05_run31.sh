#!/bin/bash
# round 5, GPU run 31: timing only: the steady iterations' wait for the gathered row removed (vmcnt(63): wrong results) -- what that wait
# still costs with the row gathered one iteration ahead
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
for rep in 1 2; do for lib in liblbhip.so liblbhip_nowait.so; do
  for cfg in "periodic 8192 53601" "periodic 8192 20833" "periodic 4096 53601" "pipe 8192 53601"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
done; done
} > gpurun_out/r05_nowait_timing.txt 2>&1
