#!/bin/bash
# round 5, GPU run 23: k_deep<6> with every stage window in LDS (RW = 0, liblbhip_rw0.so) against one in registers
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
LB_LIB=$L/liblbhip_rw0.so python3 tools/step5_check.py --six --no-time 2>&1 | grep -v "^checked"
for rep in 1 2; do for lib in liblbhip.so liblbhip_rw0.so; do
  for cfg in "periodic 8192 20833" "periodic 4096 20833" "pipe 8192 20833" "periodic 8192 53601"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
done; done
} > gpurun_out/r05_rw0_ab.txt 2>&1
