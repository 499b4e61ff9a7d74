#!/bin/bash
# round 5, GPU run 27: A/B on one box: the row gathered ahead waited for by hand (liblbhip.so: asm loads into a fixed window of
# accumulation registers, s_waitcnt vmcnt(9) = the stores of the iteration in between) against the compiler's waits (liblbhip_auto.so)
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
python3 tools/step5_check.py --six --no-time 2>&1 | grep -v "^checked"
for rep in 1 2; do for lib in liblbhip_auto.so liblbhip.so; do
  for cfg in "periodic 8192 53601" "periodic 8192 20833" "periodic 4096 53601" "pipe 8192 53601" "cavity 8192 53601"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
  echo -n "$lib periodic+mask 8192 variant 53601: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc periodic --mask --n 8192 --steps 84 --repeat 3 --variant 53601 | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_manual_wait_ab.txt 2>&1
