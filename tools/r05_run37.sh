#!/bin/bash
# round 5, GPU run 37: A/B on one box: periodic k_deep told that every row exists (three branches and the zero-filled row gone:
# liblbhip.so) against testing step1_rows' answer (liblbhip_have0.so)
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
python3 tools/step5_check.py --six --no-time 2>&1 | grep -v "^checked"
for rep in 1 2 3; do for lib in liblbhip_have0.so liblbhip.so; do
  for cfg in "periodic 8192 53601" "periodic 8192 20833" "periodic 4096 53601" "periodic 2048 53601"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
  echo -n "$lib periodic+mask 8192 variant 53601: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc periodic --mask --n 8192 --steps 84 --repeat 3 --variant 53601 | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_periodic_have_ab.txt 2>&1
exit 0
