#!/bin/bash
# round 5, GPU run 21: A/B on one box: the gather's lane offset kept local to the block (scalar-base loads) against the shipped library
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
LB_LIB=$L/liblbhip_glo.so python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
for rep in 1 2; do for lib in liblbhip.so liblbhip_glo.so; do
  for cfg in "periodic 8192 53601" "periodic 8192 20833" "periodic 4096 53601" "pipe 8192 53601" "periodic 8192 4449" "periodic 8192 9" "cavity 1024 -1"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
  echo -n "$lib pipe+tiff 4096 variant -1: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc pipe --tiff --n 4096 --steps 84 --repeat 3 | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_glo_ab.txt 2>&1
