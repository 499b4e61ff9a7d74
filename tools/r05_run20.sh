#!/bin/bash
# round 5, GPU run 20: A/B on one box: default scheduler against -amdgpu-sched-strategy=max-ilp (whole library)
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
LB_LIB=$L/liblbhip_maxilp.so python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
for rep in 1 2; do for lib in liblbhip.so liblbhip_maxilp.so; do
  for cfg in "periodic 8192 53601" "periodic 8192 20833" "periodic 4096 53601" "pipe 8192 53601" "periodic 8192 4449" "cavity 1024 512"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
  echo -n "$lib pipe+tiff 4096 variant 4449: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc pipe --tiff --n 4096 --steps 84 --repeat 3 --variant 4449 | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_maxilp_ab.txt 2>&1
