#!/bin/bash
# round 5, GPU run 41: k_step5 with strips 240 cells = 15 x 64 bytes apart (liblbhip_s5a.so: every strip's stores begin and end on
# 64-byte boundaries, 3 % more strips) against 248 apart (liblbhip.so: 32-byte boundaries), one box
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
LB_LIB=$L/liblbhip_s5a.so python3 tools/step5_check.py --no-time 2>&1 | grep -v "^checked"
for rep in 1 2; do for lib in liblbhip.so liblbhip_s5a.so; do
  for cfg in "periodic 8192 4449" "periodic 4096 4449" "periodic 2048 4449" "pipe 8192 4449" "pipe 4096 4449" "pipe 3072 4449" "cavity 3072 4449" "velocity_inlet 4096 4449"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 80 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
  echo -n "$lib pipe --tiff 4096 variant 4449: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc pipe --tiff --n 4096 --steps 80 --repeat 3 --variant 4449 | sed 's/.*\]: //'
  echo -n "$lib pipe --cyl 3751x1251 variant 4449: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc pipe --cyl --n 3751 --ny 1251 --steps 100 --repeat 3 --variant 4449 | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_step5_align64_ab.txt 2>&1
exit 0
