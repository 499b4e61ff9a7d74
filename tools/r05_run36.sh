#!/bin/bash
# round 5, GPU run 36: A/B on one box: the obstacle swap in a block that all-fluid waves skip (liblbhip.so) against unconditional selects
# (liblbhip_skip0.so): config 5's image, the reference's cylinder, a dense random mask; k_deep<7>, k_deep<6>, k_step5, k_step4, k_step
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
python3 tools/step5_check.py --no-time 2>&1 | grep -v "^checked"
for rep in 1 2; do for lib in liblbhip_skip0.so liblbhip.so; do
  for cfg in "pipe --tiff 4096 53601" "pipe --tiff 4096 20833" "pipe --tiff 4096 4449" "pipe --tiff 8192 53601" "pipe --mask 4096 53601" "pipe --mask 8192 53601" "periodic --mask 8192 53601" "pipe --cyl 4096 53601" "pipe --tiff 4096 353" "pipe --tiff 4096 9"; do set -- $cfg
      echo -n "$lib $1 $2 $3 variant $4: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 $2 --n $3 --steps 84 --repeat 3 --variant $4 | sed 's/.*\]: //'
  done
  echo -n "$lib pipe --cyl 3751x1251 variant 4449: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc pipe --cyl --n 3751 --ny 1251 --steps 100 --repeat 3 --variant 4449 | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_mask_skip2_ab.txt 2>&1
exit 0
