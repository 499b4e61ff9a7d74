#!/bin/bash
# round 5, GPU run 39: the inner half of the two skirt lanes stored as well (strips 244 apart: liblbhip.so) against whole skirt lanes
# (240 apart: liblbhip_half0.so): bitwise checks, the strip-boundary test, A/B on one box
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
python3 tools/step5_check.py --six --no-time 2>&1 | grep -v "^checked"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "strip_boundaries" 2>&1 | tail -3
for rep in 1 2; do for lib in liblbhip_half0.so liblbhip.so; do
  for cfg in "periodic 8192 53601" "periodic 8192 20833" "periodic 4096 53601" "periodic 2048 53601" "pipe 8192 53601" "cavity 8192 53601" "pipe 4096 53601"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
  echo -n "$lib periodic+mask 8192 variant 53601: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc periodic --mask --n 8192 --steps 84 --repeat 3 --variant 53601 | sed 's/.*\]: //'
  echo -n "$lib pipe --tiff 4096 variant 53601: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc pipe --tiff --n 4096 --steps 84 --repeat 3 --variant 53601 | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_half_lanes_ab.txt 2>&1
exit 0
