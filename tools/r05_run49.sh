#!/bin/bash
# round 5, GPU run 49: k_deep with strips 224 cells = 7 x 128 bytes apart (four skirt lanes per side: liblbhip_sk4.so; 37 strips at 8192
# instead of 35) against 240 = 7.5 x 128 (liblbhip.so): does the seam's position inside a 128-byte line matter?
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
LB_LIB=$L/liblbhip_sk4.so python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
for rep in 1 2; do for lib in liblbhip.so liblbhip_sk4.so; do
  for cfg in "periodic 8192 53601" "periodic 8192 20833" "periodic 4096 53601" "pipe 8192 53601"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
done; done
} > gpurun_out/r05_deep_224_ab.txt 2>&1
exit 0
