#!/bin/bash
# round 5, GPU run 28: with the row in flight in accumulation registers the two row buffers no longer swap roles: the steady iterations
# in pairs (liblbhip.so: wherever there is no mask) against pairs in periodic boxes only (pairs1) and nowhere (pairs0), one box
set -u
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9
{
for lib in liblbhip_pairs0.so liblbhip_pairs1.so; do LB_LIB=$L/$lib python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"; done
for rep in 1 2; do for lib in liblbhip.so liblbhip_pairs0.so liblbhip_pairs1.so; do
  for cfg in "periodic 8192 53601" "periodic 4096 53601" "pipe 8192 53601" "cavity 8192 53601" "pipe 4096 53601"; do set -- $cfg
      echo -n "$lib $1 $2 variant $3: "; LB_LIB=$L/$lib python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $3 | sed 's/.*\]: //'
  done
done; done
} > gpurun_out/r05_pairs_ab.txt 2>&1
