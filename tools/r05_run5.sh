#!/bin/bash
# round 5, GPU run 5: the generic deep kernel (kernels_deep.h): k_deep<D, RW, PFD> probes, periodic 8192^2 / 4096^2
set -u
cd $GRAFT_REPO_ROOT
{
for v in 620 621 611 610; do w=4; [ $v = 620 ] && w=6
  echo "== bitwise LB_DEEP=$v wpc=$w"; LB_DEEP=$v LB_STEP2_WAVES_PER_CU=$w python3 tools/step5_check.py --six --no-time 2>&1 | tail -1
done
for n in 8192 4096; do
  echo -n "k_step6 shipped  $n "; python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3 | sed 's/.*\]: //'
  for cfg in "620 6" "621 4" "611 4" "610 4" "610 6" "721 4" "711 4" "511 4" "520 8"; do set -- $cfg
    echo -n "LB_DEEP=$1 wpc=$2  $n "; LB_DEEP=$1 LB_STEP2_WAVES_PER_CU=$2 python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3 | sed 's/.*\]: //'
  done
done
} > gpurun_out/r05_deep_probe.txt 2>&1
