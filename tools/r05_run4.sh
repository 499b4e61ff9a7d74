#!/bin/bash
# round 5, GPU run 4: single-wave VALU issue rate; the skirt shifts as DPP moves instead of ds_bpermute (liblbhip_dpp.so)
set -u
cd $GRAFT_REPO_ROOT
tools/_build/valu_probe > gpurun_out/r05_valu_issue.txt 2>&1
D=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9/liblbhip_dpp.so
{
LB_LIB=$D python3 tools/step5_check.py --six --no-time 2>&1 | tail -3
LB_LIB=$D python3 tools/step5_check.py --no-time 2>&1 | tail -2
for n in 8192 4096; do
  echo -n "shipped  $n "; python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3 | sed 's/.*\]: //'
  echo -n "dpp      $n "; LB_LIB=$D python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3 | sed 's/.*\]: //'
  echo -n "shipped  $n "; python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3 | sed 's/.*\]: //'
  echo -n "dpp      $n "; LB_LIB=$D python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3 | sed 's/.*\]: //'
  echo -n "dpp pfd1 wpc4 $n "; LB_LIB=$D LB_STEP6_PFD=1 LB_STEP2_WAVES_PER_CU=4 python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3 | sed 's/.*\]: //'
  echo -n "shipped k_step5 $n "; python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3 --variant 4449 | sed 's/.*\]: //'
  echo -n "dpp k_step5     $n "; LB_LIB=$D python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3 --variant 4449 | sed 's/.*\]: //'
done
} > gpurun_out/r05_dpp.txt 2>&1
