#!/bin/bash
# round 5, GPU run 2: k_step6 with rows gathered ahead at one wave per SIMD (probe), against the shipped form
set -u
cd $GRAFT_REPO_ROOT
{
for pfd in 1 2; do LB_STEP6_PFD=$pfd LB_STEP2_WAVES_PER_CU=4 python3 tools/step5_check.py --six --no-time 2>&1 | tail -4; done
for n in 8192 4096; do
  echo "== shipped (6 waves per CU)"; python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3
  for pfd in 1 2; do for w in 4 5 6; do
    echo "== PFD=$pfd wpc=$w"; LB_STEP6_PFD=$pfd LB_STEP2_WAVES_PER_CU=$w python3 tools/run_case.py --bc periodic --n $n --steps 60 --repeat 3
  done; done
done
} > gpurun_out/r05_pfd_probe.txt 2>&1
