#!/bin/bash
# round 5, GPU run 8: which ingredient of k_deep<PIPE, MASK, 7, 2, 1> breaks it (RW = 1 / PFD = 0 alternatives); SQ counters of pipe k_deep<6>
set -u
cd $GRAFT_REPO_ROOT
{
for alt in 0 1 2; do echo "== LB_DEEP7_ALT=$alt"; LB_DEEP7_ALT=$alt python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"; done
} > gpurun_out/r05_deep7_debug.txt 2>&1
bash tools/gpu_pmc_case.sh r05d6pipe --bc pipe --n 8192 --steps 30 --variant 20833 > gpurun_out/r05_sq_deep6_pipe.txt 2>&1
bash tools/gpu_pmc_case.sh r05d6per --bc periodic --n 8192 --steps 30 --variant 20833 > gpurun_out/r05_sq_deep6_periodic.txt 2>&1
