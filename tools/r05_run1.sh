#!/bin/bash
# round 5, GPU run 1: SQ counters of k_step6 (8192^2, 4096^2 periodic) + waves-per-CU scan of the shipped kernel
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_pmc_case.sh r05s6_8192 --bc periodic --n 8192 --steps 30 > gpurun_out/r05_sq_8192.txt 2>&1
bash tools/gpu_pmc_case.sh r05s6_4096 --bc periodic --n 4096 --steps 30 > gpurun_out/r05_sq_4096.txt 2>&1
{
for w in 6 4 5; do
  echo "== LB_STEP2_WAVES_PER_CU=$w"
  LB_STEP2_WAVES_PER_CU=$w python3 tools/run_case.py --bc periodic --n 8192 --steps 60 --repeat 3
  LB_STEP2_WAVES_PER_CU=$w python3 tools/run_case.py --bc periodic --n 4096 --steps 60 --repeat 3
done
echo "== k_step5 8 waves"
python3 tools/run_case.py --bc periodic --n 8192 --steps 60 --repeat 3 --variant 4449
rocm-smi --showclocks 2>/dev/null | head -30
} > gpurun_out/r05_wpc_scan.txt 2>&1
