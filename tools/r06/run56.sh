#!/bin/bash
# round 6, GPU run 56: SQ counters of k_deep2<7> (8192^2 periodic, 140 steps per pass, four --pmc passes), k_deep<7> beside it on the same box
set -u
cd $GRAFT_REPO_ROOT
timeout 600 bash tools/gpu_pmc_case.sh r06deep2 --bc periodic --n 8192 --steps 140 --variant 119137 > gpurun_out/r06_sq_deep2.txt 2>&1
timeout 600 bash tools/gpu_pmc_case.sh r06deep7b --bc periodic --n 8192 --steps 140 --variant 53601 > gpurun_out/r06_sq_deep7b.txt 2>&1
grep "k_deep2" gpurun_out/r06_sq_deep2.txt | head -40
grep "k_deep<1, false, false, 7" gpurun_out/r06_sq_deep7b.txt | head -40
