#!/bin/bash
# round 6, GPU run 1: the relaxation with omega folded into the density (LB_RELAX_FOLD): full GPU suite + A/B against round 5's library
set -u
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r06_run1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run1_pytest.log
ROUNDS=2 timeout 600 bash tools/gpu_ab.sh gpurun_out/r06_fold_ab.txt 2d-lb_amd/LB_D2Q9/liblbhip_r05.so 2d-lb_amd/LB_D2Q9/liblbhip.so \
  "--bc periodic --n 8192 --steps 84" "--bc periodic --n 4096 --steps 84" "--bc pipe --n 8192 --steps 84" "--bc cavity --n 4096 --steps 84" \
  "--bc pipe --tiff --n 4096 --steps 84" "--bc pipe --cyl --n 3751 --ny 1251 --steps 140" "--bc cavity --n 1024 --steps 400" \
  "--bc periodic --n 8192 --ny 1024 --steps 84" > /dev/null 2>&1
tail -5 gpurun_out/r06_run1_pytest.log
cat gpurun_out/r06_fold_ab.txt.sorted
