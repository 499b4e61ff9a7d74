#!/bin/bash
# round 6, GPU run 45: the static kernel table after its round-6 revision (walled: tiles below 1450^2, k_step5 to 1700^2, k_deep2 to
# 2900^2, k_deep<7> above; periodic with a mask: k_deep from 1250^2): full GPU suite; static choice against lb_autotune across sizes
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06p_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06p_pytest_gpu.log
timeout 900 python3 tools/tune_probe.py 1280 1536 1792 2048 2560 3072 4096 2>&1 | cut -c1-120 > gpurun_out/r06p_tune_probe.txt
timeout 200 python3 tools/reference_grid_bench.py > gpurun_out/r06p_reference_grid.txt 2>&1
tail -3 gpurun_out/r06p_pytest_gpu.log
cat gpurun_out/r06p_tune_probe.txt
grep opencl gpurun_out/r06p_reference_grid.txt
