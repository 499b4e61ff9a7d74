#!/bin/bash
# round 6, GPU run 51: the slab tests again (lb_set_slab_cycle(8) beats the variant's bit)
set -u
cd $GRAFT_REPO_ROOT
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r06t_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06t_pytest_gpu.log
tail -4 gpurun_out/r06t_pytest_gpu.log
