#!/bin/bash
# round 5, GPU run 40: the new eager-macro test of the deep kernels
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_health.py -m gpu -x -q -k "deep_kernels_store" > gpurun_out/r05_eager_deep_test.txt 2>&1
exit 0
