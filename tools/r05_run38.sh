#!/bin/bash
# round 5, GPU run 38: the tune-cache test + the parity file it lives in
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tune_cache or launch_plan or quick" > gpurun_out/r05_tune_cache_test.txt 2>&1
exit 0
