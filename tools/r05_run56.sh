#!/bin/bash
# round 5, GPU run 56: the Cython-path tests on the DIAGNOSTIC build (k1_step5 lives there; its strips follow k_step5's: 240 apart)
set -u
cd $GRAFT_REPO_ROOT
LB_LIB=$GRAFT_REPO_ROOT/2d-lb_amd/LB_D2Q9/liblbhip_diag.so timeout 1500 python3 -m pytest tests/test_gpu_cython_path.py -m gpu -x -q > gpurun_out/r05_diag_cython_tests.txt 2>&1
exit 0
