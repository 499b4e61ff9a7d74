#!/bin/bash
# round 6, GPU run 9: full GPU suite on the library with k_deep2 (its variants are in the test matrices now), k_deep2 re-checked after the
# M0 save / restore
set -u
cd $GRAFT_REPO_ROOT
timeout 300 python3 tools/step5_check.py --deep2 --no-time > gpurun_out/r06_deep2_check3.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_deep2_check3.txt
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run9_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run9_pytest.log
tail -3 gpurun_out/r06_deep2_check3.txt
tail -15 gpurun_out/r06_run9_pytest.log
