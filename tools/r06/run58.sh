#!/bin/bash
# round 6, GPU run 58: the rebuilt final library (only LB_DIAG-guarded source changed since the last full suite): smoke + the parity files
set -u
cd $GRAFT_REPO_ROOT
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_health.py -m gpu -q > gpurun_out/r06z_pytest_parity.log 2>&1
echo "rc=$?"; grep -n "passed\|failed" gpurun_out/r06z_pytest_parity.log | tail -2
