#!/bin/bash
# round 6, GPU run 7: k_deep2 with bare barriers (no memory drain) and the roles interleaved over the two workgroups of a CU (against: not interleaved)
set -u
cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/step5_check.py --deep2 --sizes 8192,4096 > gpurun_out/r06_deep2_check2.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_deep2_check2.txt
echo "== roles not interleaved (LB_DEEP2_SWAP=0)" >> gpurun_out/r06_deep2_check2.txt
LB_LIB=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_d2ns.so timeout 600 python3 tools/step5_check.py --deep2 --sizes 8192,4096 >> gpurun_out/r06_deep2_check2.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_deep2_check2.txt
grep -v "^checked" gpurun_out/r06_deep2_check2.txt | tail -60
