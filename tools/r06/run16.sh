#!/bin/bash
# round 6, GPU run 16: more of the randomised checks on the final library -- 150 seeds of tests/test_gpu_random.py (kernel variants incl.
# k_deep2, random slab partitions through lb_run_group with thick bands, RCCL self-rings with split bands), tools/slab_stress.py with 200
# random partitions beside three noise processes, four rank processes on one GPU over the peer transport
set -u
cd $GRAFT_REPO_ROOT
LB_RANDOM_SEEDS=150 timeout 1500 python3 -m pytest tests/test_gpu_random.py -m gpu -q > gpurun_out/r06_random150.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_random150.txt
timeout 900 python3 tools/slab_stress.py 200 3 > gpurun_out/r06_slab_stress_200.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_slab_stress_200.txt
timeout 600 python3 tools/peer_ranks_check.py --ranks 4 > gpurun_out/r06_peer_ranks4.txt 2>&1
echo "rc=$?" >> gpurun_out/r06_peer_ranks4.txt
tail -4 gpurun_out/r06_random150.txt; tail -3 gpurun_out/r06_slab_stress_200.txt; tail -5 gpurun_out/r06_peer_ranks4.txt
