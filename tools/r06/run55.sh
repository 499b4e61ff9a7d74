#!/bin/bash
# round 6, GPU run 55: more of the randomised slab checks on the last library: 96 random RCCL self-rings (every fourth on the deep
# cycles, k_deep2 among them; odd seeds with the exchange between the launches), 120 seeds of the kernel-variant / partition tests
set -u
cd $GRAFT_REPO_ROOT
LB_RANDOM_RING_SEEDS=96 LB_RANDOM_SEEDS=120 timeout 1700 python3 -m pytest tests/test_gpu_random.py -m gpu -q > gpurun_out/r06x_random.txt 2>&1
echo "rc=$?" >> gpurun_out/r06x_random.txt
tail -4 gpurun_out/r06x_random.txt
