#!/bin/bash
# round 6, GPU run 49: k_deep2<7> inside the slab cycle (variant bit 16 on a slab handle): bitwise? faster?
set -u
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "slab_cycle_depth" > gpurun_out/r06s_pytest_slab_deep2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06s_pytest_slab_deep2.log
P=gpurun_out/r06s_slab_proxy_deep2.txt
: > $P
for rep in 1 2; do
  timeout 500 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants 53601,119137 --transports peer,rccl --reps 4 2>&1 | grep grid | cut -c1-150 >> $P
done
tail -3 gpurun_out/r06s_pytest_slab_deep2.log
cat $P
