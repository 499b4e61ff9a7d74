#!/bin/bash
# round 6, GPU run 22: which RCCL the bench and the proxy load and how many workgroups its send / receive kernel takes; proxy at 4 and 2
# slabs with the channel count capped
set -u
cd $GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 400 | grep -v "k_deep" | tail -12 > gpurun_out/r06c_bench_rccl_kernels.txt 2>&1
rm -rf gpurun_out/tl_bench
python3 - > gpurun_out/r06c_rccl_libs.txt 2>&1 <<'PY'
import os, sys
sys.path[:0] = ["2d-lb_amd", "."]
import torch
print("after import torch:", [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l][:1])
from LB_D2Q9.simulation import Simulation, comm_unique_id
s = Simulation(512, 512, 1.7, bc="periodic", halo=True)
s.comm_init(comm_unique_id(), 0, 1)
print("after lb_comm_init:", sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l)))
PY
P=gpurun_out/r06c_slab_proxy_channels.txt
: > $P
for ch in default 2 4 8 16; do
  echo "== NCCL_MAX_NCHANNELS=$ch" >> $P
  if [ $ch = default ]; then
    timeout 300 python3 tools/slab_proxy.py --parts 4,2 --steps 140 --variants -1 --transports rccl --reps 4 2>&1 | grep grid >> $P
  else
    NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=1 timeout 300 python3 tools/slab_proxy.py --parts 4,2 --steps 140 --variants -1 --transports rccl --reps 4 2>&1 | grep grid >> $P
  fi
done
cat gpurun_out/r06c_bench_rccl_kernels.txt | cut -c1-170
cat gpurun_out/r06c_rccl_libs.txt
cut -c1-150,230-330 $P
