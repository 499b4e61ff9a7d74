#!/bin/bash
# round 6, GPU run 27: the proxy inside bench.py's process structure (a torch.distributed group first): hardware queues 4 (HIP's default)
# | 8, RCCL's channels uncapped | 8, both transports, 8 | 4 | 2 slabs; two rounds
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06f_slab_proxy_queues.txt
: > $P
for rep in 1 2; do
for hq in default 8; do
for ch in default 8; do
  echo "== GPU_MAX_HW_QUEUES=$hq NCCL_MAX_NCHANNELS=$ch" >> $P
  if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
  if [ $hq = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$hq; fi
  timeout 300 python3 tools/slab_proxy.py --torch-dist --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer --reps 4 2>&1 | grep "grid" >> $P
done
done
done
cut -c1-150,230-330 $P
