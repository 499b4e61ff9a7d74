#!/bin/bash
# round 6, GPU run 28: timelines of one slab of eight inside bench.py's process structure: hardware queues 4 | 8, RCCL channels uncapped | 8
set -u
cd $GRAFT_REPO_ROOT
for hq in default 8; do
for ch in default 8; do
  if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
  if [ $hq = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$hq; fi
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_x -- python3 $GRAFT_REPO_ROOT/tools/slab_proxy.py --torch-dist --parts 8 --steps 56 --variants -1 --transports rccl --reps 1 > $GRAFT_REPO_ROOT/gpurun_out/tl_x.log 2>&1)
  python3 tools/timeline.py gpurun_out/tl_x 30 > gpurun_out/r06f_slab_timeline_rccl_8_hq${hq}_ch${ch}.txt 2>&1
  rm -rf gpurun_out/tl_x
  echo "== hq $hq ch $ch"; cut -c1-150 gpurun_out/r06f_slab_timeline_rccl_8_hq${hq}_ch${ch}.txt
done
done
