#!/bin/bash
# round 6, GPU run 21: kernel timelines of the slab cycle under RCCL at 4 and 2 slabs (where RCCL trails the peer transport)
set -u
cd $GRAFT_REPO_ROOT
for parts in 4 2; do
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_rccl_$parts -- python3 $GRAFT_REPO_ROOT/tools/slab_proxy.py --parts $parts --steps 56 --variants -1 --transports rccl --reps 1 > $GRAFT_REPO_ROOT/gpurun_out/tl_rccl_$parts.log 2>&1)
  python3 tools/timeline.py gpurun_out/tl_rccl_$parts 44 > gpurun_out/r06c_slab_timeline_rccl_$parts.txt 2>&1
  rm -rf gpurun_out/tl_rccl_$parts
done
cut -c1-170 gpurun_out/r06c_slab_timeline_rccl_4.txt
