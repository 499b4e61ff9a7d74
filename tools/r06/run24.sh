#!/bin/bash
# round 6, GPU run 24: why bench.py's RCCL self-exchange takes 32 us and the proxy's 250: the bench's timeline around an exchange
set -u
cd $GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 36 ncclDevKernel > gpurun_out/r06c_bench_timeline_rccl.txt 2>&1
rm -rf gpurun_out/tl_bench
cut -c1-170 gpurun_out/r06c_bench_timeline_rccl.txt
