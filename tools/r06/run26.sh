#!/bin/bash
# round 6, GPU run 26: bench.py over the slab path with more hardware queues per process (GPU_MAX_HW_QUEUES): does the communication
# stream get a queue of its own?
set -u
cd $GRAFT_REPO_ROOT
export LB_QUEUE_PROBE=2
for hq in 8 16; do
  export GPU_MAX_HW_QUEUES=$hq
  for ch in default 8; do
    if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
    for t in rccl peer; do
      timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06f_bench_slabpath_${t}_ch${ch}_hq$hq.json 2> gpurun_out/r06f_bench_slabpath_${t}_ch${ch}_hq$hq.err
    done
  done
done
export GPU_MAX_HW_QUEUES=8
export NCCL_MAX_NCHANNELS=8
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 30 GenericKernel > gpurun_out/r06f_bench_timeline_rccl_hq8.txt 2>&1
rm -rf gpurun_out/tl_bench
for f in gpurun_out/r06f_bench_slabpath_*.json; do
  python3 - $f <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f[31:-5], d["value"], d["slabs"]["per_rank"][0]["exchange_ms_mean"])
except Exception as e:
    print(f, "no line", e)
PY
done
cut -c1-160 gpurun_out/r06f_bench_timeline_rccl_hq8.txt
