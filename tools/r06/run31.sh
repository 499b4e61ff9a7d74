#!/bin/bash
# round 6, GPU run 31: bench.py's GPU_MAX_HW_QUEUES=8 default against HIP's four: the slab of 8 | 4 | 2 | 1 ranks in 280-step blocks, both
# transports, three rounds on one box; the plain single-GPU line both ways
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06h_bench_hw_queues.txt
: > $P
for rep in 1 2 3; do
for hq in 4 8; do
  export GPU_MAX_HW_QUEUES=$hq
  for rows in 1024 2048 4096 8192; do
    for t in rccl peer; do
      timeout 200 python3 bench.py --force-slab-path --slab-rows $rows --transport $t --steps 280 --warmup 28 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
      python3 - $hq $rows $t >> $P <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/x.json").read().strip().splitlines()[-1])
    print("GPU_MAX_HW_QUEUES=%s rows %5s %-4s: %9.1f MLUPS, exchange %.1f us mean" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"], 1e3 * d["slabs"]["per_rank"][0]["exchange_ms_mean"]))
except Exception as e:
    print("GPU_MAX_HW_QUEUES=%s rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], sys.argv[3], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
  timeout 200 python3 bench.py --steps 84 --warmup 14 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES=$hq plain single-GPU line: %9.1f MLUPS' % d['value'])" >> $P
done
done
unset GPU_MAX_HW_QUEUES
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --slab-rows 1024 --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --no-other-configs --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 30 k_halo_ > gpurun_out/r06h_bench_timeline_rccl_1024_hq8.txt 2>&1
rm -rf gpurun_out/tl_bench gpurun_out/x.json gpurun_out/x.err
sort $P | uniq -c | cat
cut -c1-150 gpurun_out/r06h_bench_timeline_rccl_1024_hq8.txt | tail -22
