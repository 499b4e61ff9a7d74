#!/bin/bash
# round 6, GPU run 25: a slab handle's three streams probed onto three hardware queues (separate_queues); RCCL's channel count through
# the environment.  bench.py over the slab path (both transports; RCCL uncapped | 8 channels), its timeline around an exchange, the
# proxy at 8 | 4 | 2 slabs with RCCL uncapped | 4 | 8 | 16 channels, the slab tests
set -u
cd $GRAFT_REPO_ROOT
export LB_QUEUE_PROBE=2
for ch in default 8; do
  for t in rccl peer; do
    if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
    timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06e_bench_slabpath_${t}_ch$ch.json 2> gpurun_out/r06e_bench_slabpath_${t}_ch$ch.err
  done
done
export NCCL_MAX_NCHANNELS=8
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 30 GenericKernel > gpurun_out/r06e_bench_timeline_rccl.txt 2>&1
rm -rf gpurun_out/tl_bench
unset NCCL_MAX_NCHANNELS
P=gpurun_out/r06e_slab_proxy_channels.txt
: > $P
for rep in 1 2; do
for ch in default 4 8 16; do
  echo "== NCCL_MAX_NCHANNELS=$ch" >> $P
  if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
  timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl --reps 4 2>&1 | grep "grid\|lb_create" >> $P
done
done
unset NCCL_MAX_NCHANNELS
timeout 1200 python3 -m pytest tests -m gpu -q -k "slab or rccl or peer or distributed or random or halo" > gpurun_out/r06_run25_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run25_pytest.log
for f in gpurun_out/r06e_bench_slabpath_*.json; do
  python3 - $f <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f[31:-5], d["value"], d["slabs"]["per_rank"][0]["exchange_ms_mean"])
except Exception as e:
    print(f, "no line", e)
PY
  grep -h lb_create ${f%.json}.err | sort | uniq -c
done
cut -c1-160 gpurun_out/r06e_bench_timeline_rccl.txt
cut -c1-150,230-330 $P
tail -3 gpurun_out/r06_run25_pytest.log
