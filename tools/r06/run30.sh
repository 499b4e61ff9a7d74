#!/bin/bash
# round 6, GPU run 30: the communication stream at the highest priority (LB_COMM_PRIO=1, a hardware-queue pool of its own): bench.py on the
# slab of 8 | 4 | 2 | 1 ranks in 280-step blocks, both transports, against the default; its timeline at 1024 rows; the slab tests with it
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06h_bench_comm_prio.txt
: > $P
for rep in 1 2 3; do
for prio in 0 1; do
  export LB_COMM_PRIO=$prio
  for rows in 1024 2048 4096 8192; do
    for t in rccl peer; do
      timeout 200 python3 bench.py --force-slab-path --slab-rows $rows --transport $t --steps 280 --warmup 28 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
      python3 - $prio $rows $t >> $P <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/x.json").read().strip().splitlines()[-1])
    print("LB_COMM_PRIO=%s rows %5s %-4s: %9.1f MLUPS, exchange %.1f us mean" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"], 1e3 * d["slabs"]["per_rank"][0]["exchange_ms_mean"]))
except Exception as e:
    print("LB_COMM_PRIO=%s rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], sys.argv[3], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
done
done
export LB_COMM_PRIO=1
for t in rccl peer; do
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --slab-rows 1024 --transport $t --steps 56 --warmup 14 --no-cpu-baseline --no-other-configs --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 30 k_halo_ > gpurun_out/r06h_bench_timeline_${t}_1024_prio.txt 2>&1
rm -rf gpurun_out/tl_bench
done
timeout 1200 python3 -m pytest tests -m gpu -q -k "slab or rccl or peer or distributed or random or halo" > gpurun_out/r06_run30_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run30_pytest.log
rm -f gpurun_out/x.json gpurun_out/x.err
sort $P | uniq | cat
cut -c1-150 gpurun_out/r06h_bench_timeline_rccl_1024_prio.txt
tail -3 gpurun_out/r06_run30_pytest.log
