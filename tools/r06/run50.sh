#!/bin/bash
# round 6, GPU run 50: k_deep2<7> in the slab cycle (lb_set_slab_cycle(8); the default under RCCL): full GPU suite (new: one slab of eight
# as a ring of its own at full size, both transports; random self-rings and rank processes with the deep variants), four rank processes
# over the peer transport, the proxy on the automatic variant, bench.py on the slab of 8 | 4 | 2 | 1 ranks with the tuner choosing
set -u
cd $GRAFT_REPO_ROOT
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r06t_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06t_pytest_gpu.log
timeout 900 python3 tools/peer_ranks_check.py --ranks 4 > gpurun_out/r06t_peer_ranks4.txt 2>&1
echo "peer ranks rc=$?" >> gpurun_out/r06t_pytest_gpu.log
P=gpurun_out/r06t_slab_proxy_final.txt
: > $P
for rep in 1 2; do
  timeout 500 python3 tools/slab_proxy.py --parts 8,4,2,1 --steps 140 --variants -1 --transports rccl,peer --reps 5 2>&1 | grep grid | cut -c1-260 >> $P
done
Q=gpurun_out/r06t_bench_placement.txt
: > $Q
for rep in 1 2; do
  for rows in 1024 2048 4096 8192; do
    for t in rccl peer; do
      timeout 200 python3 bench.py --force-slab-path --slab-rows $rows --transport $t --steps 280 --warmup 28 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
      python3 - $rows $t >> $Q <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/x.json").read().strip().splitlines()[-1])
    c = d["slabs"]["cycle_tuning"]
    print("rows %5s %-4s: %9.1f MLUPS, exchange %.1f us mean; tuner: depth %d, exchange %s; us per step beside %s | between %s" % (
        sys.argv[1], sys.argv[2], d["value"], 1e3 * d["slabs"]["per_rank"][0]["exchange_ms_mean"], c["depth"],
        "BETWEEN the launches" if c["exchange_inline"] else "beside them",
        {k: round(1e3 * v, 2) for k, v in c["ms_per_step"].items()}, {k: round(1e3 * v, 2) for k, v in c["ms_per_step_inline"].items()}))
except Exception as e:
    print("rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
done
rm -f gpurun_out/x.json gpurun_out/x.err
tail -4 gpurun_out/r06t_pytest_gpu.log; tail -2 gpurun_out/r06t_peer_ranks4.txt
cut -c1-150 $P
cat $Q
