#!/bin/bash
# round 6, GPU run 33: lb_set_exchange_inline (ABI 10) -- the slab tests (every depth x placement, both transports; random self-rings with
# odd seeds inline; four rank processes over the peer transport, inline), bench.py on the slab of 8 | 4 | 2 | 1 ranks with the collective
# tuner choosing depth AND placement (what did it choose?), proxy A/B of the placement
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q -k "slab or rccl or peer or distributed or random or halo or abi" > gpurun_out/r06_run33_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run33_pytest.log
timeout 600 python3 tools/peer_ranks_check.py --ranks 4 --inline > gpurun_out/r06i_peer_ranks4_inline.txt 2>&1
echo "peer ranks inline rc=$?" >> gpurun_out/r06_run33_pytest.log
P=gpurun_out/r06i_bench_placement.txt
: > $P
for rep in 1 2; do
  for rows in 1024 2048 4096 8192; do
    for t in rccl peer; do
      timeout 200 python3 bench.py --force-slab-path --slab-rows $rows --transport $t --steps 280 --warmup 28 --no-cpu-baseline --no-other-configs > gpurun_out/x.json 2> gpurun_out/x.err
      python3 - $rows $t >> $P <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/x.json").read().strip().splitlines()[-1])
    c = d["slabs"]["cycle_tuning"]
    print("rows %5s %-4s: %9.1f MLUPS, exchange %.1f us mean; tuner: depth %d, exchange %s; us per step beside %s | between %s" % (
        sys.argv[1], sys.argv[2], d["value"], 1e3 * d["slabs"]["per_rank"][0]["exchange_ms_mean"], c["depth"],
        "between the launches" if c["exchange_inline"] else "beside them",
        {k: round(1e3 * v, 2) for k, v in c["ms_per_step"].items()}, {k: round(1e3 * v, 2) for k, v in c["ms_per_step_inline"].items()}))
except Exception as e:
    print("rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
done
Q=gpurun_out/r06i_slab_proxy_placement.txt
: > $Q
for rep in 1 2; do
  for inl in "" "--inline"; do
    echo "== placement: ${inl:-beside}" >> $Q
    timeout 400 python3 tools/slab_proxy.py --torch-dist $inl --parts 8,4,2,1 --steps 140 --variants -1 --transports rccl,peer --reps 4 2>&1 | grep grid >> $Q
  done
done
rm -f gpurun_out/x.json gpurun_out/x.err
tail -4 gpurun_out/r06_run33_pytest.log; tail -2 gpurun_out/r06i_peer_ranks4_inline.txt
cat $P
cut -c1-150 $Q
