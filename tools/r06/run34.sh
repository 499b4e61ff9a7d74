#!/bin/bash
# round 6, GPU run 34: the collective tuner with twenty cycles per candidate: what it chooses on the slab of 8 | 4 | 2 | 1 ranks, and
# the rate of the 280-step blocks behind it; three rounds
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06j_bench_placement.txt
: > $P
for rep in 1 2 3; do
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
        "BETWEEN the launches" if c["exchange_inline"] else "beside them",
        {k: round(1e3 * v, 2) for k, v in c["ms_per_step"].items()}, {k: round(1e3 * v, 2) for k, v in c["ms_per_step_inline"].items()}))
except Exception as e:
    print("rows %s %s: no line (%s)" % (sys.argv[1], sys.argv[2], e)); print(open("gpurun_out/x.err").read()[-800:])
PY
    done
  done
done
rm -f gpurun_out/x.json gpurun_out/x.err
sort $P
