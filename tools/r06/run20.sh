#!/bin/bash
# round 6, GPU run 20: split bands for both transports (the default now): full GPU suite, slab proxy both transports, bench over the slab path
set -u
cd $GRAFT_REPO_ROOT
timeout 1300 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run20_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run20_pytest.log
P=gpurun_out/r06c_slab_proxy_final.txt
: > $P
for rep in 1 2; do
  timeout 500 python3 tools/slab_proxy.py --parts 1,2,4,8 --steps 140 --variants -1 --transports rccl,peer --reps 5 2>&1 | grep grid >> $P
done
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06c_bench_slabpath_$t.json 2> gpurun_out/r06c_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06_run20_pytest.log
done
timeout 600 python3 tools/peer_ranks_check.py --ranks 4 > gpurun_out/r06c_peer_ranks4.txt 2>&1
echo "peer ranks rc=$?" >> gpurun_out/r06_run20_pytest.log
tail -5 gpurun_out/r06_run20_pytest.log
cut -c1-150 $P
tail -5 gpurun_out/r06c_peer_ranks4.txt
python3 - <<'PY'
import json
for t in ("rccl","peer"):
    try:
        d=json.loads(open("gpurun_out/r06c_bench_slabpath_%s.json"%t).read().strip().splitlines()[-1])
        print(t, d["value"], d.get("slabs",{}).get("per_rank"))
    except Exception as e:
        print(t, "no line:", e)
PY
