#!/bin/bash
# round 6, GPU run 54: the last library: full GPU suite, smoke, the driver's bench command and the default one, bench over the slab path
set -u
cd $GRAFT_REPO_ROOT
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r06w_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06w_pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06w_smoke.txt 2>&1
echo "smoke rc=$?" >> gpurun_out/r06w_smoke.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06w_bench_steps20.json 2> gpurun_out/r06w_bench_steps20.err
echo "bench rc=$?" >> gpurun_out/r06w_smoke.txt
timeout 600 python3 bench.py > gpurun_out/r06w_bench_default.json 2> gpurun_out/r06w_bench_default.err
echo "bench default rc=$?" >> gpurun_out/r06w_smoke.txt
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06w_bench_slabpath_$t.json 2> gpurun_out/r06w_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06w_smoke.txt
done
timeout 200 python3 bench.py --variant 119137 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/r06w_bench_forced_deep2.json 2> gpurun_out/r06w_bench_forced_deep2.err
tail -3 gpurun_out/r06w_pytest_gpu.log; cat gpurun_out/r06w_smoke.txt | tail -6
python3 - <<'PY'
import json
for f in ("r06w_bench_steps20", "r06w_bench_default", "r06w_bench_forced_deep2"):
    d=json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, d["value"], r["frac"], r.get("frac_plain_launch"), r["launch_ms"], r.get("block_plan"), r["kernel"][:12], [(o["config"], o.get("value"), (o.get("kernel") or "")[:9], o.get("error")) for o in d.get("other_configs", [])], {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "kind", "cores")})
for t in ("rccl","peer"):
    e=json.loads(open("gpurun_out/r06w_bench_slabpath_%s.json"%t).read().strip().splitlines()[-1])
    print(t, e["value"], e["slabs"]["per_rank"], {k: e["slabs"]["cycle_tuning"][k] for k in ("depth","exchange_inline")})
PY
