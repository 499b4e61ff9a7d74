#!/bin/bash
# round 6, GPU run 35: the library of commit "ABI 10": full GPU suite, smoke, the driver's bench command, bench over the slab path (both
# transports), rocprofv3 kernel-trace summary of the driver's command
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06k_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06k_pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06k_smoke.txt 2>&1
echo "smoke rc=$?" >> gpurun_out/r06k_smoke.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06k_bench_steps20.json 2> gpurun_out/r06k_bench_steps20.err
echo "bench rc=$?" >> gpurun_out/r06k_smoke.txt
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06k_bench_slabpath_$t.json 2> gpurun_out/r06k_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06k_smoke.txt
done
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06k_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/r06k_prof.log 2>&1)
f=$(find gpurun_out/r06k_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" > gpurun_out/r06k_rocprof_kernel_stats.csv
rm -rf gpurun_out/r06k_prof
tail -3 gpurun_out/r06k_pytest_gpu.log; cat gpurun_out/r06k_smoke.txt | tail -5
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r06k_bench_steps20.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_plain_launch"), d["roofline"]["kernel"][:60])
print({k: v for k, v in d.items() if k in ("cpu_baseline",)})
print([ (o["config"], o["value"]) for o in d.get("other_configs", [])])
for t in ("rccl","peer"):
    e=json.loads(open("gpurun_out/r06k_bench_slabpath_%s.json"%t).read().strip().splitlines()[-1])
    print(t, e["value"], e["slabs"]["per_rank"], {k: e["slabs"]["cycle_tuning"][k] for k in ("depth","exchange_inline")})
PY
cat gpurun_out/r06k_rocprof_kernel_stats.csv | cut -c1-200
