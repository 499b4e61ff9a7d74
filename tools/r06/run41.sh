#!/bin/bash
# round 6, GPU run 41: k_deep2<7> among lb_autotune's candidates in walled boxes / with a mask; non-temporal stores from 320 MB: full GPU
# suite, what the tuner picks across sizes and families, the driver's bench command, the reference-grid tool
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06n_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06n_pytest_gpu.log
timeout 600 python3 tools/tune_probe.py 2048 3072 4096 8192 2>&1 | cut -c1-150 > gpurun_out/r06n_tune_probe.txt
for rep in 1 2; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06n_bench_steps20_$rep.json 2> gpurun_out/r06n_bench_steps20_$rep.err
  timeout 200 python3 tools/reference_grid_bench.py > gpurun_out/r06n_reference_grid_$rep.txt 2>&1
done
tail -3 gpurun_out/r06n_pytest_gpu.log
cat gpurun_out/r06n_tune_probe.txt
python3 - <<'PY'
import json
for r in (1, 2):
    d=json.loads(open("gpurun_out/r06n_bench_steps20_%d.json" % r).read().strip().splitlines()[-1])
    print(d["value"], d["roofline"]["frac"], [(o["config"], o.get("path"), o.get("value"), (o.get("kernel") or "")[:12]) for o in d.get("other_configs", [])])
PY
cat gpurun_out/r06n_reference_grid_1.txt gpurun_out/r06n_reference_grid_2.txt | grep opencl
