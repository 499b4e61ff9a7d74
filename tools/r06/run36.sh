#!/bin/bash
# round 6, GPU run 36: the driver's bench command again (the other configurations had failed on an undefined name), and the default run
set -u
cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06k_bench_steps20.json 2> gpurun_out/r06k_bench_steps20.err
echo "rc=$?"
timeout 600 python3 bench.py > gpurun_out/r06k_bench_default.json 2> gpurun_out/r06k_bench_default.err
echo "rc=$?"
python3 - <<'PY'
import json
for f in ("gpurun_out/r06k_bench_steps20.json", "gpurun_out/r06k_bench_default.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["steps"], d["roofline"]["frac"], d["roofline"].get("frac_plain_launch"), d["roofline"]["launch_ms"])
    print([(o["config"], o.get("path"), o.get("value"), o.get("error")) for o in d.get("other_configs", [])])
PY
