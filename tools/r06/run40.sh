#!/bin/bash
# round 6, GPU run 40: non-temporal stores from 320 MB per lattice pair: the driver's bench command (reference case behind it), the
# reference-grid tool, twice
set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06m_bench_steps20_$rep.json 2> gpurun_out/r06m_bench_steps20_$rep.err
  timeout 200 python3 tools/reference_grid_bench.py > gpurun_out/r06m_reference_grid_$rep.txt 2>&1
done
python3 - <<'PY'
import json
for r in (1, 2):
    d=json.loads(open("gpurun_out/r06m_bench_steps20_%d.json" % r).read().strip().splitlines()[-1])
    print(d["value"], d["roofline"]["frac"], [(o["config"], o.get("path"), o.get("value"), o.get("steps_per_launch")) for o in d.get("other_configs", [])])
PY
cat gpurun_out/r06m_reference_grid_1.txt gpurun_out/r06m_reference_grid_2.txt
