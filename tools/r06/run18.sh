#!/bin/bash
# round 6, GPU run 18: last sanity pass on the final library -- smoke(), bench.py through the slab path (both transports), the driver's command
set -u
cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r06_final_smoke.txt
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06b_bench_slabpath_$t.json 2> gpurun_out/r06b_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06_final_smoke.txt
done
timeout 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06b_bench_steps20.json 2> gpurun_out/r06b_bench_steps20.err
echo "bench steps20 rc=$?" >> gpurun_out/r06_final_smoke.txt
cat gpurun_out/r06_final_smoke.txt
python3 - <<'PY'
import json
for f in ("r06b_bench_slabpath_rccl","r06b_bench_slabpath_peer","r06b_bench_steps20"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"].get("frac_plain_launch"), (d.get("slabs") or {}).get("per_rank"), [ (o.get("config"), o.get("value")) for o in d.get("other_configs",[])])
    except Exception as e:
        print(f, "no line", e)
PY
