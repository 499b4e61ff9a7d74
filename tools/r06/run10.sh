#!/bin/bash
# round 6, GPU run 10: the lines and the profiles of the final library -- bench.py (default command and the driver's), rocprofv3 kernel
# trace + FETCH_SIZE / WRITE_SIZE passes of configurations 4, 5, 3, 2 with the tuner's choice pinned (tools/gpu_profile.sh)
set -u
cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py > gpurun_out/r06a_bench_default.json 2> gpurun_out/r06a_bench_default.err
echo "bench default rc=$?"
timeout 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06a_bench_steps20.json 2> gpurun_out/r06a_bench_steps20.err
echo "bench steps20 rc=$?"
timeout 500 bash tools/gpu_profile.sh r06c4 > gpurun_out/r06_profile_c4.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06c5 --config 5 > gpurun_out/r06_profile_c5.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06c3 --config 3 > gpurun_out/r06_profile_c3.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06c2 --config 2 > gpurun_out/r06_profile_c2.log 2>&1
python3 - <<'PY'
import json
for f in ("gpurun_out/r06a_bench_default.json","gpurun_out/r06a_bench_steps20.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, d["value"], "MLUPS; launch", r["launch_ms"], "frac", r["frac"], "plain", r.get("frac_plain_launch"), "plan", r.get("block_plan"), "six", (r.get("six_step_kernel") or {}).get("MLUPS"))
        for o in d.get("other_configs",[]): print("   ", o.get("config"), o.get("path",""), o.get("value"), o.get("roofline_frac"), str(o.get("kernel"))[:50], o.get("error",""))
        print("    cpu:", {k:v for k,v in (d.get("cpu_baseline") or {}).items() if k in ("value","cores","kind")})
    except Exception as e:
        print(f, "no line", e)
PY
ls gpurun_out/prof_r06c*/ | head -30
