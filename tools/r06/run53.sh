#!/bin/bash
# round 6, GPU run 53: k_deep2<7> among the tuner's candidates in every family: the driver's bench command three times (what does the
# headline pick?), the default command, pinned profiles of configuration 4 by whichever kernel the tuner picks AND by the other one
set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06v_bench_steps20_$rep.json 2> gpurun_out/r06v_bench_steps20_$rep.err
done
timeout 600 python3 bench.py > gpurun_out/r06v_bench_default.json 2> gpurun_out/r06v_bench_default.err
rm -rf gpurun_out/prof_r06vc4 gpurun_out/prof_r06vc4d2 gpurun_out/prof_r06vc4d1
timeout 500 bash tools/gpu_profile.sh r06vc4 > gpurun_out/r06v_profile_c4.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06vc4d2 --variant 119137 > gpurun_out/r06v_profile_c4d2.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06vc4d1 --variant 53601 > gpurun_out/r06v_profile_c4d1.log 2>&1
python3 - <<'PY'
import json
for f in ["gpurun_out/r06v_bench_steps20_%d.json" % r for r in (1, 2, 3)] + ["gpurun_out/r06v_bench_default.json", "gpurun_out/prof_r06vc4/unprofiled.json", "gpurun_out/prof_r06vc4d2/unprofiled.json", "gpurun_out/prof_r06vc4d1/unprofiled.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f[11:], d["value"], d["roofline"]["frac"], d["roofline"].get("frac_plain_launch"), d["roofline"]["kernel"][:12], [(o["config"], o.get("value"), (o.get("kernel") or "")[:9]) for o in d.get("other_configs", [])])
    except Exception as e:
        print(f, "no line", e)
PY
