#!/bin/bash
# round 6, GPU run 42: profiles of configurations 4 and 5 on the final library (tuner pinned: the summary names the kernel the line names)
set -u
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r06nc4 gpurun_out/prof_r06nc5
timeout 500 bash tools/gpu_profile.sh r06nc4 > gpurun_out/r06n_profile_c4.log 2>&1
timeout 500 bash tools/gpu_profile.sh r06nc5 --config 5 > gpurun_out/r06n_profile_c5.log 2>&1
cat gpurun_out/prof_r06nc5/tune_cache.txt
python3 - <<'PY'
import json
for t in ("r06nc4", "r06nc5"):
    d=json.loads(open("gpurun_out/prof_%s/unprofiled.json" % t).read().strip().splitlines()[-1])
    print(t, d["value"], d["roofline"]["kernel"][:70], d["roofline"]["launch_ms"])
PY
du -sh gpurun_out/prof_r06nc4 gpurun_out/prof_r06nc5
