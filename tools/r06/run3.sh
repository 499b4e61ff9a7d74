#!/bin/bash
# round 6, GPU run 3: k_deep's mask-free march per workgroup (A/B through LB_MASK_CLEAN_PATH), ABI 9 (slab cycle tuning, exchange timing):
# full GPU suite, the masked cases with and without the clean path, bench.py through the slab path on one GPU (both transports)
set -u
cd $GRAFT_REPO_ROOT
timeout 1100 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run3_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run3_pytest.log
L=2d-lb_amd/LB_D2Q9/liblbhip.so
AB_ENV_A="LB_MASK_CLEAN_PATH=0" ROUNDS=2 timeout 600 bash tools/gpu_ab.sh gpurun_out/r06_mask_clean_ab.txt $L $L \
  "--bc pipe --cyl --n 3751 --ny 1251 --steps 140" "--bc pipe --cyl --n 4096 --steps 84" "--bc pipe --tiff --n 4096 --steps 84" \
  "--bc periodic --mask --n 8192 --steps 84" "--bc cavity --cyl --n 6144 --steps 84" > /dev/null 2>&1
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06_bench_slabpath_$t.json 2> gpurun_out/r06_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06_run3_pytest.log
done
tail -8 gpurun_out/r06_run3_pytest.log
cat gpurun_out/r06_mask_clean_ab.txt.sorted
python3 - <<'PY'
import json
for t in ("rccl","peer"):
    try:
        d=json.loads(open("gpurun_out/r06_bench_slabpath_%s.json"%t).read().strip().splitlines()[-1])
        print(t, d["value"], d["roofline"]["kernel"][:40], d.get("slabs"))
    except Exception as e:
        print(t, "no line:", e); print(open("gpurun_out/r06_bench_slabpath_%s.err"%t).read()[-1500:])
PY
