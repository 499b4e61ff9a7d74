#!/bin/bash
# round 6, GPU run 4: the row in flight at a[0:42] (a wave takes ~280 registers, not a SIMD's whole file), split edge bands with the
# exchange on the communication stream: full GPU suite, slab proxy (split on / off, both transports), plain-grid A/B against round 5
set -u
cd $GRAFT_REPO_ROOT
timeout 1100 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run4_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run4_pytest.log
P=gpurun_out/r06_slab_proxy_split.txt
: > $P
for rep in 1 2; do
  echo "== split bands (default)" >> $P
  timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer 2>&1 | grep grid >> $P
  echo "== one launch per band (LB_SPLIT_BANDS=0)" >> $P
  LB_SPLIT_BANDS=0 timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer 2>&1 | grep grid >> $P
done
echo "== round-5 bands (LB_BAND_EXTRA=0), split schedule" >> $P
LB_BAND_EXTRA=0 timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer 2>&1 | grep grid >> $P
for sl in 0 6 10; do
  echo "== split bands, LB_BAND_SLACK=$sl" >> $P
  LB_BAND_SLACK=$sl timeout 200 python3 tools/slab_proxy.py --parts 8,4 --steps 140 --variants -1 --transports rccl,peer 2>&1 | grep grid >> $P
done
ROUNDS=2 timeout 400 bash tools/gpu_ab.sh gpurun_out/r06_agpr_low_ab.txt 2d-lb_amd/LB_D2Q9/liblbhip_r05.so 2d-lb_amd/LB_D2Q9/liblbhip.so \
  "--bc periodic --n 8192 --steps 84" "--bc periodic --n 4096 --steps 84" "--bc pipe --n 8192 --steps 84" "--bc pipe --tiff --n 4096 --steps 84" > /dev/null 2>&1
for t in rccl peer; do
  timeout 300 python3 bench.py --force-slab-path --transport $t --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06_bench_slabpath_$t.json 2> gpurun_out/r06_bench_slabpath_$t.err
  echo "bench $t rc=$?" >> gpurun_out/r06_run4_pytest.log
done
tail -6 gpurun_out/r06_run4_pytest.log
cat $P
cut -c1-60,190-260 gpurun_out/r06_agpr_low_ab.txt.sorted
python3 - <<'PY'
import json
for t in ("rccl","peer"):
    try:
        d=json.loads(open("gpurun_out/r06_bench_slabpath_%s.json"%t).read().strip().splitlines()[-1])
        print(t, d["value"], d["roofline"]["kernel"][:40], d.get("slabs",{}).get("per_rank"), d.get("slabs",{}).get("cycle_tuning"))
    except Exception as e:
        print(t, "no line:", e); print(open("gpurun_out/r06_bench_slabpath_%s.err"%t).read()[-1500:])
PY
