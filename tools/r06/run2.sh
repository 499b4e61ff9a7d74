#!/bin/bash
# round 6, GPU run 2: thick edge bands (band_extra), the edge stream at normal priority: full GPU suite, slab proxy A/B, slab stress under contention
set -u
cd $GRAFT_REPO_ROOT
timeout 1100 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run2_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run2_pytest.log
P=gpurun_out/r06_slab_proxy_bands.txt
: > $P
for rep in 1 2; do
  echo "== bands as in round 5 (LB_BAND_EXTRA=0)" >> $P
  LB_BAND_EXTRA=0 timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer >> $P 2>&1
  echo "== thick bands (default slack 8)" >> $P
  timeout 300 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer >> $P 2>&1
done
for sl in 2 5 12 16; do
  echo "== thick bands, LB_BAND_SLACK=$sl" >> $P
  LB_BAND_SLACK=$sl timeout 200 python3 tools/slab_proxy.py --parts 8,4 --steps 140 --variants -1 --transports rccl >> $P 2>&1
done
timeout 420 python3 tools/slab_stress.py 80 3 > gpurun_out/r06_slab_stress_normal_prio.txt 2>&1
echo "stress rc=$?" >> gpurun_out/r06_slab_stress_normal_prio.txt
tail -6 gpurun_out/r06_run2_pytest.log
cat $P
tail -5 gpurun_out/r06_slab_stress_normal_prio.txt
