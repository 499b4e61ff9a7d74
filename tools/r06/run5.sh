#!/bin/bash
# round 6, GPU run 5: new tests (planar 8192^2 under k_deep, lb_set_slab_cycle / exchange timing), the two-waves-per-SIMD probe, round 5's
# RW = 2 miscompare, slab proxy with exchange times (split under RCCL, one launch per band under the peer transport)
set -u
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -m gpu -q -k "planar_layout_8192 or slab_cycle_depth or self_ring or peer_transport or slab_schedule" > gpurun_out/r06_run5_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run5_pytest.log
timeout 400 bash tools/r06/occ2_probe.sh > /dev/null 2>&1
timeout 700 bash tools/r06/rw2_check.sh > /dev/null 2>&1
P=gpurun_out/r06_slab_proxy_policy.txt
: > $P
for rep in 1 2; do
  echo "== default policy" >> $P
  timeout 400 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer --reps 5 2>&1 | grep grid >> $P
done
echo "== round-5 bands (LB_BAND_EXTRA=0)" >> $P
LB_BAND_EXTRA=0 timeout 400 python3 tools/slab_proxy.py --parts 8,4,2 --steps 140 --variants -1 --transports rccl,peer --reps 5 2>&1 | grep grid >> $P
echo "== RCCL, split, slack 12" >> $P
LB_BAND_SLACK=12 timeout 300 python3 tools/slab_proxy.py --parts 8,4 --steps 140 --variants -1 --transports rccl --reps 5 2>&1 | grep grid >> $P
tail -5 gpurun_out/r06_run5_pytest.log
cat gpurun_out/r06_occ2_probe.txt
cat gpurun_out/r06_rw2_check.txt | tail -40
cat $P
