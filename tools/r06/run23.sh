#!/bin/bash
# round 6, GPU run 23: the halo communicator capped at 8 channels through ncclConfig_t::maxCTAs (does this RCCL honour it?): the send /
# receive kernel's workgroups in the timeline, proxy capped | uncapped | other caps, bench over the slab path, the RCCL tests
set -u
cd $GRAFT_REPO_ROOT
for cap in 8 0; do
  (cd /tmp && export TMPDIR=/tmp && LB_RCCL_MAX_CTAS=$cap timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_cap$cap -- python3 $GRAFT_REPO_ROOT/tools/slab_proxy.py --parts 4 --steps 56 --variants -1 --transports rccl --reps 1 > $GRAFT_REPO_ROOT/gpurun_out/tl_cap$cap.log 2>&1)
  python3 tools/timeline.py gpurun_out/tl_cap$cap 44 > gpurun_out/r06c_slab_timeline_rccl_4_cap$cap.txt 2>&1
  rm -rf gpurun_out/tl_cap$cap
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline --min-blocks 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.log 2>&1)
python3 tools/timeline.py gpurun_out/tl_bench 4000 | grep -i "nccl" | tail -6 > gpurun_out/r06c_bench_rccl_kernels.txt 2>&1
rm -rf gpurun_out/tl_bench
P=gpurun_out/r06c_slab_proxy_maxctas.txt
: > $P
for rep in 1 2; do
for cap in 8 0 4 16; do
  echo "== LB_RCCL_MAX_CTAS=$cap" >> $P
  LB_RCCL_MAX_CTAS=$cap timeout 300 python3 tools/slab_proxy.py --parts 8,4,2,1 --steps 140 --variants -1 --transports rccl --reps 4 2>&1 | grep grid >> $P
done
done
timeout 900 python3 -m pytest tests -m gpu -q -k "rccl or slab or comm or distributed" > gpurun_out/r06_run23_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run23_pytest.log
timeout 300 python3 bench.py --force-slab-path --transport rccl --steps 56 --warmup 14 --no-cpu-baseline > gpurun_out/r06d_bench_slabpath_rccl.json 2> gpurun_out/r06d_bench_slabpath_rccl.err
grep -i nccl gpurun_out/r06c_slab_timeline_rccl_4_cap8.txt | tail -2 | cut -c1-150
grep -i nccl gpurun_out/r06c_slab_timeline_rccl_4_cap0.txt | tail -2 | cut -c1-150
cut -c1-150 gpurun_out/r06c_bench_rccl_kernels.txt
cut -c1-150,230-330 $P
tail -3 gpurun_out/r06_run23_pytest.log
python3 -c "
import json
d=json.loads(open('gpurun_out/r06d_bench_slabpath_rccl.json').read().strip().splitlines()[-1]); print(d['value'], d['slabs']['per_rank'])"
