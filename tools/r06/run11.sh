#!/bin/bash
# round 6, GPU run 11: full GPU suite on the final library (launchers report an unsupported family), SQ counters of k_deep<7> 8192^2,
# kernel timeline of the slab cycle (one of eight slabs, both transports)
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06_run11_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_run11_pytest.log
timeout 600 bash tools/gpu_pmc_case.sh r06deep7 --bc periodic --n 8192 --steps 140 > gpurun_out/r06_sq_deep7.txt 2>&1
for t in rccl peer; do
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_$t -- python3 $GRAFT_REPO_ROOT/tools/slab_proxy.py --parts 8 --steps 56 --variants -1 --transports $t --reps 1 > $GRAFT_REPO_ROOT/gpurun_out/tl_$t.log 2>&1)
  python3 tools/timeline.py gpurun_out/tl_$t 40 > gpurun_out/r06_slab_timeline_$t.txt 2>&1
done
tail -4 gpurun_out/r06_run11_pytest.log
grep "k_deep<1, false, false, 7" gpurun_out/r06_sq_deep7.txt | head -40
head -45 gpurun_out/r06_slab_timeline_rccl.txt
