#!/bin/bash
# round 6, GPU run 32: does rocprofv3's PC sampling (beta) work on this box?  What the agent offers; a host-trap sample of k_deep<7> 8192^2
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 60 rocprofv3 -L > $R/gpurun_out/r06_rocprof_avail.txt 2>&1
grep -i -B2 -A12 "pc.sampl" $R/gpurun_out/r06_rocprof_avail.txt | head -60
timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 1 --output-format csv -d $R/gpurun_out/pcs -- python3 $R/tools/run_case.py --bc periodic --n 8192 --steps 140 > $R/gpurun_out/pcs.log 2>&1
echo "rc=$?"
tail -5 $R/gpurun_out/pcs.log
find $R/gpurun_out/pcs -type f | head; for f in $(find $R/gpurun_out/pcs -name "*pc_sampling*.csv"); do wc -l $f; head -5 $f; done
