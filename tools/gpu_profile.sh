#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + PMC passes of the bench workload.
# Usage: tools/gpu_profile.sh <tag> [bench args...]     outputs under gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --calibrate 5 --min-timed-s 0.2 $*"
# The tuner's choice is pinned: an un-profiled run tunes and writes LB_TUNE_CACHE, the profiled runs read it (lb_autotune takes a cached
# result over) -- the summaries then hold the kernel the un-profiled line names, not whatever wins under the profiler's overhead.
export LB_TUNE_CACHE=$OUT/tune_cache.txt
$BENCH > $OUT/unprofiled.json 2> $OUT/unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
