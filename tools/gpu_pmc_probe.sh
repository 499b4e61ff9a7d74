#!/bin/bash
# Extra PMC passes for kernel diagnosis (one counter group per pass).  Usage: tools/gpu_pmc_probe.sh <tag> <bench args>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 10 --warmup 4 --no-cpu-baseline $*"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- $BENCH > $OUT/g$i.log 2>&1
done
python3 - <<PY
import csv,glob,statistics as st
for d in sorted(glob.glob("$OUT/g*/")):
    for f in glob.glob(d+"**/*_counter_collection.csv", recursive=True):
        acc={}
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0]
            if "k_step" in k: acc.setdefault((k,r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
        for (k,c),v in sorted(acc.items()): print("%-60s %-26s n=%3d mean=%.6g"%(k,c,len(v),st.mean(v)))
PY
