#!/bin/bash
# round 5, GPU run 29: SQ counters of k_deep<7> with the row in flight waited for by hand (compare profiles/r05_sq_deep7_8192.txt)
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_r05sq2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $REPO/bench.py --steps 14 --warmup 7 --no-cpu-baseline > $OUT/g$i.log 2>&1
done
python3 - > $REPO/gpurun_out/r05_sq_deep7_manual_wait.txt <<PY
import csv,glob,statistics as st
for d in sorted(glob.glob("$OUT/g*/")):
    for f in glob.glob(d+"**/*_counter_collection.csv", recursive=True):
        acc={}
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0]
            if "k_deep" in k: acc.setdefault((k,r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
        for (k,c),v in sorted(acc.items()): print("%-50s %-26s n=%3d mean=%.6g"%(k,c,len(v),st.mean(v)))
PY
tail -3 $OUT/g1.log >> $REPO/gpurun_out/r05_sq_deep7_manual_wait.txt
