#!/bin/bash
# SQ counter groups for one tools/run_case.py configuration.  Usage: tools/gpu_pmc_case.sh <tag> <run_case args>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/run_case.py $*"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_I8 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_EXP_GDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- $CMD > $OUT/g$i.log 2>&1
done
python3 - <<PY
import csv,glob,statistics as st
for d in sorted(glob.glob("$OUT/g*/")):
    for f in glob.glob(d+"**/*_counter_collection.csv", recursive=True):
        acc={}
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0]
            if "k_step" in k or "k_deep" in k or "k_tile" in k or "k1_" in k: acc.setdefault((k,r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
        for (k,c),v in sorted(acc.items()):
            if len(v) >= 3: print("$TAG %-44s %-22s n=%3d mean=%.6g"%(k,c,len(v),st.mean(v)))
PY
