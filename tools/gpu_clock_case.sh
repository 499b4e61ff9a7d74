#!/bin/bash
# Effective shader clock of the hot kernel of one tools/run_case.py configuration: GRBM_GUI_ACTIVE (sum over the 8 XCDs)
# / 8 / kernel duration, next to SQ_BUSY_CYCLES and SQ_WAVE_CYCLES.  Usage: tools/gpu_clock_case.sh <tag> <run_case args>
# (LB_LIB / LB_DIAG are taken from the environment: export them first.)
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/clk_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/run_case.py $*"
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/g1 -- $CMD > $OUT/g1.log 2>&1
python3 - <<PY
import csv,glob,statistics as st
cc=glob.glob("$OUT/g1/**/*_counter_collection.csv", recursive=True)
kt=glob.glob("$OUT/g1/**/*_kernel_trace.csv", recursive=True)
dur={}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
acc={}
for f in cc:
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0]
        if "k_step" not in k: continue
        acc.setdefault((k,r["Dispatch_Id"]),{})[r["Counter_Name"]]=float(r["Counter_Value"])
by={}
for (k,d),c in acc.items():
    if d in dur and "GRBM_GUI_ACTIVE" in c:
        by.setdefault(k,[]).append((dur[d], c["GRBM_GUI_ACTIVE"]/8/dur[d]/1e3, c.get("SQ_BUSY_CYCLES",0), c.get("SQ_WAVE_CYCLES",0), c.get("SQ_WAVES",0)))
for k,v in by.items():
    if len(v)>=3:
        print("$TAG %-40s n=%3d  dur %.1f us  clock %.3f GHz  SQ_BUSY_CYCLES %.4g  SQ_WAVE_CYCLES %.4g  SQ_WAVES %.0f" % (k,len(v),st.median(x[0] for x in v),st.median(x[1] for x in v),st.median(x[2] for x in v),st.median(x[3] for x in v),st.median(x[4] for x in v)))
PY
