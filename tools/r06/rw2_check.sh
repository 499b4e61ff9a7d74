#!/bin/bash
# round 6: round 5's unexplained miscompare (VERDICT r5 weak #1b(i)): k_deep<7> with TWO register windows and the row in flight left to
# the compiler (LB_DEEP_RW=2, LB_DEEP_MANUAL=0: 256 VGPR + 70-146 AGPR, and 32-48 B of scratch per lane in the pipe / cavity + mask
# kernels) against the single-step kernel, bit for bit (tools/step5_check.py --seven), built twice: spills to accumulation registers
# (the compiler's default) and to scratch only (-mllvm -amdgpu-spill-vgpr-to-agpr=0).
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/r06_rw2_check.txt
: > $out
for v in rw2c rw2s; do
  echo "== liblbhip_$v.so" >> $out
  LB_LIB=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_$v.so timeout 300 python3 tools/step5_check.py --seven --no-time >> $out 2>&1
  echo "rc=$?" >> $out
done
cat $out
