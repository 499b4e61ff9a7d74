#!/bin/bash
# round 6, GPU run 14: where the waves of the occupancy probe's one-wave-per-SIMD build sat (k_deep<4> behind the six-step launcher, four
# waves per CU) -- per-wave records of one launch (LB_DIAG bit 12), against the product's k_deep<6>
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_occ2_placement.txt
echo "== k_deep<4> behind the six-step launcher, __launch_bounds__(128, 1), four waves per CU (liblbhip_p4a.so)" > $out
LB_TIMELINE_LIB=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_p4a.so LB_TIMELINE_DEPTH=6 timeout 200 python3 tools/wave_timeline.py 8192 4 2>&1 | grep -E "SIMD|wave slot|launch span|residency" >> $out
echo "== k_deep<6> of the diagnostic build, four waves per CU" >> $out
LB_TIMELINE_DEPTH=6 timeout 200 python3 tools/wave_timeline.py 8192 4 2>&1 | grep -E "SIMD|wave slot|launch span|residency" >> $out
cat $out
