#!/bin/bash
# k_step5 tuning scan (one GPU): edge-strip cost of the walled families, priority turns, non-temporal stores, waves per CU
out=gpurun_out/r04_step5_scan.txt
: > $out
python tools/step5_check.py --no-time >> $out 2>&1
for ec in 1.0001 1.2 1.4 1.6 1.8 2.2; do
  LB_EDGE_COST=$ec python tools/variant_time.py 8192 pipe 353,4449 --reps 1 >> $out 2>&1
  LB_EDGE_COST=$ec python tools/variant_time.py 4096 cavity 4449 --reps 1 >> $out 2>&1
done
# periodic: NT stores off (4448), priority turns off (+2048), waves per CU
python tools/variant_time.py 8192 periodic 353,4449,4448,6497 --reps 2 >> $out 2>&1
python tools/variant_time.py 4096 periodic 353,4449,4448,6497 --reps 2 >> $out 2>&1
for w in 6 4; do
  LB_STEP2_WAVES_PER_CU=$w python tools/variant_time.py 8192 periodic 4449 --reps 1 >> $out 2>&1
done
python tools/variant_time.py 2048 periodic 353,4449,512 --reps 2 >> $out 2>&1
python tools/variant_time.py 3072 periodic 353,4449 --reps 2 >> $out 2>&1
python tools/variant_time.py 8192 periodic 353,4449 --reps 1 --mask >> $out 2>&1
python tools/variant_time.py 8192 pipe 353,4449 --reps 1 --mask >> $out 2>&1
cat $out
