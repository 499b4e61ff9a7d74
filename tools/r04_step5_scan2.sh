out=gpurun_out/r04_step5_scan2.txt
: > $out
python tools/step5_check.py --no-time >> $out 2>&1
for i in 1 2; do
python tools/variant_time.py 8192 periodic 353,4449,6497 --reps 1 >> $out 2>&1
LB_STEP5_PF=1 python tools/variant_time.py 8192 periodic 4449,6497 --reps 1 >> $out 2>&1
python tools/variant_time.py 4096 periodic 353,4449 --reps 1 >> $out 2>&1
LB_STEP5_PF=1 python tools/variant_time.py 4096 periodic 4449 --reps 1 >> $out 2>&1
done
python tools/variant_time.py 8192 pipe 353,4449 --reps 2 >> $out 2>&1
python tools/variant_time.py 8192 cavity 353,4449 --reps 1 >> $out 2>&1
cat $out
