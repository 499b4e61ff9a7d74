#!/bin/bash
# where k_step5 takes over from the LDS tiles (k_tile4) and from k_step4: whole grids, one GPU
out=gpurun_out/r04_step5_sweep.txt
: > $out
for n in 1024 1280 1536 1792 2048 2560; do
  for bc in periodic cavity; do
    python tools/variant_time.py $n $bc 512,353,4449 --reps 1 >> $out 2>&1
  done
done
python tools/variant_time.py 3751x1251 pipe 353,4449 --reps 2 >> $out 2>&1
python tools/variant_time.py 3751x1251 pipe 353,4449 --reps 1 --mask >> $out 2>&1
python tools/variant_time.py 16384 periodic 353,4449 --reps 1 >> $out 2>&1
cat $out
