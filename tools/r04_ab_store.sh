#!/bin/bash
# A/B on one box: library with the nine store addresses as 64-bit per-lane values (A, liblbhip_va.so) against
# scalar base + 32-bit lane offset (B, the product build)
out=gpurun_out/r04_ab_store.txt
: > $out
A=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_va.so
for i in 1 2 3; do
  for bc in periodic pipe; do
    echo "A" >> $out; LB_LIB=$A python tools/variant_time.py 8192 $bc 353,4449 --reps 1 >> $out 2>&1
    echo "B" >> $out; python tools/variant_time.py 8192 $bc 353,4449 --reps 1 >> $out 2>&1
  done
  echo "A" >> $out; LB_LIB=$A python tools/variant_time.py 4096 periodic 353,4449 --reps 1 >> $out 2>&1
  echo "B" >> $out; python tools/variant_time.py 4096 periodic 353,4449 --reps 1 >> $out 2>&1
done
cat $out
