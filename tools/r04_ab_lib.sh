#!/bin/bash
# A/B on one box: 2d-lb_amd/LB_D2Q9/liblbhip_va.so (A) against the product build (B), default kernels, alternating
# usage: tools/r04_ab_lib.sh "<size> <bc>" ...
out=gpurun_out/r04_ab_lib.txt
: > $out
A=$PWD/2d-lb_amd/LB_D2Q9/liblbhip_va.so
for i in 1 2 3; do
  for c in "$@"; do
    echo "A" >> $out; LB_LIB=$A python tools/variant_time.py $c 4449 --reps 1 >> $out 2>&1
    echo "B" >> $out; python tools/variant_time.py $c 4449 --reps 1 >> $out 2>&1
  done
done
paste - - < $out
