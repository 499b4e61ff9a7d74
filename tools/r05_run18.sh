#!/bin/bash
# round 5, GPU run 18: shifted LDS window reads (unaligned ds_read_b128 through asm) in k_deep: bitwise, timing
set -u
cd $GRAFT_REPO_ROOT
{
python3 tools/step5_check.py --six --no-time 2>&1 | grep -v "^checked"
python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
for cfg in "periodic 8192" "periodic 4096" "pipe 8192" "cavity 8192"; do set -- $cfg
  for v in 20833 53601; do
    echo -n "$1 $2 variant $v: "; python3 tools/run_case.py --bc $1 --n $2 --steps 84 --repeat 3 --variant $v | sed 's/.*\]: //'
  done
done
for v in 4449 53601; do
  echo -n "periodic+mask 8192 variant $v: "; python3 tools/run_case.py --bc periodic --mask --n 8192 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
  echo -n "pipe+tiff 4096 variant $v: "; python3 tools/run_case.py --bc pipe --tiff --n 4096 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
done
} > gpurun_out/r05_shifted_reads.txt 2>&1
