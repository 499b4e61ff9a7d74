#!/bin/bash
# round 5, GPU run 33: size sweep again with the hand-waited gather (k_step5 4449 / k_deep<6> 20833 / k_deep<7> 53601): the thresholds of
# effective_variant; and config 5's own image at 4096^2 and the reference case's geometry
set -u
cd $GRAFT_REPO_ROOT
{
for n in 1536 2048 2560 3072 3584 4096 5120 6144; do for bc in periodic pipe cavity; do for m in "" "--mask"; do
  for v in 4449 20833 53601; do
    echo -n "$bc $n $m variant $v: "; python3 tools/run_case.py --bc $bc $m --n $n --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
  done
done; done; done
for v in 4449 20833 53601; do echo -n "pipe 4096 --tiff variant $v: "; python3 tools/run_case.py --bc pipe --tiff --n 4096 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'; done
for v in 4449 20833 53601; do echo -n "pipe --cyl variant $v: "; python3 tools/run_case.py --bc pipe --cyl --n 3751 --ny 1251 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'; done
} > gpurun_out/r05_size_sweep2.txt 2>&1
