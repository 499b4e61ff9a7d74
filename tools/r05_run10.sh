#!/bin/bash
# round 5, GPU run 10: families timed with the searched wall-strip split; then the GPU test suite
set -u
cd $GRAFT_REPO_ROOT
{
for bc in pipe cavity; do for v in 4449 20833 53601; do
  echo -n "$bc 8192 variant $v: "; python3 tools/run_case.py --bc $bc --n 8192 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
  echo -n "$bc 4096 variant $v: "; python3 tools/run_case.py --bc $bc --n 4096 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
done; done
for v in 4449 20833 53601; do
  echo -n "pipe+mask 4096 variant $v: "; python3 tools/run_case.py --bc pipe --mask --n 4096 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
  echo -n "periodic+mask 8192 variant $v: "; python3 tools/run_case.py --bc periodic --mask --n 8192 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
done
for n in 2048 2560 3072; do for v in 4449 20833 53601; do
  echo -n "periodic $n variant $v: "; python3 tools/run_case.py --bc periodic --n $n --steps 84 --repeat 3 --variant $v | sed 's/.*\]: //'
done; done
} > gpurun_out/r05_families_timing.txt 2>&1
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r05_gputest_b.txt
