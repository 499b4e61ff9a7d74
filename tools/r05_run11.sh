#!/bin/bash
# round 5, GPU run 11: walled families with the wall-column test out of the interior strips' code
set -u
cd $GRAFT_REPO_ROOT
{
python3 tools/step5_check.py --six --no-time 2>&1 | grep -v "^checked"
python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
for bc in pipe cavity; do for v in 4449 20833 53601; do
  echo -n "$bc 8192 variant $v: "; python3 tools/run_case.py --bc $bc --n 8192 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
  echo -n "$bc 4096 variant $v: "; python3 tools/run_case.py --bc $bc --n 4096 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
done; done
for v in 4449 20833 53601; do
  echo -n "pipe+mask 4096 variant $v: "; python3 tools/run_case.py --bc pipe --mask --n 4096 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
done
echo -n "periodic 8192 variant 53601: "; python3 tools/run_case.py --bc periodic --n 8192 --steps 84 --repeat 2 --variant 53601 | sed 's/.*\]: //'
} > gpurun_out/r05_families_timing2.txt 2>&1
