#!/bin/bash
# round 5, GPU run 53: k_step5 / k_deep<6> / k_deep<7> in the walled families around their thresholds, after the wall strips' cost went to 2.1
set -u
cd $GRAFT_REPO_ROOT
{
for n in 1280 1536 1792 2048 2304; do for bc in pipe cavity; do for m in "" "--mask"; do
  for v in 4449 20833 53601; do
    echo -n "$bc $n $m variant $v: "; python3 tools/run_case.py --bc $bc $m --n $n --steps 84 --repeat 3 --variant $v | sed 's/.*\]: //'
  done
done; done; done
} > gpurun_out/r05_size_sweep4.txt 2>&1
exit 0
