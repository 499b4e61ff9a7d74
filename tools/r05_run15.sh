#!/bin/bash
# round 5, GPU run 15: size sweep k_step5 / k_deep<6> / k_deep<7> / tiles, per family, with and without masks -> thresholds of effective_variant
set -u
cd $GRAFT_REPO_ROOT
{
for n in 1536 2048 2560 3072 4096 6144 8192; do for bc in periodic pipe cavity; do for m in "" "--mask"; do
  for v in 512 4449 20833 53601; do
    [ $v = 512 ] && [ $n -gt 2560 ] && continue
    echo -n "$bc $n $m variant $v: "; python3 tools/run_case.py --bc $bc $m --n $n --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
  done
done; done; done
} > gpurun_out/r05_size_sweep.txt 2>&1
