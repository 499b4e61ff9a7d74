#!/bin/bash
# round 5, GPU run 16: the obstacle swap behind a wave-uniform test: random 1 % mask (92 % of the wave rows hold a solid cell), a disc, the porous-medium image
set -u
cd $GRAFT_REPO_ROOT
{
python3 tools/step5_check.py --seven --no-time 2>&1 | grep -v "^checked"
for cfg in "pipe 4096" "pipe 8192" "periodic 8192"; do set -- $cfg
 for m in "" "--mask" "--cyl" "--tiff"; do for v in 4449 20833 53601; do
  echo -n "$1 $2 $m variant $v: "; python3 tools/run_case.py --bc $1 $m --n $2 --steps 84 --repeat 2 --variant $v | sed 's/.*\]: //'
 done; done
done
} > gpurun_out/r05_mask_skip.txt 2>&1
