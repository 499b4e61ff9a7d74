#!/bin/bash
# round 6, GPU run 38: non-temporal stores (variant bit 0) below the 450 MB lattice-pair threshold of round 3 (measured with k_step4 then):
# k_deep<7> / k_deep<6> / k_step5 with plain | non-temporal stores on grids of 1.5 M - 6 M cells; 840 steps, best of 3
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06l_nt_stores_midsize.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for v in 53616 53601 20848 20833 4464 4449; do
  run --bc pipe --cyl --n 3751 --ny 1251 --variant $v
done
for n in 1536 2048 2400; do
  for v in 53616 53601 20848 20833 4464 4449; do run --bc periodic --n $n --variant $v; done
done
for n in 2048 2400; do
  for v in 53616 53601 4464 4449; do run --bc pipe --n $n --variant $v; done
  for v in 53616 53601 4464 4449; do run --bc cavity --mask --n $n --variant $v; done
done
cat $P
