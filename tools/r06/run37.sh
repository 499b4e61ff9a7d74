#!/bin/bash
# round 6, GPU run 37: the reference's case (3751 x 1251 pipe with a disc) over the kernel families, 600 steps each, best of 3
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06l_reference_case_variants.txt
: > $P
for v in -1 53601 20833 4449 353 609 119137; do
  timeout 120 python3 tools/run_case.py --bc pipe --cyl --n 3751 --ny 1251 --steps 840 --repeat 3 --variant $v >> $P 2>&1
done
timeout 200 python3 tools/reference_grid_bench.py >> $P 2>&1
cat $P
