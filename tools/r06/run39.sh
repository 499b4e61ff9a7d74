#!/bin/bash
# round 6, GPU run 39: the reference's case: what separates the automatic variant from the forced ones (bits 0 and 4)?
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06l_reference_case_bits.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 840 --repeat 3 2>&1 | tail -1 | sed -e 's/\[k_\([a-z0-9<>]*\)[^]]*\]/[\1]/' >> $P; }
for rep in 1 2; do
for v in -1 53600 53616 53601 53617 20832 20833; do
  run --bc pipe --cyl --n 3751 --ny 1251 --variant $v
done
done
cat $P
