#!/bin/bash
# round 6, GPU run 47: the static table after the periodic revision (k_deep<6> from 1100^2, k_deep<7> from 1900^2, tiles below 1100^2): GPU
# suite; the automatic variant across sizes and families (1680-step runs, best of 3)
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r06r_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06r_pytest_gpu.log
P=gpurun_out/r06r_static_choice.txt
: > $P
run() { timeout 100 python3 tools/run_case.py "$@" --steps 1680 --repeat 3 2>&1 | tail -1 | cut -c1-150 >> $P; }
for n in 1024 1152 1280 1536 1792 2048 2560; do
  for fam in "--bc periodic" "--bc pipe" "--bc cavity --mask"; do
    run $fam --n $n --variant -1
  done
done
run --bc pipe --cyl --n 3751 --ny 1251 --variant -1
tail -3 gpurun_out/r06r_pytest_gpu.log
cat $P
