#!/bin/bash
# Round 3: from which lattice size do non-temporal stores pay in k_step4?  (352 = four-step kernel, plain stores; 353 = + non-temporal stores)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for shape in "8192 256" "8192 512" "8192 1024" "8192 2048" "4096 1024" "4096 2048" "4096 4096" "2048 2048" "3072 3072"; do
  set -- $shape
  for v in 352 353; do
    python tools/run_case.py --n $1 --ny $2 --steps 96 --repeat 5 --variant $v
  done
done
