#!/bin/bash
# round 6, GPU run 17: the halo cycle replayed from a captured hipGraph (LB_CYCLE_GRAPH=1, peer transport) against eager launches
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06_slab_proxy_graph.txt
: > $P
for rep in 1 2 3; do
  echo "== eager" >> $P
  timeout 300 python3 tools/slab_proxy.py --parts 8,4 --steps 280 --variants -1 --transports peer --reps 5 2>&1 | grep grid >> $P
  echo "== LB_CYCLE_GRAPH=1" >> $P
  LB_CYCLE_GRAPH=1 timeout 300 python3 tools/slab_proxy.py --parts 8,4 --steps 280 --variants -1 --transports peer --reps 5 2>&1 | grep grid >> $P
done
cut -c1-160 $P
