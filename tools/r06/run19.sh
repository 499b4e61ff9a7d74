#!/bin/bash
# round 6, GPU run 19: peer transport, bands in one launch against split bands, ONE box, parts 1 (the whole grid as one slab), 2, 4, 8
set -u
cd $GRAFT_REPO_ROOT
P=gpurun_out/r06_slab_proxy_peer_split_ab.txt
: > $P
for rep in 1 2 3; do
  for sp in 0 1; do
    echo "== LB_SPLIT_BANDS=$sp" >> $P
    LB_SPLIT_BANDS=$sp timeout 400 python3 tools/slab_proxy.py --parts 1,2,4,8 --steps 140 --variants -1 --transports peer --reps 5 2>&1 | grep grid | cut -c1-140 >> $P
  done
done
cat $P
