#!/bin/bash
# round 5, GPU run 55: the slab of one of N GPUs through the slab path on the final library (one-GPU proxy, 1-rank RCCL ring)
set -u
cd $GRAFT_REPO_ROOT
python3 tools/slab_proxy.py --steps 140 > gpurun_out/r05_slab_proxy_final.txt 2>&1
exit 0
