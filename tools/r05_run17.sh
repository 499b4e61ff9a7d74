#!/bin/bash
# round 5, GPU run 17: slabs on k_deep (twelve- / fourteen-step halo cycles): slab tests, then the one-GPU proxy of the strong-scaled 8192^2
set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_peer_ranks.py tests/test_gpu_random.py tests/test_gpu_multirank.py tests/test_gpu_fullsize.py -m gpu -q -x -k "slab or ring or rank or peer or random_partition or rccl" 2>&1 | tail -15 > gpurun_out/r05_gputest_slabs.txt
python3 tools/slab_proxy.py > gpurun_out/r05_slab_proxy.txt 2>&1
