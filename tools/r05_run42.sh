#!/bin/bash
# round 5, GPU run 42 (last form): the GPU suite, smoke() and the two bench lines on the final library
set -u
cd $GRAFT_REPO_ROOT
timeout 3000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_j.txt 2>&1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_smoke_j.txt 2>&1
python3 bench.py > gpurun_out/r05i_bench_default.json 2> gpurun_out/r05i_bench_default.err
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05i_bench_steps20.json 2> gpurun_out/r05i_bench_steps20.err
exit 0
