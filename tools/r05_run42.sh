#!/bin/bash
# round 5, GPU run 42: the GPU suite with k_step5's strips 240 apart; the velocity-inlet family and the slab proxy on it; the bench lines
set -u
cd $GRAFT_REPO_ROOT
timeout 3000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_g.txt 2>&1
{
for cfg in "velocity_inlet 8192 -1" "velocity_inlet 6144 -1" "velocity_inlet 4096 -1" "pipe 3072 -1" "cavity 2048 -1" "periodic 1280 -1"; do set -- $cfg
  echo -n "$1 $2 variant $3: "; python3 tools/run_case.py --bc $1 --n $2 --steps 80 --repeat 3 --variant $3 | sed 's/.*\]: //'
done
} > gpurun_out/r05_step5_align64_families.txt 2>&1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_smoke_g.txt 2>&1
python3 bench.py > gpurun_out/r05f_bench_default.json 2> gpurun_out/r05f_bench_default.err
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05f_bench_steps20.json 2> gpurun_out/r05f_bench_steps20.err
exit 0
