#!/bin/bash
# round 5, GPU run 35: soak of the final library (hand-waited gather): automatic kernel choice against the single-step kernel, bit for
# bit -- alone, and again while a second process keeps the memory system busy (the gather's arrival times move)
set -u
cd $GRAFT_REPO_ROOT
{
echo "== alone"; python3 tools/soak_bitwise.py --more
echo "== beside a second process streaming an 8192^2 lattice"
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do python3 tools/run_case.py --bc periodic --n 8192 --steps 2000 --variant 9 > /dev/null 2>&1; done ) &
BG=$!
python3 tools/soak_bitwise.py --more
kill $BG 2>/dev/null; wait $BG 2>/dev/null
} > gpurun_out/r05_soak.txt 2>&1
exit 0
