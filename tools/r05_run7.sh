#!/bin/bash
# round 5, GPU run 7: k_deep<6> / k_deep<7> in every family against the single-step kernel (bitwise), and timed against k_step5
set -u
cd $GRAFT_REPO_ROOT
{
python3 tools/step5_check.py --six --no-time 2>&1 | tail -12
python3 tools/step5_check.py --seven --sizes 8192,4096 2>&1 | tail -40
} > gpurun_out/r05_deep_families.txt 2>&1
