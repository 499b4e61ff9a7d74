#!/bin/bash
# What would a fifth fused time step cost?  Diagnostic build, LB_DIAG bit 20 adds a fifth stage's worth of work to every row of
# k_step4 (wrong results, timing only); LB_STEP2_WAVES_PER_CU = 6 is the occupancy a third LDS window per wave would leave.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
CHILD='
import os, sys
sys.path[:0] = [os.path.join(sys.argv[1], "2d-lb_amd"), sys.argv[1]]
from LB_D2Q9.simulation import Simulation
from bench import shear_layer
n = int(sys.argv[2])
sim = Simulation(n, n, 1.7, bc="periodic"); sim.set_variant(int(sys.argv[3]))
sim.init_equilibrium(*shear_layer(n, n, 0, n))
sim.run(8)
ms = min(sim.timed_run(40) for _ in range(3)) / 10.0
print("%.4f ms per launch" % ms)
'
for n in 8192 4096; do
  for w in 8 6; do
    for v in 353 1377; do      # 1377 = 353 | 1024: no one-row-ahead gather (the probe spills 56 B with it)
    for d in 0 1048576; do
      echo -n "n=$n waves/CU=$w variant=$v diag=$d: "
      LB_LIB=$REPO/2d-lb_amd/LB_D2Q9/liblbhip_diag.so LB_DIAG=$d LB_STEP2_WAVES_PER_CU=$w python3 -c "$CHILD" $REPO $n $v 2>/dev/null | tail -1
    done
    done
  done
done
