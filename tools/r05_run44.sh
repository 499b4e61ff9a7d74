#!/bin/bash
# round 5, GPU run 44: the velocity-inlet family at 4096^2 / 8192^2 under k_step5: edge-cost scan with the search-based segment split
set -u
cd $GRAFT_REPO_ROOT
{
for c in 0 1.2 1.6 2.0 2.3 2.6 3.0 3.5; do for n in 4096 8192; do
  echo -n "LB_EDGE_COST=$c velocity_inlet $n: "; LB_EDGE_COST=$c python3 tools/run_case.py --bc velocity_inlet --n $n --steps 80 --repeat 3 | sed 's/.*\]: //'
done; done
echo -n "autotuned 4096: "; python3 - <<'PY'
import os, sys, time, numpy as np
sys.path[:0] = [os.path.join(os.environ["GRAFT_REPO_ROOT"], "2d-lb_amd")]
from LB_D2Q9.simulation import Simulation
n = 4096
s = Simulation(n, n, 1.2, bc="velocity_inlet", inlet_u=0.02)
s.init_equilibrium(np.ones((n, n), np.float32), np.full((n, n), 0.02, np.float32), np.zeros((n, n), np.float32))
s.autotune(); s.run(100)
t0 = time.perf_counter(); s.run(400); dt = time.perf_counter() - t0
print(s.hot_kernel(), "%.1f MLUPS" % (n * n * 400 / dt / 1e6))
PY
} > gpurun_out/r05_vel_edge_scan.txt 2>&1
exit 0
