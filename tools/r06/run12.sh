#!/bin/bash
# round 6, GPU run 12: soak of the final library -- automatic choice against the single-step kernel, bit for bit, alone and beside a second
# process streaming an 8192^2 lattice; the same with k_deep2 forced (LB_VARIANT=119137)
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_soak.txt
echo "== alone" > $out
timeout 900 python3 tools/soak_bitwise.py >> $out 2>&1
echo "== beside a second process streaming an 8192^2 lattice" >> $out
python3 - <<'PY' &
import os, sys
sys.path[:0] = [os.path.join(os.environ["GRAFT_REPO_ROOT"], "2d-lb_amd"), os.environ["GRAFT_REPO_ROOT"]]
from LB_D2Q9.simulation import Simulation
from bench import shear_layer
s = Simulation(8192, 8192, 1.7, bc="periodic"); s.init_equilibrium(*shear_layer(8192, 8192, 0, 8192))
import time
t0 = time.time()
while time.time() - t0 < 420: s.run(140)
PY
NOISE=$!
sleep 20
timeout 600 python3 tools/soak_bitwise.py >> $out 2>&1
kill $NOISE 2>/dev/null; wait $NOISE 2>/dev/null
echo "== k_deep2 forced for the seven-step launches (LB_SOAK_VARIANT=119137)" >> $out
LB_SOAK_VARIANT=119137 timeout 600 python3 tools/soak_bitwise.py >> $out 2>&1
cat $out
