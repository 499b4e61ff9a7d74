#!/usr/bin/env python3
"""Whole-grid timing of arbitrary nx x ny periodic boxes per kernel variant (tuning aid).
Usage: tools/rect_probe.py nx ny variant[,variant...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]
from LB_D2Q9.simulation import Simulation   # noqa: E402
from bench import shear_layer               # noqa: E402

nx, ny = int(sys.argv[1]), int(sys.argv[2])
for v in [int(x) for x in sys.argv[3].split(",")]:
    sim = Simulation(nx, ny, 1.7, bc="periodic")
    sim.set_variant(v)
    sim.init_equilibrium(*shear_layer(nx, ny, 0, ny))
    sim.run(12)
    ms = sorted(sim.timed_run(48) for _ in range(5))
    print("%d x %d variant %d: best %.1f median %.1f MLUPS" % (nx, ny, v, nx * ny * 48 / ms[0] / 1e3, nx * ny * 48 / ms[2] / 1e3), flush=True)
    sim.close()
