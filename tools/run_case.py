#!/usr/bin/env python3
"""Run one configuration of the fused kernels for a few steps (a target for rocprofv3).
Usage: tools/run_case.py --bc pipe [--mask] [--n 8192] [--steps 12] [--variant 361]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bc", default="periodic")
    ap.add_argument("--mask", action="store_true")
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--ny", type=int, default=0, help="rows (default: n)")
    ap.add_argument("--repeat", type=int, default=1, help="timed runs; the best is printed")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--variant", type=int, default=-1)
    a = ap.parse_args()
    from LB_D2Q9.simulation import Simulation
    from bench import shear_layer
    mask = None
    ny = a.ny or a.n
    if a.mask:
        mask = np.random.default_rng(0).random((a.n, ny)) < 0.01
    sim = Simulation(a.n, ny, 1.7, bc=a.bc, inlet_rho=1.0005, obstacle_mask=mask)
    sim.set_variant(a.variant)
    sim.init_equilibrium(*shear_layer(a.n, ny, 0, ny))
    sim.run(a.steps)
    ms = min(sim.timed_run(a.steps) for _ in range(a.repeat))
    print("%s %dx%d mask=%d variant=%d [%s]: %.1f MLUPS, %.1f us per step" % (
        a.bc, a.n, ny, a.mask, a.variant, sim.hot_kernel(), a.n * ny * a.steps / ms / 1e3, ms * 1e3 / a.steps))


if __name__ == "__main__":
    main()
