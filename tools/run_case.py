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
    ap.add_argument("--mask", action="store_true", help="random obstacle cells, 1 %")
    ap.add_argument("--cyl", action="store_true", help="one disc of radius n/10 (the reference's kind of obstacle)")
    ap.add_argument("--tiff", action="store_true", help="the porous-medium image of BASELINE config 5")
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
    if a.cyl:
        x, y = np.meshgrid(np.arange(a.n), np.arange(ny), indexing="ij")
        mask = (x - a.n / 4) ** 2 + (y - ny / 2) ** 2 < (min(a.n, ny) / 10) ** 2
    if a.tiff:
        from LB_D2Q9.masks import obstacle_mask_from_tiff
        mask = np.array(obstacle_mask_from_tiff(os.path.join(ROOT, "tests", "golden", "CS205_obstacle_4.tif"), (a.n, ny)), dtype=bool)
        mask[0, :] = mask[-1, :] = False
        mask[:, 0] = mask[:, -1] = False
    sim = Simulation(a.n, ny, 1.7, bc=a.bc, inlet_rho=1.0005, obstacle_mask=mask)
    sim.set_variant(a.variant)
    sim.init_equilibrium(*shear_layer(a.n, ny, 0, ny))
    sim.run(a.steps)
    ms = min(sim.timed_run(a.steps) for _ in range(a.repeat))
    print("%s %dx%d mask=%d variant=%d [%s]: %.1f MLUPS, %.1f us per step" % (
        a.bc, a.n, ny, bool(a.mask or a.cyl or a.tiff), a.variant, sim.hot_kernel(), a.n * ny * a.steps / ms / 1e3, ms * 1e3 / a.steps))


if __name__ == "__main__":
    main()
