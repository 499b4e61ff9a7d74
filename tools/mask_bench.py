#!/usr/bin/env python3
"""MLUPS of the pipe family with obstacle masks of different texture: the reference's porous-medium image (config 5: docs/
CS205_obstacle_4.tif rescaled, clustered, 1.1 % solid), a disc (Pipe_Flow_Cylinder), 1 % white noise.    python tools/mask_bench.py [n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    from LB_D2Q9.masks import obstacle_mask_from_tiff
    from LB_D2Q9.simulation import Simulation
    from bench import shear_layer
    tif = obstacle_mask_from_tiff(os.path.join(ROOT, "tests", "golden", "CS205_obstacle_4.tif"), (n, n))
    x, y = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    disc = ((x - n // 4) ** 2 + (y - n // 2) ** 2 < (n // 16) ** 2)
    noise = np.random.default_rng(0).random((n, n)) < 0.01
    for name, mask in (("none", None), ("porous image", tif), ("disc", disc), ("1 % noise", noise)):
        if mask is not None:
            mask = np.array(mask, dtype=bool)
            mask[0, :] = mask[-1, :] = False
            mask[:, 0] = mask[:, -1] = False
        sim = Simulation(n, n, 1.0, bc="pipe", inlet_rho=1.001, obstacle_mask=mask)
        sim.init_equilibrium(*shear_layer(n, n, 0, n))
        sim.run(8)
        sim.autotune()
        best = min(sim.timed_run(40) for _ in range(4))
        print("%d^2 pipe, mask %-13s (%.2f %% solid): %8.1f MLUPS  [%s]" % (
            n, name, 0. if mask is None else 100 * mask.mean(), n * n * 40 / (best * 1e-3) / 1e6, sim.hot_kernel()), flush=True)
        sim.close()


if __name__ == "__main__":
    main()
