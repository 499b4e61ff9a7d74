#!/usr/bin/env python3
"""Kernel-variant sweep on one GPU (tuning aid, not part of the product).

Times the fused step for every (variant, grid) pair inside ONE process, interleaved over rounds
(cdna_hip_programming.md section 5.4 rule 24), and prints MLUPS / GB/s / fraction of 8 TB/s.
variant bits: 0 = non-temporal stores, 1 = non-temporal loads, 2-3 = rows per block (0:4, 1:1, 2:2),
4 = XCD-aware tile order, 5 = two time steps per pass (k_step2).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096,8192")
    ap.add_argument("--variants", default="0,1,9,32,33")
    ap.add_argument("--bc", default="periodic")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--mask", action="store_true")
    args = ap.parse_args()
    from LB_D2Q9.simulation import Simulation
    from bench import shear_layer
    for n in [int(s) for s in args.sizes.split(",")]:
        mask = None
        if args.mask:
            rng = np.random.default_rng(0)
            mask = rng.random((n, n)) < 0.01
        sim = Simulation(n, n, 1.7, bc=args.bc, inlet_rho=1.0005, obstacle_mask=mask)
        sim.init_equilibrium(*shear_layer(n, n, 0, n))
        variants = [int(v) for v in args.variants.split(",")]
        res = {v: [] for v in variants}
        for r in range(args.rounds):
            for v in variants:
                sim.set_variant(v)
                sim.run(4)
                ms = sim.timed_run(args.steps)
                res[v].append(n * n * args.steps / (ms * 1e-3) / 1e6)
        for nt in (False, True):
            gbs, nb = sim.copy_calibration(10, nt)
            print("n=%5d float4 copy (nt=%d): %.1f GB/s (%.0f MB per launch)" % (n, nt, gbs, nb / 1e6), flush=True)
        for v in variants:
            best, med = max(res[v]), float(np.median(res[v]))
            print("n=%5d bc=%s mask=%d variant=%2d  MLUPS best %9.1f median %9.1f  -> %7.1f GB/s  %.3f of 8 TB/s"
                  % (n, args.bc, int(args.mask), v, best, med, med * 72e-3, med * 72e-3 / 8000.), flush=True)
        sim.close()


if __name__ == "__main__":
    main()
