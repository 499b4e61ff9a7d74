#!/usr/bin/env python3
"""Does splitting ONE GPU's grid into G virtual row slabs (lb_run_group: each slab its own streams, the halo cycle between
them) fill the tail of the marching kernel's launches?  A launch of k_step4 is one round of resident waves and ends with its
slowest wave (wave slots busy ~92 % of a launch); with G slabs on G compute streams the next launch of one slab can start while
another slab's launch drains.  Prints MLUPS of the whole grid for G = 1 (plain handle) and G in --slabs.
    python tools/split_probe.py [--n 8192] [--ny 8192] [--slabs 2,4] [--steps 192]
(LB_STEP2_WAVES_PER_CU in the environment sizes every launch for that many waves per CU.)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--slabs", default="2,4")
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--bc", default="periodic")
    a = ap.parse_args()
    ny = a.ny or a.n
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import LocalSlabRing, partition_rows
    from bench import shear_layer
    one = Simulation(a.n, ny, 1.7, bc=a.bc)
    one.init_equilibrium(*shear_layer(a.n, ny, 0, ny))
    one.run(16)
    best = 0.0
    for _ in range(3):
        ms = one.timed_run(a.steps)
        best = max(best, a.n * ny * a.steps / (ms * 1e-3) / 1e6)
    print("%d x %d %s, wpc=%s: plain handle %.1f MLUPS [%s]" % (a.n, ny, a.bc, os.environ.get("LB_STEP2_WAVES_PER_CU", "8"), best, one.hot_kernel()), flush=True)
    one.close()
    for g in [int(x) for x in a.slabs.split(",")]:
        ring = LocalSlabRing(a.n, ny, 1.7, g, bc=a.bc)
        for s, (y0, h) in zip(ring.slabs, ring.parts):
            s.init_equilibrium(*shear_layer(a.n, ny, y0, h))
        ring.run_in_library(16)
        for s in ring.slabs:
            s.sync()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            ring.run_in_library(a.steps)
            for s in ring.slabs:
                s.sync()
            el = time.perf_counter() - t0
            best = max(best, a.n * ny * a.steps / el / 1e6)
        print("%d x %d %s, wpc=%s: %d virtual slabs through lb_run_group %.1f MLUPS (host clock, %d steps; %d steps per launch)"
              % (a.n, ny, a.bc, os.environ.get("LB_STEP2_WAVES_PER_CU", "8"), g, best, a.steps, ring.slabs[0].steps_per_launch()), flush=True)
        for s in ring.slabs:
            s.close()


if __name__ == "__main__":
    main()
