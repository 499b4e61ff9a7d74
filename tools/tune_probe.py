#!/usr/bin/env python3
"""Static kernel choice vs lb_autotune's across grid sizes and families, after a warm-up long enough for the clocks to settle
(a 4096^2 step takes 60 us: the first few hundred steps of a process run ~10 % slow).    python tools/tune_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]


def main():
    from LB_D2Q9.simulation import Simulation
    from bench import shear_layer
    sizes = [int(a) for a in sys.argv[1:]] or [2048, 3072, 4096, 6144, 8192]
    for n in sizes:
        for bc in ("periodic", "pipe", "cavity", "velocity_inlet"):
            sim = Simulation(n, n, 1.0, bc=bc, inlet_rho=1.001, lid_u=0.05, inlet_u=0.02)
            sim.init_equilibrium(*shear_layer(n, n, 0, n))
            sim.run(400)
            static = min(sim.timed_run(80) for _ in range(4))
            sim.autotune()
            tuned = min(sim.timed_run(80) for _ in range(4))
            print("%5d %-14s static %8.1f  tuned %8.1f MLUPS  %s" % (n, bc, n * n * 80 / (static * 1e-3) / 1e6,
                                                                     n * n * 80 / (tuned * 1e-3) / 1e6, sim.hot_kernel()), flush=True)
            sim.close()


if __name__ == "__main__":
    main()
