#!/usr/bin/env python3
"""Ablation timing of the fused kernels (diagnostic build, wrong results by design, timing only).
LB_DIAG bits: 1 = skip step-1 collide, 2 = skip step-2 collide, 4 = no stores (k_step2) / skip step-3 collide
(k_step3), 8 = all loads aligned, 512 = no boundary rule in the 4-cell path.
Each configuration runs in its own process (the switches are read at lb_create)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path[:0] = [os.path.join(%r, "2d-lb_amd"), %r]
from LB_D2Q9.simulation import Simulation
from bench import shear_layer
n = int(sys.argv[1]); variant = int(sys.argv[2])
sim = Simulation(n, n, 1.7, bc="periodic"); sim.set_variant(variant)
sim.init_equilibrium(*shear_layer(n, n, 0, n))
sim.run(6)
best = min(sim.timed_run(20) for _ in range(3))
print("%%.1f" %% (n * n * 20 / (best * 1e-3) / 1e6))
''' % (ROOT, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8192
    lib = os.path.join(ROOT, "2d-lb_amd", "LB_D2Q9", "liblbhip_diag.so")
    configs = ((33, "k_step2"), (9, "k_step"))
    if "--step3" in sys.argv:
        configs = ((97, "k_step3"),)
    if "--tile" in sys.argv:
        # k_tile4 hooks: 1 = no arithmetic at all (load, 4 x {LDS pull, barrier, write-back, barrier}, store), 2 = load, one step,
        # store (no steps inside LDS), 3 = both: what a launch costs in data movement alone; 8 = every other workgroup of an XCD steps
        # first and loads afterwards (the workgroups of a CU out of phase: what overlapping loads with steps would be worth)
        bc = "cavity" if "--cavity" in sys.argv else "periodic"
        child = CHILD.replace('bc="periodic"', 'bc="%s", lid_u=0.1' % bc).replace("sim.timed_run(20)", "sim.timed_run(400)").replace("n * n * 20", "n * n * 400")
        for diag, what in ((0, "full"), (1, "no arithmetic"), (2, "one step per launch (no steps in LDS)"),
                           (3, "load + one pull + store, no arithmetic"), (8, "every other workgroup out of phase"), (0, "full again")):
            env = dict(os.environ, LB_LIB=lib, LB_DIAG=str(diag))
            out = subprocess.run([sys.executable, "-c", child, str(n), str(864)], env=env, capture_output=True, text=True)
            val = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-200:]
            print("k_tile4 %s %d^2 diag=%d %-42s %s MLUPS-equivalent (4 steps per launch counted)" % (bc, n, diag, what, val), flush=True)
        return
    if "--step4" in sys.argv:
        # k_step4 hooks: 1 / 2 / 4 / 2048 = skip the collide of stage 1 / 2 / 3 / 4, 1024 = skip the halo cells' stages
        pf = 1024 if "--pf" in sys.argv else 0
        for diag, what in ((0, "full"), (1024, "no halo-cell stages"), (2055, "no collide in any stage"),
                           (3079, "no collide, no halo cells"), (1, "no stage-1 collide"), (0, "full again")):
            env = dict(os.environ, LB_LIB=lib, LB_DIAG=str(diag))
            out = subprocess.run([sys.executable, "-c", CHILD, str(n), str(353 | pf)], env=env, capture_output=True, text=True)
            val = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-200:]
            print("%-8s diag=%4d %-28s %s MLUPS-equivalent" % ("k_step4" + ("+pf" if pf else ""), diag, what, val), flush=True)
        return
    for variant, name in configs:
        for diag, what in ((0, "full"), (1, "no step-1 collide"), (2, "no step-2 collide"), (3, "no collide at all"),
                           (4, "no stores"), (8, "aligned loads"), (11, "no collide, aligned loads"),
                           (7, "loads only"), (16, "aligned 256-cell strips"), (19, "aligned strips, no collide"),
                           (17, "aligned strips, no step-1 collide")) + \
                (((7, "no collide at all (3 steps)"), (0, "full again"))
                 if name == "k_step3" else ()):
            if name == "k_step" and diag in (2, 3, 4, 7, 11, 16, 19, 17):
                continue
            if name == "k_step3" and diag not in (0, 1, 7):
                continue
            env = dict(os.environ, LB_LIB=lib, LB_DIAG=str(diag))
            out = subprocess.run([sys.executable, "-c", CHILD, str(n), str(variant)], env=env, capture_output=True, text=True)
            val = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-200:]
            print("%-8s diag=%2d %-28s %s MLUPS-equivalent" % (name, diag, what, val), flush=True)


if __name__ == "__main__":
    main()
