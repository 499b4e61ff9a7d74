#!/usr/bin/env python3
"""The reference's own benchmark case on this engine: Pipe_Flow_Cylinder(D=1, rho=1, nu=1, gradP=-10,
len=3, N=125, cylinder r=0.1 at (.75,.5)) = 3751 x 1251 = 4.693e6 cells, as timed in
docs/python_cython_opencl_comparison.ipynb (:136, 233, 271-273: OpenCL 317.5 MLUPS / 1000 steps on a GTX
Titan Black; :404-406: Cython 5.9 MLUPS / 20 steps).  Runs both drop-in classes (OpenCL-path semantics =
fused kernels; Cython-path semantics = boundary phase + one fused pass per step) and prints MLUPS the way the
notebook computes it (wall clock around run(), nx*ny*steps/t/1e6)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]


def main():
    from LB_D2Q9.dimensionless import cython_dim, opencl_dim
    kw = dict(diameter=1., rho=1., viscosity=1., pressure_grad=-10., pipe_length=3., N=125,
              cylinder_center=[.75, .5], cylinder_radius=.1, verbose=False)
    for name, mod, steps in (("opencl_dim (fused HIP kernels)", opencl_dim, 1000),
                             ("cython_dim (Cython-path semantics, bcs + one fused pass)   ", cython_dim, 200)):
        sim = mod.Pipe_Flow_Cylinder(**kw)
        sim.run(130)                             # long enough for the engine to pick its kernel configuration
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            sim.run(steps)                       # returns with the work complete, like the reference
            dt = time.perf_counter() - t0
            best = max(best, sim.nx * sim.ny * steps / dt / 1e6)
        print("%-62s %4d x %4d, %4d steps: %10.1f MLUPS" % (name, sim.nx, sim.ny, steps, best), flush=True)
    print("reference (its own numbers): OpenCL path 317.5 MLUPS on a GTX Titan Black, Cython path 5.9 MLUPS")


if __name__ == "__main__":
    main()
