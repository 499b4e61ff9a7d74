#!/usr/bin/env python3
"""The reference's verification study (docs/opencl_dimensionless_verification.ipynb cells 7-17, 35: plane Poiseuille flow,
D = 1.5 m, rho = 10 kg/m^3, nu = 5 m^2/s, grad P = -100 Pa/m, N = 10 / 50 / 200 lattice points across the diameter, each
run to dimensionless time 10 = 999 / 25 000 / 400 000 steps) through the drop-in classes, on the GPU.

    python examples/poiseuille_convergence.py          # prints the RMS deviation from u(y) = grad P/(2 rho nu) y (y - D)

On the reference's GTX Titan Black the N = 200 case alone took minutes; here the whole study takes a few seconds."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "2d-lb_amd"))

D, RHO, NU, GRAD_P = 1.5, 10., 5., -100.


def study(resolutions=(10, 50, 200), time_to_run=10., verbose=False):
    from LB_D2Q9.dimensionless import opencl_dim as lb_cl
    out = []
    for N in resolutions:
        sim = lb_cl.Pipe_Flow(diameter=D, rho=RHO, viscosity=NU, pressure_grad=GRAD_P, pipe_length=2 * D, N=N,
                              time_prefactor=1., verbose=verbose)
        num_steps = int(time_to_run / sim.delta_t)
        t0 = time.perf_counter()
        sim.run(num_steps)
        wall = time.perf_counter() - t0
        u = sim.get_physical_fields()["u"]
        y = np.arange(sim.ny) * sim.delta_x * sim.L                    # cell 35 of the notebook
        mean_u = u.T.mean(axis=1)
        theory = (1. / (2 * RHO * NU)) * GRAD_P * y * (y - D)
        rms = float(np.sqrt(((mean_u - theory) ** 2).mean()))
        out.append(dict(N=N, steps=num_steps, nx=sim.nx, ny=sim.ny, rms=rms, peak=float(theory.max()), wall_s=wall,
                        mlups=sim.nx * sim.ny * num_steps / wall / 1e6))
    return out


if __name__ == "__main__":
    for r in study():
        print("N = %3d  grid %4d x %3d  %6d steps  RMS deviation %.3e m/s (%.3f %% of the peak %.4f m/s)  %.2f s, %.0f MLUPS"
              % (r["N"], r["nx"], r["ny"], r["steps"], r["rms"], 100 * r["rms"] / r["peak"], r["peak"], r["wall_s"], r["mlups"]))
