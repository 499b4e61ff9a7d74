"""Periodic multi-population lattices: N D2Q9 BGK populations on one periodic grid, stepped together.

The reference's research forks (``LB_D2Q9/porous_media/single_component.py``, ``multicomponent_multiphase/multi.py``)
keep ``num_populations`` fluids in arrays shaped ``(nx, ny, num_populations[, num_jumpers])`` (Fortran order:
single_component.py:303-326) and stream them one population at a time with the kernel ``move_periodic``
(single_component.cl:338-375: ``f_streamed[jump][field][y+cy][x+cx] = f[jump][field][y][x]`` with periodic wrap).
Their forcing / interaction physics is out of scope; this module offers what that streaming sits on: every population is
one engine lattice (``Simulation(bc='periodic')``, its own ``omega``), ``move()`` is the reference's periodic streaming
of every population (``lb_move``: pure data movement, exact), and ``run(n)`` advances all populations with ONE fused
stream + collide launch per time step (``lb_run_batch``, bitwise equal to running each population alone).
fp32 like the rest of the engine (the forks are float64).
"""
import ctypes as ct

import numpy as np

from . import _native
from ._native import check
from .simulation import NUM_JUMPERS, Simulation


class Periodic_Populations(object):
    MAX_POPULATIONS = 8

    def __init__(self, nx, ny, omegas, obstacle_mask=None, device=0):
        """
        :param omegas: one BGK relaxation rate per population (the forks' ``Fluid.tau`` per field).
        :param obstacle_mask: optional (nx, ny) bounce-back mask shared by all populations.
        """
        omegas = list(omegas)
        if not 1 <= len(omegas) <= self.MAX_POPULATIONS:
            raise ValueError("1..%d populations" % self.MAX_POPULATIONS)
        self.nx, self.ny, self.num_populations, self.num_jumpers = int(nx), int(ny), len(omegas), NUM_JUMPERS
        self.omegas = omegas
        self.populations = [Simulation(nx, ny, om, bc="periodic", obstacle_mask=obstacle_mask, device=device)
                            for om in omegas]
        self._lib = _native.lib()
        self._handles = (ct.c_void_p * len(omegas))(*[p._h for p in self.populations])

    def close(self):
        for p in self.populations:
            p.close()

    # -- state: arrays shaped like the forks' (nx, ny, num_populations[, num_jumpers]), Fortran order ---------------
    def set_f(self, f):
        f = np.asarray(f)
        if f.shape != (self.nx, self.ny, self.num_populations, self.num_jumpers):
            raise ValueError("expected shape (nx, ny, num_populations, num_jumpers)")
        for i, p in enumerate(self.populations):
            p.set_f(f[:, :, i, :])

    def init_equilibrium(self, rho, u, v):
        """rho, u, v: (nx, ny, num_populations); f = feq on every population."""
        for i, p in enumerate(self.populations):
            p.init_equilibrium(rho[:, :, i], u[:, :, i], v[:, :, i])

    def get_fields(self, which=("f", "rho", "u", "v")):
        out = {}
        per = [p.get_fields(which) for p in self.populations]
        for k in which:
            out[k] = np.asfortranarray(np.stack([g[k] for g in per], axis=2))
        return out

    # -- stepping ------------------------------------------------------------------------------------------------------
    def move(self):
        """`move_periodic` for every population (single_component.cl:338-375): periodic streaming only."""
        for p in self.populations:
            p.move()

    def run(self, num_iterations, wait=True):
        """num_iterations fused time steps of all populations, one launch per step."""
        check(self._lib.lb_run_batch(self._handles, self.num_populations, int(num_iterations)))
        if wait:
            for p in self.populations:
                p.sync()

    def step(self):
        self.run(1)
