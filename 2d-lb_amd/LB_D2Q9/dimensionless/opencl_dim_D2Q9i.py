"""HIP-backed stand-in for the reference's ``LB_D2Q9.dimensionless.opencl_dim_D2Q9i``.

The reference module is ``opencl_dim.py`` with three changes (diff of the two files): it builds ``D2Q9i.cl`` -- the
"incompressible" fork of the kernels: ``update_hydro`` stores momentum instead of velocity (D2Q9i.cl:90-94), ``update_feq``
uses ``rho + 3 cu + 4.5 cu^2 - 1.5 u^2`` (:58), ``move_bcs`` has re-derived inlet / outlet formulas (:194-205) --; it
takes the Cython classes' non-dimensionalisation (``T = 8 rho nu / (|grad P| L)``, Reynolds number, ``omega =
1/(nu_lb/cs^2 + 1/2)``, ``delta rho`` scaled by ``T^2/(rho L) grad P``: opencl_dim_D2Q9i.py:98-120, 180, 253-255, 440);
and its cylinder class zeroes u, v inside the obstacle after every ``update_hydro`` (:490-503).

Here: the same classes on the engine's ``semantics='d2q9i'`` kernels (fused ``run``, one kernel per phase method).
The fork is reproduced as it is, including its instability (executed faithfully, |u| grows about tenfold in ten steps
from a 2e-4 density drop and overflows within a hundred: tests/golden/o2_d2q9i_53x27, tests/test_gpu_d2q9i.py) -- no
notebook of the reference uses it.  ``run`` behaves as the reference's by default: it never raises, a diverged run hands
back NaN fields.  Opt in with ``raise_on_divergence=True`` (constructor keyword, not in the reference): ``run`` then ends
with the device-side health check (one more pass over the populations and a host synchronisation) and raises
``FloatingPointError`` once the lattice holds non-finite cells.
"""
import numpy as np

from ..simulation import Simulation
from . import hip_dim
from .hip_dim import NUM_JUMPERS, cs, cs2, cs22, cx, cy, two_cs4, w, w0, w1, w2   # noqa: F401  (the module's constants)


class Pipe_Flow(hip_dim.Pipe_Flow):
    def __init__(self, *args, **kwargs):
        # not a reference keyword: False = the reference's behaviour (run() never raises)
        self.raise_on_divergence = bool(kwargs.pop('raise_on_divergence', False))
        super(Pipe_Flow, self).__init__(*args, **kwargs)

    def _derive_lattice_parameters(self, N, time_prefactor):
        """opencl_dim_D2Q9i.py:98-120: Reynolds number in place of the OpenCL class's W, omega with 1/cs^2."""
        self.Re = self.L ** 2 / (self.phys_visc * self.T ** 2)
        self._say('Reynolds number:', self.Re)
        self.N = N
        self.delta_x = 1. / N
        self.delta_t = time_prefactor * self.delta_x ** 2
        self.lb_viscosity = (self.delta_t / self.delta_x ** 2) * (1. / self.Re)
        self.omega = (self.lb_viscosity / cs2 + 0.5) ** -1.
        self._say('omega', self.omega)
        assert self.omega < 2.

    def set_characteristic_length_time(self):
        """L = diameter, T = 8 rho nu / (|grad P| L) (opencl_dim_D2Q9i.py:175-180)."""
        self.L = self.phys_diameter
        self.T = (8 * self.phys_rho * self.phys_visc) / (np.abs(self.phys_pressure_grad) * self.L)

    def _boundary_densities(self):
        """delta rho = nx (dt^2/dx)/cs^2 (T^2/(rho L)) grad P (opencl_dim_D2Q9i.py:253-259)."""
        nondim_deltaP = (self.T ** 2 / (self.phys_rho * self.L)) * self.phys_pressure_grad
        delta_rho = self.nx * (self.delta_t ** 2 / self.delta_x) * (1. / cs2) * nondim_deltaP
        return 1. + np.abs(delta_rho), 1.

    def _engine(self):
        rin, rout = self._boundary_densities()
        return Simulation(self.nx, self.ny, self.omega, bc='pipe', inlet_rho=rin, outlet_rho=rout, device=self.device,
                          semantics='d2q9i')

    def run(self, num_iterations):
        """As the base class (and as the reference: a diverged run returns NaN fields).  With raise_on_divergence=True: then
        one device pass over the populations (Simulation.check); the fork is unstable, and a run that has produced
        non-finite cells raises FloatingPointError here instead of returning NaN fields later."""
        super(Pipe_Flow, self).run(num_iterations)
        if self.raise_on_divergence:
            self._sim.check(raise_nonfinite=True)


class Pipe_Flow_Cylinder(Pipe_Flow, hip_dim.Pipe_Flow_Cylinder):
    """Cylinder class of the fork; u, v are re-zeroed inside the obstacle after every update_hydro
    (opencl_dim_D2Q9i.py:490-503: the engine does it inside update_hydro and inside the fused step)."""

    def set_characteristic_length_time(self):
        """L = cylinder radius, T = 8 rho nu L / (|grad P| D^2) (opencl_dim_D2Q9i.py:436-440)."""
        self.L = self.phys_cylinder_radius
        self.T = (8 * self.phys_rho * self.phys_visc * self.L) / (np.abs(self.phys_pressure_grad) * self.phys_diameter ** 2)
