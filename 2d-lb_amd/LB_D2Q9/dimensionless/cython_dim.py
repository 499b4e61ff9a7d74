"""GPU-backed stand-in for the reference's CPU module ``LB_D2Q9.dimensionless.cython_dim``.

Same classes, constructor keywords, derived constants, step order and field shapes/dtypes as
LB_D2Q9/dimensionless/cython_dim.pyx:31-513 of latticeboltzmann/2d-lb, with the numpy/Cython loops
replaced by the engine's Cython-path kernels (``semantics='cython'``: boundary rules before streaming fed
by the stored inlet/outlet velocity, plain bounce-back walls, the restricted in-place streaming, the
moment overrides).  The reference's two paths are different discretisations (they agree in the interior
and drift ~1 % apart at walls over 1000 steps), so users of the CPU classes get THIS module
and users of the OpenCL class get ``hip_dim`` / ``opencl_dim``.

It is a compatibility path: the phase methods launch one kernel per phase, ``run(n)`` a boundary kernel plus ONE fused
kernel per step (bitwise equal to the five phase calls; ~3x slower than the fused OpenCL-path kernels, four orders of
magnitude faster than the reference's own CPU loop), fp32 throughout where the reference keeps ``u, v`` and the
equilibrium temporaries in float64 (differences <= 1e-6 per step, tests).

Differences a caller can see: ``self.f``, ``self.rho`` ... are ``DeviceField`` objects (``.get()`` or
``np.asarray(...)`` give host copies) instead of live numpy arrays - assign state through
``set_fields`` / ``set_f``; ``get_nondim_fields`` scales copies (the reference scales its live ``u``,
``v`` in place, cython_dim.pyx:373-374, which corrupts the running simulation).
"""
import numpy as np

from ..masks import disc_pixels
from ..simulation import Simulation

# ---- D2Q9 constants (names/values of cython_dim.pyx:16-29) -----------------------------------------
NUM_JUMPERS = 9
w = np.array([4. / 9.] + 4 * [1. / 9.] + 4 * [1. / 36.])
cx = np.array([0, 1, 0, -1, 0, 1, -1, -1, 1])
cy = np.array([0, 0, 1, 0, -1, 1, 1, -1, -1])
cs = 1 / np.sqrt(3)
cs2 = cs ** 2
cs22 = 2 * cs2
cssq = 2.0 / 9.0
w0, w1, w2 = 4. / 9., 1. / 9., 1. / 36.


class DeviceField(object):
    """Read access to engine state under the reference's attribute names."""

    def __init__(self, owner, key):
        self._owner, self._key = owner, key

    def get(self):
        return self._owner.get_fields()[self._key]

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype)


class Pipe_Flow(object):
    """Pressure-driven pipe flow, CPU-class semantics, on the GPU."""

    def __init__(self, diameter=None, rho=None, viscosity=None, pressure_grad=1., pipe_length=None,
                 N=100, time_prefactor=1., device=0, verbose=True):
        self.verbose = verbose
        self.device = device
        self.phys_diameter = diameter
        self.phys_rho = rho
        self.phys_visc = viscosity
        self.phys_pressure_grad = pressure_grad
        self.phys_pipe_length = pipe_length

        self.L = self.T = None
        self.set_characteristic_length_time()
        self._say('Characteristic L:', self.L)
        self._say('Characteristic T:', self.T)
        self.Re = self.L ** 2 / (self.phys_visc * self.T ** 2)            # :69
        self._say('Reynolds number:', self.Re)

        self.N = N
        self.delta_x = 1. / N
        self.delta_t = time_prefactor * self.delta_x ** 2

        self.lx = self.ly = self.nx = self.ny = None
        self.initialize_grid_dims()

        self.lb_viscosity = (self.delta_t / self.delta_x ** 2) * (1. / self.Re)
        self.omega = (self.lb_viscosity / cs2 + 0.5) ** -1.                # :91
        self._say('omega', self.omega)
        assert self.omega < 2.

        self.inlet_rho = self.outlet_rho = None
        self._sim = None
        self.rho, self.u, self.v = (DeviceField(self, k) for k in ('rho', 'u', 'v'))
        self.f, self.feq = DeviceField(self, 'f'), DeviceField(self, 'feq')
        self.init_hydro()
        self.update_feq()
        self.init_pop()

    def _say(self, *args):
        if self.verbose:
            print(*args)

    def _boundary_densities(self):
        """delta rho = nx (dt^2/dx)/cs^2 (T^2/(rho L)) gradP (cython_dim.pyx:138-144)."""
        nondim_deltaP = (self.T ** 2 / (self.phys_rho * self.L)) * self.phys_pressure_grad
        delta_rho = self.nx * (self.delta_t ** 2 / self.delta_x) * (1. / cs2) * nondim_deltaP
        return 1. + np.abs(delta_rho), 1.

    def _obstacle(self):
        return None

    # ---- hooks with the reference's names ---------------------------------------------------------
    def set_characteristic_length_time(self):
        """L = diameter, T = 8 rho nu / (|gradP| L) (cython_dim.pyx:107-115)."""
        self.L = self.phys_diameter
        self.T = (8 * self.phys_rho * self.phys_visc) / (np.abs(self.phys_pressure_grad) * self.L)

    def initialize_grid_dims(self):
        self.lx = int(np.ceil((self.phys_pipe_length / self.L) * self.N))
        self.ly = self.N
        self.nx, self.ny = self.lx + 1, self.ly + 1

    def init_hydro(self):
        """Density ramp, fluid at rest, velocity zero in the obstacle (cython_dim.pyx:129-157, 451-457)."""
        self.inlet_rho, self.outlet_rho = self._boundary_densities()
        self._say('inlet rho:', self.inlet_rho)
        self._say('outlet rho:', self.outlet_rho)
        if self._sim is None:
            self._sim = Simulation(self.nx, self.ny, self.omega, bc='pipe', inlet_rho=self.inlet_rho,
                                   outlet_rho=self.outlet_rho, device=self.device, semantics='cython')
        self._sim.set_obstacle_mask(self._obstacle())
        i = np.arange(self.nx, dtype=np.float64)[:, None]
        ramp = self.inlet_rho - i * (self.inlet_rho - self.outlet_rho) / float(self.nx)
        rho_host = np.broadcast_to(ramp, (self.nx, self.ny)).astype(np.float32)
        zero = np.zeros((self.nx, self.ny), np.float32)
        self._sim.set_fields(rho_host, zero, zero)

    def update_feq(self):
        self._sim.update_feq()

    def init_pop(self, amplitude=.001):
        """f = feq (1 + amplitude N(0,1)), one draw per CELL shared by the nine links (cython_dim.pyx:191-202)."""
        perturb = None
        if amplitude:
            perturb = (1. + amplitude * np.random.randn(self.nx, self.ny))[:, :, None]
        self._sim.init_pop(perturb)

    def move_bcs(self):
        self._sim.move_bcs()

    def move(self):
        self._sim.move()

    def update_hydro(self):
        self._sim.update_hydro()

    def collide_particles(self):
        self._sim.collide_particles()

    def run(self, num_iterations):
        """move_bcs, move, update_hydro, update_feq, collide_particles per iteration (cython_dim.pyx:346-359)."""
        self._sim.run(num_iterations)

    def step(self):
        self._sim.run(1)

    # ---- state in / out ----------------------------------------------------------------------------
    def set_f(self, f):
        """f: (9, nx, ny) like the reference's ``self.f``."""
        self._sim.set_f(np.asarray(f, np.float32).transpose(1, 2, 0))

    def set_fields(self, rho, u, v):
        self._sim.set_fields(rho, u, v)

    def get_fields(self):
        """f, feq: (9, nx, ny) float32; rho (nx, ny) float32; u, v (nx, ny) float64 - the reference's
        shapes and dtypes (cython_dim.pyx:101-102, 150, 156-157, 361-371), as host copies."""
        g = self._sim.get_fields()
        return {'f': np.ascontiguousarray(g['f'].transpose(2, 0, 1)),
                'feq': np.ascontiguousarray(g['feq'].transpose(2, 0, 1)),
                'rho': np.ascontiguousarray(g['rho']),
                'u': np.ascontiguousarray(g['u'], dtype=np.float64),
                'v': np.ascontiguousarray(g['v'], dtype=np.float64)}

    def get_nondim_fields(self):
        fields = self.get_fields()
        fields['u'] *= self.delta_x / self.delta_t
        fields['v'] *= self.delta_x / self.delta_t
        return fields

    def get_physical_fields(self):
        fields = self.get_nondim_fields()
        fields['u'] *= (self.L / self.T)
        fields['v'] *= (self.L / self.T)
        return fields


class Pipe_Flow_Cylinder(Pipe_Flow):
    """Pipe flow past a cylinder, CPU-class semantics (cython_dim.pyx:398-513)."""

    def __init__(self, cylinder_center=None, cylinder_radius=None, **kwargs):
        assert cylinder_center is not None
        assert cylinder_radius is not None
        self.phys_cylinder_center = cylinder_center
        self.phys_cylinder_radius = cylinder_radius
        self.obstacle_mask = None
        super(Pipe_Flow_Cylinder, self).__init__(**kwargs)
        self.obstacle_pixels = np.where(self.obstacle_mask)

    def set_characteristic_length_time(self):
        """L = cylinder radius, T = 8 rho nu L / (|gradP| D^2) (cython_dim.pyx:405-412)."""
        self.L = self.phys_cylinder_radius
        self.T = (8 * self.phys_rho * self.phys_visc * self.L) / (np.abs(self.phys_pressure_grad) * self.phys_diameter ** 2)

    def initialize_grid_dims(self):
        self.lx = int(np.ceil((self.phys_pipe_length / self.L) * self.N))
        self.ly = int(np.ceil((self.phys_diameter / self.L) * self.N))
        self.nx, self.ny = self.lx + 1, self.ly + 1
        self.obstacle_mask = np.zeros((self.nx, self.ny), dtype=bool, order='F')
        xs, ys = disc_pixels(self.N * self.phys_cylinder_center[0] / self.L,
                             self.N * self.phys_cylinder_center[1] / self.L, self.N, (self.nx, self.ny))
        self.obstacle_mask[xs, ys] = True

    def _obstacle(self):
        return self.obstacle_mask
