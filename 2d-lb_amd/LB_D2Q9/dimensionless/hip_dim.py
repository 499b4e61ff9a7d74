"""Drop-in for the reference's ``LB_D2Q9.dimensionless.opencl_dim`` on MI355X.

Same classes (``Pipe_Flow``, ``Pipe_Flow_Cylinder``), constructor keywords, attributes, hook
methods and field shapes as LB_D2Q9/dimensionless/opencl_dim.py:58-518 of latticeboltzmann/2d-lb,
with the pyopencl context/queue/program/buffer plumbing (:203-255) replaced by a ``liblbhip``
handle (HIP device arrays behind the C ABI of include/lb_hip.h).  Numerical semantics are those
of the OpenCL path (step order move -> move_bcs -> update_hydro -> update_feq -> collide_particles,
wall rule of D2Q9.cl:213-223); ``run(n)`` executes the fused HIP kernel, the five phase methods
execute one un-fused kernel each.

Differences a caller can see:
  * ``self.f``, ``self.rho`` ... are ``DeviceField`` objects (``.get()`` -> numpy) instead of
    ``cl.Buffer``; ``self.context`` / ``self.queue`` / ``self.kernels`` do not exist.
  * ``step()`` (== ``run(1)``) and ``verbose=`` are additions; ``two_d_local_size``,
    ``three_d_local_size`` and ``use_interop`` are accepted and ignored (launch geometry is
    the engine's business; there is no CL/GL sharing).
"""
import numpy as np

from .. import _native
from ..masks import disc_pixels
from ..simulation import Simulation

# ---- D2Q9 lattice constants (same names/values as opencl_dim.py:22-36) ----------------------
NUM_JUMPERS = 9
w = np.array([4. / 9.] + 4 * [1. / 9.] + 4 * [1. / 36.], order='F', dtype=np.float32)
cx = np.array([0, 1, 0, -1, 0, 1, -1, -1, 1], order='F', dtype=np.int32)
cy = np.array([0, 0, 1, 0, -1, 1, 1, -1, -1], order='F', dtype=np.int32)
cs = 1. / np.sqrt(3)
cs2 = cs ** 2
cs22 = 2 * cs2
cssq = 2.0 / 9.0
two_cs4 = 2 * cs ** 4
w0, w1, w2 = 4. / 9., 1. / 9., 1. / 36.


def get_divisible_global(global_size, local_size):
    """Smallest multiple of local_size that covers global_size, per dimension
    (kept for callers that print or reuse it; opencl_dim.py:39-56)."""
    return tuple(-(-g // l) * l for g, l in zip(global_size, local_size))


class DeviceField(object):
    """Stand-in for the reference's ``cl.Buffer`` attributes: a named view of engine state."""

    def __init__(self, owner, key):
        self._owner, self._key = owner, key

    def get(self):
        return self._owner._sim.get_fields((self._key,))[self._key]


class Pipe_Flow(object):
    """Pressure-driven flow between two plates on the D2Q9 lattice (the reference's verification case)."""

    def __init__(self, diameter=None, rho=None, viscosity=None, pressure_grad=None, pipe_length=None,
                 N=200, time_prefactor=1.,
                 two_d_local_size=(32, 32), three_d_local_size=(32, 32, 1), use_interop=False,
                 device=0, verbose=True):
        self.verbose = verbose
        self.device = device
        # physical inputs
        self.phys_diameter = diameter
        self.phys_rho = rho
        self.phys_visc = viscosity
        self.phys_pressure_grad = pressure_grad
        self.phys_pressure_grad_div_rho = pressure_grad / rho
        self.phys_pipe_length = pipe_length
        self.use_interop = use_interop

        # characteristic scales and the dimensionless group of the OpenCL class (opencl_dim.py:93-104)
        self.L = None
        self.T = None
        self.set_characteristic_length_time()
        self._say('Characteristic L:', self.L)
        self._say('Characteristic T:', self.T)
        self._derive_lattice_parameters(N, time_prefactor)

        self.lx = self.ly = self.nx = self.ny = None
        self.initialize_grid_dims()

        # launch geometry of the reference, reported for compatibility only
        self.two_d_local_size = two_d_local_size
        self.three_d_local_size = three_d_local_size
        self.two_d_global_size = get_divisible_global((self.nx, self.ny), two_d_local_size)
        self.three_d_global_size = get_divisible_global((self.nx, self.ny, 9), three_d_local_size)
        self._say('2d global:', self.two_d_global_size)
        self._say('2d local:', self.two_d_local_size)
        self._say('3d global:', self.three_d_global_size)
        self._say('3d local:', self.three_d_local_size)

        self._sim = None
        self.init_hip()
        self.allocate_constants()

        self.inlet_rho = self.outlet_rho = None
        self.rho, self.u, self.v = (DeviceField(self, k) for k in ('rho', 'u', 'v'))
        self.init_hydro()

        self.feq = DeviceField(self, 'feq')
        self.update_feq()
        self.f = DeviceField(self, 'f')
        self.f_streamed = self.f
        self.init_pop()

    # ---- helpers ------------------------------------------------------------------------------
    def _derive_lattice_parameters(self, N, time_prefactor):
        """The dimensionless group and the lattice units of the OpenCL class (opencl_dim.py:102-120)."""
        self.W = (np.abs(self.phys_pressure_grad_div_rho) * self.L * self.T) / self.phys_visc
        self._say('Weinstein number:', self.W)

        # lattice units (:106-120)
        self.N = N
        self.delta_x = 1. / N
        self.delta_t = time_prefactor * self.delta_x ** 2
        self.ulb = self.delta_t / self.delta_x
        self._say('u_lb:', self.ulb)
        self.lb_viscosity = (self.delta_t / self.delta_x ** 2) * (1. / self.W)
        self.omega = (3 * self.lb_viscosity + 0.5) ** -1.
        self._say('omega', self.omega)
        assert self.omega < 2.

    def _say(self, *args):
        if self.verbose:
            print(*args)

    def _boundary_densities(self):
        """rho_out = 1, rho_in = 1 + |nx (dt^2/dx) / cs^2| (opencl_dim.py:266-273)."""
        delta_rho = self.nx * (self.delta_t ** 2 / self.delta_x) * (1. / cs2) * 1.
        return 1. + np.abs(delta_rho), 1.

    # ---- hooks with the reference's names -----------------------------------------------------
    def set_characteristic_length_time(self):
        """L = pipe diameter, T = sqrt(L / (|grad P| / rho)) (opencl_dim.py:180-189)."""
        self.L = self.phys_diameter
        self.T = np.sqrt(self.phys_diameter / (np.abs(self.phys_pressure_grad) / self.phys_rho))

    def initialize_grid_dims(self):
        """lx = ceil(pipe_length / L * N), ly = N; one boundary node more in each direction (:191-201)."""
        self.lx = int(np.ceil((self.phys_pipe_length / self.L) * self.N))
        self.ly = self.N
        self.nx, self.ny = self.lx + 1, self.ly + 1

    # The reference-named classes keep the reference's observable fields: rho, u, v are what update_hydro stored in the last
    # step (the moments of the PRE-collision populations, opencl_dim.py:384-385), written by the last launch of every run().
    # The lattice-unit class (LB_D2Q9.simulation.Simulation) defaults to rebuilding them on demand from the post-collision
    # populations instead (equal up to rounding, <= 3.6e-7; include/lb_hip.h LB_FLAG_EAGER_MACRO); set this attribute to
    # False on a subclass or an instance before init_hip() to get that behaviour here.
    eager_macro = True

    def _engine(self):
        rin, rout = self._boundary_densities()
        return Simulation(self.nx, self.ny, self.omega, bc='pipe', inlet_rho=rin, outlet_rho=rout, device=self.device,
                          eager_macro=self.eager_macro)

    def init_hip(self):
        """Replaces init_opencl (:203-242): report the HIP devices and create the engine handle."""
        ndev = _native.device_count()
        self._say('HIP devices visible:', ndev, '- using device', self.device)
        self._sim = self._engine()

    init_opencl = init_hip

    def allocate_constants(self):
        """The lattice constants are compiled into the kernels; nothing to allocate (:244-255)."""

    def init_hydro(self):
        """Linear density ramp from inlet to outlet, fluid at rest (:258-293)."""
        self.inlet_rho, self.outlet_rho = self._boundary_densities()
        self._say('inlet rho:', self.inlet_rho)
        self._say('outlet rho:', self.outlet_rho)
        i = np.arange(self.nx, dtype=np.float64)[:, None]
        ramp = self.inlet_rho - i * (self.inlet_rho - self.outlet_rho) / float(self.nx)
        rho_host = np.asfortranarray(np.broadcast_to(ramp, (self.nx, self.ny)).astype(np.float32))
        zero = np.zeros((self.nx, self.ny), np.float32, order='F')
        self._sim.set_fields(rho_host, zero, zero)

    def update_feq(self):
        self._sim.update_feq()

    def init_pop(self, amplitude=.001):
        """f = f_streamed = feq (1 + amplitude N(0,1)) per population, drawn from numpy's global RNG
        like the reference (:308-327); amplitude=0 gives the unperturbed equilibrium."""
        perturb = (1. + amplitude * np.random.randn(self.nx, self.ny, NUM_JUMPERS)) if amplitude else None
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
        """num_iterations time steps: move, move_bcs, update_hydro, update_feq, collide_particles
        (:372-387), fused into one HIP launch per step."""
        self._sim.run(num_iterations)

    def step(self):
        self._sim.run(1)

    def check(self, **kw):
        """Device-side health check (Simulation.check): non-finite cells, max Mach number, total mass."""
        return self._sim.check(**kw)

    # ---- state I/O (new: the reference has no checkpointing) ----------------------------------------
    def save_checkpoint(self, path):
        self._sim.save_checkpoint(path)

    def load_checkpoint(self, path):
        self._sim.load_checkpoint(path)

    # ---- read-back (:390-438) -------------------------------------------------------------------
    def get_fields(self):
        return self._sim.get_fields()

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
    """Pipe flow past a cylinder (any obstacle, really: assign ``obstacle_mask_host`` and call
    ``init_hydro(); update_feq(); init_pop()`` as docs/cs205_movie.ipynb:292-295 does)."""

    def __init__(self, cylinder_center=None, cylinder_radius=None, **kwargs):
        assert cylinder_center is not None
        assert cylinder_radius is not None
        self.phys_cylinder_center = cylinder_center
        self.phys_cylinder_radius = cylinder_radius
        self.obstacle_mask_host = None
        self.obstacle_mask = None
        super(Pipe_Flow_Cylinder, self).__init__(**kwargs)

    def set_characteristic_length_time(self):
        """L = cylinder radius, T = sqrt(L / (|grad P| / rho)) (opencl_dim.py:447-456)."""
        self.L = self.phys_cylinder_radius
        self.T = np.sqrt(self.phys_cylinder_radius / (np.abs(self.phys_pressure_grad) / self.phys_rho))

    def initialize_grid_dims(self):
        """Grid from pipe length and diameter in units of the radius; disc of N cells radius (:458-475)."""
        self.lx = int(np.ceil((self.phys_pipe_length / self.L) * self.N))
        self.ly = int(np.ceil((self.phys_diameter / self.L) * self.N))
        self.nx, self.ny = self.lx + 1, self.ly + 1
        self.obstacle_mask_host = np.zeros((self.nx, self.ny), dtype=np.int32, order='F')
        xs, ys = disc_pixels(self.N * self.phys_cylinder_center[0] / self.L,
                             self.N * self.phys_cylinder_center[1] / self.L, self.N, (self.nx, self.ny))
        self.obstacle_mask_host[xs, ys] = 1

    def init_hydro(self):
        """As the base class, then upload the mask and zero u, v inside it (:495-508)."""
        super(Pipe_Flow_Cylinder, self).init_hydro()
        self._sim.set_obstacle_mask(self.obstacle_mask_host)
        self.obstacle_mask = DeviceField(self, 'mask')
        self._sim.zero_velocity_in_obstacle()


class Pipe_Flow_PeriodicBC_VelocityInlet(Pipe_Flow):
    """The reference's second rule set (kernels ``move_bcs_PeriodicBC_VelocityInlet`` and
    ``update_hydro_PeriodicBC_VelocityInlet``, D2Q9.cl:263-374): imposed speed ``u_w`` at the inlet and the
    outlet, north/south rows fed from the opposite wall row.  In the reference only
    ``LB_D2Q9/OLD/opencl.py:281-327`` drives these kernels (its callers in ``dimensionless`` are commented
    out); this class puts the same overrides (``move_bcs``, ``init_hydro``: rho=1, u=u_w, v=0,
    ``update_hydro``) on the dimensionless constructor.  ``run`` is fused (one or two time steps per launch)."""

    def __init__(self, u_w=0.1, **kwargs):
        self.u_w = u_w
        self.u_e = u_w
        super(Pipe_Flow_PeriodicBC_VelocityInlet, self).__init__(**kwargs)

    def _engine(self):
        return Simulation(self.nx, self.ny, self.omega, bc='velocity_inlet', inlet_u=self.u_w, outlet_u=self.u_e,
                          device=self.device)

    def init_hydro(self):
        self.inlet_rho, self.outlet_rho = self._boundary_densities()     # kept as attributes; unused by this family
        rho_host = np.ones((self.nx, self.ny), np.float32, order='F')
        u_host = np.asfortranarray((np.ones((self.nx, self.ny)) * self.u_w).astype(np.float32))
        v_host = np.zeros((self.nx, self.ny), np.float32, order='F')
        self._sim.set_fields(rho_host, u_host, v_host)
