"""ctypes front-end of the CPU oracle (oracle/d2q9_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the product package (2d-lb_amd/).

Two classes restate the host side of the two reference paths:

* ``O2Sim``  - the OpenCL path: LB_D2Q9/dimensionless/opencl_dim.py:58-518
  (parameter derivation :86-134, ``init_hydro`` :258-293, ``run`` :372-387)
  driving the C restatement of LB_D2Q9/D2Q9.cl.
* ``O1Sim``  - the Cython path: LB_D2Q9/dimensionless/cython_dim.pyx:31-513.

Array conventions follow the reference: O2 fields are logically (nx, ny[, 9])
Fortran-ordered (memory = C-ordered (9, ny, nx)); O1 fields are C-ordered
(9, nx, ny) / (nx, ny).
"""
import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libd2q9_oracle.so")
_LIB_OMP_PATH = os.path.join(_HERE, "_build", "libd2q9_oracle_omp.so")   # same source, -fopenmp (CPU baseline only)

# D2Q9 constants exactly as the reference forms them (opencl_dim.py:22-36)
cs = 1.0 / np.sqrt(3)
cs2 = cs ** 2
cs22 = 2 * cs2
two_cs4 = 2 * cs ** 4
CX = np.array([0, 1, 0, -1, 0, 1, -1, -1, 1], dtype=np.int32)
CY = np.array([0, 0, 1, 0, -1, 1, 1, -1, -1], dtype=np.int32)
W = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4, dtype=np.float32)

BC_PIPE, BC_PERIODIC, BC_CAVITY, BC_VELOCITY_INLET = 0, 1, 2, 3


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "d2q9_oracle.c")
    stale = lambda p: not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src)
    if force or stale(_LIB_PATH) or stale(_LIB_OMP_PATH):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_fp = ct.POINTER(ct.c_float)
_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int32)
_bp = ct.POINTER(ct.c_uint8)


class _O2State(ct.Structure):
    _fields_ = [("nx", ct.c_int32), ("ny", ct.c_int32), ("bc_mode", ct.c_int32), ("_pad", ct.c_int32),
                ("omega", ct.c_float), ("rho_in", ct.c_float), ("rho_out", ct.c_float),
                ("lid_u", ct.c_float), ("rho0", ct.c_float), ("u_w", ct.c_float), ("u_e", ct.c_float),
                ("cs2", ct.c_float), ("two_cs2", ct.c_float), ("two_cs4", ct.c_float),
                ("f", _fp), ("fs", _fp), ("feq", _fp), ("rho", _fp), ("u", _fp), ("v", _fp),
                ("mask", _ip), ("d2q9i", ct.c_int32), ("_pad2", ct.c_int32)]


class _O1State(ct.Structure):
    _fields_ = [("nx", ct.c_int32), ("ny", ct.c_int32), ("numpy2", ct.c_int32), ("_pad", ct.c_int32),
                ("omega", ct.c_double), ("rho_in", ct.c_double), ("rho_out", ct.c_double),
                ("f", _fp), ("feq", _fp), ("rho", _fp), ("u", _dp), ("v", _dp), ("mask", _bp)]


_lib = None
_lib_omp = None


def lib_omp():
    """The -fopenmp build of the same source: o1_run for timing the Cython-path port on all host cores (bench.py
    cpu_baseline), o2_run so that the full-size GPU tests can hold an 8192^2 run to the oracle in seconds (the loops that
    carry a pragma are independent per cell: same bits as the serial build, tests/test_oracle_golden.py)."""
    global _lib_omp
    if _lib_omp is None:
        build()
        L = ct.CDLL(_LIB_OMP_PATH)
        L.o1_run.argtypes = [ct.POINTER(_O1State), ct.c_int]
        L.o1_run.restype = None
        L.o2_run.argtypes = [ct.POINTER(_O2State), ct.c_int]
        L.o2_run.restype = None
        _lib_omp = L
    return _lib_omp


def lib():
    global _lib
    if _lib is None:
        build()
        L = ct.CDLL(_LIB_PATH)
        for name in ("o2_phase_move", "o2_phase_bcs"):
            getattr(L, name).argtypes = [ct.POINTER(_O2State)]
            getattr(L, name).restype = None
        L.o2_run.argtypes = [ct.POINTER(_O2State), ct.c_int]
        L.o2_run.restype = None
        L.o2_stream.argtypes = [_fp, _fp, ct.c_int, ct.c_int, ct.c_int, ct.c_int]
        L.o2_copy.argtypes = [_fp, _fp, ct.c_int, ct.c_int]
        L.o2_bc_pipe.argtypes = [_fp, ct.c_float, ct.c_float, ct.c_int, ct.c_int]
        L.o2_bc_cavity.argtypes = [_fp, ct.c_float, ct.c_float, ct.c_int, ct.c_int]
        L.o2i_bc_pipe.argtypes = [_fp, ct.c_float, ct.c_float, ct.c_int, ct.c_int]
        L.o2i_moments.argtypes = [_fp, _fp, _fp, _fp, ct.c_int, ct.c_int]
        L.o2i_feq.argtypes = [_fp, _fp, _fp, _fp, ct.c_int, ct.c_int]
        L.o2_bc_velocity_inlet.argtypes = [_fp, ct.c_float, ct.c_float, ct.c_int, ct.c_int]
        L.o2_moments_velocity_inlet.argtypes = [_fp, _fp, _fp, _fp, ct.c_float, ct.c_float, ct.c_int, ct.c_int]
        L.o2_bounceback.argtypes = [_ip, _fp, ct.c_int, ct.c_int]
        L.o2_zero_velocity.argtypes = [_ip, _fp, _fp, ct.c_int, ct.c_int]
        L.o2_moments.argtypes = [_fp, _fp, _fp, _fp, ct.c_int, ct.c_int]
        L.o2_feq.argtypes = [_fp, _fp, _fp, _fp, ct.c_float, ct.c_float, ct.c_float, ct.c_int, ct.c_int]
        L.o2_collide.argtypes = [_fp, _fp, ct.c_float, ct.c_int, ct.c_int]
        for name in ("o1_move_bcs", "o1_move", "o1_update_hydro", "o1_update_feq", "o1_collide"):
            getattr(L, name).argtypes = [ct.POINTER(_O1State)]
            getattr(L, name).restype = None
        L.o1_run.argtypes = [ct.POINTER(_O1State), ct.c_int]
        L.o1_run.restype = None
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(_fp)


def disc_mask(nx, ny, xc, yc, radius):
    """Pixels of ``skimage.draw.circle(xc, yc, radius)`` (scikit-image, un-pinned
    and absent from the reference tree: setup.py:29): the strict-inequality disc
    (x-xc)^2 + (y-yc)^2 < radius^2, clipped to the array.  Returns bool (nx, ny)."""
    x = np.arange(nx, dtype=np.float64)[:, None]
    y = np.arange(ny, dtype=np.float64)[None, :]
    return ((x - xc) ** 2 + (y - yc) ** 2) < float(radius) ** 2


def opencl_pipe_parameters(diameter, rho, viscosity, pressure_grad, pipe_length, N=200,
                           time_prefactor=1., cylinder_radius=None):
    """Physical -> lattice parameters of the OpenCL class (opencl_dim.py:86-134,
    180-201, 266-273; cylinder variant :447-460)."""
    p = {}
    zeta = np.abs(pressure_grad) / rho
    if cylinder_radius is None:
        p["L"] = diameter
        p["T"] = np.sqrt(diameter / zeta)
    else:
        p["L"] = cylinder_radius
        p["T"] = np.sqrt(cylinder_radius / zeta)
    p["W"] = (np.abs(pressure_grad / rho) * p["L"] * p["T"]) / viscosity
    p["N"] = N
    p["delta_x"] = 1. / N
    p["delta_t"] = time_prefactor * p["delta_x"] ** 2
    p["ulb"] = p["delta_t"] / p["delta_x"]
    p["lb_viscosity"] = (p["delta_t"] / p["delta_x"] ** 2) * (1. / p["W"])
    p["omega"] = (3 * p["lb_viscosity"] + 0.5) ** -1.
    p["lx"] = int(np.ceil((pipe_length / p["L"]) * N))
    p["ly"] = N if cylinder_radius is None else int(np.ceil((diameter / p["L"]) * N))
    p["nx"], p["ny"] = p["lx"] + 1, p["ly"] + 1
    delta_rho = p["nx"] * (p["delta_t"] ** 2 / p["delta_x"]) * (1. / cs2) * 1.
    p["outlet_rho"] = 1.
    p["inlet_rho"] = 1. + np.abs(delta_rho)
    return p


def cython_pipe_parameters(diameter, rho, viscosity, pressure_grad, pipe_length, N=100,
                           time_prefactor=1., cylinder_radius=None):
    """Physical -> lattice parameters of the Cython class (cython_dim.pyx:56-94,
    138-147; cylinder variant :405-420)."""
    p = {}
    if cylinder_radius is None:
        p["L"] = diameter
        p["T"] = (8 * rho * viscosity) / (np.abs(pressure_grad) * p["L"])
    else:
        p["L"] = cylinder_radius
        p["T"] = (8 * rho * viscosity * p["L"]) / (np.abs(pressure_grad) * diameter ** 2)
    p["Re"] = p["L"] ** 2 / (viscosity * p["T"] ** 2)
    p["N"] = N
    p["delta_x"] = 1. / N
    p["delta_t"] = time_prefactor * p["delta_x"] ** 2
    p["lx"] = int(np.ceil((pipe_length / p["L"]) * N))
    p["ly"] = N if cylinder_radius is None else int(np.ceil((diameter / p["L"]) * N))
    p["nx"], p["ny"] = p["lx"] + 1, p["ly"] + 1
    p["lb_viscosity"] = (p["delta_t"] / p["delta_x"] ** 2) * (1. / p["Re"])
    p["omega"] = (p["lb_viscosity"] / cs2 + 0.5) ** -1.
    nondim_deltaP = (p["T"] ** 2 / (rho * p["L"])) * pressure_grad
    delta_rho = p["nx"] * (p["delta_t"] ** 2 / p["delta_x"]) * (1. / cs2) * nondim_deltaP
    p["outlet_rho"] = 1.
    p["inlet_rho"] = 1. + np.abs(delta_rho)
    return p


def density_ramp(nx, ny, inlet_rho, outlet_rho):
    """rho[i,:] = rho_in - i (rho_in - rho_out)/nx, float32 (opencl_dim.py:279-283,
    cython_dim.pyx:150-154).  Returned as (nx, ny) C-ordered."""
    i = np.arange(nx, dtype=np.float64)[:, None]
    r = inlet_rho - i * (inlet_rho - outlet_rho) / float(nx)
    return np.ascontiguousarray(np.broadcast_to(r, (nx, ny)).astype(np.float32))


class O2Sim(object):
    """OpenCL-path oracle.  Internal arrays are C-ordered (9, ny, nx) / (ny, nx)
    == the reference's F-ordered (nx, ny, 9) / (nx, ny) buffers."""

    def __init__(self, nx, ny, omega, bc_mode=BC_PIPE, inlet_rho=1., outlet_rho=1.,
                 lid_u=0., rho0=1., mask=None, u_w=0., u_e=None, d2q9i=False):
        self.u_w, self.u_e = u_w, (u_w if u_e is None else u_e)
        self.d2q9i = bool(d2q9i)         # the kernels of D2Q9i.cl instead of D2Q9.cl
        self.nx, self.ny = int(nx), int(ny)
        self.omega = omega
        self.bc_mode = bc_mode
        self.inlet_rho, self.outlet_rho = inlet_rho, outlet_rho
        self.lid_u, self.rho0 = lid_u, rho0
        shp3, shp2 = (9, self.ny, self.nx), (self.ny, self.nx)
        self.f = np.zeros(shp3, np.float32)
        self.fs = np.zeros(shp3, np.float32)
        self.feq = np.zeros(shp3, np.float32)
        self.rho = np.ones(shp2, np.float32)
        self.u = np.zeros(shp2, np.float32)
        self.v = np.zeros(shp2, np.float32)
        self.mask = None
        if mask is not None:
            self.set_mask(mask)

    # -- state -----------------------------------------------------------
    def set_mask(self, mask_xy):
        """mask_xy: (nx, ny) array, non-zero = solid (int32 on the device, opencl_dim.py:468)."""
        m = np.asarray(mask_xy)
        assert m.shape == (self.nx, self.ny)
        self.mask = np.ascontiguousarray((m != 0).astype(np.int32).T)

    def set_macro(self, rho_xy, u_xy, v_xy):
        self.rho[...] = np.asarray(rho_xy, np.float32).T
        self.u[...] = np.asarray(u_xy, np.float32).T
        self.v[...] = np.asarray(v_xy, np.float32).T

    def set_f(self, f_xyk):
        """f_xyk: (nx, ny, 9).  Also fills the streaming buffer (opencl_dim.py:323-327)."""
        self.f[...] = np.asarray(f_xyk, np.float32).transpose(2, 1, 0)
        self.fs[...] = self.f

    def _state(self):
        s = _O2State()
        s.nx, s.ny, s.bc_mode = self.nx, self.ny, self.bc_mode
        s.omega = np.float32(self.omega)
        s.rho_in, s.rho_out = np.float32(self.inlet_rho), np.float32(self.outlet_rho)
        s.lid_u, s.rho0 = np.float32(self.lid_u), np.float32(self.rho0)
        s.u_w, s.u_e = np.float32(self.u_w), np.float32(self.u_e)
        s.cs2, s.two_cs2, s.two_cs4 = np.float32(cs2), np.float32(cs22), np.float32(two_cs4)
        s.f, s.fs, s.feq = _f(self.f), _f(self.fs), _f(self.feq)
        s.rho, s.u, s.v = _f(self.rho), _f(self.u), _f(self.v)
        s.mask = self.mask.ctypes.data_as(_ip) if self.mask is not None else None
        s.d2q9i = int(self.d2q9i)
        return s

    # -- the reference's methods ----------------------------------------
    def move(self):
        lib().o2_phase_move(ct.byref(self._state()))

    def move_bcs(self):
        lib().o2_phase_bcs(ct.byref(self._state()))

    def update_hydro(self):
        if self.bc_mode == BC_VELOCITY_INLET:
            lib().o2_moments_velocity_inlet(_f(self.f), _f(self.rho), _f(self.u), _f(self.v),
                                            np.float32(self.u_w), np.float32(self.u_e), self.nx, self.ny)
        elif self.d2q9i:
            lib().o2i_moments(_f(self.f), _f(self.rho), _f(self.u), _f(self.v), self.nx, self.ny)
            self.zero_velocity_in_obstacle()
        else:
            lib().o2_moments(_f(self.f), _f(self.rho), _f(self.u), _f(self.v), self.nx, self.ny)

    def update_feq(self):
        if self.d2q9i:
            lib().o2i_feq(_f(self.feq), _f(self.rho), _f(self.u), _f(self.v), self.nx, self.ny)
            return
        lib().o2_feq(_f(self.feq), _f(self.rho), _f(self.u), _f(self.v),
                     np.float32(cs2), np.float32(cs22), np.float32(two_cs4), self.nx, self.ny)

    def collide_particles(self):
        lib().o2_collide(_f(self.f), _f(self.feq), np.float32(self.omega), self.nx, self.ny)

    def zero_velocity_in_obstacle(self):
        if self.mask is not None:
            lib().o2_zero_velocity(self.mask.ctypes.data_as(_ip), _f(self.u), _f(self.v), self.nx, self.ny)

    def init_pop(self, perturb_xyk=None):
        """f = feq * perturb (opencl_dim.py:308-327); the reference draws
        perturb = 1 + 0.001 randn(nx,ny,9) from the unseeded global RNG, the
        oracle takes it as an argument (None = no perturbation)."""
        self.f[...] = self.feq
        if perturb_xyk is not None:
            self.f *= np.asarray(perturb_xyk, np.float64).transpose(2, 1, 0)
        self.fs[...] = self.f

    def run(self, n, openmp=False):
        (lib_omp() if openmp else lib()).o2_run(ct.byref(self._state()), int(n))

    def get_fields(self):
        """Same shapes/orders as opencl_dim.py:390-415."""
        return {"f": self.f.transpose(2, 1, 0).copy(order="F"),
                "feq": self.feq.transpose(2, 1, 0).copy(order="F"),
                "u": self.u.T.copy(order="F"), "v": self.v.T.copy(order="F"),
                "rho": self.rho.T.copy(order="F")}

    @classmethod
    def pipe_flow(cls, cylinder_center=None, cylinder_radius=None, perturb=None, **kw):
        """Restates opencl_dim.Pipe_Flow.__init__ / Pipe_Flow_Cylinder.__init__."""
        p = opencl_pipe_parameters(cylinder_radius=cylinder_radius, **kw)
        assert p["omega"] < 2.
        sim = cls(p["nx"], p["ny"], p["omega"], BC_PIPE, p["inlet_rho"], p["outlet_rho"])
        sim.params = p
        if cylinder_radius is not None:
            N, L = p["N"], p["L"]
            sim.set_mask(disc_mask(p["nx"], p["ny"], N * cylinder_center[0] / L,
                                   N * cylinder_center[1] / L, N))
        ramp = density_ramp(p["nx"], p["ny"], p["inlet_rho"], p["outlet_rho"])
        sim.set_macro(ramp, np.zeros_like(ramp), np.zeros_like(ramp))
        sim.zero_velocity_in_obstacle()
        sim.update_feq()
        sim.init_pop(perturb)
        return sim


class O1Sim(object):
    """Cython-path oracle: arrays C-ordered (9, nx, ny) / (nx, ny); u, v float64."""

    def __init__(self, nx, ny, omega, inlet_rho, outlet_rho=1., mask=None, numpy2=True):
        self.nx, self.ny = int(nx), int(ny)
        self.lx, self.ly = self.nx - 1, self.ny - 1
        self.omega, self.inlet_rho, self.outlet_rho = float(omega), float(inlet_rho), float(outlet_rho)
        self.numpy2 = bool(numpy2)
        self.f = np.zeros((9, nx, ny), np.float32)
        self.feq = np.zeros((9, nx, ny), np.float32)
        self.rho = np.ones((nx, ny), np.float32)
        self.u = np.zeros((nx, ny), np.float64)
        self.v = np.zeros((nx, ny), np.float64)
        self.mask = None if mask is None else np.ascontiguousarray(np.asarray(mask) != 0).astype(np.uint8)

    def _state(self):
        s = _O1State()
        s.nx, s.ny, s.numpy2 = self.nx, self.ny, int(self.numpy2)
        s.omega, s.rho_in, s.rho_out = self.omega, self.inlet_rho, self.outlet_rho
        s.f, s.feq, s.rho = _f(self.f), _f(self.feq), _f(self.rho)
        s.u, s.v = self.u.ctypes.data_as(_dp), self.v.ctypes.data_as(_dp)
        s.mask = self.mask.ctypes.data_as(_bp) if self.mask is not None else None
        return s

    def move_bcs(self):
        lib().o1_move_bcs(ct.byref(self._state()))

    def move(self):
        lib().o1_move(ct.byref(self._state()))

    def update_hydro(self):
        lib().o1_update_hydro(ct.byref(self._state()))

    def update_feq(self):
        lib().o1_update_feq(ct.byref(self._state()))

    def collide_particles(self):
        lib().o1_collide(ct.byref(self._state()))

    def run(self, n, openmp=False):
        (lib_omp() if openmp else lib()).o1_run(ct.byref(self._state()), int(n))

    def init_pop(self, perturb_xy=None):
        """f = feq * perturb, perturb (nx,ny) shared by the 9 links (cython_dim.pyx:191-202)."""
        self.f[...] = self.feq
        if perturb_xy is not None:
            self.f[...] = (self.f * np.asarray(perturb_xy, np.float64)[None]).astype(np.float32)

    def get_fields(self):
        return {"f": self.f, "feq": self.feq, "u": self.u, "v": self.v, "rho": self.rho}

    @classmethod
    def pipe_flow(cls, cylinder_center=None, cylinder_radius=None, perturb=None, numpy2=True, **kw):
        """Restates cython_dim.Pipe_Flow.__init__ / Pipe_Flow_Cylinder.__init__."""
        p = cython_pipe_parameters(cylinder_radius=cylinder_radius, **kw)
        assert p["omega"] < 2.
        mask = None
        if cylinder_radius is not None:
            N, L = p["N"], p["L"]
            mask = disc_mask(p["nx"], p["ny"], N * cylinder_center[0] / L, N * cylinder_center[1] / L, N)
        sim = cls(p["nx"], p["ny"], p["omega"], p["inlet_rho"], p["outlet_rho"], mask, numpy2)
        sim.params = p
        sim.rho[...] = density_ramp(p["nx"], p["ny"], p["inlet_rho"], p["outlet_rho"])
        sim.update_feq()
        sim.init_pop(perturb)
        return sim
