"""Generic D2Q9 BGK lattice on one MI355X (or one row slab of a multi-GPU run).

``Simulation`` owns a ``liblbhip`` handle and mirrors the method surface of the reference's
``LB_D2Q9.dimensionless.opencl_dim.Pipe_Flow`` (opencl_dim.py:295-438): ``move``, ``move_bcs``,
``update_hydro``, ``update_feq``, ``collide_particles``, ``run(n)``, ``get_fields()``, plus
``step()`` == ``run(1)``.  It takes lattice parameters directly (nx, ny, omega, boundary family);
the physical-units constructors live in ``LB_D2Q9.dimensionless.hip_dim``.

Host arrays use the reference's conventions: F-ordered float32 ``(nx, ny)`` / ``(nx, ny, 9)``
(opencl_dim.py:165, 279, 390-415); for a row slab ``ny`` is the slab height.
"""
import ctypes as ct
import os

import numpy as np

from . import _native
from ._native import LbError, check

NUM_JUMPERS = 9


def _f_order(a, shape, dtype=np.float32):
    a = np.asarray(a)
    if a.shape != shape:
        raise ValueError("expected shape %s, got %s" % (shape, a.shape))
    return np.asfortranarray(a.astype(dtype, copy=False))


def _address(buf):
    if isinstance(buf, np.ndarray):
        assert buf.dtype == np.float32 and buf.flags.c_contiguous
        return buf.ctypes.data
    return int(buf)


def run_group(sims, num_iterations, wait=True):
    """Advance slab handles that tile one grid on one device in lock step (lb_run_group)."""
    arr = (ct.c_void_p * len(sims))(*[s._h for s in sims])
    check(_native.lib().lb_run_group(arr, len(sims), int(num_iterations)))
    if wait:
        for s in sims:
            s.sync()


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 creates it, the caller broadcasts it)."""
    uid = (ct.c_char * 128)()
    check(_native.lib().lb_comm_unique_id(uid))
    return bytes(uid.raw)


class Simulation(object):
    def __init__(self, nx, ny, omega, bc="pipe", inlet_rho=1., outlet_rho=1., lid_u=0., rho0=1.,
                 obstacle_mask=None, device=0, y0=0, local_ny=None, halo=False, semantics="opencl",
                 inlet_u=0., outlet_u=None, planar=None, eager_macro=False):
        """
        :param nx, ny: global grid size (cells, boundary nodes included).
        :param omega: BGK relaxation rate, 0 < omega < 2.
        :param bc: 'pipe' (the reference's pressure inlet/outlet + no-slip walls, D2Q9.cl:173-261),
                   'periodic' or 'cavity' (lid-driven; build-defined, see oracle/d2q9_oracle.c).
        :param inlet_u, outlet_u: bc='velocity_inlet' (D2Q9.cl:263-374; un-fused kernels): imposed speed at
               x=0 / x=nx-1 (outlet_u defaults to inlet_u, as OLD/opencl.py:283-286 sets u_e = u_w).
        :param obstacle_mask: optional (nx, ny) array, non-zero = solid (bounce-back, D2Q9.cl:398-433).
        :param device: HIP device ordinal; -1 (LB_DEVICE_CPU) = the product's own CPU backend: semantics='cython', bc='pipe',
               whole grid, single thread, the reference's mixed float32 / float64 arithmetic (include/lb_hip.h).  Chosen by
               this value only: a GPU handle never falls back to the host.
        :param y0, local_ny: the row slab this object owns (multi-GPU); default = whole grid.
        :param halo: fill the ghost rows through the halo interface even for a whole-grid handle.
        :param semantics: 'opencl' (D2Q9.cl, fused fast path), 'cython' (cython_dim.pyx: pipe family, whole grid,
               compatibility path; the two reference paths differ at walls and inlets) or 'd2q9i' (the reference's
               D2Q9i.cl fork of the OpenCL path: pipe family, whole grid, fused).
        :param planar: device layout of the lattices: False = the nine planes of a row stored together (default),
               True = each plane contiguous (LB_FLAG_PLANAR); results are identical.  None: environment variable
               LB_LAYOUT=planar selects True (tuning aid).
        :param eager_macro: the last launch of every run() stores rho, u, v itself (LB_FLAG_EAGER_MACRO).  Default: the
               plain families leave them to be rebuilt from the populations -- whose moments they are, BGK relaxation
               conserving both -- the first time get_fields / update_feq / ... needs them (include/lb_hip.h).
        """
        if isinstance(bc, str):
            if bc not in _native.BC_NAMES:
                raise ValueError("bc must be one of %s" % sorted(_native.BC_NAMES))
            bc = _native.BC_NAMES[bc]
        self.nx, self.ny = int(nx), int(ny)
        self.y0 = int(y0)
        self.local_ny = self.ny if local_ny is None else int(local_ny)
        self.omega = omega
        self.bc_mode = bc
        self.inlet_rho, self.outlet_rho = inlet_rho, outlet_rho
        self.lid_u, self.rho0 = lid_u, rho0
        self.device = int(device)
        self._lib = _native.lib()            # raises if liblbhip.so is missing: no CPU fallback
        p = _native.LbParams()
        p.nx, p.ny, p.y0, p.local_ny = self.nx, self.ny, self.y0, self.local_ny
        p.bc_mode, p.device = bc, self.device
        if planar is None:
            planar = os.environ.get("LB_LAYOUT", "") == "planar"
        self.planar = bool(planar)
        self.eager_macro = bool(eager_macro)
        p.flags = ((_native.LB_FLAG_HALO if halo else 0) | (_native.LB_FLAG_PLANAR if self.planar else 0) |
                   (_native.LB_FLAG_EAGER_MACRO if self.eager_macro else 0))
        sem = {"opencl": _native.LB_SEM_OPENCL, "cython": _native.LB_SEM_CYTHON, "d2q9i": _native.LB_SEM_OPENCL_D2Q9I}
        if semantics not in sem:
            raise ValueError("semantics must be one of %s" % sorted(sem))
        p.semantics = sem[semantics]
        self.semantics = semantics
        self._halo = bool(halo)
        p.omega = np.float32(omega)
        p.inlet_rho, p.outlet_rho = np.float32(inlet_rho), np.float32(outlet_rho)
        p.lid_u, p.rho0 = np.float32(lid_u), np.float32(rho0)
        self.inlet_u, self.outlet_u = inlet_u, (inlet_u if outlet_u is None else outlet_u)
        p.inlet_u, p.outlet_u = np.float32(self.inlet_u), np.float32(self.outlet_u)
        self._h = ct.c_void_p()
        check(self._lib.lb_create(ct.byref(p), ct.byref(self._h)))
        if self.device == _native.LB_DEVICE_CPU:
            # the CPU backend keeps the reference's float64 scalars (lb_params carries float32: include/lb_hip.h)
            check(self._lib.lb_set_params_f64(self._h, float(omega), float(inlet_rho), float(outlet_rho)))
        self._shape2 = (self.nx, self.local_ny)
        self._shape3 = (self.nx, self.local_ny, NUM_JUMPERS)
        self._mask_host = None
        self._mask_halo_host = None      # (south_rows, north_rows) as last given to set_obstacle_mask_halo
        if obstacle_mask is not None:
            self.set_obstacle_mask(obstacle_mask)

    # -- lifetime ----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.lb_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._lib.lb_sync(self._h))

    # -- state -------------------------------------------------------------
    def set_obstacle_mask(self, mask):
        """(nx, local_ny) array; cells equal to 1/True are solid.  None removes the obstacle."""
        if mask is None:
            check(self._lib.lb_set_mask(self._h, None))
            self._mask_host = None
            return
        m = _f_order(np.asarray(mask) != 0, self._shape2, np.int32)
        self._mask_host = m
        check(self._lib.lb_set_mask(self._h, m.ctypes.data))
        if self._halo and self.local_ny == self.ny and self.bc_mode == _native.LB_BC_PERIODIC:
            # a whole periodic grid run through the halo path is its own neighbour
            rows = np.arange(self.MASK_HALO_ROWS)
            self.set_obstacle_mask_halo(m[:, (rows - self.MASK_HALO_ROWS) % self.ny].T, m[:, rows % self.ny].T)

    def set_fields(self, rho, u, v):
        """Upload the macroscopic fields (what init_hydro does, opencl_dim.py:291-293)."""
        r, uu, vv = (_f_order(a, self._shape2) for a in (rho, u, v))
        check(self._lib.lb_set_macro(self._h, r.ctypes.data, uu.ctypes.data, vv.ctypes.data))

    def set_f(self, f):
        """Upload the populations, (nx, local_ny, 9); also fills the streaming buffer."""
        ff = _f_order(f, self._shape3)
        check(self._lib.lb_set_f(self._h, ff.ctypes.data))

    def init_pop(self, perturb=None):
        """f = f_streamed = feq * perturb (opencl_dim.py:308-327); perturb None = exactly feq."""
        if perturb is None:
            check(self._lib.lb_init_pop(self._h))
            self.sync()
        else:
            f = self.get_fields(("feq",))["feq"]
            f *= np.asarray(perturb)
            self.set_f(f)

    def init_equilibrium(self, rho, u, v, perturb=None):
        """rho,u,v -> feq -> f = feq * perturb (the construction sequence of opencl_dim.py:163-178)."""
        self.set_fields(rho, u, v)
        self.update_feq()
        self.init_pop(perturb)

    # -- the reference's phases, one kernel each (slow path, API parity) -----
    def move(self):
        check(self._lib.lb_move(self._h))
        self.sync()

    def move_bcs(self):
        check(self._lib.lb_move_bcs(self._h))
        self.sync()

    def update_hydro(self):
        check(self._lib.lb_update_hydro(self._h))
        self.sync()

    def update_feq(self):
        check(self._lib.lb_update_feq(self._h))
        self.sync()

    def collide_particles(self):
        check(self._lib.lb_collide_particles(self._h))
        self.sync()

    def zero_velocity_in_obstacle(self):
        check(self._lib.lb_zero_velocity_in_obstacle(self._h))
        self.sync()

    # -- the hot path --------------------------------------------------------
    def run(self, num_iterations, wait=True):
        """num_iterations fused time steps.  The reference returns with the work complete (it waits after
        every kernel); pass wait=False to only enqueue (never blocks the host).  A blocking run of at least four
        times the tuning pass (4 x 361 + 7 = 1451 steps; 4 x 889 + 7 on grids <= 768^2) first times the candidate kernel
        configurations on its own first steps (lb_autotune_quick: they are bitwise equivalent, the trajectory is
        unchanged) and keeps the fastest for this grid; shorter runs use the size heuristic (or call autotune())."""
        n = int(num_iterations)
        if wait and n > 0:
            # (the pass costs 361 steps, 889 on grids <= 768^2, some of them in configurations several times slower than
            #  the best: it only pays for itself in a run several times that long)
            used = self._lib.lb_autotune_quick(self._h, (n - 7) // 4)
            if used < 0:
                check(used)
            n -= used
        check(self._lib.lb_run(self._h, n))
        if wait:
            self.sync()

    def step(self):
        self.run(1)

    # -- health -----------------------------------------------------------------
    MACH_TOLERANCE = 0.1        # the forks' default `mach_tolerance` (porous_media/single_component.py:254)

    def check(self, across_ranks=False, warn=False, raise_nonfinite=False):
        """One device pass over the populations (lb_check): {'n_nonfinite': cells whose density or velocity is not
        finite, 'max_mach': max |u| / c_s, 'sum_rho': total mass of the finite cells}.  The reference's forks have the
        two halves of this as `check_max_ulb` (a warning when max |u| > c_s * mach_tolerance,
        porous_media/single_component.py:221-225) and `check_fields()` (:753-766); `warn=True` issues that warning,
        `raise_nonfinite=True` raises FloatingPointError when the run has diverged.  across_ranks: combined over the
        ranks of the communicator (collective)."""
        n, m, r = ct.c_int64(), ct.c_float(), ct.c_double()
        check(self._lib.lb_check(self._h, int(bool(across_ranks)), ct.byref(n), ct.byref(m), ct.byref(r)))
        out = {"n_nonfinite": n.value, "max_mach": m.value, "sum_rho": r.value}
        if raise_nonfinite and n.value:
            raise FloatingPointError("lattice has %d non-finite cell(s): the run has diverged" % n.value)
        if warn and m.value > self.MACH_TOLERANCE:
            import warnings
            warnings.warn("max_ulb is greater than cs/10! Ma= %g" % m.value, RuntimeWarning, stacklevel=2)
        return out

    def timer_start(self):
        """Record the start HIP event on the engine's stream."""
        check(self._lib.lb_timer_start(self._h))

    def timer_stop(self):
        """Record the stop event, wait for it, return the milliseconds since timer_start()."""
        ms = ct.c_float()
        check(self._lib.lb_timer_stop(self._h, ct.byref(ms)))
        return ms.value

    def timed_run(self, num_iterations):
        """run() bracketed by HIP events on the engine's stream; returns milliseconds."""
        self.timer_start()
        check(self._lib.lb_run(self._h, int(num_iterations)))
        return self.timer_stop()

    # -- read-back -------------------------------------------------------------
    def get_fields(self, which=("f", "feq", "u", "v", "rho")):
        """Dictionary of host copies, same keys/shapes/orders as opencl_dim.py:390-415."""
        out = {}
        if "f" in which:
            out["f"] = np.zeros(self._shape3, np.float32, order="F")
            check(self._lib.lb_get_f(self._h, out["f"].ctypes.data))
        if "feq" in which:
            out["feq"] = np.zeros(self._shape3, np.float32, order="F")
            check(self._lib.lb_get_feq(self._h, out["feq"].ctypes.data))
        macro = [k for k in ("rho", "u", "v") if k in which]
        if macro:
            for k in macro:
                out[k] = np.zeros(self._shape2, np.float32, order="F")
            ptr = lambda k: out[k].ctypes.data if k in out else None
            check(self._lib.lb_get_macro(self._h, ptr("rho"), ptr("u"), ptr("v")))
        return out

    # -- state I/O (the reference has none: state only leaves through get_fields) --------------
    CHECKPOINT_VERSION = 2
    _CKPT_SCALARS = ("nx", "ny", "y0", "local_ny", "bc_mode", "omega", "inlet_rho", "outlet_rho", "lid_u", "rho0",
                     "inlet_u", "outlet_u")

    @staticmethod
    def _ckpt_path(path):
        """np.savez appends '.npz' to a bare path; np.load does not: use one spelling for both."""
        path = str(path)
        return path if path.endswith(".npz") else path + ".npz"

    def checkpoint_arrays(self):
        """Everything needed to continue this run bit for bit, as a dict of arrays: populations, the macroscopic
        fields of the last step, the obstacle mask, every lattice parameter of the handle (boundary family,
        semantics, imposed speeds, halo flag) and a format version."""
        g = self.get_fields(("f", "rho", "u", "v"))
        d = dict(f=g["f"], rho=g["rho"], u=g["u"], v=g["v"],
                 mask=(self._mask_host if self._mask_host is not None else np.zeros((0, 0), np.int32)),
                 version=self.CHECKPOINT_VERSION, semantics=np.array(self.semantics), halo=int(self._halo))
        for k in self._CKPT_SCALARS:
            d[k] = getattr(self, k)
        if self.bc_mode == _native.LB_BC_VELOCITY_INLET:
            d["corner_state"] = self.get_corner_state()
        if self._mask_halo_host is not None:         # a slab: the neighbours' obstacle rows it was given
            empty = np.zeros((0, 0), np.int32)
            d["mask_halo_south"] = empty if self._mask_halo_host[0] is None else self._mask_halo_host[0]
            d["mask_halo_north"] = empty if self._mask_halo_host[1] is None else self._mask_halo_host[1]
        return d

    def save_checkpoint(self, path):
        """Write checkpoint_arrays() to `path` (.npz appended when missing)."""
        np.savez(self._ckpt_path(path), **self.checkpoint_arrays())

    def _check_compatible(self, d):
        if int(d["version"]) != self.CHECKPOINT_VERSION:
            raise ValueError("checkpoint format %d, this build reads %d" % (int(d["version"]), self.CHECKPOINT_VERSION))
        if (int(d["nx"]), int(d["ny"]), int(d["y0"]), int(d["local_ny"])) != (self.nx, self.ny, self.y0, self.local_ny):
            raise ValueError("checkpoint is for a %dx%d lattice (slab %d+%d), this one is %dx%d (slab %d+%d)"
                             % (int(d["nx"]), int(d["ny"]), int(d["y0"]), int(d["local_ny"]),
                                self.nx, self.ny, self.y0, self.local_ny))
        if int(d["bc_mode"]) != self.bc_mode:
            raise ValueError("checkpoint was written with a different boundary family")
        if str(d["semantics"]) != self.semantics:
            raise ValueError("checkpoint was written with semantics=%r, this lattice has %r" % (str(d["semantics"]), self.semantics))
        for k in ("omega", "inlet_rho", "outlet_rho", "lid_u", "rho0", "inlet_u", "outlet_u"):
            if np.float32(d[k]) != np.float32(getattr(self, k)):
                raise ValueError("checkpoint has %s = %r, this lattice %r" % (k, float(d[k]), getattr(self, k)))

    def load_checkpoint(self, path):
        """Restore a state written by save_checkpoint into this lattice; every parameter must match
        (a different omega or boundary family would silently change the physics)."""
        with np.load(self._ckpt_path(path)) as d:
            self.restore_arrays(d)

    def restore_arrays(self, d):
        self._check_compatible(d)
        self.set_obstacle_mask(d["mask"] if d["mask"].size else None)
        if "mask_halo_south" in d:
            so, no = d["mask_halo_south"], d["mask_halo_north"]
            self.set_obstacle_mask_halo(so if so.size else None, no if no.size else None)
        self.set_fields(d["rho"], d["u"], d["v"])
        self.set_f(d["f"])
        if self.bc_mode == _native.LB_BC_VELOCITY_INLET:
            self.set_corner_state(d["corner_state"])           # (after set_f, which resets them to f's own corners)

    def get_corner_state(self):
        """bc='velocity_inlet': the eight corner links no kernel of that rule set writes (include/lb_hip.h)."""
        out = np.zeros(8, np.float32)
        check(self._lib.lb_get_corner_state(self._h, out.ctypes.data))
        return out

    def set_corner_state(self, values):
        v = np.ascontiguousarray(values, np.float32)
        if v.shape != (8,):
            raise ValueError("corner state = 8 floats")
        check(self._lib.lb_set_corner_state(self._h, v.ctypes.data))

    @classmethod
    def from_checkpoint(cls, path, device=0, eager_macro=False):
        """Build a new Simulation from a checkpoint file.  eager_macro is a property of the handle, not of the state: a
        checkpoint written by a handle that rebuilds rho, u, v on demand holds the moments of the post-collision
        populations, one written by an eager handle the pre-collision ones (equal up to rounding, include/lb_hip.h); the
        populations, and hence every later step, are the same bits either way."""
        with np.load(cls._ckpt_path(path)) as d:
            if int(d["version"]) != cls.CHECKPOINT_VERSION:
                raise ValueError("checkpoint format %d, this build reads %d" % (int(d["version"]), cls.CHECKPOINT_VERSION))
            sim = cls(int(d["nx"]), int(d["ny"]), float(d["omega"]), bc=int(d["bc_mode"]),
                      inlet_rho=float(d["inlet_rho"]), outlet_rho=float(d["outlet_rho"]), lid_u=float(d["lid_u"]),
                      rho0=float(d["rho0"]), device=device, y0=int(d["y0"]), local_ny=int(d["local_ny"]),
                      halo=bool(int(d["halo"])), semantics=str(d["semantics"]),
                      inlet_u=float(d["inlet_u"]), outlet_u=float(d["outlet_u"]), eager_macro=eager_macro)
            sim.restore_arrays(d)
        return sim

    # -- row-slab stepping (driven by LB_D2Q9.slabs) -------------------------------
    def step_boundary(self, write_macro=False):
        check(self._lib.lb_step_boundary(self._h, int(bool(write_macro))))

    def step_interior(self, write_macro=False):
        check(self._lib.lb_step_interior(self._h, int(bool(write_macro))))

    def step_finish(self):
        check(self._lib.lb_step_finish(self._h))

    HALO_SEGMENTS = 18

    def halo_floats(self):
        """Length of one halo buffer in floats (18 row segments of nx, three rows deep)."""
        return self.HALO_SEGMENTS * self.nx

    MASK_HALO_ROWS = _native.LB_MASK_HALO_ROWS

    def set_obstacle_mask_halo(self, south_rows=None, north_rows=None):
        """Mask rows of the neighbouring slabs next to this one: south_rows = global rows y0-MASK_HALO_ROWS .. y0-1
        (nearest last), north_rows = rows y0+H .. y0+H+MASK_HALO_ROWS-1 (nearest first), each (MASK_HALO_ROWS, nx),
        non-zero = solid; None = no solid cells."""
        rows = []
        for r in (south_rows, north_rows):
            if r is not None and np.asarray(r).shape != (self.MASK_HALO_ROWS, self.nx):
                raise ValueError("mask halo must have shape (%d, nx)" % self.MASK_HALO_ROWS)
            rows.append(None if r is None else np.ascontiguousarray((np.asarray(r) != 0).astype(np.int32)))
        ptr = lambda a: None if a is None else a.ctypes.data
        check(self._lib.lb_set_mask_halo(self._h, ptr(rows[0]), ptr(rows[1])))
        self._mask_halo_host = tuple(rows)

    def halo_export(self, side, buf):
        """Copy the halo leaving through edge `side` (0 south, 1 north) into buf[18*nx]
        (numpy float32 array or a raw host/device address); layout: include/lb_hip.h."""
        check(self._lib.lb_halo_export(self._h, int(side), _address(buf)))

    def halo_import(self, side, buf):
        check(self._lib.lb_halo_import(self._h, int(side), _address(buf)))

    def use_stream(self, hip_stream):
        """Run on an external hipStream_t (integer handle, e.g. torch's current stream); None = own."""
        check(self._lib.lb_set_stream(self._h, hip_stream))

    def comm_init(self, unique_id, rank, nranks):
        """Attach an RCCL communicator: run() then exchanges halos itself (lb_comm_init)."""
        check(self._lib.lb_comm_init(self._h, unique_id, int(rank), int(nranks)))

    def peer_export(self):
        """This handle's descriptor for the peer transport (lb_peer_export): IPC handles of its lattices and flag block,
        LB_PEER_HANDLE_BYTES bytes to be carried to the two neighbouring ranks."""
        buf = (ct.c_char * _native.LB_PEER_HANDLE_BYTES)()
        check(self._lib.lb_peer_export(self._h, buf))
        return bytes(buf.raw)

    def peer_connect(self, rank, nranks, south, north, min_h):
        """Map the neighbours' descriptors (bytes from their peer_export(); None at a wall) and switch run() on this slab
        to the peer transport: halo rows stored straight into the neighbours' ghost rows (lb_peer_connect)."""
        check(self._lib.lb_peer_connect(self._h, int(rank), int(nranks), south, north, int(min_h)))

    # -- tuning / introspection ------------------------------------------------
    def copy_calibration(self, iters=10, nontemporal=False):
        """Time `iters` plain float4 copies of one lattice into the other (known bytes).
        Returns (GB/s, bytes per launch).  Only between steps: it overwrites the scratch lattice."""
        nbytes, ms = ct.c_int64(), ct.c_float()
        check(self._lib.lb_copy_calibration(self._h, int(nontemporal), ct.byref(nbytes)))
        check(self._lib.lb_timer_start(self._h))
        for _ in range(iters):
            check(self._lib.lb_copy_calibration(self._h, int(nontemporal), ct.byref(nbytes)))
        check(self._lib.lb_timer_stop(self._h, ct.byref(ms)))
        return nbytes.value * iters / (ms.value * 1e-3) / 1e9, nbytes.value

    def set_variant(self, variant):
        check(self._lib.lb_set_variant(self._h, int(variant)))

    def set_slab_cycle(self, depth):
        """Slab handles: depth of the fused kernel the halo cycle runs on (0 = automatic, 3 ... 7; 8 = seven steps by k_deep2, two waves
        per SIMD); every rank of a run must set the same value (lb_set_slab_cycle)."""
        check(self._lib.lb_set_slab_cycle(self._h, int(depth)))

    def set_exchange_inline(self, on):
        """Slab handles: the halo exchange of a cycle on the compute stream, between the interior launches (True), instead of beside
        them on the communication stream (False, the default); every rank of a run must set the same value (lb_set_exchange_inline)."""
        check(self._lib.lb_set_exchange_inline(self._h, int(bool(on))))

    def exchange_timing(self, enable=True):
        """Slab handles: time every halo exchange of run() on its stream from now on (lb_exchange_timing)."""
        check(self._lib.lb_exchange_timing(self._h, int(bool(enable))))

    def exchange_stats(self):
        """The exchanges timed since the last call: {"n", "total_ms", "max_ms", "cycle_depth", "band_rows"} (lb_exchange_stats)."""
        n, tot, mx, d, b = ct.c_int64(), ct.c_double(), ct.c_double(), ct.c_int(), ct.c_int()
        check(self._lib.lb_exchange_stats(self._h, ct.byref(n), ct.byref(tot), ct.byref(mx), ct.byref(d), ct.byref(b)))
        return {"n": n.value, "total_ms": tot.value, "max_ms": mx.value, "cycle_depth": d.value, "band_rows": b.value}

    def autotune(self):
        """Time the candidate fused-kernel configurations on a few live steps and keep the fastest for this
        grid (bitwise-equivalent candidates).  Returns the number of time steps the simulation advanced."""
        n = self._lib.lb_autotune(self._h)
        if n < 0:
            check(n)
        return n

    def hot_kernel(self):
        """Name of the kernel run() spends its time in for this grid / variant / tuning (lb_hot_kernel)."""
        buf = ct.create_string_buffer(256)
        check(self._lib.lb_hot_kernel(self._h, buf, len(buf)))
        return buf.value.decode()

    def plan_launches(self, num_iterations):
        """The time steps of each launch run(num_iterations) would make, in order (whole-grid OpenCL-path GPU handles; None
        elsewhere): every launch of a marching kernel moves the same bytes whatever number of steps it fuses."""
        buf = (ct.c_int * 256)()
        n = self._lib.lb_plan_launches(self._h, int(num_iterations), buf, 256)
        if n < 0:
            return None
        return [int(buf[i]) for i in range(min(n, 256))]

    def steps_per_launch(self):
        """Time steps one launch of run()'s hot kernel advances for this grid / variant / tuning: 4, 3, 2 or 1."""
        n = self._lib.lb_steps_per_launch(self._h)
        if n < 0:
            check(n)
        return n

    def layout(self):
        a, b, c = ct.c_int64(), ct.c_int64(), ct.c_int64()
        check(self._lib.lb_layout(self._h, ct.byref(a), ct.byref(b), ct.byref(c)))
        return {"pitch": a.value, "plane_stride": b.value, "bytes": c.value, "planar": self.planar}
