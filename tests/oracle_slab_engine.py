"""CPU stand-in for the slab engine interface of LB_D2Q9.slabs (test infrastructure).

It implements the same methods as LB_D2Q9.simulation.Simulation's slab API (step_boundary,
step_interior, step_finish, halo_export, halo_import, set_f, get_fields, sync) on top of the
oracle's C functions, so that the partition / neighbour / exchange logic of DistributedSlab can run
under torch.distributed+gloo on a machine without a GPU.  The product never imports this file.
"""
import ctypes as ct

import numpy as np

from oracle import oracle as O

K_UP, K_DOWN = (2, 5, 6), (4, 7, 8)
# halo buffer layout of include/lb_hip.h: 18 row segments, three rows deep.  This single-step CPU engine
# only consumes the nearest ghost row's cy=+-1 links but produces and accepts the full format.
NORTH_OUT = ((2, -3), (5, -3), (6, -3), (0, -2), (1, -2), (3, -2), (2, -2), (5, -2), (6, -2),
             (0, -1), (1, -1), (2, -1), (3, -1), (4, -1), (5, -1), (6, -1), (7, -1), (8, -1))      # rows from H
SOUTH_OUT = ((0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (7, 0), (8, 0),
             (0, 1), (1, 1), (3, 1), (4, 1), (7, 1), (8, 1), (4, 2), (7, 2), (8, 2))
NSEG = 18


class OracleSlabEngine(object):
    def __init__(self, nx, ny, omega, bc="pipe", obstacle_mask=None, device=0, y0=0, local_ny=None,
                 halo=True, inlet_rho=1., outlet_rho=1., lid_u=0., rho0=1.):
        self.nx, self.ny, self.y0 = nx, ny, y0
        self.h = ny if local_ny is None else local_ny
        self.device = device
        self.bc = {"pipe": O.BC_PIPE, "periodic": O.BC_PERIODIC, "cavity": O.BC_CAVITY}[bc]
        periodic = self.bc == O.BC_PERIODIC
        # a ghost row only where a neighbour exists, so that the array's edge rows are the walls
        self.gs = 1 if (periodic or y0 > 0) else 0
        self.gn = 1 if (periodic or y0 + self.h < ny) else 0
        self.rows = self.h + self.gs + self.gn
        self.omega, self.rin, self.rout, self.lid, self.rho0 = omega, inlet_rho, outlet_rho, lid_u, rho0
        z3 = lambda: np.zeros((9, self.rows, nx), np.float32)
        self.cur, self.new = z3(), None
        self.feq = z3()
        self.rho, self.u, self.v = (np.zeros((self.rows, nx), np.float32) for _ in range(3))
        self.mask = None
        self.set_obstacle_mask(obstacle_mask)

    def set_obstacle_mask(self, obstacle_mask):
        self.mask = None
        if obstacle_mask is not None:
            m = np.zeros((self.rows, self.nx), np.int32)
            m[self.gs:self.gs + self.h] = (np.asarray(obstacle_mask) != 0).T
            self.mask = m

    def sync(self):
        pass

    def set_f(self, f_slab):
        self.cur[:, self.gs:self.gs + self.h, :] = np.asarray(f_slab, np.float32).transpose(2, 1, 0)

    def _step(self):
        L, nx, rows = O.lib(), self.nx, self.rows
        new = self.cur.copy()                           # stale entries are overwritten or never read
        wrap_x = 1 if self.bc == O.BC_PERIODIC else 0
        L.o2_stream(O._f(self.cur), O._f(new), nx, rows, wrap_x, 0)
        if self.bc == O.BC_PIPE:
            L.o2_bc_pipe(O._f(new), np.float32(self.rin), np.float32(self.rout), nx, rows)
        elif self.bc == O.BC_CAVITY:
            L.o2_bc_cavity(O._f(new), np.float32(self.lid), np.float32(self.rho0), nx, rows)
        if self.mask is not None:
            L.o2_bounceback(self.mask.ctypes.data_as(O._ip), O._f(new), nx, rows)
        L.o2_moments(O._f(new), O._f(self.rho), O._f(self.u), O._f(self.v), nx, rows)
        L.o2_feq(O._f(self.feq), O._f(self.rho), O._f(self.u), O._f(self.v), np.float32(O.cs2),
                 np.float32(O.cs22), np.float32(O.two_cs4), nx, rows)
        L.o2_collide(O._f(new), O._f(self.feq), np.float32(self.omega), nx, rows)
        return new

    def step_boundary(self, write_macro=False):
        self.new = self._step()                         # whole slab at once; the split is a GPU concern

    def step_interior(self, write_macro=False):
        pass

    def step_finish(self):
        self.cur, self.new = self.new, None

    def _target(self):
        return self.new if self.new is not None else self.cur

    @staticmethod
    def _as_array(buf, n):
        if isinstance(buf, np.ndarray):
            return buf.reshape(NSEG, n)
        return np.ctypeslib.as_array((ct.c_float * (NSEG * n)).from_address(int(buf))).reshape(NSEG, n)

    def halo_export(self, side, buf):
        a, out = self._target(), self._as_array(buf, self.nx)
        for i, (k, row) in enumerate(NORTH_OUT if side else SOUTH_OUT):
            out[i] = a[k, self.gs + (self.h if side else 0) + row]

    def halo_import(self, side, buf):
        a, src = self._target(), self._as_array(buf, self.nx)
        if (side == 0 and not self.gs) or (side == 1 and not self.gn):
            return
        # one ghost row per side here: keep the nearest-row segments only
        if side == 0:          # south ghost row -1  <- neighbour's NORTH_OUT entries with row -1
            for i, (k, row) in enumerate(NORTH_OUT):
                if row == -1 and k in K_UP:
                    a[k, 0] = src[i]
        else:                  # north ghost row H   <- neighbour's SOUTH_OUT entries with row 0
            for i, (k, row) in enumerate(SOUTH_OUT):
                if row == 0 and k in K_DOWN:
                    a[k, self.rows - 1] = src[i]

    def check(self, across_ranks=False, warn=False, raise_nonfinite=False):
        """Simulation.check on the host: moments of the current populations of this slab's rows."""
        f = self.cur[:, self.gs:self.gs + self.h].astype(np.float64)
        rho = f.sum(axis=0)
        with np.errstate(all="ignore"):
            ux = (f[1] - f[3] + f[5] - f[6] - f[7] + f[8]) / rho
            uy = (f[5] + f[2] + f[6] - f[7] - f[4] - f[8]) / rho
        ok = np.isfinite(rho) & np.isfinite(ux) & np.isfinite(uy)
        usq = np.where(ok, ux * ux + uy * uy, 0.0)
        return {"n_nonfinite": int((~ok).sum()), "max_mach": float(np.sqrt(3.0 * usq.max())),
                "sum_rho": float(rho[ok].sum())}

    def get_fields(self, which=("f", "u", "v", "rho")):
        sl = slice(self.gs, self.gs + self.h)
        out = {}
        if "f" in which:
            out["f"] = np.asfortranarray(self.cur[:, sl].transpose(2, 1, 0))
        for k in ("rho", "u", "v"):
            if k in which:
                out[k] = np.asfortranarray(getattr(self, k)[sl].T)
        return out
