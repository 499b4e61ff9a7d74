#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN SOURCES.

Runs only in the build container (needs /root/reference, gcc, cython); nothing
here runs on the GPU box and no reference source text is written into the
repository: all compilation happens in a fresh temp dir under /tmp, the only
outputs are numeric fixtures (inputs + expected outputs) under tests/golden/.

Two reference executions are used:

O1  LB_D2Q9/dimensionless/cython_dim.pyx is cythonized (`cython -2`, the
    reference is Python-2 syntax) and imported.  The module imports `skimage`
    (absent here, un-pinned in the reference's setup.py:29); an otherwise empty
    placeholder package satisfies the import, and for the cylinder class its
    `draw.circle` is the published strict-inequality disc.  The obstacle masks
    actually used are stored in the fixtures as data.
O2  LB_D2Q9/D2Q9.cl is OpenCL C and there is no OpenCL device or pyopencl in the
    image.  Its kernels are plain C99 apart from address-space qualifiers and
    the work-item id built-ins, so the file is #included (by absolute path, at
    generation time) behind a dozen #defines and each kernel is invoked once
    per work-item of the NDRange the host class would have launched
    (opencl_dim.py:130-134, 32x32[x1] work-groups).  The host-side launch
    order is restated from opencl_dim.py:372-387 / :510-518.  Compiled with
    -ffp-contract=off.  A real OpenCL device may contract a*b+c to FMA, so O2
    fixtures pin the arithmetic up to that freedom (tolerances in the tests).

Usage:  python oracle/make_golden.py
"""
import ctypes as ct
import importlib
import os
import subprocess
import sys
import sysconfig
import tempfile

import numpy as np

REF = "/root/reference/LB_D2Q9"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

cs = 1.0 / np.sqrt(3)
cs2, cs22, two_cs4 = cs ** 2, 2 * cs ** 2, 2 * cs ** 4
CXv = np.array([0, 1, 0, -1, 0, 1, -1, -1, 1])
CYv = np.array([0, 0, 1, 0, -1, 1, 1, -1, -1])
Wv = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)

# --------------------------------------------------------------------------
#  O2: D2Q9.cl executed work-item by work-item on the CPU
# --------------------------------------------------------------------------
O2_DRIVER = r'''
#include <stddef.h>
static int g_gid[3], g_lid[3], g_lsz[3] = {32, 32, 1};
#define __kernel
#define __global
#define __local
#define __constant const
#define __read_only
#define __write_only
#define CLK_LOCAL_MEM_FENCE 0
static inline int get_global_id(int d)  { return g_gid[d]; }
static inline int get_local_id(int d)   { return g_lid[d]; }
static inline int get_local_size(int d) { return g_lsz[d]; }
static inline void barrier(int flags)   { (void)flags; }
#include "%(cl)s"

static int pad32(int n) { return (n + 31) / 32 * 32; }
/* visit every work-item of the padded NDRange, z planes 0..nz-1 */
#define NDRANGE(nz, CALL)                                                     \
    for (int z = 0; z < (nz); ++z)                                            \
        for (int y = 0; y < pad32(ny); ++y)                                   \
            for (int x = 0; x < pad32(nx); ++x) {                             \
                g_gid[0] = x; g_gid[1] = y; g_gid[2] = z;                     \
                g_lid[0] = x %% 32; g_lid[1] = y %% 32; g_lid[2] = 0;         \
                CALL;                                                         \
            }

static float lu[1024], lv[1024], lr[1024];
static const float Wc[9] = {4.f/9.f, 1.f/9.f, 1.f/9.f, 1.f/9.f, 1.f/9.f, 1.f/36.f, 1.f/36.f, 1.f/36.f, 1.f/36.f};
static const int CXc[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
static const int CYc[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};

void k_update_feq(float *feq, float *u, float *v, float *rho, float cs, float cs2, float cs22,
                  float two_cs4, int nx, int ny)
{ NDRANGE(9, update_feq(feq, u, v, rho, lu, lv, lr, Wc, CXc, CYc, cs, cs2, cs22, two_cs4, nx, ny)) }
void k_update_hydro(float *f, float *u, float *v, float *rho, float rin, float rout, int nx, int ny)
{ NDRANGE(1, update_hydro(f, u, v, rho, rin, rout, nx, ny)) }
void k_collide(float *f, float *feq, float omega, int nx, int ny)
{ NDRANGE(9, collide_particles(f, feq, omega, nx, ny)) }
void k_copy(float *from, float *to, int nx, int ny)
{ NDRANGE(9, copy_buffer(from, to, nx, ny)) }
void k_move(float *f, float *fs, int nx, int ny)
{ NDRANGE(9, move(f, fs, CXc, CYc, nx, ny)) }
void k_move_bcs(float *f, float *u, float rin, float rout, int nx, int ny)
{ NDRANGE(1, move_bcs(f, u, rin, rout, nx, ny)) }
void k_move_bcs_vel(float *f, float *u, float uw, float ue, int nx, int ny)
{ NDRANGE(1, move_bcs_PeriodicBC_VelocityInlet(f, u, uw, ue, nx, ny)) }
void k_update_hydro_vel(float *f, float *u, float *v, float *rho, float uw, float ue, int nx, int ny)
{ NDRANGE(1, update_hydro_PeriodicBC_VelocityInlet(f, u, v, rho, uw, ue, nx, ny)) }
void k_zero_vel(int *mask, float *u, float *v, int nx, int ny)
{ NDRANGE(1, set_zero_velocity_in_obstacle(mask, u, v, nx, ny)) }
void k_bounceback(int *mask, float *f, int nx, int ny)
{ NDRANGE(1, bounceback_in_obstacle(mask, f, nx, ny)) }
'''


def build_o2(tmp, cl_file="D2Q9.cl"):
    src = os.path.join(tmp, "o2_driver_%s.c" % cl_file.replace(".", "_"))
    driver = O2_DRIVER
    if cl_file == "D2Q9i.cl":       # the fork has no velocity-inlet kernels: drop their wrappers
        a, b = driver.index("void k_move_bcs_vel("), driver.index("void k_zero_vel(")
        driver = driver[:a] + driver[b:]
    with open(src, "w") as fh:
        fh.write(driver % {"cl": os.path.join(REF, cl_file)})
    so = os.path.join(tmp, "libo2ref_%s.so" % cl_file.replace(".", "_"))
    subprocess.check_call(["gcc", "-O1", "-std=gnu99", "-ffp-contract=off", "-fPIC", "-shared",
                           "-w", src, "-o", so, "-lm"])
    L = ct.CDLL(so)
    fp, ip, F, I = ct.POINTER(ct.c_float), ct.POINTER(ct.c_int), ct.c_float, ct.c_int
    L.k_update_feq.argtypes = [fp, fp, fp, fp, F, F, F, F, I, I]
    L.k_update_hydro.argtypes = [fp, fp, fp, fp, F, F, I, I]
    L.k_collide.argtypes = [fp, fp, F, I, I]
    L.k_copy.argtypes = [fp, fp, I, I]
    L.k_move.argtypes = [fp, fp, I, I]
    L.k_move_bcs.argtypes = [fp, fp, F, F, I, I]
    L.k_zero_vel.argtypes = [ip, fp, fp, I, I]
    if cl_file != "D2Q9i.cl":
        L.k_move_bcs_vel.argtypes = [fp, fp, F, F, I, I]
        L.k_update_hydro_vel.argtypes = [fp, fp, fp, fp, F, F, I, I]
    L.k_bounceback.argtypes = [ip, fp, I, I]
    return L


def P(a):
    return a.ctypes.data_as(ct.POINTER(ct.c_float if a.dtype == np.float32 else ct.c_int))


class RefOpenCL(object):
    """Host order of opencl_dim.Pipe_Flow(.Cylinder) around the executed kernels.
    Buffers are F-ordered (nx, ny[, 9]) exactly like the reference's host arrays."""

    def __init__(self, L, nx, ny, omega, rin, rout, mask=None):
        self.L, self.nx, self.ny = L, nx, ny
        self.omega, self.rin, self.rout = np.float32(omega), np.float32(rin), np.float32(rout)
        z3 = lambda: np.zeros((nx, ny, 9), np.float32, order="F")
        z2 = lambda: np.zeros((nx, ny), np.float32, order="F")
        self.f, self.fs, self.feq = z3(), z3(), z3()
        self.rho, self.u, self.v = z2(), z2(), z2()
        self.mask = None if mask is None else np.asfortranarray(mask.astype(np.int32))

    def update_feq(self):
        self.L.k_update_feq(P(self.feq), P(self.u), P(self.v), P(self.rho), np.float32(cs),
                            np.float32(cs2), np.float32(cs22), np.float32(two_cs4), self.nx, self.ny)

    def move(self):
        self.L.k_move(P(self.f), P(self.fs), self.nx, self.ny)
        self.L.k_copy(P(self.fs), P(self.f), self.nx, self.ny)

    def move_bcs(self):
        self.L.k_move_bcs(P(self.f), P(self.u), self.rin, self.rout, self.nx, self.ny)
        if self.mask is not None:
            self.L.k_bounceback(P(self.mask), P(self.f), self.nx, self.ny)

    def update_hydro(self):
        self.L.k_update_hydro(P(self.f), P(self.u), P(self.v), P(self.rho), self.rin, self.rout,
                              self.nx, self.ny)

    def collide(self):
        self.L.k_collide(P(self.f), P(self.feq), self.omega, self.nx, self.ny)

    def zero_vel(self):
        if self.mask is not None:
            self.L.k_zero_vel(P(self.mask), P(self.u), P(self.v), self.nx, self.ny)

    def run(self, n):
        for _ in range(n):                       # opencl_dim.py:380-387
            self.move()
            self.move_bcs()
            self.update_hydro()
            self.update_feq()
            self.collide()

    def snap(self):
        return {k: getattr(self, k).copy(order="F") for k in ("f", "feq", "rho", "u", "v")}


def ramp(nx, ny, rin, rout):
    r = np.zeros((nx, ny), np.float32, order="F")
    for i in range(nx):
        r[i, :] = rin - i * (rin - rout) / float(nx)
    return r


def disc(nx, ny, xc, yc, R):
    x = np.arange(nx)[:, None].astype(float)
    y = np.arange(ny)[None, :].astype(float)
    return ((x - xc) ** 2 + (y - yc) ** 2) < R ** 2


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024.))


def flat(prefix, d):
    return {"%s_%s" % (prefix, k): v for k, v in d.items()}


def gen_o2(L):
    # ---- a. one call of every kernel on the same random state -------------
    nx, ny, omega, rin, rout = 37, 19, 1.2, 1.004, 1.0
    rng = np.random.default_rng(2015)
    mask = disc(nx, ny, 12.0, 9.0, 4.0)
    f0 = np.asfortranarray((Wv[None, None, :] * (1 + 0.05 * rng.standard_normal((nx, ny, 9)))).astype(np.float32))
    fs0 = np.asfortranarray(rng.standard_normal((nx, ny, 9)).astype(np.float32))   # stale content of f_streamed
    out = {"nx": nx, "ny": ny, "omega": omega, "inlet_rho": rin, "outlet_rho": rout,
           "mask": mask, "f0": f0, "fs0": fs0}
    s = RefOpenCL(L, nx, ny, omega, rin, rout, mask)
    s.f[...] = f0; s.fs[...] = fs0
    L.k_move(P(s.f), P(s.fs), nx, ny)
    out["after_move_fs"] = s.fs.copy(order="F")
    s.f[...] = f0
    L.k_move_bcs(P(s.f), P(s.u), s.rin, s.rout, nx, ny)
    out["after_bcs_f"] = s.f.copy(order="F")
    s.f[...] = f0
    L.k_bounceback(P(s.mask), P(s.f), nx, ny)
    out["after_bounce_f"] = s.f.copy(order="F")
    s.f[...] = f0
    s.update_hydro()
    out["hydro_rho"], out["hydro_u"], out["hydro_v"] = s.rho.copy(order="F"), s.u.copy(order="F"), s.v.copy(order="F")
    s.update_feq()
    out["feq"] = s.feq.copy(order="F")
    s.collide()
    out["after_collide_f"] = s.f.copy(order="F")
    s.zero_vel()
    out["zeroed_u"], out["zeroed_v"] = s.u.copy(order="F"), s.v.copy(order="F")
    save("o2_kernels_37x19", **out)

    # ---- b. the reference's Poiseuille verification case, N=10 ------------
    # opencl Pipe_Flow(diameter=1.5, rho=10, viscosity=5, pressure_grad=-100, pipe_length=3, N=10)
    # (docs/opencl_dimensionless_verification.ipynb:94-115); host arithmetic of opencl_dim.py:86-134, 266-283
    D, rho_p, nu, gradP, plen, N, tp = 1.5, 10., 5., -100., 3., 10, 1.
    Lc = D; zeta = np.abs(gradP) / rho_p; T = np.sqrt(D / zeta)
    Wn = (np.abs(gradP / rho_p) * Lc * T) / nu
    dx = 1. / N; dt = tp * dx ** 2
    omega = (3 * ((dt / dx ** 2) * (1. / Wn)) + 0.5) ** -1.
    lx = int(np.ceil((plen / Lc) * N)); ly = N; nx, ny = lx + 1, ly + 1
    rin = 1. + np.abs(nx * (dt ** 2 / dx) * (1. / cs2) * 1.); rout = 1.
    s = RefOpenCL(L, nx, ny, omega, rin, rout)
    s.rho[...] = ramp(nx, ny, rin, rout)
    s.update_feq()
    s.f[...] = s.feq; s.fs[...] = s.feq
    out = {"nx": nx, "ny": ny, "omega": omega, "inlet_rho": rin, "outlet_rho": rout,
           "T": T, "W": Wn, "delta_t": dt, "delta_x": dx, "L": Lc, "f0": s.f.copy(order="F")}
    done = 0
    for n in (1, 10, 200, 999):
        s.run(n - done); done = n
        out.update(flat("s%d" % n, s.snap()))
    save("o2_pipe_N10", **out)

    # ---- c. noisy pipe, odd sizes -----------------------------------------
    nx, ny, omega, rin, rout = 49, 25, 1.0, 1.004, 1.0
    rng = np.random.default_rng(7)
    s = RefOpenCL(L, nx, ny, omega, rin, rout)
    s.rho[...] = ramp(nx, ny, rin, rout)
    s.update_feq()
    f0 = np.asfortranarray((s.feq * (1. + 0.001 * rng.standard_normal((nx, ny, 9)))).astype(np.float32))
    s.f[...] = f0; s.fs[...] = f0
    out = {"nx": nx, "ny": ny, "omega": omega, "inlet_rho": rin, "outlet_rho": rout, "f0": f0}
    done = 0
    for n in (1, 100, 1000):
        s.run(n - done); done = n
        out.update(flat("s%d" % n, s.snap()))
    save("o2_pipe_noise_49x25", **out)

    # ---- d. cylinder in the pipe -------------------------------------------
    nx, ny, omega, rin, rout = 61, 31, 1.7, 1.003, 1.0
    rng = np.random.default_rng(11)
    mask = disc(nx, ny, 15.0, 15.0, 5.0)
    s = RefOpenCL(L, nx, ny, omega, rin, rout, mask)
    s.rho[...] = ramp(nx, ny, rin, rout)
    s.zero_vel()
    s.update_feq()
    f0 = np.asfortranarray((s.feq * (1. + 0.001 * rng.standard_normal((nx, ny, 9)))).astype(np.float32))
    s.f[...] = f0; s.fs[...] = f0
    out = {"nx": nx, "ny": ny, "omega": omega, "inlet_rho": rin, "outlet_rho": rout, "mask": mask, "f0": f0}
    done = 0
    for n in (1, 100, 500):
        s.run(n - done); done = n
        out.update(flat("s%d" % n, s.snap()))
    save("o2_cyl_61x31", **out)


def gen_o2_velocity_inlet(L):
    """The two velocity-inlet kernels of D2Q9.cl (:263-374), driven in the order of
    OLD/opencl.py:281-327 (Pipe_Flow_PeriodicBC_VelocityInlet: rho=1, u=u_w, v=0 at start)."""
    nx, ny, omega, uw = 45, 23, 1.1, 0.05
    rng = np.random.default_rng(21)
    s = RefOpenCL(L, nx, ny, omega, 1.0, 1.0)
    s.rho[...] = 1.0
    s.u[...] = uw
    s.update_feq()
    f0 = np.asfortranarray((s.feq * (1. + 0.001 * rng.standard_normal((nx, ny, 9)))).astype(np.float32))
    s.f[...] = f0; s.fs[...] = f0
    out = {"nx": nx, "ny": ny, "omega": omega, "u_w": uw, "u_e": uw, "f0": f0}
    # one call of each kernel on the initial state
    L.k_move_bcs_vel(P(s.f), P(s.u), np.float32(uw), np.float32(uw), nx, ny)
    out["after_bcs_f"] = s.f.copy(order="F")
    s.f[...] = f0
    L.k_update_hydro_vel(P(s.f), P(s.u), P(s.v), P(s.rho), np.float32(uw), np.float32(uw), nx, ny)
    out.update(hydro_rho=s.rho.copy(order="F"), hydro_u=s.u.copy(order="F"), hydro_v=s.v.copy(order="F"))
    # runs
    s.rho[...] = 1.0; s.u[...] = uw; s.v[...] = 0.0
    done = 0
    for n in (1, 20, 200):
        for _ in range(n - done):
            s.move()
            L.k_move_bcs_vel(P(s.f), P(s.u), np.float32(uw), np.float32(uw), nx, ny)
            L.k_update_hydro_vel(P(s.f), P(s.u), P(s.v), P(s.rho), np.float32(uw), np.float32(uw), nx, ny)
            s.update_feq()
            s.collide()
        done = n
        out.update(flat("s%d" % n, s.snap()))
    save("o2_velocity_inlet_45x23", **out)


def gen_o2_d2q9i(L):
    """LB_D2Q9/D2Q9i.cl (the "incompressible" fork) driven as dimensionless/opencl_dim_D2Q9i.py drives it:
    move, move_bcs (+ bounce-back), update_hydro (+ zero velocity in the obstacle), update_feq, collide.
    Executed faithfully, the fork diverges within tens of steps for every omega / pressure drop tried
    (its equilibrium is w*rho*(rho + ...), quadratic in rho): the fixture records that too."""
    nx, ny, omega, rin, rout = 53, 27, 1.0, 1.0002, 1.0
    rng = np.random.default_rng(31)
    mask = disc(nx, ny, 14.0, 13.0, 4.0)
    s = RefOpenCL(L, nx, ny, omega, rin, rout, mask)
    s.rho[...] = ramp(nx, ny, rin, rout)
    s.zero_vel()
    s.update_feq()
    f0 = np.asfortranarray((s.feq * (1. + 0.001 * rng.standard_normal((nx, ny, 9)))).astype(np.float32))
    s.f[...] = f0; s.fs[...] = f0
    out = {"nx": nx, "ny": ny, "omega": omega, "inlet_rho": rin, "outlet_rho": rout, "mask": mask, "f0": f0,
           "feq0": s.feq.copy(order="F")}
    L.k_move_bcs(P(s.f), P(s.u), s.rin, s.rout, nx, ny)
    out["after_bcs_f"] = s.f.copy(order="F")
    s.f[...] = f0
    s.update_hydro()
    out.update(hydro_rho=s.rho.copy(order="F"), hydro_u=s.u.copy(order="F"), hydro_v=s.v.copy(order="F"))
    s.update_feq()
    out["feq1"] = s.feq.copy(order="F")
    s.f[...] = f0; s.fs[...] = f0
    done = 0
    for n in (1, 10, 60):                    # the fork is numerically unstable: NaN well before step 60
        for _ in range(n - done):
            s.move(); s.move_bcs(); s.update_hydro(); s.zero_vel(); s.update_feq(); s.collide()
        done = n
        out.update(flat("s%d" % n, s.snap()))
    save("o2_d2q9i_53x27", **out)


# --------------------------------------------------------------------------
#  O1: cython_dim.pyx compiled and imported
# --------------------------------------------------------------------------
def build_o1(tmp):
    subprocess.check_call(["cython", "-2", os.path.join(REF, "dimensionless", "cython_dim.pyx"),
                           "-o", os.path.join(tmp, "cython_dim.c")])
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-w",
                           "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include(),
                           os.path.join(tmp, "cython_dim.c"), "-o", os.path.join(tmp, "cython_dim" + ext)])
    stub = os.path.join(tmp, "stub", "skimage")
    os.makedirs(stub)
    with open(os.path.join(stub, "__init__.py"), "w") as fh:
        fh.write("from . import draw\n")
    with open(os.path.join(stub, "draw.py"), "w") as fh:
        fh.write("import numpy as np\n"
                 "def circle(r, c, radius):\n"
                 "    lo_r, hi_r = int(np.floor(r - radius)), int(np.ceil(r + radius)) + 1\n"
                 "    lo_c, hi_c = int(np.floor(c - radius)), int(np.ceil(c + radius)) + 1\n"
                 "    rr, cc = np.mgrid[lo_r:hi_r, lo_c:hi_c]\n"
                 "    keep = ((rr - r) ** 2 + (cc - c) ** 2) < radius ** 2\n"
                 "    return rr[keep], cc[keep]\n")
    sys.path[:0] = [tmp, os.path.join(tmp, "stub")]
    if not hasattr(np, "bool"):
        np.bool = bool          # cython_dim.pyx:424 uses the removed alias
    return importlib.import_module("cython_dim")


def o1_snap(s):
    return {"f": np.array(s.f, copy=True), "feq": np.array(s.feq, copy=True),
            "rho": np.array(s.rho, copy=True), "u": np.array(s.u, copy=True), "v": np.array(s.v, copy=True)}


def o1_seed_state(s, rng):
    """Replace the unseeded init_pop perturbation by a seeded one of the same form."""
    s.update_feq()
    perturb = 1. + 0.001 * rng.standard_normal((s.nx, s.ny))
    s.f = (s.feq * perturb[None]).astype(np.float32)
    return perturb


def gen_o1(m):
    # ---- a. pipe, per-method trace of one step then long runs --------------
    kw = dict(diameter=1., rho=1., viscosity=0.2, pressure_grad=-1., pipe_length=2., N=16, time_prefactor=0.2)
    s = m.Pipe_Flow(**kw)
    rng = np.random.default_rng(3)
    perturb = o1_seed_state(s, rng)
    out = {"kw_names": np.array(list(kw.keys())), "kw_vals": np.array(list(kw.values()), float),
           "nx": s.nx, "ny": s.ny, "omega": s.omega, "inlet_rho": s.inlet_rho, "outlet_rho": s.outlet_rho,
           "T": s.T, "Re": s.Re, "perturb": perturb, "f0": s.f.copy(), "rho0": s.rho.copy()}
    # warm the stored u (move_bcs reads it) with two full steps, then trace the third
    s.run(2)
    out.update(flat("pre", o1_snap(s)))
    s.move_bcs();          out["t_bcs_f"] = s.f.copy()
    s.move();              out["t_move_f"] = s.f.copy()
    s.update_hydro();      out.update(flat("t_hydro", {"rho": s.rho.copy(), "u": s.u.copy(), "v": s.v.copy()}))
    s.update_feq();        out["t_feq"] = s.feq.copy()
    s.collide_particles(); out["t_collide_f"] = s.f.copy()
    done = 3
    for n in (50, 500):
        s.run(n - done); done = n
        out.update(flat("s%d" % n, o1_snap(s)))
    save("o1_pipe_33x17", **out)

    # ---- b. cylinder --------------------------------------------------------
    kw = dict(diameter=1., rho=1., viscosity=2., pressure_grad=-10., pipe_length=1.5, N=4, time_prefactor=0.02)
    s = m.Pipe_Flow_Cylinder(cylinder_center=[.4, .5], cylinder_radius=.1, **kw)
    rng = np.random.default_rng(5)
    perturb = o1_seed_state(s, rng)
    out = {"kw_names": np.array(list(kw.keys())), "kw_vals": np.array(list(kw.values()), float),
           "cylinder_center": np.array([.4, .5]), "cylinder_radius": .1,
           "nx": s.nx, "ny": s.ny, "omega": s.omega, "inlet_rho": s.inlet_rho, "outlet_rho": s.outlet_rho,
           "T": s.T, "Re": s.Re, "mask": np.array(s.obstacle_mask), "perturb": perturb,
           "f0": s.f.copy(), "rho0": s.rho.copy()}
    done = 0
    for n in (1, 50, 300):
        s.run(n - done); done = n
        out.update(flat("s%d" % n, o1_snap(s)))
    save("o1_cyl_61x41", **out)

    # ---- c. derived constants of the notebooks' cylinder cases (Appendix C) --
    rows = []
    for N, r in ((25, 1. / 25), (25, 1. / 10)):
        c = m.Pipe_Flow_Cylinder(cylinder_center=[.75, .5], cylinder_radius=r, diameter=1., rho=1.,
                                 viscosity=1., pressure_grad=-100., pipe_length=3., N=N)
        rows.append([N, r, c.L, c.T, c.Re, c.omega, c.inlet_rho, c.nx, c.ny])
    save("o1_constants", table=np.array(rows, float),
         columns=np.array(["N", "radius", "L", "T", "Re", "omega", "inlet_rho", "nx", "ny"]))


def gen_o1_config1(m):
    """BASELINE config 1 at full size through the imported reference: 256 x 256 Poiseuille start-up flow,
    `Pipe_Flow.run` (cython_dim.pyx:346-359), 1000 steps from f = feq (the perturbation of init_pop replaced by
    none).  Stored: x-means of u and rho, every fourth row / column of rho, u, v, and the derived constants."""
    kw = dict(diameter=1., rho=1., viscosity=.05, pressure_grad=-1., pipe_length=1., N=255, time_prefactor=25.5)
    s = m.Pipe_Flow(**kw)
    s.update_feq()
    s.f = s.feq.copy()
    s.run(1000)
    rho, u, v = np.array(s.rho), np.array(s.u), np.array(s.v)
    save("o1_config1_256", kw_names=np.array(list(kw.keys())), kw_vals=np.array(list(kw.values()), float),
         nx=s.nx, ny=s.ny, omega=s.omega, inlet_rho=s.inlet_rho, outlet_rho=s.outlet_rho, T=s.T, Re=s.Re,
         steps=1000, stride=4, u_xmean=u.mean(axis=0), v_xmean=v.mean(axis=0), rho_ymean=rho.astype(np.float64).mean(axis=1),
         rho_sub=rho[::4, ::4].copy(), u_sub=u[::4, ::4].copy(), v_sub=v[::4, ::4].copy(),
         u_col0=u[0].copy(), u_collast=u[-1].copy())


# --------------------------------------------------------------------------
#  move_periodic of the research forks (porous_media/single_component.cl:338-375), executed
# --------------------------------------------------------------------------
MOVE_PERIODIC_DRIVER = r'''
#include <stddef.h>
#include <math.h>
static int g_gid[3], g_lid[3], g_lsz[3] = {32, 32, 1};
#define cl_khr_fp64 1            /* the capability macro an fp64-capable OpenCL compiler predefines (the file #errors without) */
#define __kernel
#define __global
#define __local
#define __constant const
#define __read_only
#define __write_only
#define CLK_LOCAL_MEM_FENCE 0
static inline int get_global_id(int d)  { return g_gid[d]; }
static inline int get_local_id(int d)   { return g_lid[d]; }
static inline int get_local_size(int d) { return g_lsz[d]; }
static inline void barrier(int flags)   { (void)flags; }
#include "%(cl)s"

static int pad32(int n) { return (n + 31) / 32 * 32; }
static const int CXc[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
static const int CYc[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
/* single_component.py:177-184: one launch per population over the padded 2-d NDRange (32 x 32 work-groups) */
void k_move_periodic(double *f, double *fs, int nx, int ny, int cur_field, int num_populations, int num_jumpers)
{
    for (int y = 0; y < pad32(ny); ++y)
        for (int x = 0; x < pad32(nx); ++x) {
            g_gid[0] = x; g_gid[1] = y; g_gid[2] = 0;
            g_lid[0] = x %% 32; g_lid[1] = y %% 32; g_lid[2] = 0;
            move_periodic(f, fs, CXc, CYc, nx, ny, cur_field, num_populations, num_jumpers);
        }
}
'''


def gen_move_periodic(tmp):
    """Execute the forks' `move_periodic` (porous_media/single_component.cl:338-375; host call single_component.py:177-184)
    on small [jumper][population][y][x] arrays of distinct integer-valued doubles -- pure index arithmetic -- and store
    input and output.  The product's populations module streams fp32 lattices; integers below 2^24 survive the cast."""
    src = os.path.join(tmp, "move_periodic_driver.c")
    with open(src, "w") as fh:
        fh.write(MOVE_PERIODIC_DRIVER % {"cl": os.path.join(REF, "porous_media", "single_component.cl")})
    so = os.path.join(tmp, "libmoveperiodic.so")
    subprocess.check_call(["gcc", "-O1", "-std=gnu99", "-ffp-contract=off", "-fPIC", "-shared", "-w", src, "-o", so, "-lm"])
    L = ct.CDLL(so)
    dp = ct.POINTER(ct.c_double)
    L.k_move_periodic.argtypes = [dp, dp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int]
    out = {}
    for tag, (nx, ny, P) in (("a", (32, 24, 2)), ("b", (37, 19, 2)), ("c", (5, 7, 1))):
        rng = np.random.default_rng(nx * 100 + ny)
        n = nx * ny * P * 9
        # the forks' shape (nx, ny, num_populations, num_jumpers), Fortran order = the kernel's jump*P*nx*ny + field*nx*ny + y*nx + x
        f = np.asfortranarray(rng.permutation(n).astype(np.float64).reshape((nx, ny, P, 9), order="F"))
        fs = np.asfortranarray(np.full((nx, ny, P, 9), -1.0))
        for field in range(P):
            L.k_move_periodic(f.ctypes.data_as(dp), fs.ctypes.data_as(dp), nx, ny, field, P, 9)
        assert fs.min() >= 0 and np.array_equal(np.sort(fs.ravel()), np.arange(n))     # a permutation: every entry written once
        out["f_" + tag], out["streamed_" + tag] = f.astype(np.int32), fs.astype(np.int32)     # (integers: stored as such)
    save("o2_move_periodic", **out)


def main():
    tmp = tempfile.mkdtemp(prefix="lb_golden_", dir="/tmp")
    print("scratch dir", tmp)
    if "--only-move-periodic" in sys.argv:
        gen_move_periodic(tmp)
        return
    if "--only-config1" in sys.argv:
        gen_o1_config1(build_o1(tmp))
        return
    L = build_o2(tmp)
    if "--only-velocity-inlet" not in sys.argv and "--only-d2q9i" not in sys.argv:
        gen_o2(L)
    if "--only-d2q9i" not in sys.argv:
        gen_o2_velocity_inlet(L)
    if "--only-velocity-inlet" in sys.argv:
        return
    gen_o2_d2q9i(build_o2(tmp, "D2Q9i.cl"))
    if "--only-d2q9i" in sys.argv:
        return
    gen_move_periodic(tmp)
    m = build_o1(tmp)
    gen_o1(m)
    gen_o1_config1(m)


if __name__ == "__main__":
    main()
