"""BASELINE config 1 at its full size: 256 x 256 Poiseuille pipe flow with the semantics of the reference's
CPU path (`cython_dim.Pipe_Flow.run`, cython_dim.pyx:346-359), 1000 steps from f = feq.

Three anchors:
  * tests/golden/o1_config1_256.npz - the imported (cythonized) reference itself run at this size
    (oracle/make_golden.py gen_o1_config1): x-means and every fourth row / column of rho, u, v;
  * the oracle's restatement (O1Sim), full fields;
  * the analytic start-up solution of plane Poiseuille flow (the reference's own known-answer method,
    docs/opencl_dimensionless_verification.ipynb:632-696, here at t = 1000 steps instead of the steady state):
      u(y,t) = G/(2 nu) y (D-y) - sum_{n odd} 4 G D^2 / (nu n^3 pi^3) sin(n pi y / D) exp(-n^2 pi^2 nu t / D^2),
    G = cs^2 (rho_in - 1) / nx, D = ly, nu = lb_viscosity.

Tolerances (fp32 device vs the reference's float64 temporaries; SURVEY 8c, <= 1000 laminar steps):
|d rho| <= 1e-5, |d u|, |d v| <= 5e-6; analytic profile: 2e-5 (0.3 % of the peak velocity: the lattice
solution carries its own discretisation and compressibility error).
"""
import numpy as np
import pytest

from conftest import golden

KW = dict(diameter=1., rho=1., viscosity=.05, pressure_grad=-1., pipe_length=1., N=255, time_prefactor=25.5)
STEPS = 1000


def md(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def startup_profile(ny, nx, inlet_rho, nu, t):
    D = ny - 1
    G = (1. / 3.) * (inlet_rho - 1.) / nx
    y = np.arange(ny, dtype=np.float64)
    u = G / (2 * nu) * y * (D - y)
    for n in range(1, 400, 2):
        u -= 4 * G * D * D / (nu * (n * np.pi) ** 3) * np.sin(n * np.pi * y / D) * np.exp(-(n * np.pi) ** 2 * nu * t / D ** 2)
    return u


def check_against_reference_fixture(d, rho, u, v, tol_rho, tol_u):
    st = int(d["stride"])
    assert md(rho[::st, ::st], d["rho_sub"]) <= tol_rho
    assert md(u[::st, ::st], d["u_sub"]) <= tol_u and md(v[::st, ::st], d["v_sub"]) <= tol_u
    assert md(u.mean(axis=0), d["u_xmean"]) <= tol_u and md(v.mean(axis=0), d["v_xmean"]) <= tol_u
    assert md(rho.astype(np.float64).mean(axis=1), d["rho_ymean"]) <= tol_rho
    assert md(u[0], d["u_col0"]) <= tol_u and md(u[-1], d["u_collast"]) <= tol_u


def test_config1_oracle_vs_imported_reference_and_analytic(oracle):
    """The oracle's Cython-path port at config 1: bit-exact against the imported reference's own run, and on the
    analytic start-up profile."""
    d = golden("o1_config1_256")
    assert dict(zip(d["kw_names"], d["kw_vals"])) == KW and int(d["steps"]) == STEPS
    s = oracle.O1Sim.pipe_flow(numpy2=True, **KW)
    assert (s.nx, s.ny) == (256, 256) == (int(d["nx"]), int(d["ny"]))
    assert s.omega == float(d["omega"]) == pytest.approx(0.8992805755, rel=1e-9)
    assert s.inlet_rho == float(d["inlet_rho"]) == pytest.approx(1.0048188235, rel=1e-9)
    s.run(STEPS)
    check_against_reference_fixture(d, s.rho, s.u, s.v, 0.0, 0.0)          # bit-exact
    ua = startup_profile(s.ny, s.nx, s.inlet_rho, s.params["lb_viscosity"], STEPS)
    assert md(s.u.mean(axis=0), ua) <= 2e-5 and ua.max() == pytest.approx(6.2745e-3, rel=1e-3)
    # the float32-collide mode (NumPy 1 behaviour, what a float32 device is closest to) stays within the tolerances
    s1 = oracle.O1Sim.pipe_flow(numpy2=False, **KW)
    s1.run(STEPS)
    check_against_reference_fixture(d, s1.rho, s1.u, s1.v, 1e-5, 5e-6)


@pytest.mark.gpu
def test_config1_gpu_cython_path_256_poiseuille(lbhip, oracle):
    """The same configuration through the product: `LB_D2Q9.dimensionless.cython_dim.Pipe_Flow` on the GPU
    (Cython-path semantics: k1_bcs + fused k1_step per step), `init_pop(amplitude=0)`, 1000 steps."""
    from LB_D2Q9.dimensionless import cython_dim as lb
    d = golden("o1_config1_256")
    sim = lb.Pipe_Flow(verbose=False, **KW)
    assert (sim.nx, sim.ny) == (256, 256)
    assert sim.omega == float(d["omega"]) and sim.inlet_rho == float(d["inlet_rho"]) and sim.Re == float(d["Re"])
    sim.init_pop(amplitude=0.)
    ref = oracle.O1Sim.pipe_flow(numpy2=False, **KW)
    assert md(sim.get_fields()["f"], ref.f) == 0.0
    sim.run(STEPS)
    ref.run(STEPS)
    g = sim.get_fields()
    assert g["u"].dtype == np.float64 and g["rho"].shape == (256, 256) and g["f"].shape == (9, 256, 256)
    # the imported reference's own run
    check_against_reference_fixture(d, g["rho"], g["u"], g["v"], 1e-5, 5e-6)
    # the oracle, full fields
    assert md(g["rho"], ref.rho) <= 1e-5 and md(g["u"], ref.u) <= 5e-6 and md(g["v"], ref.v) <= 5e-6
    assert md(g["f"], ref.f) <= 1e-5
    # the parabola (start-up profile)
    ua = startup_profile(sim.ny, sim.nx, sim.inlet_rho, sim.lb_viscosity, STEPS)
    assert md(g["u"].mean(axis=0), ua) <= 2e-5
    # wall rows: u = v = 0 (cython_dim.pyx:318-321); the four corner cells then take the inlet / outlet formula (:326-333)
    assert np.all(g["u"][1:-1, 0] == 0) and np.all(g["u"][1:-1, -1] == 0) and np.all(g["v"][:, 0] == 0)
    assert md(g["u"][[0, -1]][:, [0, -1]], ref.u[[0, -1]][:, [0, -1]]) <= 5e-6
