"""The reference's "incompressible" fork (LB_D2Q9/D2Q9i.cl, dimensionless/opencl_dim_D2Q9i.py) on the GPU: engine
semantics 'd2q9i', module LB_D2Q9.dimensionless.opencl_dim_D2Q9i.  Checked against the fixture made by EXECUTING
D2Q9i.cl (oracle/make_golden.py gen_o2_d2q9i) over the fork's stable window -- it is unstable: the fixture's |u| grows
tenfold between step 1 and step 10 and is NaN by step 60 -- and against the oracle's bit-exact restatement.

Tolerances: one phase / one step as everywhere (|d f| <= 2.5e-7, |d rho| <= 5e-7, |d u|, |d v| <= 1e-6); at step 10 the
same single-step rounding differences have been amplified by the instability like the solution itself, so the bound is
the multi-step one (|d f|, |d rho| <= 1e-5, |d u|, |d v| <= 5e-6) or 1e-4 of the field's own range, whichever is larger
(measured at step 10 of the fixture: 1.5e-6 on v, whose range is 1.2e-2)."""
import numpy as np
import pytest

from conftest import golden
from test_gpu_parity import TOL1, TOLN, assert_fields_close, maxdiff, _random_state

pytestmark = pytest.mark.gpu


def _sim(d, **kw):
    from LB_D2Q9.simulation import Simulation
    return Simulation(int(d["nx"]), int(d["ny"]), float(d["omega"]), bc="pipe", inlet_rho=float(d["inlet_rho"]),
                      outlet_rho=float(d["outlet_rho"]), obstacle_mask=d["mask"], semantics="d2q9i", **kw)


def test_d2q9i_phases_vs_executed_fork(lbhip):
    d = golden("o2_d2q9i_53x27")
    m = d["mask"].astype(bool)
    sim = _sim(d)
    sim.set_f(d["f0"])
    sim.move_bcs()                                   # move_bcs + bounceback_in_obstacle: compose the golden on the host
    want = d["after_bcs_f"].copy()
    for a, b in ((1, 3), (2, 4), (5, 7), (6, 8)):
        ta, tb = want[..., a][m].copy(), want[..., b][m].copy()
        want[..., a][m], want[..., b][m] = tb, ta
    assert maxdiff(sim.get_fields(("f",))["f"], want) <= 2.5e-7
    sim.set_f(d["f0"])
    sim.update_hydro()                               # D2Q9i.cl:67-97 + u, v zeroed in the obstacle (the cylinder class)
    g = sim.get_fields(("rho", "u", "v"))
    hu, hv = d["hydro_u"].copy(), d["hydro_v"].copy()
    hu[m] = 0; hv[m] = 0
    assert maxdiff(g["rho"], d["hydro_rho"]) <= 5e-7 and maxdiff(g["u"], hu) <= 1e-6 and maxdiff(g["v"], hv) <= 1e-6
    sim.set_fields(d["hydro_rho"], d["hydro_u"], d["hydro_v"])
    sim.update_feq()
    assert maxdiff(sim.get_fields(("feq",))["feq"], d["feq1"]) <= 2.5e-7


@pytest.mark.parametrize("variant", [-1, 0])
def test_d2q9i_fused_run_vs_executed_fork_over_its_stable_window(lbhip, variant):
    d = golden("o2_d2q9i_53x27")
    sim = _sim(d)
    sim.set_variant(variant)
    sim.set_f(d["f0"])
    sim.run(1)
    assert_fields_close(sim.get_fields(), d, TOL1, "s1_")
    sim.run(9)
    g = sim.get_fields()
    for k in ("f", "rho", "u", "v"):
        want = d["s10_" + k]
        span = float(np.abs(want - want.mean()).max())
        assert maxdiff(g[k], want) <= max(1e-4 * span, TOLN[k]), k
    assert np.abs(d["s10_u"]).max() > 10 * np.abs(d["s1_u"]).max()        # (the reference execution is running away)
    # fused == the fork's own phase order, un-fused: move, move_bcs, update_hydro, update_feq, collide_particles
    a, b = _sim(d), _sim(d)
    a.set_f(d["f0"]); b.set_f(d["f0"])
    a.run(3)
    for _ in range(3):
        b.move(); b.move_bcs(); b.update_hydro(); b.update_feq(); b.collide_particles()
    ga, gb = a.get_fields(), b.get_fields()
    for k in ("f", "feq", "rho", "u", "v"):
        assert maxdiff(ga[k], gb[k]) <= 1e-6, k


@pytest.mark.parametrize("nx,ny,masked", [(1003, 177, True), (1024, 160, False), (96, 64, True)])
def test_d2q9i_every_fused_kernel_bitwise_and_vs_oracle(lbhip, oracle, nx, ny, masked):
    """k_step, k_step2 ... k_step5, k_deep<6>, k_deep<7> and the LDS-tile kernel with the fork's cell routines: bitwise equal, and on the oracle's
    restatement (bit-exact against the executed fork) after 8 steps of a near-equilibrium state."""
    from LB_D2Q9.simulation import Simulation
    rng = np.random.default_rng(nx + ny)
    f0 = _random_state(rng, nx, ny, amp=0.001)
    mask = None
    if masked:
        mask = rng.random((nx, ny)) < 0.03
        mask[0, :] = mask[-1, :] = False
        mask[:, 0] = mask[:, -1] = False
    outs = []
    variants = (0, 33, 97, 353, 864, 353 | 4096, 353 | 4096 | 16384, 353 | 4096 | 16384 | 32768, 353 | 4096 | 16384 | 32768 | 65536) if nx >= 512 else (0, -1, 512)
    for variant in variants:
        s = Simulation(nx, ny, 1.0, bc="pipe", inlet_rho=1.0002, obstacle_mask=mask, semantics="d2q9i")
        s.set_variant(variant)
        if nx >= 512:
            assert s.steps_per_launch() == {0: 1, 33: 2, 97: 3, 353: 4, 864: 4, 4449: 5, 20833: 6, 53601: 7, 119137: 7}[variant]
        assert "D2Q9i" in s.hot_kernel()
        s.set_f(f0)
        s.run(5); s.run(3)
        outs.append(s.get_fields(("f", "rho", "u", "v")))
    for o in outs[1:]:
        for k in o:
            assert np.array_equal(outs[0][k], o[k]), k
    ref = oracle.O2Sim(nx, ny, 1.0, oracle.BC_PIPE, 1.0002, 1., mask=mask, d2q9i=True)
    ref.set_f(f0)
    ref.run(8)
    w = ref.get_fields()
    for k in ("f", "rho", "u", "v"):
        span = float(np.abs(w[k] - w[k].mean()).max())
        assert maxdiff(outs[0][k], w[k]) <= max(1e-4 * span, TOLN[k]), k


def test_d2q9i_module_classes(lbhip, oracle):
    """LB_D2Q9.dimensionless.opencl_dim_D2Q9i: the Cython classes' non-dimensionalisation on the fork's kernels
    (opencl_dim_D2Q9i.py:98-120, 175-180, 253-259, 436-440)."""
    from LB_D2Q9.dimensionless import opencl_dim_D2Q9i as lb
    kw = dict(diameter=1., rho=1., viscosity=1., pressure_grad=-10., pipe_length=3., N=8)
    np.random.seed(4)
    c = lb.Pipe_Flow_Cylinder(cylinder_center=[.75, .5], cylinder_radius=.1, verbose=False, **kw)
    p = oracle.cython_pipe_parameters(cylinder_radius=.1, **kw)
    assert (c.nx, c.ny) == (p["nx"], p["ny"]) == (241, 81)
    assert c.omega == p["omega"] == pytest.approx(0.413223140496, rel=1e-11) and c.Re == p["Re"] and c.T == p["T"]
    assert c.inlet_rho == p["inlet_rho"] and int(np.asarray(c.obstacle_mask_host).sum()) == 193
    g0 = c.get_fields()
    assert np.all(g0["u"] == 0) and g0["f"].shape == (241, 81, 9)
    ref = oracle.O2Sim(c.nx, c.ny, c.omega, oracle.BC_PIPE, c.inlet_rho, c.outlet_rho, mask=c.obstacle_mask_host, d2q9i=True)
    ref.set_f(g0["f"])
    c.run(5); ref.run(5)
    g, w = c.get_fields(), ref.get_fields()
    for k in ("f", "rho", "u", "v"):
        span = float(np.abs(w[k] - w[k].mean()).max())
        assert maxdiff(g[k], w[k]) <= max(1e-4 * span, TOLN[k]), k
    m = np.asarray(c.obstacle_mask_host).astype(bool)
    assert np.all(g["u"][m] == 0) and np.all(g["v"][m] == 0)
    s = lb.Pipe_Flow(diameter=1., rho=1., viscosity=.05, pressure_grad=-1., pipe_length=1., N=31, time_prefactor=3.1,
                     verbose=False)
    q = oracle.cython_pipe_parameters(diameter=1., rho=1., viscosity=.05, pressure_grad=-1., pipe_length=1., N=31,
                                      time_prefactor=3.1)
    assert (s.omega, s.inlet_rho, s.nx, s.ny) == (q["omega"], q["inlet_rho"], q["nx"], q["ny"])
