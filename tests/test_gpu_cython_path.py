"""The reference's CPU ("Cython") path on the GPU (semantics='cython', LB_D2Q9.dimensionless.cython_dim)
against the fixtures produced by the imported reference module (oracle/make_golden.py) and against
the oracle's restatement.  fp32 on the device vs float64 temporaries in the reference: tolerances as for
the OpenCL path (single phase 1e-6; <= 500 steps: rho 1e-5, u/v 5e-6)."""
import numpy as np
import pytest

from conftest import golden
from test_oracle_golden import kwargs_of

pytestmark = pytest.mark.gpu


def md(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def test_cython_pipe_trace_and_runs_vs_reference_fixture(lbhip):
    from LB_D2Q9.dimensionless import cython_dim as lb
    d = golden("o1_pipe_33x17")
    sim = lb.Pipe_Flow(verbose=False, **kwargs_of(d))
    assert (sim.nx, sim.ny) == (int(d["nx"]), int(d["ny"]))
    assert sim.omega == float(d["omega"]) and sim.inlet_rho == float(d["inlet_rho"]) and sim.Re == float(d["Re"])
    sim.init_pop(amplitude=0.)
    assert md(sim.get_fields()["rho"], d["rho0"]) == 0.0
    sim.set_f(d["f0"])
    sim.run(2)
    g = sim.get_fields()
    assert g["f"].shape == (9, sim.nx, sim.ny) and g["f"].dtype == np.float32 and g["u"].dtype == np.float64
    for k, tol in (("f", 5e-7), ("feq", 5e-7), ("rho", 1e-6), ("u", 1e-6), ("v", 1e-6)):
        assert md(g[k], d["pre_" + k]) <= tol, k
    # one step phase by phase, restarting every phase from the reference's own state
    sim.set_f(d["pre_f"]); sim.set_fields(d["pre_rho"], d["pre_u"], d["pre_v"])
    sim.move_bcs();          assert md(sim.get_fields()["f"], d["t_bcs_f"]) <= 2.5e-7
    sim.set_f(d["t_bcs_f"])
    sim.move();              assert md(sim.get_fields()["f"], d["t_move_f"]) == 0.0        # pure data movement
    sim.update_hydro()
    g = sim.get_fields()
    assert md(g["rho"], d["t_hydro_rho"]) <= 5e-7 and md(g["u"], d["t_hydro_u"]) <= 1e-6 and md(g["v"], d["t_hydro_v"]) <= 1e-6
    sim.set_fields(d["t_hydro_rho"], d["t_hydro_u"], d["t_hydro_v"])
    sim.update_feq();        assert md(sim.get_fields()["feq"], d["t_feq"]) <= 2.5e-7
    sim.collide_particles(); assert md(sim.get_fields()["f"], d["t_collide_f"]) <= 2.5e-7
    # long runs from the start
    sim.set_f(d["f0"]); sim.set_fields(d["rho0"], np.zeros_like(d["rho0"]), np.zeros_like(d["rho0"]))
    done = 0
    for n in (50, 500):
        sim.run(n - done)
        done = n
        g = sim.get_fields()
        assert md(g["rho"], d["s%d_rho" % n]) <= 1e-5, n
        assert md(g["u"], d["s%d_u" % n]) <= 5e-6 and md(g["v"], d["s%d_v" % n]) <= 5e-6, n
        assert md(g["f"], d["s%d_f" % n]) <= 1e-5, n


def test_cython_cylinder_vs_reference_fixture(lbhip):
    from LB_D2Q9.dimensionless import cython_dim as lb
    d = golden("o1_cyl_61x41")
    sim = lb.Pipe_Flow_Cylinder(cylinder_center=list(d["cylinder_center"]), cylinder_radius=float(d["cylinder_radius"]),
                                verbose=False, **kwargs_of(d))
    assert np.array_equal(sim.obstacle_mask, d["mask"])
    assert sim.omega == float(d["omega"]) and sim.inlet_rho == float(d["inlet_rho"])
    sim.set_f(d["f0"])
    done = 0
    for n in (1, 50, 300):
        sim.run(n - done)
        done = n
        g = sim.get_fields()
        assert md(g["rho"], d["s%d_rho" % n]) <= 1e-5 and md(g["u"], d["s%d_u" % n]) <= 5e-6, n
        assert md(g["v"], d["s%d_v" % n]) <= 5e-6 and md(g["f"], d["s%d_f" % n]) <= 1e-5, n
        assert np.all(g["u"][d["mask"]] == 0) and np.all(g["v"][d["mask"]] == 0)


def test_cython_path_larger_grid_vs_oracle(lbhip, oracle):
    """257 x 129 pipe with a disc, 200 steps, against the oracle's restatement (numpy-1 float32 mode, which
    is what a float32 device is closest to)."""
    from LB_D2Q9.dimensionless import cython_dim as lb
    kw = dict(diameter=1., rho=1., viscosity=0.2, pressure_grad=-1., pipe_length=2., N=128, time_prefactor=12.8)
    cyl = dict(cylinder_center=[.5, .5], cylinder_radius=1.)
    np.random.seed(5)
    sim = lb.Pipe_Flow(verbose=False, **kw)
    np.random.seed(5)
    perturb = 1. + .001 * np.random.randn(sim.nx, sim.ny)
    ref = oracle.O1Sim.pipe_flow(perturb=perturb, numpy2=False, **kw)
    assert sim.omega == ref.omega and sim.inlet_rho == ref.inlet_rho
    assert md(sim.get_fields()["f"], ref.f) == 0.0
    sim.run(200); ref.run(200)
    g = sim.get_fields()
    assert md(g["rho"], ref.rho) <= 1e-5 and md(g["u"], ref.u) <= 5e-6 and md(g["v"], ref.v) <= 5e-6


def test_cython_semantics_restrictions(lbhip):
    from LB_D2Q9 import _native
    from LB_D2Q9.simulation import Simulation
    with pytest.raises(_native.LbError):
        Simulation(64, 64, 1.0, bc="periodic", semantics="cython")
    with pytest.raises(_native.LbError):
        Simulation(64, 64, 1.0, bc="pipe", semantics="cython", y0=0, local_ny=32)


def test_cython_path_fused_run_equals_phase_calls(lbhip):
    """run(n) on the Cython-path classes = the first step's boundary phase + ONE fused pass per step (restricted pull,
    moments, equilibrium, relaxation, the next step's boundary rule; four cells per lane); it must equal the five phase calls of the reference's loop (cython_dim.pyx:346-359) bit
    for bit, with and without an obstacle, on sizes that are not multiples of the launch shape."""
    from LB_D2Q9.dimensionless import cython_dim as lb
    # (N = 37: 101 x 38 cells, too small for LDS tiles -> single-step passes k1_fstep only; the other two: four steps per
    #  launch through k1_tile4 + single steps for the remainder of a run)
    for cls, extra in ((lb.Pipe_Flow, {}), (lb.Pipe_Flow, dict(N=150)),
                       (lb.Pipe_Flow_Cylinder, dict(N=90, cylinder_center=[.6, .5], cylinder_radius=.12))):
        kw = dict(diameter=1., rho=1., viscosity=.2, pressure_grad=-1.5, pipe_length=2.7, N=37, time_prefactor=.2,
                  verbose=False)
        kw.update(extra)                             # (N counts lattice points per cylinder radius in the cylinder class)
        np.random.seed(3)
        a = cls(**kw)
        assert ("k1_tile4" in a._sim.hot_kernel()) == bool(extra) and a._sim.steps_per_launch() == (4 if extra else 1)
        f0 = a.get_fields()["f"]
        b = cls(**kw)
        b.set_f(f0)
        g0 = a.get_fields()
        b.set_fields(g0["rho"], g0["u"], g0["v"])
        a.run(1); a.run(10); a.run(12)              # (every run starts with the boundary phase as a launch of its own and ends
        for _ in range(23):                          #  with a pass that leaves the post-collision populations alone)
            b.move_bcs(); b.move(); b.update_hydro(); b.update_feq(); b.collide_particles()
        ga, gb = a.get_fields(), b.get_fields()
        for k in ("f", "rho", "u", "v", "feq"):
            assert np.array_equal(ga[k], gb[k]), (cls.__name__, kw["N"], k)
        assert np.all(np.isfinite(ga["f"]))
        # single steps only (explicit variant without the tile bit): the same bits again
        c = cls(**kw)
        c._sim.set_variant(0)
        c.set_f(f0)
        c.set_fields(g0["rho"], g0["u"], g0["v"])
        c.run(23)
        gc = c.get_fields()
        for k in ("f", "rho", "u", "v"):
            assert np.array_equal(ga[k], gc[k]), (cls.__name__, kw["N"], "single steps", k)
