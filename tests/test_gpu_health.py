"""rho, u, v on demand and the device-side health check.

* lb_run does not store rho, u, v in the plain families (include/lb_hip.h, LB_FLAG_EAGER_MACRO): they are the moments of the
  post-collision populations, rebuilt by one device pass when asked for.  BGK relaxation conserves rho and rho*u, so
  this differs from what the fused kernel would have stored (the pre-collision moments, as the reference's update_hydro
  does, opencl_dim.py:384-385) by rounding only: bounds below, and every fixture of tests/test_gpu_parity.py is held to
  the single-step tolerances through this path.
* lb_check: non-finite cells, max Mach number, total mass (what the reference's forks print / warn about:
  porous_media/single_component.py:221-225, 753-766) against numpy on the downloaded populations.
"""
import warnings

import numpy as np
import pytest

from conftest import golden
from test_gpu_parity import _random_state, maxdiff

pytestmark = pytest.mark.gpu

FAMILIES = [("periodic", {}), ("cavity", {"lid_u": 0.08, "rho0": 1.0}), ("pipe", {"inlet_rho": 1.01, "outlet_rho": 1.0})]


def host_health(f):
    """n_nonfinite, max Mach, sum rho of an (nx, ny, 9) array, float64 on the host."""
    f = np.asarray(f, np.float64)
    with np.errstate(all="ignore"):
        rho = f.sum(axis=2)
        ux = (f[..., 1] - f[..., 3] + f[..., 5] - f[..., 6] - f[..., 7] + f[..., 8]) / rho
        uy = (f[..., 5] + f[..., 2] + f[..., 6] - f[..., 7] - f[..., 4] - f[..., 8]) / rho
        usq = ux * ux + uy * uy
    ok = np.isfinite(rho) & np.isfinite(usq)
    return int((~ok).sum()), float(np.sqrt(3.0 * usq[ok].max())), float(rho[ok].sum())


@pytest.mark.parametrize("bc", ["periodic", "pipe", "cavity"])
@pytest.mark.parametrize("masked", [False, True])
def test_deep_kernels_store_the_single_step_kernels_fields_when_eager(lbhip, bc, masked):
    """k_deep<6> / k_deep<7> instantiated with MACRO (a handle that stores rho, u, v with the last launch of every run: the reference-named
    classes do) -- the launch whose hand-written wait counts twelve stores instead of nine -- against the single-step kernel on such a
    handle: populations and stored fields, bit for bit, over runs whose last launch is a deep one."""
    from LB_D2Q9.simulation import Simulation
    nx, ny = 1216, 320
    rng = np.random.default_rng(77)
    f0 = _random_state(rng, nx, ny)
    mask = None
    if masked:
        mask = rng.random((nx, ny)) < 0.02
        if bc != "periodic":
            mask[0, :] = mask[-1, :] = False
            mask[:, 0] = mask[:, -1] = False
    out = []
    for variant in (0, 97 | 256 | 4096 | 16384, 97 | 256 | 4096 | 16384 | 32768, 97 | 256 | 4096 | 16384 | 32768 | 65536):
        s = Simulation(nx, ny, 1.6, bc=bc, obstacle_mask=mask, eager_macro=True, inlet_rho=1.002, lid_u=0.04)
        s.set_variant(variant)
        if variant:
            assert "k_deep" in s.hot_kernel()
        s.set_f(f0)
        for n in (7, 14, 6, 13, 1):
            s.run(n)
        out.append(s.get_fields(("f", "rho", "u", "v")))
        s.close()
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k]) and np.array_equal(out[0][k], out[2][k]), k


@pytest.mark.parametrize("bc,kw", FAMILIES)
@pytest.mark.parametrize("nx,ny,variant", [(67, 29, 0), (1030, 130, -1), (1024, 256, 353), (300, 200, 512)])
def test_fields_on_demand_equal_stored_fields_within_rounding(lbhip, bc, kw, nx, ny, variant):
    from LB_D2Q9.simulation import Simulation
    if bc == "periodic" and nx % 4 and variant > 0:
        pytest.skip("marching kernels need nx % 4 == 0 in a periodic box")
    rng = np.random.default_rng(nx + ny)
    f0 = _random_state(rng, nx, ny)
    mask = rng.random((nx, ny)) < 0.03
    if bc != "periodic":
        mask[0, :] = mask[-1, :] = False
        mask[:, 0] = mask[:, -1] = False
    out = []
    for eager in (False, True):
        s = Simulation(nx, ny, 1.6, bc=bc, obstacle_mask=mask, eager_macro=eager, **kw)
        s.set_variant(variant)
        s.set_f(f0)
        s.run(9)
        s.run(4)
        out.append(s.get_fields(("f", "rho", "u", "v", "feq")))
        s.close()
    lazy, eager = out
    assert np.array_equal(lazy["f"], eager["f"])                     # the populations do not know the difference
    # rounding of nine relaxed populations (|f| <= 0.45: 3e-8 each) + the sum's own + (round 6) that of omega * rho, through which omega
    # enters the nine equilibria at once: a few ulp of 1 -- the contract's single-step bound on rho, measured 4.2e-7
    assert maxdiff(lazy["rho"], eager["rho"]) <= 5e-7
    assert maxdiff(lazy["u"], eager["u"]) <= 2.5e-7 and maxdiff(lazy["v"], eager["v"]) <= 2.5e-7
    assert maxdiff(lazy["feq"], eager["feq"]) <= 2.5e-7


def test_fields_on_demand_are_those_of_the_last_step_whatever_happens_next(lbhip):
    """The fields are rebuilt before anything else overwrites the populations they derive from (lb_set_f, the un-fused
    phases): observable behaviour = the reference's, whose rho, u, v buffers only change in update_hydro."""
    from LB_D2Q9.simulation import Simulation
    nx, ny = 130, 70
    rng = np.random.default_rng(5)
    f0, f1 = _random_state(rng, nx, ny), _random_state(rng, nx, ny, amp=0.05)
    a = Simulation(nx, ny, 1.2, bc="pipe", inlet_rho=1.01)
    b = Simulation(nx, ny, 1.2, bc="pipe", inlet_rho=1.01)
    for s in (a, b):
        s.set_f(f0)
        s.run(6)
    want = a.get_fields(("rho", "u", "v"))           # rebuilt right after the run
    b.set_f(f1)                                       # ... here: before the populations are replaced
    got = b.get_fields(("rho", "u", "v", "f"))
    assert np.array_equal(got["f"], f1)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    b.move(); b.move_bcs()                            # the phases leave rho, u, v alone, as in the reference
    for k in want:
        assert np.array_equal(b.get_fields((k,))[k], want[k]), k
    b.update_hydro()                                  # ... until update_hydro
    assert maxdiff(b.get_fields(("rho",))["rho"], want["rho"]) > 1e-4
    # a run that nobody looks at costs nothing: two runs back to back, fields of the second
    a.run(3); a.run(2)
    c = Simulation(nx, ny, 1.2, bc="pipe", inlet_rho=1.01)
    c.set_f(f0); c.run(11)
    for k in ("f", "rho", "u", "v"):
        assert np.array_equal(a.get_fields((k,))[k], c.get_fields((k,))[k]), k


def test_checkpoint_written_lazy_restored_eager_continues_bitwise(lbhip, tmp_path):
    """A checkpoint written by a handle that rebuilds rho, u, v on demand, restored into one that stores them (and the
    other way round): the populations of every later step are the same bits; only the stored fields differ, by rounding."""
    from LB_D2Q9.simulation import Simulation
    nx, ny = 192, 160
    rng = np.random.default_rng(11)
    mask = (rng.random((nx, ny)) < 0.03).astype(np.int32)
    mask[0, :] = mask[-1, :] = 0
    mask[:, 0] = mask[:, -1] = 0
    f0 = _random_state(rng, nx, ny)
    ref = Simulation(nx, ny, 1.5, bc="pipe", inlet_rho=1.004, outlet_rho=1.0, obstacle_mask=mask)
    ref.set_f(f0)
    ref.run(9 + 14)
    want = ref.get_fields(("f",))["f"]
    for first, second in ((False, True), (True, False)):
        a = Simulation(nx, ny, 1.5, bc="pipe", inlet_rho=1.004, outlet_rho=1.0, obstacle_mask=mask, eager_macro=first)
        a.set_f(f0)
        a.run(9)
        path = str(tmp_path / ("ck_%d" % int(first)))
        a.save_checkpoint(path)
        b = Simulation.from_checkpoint(path, eager_macro=second)
        assert b.eager_macro == second
        b.run(14)
        got = b.get_fields(("f", "rho", "u", "v"))
        assert np.array_equal(got["f"], want)
        full = ref.get_fields(("rho", "u", "v"))
        assert maxdiff(got["rho"], full["rho"]) <= 3.6e-7 and maxdiff(got["u"], full["u"]) <= 2.5e-7


@pytest.mark.parametrize("bc,kw", FAMILIES + [("velocity_inlet", {"inlet_u": 0.04})])
@pytest.mark.parametrize("nx,ny", [(67, 29), (1030, 70), (1024, 300)])
def test_health_check_vs_numpy(lbhip, bc, kw, nx, ny):
    from LB_D2Q9.simulation import Simulation
    rng = np.random.default_rng(nx * 7 + ny)
    f0 = _random_state(rng, nx, ny, amp=0.05)
    s = Simulation(nx, ny, 1.5, bc=bc, **kw)
    s.set_f(f0)
    c = s.check()
    n, mach, mass = host_health(f0)
    assert c["n_nonfinite"] == n == 0
    assert abs(c["max_mach"] - mach) <= 2e-6 and abs(c["sum_rho"] - mass) <= 2e-7 * mass
    assert s.check() == c                              # reproducible: fixed reduction order
    # after a run (fields not yet rebuilt in the plain families: the same pass does both)
    s.run(7)
    c = s.check()
    g = s.get_fields(("f", "rho"))
    n, mach, mass = host_health(g["f"])
    assert c["n_nonfinite"] == 0 and abs(c["max_mach"] - mach) <= 2e-6 and abs(c["sum_rho"] - mass) <= 2e-7 * mass
    if bc != "velocity_inlet":
        assert abs(c["sum_rho"] - float(g["rho"].astype(np.float64).sum())) <= 1e-9 * mass    # the stored rho IS what was summed
    # poisoned cells: NaN, +-Inf, and a cell of zero density (velocity undefined)
    bad = f0.copy()
    bad[3, 4, 2] = np.nan
    bad[nx - 1, ny - 1, 0] = np.inf
    bad[10, 0, 7] = -np.inf
    bad[20, 5, :] = 0.0
    bad[21, 5, :] = np.nan
    s.set_f(bad)
    c = s.check()
    n, mach, mass = host_health(bad)
    assert c["n_nonfinite"] == n == 5
    assert abs(c["max_mach"] - mach) <= 2e-6 and abs(c["sum_rho"] - mass) <= 2e-7 * mass
    with pytest.raises(FloatingPointError):
        s.check(raise_nonfinite=True)
    s.close()


def test_mach_warning_like_the_forks(lbhip):
    """porous_media/single_component.py:221-225: 'max_ulb is greater than cs/10!' when max |u| > c_s * mach_tolerance."""
    from LB_D2Q9.simulation import Simulation
    nx, ny = 64, 64
    s = Simulation(nx, ny, 1.0, bc="periodic")
    rho = np.ones((nx, ny), np.float32)
    for speed, expect in ((0.01, False), (0.2, True)):
        s.init_equilibrium(rho, np.full((nx, ny), speed, np.float32), np.zeros((nx, ny), np.float32))
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            c = s.check(warn=True)
        assert abs(c["max_mach"] - speed * np.sqrt(3.0)) <= 1e-5
        assert bool(w) == expect
        if expect:
            assert "cs/10" in str(w[0].message)


def test_d2q9i_class_raises_when_the_fork_diverges(lbhip):
    """The executed D2Q9i fork overflows within tens of steps (tests/golden/o2_d2q9i_53x27 is NaN by step 60; this docs-sized
    cylinder case, inlet density 1.9, within ten: oracle, float64 moments).  By default the module's classes behave like
    the reference's (run() never raises, NaN fields come back); with raise_on_divergence=True they check after every run()
    and raise on the first one that leaves non-finite cells."""
    from LB_D2Q9.dimensionless import opencl_dim_D2Q9i as lb
    args = dict(cylinder_center=[.75, .5], cylinder_radius=.1, verbose=False, diameter=1., rho=1., viscosity=1.,
                pressure_grad=-10., pipe_length=3., N=8)
    np.random.seed(4)
    ref_like = lb.Pipe_Flow_Cylinder(**args)
    ref_like.run(60)                                   # the reference's behaviour: no exception ...
    assert not np.all(np.isfinite(ref_like.get_fields()["f"]))      # ... and non-finite populations handed back
    np.random.seed(4)
    c = lb.Pipe_Flow_Cylinder(raise_on_divergence=True, **args)
    c.run(2)                                           # still finite
    assert np.all(np.isfinite(c.get_fields()["f"])) and c.check()["n_nonfinite"] == 0
    steps = 2
    with pytest.raises(FloatingPointError):
        while steps < 60:
            c.run(1)
            steps += 1
    assert 3 <= steps < 60
    assert c.check()["n_nonfinite"] > 0               # (populations ~1e19 and growing: the velocity's square has overflowed)


def test_check_across_ranks_single_rank_ring(lbhip):
    """lb_check(across_ranks=1): ncclAllReduce of the three scalars over the communicator of the slab path (one rank here)."""
    from LB_D2Q9.simulation import Simulation, comm_unique_id
    nx, ny = 1024, 160
    rng = np.random.default_rng(9)
    f0 = _random_state(rng, nx, ny)
    s = Simulation(nx, ny, 1.3, bc="periodic", halo=True)
    s.comm_init(comm_unique_id(), 0, 1)
    s.set_f(f0)
    s.run(20)
    local, glob = s.check(), s.check(across_ranks=True)
    assert local["n_nonfinite"] == glob["n_nonfinite"] == 0
    assert local["sum_rho"] == glob["sum_rho"] and local["max_mach"] == glob["max_mach"]
    n, mach, mass = host_health(s.get_fields(("f",))["f"])
    assert abs(glob["max_mach"] - mach) <= 2e-6 and abs(glob["sum_rho"] - mass) <= 2e-7 * mass
    with pytest.raises(Exception):
        Simulation(64, 64, 1.0, bc="periodic").check(across_ranks=True)      # no communicator
