"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (fp32, stated by SURVEY.md section 8c / Appendix D and re-measured here):
  single phase / single step from identical f : |d rho| <= 5e-7, |d u|,|d v| <= 1e-6, |d f| <= 2.5e-7
                                                 (2.5e-7 = 4 ulp of the rest population 4/9)
  <= 1000 laminar steps                        : |d rho| <= 1e-5, |d u|,|d v| <= 5e-6
The HIP kernels use idiomatic fp32 (FMA, 3*cu instead of cu/cs2, float literals) where the reference
OpenCL source has double literals and divisions, hence "within tolerance", not bit equality.
"""
import os

import numpy as np
import pytest

from conftest import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

TOL1 = dict(f=2.5e-7, feq=2.5e-7, rho=5e-7, u=1e-6, v=1e-6)
TOLN = dict(f=1e-5, feq=1e-5, rho=1e-5, u=5e-6, v=5e-6)


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def assert_fields_close(got, want, tol, prefix=""):
    """max |got - want| <= tol per field; the message (and, with -s / -rP, the test's output) carries the measured margins."""
    meas = {k: maxdiff(got[k], want[prefix + k]) for k in tol}
    report = ", ".join("%s %.2e / %.1e (%.0f %%)" % (k, meas[k], tol[k], 100. * meas[k] / tol[k]) for k in tol)
    print("measured / bound:", report)
    for k, t in tol.items():
        assert meas[k] <= t, "%s: max abs diff %.3e > %.1e  [all fields: %s]" % (k, meas[k], t, report)


def contract_tol(n_steps):
    """The parity contract as a function of the number of time steps compared, written ONCE: the
    single-step bounds -- max |df| <= 2.5e-7, |drho| <= 5e-7, |du|, |dv| <= 1e-6 from identical populations, the last being the
    reference's own criterion (testing/Bryan/opencl_check_03.ipynb:593, 778) -- times n, capped at the contract's envelope for
    <= 1000 laminar steps (|drho| <= 1e-5, |du|, |dv| <= 5e-6; f as u).  Not fitted to any kernel: a kernel whose arithmetic
    cannot meet n x the single-step bound fails here."""
    n = max(1, int(n_steps))
    return dict(f=min(2.5e-7 * n, 5e-6), rho=min(5e-7 * n, 1e-5), u=min(1e-6 * n, 5e-6), v=min(1e-6 * n, 5e-6))


def make_sim(d, lbhip, bc="pipe", **kw):
    from LB_D2Q9.simulation import Simulation
    mask = d["mask"] if "mask" in d.files else None
    return Simulation(int(d["nx"]), int(d["ny"]), float(d["omega"]), bc=bc, inlet_rho=float(d["inlet_rho"]),
                      outlet_rho=float(d["outlet_rho"]), obstacle_mask=mask, **kw)


# ---- every reference kernel, one call each, against the executed D2Q9.cl -------------------------
def test_unfused_phases_match_reference_kernels(lbhip):
    d = golden("o2_kernels_37x19")
    sim = make_sim(d, lbhip)
    sim.set_f(d["f0"])
    sim.move_bcs()                                   # move_bcs + bounceback_in_obstacle
    # the golden has them separately: compose on the host
    want = d["after_bcs_f"].copy()
    m = d["mask"].astype(bool)
    for a, b in ((1, 3), (2, 4), (5, 7), (6, 8)):
        ta, tb = want[..., a][m].copy(), want[..., b][m].copy()
        want[..., a][m], want[..., b][m] = tb, ta
    assert maxdiff(sim.get_fields(("f",))["f"], want) <= 2.5e-7

    sim.set_f(d["f0"])
    sim.update_hydro()
    g = sim.get_fields(("rho", "u", "v"))
    assert maxdiff(g["rho"], d["hydro_rho"]) <= 5e-7
    assert maxdiff(g["u"], d["hydro_u"]) <= 1e-6
    assert maxdiff(g["v"], d["hydro_v"]) <= 1e-6
    sim.update_feq()
    assert maxdiff(sim.get_fields(("feq",))["feq"], d["feq"]) <= 2.5e-7
    sim.collide_particles()
    assert maxdiff(sim.get_fields(("f",))["f"], d["after_collide_f"]) <= 2.5e-7
    sim.zero_velocity_in_obstacle()
    g = sim.get_fields(("u", "v"))
    assert maxdiff(g["u"], d["zeroed_u"]) <= 1e-6 and maxdiff(g["v"], d["zeroed_v"]) <= 1e-6
    assert np.all(g["u"][m] == 0) and np.all(g["v"][m] == 0)


def test_move_keeps_stale_entries_like_reference(lbhip, oracle):
    """kernels.move drops out-of-box targets; f_streamed keeps its previous value there."""
    d = golden("o2_kernels_37x19")
    sim = make_sim(d, lbhip)
    sim.set_f(d["f0"])                                # f_streamed = f0 as well
    sim.move()
    o = oracle.O2Sim(int(d["nx"]), int(d["ny"]), 1.0)
    o.set_f(d["f0"])
    o.move()
    assert maxdiff(sim.get_fields(("f",))["f"], o.get_fields()["f"]) == 0.0


# ---- fused run() against the executed reference --------------------------------------------------
@pytest.mark.parametrize("name,steps", [("o2_pipe_N10", (1, 10, 200, 999)),
                                        ("o2_pipe_noise_49x25", (1, 100, 1000)),
                                        ("o2_cyl_61x31", (1, 100, 500))])
def test_fused_run_matches_reference(lbhip, name, steps):
    d = golden(name)
    sim = make_sim(d, lbhip)
    sim.set_f(d["f0"])
    done = 0
    for n in steps:
        sim.run(n - done)
        done = n
        assert_fields_close(sim.get_fields(), d, TOL1 if n == 1 else TOLN, "s%d_" % n)


def test_fused_equals_unfused_sequence(lbhip):
    """run(1) == move, move_bcs, update_hydro, update_feq, collide_particles (opencl_dim.py:380-387)."""
    d = golden("o2_cyl_61x31")
    a, b = make_sim(d, lbhip), make_sim(d, lbhip)
    a.set_f(d["f0"]); b.set_f(d["f0"])
    for _ in range(5):
        a.run(1)
        b.move(); b.move_bcs(); b.update_hydro(); b.update_feq(); b.collide_particles()
    ga, gb = a.get_fields(), b.get_fields()
    for k in ("f", "feq", "rho", "u", "v"):
        assert maxdiff(ga[k], gb[k]) <= 1e-6, k


def test_fused_opencl_path_kernel_vs_imported_cython_reference_interior(lbhip):
    """The strictly pinned oracle leg is O1 (the imported, cythonized reference).  Away from walls the two reference
    paths are the same scheme (stream + BGK), which is how the reference authors checked their OpenCL port against
    their CPU code: |difference| <= 1e-6 outside a 3-cell margin (testing/Bryan/opencl_check_03.ipynb:593, 778).
    Here the fused HIP kernel takes one step from the state the imported reference had before its third step
    (fixture o1_pipe_33x17: pre_* -> t_collide_f / t_hydro_*) and must agree with it in the interior."""
    from LB_D2Q9.simulation import Simulation
    d = golden("o1_pipe_33x17")
    nx, ny = int(d["nx"]), int(d["ny"])
    for variant in (-1, 0, 512):                       # automatic choice, single-step kernel, LDS-tile kernel (1 step = k_step)
        sim = Simulation(nx, ny, float(d["omega"]), bc="pipe", inlet_rho=float(d["inlet_rho"]), outlet_rho=float(d["outlet_rho"]))
        sim.set_variant(variant)
        sim.set_f(np.asarray(d["pre_f"]).transpose(1, 2, 0))        # reference layout (9, nx, ny) -> (nx, ny, 9)
        sim.run(1)
        g = sim.get_fields(("f", "rho", "u", "v"))
        inner = (slice(3, -3), slice(3, -3))
        assert maxdiff(g["f"][inner], np.asarray(d["t_collide_f"]).transpose(1, 2, 0)[inner]) <= 1e-6
        assert maxdiff(g["rho"][inner], d["t_hydro_rho"][inner]) <= 1e-6
        assert maxdiff(g["u"][inner], d["t_hydro_u"][inner]) <= 1e-6
        assert maxdiff(g["v"][inner], d["t_hydro_v"][inner]) <= 1e-6
        sim.close()


# ---- build-defined boundary families against the oracle -------------------------------------------
def _random_state(rng, nx, ny, amp=0.02):
    w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    return (w[None, None, :] * (1 + amp * rng.standard_normal((nx, ny, 9)))).astype(np.float32)


@pytest.mark.parametrize("bc,kw", [("periodic", {}), ("cavity", {"lid_u": 0.08, "rho0": 1.0}),
                                   ("pipe", {"inlet_rho": 1.01, "outlet_rho": 1.0})])
@pytest.mark.parametrize("nx,ny", [(64, 32), (67, 29), (5, 7), (256, 3), (130, 130)])
def test_bc_families_vs_oracle(lbhip, oracle, bc, kw, nx, ny):
    from LB_D2Q9.simulation import Simulation
    rng = np.random.default_rng(nx * 1000 + ny)
    f0 = _random_state(rng, nx, ny)
    mask = (rng.random((nx, ny)) < 0.05)
    mask[0, :] = mask[-1, :] = False
    mask[:, 0] = mask[:, -1] = False
    omega = 1.3
    sim = Simulation(nx, ny, omega, bc=bc, obstacle_mask=mask, **kw)
    code = {"pipe": oracle.BC_PIPE, "periodic": oracle.BC_PERIODIC, "cavity": oracle.BC_CAVITY}[bc]
    o = oracle.O2Sim(nx, ny, omega, code, kw.get("inlet_rho", 1.), kw.get("outlet_rho", 1.),
                     kw.get("lid_u", 0.), kw.get("rho0", 1.), mask=mask)
    sim.set_f(f0); o.set_f(f0)
    sim.run(1); o.run(1)
    assert_fields_close(sim.get_fields(), o.get_fields(), TOL1)
    sim.run(49); o.run(49)
    assert_fields_close(sim.get_fields(), o.get_fields(), TOLN)


def test_periodic_mass_drift_tracks_reference(lbhip, oracle):
    """Size-independent property: a periodic box conserves sum(rho) up to fp32 rounding bias.  The
    reference arithmetic itself drifts by about +9e-9 per step (sum of the float32 weights is
    1 + 7.5e-9); the engine must stay within 2e-8 per step of exact conservation and, on a grid the
    oracle can run, within 5e-9 per step of the oracle's own drift."""
    from LB_D2Q9.simulation import Simulation
    steps = 200
    for n in (256, 1024):
        rng = np.random.default_rng(1)
        f0 = _random_state(rng, n, n, 0.01)
        sim = Simulation(n, n, 1.7, bc="periodic")
        sim.set_f(f0)
        m0 = f0.astype(np.float64).sum()
        sim.run(steps)
        drift = (sim.get_fields(("f",))["f"].astype(np.float64).sum() - m0) / m0 / steps
        assert abs(drift) < 2e-8, drift
        if n == 256:
            o = oracle.O2Sim(n, n, 1.7, oracle.BC_PERIODIC)
            o.set_f(f0)
            o.run(steps)
            odrift = (o.f.astype(np.float64).sum() - m0) / m0 / steps
            assert abs(drift - odrift) < 5e-9, (drift, odrift)


# ---- the two-steps-per-pass kernel ------------------------------------------------------------------
@pytest.mark.parametrize("bc,nx,ny", [("periodic", 1000, 130), ("periodic", 512, 128), ("pipe", 1003, 177),
                                      ("cavity", 777, 201), ("pipe", 2048, 300)])
@pytest.mark.parametrize("masked", [False, True])
def test_two_step_kernel_equals_single_step_kernel(lbhip, oracle, bc, nx, ny, masked):
    """variant bit 5 selects k_step2 (two time steps per pass, register window + lane shuffles), bits 6 / 8 / 12 the three-,
    four- and five-step kernels, bit 9 the LDS tiles.  Same
    per-cell arithmetic as k_step and the library is built with -ffp-contract=on, so the fields must
    be bitwise equal to the single-step kernel's, and match the oracle."""
    from LB_D2Q9.simulation import Simulation
    rng = np.random.default_rng(nx + ny)
    f0 = _random_state(rng, nx, ny)
    mask = None
    if masked:
        mask = rng.random((nx, ny)) < 0.03
        mask[0, :] = mask[-1, :] = False
        mask[:, 0] = mask[:, -1] = False
    kw = dict(inlet_rho=1.004, lid_u=0.06)
    sims = []
    # single step / two-step / + NT stores / three-step (+ two-step remainder) / four-step (+ remainders) /
    # four steps through LDS tiles (+ single-step remainders) / five-step on overlapping strips (+ remainders)
    for variant in (0, 32, 33, 97, 97 | 256, 512, 97 | 256 | 4096, 97 | 256 | 4096 | 16384, 97 | 256 | 4096 | 16384 | 32768,
                    97 | 256 | 4096 | 16384 | 32768 | 65536):                     # (the last one: k_deep2)
        s = Simulation(nx, ny, 1.6, bc=bc, obstacle_mask=mask, **kw)
        s.set_variant(variant)
        assert s.steps_per_launch() == {0: 1, 32: 2, 33: 2, 97: 3, 353: 4, 512: 4, 4449: 5, 20833: 6, 53601: 7, 119137: 7}[variant]
        s.set_f(f0)
        s.run(7)                      # 7 = 1+2+2+2 (two-step) = 1+3+3 (three-step) = 3+4 (four-step) = 2+5 (five-step) = 1+6 = 7
        s.run(4)                      # 4 = 2+2 = 1+3 = 4
        sims.append(s.get_fields(("f", "rho", "u", "v")))
    for k in ("f", "rho", "u", "v"):
        assert np.array_equal(sims[0][k], sims[1][k]), k          # -ffp-contract=on: same rounding in every kernel
        assert np.array_equal(sims[1][k], sims[2][k]), k          # NT stores never change results
        assert np.array_equal(sims[0][k], sims[3][k]), k          # three steps per pass: still bitwise
        assert np.array_equal(sims[0][k], sims[4][k]), k          # four steps per pass (LDS windows): still bitwise
        assert np.array_equal(sims[0][k], sims[5][k]), k          # four steps per pass (LDS tiles): still bitwise
        assert np.array_equal(sims[0][k], sims[6][k]), k          # five steps per pass (overlapping strips): still bitwise
        assert np.array_equal(sims[0][k], sims[7][k]), k          # six steps per pass (k_deep<6>, one wave per SIMD): still bitwise
        assert np.array_equal(sims[0][k], sims[8][k]), k          # seven steps per pass (k_deep<7>): still bitwise
    code = {"pipe": oracle.BC_PIPE, "periodic": oracle.BC_PERIODIC, "cavity": oracle.BC_CAVITY}[bc]
    o = oracle.O2Sim(nx, ny, 1.6, code, 1.004, 1., 0.06, 1., mask=mask)
    o.set_f(f0)
    o.run(11)
    assert_fields_close(sims[1], o.get_fields(), dict(f=2e-6, rho=2e-6, u=2e-6, v=2e-6))


def test_run_is_split_into_the_cheapest_launches(lbhip):
    """lb_plan_launches: how lb_run(n) splits n steps into launches -- a launch of a marching kernel costs about the same whatever
    it fuses, so the split minimises their summed cost (shallow ones first; costs: what lb_autotune timed on the handle, else the
    seeds of launch_costs): 20 steps with depths up to seven = 6 + 7 + 7, up to six = 4 + 4 + 6 + 6, up to five = 4 x 5; the plan
    sums to n, and run(n) of every plan equals the single-step kernel (the other tests)."""
    from LB_D2Q9.simulation import Simulation
    s = Simulation(2560, 2560, 1.5, bc="periodic")
    assert s.steps_per_launch() == 7
    assert s.plan_launches(20) == [6, 7, 7] and s.plan_launches(56) == [7] * 8 and s.plan_launches(5) == [5]
    assert s.plan_launches(8) == [4, 4] and s.plan_launches(0) == [] and s.plan_launches(27) == [6, 7, 7, 7]
    for n in (1, 2, 3, 11, 13, 29, 64, 65, 100, 131):
        p = s.plan_launches(n)
        assert sum(p) == n and all(1 <= d <= 7 for d in p) and (n > 64 or p == sorted(p)), (n, p)
    s.set_variant(353 | 4096 | 16384)
    assert s.plan_launches(20) == [4, 4, 6, 6] and s.plan_launches(60) == [6] * 10 and s.plan_launches(23) == [5, 6, 6, 6]
    s.set_variant(353 | 4096)
    assert s.plan_launches(20) == [5] * 4 and s.plan_launches(23) == [4, 4, 5, 5, 5]
    s.set_variant(9)
    assert s.plan_launches(4) == [1, 1, 1, 1]
    slab = Simulation(1024, 256, 1.5, bc="periodic", y0=0, local_ny=128)
    assert slab.plan_launches(8) is None


def test_launch_plan_follows_the_costs_autotune_measured(lbhip):
    """lb_autotune leaves the handle the launch cost of every depth it timed (launch_costs in lb_hip.cpp); lb_plan_launches then
    splits runs by THOSE costs: the plan still sums to n, uses no depth beyond the tuned one, and run(n) by that plan gives the
    single-step kernel's bits."""
    from LB_D2Q9.simulation import Simulation
    nx, ny = 2560, 2304
    rng = np.random.default_rng(5)
    f0 = _random_state(rng, nx, ny)
    a = Simulation(nx, ny, 1.6, bc="periodic")
    a.set_f(f0)
    used = a.autotune()
    deepest = a.steps_per_launch()
    assert used > 0 and 1 <= deepest <= 7
    for n in (1, 5, 13, 20, 41, 64, 84):
        p = a.plan_launches(n)
        assert sum(p) == n and max(p) <= deepest and min(p) >= 1, (n, p)      # (the order follows the measured costs)
    a.run(20)
    a.run(13)
    b = Simulation(nx, ny, 1.6, bc="periodic")
    b.set_variant(0)
    b.set_f(f0)
    b.run(used + 33)
    ga, gb = a.get_fields(("f",)), b.get_fields(("f",))
    assert np.array_equal(ga["f"], gb["f"])


def test_tune_cache_hands_a_result_to_the_next_handle_of_the_same_shape(lbhip, tmp_path, monkeypatch):
    """LB_TUNE_CACHE=<file>: what lb_autotune found on one handle (kernel depth, waves per CU, the measured launch costs the plan is
    made from) is taken over by the first run of a later handle of the same shape -- in this process and, through the file, in another
    one -- without a tuning step; another shape is not touched; results stay the single-step kernel's bits."""
    import subprocess
    import sys
    from LB_D2Q9.simulation import Simulation
    cache = tmp_path / "tune.txt"
    monkeypatch.setenv("LB_TUNE_CACHE", str(cache))
    nx, ny = 1792, 1536
    rng = np.random.default_rng(11)
    f0 = _random_state(rng, nx, ny)
    a = Simulation(nx, ny, 1.5, bc="periodic")
    a.set_f(f0)
    used = a.autotune()
    assert used > 0
    want = (a.steps_per_launch(), a.hot_kernel(), a.plan_launches(41))
    lines = cache.read_text().strip().splitlines()
    assert len(lines) == 1 and ":%dx%d:" % (nx, ny) in lines[0]
    b = Simulation(nx, ny, 1.5, bc="periodic")          # same shape: tuned by its first run, no step spent on it
    b.set_f(f0)
    b.run(used + 9)
    assert (b.steps_per_launch(), b.hot_kernel(), b.plan_launches(41)) == want
    a.run(9)
    assert np.array_equal(a.get_fields(("f",))["f"], b.get_fields(("f",))["f"])
    c = Simulation(nx, ny + 64, 1.5, bc="periodic")     # another shape: the heuristic, as without the cache
    c.init_equilibrium(np.ones((nx, ny + 64), np.float32), np.zeros((nx, ny + 64), np.float32), np.zeros((nx, ny + 64), np.float32))
    c.run(3)
    monkeypatch.delenv("LB_TUNE_CACHE")
    d = Simulation(nx, ny + 64, 1.5, bc="periodic")
    d.run(3)
    assert c.hot_kernel() == d.hot_kernel() and c.plan_launches(41) == d.plan_launches(41)
    # another process reads the file
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from LB_D2Q9.simulation import Simulation; "
            "s = Simulation(%d, %d, 1.5, bc='periodic'); s.run(2); print(s.steps_per_launch(), '|', s.hot_kernel(), '|', s.plan_launches(41))"
            % (os.path.join(ROOT, "2d-lb_amd"), nx, ny))
    env = dict(os.environ, LB_TUNE_CACHE=str(cache))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "%s | %s | %s" % want


@pytest.mark.parametrize("nx", [512, 716, 720, 724, 740, 744, 748, 960, 964, 992, 996, 1000, 1196, 1236, 1241, 1440, 1488])
@pytest.mark.parametrize("bc", ["periodic", "pipe", "cavity"])
def test_five_and_six_step_kernel_strip_boundaries(lbhip, bc, nx):
    """k_step5 / k_deep<6>, k_deep<7> march overlapping strips laid 240 cells apart (k_step5 until the end of round 5: 248), each starting 8
    cells early: widths around the multiples of 248 and 240 (the last strip stores a few cells, or none + a whole strip; odd widths in the walled families),
    heights around the segment sizes, with an obstacle mask whose solid cells sit on the strip seams, against the single-step
    kernel, bit for bit."""
    from LB_D2Q9.simulation import Simulation
    if bc == "periodic" and nx % 4:
        nx += 4 - nx % 4
    ny = 128 + (nx % 7) * 9
    rng = np.random.default_rng(nx)
    f0 = _random_state(rng, nx, ny)
    mask = rng.random((nx, ny)) < 0.02
    for seam in list(range(244, nx - 1, 248)) + list(range(236, nx - 1, 240)):     # the last stored cells of a strip and the first of the next
        mask[seam:seam + 8, ::3] = True
    if bc != "periodic":
        mask[0, :] = mask[-1, :] = False
        mask[:, 0] = mask[:, -1] = False
    out = []
    for variant in (0, 97 | 256 | 4096, 97 | 256 | 4096 | 16384, 97 | 256 | 4096 | 16384 | 32768, 97 | 256 | 4096 | 16384 | 32768 | 65536):
        s = Simulation(nx, ny, 1.55, bc=bc, obstacle_mask=mask, inlet_rho=1.003, lid_u=0.05)
        s.set_variant(variant)
        if variant:
            spl = 7 if variant & 32768 else (6 if variant & 16384 else 5)
            assert s.steps_per_launch() == spl and ("k_step5" if spl == 5 else ("k_deep2<7>" if variant & 65536 else "k_deep<%d>" % spl)) in s.hot_kernel()
        s.set_f(f0)
        s.run(12)
        s.run(7)
        out.append(s.get_fields(("f", "rho", "u", "v")))
        s.close()
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k]) and np.array_equal(out[0][k], out[2][k]) and np.array_equal(out[0][k], out[3][k]), k


@pytest.mark.parametrize("bc,nx,ny", [("pipe", 96, 64), ("periodic", 64, 96), ("cavity", 130, 70), ("periodic", 256, 256),
                                      ("pipe", 301, 101)])
@pytest.mark.parametrize("masked", [False, True])
def test_tile_kernel_equals_single_step_kernel_small_grids(lbhip, bc, nx, ny, masked):
    """k_tile4 (variant bit 9): four time steps per pass inside 32 x 16 LDS tiles, for the small grids the
    marching kernels do not serve; tile edges that are not multiples of 32, periodic images, walls, masks; one band of
    tile rows per XCD (default; tile counts that do not divide by eight) and tiles in launch order (variant bit 13)."""
    from LB_D2Q9.simulation import Simulation
    rng = np.random.default_rng(7 * nx + ny)
    f0 = _random_state(rng, nx, ny)
    mask = None
    if masked:
        mask = rng.random((nx, ny)) < 0.05
        if bc != "periodic":
            mask[0, :] = mask[-1, :] = False
            mask[:, 0] = mask[:, -1] = False
    kw = dict(inlet_rho=1.004, lid_u=0.06)
    out = []
    for variant in (0, 512, 512 | 8192):
        s = Simulation(nx, ny, 1.45, bc=bc, obstacle_mask=mask, **kw)
        s.set_variant(variant)
        s.set_f(f0)
        s.run(9)                      # 9 = 1 + 4 + 4
        s.run(8)
        out.append(s.get_fields(("f", "rho", "u", "v")))
        s.close()
    for k in ("f", "rho", "u", "v"):
        assert np.array_equal(out[0][k], out[1][k]), k
        assert np.array_equal(out[0][k], out[2][k]), k


# ---- row slabs ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("bc", ["periodic", "pipe", "cavity"])
@pytest.mark.parametrize("nslabs", [2, 3])
def test_virtual_slabs_equal_single_slab_bitwise(lbhip, bc, nslabs):
    """G row slabs on one device, halos moved with lb_halo_export/import, must equal the
    one-slab run bit for bit (same arithmetic, different partition)."""
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import LocalSlabRing
    nx, ny = 96, 50
    rng = np.random.default_rng(5)
    f0 = _random_state(rng, nx, ny)
    mask = rng.random((nx, ny)) < 0.04
    mask[:, 0] = mask[:, -1] = False
    kw = dict(inlet_rho=1.01, lid_u=0.05)
    one = Simulation(nx, ny, 1.5, bc=bc, obstacle_mask=mask, **kw)
    one.set_f(f0)
    ring = LocalSlabRing(nx, ny, 1.5, nslabs, bc=bc, obstacle_mask=mask, **kw)
    ring.set_f(f0)
    one.run(25)
    ring.run(25)
    a, b = one.get_fields(("f", "rho", "u", "v")), ring.get_fields(("f", "rho", "u", "v"))
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("bc", ["periodic", "pipe", "cavity"])
@pytest.mark.parametrize("nslabs", [2, 3])
def test_in_library_slab_schedule_with_two_step_kernel_bitwise(lbhip, bc, nslabs):
    """lb_run_group = the multi-GPU schedule (edge bands on a priority stream, interior on the compute
    stream, six-step halo cycles with the 6-deep halo, launch-by-launch exchange of the 3-deep halo for
    the remainder) with device-to-device copies instead of RCCL.  Must equal the undivided run bit for
    bit, for step counts that are and are not multiples of six, three and two."""
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import LocalSlabRing
    nx, ny = 1000, 137 if nslabs == 2 else 345          # (three slabs of 115 rows: the fourteen-step cycle needs >= 112)
    rng = np.random.default_rng(17)
    f0 = _random_state(rng, nx, ny)
    mask = rng.random((nx, ny)) < 0.03
    mask[0, :] = mask[-1, :] = False
    if bc != "periodic":
        mask[:, 0] = mask[:, -1] = False
    kw = dict(inlet_rho=1.006, lid_u=0.05)
    one = Simulation(nx, ny, 1.55, bc=bc, obstacle_mask=mask, **kw)
    one.set_variant(0)
    one.set_f(f0)
    # fourteen- / twelve-step cycles of k_deep (slabs of >= 112 / 96 rows; else what fits) / ten-step (k_step5) / eight-step
    # cycles (slabs of >= 64 rows, else six-step) / six-step cycles / three-step launches without the cycle / two-step /
    # single-step kernels on slabs
    first = True
    for variant in (97 | 256 | 4096 | 16384 | 32768, 97 | 256 | 4096 | 16384, 97 | 256 | 4096, 97 | 256, 97, 97 | 128, 33, 1):
        ring = LocalSlabRing(nx, ny, 1.55, nslabs, bc=bc, obstacle_mask=mask, **kw)
        ring.set_variant(variant)
        ring.set_f(f0)
        ring.run_in_library(20)                   # e.g. 3 six-step cycles + 2 steps; one fourteen-step cycle + a lone... + remainder
        ring.run_in_library(7)                    # 1 cycle + 1 step / one lone seven-step half
        ring.run_in_library(4)
        if first:
            one.run(31)
            first = False
        a, b = one.get_fields(("f", "rho", "u", "v")), ring.get_fields(("f", "rho", "u", "v"))
        for k in a:
            assert np.array_equal(a[k], b[k]), (variant, k)


def test_rccl_self_ring_cycles_with_mask(lbhip):
    """1-rank periodic ring over RCCL inside lb_run, obstacle mask, every transition between the six-step
    cycle and the launch-by-launch schedule."""
    from LB_D2Q9.simulation import Simulation, comm_unique_id
    nx, ny = 1024, 160
    rng = np.random.default_rng(23)
    f0 = _random_state(rng, nx, ny)
    mask = rng.random((nx, ny)) < 0.02
    one = Simulation(nx, ny, 1.3, bc="periodic", obstacle_mask=mask)
    one.set_variant(0)
    one.set_f(f0)
    one.run(61 + 29 + 4 + 5 + 16 + 7)
    for variant in (97 | 256 | 4096 | 16384 | 32768, 97 | 256 | 4096 | 16384, 97 | 256, 97, 97 | 128):      # fourteen-, twelve-, eight-step cycle, six-step cycle, no cycle
        two = Simulation(nx, ny, 1.3, bc="periodic", obstacle_mask=mask, halo=True)
        two.set_variant(variant)
        two.comm_init(comm_unique_id(), 0, 1)
        two.set_f(f0)
        two.run(5)                                # launch by launch: 3 + 2
        two.run(14)                               # 2 cycles + 2 steps (3-deep ghosts -> deep exchange first)
        two.run(6)                                # 1 cycle
        two.run(12)                               # 2 cycles on valid 6-deep ghosts
        two.run(4)
        two.run(20)                               # the bench block: two eight-step cycles + one lone four-step half (or 3 x 6 + 2)
        two.run(29)                               # ... + a lone first half that is not the last launch + launch by launch
        two.run(4)                                # six-step cycle: a first half that is not the last launch, from 3-deep ghosts
        two.run(5)                                #   (it reads 2D deep: must exchange first -- was wrong until tools/ring_stress.py)
        two.run(16)                               # full cycles only: leaves 2D-deep ghosts
        two.run(7)                                # eight-step cycle: first half, not last, from 8-deep ghosts
        a, b = one.get_fields(("f", "rho", "u", "v")), two.get_fields(("f", "rho", "u", "v"))
        for k in a:
            assert np.array_equal(a[k], b[k]), (variant, k)
        two.close()


@pytest.mark.parametrize("bc", ["pipe", "cavity"])
def test_slab_schedule_inside_lb_run_wall_families_single_rank(lbhip, bc):
    """lb_run's own slab schedule (edge bands, halo cycles, the lone first half, launch-by-launch remainders, MACRO on the
    last launch) for the walled families: a whole grid flagged as a slab with a 1-rank RCCL communicator has no neighbour
    (the exchange degenerates to an empty group), so the result must equal the plain run bit for bit."""
    from LB_D2Q9.simulation import Simulation, comm_unique_id
    nx, ny = 1024, 200
    rng = np.random.default_rng(31)
    f0 = _random_state(rng, nx, ny)
    mask = rng.random((nx, ny)) < 0.02
    mask[0, :] = mask[-1, :] = False
    mask[:, 0] = mask[:, -1] = False
    kw = dict(inlet_rho=1.004, lid_u=0.05)
    one = Simulation(nx, ny, 1.4, bc=bc, obstacle_mask=mask, **kw)
    one.set_variant(0)
    one.set_f(f0)
    one.run(20 + 7 + 4 + 9)
    want = one.get_fields(("f", "rho", "u", "v"))
    for variant in (97 | 256 | 4096 | 16384 | 32768 | 65536, 97 | 256 | 4096 | 16384 | 32768, 97 | 256 | 4096 | 16384, 97 | 256, 97, 33):
        s = Simulation(nx, ny, 1.4, bc=bc, obstacle_mask=mask, halo=True, **kw)
        s.set_variant(variant)
        s.comm_init(comm_unique_id(), 0, 1)
        s.set_f(f0)
        for n in (20, 7, 4, 9):
            s.run(n)
        got = s.get_fields(("f", "rho", "u", "v"))
        for k in want:
            assert np.array_equal(got[k], want[k]), (variant, k)
        s.close()


def test_rccl_self_exchange_single_rank(lbhip):
    """One rank, periodic box split as a 'slab' talking to itself over RCCL: exercises
    lb_comm_init + the in-run exchange path on a single GPU."""
    import ctypes as ct
    from LB_D2Q9 import _native
    from LB_D2Q9.simulation import Simulation
    nx, ny = 128, 64
    rng = np.random.default_rng(9)
    f0 = _random_state(rng, nx, ny)
    one = Simulation(nx, ny, 1.2, bc="periodic")
    one.set_f(f0)
    one.run(10)
    # halo=True: the whole-grid handle fills its ghost rows through the halo path, i.e. a 1-rank
    # periodic ring that sends to itself over RCCL inside lb_run
    two = Simulation(nx, ny, 1.2, bc="periodic", halo=True)
    uid = (ct.c_char * 128)()
    _native.check(lbhip.lb_comm_unique_id(uid))
    _native.check(lbhip.lb_comm_init(two._h, uid, 0, 1))
    two.set_f(f0)
    two.run(10)
    assert np.array_equal(one.get_fields(("f",))["f"], two.get_fields(("f",))["f"])


@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_slab_cycle_depth_set_by_the_caller_and_exchange_timing(lbhip, transport):
    """lb_set_slab_cycle (ABI 9), lb_set_exchange_inline (ABI 10): the halo cycle of lb_run on the depth the caller fixes -- what DistributedSlab.autotune does after
    the ranks have timed the candidates together -- gives the plain run's bits at every depth, over both transports (thick edge bands;
    under RCCL split in two launches with the exchange on the communication stream: lb_hip.cpp, slab_cycle_first), with an obstacle
    mask; lb_exchange_timing / lb_exchange_stats count the exchanges and report the cycle in use."""
    from LB_D2Q9.simulation import Simulation, comm_unique_id
    from LB_D2Q9.slabs import _SlabSet
    nx, ny = 1024, 480
    rng = np.random.default_rng(77)
    f0 = _random_state(rng, nx, ny)
    mask = rng.random((nx, ny)) < 0.02
    one = Simulation(nx, ny, 1.5, bc="periodic", obstacle_mask=mask)
    one.set_variant(0)
    one.set_f(f0)
    one.run(3 * 28 + 9)
    want = one.get_fields(("f",))["f"]
    one.close()
    # (ABI 10: lb_set_exchange_inline -- the exchange between the interior launches on the compute stream instead of beside them)
    # (depth 8: the seven-step cycle by k_deep2, interior and edge bands alike, as the collective tuner sets it)
    for depth, inline in ((7, False), (7, True), (8, False), (8, True), (6, False), (6, True), (5, True), (5, False), (4, False), (3, False)):
        s = Simulation(nx, ny, 1.5, bc="periodic", obstacle_mask=mask, halo=True)
        s.set_obstacle_mask_halo(*_SlabSet._mask_halo_rows(mask, 0, ny, ny, True))
        s.set_variant(97 | 256 | 4096 | 16384 | 32768)        # every marching kernel allowed; the cycle's depth (and k_deep / k_deep2) is the caller's
        if transport == "rccl":
            s.comm_init(comm_unique_id(), 0, 1)
        else:
            d = s.peer_export()
            s.peer_connect(0, 1, d, d, ny)
        s.set_slab_cycle(depth)
        s.set_exchange_inline(inline)
        s.exchange_timing(True)
        s.set_f(f0)
        for n in (28, 28, 28, 9):
            s.run(n)
        st = s.exchange_stats()
        assert st["cycle_depth"] == min(depth, 7) and st["n"] >= 3 and st["total_ms"] > 0 and st["max_ms"] <= st["total_ms"]
        assert st["band_rows"] >= 2 * min(depth, 7)
        assert s.exchange_stats()["n"] == 0                    # (the query starts over)
        got = s.get_fields(("f",))["f"]
        assert np.array_equal(got, want), (transport, depth, inline)
        s.close()


# ---- the drop-in classes ---------------------------------------------------------------------------
def test_pipe_flow_class_poiseuille_kat(lbhip):
    """The reference's own known-answer test (docs/opencl_dimensionless_verification.ipynb:632-696):
    steady plane-Poiseuille profile u(y) = (1/(2 rho nu)) gradP y (y - D), N = 10."""
    from LB_D2Q9.dimensionless import opencl_dim as lb
    D, rho, nu, gradP = 1.5, 10., 5., -100.
    sim = lb.Pipe_Flow(diameter=D, rho=rho, viscosity=nu, pressure_grad=gradP, pipe_length=3., N=10,
                       verbose=False)
    sim.init_pop(amplitude=0.)
    steps = int(10. / (sim.delta_t * sim.T)) if False else 999
    sim.run(steps)
    u = sim.get_physical_fields()["u"]
    prof = u[sim.nx // 2, :]
    y = np.linspace(0, D, sim.ny)
    theory = (1. / (2 * rho * nu)) * gradP * y * (y - D)
    assert prof[0] == pytest.approx(0., abs=1e-6) and prof[-1] == pytest.approx(0., abs=1e-6)
    assert np.max(np.abs(prof - theory)) < 0.05 * theory.max()     # N=10 is coarse: 5 % of the peak
    d = golden("o2_pipe_N10")
    g = sim.get_fields()
    assert_fields_close(g, d, TOLN, "s999_")


# ---- state I/O and the frame dumper on the real engine ---------------------------------------------------
def test_checkpoint_restart_is_bitwise(lbhip, tmp_path):
    from LB_D2Q9.simulation import Simulation
    nx, ny = 600, 90
    rng = np.random.default_rng(4)
    mask = rng.random((nx, ny)) < 0.03
    mask[0, :] = mask[-1, :] = False
    mask[:, 0] = mask[:, -1] = False
    a = Simulation(nx, ny, 1.45, bc="pipe", inlet_rho=1.003, obstacle_mask=mask)
    a.set_variant(33)
    a.set_f(_random_state(rng, nx, ny))
    a.run(9)
    path = str(tmp_path / "state.npz")
    a.save_checkpoint(path)
    a.run(10)
    b = Simulation.from_checkpoint(path)
    g = b.get_fields(("rho", "u"))
    b.run(10)
    ga, gb = a.get_fields(("f", "rho", "u", "v")), b.get_fields(("f", "rho", "u", "v"))
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k
    assert g["rho"].shape == (nx, ny)


def test_checkpoint_round_trips_every_handle_parameter(lbhip, tmp_path):
    """A checkpoint rebuilds the handle it was taken from: Cython-path semantics, the velocity-inlet family with its
    imposed speeds, a bare path (np.savez appends .npz), a cleared obstacle; loading into a lattice with another
    omega / semantics is refused instead of silently changing the physics."""
    from LB_D2Q9.simulation import Simulation
    nx, ny = 70, 40
    rng = np.random.default_rng(8)
    f0 = _random_state(rng, nx, ny)
    for kw in (dict(bc="pipe", inlet_rho=1.004, semantics="cython"),
               dict(bc="velocity_inlet", inlet_u=0.03, outlet_u=0.025),
               dict(bc="cavity", lid_u=0.07, rho0=1.01)):
        a = Simulation(nx, ny, 1.2, **kw)
        a.set_f(f0)
        a.run(6)
        path = str(tmp_path / ("ckpt_" + kw["bc"] + kw.get("semantics", "")))       # no suffix
        a.save_checkpoint(path)
        b = Simulation.from_checkpoint(path)
        assert b.semantics == a.semantics and b.bc_mode == a.bc_mode
        assert (b.inlet_u, b.outlet_u, b.lid_u, b.rho0) == (a.inlet_u, a.outlet_u, a.lid_u, a.rho0)
        a.run(7); b.run(7)
        ga, gb = a.get_fields(("f", "rho", "u", "v")), b.get_fields(("f", "rho", "u", "v"))
        for k in ga:
            assert np.array_equal(ga[k], gb[k]), (kw, k)
    # mismatches are refused
    a = Simulation(nx, ny, 1.2, bc="pipe", inlet_rho=1.004)
    a.set_f(f0)
    a.save_checkpoint(str(tmp_path / "p"))
    for other in (dict(omega=1.3), dict(semantics="cython"), dict(inlet_rho=1.005)):
        kw = dict(omega=1.2, bc="pipe", inlet_rho=1.004)
        kw.update(other)
        with pytest.raises(ValueError):
            Simulation(nx, ny, **kw).load_checkpoint(str(tmp_path / "p"))
    # a checkpoint without an obstacle clears the obstacle of the lattice it is loaded into
    mask = rng.random((nx, ny)) < 0.1
    c = Simulation(nx, ny, 1.2, bc="pipe", inlet_rho=1.004, obstacle_mask=mask)
    c.load_checkpoint(str(tmp_path / "p.npz"))
    a.run(5); c.run(5)
    assert np.array_equal(a.get_fields(("f",))["f"], c.get_fields(("f",))["f"])


def test_frame_dumper_on_pipe_flow_cylinder(lbhip, tmp_path):
    from LB_D2Q9.dimensionless import opencl_dim as lb
    from LB_D2Q9.frames import Frame_Dumper
    sim = lb.Pipe_Flow_Cylinder(diameter=1., rho=1., viscosity=1., pressure_grad=-10., pipe_length=3., N=8,
                                cylinder_center=[.75, .5], cylinder_radius=.1, verbose=False)
    d = Frame_Dumper(sim, sim.u, num_steps_per_draw=10, max_magnitude=0.05, render_folder=str(tmp_path))
    d.run(2)
    assert d.total_num_steps == 20 and d.I.shape == (sim.nx, sim.ny) and np.isfinite(d.I).all()
    assert np.array_equal(d.I, sim.get_fields()["u"])
    assert all(open(f, "rb").read(8) == b"\x89PNG\r\n\x1a\n" for f in d.frames_written)


def test_pipe_flow_cylinder_docs_case_vs_oracle(lbhip, oracle):
    """The reference's movie notebook case (docs/vortex_sheet_movie.ipynb: D=1, rho=1, nu=1, gradP=-100,
    len=3, N=25, cylinder r=1/25 at (.75,.5), 62 steps per 0.1 t) through the drop-in class, against the
    oracle's restatement of the same constructor with the same numpy RNG stream for init_pop."""
    from LB_D2Q9.dimensionless import opencl_dim as lb
    kw = dict(diameter=1., rho=1., viscosity=1., pressure_grad=-100., pipe_length=3., N=25)
    cyl = dict(cylinder_center=[.75, .5], cylinder_radius=1. / 25)
    np.random.seed(1234)
    sim = lb.Pipe_Flow_Cylinder(verbose=False, **cyl, **kw)
    np.random.seed(1234)
    perturb = 1. + .001 * np.random.randn(sim.nx, sim.ny, 9)
    ref = oracle.O2Sim.pipe_flow(perturb=perturb, **cyl, **kw)
    assert (sim.nx, sim.ny) == (ref.nx, ref.ny) == (1876, 626)
    assert sim.omega == ref.params["omega"] and sim.inlet_rho == ref.params["inlet_rho"]
    assert np.array_equal(sim.obstacle_mask_host.astype(bool), ref.mask.T.astype(bool))
    g0 = sim.get_fields()
    assert np.array_equal(g0["f"], ref.get_fields()["f"])          # identical initial populations
    sim.run(62)
    ref.run(62)
    assert_fields_close(sim.get_fields(), ref.get_fields(), TOLN)
    phys = sim.get_physical_fields()
    assert np.allclose(phys["u"], sim.get_fields()["u"] * (sim.delta_x / sim.delta_t) * (sim.L / sim.T))


@pytest.mark.parametrize("bc", ["pipe", "periodic", "cavity"])
def test_small_grid_graph_replay_equals_eager_steps(lbhip, bc):
    """Grids <= 768^2 replay 16 captured single-step launches per hipGraph launch inside run(n);
    run(1) never does.  Same kernels, so the two must agree bit for bit (also across a mask change,
    which invalidates the capture)."""
    from LB_D2Q9.simulation import Simulation
    nx, ny = 200, 150
    rng = np.random.default_rng(8)
    f0 = _random_state(rng, nx, ny)
    mask = rng.random((nx, ny)) < 0.04
    mask[0, :] = mask[-1, :] = False
    mask[:, 0] = mask[:, -1] = False
    kw = dict(inlet_rho=1.004, lid_u=0.05)
    a, b = Simulation(nx, ny, 1.4, bc=bc, **kw), Simulation(nx, ny, 1.4, bc=bc, **kw)
    a.set_f(f0); b.set_f(f0)
    a.run(101)                                   # 6 graph replays + 5 eager steps
    for _ in range(101):
        b.run(1)
    for k in ("f", "rho", "u", "v"):
        assert np.array_equal(a.get_fields((k,))[k], b.get_fields((k,))[k]), k
    a.set_obstacle_mask(mask); b.set_obstacle_mask(mask)
    a.run(50)
    for _ in range(50):
        b.run(1)
    for k in ("f", "rho", "u", "v"):
        assert np.array_equal(a.get_fields((k,))[k], b.get_fields((k,))[k]), k


def test_velocity_inlet_family_vs_reference_kernels(lbhip, oracle):
    """The reference's velocity-inlet rule set (D2Q9.cl:263-374) on the GPU, phase by phase and over 200
    steps, against the fixture produced by executing those kernels, then through the class."""
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.dimensionless import opencl_dim as lb
    d = golden("o2_velocity_inlet_45x23")
    nx, ny, uw = int(d["nx"]), int(d["ny"]), float(d["u_w"])
    sim = Simulation(nx, ny, float(d["omega"]), bc="velocity_inlet", inlet_u=uw)
    sim.set_f(d["f0"])
    sim.move_bcs()
    assert maxdiff(sim.get_fields(("f",))["f"], d["after_bcs_f"]) <= 2.5e-7
    sim.set_f(d["f0"])
    sim.set_fields(np.ones((nx, ny)), np.full((nx, ny), uw), np.zeros((nx, ny)))
    sim.update_hydro()
    g = sim.get_fields(("rho", "u", "v"))
    assert maxdiff(g["rho"], d["hydro_rho"]) <= 5e-7 and maxdiff(g["u"], d["hydro_u"]) <= 1e-6
    assert maxdiff(g["v"], d["hydro_v"]) <= 1e-6
    sim.set_fields(np.ones((nx, ny)), np.full((nx, ny), uw), np.zeros((nx, ny)))
    done = 0
    for n in (1, 20, 200):
        sim.run(n - done)
        done = n
        assert_fields_close(sim.get_fields(), d, TOL1 if n == 1 else TOLN, "s%d_" % n)
    # the class: same overrides as OLD/opencl.py:281-327 on the dimensionless constructor
    np.random.seed(3)
    c = lb.Pipe_Flow_PeriodicBC_VelocityInlet(u_w=0.04, diameter=1., rho=1., viscosity=0.1, pressure_grad=-1.,
                                              pipe_length=2., N=24, verbose=False)
    np.random.seed(3)
    perturb = 1. + .001 * np.random.randn(c.nx, c.ny, 9)
    o = oracle.O2Sim(c.nx, c.ny, c.omega, oracle.BC_VELOCITY_INLET, u_w=0.04)
    o.set_macro(np.ones((c.nx, c.ny)), np.full((c.nx, c.ny), 0.04), np.zeros((c.nx, c.ny)))
    o.update_feq(); o.init_pop(perturb)
    assert maxdiff(c.get_fields()["f"], o.get_fields()["f"]) <= 2.5e-7
    c.run(100); o.run(100)
    assert_fields_close(c.get_fields(), o.get_fields(), TOLN)


@pytest.mark.parametrize("nx,ny,masked", [(45, 23, False), (67, 31, True), (1024, 160, False), (1003, 131, True), (2048, 70, True),
                                          (1536, 300, True)])
def test_velocity_inlet_fused_kernels_vs_oracle_and_unfused(lbhip, oracle, nx, ny, masked):
    """lb_run on the velocity-inlet family = fused kernels (k_step; k_step2 from nx >= 512, >= 64 rows; from 128 rows also
    k_step3 / k_step4 / k_step5 on the rows no wall-row link reaches + the wall-row bands advanced as a small lattice of their own): against the
    oracle's restatement of D2Q9.cl:263-374 driven as OLD/opencl.py:281-327 drives it (pinned bit-exact to the executed
    kernels by o2_velocity_inlet_45x23), against the engine's own un-fused phase sequence, and the two fused kernels
    against each other bit for bit.  Random initial populations, so that the four corner cells' never-written links
    (lb_get_corner_state) and the never-written u, v of the inlet / outlet columns matter."""
    from LB_D2Q9.simulation import Simulation
    rng = np.random.default_rng(nx + 3 * ny)
    f0 = _random_state(rng, nx, ny)
    mask = None
    if masked:
        mask = rng.random((nx, ny)) < 0.03
        mask[0, :] = mask[-1, :] = False
        mask[:, 0] = mask[:, -1] = False
    uw, ue, omega = 0.03, 0.028, 1.25
    u0 = (0.01 * rng.standard_normal((nx, ny))).astype(np.float32)
    v0 = (0.01 * rng.standard_normal((nx, ny))).astype(np.float32)
    o = oracle.O2Sim(nx, ny, omega, oracle.BC_VELOCITY_INLET, u_w=uw, u_e=ue, mask=mask)
    o.set_macro(np.ones((nx, ny)), u0, v0)
    o.set_f(f0)
    outs = {}
    variants = [("single", 0), ("two", 33), ("auto", -1)]
    if nx >= 512 and ny >= 128:
        variants += [("three", 97), ("four", 353), ("five", 353 | 4096)]
    for name, variant in variants:
        s = Simulation(nx, ny, omega, bc="velocity_inlet", inlet_u=uw, outlet_u=ue, obstacle_mask=mask)
        s.set_variant(variant)
        if variant == 33 and nx >= 512 and ny >= 64:
            assert s.steps_per_launch() == 2 and "k_step2" in s.hot_kernel()
        if variant in (97, 353, 353 | 4096):
            assert s.steps_per_launch() == {97: 3, 353: 4, 4449: 5}[variant]
        s.set_fields(np.ones((nx, ny)), u0, v0)
        s.set_f(f0)
        s.run(1)
        if name == "single":
            o.run(1)
            assert_fields_close(s.get_fields(), o.get_fields(), TOL1)
        s.run(7); s.run(4)
        outs[name] = s.get_fields(("f", "rho", "u", "v"))
        assert np.array_equal(s.get_corner_state(), np.array([f0[0, 0, 1], f0[0, 0, 8], f0[0, -1, 1], f0[0, -1, 5],
                                                              f0[-1, 0, 3], f0[-1, 0, 7], f0[-1, -1, 3], f0[-1, -1, 6]]))
    o.run(11)
    assert_fields_close(outs["single"], o.get_fields(), dict(f=2e-6, rho=2e-6, u=2e-6, v=2e-6))
    for name in outs:
        for k in ("f", "rho", "u", "v"):
            assert np.array_equal(outs["single"][k], outs[name][k]), (name, k)
    # the un-fused phase sequence (opencl_dim.py:380-387 order), also after fused steps have swapped the lattices
    u = Simulation(nx, ny, omega, bc="velocity_inlet", inlet_u=uw, outlet_u=ue, obstacle_mask=mask)
    u.set_fields(np.ones((nx, ny)), u0, v0)
    u.set_f(f0)
    u.run(3)
    for _ in range(9):
        u.move(); u.move_bcs(); u.update_hydro(); u.update_feq(); u.collide_particles()
    g = u.get_fields(("f", "rho", "u", "v"))
    for k in g:
        assert maxdiff(g[k], outs["single"][k]) <= 2e-6, k


def test_autotune_is_transparent(lbhip, oracle):
    """lb_autotune times the candidate fused-kernel configurations on live steps; whatever it picks, the
    trajectory is the one the single-step kernel produces (bitwise), also when a long run triggers it."""
    from LB_D2Q9.simulation import Simulation
    nx, ny = 1500, 300
    rng = np.random.default_rng(12)
    f0 = _random_state(rng, nx, ny)
    mask = rng.random((nx, ny)) < 0.02
    mask[0, :] = mask[-1, :] = False
    mask[:, 0] = mask[:, -1] = False
    ref = Simulation(nx, ny, 1.3, bc="pipe", inlet_rho=1.002, obstacle_mask=mask)
    ref.set_variant(0)
    ref.set_f(f0)
    a = Simulation(nx, ny, 1.3, bc="pipe", inlet_rho=1.002, obstacle_mask=mask)
    a.set_f(f0)
    used = a.autotune()
    assert used > 0 and a.steps_per_launch() in (1, 2, 3, 4)
    name = a.hot_kernel()                         # names the kernel; a marching kernel also its tuned waves per CU
    assert name.startswith("k_") and "<PIPE, MASK>" in name
    assert ("tuned:" in name) == (a.steps_per_launch() > 1 and "k_tile4" not in name)
    ref.run(used)
    for k in ("f", "rho", "u", "v"):
        assert np.array_equal(a.get_fields((k,))[k], ref.get_fields((k,))[k]), k
    a.run(31); ref.run(31)
    assert np.array_equal(a.get_fields(("f",))["f"], ref.get_fields(("f",))["f"])
    b = Simulation(nx, ny, 1.3, bc="pipe", inlet_rho=1.002, obstacle_mask=mask)
    b.set_f(f0)
    b.run(2900)                                   # long blocking first run (a small grid: >= 4 x 721 + 7 steps): tunes itself on the way
    assert b.steps_per_launch() in (1, 2, 3, 4)
    ref2 = Simulation(nx, ny, 1.3, bc="pipe", inlet_rho=1.002, obstacle_mask=mask)
    ref2.set_variant(0); ref2.set_f(f0); ref2.run(2900)
    for k in ("f", "rho", "u", "v"):
        assert np.array_equal(b.get_fields((k,))[k], ref2.get_fields((k,))[k]), k


@pytest.mark.parametrize("family", ["pipe_mask", "velocity_inlet", "cython", "d2q9i", "periodic_halo"])
def test_planar_layout_equals_interleaved_rows(lbhip, family):
    """The two device layouts of the lattices (rows interleaving the nine planes = default; LB_FLAG_PLANAR = each plane
    contiguous) are a matter of strides only: same kernels, same bits -- every family, the phase kernels, set / get of f
    and feq, the halo interface."""
    from LB_D2Q9.simulation import Simulation
    rng = np.random.default_rng(77)
    nx, ny, steps = 600, 140, 11
    kw = dict(bc="pipe", inlet_rho=1.004)
    if family == "pipe_mask":
        mask = rng.random((nx, ny)) < 0.04
        mask[0, :] = mask[-1, :] = False
        mask[:, 0] = mask[:, -1] = False
        kw["obstacle_mask"] = mask
    elif family == "velocity_inlet":
        kw = dict(bc="velocity_inlet", inlet_u=0.03)
    elif family in ("cython", "d2q9i"):
        kw["semantics"] = family
    elif family == "periodic_halo":
        kw = dict(bc="periodic", halo=True)
    f0 = _random_state(rng, nx, ny)
    out = []
    for planar in (False, True):
        s = Simulation(nx, ny, 1.3, planar=planar, **kw)
        assert s.layout()["planar"] == planar
        assert s.layout()["plane_stride"] == (s.layout()["pitch"] * (ny + 28) if planar else s.layout()["pitch"])      # (14 ghost rows per side)
        s.set_f(f0)
        assert np.array_equal(s.get_fields(("f",))["f"], f0)
        if family == "periodic_halo":
            buf = np.zeros((2, s.halo_floats()), np.float32)

            def exchange():                         # the ring of one slab with itself through the halo interface
                s.halo_export(0, buf[0]); s.halo_export(1, buf[1])
                s.sync()
                s.halo_import(0, buf[1]); s.halo_import(1, buf[0])
                s.sync()
            exchange()
            for it in range(steps):
                s.step_boundary(it == steps - 1); s.step_interior(it == steps - 1)
                exchange()
                s.step_finish()
            s.update_feq()
        else:
            s.run(steps)
            s.move_bcs(); s.update_hydro(); s.update_feq()          # the un-fused phases, too
        out.append(s.get_fields(("f", "feq", "rho", "u", "v")))
        s.close()
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k]), (family, k)
