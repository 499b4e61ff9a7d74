"""BASELINE.json's configurations at their full sizes on the GPU.  Where the oracle finishes in
seconds (<= 4096^2, a few steps) the comparison is direct; beyond that the checks are size-independent
properties of the lattice-Boltzmann step (exact x-periodicity of a periodic initial state, mass drift,
partition independence)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_gpu_parity import TOL1, assert_fields_close, contract_tol, maxdiff

pytestmark = pytest.mark.gpu


def equilibrium(rho, u, v):
    """f = feq(rho, u, v) in float64 -> float32, (nx, ny, 9) F-ordered."""
    w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    cx = np.array([0, 1, 0, -1, 0, 1, -1, -1, 1])
    cy = np.array([0, 0, 1, 0, -1, 1, 1, -1, -1])
    f = np.empty(rho.shape + (9,), np.float32, order="F")
    usq = u.astype(np.float64) ** 2 + v.astype(np.float64) ** 2
    for k in range(9):
        cu = cx[k] * u.astype(np.float64) + cy[k] * v.astype(np.float64)
        f[..., k] = w[k] * rho * (1 + 3 * cu + 4.5 * cu * cu - 1.5 * usq)
    return f


def test_config2_lid_driven_cavity_1024_vs_oracle(lbhip, oracle):
    """1024x1024 lid-driven cavity, Re = U L / nu = 1000 with U = 0.1: omega = 1/(3 nu + 1/2)."""
    from LB_D2Q9.simulation import Simulation
    n, U = 1024, 0.1
    nu = U * (n - 1) / 1000.
    omega = 1. / (3 * nu + 0.5)
    assert omega == pytest.approx(1.2393, abs=1e-3)
    rng = np.random.default_rng(2)
    f0 = equilibrium(np.ones((n, n)), 1e-3 * rng.standard_normal((n, n)), 1e-3 * rng.standard_normal((n, n)))
    sim = Simulation(n, n, omega, bc="cavity", lid_u=U, rho0=1.)
    ref = oracle.O2Sim(n, n, omega, oracle.BC_CAVITY, lid_u=U, rho0=1.)
    sim.set_f(f0); ref.set_f(f0)
    sim.run(1); ref.run(1)
    assert_fields_close(sim.get_fields(("f", "rho", "u", "v")), ref.get_fields(), dict(f=2.5e-7, rho=5e-7, u=1e-6, v=1e-6))
    sim.run(19); ref.run(19)
    g, w = sim.get_fields(("f", "rho", "u", "v")), ref.get_fields()
    assert_fields_close(g, w, dict(f=2e-6, rho=2e-6, u=2e-6, v=2e-6))
    assert g["u"][n // 2, -1] > 0.05          # the lid drags the top row along


@pytest.mark.parametrize("variant,steps", [(9, 2), (33, 2), (97, 3), (353, 4), (353, 8), (4449, 5), (4449, 10), (20833, 6), (20833, 12),
                                           (-1, 4), (-1, 7), (-1, 14)])
def test_config3_kelvin_helmholtz_4096_vs_oracle(lbhip, oracle, variant, steps):
    """4096x4096 periodic double shear layer against the oracle: single-, two-, three-, four- and five-step kernels
    (353 = k_step4, 4449 = k_step5, 20833 = k_deep<6> forced, -1 = the automatic choice, which is k_deep<7> at this size: the kernel
    bench.py times; 4 steps of it = its remainder launch, k_step4), one and two launches of the four- ... seven-step kernel."""
    from LB_D2Q9.simulation import Simulation
    import bench
    n = 4096
    rho, u, v = bench.shear_layer(n, n, 0, n, U=0.05)
    f0 = equilibrium(rho, u, v)
    sim = Simulation(n, n, 1.8, bc="periodic")
    sim.set_variant(variant)
    if variant in (353, 4449, 20833, -1):
        assert sim.steps_per_launch() == {353: 4, 4449: 5, 20833: 6, -1: 7}[variant]
    sim.set_f(f0)
    ref = oracle.O2Sim(n, n, 1.8, oracle.BC_PERIODIC)
    ref.set_f(f0)
    sim.run(steps); ref.run(steps)
    # the contract's bound for this many steps (contract_tol: n x the single-step bounds of SURVEY 8c, not fitted to a kernel);
    # measured at 4096^2, round 5's arithmetic (= round 4's): 4 steps f 3.3e-7 rho 1.01e-6; 8 steps f 6.9e-7 rho 1.25e-6
    tol = contract_tol(steps)
    assert_fields_close(sim.get_fields(("f", "rho", "u", "v")), ref.get_fields(), tol)


def test_config4_shear_layer_8192_properties(lbhip):
    """8192x8192 periodic shear layer (the bench workload).  The initial state has period nx/4 in x,
    every cell is updated by the same arithmetic from identical neighbourhoods, so the state must stay
    EXACTLY nx/4-periodic; total mass may only drift by fp32 rounding bias (the float32 weights sum to
    1 + 7.5e-9, so the reference arithmetic itself gains ~1e-8 x omega per step: bound 5e-8 per step)."""
    from LB_D2Q9.simulation import Simulation
    import bench
    n, steps = 8192, 1005                                # 1005 = 4 + 143 x 7: four-step and seven-step kernels both run
    sim = Simulation(n, n, 1.7, bc="periodic")
    sim.init_equilibrium(*bench.shear_layer(n, n, 0, n))
    assert sim.steps_per_launch() == 7                   # periodic whole-grid handle of >= 2560^2 cells: k_deep<7>
    rho0 = sim.get_fields(("rho",))["rho"].astype(np.float64).sum()
    sim.run(steps)
    g = sim.get_fields(("rho", "u", "v"))
    q = n // 4
    for k in ("rho", "u", "v"):
        a = g[k]
        assert np.array_equal(a[:q], a[q:2 * q]) and np.array_equal(a[:q], a[3 * q:]), k
        assert np.all(np.isfinite(a))
    drift = (g["rho"].astype(np.float64).sum() - rho0) / rho0 / steps
    assert abs(drift) < 5e-8, drift
    assert abs(g["u"]).max() < 0.06 and abs(g["v"]).max() < 0.01


def test_config4_shear_layer_8192_default_kernel_vs_oracle(lbhip, oracle):
    """8192x8192, the bench workload, the bench's initial state, the kernel the bench times (k_deep<7> on segment pairs: one
    launch = 7 steps, two launches = 14) DIRECTLY against the oracle at the size the metric is quoted on -- the same-size field
    comparison the reference's own check makes (testing/Bryan/opencl_check_03.ipynb:593, 778).  The oracle runs its
    -fopenmp build (same bits as the serial one: tests/test_oracle_golden.py).  Bounds: the contract's, n x the single-step ones
    (contract_tol); the measured margins are printed."""
    from LB_D2Q9.simulation import Simulation
    import bench
    n = 8192
    sim = Simulation(n, n, 1.7, bc="periodic")
    assert sim.steps_per_launch() == 7 and "k_deep<7>" in sim.hot_kernel()
    sim.init_equilibrium(*bench.shear_layer(n, n, 0, n))          # f = feq, built on the device, as bench.py does
    f0 = sim.get_fields(("f",))["f"]
    ref = oracle.O2Sim(n, n, 1.7, oracle.BC_PERIODIC)
    ref.set_f(f0)
    del f0
    done = 0
    for steps in (7, 14):
        tol = contract_tol(steps)
        sim.run(steps - done)
        ref.run(steps - done, openmp=True)
        done = steps
        g = sim.get_fields(("f", "rho", "u", "v"))
        # (nx, ny, 9) F-ordered on the GPU side is the oracle's (9, ny, nx) C-ordered array: compare without copies
        gf = np.asarray(g["f"]).transpose(2, 1, 0)
        assert gf.flags.c_contiguous
        meas = {"f": max(maxdiff(gf[k9], ref.f[k9]) for k9 in range(9))}      # (plane by plane: no 5 GB float64 temporaries)
        for k in ("rho", "u", "v"):
            meas[k] = maxdiff(np.asarray(g[k]).T, getattr(ref, k))
        report = ", ".join("%s %.2e / %.1e" % (k, meas[k], tol[k]) for k in ("f", "rho", "u", "v"))
        print("8192^2, %d steps, measured / bound: %s" % (steps, report))
        for k in meas:
            assert meas[k] <= tol[k], (steps, k, report)
        del g
    assert float(np.abs(ref.u).max()) > 0.03                       # the shear layer is there


def test_config4_shear_layer_8192_default_and_four_step_kernel_equal_single_step_kernel_bitwise(lbhip):
    """8192x8192, the bench workload: the default kernel (k_deep<7>: 4 + 4 steps, then the driver's 5 + 10 x 20 steps = 6 + 7 + 7
    each), the six-step kernel (20833: k_deep<6>), the five-step kernel (variant 4449: the velocity-inlet family's) and the four-step kernel (353) against
    the single-step kernel (variant 9) on the populations themselves, bit for bit.  The single-step kernel is the one the
    oracle comparisons at <= 4096^2 pin; this carries them to the size the metric is quoted on."""
    from LB_D2Q9.simulation import Simulation
    import bench
    n = 8192
    ref = None
    for variant in (9, -1, 20833, 4449, 353, 119137):
        sim = Simulation(n, n, 1.7, bc="periodic")
        sim.set_variant(variant)
        assert sim.steps_per_launch() == {9: 1, -1: 7, 20833: 6, 4449: 5, 353: 4, 119137: 7}[variant]
        sim.init_equilibrium(*bench.shear_layer(n, n, 0, n))
        f0 = sim.get_fields(("f",))["f"] if ref is None else None
        sim.run(8)
        f8 = sim.get_fields(("f",))["f"]
        # ... and over the driver's whole bench run: 5 warm-up steps, then blocks of 20 (6 + 7 + 7 steps of the default kernel,
        # segment pairs and all), 213 steps in all: still the single-step kernel's bits
        sim.run(5)
        for _ in range(10):
            sim.run(20)
        f213 = sim.get_fields(("f",))["f"]
        chk = sim.check()
        sim.close()
        if ref is None:
            assert np.all(np.isfinite(f8)) and f8.std() > 0 and not np.array_equal(f0, f8)     # the flow has evolved
            del f0
            ref = (f8, f213, chk)
        else:
            assert np.array_equal(f8, ref[0])
            assert np.array_equal(f213, ref[1]) and chk == ref[2]
            assert chk["n_nonfinite"] == 0 and abs(chk["sum_rho"] / (n * n) - 1.0) < 1e-4


def test_planar_layout_8192_deep_kernels_equal_single_step_kernel_bitwise(lbhip):
    """LB_FLAG_PLANAR at 8192^2: each plane contiguous, 8220 rows x 8192 floats = 269 MB, so k_deep's scalar plane offsets reach
    8 x 269 MB = 2.16 GB -- beyond the 2 GiB its buffer resources spanned until round 6 (a raw buffer access is range-checked as
    offset >= num_records - soffset: plane 8 would have read zeros and dropped its stores, silently).  k_deep<7> and
    k_deep<6> against the single-step kernel, bit for bit, on the bench's initial state."""
    from LB_D2Q9.simulation import Simulation
    import bench
    n = 8192
    ref = None
    for variant in (9, -1, 20833):
        sim = Simulation(n, n, 1.7, bc="periodic", planar=True)
        lay = sim.layout()
        assert lay["planar"] and 8 * lay["plane_stride"] * 4 > 2 ** 31
        sim.set_variant(variant)
        assert sim.steps_per_launch() == {9: 1, -1: 7, 20833: 6}[variant]
        sim.init_equilibrium(*bench.shear_layer(n, n, 0, n))
        sim.run(14)
        f = sim.get_fields(("f",))["f"]
        chk = sim.check()
        sim.close()
        if ref is None:
            assert np.all(np.isfinite(f[:, :, 8])) and f[:, :, 8].std() > 0
            ref = (f, chk)
        else:
            for k in range(9):
                assert np.array_equal(f[:, :, k], ref[0][:, :, k]), (variant, "plane", k)
            assert chk == ref[1]


@pytest.mark.parametrize("bc,masked", [("pipe", True), ("cavity", False), ("periodic", True)])
def test_full_size_families_default_and_four_step_kernel_equal_single_step_kernel_bitwise(lbhip, bc, masked):
    """8192 x 8192 in the other boundary families, with and without an obstacle mask (the instantiations of k_step5 and k_step4
    the periodic bench never runs: wall rules on the boundary cell, lanes beyond the box, mask history registers): default
    kernel (k_deep<7> at this size: 4 + 4, 3 steps), six-, five- and four-step kernel against the single-step kernel, bit for bit."""
    from LB_D2Q9.simulation import Simulation
    import bench
    n = 8192
    mask = None
    if masked:
        mask = np.random.default_rng(5).random((n, n)) < 0.01
        if bc != "periodic":
            mask[0, :] = mask[-1, :] = False
            mask[:, 0] = mask[:, -1] = False
    out = []
    for variant in (-1, 9, 353, 4449, 20833):
        sim = Simulation(n, n, 1.6, bc=bc, inlet_rho=1.0005, lid_u=0.05, obstacle_mask=mask)
        sim.set_variant(variant)
        assert sim.steps_per_launch() == {9: 1, -1: 7, 353: 4, 4449: 5, 20833: 6}[variant]
        sim.init_equilibrium(*bench.shear_layer(n, n, 0, n))
        sim.run(14)
        sim.run(3)
        out.append(sim.get_fields(("f",))["f"])
        sim.close()
    assert np.all(np.isfinite(out[0])) and all(np.array_equal(o, out[1]) for o in out)


@pytest.mark.parametrize("bc,nx,ny,masked", [("pipe", 3751, 1251, True), ("cavity", 2048, 2048, False),
                                             ("velocity_inlet", 4096, 1000, False), ("pipe", 5000, 700, False),
                                             ("velocity_inlet", 2304, 3000, True), ("pipe", 1024, 6000, False)])
def test_wall_column_strips_with_shorter_segments_bitwise(lbhip, bc, nx, ny, masked):
    """In the wall families k_step4 and k_step5 give the first and the last strip (the wall columns) shorter segments than the
    others (lb_hip.cpp launch_step2: `edge_seg_rows`); grids of many shapes -- few / many strips, short / tall, odd widths, with a
    mask, the velocity-inlet family's interior pass -- against the single-step kernel, bit for bit, 11 steps."""
    from LB_D2Q9.simulation import Simulation
    rng = np.random.default_rng(nx + ny)
    mask = None
    if masked:
        mask = rng.random((nx, ny)) < 0.01
        mask[0, :] = mask[-1, :] = False
        mask[:, 0] = mask[:, -1] = False
    rho = (1.0 + 1e-3 * rng.standard_normal((nx, ny))).astype(np.float32)
    u = (0.02 + 1e-3 * rng.standard_normal((nx, ny))).astype(np.float32)
    v = (1e-3 * rng.standard_normal((nx, ny))).astype(np.float32)
    out = []
    for variant in (353, 9, 353 | 4096, 353 | 4096 | 16384, 353 | 4096 | 16384 | 32768, 353 | 4096 | 16384 | 32768 | 65536):
        sim = Simulation(nx, ny, 1.5, bc=bc, inlet_rho=1.0005, lid_u=0.05, inlet_u=0.02, obstacle_mask=mask)
        sim.set_variant(variant)
        if variant & 4096:
            assert sim.steps_per_launch() == ((7 if variant & 32768 else 6) if (variant & 16384) and bc != "velocity_inlet" else 5)
        sim.init_equilibrium(rho, u, v)
        sim.run(14)
        sim.run(3)
        out.append(sim.get_fields(("f", "rho", "u", "v")))
        sim.close()
    for k in out[0]:
        assert np.all(np.isfinite(out[0][k])) and all(np.array_equal(o[k], out[1][k]) for o in out), k


def test_config4_eight_slabs_equal_one_gpu_run_bitwise(lbhip):
    """The 8-GPU decomposition of the bench workload (8 slabs of 1024 rows, k_deep<7>, 14-deep halo)
    executed as in-library virtual slabs on one device: bitwise equal to the undivided run."""
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import LocalSlabRing, partition_rows
    import bench
    n, steps = 8192, 38                           # 2 fourteen-step halo cycles + a lone seven-step half + 3 steps
    one = Simulation(n, n, 1.7, bc="periodic")
    one.init_equilibrium(*bench.shear_layer(n, n, 0, n))
    one.run(steps)
    want = one.get_fields(("rho", "u", "v"))
    one.close()
    ring = LocalSlabRing(n, n, 1.7, 8, bc="periodic")
    assert ring.parts == partition_rows(n, 8) and ring.slabs[0].steps_per_launch() == 7
    for s, (y0, h) in zip(ring.slabs, ring.parts):
        s.init_equilibrium(*bench.shear_layer(n, n, y0, h))
    ring.run_in_library(steps)
    got = ring.get_fields(("rho", "u", "v"))
    for k in want:
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("transport,cycle,inline", [("rccl", 0, False), ("rccl", 7, True), ("peer", 0, False), ("peer", 8, False)])
def test_one_slab_of_eight_as_a_ring_of_its_own_equals_the_plain_grid_bitwise(lbhip, transport, cycle, inline):
    """The slab one of eight ranks holds (8192 x 1024, periodic in itself: a one-rank ring) through lb_run's halo cycle at the sizes
    where the automatic choices apply -- fourteen-step cycle, thick split edge bands, under RCCL k_deep2<7> by default (under the peer
    transport k_deep<7>; lb_set_slab_cycle(8) / (7) say so explicitly), the exchange beside or between the interior launches --
    against the plain 8192 x 1024 grid, bit for bit."""
    from LB_D2Q9.simulation import Simulation, comm_unique_id
    import bench
    nx, ny, steps = 8192, 1024, 14 * 3 + 7 + 3
    one = Simulation(nx, ny, 1.7, bc="periodic")
    one.init_equilibrium(*bench.shear_layer(nx, ny, 0, ny))
    one.run(steps)
    want = one.get_fields(("f",))["f"]
    one.close()
    s = Simulation(nx, ny, 1.7, bc="periodic", halo=True)
    if transport == "rccl":
        s.comm_init(comm_unique_id(), 0, 1)
    else:
        d = s.peer_export()
        s.peer_connect(0, 1, d, d, ny)
    s.set_slab_cycle(cycle)
    s.set_exchange_inline(inline)
    s.exchange_timing(True)
    s.init_equilibrium(*bench.shear_layer(nx, ny, 0, ny))
    s.run(steps)
    st = s.exchange_stats()
    assert st["cycle_depth"] == 7 and st["n"] >= 3
    got = s.get_fields(("f",))["f"]
    s.close()
    assert np.array_equal(got, want)


def test_config5_porous_obstacles_4096_vs_oracle(lbhip, oracle):
    """4096x4096 pipe flow through the reference's obstacle image (docs/CS205_obstacle_4.tif rescaled
    by nearest neighbour), bounce-back mask, two steps against the oracle + a longer sanity run."""
    from LB_D2Q9.masks import obstacle_mask_from_tiff
    from LB_D2Q9.simulation import Simulation
    n = 4096
    mask = obstacle_mask_from_tiff(os.path.join(GOLDEN, "CS205_obstacle_4.tif"), (n, n))
    mask[0, :] = mask[-1, :] = 0
    mask[:, 0] = mask[:, -1] = 0
    assert 0.005 < mask.mean() < 0.02
    rin = 1.001
    ramp = oracle.density_ramp(n, n, rin, 1.)
    f0 = equilibrium(ramp.astype(np.float64), np.zeros((n, n)), np.zeros((n, n)))
    sim = Simulation(n, n, 1.0, bc="pipe", inlet_rho=rin, outlet_rho=1., obstacle_mask=mask)
    ref = oracle.O2Sim(n, n, 1.0, oracle.BC_PIPE, rin, 1., mask=mask)
    sim.set_f(f0); ref.set_f(f0)
    assert sim.steps_per_launch() == 7 and "k_deep<7>" in sim.hot_kernel()     # (walled + mask: k_deep from 4000^2 cells)
    sim.run(3); ref.run(3)                            # remainder launch: k_step3
    assert_fields_close(sim.get_fields(("f", "rho", "u", "v")), ref.get_fields(), dict(f=5e-7, rho=1e-6, u=1e-6, v=1e-6))
    sim.run(4); ref.run(4)                            # remainder launch: k_step4<PIPE, MASK>
    assert_fields_close(sim.get_fields(("f", "rho", "u", "v")), ref.get_fields(), dict(f=1e-6, rho=1e-6, u=1e-6, v=1e-6))
    sim.run(10); ref.run(10)                          # the plan's launches of k_deep<PIPE, MASK> (and a remainder launch)
    assert_fields_close(sim.get_fields(("f", "rho", "u", "v")), ref.get_fields(), dict(f=2.5e-6, rho=2.5e-6, u=2.5e-6, v=2.5e-6))
    sim.run(183)
    g = sim.get_fields(("rho", "u", "v"))
    assert np.all(np.isfinite(g["rho"])) and abs(g["rho"].mean() - 1.0005) < 1e-3
    assert g["u"].mean() > 0                         # flow from inlet to outlet
