"""Periodic multi-population sets (LB_D2Q9.populations): the streaming of the reference's research forks
(porous_media/single_component.cl:338-375 `move_periodic`) and the batched fused step."""
import numpy as np
import pytest

from conftest import golden
from test_gpu_parity import TOLN, assert_fields_close, _random_state

pytestmark = pytest.mark.gpu


def test_population_set_streaming_vs_executed_move_periodic(lbhip):
    """Periodic_Populations.move() against the reference kernel itself: fixture o2_move_periodic = input and output of
    porous_media/single_component.cl:338-375 executed work-item by work-item (oracle/make_golden.py gen_move_periodic),
    on arrays of distinct integers (exact in fp32)."""
    from LB_D2Q9.populations import Periodic_Populations
    d = golden("o2_move_periodic")
    for tag in "abc":
        f, want = d["f_" + tag].astype(np.float32), d["streamed_" + tag].astype(np.float32)
        nx, ny, P, _ = f.shape
        pops = Periodic_Populations(nx, ny, [1.0] * P)
        pops.set_f(f)
        pops.move()
        got = pops.get_fields(("f",))["f"]
        assert got.shape == want.shape and np.array_equal(got, want), tag
        pops.close()


def move_periodic_reference(f):
    """single_component.cl:338-375 on a host array f[x, y, field, jump] (Fortran order = the kernel's
    jump*P*nx*ny + field*nx*ny + y*nx + x): f_streamed[(x+cx) % nx, (y+cy) % ny, field, jump] = f[x, y, field, jump].
    Restated with np.roll for the sizes the fixture does not hold; the restatement itself is held to the executed kernel's
    fixture in tests/test_oracle_golden.py::test_periodic_streaming_is_the_forks_move_periodic."""
    cx = [0, 1, 0, -1, 0, 1, -1, -1, 1]
    cy = [0, 0, 1, 0, -1, 1, 1, -1, -1]
    out = np.empty_like(f)
    for j in range(9):
        out[:, :, :, j] = np.roll(np.roll(f[:, :, :, j], cx[j], axis=0), cy[j], axis=1)
    return out


@pytest.mark.parametrize("nx,ny,masked", [(64, 48, False), (250, 250, True), (1030, 70, False)])
def test_population_set_streaming_and_batched_step(lbhip, oracle, nx, ny, masked):
    from LB_D2Q9.populations import Periodic_Populations
    from LB_D2Q9.simulation import Simulation
    omegas = [1.7, 0.9, 1.25]
    rng = np.random.default_rng(nx)
    f0 = np.asfortranarray(np.stack([_random_state(rng, nx, ny) for _ in omegas], axis=2))     # (nx, ny, 3, 9)
    mask = (rng.random((nx, ny)) < 0.04) if masked else None
    pops = Periodic_Populations(nx, ny, omegas, obstacle_mask=mask)
    assert (pops.num_populations, pops.num_jumpers) == (3, 9)
    # move_periodic: pure data movement, exact
    pops.set_f(f0)
    pops.move()
    got = pops.get_fields(("f",))["f"]
    assert got.shape == (nx, ny, 3, 9) and got.flags.f_contiguous
    assert np.array_equal(got, move_periodic_reference(f0))
    # one fused launch per step for the whole set == every population run alone, bit for bit; and the oracle
    pops.set_f(f0)
    pops.run(7)
    pops.run(4)
    g = pops.get_fields()
    for i, om in enumerate(omegas):
        one = Simulation(nx, ny, om, bc="periodic", obstacle_mask=mask)
        one.set_variant(0)
        one.set_f(f0[:, :, i, :])
        one.run(11)
        h = one.get_fields(("f", "rho", "u", "v"))
        for k in h:
            assert np.array_equal(g[k][:, :, i], h[k]), (i, k)
        if nx * ny <= 70000:
            o = oracle.O2Sim(nx, ny, om, oracle.BC_PERIODIC, mask=mask)
            o.set_f(f0[:, :, i, :])
            o.run(11)
            assert_fields_close(h, o.get_fields(), dict(f=2e-6, rho=2e-6, u=2e-6, v=2e-6))
    pops.close()


def test_population_set_argument_checks(lbhip):
    import ctypes as ct
    from LB_D2Q9 import _native
    from LB_D2Q9.populations import Periodic_Populations
    from LB_D2Q9.simulation import Simulation
    with pytest.raises(ValueError):
        Periodic_Populations(32, 32, [])
    a, b = Simulation(64, 64, 1.0, bc="periodic"), Simulation(64, 32, 1.0, bc="periodic")
    c = Simulation(64, 64, 1.0, bc="pipe")
    for pair in ((a, b), (a, c), (a, a)):
        arr = (ct.c_void_p * 2)(pair[0]._h, pair[1]._h)
        assert _native.lib().lb_run_batch(arr, 2, 1) == -1
