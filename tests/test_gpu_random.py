"""Randomised GPU parity sweep: random grid shapes (including tiny, odd, non-multiple-of-4 and
strip-boundary widths), boundary families, masks, kernel variants and step counts.  Every variant must
equal the single-step kernel bit for bit and match the oracle within the fp32 tolerance."""
import os

import numpy as np
import pytest

from test_gpu_parity import assert_fields_close, _random_state

pytestmark = pytest.mark.gpu

# NT / tile shapes / XCD order / two-, three-, four-step marching kernels / LDS-tile kernel / the automatic choice
VARIANTS = (1, 9, 16, 24, 33, 41, 97, 105, 97 | 256, 105 | 256, 97 | 256 | 4096, 105 | 256 | 4096, 97 | 256 | 4096 | 16384, 105 | 256 | 4096 | 16384, 97 | 256 | 4096 | 16384 | 32768, 105 | 256 | 4096 | 16384 | 32768,
            97 | 256 | 4096 | 16384 | 32768 | 65536, 105 | 256 | 4096 | 16384 | 32768 | 65536, 512, 512 | 1, -1)
WIDTHS = (2, 3, 5, 63, 64, 65, 255, 256, 257, 511, 512, 513, 600, 768, 1021, 1024, 1028, 1280)


@pytest.mark.parametrize("seed", range(int(os.environ.get("LB_RANDOM_SEEDS", "36"))))   # more for a soak
def test_random_configuration(lbhip, oracle, seed):
    from LB_D2Q9.simulation import Simulation
    rng = np.random.default_rng(1000 + seed)
    bc = ("pipe", "periodic", "cavity")[seed % 3]
    nx = int(rng.choice(WIDTHS))
    ny = int(rng.choice((2, 3, 7, 33, 64, 129, 130, 200, 257, 300)))
    if bc == "periodic" and nx % 4 and nx >= 512:
        nx += 4 - nx % 4                                   # the marching kernels need nx % 4 == 0 when periodic
    steps = int(rng.integers(1, 14))
    omega = float(rng.uniform(0.6, 1.8))
    masked = bool(rng.integers(0, 2)) and nx > 4 and ny > 4
    mask = None
    if masked:
        mask = rng.random((nx, ny)) < 0.05
        mask[0, :] = mask[-1, :] = False
        mask[:, 0] = mask[:, -1] = False
    kw = dict(inlet_rho=1.0 + float(rng.uniform(0, 0.01)), lid_u=float(rng.uniform(0, 0.08)))
    f0 = _random_state(rng, nx, ny)
    base = Simulation(nx, ny, omega, bc=bc, obstacle_mask=mask, **kw)
    base.set_variant(0)
    base.set_f(f0)
    base.run(steps)
    want = base.get_fields(("f", "rho", "u", "v"))
    for i, variant in enumerate(rng.choice(VARIANTS, size=4, replace=False)):
        # (every other one with each plane contiguous in device memory instead of interleaved rows: LB_FLAG_PLANAR)
        s = Simulation(nx, ny, omega, bc=bc, obstacle_mask=mask, planar=bool(i & 1), **kw)
        s.set_variant(int(variant))
        s.set_f(f0)
        s.run(steps)
        got = s.get_fields(("f", "rho", "u", "v"))
        for k in want:
            assert np.array_equal(got[k], want[k]), (bc, nx, ny, steps, int(variant), bool(i & 1), k)
        s.close()
    code = {"pipe": oracle.BC_PIPE, "periodic": oracle.BC_PERIODIC, "cavity": oracle.BC_CAVITY}[bc]
    o = oracle.O2Sim(nx, ny, omega, code, kw["inlet_rho"], 1., kw["lid_u"], 1., mask=mask)
    o.set_f(f0)
    o.run(steps)
    assert_fields_close(want, o.get_fields(), dict(f=2e-6, rho=2e-6, u=2e-6, v=2e-6))


@pytest.mark.parametrize("sync_bits", [0, 2])
@pytest.mark.parametrize("seed", range(int(os.environ.get("LB_RANDOM_SLAB_SEEDS", "12"))))
def test_random_slab_partition(lbhip, seed, sync_bits):
    """Random row-slab partitions run through the in-library multi-GPU schedule (lb_run_group: halo cycles of two
    five-, four- or three-step launches, launch-by-launch remainders, walls with bands of unequal height, masks) must
    equal the undivided run bit for bit -- with the members' streams ordered by events alone (sync_bits 0: what lb_run
    relies on with RCCL) and with a device join after every exchange (2)."""
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import LocalSlabRing
    prev = lbhip.lb_set_debug_sync(sync_bits)
    try:
        _random_slab_partition_case(seed)
    finally:
        lbhip.lb_set_debug_sync(prev)


def _random_slab_partition_case(seed):
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import LocalSlabRing
    rng = np.random.default_rng(5000 + seed)
    bc = ("periodic", "pipe", "cavity")[seed % 3]
    nx = int(rng.choice((512, 516, 768, 1000, 1024, 1284)))
    nslabs = int(rng.integers(2, 6))
    ny = int(rng.integers(nslabs * 7, 700))                  # slab heights from 7 rows (no fused kernel) to 350
    variant = int(rng.choice((-1, 97 | 256 | 4096, 97 | 256, 97, 97 | 128, 33, 1)))
    if seed % 4 == 3:                                # (the twelve- / fourteen-step cycle: k_deep<6>, k_deep<7>, k_deep2<7> -- from 96 / 112 rows)
        variant = (97 | 256 | 4096 | 16384, 97 | 256 | 4096 | 16384 | 32768, 97 | 256 | 4096 | 16384 | 32768 | 65536)[(seed // 4) % 3]
    mask = None
    if rng.integers(0, 2):
        mask = rng.random((nx, ny)) < 0.03
        mask[0, :] = mask[-1, :] = False
        if bc != "periodic":
            mask[:, 0] = mask[:, -1] = False
    kw = dict(inlet_rho=1.005, lid_u=0.05)
    f0 = _random_state(rng, nx, ny)
    one = Simulation(nx, ny, 1.5, bc=bc, obstacle_mask=mask, **kw)
    one.set_variant(0)
    one.set_f(f0)
    ring = LocalSlabRing(nx, ny, 1.5, nslabs, bc=bc, obstacle_mask=mask, **kw)
    ring.set_variant(variant)
    ring.set_f(f0)
    total = 0
    for n in rng.integers(1, 30, size=3):
        ring.run_in_library(int(n))
        total += int(n)
    one.run(total)
    a, b = one.get_fields(("f", "rho", "u", "v")), ring.get_fields(("f", "rho", "u", "v"))
    for k in a:
        assert np.array_equal(a[k], b[k]), (bc, nx, ny, nslabs, variant, total, k)


@pytest.mark.parametrize("seed", range(int(os.environ.get("LB_RANDOM_RING_SEEDS", "16"))))
def test_random_self_ring(lbhip, seed):
    """The production slab path (lb_run with the RCCL exchange) as a one-rank periodic ring that sends its halo to itself:
    random shapes, variants (ten- / eight- / six-step cycle, no cycle, two- and single-step launches), masks and sequences of run
    lengths (every transition between cycles, lone first halves and launch-by-launch steps, with whatever ghost depth the
    previous run left) against the plain whole-grid handle, bit for bit.  (tools/ring_stress.py is the same in a loop.)"""
    from LB_D2Q9.simulation import Simulation, comm_unique_id
    rng = np.random.default_rng(9000 + (71 if seed == 0 else seed))      # 9071: the case that found the ghost-depth bug
    nx = int(rng.choice((512, 516, 768, 1000, 1024, 1284, 2048)))
    ny = int(rng.integers(8, 700))
    variant = int(rng.choice((-1, 97 | 256 | 4096, 97 | 256, 97, 97 | 128, 33, 1)))
    mask = None
    if rng.integers(0, 2):
        mask = rng.random((nx, ny)) < 0.03
    f0 = _random_state(rng, nx, ny)
    one = Simulation(nx, ny, 1.5, bc="periodic", obstacle_mask=mask)
    one.set_variant(0)
    one.set_f(f0)
    ring = Simulation(nx, ny, 1.5, bc="periodic", obstacle_mask=mask, halo=True)
    ring.comm_init(comm_unique_id(), 0, 1)
    ring.set_variant(variant)
    ring.set_exchange_inline(seed % 2 == 1)          # (odd seeds: the exchange between the interior launches, lb_set_exchange_inline)
    ring.set_f(f0)
    runs = [int(n) for n in rng.integers(1, 40, size=3)]
    for n in runs:
        ring.run(n)
    one.run(sum(runs))
    a, b = one.get_fields(("f", "rho", "u", "v")), ring.get_fields(("f", "rho", "u", "v"))
    for k in a:
        assert np.array_equal(a[k], b[k]), (nx, ny, variant, runs, k)
    one.close()
    ring.close()


@pytest.mark.timeout(1500, method="thread")
def test_slab_schedule_under_contention_200_partitions(lbhip):
    """The gating stress job for the slab path's stream / event web (with a high-priority edge stream ~2 % of
    random partitions mis-ordered when other processes shared the GPU; that stream is gone from the product and an audit of the
    cycle's read-after-write / write-after-read pairs finds every one ordered): 200 random partitions through lb_run_group with the
    members' streams ordered by EVENTS ALONE (lb_set_debug_sync(0), the schedule lb_run relies on), while two other processes
    -- started as children of the stress tool -- keep the GPU busy.  Every partition must equal the undivided run bit for bit."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, LB_DEBUG_SYNC="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "slab_stress.py"), "200", "2"], capture_output=True, text=True,
                       timeout=1400, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "200 seeds, 0 mismatching fields" in p.stdout
