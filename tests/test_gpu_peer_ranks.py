"""lb_run's multi-rank halo schedule executed by real rank PROCESSES that share the one GPU of the box (peer transport:
include/lb_hip.h lb_peer_export / lb_peer_connect; driver: tools/peer_ranks_check.py), each rank bitwise equal to the
undivided run -- and the transport's single-process form, a periodic ring that closes on itself."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LB_PEER_CHILD"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("ranks", [2, 4])
@pytest.mark.timeout(900, method="thread")
def test_rank_processes_on_one_gpu_equal_the_undivided_run_bitwise(lbhip, ranks):
    """2 and 4 rank processes on GPU 0: ten-, eight- and six-step halo cycles, launch-by-launch schedules (three-, two-, single-step
    kernels), lone first halves and remainders (runs of 20 + 7 + 4 steps), three boundary families, obstacle masks, slabs of
    unequal height; the halo rows travel through memory mapped across the processes, the ranks meet at device-side flags."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "peer_ranks_check.py"), "--ranks", str(ranks)],
                       capture_output=True, text=True, timeout=800, env=_env())
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "all cases bitwise equal" in p.stdout
    assert "= False" not in p.stdout
    assert p.stdout.count("= True") >= 15


def test_peer_transport_self_ring_equals_plain_run(lbhip):
    """One rank, periodic: the slab's neighbours are itself (its own descriptor is used in place, no IPC mapping): lb_run's cycle
    through the peer transport's kernels against the plain whole-grid handle, bit for bit; runs that walk through every
    transition between the cycle, the lone first half and the launch-by-launch schedule."""
    from LB_D2Q9.simulation import Simulation
    nx, ny = 1024, 160
    rng = np.random.default_rng(8)
    w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    f0 = (w[None, None, :] * (1 + 0.02 * rng.standard_normal((nx, ny, 9)))).astype(np.float32)
    mask = rng.random((nx, ny)) < 0.02
    for variant in (97 | 256 | 4096 | 16384 | 32768, 97 | 256 | 4096 | 16384, 97 | 256 | 4096, 97 | 256, 97, 33, 1):
        one = Simulation(nx, ny, 1.6, bc="periodic", obstacle_mask=mask)
        one.set_variant(0)
        ring = Simulation(nx, ny, 1.6, bc="periodic", obstacle_mask=mask, halo=True)
        ring.set_variant(variant)
        d_ = ring.MASK_HALO_ROWS
        ring.set_obstacle_mask_halo(mask[:, -d_:].T.copy(), mask[:, :d_].T.copy())
        d = ring.peer_export()
        ring.peer_connect(0, 1, d, d, ny)
        one.set_f(f0)
        ring.set_f(f0)
        for n in (29, 4, 16, 1, 8):
            one.run(n)
            ring.run(n)
        a, b = one.get_fields(("f", "rho", "u", "v")), ring.get_fields(("f", "rho", "u", "v"))
        for k in a:
            assert np.array_equal(a[k], b[k]), (variant, k)
        one.close()
        ring.close()


def test_peer_connect_refuses_what_does_not_fit(lbhip):
    from LB_D2Q9 import _native
    from LB_D2Q9.simulation import Simulation
    a = Simulation(640, 64, 1.2, bc="periodic", halo=True)
    b = Simulation(704, 64, 1.2, bc="periodic", halo=True)
    whole = Simulation(640, 64, 1.2, bc="periodic")
    with pytest.raises(_native.LbError):
        whole.peer_export()                                        # not a slab handle
    da, db = a.peer_export(), b.peer_export()
    with pytest.raises(_native.LbError):
        a.peer_connect(0, 1, db, db, 64)                           # another geometry
    with pytest.raises(_native.LbError):
        a.peer_connect(0, 1, b"\0" * len(da), None, 64)            # not a descriptor
    a.peer_connect(0, 1, da, da, 64)
    with pytest.raises(_native.LbError):
        a.peer_connect(0, 1, da, da, 64)                           # already attached
    a.run(8)


@pytest.mark.parametrize("transport", ["rccl", "peer"])
@pytest.mark.timeout(600, method="thread")
def test_bench_through_the_slab_path_on_one_gpu(lbhip, transport):
    """`bench.py --force-slab-path`: the code path `bench.py --gpus N` takes (DistributedSlab inside a torch.distributed group, the
    halo cycle inside lb_run, the collective health check) with ONE rank whose neighbours are itself, over either transport: one
    JSON line that names the transport."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-slab-path", "--transport", transport, "--size", "2048",
                        "--steps", "24", "--warmup", "8", "--min-timed-s", "0.1", "--no-cpu-baseline", "--calibrate", "0"],
                       capture_output=True, text=True, timeout=500, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and ("halo via " + transport) in line["config"]["workload"]
    assert line["health"]["n_nonfinite"] == 0 and 0 < line["roofline"]["frac"] <= 1.0
