"""The RCCL halo path with MORE THAN ONE rank (one process per GPU).  Needs a node with >= 2 GPUs: on the
1-GPU boxes these tests skip, and the path is covered by its single-device stand-ins (in-library virtual slabs,
the 1-rank RCCL self-ring: tests/test_gpu_parity.py, tests/test_gpu_fullsize.py)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _gpus(lbhip):
    n = lbhip.lb_device_count()
    return max(n, 0)


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


# (the subprocess timeouts stay below the pytest timeout of these tests, so that a hung multi-rank run is reaped by
#  subprocess.run -- which kills its child -- before pytest's thread-method timeout ends the whole session)
@pytest.mark.timeout(2400, method="thread")
def test_slabs_over_rccl_equal_the_undivided_run_bitwise(lbhip):
    """tools/multi_gpu_check.py under torch.distributed.run: automatic kernel choice (ten-step halo cycle on
    k_step5) and every explicit schedule x three boundary families x obstacle mask, each rank against the
    undivided single-step run, bit for bit."""
    n = _gpus(lbhip)
    if n < 2:
        pytest.skip("needs >= 2 GPUs (%d visible)" % n)
    n = min(n, 8)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                        "--master-addr", "127.0.0.1", "--master-port", "29611",
                        os.path.join(ROOT, "tools", "multi_gpu_check.py")],
                       capture_output=True, text=True, timeout=1800, env=_env())
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "all cases bitwise equal" in p.stdout
    assert "= False" not in p.stdout


@pytest.mark.timeout(2400, method="thread")
def test_slabs_over_the_peer_transport_across_gpus_equal_the_undivided_run_bitwise(lbhip):
    """tools/peer_ranks_check.py with one rank process per GPU: the same schedules over the peer transport (halo rows stored into
    the neighbour GPU's ghost rows over xGMI through IPC-mapped memory, flags in fine-grained memory).  On one GPU the same tool
    runs in tests/test_gpu_peer_ranks.py; this is its multi-GPU form."""
    n = _gpus(lbhip)
    if n < 2:
        pytest.skip("needs >= 2 GPUs (%d visible)" % n)
    n = min(n, 8)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "peer_ranks_check.py"), "--ranks", str(n), "--gpus", str(n)],
                       capture_output=True, text=True, timeout=1800, env=_env())
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "all cases bitwise equal" in p.stdout and "= False" not in p.stdout


@pytest.mark.timeout(1800, method="thread")
def test_bench_spawns_its_own_ranks(lbhip):
    """`python bench.py --gpus N` as the driver calls it (no launcher): one JSON line, n_gpus = N, RCCL transport."""
    n = _gpus(lbhip)
    if n < 2:
        pytest.skip("needs >= 2 GPUs (%d visible)" % n)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "8",
                        "--size", "4096", "--min-timed-s", "0.1"], capture_output=True, text=True, timeout=1200, env=_env())
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "halo via rccl" in line["config"]["workload"]
    assert 0 < line["roofline"]["frac"] <= 1.0


def test_distributed_checkpoint_on_the_real_engine_single_rank(lbhip, tmp_path):
    """DistributedSlab.save_checkpoint / from_checkpoint with the HIP engine behind it (one rank, gloo group,
    python-driven halo exchange with itself): the resumed run continues bit for bit, obstacle rows of the
    'neighbour' included.  (The re-cut onto another rank count runs under gloo in tests/test_slabs_gloo.py.)"""
    import socket
    import numpy as np
    import torch.distributed as dist
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import DistributedSlab
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        nx, ny = 640, 96
        rng = np.random.default_rng(11)
        w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
        f0 = (w[None, None, :] * (1 + 0.02 * rng.standard_normal((nx, ny, 9)))).astype(np.float32)
        mask = rng.random((nx, ny)) < 0.03
        one = Simulation(nx, ny, 1.35, bc="periodic", obstacle_mask=mask)
        one.set_f(f0)
        one.run(21)
        a = DistributedSlab(nx, ny, 1.35, bc="periodic", obstacle_mask=mask, transport="torch", device=0)
        a.set_f(f0)
        a.run(9)
        a.save_checkpoint(str(tmp_path / "ck"))
        b = DistributedSlab.from_checkpoint(str(tmp_path / "ck"), transport="torch", device=0)
        b.run(12)
        g, h = b.get_local_fields(("f", "rho", "u", "v")), one.get_fields(("f", "rho", "u", "v"))
        for k in g:
            assert np.array_equal(g[k], h[k]), k
    finally:
        dist.destroy_process_group()
