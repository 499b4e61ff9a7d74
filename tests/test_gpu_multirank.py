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


def test_slabs_over_rccl_equal_the_undivided_run_bitwise(lbhip):
    """tools/multi_gpu_check.py under torch.distributed.run: automatic kernel choice (eight-step halo cycle on
    k_step4) and every explicit schedule x three boundary families x obstacle mask, each rank against the
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
