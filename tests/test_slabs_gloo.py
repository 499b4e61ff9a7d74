"""Multi-process (world_size 2 and 3, gloo, CPU) test of the row-slab driver: DistributedSlab's
partition, neighbour and halo-exchange logic with the oracle standing in for the GPU engine must
reproduce the undivided oracle run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

torch = pytest.importorskip("torch")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, bc, steps, out_dir):
    for p in (os.path.join(ROOT, "2d-lb_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from LB_D2Q9.slabs import DistributedSlab
    from oracle_slab_engine import OracleSlabEngine
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        nx, ny = 40, 23
        rng = np.random.default_rng(42)
        w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
        f0 = (w[None, None, :] * (1 + 0.02 * rng.standard_normal((nx, ny, 9)))).astype(np.float32)
        mask = rng.random((nx, ny)) < 0.05
        mask[:, 0] = mask[:, -1] = False
        slab = DistributedSlab(nx, ny, 1.4, bc=bc, obstacle_mask=mask, transport="torch",
                               engine_factory=OracleSlabEngine, inlet_rho=1.01, lid_u=0.05)
        slab.set_f(f0)
        slab.run(steps)
        g = slab.get_fields(("f", "rho", "u", "v"))
        chk = slab.check()                  # collective: the three health scalars combined over the ranks
        if rank == 0:
            np.savez(os.path.join(out_dir, "out_%s_%d.npz" % (bc, world)), check=np.array(
                [chk["n_nonfinite"], chk["max_mach"], chk["sum_rho"]], np.float64), **g)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("bc", ["periodic", "pipe", "cavity"])
def test_distributed_slabs_equal_single_domain(oracle, tmp_path, bc, world):
    import torch.multiprocessing as mp
    steps = 12
    mp.spawn(_worker, args=(world, _free_port(), bc, steps, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "out_%s_%d.npz" % (bc, world)))
    nx, ny = 40, 23
    rng = np.random.default_rng(42)
    w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    f0 = (w[None, None, :] * (1 + 0.02 * rng.standard_normal((nx, ny, 9)))).astype(np.float32)
    mask = rng.random((nx, ny)) < 0.05
    mask[:, 0] = mask[:, -1] = False
    code = {"pipe": oracle.BC_PIPE, "periodic": oracle.BC_PERIODIC, "cavity": oracle.BC_CAVITY}[bc]
    ref = oracle.O2Sim(nx, ny, 1.4, code, 1.01, 1., 0.05, 1., mask=mask)
    ref.set_f(f0)
    ref.run(steps)
    want = ref.get_fields()
    for k in ("f", "rho", "u", "v"):
        assert np.array_equal(got[k], want[k]), (bc, world, k)
    # DistributedSlab.check: the per-rank scalars summed / maximised over the ranks == the whole lattice's
    f = want["f"].astype(np.float64)
    rho = f.sum(axis=2)
    ux = (f[..., 1] - f[..., 3] + f[..., 5] - f[..., 6] - f[..., 7] + f[..., 8]) / rho
    uy = (f[..., 5] + f[..., 2] + f[..., 6] - f[..., 7] - f[..., 4] - f[..., 8]) / rho
    n_bad, mach, mass = got["check"]
    assert n_bad == 0 and abs(mass - rho.sum()) <= 1e-9 * rho.sum()
    assert abs(mach - np.sqrt(3.0 * (ux * ux + uy * uy).max())) <= 1e-6


# ---- multi-rank checkpoint: written by 2 ranks, continued by 3 ---------------------------------------------------
def _case(bc):
    nx, ny = 40, 23
    rng = np.random.default_rng(42)
    w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    f0 = (w[None, None, :] * (1 + 0.02 * rng.standard_normal((nx, ny, 9)))).astype(np.float32)
    mask = rng.random((nx, ny)) < 0.05
    mask[:, 0] = mask[:, -1] = False
    return nx, ny, f0, mask


def _ckpt_worker(rank, world, port, bc, phase, steps, ckpt_dir, out_dir):
    for p in (os.path.join(ROOT, "2d-lb_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from LB_D2Q9.slabs import DistributedSlab
    from oracle_slab_engine import OracleSlabEngine
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        nx, ny, f0, mask = _case(bc)
        if phase == "save":
            # (numpy scalars as parameters: they must reach the manifest like Python floats; np.float32(1.01) is not 1.01)
            slab = DistributedSlab(nx, ny, 1.4, bc=bc, obstacle_mask=mask, transport="torch",
                                   engine_factory=OracleSlabEngine, inlet_rho=np.float64(1.01), lid_u=np.float32(0.05))
            slab.set_f(f0)
            slab.run(steps)
            slab.save_checkpoint(ckpt_dir)
        else:
            slab = DistributedSlab.from_checkpoint(ckpt_dir, transport="torch", engine_factory=OracleSlabEngine)
            assert slab.nranks == world and (slab.nx, slab.ny) == (nx, ny)
            from LB_D2Q9.slabs import _bc_code
            assert _bc_code(bc) == _bc_code(_bc_code(bc))        # a family named by number or by name is the same family
            slab.run(steps)
            g = slab.get_fields(("f", "rho", "u", "v"))
            if rank == 0:
                np.savez(os.path.join(out_dir, "resumed_%s.npz" % bc), **g)
            # a run with another omega refuses the checkpoint on every rank
            other = DistributedSlab(nx, ny, 1.5, bc=bc, obstacle_mask=mask, transport="torch",
                                    engine_factory=OracleSlabEngine, inlet_rho=1.01, lid_u=0.05)
            try:
                other.load_checkpoint(ckpt_dir)
                raise AssertionError("omega mismatch accepted")
            except ValueError:
                pass
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bc", ["periodic", "pipe"])
def test_multirank_checkpoint_written_by_two_ranks_resumes_on_three(oracle, tmp_path, bc):
    """DistributedSlab.save_checkpoint (per-rank shards + manifest) after 5 steps on 2 ranks; from_checkpoint on 3
    ranks re-cuts the rows and runs 7 more: bitwise equal to 12 steps of the single-domain oracle."""
    import json
    import torch.multiprocessing as mp
    ckpt = str(tmp_path / ("ckpt_" + bc))
    mp.spawn(_ckpt_worker, args=(2, _free_port(), bc, "save", 5, ckpt, str(tmp_path)), nprocs=2, join=True)
    man = json.load(open(os.path.join(ckpt, "manifest.json")))
    assert man["nranks"] == 2 and man["partition_rows"] == [[0, 12], [12, 11]] and man["has_mask"]
    assert man["params"] == {"inlet_rho": 1.01, "lid_u": float(np.float32(0.05))}            # numpy scalars are kept
    assert sorted(os.listdir(ckpt)) == ["manifest.json", "shard_0000.npz", "shard_0001.npz"]
    mp.spawn(_ckpt_worker, args=(3, _free_port(), bc, "resume", 7, ckpt, str(tmp_path)), nprocs=3, join=True)
    got = np.load(os.path.join(str(tmp_path), "resumed_%s.npz" % bc))
    nx, ny, f0, mask = _case(bc)
    code = {"pipe": oracle.BC_PIPE, "periodic": oracle.BC_PERIODIC}[bc]
    ref = oracle.O2Sim(nx, ny, 1.4, code, 1.01, 1., 0.05, 1., mask=mask)
    ref.set_f(f0)
    ref.run(12)
    want = ref.get_fields()
    for k in ("f", "rho", "u", "v"):
        assert np.array_equal(got[k], want[k]), (bc, k)


# ---- DistributedSlab.autotune: the ranks time the halo cycle's candidate depths TOGETHER and agree --------------------------------
class _TimedStubEngine(object):
    """An engine that only knows how long a run 'takes' on this rank at each depth of the halo cycle: enough for the collective
    choice (the kernels' own equivalence at every depth is the GPU suite's: tests/test_gpu_parity.py)."""
    MS_PER_STEP = {0: {8: 0.20, 7: 0.10, 6: 0.12, 5: 0.15}, 1: {8: 0.21, 7: 0.30, 6: 0.13, 5: 0.14}}       # rank 1 is slow at depth 7 (8: seven steps by k_deep2)

    def __init__(self, **kw):
        self.rank_of = None
        self.depth = 0
        self.steps = 0
        self._n = 0

    def set_slab_cycle(self, depth):
        self.depth = depth

    def sync(self):
        pass

    def timer_start(self):
        self._n = 0

    def run(self, n, wait=True):
        self._n += n
        self.steps += n

    def timer_stop(self):
        return self.MS_PER_STEP[self.rank_of][self.depth] * self._n


class _PlacedStubEngine(_TimedStubEngine):
    """... and at each placement of the exchange (lb_set_exchange_inline): rank 1 is slow at depth 7 only while the exchange overlaps."""
    MS_INLINE = {0: {8: 0.22, 7: 0.11, 6: 0.13, 5: 0.16}, 1: {8: 0.23, 7: 0.115, 6: 0.14, 5: 0.15}}

    def __init__(self, **kw):
        _TimedStubEngine.__init__(self, **kw)
        self.inline = False

    def set_exchange_inline(self, on):
        self.inline = bool(on)

    def timer_stop(self):
        return (self.MS_INLINE if self.inline else self.MS_PER_STEP)[self.rank_of][self.depth] * self._n


def _tune_worker(rank, world, port, out_dir, placed=False):
    for p in (os.path.join(ROOT, "2d-lb_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from LB_D2Q9.slabs import DistributedSlab
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        slab = DistributedSlab(64, 400, 1.4, bc="periodic", transport="torch",
                               engine_factory=_PlacedStubEngine if placed else _TimedStubEngine)
        slab.engine.rank_of = rank
        slab.transport = "peer"             # (what autotune asks: the schedule runs inside the engine)
        got = slab.autotune()
        np.savez(os.path.join(out_dir, "tune_%d.npz" % rank), depth=got["depth"], steps=got["steps"], engine_depth=slab.engine.depth,
                 engine_steps=slab.engine.steps, t7=got["ms_per_step"][7], t6=got["ms_per_step"][6], t5=got["ms_per_step"][5],
                 inline=got["exchange_inline"], engine_inline=getattr(slab.engine, "inline", False),
                 i7=got.get("ms_per_step_inline", {}).get(7, -1.0))
    finally:
        dist.destroy_process_group()


def test_distributed_slab_autotune_agrees_on_the_slowest_ranks_best_depth(tmp_path):
    """Two ranks, 200 rows each (>= 16 x 7: every candidate applies): rank 0 alone would pick depth 7, rank 1 is three times
    slower there -- the MAX over ranks decides, both set depth 6 and both have advanced the same number of live steps."""
    import torch.multiprocessing as mp
    mp.spawn(_tune_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [np.load(os.path.join(str(tmp_path), "tune_%d.npz" % k)) for k in range(2)]
    for k in range(2):
        assert int(r[k]["depth"]) == 6 and int(r[k]["engine_depth"]) == 6
        assert abs(float(r[k]["t7"]) - 0.30) < 1e-9 and abs(float(r[k]["t6"]) - 0.13) < 1e-9 and abs(float(r[k]["t5"]) - 0.15) < 1e-9
        assert int(r[k]["steps"]) == int(r[k]["engine_steps"]) == 3 * 20 * 2 * (7 + 7 + 6 + 5)
        assert not bool(r[k]["inline"])                      # (an engine without lb_set_exchange_inline: nothing to place)


def test_distributed_slab_autotune_places_the_exchange_where_the_slowest_rank_is_fastest(tmp_path):
    """The same two ranks on an engine that can also run the exchange between the interior launches: there rank 1 is not slow at
    depth 7 -- 0.115 ms per step is the best MAX over ranks of the six candidates; both ranks set depth 7 AND the inline exchange."""
    import torch.multiprocessing as mp
    mp.spawn(_tune_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
    r = [np.load(os.path.join(str(tmp_path), "tune_%d.npz" % k)) for k in range(2)]
    for k in range(2):
        assert int(r[k]["depth"]) == 7 and int(r[k]["engine_depth"]) == 7
        assert bool(r[k]["inline"]) and bool(r[k]["engine_inline"])
        assert abs(float(r[k]["t7"]) - 0.30) < 1e-9 and abs(float(r[k]["i7"]) - 0.115) < 1e-9
        assert int(r[k]["steps"]) == int(r[k]["engine_steps"]) == 2 * 3 * 20 * 2 * (7 + 7 + 6 + 5)
