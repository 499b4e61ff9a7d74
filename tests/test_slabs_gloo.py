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
