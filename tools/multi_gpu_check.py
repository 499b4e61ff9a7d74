#!/usr/bin/env python3
"""Multi-GPU correctness check for a node with >= 2 GPUs (none was available while this was written):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/multi_gpu_check.py

Rank r drives GPU r; the slabs exchange halos over RCCL inside lb_run (two-step kernel, odd and even
step counts) and the gathered result is compared bitwise with the undivided run on rank 0's GPU.
With --same-gpu all ranks use GPU 0: RCCL rejects that ("invalid usage", duplicate device) - tried on
the 1-GPU pool, which is why the in-library virtual slabs and the 1-rank self-ring exist."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]


def main():
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import DistributedSlab
    nx, ny = 1024, 256
    rng = np.random.default_rng(3)
    w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    f0 = (w[None, None, :] * (1 + 0.02 * rng.standard_normal((nx, ny, 9)))).astype(np.float32)
    same = "--same-gpu" in sys.argv
    dev = 0 if same else int(os.environ.get("LOCAL_RANK", "0"))
    for bc in ("periodic", "pipe", "cavity"):
        slab = DistributedSlab(nx, ny, 1.5, bc=bc, transport="rccl", device=dev, inlet_rho=1.004, lid_u=0.05)
        slab.engine.set_variant(33)
        slab.set_f(f0)
        slab.run(7)
        slab.run(4)
        g = slab.get_fields(("f", "rho"))
        if rank == 0:
            one = Simulation(nx, ny, 1.5, bc=bc, inlet_rho=1.004, lid_u=0.05, device=dev)
            one.set_variant(0)
            one.set_f(f0)
            one.run(11)
            h = one.get_fields(("f", "rho"))
            print("%d ranks, bc=%s: bitwise equal to the undivided run = %s" % (world, bc, all(np.array_equal(g[k], h[k]) for k in g)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
