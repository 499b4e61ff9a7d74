#!/usr/bin/env python3
"""Multi-GPU correctness check for a node with >= 2 GPUs:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/multi_gpu_check.py

Rank r drives GPU r; the slabs exchange halos over RCCL inside lb_run.  Every rank also runs the UNDIVIDED grid
on its own GPU with the single-step kernel and compares its rows of the slab run with it bit for bit (no gather).
Cases: the automatic kernel choice (variant -1) on slabs of 2048 x 1024 cells -- the deepest halo cycle that fits, on
k_step5 that bench.py --gpus N runs -- and every explicit schedule (ten-step, eight-step, six-step cycle, three-step
launches without the cycle, two-step, single-step) on small slabs; three boundary families, with an obstacle mask;
step counts that are and are not multiples of the cycle.  Prints one line per case and exits non-zero on a mismatch.
With --same-gpu all ranks use GPU 0: RCCL rejects that ("invalid usage", duplicate device)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]


def main():
    import torch
    import torch.distributed as dist
    same = "--same-gpu" in sys.argv
    quick = "--quick" in sys.argv
    dev = 0 if same else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    rank, world = dist.get_rank(), dist.get_world_size()
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import DistributedSlab
    w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    bad = 0
    cases = [(2048, 1024 * world, -1, (23, 10, 5))]                      # automatic: ten-step cycle on k_step5
    if not quick:
        cases += [(1024, 128 * world, v, (20, 7, 4)) for v in (97 | 256 | 4096 | 16384 | 32768, 97 | 256 | 4096 | 16384, 97 | 256 | 4096, 97 | 256, 97, 97 | 128, 33, 1)]
    for nx, ny, variant, runs in cases:
        rng = np.random.default_rng(3)
        f0 = (w[None, None, :] * (1 + 0.02 * rng.standard_normal((nx, ny, 9)))).astype(np.float32)
        mask = rng.random((nx, ny)) < 0.02
        mask[0, :] = mask[-1, :] = False
        for bc in ("periodic", "pipe", "cavity"):
            m = mask.copy()
            if bc != "periodic":
                m[:, 0] = m[:, -1] = False
            kw = dict(inlet_rho=1.004, lid_u=0.05)
            slab = DistributedSlab(nx, ny, 1.5, bc=bc, obstacle_mask=m, transport="rccl", device=dev, **kw)
            slab.engine.set_variant(variant)
            spl = slab.engine.steps_per_launch()
            slab.set_f(f0)
            for n in runs:
                slab.run(n)
            g = slab.get_local_fields(("f", "rho", "u", "v"))
            one = Simulation(nx, ny, 1.5, bc=bc, obstacle_mask=m, device=dev, **kw)
            one.set_variant(0)
            one.set_f(f0)
            one.run(sum(runs))
            h = one.get_fields(("f", "rho", "u", "v"))
            ok = all(np.array_equal(g[k], h[k][:, slab.y0:slab.y0 + slab.h]) for k in g)
            t = torch.tensor([0 if ok else 1], dtype=torch.int32, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bad += int(t[0])
            if rank == 0:
                print("%d ranks, %dx%d, bc=%s, variant=%d (%d steps per launch), runs=%s: bitwise equal to the undivided "
                      "run = %s" % (world, nx, ny, bc, variant, spl, list(runs), not int(t[0])), flush=True)
            slab.engine.close()
            one.close()
    dist.barrier()
    dist.destroy_process_group()
    if bad:
        raise SystemExit("multi_gpu_check: %d case(s) differ" % bad)
    if rank == 0:
        print("multi_gpu_check: all cases bitwise equal", flush=True)


if __name__ == "__main__":
    main()
