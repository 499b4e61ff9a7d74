#!/usr/bin/env python3
"""lb_run's multi-rank schedule executed by REAL rank processes, on whatever GPUs there are -- one GPU is enough:

    python tools/peer_ranks_check.py --ranks 4 [--quick] [--gpus 1]

The parent starts `--ranks` child processes (rank r on GPU r % gpus; the default is to put them all on GPU 0) and waits.
Every child joins a gloo process group (CPU tensors: the bootstrap only carries the peer descriptors), builds its slab with
DistributedSlab(transport='peer') -- the halo rows are stored straight into the neighbours' ghost rows through device memory
mapped across the processes, the ranks meet at device-side flags (include/lb_hip.h, lb_peer_connect) -- runs the cases of
tools/multi_gpu_check.py and compares its rows with the UNDIVIDED grid, which it computes itself with the single-step kernel,
bit for bit.  RCCL refuses several ranks on one device; this transport does not, so the ten- / eight- / six-step halo cycles, the
lone first half, the launch-by-launch remainder and the initial exchange of lb_run all execute with nranks > 1 on a 1-GPU box.
Prints one line per case (rank 0) and exits non-zero on a mismatch."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]


def child():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ["LB_CHECK_DEVICE"])
    quick = "--quick" in sys.argv
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import DistributedSlab
    w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    bad = 0
    cases = [(2048, 1024 * world, -1, (33, 14, 5), ("periodic",))] if not quick else []       # automatic: fourteen-step cycle on k_deep<7>
    # explicit schedules on small slabs: ten-step cycle, eight-step cycle, six-step cycle, three-step launches without the cycle,
    # two-step, single-step
    cases += [(1024, 128 * world + 5, v, (20, 7, 4), ("periodic", "pipe", "cavity"))
              for v in ((97 | 256 | 4096 | 16384 | 32768 | 65536, 97 | 256 | 4096 | 16384 | 32768, 97 | 256 | 4096 | 16384, 97 | 256 | 4096, 97 | 256, 97, 97 | 128, 33, 1) if not quick else (97 | 256 | 4096 | 16384 | 32768 | 65536, 97 | 256 | 4096 | 16384 | 32768, 97 | 256 | 4096, 97 | 256, 1))]
    for nx, ny, variant, runs, families in cases:
        rng = np.random.default_rng(3)
        f0 = (w[None, None, :] * (1 + 0.02 * rng.standard_normal((nx, ny, 9)))).astype(np.float32)
        mask = rng.random((nx, ny)) < 0.02
        mask[0, :] = mask[-1, :] = False
        for bc in families:
            m = mask.copy()
            if bc != "periodic":
                m[:, 0] = m[:, -1] = False
            kw = dict(inlet_rho=1.004, lid_u=0.05)
            slab = DistributedSlab(nx, ny, 1.5, bc=bc, obstacle_mask=m, transport="peer", device=dev, **kw)
            slab.engine.set_variant(variant)
            slab.engine.set_exchange_inline("--inline" in sys.argv)       # (the exchange between the interior launches, every rank alike)
            spl = slab.engine.steps_per_launch()
            slab.set_f(f0)
            for n in runs:
                slab.run(n)                      # (waits: lb_sync reports a neighbour that never arrived)
            g = slab.get_local_fields(("f", "rho", "u", "v"))
            one = Simulation(nx, ny, 1.5, bc=bc, obstacle_mask=m, device=dev, **kw)
            one.set_variant(0)
            one.set_f(f0)
            one.run(sum(runs))
            h = one.get_fields(("f", "rho", "u", "v"))
            ok = all(np.array_equal(g[k], h[k][:, slab.y0:slab.y0 + slab.h]) for k in g)
            t = torch.tensor([0 if ok else 1], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bad += int(t[0])
            if rank == 0:
                print("%d rank processes (peer transport), %dx%d, bc=%s, variant=%d (%d steps per launch), runs=%s: bitwise equal "
                      "to the undivided run = %s" % (world, nx, ny, bc, variant, spl, list(runs), not int(t[0])), flush=True)
            dist.barrier()                       # nobody unmaps a lattice a neighbour may still be storing into
            slab.engine.close()
            one.close()
    dist.barrier()
    dist.destroy_process_group()
    if bad:
        raise SystemExit("peer_ranks_check: %d case(s) differ" % bad)
    if rank == 0:
        print("peer_ranks_check: all cases bitwise equal", flush=True)


def parent():
    import socket
    n = int(sys.argv[sys.argv.index("--ranks") + 1]) if "--ranks" in sys.argv else 2
    gpus = int(sys.argv[sys.argv.index("--gpus") + 1]) if "--gpus" in sys.argv else 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LB_CHECK_DEVICE=str(r % gpus), LB_PEER_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    failed = None
    while failed is None:                        # a rank that dies leaves its peers waiting: take them down with it
        codes = [p.poll() for p in procs]
        failed = next((r for r, c in enumerate(codes) if c not in (None, 0)), None)
        if failed is None and all(c == 0 for c in codes):
            return 0
        time.sleep(0.2)
    for p in procs:                              # (exact PIDs of the children started above)
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    print("peer_ranks_check: rank %d exited with status %s" % (failed, procs[failed].returncode), file=sys.stderr)
    return 1


if __name__ == "__main__":
    if os.environ.get("LB_PEER_CHILD"):
        child()
    else:
        sys.exit(parent())
