#!/usr/bin/env python3
"""One-GPU proxy for the per-GPU work of an N-GPU strong-scaling run: a periodic (nx x ny/N) grid
run (a) as a plain whole-grid handle and (b) through the slab path as a 1-rank RCCL ring that sends
its halo to itself (edge bands first, exchange on the communication stream, two-step kernel).
Prints MLUPS for both; N x (b) is what N GPUs could reach if peer exchange costs what self exchange does."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=8192)
    ap.add_argument("--parts", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--variants", default="-1,9")
    ap.add_argument("--reps", type=int, default=3, help="timed runs per configuration; the best counts")
    ap.add_argument("--transports", default="rccl", help="comma list of rccl, peer (the slab path as a 1-rank ring over that transport)")
    ap.add_argument("--inline", action="store_true", help="the exchange between the interior launches (lb_set_exchange_inline)")
    ap.add_argument("--torch-dist", action="store_true",
                    help="first join a one-rank torch.distributed group over RCCL, as bench.py does: torch's and RCCL's own streams then exist "
                         "before the handles' and compete for the process's hardware queues (GPU_MAX_HW_QUEUES)")
    args = ap.parse_args()
    if args.torch_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29537")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        dist.barrier()
    from LB_D2Q9.simulation import Simulation, comm_unique_id
    from bench import shear_layer
    for parts in [int(p) for p in args.parts.split(",")]:
        ny = args.nx // parts
        for variant in [int(v) for v in args.variants.split(",")]:
            res = {}
            modes = ["plain"] + ["slab+%s-self" % t for t in args.transports.split(",")]
            for mode in modes:
                sim = Simulation(args.nx, ny, 1.7, bc="periodic", halo=(mode != "plain"))
                if variant >= 0:
                    sim.set_variant(variant)
                if mode == "slab+rccl-self":
                    sim.comm_init(comm_unique_id(), 0, 1)
                elif mode == "slab+peer-self":
                    d = sim.peer_export()
                    sim.peer_connect(0, 1, d, d, ny)
                if mode != "plain" and args.inline:
                    sim.set_exchange_inline(True)
                sim.init_equilibrium(*shear_layer(args.nx, ny, 0, ny))
                sim.run(10)
                if mode != "plain":
                    sim.exchange_timing(True)
                best = 0.0
                host = 1e9
                for _ in range(args.reps):
                    import time
                    t0 = time.perf_counter()
                    sim.run(args.steps, wait=False)      # host enqueue time only
                    host = min(host, (time.perf_counter() - t0) / args.steps * 1e6)
                    sim.sync()
                    ms = sim.timed_run(args.steps)
                    sim.sync()
                    best = max(best, args.nx * ny * args.steps / (ms * 1e-3) / 1e6)
                res[mode] = best
                res[mode + "_host_us"] = host
                if mode != "plain":
                    st = sim.exchange_stats()
                    res[mode + "_xchg"] = "halo cycle on depth %d, edge bands of %d rows, exchange %.0f us mean / %.0f us max" % (
                        st["cycle_depth"], st["band_rows"], 1e3 * st["total_ms"] / max(1, st["n"]), 1e3 * st["max_ms"])
                sim.close()
            for mode in modes[1:]:
                print("grid %5d x %5d (1/%d of %d^2) variant %3d: plain %9.1f MLUPS, slab path (%s) %9.1f MLUPS "
                      "(%.1f us/step GPU, %.0f us/step host enqueue; %s) -> x%d = %9.1f"
                      % (args.nx, ny, parts, args.nx, variant, res["plain"], mode[5:], res[mode],
                         args.nx * ny / res[mode], res[mode + "_host_us"], res[mode + "_xchg"], parts, parts * res[mode]), flush=True)


if __name__ == "__main__":
    main()
