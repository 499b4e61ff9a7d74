#!/usr/bin/env python3
"""The production slab path (lb_run with the RCCL exchange) as a one-rank periodic ring that sends its halo to itself, random
shapes / variants / run lengths, against the plain whole-grid handle, bit for bit -- meant to be run several times at once so
that the processes disturb each other's timing.    python tools/ring_stress.py [seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT, os.path.join(ROOT, "tests")]


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    from LB_D2Q9.simulation import Simulation, comm_unique_id
    from test_gpu_parity import _random_state
    bad = 0
    for seed in range(seeds):
        rng = np.random.default_rng(9000 + seed)
        nx = int(rng.choice((512, 516, 768, 1000, 1024, 1284, 2048)))
        ny = int(rng.integers(8, 700))
        variant = int(rng.choice((-1, 97 | 256 | 4096, 97 | 256, 97, 97 | 128, 33, 1)))
        mask = None
        if rng.integers(0, 2):
            mask = rng.random((nx, ny)) < 0.03
        f0 = _random_state(rng, nx, ny)
        # every third case a walled family: a whole grid flagged as a slab with a 1-rank communicator has no neighbour (the
        # exchange degenerates to an empty group) but runs lb_run's slab schedule all the same
        bc = ("periodic", "periodic", ("pipe", "cavity")[(seed // 3) % 2])[seed % 3]
        kw = dict(inlet_rho=1.004, lid_u=0.05)
        if mask is not None and bc != "periodic":
            mask[0, :] = mask[-1, :] = False
            mask[:, 0] = mask[:, -1] = False
        one = Simulation(nx, ny, 1.5, bc=bc, obstacle_mask=mask, **kw)
        one.set_variant(0)
        one.set_f(f0)
        ring = Simulation(nx, ny, 1.5, bc=bc, obstacle_mask=mask, halo=True, **kw)
        ring.comm_init(comm_unique_id(), 0, 1)
        ring.set_variant(variant)
        ring.set_f(f0)
        runs = [int(n) for n in rng.integers(1, 40, size=3)]
        for n in runs:
            ring.run(n)
        one.run(sum(runs))
        a, b = one.get_fields(("f", "rho", "u", "v")), ring.get_fields(("f", "rho", "u", "v"))
        for k in a:
            if not np.array_equal(a[k], b[k]):
                bad += 1
                d = a[k] != b[k]
                if d.ndim == 3:
                    d = d.any(axis=2)
                rows = np.nonzero(d.any(axis=0))[0]
                print("seed %d %s %dx%d variant %d runs %s mask %d: %s differs in %d cells, rows %d..%d" % (
                    seed, bc, nx, ny, variant, runs, mask is not None, k, int(d.sum()), rows.min(), rows.max()), flush=True)
        one.close()
        ring.close()
    print("%d seeds, %d mismatching fields" % (seeds, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
