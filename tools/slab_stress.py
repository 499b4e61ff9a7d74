#!/usr/bin/env python3
"""Random slab partitions (tests/test_gpu_random.py::test_random_slab_partition) while other processes keep the GPU busy: the
multi-stream schedule of lb_run_group must not depend on timing.  Prints where a mismatch sits.
    python tools/slab_stress.py [seeds] [noise processes]"""
import os
import signal
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT, os.path.join(ROOT, "tests")]

NOISE = r'''
import os, sys
sys.path[:0] = [os.path.join(%r, "2d-lb_amd"), %r]
from LB_D2Q9.simulation import Simulation
from bench import shear_layer
s = Simulation(2048, 2048, 1.7, bc="periodic"); s.init_equilibrium(*shear_layer(2048, 2048, 0, 2048))
while True:
    s.run(200)
''' % (ROOT, ROOT)


def main():
    signal.signal(signal.SIGTERM, lambda *a: sys.exit(143))       # (`timeout` must not leave the noise processes behind)
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    nnoise = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import LocalSlabRing, partition_rows
    from test_gpu_parity import _random_state
    noise = [subprocess.Popen([sys.executable, "-c", NOISE]) for _ in range(nnoise)]
    bad = 0
    try:
        for seed in range(seeds):
            rng = np.random.default_rng(5000 + seed)
            bc = ("periodic", "pipe", "cavity")[seed % 3]
            nx = int(rng.choice((512, 516, 768, 1000, 1024, 1284)))
            nslabs = int(rng.integers(2, 6))
            ny = int(rng.integers(nslabs * 7, 700))
            variant = int(rng.choice((-1, 97 | 256, 97, 97 | 128, 33, 1)))
            mask = None
            if rng.integers(0, 2):
                mask = rng.random((nx, ny)) < 0.03
                mask[0, :] = mask[-1, :] = False
                if bc != "periodic":
                    mask[:, 0] = mask[:, -1] = False
            kw = dict(inlet_rho=1.005, lid_u=0.05)
            f0 = _random_state(rng, nx, ny)
            one = Simulation(nx, ny, 1.5, bc=bc, obstacle_mask=mask, **kw)
            one.set_variant(0)
            one.set_f(f0)
            ring = LocalSlabRing(nx, ny, 1.5, nslabs, bc=bc, obstacle_mask=mask, **kw)
            ring.set_variant(variant)
            ring.set_f(f0)
            runs = [int(n) for n in rng.integers(1, 30, size=3)]
            for n in runs:
                ring.run_in_library(n)
            one.run(sum(runs))
            a, b = one.get_fields(("f", "rho", "u", "v")), ring.get_fields(("f", "rho", "u", "v"))
            parts = partition_rows(ny, nslabs)
            for k in a:
                if not np.array_equal(a[k], b[k]):
                    bad += 1
                    d = a[k] != b[k]
                    if d.ndim == 3:
                        d = d.any(axis=2)
                    rows = np.nonzero(d.any(axis=0))[0]
                    cols = np.nonzero(d.any(axis=1))[0]
                    runs_of_rows = []                         # maximal runs of consecutive differing rows
                    for r in rows:
                        if runs_of_rows and r == runs_of_rows[-1][1] + 1:
                            runs_of_rows[-1][1] = int(r)
                        else:
                            runs_of_rows.append([int(r), int(r)])
                    print("seed %d %s %dx%d slabs %d variant %d runs %s mask %d: %s differs in %d cells, rows %s, cols %d..%d; "
                          "partition %s" % (seed, bc, nx, ny, nslabs, variant, runs, mask is not None, k, int(d.sum()),
                                            runs_of_rows, cols.min(), cols.max(), parts), flush=True)
            one.close()
            for s in ring.slabs:
                s.close()
    finally:
        for p in noise:
            p.kill()
    print("%d seeds, %d mismatching fields" % (seeds, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
