#!/usr/bin/env python3
"""Time whole-grid runs of explicit kernel variants: tools/variant_time.py SIZE BC VARIANT[,VARIANT...] [--reps N] [--mask]

Prints k MLUPS and ms per launch (best of three timed blocks of 20 launches) per variant, alternating `reps` times."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2d-lb_amd"))
from LB_D2Q9.simulation import Simulation  # noqa: E402


def main():
    size, bc = sys.argv[1], sys.argv[2]
    nx, ny = (int(v) for v in size.split("x")) if "x" in size else (int(size), int(size))
    variants = [int(v) for v in sys.argv[3].split(",")]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
    mask = None
    if "--mask" in sys.argv:
        yy, xx = np.meshgrid(np.arange(ny), np.arange(nx))
        mask = (xx - nx // 4) ** 2 + (yy - ny // 2) ** 2 < (ny // 10) ** 2
    for _ in range(reps):
        for v in variants:
            s = Simulation(nx, ny, 1.7, bc=bc, inlet_rho=1.003, lid_u=0.05, obstacle_mask=mask)
            s.set_variant(v)
            spl = s.steps_per_launch()
            launches = 20
            s.run(2 * spl)
            s.sync()
            best = min(s.timed_run(launches * spl) for _ in range(3))
            print("%s %-8s variant %5d steps/launch %d  %7.1f k MLUPS  %.4f ms per launch  %s" % (
                size, bc, v, spl, nx * ny * launches * spl / best / 1e6, best / launches,
                " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("LB_") and k != "LB_LIB")), flush=True)
            s.close()


if __name__ == "__main__":
    main()
