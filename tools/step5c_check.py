#!/usr/bin/env python3
"""k1_step5 (the Cython path's five-step marching kernel, variant bit 12) against the single-step pass k1_fstep (variant 0), bit for
bit, and timed against the LDS tiles (variant 512):  python tools/step5c_check.py [--no-time]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2d-lb_amd"))
from LB_D2Q9.simulation import Simulation  # noqa: E402

W = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)


def make(nx, ny, variant, mask, f0, rho0, u0, v0):
    s = Simulation(nx, ny, 1.3, bc="pipe", semantics="cython", inlet_rho=1.004, outlet_rho=1.0, obstacle_mask=mask)
    s.set_variant(variant)
    s.set_fields(rho0, u0, v0)
    s.set_f(f0)
    return s


def main():
    bad = 0
    for nx, ny, masked in ((1003, 177, False), (1003, 177, True), (512, 128, True), (744, 300, False), (2048, 640, True),
                           (3751, 1251, True)):
        rng = np.random.default_rng(nx + ny)
        f0 = (W[None, None, :] * (1 + 0.02 * rng.standard_normal((nx, ny, 9)))).astype(np.float32)
        rho0 = f0.sum(axis=2)
        u0 = (0.01 * rng.standard_normal((nx, ny))).astype(np.float32)
        v0 = (0.01 * rng.standard_normal((nx, ny))).astype(np.float32)
        mask = None
        if masked:
            mask = rng.random((nx, ny)) < 0.02
            mask[0, :] = mask[-1, :] = False
            mask[:, 0] = mask[:, -1] = False
        out = []
        for variant in (0, 4096 | 512):
            s = make(nx, ny, variant, mask, f0, rho0, u0, v0)
            if variant:
                assert s.steps_per_launch() == 5 and "k1_step5" in s.hot_kernel(), s.hot_kernel()
            s.run(5); s.run(13); s.run(9); s.run(10)
            out.append(s.get_fields(("f", "rho", "u", "v")))
            s.close()
        for k in out[0]:
            if not np.array_equal(out[0][k], out[1][k], equal_nan=True):
                bad += 1
                d = np.abs(out[0][k].astype(np.float64) - out[1][k])
                idx = np.argwhere(~(d == 0))
                print("MISMATCH", nx, ny, masked, k, "max", np.nanmax(d), "n", len(idx), "first", idx[:6].tolist(), "last", idx[-3:].tolist())
        print("checked", nx, ny, "mask" if masked else "", "finite" if np.all(np.isfinite(out[0]["f"])) else "NON-FINITE", flush=True)
    print("bitwise mismatches:", bad)
    if "--no-time" not in sys.argv:
        for nx, ny in ((3751, 1251), (4096, 4096), (8192, 8192)):
            yy, xx = np.meshgrid(np.arange(ny), np.arange(nx))
            mask = (xx - nx // 4) ** 2 + (yy - ny // 2) ** 2 < (ny // 10) ** 2          # a cylinder, as in the reference's case
            for name, variant in (("k1_tile4", 512), ("k1_step5", 4096 | 512), ("k1_tile4", 512), ("k1_step5", 4096 | 512)):
                s = Simulation(nx, ny, 1.3, bc="pipe", semantics="cython", inlet_rho=1.004, outlet_rho=1.0, obstacle_mask=mask)
                s.set_variant(variant)
                spl = s.steps_per_launch()
                n = 200
                s.run(n)
                s.sync()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    s.run(n)
                    s.sync()
                    best = min(best, time.perf_counter() - t0)
                print("%5d x %5d %-9s steps/launch %d  %7.1f k MLUPS (wall clock around run(200))" % (
                    nx, ny, name, spl, nx * ny * n / best / 1e9), flush=True)
                s.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
