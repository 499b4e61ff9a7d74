#!/usr/bin/env python3
"""k_step5 (variant bit 12) against the single-step kernel (bitwise) and against k_step4 (time).

    python tools/step5_check.py [--no-time] [--sizes 8192,4096]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2d-lb_amd"))
from LB_D2Q9.simulation import Simulation  # noqa: E402

W = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
SIX = "--six" in sys.argv                       # k_deep<6> (variant bit 14) instead of k_step5 (bit 12)
SEVEN = "--seven" in sys.argv                   # k_deep<7> (variant bit 15)
DEEP2 = "--deep2" in sys.argv                   # k_deep2 (variant bit 16: the seven-step launches by two waves per strip and direction)
SEVEN = SEVEN or DEEP2
DEEP = 97 | 256 | 4096 | (16384 if SIX or SEVEN else 0) | (32768 if SEVEN else 0) | (65536 if DEEP2 else 0)
DEEP_SPL = 7 if SEVEN else (6 if SIX else 5)


def state(rng, nx, ny, amp=0.02):
    return (W[None, None, :] * (1 + amp * rng.standard_normal((nx, ny, 9)))).astype(np.float32)


def bitwise():
    bad = 0
    for bc, nx, ny, masked in (("periodic", 1000, 130, False), ("periodic", 512, 128, True), ("pipe", 1003, 177, False),
                               ("pipe", 1003, 177, True), ("cavity", 777, 201, True), ("cavity", 777, 201, False),
                               ("pipe", 2048, 300, False), ("periodic", 2048, 1024, False), ("pipe_i", 1024, 640, True)):
        rng = np.random.default_rng(nx + ny)
        f0 = state(rng, nx, ny)
        mask = None
        if masked:
            mask = rng.random((nx, ny)) < 0.03
            mask[0, :] = mask[-1, :] = False
            mask[:, 0] = mask[:, -1] = False
        out = []
        for variant in (0, DEEP):
            kw = dict(inlet_rho=1.004, lid_u=0.06)
            if bc == "pipe_i":
                s = Simulation(nx, ny, 1.6, bc="pipe", obstacle_mask=mask, semantics="d2q9i", **kw)
            else:
                s = Simulation(nx, ny, 1.6, bc=bc, obstacle_mask=mask, **kw)
            s.set_variant(variant)
            if variant:
                assert s.steps_per_launch() == DEEP_SPL, s.steps_per_launch()
                assert ("k_deep2" in s.hot_kernel()) == DEEP2, s.hot_kernel()
            s.set_f(f0)
            s.run(7)
            s.run(13)
            s.run(5)
            out.append(s.get_fields(("f", "rho", "u", "v")))
        for k in ("f", "rho", "u", "v"):
            same = np.array_equal(out[0][k], out[1][k], equal_nan=True)
            if not same:
                bad += 1
                d = np.abs(out[0][k].astype(np.float64) - out[1][k])
                idx = np.argwhere(d > 0)
                print("MISMATCH", bc, nx, ny, masked, k, "max", d.max(), "n", len(idx), "first", idx[:5].tolist())
        print("checked", bc, nx, ny, "mask" if masked else "")
    return bad


def timing(sizes):
    for n in sizes:
        for bc in ("periodic", "pipe"):
            for name, variant in (("k_deep<6>", 353 | 4096 | 16384), ("k_deep<7>", 353 | 4096 | 16384 | 32768), ("k_deep2<7>", 353 | 4096 | 16384 | 32768 | 65536),
                                  ("k_deep<6>", 353 | 4096 | 16384), ("k_deep<7>", 353 | 4096 | 16384 | 32768), ("k_deep2<7>", 353 | 4096 | 16384 | 32768 | 65536)):
                s = Simulation(n, n, 1.7, bc=bc, inlet_rho=1.003)
                s.set_variant(variant)
                spl = s.steps_per_launch()
                launches = 20
                s.run(2 * spl)
                s.sync()
                best = 1e9
                for _ in range(3):
                    best = min(best, s.timed_run(launches * spl))
                print("%5d^2 %-8s %s steps/launch %d  %.1f k MLUPS  (%.3f ms per launch)" % (
                    n, bc, name, spl, n * n * launches * spl / best / 1e6, best / launches), flush=True)
                s.close()


if __name__ == "__main__":
    bad = bitwise()
    print("bitwise mismatches:", bad)
    if "--no-time" not in sys.argv:
        sizes = [8192, 4096]
        if "--sizes" in sys.argv:
            sizes = [int(x) for x in sys.argv[sys.argv.index("--sizes") + 1].split(",")]
        timing(sizes)
    sys.exit(1 if bad else 0)
