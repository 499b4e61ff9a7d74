#!/usr/bin/env python3
"""Long-run soak on the GPU box: the automatic kernel choice (autotune included) against the single-step kernel, bit for bit,
over a few hundred steps in run() calls of odd lengths, four families at large sizes.    python tools/soak_bitwise.py"""
import os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]
from LB_D2Q9.simulation import Simulation
def case(bc, nx, ny, steps, masked, **kw):
    rng = np.random.default_rng(3)
    mask = None
    if masked:
        mask = rng.random((nx, ny)) < 0.01
        mask[0, :] = mask[-1, :] = False; mask[:, 0] = mask[:, -1] = False
    rho = (1.0 + 1e-3 * rng.standard_normal((nx, ny))).astype(np.float32)
    u = (0.02 + 1e-3 * rng.standard_normal((nx, ny))).astype(np.float32)
    v = (1e-3 * rng.standard_normal((nx, ny))).astype(np.float32)
    out = []
    for variant in (int(os.environ.get("LB_SOAK_VARIANT", "-1")), 9):       # (LB_SOAK_VARIANT=119137: k_deep2 for the seven-step launches)
        s = Simulation(nx, ny, 1.2, bc=bc, obstacle_mask=mask, **kw)
        s.set_variant(variant)
        s.init_equilibrium(rho, u, v)
        for chunk in (steps // 2, steps - steps // 2 - 7, 7):
            s.run(chunk)
        out.append(s.get_fields(("f", "rho", "u", "v")))
        s.close()
    ok = all(np.array_equal(out[0][k], out[1][k]) for k in out[0]) and np.all(np.isfinite(out[0]["f"]))
    print(bc, nx, ny, steps, "mask" if masked else "", "bitwise equal" if ok else "MISMATCH", flush=True)
case("pipe", 4096, 4096, 403, True, inlet_rho=1.0005)
case("velocity_inlet", 4096, 1000, 401, False, inlet_u=0.02)
case("cavity", 3000, 3000, 402, False, lid_u=0.05)
case("periodic", 8192, 8192, 203, False)
if "--more" in sys.argv:          # (round 5: the families of k_deep's hand-waited gather at its sizes)
    case("periodic", 8192, 8192, 150, True)
    case("pipe", 8192, 8192, 150, False, inlet_rho=1.0005)
    case("cavity", 6144, 6144, 150, True, lid_u=0.05)
    case("periodic", 2048, 2048, 500, False)
    # (round 6: the sizes the revised table gives to k_deep2<7> -- walled, 1700^2 ... 2900^2 cells --, k_deep<6> -- periodic from 1100^2 --
    #  and the reference's own case)
    case("pipe", 2048, 2048, 500, False, inlet_rho=1.0005)
    case("cavity", 2304, 2304, 430, True, lid_u=0.05)
    case("pipe", 3751, 1251, 500, True, inlet_rho=1.0005)
    case("periodic", 1280, 1280, 700, False)
    case("periodic", 1536, 1536, 600, True)
