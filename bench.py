#!/usr/bin/env python3
"""Headline benchmark: MLUPS + achieved HBM GB/s of the fused D2Q9 step on an 8192x8192 fp32
periodic shear layer (BASELINE.json: metric / configs[3]), 1..8 MI355X, row slabs + RCCL halo.

    python bench.py --gpus N --steps K --warmup W          # N > 1: spawns one rank process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config {2,3,4,5}                     # the other single-GPU configurations of BASELINE.json
                                                           # (1-based: 2 = 1024^2 lid-driven cavity Re 1000, 3 = 4096^2
                                                           # Kelvin-Helmholtz, 4 = the default, 5 = 4096^2 pipe + TIFF obstacles),
                                                           # one timed run per case like the reference's own comparison
                                                           # (docs/python_cython_opencl_comparison.ipynb:271-273)

A "step" = one pass of the hot path (stream + BC + moments + feq + BGK collide) over the whole grid.  The
grid is fixed as N grows (strong scaling, as the north-star states its 8-GPU target).  Rank 0 prints ONE JSON
line.

Timing.  After W warm-up steps the K-step block -- barrier + device sync, K steps, barrier + device sync;
every rank's time runs from the common start to its own device synchronisation behind the K steps, MAX over ranks (the
closing barrier's own latency, 20-50 us of collective, is reported beside, not counted as time steps) -- is repeated until at least MIN_BLOCKS blocks and MIN_TIMED_S seconds have been timed;
`ms_per_step` / `value` come from the MEDIAN block (`timing` lists min / max / count), so a 5 ms sample on a
fresh box no longer decides the line.

Extra objects in the line:
  roofline     - the dominant kernel against the HBM roofline.  `achieved` = the bytes one launch MUST move --
                 72 B x the cells of its slab (nine fp32 planes read once, nine written once; 73 B with an obstacle
                 mask), whatever number of time steps the launch fuses -- / the average duration of a launch inside the
                 timed K-step blocks (HIP events on the engine's stream around every block: what `value` is made of);
                 `frac` = achieved / 8 TB/s, <= 1.  `frac_plain_launch` prices a launch in the middle of a long run
                 instead (runs of 2q and of q launches, difference / q, right after the timed region): the two differ
                 by whatever the first / last launch of a run() costs extra.
                 `effective_GBps` = 72 B x lattice UPDATES / time (what an un-blocked kernel would have to move
                 for the same MLUPS; exceeds the peak when several steps share one pass) is reported beside it,
                 never as `frac`.  `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes of
                 this same command (profiles/pmc_traffic.json; see `traffic_source`), not measured in this run.
  other_configs - (N = 1, default configuration only) the other single-GPU configurations of BASELINE.json -- 2, 3, 5 --
                 timed by this same process after the headline's timed region, through the same code path as
                 `--config N` (about a second each: median of >= 5 K-step blocks): value (MLUPS), ms_per_step,
                 launch_ms, roofline_frac, kernel.  So that the driver's clock covers every configuration.
  methodology  - a tag naming how the timed region is bracketed and which launch `roofline.frac` prices; it changes
                 whenever a line stops being comparable with an older one (rounds 1-2: "r2").
  cpu_baseline - oracle port of the reference's Cython CPU path (oracle/d2q9_oracle.c o1_run, in the mode that is
                 pinned bit-exact to the imported reference: numpy2=True), 1 core, bounded samples at 256^2, 1024^2 and
                 4096^2 (`value` = the 4096^2 sample); reported, not a target.  Rank 0, N=1 only.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (four by default), and streams that share one run in submission
# order.  A slab handle needs three that do not (interior, edge bands, halo exchange); under torch.distributed -- torch's own and RCCL's
# streams come first -- the exchange has been seen on the interior's queue, ~45 us per halo cycle that nothing hides, and with a
# priority stream in the mix the edge bands on it (-35 %): profiles/r06_experiments.txt section 10c.  The HIP runtime reads the variable
# when it is loaded, i.e. at `import torch` below; the caller's own value wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

B_ALG = 72.0              # algorithmic bytes per lattice update: 9 fp32 read + 9 fp32 written
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
COPY_CEILING_GBS = 6290.0  # measured float4-copy ceiling quoted by the same guide
MIN_BLOCKS, MIN_TIMED_S, MAX_BLOCKS = 5, 0.5, 2000
# How the numbers of a line are made (bump when that changes, so that lines of different rounds are not compared blindly):
#   r3-block-avg: a rank's time runs from the common start (barrier + device sync) to its OWN device sync behind the K steps,
#   MAX over ranks (the closing barrier is bracketing: timing.ms_per_step_with_closing_barrier has it); median block of
#   >= 5 blocks and >= 0.5 s; rho, u, v are rebuilt on demand, so no launch of a block stores them (--eager-macro: round 2's
#   behaviour); roofline.frac prices the K-step block average launch (frac_plain_launch: a launch inside a long run).
#   Rounds 1-2 ("r2"): wall time included the closing barrier, the last launch of every block stored rho, u, v, and frac
#   priced the plain launch.
#   (The default block is 60 steps since the default kernel fuses five steps -- 48 until then, with four: a block should be whole
#   launches of the deepest kernel, 60 = lcm(1..5); K is on the line.  Not a change of method.)
METHODOLOGY = "r5-block-avg-plan-priced"


def shear_layer(nx, ny, y0, h, U=0.04, seed=0):
    """Double periodic shear layer (BASELINE config 4): rho=1, u = U tanh((y - ny/4)/d) below the
    mid-line and U tanh((3ny/4 - y)/d) above, d = ny/80, v = 1e-3 U sin(2 pi 4 x / nx).
    Returns the rows [y0, y0+h) as F-ordered (nx, h) float32 arrays."""
    y = np.arange(y0, y0 + h, dtype=np.float64)
    d = ny / 80.0
    prof = np.where(y < ny / 2.0, np.tanh((y - ny / 4.0) / d), np.tanh((3.0 * ny / 4.0 - y) / d))
    x = np.arange(nx, dtype=np.float64)
    u = np.asfortranarray(np.broadcast_to((U * prof)[None, :], (nx, h)).astype(np.float32))
    v = np.asfortranarray(np.broadcast_to((1e-3 * U * np.sin(2 * np.pi * 4 * x / nx))[:, None],
                                          (nx, h)).astype(np.float32))
    rho = np.ones((nx, h), np.float32, order="F")
    return rho, u, v


def cpu_baseline(budgets=((256, 3.0), (1024, 4.0), (4096, 10.0)), all_cores_budget_s=4.0):
    """Time the oracle's restatement of the reference Cython path (cython_dim.pyx:346-359, five un-fused passes, single
    thread) in its pinned mode (numpy2=True: bit-exact against the imported reference, tests/test_oracle_golden.py) on
    n x n pipe flows for about the given seconds each: 256^2 (BASELINE config 1), 1024^2, 4096^2."""
    from oracle import oracle as O
    ncores = len(os.sched_getaffinity(0))

    def sample(n, budget_s, openmp=False):
        kw = dict(diameter=1., rho=1., viscosity=0.05, pressure_grad=-1., pipe_length=1., N=n - 1,
                  time_prefactor=(n - 1) / 10.)
        sim = O.O1Sim.pipe_flow(numpy2=True, **kw)
        sim.run(1, openmp=openmp)                    # touch every page once
        steps, t0 = 0, time.perf_counter()
        while True:
            sim.run(1, openmp=openmp)
            steps += 1
            el = time.perf_counter() - t0
            if el >= budget_s or steps >= 20000:
                break
        return sim.nx * sim.ny * steps / el / 1e6, sim.nx, sim.ny, steps, el

    sizes = []
    for n, budget in budgets:
        mlups, nx, ny, steps, el = sample(n, budget)
        sizes.append({"grid": [nx, ny], "value": round(mlups, 3), "steps": steps, "seconds": round(el, 2)})
    big = sizes[-1]
    out = {"value": big["value"], "unit": "MLUPS", "cores": 1, "kind": "port",
           "sample": "oracle o1_run (C port of cython_dim.pyx Pipe_Flow.run, numpy2=True = the mode pinned bit-exact to the "
                     "imported reference), %dx%d grid, %d steps, %.1f s, 1 thread of %d available"
                     % (big["grid"][0], big["grid"][1], big["steps"], big["seconds"], ncores),
           "sizes": sizes}
    # the product's own CPU backend (lb_create with device = -1: 2d-lb_amd/csrc/cpu_backend.h, the same reference path restated
    # behind the C ABI, 1 thread) on the largest size -- BASELINE.md section 4's "build's own C++ restatement"
    try:
        from LB_D2Q9.dimensionless import cython_dim
        n = budgets[-1][0]
        own = cython_dim.Pipe_Flow(device=-1, verbose=False, diameter=1., rho=1., viscosity=0.05, pressure_grad=-1.,
                                   pipe_length=1., N=n - 1, time_prefactor=(n - 1) / 10.)
        own.run(1)
        steps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min(4.0, budgets[-1][1]):
            own.run(1)
            steps += 1
        el = time.perf_counter() - t0
        out["product_cpu_backend"] = {"value": round(own.nx * own.ny * steps / el / 1e6, 3), "unit": "MLUPS", "cores": 1,
                                      "kind": "own", "sample": "cython_dim.Pipe_Flow(device=-1) = lb_create(device = LB_DEVICE_CPU), "
                                                               "%dx%d, %d steps, %.1f s" % (own.nx, own.ny, steps, el)}
        own._sim.close()
    except Exception as exc:                             # noqa: BLE001 - the baseline leg must not take the line down
        out["product_cpu_backend"] = {"error": str(exc)}
    if all_cores_budget_s > 0:
        # "honest best CPU" line: the same port with OpenMP over the independent
        # cell loops (the in-place streaming only splits four ways), all host cores
        mlups, nx, ny, steps, el = sample(budgets[-1][0], all_cores_budget_s, openmp=True)
        out["all_cores"] = {"value": round(mlups, 3), "unit": "MLUPS", "cores": ncores,
                            "sample": "same port, -fopenmp, %dx%d, %d steps, %.1f s" % (nx, ny, steps, el)}
    return out


def load_pmc_traffic(key, plan, hot_kernel=""):
    """HBM bytes per launch, averaged over the launches of one timed block, from the committed rocprofv3 --pmc summaries of this
    workload (profiles/pmc_traffic.json, produced by tools/pmc_summary.py; key = grid side, or "c<config>/<side>" for the
    configurations other than the default; one entry per fused depth, "<key>/<steps per launch>").  `plan` = the depths of the
    block's launches (lb_plan_launches): a block of 20 steps is 6 + 7 + 7, i.e. one launch of the six-step and two of the
    seven-step kernel, each priced with its own counted bytes.  Returns (mean bytes per launch, source, per-depth bytes);
    bytes None with the reason in `source` when a depth of the plan has no committed profile."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
    except (OSError, ValueError) as exc:
        return None, "no traffic figure: %s unreadable (%s)" % (path, exc), None
    per_depth, srcs = {}, []
    for depth in sorted(set(plan)):
        # (seven steps per launch are k_deep<7>'s or k_deep2<7>'s: the profile of the kernel this line ran, if there is one)
        family = "k_deep2" if (depth == 7 and "k_deep2" in hot_kernel) else ("k_deep" if depth >= 6 else None)
        ent = (d.get("%s/%d:%s" % (key, depth, family)) if family else None) or d.get("%s/%d" % (key, depth))
        if not ent or not ent.get("hbm_bytes_per_launch"):
            return None, "no traffic figure: profiles/pmc_traffic.json holds no --pmc profile for %r (has: %s)" % (
                "%s/%d" % (key, depth), ", ".join(sorted(d))), None
        per_depth[depth] = ent["hbm_bytes_per_launch"]
        srcs.append(ent.get("source", "profiles/pmc_traffic.json"))
    mean = sum(per_depth[x] for x in plan) / float(len(plan))
    return mean, "committed profiles %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, read side calibrated on k_copy4), the " \
        "block's launches %s each priced with its own kernel's bytes; not measured in this run" % (
            ", ".join(sorted(set(srcs))), "+".join(str(x) for x in plan)), per_depth


def workload(config, n, omega, local_rank, eager_macro=False):
    """The single-GPU configurations of BASELINE.json (1-based as listed there): returns (simulation, description,
    bytes per cell and launch).  Initial states are built on the device from uploaded rho, u, v (f = feq)."""
    from LB_D2Q9.simulation import Simulation
    if config == 2:
        # 1024^2 lid-driven cavity, Re = U L / nu = 1000 with lid speed U = 0.1 (tests/test_gpu_fullsize.py: same case)
        U = 0.1
        nu = U * (n - 1) / 1000.0
        om = 1.0 / (3.0 * nu + 0.5) if omega is None else omega
        sim = Simulation(n, n, om, bc="cavity", lid_u=U, rho0=1.0, device=local_rank, eager_macro=eager_macro)
        z = np.zeros((n, n), np.float32, order="F")
        sim.init_equilibrium(np.ones((n, n), np.float32, order="F"), z, z)
        return sim, "%dx%d lid-driven cavity, Re=1000 (U=%.2g, omega=%.4f), D2Q9 BGK fp32" % (n, n, U, om), B_ALG
    if config == 3:
        om = 1.8 if omega is None else omega
        sim = Simulation(n, n, om, bc="periodic", device=local_rank, eager_macro=eager_macro)
        sim.init_equilibrium(*shear_layer(n, n, 0, n, U=0.05))
        return sim, "%dx%d Kelvin-Helmholtz double vortex sheet (periodic, U=0.05, omega=%g), D2Q9 BGK fp32" % (n, n, om), B_ALG
    if config == 5:
        from LB_D2Q9.masks import obstacle_mask_from_tiff
        om = 1.0 if omega is None else omega
        mask = np.array(obstacle_mask_from_tiff(os.path.join(ROOT, "tests", "golden", "CS205_obstacle_4.tif"), (n, n)), dtype=bool)
        mask[0, :] = mask[-1, :] = False
        mask[:, 0] = mask[:, -1] = False
        sim = Simulation(n, n, om, bc="pipe", inlet_rho=1.001, outlet_rho=1.0, obstacle_mask=mask, device=local_rank,
                         eager_macro=eager_macro)
        z = np.zeros((n, n), np.float32, order="F")
        sim.init_equilibrium(np.ones((n, n), np.float32, order="F"), z, z)
        sim.zero_velocity_in_obstacle()
        return sim, ("%dx%d porous-media obstacle flow: pressure-driven pipe, bounce-back mask from docs/CS205_obstacle_4.tif "
                     "(%.2f %% solid), omega=%g, D2Q9 BGK fp32" % (n, n, 100.0 * mask.mean(), om)), B_ALG + 1.0
    raise SystemExit("--config must be 2, 3, 4 or 5 (config 1 is the CPU plumbing case: tests/test_config1.py)")


def measure_config(config, local_rank, steps, warmup, min_blocks, min_timed_s, size=None):
    """One single-GPU configuration through the code path of `bench.py --config N`: build the workload on the device,
    tune, warm up, time K-step blocks (device sync on both sides; HIP events on the engine's stream around each), median
    block.  Returns the summary dict that goes into `other_configs`."""
    import torch
    n = size or {2: 1024, 3: 4096, 5: 4096}[config]
    sim, what, bytes_per_cell = workload(config, n, None, local_rank)
    try:
        sim.autotune()
        sim.run(warmup, wait=False)
        walls, evs, total = [], [], 0.0
        while len(walls) < MAX_BLOCKS and (len(walls) < min_blocks or total < min_timed_s):
            sim.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev_ms = sim.timed_run(steps)
            sim.sync()
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
            evs.append(ev_ms)
            total += walls[-1]
        wall, ev_ms = statistics.median(walls), statistics.median(evs)
        health = sim.check()
        mean_rho = health["sum_rho"] / (float(n) * n)
        if health["n_nonfinite"] or abs(mean_rho - 1.0) > 1e-3 or not health["max_mach"] < 0.3:
            raise SystemExit("bench: non-physical state after config %d (%r)" % (config, health))
        spl = sim.steps_per_launch()
        plan = sim.plan_launches(steps) if hasattr(sim, "plan_launches") else None
        launch_s = ev_ms / 1e3 / (len(plan) if plan else steps / float(spl))
        achieved = bytes_per_cell * n * n / launch_s / 1e9
        return {"config": config, "workload": what, "value": round(n * float(n) * steps / wall / 1e6, 1), "unit": "MLUPS",
                "ms_per_step": round(wall * 1e3 / steps, 6), "steps": steps, "warmup": warmup, "blocks": len(walls),
                "timed_s": round(total, 3), "launch_ms": round(launch_s * 1e3, 6), "steps_per_launch": spl, "block_plan": plan,
                "roofline_frac": round(achieved / HBM_PEAK_GBS, 4), "achieved_GBps": round(achieved, 1),
                "bytes_per_cell_per_launch": bytes_per_cell, "kernel": sim.hot_kernel(), "health": health}
    finally:
        sim.close()


def reference_case(path, local_rank):
    """The reference's ONE published benchmark on this engine, on the driver's clock: Pipe_Flow_Cylinder(diameter=1, rho=1,
    viscosity=1, pressure_grad=-10, pipe_length=3, N=125, cylinder_center=[.75, .5], cylinder_radius=.1) = 3751 x 1251 cells,
    1000 steps timed by the wall clock around run() and MLUPS = nx ny steps / t / 1e6, exactly as the notebook does
    (docs/python_cython_opencl_comparison.ipynb:136, 233, 271-273: 317.5 MLUPS on a GTX Titan Black; Cython class :404-406:
    5.9 MLUPS, 20 steps) -- through the drop-in classes: path = "opencl" (hip_dim: the fused kernels) or "cython"
    (cython_dim: the reference's CPU semantics on the GPU)."""
    from LB_D2Q9.dimensionless import cython_dim, hip_dim
    mod = hip_dim if path == "opencl" else cython_dim
    sim = mod.Pipe_Flow_Cylinder(diameter=1., rho=1., viscosity=1., pressure_grad=-10., pipe_length=3., N=125,
                                 cylinder_center=[.75, .5], cylinder_radius=.1, verbose=False)
    eng = getattr(sim, "_sim", None)
    try:
        if eng is not None and hasattr(eng, "autotune"):
            eng.autotune()
        steps = 1000
        sim.run(140)                                 # warm-up (untimed)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            sim.run(steps)                           # returns with the work complete, like the reference's run()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        nx, ny = int(sim.nx), int(sim.ny)
        mlups = nx * ny * steps / best / 1e6
        spl = eng.steps_per_launch() if eng is not None else None
        plan = eng.plan_launches(steps) if eng is not None and hasattr(eng, "plan_launches") else None
        launches = len(plan) if plan else (steps / float(spl) if spl else None)
        out = {"config": "reference_case", "path": path,
               "workload": "the reference's published benchmark: Pipe_Flow_Cylinder N=125, %d x %d cells, %d steps, wall clock around "
                           "run() (docs/python_cython_opencl_comparison.ipynb:136, 233, 271-273), %s-path drop-in class" % (nx, ny, steps, path),
               "value": round(mlups, 1), "unit": "MLUPS", "seconds": round(best, 4), "steps": steps,
               "reference_published_MLUPS": 317.52 if path == "opencl" else 5.911,
               "steps_per_launch": spl, "kernel": eng.hot_kernel() if eng is not None else None}
        if launches:
            out["launch_ms"] = round(best * 1e3 / launches, 6)
            out["roofline_frac"] = round((B_ALG + 1.0) * nx * ny / (best / launches) / 1e9 / HBM_PEAK_GBS, 4)
        return out
    finally:
        if eng is not None:
            eng.close()


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start one rank process per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* as torch.distributed.run would set them), wait, forward rank 0's line.  This parent
    never touches HIP or torch.cuda (a process that initialised the GPU must not start replacing itself, and
    has no need to: the children are ordinary subprocesses)."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    # rank 0's stdout goes through a file, not a pipe: nobody reads while the ranks run, and a chatty runtime
    # (NCCL_DEBUG=INFO ...) must not be able to fill a pipe and stall the rank
    out0 = tempfile.TemporaryFile(mode="w+b")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("NCCL_SOCKET_IFNAME", "lo")       # one node: bootstrap over the loopback interface
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves its peers blocked in a collective: poll, and take the others down with it
    failed = None
    while failed is None:
        codes = [p.poll() for p in procs]
        failed = next((r for r, c in enumerate(codes) if c not in (None, 0)), None)
        if failed is None and all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:                       # (exact PIDs of the children started above)
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    out0.seek(0)
    out = out0.read().decode(errors="replace")
    out0.close()
    if failed is not None:
        sys.stderr.write(out)
        raise SystemExit("bench: rank %d exited with status %s" % (failed, procs[failed].returncode))
    # exactly ONE line on stdout: the result; whatever else rank 0 wrote there (a runtime's banner) goes to stderr
    lines = [l for l in out.splitlines() if l.startswith("{")]
    sys.stderr.write("".join(l + "\n" for l in out.splitlines() if not l.startswith("{")))
    if not lines:
        raise SystemExit("bench: rank 0 printed no result line")
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=84)            # (84 = whole launches of every depth up to seven)
    ap.add_argument("--warmup", type=int, default=14)
    ap.add_argument("--config", type=int, default=4, choices=[2, 3, 4, 5],
                    help="BASELINE.json configuration, 1-based: 4 (default) = 8192^2 periodic shear layer, the one the metric "
                         "is quoted on; 2 = 1024^2 lid-driven cavity Re=1000; 3 = 4096^2 Kelvin-Helmholtz; 5 = 4096^2 pipe "
                         "flow through the porous-medium image.  2, 3, 5 are single-GPU cases")
    ap.add_argument("--size", type=int, default=None, help="grid side (default: the configuration's own: 1024 / 4096 / 8192 / 4096)")
    ap.add_argument("--omega", type=float, default=None, help="BGK relaxation rate (default: the configuration's own; 1.7 for config 4)")
    ap.add_argument("--transport", default=os.environ.get("LB_HALO_TRANSPORT", "rccl"), choices=["rccl", "peer", "torch"],
                    help="halo transport of the multi-rank run: rccl (default), peer (direct stores into the neighbours' ghost rows "
                         "through IPC-mapped memory), torch (python-driven); one that cannot be set up falls back to the next")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="do not time BASELINE configurations 2, 3 and 5 behind the headline (N = 1, default configuration)")
    ap.add_argument("--variant", type=int, default=None, help="kernel variant (tuning)")
    ap.add_argument("--eager-macro", action="store_true",
                    help="the last launch of every run() stores rho, u, v (LB_FLAG_EAGER_MACRO; round-2 behaviour) instead of "
                         "leaving them to be rebuilt from the populations on demand")
    ap.add_argument("--force-slab-path", action="store_true",
                    help="run through DistributedSlab / the RCCL halo path even with one rank (a 1-rank periodic "
                         "ring exchanging with itself): exercises the multi-GPU code on a single GPU")
    ap.add_argument("--slab-rows", type=int, default=0,
                    help="diagnostic, with --force-slab-path: the lattice is SIZE columns x this many rows -- the slab one of N ranks "
                         "of a strong-scaled SIZE^2 lattice would hold, in this process's structure (torch.distributed, RCCL's streams)")
    ap.add_argument("--calibrate", type=int, default=3,
                    help="launch N plain float4 copies of known size before the timed region (the device's own "
                         "streaming rate, `copy_GBps`; also the FETCH_SIZE calibration of rocprofv3 --pmc runs); 0 = skip")
    ap.add_argument("--min-timed-s", type=float, default=MIN_TIMED_S)
    ap.add_argument("--min-blocks", type=int, default=MIN_BLOCKS)
    args = ap.parse_args()
    if args.steps < 1:
        raise SystemExit("--steps must be >= 1")
    if args.size is None:
        args.size = {2: 1024, 3: 4096, 4: 8192, 5: 4096}[args.config]
    if args.config != 4 and (args.gpus > 1 or args.force_slab_path):
        raise SystemExit("--config %d is a single-GPU case" % args.config)
    if args.config == 4 and args.omega is None:
        args.omega = 1.7

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])
        return

    # stdout carries ONE line, the result.  RCCL prints a version banner to the C-level stdout of every process that
    # creates a communicator (it surfaces at exit, behind the result): file descriptor 1 is pointed at stderr for
    # everything else, the line itself goes to a duplicate of the original stdout.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_slab_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1"):
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")   # rendezvous on this node: bootstrap over the loopback interface
                                                                # (a multi-node launch names a routable MASTER_ADDR and keeps RCCL's choice)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from LB_D2Q9.simulation import Simulation
    from LB_D2Q9.slabs import DistributedSlab

    n = args.size
    ny_all = args.slab_rows if (args.slab_rows and args.force_slab_path and args.config == 4) else n     # rows of the whole lattice
    bytes_per_cell, what = B_ALG, None
    if args.config != 4:
        sim, what, bytes_per_cell = workload(args.config, n, args.omega, local_rank, eager_macro=args.eager_macro)
        eng, y0, h = sim, 0, n
    elif world == 1 and not args.force_slab_path:
        sim = Simulation(n, n, args.omega, bc="periodic", device=local_rank, eager_macro=args.eager_macro)
        eng, y0, h = sim, 0, n
    else:
        # RCCL halo exchange inside the engine; if any rank cannot set it up, every rank falls back -- first to the peer
        # transport (the same schedule inside lb_run, halo rows stored straight into the neighbours' ghost rows through
        # IPC-mapped device memory), then to the torch.distributed-driven exchange (same halo format, single-step kernel,
        # not overlapped)
        chain = {"rccl": ["rccl", "peer", "torch"], "peer": ["peer", "torch"], "torch": ["torch"]}[args.transport]
        slab = None
        for transport in chain:
            failed = 0
            try:
                slab = DistributedSlab(n, ny_all, args.omega, bc="periodic", transport=transport, device=local_rank,
                                       eager_macro=args.eager_macro)
            except Exception as exc:                                   # noqa: BLE001 - reported below
                failed = 1
                print("rank %d: %s halo transport unavailable (%s)" % (rank, transport, exc), file=sys.stderr, flush=True)
            flag = torch.tensor([failed], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if not int(flag[0]):
                args.transport = transport
                break
            if slab is not None:
                slab.engine.close()
                slab = None
        if slab is None:
            raise SystemExit("bench: no usable halo transport")
        sim, eng, y0, h = slab, slab.engine, slab.y0, slab.h
    if args.variant is not None:
        eng.set_variant(args.variant)
    if args.config == 4:
        rho, u, v = shear_layer(n, ny_all, y0, h)
        eng.init_equilibrium(rho, u, v)        # feq and f = feq are built on the device
        del rho, u, v

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        eng.sync()
        torch.cuda.synchronize()

    slab_tuning = None
    if dist is None:
        eng.autotune()                         # picks the fused-kernel configuration for this grid (untimed)
    elif args.config == 4 and args.variant is None and hasattr(sim, "autotune"):
        # slabs: the ranks time the halo cycle on each depth of the fused kernel together and agree (untimed, live steps)
        # (depths 7, 6, 5 x the exchange beside / between the interior launches; twenty cycles per candidate: what RCCL's kernel costs
        #  the launches it runs beside shows in a steady state only -- six cycles had "beside" 7 % ahead where 140-step runs have it 3-6 %
        #  behind: profiles/r06i_bench_placement.txt, r06i_slab_proxy_placement.txt)
        slab_tuning = sim.autotune()               # (cycles=20, rounds=2: the defaults)
    if dist is not None and hasattr(eng, "exchange_timing") and args.transport in ("rccl", "peer"):
        eng.exchange_timing(True)
    copy_gbs = None
    if args.calibrate:
        copy_gbs = {"plain": round(eng.copy_calibration(args.calibrate, False)[0], 1),
                    "nontemporal": round(eng.copy_calibration(args.calibrate, True)[0], 1)}
    sim.run(args.warmup, wait=False)

    # ---- timed region: blocks of exactly K steps, each bracketed by barrier + device sync -------------------
    walls, evs, walls_incl, total = [], [], [], 0.0
    while len(walls) < MAX_BLOCKS and (len(walls) < args.min_blocks or total < args.min_timed_s):
        barrier()
        t0 = time.perf_counter()
        ev_ms = sim.timed_run(args.steps)      # enqueue K steps between two HIP events on the engine's stream, wait for the last
        eng.sync()
        torch.cuda.synchronize()               # the device has finished this rank's K steps ...
        wall = time.perf_counter() - t0        # ... = this rank's time, from the common start to its own synchronised completion
        barrier()                              # closing bracket; the collective itself is not a time step (MAX over ranks below says
        wall_incl = time.perf_counter() - t0   #   when the slowest rank was done); the time with it: timing.ms_per_step_with_closing_barrier
        if dist is not None:                   # MAX over ranks; also makes every rank take the same loop decision
            t = torch.tensor([wall, ev_ms, wall_incl], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, ev_ms, wall_incl = float(t[0]), float(t[1]), float(t[2])
        walls.append(wall)
        evs.append(ev_ms)
        walls_incl.append(wall_incl)
        total += wall
    wall, ev_ms = statistics.median(walls), statistics.median(evs)
    # per rank: what the halo exchanges of the timed blocks took on their stream (lb_exchange_stats), gathered on rank 0
    per_rank = None
    if dist is not None and hasattr(eng, "exchange_stats") and args.transport in ("rccl", "peer"):
        st = eng.exchange_stats()
        mine = torch.tensor([float(rank), float(h), float(st["n"]), st["total_ms"], st["max_ms"], float(st["cycle_depth"]),
                             float(st["band_rows"]), statistics.median(evs)], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": int(t[0]), "rows": int(t[1]), "exchanges_timed": int(t[2]),
                     "exchange_ms_mean": round(float(t[3]) / max(1.0, float(t[2])), 4), "exchange_ms_max": round(float(t[4]), 4),
                     "cycle_depth": int(t[5]), "edge_band_rows": int(t[6])} for t in every]

    # ---- duration of ONE launch of the dominant kernel, for the roofline (outside the timed region) -----------
    # A run() ends with the launch that also stores rho, u, v (12 B per cell more, ~11 % longer at 8192^2), so the K-step
    # block above averages two kinds of launches.  The plain launch's duration is the difference between runs of 2q and
    # of q launches, divided by q (HIP events on the engine's stream, median of 5 each) -- the figure rocprofv3 reports
    # as the kernel's average (profiles/r02_rocprof_summary.md).
    plain_ms = macro_extra_ms = None
    if dist is None:
        # (whole launches of the deepest kernel, at least four per run: with --steps 20 the 2 x 7 against 4 x 7 steps of round 5
        #  differed by less than their noise and the probe rejected itself -- frac_plain_launch: null on the driver's line)
        spl_probe = eng.steps_per_launch()
        q = max(4, -(-args.steps // spl_probe))
        t1 = statistics.median(sim.timed_run(spl_probe * q) for _ in range(7))
        t2 = statistics.median(sim.timed_run(2 * spl_probe * q) for _ in range(7))
        a = (t2 - t1) / q
        if 0.5 * t1 / q < a <= 1.1 * t1 / q:           # (else: keep the block average below)
            plain_ms, macro_extra_ms = a, max(t1 - q * a, 0.0)

    # the six-step kernel beside the default (seven steps per launch since round 5): the launch the previous rounds' `frac` priced
    six = None
    if dist is None and args.variant is None and args.config == 4 and eng.steps_per_launch() == 7:
        eng.set_variant(353 | 4096 | 16384)            # k_deep<6> (bitwise equal: these are ordinary time steps)
        if eng.steps_per_launch() == 6:
            sim.run(12, wait=False)
            t6 = statistics.median(sim.timed_run(6 * 10) for _ in range(5)) / 10.0
            six = {"kernel": eng.hot_kernel(), "launch_ms": round(t6, 4),
                   "frac": round(bytes_per_cell * n * h / (t6 / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                   "MLUPS": round(6.0 * n * h / (t6 / 1e3) / 1e6, 1)}
        eng.set_variant(-1)

    # sanity: the run must have produced finite, physical numbers (guards against timing a broken kernel) -- one
    # device pass and 24 bytes to the host (lb_check) instead of downloading a 268 MB plane
    health = sim.check() if dist is not None else eng.check()
    mean_rho = health["sum_rho"] / (float(n) * ny_all)
    if health["n_nonfinite"] or abs(mean_rho - 1.0) > 1e-3 or not health["max_mach"] < 0.3:
        raise SystemExit("bench: non-physical state after the run (%r)" % (health,))

    if rank == 0:
        cells = float(n) * ny_all
        mlups = cells * args.steps / wall / 1e6
        # dominant kernel = the fused step over this rank's rows; rank 0's slab is representative.
        # One launch advances spl time steps: it performs spl x n x h lattice updates, but what it MUST move is
        # one read and one write of the slab's nine planes, 72 B x n x h, whatever spl is.
        python_driven = dist is not None and args.transport == "torch"      # that path is single-step
        spl = 1 if python_driven else eng.steps_per_launch()
        kname = "k_step (python-driven exchange)" if python_driven else eng.hot_kernel()
        # A block of K timed steps is a handful of launches; every launch of a marching kernel moves the same algorithmic bytes
        # whatever number of steps it fuses, so a launch is priced at block time / launches of the block.  The engine tells how
        # it splits K (lb_plan_launches; K = 20 with depths up to six: 4 + 4 + 6 + 6); where it cannot (slabs): K / spl launches.
        plan = None if python_driven or not hasattr(eng, "plan_launches") else eng.plan_launches(args.steps)
        n_launches = float(len(plan)) if plan else args.steps / float(spl)
        launch_s = ev_ms / 1e3 / n_launches
        launch_source = ("K-step block average: HIP events on the engine's stream around each timed block of %d steps / its "
                         "%g launches%s (median block)" % (args.steps, n_launches,
                                                          " of %s steps" % "+".join(str(d) for d in plan) if plan else ""))
        bytes_per_launch = bytes_per_cell * n * h
        achieved = bytes_per_launch / launch_s / 1e9
        effective = B_ALG * n * h * args.steps / (ev_ms / 1e3) / 1e9
        traffic, traffic_source, traffic_by_depth = load_pmc_traffic(n if args.config == 4 else "c%d/%d" % (args.config, n), plan or [spl], eng.hot_kernel()) \
            if dist is None else (None, "no traffic figure: counters are collected on one GPU", None)
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "frac_plain_launch": None if plain_ms is None or python_driven else round(bytes_per_launch / (plain_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                "plain_launch_ms": None if plain_ms is None or python_driven else round(plain_ms, 4),
                "plain_launch_source": "difference of runs of 2q and q launches / q, outside the timed region (a launch in the "
                                       "middle of a long run; first/last-launch extras of a run: %+.4f ms)" % (macro_extra_ms or 0.0),
                "frac_of_measured_copy": round(achieved / COPY_CEILING_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_source, "traffic_by_steps_per_launch": traffic_by_depth,
                "traffic_frac": None if traffic is None else round(traffic / launch_s / 1e9 / HBM_PEAK_GBS, 4),
                "kernel": "%s, %d x %d cells x %d step(s) per launch%s" % (
                    kname, n, h, spl, "" if not plan or set(plan) == {spl} else "; the block's launches: %s steps (the shallower ones: the "
                    "same march with fewer stages)" % "+".join(str(d) for d in plan)),
                "launch_ms": round(launch_s * 1e3, 4), "launch_ms_source": launch_source, "steps_per_launch": spl,
                "block_plan": plan,
                "six_step_kernel": six,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "effective_GBps": round(effective, 1), "effective_x_roofline": round(effective / HBM_PEAK_GBS, 4),
                "macro_fields": "rebuilt on demand from the populations (lb_get_macro / lb_check), not stored by run()"
                                if not getattr(eng, "eager_macro", False) else "stored by the last launch of every run()",
                "note": "achieved = %g B x cells of one launch (compulsory: each plane read once, written once%s) / "
                        "launch time; effective_GBps = 72 B x lattice updates / time is NOT an HBM rate when "
                        "steps_per_launch > 1 (effective_x_roofline = value against the single-pass roofline 8 TB/s / 72 B per "
                        "update); frac prices ONE launch, whatever number of time steps it fuses: seven since round 5 (k_deep<7>: the launch "
                        "is bound by instruction issue, not by HBM -- `compute`), six / five in round 4, four before (k_step4: launch "
                        "0.85 ms, frac 0.71, 314 k MLUPS)"
                        % (bytes_per_cell, "" if bytes_per_cell == B_ALG else ", + 1 B obstacle mask")}
        # what the kernel is actually bound by since round 5: vector-ALU issue (profiles/r05_experiments.txt).  68 fp32 operations per
        # cell update as written (d2q9_cell.h: 23 of them FMAs) = 91 flop; the packed pipe's peak is the guide's fp32 vector figure.
        # (round 6: omega enters once, through the density -- 60 operations per update instead of 68, d2q9_cell.h: equilibrate_t)
        flop_per_update, fp32_peak = 83.0, 157.3e12
        compute = {"bound": "valu", "achieved": round(mlups * 1e6 * flop_per_update / 1e12, 2), "peak": fp32_peak / 1e12, "unit": "TFLOP/s",
                   "frac": round(mlups * 1e6 * flop_per_update / fp32_peak, 4),
                   "note": "informational: %g flop per lattice update (60 fp32 operations, 23 of them FMAs, all issued as v_pk_*_f32) "
                           "against the fp32 vector peak; the collision is %d of the %d instructions a wave of k_deep<7> issues per row "
                           "(tools/isa_stats.py)" % (flop_per_update, 839, 1342)}
        line = {
            "metric": "MLUPS (million lattice updates per second), fused D2Q9 BGK step",
            "value": round(mlups, 1), "unit": "MLUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3 / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "achieved_hbm_GBps": round(world * achieved, 1),
            "timing": {"blocks": len(walls), "block_steps": args.steps, "statistic": "median",
                       "min_ms_per_step": round(min(walls) * 1e3 / args.steps, 4),
                       "max_ms_per_step": round(max(walls) * 1e3 / args.steps, 4),
                       "ms_per_step_with_closing_barrier": round(statistics.median(walls_incl) * 1e3 / args.steps, 4),
                       "timed_s": round(total, 3)},
            "config": {"workload": what or "%dx%d periodic double shear layer, D2Q9 BGK fp32, omega=%g, "
                                           "%d row slab(s) of %d rows%s" % (n, ny_all, args.omega, world, h,
                                                                            "" if dist is None else ", halo via " + args.transport),
                       "baseline_config": args.config, "grid": [n, ny_all], "bytes_per_lattice_update": B_ALG},
            "health": health,
            "roofline": roof,
            "compute": compute,
        }
        if per_rank is not None:
            line["slabs"] = {"per_rank": per_rank, "cycle_tuning": slab_tuning,
                             "note": "exchange_ms = one halo exchange on its stream (pack / push, transfer, wait for the neighbours, unpack; "
                                     "lb_exchange_stats); the edge bands are cut so that their waves finish 8 iterations (~2 x 25 us per "
                                     "cycle) before the interior's: an exchange longer than that delays the compute stream"}
        line["methodology"] = METHODOLOGY
        if copy_gbs is not None:
            line["copy_GBps"] = copy_gbs
            roof["frac_of_copy_on_this_device"] = round(achieved / max(copy_gbs.values()), 4)
        if world == 1 and dist is None and args.config == 4 and not args.no_other_configs:
            # the other single-GPU configurations, on the driver's clock too: after the headline's timed region, the
            # headline's lattice released first (same code path as `--config N`; about a second each)
            eng.close()
            line["other_configs"] = []
            for c in (2, 3, 5):
                try:
                    line["other_configs"].append(measure_config(c, local_rank, 84, 14, args.min_blocks, args.min_timed_s))    # (84 = whole launches of depths 7, 6, 4, 3, 2, 1; k_step5 blocks end in a four-step launch)
                except (Exception, SystemExit) as exc:             # noqa: BLE001 - a side line must not take the headline down
                    line["other_configs"].append({"config": c, "error": str(exc)})
            for path in ("opencl", "cython"):
                try:
                    line["other_configs"].append(reference_case(path, local_rank))
                except (Exception, SystemExit) as exc:             # noqa: BLE001
                    line["other_configs"].append({"config": "reference_case", "path": path, "error": str(exc)})
        if world == 1 and not args.no_cpu_baseline and args.config == 4:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), file=result_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
