"""Row-slab decomposition of the lattice across GPUs (new work: the reference is single-device).

The grid is cut along y (the slow axis of the device layout) into contiguous row slabs, one per
GPU / process.  Streaming reaches one cell, so per step each slab needs from its neighbours only
the populations that cross the shared edge (k=2,5,6 travel north, k=4,7,8 travel south).  The engine
keeps the ghost zone fourteen rows deep: inside ``lb_run`` the seven-step kernel (k_deep) runs twice per exchange on large
slabs (117 row segments of nx floats per direction; smaller slabs and explicit variants: six- / five- / four- / three-step kernels,
12 / 10 / 8 / 6 rows, 99 / 81 / 63 / 45 segments); the halo this module moves itself (lb_halo_export /
import) is the 3-deep one (18 row segments, include/lb_hip.h).  There is no collective on the data path.

* ``partition_rows`` / ``neighbours``  - the arithmetic.
* ``LocalSlabRing``     - G slabs on ONE device, halos copied with lb_halo_export/import.  Used to
                          prove that a partitioned run equals the single-slab run bit for bit.
* ``DistributedSlab``   - one process per GPU under ``torch.distributed``.  transport='rccl': the
                          engine exchanges halos itself inside ``lb_run`` (RCCL send/recv on a side
                          HIP stream, overlapped with the interior rows).  transport='peer': the same
                          schedule, the halo rows stored straight into the neighbours' ghost rows through
                          device memory mapped across the rank processes (lb_peer_connect; works between
                          processes that share ONE GPU, which RCCL refuses).  transport='torch': the
                          exchange is driven from here with ``torch.distributed`` point-to-point ops
                          (any backend; also the path the CPU/gloo tests exercise with a stand-in
                          engine injected through ``engine_factory``).
"""
import numpy as np

from . import _native

SOUTH, NORTH = 0, 1
MASK_HALO_ROWS = _native.LB_MASK_HALO_ROWS


def partition_rows(ny, nparts):
    """Balanced contiguous split of ny rows: list of (y0, height); the first ny % nparts slabs get
    one extra row."""
    if nparts < 1 or nparts > ny:
        raise ValueError("cannot cut %d rows into %d slabs" % (ny, nparts))
    base, extra = divmod(ny, nparts)
    out, y0 = [], 0
    for r in range(nparts):
        h = base + (1 if r < extra else 0)
        out.append((y0, h))
        y0 += h
    return out


def neighbours(rank, nranks, periodic):
    """(south, north) ranks of slab `rank`; -1 where the slab touches a wall."""
    south = rank - 1 if rank > 0 else (nranks - 1 if periodic else -1)
    north = rank + 1 if rank < nranks - 1 else (0 if periodic else -1)
    return south, north


def _is_periodic(bc):
    return bc in ("periodic", _native.LB_BC_PERIODIC) and not isinstance(bc, bool)


def _bc_code(bc):
    """Boundary family as its lb_bc_mode number, whether it was given by name or by number."""
    return _native.BC_NAMES[bc] if isinstance(bc, str) else int(bc)


def _default_engine(**kw):
    from .simulation import Simulation
    return Simulation(**kw)


class _SlabSet(object):
    """Shared scatter/gather helpers: global F-ordered (nx, ny[, 9]) arrays <-> per-slab pieces."""

    def _cut(self, a, y0, h):
        return np.asfortranarray(np.asarray(a)[:, y0:y0 + h])

    @staticmethod
    def _mask_halo_rows(mask, y0, h, ny, periodic):
        """The MASK_HALO_ROWS mask rows below / above the slab [y0, y0+h), each (rows, nx) (None at a wall;
        rows that fall outside a non-periodic box are empty)."""
        m = np.asarray(mask) != 0
        d = MASK_HALO_ROWS

        def rows(ys):
            out = np.zeros((d, m.shape[0]), bool)
            for i, y in enumerate(ys):
                if periodic:
                    out[i] = m[:, y % ny]
                elif 0 <= y < ny:
                    out[i] = m[:, y]
            return out
        south = rows(range(y0 - d, y0)) if (periodic or y0 > 0) else None
        north = rows(range(y0 + h, y0 + h + d)) if (periodic or y0 + h < ny) else None
        return south, north


class LocalSlabRing(_SlabSet):
    """G virtual slabs on one device.  Not a performance path: it exists so that
    'partitioned == unpartitioned, bit for bit' can be tested on a single GPU."""

    def __init__(self, nx, ny, omega, nslabs, bc="pipe", obstacle_mask=None, device=0, **kw):
        self.nx, self.ny, self.bc = nx, ny, bc
        self.parts = partition_rows(ny, nslabs)
        self.periodic = _is_periodic(bc)
        self.slabs = []
        for (y0, h) in self.parts:
            m = None if obstacle_mask is None else self._cut(obstacle_mask, y0, h)
            eng = _default_engine(nx=nx, ny=ny, omega=omega, bc=bc, obstacle_mask=m,
                                  device=device, y0=y0, local_ny=h, halo=True, **kw)
            if obstacle_mask is not None:
                eng.set_obstacle_mask_halo(*self._mask_halo_rows(obstacle_mask, y0, h, ny, self.periodic))
            self.slabs.append(eng)
        self._buf = np.zeros((len(self.slabs), 2, 18 * nx), np.float32)
        self._ghosts_valid = False

    def set_f(self, f):
        for s, (y0, h) in zip(self.slabs, self.parts):
            s.set_f(self._cut(f, y0, h))
        self._ghosts_valid = False

    def _exchange(self):
        n = len(self.slabs)
        for r, s in enumerate(self.slabs):
            s.halo_export(SOUTH, self._buf[r, SOUTH])
            s.halo_export(NORTH, self._buf[r, NORTH])
        for s in self.slabs:
            s.sync()
        for r, s in enumerate(self.slabs):
            south, north = neighbours(r, n, self.periodic)
            if south >= 0:
                s.halo_import(SOUTH, self._buf[south, NORTH])   # what the southern slab sent north
            if north >= 0:
                s.halo_import(NORTH, self._buf[north, SOUTH])
        for s in self.slabs:
            s.sync()

    def run_in_library(self, n):
        """The same slabs advanced by lb_run_group: the multi-GPU schedule (edge bands first, two-step
        kernel where applicable, halo packed / unpacked on a side stream) with the neighbour's buffer read directly."""
        from .simulation import run_group
        run_group(self.slabs, n)
        self._ghosts_valid = False      # lb_run_group refreshes the ghosts itself

    def set_variant(self, variant):
        for s in self.slabs:
            s.set_variant(variant)

    def run(self, n):
        if not self._ghosts_valid:
            self._exchange()
        for it in range(n):
            macro = (it == n - 1)
            for s in self.slabs:
                s.step_boundary(macro)
                s.step_interior(macro)
            self._exchange()            # halos of the lattice just written
            for s in self.slabs:
                s.step_finish()
        self._ghosts_valid = True

    def get_fields(self, which=("f", "feq", "u", "v", "rho")):
        pieces = [s.get_fields(which) for s in self.slabs]
        return {k: np.asfortranarray(np.concatenate([p[k] for p in pieces], axis=1)) for k in which}


class DistributedSlab(_SlabSet):
    """This process's slab of a multi-GPU lattice.  Requires an initialised
    ``torch.distributed`` process group (backend 'nccl' == RCCL on ROCm, or 'gloo')."""

    def __init__(self, nx, ny, omega, bc="pipe", obstacle_mask=None, transport="rccl", device=None,
                 engine_factory=None, group=None, **kw):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)
        self.nx, self.ny, self.bc = nx, ny, bc
        self.periodic = _is_periodic(bc)
        self.parts = partition_rows(ny, self.nranks)
        self.y0, self.h = self.parts[self.rank]
        self.south, self.north = neighbours(self.rank, self.nranks, self.periodic)
        self.transport = transport
        make = engine_factory or _default_engine
        m = None if obstacle_mask is None else self._cut(obstacle_mask, self.y0, self.h)
        self._mask_local = m
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", "0")) if engine_factory is None else 0
        self.omega, self._engine_kw = omega, dict(kw)
        self._has_mask = obstacle_mask is not None
        self.engine = make(nx=nx, ny=ny, omega=omega, bc=bc, obstacle_mask=m, device=device,
                           y0=self.y0, local_ny=self.h, halo=True, **kw)
        if obstacle_mask is not None and hasattr(self.engine, "set_obstacle_mask_halo"):
            self.engine.set_obstacle_mask_halo(*self._mask_halo_rows(obstacle_mask, self.y0, self.h, ny, self.periodic))
        self._ghosts_valid = False
        self._bufs = None
        if transport == "rccl":
            self._attach_rccl()
        elif transport == "peer":
            self._attach_peer()
        elif transport != "torch":
            raise ValueError("transport must be 'rccl', 'peer' or 'torch'")

    # -- RCCL inside the engine ------------------------------------------------------------------
    def _attach_rccl(self):
        """Collective.  A rank that cannot set RCCL up must not leave its peers blocked in a broadcast or inside
        ncclCommInitRank, so every stage ends in an agreement over torch.distributed before the next one starts:
        (1) librccl loads everywhere, (2) rank 0 made a unique id (a flag byte travels with it), (3) every rank's
        lb_comm_init returned.  On failure EVERY rank raises LbError (bench.py then falls back to transport='torch')."""
        import torch
        from .simulation import comm_unique_id
        dist = self._dist
        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", self.engine.device) if on_gpu else torch.device("cpu")
        if on_gpu:
            torch.cuda.set_device(dev)

        def all_ok(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            return bool(int(t[0]))

        err = None
        try:
            _native.check(_native.lib().lb_comm_available())
        except _native.LbError as exc:
            err = exc
        if not all_ok(err is None):
            raise _native.LbError("RCCL cannot be loaded on every rank (%s)" % (err or "a peer failed"))
        uid = torch.zeros(129, dtype=torch.uint8, device=dev)              # [128] = 1: the id is valid
        if self.rank == 0:
            try:
                uid = torch.tensor(list(comm_unique_id()) + [1], dtype=torch.uint8, device=dev)
            except _native.LbError as exc:
                err = exc
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(uid, src=src, group=self.group)
        uid = uid.cpu().numpy()
        if not uid[128]:
            raise _native.LbError("rank 0 could not create an RCCL unique id (%s)" % (err or "see rank 0"))
        try:
            self.engine.comm_init(uid[:128].tobytes(), self.rank, self.nranks)
        except _native.LbError as exc:
            err = exc
        if not all_ok(err is None):
            raise _native.LbError("lb_comm_init failed on a rank (%s)" % (err or "a peer failed"))

    # -- peer transport inside the engine --------------------------------------------------------------
    def _attach_peer(self):
        """Collective.  Every rank exports its descriptor (IPC handles of its lattices and flag block), the descriptors
        travel as bytes over torch.distributed (any backend: CPU tensors under gloo, which is what several rank processes
        sharing one GPU use), each rank maps its two neighbours.  As with RCCL, every stage ends in an agreement so that a
        rank that fails raises on ALL ranks instead of leaving its peers waiting."""
        import torch
        dist = self._dist
        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", self.engine.device) if on_gpu else torch.device("cpu")
        if on_gpu:
            torch.cuda.set_device(dev)

        def all_ok(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            return bool(int(t[0]))

        err, mine = None, bytes(_native.LB_PEER_HANDLE_BYTES)
        try:
            mine = self.engine.peer_export()
        except _native.LbError as exc:
            err = exc
        if not all_ok(err is None):
            raise _native.LbError("lb_peer_export failed on a rank (%s)" % (err or "a peer failed"))
        t = torch.tensor(list(mine), dtype=torch.uint8, device=dev)
        every = [torch.zeros_like(t) for _ in range(self.nranks)]
        dist.all_gather(every, t, group=self.group)
        descs = [bytes(e.cpu().numpy().tobytes()) for e in every]
        try:
            self.engine.peer_connect(self.rank, self.nranks, descs[self.south] if self.south >= 0 else None,
                                     descs[self.north] if self.north >= 0 else None, min(h for _, h in self.parts))
        except _native.LbError as exc:
            err = exc
        if not all_ok(err is None):
            raise _native.LbError("lb_peer_connect failed on a rank (%s)" % (err or "a peer failed"))

    # -- torch.distributed driven exchange ----------------------------------------------------------
    def _torch_buffers(self):
        if self._bufs is None:
            import torch
            on_gpu = self._dist.get_backend(self.group) == "nccl"
            dev = torch.device("cuda", self.engine.device) if on_gpu else torch.device("cpu")
            if on_gpu:
                torch.cuda.set_device(dev)
                # one stream for kernels, halo copies and torch's collectives' stream dependencies
                self.engine.use_stream(torch.cuda.current_stream().cuda_stream)
            mk = lambda: torch.zeros(18 * self.nx, dtype=torch.float32, device=dev)
            self._bufs = {"send_s": mk(), "send_n": mk(), "recv_s": mk(), "recv_n": mk(), "gpu": on_gpu}
        return self._bufs

    @staticmethod
    def _ptr(t):
        return t.data_ptr()

    def _exchange_torch(self):
        """Edge rows of the lattice the next step reads -> neighbours' ghost rows."""
        import torch
        dist, b = self._dist, self._torch_buffers()
        eng = self.engine
        eng.halo_export(SOUTH, self._ptr(b["send_s"]))
        eng.halo_export(NORTH, self._ptr(b["send_n"]))
        if not b["gpu"]:
            eng.sync()
        grank = (lambda r: dist.get_global_rank(self.group, r)) if self.group is not None else (lambda r: r)
        if self.nranks == 1:
            if self.periodic:
                b["recv_s"].copy_(b["send_n"])
                b["recv_n"].copy_(b["send_s"])
        else:
            # posting order: sends north then south, receives south then north, so that with two
            # ranks (both neighbours are the same peer) the n-th send meets the n-th receive
            ops = []
            if self.north >= 0:
                ops.append(dist.P2POp(dist.isend, b["send_n"], grank(self.north), self.group))
            if self.south >= 0:
                ops.append(dist.P2POp(dist.isend, b["send_s"], grank(self.south), self.group))
            if self.south >= 0:
                ops.append(dist.P2POp(dist.irecv, b["recv_s"], grank(self.south), self.group))
            if self.north >= 0:
                ops.append(dist.P2POp(dist.irecv, b["recv_n"], grank(self.north), self.group))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.south >= 0:
            eng.halo_import(SOUTH, self._ptr(b["recv_s"]))
        if self.north >= 0:
            eng.halo_import(NORTH, self._ptr(b["recv_n"]))
        if not b["gpu"]:
            eng.sync()

    # -- public API -------------------------------------------------------------------------------------
    def set_f(self, f_global):
        """Every rank passes the same global (nx, ny, 9) array and keeps its rows."""
        self.engine.set_f(self._cut(f_global, self.y0, self.h))
        self._ghosts_valid = False

    def set_f_local(self, f_slab):
        self.engine.set_f(f_slab)
        self._ghosts_valid = False

    def run(self, n, wait=True):
        if self.transport in ("rccl", "peer"):
            self.engine.run(n, wait=wait)
            return
        if n and not self._ghosts_valid:
            self._exchange_torch()
        for it in range(n):
            macro = (it == n - 1)
            self.engine.step_boundary(macro)
            self.engine.step_interior(macro)
            self._exchange_torch()
            self.engine.step_finish()
        self._ghosts_valid = True
        if wait:
            self.engine.sync()

    def timed_run(self, n):
        """HIP-event time of run(n) on this rank's stream, ms."""
        self.engine.timer_start()
        self.run(n, wait=False)
        return self.engine.timer_stop()

    def autotune(self, depths=(8, 7, 6, 5), cycles=20, rounds=2, placements=(False, True)):
        """Collective: the ranks time the halo cycle on each candidate TOGETHER -- a depth of the fused kernel (8: seven steps per
        launch by k_deep2, two waves per SIMD, which RCCL's kernel slows far less than it slows k_deep's lone waves) x where the exchange
        runs (beside the interior launches on its own stream, or between them on the compute stream: lb_set_exchange_inline) --,
        `cycles` cycles of 2 x depth live time steps each (every candidate gives the same bits: tuning advances the simulation; twenty,
        because what a transport's kernels cost the launches they run beside shows in a steady state only), the
        slowest rank's time counts, the best of `rounds`, and all set the candidate that is fastest per time step (lb_set_slab_cycle,
        lb_set_exchange_inline).  A whole-grid handle tunes itself (Simulation.autotune); slabs cannot: every rank must run the
        same schedule.  Returns {"depth": chosen, "exchange_inline": chosen, "ms_per_step": {depth: slowest rank's, the better
        placement}, "ms_per_step_inline": {depth: ...}, "steps": time steps advanced}; nothing to choose on the python-driven
        transport."""
        if self.transport not in ("rccl", "peer") or not hasattr(self.engine, "set_slab_cycle"):
            return {"depth": 0, "exchange_inline": False, "ms_per_step": {}, "steps": 0}
        import torch
        dist = self._dist
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        hmin = min(h for _, h in self.parts)
        can_place = hasattr(self.engine, "set_exchange_inline")
        cands = [(d, bool(pl)) for d in depths if hmin >= 16 * min(d, 7) for pl in (placements if can_place else (False,))]
        times, steps = {}, 0
        for r in range(rounds + 1):                      # (round 0 warms every candidate up)
            for d, pl in cands:
                self.engine.set_slab_cycle(d)
                if can_place:
                    self.engine.set_exchange_inline(pl)
                n = 2 * min(d, 7) * cycles
                self.engine.sync()
                dist.barrier(self.group)
                ms = self.timed_run(n)
                steps += n
                t = torch.tensor([ms / n], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                if r:
                    times[(d, pl)] = min(times.get((d, pl), 1e30), float(t[0]))
        best = min(times, key=times.get) if times else (0, False)
        self.engine.set_slab_cycle(best[0])
        if can_place:
            self.engine.set_exchange_inline(best[1])
        return {"depth": best[0], "exchange_inline": best[1],
                "ms_per_step": {d: round(v, 5) for (d, pl), v in times.items() if not pl},
                "ms_per_step_inline": {d: round(v, 5) for (d, pl), v in times.items() if pl}, "steps": steps}

    def get_local_fields(self, which=("f", "feq", "u", "v", "rho")):
        return self.engine.get_fields(which)

    def check(self, warn=False, raise_nonfinite=False, in_engine=False):
        """Collective.  Health of the whole lattice (Simulation.check over every rank's rows): non-finite cells and
        mass summed, Mach number maximised over the ranks -- with torch.distributed on the three scalars (any backend), or,
        in_engine=True on the RCCL transport, inside the engine with ncclAllReduce on its own communicator
        (lb_check(across_ranks = 1))."""
        if in_engine and self.transport == "rccl":
            return self.engine.check(across_ranks=True, warn=warn, raise_nonfinite=raise_nonfinite)
        import torch
        c = self.engine.check()
        on_gpu = self._dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", self.engine.device) if on_gpu else torch.device("cpu")
        t = torch.tensor([c["sum_rho"], float(c["n_nonfinite"])], dtype=torch.float64, device=dev)
        m = torch.tensor([c["max_mach"]], dtype=torch.float32, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        self._dist.all_reduce(m, op=self._dist.ReduceOp.MAX, group=self.group)
        out = {"n_nonfinite": int(t[1]), "max_mach": float(m[0]), "sum_rho": float(t[0])}
        if raise_nonfinite and out["n_nonfinite"]:
            raise FloatingPointError("lattice has %d non-finite cell(s): the run has diverged" % out["n_nonfinite"])
        if warn and out["max_mach"] > 0.1:
            import warnings
            warnings.warn("max_ulb is greater than cs/10! Ma= %g" % out["max_mach"], RuntimeWarning, stacklevel=2)
        return out

    # -- state I/O across ranks (the reference has none) --------------------------------------
    CHECKPOINT_VERSION = 1

    def save_checkpoint(self, path):
        """Collective.  `path` becomes a directory: one shard per rank (`shard_0003.npz`: that slab's populations,
        macroscopic fields and obstacle rows) and `manifest.json` written by rank 0 (grid, boundary family, lattice
        parameters, the row partition).  Nothing is gathered: every rank writes its own rows."""
        import json
        import os
        os.makedirs(path, exist_ok=True)
        g = self.engine.get_fields(("f", "rho", "u", "v"))
        mask = getattr(self.engine, "_mask_host", None)
        if mask is None and self._has_mask:
            mask = self._mask_local
        np.savez(os.path.join(path, "shard_%04d.npz" % self.rank), f=g["f"], rho=g["rho"], u=g["u"], v=g["v"],
                 mask=(np.zeros((0, 0), np.uint8) if mask is None else (np.asarray(mask) != 0).astype(np.uint8)),
                 y0=self.y0, h=self.h)
        if self.rank == 0:
            import numbers
            scalar = lambda v: isinstance(v, (numbers.Real, np.generic)) and not isinstance(v, (bool, np.bool_))
            man = {"version": self.CHECKPOINT_VERSION, "nx": self.nx, "ny": self.ny, "bc": self.bc if isinstance(self.bc, str) else int(self.bc),
                   "omega": float(self.omega), "params": {k: float(v) for k, v in self._engine_kw.items() if scalar(v)},
                   "nranks": self.nranks, "partition_rows": [list(p) for p in self.parts], "has_mask": bool(self._has_mask)}
            with open(os.path.join(path, "manifest.json"), "w") as fh:
                json.dump(man, fh, indent=1)
        self._dist.barrier(group=self.group)

    @staticmethod
    def read_manifest(path):
        import json
        import os
        with open(os.path.join(path, "manifest.json")) as fh:
            man = json.load(fh)
        if man["version"] != DistributedSlab.CHECKPOINT_VERSION:
            raise ValueError("checkpoint format %r, this build reads %d" % (man["version"], DistributedSlab.CHECKPOINT_VERSION))
        return man

    def load_checkpoint(self, path):
        """Collective.  Restore a state written by save_checkpoint -- by ANY number of ranks: every rank re-cuts
        its own rows [y0, y0+h) out of the shards that overlap them (and the obstacle rows of its neighbours)."""
        import os
        man = self.read_manifest(path)
        if (man["nx"], man["ny"]) != (self.nx, self.ny):
            raise ValueError("checkpoint is for a %dx%d grid, this run is %dx%d" % (man["nx"], man["ny"], self.nx, self.ny))
        if _bc_code(man["bc"]) != _bc_code(self.bc):
            raise ValueError("checkpoint was written with boundary family %r, this run has %r" % (man["bc"], self.bc))
        if np.float32(man["omega"]) != np.float32(self.omega):
            raise ValueError("checkpoint has omega = %r, this run %r" % (man["omega"], self.omega))
        for k, v in man["params"].items():
            if k in self._engine_kw and np.float32(self._engine_kw[k]) != np.float32(v):
                raise ValueError("checkpoint has %s = %r, this run %r" % (k, v, self._engine_kw[k]))
        lo, hi = self.y0, self.y0 + self.h
        out = {"f": np.zeros((self.nx, self.h, 9), np.float32, order="F")}
        for k in ("rho", "u", "v"):
            out[k] = np.zeros((self.nx, self.h), np.float32, order="F")
        gmask = np.zeros((self.nx, self.ny), bool) if man["has_mask"] else None
        for r, (y0, h) in enumerate(man["partition_rows"]):
            a, b = max(lo, y0), min(hi, y0 + h)
            if a >= b and gmask is None:
                continue
            with np.load(os.path.join(path, "shard_%04d.npz" % r)) as d:
                if (int(d["y0"]), int(d["h"])) != (y0, h):
                    raise ValueError("shard %d does not match the manifest" % r)
                if gmask is not None and d["mask"].size:
                    gmask[:, y0:y0 + h] = d["mask"] != 0
                if a < b:
                    for k in out:
                        out[k][:, a - lo:b - lo] = d[k][:, a - y0:b - y0]
        eng = self.engine
        if hasattr(eng, "set_obstacle_mask"):
            eng.set_obstacle_mask(None if gmask is None else self._cut(gmask, lo, self.h))
            if gmask is not None and hasattr(eng, "set_obstacle_mask_halo"):
                eng.set_obstacle_mask_halo(*self._mask_halo_rows(gmask, lo, self.h, self.ny, self.periodic))
        elif gmask is not None or self._has_mask:
            raise ValueError("this engine cannot change its obstacle mask")
        self._has_mask = gmask is not None
        self._mask_local = None if gmask is None else self._cut(gmask, lo, self.h)
        if hasattr(eng, "set_fields"):
            eng.set_fields(out["rho"], out["u"], out["v"])
        eng.set_f(out["f"])
        self._ghosts_valid = False
        self._dist.barrier(group=self.group)

    @classmethod
    def from_checkpoint(cls, path, **kw):
        """Build this rank's slab of a run continued from `path` on the CURRENT process group (any rank count)."""
        man = cls.read_manifest(path)
        params = dict(man["params"])
        params.update(kw)
        placeholder = np.zeros((man["nx"], man["ny"]), bool) if man["has_mask"] else None
        slab = cls(man["nx"], man["ny"], man["omega"], bc=man["bc"], obstacle_mask=placeholder, **params)
        slab.load_checkpoint(path)
        return slab

    def get_fields(self, which=("f", "u", "v", "rho")):
        """Gather the slabs of every rank (all ranks receive the global arrays).  For tests and
        small grids; large runs should read ``get_local_fields``."""
        local = self.get_local_fields(which)
        gathered = [None] * self.nranks
        self._dist.all_gather_object(gathered, local, group=self.group)
        return {k: np.asfortranarray(np.concatenate([g[k] for g in gathered], axis=1)) for k in which}
