/*
 * lb_hip.h -- C ABI of liblbhip.so, the MI355X (gfx950) D2Q9 lattice-Boltzmann engine.
 *
 * This is the drop-in boundary for the collide-and-stream path of
 * latticeboltzmann/2d-lb.  In the reference that path sits behind pyopencl:
 * `cl.Program(...).build()` gives `self.kernels`, every step method is
 * `self.kernels.<name>(queue, global, local, *buffers, scalars).wait()` and all
 * state lives in `cl.Buffer`s (LB_D2Q9/dimensionless/opencl_dim.py:203-255,
 * 295-370, 390-415, 495-518).  Each entry point below names the reference
 * interface it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++ or torch types cross this boundary.
 *  - every function returns 0 on success or a negative lb_status; the message
 *    of the last failure on the calling thread is lb_last_error().
 *  - host arrays are borrowed for the duration of the call.  Field layout on
 *    the host is the reference's device layout: plane-major, x fastest,
 *    idx(k,x,y) = k*nx*H + y*nx + x (== the F-ordered (nx,ny[,9]) numpy arrays
 *    of opencl_dim.py:165,279,390-415), H = rows of the slab this handle owns.
 *  - work is enqueued asynchronously on the handle's HIP stream; lb_sync() and
 *    every lb_get_* wait for it (the reference waits after every kernel).
 *  - one host thread per handle.
 */
#ifndef LB_HIP_H
#define LB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LB_ABI_VERSION 10

typedef enum {
    LB_OK = 0,
    LB_ERR_ARG = -1,      /* bad argument / unsupported combination           */
    LB_ERR_HIP = -2,      /* a HIP runtime call failed (no device, OOM, ...)  */
    LB_ERR_STATE = -3,    /* call not valid in the handle's current state     */
    LB_ERR_COMM = -4      /* RCCL failure                                     */
} lb_status;

/* Boundary-condition families.  PIPE is the reference's `move_bcs`
 * (D2Q9.cl:173-261); PERIODIC and CAVITY are build-defined (BASELINE configs
 * 2-4) and specified by oracle/d2q9_oracle.c; VELOCITY_INLET is the reference's
 * second rule set of D2Q9.cl. */
typedef enum {
    LB_BC_PIPE = 0,       /* pressure inlet x=0 / outlet x=nx-1, no-slip y=0,ny-1 */
    LB_BC_PERIODIC = 1,   /* periodic in x and y                               */
    LB_BC_CAVITY = 2,     /* four no-slip walls, north wall moving with lid_u  */
    LB_BC_VELOCITY_INLET = 3  /* D2Q9.cl:263-374: imposed speed inlet_u at x=0 / outlet_u at x=nx-1, north and
                             south rows copy their missing links from the opposite wall row.  Dead code in
                             the reference's `dimensionless` package (only OLD/opencl.py:281-327 launches
                             it).  Whole-grid handles; lb_run fuses it (up to four time steps per launch; from
                             three on the wall-row bands are advanced as a small lattice of their own), the phase
                             entry points run it un-fused. */
} lb_bc_mode;

/* Which of the reference's two (numerically different) paths the handle reproduces.
 * OPENCL: LB_D2Q9/D2Q9.cl driven as opencl_dim.py:372-387 does -- the fused, fast path.
 * CYTHON: LB_D2Q9/dimensionless/cython_dim.pyx:160-359 -- boundary rules before streaming, bounce-back
 *         walls, restricted in-place streaming, moment overrides; PIPE family, whole-grid handles
 *         (a compatibility path for users of the reference's CPU classes: lb_run launches the boundary
 *         phase + one fused pass per step, the phase entry points one kernel each).
 * OPENCL_D2Q9I: see below. */
typedef enum {
    LB_SEM_OPENCL = 0,
    LB_SEM_CYTHON = 1,
    LB_SEM_OPENCL_D2Q9I = 2   /* LB_D2Q9/D2Q9i.cl driven as dimensionless/opencl_dim_D2Q9i.py does: the "incompressible"
                                 fork of the OpenCL path -- momentum in place of velocity (D2Q9i.cl:90-94), inner =
                                 rho + 3 cu + 4.5 cu^2 - 1.5 usq (:58), re-derived inlet / outlet (:194-205), u, v
                                 re-zeroed in the obstacle every step.  PIPE family, whole-grid handles; fused like the
                                 OpenCL path.  Restated as the fork has it: it is unstable (tests/golden/o2_d2q9i_53x27). */
} lb_semantics;

typedef struct {
    int32_t nx, ny;           /* global grid (reference: self.nx, self.ny, opencl_dim.py:191-201) */
    int32_t y0, local_ny;     /* row slab [y0, y0+local_ny) owned by this handle; 0, ny for one GPU */
    int32_t bc_mode;          /* lb_bc_mode */
    int32_t device;           /* HIP device ordinal, or LB_DEVICE_CPU */
    float omega;              /* np.float32(self.omega), opencl_dim.py:369 */
    float inlet_rho;          /* np.float32(self.inlet_rho), :336 */
    float outlet_rho;
    float lid_u;              /* CAVITY only */
    float rho0;               /* CAVITY corner closure density */
    int32_t flags;            /* LB_FLAG_* */
    int32_t semantics;        /* lb_semantics; 0 = the OpenCL path */
    float inlet_u, outlet_u;  /* VELOCITY_INLET only: u_w, u_e of OLD/opencl.py:283-286 */
    int32_t reserved[1];      /* must be zero */
} lb_params;

/* lb_params.device = LB_DEVICE_CPU: the product's own CPU backend (2d-lb_amd/csrc/cpu_backend.h): the reference's CPU class
 * LB_D2Q9/dimensionless/cython_dim.pyx Pipe_Flow.run (:346-359) restated for the host, single-threaded like the reference, in
 * the reference's mixed float32 / float64 arithmetic.  Needs semantics = LB_SEM_CYTHON, bc_mode = LB_BC_PIPE, a whole-grid
 * handle and no flags.  BASELINE.json's first configuration ("256 x 256 Poiseuille, CPU path, no GPU") runs on it without a
 * GPU in the box, and bench.py times it next to the GPU numbers.  It is selected by this value and by nothing else: a handle
 * with a device ordinal >= 0 never falls back to the host.  Entry points that only make sense on a device (lb_set_stream,
 * the slab / halo / comm / peer calls, lb_run_group, lb_run_batch, lb_autotune*, lb_copy_calibration, the corner state) return
 * LB_ERR_STATE on such a handle; u, v are float64 inside, float32 across the ABI like everywhere else. */
#define LB_DEVICE_CPU (-1)

/* (declared below: lb_set_params_f64 hands a CPU-backend handle the reference's float64 omega / inlet_rho / outlet_rho, which
 *  lb_params carries as float32) */

/* Treat the handle as a row slab with ghost rows even when it owns the whole grid: its halo
 * is then filled by lb_halo_import / the RCCL exchange (a 1-rank periodic ring sends to
 * itself).  Lets the multi-GPU code path run, and be tested, on a single GPU. */
#define LB_FLAG_HALO 1
/* Device layout of the two lattices.  Default: the nine plane-rows of a lattice row are stored together
 * ([row][plane][pitch]); with this flag each plane is contiguous ([plane][row][pitch], the round-1 layout).  Results are
 * bit-identical; the marching kernels stream ~12 % faster from interleaved rows at 8192^2. */
#define LB_FLAG_PLANAR 2
/* rho, u, v after lb_run.  BGK relaxation conserves rho and rho*u, so the moments of the post-collision populations a run
 * leaves behind are the rho, u, v of its last step (the reference stores the pre-collision moments, opencl_dim.py:384-385:
 * equal up to rounding).  By default lb_run therefore stores nothing but the populations in the PIPE / PERIODIC / CAVITY
 * families of the OpenCL path, and the fields are rebuilt from them by one device pass the first time they are asked for
 * (lb_get_macro, lb_update_feq, ..., or before anything else overwrites the populations).  With this flag the last launch
 * of every lb_run stores them itself, as round 2 did (12 B per cell more in that launch).  The other families /
 * semantics always store: their fields are not plain moments (imposed inlet speeds, wall overrides, momentum). */
#define LB_FLAG_EAGER_MACRO 4

/* Obstacle-mask rows a slab keeps of each neighbour (lb_set_mask_halo): the fourteen-step halo cycle (two
 * seven-step launches per exchange, ABI 8) recomputes seven of the neighbour's rows and reads the mask six rows
 * beyond them.  (ABI 7: 9 rows, the ten-step cycle.) */
#define LB_MASK_HALO_ROWS 13

typedef struct lb_sim lb_sim; /* opaque: device buffers, streams, events, RCCL communicator */

/* ---- lifetime (replaces init_opencl + allocate_constants + the cl.Buffer
 *      allocations of __init__, opencl_dim.py:165-176, 203-255) ------------- */
int lb_abi_version(void);
int lb_device_count(void);                     /* <0 on error */
const char *lb_last_error(void);
int lb_create(const lb_params *p, lb_sim **out);
int lb_destroy(lb_sim *s);
/* CPU backend only (LB_DEVICE_CPU): the reference keeps omega, inlet_rho and outlet_rho as np.float64 and its mixed-precision
 * expressions see them as such (cython_dim.pyx:86-95, 139-141); lb_params rounds them to float32.  This call restores the
 * float64 values, which makes the backend reproduce the reference's populations bit for bit.  LB_ERR_STATE on a GPU handle
 * (the device kernels compute in float32 throughout). */
int lb_set_params_f64(lb_sim *s, double omega, double inlet_rho, double outlet_rho);
int lb_sync(lb_sim *s);
/* Run everything on an externally owned hipStream_t (e.g. torch's current
 * stream) instead of the handle's own; pass NULL to go back. */
int lb_set_stream(lb_sim *s, void *hip_stream);

/* ---- host <-> device state (replaces cl.Buffer(COPY_HOST_PTR, hostbuf=...)
 *      and cl.enqueue_copy, opencl_dim.py:291-293, 323-327, 395-407, 502) ---- */
int lb_set_macro(lb_sim *s, const float *rho, const float *u, const float *v);  /* each [H][nx]   */
int lb_get_macro(lb_sim *s, float *rho, float *u, float *v);
int lb_set_f(lb_sim *s, const float *f);       /* [9][H][nx]; also fills f_streamed (:323-327) */
int lb_get_f(lb_sim *s, float *f);
int lb_get_feq(lb_sim *s, float *feq);         /* [9][H][nx] */
int lb_set_mask(lb_sim *s, const int32_t *mask); /* [H][nx], 1 = solid (opencl_dim.py:468, 502); NULL clears */
/* VELOCITY_INLET only: the eight corner links that no kernel of that rule set ever writes (the reference's push
 * `move` drops what would enter from outside the box, D2Q9.cl:151-169, and the rules skip them): they keep the
 * values f had at the last lb_set_f / lb_init_pop -- in the reference inside its f_streamed buffer.  Order:
 * f1(0,0), f8(0,0), f1(0,ny-1), f5(0,ny-1), f3(nx-1,0), f7(nx-1,0), f3(nx-1,ny-1), f6(nx-1,ny-1).  A checkpoint
 * needs them next to f (lb_set_f resets them). */
int lb_get_corner_state(lb_sim *s, float *out8);
int lb_set_corner_state(lb_sim *s, const float *in8);

/* ---- the reference's per-phase methods, one kernel each (slow, un-fused;
 *      API and test parity).  Single-slab handles only. ------------------- */
int lb_move(lb_sim *s);                 /* kernels.move + kernels.copy_buffer, opencl_dim.py:339-353 */
int lb_move_bcs(lb_sim *s);             /* kernels.move_bcs (+ bounceback_in_obstacle), :329-337, 510-518 */
int lb_update_hydro(lb_sim *s);         /* kernels.update_hydro, :355-362 */
int lb_update_feq(lb_sim *s);           /* kernels.update_feq, :295-306 */
int lb_collide_particles(lb_sim *s);    /* kernels.collide_particles, :364-370 */
int lb_zero_velocity_in_obstacle(lb_sim *s); /* kernels.set_zero_velocity_in_obstacle, :506-508 */
int lb_init_pop(lb_sim *s);             /* f = f_streamed = feq (device side of init_pop, :308-327) */

/* ---- the hot path: n time steps (replaces the body of Pipe_Flow.run, opencl_dim.py:372-387:
 *      move -> move_bcs(+obstacle) -> update_hydro -> update_feq -> collide_particles, 6-8 launches and
 *      as many host waits per step) by fused launches that advance one to four time steps each
 *      (k_step, k_step2, k_step3, k_step4, k_tile4; results bitwise independent of which) and no host wait.
 *      rho,u,v are those of the LAST step: rebuilt on demand from the populations it left in the plain
 *      families (see LB_FLAG_EAGER_MACRO), stored by the last launch in the others; feq is rebuilt from them on
 *      demand.  Handles with LB_SEM_CYTHON run the first step's boundary phase as a launch of its own, then one
 *      pass per step (k1_fstep) or per four steps (k1_tile4, LDS tiles), each pass ending with the next step's
 *      boundary rule.
 *      Multi-slab handles exchange their halo rows inside lb_run when a communicator is attached
 *      (lb_comm_init), otherwise the caller drives lb_step_boundary / lb_halo_export /
 *      lb_halo_import / lb_step_interior. */
int lb_run(lb_sim *s, int n_steps);

/* ---- row-slab decomposition (new: the reference is single-device) -------- */
/* One fused step split in two launches so the halo exchange can overlap the
 * interior: boundary = local rows {0, H-1}, interior = rows 1..H-2.
 * lb_step_finish swaps the lattices.  write_macro != 0 also stores rho,u,v. */
int lb_step_boundary(lb_sim *s, int write_macro);
int lb_step_interior(lb_sim *s, int write_macro);
int lb_step_finish(lb_sim *s);
/* Halo rows of the lattice that the NEXT step will read (= the one being written between
 * lb_step_boundary and lb_step_finish, the current one otherwise).  side 0 = south edge, 1 = north
 * edge.  A halo is lb_halo_floats() = 18*nx floats: 18 row segments, three rows deep (what the
 * three-step kernel needs; a superset of what the two- and single-step kernels read):
 *   north edge out / south ghost in: row H-3 (resp. -3): k=2,5,6; row H-2 (resp. -2): k=0,1,3,2,5,6;
 *                                    row H-1 (resp. -1): k=0..8
 *   south edge out / north ghost in: row 0 (resp. H): k=0..8; row 1 (resp. H+1): k=0,1,3,4,7,8;
 *                                    row 2 (resp. H+2): k=4,7,8
 * export copies the edge rows into buf, import copies buf into the ghost rows; the buffer a slab
 * exports on its north side is what its northern neighbour imports on its south side.  buf may be
 * host or device memory (hipMemcpyDefault). */
int lb_halo_floats(lb_sim *s);
int lb_halo_export(lb_sim *s, int side, void *buf);
int lb_halo_import(lb_sim *s, int side, const void *buf);
/* Obstacle-mask rows of the neighbouring slabs next to this one: south_rows = global rows
 * y0-LB_MASK_HALO_ROWS .. y0-1 (nearest last), north_rows = rows y0+H .. y0+H+LB_MASK_HALO_ROWS-1 (nearest
 * first), each [LB_MASK_HALO_ROWS][nx] int32, NULL = no solid cells.  The multi-step kernels recompute
 * the neighbours' edge rows and need their masks. */
int lb_set_mask_halo(lb_sim *s, const int32_t *south_rows, const int32_t *north_rows);
/* Advance `count` slab handles that tile one grid on ONE device in lock step: the multi-GPU schedule and kernels
 * without a second GPU (verification).  Halos move through the pack / unpack kernels of the RCCL path, the receiver
 * reading the sender's buffer; the members' streams are ordered by events alone, as lb_run's are (lb_set_debug_sync adds device joins for diagnosis). */
int lb_run_group(lb_sim **sims, int count, int n_steps);

/* Population sets (the periodic multi-population lattices of the reference's research forks: porous_media/
 * single_component.cl:338-375 `move_periodic` streams population `cur_field` of a [jumper][population][y][x] array with
 * periodic wrap; single_component.py:679-751 steps every population per iteration).  `count` (<= 8) whole-grid PERIODIC
 * handles of one geometry on one device, one per population, each with its own omega, advance n_steps in lock step
 * with ONE fused launch per time step for all of them.  Bitwise equal to lb_run on each handle. */
int lb_run_batch(lb_sim **sims, int count, int n_steps);

/* RCCL point-to-point halo exchange over xGMI, one rank per GPU.  Rank r owns
 * slab r; neighbours are r-1 (south) and r+1 (north), wrapping for PERIODIC.
 * unique_id is the 128-byte ncclUniqueId: rank 0 obtains it with
 * lb_comm_unique_id and the caller broadcasts it (torch.distributed).
 * lb_comm_init is collective (every rank of the communicator calls it: the ranks agree on the smallest
 * slab height there, which decides the kernels and the exchange rhythm).  lb_check(across_ranks = 1) is the only
 * other collective (two all-reduces of three scalars, outside the data path).  Afterwards lb_run on a slab
 * handle exchanges halos itself: two seven- / six-step (large slabs of >= 112 / 96 rows), five-step (>= 80 rows), four-step
 * (>= 64 rows) or three-step (>= 32 rows) launches per exchange with ghost zones 14 / 12 / 10 / 8 / 6 rows deep when nx >= 512,
 * otherwise one exchange of the 3-deep halo per launch. */
int lb_comm_available(void);               /* 0 when librccl can be loaded in this process (no communicator is made) */
int lb_comm_unique_id(void *unique_id_128);
int lb_comm_init(lb_sim *s, const void *unique_id_128, int rank, int nranks);

/* Peer transport: the halo rows are stored DIRECTLY into the neighbours' ghost rows, through device memory mapped from
 * one rank's process into the other's (hipIpcMemHandle; over xGMI between GPUs, plain device memory between processes that
 * share a GPU), and the ranks meet at device-side flags (sequence numbers in fine-grained device memory, written with
 * system-scope release stores by the neighbour's kernels, polled by one lane) -- no packing, no unpacking, no library
 * call on the data path, nothing the host waits for: the alternative SURVEY.md section 8(e) lists beside RCCL (the
 * reference itself is single-device: opencl_dim.py:229-240).  lb_run's schedule is the one it runs over RCCL.
 *   lb_peer_export   fills LB_PEER_HANDLE_BYTES with this handle's descriptor (IPC handles of its two lattices and its
 *                    flag block, its geometry).  The caller carries it to the two neighbouring ranks by any means
 *                    (torch.distributed all_gather of bytes, a file, a pipe).
 *   lb_peer_connect  maps the neighbours (NULL = wall; south = rank-1, north = rank+1, wrapping for PERIODIC; a
 *                    descriptor exported by THIS process is used in place, so a 1-rank periodic ring talks to itself) and
 *                    switches lb_run on this handle to the peer transport.  min_h = the smallest slab height over all ranks
 *                    (decides the kernels and the exchange rhythm; every rank passes the same number).  Not a collective by
 *                    itself, but every rank must have exported before anyone connects, and all ranks must call lb_run with
 *                    the same step counts (as over RCCL).
 * A rank whose neighbour does not arrive within LB_PEER_TIMEOUT_S seconds (environment, default 20) gives up waiting on
 * the device, and the next lb_sync / lb_check on that handle fails with LB_ERR_COMM.  One handle per process and GPU (a
 * handle waits on the device for its neighbours' kernels: several such handles in ONE process could share a hardware
 * queue and wait for each other forever -- lb_run_group is the in-process stand-in). */
#define LB_PEER_HANDLE_BYTES 384
int lb_peer_export(lb_sim *s, void *handle_out);
int lb_peer_connect(lb_sim *s, int rank, int nranks, const void *south_handle, const void *north_handle, int min_h);

/* ---- health check (the reference's forks warn when max |u| exceeds a tenth of the speed of sound,
 *      porous_media/single_component.py:221-225, and print field sums while debugging, check_fields() :753-766; the
 *      `dimensionless` classes have nothing, and a diverged run is only noticed after downloading a field) ------------
 * One device pass over the current populations of this handle's rows: the number of cells whose density or velocity is
 * not finite, the largest Mach number max |u| / c_s (c_s = 1/sqrt(3); velocity = first moment / density in every
 * semantics) and the total mass, sum of rho, over the finite cells.  Reduced on the device (wave64 cross-lane
 * reduction, one partial per workgroup, folded in a fixed order: the result is reproducible), 24 bytes travel to the
 * host.  Waits for the handle's work.  across_ranks != 0 on a handle with a communicator (lb_comm_init): the three
 * values are combined over all ranks with ncclAllReduce (sum, max, sum) -- a collective, every rank calls it.
 * Any output pointer may be NULL. */
int lb_check(lb_sim *s, int across_ranks, int64_t *n_nonfinite, float *max_mach, double *sum_rho);
/* lb_run_group joins the device at chosen points (bits: 1 after every launch phase, 2 after every exchange, 4 after every
 * step, 8 at entry and exit); 0 = events only, the schedule lb_run itself relies on.  Process-wide; initial value from the
 * environment variable LB_DEBUG_SYNC, default 0.  Returns the previous value. */
int lb_set_debug_sync(int bits);

/* ---- measurement --------------------------------------------------------- */
/* hipEvent pair on the handle's stream: start, [enqueue work], stop -> ms. */
int lb_timer_start(lb_sim *s);
int lb_timer_stop(lb_sim *s, float *elapsed_ms);
/* Device layout facts for DESIGN.md / bench.py: pitch = padded row width (floats) = row pitch of rho, u, v and (bytes) of
 * the mask; plane stride (floats) of the lattices: = pitch with interleaved rows (element (k, y, x) at
 * (y * 9 + k) * pitch + x), = (local_ny + 28) * pitch with LB_FLAG_PLANAR; bytes allocated. */
int lb_layout(lb_sim *s, int64_t *pitch, int64_t *plane_stride, int64_t *bytes_allocated);
/* Time steps advanced by one launch of the hot kernel in lb_run with the current variant / tuning:
 * 7 ... 2 when a seven- ... two-steps-per-pass kernel is in use, else 1 (bench.py prices a launch
 * with it). */
int lb_steps_per_launch(lb_sim *s);
/* How lb_run(n_steps) on a whole-grid OpenCL-path GPU handle splits the run into launches with the current variant / tuning:
 * returns the number of launches and writes the time steps of each (in launch order) into depths[0 .. max_launches-1] (NULL:
 * count only).  Every launch of a marching kernel moves the same bytes, so bench.py prices a block of K steps by its launches.
 * LB_ERR_STATE (no message) on slab, Cython-path and CPU handles. */
int lb_plan_launches(lb_sim *s, int n_steps, int *depths, int max_launches);
/* Name of that kernel, e.g. "k_deep<7> (...)<PERIODIC>", written into buf (NUL-terminated, truncated to buflen). */
int lb_hot_kernel(lb_sim *s, char *buf, int buflen);
/* Pick the fastest configuration of the fused kernels for THIS grid by timing each candidate on a few
 * live time steps (all candidates give bitwise identical results, so this simply advances the
 * simulation): returns the number of steps advanced (0 when there is nothing to choose -- also under a variant forced with
 * lb_set_variant, which fixes the kernels), <0 on error.
 * Blocks the host (it reads HIP event times).  lb_run never tunes by itself.  A runner-up within 5 % of the winner is timed against it
 * once more over longer samples.  With LB_TUNE_CACHE set (below) a result remembered for this shape is taken over instead (returns 0). */
int lb_autotune(lb_sim *s);
/* The same with one sample per candidate, for callers that are about to run max_steps steps anyway and
 * will wait for them (the Python classes' blocking run()): tunes only when the handle is untuned, the
 * variant automatic and the pass (361 steps; 889 on grids <= 768^2) fits into max_steps; returns the number
 * of steps advanced, 0 when it did nothing.
 *
 * Environment: LB_TUNE_CACHE=<file> (or "mem": this process only) remembers every result of lb_autotune / lb_autotune_quick under
 * the handle's shape (GPU, grid, rows owned, boundary family, mask or not, layout flags, semantics; one text line each) and lets the
 * first lb_run / lb_autotune_quick of a later handle of that shape take it over -- kernel, waves per CU and the measured launch
 * costs lb_plan_launches splits runs by -- without spending a step on tuning.  Unset (the default): every handle starts from the
 * size heuristic.  Only speed depends on it: every candidate gives the same bits. */
int lb_autotune_quick(lb_sim *s, int max_steps);
/* Calibration launch: a plain 16-byte-per-lane copy of the current lattice into the other one
 * (which is scratch between steps).  *bytes_moved = bytes read + written.  Known traffic in the
 * fused kernel's access shape: corrects rocprofv3 FETCH_SIZE on gfx950 and gives the device's
 * own streaming ceiling. */
int lb_copy_calibration(lb_sim *s, int nontemporal, int64_t *bytes_moved);
/* Kernel variant selector for tuning experiments: -1 = automatic (default); otherwise bit 0
 * non-temporal stores, bit 1 non-temporal loads, bits 2-3 rows per workgroup (0: 4, 1: 1, 2: 2),
 * bit 4 XCD-aware tile order, bit 5 two time steps per pass where applicable (nx >= 512), bit 6 three
 * time steps per pass, bit 7 slabs exchange their halo after every launch instead of every two (no
 * halo cycle), bit 8 four time steps per pass (nx >= 512; whole-grid handles of >= 128 rows, slabs of >= 64), bit 9
 * four time steps per pass through 32 x 16 LDS tiles (whole-grid handles of >= 64 x 64 cells; for small grids), bit 10
 * k_step4 without its one-row-ahead gather, bit 11 k_step4 / k_step5 without the priority turns of the two waves of a SIMD,
 * bit 12 five time steps per pass on overlapping strips (k_step5: whole-grid handles where bit 8 applies and slabs of >= 80
 * rows -- the ten-step halo cycle; what the automatic choice takes from 1200^2 periodic / 1850^2 walled cells of a whole
 * grid, 1280^2 cells of a slab or of the velocity-inlet family), bit 14 six and bit 15 (with bit 14) seven time steps per pass
 * (k_deep, one wave per SIMD: whole-grid handles and slabs of >= 96 / 112 rows -- the twelve- / fourteen-step halo cycle --, not
 * the velocity-inlet family; automatic from 1100^2 (six steps; seven from 1900^2) periodic (1250^2 with obstacle-mask cells; slabs: 2400^2) / 1700^2 walled (slabs: 3800^2) cells), bit 16
 * (with bits 14, 15) the seven-step launches by k_deep2 -- two waves per strip and direction, two waves per SIMD (round 6; automatic on walled whole grids of 1700^2 ... 2900^2 cells, one of lb_autotune's candidates), bit 13
 * the LDS-tile kernel takes its tiles in launch order instead of one band of tile rows per XCD (bits 10, 11, 13: A/B
 * switches of things on by default).  Results never depend on it (bitwise); the ranks of one run must use the same value. */
int lb_set_variant(lb_sim *s, int variant);
/* Slab handles (round 6; new work, the reference is single-device: opencl_dim.py:229-240).  Depth of the fused kernel the halo cycle
 * of lb_run runs on (the cycle is 2 x depth time steps between two exchanges): 0 = automatic (the size thresholds of
 * lb_set_variant's bits 12, 14, 15), 3 ... 7 = that depth wherever the smallest slab of the run has >= 16 x depth rows, 8 = seven
 * steps per launch by k_deep2<7> (two waves per strip and direction, in pairs per SIMD) instead of k_deep<7> (lone waves: RCCL's
 * send / receive kernel slows the launches it runs beside by 5-30 %; k_deep2's do not run longer for it).  With 0 the
 * seven-step cycle runs on k_deep2 under the RCCL transport and on k_deep otherwise.  Results never depend on it (bitwise); EVERY
 * rank of a run must set the same value -- the ranks time the candidates together and agree (LB_D2Q9/slabs.py:
 * DistributedSlab.autotune). */
int lb_set_slab_cycle(lb_sim *s, int depth);
/* Where the halo exchange of a cycle runs (slabs; ABI 10).  0, the default: on the handle's communication stream, beside the second
 * interior launch and the inner part of the edge bands -- hidden, if its kernels find room beside kernels that fill the chip.  1: on the
 * COMPUTE stream, behind the second interior launch and in front of the next first one -- exposed (pack + transfer + unpack, ~45 us of a
 * self-exchange) but beside nothing.  RCCL's send / receive kernel (59 workgroups with 20 KB of LDS each) takes 15 us alone and up to a
 * whole launch beside k_deep, which it slows by 5-30 %; which placement wins depends on the slab's height and on the link, so the ranks
 * time both together and agree (DistributedSlab.autotune).  Results never depend on it (bitwise); every rank sets the same value.
 * Whole-grid handles and the CPU backend ignore it. */
int lb_set_exchange_inline(lb_sim *s, int on);
/* Diagnosis of a multi-GPU run.  lb_exchange_timing(s, 1) brackets every halo exchange of lb_run with a pair of timing events on the
 * stream that carries it (up to 256 exchanges between two queries; more are counted as dropped, not timed); lb_exchange_stats waits
 * for the exchanges recorded so far and returns their number, their total and longest duration in milliseconds -- pack / push, the
 * transfer, the wait for the neighbours' matching calls, unpack --, the depth of the halo cycle in use and the rows of one edge band
 * of its second launch (2 x depth + the extra rows that keep the band's waves busy as long as the interior's, lb_hip.cpp:
 * band_extra), then starts over.  The bands are split -- only their outer 2 x depth rows wait for the exchange, which runs on a stream
 * of its own beside the rest --, so an exchange delays the compute stream once it takes longer than about a whole launch; bench.py
 * --gpus N prints the figures per rank.  (The three streams of a slab handle must not share a hardware queue: a process that creates
 * many streams wants GPU_MAX_HW_QUEUES=8 set before the HIP runtime loads -- INTEGRATION.md.) */
int lb_exchange_timing(lb_sim *s, int enable);
int lb_exchange_stats(lb_sim *s, int64_t *n_exchanges, double *total_ms, double *max_ms, int *cycle_depth, int *band_rows);

#ifdef __cplusplus
}
#endif
#endif /* LB_HIP_H */
