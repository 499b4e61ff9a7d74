// kernels_phases.h -- one kernel per reference kernel (un-fused, API / test parity) and the halo
// pack / unpack kernels of the row-slab decomposition.  Included by lb_hip.cpp only.
#pragma once
#include "d2q9_cell.h"

namespace {

// ---- un-fused kernels: the reference's phases one by one (API / test parity) ----------------
struct PhaseArgs {
    float u_w, u_e;        // VELOCITY_INLET
    float *f, *fs, *feq;   // plane 0, row 0, x 0
    float *rho, *u, *v;
    const uint8_t *mask;
    long long plane;       // lattice: plane stride
    int pitch;             // lattice: row stride (StepArgs)
    int fpitch;            // fields and mask: row pitch
    int nx, ny, bc;
    float omega, rho_in, rho_out, lid_u, rho0;
};

__device__ __constant__ int d_cx[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
__device__ __constant__ int d_cy[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
__device__ __constant__ float d_w[9] = {4.f / 9.f,  1.f / 9.f,  1.f / 9.f,  1.f / 9.f, 1.f / 9.f,
                                        1.f / 36.f, 1.f / 36.f, 1.f / 36.f, 1.f / 36.f};

// D2Q9.cl:139-171 in pull form: a cell whose upstream neighbour is outside the box keeps the
// stale content of f_streamed, exactly like the reference's dropped push.
__global__ void k_move(const PhaseArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= a.nx) return;
    int sx = x - d_cx[k], sy = y - d_cy[k];
    if (a.bc == LB_BC_PERIODIC) {
        sx = (sx + a.nx) % a.nx;
        sy = (sy + a.ny) % a.ny;
    }
    if (sx < 0 || sx >= a.nx || sy < 0 || sy >= a.ny) return;
    a.fs[k * a.plane + (long long)y * a.pitch + x] = a.f[k * a.plane + (long long)sy * a.pitch + sx];
}

__global__ void k_bcs(const PhaseArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x, S = a.plane;
    float *f = a.f + o;
    Cell c = {f[0], f[S], f[2 * S], f[3 * S], f[4 * S], f[5 * S], f[6 * S], f[7 * S], f[8 * S]};
    if (x == 0 || x == a.nx - 1 || y == 0 || y == a.ny - 1) {
        const bool w = (x == 0), e = (x == a.nx - 1), so = (y == 0), no = (y == a.ny - 1);
        if (a.bc == LB_BC_PIPE) bc_pipe_cell(c, w, e, so, no, a.rho_in, a.rho_out);
        if (a.bc == LB_BC_PIPE_I) bc_pipe_i_cell(c, w, e, so, no, a.rho_in, a.rho_out);
        if (a.bc == LB_BC_CAVITY) bc_cavity_cell(c, w, e, so, no, a.lid_u, a.rho0);
    }
    bounce_cell(c, a.mask && a.mask[(long long)y * a.fpitch + x]);
    f[S] = c.f1; f[2 * S] = c.f2; f[3 * S] = c.f3; f[4 * S] = c.f4;
    f[5 * S] = c.f5; f[6 * S] = c.f6; f[7 * S] = c.f7; f[8 * S] = c.f8;
}

__global__ void k_hydro(const PhaseArgs a)   // D2Q9.cl:67-100
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = a.f[k * a.plane + o];
    const float rho = f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7] + f[8];
    const float inv = 1.0f / rho;
    const long long m = (long long)y * a.fpitch + x;
    a.rho[m] = rho;
    a.u[m] = (f[1] - f[3] + f[5] - f[6] - f[7] + f[8]) * inv;
    a.v[m] = (f[5] + f[2] + f[6] - f[7] - f[4] - f[8]) * inv;
}

__global__ void k_hydro_i(const PhaseArgs a)   // D2Q9i.cl:67-97
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x, S = a.plane;
    const float *f = a.f + o;
    const Cell c = {f[0], f[S], f[2 * S], f[3 * S], f[4 * S], f[5 * S], f[6 * S], f[7 * S], f[8 * S]};
    float rho, ux, uy;
    moments_i_cell(c, rho, ux, uy);
    const long long m = (long long)y * a.fpitch + x;
    a.rho[m] = rho; a.u[m] = ux; a.v[m] = uy;
}

__global__ void k_feq_i(const PhaseArgs a)     // D2Q9i.cl:2-64
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x, m = (long long)y * a.fpitch + x;
    const float rho = a.rho[m], ux = a.u[m], uy = a.v[m];
    const float usq = ux * ux + uy * uy;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float cu = d_cx[k] * ux + d_cy[k] * uy;
        a.feq[k * a.plane + o] = d_w[k] * rho * (rho + 3.f * cu + 4.5f * cu * cu - 1.5f * usq);
    }
}

// Equilibrium of one cell, all nine links: w_k rho (1 + 3 cu + 4.5 cu^2 - 1.5 u^2), written the way the fused kernels'
// equilibrate_cell writes it -- 1 - 1.5 u^2 once, the weight times rho once per weight class, the two links of a direction
// as a pair (feq_pair, d2q9_cell.h) -- so that the un-fused k_feq, the fused Cython-path passes and the fused OpenCL-path
// kernels all round alike (k_feq's output is bitwise what k_step's relaxation used).  ~30 vector instructions per cell.
__device__ __forceinline__ void feq_cell(float rho, float ux, float uy, float (&fe)[9])
{
    const float usq = lb_fma(ux, ux, uy * uy);
    const float base = lb_fma(-1.5f, usq, 1.f);
    const float r0 = (4.f / 9.f) * rho, r1 = (1.f / 9.f) * rho, r2 = (1.f / 36.f) * rho;
    const float r13 = 3.f * r1, r23 = 3.f * r2;
    fe[0] = r0 * base;
    feq_pair<float>(r1, r13, ux, base, fe[1], fe[3]);
    feq_pair<float>(r1, r13, uy, base, fe[2], fe[4]);
    feq_pair<float>(r2, r23, ux + uy, base, fe[5], fe[7]);
    feq_pair<float>(r2, r23, ux - uy, base, fe[8], fe[6]);
}

__global__ void k_feq(const PhaseArgs a)     // D2Q9.cl:2-64
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x, m = (long long)y * a.fpitch + x;
    float fe[9];
    feq_cell(a.rho[m], a.u[m], a.v[m], fe);
#pragma unroll
    for (int k = 0; k < 9; ++k) a.feq[k * a.plane + o] = fe[k];
}

__global__ void k_collide(const PhaseArgs a) // D2Q9.cl:102-121
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= a.nx) return;
    const long long o = k * a.plane + (long long)y * a.pitch + x;
    a.f[o] = a.f[o] * (1.f - a.omega) + a.omega * a.feq[o];
}

__global__ void k_zero_vel(const PhaseArgs a) // D2Q9.cl:377-396
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long m = (long long)y * a.fpitch + x;
    if (a.mask[m]) { a.u[m] = 0.f; a.v[m] = 0.f; }
}

// D2Q9.cl:263-321 `move_bcs_PeriodicBC_VelocityInlet`, in place like the reference: every thread stores only the
// planes its rule sets.  The north row reads planes 4,7,8 of row 0 and the south row planes 2,5,6 of row ny-1 --
// planes this launch never writes on those rows (row 0 stores 2,5,6 [+1,5,8 / 3,6,7 never: the inlet / outlet rules
// skip the wall rows], row ny-1 stores 4,7,8) -- so the launch is race-free.  The obstacle swap is a second launch
// (k_bounce), as in the reference, where `bounceback_in_obstacle` runs after `move_bcs` (OLD obstacle subclass).
__global__ void k_bcs_vel(const PhaseArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    if (!(x == 0 || x == a.nx - 1 || y == 0 || y == a.ny - 1)) return;
    const long long o = (long long)y * a.pitch + x, S = a.plane;
    float *f = a.f + o;
    const float f0 = f[0], f1 = f[S], f2 = f[2 * S], f3 = f[3 * S], f4 = f[4 * S], f5 = f[5 * S], f6 = f[6 * S],
                f7 = f[7 * S], f8 = f[8 * S];
    if (x == 0 && y >= 1 && y < a.ny - 1) {
        const float rho_w = (1.f / (1.f - a.u_w)) * (f0 + f2 + f4 + 2.f * (f3 + f6 + f7));
        const float h = 0.5f * (f2 - f4), t = (1.f / 6.f) * rho_w * a.u_w;
        f[S] = f3 + (2.f / 3.f) * rho_w * a.u_w;
        f[5 * S] = f7 - h + t;
        f[8 * S] = f6 + h + t;
    }
    if (x == a.nx - 1 && y >= 1 && y < a.ny - 1) {
        const float rho_e = (1.f / (1.f + a.u_e)) * (f0 + f2 + f4 + 2.f * (f1 + f5 + f8));
        const float h = 0.5f * (f2 - f4), t = (1.f / 6.f) * rho_e * a.u_e;
        f[3 * S] = f1 - (2.f / 3.f) * rho_e * a.u_e;
        f[6 * S] = f5 + h - t;
        f[7 * S] = f8 - h - t;
    }
    if (y == a.ny - 1) {
        f[4 * S] = a.f[4 * S + x]; f[8 * S] = a.f[8 * S + x]; f[7 * S] = a.f[7 * S + x];
    }
    if (y == 0) {
        const long long top = (long long)(a.ny - 1) * a.pitch + x;
        f[2 * S] = a.f[2 * S + top]; f[6 * S] = a.f[6 * S + top]; f[5 * S] = a.f[5 * S + top];
    }
}

// D2Q9.cl:398-433 `bounceback_in_obstacle` on its own (after k_bcs_vel; k_bcs / k1_bcs fuse it: their rules are
// cell-local)
__global__ void k_bounce(const PhaseArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x, S = a.plane;
    if (!a.mask[(long long)y * a.fpitch + x]) return;
    float *f = a.f + o;
    const float f1 = f[S], f2 = f[2 * S], f3 = f[3 * S], f4 = f[4 * S], f5 = f[5 * S], f6 = f[6 * S], f7 = f[7 * S],
                f8 = f[8 * S];
    f[S] = f3; f[3 * S] = f1; f[2 * S] = f4; f[4 * S] = f2;
    f[5 * S] = f7; f[7 * S] = f5; f[6 * S] = f8; f[8 * S] = f6;
}

// D2Q9.cl:323-374 `update_hydro_PeriodicBC_VelocityInlet`: v on the inlet / outlet columns and u, v
// on the four corner cells are left as they were.
__global__ void k_hydro_vel(const PhaseArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x, S = a.plane;
    const float *f = a.f + o;
    const float f0 = f[0], f1 = f[S], f2 = f[2 * S], f3 = f[3 * S], f4 = f[4 * S], f5 = f[5 * S], f6 = f[6 * S],
                f7 = f[7 * S], f8 = f[8 * S];
    float rho = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + f8;
    const float inv = 1.0f / rho;
    const long long m = (long long)y * a.fpitch + x;
    if (x != 0 && x != a.nx - 1) {
        a.u[m] = (f1 - f3 + f5 - f6 - f7 + f8) * inv;
        a.v[m] = (f5 + f2 + f6 - f7 - f4 - f8) * inv;
    }
    if (x == 0 && y != 0 && y < a.ny - 1) {
        rho = (1.f / (1.f - a.u_w)) * (f0 + f2 + f4 + 2.f * (f3 + f6 + f7));
        a.u[m] = a.u_w;
    }
    if (x == a.nx - 1 && y != 0 && y < a.ny - 1) {
        rho = (1.f / (1.f + a.u_e)) * (f0 + f2 + f4 + 2.f * (f1 + f5 + f8));
        a.u[m] = a.u_e;
    }
    a.rho[m] = rho;
}

// ---- the reference's CPU ("Cython") path as GPU kernels -------------------------------------------
// LB_D2Q9/dimensionless/cython_dim.pyx is a different discretisation of the same pipe flow:
// boundary rules BEFORE streaming and fed by the stored inlet/outlet velocity of the previous step, plain
// bounce-back walls, an in-place streaming whose loop bounds leave four tangential links unmoved on one
// wall row/column each, and overrides in the moment update.  These three kernels restate it phase by
// phase (one thread per cell); k_feq and k_collide are shared.  lb_run fuses them into k1_fstep / k1_tile4 below.
// x = i (0..lx), y = j (0..ly) in the .pyx's notation.

// cython_dim.pyx:204-269 `move_bcs` (+ :468-513 obstacle swap) of one cell; u_here = the stored u of this cell (inlet /
// outlet columns: the value the previous update_hydro left there)
template <typename A>      // (PhaseArgs, or the marching kernels' StepArgs: nx, ny, rho_in, rho_out)
__device__ __forceinline__ void c1_bcs_cell(const A &a, int x, int y, float u_here, bool solid, Cell &c)
{
    const int lx = a.nx - 1, ly = a.ny - 1;
    if (x == 0 && y >= 1 && y < ly) {                         // inlet, stored u of the previous update_hydro
        const float u0 = u_here, t = (1.f / 6.f) * u0 * a.rho_in;
        const float f2 = c.f2, f4 = c.f4;
        c.f1 = c.f3 + (2.f / 3.f) * a.rho_in * u0;
        c.f5 = (-.5f * f2 + .5f * f4) + c.f7 + t;
        c.f8 = (.5f * f2 - .5f * f4) + c.f6 + t;
    } else if (x == lx && y >= 1 && y < ly) {                 // outlet
        const float ul = u_here, t = (1.f / 6.f) * ul * a.rho_out;
        const float f2 = c.f2, f4 = c.f4;
        c.f3 = c.f1 - (2.f / 3.f) * a.rho_out * ul;
        c.f6 = (-.5f * f2 + .5f * f4) + c.f8 - t;
        c.f7 = (.5f * f2 - .5f * f4) + c.f5 - t;
    } else if (y == ly && x >= 1 && x < lx) {                 // north wall: plain bounce-back
        c.f4 = c.f2; c.f8 = c.f6; c.f7 = c.f5;
    } else if (y == 0 && x >= 1 && x < lx) {                  // south wall
        c.f2 = c.f4; c.f6 = c.f8; c.f5 = c.f7;
    } else if (x == 0 && y == 0) {                            // corners :242-269
        const float t = .5f * (-c.f0 - 2.f * c.f3 - 2.f * c.f4 - 2.f * c.f7 + a.rho_in);
        c.f1 = c.f3; c.f2 = c.f4; c.f5 = c.f7; c.f6 = t; c.f8 = t;
    } else if (x == 0 && y == ly) {
        const float t = .5f * (-c.f0 - 2.f * c.f2 - 2.f * c.f3 - 2.f * c.f6 + a.rho_in);
        c.f1 = c.f3; c.f4 = c.f2; c.f5 = t; c.f7 = t; c.f8 = c.f6;
    } else if (x == lx && y == 0) {
        const float t = .5f * (-c.f0 - 2.f * c.f1 - 2.f * c.f4 - 2.f * c.f8 + a.rho_out);
        c.f3 = c.f1; c.f2 = c.f4; c.f6 = c.f8; c.f5 = t; c.f7 = t;
    } else if (x == lx && y == ly) {
        const float t = .5f * (-c.f0 - 2.f * c.f1 - 2.f * c.f2 - 2.f * c.f5 + a.rho_out);
        c.f3 = c.f1; c.f4 = c.f2; c.f6 = t; c.f7 = c.f5; c.f8 = t;
    }
    bounce_cell(c, solid);
}

// ... as a phase of its own, in place
__global__ void k1_bcs(const PhaseArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const int lx = a.nx - 1, ly = a.ny - 1;
    const long long o = (long long)y * a.pitch + x, S = a.plane, m = (long long)y * a.fpitch + x;
    float *f = a.f + o;
    const bool edge = (x == 0 || x == lx || y == 0 || y == ly);
    const bool solid = a.mask && a.mask[m];
    if (!edge && !solid) return;
    Cell c = {f[0], f[S], f[2 * S], f[3 * S], f[4 * S], f[5 * S], f[6 * S], f[7 * S], f[8 * S]};
    c1_bcs_cell(a, x, y, (x == 0 || x == lx) ? a.u[m] : 0.f, solid, c);
    f[S] = c.f1; f[2 * S] = c.f2; f[3 * S] = c.f3; f[4 * S] = c.f4;
    f[5 * S] = c.f5; f[6 * S] = c.f6; f[7 * S] = c.f7; f[8 * S] = c.f8;
}

// cython_dim.pyx:271-299 `move`: the in-place loops read sources before they are overwritten, i.e. they
// are a simultaneous pull restricted to the loops' index ranges; a link outside its range keeps its value.
__global__ void k1_move(const PhaseArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= a.nx) return;
    const int lx = a.nx - 1, ly = a.ny - 1;
    bool moved = false;
    switch (k) {
    case 2: case 6: moved = (y >= 1 && x <= lx - 1); break;     // j = ly..1, i = 0..lx-1
    case 1: case 5: moved = (y >= 1 && x >= 1); break;          // j = ly..1, i = lx..1
    case 4: case 8: moved = (y <= ly - 1 && x >= 1); break;     // j = 0..ly-1, i = lx..1
    case 3: case 7: moved = (y <= ly - 1 && x <= lx - 1); break; // j = 0..ly-1, i = 0..lx-1
    default: break;                                             // the rest population never moves
    }
    const int sx = moved ? x - d_cx[k] : x, sy = moved ? y - d_cy[k] : y;
    a.fs[k * a.plane + (long long)y * a.pitch + x] = a.f[k * a.plane + (long long)sy * a.pitch + sx];
}

// cython_dim.pyx:302-333 `update_hydro` (+ :459-466 obstacle zeroing) of one cell
template <typename A>
__device__ __forceinline__ void c1_moments(const A &a, int x, int y, bool solid, float f0, float f1, float f2,
                                           float f3, float f4, float f5, float f6, float f7, float f8, float &rho,
                                           float &ux, float &uy)
{
    const int lx = a.nx - 1, ly = a.ny - 1;
    rho = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + f8;
    const float inv = __builtin_amdgcn_rcpf(rho);               // (v_rcp_f32, 1 ulp, as moments_cell; the reference divides in float64)
    ux = (f1 - f3 + f5 - f6 - f7 + f8) * inv;
    uy = (f5 + f2 + f6 - f7 - f4 - f8) * inv;
    if (y == 0 || y == ly) { ux = 0.f; uy = 0.f; }              // walls
    if (x == 0) {                                                // pressure inlet: rho pinned, u from the knowns
        rho = a.rho_in;
        ux = 1.f - ((f0 + f2 + f4) + 2.f * (f3 + f6 + f7)) / a.rho_in;
    }
    if (x == lx) {
        rho = a.rho_out;
        ux = -1.f + ((f0 + f2 + f4) + 2.f * (f1 + f5 + f8)) / a.rho_out;
    }
    if (solid) { ux = 0.f; uy = 0.f; }
}

__global__ void k1_hydro(const PhaseArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x, S = a.plane;
    const float *f = a.f + o;
    float rho, ux, uy;
    const long long m = (long long)y * a.fpitch + x;
    c1_moments(a, x, y, a.mask && a.mask[m], f[0], f[S], f[2 * S], f[3 * S], f[4 * S], f[5 * S], f[6 * S], f[7 * S],
               f[8 * S], rho, ux, uy);
    a.rho[m] = rho; a.u[m] = ux; a.v[m] = uy;
}

// The Cython path, one launch per time step, four cells per lane.  Its step is rule -> stream -> moments -> relax
// (cython_dim.pyx:346-359); the rule of step n+1 only touches the cell's own populations (and, on the inlet / outlet
// columns, the u that step n's moments produced for this very cell), so it rides at the END of step n's pass: restricted
// pull from a lattice whose boundary cells have already been through the rule, moments with their overrides, equilibrium,
// relaxation, then -- RULE -- next step's rule on the cells it concerns, then nine aligned 16-byte stores.  A run is
// k1_bcs once (the first step's rule), n passes, the last one without RULE so that the populations it leaves are the
// post-collision ones the reference holds after run().  Same expressions as k1_bcs + k1_move + k1_hydro + k_feq +
// k_collide (c1_bcs_cell, c1_moments, feq_cell): the same bits (test_cython_path_fused_run_equals_phase_calls).
// Loads: the three links with cx = 0 aligned, the six others through 16-byte loads displaced by one element, as in
// k_step; "a link outside its loop range keeps its value" becomes: rows 0 / ny-1 read their own row for the links that do
// not move there (wave-uniform), and the single cells at x = 0 / x = nx-1 get their own value patched in.
template <bool MASK, bool RULE, bool MACRO>
__global__ __launch_bounds__(256) void k1_fstep(const PhaseArgs a)
{
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y * blockDim.y + __builtin_amdgcn_readfirstlane(threadIdx.y);
    if (x4 >= a.fpitch || y >= a.ny) return;
    const int lx = a.nx - 1, ly = a.ny - 1;
    const long long S = a.plane, P = a.pitch;
    const bool up = (y >= 1), dn = (y <= ly - 1);              // wave-uniform: links 1,2,5,6 move iff up, 3,4,7,8 iff dn
    const float *r0 = a.f + (long long)y * P;                   // own row; rows the moving links come from:
    const float *rm = up ? r0 - P : r0, *rp = dn ? r0 + P : r0;
    const int su = up ? 1 : 0, sd = dn ? 1 : 0;                 // x displacement of the diagonal / horizontal links when they move
    // the cells at x = 0 (links 1,5,4,8 stay) and x = lx (links 2,6,3,7 stay): their own values, fetched first
    const bool first = (x4 == 0), last = (x4 <= lx && lx < x4 + 4);
    const int jl = lx & 3;
    float p1 = 0.f, p5 = 0.f, p4 = 0.f, p8 = 0.f, p2 = 0.f, p6 = 0.f, p3 = 0.f, p7 = 0.f;
    if (first) { p1 = r0[1 * S]; p5 = r0[5 * S]; p4 = r0[4 * S]; p8 = r0[8 * S]; }
    if (last) { p2 = r0[2 * S + lx]; p6 = r0[6 * S + lx]; p3 = r0[3 * S + lx]; p7 = r0[7 * S + lx]; }
    f4a q[9];
    q[0] = load4<false>(lane_ptr(r0, x4));
    q[1] = load4u<false>(lane_ptr(r0 + 1 * S - su, x4));        // 1: from (x-1, y)     iff y >= 1 and x >= 1
    q[5] = load4u<false>(lane_ptr(rm + 5 * S - su, x4));        // 5: from (x-1, y-1)
    q[2] = load4<false>(lane_ptr(rm + 2 * S, x4));              // 2: from (x, y-1)     iff y >= 1 and x <= lx-1
    q[6] = load4u<false>(lane_ptr(rm + 6 * S + su, x4));        // 6: from (x+1, y-1)
    q[4] = load4<false>(lane_ptr(rp + 4 * S, x4));              // 4: from (x, y+1)     iff y <= ly-1 and x >= 1
    q[8] = load4u<false>(lane_ptr(rp + 8 * S - sd, x4));        // 8: from (x-1, y+1)
    q[3] = load4u<false>(lane_ptr(r0 + 3 * S + sd, x4));        // 3: from (x+1, y)     iff y <= ly-1 and x <= lx-1
    q[7] = load4u<false>(lane_ptr(rp + 7 * S + sd, x4));        // 7: from (x+1, y+1)
    uc4 mk = {0, 0, 0, 0};
    if (MASK) mk = *reinterpret_cast<const uc4 *>(lane_ptr(a.mask + (long long)y * a.fpitch, x4));
    if (first) { q[1].x = p1; q[5].x = p5; q[4].x = p4; q[8].x = p8; }
    if (last) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j == jl) { q[2][j] = p2; q[6][j] = p6; q[3][j] = p3; q[7][j] = p7; }
    }
    f4a r4, u4, v4;
    const bool wall_row = (y == 0 || y == ly);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = x4 + j;
        const bool solid = MASK && mk[j] != 0;
        float rho, ux, uy;
        c1_moments(a, x, y, solid, q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j], q[6][j], q[7][j], q[8][j], rho, ux, uy);
        r4[j] = rho; u4[j] = ux; v4[j] = uy;
        float fe[9];
        feq_cell(rho, ux, uy, fe);
#pragma unroll
        for (int k = 0; k < 9; ++k) q[k][j] = q[k][j] * (1.f - a.omega) + a.omega * fe[k];
        if (RULE && x <= lx && (wall_row || x == 0 || x == lx || solid)) {
            Cell c = {q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j], q[6][j], q[7][j], q[8][j]};
            c1_bcs_cell(a, x, y, ux, solid, c);                 // (ux: what this pass would store as u on the inlet / outlet columns)
            q[1][j] = c.f1; q[2][j] = c.f2; q[3][j] = c.f3; q[4][j] = c.f4;
            q[5][j] = c.f5; q[6][j] = c.f6; q[7][j] = c.f7; q[8][j] = c.f8;
        }
    }
    float *d = a.fs + (long long)y * P;
#pragma unroll
    for (int k = 0; k < 9; ++k) store4<false>(lane_ptr(d + k * S, x4), q[k]);
    if (MACRO) {
        const long long m = (long long)y * a.fpitch;
        store4<false>(lane_ptr(a.rho + m, x4), r4);
        store4<false>(lane_ptr(a.u + m, x4), u4);
        store4<false>(lane_ptr(a.v + m, x4), v4);
    }
}

// The same four time steps at a time through LDS tiles (k_tile4's scheme, kernels_tile.h): a workgroup loads a 32 x 16
// tile + 4 halo cells on every side of all nine planes into LDS once, advances it four Cython-path steps there -- after
// step s the outermost s rings are stale and no longer computed -- and stores the tile.  A single-step pass of this path
// moves 72 B per cell and sits at the streaming ceiling (79 k MLUPS on the reference's 3751 x 1251 case); four steps per
// pass is the only way past it, and because this path's rule is cell-local and its pull is restricted at the box's
// edges (no cell ever pulls from outside), the tile form needs no wall pass.  RULE_LAST: the fourth step is also followed
// by the next step's boundary rule (false for the last launch of a run).  Same cell functions: same bits.
template <bool MASK, bool MACRO, bool RULE_LAST>
__global__ __launch_bounds__((TileShape<32, 16, 2>::THREADS), 8) void k1_tile4(const PhaseArgs a, int tiles_x, int n_tiles)   // (8 waves per SIMD: four workgroups per CU, as the LDS allows)
{
    typedef TileShape<32, 16, 2> T;
    constexpr int L = T::LW, LH = T::LH, CELLS = T::CELLS, THREADS = T::THREADS, CPT = T::CPT;
    __shared__ float lds[9][CELLS];
    __shared__ unsigned char lmask[CELLS];
    const int tid = threadIdx.x;
    const int tile = xcd_band_tile(blockIdx.x, n_tiles);             // (one band of tile rows per XCD: kernels_tile.h)
    if (tile >= n_tiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int gx0 = tx * 32 - TILE_T, gy0 = ty * 16 - TILE_T;        // global coordinates of region cell (0,0)
    const int lx = a.nx - 1, ly = a.ny - 1;
    const long long P = a.pitch, S = a.plane;
    // the region into LDS in linear order ...
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + i * THREADS;
        if (c < CELLS) {
            const int gx = gx0 + c % L, gy = gy0 + c / L;
            // (cells outside the box are computed from copies of the nearest cells inside; nothing consumes them: a cell on
            //  the box's edge keeps its own value for every link that would come from outside)
            const int sx = min(max(gx, 0), lx), sy = min(max(gy, 0), ly);
            const long long o = (long long)sy * P + sx;
#pragma unroll
            for (int k = 0; k < 9; ++k) lds[k][c] = a.f[k * S + o];
            lmask[c] = (MASK && sx == gx && sy == gy) ? a.mask[(long long)sy * a.fpitch + sx] : 0;
        }
    }
    // ... and the cells I step: the tile itself and the halo rings from the inside out (tile_step_cell, kernels_tile.h)
    int cc[CPT], gxs[CPT], gys[CPT], ring[CPT];
    bool mine[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        int cx, cy;
        const bool have = tile_step_cell<32, 16>(tid, i, cx, cy);
        const int gx = gx0 + cx, gy = gy0 + cy;
        cc[i] = cy * L + cx;
        gxs[i] = gx; gys[i] = gy;
        ring[i] = have ? min(min(cx, L - 1 - cx), min(cy, LH - 1 - cy)) : -1;
        mine[i] = gx >= 0 && gx <= lx && gy >= 0 && gy <= ly;
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 1; s <= TILE_T; ++s) {
        const bool last = (s == TILE_T);
        float out[CPT][9], rr[CPT], uu[CPT], vv[CPT];
        bool act[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = cc[i], x = gxs[i], y = gys[i];
            act[i] = ring[i] >= s;
            if (!act[i]) continue;
            const bool up = (y >= 1), dn = (y <= ly - 1), le = (x >= 1), ri = (x <= lx - 1);      // cython_dim.pyx:271-299
            const float f0 = lds[0][c];
            const float f1 = lds[1][(up && le) ? c - 1 : c];
            const float f5 = lds[5][(up && le) ? c - L - 1 : c];
            const float f2 = lds[2][(up && ri) ? c - L : c];
            const float f6 = lds[6][(up && ri) ? c - L + 1 : c];
            const float f4 = lds[4][(dn && le) ? c + L : c];
            const float f8 = lds[8][(dn && le) ? c + L - 1 : c];
            const float f3 = lds[3][(dn && ri) ? c + 1 : c];
            const float f7 = lds[7][(dn && ri) ? c + L + 1 : c];
            const bool solid = MASK && lmask[c] != 0;
            float rho, ux, uy;
            c1_moments(a, x, y, solid, f0, f1, f2, f3, f4, f5, f6, f7, f8, rho, ux, uy);
            const float fk[9] = {f0, f1, f2, f3, f4, f5, f6, f7, f8};
            float fe[9];
            feq_cell(rho, ux, uy, fe);
#pragma unroll
            for (int k = 0; k < 9; ++k) out[i][k] = fk[k] * (1.f - a.omega) + a.omega * fe[k];
            rr[i] = rho; uu[i] = ux; vv[i] = uy;
            if ((!last || RULE_LAST) && mine[i] && (x == 0 || x == lx || y == 0 || y == ly || solid)) {
                Cell q = {out[i][0], out[i][1], out[i][2], out[i][3], out[i][4], out[i][5], out[i][6], out[i][7], out[i][8]};
                c1_bcs_cell(a, x, y, ux, solid, q);
                out[i][1] = q.f1; out[i][2] = q.f2; out[i][3] = q.f3; out[i][4] = q.f4;
                out[i][5] = q.f5; out[i][6] = q.f6; out[i][7] = q.f7; out[i][8] = q.f8;
            }
            if (last && mine[i]) {                      // (the cells still computed in the fourth step are exactly the tile)
                float *d = a.fs + (long long)y * P + x;
#pragma unroll
                for (int k = 0; k < 9; ++k) d[k * S] = out[i][k];
                if (MACRO) { const long long m = (long long)y * a.fpitch + x; a.rho[m] = rr[i]; a.u[m] = uu[i]; a.v[m] = vv[i]; }
            }
        }
        if (last) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            if (act[i]) {
                const int c = cc[i];
#pragma unroll
                for (int k = 0; k < 9; ++k) lds[k][c] = out[i][k];
            }
        __syncthreads();
    }
}

// ---- macroscopic fields on demand + device-side health check ---------------------------------------------------
// BGK relaxation conserves rho and rho*u, so the moments of the post-collision populations a run() leaves behind
// ARE the rho, u, v of its last step (the reference stores the pre-collision ones, opencl_dim.py:384-385: equal up to
// rounding).  lb_run therefore does not store them in the plain families; this kernel rebuilds them the first time
// somebody asks (lb_get_macro, lb_update_feq, ...).  The same pass reduces what the reference's forks print or warn
// about while they run -- max |u| against the speed of sound (porous_media/single_component.py:221-225), the sums of
// check_fields() (:753-766) -- plus a count of non-finite cells: per workgroup a partial (wave reduction through
// cross-lane moves, then four waves through LDS), and k_check_final folds the partials in a fixed order, so the
// result does not depend on the order in which workgroups retire.
struct CheckPartial {
    double sum_rho;                 // over the finite cells
    unsigned long long nonfinite;   // cells whose rho, u or v is not finite
    float max_usq;                  // max u^2 + v^2 (lattice units)
    int pad;
};

__device__ __forceinline__ void check_reduce_wave(double &s, unsigned long long &n, float &m)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_xor(s, d);
        n += __shfl_xor(n, d);
        m = fmaxf(m, __shfl_xor(m, d));
    }
}

// grid = (ceil(fpitch / 1024), H), 256 threads, 4 cells per lane; origin = plane 0, row 0 of the current lattice
template <bool STORE>
__global__ __launch_bounds__(256) void k_macro_check(const float *origin, long long plane, int pitch, int fpitch, int nx,
                                                     float *rho, float *u, float *v, CheckPartial *part)
{
    __shared__ CheckPartial sh[4];
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    double s = 0.0;
    unsigned long long n = 0;
    float m = 0.f;
    if (x4 < fpitch) {
        const float *r = origin + (long long)y * pitch + x4;
        f4a q[9], r4, u4, v4;
#pragma unroll
        for (int k = 0; k < 9; ++k) q[k] = *reinterpret_cast<const f4a *>(r + k * plane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Cell c = {q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j], q[6][j], q[7][j], q[8][j]};
            float rr, ux, uy;
            moments_cell(c, rr, ux, uy);
            r4[j] = rr; u4[j] = ux; v4[j] = uy;
            if (x4 + j < nx) {
                const float usq = ux * ux + uy * uy;
                const bool ok = fabsf(rr) <= 3.0e38f && fabsf(usq) <= 3.0e38f;      // (false for NaN)
                if (ok) { s += (double)rr; m = fmaxf(m, usq); }
                else ++n;
            }
        }
        if (STORE) {
            const long long o = (long long)y * fpitch + x4;
            *reinterpret_cast<f4a *>(rho + o) = r4;
            *reinterpret_cast<f4a *>(u + o) = u4;
            *reinterpret_cast<f4a *>(v + o) = v4;
        }
    }
    check_reduce_wave(s, n, m);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[wave] = CheckPartial{s, n, m, 0};
    __syncthreads();
    if (threadIdx.x == 0) {
        CheckPartial t = sh[0];
        for (int i = 1; i < 4; ++i) { t.sum_rho += sh[i].sum_rho; t.nonfinite += sh[i].nonfinite; t.max_usq = fmaxf(t.max_usq, sh[i].max_usq); }
        part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

// one workgroup of 1024 threads: thread t folds partials t, t + 1024, ... in that order, then a fixed tree
__global__ __launch_bounds__(1024) void k_check_final(const CheckPartial *part, long long count, CheckPartial *out)
{
    __shared__ CheckPartial sh[16];
    double s = 0.0;
    unsigned long long n = 0;
    float m = 0.f;
    for (long long i = threadIdx.x; i < count; i += 1024) {
        s += part[i].sum_rho; n += part[i].nonfinite; m = fmaxf(m, part[i].max_usq);
    }
    check_reduce_wave(s, n, m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = CheckPartial{s, n, m, 0};
    __syncthreads();
    if (threadIdx.x == 0) {
        CheckPartial t = sh[0];
        for (int i = 1; i < 16; ++i) { t.sum_rho += sh[i].sum_rho; t.nonfinite += sh[i].nonfinite; t.max_usq = fmaxf(t.max_usq, sh[i].max_usq); }
        *out = t;
    }
}

// the folded record laid out for two all-reduces: {sum_rho, count} as doubles, max u^2 as a float
__global__ void k_check_spread(const CheckPartial *res, double *d, float *m)
{
    d[0] = res->sum_rho;
    d[1] = (double)res->nonfinite;
    m[0] = res->max_usq;
}

// Halo pack / unpack: the halo of one edge is 18 (3 rows deep), 45 (6 deep), 63 (8 deep) or 81 (10 deep) row segments scattered
// over the planes (HaloTables on the host side).  One tiny kernel gathers both edges into two contiguous
// buffers (so that an exchange is one send + one receive per neighbour), one scatters the received
// buffers into the ghost rows.  `neg` lists rows -D..-1 (leaves north, counted from row H / arrives
// south, counted from row 0), `pos` rows 0..D-1 (leaves south / arrives north).
struct HaloTable {
    signed char k[117], row[117];    // (D = 14: 9 x 14 - 9 row segments)
    int n;
};

// Copy two runs of whole rows (`width` floats each, rows `rs_src` / `rs_dst` floats apart) of `gridDim.z` planes from one plane set into
// another: run A = n_a rows src row sa.. -> dst row da.., run B = n_b rows sb.. -> db..; 16 bytes per lane (width % 64 == 0).
// The velocity-inlet family's band scheme (lb_hip.cpp vel_band_pass) moves its wall-row bands with it.
__global__ void k_rows_copy(const float *src, float *dst, long long plane_src, long long plane_dst, int width, long long rs_src,
                            long long rs_dst, int n_a, int sa, int da, int n_b, int sb, int db)
{
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, j = blockIdx.y, k = blockIdx.z;
    if (x4 >= width || j >= n_a + n_b) return;
    const int rs = j < n_a ? sa + j : sb + (j - n_a), rd = j < n_a ? da + j : db + (j - n_a);
    *reinterpret_cast<f4a *>(dst + k * plane_dst + rd * rs_dst + x4) =
        *reinterpret_cast<const f4a *>(src + k * plane_src + rs * rs_src + x4);
}

// One wave moves 1 KiB of a row segment: 16 bytes per lane when nx is a multiple of 4 (row starts and buffer segments are
// then 16-byte aligned: pitch % 64 == 0), a dword per lane otherwise.  grid = (ceil(nx / (256 * V)), segments, 2 edges).
template <int V>
__global__ void k_halo_pack(const float *origin, long long plane, int pitch, int h, int nx, float *buf_n, float *buf_s,
                            const HaloTable neg, const HaloTable pos)
{
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * V, seg = blockIdx.y, north = (blockIdx.z == 0);
    if (x >= nx) return;
    float *buf = north ? buf_n : buf_s;
    if (!buf) return;
    const int k = north ? neg.k[seg] : pos.k[seg];
    const long long row = north ? h + neg.row[seg] : pos.row[seg];
    const float *src = origin + k * plane + row * pitch + x;
    float *dst = buf + (long long)seg * nx + x;
    if (V == 4) *reinterpret_cast<f4a *>(dst) = *reinterpret_cast<const f4a *>(src);
    else *dst = *src;
}

template <int V>
__global__ void k_halo_unpack(float *origin, long long plane, int pitch, int h, int nx, const float *buf_s,
                              const float *buf_n, const HaloTable neg, const HaloTable pos)
{
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * V, seg = blockIdx.y, north = (blockIdx.z == 0);
    if (x >= nx) return;
    const float *buf = north ? buf_n : buf_s;
    if (!buf) return;
    const int k = north ? pos.k[seg] : neg.k[seg];
    const long long row = north ? h + pos.row[seg] : neg.row[seg];
    float *dst = origin + k * plane + row * pitch + x;
    const float *src = buf + (long long)seg * nx + x;
    if (V == 4) *reinterpret_cast<f4a *>(dst) = *reinterpret_cast<const f4a *>(src);
    else *dst = *src;
}

// ---- peer transport (lb_peer_export / lb_peer_connect): halo rows stored straight into the neighbours' ghost rows -------
// Every rank owns a block of flags (fine-grained device memory, mapped by both neighbours); one 64-byte line per flag:
//   READY_FROM_SOUTH / _NORTH  written by that neighbour: 2 e + w = "exchange e may be stored into my lattice w"
//                              (its kernels that read the ghost rows of exchange e - 1 are complete)
//   DATA_FROM_SOUTH / _NORTH   written by that neighbour: e = "my rows of exchange e are in your ghost rows"
//   ERR                        local: a wait gave up (lb_sync reports it)
//   COUNT, WHICH_S, WHICH_N    local: the exchange counter; the lattice index each neighbour announced for this exchange
// An exchange on the edge stream: k_peer_pre (announce + wait for the neighbours' announcements), k_halo_push (the
// stores), k_peer_post (publish + wait for the neighbours' rows).  A rank signals before it waits, in both kernels, and every
// rank runs the same sequence of exchanges, so nobody waits for somebody who waits for him.  The bulk rows are ordinary
// stores made visible by the end of k_halo_push (stream order, kernel-boundary release) before k_peer_post publishes
// them; the kernels that read them start after k_peer_post has seen the flag.  Counters live on the device: the kernel
// arguments of a cycle never change, so a captured cycle can be replayed.
enum { PEER_READY_FROM_SOUTH = 0, PEER_READY_FROM_NORTH = 8, PEER_DATA_FROM_SOUTH = 16, PEER_DATA_FROM_NORTH = 24, PEER_ERR = 32,
       PEER_COUNT = 40, PEER_WHICH_S = 48, PEER_WHICH_N = 56, PEER_FLAG_WORDS = 64 };      // (uint64 indices: 64 bytes apart)

struct PeerArgs {
    unsigned long long *mine, *south, *north;      // flag blocks: my own, my neighbours' (nullptr = wall)
    unsigned long long timeout_ticks;              // of the 100 MHz clock
    int which;                                     // pre: the lattice of mine that receives this exchange
};

__device__ __forceinline__ void peer_signal(unsigned long long *flag, unsigned long long v)
{
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// spin until *flag >= want (shifted right by `shift` first); returns the value seen, or 0 after the timeout
__device__ __forceinline__ unsigned long long peer_wait(unsigned long long *flag, unsigned long long want, int shift,
                                                        unsigned long long timeout_ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned long long v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((v >> shift) >= want) return v;
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) return 0;
        __builtin_amdgcn_s_sleep(16);
    }
}

__global__ void k_peer_pre(const PeerArgs p)
{
    if (threadIdx.x != 0) return;
    const unsigned long long e = p.mine[PEER_COUNT] + 1;
    p.mine[PEER_COUNT] = e;
    const unsigned long long v = 2 * e + (unsigned)p.which;
    if (p.south) peer_signal(p.south + PEER_READY_FROM_NORTH, v);       // (I am my southern neighbour's northern one)
    if (p.north) peer_signal(p.north + PEER_READY_FROM_SOUTH, v);
    if (p.south) {
        const unsigned long long r = peer_wait(p.mine + PEER_READY_FROM_SOUTH, e, 1, p.timeout_ticks);
        if (!r) p.mine[PEER_ERR] = e;
        p.mine[PEER_WHICH_S] = r & 1;
    }
    if (p.north) {
        const unsigned long long r = peer_wait(p.mine + PEER_READY_FROM_NORTH, e, 1, p.timeout_ticks);
        if (!r) p.mine[PEER_ERR] = e;
        p.mine[PEER_WHICH_N] = r & 1;
    }
}

__global__ void k_peer_post(const PeerArgs p)
{
    if (threadIdx.x != 0) return;
    const unsigned long long e = p.mine[PEER_COUNT];
    __threadfence_system();
    if (p.south) peer_signal(p.south + PEER_DATA_FROM_NORTH, e);
    if (p.north) peer_signal(p.north + PEER_DATA_FROM_SOUTH, e);
    if (p.south && !peer_wait(p.mine + PEER_DATA_FROM_SOUTH, e, 0, p.timeout_ticks)) p.mine[PEER_ERR] = e;
    if (p.north && !peer_wait(p.mine + PEER_DATA_FROM_NORTH, e, 0, p.timeout_ticks)) p.mine[PEER_ERR] = e;
}

// My edge rows -> the neighbours' ghost rows.  Same segment tables as k_halo_pack / k_halo_unpack (entry i of an OUT table
// pairs with entry i of the neighbour's IN table); the destination lattice of each neighbour is the one it announced.
struct PeerDst {
    float *lat[2];          // plane 0, row 0, x 0 of the neighbour's two lattices (mapped into this process); nullptr = wall
    long long plane, rowp;  // its strides
    int h;                  // its height (a southern neighbour's north ghost rows start at its row h)
};
template <int V>
__global__ void k_halo_push(const float *origin, long long plane, int pitch, int h, int nx, const unsigned long long *mine,
                            const PeerDst to_n, const PeerDst to_s, const HaloTable neg, const HaloTable pos)
{
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * V, seg = blockIdx.y, north = (blockIdx.z == 0);
    if (x >= nx) return;
    const PeerDst &d = north ? to_n : to_s;
    if (!d.lat[0]) return;
    const int w = (int)mine[north ? PEER_WHICH_N : PEER_WHICH_S];
    // north edge out = my rows h-D..h-1 (neg, +h) -> its south ghost rows -D..-1 (neg, +0);
    // south edge out = my rows 0..D-1 (pos) -> its north ghost rows (pos, + its h)
    const int k = north ? neg.k[seg] : pos.k[seg];
    const long long rs = north ? h + neg.row[seg] : pos.row[seg];
    const long long rd = north ? neg.row[seg] : d.h + pos.row[seg];
    const float *src = origin + k * plane + rs * pitch + x;
    float *dst = d.lat[w] + k * d.plane + rd * d.rowp + x;
    if (V == 4) *reinterpret_cast<f4a *>(dst) = *reinterpret_cast<const f4a *>(src);
    else *dst = *src;
}

}  // namespace
