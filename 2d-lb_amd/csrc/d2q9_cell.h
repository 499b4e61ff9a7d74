// d2q9_cell.h -- device-side arithmetic of one D2Q9 cell (boundary rules, obstacle swap, moments,
// equilibrium, BGK relaxation) and the vector load/store helpers shared by every kernel.
// Included by lb_hip.cpp only (one translation unit); see the header comment there for the
// reference lines each function follows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lb_hip.h"

namespace {

typedef float f4a __attribute__((ext_vector_type(4)));              // 16-byte aligned
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // dword aligned
typedef unsigned char uc4 __attribute__((ext_vector_type(4)));

// One cell's nine populations as named scalars (never an indexable array: the boundary rules
// below assign different members on different branches, and an array would be demoted to scratch).
struct Cell {
    float f0, f1, f2, f3, f4, f5, f6, f7, f8;
};

// D2Q9.cl:173-261 (move_bcs) for one cell, float arithmetic (the reference's double literals
// are not mimicked: SURVEY.md Appendix D.1 measured that difference at <= 1e-6 over 5000 steps).
// (w, e, s, n: the cell lies in column 0 / nx-1, row 0 / ny-1)
__device__ __forceinline__ void bc_pipe_cell(Cell &c, bool w, bool e, bool s, bool n, float rin, float rout)
{
    const float f0 = c.f0, f1 = c.f1, f2 = c.f2, f3 = c.f3, f4 = c.f4, f5 = c.f5, f6 = c.f6, f7 = c.f7, f8 = c.f8;
    if (w && !s && !n) {                                   // inlet :198-203
        const float uu = -((f0 + f2 + 2.f * f3 + f4 + 2.f * f6 + 2.f * f7 - rin) / rin);
        const float a = (1.f / 6.f) * uu * rin;
        c.f1 = f3 + (2.f / 3.f) * rin * uu;
        c.f5 = -.5f * f2 + .5f * f4 + f7 + a;
        c.f8 = .5f * f2 - .5f * f4 + f6 + a;
    }
    if (e && !s && !n) {                                   // outlet :205-210
        const float uu = -1.f + (f0 + 2.f * f1 + f2 + f4 + 2.f * f5 + 2.f * f8) / rout;
        const float a = (1.f / 6.f) * uu * rout;
        c.f3 = f1 - (2.f / 3.f) * rout * uu;
        c.f6 = -.5f * f2 + .5f * f4 + f8 - a;
        c.f7 = .5f * f2 - .5f * f4 + f5 - a;
    }
    if (n && !w && !e) {                                   // north wall :213-217
        c.f4 = f2;
        c.f8 = .5f * (-f1 + f3 + 2.f * f6);
        c.f7 = .5f * (f1 - f3 + 2.f * f5);
    }
    if (s && !w && !e) {                                   // south wall :219-223
        c.f2 = f4;
        c.f6 = .5f * (f1 - f3 + 2.f * f8);
        c.f5 = .5f * (-f1 + f3 + 2.f * f7);
    }
    if (w && s) {                                          // corners :228-259
        const float t = .5f * (-f0 - 2.f * f3 - 2.f * f4 - 2.f * f7 + rin);
        c.f1 = f3; c.f2 = f4; c.f5 = f7; c.f6 = t; c.f8 = t;
    }
    if (w && n) {
        const float t = .5f * (-f0 - 2.f * f2 - 2.f * f3 - 2.f * f6 + rin);
        c.f1 = f3; c.f4 = f2; c.f8 = f6; c.f5 = t; c.f7 = t;
    }
    if (e && s) {
        const float t = .5f * (-f0 - 2.f * f1 - 2.f * f4 - 2.f * f8 + rout);
        c.f3 = f1; c.f2 = f4; c.f6 = f8; c.f5 = t; c.f7 = t;
    }
    if (e && n) {
        const float t = .5f * (-f0 - 2.f * f1 - 2.f * f2 - 2.f * f5 + rout);
        c.f3 = f1; c.f4 = f2; c.f7 = f5; c.f6 = t; c.f8 = t;
    }
}

// Build-defined lid-driven cavity closure, oracle/d2q9_oracle.c o2_bc_cavity.
__device__ __forceinline__ void bc_cavity_cell(Cell &c, bool w, bool e, bool s, bool n, float lid, float rho0)
{
    const float f0 = c.f0, f1 = c.f1, f2 = c.f2, f3 = c.f3, f4 = c.f4, f5 = c.f5, f6 = c.f6, f7 = c.f7, f8 = c.f8;
    if (n && !w && !e) {
        const float rw = f0 + f1 + f3 + 2.f * (f2 + f5 + f6);
        c.f4 = f2;
        c.f7 = 0.5f * (f1 - f3 + 2.f * f5) - 0.5f * rw * lid;
        c.f8 = 0.5f * (-f1 + f3 + 2.f * f6) + 0.5f * rw * lid;
    }
    if (s && !w && !e) {
        c.f2 = f4;
        c.f6 = 0.5f * (f1 - f3 + 2.f * f8);
        c.f5 = 0.5f * (-f1 + f3 + 2.f * f7);
    }
    if (w && !s && !n) {
        c.f1 = f3;
        c.f5 = 0.5f * (-f2 + f4 + 2.f * f7);
        c.f8 = 0.5f * (f2 - f4 + 2.f * f6);
    }
    if (e && !s && !n) {
        c.f3 = f1;
        c.f6 = 0.5f * (-f2 + f4 + 2.f * f8);
        c.f7 = 0.5f * (f2 - f4 + 2.f * f5);
    }
    if (w && s) {
        const float t = 0.5f * (-f0 - 2.f * f3 - 2.f * f4 - 2.f * f7 + rho0);
        c.f1 = f3; c.f2 = f4; c.f5 = f7; c.f6 = t; c.f8 = t;
    }
    if (w && n) {
        const float t = 0.5f * (-f0 - 2.f * f2 - 2.f * f3 - 2.f * f6 + rho0);
        c.f1 = f3; c.f4 = f2; c.f8 = f6; c.f5 = t; c.f7 = t;
    }
    if (e && s) {
        const float t = 0.5f * (-f0 - 2.f * f1 - 2.f * f4 - 2.f * f8 + rho0);
        c.f3 = f1; c.f2 = f4; c.f6 = f8; c.f5 = t; c.f7 = t;
    }
    if (e && n) {
        const float t = 0.5f * (-f0 - 2.f * f1 - 2.f * f2 - 2.f * f5 + rho0);
        c.f3 = f1; c.f4 = f2; c.f7 = f5; c.f6 = t; c.f8 = t;
    }
}

// D2Q9.cl:398-433 (bounceback_in_obstacle): exchange opposite links on a solid cell.
__device__ __forceinline__ void bounce_cell(Cell &c, bool solid)
{
    const float f1 = c.f1, f2 = c.f2, f3 = c.f3, f4 = c.f4, f5 = c.f5, f6 = c.f6, f7 = c.f7, f8 = c.f8;
    c.f1 = solid ? f3 : f1; c.f3 = solid ? f1 : f3;
    c.f2 = solid ? f4 : f2; c.f4 = solid ? f2 : f4;
    c.f5 = solid ? f7 : f5; c.f7 = solid ? f5 : f7;
    c.f6 = solid ? f8 : f6; c.f8 = solid ? f6 : f8;
}

// Moments (D2Q9.cl:92-97), equilibrium (:55-60) and BGK relaxation (:119) of one cell.
// The products keep the reference's structure -- feq_k = (w_k rho) * inner_k with the float32
// weights, then f (1-omega) + omega feq -- because the rounding of the weights is a *systematic*
// mass bias (sum_k fl(w_k) = 1 + 7.5e-9); folding omega into the weights would change that bias
// and make rho drift away from the reference's by ~3e-8 per step in a periodic box.
// The arithmetic is written once for a scalar cell (T = float: halo cells, tiles, phase kernels) and for a PAIR of
// x-adjacent cells (T = f2a: the four cells a lane of the row kernels holds are two such pairs, which live in the aligned
// 64-bit register pairs of the 16-byte loads -- every operation below is then one v_pk_*_f32, with no component shuffled
// into place first: left to the SLP vectorizer, the scalar form cost ~30 v_mov per row and stage to re-pair its operands).
// Same operations, same order, explicit fma: the two instantiations round alike, so a cell gets the same bits whichever
// form (and whichever kernel) computes it.
typedef float f2a __attribute__((ext_vector_type(2)));
#ifndef LB_RELAX_FOLD
#define LB_RELAX_FOLD 1            // (0: omega * feq_k as nine products of their own -- rounds 2-5; A/B builds)
#endif
__device__ __forceinline__ float lb_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ f2a lb_rcp(f2a x) { return f2a{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ float lb_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f2a lb_fma(f2a a, f2a b, f2a c) { return __builtin_elementwise_fma(a, b, c); }
template <typename T>
__device__ __forceinline__ T lb_splat(float x);
template <>
__device__ __forceinline__ float lb_splat<float>(float x) { return x; }
template <>
__device__ __forceinline__ f2a lb_splat<f2a>(float x) { return f2a{x, x}; }

template <typename T>
__device__ __forceinline__ void moments_t(T f0, T f1, T f2, T f3, T f4, T f5, T f6, T f7, T f8, T &rho, T &ux, T &uy)
{
    rho = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + f8;
    // v_rcp_f32 (1 ulp) instead of the ten-instruction IEEE division: the three- and four-step kernels are
    // bound by vector-ALU issue, and OpenCL's own '/' (D2Q9.cl:95-96) is only specified to 2.5 ulp
    const T inv = lb_rcp(rho);
    // the two diagonal differences are shared by both components: (f1-f3) + (f5-f7) + (f8-f6) and (f2-f4) + (f5-f7) - (f8-f6)
    // -- 8 additions instead of 10; the reference's left-to-right sums (D2Q9.cl:95-96) differ from this by rounding only
    const T a = f5 - f7, b = f8 - f6;
    ux = ((f1 - f3) + a + b) * inv;
    uy = ((f2 - f4) + a - b) * inv;
}

// Equilibrium of the two links +-c of one direction, weight class r = fl(w rho): feq = r (base + 4.5 cu^2 +- 3 cu) as
//   sh = fma(4.5 cu, cu, base);  rs = r sh;  feq+- = fma(+-3r, cu, rs)
// -- five operations for the pair where r * (base +- 3 cu + 4.5 cu^2) takes eight.  Every product still carries the
// float32 weight through r (the systematic bias of sum_k fl(w_k), see above, is the reference's); measured on the mass
// drift of a periodic box (test_periodic_mass_drift_tracks_reference): +8.1e-9 per step against the reference's +8.9e-9,
// the previous form +8.0e-9; with omega folded into r as well it is +3.5e-9 -- which is why the relaxation below keeps
// omega * feq as a product of its own.
template <typename T>
__device__ __forceinline__ void feq_pair(T r, T r3, T cu, T base, T &fp, T &fm)
{
    const T sh = lb_fma(4.5f * cu, cu, base);
    const T rs = r * sh;
    fp = lb_fma(r3, cu, rs);
    fm = lb_fma(-r3, cu, rs);
}

template <typename T>
__device__ __forceinline__ void equilibrate_t(T &f0, T &f1, T &f2, T &f3, T &f4, T &f5, T &f6, T &f7, T &f8, float omega, T rho,
                                              T ux, T uy)
{
    const T one = lb_splat<T>(1.f);
    const T usq = lb_fma(ux, ux, uy * uy);
    const T base = lb_fma(lb_splat<T>(-1.5f), usq, one);
    const T keep = lb_splat<T>(1.f - omega);
#if LB_RELAX_FOLD
    // omega enters ONCE, through the density: rw = omega rho, r_k = fl(w_k) rw, and every equilibrium below comes out as
    // omega feq_k -- 61 operations per cell update instead of 69 (the nine products omega * feq_k become one).  The float32
    // weights still multiply a run-time value, so their systematic bias (see above) is the reference's; the one extra rounding,
    // omega * rho, depends on the cell's density and averages out (measured on the mass drift of a periodic box,
    // test_periodic_mass_drift_tracks_reference / profiles/r06_experiments.txt section 1).
    const T rw = lb_splat<T>(omega) * rho;
    const T r0 = (4.f / 9.f) * rw, r1 = (1.f / 9.f) * rw, r2 = (1.f / 36.f) * rw;
    const T r13 = 3.f * r1, r23 = 3.f * r2;
    T e1, e2, e3, e4, e5, e6, e7, e8;
    feq_pair<T>(r1, r13, ux, base, e1, e3);
    feq_pair<T>(r1, r13, uy, base, e2, e4);
    feq_pair<T>(r2, r23, ux + uy, base, e5, e7);
    feq_pair<T>(r2, r23, ux - uy, base, e8, e6);
    f0 = lb_fma(f0, keep, r0 * base);
    f1 = lb_fma(f1, keep, e1);
    f3 = lb_fma(f3, keep, e3);
    f2 = lb_fma(f2, keep, e2);
    f4 = lb_fma(f4, keep, e4);
    f5 = lb_fma(f5, keep, e5);
    f7 = lb_fma(f7, keep, e7);
    f8 = lb_fma(f8, keep, e8);
    f6 = lb_fma(f6, keep, e6);
#else
    const T om = lb_splat<T>(omega);
    const T r0 = (4.f / 9.f) * rho, r1 = (1.f / 9.f) * rho, r2 = (1.f / 36.f) * rho;
    const T r13 = 3.f * r1, r23 = 3.f * r2;
    T e1, e2, e3, e4, e5, e6, e7, e8;
    feq_pair<T>(r1, r13, ux, base, e1, e3);
    feq_pair<T>(r1, r13, uy, base, e2, e4);
    feq_pair<T>(r2, r23, ux + uy, base, e5, e7);
    feq_pair<T>(r2, r23, ux - uy, base, e8, e6);
    f0 = lb_fma(f0, keep, om * (r0 * base));
    f1 = lb_fma(f1, keep, om * e1);
    f3 = lb_fma(f3, keep, om * e3);
    f2 = lb_fma(f2, keep, om * e2);
    f4 = lb_fma(f4, keep, om * e4);
    f5 = lb_fma(f5, keep, om * e5);
    f7 = lb_fma(f7, keep, om * e7);
    f8 = lb_fma(f8, keep, om * e8);
    f6 = lb_fma(f6, keep, om * e6);
#endif
}

__device__ __forceinline__ void moments_cell(const Cell &c, float &rho, float &ux, float &uy)
{
    moments_t<float>(c.f0, c.f1, c.f2, c.f3, c.f4, c.f5, c.f6, c.f7, c.f8, rho, ux, uy);
}

__device__ __forceinline__ void equilibrate_cell(Cell &c, float omega, float rho, float ux, float uy)
{
    equilibrate_t<float>(c.f0, c.f1, c.f2, c.f3, c.f4, c.f5, c.f6, c.f7, c.f8, omega, rho, ux, uy);
}

__device__ __forceinline__ void relax_cell(Cell &c, float omega, float &rho, float &ux, float &uy)
{
    moments_cell(c, rho, ux, uy);
    equilibrate_cell(c, omega, rho, ux, uy);
}

// ---- the reference's second rule set: imposed-speed inlet / outlet, north and south rows sharing their vertical
// links (D2Q9.cl:263-374, `move_bcs_PeriodicBC_VelocityInlet` / `update_hydro_PeriodicBC_VelocityInlet`) -----------
// In pull form.  The reference's push `move` never writes the links that would enter from outside the box; after
// `copy_buffer` they hold whatever f_streamed held there ("stale").  The rules then set, with the post-stream values:
//   inlet (x = 0, 1 <= y <= ny-2): f1, f5, f8 from the imposed speed;  outlet likewise;
//   north row: f4, f8, f7 := the values ROW 0 received (streamed out of row 1);  south row: f2, f6, f5 := the values
//   row ny-1 received (out of row ny-2) -- i.e. the pull of a wall row reads row 1 / ny-2 in place of the row outside
//   (the kernels' source-row mapping, `wrap_y == 2`), except where THAT link itself entered from outside:
// which leaves exactly eight links of the four corner cells that nothing ever writes.  They keep the values they had
// when the populations were last set (lb_set_f / lb_init_pop copy them into `corner`):
//   corner[0] = f1(0,0)  [1] = f8(0,0)  [2] = f1(0,ny-1)  [3] = f5(0,ny-1)
//   corner[4] = f3(nx-1,0)  [5] = f7(nx-1,0)  [6] = f3(nx-1,ny-1)  [7] = f6(nx-1,ny-1)
__device__ __forceinline__ void bc_vel_cell(Cell &c, bool w, bool e, bool s, bool n, float u_w, float u_e,
                                            const float *corner)
{
    const float f0 = c.f0, f1 = c.f1, f2 = c.f2, f3 = c.f3, f4 = c.f4, f5 = c.f5, f6 = c.f6, f7 = c.f7, f8 = c.f8;
    if (w && !s && !n) {                                   // inlet :290-295
        const float rho_w = (1.f / (1.f - u_w)) * (f0 + f2 + f4 + 2.f * (f3 + f6 + f7));
        const float h = 0.5f * (f2 - f4), t = (1.f / 6.f) * rho_w * u_w;
        c.f1 = f3 + (2.f / 3.f) * rho_w * u_w;
        c.f5 = f7 - h + t;
        c.f8 = f6 + h + t;
    }
    if (e && !s && !n) {                                   // outlet :297-302
        const float rho_e = (1.f / (1.f + u_e)) * (f0 + f2 + f4 + 2.f * (f1 + f5 + f8));
        const float h = 0.5f * (f2 - f4), t = (1.f / 6.f) * rho_e * u_e;
        c.f3 = f1 - (2.f / 3.f) * rho_e * u_e;
        c.f6 = f5 + h - t;
        c.f7 = f8 - h - t;
    }
    if (w && s) { c.f1 = corner[0]; c.f8 = corner[1]; c.f5 = corner[3]; }
    if (w && n) { c.f1 = corner[2]; c.f5 = corner[3]; c.f8 = corner[1]; }
    if (e && s) { c.f3 = corner[4]; c.f7 = corner[5]; c.f6 = corner[7]; }
    if (e && n) { c.f3 = corner[6]; c.f6 = corner[7]; c.f7 = corner[5]; }
}

// D2Q9.cl:323-374: on the inlet / outlet columns the moments are not the plain sums.  rho: the rule's density on
// the inlet / outlet cells proper (not the corners); u: the imposed speed there, and whatever the u array holds on
// the corner cells (the kernel never writes it); v: whatever the v array holds, on the whole column (never written).
// (c: the cell after the boundary rule and the obstacle swap; u_prev, v_prev: the stored fields at this cell.)
__device__ __forceinline__ void vel_moments_cell(const Cell &c, bool w, bool s, bool n, float u_w, float u_e, float u_prev,
                                                 float v_prev, float &rho, float &ux, float &uy)
{
    ux = u_prev;
    uy = v_prev;
    if (!s && !n) {
        if (w) { rho = (1.f / (1.f - u_w)) * (c.f0 + c.f2 + c.f4 + 2.f * (c.f3 + c.f6 + c.f7)); ux = u_w; }
        else   { rho = (1.f / (1.f + u_e)) * (c.f0 + c.f2 + c.f4 + 2.f * (c.f1 + c.f5 + c.f8)); ux = u_e; }
    }
}

// ---- the reference's "incompressible" fork, LB_D2Q9/D2Q9i.cl (host: dimensionless/opencl_dim_D2Q9i.py) -------------
// Three routines differ from D2Q9.cl: the inlet / outlet of `move_bcs` (D2Q9i.cl:194-205), `update_hydro` (:90-94:
// momentum, not divided by rho) and `update_feq` (:58: inner = rho + 3 cu + 4.5 cu^2 - 1.5 usq, still multiplied by
// w rho).  Restated as the fork has them -- executed faithfully it is unstable (|u| grows ~10x in ten steps from a 2e-4
// density drop, tests/golden/o2_d2q9i_53x27) --; float arithmetic where the fork's literals are double.
__device__ __forceinline__ void bc_pipe_i_cell(Cell &c, bool w, bool e, bool s, bool n, float rin, float rout)
{
    const float f0 = c.f0, f1 = c.f1, f2 = c.f2, f3 = c.f3, f4 = c.f4, f5 = c.f5, f6 = c.f6, f7 = c.f7, f8 = c.f8;
    if (w && !s && !n) {                                   // inlet :194-198
        const float uu = -f0 - f2 - 2.f * f3 - f4 - 2.f * f6 - 2.f * f7 + rin;
        c.f1 = (1.f / 3.f) * (3.f * f3 + 2.f * uu);
        c.f5 = (1.f / 6.f) * (-3.f * f2 + 3.f * f4 + 6.f * f7 + uu);
        c.f8 = (1.f / 6.f) * (3.f * f2 - 3.f * f4 + 6.f * f6 + uu);
    } else if (e && !s && !n) {                            // outlet :201-205
        const float uu = f0 + 2.f * f1 + f2 + f4 + 2.f * f5 + 2.f * f8 - rout;
        c.f3 = (1.f / 3.f) * (3.f * f1 - 2.f * uu);
        c.f6 = (1.f / 6.f) * (-3.f * f2 + 3.f * f4 + 6.f * f8 - uu);
        c.f7 = (1.f / 6.f) * (3.f * f2 - 3.f * f4 + 6.f * f5 - uu);
    } else {
        bc_pipe_cell(c, w, e, s, n, rin, rout);            // walls and corners: as D2Q9.cl
    }
}

__device__ __forceinline__ void moments_i_cell(const Cell &c, float &rho, float &ux, float &uy)
{
    rho = c.f0 + c.f1 + c.f2 + c.f3 + c.f4 + c.f5 + c.f6 + c.f7 + c.f8;
    ux = (c.f1 + c.f5 + c.f8 - c.f6 - c.f3 - c.f7);
    uy = (c.f6 + c.f2 + c.f5 - c.f7 - c.f4 - c.f8);
}

__device__ __forceinline__ void equilibrate_i_cell(Cell &c, float omega, float rho, float ux, float uy)
{
    const float usq = ux * ux + uy * uy;
    const float base = rho - 1.5f * usq;
    const float keep = 1.f - omega;
    const float r0 = (4.f / 9.f) * rho, r1 = (1.f / 9.f) * rho, r2 = (1.f / 36.f) * rho;
    c.f0 = c.f0 * keep + omega * (r0 * base);
    c.f1 = c.f1 * keep + omega * (r1 * (base + 3.f * ux + 4.5f * ux * ux));
    c.f3 = c.f3 * keep + omega * (r1 * (base - 3.f * ux + 4.5f * ux * ux));
    c.f2 = c.f2 * keep + omega * (r1 * (base + 3.f * uy + 4.5f * uy * uy));
    c.f4 = c.f4 * keep + omega * (r1 * (base - 3.f * uy + 4.5f * uy * uy));
    const float p = ux + uy, m = ux - uy;
    c.f5 = c.f5 * keep + omega * (r2 * (base + 3.f * p + 4.5f * p * p));
    c.f7 = c.f7 * keep + omega * (r2 * (base - 3.f * p + 4.5f * p * p));
    c.f8 = c.f8 * keep + omega * (r2 * (base + 3.f * m + 4.5f * m * m));
    c.f6 = c.f6 * keep + omega * (r2 * (base - 3.f * m + 4.5f * m * m));
}

template <bool NT>
__device__ __forceinline__ f4a load4(const float *p)
{
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4a *>(p));
    return *reinterpret_cast<const f4a *>(p);
}
template <bool NT>
__device__ __forceinline__ f4a load4u(const float *p)
{
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4u *>(p));
    return *reinterpret_cast<const f4u *>(p);
}
template <bool NT>
__device__ __forceinline__ void store4(float *p, f4a v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4a *>(p));
    else *reinterpret_cast<f4a *>(p) = v;
}

}  // namespace
