// kernels_step5c.h -- the Cython path (cython_dim.pyx Pipe_Flow, kernels_phases.h k1_*) five time steps per pass: k_step5's march
// on overlapping strips with that path's cell functions.  Included by lb_hip.cpp after kernels_phases.h (c1_bcs_cell, c1_moments,
// feq_cell) and kernels_step5.h (whose geometry, windows and hand-over it shares: read that header first).
//
// The path's step is rule -> stream -> moments -> relax (cython_dim.pyx:346-359); as in k1_fstep the rule of step n + 1 rides at the
// end of step n (it is cell-local), so a stage is: restricted pull, moments with their overrides, equilibrium, relaxation, next
// step's rule.  What is new against k_step5 is the RESTRICTED pull (cython_dim.pyx:271-299: the in-place loops leave a link where
// it is when its source lies outside the loop range -- links 1,2,5,6 on row 0, links 3,4,7,8 on row ny-1, links 1,5,4,8 in column 0,
// links 2,6,3,7 in column nx-1): a stage then needs values of the row ITSELF that the marching windows do not hold --
//   * row 0 (the far end of a downward march) / row ny-1 (of an upward one): the row's own from-ahead links.  Nobody will ever pull
//     from that row's from-behind links (the rows beyond a wall do not exist), so when the wall row enters a window its from-ahead
//     links take their place (the `e` part of a register window, the ring slot of an LDS window) and are taken from there, unshifted,
//     one iteration later; the link with cy = 0 that stays (1 on row 0, 3 on row ny-1) is the window's own `d` value, unshifted.
//   * columns 0 / nx-1: one cell per row; its three own values travel in a per-stage delay line (three registers per stage boundary:
//     the lane holding x = 0 keeps links 5,4,8, the lane holding x = nx-1 links 2,6,7); links 1 / 3 are the window's own `d` values.
// Same cell functions as k1_fstep, same expressions: the same bits (tests/test_gpu_cython_path.py).
#pragma once

namespace {

struct Col3 {
    float a, b, c;          // lane holding x = 0: own links 5, 4, 8; lane holding x = nx-1: own links 2, 6, 7
};

struct March5cState {
    Window w1, w2;
    Col3 col[4];            // the wall-column cells' own links after steps 1..4 of the row each window gathers next
    unsigned mhist;
};
struct March5cCtx {
    int lane, x4, ym, n_iter, wy;
    bool store_lane, first, last;       // my lane holds x = 0 / x = nx-1 (in cell jl)
    int jl;
    unsigned slot;
    f4a (*W3)[64], (*W4)[64], (*P3)[64], (*P4)[64];
};

// step 1 of row y from memory: k1_fstep's restricted pull (kernels_phases.h)
template <bool MASK>
__device__ __forceinline__ void c1_row1_load(const StepArgs &a, int y, int x4, bool first, bool last, int jl, Row1 &o)
{
    const int lx = a.nx - 1, ly = a.ny - 1;
    o.have = (y >= 0 && y <= ly);
    o.rr = y;
    o.mk = uc4{0, 0, 0, 0};
    if (!o.have) {
#pragma unroll
        for (int k = 0; k < 9; ++k) o.q[k] = f4a{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const long long S = a.plane, P = a.pitch;
    const bool up = (y >= 1), dn = (y <= ly - 1);              // wave-uniform: links 1,2,5,6 move iff up, 3,4,7,8 iff dn
    const float *r0 = a.src + (long long)y * P;
    const float *rm = up ? r0 - P : r0, *rp = dn ? r0 + P : r0;
    const int su = up ? 1 : 0, sd = dn ? 1 : 0;
    float p1 = 0.f, p5 = 0.f, p4 = 0.f, p8 = 0.f, p2 = 0.f, p6 = 0.f, p3 = 0.f, p7 = 0.f;
    if (first) { p1 = r0[1 * S]; p5 = r0[5 * S]; p4 = r0[4 * S]; p8 = r0[8 * S]; }
    if (last) { p2 = r0[2 * S + lx]; p6 = r0[6 * S + lx]; p3 = r0[3 * S + lx]; p7 = r0[7 * S + lx]; }
    f4a (&q)[9] = o.q;
    q[0] = load4<false>(lane_ptr(r0, x4));
    q[1] = load4u<false>(lane_ptr(r0 + 1 * S - su, x4));
    q[5] = load4u<false>(lane_ptr(rm + 5 * S - su, x4));
    q[2] = load4<false>(lane_ptr(rm + 2 * S, x4));
    q[6] = load4u<false>(lane_ptr(rm + 6 * S + su, x4));
    q[4] = load4<false>(lane_ptr(rp + 4 * S, x4));
    q[8] = load4u<false>(lane_ptr(rp + 8 * S - sd, x4));
    q[3] = load4u<false>(lane_ptr(r0 + 3 * S + sd, x4));
    q[7] = load4u<false>(lane_ptr(rp + 7 * S + sd, x4));
    if (MASK) o.mk = *reinterpret_cast<const uc4 *>(lane_ptr(a.mask + (long long)y * a.fpitch, x4));
    if (first) { q[1].x = p1; q[5].x = p5; q[4].x = p4; q[8].x = p8; }
    if (last) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j == jl) { q[2][j] = p2; q[6][j] = p6; q[3][j] = p3; q[7][j] = p7; }
    }
}

// cell j (run-time index) of my four: select chains -- these run in the rare branches only
__device__ __forceinline__ float c1_pick(const f4a &v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
__device__ __forceinline__ void c1_put(f4a &v, int j, float x)
{
    v.x = j == 0 ? x : v.x; v.y = j == 1 ? x : v.y; v.z = j == 2 ? x : v.z; v.w = j == 3 ? x : v.w;
}

// Moments with their overrides, equilibrium, relaxation and -- `rule` -- the next step's rule of my four cells of row y, in place:
// k1_fstep's loop body (kernels_phases.h), arranged as collide_row arranges the OpenCL path's.
//   Straight-line, all four cells as two packed pairs: c1_moments' sums in ITS order (rho = f0 + ... + f8 left to right, the
//   velocity sums as written there), the wall rows' and the solid cells' zero velocity as selects, then equilibrate_t -- the
//   same operations as feq_cell + "f (1 - omega) + omega feq" (held bitwise by the tests).
//   Rare, one copy each: the cell in column 0 / nx-1 again as a scalar cell from its ORIGINAL links (its moments are overridden by
//   the pinned density: c1_moments in full), and the rule -- cell by cell in a wall row, else that one cell -- through c1_bcs_cell.
//   The rule's obstacle swap for solid cells anywhere else: selects on the pairs.
template <bool MASK>
__device__ __forceinline__ void c1_collide_row(const StepArgs &a, int x4, int y, f4a (&q)[9], uc4 mk, bool rule, bool first,
                                               bool last, int jl, f4a &r4, f4a &u4, f4a &v4)
{
    const int lx = a.nx - 1, ly = a.ny - 1;
    const bool wall_row = (y == 0 || y == ly);
    const int jc = first ? 0 : jl;                  // my wall-column cell, if I hold one
    Cell own = {};
    if (first || last)
        own = Cell{c1_pick(q[0], jc), c1_pick(q[1], jc), c1_pick(q[2], jc), c1_pick(q[3], jc), c1_pick(q[4], jc),
                   c1_pick(q[5], jc), c1_pick(q[6], jc), c1_pick(q[7], jc), c1_pick(q[8], jc)};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f2a f[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) f[k] = h ? q[k].zw : q[k].xy;
        f2a rho = f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7] + f[8];
        const f2a inv = lb_rcp(rho);
        f2a ux = (f[1] - f[3] + f[5] - f[6] - f[7] + f[8]) * inv;
        f2a uy = (f[5] + f[2] + f[6] - f[7] - f[4] - f[8]) * inv;
        if (wall_row) { ux = f2a{0.f, 0.f}; uy = f2a{0.f, 0.f}; }
        if (MASK) {
            const bool s0 = mk[2 * h] != 0, s1 = mk[2 * h + 1] != 0;
            ux = f2a{s0 ? 0.f : ux.x, s1 ? 0.f : ux.y};
            uy = f2a{s0 ? 0.f : uy.x, s1 ? 0.f : uy.y};
        }
        {   // feq_cell + "f (1 - omega) + omega feq" on a pair
            const f2a one = lb_splat<f2a>(1.f);
            const f2a usq = lb_fma(ux, ux, uy * uy);
            const f2a base = lb_fma(lb_splat<f2a>(-1.5f), usq, one);
            const f2a keep = lb_splat<f2a>(1.f - a.omega), om = lb_splat<f2a>(a.omega);
            const f2a r0 = (4.f / 9.f) * rho, r1 = (1.f / 9.f) * rho, r2 = (1.f / 36.f) * rho;
            const f2a r13 = 3.f * r1, r23 = 3.f * r2;
            f2a e[9];
            e[0] = r0 * base;
            feq_pair<f2a>(r1, r13, ux, base, e[1], e[3]);
            feq_pair<f2a>(r1, r13, uy, base, e[2], e[4]);
            feq_pair<f2a>(r2, r23, ux + uy, base, e[5], e[7]);
            feq_pair<f2a>(r2, r23, ux - uy, base, e[8], e[6]);
            // (two roundings, no fma: what the compiler makes of k1_fstep's "f (1 - omega) + omega feq" -- held by the bitwise tests)
#pragma unroll
            for (int k = 0; k < 9; ++k) { const f2a a_ = f[k] * keep; const f2a b_ = om * e[k]; f[k] = a_ + b_; }
        }
        if (h) { r4.zw = rho; u4.zw = ux; v4.zw = uy; }
        else   { r4.xy = rho; u4.xy = ux; v4.xy = uy; }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (h) q[k].zw = f[k];
            else q[k].xy = f[k];
        }
    }
    if (first || last) {                            // the inlet / outlet column's cell: pinned density, its own velocity
        const int x = x4 + jc;
        const bool solid = MASK && (jc == 0 ? mk.x : (jc == 1 ? mk.y : (jc == 2 ? mk.z : mk.w))) != 0;
        float rho, ux, uy;
        c1_moments(a, x, y, solid, own.f0, own.f1, own.f2, own.f3, own.f4, own.f5, own.f6, own.f7, own.f8, rho, ux, uy);
        float fe[9];
        feq_cell(rho, ux, uy, fe);
        const float o[9] = {own.f0, own.f1, own.f2, own.f3, own.f4, own.f5, own.f6, own.f7, own.f8};
#pragma unroll
        for (int k = 0; k < 9; ++k) c1_put(q[k], jc, o[k] * (1.f - a.omega) + a.omega * fe[k]);
        c1_put(r4, jc, rho); c1_put(u4, jc, ux); c1_put(v4, jc, uy);
    }
    if (!rule) return;
    if (wall_row || first || last) {
        const int j0 = wall_row ? 0 : jc, j1 = wall_row ? 3 : jc;
#pragma unroll 1
        for (int j = j0; j <= j1; ++j) {
            const int x = x4 + j;
            if (x > lx) break;
            const bool solid = MASK && (j == 0 ? mk.x : (j == 1 ? mk.y : (j == 2 ? mk.z : mk.w))) != 0;
            Cell c = {c1_pick(q[0], j), c1_pick(q[1], j), c1_pick(q[2], j), c1_pick(q[3], j), c1_pick(q[4], j),
                      c1_pick(q[5], j), c1_pick(q[6], j), c1_pick(q[7], j), c1_pick(q[8], j)};
            c1_bcs_cell(a, x, y, c1_pick(u4, j), solid, c);
            c1_put(q[1], j, c.f1); c1_put(q[2], j, c.f2); c1_put(q[3], j, c.f3); c1_put(q[4], j, c.f4);
            c1_put(q[5], j, c.f5); c1_put(q[6], j, c.f6); c1_put(q[7], j, c.f7); c1_put(q[8], j, c.f8);
        }
    }
    if (MASK) {                                     // solid cells that the branch above has not been through: the rule = the swap
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bool s0 = mk[2 * h] != 0, s1 = mk[2 * h + 1] != 0;
            if (wall_row) s0 = s1 = false;
            if (first || last) {
                if (jc == 2 * h) s0 = false;
                if (jc == 2 * h + 1) s1 = false;
            }
            auto swap2 = [&](f4a &p, f4a &o) {
                const f2a pp = h ? p.zw : p.xy, oo = h ? o.zw : o.xy;
                const f2a pn = f2a{s0 ? oo.x : pp.x, s1 ? oo.y : pp.y}, on = f2a{s0 ? pp.x : oo.x, s1 ? pp.y : oo.y};
                if (h) { p.zw = pn; o.zw = on; }
                else { p.xy = pn; o.xy = on; }
            };
            swap2(q[1], q[3]); swap2(q[2], q[4]); swap2(q[5], q[7]); swap2(q[6], q[8]);
        }
    }
}

// the wall-column cells' own links of a row that has just been through a stage (what the next stage's pull leaves in place)
__device__ __forceinline__ Col3 c1_col_of(const f4a (&q)[9], bool first, bool last, int jl)
{
    Col3 c = {0.f, 0.f, 0.f};
    if (first) c = Col3{q[5].x, q[4].x, q[8].x};
    if (last) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j == jl) c = Col3{q[2][j], q[6][j], q[7][j]};
    }
    return c;
}

// The next stage's restricted pull for my four cells of row y: window {d: links 0,1,3 of row y; e: its from-behind links -- or, y a
// far wall row, its from-ahead links --; g: the from-behind links of the row behind}, newest row q (the row ahead), the wall-column
// cells' own links `col`.
template <bool DOWN>
__device__ __forceinline__ void c1_gather(const StepArgs &a, const Window &w, const f4a &eA, const f4a &eB, const f4a &eC,
                                          const f4a (&q)[9], const Col3 &col, int y, bool first, bool last, int jl, f4a (&t)[9])
{
    typedef Dir<DOWN> D;
    const int ly = a.ny - 1;
    const bool far_wall = DOWN ? (y == 0) : (y == ly);     // (wave-uniform) the from-ahead links and one cy = 0 link stay
    t[0] = w.d0;
    t[1] = skirt_left(w.d1);
    t[3] = skirt_right(w.d3);
    if (far_wall) {
        if (DOWN) t[1] = w.d1;                              // row 0: link 1 stays
        else t[3] = w.d3;                                   // row ny-1: link 3 stays
    }
    t[D::A] = w.g2;
    t[D::B] = skirt_left(w.g5);
    t[D::C] = skirt_right(w.g6);
    if (far_wall) {
        t[D::An] = eA; t[D::Bn] = eB; t[D::Cn] = eC;        // (the row's own, put there when it entered the window)
    } else {
        t[D::An] = q[D::An];
        t[D::Cn] = skirt_right(q[D::Cn]);
        t[D::Bn] = skirt_left(q[D::Bn]);
    }
    // columns 0 / nx-1: links 1,5,4,8 / 2,6,3,7 stay
    if (first) { t[1].x = w.d1.x; t[5].x = col.a; t[4].x = col.b; t[8].x = col.c; }
    if (last) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j == jl) { t[3][j] = w.d3[j]; t[2][j] = col.a; t[6][j] = col.b; t[7][j] = col.c; }
    }
}

// a row enters a register window; a far wall row with its from-ahead links in the place of its from-behind ones
template <bool DOWN>
__device__ __forceinline__ void c1_window_push(Window &w, const f4a (&q)[9], bool far_wall)
{
    typedef Dir<DOWN> D;
    w.g2 = w.e2; w.g5 = w.e5; w.g6 = w.e6;
    w.e2 = far_wall ? q[D::An] : q[D::A];
    w.e5 = far_wall ? q[D::Bn] : q[D::B];
    w.e6 = far_wall ? q[D::Cn] : q[D::C];
    w.d0 = q[0]; w.d1 = q[1]; w.d3 = q[3];
}
template <bool DOWN>
__device__ __forceinline__ void c1_lds_window_push(f4a (*W)[64], int lane, int it, const f4a (&q)[9], bool far_wall)
{
    typedef Dir<DOWN> D;
    const int gs = 3 + 3 * (it & 1);
    W[0][lane] = q[0]; W[1][lane] = q[1]; W[2][lane] = q[3];
    W[gs][lane] = far_wall ? q[D::An] : q[D::A];
    W[gs + 1][lane] = far_wall ? q[D::Bn] : q[D::B];
    W[gs + 2][lane] = far_wall ? q[D::Cn] : q[D::C];
}

template <bool MASK, bool MACRO, bool DOWN, int NST>
__device__ __forceinline__ void march5c_iter(const StepArgs &a, const March5cCtx &cx, const int i_, March5cState &st)
{
    const int lane = cx.lane, x4 = cx.x4, jl = cx.jl;
    const bool first = cx.first, last = cx.last;
    const long long S = a.plane;
    const int i = NST < 5 ? NST - 1 : i_;
    const int it = i;
    const int ly = a.ny - 1;
    auto row_at = [&](int p) { return DOWN ? cx.ym - 1 - p : cx.ym + p; };
    auto far_wall = [&](int y) { return DOWN ? (y == 0) : (y == ly); };
    f4a(*W3)[64] = cx.W3;
    f4a(*W4)[64] = cx.W4;
    Window &w1 = st.w1, &w2 = st.w2;

    if (a.prio_turns > 0 && (i & 3) == 0) {
        const unsigned turn = (unsigned)(__builtin_amdgcn_s_memrealtime() >> a.prio_turns) & 1u;
        if (turn == cx.slot) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
    // ---- what the other wave published for "position -1" (k_step5) -------------------------------------------
    if (NST == 2) { w1.g2 = W4[6][lane]; w1.g5 = W4[7][lane]; w1.g6 = W4[8][lane]; }
    if (NST == 3) { w2.g2 = W4[3][lane]; w2.g5 = W4[4][lane]; w2.g6 = W4[5][lane]; }
    // ---- step 1 of position i (from memory) --------------------------------------------------------------------
    Row1 cur;
    const int y1 = row_at(i);
    c1_row1_load<MASK>(a, y1, x4, first, last, jl, cur);
    f4a (&q1)[9] = cur.q;
    f4a r4, u4, v4;
    const uc4 mk = cur.mk;
    if (cur.have) c1_collide_row<MASK>(a, x4, y1, q1, mk, true, first, last, jl, r4, u4, v4);
    if (NST == 1) lds_publish<DOWN>(cx.P4, lane, 6, q1);
    const Col3 c1n = c1_col_of(q1, first, last, jl);
    // ---- step 2 of position i-1 (window 1, registers) ----------------------------------------------------------
    f4a q2[9];
    Col3 c2n = {0.f, 0.f, 0.f};
    if (NST >= 2) {
        const int y2 = row_at(i - 1);
        c1_gather<DOWN>(a, w1, w1.e2, w1.e5, w1.e6, q1, st.col[0], y2, first, last, jl, q2);
        c1_window_push<DOWN>(w1, q1, far_wall(y1));
        c1_collide_row<MASK>(a, x4, y2, q2, mask_bits(st.mhist, 1), true, first, last, jl, r4, u4, v4);
        if (NST == 2) lds_publish<DOWN>(cx.P4, lane, 3, q2);
        c2n = c1_col_of(q2, first, last, jl);
    } else {
        c1_window_push<DOWN>(w1, q1, far_wall(y1));
    }
    st.col[0] = c1n;
    // ---- step 3 of position i-2 (window 2, registers) ----------------------------------------------------------
    f4a q3[9];
    Col3 c3n = {0.f, 0.f, 0.f};
    if (NST >= 3) {
        const int y3 = row_at(i - 2);
        c1_gather<DOWN>(a, w2, w2.e2, w2.e5, w2.e6, q2, st.col[1], y3, first, last, jl, q3);
        c1_window_push<DOWN>(w2, q2, far_wall(row_at(i - 1)));
        c1_collide_row<MASK>(a, x4, y3, q3, mask_bits(st.mhist, 2), true, first, last, jl, r4, u4, v4);
        if (NST == 3) lds_publish<DOWN>(cx.P3, lane, 6, q3);
        c3n = c1_col_of(q3, first, last, jl);
    } else if (NST == 2) {
        c1_window_push<DOWN>(w2, q2, far_wall(row_at(i - 1)));
    }
    if (NST >= 2) st.col[1] = c2n;
    // ---- step 4 of position i-3 (window 3, LDS) ----------------------------------------------------------------
    f4a q4[9];
    Col3 c4n = {0.f, 0.f, 0.f};
    if (NST >= 4) {
        const int y4 = row_at(i - 3);
        Window w3;
        lds_window_load(W3, lane, it, w3);
        f4a eA = w3.g2, eB = w3.g5, eC = w3.g6;
        if (far_wall(y4)) {                          // (its own from-ahead links: the ring slot written one iteration ago)
            const int es = 3 + 3 * ((it & 1) ^ 1);
            eA = W3[es][lane]; eB = W3[es + 1][lane]; eC = W3[es + 2][lane];
        }
        c1_gather<DOWN>(a, w3, eA, eB, eC, q3, st.col[2], y4, first, last, jl, q4);
        c1_lds_window_push<DOWN>(W3, lane, it, q3, far_wall(row_at(i - 2)));
        c1_collide_row<MASK>(a, x4, y4, q4, mask_bits(st.mhist, 3), true, first, last, jl, r4, u4, v4);
        if (NST == 4) lds_publish<DOWN>(cx.P4, lane, 3, q4);
        c4n = c1_col_of(q4, first, last, jl);
    } else if (NST == 3) {
        c1_lds_window_push<DOWN>(W3, lane, it, q3, far_wall(row_at(i - 2)));
    }
    if (NST >= 3) st.col[2] = c3n;
    // ---- step 5 of position i-4 (window 4, LDS), stored --------------------------------------------------------
    if (NST >= 5) {
        const int y5 = row_at(i - 4);
        Window w4;
        lds_window_load(W4, lane, it, w4);
        f4a eA = w4.g2, eB = w4.g5, eC = w4.g6;
        if (far_wall(y5)) {
            const int es = 3 + 3 * ((it & 1) ^ 1);
            eA = W4[es][lane]; eB = W4[es + 1][lane]; eC = W4[es + 2][lane];
        }
        f4a t[9];
        c1_gather<DOWN>(a, w4, eA, eB, eC, q4, st.col[3], y5, first, last, jl, t);
        c1_lds_window_push<DOWN>(W4, lane, it, q4, far_wall(row_at(i - 3)));
        const bool in_grid = (y5 >= 0 && y5 <= ly);
        if (in_grid) c1_collide_row<MASK>(a, x4, y5, t, mask_bits(st.mhist, 4), a.rule_last != 0, first, last, jl, r4, u4, v4);
        if (cx.store_lane && in_grid) {
            float *d = a.dst + (long long)y5 * a.pitch;
            store_row9<false>(a.nts != 0, d, S, x4, t);
            if (MACRO) {
                const long long m = (long long)y5 * a.fpitch;
                store4<false>(lane_ptr(a.rho + m, x4), r4);
                store4<false>(lane_ptr(a.u + m, x4), u4);
                store4<false>(lane_ptr(a.v + m, x4), v4);
            }
        }
    } else if (NST == 4) {
        c1_lds_window_push<DOWN>(W4, lane, it, q4, far_wall(row_at(i - 3)));
    }
    if (NST >= 4) st.col[3] = c4n;
    if (MASK) st.mhist = ((st.mhist | mask_word(mk)) << 1) & 0x1e1e1e1eu;
    if (NST < 5) __syncthreads();
}

template <bool MASK, bool MACRO, bool DOWN>
__device__ __forceinline__ void march5c(const StepArgs &a, const int x0, const int ym, const int len, const int wy,
                                        f4a (*lds_win)[2][9][64], const unsigned slot)
{
    March5cCtx cx;
    cx.lane = threadIdx.x;
    const int xr = x0 + cx.lane * 4;
    const int lx = a.nx - 1;
    cx.x4 = min(max(xr, 0), lx & ~3);                // lanes beyond the box: copies of the lane at that end
    cx.store_lane = cx.lane >= STEP5_SKIRT / 4 && cx.lane <= 63 - STEP5_SKIRT / 4 && xr < a.nx;
    cx.first = (cx.x4 == 0);
    cx.last = (cx.x4 <= lx && lx < cx.x4 + 4);
    cx.jl = lx & 3;
    cx.ym = ym; cx.n_iter = len + 4; cx.wy = wy; cx.slot = slot;
    cx.W3 = lds_win[wy][0];
    cx.W4 = lds_win[wy][1];
    cx.P3 = lds_win[wy ^ 1][0];
    cx.P4 = lds_win[wy ^ 1][1];
    March5cState st = {};
    march5c_iter<MASK, MACRO, DOWN, 1>(a, cx, 0, st);
    march5c_iter<MASK, MACRO, DOWN, 2>(a, cx, 1, st);
    march5c_iter<MASK, MACRO, DOWN, 3>(a, cx, 2, st);
    march5c_iter<MASK, MACRO, DOWN, 4>(a, cx, 3, st);
    for (int i = 4; i < cx.n_iter; ++i) march5c_iter<MASK, MACRO, DOWN, 5>(a, cx, i, st);
}

// Launch geometry: k_step5's (segment pairs, XCD-transposed order, shorter segments for the wall-column strips).
template <bool MASK, bool MACRO>
__global__ __launch_bounds__(64 * STEP4_WAVES, 2) void k1_step5(const StepArgs a, int strips, int seg_rows, int nsegs, int row_end)
{
    __shared__ f4a lds_win[STEP4_WAVES][2][9][64];
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int item = xcd_item(blockIdx.x, gridDim.x);
    const unsigned slot = __builtin_amdgcn_s_getreg((4 << 11) | 4) & 1u;
    int sx, sy;
    if (item < strips * nsegs) {
        sx = item % strips;
        sy = item / strips;
    } else {
        if (!a.edge_seg_rows) return;
        const int j = item - strips * nsegs;
        sx = (j & 1) ? strips - 1 : 0;
        sy = nsegs + (j >> 1);
    }
    int stride = a.seg_stride;
    if (a.edge_seg_rows && (sx == 0 || sx == strips - 1)) stride = seg_rows = a.edge_seg_rows;
    const int ya = a.row_begin + sy * stride;
    if (ya >= row_end) return;
    const int yb = min(ya + seg_rows, row_end);
    const int ym = ya + (yb - ya) / 2;
    const int x0 = sx * STEP5_VALID - STEP5_SKIRT;
    if (wy == 0) march5c<MASK, MACRO, true>(a, x0, ym, ym - ya, 0, lds_win, slot);
    else march5c<MASK, MACRO, false>(a, x0, ym, yb - ym, 1, lds_win, slot);
}

}  // namespace
