// kernels_step4.h -- four time steps per pass.  Included by lb_hip.cpp after kernels_fused.h.
//
// The march of k_step3 with one more stage.  Per iteration i a wave loads one row and runs step 1 on it, step 2 on the
// row loaded one iteration earlier, step 3 on the one before, step 4 on the one before that (stored).  The window
// between steps 1 and 2 stays in registers; the windows between steps 2/3 and 3/4 live in LDS, wave-private (9 slots of
// 64 x 16 B each: links 0,1,3 of the previous row, the three links pulled from behind of the previous two rows in a
// 2-deep ring), which is what keeps the kernel at two waves per SIMD -- an LDS-resident window costs ~1 % against
// registers (measured on k_step3, profiles/r01_ablation.txt).
//
// Segment pairs (round 3).  A wave that starts a segment cold must run step 1 on three rows, step 2 on two and step 3
// on one row BELOW its first output row before step 4 has anything to consume (and the same above its last row): six
// redundant rows per segment, +37 % rows read on the 16-row segments of an 8-GPU slab.  Now the two waves of a
// workgroup share one strip and start back to back at the middle row of a PAIR of segments, one marching up, the other
// down.  What the upward wave needs from below its first row -- steps 1, 2, 3 of the row just under it -- is exactly
// what the downward wave computes in its first three iterations as useful work, and vice versa: each publishes the three
// links that cross the middle line (and its halo cells' share) into the other's window slots, three workgroup barriers
// in the first three iterations, none afterwards.  Redundant rows per segment: three (beyond its far end) instead of six,
// and those are rows the neighbouring pair's wave reads at about the same time.  The two directions are two
// instantiations of one body (template DOWN): link roles are by direction of travel relative to the march ("pulled
// from behind" = cy = +1 links 2,5,6 marching up, cy = -1 links 4,8,7 marching down), cell arithmetic always sees the
// physical links, so results stay bitwise those of k_step.
//
// The cells beyond the strip are recomputed as scalar cells, as in k_step3, one more ring per stage: step 1
// for x0-3..x0-1 | x0+256..x0+258, step 2 for the inner two, step 3 for the innermost -- here by six
// "halo lanes", one cell each (see march4).  A halo cell at distance d takes its centre links from itself, the links moving toward the
// strip from the cell at d+1, the links moving away from it from the cell at d-1 (d = 1: the strip's own
// edge cell, out of the vector registers).
//
// HBM traffic per four updates of a cell: 9 reads + 9 writes (+3 rows per segment): ~18.5 B per lattice update.
#pragma once

namespace {

// Link roles relative to the direction of the march.  A, B, C: the links a row pulls from the row BEHIND it (vertical,
// from the left = cx +1, from the right = cx -1); An, Bn, Cn: those it pulls from the row AHEAD.
template <bool DOWN>
struct Dir {
    static constexpr int A = DOWN ? 4 : 2, B = DOWN ? 8 : 5, C = DOWN ? 7 : 6;
    static constexpr int An = DOWN ? 2 : 4, Bn = DOWN ? 5 : 8, Cn = DOWN ? 6 : 7;
};
template <int K>
__device__ __forceinline__ float cget(const Cell &c)
{
    if constexpr (K == 0) return c.f0;
    else if constexpr (K == 1) return c.f1;
    else if constexpr (K == 2) return c.f2;
    else if constexpr (K == 3) return c.f3;
    else if constexpr (K == 4) return c.f4;
    else if constexpr (K == 5) return c.f5;
    else if constexpr (K == 6) return c.f6;
    else if constexpr (K == 7) return c.f7;
    else return c.f8;
}
template <int K>
__device__ __forceinline__ void cset(Cell &c, float v)
{
    if constexpr (K == 0) c.f0 = v;
    else if constexpr (K == 1) c.f1 = v;
    else if constexpr (K == 2) c.f2 = v;
    else if constexpr (K == 3) c.f3 = v;
    else if constexpr (K == 4) c.f4 = v;
    else if constexpr (K == 5) c.f5 = v;
    else if constexpr (K == 6) c.f6 = v;
    else if constexpr (K == 7) c.f7 = v;
    else c.f8 = v;
}

// what later stages can ask of a halo cell: centre links (cx = 0), links moving toward the strip
// (cx = +1 on the left side, cx = -1 on the right) and away from it, each by where it is pulled from:
// the same row (0), the row behind (p), the row ahead (m)
struct HaloCell9 {
    float c0, cp, cm;
    float t0, tp, tm;
    float a0, ap, am;
};
template <bool DOWN>
__device__ __forceinline__ HaloCell9 halo_all(const Cell &c, bool left)
{
    typedef Dir<DOWN> D;
    HaloCell9 h;
    h.c0 = c.f0; h.cp = cget<D::A>(c); h.cm = cget<D::An>(c);
    h.t0 = left ? c.f1 : c.f3; h.tp = left ? cget<D::B>(c) : cget<D::C>(c); h.tm = left ? cget<D::Bn>(c) : cget<D::Cn>(c);
    h.a0 = left ? c.f3 : c.f1; h.ap = left ? cget<D::C>(c) : cget<D::B>(c); h.am = left ? cget<D::Cn>(c) : cget<D::Bn>(c);
    return h;
}
// delay line of one link triple: same-row link of the previous row, from-behind link of the previous two rows
// (the from-ahead link is consumed in the iteration that produces it)
struct Tri {
    float d, e, g;
};
__device__ __forceinline__ void tri_push(Tri &w, float x0, float xp)
{
    w.g = w.e; w.e = xp; w.d = x0;
}

// Next stage of the halo cell at (hx, row yg): pre-collision links from the delay lines, then boundary
// rule, obstacle swap and relaxation as everywhere else.  centre/toward: delay lines + this iteration's
// from-ahead links (cm_new, tm_new); the away links of the closer cell: a0 (same row), ap (from behind), am (from ahead).
template <int BC, bool MASK, bool DOWN>
__device__ __forceinline__ void halo_cell_next(const StepArgs &a, int hx, int yg, bool left, bool solid,
                                               const Tri &centre, float cm_new, const Tri &toward, float tm_new,
                                               float a0, float ap, float am, Cell &c)
{
    typedef Dir<DOWN> D;
    c.f0 = centre.d; cset<D::A>(c, centre.g); cset<D::An>(c, cm_new);
    const float t0 = toward.d, tp = toward.g;
    c.f1 = left ? t0 : a0; c.f3 = left ? a0 : t0;
    cset<D::B>(c, left ? tp : ap); cset<D::C>(c, left ? ap : tp);
    cset<D::Bn>(c, left ? tm_new : am); cset<D::Cn>(c, left ? am : tm_new);
    int xc = hx;
    if (BC == LB_BC_PERIODIC) xc = hx < 0 ? hx + a.nx : (hx >= a.nx ? hx - a.nx : hx);
    else if (hx < 0 || hx >= a.nx) return;              // outside the box: don't-care
    if (BC != LB_BC_PERIODIC) {
        const bool w = (xc == 0), e = (xc == a.nx - 1), so = (yg == 0), no = (yg == a.ny - 1);
        if (w || e || so || no) boundary_rule<BC>(a, c, w, e, so, no);
    }
    float rho, ux, uy;
    finish_cell<BC, MASK>(a, xc, yg - a.y0, c, solid, rho, ux, uy);
}

// register window of one stage: d = links 0,1,3 of the previous row; e / g = links A,B,C of the previous row / the one before
template <bool DOWN>
__device__ __forceinline__ void window_push_dir(Window &w, const f4a (&q)[9])
{
    typedef Dir<DOWN> D;
    w.g2 = w.e2; w.g5 = w.e5; w.g6 = w.e6;
    w.e2 = q[D::A]; w.e5 = q[D::B]; w.e6 = q[D::C];
    w.d0 = q[0]; w.d1 = q[1]; w.d3 = q[3];
}

// the LDS-resident window: slots 0,1,2 = links 0,1,3 of the previous row; 3..5 / 6..8 = links A,B,C of the
// previous two rows (ring, slot chosen by the iteration's parity)
__device__ __forceinline__ void lds_window_load(f4a (*W)[64], int lane, int it, Window &w)
{
    const int gs = 3 + 3 * (it & 1);                    // the older of the two ring rows
    w.d0 = W[0][lane]; w.d1 = W[1][lane]; w.d3 = W[2][lane];
    w.g2 = W[gs][lane]; w.g5 = W[gs + 1][lane]; w.g6 = W[gs + 2][lane];
}
template <bool DOWN>
__device__ __forceinline__ void lds_window_push(f4a (*W)[64], int lane, int it, const f4a (&q)[9])
{
    typedef Dir<DOWN> D;
    const int gs = 3 + 3 * (it & 1);                    // overwrite the row just consumed
    W[0][lane] = q[0]; W[1][lane] = q[1]; W[2][lane] = q[3];
    W[gs][lane] = q[D::A]; W[gs + 1][lane] = q[D::B]; W[gs + 2][lane] = q[D::C];
}
// the three links of row `q` that cross the pair's middle line, into the OTHER wave's window at ring slot `gs` (3 or 6):
// they are what that wave pulls from behind its first row
template <bool DOWN>
__device__ __forceinline__ void lds_publish(f4a (*W)[64], int lane, int gs, const f4a (&q)[9])
{
    typedef Dir<DOWN> D;
    W[gs][lane] = q[D::An]; W[gs + 1][lane] = q[D::Bn]; W[gs + 2][lane] = q[D::Cn];
}

// gather of the next stage for my 4 cells from a window holding {d0,d1,d3,g2,g5,g6}, the newest row q and
// the innermost halo cell's toward links (delay line + this iteration's from-ahead link)
template <bool DOWN>
__device__ __forceinline__ void stage_gather(const Window &w, const f4a (&q)[9], const Tri &ht, float htm_new, int lane,
                                             f4a (&t)[9])
{
    typedef Dir<DOWN> D;
    t[0] = w.d0;
    t[1] = from_left(w.d1, ht.d, lane);
    t[3] = from_right(w.d3, ht.d, lane);
    t[D::A] = w.g2;
    t[D::B] = from_left(w.g5, ht.g, lane);
    t[D::C] = from_right(w.g6, ht.g, lane);
    t[D::An] = q[D::An];
    t[D::Cn] = from_right(q[D::Cn], htm_new, lane);
    t[D::Bn] = from_left(q[D::Bn], htm_new, lane);
}

constexpr int STEP4_WAVES = 2;      // waves per workgroup = the two directions of a segment pair: 2 x 2 windows x 9 KiB of LDS

// what the halo lanes of the two waves hand each other across the middle line: [receiving wave][stage][value][halo lane]
struct HaloXchg {
    float v[STEP4_WAVES][3][3][8];
};

// Everything step 1 of one row takes from memory: the nine gathered planes of my four cells, their obstacle
// flags, and (halo lanes) the raw populations of my halo cell.
struct Row1 {
    f4a q[9];
    WrapPatch wp;       // periodic boxes: the wrap elements of the lanes at x = 0 / nx-1, merged at the point of use
    uc4 mk;
    Cell hc;
    int hxc;            // wrapped column of the halo cell, -1 = outside a walled box
    int rr;             // local row the data belongs to (wrapped where the box is periodic)
    bool hsolid, have;
};

template <int BC, bool MASK>
__device__ __forceinline__ void row1_load(const StepArgs &a, int r, int x4, bool halo1, int hx, Row1 &o)
{
    int ym, yp;
    o.have = step1_rows(a, r, o.rr, ym, yp);
    o.mk = uc4{0, 0, 0, 0};
    o.hsolid = false;
    o.hxc = -1;
    // (o.hc is only read by the halo lanes of a row that was loaded: no zero fill for everybody else -- nine v_mov per row)
    if (o.have) {
        gather_issue<BC, MASK, false>(a, x4, o.rr, ym, yp, o.q, o.mk, o.wp);
#ifdef LB_DIAG
        if (!(a.diag & 262144))                         // timing only: no halo-cell loads at all
#endif
        if (halo1) halo_cell_load<BC, MASK>(a, hx, o.rr, ym, yp, o.hc, o.hsolid, o.hxc);
    } else {
        o.wp = WrapPatch{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 9; ++k) o.q[k] = f4a{0.f, 0.f, 0.f, 0.f};
    }
}

// Halo lanes.  The three cells beyond each end of the strip are spread over six lanes -- lanes 0,1,2 own
// the cells at distance 1,2,3 on the left, lanes 63,62,61 those on the right -- so that one scalar-cell
// call per stage serves all of them at once (six sequential calls by two lanes made the kernel bound by
// vector-ALU issue).  A halo cell takes the links moving toward the strip from the lane that owns the
// cell farther out (`lane_out`) and the links moving away from it from the lane that owns the cell
// closer in (`lane_in`; for distance 1 that is the strip's own edge cell, in the same lane), through
// ds_bpermute, which does not occupy the vector ALU.
//
// One wave's march: strip [x0, x0 + 256), own rows = `len` rows starting at the pair's middle line `ym` and going up
// (rows ym .. ym+len-1) or down (rows ym-1 .. ym-len).  Position p of the march = row ym + p / ym - 1 - p; iteration i
// loads position i and runs step 1 on it, step 2 on position i-1, step 3 on i-2, step 4 on i-3 (stored): len + 3
// iterations.  What the steps of position 0 pull from "position -1" is the other wave's position 0 (see the header).
// (Touching the next row's 90 cache lines a row ahead with two one-lane-per-line loads, instead of the register prefetch
//  PF below, costs more in the texture addresser than the wait it saves: 243 -> 162 k MLUPS at 8192^2,
//  profiles/r02_experiments.txt.)
// Everything one wave's march carries from one iteration to the next (besides the two Row1 buffers of the one-row-ahead gather).
struct March4State {
    Window w1;                                          // stage window between steps 1 and 2 (registers)
    // delay lines of MY halo cell: stage 1 (centre, toward, away), stage 2 (centre, toward), stage 3 (toward)
    Tri s1c, s1t, s1a, s2c, s2t, s3t;
    // obstacle-mask history: my four cells (per byte: bit 1 = the row loaded one iteration ago, bit 2 = two, bit 3 = three
    // ago); my halo cell (bit 1, bit 2 likewise)
    unsigned mhist, hmask;
};
// ... and what stays fixed
struct March4Ctx {
    int lane, x4, hx, hd, lane_out, lane_in, hslot, ym, n_iter, wy;
    bool store_lane, left, halo1, halo2, halo3;
    unsigned slot;
    f4a (*W2)[64], (*W3)[64], (*P2)[64], (*P3)[64];     // my two LDS windows, the other wave's
    HaloXchg *xchg;
};

// One iteration of the march (see march4 below): loads position i (or takes it from `cur`, gathered an iteration ago, and
// gathers position i + 1 into `nxt`), step 1 on it, step 2 on position i - 1, step 3 on i - 2, step 4 on i - 3 (stored).
//
// STEADY = the iterations 3 <= i < len, where every stage has a row, the row loaded is one of the wave's own (inside the
// slab: `have` is true), the other wave has nothing to hand over any more and the next position exists: no condition on i is
// left in the body, so nothing has to be zero-filled "for the branch not taken" and no value is copied into a join register
// (the generic form spends ~110 v_mov per row on those joins: the q2 / q3 arrays of a stage that may not run yet).  march4
// runs the steady iterations in PAIRS with the two Row1 buffers swapping roles (PAR = i & 1 is then a constant: the LDS ring
// slots are immediate offsets), so that "cur = nxt" and the windows' rotation are register names, not moves.
template <int BC, bool MASK, bool MACRO, bool NTS, bool PF, bool DOWN, int NST>
__device__ __forceinline__ void march4_iter(const StepArgs &a, const March4Ctx &cx, const int i_, March4State &st, Row1 &cur,
                                            Row1 &nxt)
{
    typedef Dir<DOWN> D;
    const int lane = cx.lane, x4 = cx.x4;
    const bool left = cx.left, halo1 = cx.halo1, halo2 = cx.halo2, halo3 = cx.halo3;
    const int lane_out = cx.lane_out, lane_in = cx.lane_in, hslot = cx.hslot, wy = cx.wy;
    const long long S = a.plane;
    const int i = NST < 4 ? NST - 1 : i_;               // (the filling iterations: i is a constant)
    const int it = i;
    auto row_at = [&](int p) { return DOWN ? cx.ym - 1 - p : cx.ym + p; };
    f4a(*W2)[64] = cx.W2;
    f4a(*W3)[64] = cx.W3;
    Window &w1 = st.w1;

    // Two waves share a SIMD, and its arbiters serve the OLDER one first: measured per wave (tools/wave_timeline.py), the
    // wave in slot 0 of every SIMD finished its 134 rows after ~945 us, the one in slot 1 after ~1190 us, i.e. for the last
    // fifth of the launch every SIMD ran a single wave.  Priority outranks age, so the two take turns: both read the same
    // 100 MHz clock at the top of every fourth row and the wave whose slot parity matches bit 13 of it (82 us per turn, ten
    // rows or so) raises its priority -- complementary at (almost) all times without the waves knowing of each other.
    if (a.prio_turns > 0 && (i & 3) == 0) {            // (every fourth row: reading the clock drains the wave's LDS queue)
        const unsigned turn = (unsigned)(__builtin_amdgcn_s_memrealtime() >> a.prio_turns) & 1u;
        if (turn == cx.slot) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
    // ---- what the other wave published for "position -1" in its previous iteration ---------------------------
    if (NST >= 2 && i >= 1 && i <= 3) {
        const float *hv = &cx.xchg->v[wy][i - 1][0][hslot];
        if (i == 1) {
            w1.g2 = W3[6][lane]; w1.g5 = W3[7][lane]; w1.g6 = W3[8][lane];     // (W3 is idle until iteration 2: a mailbox)
            if (halo1) { st.s1c.g = hv[0]; st.s1t.g = hv[8]; st.s1a.g = hv[16]; }
        } else if (i == 2) {
            if (halo1) { st.s2c.g = hv[0]; st.s2t.g = hv[8]; }
        } else {
            if (halo1) st.s3t.g = hv[0];
        }
    }
    // ---- step 1 of position i (from memory) --------------------------------------------------------------------
    if (PF) {
        // (always: behind the last position the last row is gathered once more -- a cache hit that nobody consumes -- so that
        //  `nxt` is defined on every path: with "if (i + 1 < n_iter)" the compiler keeps the OLD nxt as the value of the path
        //  not taken, i.e. copies 50 registers into the join registers at the top of every iteration, behind an
        //  s_waitcnt vmcnt(0) that also waits for the previous row's nine stores)
        row1_load<BC, MASK>(a, row_at(min(i + 1, cx.n_iter - 1)), x4, halo1, cx.hx, nxt);
    } else {
        row1_load<BC, MASK>(a, row_at(i), x4, halo1, cx.hx, cur);
    }
    f4a (&q1)[9] = cur.q;
    f4a r4, u4, v4;
    const uc4 mk = cur.mk;
    const bool hsolid = cur.hsolid;
    HaloCell9 n1 = {};                              // stage-1 links of my halo cell at position i
    if (cur.have) {
        gather_merge<BC, true>(a, x4, q1, cur.wp);    // (periodic wrap elements: merged here, not behind the loads; nx % 4 == 0: step4_applicable)
#ifdef LB_DIAG
        if (!(a.diag & 1024))
#endif
        if (halo1) {
            Cell c = cur.hc;
            halo_cell_finish<BC, MASK>(a, cur.hxc, cur.rr, c, hsolid);
            n1 = halo_all<DOWN>(c, left);
        }
#ifdef LB_DIAG
        if (!(a.diag & 1))
#endif
        collide_row<BC, MASK>(a, x4, a.y0 + cur.rr, q1, mk, r4, u4, v4);
    }
    if (NST == 1) {                        // my position 0 after step 1 -> the other wave's register window
        lds_publish<DOWN>(cx.P3, lane, 6, q1);
        if (halo1) {
            float *hv = &cx.xchg->v[wy ^ 1][0][0][hslot];
            hv[0] = n1.cm; hv[8] = n1.tm; hv[16] = n1.am;
        }
    }
    // ---- step 2 of position i-1 (window 1, registers) ----------------------------------------------------------
    f4a q2[9];
    HaloCell9 n2 = {};                              // stage-2 links of my halo cell at position i-1
    if (NST >= 2) {
        int r2, t0_, t1_;
        (void)step1_rows(a, row_at(i - 1), r2, t0_, t1_);
        stage_gather<DOWN>(w1, q1, st.s1t, n1.tm, lane, q2);   // (lanes 0 / 63 own the innermost cells)
        const float w1_d3x = w1.d3.x, w1_d1w = w1.d1.w, w1_g6x = w1.g6.x, w1_g5w = w1.g5.w;   // the halo stage's share
        // every window takes its new row as soon as its old one has been gathered from, not at the end of the
        // iteration: q1 / q2 / q3 (36 registers each) then die here instead of living through the stages below
        window_push_dir<DOWN>(w1, q1);
        // toward links of the cell farther out, away links of the cell closer in (all lanes take part)
        Tri tw = {__shfl(st.s1t.d, lane_out), 0.f, __shfl(st.s1t.g, lane_out)};
        const float tm_new = __shfl(n1.tm, lane_out);
        float a0 = __shfl(st.s1a.d, lane_in), ap = __shfl(st.s1a.g, lane_in), am = __shfl(n1.am, lane_in);
        if (cx.hd == 1) {                           // closer in = my own edge cell
            a0 = left ? w1_d3x : w1_d1w; ap = left ? w1_g6x : w1_g5w; am = left ? q1[D::Cn].x : q1[D::Bn].w;
        }
#ifdef LB_DIAG
        if (!(a.diag & 1024))
#endif
        if (halo2) {
            Cell c;
            halo_cell_next<BC, MASK, DOWN>(a, cx.hx, a.y0 + r2, left, (st.hmask & 2u) != 0, st.s1c, n1.cm, tw, tm_new, a0, ap, am, c);
            n2 = halo_all<DOWN>(c, left);
        }
#ifdef LB_DIAG
        if (!(a.diag & 2))
#endif
        collide_row<BC, MASK>(a, x4, a.y0 + r2, q2, mask_bits(st.mhist, 1), r4, u4, v4);
        if (NST == 2) {                    // my position 0 after step 2 -> the other wave's window 2
            lds_publish<DOWN>(cx.P2, lane, 3, q2);
            if (halo1) {
                float *hv = &cx.xchg->v[wy ^ 1][1][0][hslot];
                hv[0] = n2.cm; hv[8] = n2.tm;
            }
        }
    } else {
        window_push_dir<DOWN>(w1, q1);
#pragma unroll
        for (int k = 0; k < 9; ++k) q2[k] = f4a{0.f, 0.f, 0.f, 0.f};
    }
    // ---- step 3 of position i-2 (window 2, LDS) ----------------------------------------------------------------
    f4a q3[9];
    HaloCell9 n3 = {};                              // stage-3 links of my halo cell at position i-2
    if (NST >= 3) {
        int r3, t0_, t1_;
        (void)step1_rows(a, row_at(i - 2), r3, t0_, t1_);
        Window w2;
        lds_window_load(W2, lane, it, w2);
        stage_gather<DOWN>(w2, q2, st.s2t, n2.tm, lane, q3);
        const float e0 = left ? w2.d3.x : w2.d1.w, ep = left ? w2.g6.x : w2.g5.w, em = left ? q2[D::Cn].x : q2[D::Bn].w;
        lds_window_push<DOWN>(W2, lane, it, q2);
        Tri tw = {__shfl(st.s2t.d, lane_out), 0.f, __shfl(st.s2t.g, lane_out)};
        const float tm_new = __shfl(n2.tm, lane_out);
#ifdef LB_DIAG
        if (!(a.diag & 1024))
#endif
        if (halo3) {
            Cell c;
            halo_cell_next<BC, MASK, DOWN>(a, cx.hx, a.y0 + r3, left, (st.hmask & 4u) != 0, st.s2c, n2.cm, tw, tm_new, e0, ep, em, c);
            n3 = halo_all<DOWN>(c, left);
        }
#ifdef LB_DIAG
        if (!(a.diag & 4))
#endif
        collide_row<BC, MASK>(a, x4, a.y0 + r3, q3, mask_bits(st.mhist, 2), r4, u4, v4);
#ifdef LB_DIAG
        if (a.diag & 1048576) {
            // timing only (wrong results): a FIFTH stage's worth of work -- a window load and push on W2 again, the gather with its
            // six cross-lane moves, a halo-cell stage, a collide -- to price five steps per pass before building them
            // (tools/r04_fifth_stage.sh, profiles/r04_experiments.txt section 9)
            Window w2b;
            lds_window_load(W2, lane, it, w2b);
            f4a q3b[9];
            stage_gather<DOWN>(w2b, q3, st.s2t, n3.tm, lane, q3b);
            lds_window_push<DOWN>(W2, lane, it, q3);
            if (halo3) {
                Cell c;
                Tri tw5 = {__shfl(st.s2t.d, lane_out), 0.f, __shfl(st.s2t.g, lane_out)};
                halo_cell_next<BC, MASK, DOWN>(a, cx.hx, a.y0 + r3, left, false, st.s2c, n3.cm, tw5, n3.tm, n3.a0, n3.ap, n3.am, c);
                n3 = halo_all<DOWN>(c, left);
            }
            collide_row<BC, MASK>(a, x4, a.y0 + r3, q3b, mask_bits(st.mhist, 2), r4, u4, v4);
#pragma unroll
            for (int k = 0; k < 9; ++k) q3[k] = q3b[k];
        }
#endif
        if (NST == 3) {                    // my position 0 after step 3 -> the other wave's window 3
            lds_publish<DOWN>(cx.P3, lane, 6, q3);
            if (halo1) cx.xchg->v[wy ^ 1][2][0][hslot] = n3.tm;
        }
    } else {
        if (i == 1) lds_window_push<DOWN>(W2, lane, it, q2);    // (iteration 0 has nothing to push: slots 3..5 belong to the mailbox)
#pragma unroll
        for (int k = 0; k < 9; ++k) q3[k] = f4a{0.f, 0.f, 0.f, 0.f};
    }
    // ---- step 4 of position i-3 (window 3, LDS), stored --------------------------------------------------------
    if (NST >= 4) {
        int r4_, t0_, t1_;
        (void)step1_rows(a, row_at(i - 3), r4_, t0_, t1_);
        Window w3;
        lds_window_load(W3, lane, it, w3);
        f4a t[9];
        stage_gather<DOWN>(w3, q3, st.s3t, n3.tm, lane, t);
        lds_window_push<DOWN>(W3, lane, it, q3);
#ifdef LB_DIAG
        if (!(a.diag & 2048))
#endif
        collide_row<BC, MASK>(a, x4, a.y0 + r4_, t, mask_bits(st.mhist, 3), r4, u4, v4);
        if (cx.store_lane) {
            const long long o = (long long)r4_ * a.pitch;   // row start, uniform
            float *d = a.dst + o;
            store_row9<NTS>(a.nts != 0, d, S, x4, t);
            if (MACRO) {
                const long long m = (long long)r4_ * a.fpitch;
                store4<false>(lane_ptr(a.rho + m, x4), r4);
                store4<false>(lane_ptr(a.u + m, x4), u4);
                store4<false>(lane_ptr(a.v + m, x4), v4);
            }
        }
    } else if (i == 2) {
        lds_window_push<DOWN>(W3, lane, it, q3);    // position 0 after step 3: the d slots and ring slot 3 (the other
    }                                               // wave fills ring slot 6; iterations 0, 1 have nothing to push)
    // ---- slide the halo cells' delay lines -----------------------------------------------------------------------
    tri_push(st.s1c, n1.c0, n1.cp); tri_push(st.s1t, n1.t0, n1.tp); tri_push(st.s1a, n1.a0, n1.ap);
    tri_push(st.s2c, n2.c0, n2.cp); tri_push(st.s2t, n2.t0, n2.tp);
    tri_push(st.s3t, n3.t0, n3.tp);
    if (MASK) {
        st.mhist = ((st.mhist | mask_word(mk)) << 1) & 0x0e0e0e0eu;
        st.hmask = ((st.hmask | (hsolid ? 1u : 0u)) << 1) & 0x6u;
    }
    if (NST < 4) __syncthreads();         // what was published in this iteration is consumed in the next
}

// Halo lanes.  The three cells beyond each end of the strip are spread over six lanes -- lanes 0,1,2 own
// the cells at distance 1,2,3 on the left, lanes 63,62,61 those on the right -- so that one scalar-cell
// call per stage serves all of them at once (six sequential calls by two lanes made the kernel bound by
// vector-ALU issue).  A halo cell takes the links moving toward the strip from the lane that owns the
// cell farther out (`lane_out`) and the links moving away from it from the lane that owns the cell
// closer in (`lane_in`; for distance 1 that is the strip's own edge cell, in the same lane), through
// ds_bpermute, which does not occupy the vector ALU.
//
// One wave's march: strip [x0, x0 + 256), own rows = `len` rows starting at the pair's middle line `ym` and going up
// (rows ym .. ym+len-1) or down (rows ym-1 .. ym-len).  Position p of the march = row ym + p / ym - 1 - p; iteration i
// loads position i and runs step 1 on it, step 2 on position i-1, step 3 on i-2, step 4 on i-3 (stored): len + 3
// iterations.  What the steps of position 0 pull from "position -1" is the other wave's position 0 (see the header).
// Iterations 0..2 (the pipeline fills, the two waves hand over across the middle line) and len..len+2 (the three rows beyond
// the far end, which may lie outside a wall; no next row to gather) run the generic body; the iterations in between, two by
// two, the steady one (march4_iter).
// (Touching the next row's 90 cache lines a row ahead with two one-lane-per-line loads, instead of the register prefetch
//  PF below, costs more in the texture addresser than the wait it saves: 243 -> 162 k MLUPS at 8192^2,
//  profiles/r02_experiments.txt.)
template <int BC, bool MASK, bool MACRO, bool NTS, bool PF, bool DOWN>
__device__ __forceinline__ void march4(const StepArgs &a, const int x0, const int ym, const int len, const int wy,
                                       f4a (*lds_win)[2][9][64], HaloXchg &xchg, const unsigned slot)
{
    March4Ctx cx;
    cx.lane = threadIdx.x;
    const int xr = x0 + cx.lane * 4;
    cx.x4 = xr;
    if (BC == LB_BC_PERIODIC && xr >= a.nx) cx.x4 = xr - a.nx;
    cx.store_lane = xr < a.nx;
    cx.left = (cx.lane < 32);
    cx.hd = cx.left ? cx.lane + 1 : 64 - cx.lane;    // distance of my halo cell from the strip (1..3 in halo lanes)
    cx.halo1 = (cx.hd <= 3); cx.halo2 = (cx.hd <= 2); cx.halo3 = (cx.hd == 1);   // lanes taking part in halo stages 1, 2, 3
    cx.hx = cx.left ? x0 - cx.hd : x0 + STRIP_W - 1 + cx.hd;                     // my halo cell
    cx.lane_out = cx.left ? cx.lane + 1 : cx.lane - 1;          // owner of the cell one farther out
    cx.lane_in = cx.left ? max(cx.lane - 1, 0) : min(cx.lane + 1, 63);      // owner of the cell one closer in
    cx.hslot = cx.left ? cx.lane : cx.lane - 56;     // my place in the halo exchange (halo lanes: 0,1,2 | 5,6,7)
    cx.ym = ym; cx.n_iter = len + 3; cx.wy = wy; cx.slot = slot;
    cx.W2 = lds_win[wy][0];
    cx.W3 = lds_win[wy][1];
    cx.P2 = lds_win[wy ^ 1][0];                      // the other wave's windows
    cx.P3 = lds_win[wy ^ 1][1];
    cx.xchg = &xchg;
    March4State st = {};
    auto row_at = [&](int p) { return DOWN ? ym - 1 - p : ym + p; };

    // PF: the gather of the next row is issued before the current one is computed, so that a wave does not wait a full
    // memory latency per row (it spent 25 % of its cycles there: profiles/r02_experiments.txt).  46 more registers: only
    // the instantiations that stay under 256 without spilling are launched with it (launch_step2_bc).
    Row1 ra, rb;
    if (PF) row1_load<BC, MASK>(a, row_at(0), cx.x4, cx.halo1, cx.hx, ra);
    // the pipeline fills: iterations 0, 1, 2 run one, two, three stages (every launch gives a wave at least 4 rows: n_iter >= 7)
    march4_iter<BC, MASK, MACRO, NTS, PF, DOWN, 1>(a, cx, 0, st, ra, PF ? rb : ra);
    if (PF) ra = rb;
    march4_iter<BC, MASK, MACRO, NTS, PF, DOWN, 2>(a, cx, 1, st, ra, PF ? rb : ra);
    if (PF) ra = rb;
    march4_iter<BC, MASK, MACRO, NTS, PF, DOWN, 3>(a, cx, 2, st, ra, PF ? rb : ra);
    if (PF) ra = rb;
    // The full pipeline.  (Round 4 also ran these iterations in PAIRS, the two row buffers swapping roles so that "this row =
    // the row gathered an iteration ago" is a register name instead of 60 v_mov per row: 927 instead of ~990 vector
    // instructions per wave and row, no scratch at 251-253 registers -- and 1-1.5 % SLOWER at 8192^2 on every box tried
    // (17 KB of loop body per direction instead of 9; profiles/r04_experiments.txt).  The kernel is not bound by vector-ALU
    // issue any more; the pairs are not kept.  PAIRS = true restores them.)
    constexpr bool PAIRS = false;
    int i = 3;
    if (PAIRS) {
        for (; i + 1 < cx.n_iter; i += 2) {
            march4_iter<BC, MASK, MACRO, NTS, PF, DOWN, 4>(a, cx, i, st, ra, rb);
            march4_iter<BC, MASK, MACRO, NTS, PF, DOWN, 4>(a, cx, i + 1, st, rb, ra);
        }
    }
    for (; i < cx.n_iter; ++i) {
        march4_iter<BC, MASK, MACRO, NTS, PF, DOWN, 4>(a, cx, i, st, ra, PF ? rb : ra);
        if (PF) ra = rb;
    }
}

// Which instantiations gather one row ahead (template flag PF): those that stay within 256 registers without scratch
// (tools/kernel_resources.py k_step4).
constexpr bool step4_prefetch(int bc, bool mask, bool macro)
{
    // (with a mask: it fits since round 4 -- pipe 254 registers, cavity 256, no scratch -- and buys nothing: config 5
    //  262 k MLUPS with it, 267 k without, one box: profiles/r04_experiments.txt)
    // (the D2Q9i fork's launch that also stores rho, u, v: 20 B of scratch with the prefetch)
    return bc != LB_BC_VELOCITY_INLET && !mask && !(bc == LB_BC_PIPE_I && macro);
}

// XW = strips per workgroup.  1 (what is launched): a workgroup is the two waves of one segment pair.  2 was an experiment of
// round 4: the pairs of two x-ADJACENT strips share a workgroup -- they start together on one CU, so the cache lines a strip's
// halo cells and displaced loads take from its neighbour's strip (the read traffic beyond the compulsory bytes: +12.7 % at
// 8192^2) would be lines the neighbour's waves fetch at about the same time on the same L2.  Bitwise equal, 74.9 KB of LDS per
// workgroup, and no faster: 8192^2 314.7 / 316.3 / 314.3 k MLUPS against 318.7 / 316.2 / 319.1 k, 4096^2 287-292 k against
// 285-299 k (one box, alternating: profiles/r04_experiments.txt).  The template parameter stays; nothing instantiates 2.
template <int BC, bool MASK, bool MACRO, bool NTS, bool PF, int XW = 1>
__global__ __launch_bounds__(64 * STEP4_WAVES * XW, 2) void k_step4(const StepArgs a, int strips, int seg_rows, int nsegs,
                                                                    int row_end)
{
    __shared__ f4a lds_win_all[STEP4_WAVES * XW][2][9][64];
    __shared__ HaloXchg xchg_all[XW];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int wy = w & 1, wx = w >> 1;
    f4a (*lds_win)[2][9][64] = lds_win_all + STEP4_WAVES * wx;
    HaloXchg &xchg = xchg_all[wx];
    // one workgroup = one pair of segments of one strip (XCD-transposed order, as k_step3: eight x-adjacent strips share an L2)
    int item = xcd_item(blockIdx.x, gridDim.x);
#ifdef LB_DIAG
    if (a.diag & 65536) item = blockIdx.x;              // experiment: no XCD transposition
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
#endif
    // items [0, strips * nsegs): pair item / strips of strip item % strips.  Behind them, where the box has walls at its
    // left and right end (a.edge_seg_rows > 0): further pairs of the first and the last strip only.  Those two strips run
    // the inlet / outlet / wall rule for one cell per row and stage, which makes their rows ~18 % dearer; with equal segments
    // their waves were the last to finish by that margin in every launch (tools/wave_timeline.py), so they get shorter ones.
    // (XW = 2: an item is a pair of strips -- 2 i and 2 i + 1 --; behind them, items that hold one further pair of BOTH edge strips)
    int sx, sy;
    const unsigned slot = __builtin_amdgcn_s_getreg((4 << 11) | 4) & 1u;       // HW_REG_HW_ID wave_id bit 0: my slot on the SIMD
    const int sgroups = (strips + XW - 1) / XW;
    if (item < sgroups * nsegs) {
        sx = (item % sgroups) * XW + wx;
        sy = item / sgroups;
        if (sx >= strips) return;                       // (an odd number of strips: the last group holds one)
    } else {
        if (!a.edge_seg_rows) return;
        const int j = item - sgroups * nsegs;
        if (XW == 1) {
            sx = (j & 1) ? strips - 1 : 0;
            sy = nsegs + (j >> 1);
        } else {
            sx = wx ? strips - 1 : 0;
            sy = nsegs + j;
        }
    }
    int stride = a.seg_stride, rows = seg_rows;
    if (a.edge_seg_rows && (sx == 0 || sx == strips - 1)) stride = rows = a.edge_seg_rows;
    const int ya = a.row_begin + sy * stride;
    if (ya >= row_end) return;                          // (both waves of the workgroup: the barriers below stay matched)
    const int yb = min(ya + rows, row_end);
    const int ym = ya + (yb - ya) / 2;                  // the pair's middle line: wave 0 marches down from it, wave 1 up
    if (wy == 0) march4<BC, MASK, MACRO, NTS, PF, true>(a, sx * STRIP_W, ym, ym - ya, 0, lds_win, xchg, slot);
    else march4<BC, MASK, MACRO, NTS, PF, false>(a, sx * STRIP_W, ym, yb - ym, 1, lds_win, xchg, slot);
#ifdef LB_DIAG
    if ((a.diag & 4096) && threadIdx.x == 0) {
        // per-wave timeline into the (otherwise unused) rho array: start, end (100 MHz ticks), XCC id, HW id, item
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        unsigned *o = reinterpret_cast<unsigned *>(a.rho) + 8 * (item * STEP4_WAVES + wy);
        o[0] = (unsigned)diag_t0; o[1] = (unsigned)(diag_t0 >> 32); o[2] = (unsigned)t1; o[3] = (unsigned)(t1 >> 32);
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
        o[6] = (unsigned)(item * STEP4_WAVES + wy); o[7] = (unsigned)(wy ? yb - ym : ym - ya);
    }
#endif
}

}  // namespace
