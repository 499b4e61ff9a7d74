// kernels_step4.h -- four time steps per pass.  Included by lb_hip.cpp after kernels_fused.h.
//
// The march of k_step3 with one more stage.  Per row r: step 1 of row r (from memory), step 2 of row
// r-1, step 3 of row r-2, step 4 of row r-3 (stored).  The window between steps 1 and 2 stays in
// registers; the windows between steps 2/3 and 3/4 live in LDS, wave-private (9 slots of 64 x 16 B each:
// links 0,1,3 of the previous row, links 2,5,6 of the previous two rows in a 2-deep ring), which is what
// keeps the kernel at two waves per SIMD -- an LDS-resident window costs ~1 % against registers
// (measured on k_step3, profiles/r01_ablation.txt).  No barriers: a wave only ever reads what it wrote.
//
// The cells beyond the strip are recomputed by the edge lanes as scalar cells, as in k_step3, one more
// ring per stage: step 1 for x0-3..x0-1 | x0+256..x0+258, step 2 for the inner two, step 3 for the
// innermost.  A halo cell at distance d takes its centre links from itself, the links moving toward the
// strip from the cell at d+1, the links moving away from it from the cell at d-1 (d = 1: the strip's own
// edge cell, out of the vector registers).
//
// HBM traffic per four updates of a cell: 9 reads + 9 writes (+6 rows per segment): ~19 B per lattice update.
#pragma once

namespace {

// what later stages can ask of a halo cell: centre links (cx = 0), links moving toward the strip
// (cx = +1 on the left side: 1,5,8; cx = -1 on the right: 3,6,7) and away from it, each by cy = 0, +1, -1
struct HaloCell9 {
    float c0, cp, cm;      // links 0, 2, 4
    float t0, tp, tm;
    float a0, ap, am;
};
__device__ __forceinline__ HaloCell9 halo_all(const Cell &c, bool left)
{
    HaloCell9 h;
    h.c0 = c.f0; h.cp = c.f2; h.cm = c.f4;
    h.t0 = left ? c.f1 : c.f3; h.tp = left ? c.f5 : c.f6; h.tm = left ? c.f8 : c.f7;
    h.a0 = left ? c.f3 : c.f1; h.ap = left ? c.f6 : c.f5; h.am = left ? c.f7 : c.f8;
    return h;
}
// delay line of one link triple: cy = 0 link of the previous row, cy = +1 link of the previous two rows
// (the cy = -1 link is consumed in the iteration that produces it)
struct Tri {
    float d, e, g;
};
__device__ __forceinline__ void tri_push(Tri &w, float x0, float xp)
{
    w.g = w.e; w.e = xp; w.d = x0;
}

// Next stage of the halo cell at (hx, row yg_row): pre-collision links from the delay lines, then boundary
// rule, obstacle swap and relaxation as everywhere else.  centre/toward: delay lines + this iteration's
// cy = -1 links (cm_new, tm_new); the away links of the closer cell: a0 (cy = 0), ap (cy = +1), am (cy = -1).
template <int BC, bool MASK>
__device__ __forceinline__ void halo_cell_next(const StepArgs &a, int hx, int yg, bool left, bool solid,
                                               const Tri &centre, float cm_new, const Tri &toward, float tm_new,
                                               float a0, float ap, float am, Cell &c)
{
    c.f0 = centre.d; c.f2 = centre.g; c.f4 = cm_new;
    const float t0 = toward.d, tp = toward.g;
    c.f1 = left ? t0 : a0; c.f3 = left ? a0 : t0;
    c.f5 = left ? tp : ap; c.f6 = left ? ap : tp;
    c.f8 = left ? tm_new : am; c.f7 = left ? am : tm_new;
    int xc = hx;
    if (BC == LB_BC_PERIODIC) xc = hx < 0 ? hx + a.nx : (hx >= a.nx ? hx - a.nx : hx);
    else if (hx < 0 || hx >= a.nx) return;              // outside the box: don't-care
    if (BC != LB_BC_PERIODIC) {
        const bool w = (xc == 0), e = (xc == a.nx - 1), so = (yg == 0), no = (yg == a.ny - 1);
        if (w || e || so || no) {
            if (BC == LB_BC_PIPE) bc_pipe_cell(c, w, e, so, no, a.rho_in, a.rho_out);
            if (BC == LB_BC_CAVITY) bc_cavity_cell(c, w, e, so, no, a.lid_u, a.rho0);
        }
    }
    if (MASK) bounce_cell(c, solid);
    float rho, ux, uy;
    relax_cell(c, a.omega, rho, ux, uy);
}

// the LDS-resident window: slots 0,1,2 = links 0,1,3 of the previous row; 3..5 / 6..8 = links 2,5,6 of the
// previous two rows (ring, slot chosen by the iteration's parity)
__device__ __forceinline__ void lds_window_load(f4a (*W)[64], int lane, int it, Window &w)
{
    const int gs = 3 + 3 * (it & 1);                    // the older of the two ring rows
    w.d0 = W[0][lane]; w.d1 = W[1][lane]; w.d3 = W[2][lane];
    w.g2 = W[gs][lane]; w.g5 = W[gs + 1][lane]; w.g6 = W[gs + 2][lane];
}
__device__ __forceinline__ void lds_window_push(f4a (*W)[64], int lane, int it, const f4a (&q)[9])
{
    const int gs = 3 + 3 * (it & 1);                    // overwrite the row just consumed
    W[0][lane] = q[0]; W[1][lane] = q[1]; W[2][lane] = q[3];
    W[gs][lane] = q[2]; W[gs + 1][lane] = q[5]; W[gs + 2][lane] = q[6];
}

// gather of the next stage for my 4 cells from a window holding {d0,d1,d3,g2,g5,g6}, the newest row q and
// the innermost halo cell's toward links (delay line + this iteration's cy = -1 link)
__device__ __forceinline__ void stage_gather(const Window &w, const f4a (&q)[9], const Tri &ht, float htm_new, int lane,
                                             f4a (&t)[9])
{
    t[0] = w.d0;
    t[1] = from_left(w.d1, ht.d, lane);
    t[3] = from_right(w.d3, ht.d, lane);
    t[2] = w.g2;
    t[5] = from_left(w.g5, ht.g, lane);
    t[6] = from_right(w.g6, ht.g, lane);
    t[4] = q[4];
    t[7] = from_right(q[7], htm_new, lane);
    t[8] = from_left(q[8], htm_new, lane);
}

constexpr int STEP4_WAVES = 2;      // waves per workgroup: 2 x 2 windows x 9 KiB = 36 KiB of LDS

template <int BC, bool MASK, bool MACRO, bool NTS>
__global__ __launch_bounds__(64 * STEP4_WAVES, 2) void k_step4(const StepArgs a, int strips, int seg_rows, int nsegs,
                                                               int row_end)
{
    __shared__ f4a lds_win[STEP4_WAVES][2][9][64];
    const int lane = threadIdx.x;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int item = xcd_item(blockIdx.x, gridDim.x) * STEP4_WAVES + wy;   // (XCD-transposed order, as k_step3)
    const int sx = item % strips, sy = item / strips;
    if (sy >= nsegs) return;
    const int ya = a.row_begin + sy * a.seg_stride;
    if (ya >= row_end) return;
    const int yb = min(ya + seg_rows, row_end);
    const int x0 = sx * STRIP_W;
    const int xr = x0 + lane * 4;
    int x4 = xr;
    if (BC == LB_BC_PERIODIC && xr >= a.nx) x4 = xr - a.nx;
    const bool store_lane = xr < a.nx;
    const bool left = (lane == 0);
    const bool edge_lane = left || (lane == 63);
    const int hx1 = left ? x0 - 1 : x0 + STRIP_W;       // halo cells at distance 1, 2, 3 from the strip
    const int hx2 = left ? x0 - 2 : x0 + STRIP_W + 1;
    const int hx3 = left ? x0 - 3 : x0 + STRIP_W + 2;
    const long long S = a.plane;

    f4a(*W2)[64] = lds_win[wy][0];
    f4a(*W3)[64] = lds_win[wy][1];
    {
        const f4a z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 9; ++k) { W2[k][lane] = z; W3[k][lane] = z; }
    }
    Window w1 = {};
    // delay lines of the halo cells: stage 1: cell 1 (centre, toward, away), cell 2 (centre, toward), cell 3
    // (toward); stage 2: cell 1 (centre, toward), cell 2 (toward); stage 3: cell 1 (toward)
    Tri s1c1 = {}, s1t1 = {}, s1a1 = {}, s1c2 = {}, s1t2 = {}, s1t3 = {};
    Tri s2c1 = {}, s2t1 = {}, s2t2 = {};
    Tri s3t1 = {};
    // obstacle-mask history: my cells (per byte: bit 1 = row r-1, bit 2 = r-2, bit 3 = r-3); halo cells
    // (bits 0..2 = cells 1..3 in row r-1, bits 3..5 = in row r-2)
    unsigned mhist = 0, hmask = 0;
    int it = 0;

    for (int r = ya - 3; r <= yb + 2; ++r, ++it) {
        // ---- step 1 of row r (from memory) ---------------------------------------------------------------
        f4a q1[9], r4, u4, v4;
        uc4 mk = {0, 0, 0, 0};
        int rr, ym, yp;
        const bool have = step1_rows(a, r, rr, ym, yp);
        HaloCell9 n11 = {}, n12 = {}, n13 = {};         // stage-1 links of halo cells 1, 2, 3 in row r
        unsigned hcur = 0;
        if (have) {
            gather_row<BC, MASK, false>(a, x4, rr, ym, yp, q1, mk);
            if (edge_lane) {
                Cell c;
                bool sol;
                halo_cell_step1<BC, MASK>(a, hx1, rr, ym, yp, c, sol);
                n11 = halo_all(c, left); hcur |= sol ? 1u : 0u;
                halo_cell_step1<BC, MASK>(a, hx2, rr, ym, yp, c, sol);
                n12 = halo_all(c, left); hcur |= sol ? 2u : 0u;
                halo_cell_step1<BC, MASK>(a, hx3, rr, ym, yp, c, sol);
                n13 = halo_all(c, left); hcur |= sol ? 4u : 0u;
            }
            collide_row<BC, MASK>(a, x4, a.y0 + rr, q1, mk, r4, u4, v4);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) q1[k] = f4a{0.f, 0.f, 0.f, 0.f};
        }
        // ---- step 2 of row r-1 (window 1, registers) -----------------------------------------------------
        f4a q2[9];
        HaloCell9 n21 = {}, n22 = {};                   // stage-2 links of halo cells 1, 2 in row r-1
        if (r >= ya - 1) {
            int r2, t0_, t1_;
            (void)step1_rows(a, r - 1, r2, t0_, t1_);
            stage_gather(w1, q1, s1t1, n11.tm, lane, q2);
            if (edge_lane) {
                const int yg = a.y0 + r2;
                Cell c;
                // cell 1: away links of the closer cell = my own edge cell
                halo_cell_next<BC, MASK>(a, hx1, yg, left, (hmask & 1u) != 0, s1c1, n11.cm, s1t2, n12.tm,
                                         left ? w1.d3.x : w1.d1.w, left ? w1.g6.x : w1.g5.w, left ? q1[7].x : q1[8].w, c);
                n21 = halo_all(c, left);
                // cell 2: away links of the closer cell = halo cell 1
                halo_cell_next<BC, MASK>(a, hx2, yg, left, (hmask & 2u) != 0, s1c2, n12.cm, s1t3, n13.tm,
                                         s1a1.d, s1a1.g, n11.am, c);
                n22 = halo_all(c, left);
            }
            collide_row<BC, MASK>(a, x4, a.y0 + r2, q2, mask_bits(mhist, 1), r4, u4, v4);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) q2[k] = f4a{0.f, 0.f, 0.f, 0.f};
        }
        // ---- step 3 of row r-2 (window 2, LDS) -----------------------------------------------------------
        f4a q3[9];
        HaloCell9 n31 = {};                             // stage-3 links of halo cell 1 in row r-2
        if (r >= ya + 1) {
            int r3, t0_, t1_;
            (void)step1_rows(a, r - 2, r3, t0_, t1_);
            Window w2;
            lds_window_load(W2, lane, it, w2);
            stage_gather(w2, q2, s2t1, n21.tm, lane, q3);
            if (edge_lane) {
                Cell c;
                halo_cell_next<BC, MASK>(a, hx1, a.y0 + r3, left, (hmask & 8u) != 0, s2c1, n21.cm, s2t2, n22.tm,
                                         left ? w2.d3.x : w2.d1.w, left ? w2.g6.x : w2.g5.w, left ? q2[7].x : q2[8].w, c);
                n31 = halo_all(c, left);
            }
            collide_row<BC, MASK>(a, x4, a.y0 + r3, q3, mask_bits(mhist, 2), r4, u4, v4);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) q3[k] = f4a{0.f, 0.f, 0.f, 0.f};
        }
        // ---- step 4 of row r-3 (window 3, LDS), stored ----------------------------------------------------
        if (r >= ya + 3) {
            int r4_, t0_, t1_;
            (void)step1_rows(a, r - 3, r4_, t0_, t1_);
            Window w3;
            lds_window_load(W3, lane, it, w3);
            f4a t[9];
            stage_gather(w3, q3, s3t1, n31.tm, lane, t);
            collide_row<BC, MASK>(a, x4, a.y0 + r4_, t, mask_bits(mhist, 3), r4, u4, v4);
            if (store_lane) {
                const long long o = (long long)r4_ * a.pitch;   // row start, uniform
                float *d = a.dst + o;
#pragma unroll
                for (int k = 0; k < 9; ++k) store4<NTS>(lane_ptr(d + k * S, x4), t[k]);
                if (MACRO) {
                    store4<false>(lane_ptr(a.rho + o, x4), r4);
                    store4<false>(lane_ptr(a.u + o, x4), u4);
                    store4<false>(lane_ptr(a.v + o, x4), v4);
                }
            }
        }
        // ---- slide everything -------------------------------------------------------------------------------
        window_push(w1, q1);
        lds_window_push(W2, lane, it, q2);
        lds_window_push(W3, lane, it, q3);
        tri_push(s1c1, n11.c0, n11.cp); tri_push(s1t1, n11.t0, n11.tp); tri_push(s1a1, n11.a0, n11.ap);
        tri_push(s1c2, n12.c0, n12.cp); tri_push(s1t2, n12.t0, n12.tp);
        tri_push(s1t3, n13.t0, n13.tp);
        tri_push(s2c1, n21.c0, n21.cp); tri_push(s2t1, n21.t0, n21.tp);
        tri_push(s2t2, n22.t0, n22.tp);
        tri_push(s3t1, n31.t0, n31.tp);
        if (MASK) {
            mhist = ((mhist | mask_word(mk)) << 1) & 0x0e0e0e0eu;
            hmask = ((hmask << 3) | hcur) & 0x3fu;
        }
    }
}

}  // namespace
