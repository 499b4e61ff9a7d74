// kernels_step5.h -- FIVE time steps per pass.  Included by lb_hip.cpp after kernels_step4.h, whose building blocks it uses
// (segment pairs, Window, the LDS window helpers, Row1 / row1_load: read that header first).
//
// Why.  Round 4 took a fifth of k_step4's vector instructions away and its launches took what they had taken: the four-step
// kernel moves its bytes at the ceiling of the marching waves' access pattern (~6.0 TB/s counted, 1.08 x the compulsory bytes),
// and its vector ALU idles a third of the time.  A launch costs about the same whatever it computes per row, so the way up is a
// FIFTH time step per pass -- provided the kernel keeps eight waves per CU (timing probe on the diagnostic build,
// tools/r04_fifth_stage.sh: with a third LDS window, i.e. six waves per CU, the gain is gone -- for k_step4's halo-lane form, which
// the probe ran; THIS kernel turned out to run as fast at six waves per CU, which is what kernels_step6.h builds on).  Hence: the windows between steps
// 1/2 AND 2/3 in registers (2 x 36), those between 3/4 and 4/5 in LDS (the 36 KB per workgroup k_step4 uses).
//
// Overlapping strips instead of halo lanes.  k_step4 recomputes the cells beyond its 256-cell strip as scalar cells in "halo
// lanes" -- a third of its vector instructions, 27 registers of delay lines, and with four cells per side and stage the first
// version of this kernel spilled (1.37 ms instead of 0.88 ms per 8192^2 launch: profiles/r04_experiments.txt section 10).  Here a
// wave's 64 lanes x 4 cells ARE the strip and its skirt: strips were laid 248 cells apart and started 4 cells early (since the end of
// round 5: 240 apart, 8 early, LB_STEP5_ALIGN64 below), step s is
// right for the cells at least s - 1 away from either end (what is wrong creeps in one cell per step), so after step 5 lanes
// 1..62 hold 248 good cells and lanes 0 and 63 are never stored.  No halo cells, no delay lines, no exchange between the halo
// lanes of a pair; the price is 256 / 248 = +3.2 % rows read and computed (the halo lanes read those cells, too) and row starts
// that are 32-byte- instead of 1-KB-aligned (valid regions start at multiples of 992 B; six of the nine planes are read displaced
// by one cell anyway).  Walls: the rule of a wall column rebuilds what it pulled from outside the box, so x = 0 and x = nx - 1
// are right at every step whatever the lanes beyond them hold.
//
// The launch moves k_step4's 72 B per cell for five steps instead of four.  Same cell functions: bitwise equal to k_step.
#pragma once

namespace {

// LB_STEP5_ALIGN64 (round 5, default): strips 240 cells = 15 x 64 bytes apart instead of 248 = 15.5 x 64 -- two lanes of skirt per
// side, of which the inner one is computed right and not stored --, so that every strip's stores begin and end on 64-byte boundaries.
// With 248 every other seam cut a 64-byte sector into two parts written at different times; k_deep with its seams at 8-byte offsets
// lost 12-23 % to that (profiles/r05_experiments.txt section 23).  (The counted HBM bytes do not show it -- 1.06 x compulsory before
// and after, profiles/pmc_traffic.json --, the launch time does: 8192^2 1030 -> 894 us by rocprofv3.)  One box, 248 | 240 apart, k MLUPS: periodic 8192^2 334-350 | 375, pipe 8192^2
// 351 | 369-376, cavity 3072^2 287-291 | 296-297, velocity inlet 4096^2 266-267 | 270-273; 4096^2 and below +-1 %
// (profiles/r05_step5_align64_ab.txt): 3 % more strips, no partial sectors.
#ifndef LB_STEP5_ALIGN64
#define LB_STEP5_ALIGN64 1
#endif
constexpr int STEP5_SKIRT = LB_STEP5_ALIGN64 ? 8 : 4;   // cells a strip starts before / ends behind its stored cells (= one lane)
constexpr int STEP5_VALID = STRIP_W - 2 * STEP5_SKIRT;  // 248 cells stored per strip and row

struct March5State {
    Window w1, w2;                      // stage windows between steps 1/2 and 2/3 (registers)
    unsigned mhist;                     // obstacle-mask history of my four cells (per byte: bit j = the row loaded j iterations ago, j = 1..4)
};
struct March5Ctx {
    int lane, x4, ym, n_iter, wy;
    bool store_lane;
    unsigned slot;
    f4a (*W3)[64], (*W4)[64], (*P3)[64], (*P4)[64];     // my two LDS windows (steps 3/4, 4/5), the other wave's
    f4a (*R2)[64], (*Q2)[64];                           // PF: the ring of window 2 (six slots), the other wave's
};

// gather of the next stage for my 4 cells from a window {d0,d1,d3,g2,g5,g6} and the newest row q; what lanes 0 / 63 take
// from beyond the wave is their own value: wrong, and never within reach of a stored cell
// (the neighbour lane's element by a DPP move -- wave_shr:1 / wave_shl:1, one vector-ALU pass; lanes 0 / 63, which have no such
//  neighbour, get 0.0: as wrong as their own value, and as far from every stored cell.  Round 5; until then -- and with
//  -DLB_SKIRT_BPERMUTE -- by ds_bpermute: an LDS-queue round trip per element, 36 per row of six stages.  Bitwise equal, same speed
//  at two waves per SIMD, +2 % at one: profiles/r05_experiments.txt)
#ifndef LB_SKIRT_BPERMUTE
#define LB_SKIRT_DPP 1
#endif
__device__ __forceinline__ float wave_from_left(float v)
{
#ifdef LB_SKIRT_DPP
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0x138, 0xf, 0xf, true));
#else
    return __shfl_up(v, 1);
#endif
}
__device__ __forceinline__ float wave_from_right(float v)
{
#ifdef LB_SKIRT_DPP
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0x130, 0xf, 0xf, true));
#else
    return __shfl_down(v, 1);
#endif
}
__device__ __forceinline__ f4a skirt_left(f4a v) { return f4a{wave_from_left(v.w), v.x, v.y, v.z}; }
__device__ __forceinline__ f4a skirt_right(f4a v) { return f4a{v.y, v.z, v.w, wave_from_right(v.x)}; }
template <bool DOWN>
__device__ __forceinline__ void skirt_gather(const Window &w, const f4a (&q)[9], f4a (&t)[9])
{
    typedef Dir<DOWN> D;
    t[0] = w.d0;
    t[1] = skirt_left(w.d1);
    t[3] = skirt_right(w.d3);
    t[D::A] = w.g2;
    t[D::B] = skirt_left(w.g5);
    t[D::C] = skirt_right(w.g6);
    t[D::An] = q[D::An];
    t[D::Cn] = skirt_right(q[D::Cn]);
    t[D::Bn] = skirt_left(q[D::Bn]);
}

// One iteration: position i is loaded and takes step 1, position i-1 step 2 (window 1), i-2 step 3 (window 2), i-3 step 4 (LDS
// window 3), i-4 step 5 (LDS window 4; stored).  NST = number of stages that have a row: 1..4 in iterations 0..3 (code of their
// own, i a constant: the pipeline fills, the two waves of the pair hand over), 5 in the loop.
// PF (not launched): `cur` holds position i on entry, gathered during the previous iteration, and is gathered anew -- position
// i + 1 -- as soon as step 2 has taken what it needs from it: the loads fly while steps 3, 4, 5 compute, in the registers the row
// just consumed occupied.  Beside two register windows the compiler spills the row in flight wherever the gather sits (~21 scratch
// accesses per row: 277-295 k instead of 327 k MLUPS at 8192^2), so PF also moves window 2's ring into LDS (its links 0,1,3 stay in
// registers): 227 registers, no scratch, 48 KB of LDS per workgroup, i.e. SIX waves per CU -- and the same speed as eight waves
// without the gather (8192^2: 335.4-336.7 k against 330.0-335.8 k MLUPS, 4096^2 307-310 k against 313-336 k: not a matter of
// loads in flight any more; profiles/r04_experiments.txt section 10).
template <int BC, bool MASK, bool MACRO, bool PF, bool DOWN, int NST>
__device__ __forceinline__ void march5_iter(const StepArgs &a, const March5Ctx &cx, const int i_, March5State &st, Row1 &cur)
{
    const int lane = cx.lane, x4 = cx.x4;
    const long long S = a.plane;
    const int i = NST < 5 ? NST - 1 : i_;
    const int it = i;
    auto row_at = [&](int p) { return DOWN ? cx.ym - 1 - p : cx.ym + p; };
    f4a(*W3)[64] = cx.W3;
    f4a(*W4)[64] = cx.W4;
    Window &w1 = st.w1, &w2 = st.w2;

    if (a.prio_turns > 0 && (i & 3) == 0) {            // the two waves of a SIMD take turns at the higher priority (march4_iter)
        const unsigned turn = (unsigned)(__builtin_amdgcn_s_memrealtime() >> a.prio_turns) & 1u;
        if (turn == cx.slot) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
    // ---- what the other wave published for "position -1" in its previous iteration ---------------------------
    // (steps 1 and 2 of its position 0 come through my still idle window 4 -- slots 6..8, then 3..5 -- into my register windows;
    //  steps 3 and 4 went straight into the ring slots of my LDS windows)
    if (NST == 2) { w1.g2 = W4[6][lane]; w1.g5 = W4[7][lane]; w1.g6 = W4[8][lane]; }
    if (NST == 3 && !PF) { w2.g2 = W4[3][lane]; w2.g5 = W4[4][lane]; w2.g6 = W4[5][lane]; }
    // ---- step 1 of position i (from memory) --------------------------------------------------------------------
    if (!PF) row1_load<BC, MASK>(a, row_at(i), x4, false, 0, cur);
    f4a (&q1)[9] = cur.q;
    f4a r4, u4, v4;
    const uc4 mk = cur.mk;
    if (cur.have) {
        gather_merge<BC, true>(a, x4, q1, cur.wp);
        collide_row<BC, MASK>(a, x4, a.y0 + cur.rr, q1, mk, r4, u4, v4);
    }
    // (behind the last position the last row is gathered once more -- a cache hit that nobody consumes: no condition on i)
    const int r_next = row_at(min(i + 1, cx.n_iter - 1));
    if (NST == 1) lds_publish<DOWN>(cx.P4, lane, 6, q1);        // my position 0 after step 1 -> the other wave's window 1 (mailbox)
    // ---- step 2 of position i-1 (window 1, registers) ----------------------------------------------------------
    f4a q2[9];
    if (NST >= 2) {
        int r2, t0_, t1_;
        (void)step1_rows(a, row_at(i - 1), r2, t0_, t1_);
        skirt_gather<DOWN>(w1, q1, q2);
        window_push_dir<DOWN>(w1, q1);              // (every window takes its new row as soon as its old one has been gathered from)
        if (PF) row1_load<BC, MASK>(a, r_next, x4, false, 0, cur);
        collide_row<BC, MASK>(a, x4, a.y0 + r2, q2, mask_bits(st.mhist, 1), r4, u4, v4);
        if (NST == 2) {                             // my position 0 after step 2 -> the other wave's window 2
            if (PF) lds_publish<DOWN>(cx.Q2, lane, 0, q2);      // (its ring, the slot it reads in iteration 2)
            else lds_publish<DOWN>(cx.P4, lane, 3, q2);         // (registers there: through its idle window 4)
        }
    } else {
        window_push_dir<DOWN>(w1, q1);
        if (PF) row1_load<BC, MASK>(a, r_next, x4, false, 0, cur);
    }
    // ---- step 3 of position i-2 (window 2, registers) ----------------------------------------------------------
    f4a q3[9];
    if (NST >= 3) {
        int r3, t0_, t1_;
        (void)step1_rows(a, row_at(i - 2), r3, t0_, t1_);
        if (PF) {                                   // window 2: links 0,1,3 in registers, the ring in LDS
            typedef Dir<DOWN> D;
            const int gs = 3 * (it & 1);
            w2.g2 = cx.R2[gs][lane]; w2.g5 = cx.R2[gs + 1][lane]; w2.g6 = cx.R2[gs + 2][lane];
            skirt_gather<DOWN>(w2, q2, q3);
            cx.R2[gs][lane] = q2[D::A]; cx.R2[gs + 1][lane] = q2[D::B]; cx.R2[gs + 2][lane] = q2[D::C];
            w2.d0 = q2[0]; w2.d1 = q2[1]; w2.d3 = q2[3];
        } else {
            skirt_gather<DOWN>(w2, q2, q3);
            window_push_dir<DOWN>(w2, q2);
        }
        collide_row<BC, MASK>(a, x4, a.y0 + r3, q3, mask_bits(st.mhist, 2), r4, u4, v4);
        if (NST == 3) lds_publish<DOWN>(cx.P3, lane, 6, q3);    // my position 0 after step 3 -> the other wave's window 3
    } else if (NST == 2) {
        if (PF) {                                   // (position 0 after step 2 enters window 2: ring slot of odd iterations)
            typedef Dir<DOWN> D;
            cx.R2[3][lane] = q2[D::A]; cx.R2[4][lane] = q2[D::B]; cx.R2[5][lane] = q2[D::C];
            w2.d0 = q2[0]; w2.d1 = q2[1]; w2.d3 = q2[3];
        } else {
            window_push_dir<DOWN>(w2, q2);
        }
    }
    // ---- step 4 of position i-3 (window 3, LDS) ----------------------------------------------------------------
    f4a q4[9];
    if (NST >= 4) {
        int r4_, t0_, t1_;
        (void)step1_rows(a, row_at(i - 3), r4_, t0_, t1_);
        Window w3;
        lds_window_load(W3, lane, it, w3);
        skirt_gather<DOWN>(w3, q3, q4);
        lds_window_push<DOWN>(W3, lane, it, q3);
        collide_row<BC, MASK>(a, x4, a.y0 + r4_, q4, mask_bits(st.mhist, 3), r4, u4, v4);
        if (NST == 4) lds_publish<DOWN>(cx.P4, lane, 3, q4);    // my position 0 after step 4 -> the other wave's window 4
    } else if (NST == 3) {
        lds_window_push<DOWN>(W3, lane, it, q3);    // position 0 after step 3: the d slots and ring slot 3 (the other wave fills slot 6)
    }
    // ---- step 5 of position i-4 (window 4, LDS), stored --------------------------------------------------------
    if (NST >= 5) {
        int r5, t0_, t1_;
        (void)step1_rows(a, row_at(i - 4), r5, t0_, t1_);
        Window w4;
        lds_window_load(W4, lane, it, w4);
        f4a t[9];
        skirt_gather<DOWN>(w4, q4, t);
        lds_window_push<DOWN>(W4, lane, it, q4);
        collide_row<BC, MASK>(a, x4, a.y0 + r5, t, mask_bits(st.mhist, 4), r4, u4, v4);
        if (cx.store_lane) {
            const long long o = (long long)r5 * a.pitch;    // row start, uniform
            float *d = a.dst + o;
            store_row9<false>(a.nts != 0, d, S, x4, t);
            if (MACRO) {
                const long long m = (long long)r5 * a.fpitch;
                store4<false>(lane_ptr(a.rho + m, x4), r4);
                store4<false>(lane_ptr(a.u + m, x4), u4);
                store4<false>(lane_ptr(a.v + m, x4), v4);
            }
        }
    } else if (NST == 4) {
        lds_window_push<DOWN>(W4, lane, it, q4);    // position 0 after step 4: the d slots and ring slot 6 (the other wave fills slot 3)
    }
    if (MASK) st.mhist = ((st.mhist | mask_word(mk)) << 1) & 0x1e1e1e1eu;
    if (NST < 5) __syncthreads();                   // what was published in this iteration is consumed in the next
}

// One wave's march: columns [x0, x0 + 256) of which [x0 + 4, x0 + 252) are stored, `len` rows from the pair's middle line `ym`
// upward or downward; len + 4 iterations.
template <int BC, bool MASK, bool MACRO, bool PF, bool DOWN>
__device__ __forceinline__ void march5(const StepArgs &a, const int x0, const int ym, const int len, const int wy,
                                       f4a (*mine)[64], f4a (*other)[64], const unsigned slot)
{
    March5Ctx cx;
    cx.lane = threadIdx.x;
    const int xr = x0 + cx.lane * 4;                 // true column of my first cell: -4 .. ; may lie beyond either end of the box
    // lanes beyond an end of the box: the periodic image (nx % 4 == 0), or -- walls -- copies of the lane at that end (their
    // values are never within reach of a stored cell: the wall column's rule rebuilds what it pulled from them)
    // (periodic: only the first lane beyond the last column is anybody's skirt; the lanes behind it -- the last strip of 8192
    //  columns stores 8 cells -- read what that lane reads, i.e. the same cache lines, instead of 240 more columns)
    if (BC == LB_BC_PERIODIC) cx.x4 = xr < 0 ? xr + a.nx : (xr >= a.nx ? (LB_STEP5_ALIGN64 && xr - a.nx < 8 ? xr - a.nx : (LB_STEP5_ALIGN64 ? 4 : 0)) : xr);
    else cx.x4 = min(max(xr, 0), (a.nx - 1) & ~3);
    cx.store_lane = cx.lane >= STEP5_SKIRT / 4 && cx.lane <= 63 - STEP5_SKIRT / 4 && xr < a.nx;
    cx.ym = ym; cx.n_iter = len + 4; cx.wy = wy; cx.slot = slot;
    cx.W3 = mine; cx.W4 = mine + 9; cx.R2 = mine + 18;         // (slots: window 3, window 4, PF: the ring of window 2)
    cx.P3 = other; cx.P4 = other + 9; cx.Q2 = other + 18;
    March5State st = {};
    auto row_at = [&](int p) { return DOWN ? ym - 1 - p : ym + p; };
    Row1 cur;
    if (PF) row1_load<BC, MASK>(a, row_at(0), cx.x4, false, 0, cur);
    march5_iter<BC, MASK, MACRO, PF, DOWN, 1>(a, cx, 0, st, cur);
    march5_iter<BC, MASK, MACRO, PF, DOWN, 2>(a, cx, 1, st, cur);
    march5_iter<BC, MASK, MACRO, PF, DOWN, 3>(a, cx, 2, st, cur);
    march5_iter<BC, MASK, MACRO, PF, DOWN, 4>(a, cx, 3, st, cur);
    for (int i = 4; i < cx.n_iter; ++i) march5_iter<BC, MASK, MACRO, PF, DOWN, 5>(a, cx, i, st, cur);
}

// strips a grid of nx columns is cut into
constexpr int step5_strips(int nx) { return (nx + STEP5_VALID - 1) / STEP5_VALID; }

// Launch geometry as k_step4: one workgroup = one segment pair of one strip (two waves), XCD-transposed order, shorter segments
// for the two wall-column strips.
template <int BC, bool MASK, bool MACRO, bool PF>
__global__ __launch_bounds__(64 * STEP4_WAVES, 2) void k_step5(const StepArgs a, int strips, int seg_rows, int nsegs, int row_end)
{
    __shared__ f4a lds_win[STEP4_WAVES][PF ? 24 : 18][64];
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int item = xcd_item(blockIdx.x, gridDim.x);
    const unsigned slot = __builtin_amdgcn_s_getreg((4 << 11) | 4) & 1u;
    // items [0, strips * nsegs): pair item / strips of strip item % strips; behind them, in a box with walls at its left and right
    // end (a.edge_seg_rows > 0): further pairs of the first and the last strip, which get shorter segments (k_step4)
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
    if (ya >= row_end) return;                          // (both waves of the workgroup: the barriers stay matched)
    const int yb = min(ya + seg_rows, row_end);
    const int ym = ya + (yb - ya) / 2;                  // the pair's middle line: wave 0 marches down from it, wave 1 up
    const int x0 = sx * STEP5_VALID - STEP5_SKIRT;
    if (wy == 0) march5<BC, MASK, MACRO, PF, true>(a, x0, ym, ym - ya, 0, lds_win[0], lds_win[1], slot);
    else march5<BC, MASK, MACRO, PF, false>(a, x0, ym, yb - ym, 1, lds_win[1], lds_win[0], slot);
}

}  // namespace
