// kernels_step6.h -- SIX time steps per pass.  Included by lb_hip.cpp after kernels_step5.h, whose design it extends by one stage
// (overlapping strips, segment pairs, peeled pipeline fill: read that header first).
//
// Why.  k_step5's launch costs what its access pattern costs: the same with six instead of eight waves per CU, and the same with
// or without a row-ahead gather (profiles/r04_experiments.txt sections 13, 14).  Six waves per CU leave each wave 26 KB of LDS instead
// of 20: room for one more stage window.  The windows between steps 1/2 and 2/3 stay in registers, the one between 3/4 keeps its
// link 3 in registers and the rest in LDS (its ring + links 0, 1: 8 slots), those between 4/5 and 5/6 live in LDS (9 slots each):
// 26 KB per wave, 52 KB per workgroup, three workgroups per CU.  The skirt is five cells deep, i.e. two lanes: strips are laid 240 cells apart and
// start 8 cells early, lanes 2..61 are stored.  The launch moves the same 72 B per cell for six steps.  Same cell functions:
// bitwise equal to k_step.
#pragma once

// diagnostic build (tools/ablate.py; timing only, wrong results): LB_DIAG bit 0 = no cell arithmetic in any stage, bit 21 = strips
// 256 cells apart without skirts (rows start on 1-KiB boundaries), bit 22 = no stores, bit 23 = only the first row of a segment is
// loaded
#ifdef LB_DIAG
#define LB_DIAG_NOCOLLIDE if (!(a.diag & 1))
#else
#define LB_DIAG_NOCOLLIDE
#endif

namespace {

constexpr int STEP6_SKIRT = 8;                          // cells a strip starts before / ends behind its stored cells (two lanes)
constexpr int STEP6_VALID = STRIP_W - 2 * STEP6_SKIRT;  // 240 cells stored per strip and row
constexpr int step6_strips(int nx) { return (nx + STEP6_VALID - 1) / STEP6_VALID; }

struct March6State {
    Window w1, w2;                      // stage windows between steps 1/2 and 2/3 (registers)
    f4a w3d3;                           // window 3: link 3 of its newest row (links 0, 1 and its ring: LDS)
    unsigned mhist;                     // obstacle-mask history (per byte: bit j = the row loaded j iterations ago, j = 1..5)
};
struct March6Ctx {
    int lane, x4, ym, n_iter, wy;
    bool store_lane;
    unsigned slot;
    f4a (*R3)[64], (*W4)[64], (*W5)[64];        // my window 3 (ring: 6 slots, links 0, 1: 2 slots), my windows 4 and 5 (9 slots each)
    f4a (*Q3)[64], (*P4)[64], (*P5)[64];        // the other wave's
};

// ring of a window whose links 0,1,3 live in registers: two rows of three slots
__device__ __forceinline__ void ring_load(f4a (*R)[64], int lane, int it, Window &w)
{
    const int gs = 3 * (it & 1);
    w.g2 = R[gs][lane]; w.g5 = R[gs + 1][lane]; w.g6 = R[gs + 2][lane];
}
template <bool DOWN>
__device__ __forceinline__ void ring_push(f4a (*R)[64], int lane, int it, const f4a (&q)[9])
{
    typedef Dir<DOWN> D;
    const int gs = 3 * (it & 1);
    R[gs][lane] = q[D::A]; R[gs + 1][lane] = q[D::B]; R[gs + 2][lane] = q[D::C];
}

// One iteration: position i is loaded and takes step 1, position i-1 step 2 (window 1), i-2 step 3 (window 2), i-3 step 4 (window 3),
// i-4 step 5 (window 4), i-5 step 6 (window 5; stored).  NST = number of stages that have a row: 1..5 in iterations 0..4 (code of
// their own: the pipeline fills, the two waves of the pair hand over), 6 in the loop.
// PFD = rows gathered ahead (0: the row is loaded where it is consumed; 1, 2: `cur` holds position i on entry, gathered PFD
// iterations ago, and position i + PFD is gathered into `nxt` at the top of the iteration -- the form for ONE wave per SIMD, whose
// 512 registers hold the rows in flight and who has no second wave to cover its waits).
// PAR (PFD >= 1, steady iterations in pairs): the parity of i as a constant -- the ring slots of the LDS windows become immediate
// offsets, and the two row buffers swap roles from one iteration to the next instead of being copied.
template <int BC, bool MASK, bool MACRO, int PFD, bool DOWN, int NST, int PAR = -1>
__device__ __forceinline__ void march6_iter(const StepArgs &a, const March6Ctx &cx, const int i_, March6State &st, Row1 &cur,
                                            Row1 &nxt)
{
    const int lane = cx.lane, x4 = cx.x4;
    const long long S = a.plane;
    const int i = NST < 6 ? NST - 1 : i_;
    const int it = PAR >= 0 ? PAR : i;               // (only its parity is used)
    auto row_at = [&](int p) { return DOWN ? cx.ym - 1 - p : cx.ym + p; };
    f4a(*W4)[64] = cx.W4;
    f4a(*W5)[64] = cx.W5;
    Window &w1 = st.w1, &w2 = st.w2;

    if (PFD == 0 && a.prio_turns > 0 && (i & 3) == 0) {   // (one wave per SIMD -- PFD >= 1 -- has nobody to take turns with)            // the two waves of a SIMD take turns at the higher priority (march4_iter)
        const unsigned turn = (unsigned)(__builtin_amdgcn_s_memrealtime() >> a.prio_turns) & 1u;
        if (turn == cx.slot) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
    // ---- what the other wave published for "position -1" in its previous iteration ---------------------------
    // (steps 1 and 2 of its position 0 come through my still idle window 5 -- slots 6..8, then 3..5 -- into my register windows;
    //  steps 3, 4, 5 went straight into the ring slots of my LDS windows)
    if (NST == 2) { w1.g2 = W5[6][lane]; w1.g5 = W5[7][lane]; w1.g6 = W5[8][lane]; }
    if (NST == 3) { w2.g2 = W5[3][lane]; w2.g5 = W5[4][lane]; w2.g6 = W5[5][lane]; }
    // ---- step 1 of position i (from memory) --------------------------------------------------------------------
    // (behind the last position the last row is gathered again -- a cache hit that nobody consumes: no condition on i)
#ifdef LB_DIAG
    if (!((a.diag & (1 << 23)) && i > 0))
#endif
    {
    if (PFD) row1_load<BC, MASK>(a, row_at(min(i + PFD, cx.n_iter - 1)), x4, false, 0, nxt);
    else row1_load<BC, MASK>(a, row_at(i), x4, false, 0, cur);
    }
    f4a (&q1)[9] = cur.q;
    f4a r4, u4, v4;
    const uc4 mk = cur.mk;
    if (cur.have) {
        gather_merge<BC, true>(a, x4, q1, cur.wp);
        LB_DIAG_NOCOLLIDE collide_row<BC, MASK>(a, x4, a.y0 + cur.rr, q1, mk, r4, u4, v4);
    }
    if (NST == 1) lds_publish<DOWN>(cx.P5, lane, 6, q1);        // my position 0 after step 1 -> the other wave's window 1 (mailbox)
    // ---- step 2 of position i-1 (window 1, registers) ----------------------------------------------------------
    f4a q2[9];
    if (NST >= 2) {
        int r2, t0_, t1_;
        (void)step1_rows(a, row_at(i - 1), r2, t0_, t1_);
        skirt_gather<DOWN>(w1, q1, q2);
        window_push_dir<DOWN>(w1, q1);
        LB_DIAG_NOCOLLIDE collide_row<BC, MASK>(a, x4, a.y0 + r2, q2, mask_bits(st.mhist, 1), r4, u4, v4);
        if (NST == 2) lds_publish<DOWN>(cx.P5, lane, 3, q2);    // my position 0 after step 2 -> the other wave's window 2 (mailbox)
    } else {
        window_push_dir<DOWN>(w1, q1);
    }
    // ---- step 3 of position i-2 (window 2, registers) ----------------------------------------------------------
    f4a q3[9];
    if (NST >= 3) {
        int r3, t0_, t1_;
        (void)step1_rows(a, row_at(i - 2), r3, t0_, t1_);
        skirt_gather<DOWN>(w2, q2, q3);
        window_push_dir<DOWN>(w2, q2);
        LB_DIAG_NOCOLLIDE collide_row<BC, MASK>(a, x4, a.y0 + r3, q3, mask_bits(st.mhist, 2), r4, u4, v4);
        if (NST == 3) lds_publish<DOWN>(cx.Q3, lane, 3, q3);    // my position 0 after step 3 -> the other wave's ring of window 3
    } else if (NST == 2) {
        window_push_dir<DOWN>(w2, q2);
    }
    // ---- step 4 of position i-3 (window 3: links 0,1,3 in registers, ring in LDS) ---------------------------------
    f4a q4[9];
    if (NST >= 4) {
        int r4_, t0_, t1_;
        (void)step1_rows(a, row_at(i - 3), r4_, t0_, t1_);
        Window w3;
        w3.d0 = cx.R3[6][lane]; w3.d1 = cx.R3[7][lane]; w3.d3 = st.w3d3;
        ring_load(cx.R3, lane, it, w3);
        skirt_gather<DOWN>(w3, q3, q4);
        ring_push<DOWN>(cx.R3, lane, it, q3);
        cx.R3[6][lane] = q3[0]; cx.R3[7][lane] = q3[1]; st.w3d3 = q3[3];
        LB_DIAG_NOCOLLIDE collide_row<BC, MASK>(a, x4, a.y0 + r4_, q4, mask_bits(st.mhist, 3), r4, u4, v4);
        if (NST == 4) lds_publish<DOWN>(cx.P4, lane, 3, q4);    // my position 0 after step 4 -> the other wave's window 4
    } else if (NST == 3) {
        ring_push<DOWN>(cx.R3, lane, it, q3);       // position 0 after step 3: ring row of even iterations (the other wave fills the odd one)
        cx.R3[6][lane] = q3[0]; cx.R3[7][lane] = q3[1]; st.w3d3 = q3[3];
    }
    // ---- step 5 of position i-4 (window 4, LDS) ----------------------------------------------------------------
    f4a q5[9];
    if (NST >= 5) {
        int r5, t0_, t1_;
        (void)step1_rows(a, row_at(i - 4), r5, t0_, t1_);
        Window w4;
        lds_window_load(W4, lane, it, w4);
        skirt_gather<DOWN>(w4, q4, q5);
        lds_window_push<DOWN>(W4, lane, it, q4);
        LB_DIAG_NOCOLLIDE collide_row<BC, MASK>(a, x4, a.y0 + r5, q5, mask_bits(st.mhist, 4), r4, u4, v4);
        if (NST == 5) lds_publish<DOWN>(cx.P5, lane, 6, q5);    // my position 0 after step 5 -> the other wave's window 5
    } else if (NST == 4) {
        lds_window_push<DOWN>(W4, lane, it, q4);    // position 0 after step 4: the d slots and ring slot 6 (the other wave fills slot 3)
    }
    // ---- step 6 of position i-5 (window 5, LDS), stored --------------------------------------------------------
    if (NST >= 6) {
        int r6, t0_, t1_;
        (void)step1_rows(a, row_at(i - 5), r6, t0_, t1_);
        Window w5;
        lds_window_load(W5, lane, it, w5);
        f4a t[9];
        skirt_gather<DOWN>(w5, q5, t);
        lds_window_push<DOWN>(W5, lane, it, q5);
        LB_DIAG_NOCOLLIDE collide_row<BC, MASK>(a, x4, a.y0 + r6, t, mask_bits(st.mhist, 5), r4, u4, v4);
#ifdef LB_DIAG
        if (!(a.diag & (1 << 22)))
#endif
        if (cx.store_lane) {
            const long long o = (long long)r6 * a.pitch;    // row start, uniform
            float *d = a.dst + o;
            store_row9<false>(a.nts != 0, d, S, x4, t);
            if (MACRO) {
                const long long m = (long long)r6 * a.fpitch;
                store4<false>(lane_ptr(a.rho + m, x4), r4);
                store4<false>(lane_ptr(a.u + m, x4), u4);
                store4<false>(lane_ptr(a.v + m, x4), v4);
            }
        }
    } else if (NST == 5) {
        lds_window_push<DOWN>(W5, lane, it, q5);    // position 0 after step 5: the d slots and ring slot 3 (the other wave fills slot 6)
    }
    if (MASK) st.mhist = ((st.mhist | mask_word(mk)) << 1) & 0x3e3e3e3eu;
    if (NST < 6) __syncthreads();                   // what was published in this iteration is consumed in the next
}

// One wave's march: columns [x0, x0 + 256) of which [x0 + 8, x0 + 248) are stored, `len` rows from the pair's middle line `ym`
// upward or downward; len + 5 iterations.
template <int BC, bool MASK, bool MACRO, int PFD, bool DOWN>
__device__ __forceinline__ void march6(const StepArgs &a, const int x0, const int ym, const int len, const int wy,
                                       f4a (*mine)[64], f4a (*other)[64], const unsigned slot)
{
    March6Ctx cx;
    cx.lane = threadIdx.x;
    const int xr = x0 + cx.lane * 4;                 // true column of my first cell: -8 .. ; may lie beyond either end of the box
    // lanes beyond an end of the box: the periodic images as far as the skirt reaches (behind it: the last image lane's lines), or
    // -- walls -- copies of the lane at that end
    if (BC == LB_BC_PERIODIC) cx.x4 = xr < 0 ? xr + a.nx : (xr >= a.nx ? (xr - a.nx < STEP6_SKIRT ? xr - a.nx : 4) : xr);
    else cx.x4 = min(max(xr, 0), (a.nx - 1) & ~3);
    cx.store_lane = cx.lane >= 2 && cx.lane <= 61 && xr < a.nx;
#ifdef LB_DIAG
    if (a.diag & (1 << 21)) { cx.x4 = xr; cx.store_lane = true; }
#endif
    cx.ym = ym; cx.n_iter = len + 5; cx.wy = wy; cx.slot = slot;
    cx.R3 = mine; cx.W4 = mine + 8; cx.W5 = mine + 17;          // (slots: window 3's ring + its links 0, 1; window 4; window 5)
    cx.Q3 = other; cx.P4 = other + 8; cx.P5 = other + 17;
    March6State st = {};
    auto row_at = [&](int p) { return DOWN ? ym - 1 - p : ym + p; };
    Row1 ra, rb, rc;
    if (PFD >= 1) row1_load<BC, MASK>(a, row_at(0), cx.x4, false, 0, ra);
    if (PFD >= 2) row1_load<BC, MASK>(a, row_at(1), cx.x4, false, 0, rb);
    if (PFD == 1) {
        // rows in flight: one; the two buffers swap roles every iteration (position i in ra for even i, in rb for odd i)
        march6_iter<BC, MASK, MACRO, PFD, DOWN, 1>(a, cx, 0, st, ra, rb);
        march6_iter<BC, MASK, MACRO, PFD, DOWN, 2>(a, cx, 1, st, rb, ra);
        march6_iter<BC, MASK, MACRO, PFD, DOWN, 3>(a, cx, 2, st, ra, rb);
        march6_iter<BC, MASK, MACRO, PFD, DOWN, 4>(a, cx, 3, st, rb, ra);
        march6_iter<BC, MASK, MACRO, PFD, DOWN, 5>(a, cx, 4, st, ra, rb);
        int i = 5;
        for (; i + 1 < cx.n_iter; i += 2) {
            march6_iter<BC, MASK, MACRO, PFD, DOWN, 6, 1>(a, cx, i, st, rb, ra);
            march6_iter<BC, MASK, MACRO, PFD, DOWN, 6, 0>(a, cx, i + 1, st, ra, rb);
        }
        if (i < cx.n_iter) march6_iter<BC, MASK, MACRO, PFD, DOWN, 6, 1>(a, cx, i, st, rb, ra);
        return;
    }
    // rows in flight: ra = position i, (PFD = 2: rb = i + 1,) the newest one lands in rc
#define LB_M6(NST, I)                                                                                   \
    do {                                                                                                \
        march6_iter<BC, MASK, MACRO, PFD, DOWN, NST>(a, cx, I, st, ra, PFD == 2 ? rc : ra);             \
        if (PFD == 2) { ra = rb; rb = rc; }                                                             \
    } while (0)
    LB_M6(1, 0); LB_M6(2, 1); LB_M6(3, 2); LB_M6(4, 3); LB_M6(5, 4);
    for (int i = 5; i < cx.n_iter; ++i) LB_M6(6, i);
#undef LB_M6
}

// Launch geometry as k_step5: one workgroup = one segment pair of one strip (two waves), XCD-transposed order, shorter segments
// for the two wall-column strips.  48 KB of LDS: three workgroups per CU.
template <int BC, bool MASK, bool MACRO, int PFD = 0>
__global__ __launch_bounds__(64 * STEP4_WAVES, (PFD ? 1 : 2)) void k_step6(const StepArgs a, int strips, int seg_rows, int nsegs, int row_end)
{
    __shared__ f4a lds_win[STEP4_WAVES][26][64];
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
    if (ya >= row_end) return;                          // (both waves of the workgroup: the barriers stay matched)
    const int yb = min(ya + seg_rows, row_end);
    const int ym = ya + (yb - ya) / 2;                  // the pair's middle line: wave 0 marches down from it, wave 1 up
    int x0 = sx * STEP6_VALID - STEP6_SKIRT;
#ifdef LB_DIAG
    if (a.diag & (1 << 21)) x0 = sx * STRIP_W;
#endif
    if (wy == 0) march6<BC, MASK, MACRO, PFD, true>(a, x0, ym, ym - ya, 0, lds_win[0], lds_win[1], slot);
    else march6<BC, MASK, MACRO, PFD, false>(a, x0, ym, yb - ym, 1, lds_win[1], lds_win[0], slot);
}

}  // namespace
