// kernels_deep2.h -- k_deep's march with TWO waves per strip and direction, so that two waves share every SIMD (round 6).
//
// Why.  k_deep runs one wave per SIMD -- 40 KB of LDS windows and ~280 registers per wave leave room for no second one -- and a lone
// wave issues one instruction per ~5 cycles whatever the instruction: the vector ALU is busy 65 % of the time, the launch is bound
// by instruction issue.  Round 5 priced a second wave per SIMD at "nothing" from two DIFFERENT kernels;
// round 6 ran the SAME kernel both ways (k_deep<4>, 213 registers, 16 KB: profiles/r06_occ2_probe.txt, 8192^2, arithmetic only, no
// global memory): four waves per CU 767 us per launch, eight waves per CU 415 us -- of which ~1.4 x is the second wave's doing (a
// quarter of the four-waves build's waves had doubled up on SIMDs and set its launch time: profiles/r06_occ2_placement.txt).  A SIMD fed
// by two waves issues their scalar, LDS and memory instructions beside the other's vector instructions and covers their stalls.
//
// How.  A wave's state must halve.  So the D stages of a strip's march are split between two waves that share the strip: the FRONT
// wave gathers the rows from memory (one row ahead, in a register window it waits for by hand: kernels_deep.h) and runs stages
// 1..F, the BACK wave runs stages F+1..D and stores.  Each is an ordinary k_deep march (kernels_deep.h: deep_iter with ROLE) of depth F
// / D - F + 1 -- the back wave's "step 1" being the row the front wave hands it through ONE 9.25 KB slot of LDS (nine links + the
// row's obstacle flags) --, with its own stage windows (registers + LDS) and its own partner of the other direction: a workgroup is
// four waves, front-down, front-up, back-down, back-up; the two front waves hand each other the links that cross the pair's middle
// line while their pipelines fill, and so do the two back waves, exactly as k_deep's two waves do.  The back waves run one trip
// behind the front waves.  Per trip, two workgroup barriers: (1) the back waves have taken the slot's row -- the front waves may
// overwrite it at the end of their iteration --, (2) the front waves' rows and everybody's published links are visible.
//
// Resources per wave at D = 7, F = 4: front 2 register windows + 1 LDS window (8 KB) + the slot its row gathered ahead is loaded
// into (10 KB: the row in flight is in NO register -- `buffer_load ... lds`, kernels_deep.h: deep_row_issue_lds); back 2 register windows
// + 1 LDS window (8 KB); the hand-over slot 10 KB per direction: 72 KB per workgroup, two workgroups = eight waves per CU; 256 vector
// registers per wave, no accumulation register.
//
// The launch moves the same 72 B per cell; same cell functions, same operations in the same order: bitwise equal to k_step.
#pragma once

namespace {

// registers windows of the front / back wave, per depth and split (see the header)
constexpr int deep2_split(int D) { return D >= 8 ? 4 : (D + 1) / 2; }          // F: stages of the front wave (7 -> 4, 6 -> 3)
constexpr int deep2_rw_front(int D) { return deep2_split(D) - 2 >= 2 ? 2 : 1; }
constexpr int deep2_rw_back(int D) { return (D - deep2_split(D) + 1) - 2 >= 2 ? 2 : 1; }
constexpr int DEEP2_WAVES = 4;

// The trip's two barriers (see the header).  NOT __syncthreads(): that is a fence as well -- `s_waitcnt vmcnt(0)` in front of the
// barrier --, and at every trip a back wave would wait out the nine stores it has just issued, a front wave the row it has just asked
// for: the whole workgroup stalls on memory latency twice per row (the first k_deep2, profiles/r06_deep2_first.txt: 0.92-0.95 ms per
// 8192^2 launch where k_deep<7> takes 0.89).  What the waves hand each other goes through LDS: an LDS instruction that has completed
// (lgkmcnt) is visible to the workgroup.
__device__ __forceinline__ void deep2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the front wave: stages 1..F of positions 0 .. len + D - 2, one per trip from trip 0 ---------------------------------------
template <int BC, bool MASK, int F, int RW, bool DOWN, int NST>
__device__ __forceinline__ void deep2_front_fill(const StepArgs &a, const DeepCtx &cx, DeepState<RW, F - 1 - RW> &st, Row1 &ra, Row1 &rb,
                                                 int &trip)
{
    if constexpr (NST < F) {
        deep2_barrier();
        if (trip < cx.n_iter) {
            if ((NST & 1) == 0) deep_iter<BC, MASK, false, F, RW, 1, DOWN, NST, -1, DEEP_FRONT>(a, cx, NST - 1, st, rb, ra);
            else deep_iter<BC, MASK, false, F, RW, 1, DOWN, NST, -1, DEEP_FRONT>(a, cx, NST - 1, st, ra, rb);
        }
        deep2_barrier();
        ++trip;
        deep2_front_fill<BC, MASK, F, RW, DOWN, NST + 1>(a, cx, st, ra, rb, trip);
    }
}

template <int BC, bool MASK, int D, int F, int RW, bool DOWN>
__device__ __forceinline__ void deep2_front(const StepArgs &a, const int x0, const int ym, const int len, const int trips, f4a (*mine)[64],
                                            f4a (*other)[64], f4a (*ho)[64], f4a (*dma)[64])
{
    DeepCtx cx;
    cx.lane = threadIdx.x;
    const int xr = x0 + cx.lane * 4;
    constexpr int SKL = deep_skirt_lanes(D);
    if (BC == LB_BC_PERIODIC) cx.x4 = xr < 0 ? xr + a.nx : (xr >= a.nx ? (xr - a.nx < 4 * SKL ? xr - a.nx : 4 * (SKL - 1)) : xr);
    else cx.x4 = min(max(xr, 0), (a.nx - 1) & ~3);
    cx.store_lane = false;                              // (a front wave stores nothing)
    cx.ym = ym; cx.n_iter = len + D - 1;                // every position the BACK wave's last stage needs
    cx.mine = mine; cx.other = other; cx.ho = ho; cx.dma = dma;
    // (a pointer into LDS as a 64-bit number: the aperture's base above, the byte offset inside the workgroup's LDS below)
    cx.dma_off = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<unsigned long long>(dma));
    DeepState<RW, F - 1 - RW> st = {};
    auto row_at = [&](int p) { return DOWN ? ym - 1 - p : ym + p; };
    Row1 ra, rb;
    deep_row_issue_lds<BC, MASK>(a, row_at(0), cx.x4, cx.dma_off, ra);
    int trip = 0;
    deep2_front_fill<BC, MASK, F, RW, DOWN, 1>(a, cx, st, ra, rb, trip);
    if ((F - 1) & 1) ra = rb;                           // (position F - 1 is in ra or rb by its parity)
    for (; trip < trips; ++trip) {
#ifdef LB_DIAG
        if (!(a.diag & (1 << 24)))                      // timing only: no barriers in the steady state (races: wrong results)
#endif
        deep2_barrier();
        if (trip < cx.n_iter) {
            deep_iter<BC, MASK, false, F, RW, 1, DOWN, F, -1, DEEP_FRONT>(a, cx, trip, st, ra, rb);
            ra = rb;
        }
#ifdef LB_DIAG
        if (!(a.diag & (1 << 24)))
#endif
        deep2_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the row gathered behind the last position)
}

// ---- the back wave: stages F+1..D, one trip behind: its iteration j (position j enters) runs in trip j + F ------------------------
template <bool MASK>
__device__ __forceinline__ void deep2_take(const DeepCtx &cx, Row1 &cur)
{
#pragma unroll
    for (int k = 0; k < 9; ++k) cur.q[k] = cx.ho[k][cx.lane];
    cur.mk = uc4{0, 0, 0, 0};
    if (MASK) cur.mk = __builtin_bit_cast(uc4, reinterpret_cast<const unsigned *>(cx.ho[9])[cx.lane]);
    cur.have = true;
}

template <int BC, bool MASK, bool MACRO, int DB, int RW, bool DOWN, int NST>
__device__ __forceinline__ void deep2_back_fill(const StepArgs &a, const DeepCtx &cx, DeepState<RW, DB - 1 - RW> &st, Row1 &cur, int &j)
{
    if constexpr (NST < DB) {
        if (j < cx.n_iter) deep2_take<MASK>(cx, cur);
        deep2_barrier();
        if (j < cx.n_iter) deep_iter<BC, MASK, MACRO, DB, RW, 0, DOWN, NST, -1, DEEP_BACK>(a, cx, NST - 1, st, cur, cur);
        deep2_barrier();
        ++j;
        deep2_back_fill<BC, MASK, MACRO, DB, RW, DOWN, NST + 1>(a, cx, st, cur, j);
    }
}

template <int BC, bool MASK, bool MACRO, int D, int F, int RW, bool DOWN>
__device__ __forceinline__ void deep2_back(const StepArgs &a, const int x0, const int ym, const int len, const int trips, f4a (*mine)[64],
                                           f4a (*other)[64], f4a (*ho)[64])
{
    constexpr int DB = D - F + 1;                       // the back wave's own depth: "step 1" = the row handed over
    DeepCtx cx;
    cx.lane = threadIdx.x;
    const int xr = x0 + cx.lane * 4;
    constexpr int SKL = deep_skirt_lanes(D);
    if (BC == LB_BC_PERIODIC) cx.x4 = xr < 0 ? xr + a.nx : (xr >= a.nx ? (xr - a.nx < 4 * SKL ? xr - a.nx : 4 * (SKL - 1)) : xr);
    else cx.x4 = min(max(xr, 0), (a.nx - 1) & ~3);
    cx.store_lane = cx.lane >= SKL && cx.lane <= 63 - SKL && xr < a.nx;
    cx.ym = ym; cx.n_iter = len + DB - 1;
    cx.mine = mine; cx.other = other; cx.ho = ho;
    DeepState<RW, DB - 1 - RW> st = {};
    Row1 cur;
    cur.wp = WrapPatch{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    cur.hsolid = false; cur.hxc = -1; cur.rr = 0;
    // trips 0 .. F - 1: the front wave's pipeline fills, nothing has been handed over yet
    for (int t = 0; t < F; ++t) {
        deep2_barrier();
        deep2_barrier();
    }
    int j = 0;
    deep2_back_fill<BC, MASK, MACRO, DB, RW, DOWN, 1>(a, cx, st, cur, j);
    for (; j + F < trips; ++j) {
        if (j < cx.n_iter) deep2_take<MASK>(cx, cur);
#ifdef LB_DIAG
        if (!(a.diag & (1 << 24)))
#endif
        deep2_barrier();
        if (j < cx.n_iter) deep_iter<BC, MASK, MACRO, DB, RW, 0, DOWN, DB, -1, DEEP_BACK>(a, cx, j, st, cur, cur);
#ifdef LB_DIAG
        if (!(a.diag & (1 << 24)))
#endif
        deep2_barrier();
    }
}

// Launch geometry as k_deep: one workgroup = one segment pair of one strip, now four waves; XCD-transposed order, shorter segments for
// the two wall-column strips.  LDS: front 2 x (F - 1 - RWF) windows, back 2 x (D - F - RWB) windows, 2 slots.
template <int BC, bool MASK, bool MACRO, int D>
__global__ __launch_bounds__(64 * DEEP2_WAVES, 2) void k_deep2(const StepArgs a, int strips, int seg_rows, int nsegs, int row_end)
{
    constexpr int F = deep2_split(D), RWF = deep2_rw_front(D), RWB = deep2_rw_back(D);
    constexpr int LF = (F - 1 - RWF) * DEEP_WSLOTS, LBK = (D - F - RWB) * DEEP_WSLOTS;
    static_assert(2 * (LF + LBK + 2 * DEEP_HO_SLOTS) <= 80, "two workgroups per CU: 80 KB each");
    __shared__ f4a lds_dma[2][DEEP_HO_SLOTS][64];
    __shared__ f4a lds_front[2][LF][64];
    __shared__ f4a lds_back[2][LBK][64];
    __shared__ f4a lds_ho[2][DEEP_HO_SLOTS][64];
    // 0 front-down, 1 front-up, 2 back-down, 3 back-up.  A workgroup's wave w lands on SIMD w, and the two workgroups of a CU are (in
    // launch order) 256 apart: every other 256 workgroups take the roles two waves on, so that a SIMD holds a front wave (four stages
    // and the gather) AND a back wave (three stages and the stores), not two of a kind -- LB_DEEP2_SWAP 0: off (A/B).
#ifndef LB_DEEP2_SWAP
#define LB_DEEP2_SWAP 1
#endif
    const int wy = (__builtin_amdgcn_readfirstlane(threadIdx.y) + (LB_DEEP2_SWAP ? ((blockIdx.x >> 8) & 1) * 2 : 0)) & 3;
    const int item = xcd_item(blockIdx.x, gridDim.x);
#ifdef LB_DIAG
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
#endif
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
    if (ya >= row_end) return;                          // (all four waves of the workgroup: the barriers stay matched)
    const int yb = min(ya + seg_rows, row_end);
    const int ym = ya + (yb - ya) / 2;                  // the pair's middle line: the down waves march down from it, the up waves up
    const int x0 = sx * deep_valid(D) - 4 * deep_skirt_lanes(D);
    const int trips = max(ym - ya, yb - ym) + D;        // front: len + D - 1 iterations from trip 0; back: len + D - F from trip F
    if (wy == 0) deep2_front<BC, MASK, D, F, RWF, true>(a, x0, ym, ym - ya, trips, lds_front[0], lds_front[1], lds_ho[0], lds_dma[0]);
    else if (wy == 1) deep2_front<BC, MASK, D, F, RWF, false>(a, x0, ym, yb - ym, trips, lds_front[1], lds_front[0], lds_ho[1], lds_dma[1]);
    else if (wy == 2) deep2_back<BC, MASK, MACRO, D, F, RWB, true>(a, x0, ym, ym - ya, trips, lds_back[0], lds_back[1], lds_ho[0]);
    else deep2_back<BC, MASK, MACRO, D, F, RWB, false>(a, x0, ym, yb - ym, trips, lds_back[1], lds_back[0], lds_ho[1]);
#ifdef LB_DIAG
    if ((a.diag & 4096) && threadIdx.x == 0) {
        // per-wave timeline as k_deep's (tools/wave_timeline.py, LB_TIMELINE_DEEP2=1): four records per item; [6] = item * 4 + role
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        unsigned *o = reinterpret_cast<unsigned *>(a.rho) + 8 * (item * DEEP2_WAVES + wy);
        o[0] = (unsigned)diag_t0; o[1] = (unsigned)(diag_t0 >> 32); o[2] = (unsigned)t1; o[3] = (unsigned)(t1 >> 32);
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
        o[6] = (unsigned)(item * DEEP2_WAVES + wy); o[7] = (unsigned)((wy & 1) ? yb - ym : ym - ya);
    }
#endif
}

}  // namespace
