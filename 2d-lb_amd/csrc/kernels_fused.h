// kernels_fused.h -- the hot path: k_step (one time step per launch), k_step2 / k_step3 (two / three
// time steps per launch: wave-private strips marching in y with register windows) and the float4
// copy used for calibration.  Included by lb_hip.cpp only.
#pragma once
#include <type_traits>
#include "d2q9_cell.h"

// (external linkage: the argument block crosses translation units -- the host side in lb_hip.cpp fills it, the launchers of
//  launchers.h, one translation unit per kernel family, hand it to their kernels)
struct StepArgs {
    const float *src;      // plane 0, row 0, x 0 of the lattice being read
    float *dst;            // same element of the lattice being written
    const uint8_t *mask;   // [H][fpitch] or nullptr
    float *rho, *u, *v;    // [H][fpitch]
    long long plane;       // lattice: plane stride, floats
    int pitch;             // lattice: row stride, floats (rows interleave the nine planes: 9 * fpitch; planar layout: fpitch)
    int fpitch;            // row pitch of the fields and the mask = padded row width, floats / bytes
    int nx, ny;            // global grid
    int y0, h;             // slab origin / height
    int row_begin, row_step, row_count;  // local rows visited: row_begin + i*row_step, i < row_count
    int wrap_y;            // 1: periodic in y inside one slab: wrap the source rows locally; 2: VELOCITY_INLET: the wall
                           //    rows pull from row 1 / ny-2 in place of the row outside (bc_vel_cell)
    int ghost_s, ghost_n;  // slab has a neighbour below / above: ghost rows hold its edge rows
    int seg_stride;        // k_step2: first row of segment i = row_begin + i*seg_stride
    int edge_seg_rows;     // k_step4: > 0: the first and the last strip march segments of this many rows (see the kernel)
    int diag;              // ablation switches, read only by the LB_DIAG build (tools/ablate.py)
    int prio_turns;        // k_step4: the two waves of a SIMD alternate their issue priority (bit of the 100 MHz clock)
    int nts;               // marching kernels: non-temporal stores (a run-time flag there: halves their instantiations and takes a
                           // minute off the library's build)
    int tile_launch_order; // k_tile4: 1 = tile = blockIdx (A/B switch; default: one band of tile rows per XCD)
    float omega, rho_in, rho_out, lid_u, rho0;
    float u_w, u_e;        // VELOCITY_INLET: imposed speeds
    const float *corner;   // VELOCITY_INLET: the eight never-written corner links (bc_vel_cell)
};

// several lattices of one geometry advanced by one launch (k_step_batch)
constexpr int BATCH_MAX = 8;
struct BatchArgs {
    StepArgs a[BATCH_MAX];
};

namespace {


// Template value of the PIPE family run with the kernels of the reference's D2Q9i.cl fork (lb_params.semantics =
// LB_SEM_OPENCL_D2Q9I); not a public lb_bc_mode.
constexpr int LB_BC_PIPE_I = 4;

// The family's boundary rule for one cell (w, e, s, n: it lies in column 0 / nx-1, row 0 / ny-1).
template <int BC>
__device__ __forceinline__ void boundary_rule(const StepArgs &a, Cell &c, bool w, bool e, bool s, bool n)
{
    if (BC == LB_BC_PIPE) bc_pipe_cell(c, w, e, s, n, a.rho_in, a.rho_out);
    if (BC == LB_BC_PIPE_I) bc_pipe_i_cell(c, w, e, s, n, a.rho_in, a.rho_out);
    if (BC == LB_BC_CAVITY) bc_cavity_cell(c, w, e, s, n, a.lid_u, a.rho0);
    if (BC == LB_BC_VELOCITY_INLET && (w || e)) bc_vel_cell(c, w, e, s, n, a.u_w, a.u_e, a.corner);
}

// The same rule as a FUNCTION (one copy per kernel, called): k_deep's waves run alone on their SIMDs and pay for every byte of
// their loop in the instruction cache -- inlined at four call sites per stage the rule was ~450 instructions per stage of code that
// two strips in thirty-five and two rows in eight thousand ever execute (profiles/r05_experiments.txt section 6).  Values in,
// values out (nine registers each way); p0, p1 = the family's two parameters.
template <int BC>
__device__ __noinline__ Cell boundary_rule_call(Cell c, int wesn, float p0, float p1)
{
    const bool w = wesn & 1, e = wesn & 2, s = wesn & 4, n = wesn & 8;
    if (BC == LB_BC_PIPE) bc_pipe_cell(c, w, e, s, n, p0, p1);
    if (BC == LB_BC_PIPE_I) bc_pipe_i_cell(c, w, e, s, n, p0, p1);
    if (BC == LB_BC_CAVITY) bc_cavity_cell(c, w, e, s, n, p0, p1);
    return c;
}

// Obstacle swap, moments (with the family's overrides), equilibrium and relaxation of the cell at column x (wrapped
// into the box), local row yl -- the one sequence every fused kernel runs on every cell, vector or scalar.
template <int BC, bool MASK>
__device__ __forceinline__ void finish_cell(const StepArgs &a, int x, int yl, Cell &c, bool solid, float &rho, float &ux,
                                            float &uy)
{
    if (MASK) bounce_cell(c, solid);
    if (BC == LB_BC_VELOCITY_INLET) {
        // the cell on the inlet / outlet column takes its moments from the rule and the stored fields (D2Q9.cl:323-374)
        moments_cell(c, rho, ux, uy);
        if (x == 0 || x == a.nx - 1) {
            const long long o = (long long)yl * a.fpitch + x;
            const int yg = a.y0 + yl;
            vel_moments_cell(c, x == 0, yg == 0, yg == a.ny - 1, a.u_w, a.u_e, a.u[o], a.v[o], rho, ux, uy);
        }
        equilibrate_cell(c, a.omega, rho, ux, uy);
    } else if (BC == LB_BC_PIPE_I) {
        moments_i_cell(c, rho, ux, uy);
        if (MASK && solid) { ux = 0.f; uy = 0.f; }      // opencl_dim_D2Q9i.py:494-503: u, v re-zeroed in the obstacle every step
        equilibrate_i_cell(c, a.omega, rho, ux, uy);
    } else {
        relax_cell(c, a.omega, rho, ux, uy);
    }
}

// Pull-stream gather for 4 consecutive cells (x4..x4+3) of local row yl: q[k] = f_k at (x - cx_k,
// y - cy_k) of the source lattice.  ym / yp are the source rows of the cy=+1 / cy=-1 links (already
// wrapped by the caller where the box is periodic in y within this slab).
// Element x (per lane, >= 0) of a row whose start is the same in all lanes: written as uniform pointer +
// 32-bit byte offset so that the access takes the scalar-base form (global_load v, v_off, s[base:base+1])
// instead of a 64-bit address computed per lane -- one offset register serves all nine planes.
template <typename T>
__device__ __forceinline__ T *lane_ptr(T *row, int x)
{
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
    return reinterpret_cast<T *>(reinterpret_cast<B *>(row) + (unsigned)x * (unsigned)sizeof(T));
}

// Periodic x wrap: the lane holding x = 0 / x = nx-1 fetches the one element per plane that its displaced load takes from
// the row padding.  In two halves: gather_issue issues these loads FIRST, into temporaries, then the nine plane loads;
// gather_merge puts the temporaries in place.  (Written as a patch behind the plane loads -- q[1].x = ... -- each waited for
// the plane load it overwrites: a second, serial memory round trip per row for the two strips at the box's ends, which were
// the last to finish in every launch of the marching kernels.  And merged right behind the loads, the selects make the wave
// wait for them on the spot, which undoes k_step4's one-row-ahead gather for those two strips -- again the stragglers,
// +7 % on the launch: tools/wave_timeline.py.  k_step4 therefore merges at the point of use.)
struct WrapPatch {
    float p1, p5, p8, w3, w6, w7;
};

template <int BC, bool MASK, bool NTL>
__device__ __forceinline__ void gather_issue(const StepArgs &a, int x4, int yl, int ym, int yp, f4a (&q)[9], uc4 &mk,
                                             WrapPatch &wp)
{
    const long long P = a.pitch, S = a.plane;
    const float *s = a.src;
    const float *r0 = s + (long long)yl * P, *rm = s + (long long)ym * P, *rp = s + (long long)yp * P;   // uniform
    wp = WrapPatch{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef LB_DIAG
    if (a.diag & 8) {                      // timing only: all nine planes read aligned (wrong results)
        q[0] = load4<NTL>(lane_ptr(r0, x4));          q[1] = load4<NTL>(lane_ptr(r0 + 1 * S, x4));
        q[2] = load4<NTL>(lane_ptr(rm + 2 * S, x4));  q[3] = load4<NTL>(lane_ptr(r0 + 3 * S, x4));
        q[4] = load4<NTL>(lane_ptr(rp + 4 * S, x4));  q[5] = load4<NTL>(lane_ptr(rm + 5 * S, x4));
        q[6] = load4<NTL>(lane_ptr(rm + 6 * S, x4));  q[7] = load4<NTL>(lane_ptr(rp + 7 * S, x4));
        q[8] = load4<NTL>(lane_ptr(rp + 8 * S, x4));
        mk = uc4{0, 0, 0, 0};
        return;
    }
#endif
    if (BC == LB_BC_PERIODIC) {
        const int c = a.nx - 1 - x4;
        bool wrap_w = x4 == 0, wrap_e = c >= 0 && c < 4;
#ifdef LB_DIAG
        if (a.diag & 16384) wrap_w = wrap_e = false;        // timing only: no periodic-wrap patch (wrong results)
#endif
        if (wrap_w) {
            wp.p1 = s[1 * S + (long long)yl * P + a.nx - 1];
            wp.p5 = s[5 * S + (long long)ym * P + a.nx - 1];
            wp.p8 = s[8 * S + (long long)yp * P + a.nx - 1];
        }
        if (wrap_e) {
            wp.w3 = s[3 * S + (long long)yl * P];
            wp.w6 = s[6 * S + (long long)ym * P];
            wp.w7 = s[7 * S + (long long)yp * P];
        }
    }
    q[0] = load4<NTL>(lane_ptr(r0, x4));
    q[1] = load4u<NTL>(lane_ptr(r0 + 1 * S - 1, x4));
    q[2] = load4<NTL>(lane_ptr(rm + 2 * S, x4));
    q[3] = load4u<NTL>(lane_ptr(r0 + 3 * S + 1, x4));
    q[4] = load4<NTL>(lane_ptr(rp + 4 * S, x4));
    q[5] = load4u<NTL>(lane_ptr(rm + 5 * S - 1, x4));
    q[6] = load4u<NTL>(lane_ptr(rm + 6 * S + 1, x4));
    q[7] = load4u<NTL>(lane_ptr(rp + 7 * S + 1, x4));
    q[8] = load4u<NTL>(lane_ptr(rp + 8 * S - 1, x4));
    mk = uc4{0, 0, 0, 0};
    if (MASK) mk = *reinterpret_cast<const uc4 *>(lane_ptr(a.mask + (long long)yl * a.fpitch, x4));
}

// (ALIGNED4: the caller guarantees nx % 4 == 0 -- the marching kernels in a periodic box: the lane that holds x = nx-1 holds it in
//  its last component, so three selects do what twelve do for a general width)
template <int BC, bool ALIGNED4 = false>
__device__ __forceinline__ void gather_merge(const StepArgs &a, int x4, f4a (&q)[9], const WrapPatch &wp)
{
    if (BC == LB_BC_PERIODIC && ALIGNED4) {
        bool wrap_w = x4 == 0, wrap_e = x4 == a.nx - 4;
#ifdef LB_DIAG
        if (a.diag & (8 | 16384)) wrap_w = wrap_e = false;
#endif
        q[1].x = wrap_w ? wp.p1 : q[1].x;
        q[5].x = wrap_w ? wp.p5 : q[5].x;
        q[8].x = wrap_w ? wp.p8 : q[8].x;
        q[3].w = wrap_e ? wp.w3 : q[3].w;
        q[6].w = wrap_e ? wp.w6 : q[6].w;
        q[7].w = wrap_e ? wp.w7 : q[7].w;
    } else if (BC == LB_BC_PERIODIC) {
        const int c = a.nx - 1 - x4;
        bool wrap_w = x4 == 0, wrap_e = c >= 0 && c < 4;
#ifdef LB_DIAG
        if (a.diag & (8 | 16384)) wrap_w = wrap_e = false;
#endif
        q[1].x = wrap_w ? wp.p1 : q[1].x;
        q[5].x = wrap_w ? wp.p5 : q[5].x;
        q[8].x = wrap_w ? wp.p8 : q[8].x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool hit = wrap_e && j == c;
            q[3][j] = hit ? wp.w3 : q[3][j];
            q[6][j] = hit ? wp.w6 : q[6][j];
            q[7][j] = hit ? wp.w7 : q[7][j];
        }
    }
}

template <int BC, bool MASK, bool NTL>
__device__ __forceinline__ void gather_row(const StepArgs &a, int x4, int yl, int ym, int yp, f4a (&q)[9], uc4 &mk)
{
    WrapPatch wp;
    gather_issue<BC, MASK, NTL>(a, x4, yl, ym, yp, q, mk, wp);
    gather_merge<BC>(a, x4, q, wp);
}

// The nine 16-byte stores of a row of the marching kernels, non-temporal or plain by a RUN-TIME, wave-uniform flag (as a template
// argument the choice doubled those kernels' instantiations: a minute and a half of the library's build).  The empty asm keeps
// the two branches from being merged: merged, the stores lose the non-temporal hint (the compiler keeps only what both carry)
// -- measured: 8192^2 277 k instead of 309 k MLUPS.
template <bool NTS>
__device__ __forceinline__ void store_row9(bool nts, float *d, long long S, int x4, const f4a (&t)[9])
{
    // Addressing: ONE scalar base (the row) and a 32-bit per-lane offset that carries the plane as well -- (x + k S) * 4 bytes
    // stays below 4 GB for every lattice the marching kernels are launched on (marching_planes_fit, lb_hip.cpp).  With nine
    // bases d + k S the compiler, once the two branches below share them, computes nine 64-bit per-lane addresses in front of the
    // branch (instruction selection works block by block and then sees nine opaque pointers): 18 registers and 36 vector adds
    // per row.  The lane offset passes through an empty asm in either branch so that the sums are not shared either.
    const int kS = (int)S;
    if (NTS || nts) {
        asm volatile("" : "+v"(x4) : : "memory");       // (in front as well: common code is hoisted out of branches, too)
#pragma unroll
        for (int k = 0; k < 9; ++k) store4<true>(lane_ptr(d, x4 + k * kS), t[k]);
        asm volatile("" ::: "memory");
    } else {
        asm volatile("" : "+v"(x4));
#pragma unroll
        for (int k = 0; k < 9; ++k) store4<false>(lane_ptr(d, x4 + k * kS), t[k]);
    }
}

// Boundary rule, obstacle swap, moments, equilibrium and relaxation of the 4 gathered cells, in place.
template <int BC, bool MASK, bool OOL = false>
__device__ __forceinline__ void collide_row(const StepArgs &a, int x4, int yg, f4a (&q)[9], uc4 mk, f4a &r4,
                                            f4a &u4, f4a &v4)
{
    // Phase 1, branchy and rare: the boundary rule, for the cells ON the boundary only -- all four in the
    // wall rows y = 0, ny-1 (wave-uniform), otherwise the one cell of the one lane that holds x = 0 or
    // x = nx-1.  (Running it for all four cells of those lanes made the two wall-column strips the
    // stragglers of every launch: -11 % at 8192^2, profiles/r01_ablation.txt.)
    if (BC != LB_BC_PERIODIC) {
        const bool south = (yg == 0), north = (yg == a.ny - 1);
        bool wall_row = (BC != LB_BC_VELOCITY_INLET) && (south || north);   // (that family's wall rows need no rule: their pull is remapped)
        bool first = (x4 == 0);
        bool last = (x4 <= a.nx - 1 && a.nx - 1 < x4 + 4);
        const int jl = (a.nx - 1) & 3;
#ifdef LB_DIAG
        if (a.diag & 512) wall_row = first = last = false;    // timing only: no boundary rule
#endif
        if (wall_row || first || last) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (wall_row || (first && j == 0) || (last && j == jl)) {
                    Cell c = {q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j], q[6][j], q[7][j], q[8][j]};
                    const bool w = first && j == 0, e = last && j == jl;
                    if constexpr (OOL && (BC == LB_BC_PIPE || BC == LB_BC_PIPE_I || BC == LB_BC_CAVITY)) {
                        const bool cav = (BC == LB_BC_CAVITY);
                        c = boundary_rule_call<BC>(c, (w ? 1 : 0) | (e ? 2 : 0) | (south ? 4 : 0) | (north ? 8 : 0),
                                                   cav ? a.lid_u : a.rho_in, cav ? a.rho0 : a.rho_out);
                    } else {
                        boundary_rule<BC>(a, c, w, e, south, north);
                    }
                    q[0][j] = c.f0; q[1][j] = c.f1; q[2][j] = c.f2; q[3][j] = c.f3; q[4][j] = c.f4;
                    q[5][j] = c.f5; q[6][j] = c.f6; q[7][j] = c.f7; q[8][j] = c.f8;
                }
            }
        }
    }
    // Phase 2, straight-line: obstacle swap (selects), moments, equilibrium, relaxation of the four cells.
    // Kept free of control flow so that the compiler pairs the cells into packed fp32 instructions
    // (v_pk_fma/mul/add_f32: 523 packed ops in the periodic kernel against 105 when the boundary branches
    // sat inside this loop -- the wall families ran 7 % slower for that alone).
    if (BC == LB_BC_VELOCITY_INLET) {
        // moments of the four cells, then -- rare: the one lane per row that holds x = 0 or x = nx-1 -- the column's overrides
        // (D2Q9.cl:323-374), then the relaxation of the four cells: the branch sits between the two straight-line parts
        Cell c[4];
        float rho[4], ux[4], uy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c[j] = Cell{q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j], q[6][j], q[7][j], q[8][j]};
            if (MASK) bounce_cell(c[j], mk[j] != 0);
            moments_cell(c[j], rho[j], ux[j], uy[j]);
        }
        const bool first = (x4 == 0), last = (x4 <= a.nx - 1 && a.nx - 1 < x4 + 4);
        if (first || last) {
            const int jl = (a.nx - 1) & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((first && j == 0) || (last && j == jl)) {
                    const long long o = (long long)(yg - a.y0) * a.fpitch + x4 + j;
                    vel_moments_cell(c[j], first && j == 0, yg == 0, yg == a.ny - 1, a.u_w, a.u_e, a.u[o], a.v[o], rho[j], ux[j],
                                     uy[j]);
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            equilibrate_cell(c[j], a.omega, rho[j], ux[j], uy[j]);
            r4[j] = rho[j]; u4[j] = ux[j]; v4[j] = uy[j];
            q[0][j] = c[j].f0; q[1][j] = c[j].f1; q[2][j] = c[j].f2; q[3][j] = c[j].f3; q[4][j] = c[j].f4;
            q[5][j] = c[j].f5; q[6][j] = c[j].f6; q[7][j] = c[j].f7; q[8][j] = c[j].f8;
        }
        return;
    }
    if (BC == LB_BC_PIPE_I) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Cell c = {q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j], q[6][j], q[7][j], q[8][j]};
            float rho, ux, uy;
            finish_cell<BC, MASK>(a, x4 + j, yg - a.y0, c, mk[j] != 0, rho, ux, uy);
            r4[j] = rho; u4[j] = ux; v4[j] = uy;
            q[0][j] = c.f0; q[1][j] = c.f1; q[2][j] = c.f2; q[3][j] = c.f3; q[4][j] = c.f4;
            q[5][j] = c.f5; q[6][j] = c.f6; q[7][j] = c.f7; q[8][j] = c.f8;
        }
        return;
    }
    // the plain families: the four cells as two pairs (d2q9_cell.h, T = f2a), each in the aligned register pair its 16-byte
    // load put it in; finish_cell's sequence -- obstacle swap, moments, equilibrium, relaxation -- on both cells of a pair at once
    // (round 5 tried the obstacle swap behind a wave-uniform "some lane holds a solid cell in this row" test -- 94 % of a wave's rows
    //  are all fluid under the porous-medium image of BASELINE config 5 --: the join of the two paths costs more register moves than the
    //  32 selects it skips; masked kernels 5-12 % slower, sparse masks included: profiles/r05_experiments.txt section 8.  Twice more
    //  at the end of the round, section 20: the swap in a divergent block of its own that all-fluid waves branch around -- no renamed
    //  registers, ~5 vector instructions on the all-fluid path --: still 3-7 % slower under the porous image, the cylinder and a random
    //  mask alike (a lone wave pays for every taken branch and for the scheduling barrier a block boundary is); and v_swap_b32 under the
    //  execution mask through inline asm, whose operands -- sub-registers of the 128-bit row registers -- the compiler copies in and out)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f2a f[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) f[k] = h ? q[k].zw : q[k].xy;
        if (MASK) {
            const bool s0 = mk[2 * h] != 0, s1 = mk[2 * h + 1] != 0;
            auto swap2 = [&](f2a &p, f2a &o) {              // bounce_cell on a pair: exchange opposite links on solid cells
                const f2a pp = p, oo = o;
                p = f2a{s0 ? oo.x : pp.x, s1 ? oo.y : pp.y};
                o = f2a{s0 ? pp.x : oo.x, s1 ? pp.y : oo.y};
            };
            swap2(f[1], f[3]); swap2(f[2], f[4]); swap2(f[5], f[7]); swap2(f[6], f[8]);
        }
        f2a rho, ux, uy;
        moments_t<f2a>(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], rho, ux, uy);
        equilibrate_t<f2a>(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], a.omega, rho, ux, uy);
        if (h) { r4.zw = rho; u4.zw = ux; v4.zw = uy; }
        else   { r4.xy = rho; u4.xy = ux; v4.xy = uy; }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (h) q[k].zw = f[k];
            else q[k].xy = f[k];
        }
    }
}

// The hot kernel: one full time step for 4 consecutive cells of one row per lane.
//   pull-stream (move+copy_buffer) -> boundary rule (move_bcs) -> obstacle swap
//   (bounceback_in_obstacle) -> moments (update_hydro) -> equilibrium (update_feq) ->
//   relaxation (collide_particles), then 9 aligned 16-byte stores.
// Launch: blockDim = (64, RW): a wave covers 256 cells of one row, RW rows per block.
//
// XCD-aware tile order (XCD = true): workgroups are dealt round-robin over the 8 XCDs, each with its
// own L2.  A misaligned 1 KiB wave read touches 9 cache lines, the 9th shared with the wave to its
// right; if that neighbour runs on another XCD the line is fetched from HBM twice (measured: +8.3 %
// FETCH_SIZE = 6/9 planes x 1/8).  Remapping the linear workgroup id so that every XCD sweeps its own
// contiguous band of rows keeps x-neighbours on one L2.  Only speed depends on it, never results.
template <int BC, bool MASK, bool MACRO, bool NTL, bool NTS, bool XCD>
__device__ __forceinline__ void step_body(const StepArgs &a)
{
    int bx = blockIdx.x, by = blockIdx.y;
    if (XCD) {
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned per = total >> 3;                 // tiles per XCD (tail handled below)
        if (lin < (per << 3)) {
            const unsigned t = (lin & 7u) * per + (lin >> 3);
            bx = t % gridDim.x;
            by = t / gridDim.x;
        }
    }
    const int x4 = (bx * blockDim.x + threadIdx.x) * 4;
    const int ri = by * blockDim.y + __builtin_amdgcn_readfirstlane(threadIdx.y);   // uniform: blockDim.x % 64 == 0
    if (x4 >= a.fpitch || ri >= a.row_count) return;
    const int yl = a.row_begin + ri * a.row_step;
    const int yg = a.y0 + yl;
    int ym = yl - 1, yp = yl + 1;          // source rows of the cy=+1 / cy=-1 links
    if (a.wrap_y == 1) {
        if (ym < 0) ym = a.h - 1;
        if (yp >= a.h) yp = 0;
    } else if (a.wrap_y == 2) {
        if (ym < 0) ym = a.h - 2;
        if (yp >= a.h) yp = 1;
    }
    const long long o0 = (long long)yl * a.pitch;       // row start, uniform
    f4a q[9], r4, u4, v4;
    uc4 mk;
    gather_row<BC, MASK, NTL>(a, x4, yl, ym, yp, q, mk);
#ifdef LB_DIAG
    if (!(a.diag & 1))
#endif
    collide_row<BC, MASK>(a, x4, yg, q, mk, r4, u4, v4);

    const long long S = a.plane;
    float *d = a.dst + o0;
#pragma unroll
    for (int k = 0; k < 9; ++k) store4<NTS>(lane_ptr(d + k * S, x4), q[k]);
    if (MACRO) {
        const long long m0 = (long long)yl * a.fpitch;
        store4<false>(lane_ptr(a.rho + m0, x4), r4);
        store4<false>(lane_ptr(a.u + m0, x4), u4);
        store4<false>(lane_ptr(a.v + m0, x4), v4);
    }
}

template <int BC, bool MASK, bool MACRO, bool NTL, bool NTS, bool XCD>
__global__ __launch_bounds__(256) void k_step(const StepArgs a)
{
    step_body<BC, MASK, MACRO, NTL, NTS, XCD>(a);
}

// Several lattices of one geometry advanced by ONE launch, blockIdx.z = which: the periodic multi-population sets of the
// reference's research forks (porous_media/single_component.cl:338-375 `move_periodic` streams population `cur_field`
// of a [jumper][population][y][x] array; here every population is a lattice of its own with its own relaxation rate,
// and stream + collide are fused as everywhere).  Same cell code as k_step: bitwise equal to separate launches.
template <int BC, bool MASK, bool MACRO>
__global__ __launch_bounds__(256) void k_step_batch(const BatchArgs b)
{
    step_body<BC, MASK, MACRO, false, false, false>(b.a[blockIdx.z]);
}

// ---- two time steps per pass ------------------------------------------------------------------
// Temporal blocking without LDS.  A wave owns a strip of 256 cells (64 lanes x 4, 1 KiB-aligned) and
// marches up a segment of rows.  For every row r it computes step 1 (gather from the source lattice +
// collide: exactly gather_row/collide_row above) and keeps the result in registers; the second step of
// row y = r-1 needs, per link k, the step-1 value of ONE row only (cy=-1: row r, just computed; cy=0:
// row r-1; cy=+1: row r-2), so a register window of 3+6 float4 holds everything, and the x-neighbour
// a link comes from is one element to the left/right = a 1-lane shuffle.  The two cells just outside
// the strip (x0-1 and x0+256) are recomputed by the edge lanes 0 and 63 as a fifth, scalar cell, so
// waves never exchange anything, every strip is a whole number of cache lines and every store is a
// full aligned 1 KiB.  (A first version used lanes 0/63 as halo lanes and advanced strips by 248
// cells: simpler, but its 992-byte store segments and 34-instead-of-32 strips cost 10-18 %:
// tools/ablate.py, profiles/r01_ablation.txt.)  HBM traffic per two updates of a cell: 9 reads +
// 9 writes (+2 rows per segment), i.e. ~37 B per lattice update instead of 72.
constexpr int STRIP_W = 256;       // cells per wave-row

// Workgroup -> work item order of the marching kernels.  Workgroups are dealt round-robin over the 8
// XCDs; transposing every 8x8 block of workgroup ids puts 8 consecutive items (= up to 32 adjacent
// strips, one segment row at nx = 8192) on ONE XCD, so the cache line a strip shares with its
// neighbour is fetched into one L2 instead of two (+1.8 % at 8192^2, profiles/r01_ablation.txt).
// Only speed depends on it.
__device__ __forceinline__ int xcd_item(int wg, int nwg)
{
    const int blk = wg & ~63, i = wg & 63;
    return (blk + 64 <= nwg) ? blk + (i & 7) * 8 + (i >> 3) : wg;
}

// value of the cell one to the LEFT of each of my 4 cells (links with cx = +1); lane 0 takes the
// strip's left halo cell
__device__ __forceinline__ f4a from_left(f4a v, float halo, int lane)
{
    float w = __shfl_up(v.w, 1);                // left lane's last cell
    if (lane == 0) w = halo;
    return f4a{w, v.x, v.y, v.z};
}
// value of the cell one to the RIGHT of each of my 4 cells (links with cx = -1); lane 63 takes the
// strip's right halo cell
__device__ __forceinline__ f4a from_right(f4a v, float halo, int lane)
{
    float x = __shfl_down(v.x, 1);              // right lane's first cell
    if (lane == 63) x = halo;
    return f4a{v.y, v.z, v.w, x};
}

// Resolve the rows step 1 of row r reads.  Rows -1 and H are wrapped (whole periodic grid on this
// GPU), read from the ghost rows (slab with a neighbour on that side: rows -2..H+1 hold valid halo
// data) or skipped: returns false when the row lies outside a wall (its values are never consumed
// un-overwritten).
__device__ __forceinline__ bool step1_rows(const StepArgs &a, int r, int &rr, int &ym, int &yp)
{
    // (selects, no branches: everything here is wave-uniform, and as early returns it cut the marching kernels' loop bodies into
    //  two dozen basic blocks -- instruction selection then no longer saw a row's base and its lane offset in one block)
    // wrap_y: 1 = the grid is periodic in y on this GPU (period h); 2 = VELOCITY_INLET: rows 0 and h-1 share their vertical links,
    // i.e. the rows beyond a wall row are the rows on the far side of the OTHER wall row: row -1 is row h-2, row h is row 1
    // (period h-1); 0 = rows beyond the slab are ghost rows or lie outside a wall
    const int h = a.h, w = a.wrap_y, per = h - (w == 2 ? 1 : 0);
    const int wr = r < 0 ? r + per : (r >= h ? r - per : r);
    rr = w ? wr : r;
    const int m = rr - 1, p = rr + 1;
    ym = (w && m < 0) ? h - w : m;              // (w = 1: row h-1; w = 2: row h-2)
    yp = (w && p >= h) ? w - 1 : p;             // (w = 1: row 0; w = 2: row 1)
    return w || (r < 0 ? a.ghost_s != 0 : (r >= h ? a.ghost_n != 0 : true));
}

// Step 1 of the single cell (hx, row rr): the strip's halo cell, executed by one edge lane.  Same
// arithmetic as collide_row, so the value equals what the neighbouring strip computes for that cell.
// In two halves so that the loads can be issued a row ahead of their use (k_step4's prefetch):
// halo_cell_load = the nine populations the cell pulls (+ its obstacle flag; xc = its wrapped column, -1 =
// outside a walled box: zeros, don't-care), halo_cell_finish = boundary rule, obstacle swap, relaxation.
template <int BC, bool MASK>
__device__ __forceinline__ void halo_cell_load(const StepArgs &a, int hx, int rr, int ym, int yp, Cell &c, bool &solid,
                                               int &xc)
{
    solid = false;
    xc = hx;
    int xl = hx - 1, xg = hx + 1;
    if (BC == LB_BC_PERIODIC) {
        xc = hx < 0 ? hx + a.nx : (hx >= a.nx ? hx - a.nx : hx);
        xl = xc - 1 < 0 ? a.nx - 1 : xc - 1;
        xg = xc + 1 >= a.nx ? 0 : xc + 1;
    } else if (hx < 0 || hx >= a.nx) {
        c = Cell{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};    // outside the box: don't-care
        xc = -1;
        return;
    }
    const long long P = a.pitch, S = a.plane;
    const float *s = a.src;
    const float *r0 = s + (long long)rr * P, *rm = s + (long long)ym * P, *rp = s + (long long)yp * P;   // uniform
    c.f0 = *lane_ptr(r0, xc);
    c.f1 = *lane_ptr(r0 + 1 * S, xl);
    c.f2 = *lane_ptr(rm + 2 * S, xc);
    c.f3 = *lane_ptr(r0 + 3 * S, xg);
    c.f4 = *lane_ptr(rp + 4 * S, xc);
    c.f5 = *lane_ptr(rm + 5 * S, xl);
    c.f6 = *lane_ptr(rm + 6 * S, xg);
    c.f7 = *lane_ptr(rp + 7 * S, xg);
    c.f8 = *lane_ptr(rp + 8 * S, xl);
    if (MASK) solid = *lane_ptr(a.mask + (long long)rr * a.fpitch, xc) != 0;
}

template <int BC, bool MASK>
__device__ __forceinline__ void halo_cell_finish(const StepArgs &a, int xc, int rr, Cell &c, bool solid)
{
    if (BC != LB_BC_PERIODIC && xc < 0) return;
    const int yg = a.y0 + rr;
    if (BC != LB_BC_PERIODIC && (yg == 0 || yg == a.ny - 1 || xc == 0 || xc == a.nx - 1))
        boundary_rule<BC>(a, c, xc == 0, xc == a.nx - 1, yg == 0, yg == a.ny - 1);
    float rho, ux, uy;
    finish_cell<BC, MASK>(a, xc, rr, c, solid, rho, ux, uy);
}

template <int BC, bool MASK>
__device__ __forceinline__ void halo_cell_step1(const StepArgs &a, int hx, int rr, int ym, int yp, Cell &c,
                                                bool &solid)
{
    int xc;
    halo_cell_load<BC, MASK>(a, hx, rr, ym, yp, c, solid, xc);
    halo_cell_finish<BC, MASK>(a, xc, rr, c, solid);
}

// Segment i of a launch covers output rows [row_begin + i*seg_stride, +seg_rows) clipped to row_end:
// one contiguous range cut into equal shares (seg_stride == seg_rows), or the two edge bands of a slab
// (seg_stride = their distance) that are computed first so their halo can travel early.
template <int BC, bool MASK, bool MACRO, bool NTS>
__global__ __launch_bounds__(256, 2) void k_step2(const StepArgs a, int strips, int seg_rows, int nsegs, int row_end)
{
    const int lane = threadIdx.x;                       // blockDim = (64, 4): four independent waves
    // (threadIdx.y is the same in all 64 lanes of a wave, but the compiler does not know: without the
    //  readfirstlane the segment, the row loop and every row address would live in vector registers)
    const int item = xcd_item(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int sx = item % strips, sy = item / strips;
    if (sy >= nsegs) return;
    const int ya = a.row_begin + sy * a.seg_stride;
    if (ya >= row_end) return;
    const int yb = min(ya + seg_rows, row_end);
    const int x0 = sx * STRIP_W;
    const int xr = x0 + lane * 4;                       // true x of my first cell; may lie beyond nx
    int x4 = xr;                                        // x used for addressing
    if (BC == LB_BC_PERIODIC && xr >= a.nx) x4 = xr - a.nx;   // duplicates of cells 0.. (nx % 4 == 0)
    const bool store_lane = xr < a.nx;                  // lanes past the box compute don't-care values
    const bool edge_lane = (lane == 0) || (lane == 63);
    const int hx = (lane == 0) ? x0 - 1 : x0 + STRIP_W; // my halo cell (edge lanes only)
    const long long S = a.plane;

    f4a d0 = {}, d1 = {}, d3 = {};                      // step-1 links 0,1,3 of row r-1
    f4a e2 = {}, e5 = {}, e6 = {};                      // step-1 links 2,5,6 of row r-1
    f4a g2 = {}, g5 = {}, g6 = {};                      //                     of row r-2
    // the same window for the halo cell; lane 0 keeps the links entering from the left (1,5,8),
    // lane 63 those entering from the right (3,6,7)
    float hd = 0.f, he = 0.f, hg = 0.f;                 // cy=0 link of row r-1; cy=+1 link of rows r-1, r-2
    uc4 mk_prev = {0, 0, 0, 0};                         // obstacle mask of row r-1: loaded once, with that row's populations
    for (int r = ya - 1; r <= yb; ++r) {
        // ---- step 1 of row r ---------------------------------------------------------------------
        // (an explicit software prefetch of row r+1 was tried: +36 VGPR, -2 %; removing all
        // arithmetic does not make the kernel faster either: the memory pipeline is what the waves
        // wait for -- profiles/r01_ablation.txt)
        f4a q[9], r4, u4, v4;
        uc4 mk = {0, 0, 0, 0};
        int rr, ym, yp;
        const bool have = step1_rows(a, r, rr, ym, yp);
        float hq0 = 0.f, hq1 = 0.f, hq2 = 0.f;          // halo cell, row r: cy=0, cy=+1, cy=-1 link
        if (have) {
            gather_row<BC, MASK, false>(a, x4, rr, ym, yp, q, mk);
            if (edge_lane) {
                Cell hc;
                bool hsolid;
                halo_cell_step1<BC, MASK>(a, hx, rr, ym, yp, hc, hsolid);
                hq0 = lane == 0 ? hc.f1 : hc.f3;
                hq1 = lane == 0 ? hc.f5 : hc.f6;
                hq2 = lane == 0 ? hc.f8 : hc.f7;
            }
#ifdef LB_DIAG
            if (!(a.diag & 1))
#endif
            collide_row<BC, MASK>(a, x4, a.y0 + rr, q, mk, r4, u4, v4);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) q[k] = f4a{0.f, 0.f, 0.f, 0.f};
        }
        // ---- step 2 of row y = r-1 from the register window ---------------------------------------
        if (r >= ya + 1) {
            const int y = r - 1;
            f4a t[9];
            t[0] = d0;
            t[1] = from_left(d1, hd, lane);
            t[3] = from_right(d3, hd, lane);
            t[2] = g2;
            t[5] = from_left(g5, hg, lane);
            t[6] = from_right(g6, hg, lane);
            t[4] = q[4];
            t[7] = from_right(q[7], hq2, lane);
            t[8] = from_left(q[8], hq2, lane);
            const long long o = (long long)y * a.pitch;         // row start, uniform
#ifdef LB_DIAG
            if (!(a.diag & 2))
#endif
            collide_row<BC, MASK>(a, x4, a.y0 + y, t, mk_prev, r4, u4, v4);
#ifdef LB_DIAG
            if (a.diag & 4) {              // no stores: keep the values alive instead
#pragma unroll
                for (int k = 0; k < 9; ++k) asm volatile("" ::"v"(t[k]));
            } else
#endif
            if (store_lane) {
                float *d = a.dst + o;
                store_row9<NTS>(a.nts != 0, d, S, x4, t);
                if (MACRO) {
                    const long long m = (long long)y * a.fpitch;
                    store4<false>(lane_ptr(a.rho + m, x4), r4);
                    store4<false>(lane_ptr(a.u + m, x4), u4);
                    store4<false>(lane_ptr(a.v + m, x4), v4);
                }
            }
        }
        // ---- slide the window -----------------------------------------------------------------------
        g2 = e2; g5 = e5; g6 = e6;
        e2 = q[2]; e5 = q[5]; e6 = q[6];
        d0 = q[0]; d1 = q[1]; d3 = q[3];
        hg = he; he = hq1; hd = hq0;
        mk_prev = mk;
    }
}

// ---- three time steps per pass ----------------------------------------------------------------
// The two-step kernel is bound by HBM alone (removing all of its arithmetic does not speed it up), so
// a third step per pass is free bytes: the same march with one more register window.  Per row r:
// step 1 of row r (from memory), step 2 of row r-1 (from window 1), step 3 of row r-2 (from window 2,
// stored).  The edge lanes now recompute two cells beyond the strip for step 1 (x0-2, x0-1 |
// x0+256, x0+257) and one for step 2 (x0-1 | x0+256), all as scalar cells with the same arithmetic.
// On slabs it reads the neighbours' rows from the ghost zone (lb_run's halo cycle).  ~25 B of HBM traffic per
// lattice update.

// Post-collision links of one halo cell that later stages can ask for: the centre links (cx = 0) and
// the three links that point toward the strip (cx = +1 on the left side, -1 on the right side),
// each indexed by cy = 0, +1, -1.
struct HaloLinks {
    float c0, c2, c4;      // links 0, 2 (cy=+1), 4 (cy=-1)
    float t0, tp, tm;      // toward-strip links with cy = 0, +1, -1: (1,5,8) on the left, (3,6,7) on the right
};

__device__ __forceinline__ HaloLinks halo_links(const Cell &c, bool left)
{
    HaloLinks h;
    h.c0 = c.f0; h.c2 = c.f2; h.c4 = c.f4;
    h.t0 = left ? c.f1 : c.f3;
    h.tp = left ? c.f5 : c.f6;
    h.tm = left ? c.f8 : c.f7;
    return h;
}

// The 4-cell-wide register window one stage hands to the next (see k_step2).
struct Window {
    f4a d0, d1, d3;        // links 0,1,3 of the previous row
    f4a e2, e5, e6;        // links 2,5,6 of the previous row
    f4a g2, g5, g6;        //                 of the row before that
};
__device__ __forceinline__ void window_push(Window &w, const f4a (&q)[9])
{
    w.g2 = w.e2; w.g5 = w.e5; w.g6 = w.e6;
    w.e2 = q[2]; w.e5 = q[5]; w.e6 = q[6];
    w.d0 = q[0]; w.d1 = q[1]; w.d3 = q[3];
}
// The same for one halo cell (scalars): toward links always, centre links when a later stage
// recomputes this cell.
struct HaloWindow {
    float t0_d, tp_e, tp_g;     // toward links: cy=0 of the previous row; cy=+1 of the previous row / the one before
    float c0_d, c2_e, c2_g;     // centre links, same delays
};
__device__ __forceinline__ void halo_push(HaloWindow &w, const HaloLinks &h)
{
    w.tp_g = w.tp_e; w.tp_e = h.tp; w.t0_d = h.t0;
    w.c2_g = w.c2_e; w.c2_e = h.c2; w.c0_d = h.c0;
}

// Gather for the next stage of row y from the previous stage's window `w`, its newest row `q`
// (= row y+1) and the inner halo cell's toward links (window hw, newest row hnew).
__device__ __forceinline__ void window_gather(const Window &w, const f4a (&q)[9], const HaloWindow &hw,
                                              const HaloLinks &hnew, int lane, f4a (&t)[9])
{
    t[0] = w.d0;
    t[1] = from_left(w.d1, hw.t0_d, lane);
    t[3] = from_right(w.d3, hw.t0_d, lane);
    t[2] = w.g2;
    t[5] = from_left(w.g5, hw.tp_g, lane);
    t[6] = from_right(w.g6, hw.tp_g, lane);
    t[4] = q[4];
    t[7] = from_right(q[7], hnew.tm, lane);
    t[8] = from_left(q[8], hnew.tm, lane);
}

// obstacle-mask history of the three-step kernel (see mhist there)
__device__ __forceinline__ unsigned mask_word(uc4 m)
{
    return (m.x ? 1u : 0u) | (m.y ? 0x100u : 0u) | (m.z ? 0x10000u : 0u) | (m.w ? 0x1000000u : 0u);
}
__device__ __forceinline__ uc4 mask_bits(unsigned hist, int age)
{
    const unsigned b = (hist >> age) & 0x01010101u;
    return uc4{(unsigned char)(b & 0xff), (unsigned char)((b >> 8) & 0xff), (unsigned char)((b >> 16) & 0xff),
               (unsigned char)(b >> 24)};
}

template <int BC, bool MASK, bool MACRO, bool NTS>
__global__ __launch_bounds__(256, 2) void k_step3(const StepArgs a, int strips, int seg_rows, int nsegs, int row_end)
{
    const int lane = threadIdx.x;                       // blockDim = (64, 4): four independent waves
    // (threadIdx.y is the same in all 64 lanes of a wave, but the compiler does not know: without the
    //  readfirstlane the segment, the row loop and every row address would live in vector registers)
    const int item = xcd_item(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y);
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
    const int hxi = left ? x0 - 1 : x0 + STRIP_W;       // inner halo cell (adjacent to the strip)
    const int hxo = left ? x0 - 2 : x0 + STRIP_W + 1;   // outer halo cell
    int hxi_c = hxi;                                    // inner halo cell, wrapped, for mask / boundary tests
    if (BC == LB_BC_PERIODIC) hxi_c = hxi < 0 ? hxi + a.nx : (hxi >= a.nx ? hxi - a.nx : hxi);
    const bool hxi_in = (BC == LB_BC_PERIODIC) || (hxi >= 0 && hxi < a.nx);
    const long long S = a.plane;

    Window w1 = {}, w2 = {};                            // step-1 / step-2 results of my 4 cells
    HaloWindow hi1 = {}, ho1 = {}, hi2 = {};            // step 1 of the inner / outer halo cell, step 2 of the inner one
    // obstacle mask of rows r-1 and r-2 (and of the inner halo cell in row r-1): loaded once, together with
    // that row's populations, and handed down like the windows -- a load at the point of use stalls the wave
    // for a memory round trip in steps 2 and 3 of every row (-18 % with a mask at 8192^2)
    // (one register: in every byte bit 1 = my cell in row r-1, bit 2 = in row r-2; bit 6 of byte 0 = the inner
    //  halo cell in row r-1 -- the wall + obstacle instantiations sit at the 256-register limit)
    unsigned mhist = 0;
    for (int r = ya - 2; r <= yb + 1; ++r) {
        // ---- step 1 of row r (from memory) --------------------------------------------------------
        f4a q1[9], r4, u4, v4;
        uc4 mk = {0, 0, 0, 0};
        int rr, ym, yp;
        const bool have = step1_rows(a, r, rr, ym, yp);
        HaloLinks hi_new = {}, ho_new = {};
        bool hsolid = false;
        if (have) {
            gather_row<BC, MASK, false>(a, x4, rr, ym, yp, q1, mk);   // (non-temporal loads: no gain, measured)
            if (edge_lane) {
                Cell c;
                bool osolid;
                halo_cell_step1<BC, MASK>(a, hxi, rr, ym, yp, c, hsolid);
                hi_new = halo_links(c, left);
                halo_cell_step1<BC, MASK>(a, hxo, rr, ym, yp, c, osolid);
                ho_new = halo_links(c, left);
            }
#ifdef LB_DIAG
            if (!(a.diag & 1))
#endif
            collide_row<BC, MASK>(a, x4, a.y0 + rr, q1, mk, r4, u4, v4);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) q1[k] = f4a{0.f, 0.f, 0.f, 0.f};
        }
        // ---- step 2 of row r-1 (from window 1) ----------------------------------------------------
        // (skipped while the window is still filling, r < ya: nothing consumes those rows and their
        //  mask rows r-1 < ya-1 may not exist)
        f4a q2[9];
        HaloLinks h2_new = {};
        if (r >= ya) {
            int r2, r2m, r2p;
            (void)step1_rows(a, r - 1, r2, r2m, r2p);   // r2 = local row of r-1 (wrapped when periodic)
            window_gather(w1, q1, hi1, hi_new, lane, q2);
            if (edge_lane) {
                // the inner halo cell, step 2: centre links from itself, toward links from the outer halo
                // cell, the remaining three from the strip's own edge cell
                Cell c;
                c.f0 = hi1.c0_d; c.f2 = hi1.c2_g; c.f4 = hi_new.c4;
                const float a0 = ho1.t0_d, ap = ho1.tp_g, am = ho_new.tm;          // from the outer cell
                const float b0 = left ? w1.d3.x : w1.d1.w;                         // from my edge cell: cy = 0
                const float bp = left ? w1.g6.x : w1.g5.w;                         //                   cy = +1
                const float bm = left ? q1[7].x : q1[8].w;                         //                   cy = -1
                c.f1 = left ? a0 : b0; c.f3 = left ? b0 : a0;
                c.f5 = left ? ap : bp; c.f6 = left ? bp : ap;
                c.f8 = left ? am : bm; c.f7 = left ? bm : am;
                if (hxi_in) {
                    const int yg = a.y0 + r2;
                    if (BC != LB_BC_PERIODIC && (yg == 0 || yg == a.ny - 1 || hxi_c == 0 || hxi_c == a.nx - 1))
                        boundary_rule<BC>(a, c, hxi_c == 0, hxi_c == a.nx - 1, yg == 0, yg == a.ny - 1);
                    float rho, ux, uy;
                    finish_cell<BC, MASK>(a, hxi_c, r2, c, (mhist & 0x40u) != 0, rho, ux, uy);
                }
                h2_new = halo_links(c, left);
            }
#ifdef LB_DIAG
            if (!(a.diag & 2))
#endif
            collide_row<BC, MASK>(a, x4, a.y0 + r2, q2, mask_bits(mhist, 1), r4, u4, v4);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) q2[k] = f4a{0.f, 0.f, 0.f, 0.f};
        }
        // ---- step 3 of row r-2 (from window 2), stored ---------------------------------------------
        if (r >= ya + 2) {
            int r3, r3m, r3p;
            (void)step1_rows(a, r - 2, r3, r3m, r3p);
            f4a t[9];
            window_gather(w2, q2, hi2, h2_new, lane, t);
            const long long o = (long long)r3 * a.pitch;        // row start, uniform
#ifdef LB_DIAG
            if (!(a.diag & 4))
#endif
            collide_row<BC, MASK>(a, x4, a.y0 + r3, t, mask_bits(mhist, 2), r4, u4, v4);
            if (store_lane) {
                float *d = a.dst + o;
                store_row9<NTS>(a.nts != 0, d, S, x4, t);
                if (MACRO) {
                    const long long m = (long long)r3 * a.fpitch;
                    store4<false>(lane_ptr(a.rho + m, x4), r4);
                    store4<false>(lane_ptr(a.u + m, x4), u4);
                    store4<false>(lane_ptr(a.v + m, x4), v4);
                }
            }
        }
        // ---- slide the windows -------------------------------------------------------------------------
        window_push(w1, q1);
        window_push(w2, q2);
        halo_push(hi1, hi_new);
        halo_push(ho1, ho_new);
        halo_push(hi2, h2_new);
        if (MASK) mhist = ((mhist | mask_word(mk) | (hsolid ? 0x20u : 0u)) << 1) & 0x06060646u;
    }
}

// Calibration kernel: plain 16-byte-per-lane copy of n4 float4s, ONE float4 per thread and no loop -- the shape that
// reaches the device's streaming ceiling (6.3 TB/s; grid-stride loops with 8 loads in flight per lane stop at 5.2-5.8 TB/s
// whatever the grid: tools/copy_bench.hip, profiles/r02_experiments.txt).  Known byte count, used to (a) correct rocprofv3's
// FETCH_SIZE on gfx950 and (b) measure the streaming ceiling of the device the bench runs on.
template <bool NT>
__global__ __launch_bounds__(256) void k_copy4(const f4a *__restrict__ src, f4a *__restrict__ dst, long long n4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    else dst[i] = src[i];
}

}  // namespace
