// kernels_tile.h -- four time steps per pass for SMALL grids: 2-D tiles staged in LDS.
// Included by lb_hip.cpp after kernels_fused.h.
//
// Grids below ~1600^2 cells live in the Infinity Cache and are not bandwidth-bound: a single-step launch costs
// its ~2 us of dependent-kernel boundary plus one global-memory round trip per step, and the marching kernels
// (one global round trip per row) are worse.  Here a workgroup loads a region -- a 32 x 16 tile + 4 halo cells on every side = 40 x 24
// cells -- of all nine planes into LDS once, advances it four time steps without touching
// global memory -- after step s the outermost s rings hold stale data and are no longer computed --, and
// stores the tile.  Per step: every thread pulls the nine links of its cells out of LDS into
// registers, applies the boundary rule / obstacle swap / relaxation (the same cell functions as every other
// kernel: results are bitwise identical), and after a barrier writes the post-collision values back in place.
// Redundant work with 32 x 16 tiles: (40*24 + 38*22 + 36*20 + 34*18) / (4 * 512) = 1.53.  LDS: 9 x 960 floats + 960
// mask bytes = 35.5 KB per workgroup, four workgroups per CU.
#pragma once

namespace {

constexpr int TILE_T = 4;                   // time steps per pass = halo width

// TW x TH = the tile; the region held in LDS is (TW + 8) x (TH + 8).  Two shapes, picked by the host: 32 x 16
// (two cells per thread: 512 threads; one: 960) and, so that small grids still spread over the chip, 16 x 16.
// CPT = cells per thread: few, so that the kernel stays near 45-60 VGPR and several workgroups share a CU
template <int TW, int TH, int CPT_>
struct TileShape {
    static constexpr int LW = TW + 2 * TILE_T, LH = TH + 2 * TILE_T, CELLS = LW * LH;
    static constexpr int CPT = CPT_;
    static constexpr int THREADS = ((CELLS + CPT - 1) / CPT + 63) / 64 * 64;
};

// Which tile a workgroup takes.  Workgroups go to the eight XCDs round-robin (blockIdx % 8) and every XCD has an L2 of its
// own: in launch order a tile's neighbours all sit on OTHER XCDs, and the 88 % of halo cells a 40 x 24 region shares with
// them (plus the partial cache lines of its 160-byte rows) were fetched over the fabric once per XCD that touched them --
// 2.98 x the compulsory bytes at 1024^2 (profiles/r03_experiments.txt), which is what bounded the kernel.  Here XCD j takes
// the j-th eighth of the tiles in row-major order, a band of tile rows, in the order its workgroups are started: tiles
// that share halo cells run on one XCD at about the same time.  The grid is rounded up to eight equal shares (n_tiles
// need not divide): a workgroup whose index falls behind the last tile leaves at once.
__device__ __forceinline__ int xcd_band_tile(int b, int n_tiles)
{
    const int share = (n_tiles + 7) >> 3;
    return (b & 7) * share + (b >> 3);
}

// Which region cell a thread STEPS, as opposed to loads (the region goes into LDS in linear order, rows coalesced).  With two
// cells per thread: slot 0 = the tile itself, stepped four times; slot 1 = the halo rings from the inside out -- ring r is stepped
// in steps 1..r --, so the waves of slot 1 leave the later steps as wholes instead of every wave of the region keeping a few live
// lanes: 44 wave bodies per tile and four steps instead of 52 (the kernel is bound by its vector instructions, not by bytes:
// profiles/r03_experiments.txt section 15).  Ring 0 is never stepped.  false = no cell in this slot.
template <int TW, int TH>
__device__ __forceinline__ bool tile_step_cell(int tid, int slot, int &lx, int &ly)
{
    constexpr int L = TW + 2 * TILE_T, LH = TH + 2 * TILE_T;
    lx = ly = 0;
    if (slot == 0) {
        lx = TILE_T + tid % TW; ly = TILE_T + tid / TW;
        return tid < TW * TH;
    }
    int j = tid;
#pragma unroll
    for (int r = TILE_T - 1; r >= 1; --r) {
        const int W = L - 2 * r, H = LH - 2 * r, n = 2 * (W + H) - 4;
        if (j >= 0 && j < n) {
            if (j < W) { lx = r + j; ly = r; }                                          // its row nearest row 0
            else if (j < 2 * W) { lx = r + j - W; ly = LH - 1 - r; }                    // the opposite one
            else if (j < 2 * W + H - 2) { lx = r; ly = r + 1 + (j - 2 * W); }           // the columns between them
            else { lx = L - 1 - r; ly = r + 1 + (j - 2 * W - (H - 2)); }
            return true;
        }
        j -= n;
    }
    return false;
}

template <int BC, bool MASK, bool MACRO, int TW, int TH, int CPT>
__global__ __launch_bounds__((TileShape<TW, TH, CPT>::THREADS), (CPT == 2 ? 8 : 1)) void k_tile4(    // (two cells per thread: <= 64 VGPR, four 512-thread workgroups per CU, as the LDS allows)
    const StepArgs a, int tiles_x, int n_tiles)
{
    constexpr int TILE_L = TileShape<TW, TH, CPT>::LW, TILE_LH = TileShape<TW, TH, CPT>::LH;
    constexpr int TILE_CELLS = TileShape<TW, TH, CPT>::CELLS, TILE_THREADS = TileShape<TW, TH, CPT>::THREADS;
    constexpr int TILE_CPT = TileShape<TW, TH, CPT>::CPT;
    __shared__ float lds[9][TILE_CELLS];
    __shared__ unsigned char lmask[TILE_CELLS];
    const int tid = threadIdx.x;
    const int tile = a.tile_launch_order ? (int)blockIdx.x : xcd_band_tile(blockIdx.x, n_tiles);   // (A/B switch: variant bit 13)
    if (tile >= n_tiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int gx0 = tx * TW - TILE_T, gy0 = ty * TH - TILE_T;         // global coordinates of region cell (0,0)
    const long long P = a.pitch, S = a.plane;
#ifdef LB_DIAG
    const bool flip = (a.diag & 8) && ((blockIdx.x >> 3) & 1);
#else
    constexpr bool flip = false;
#endif

    // Wall cells (x = 0, nx-1; y = 0, ny-1) are not processed in the main pass but in one extra pass in which
    // thread t takes the t-th wall cell of the region (west column, east column, south row, north row; corners
    // belong to the columns): in the linear cell order every wave of a wall tile holds a few wall cells and
    // would execute the whole boundary rule four times per step.  Positions inside the region, -1 = this tile
    // does not touch that wall; all workgroup-uniform.
    int lxw = -1, lxe = -1, lys = -1, lyn = -1;
    if (BC != LB_BC_PERIODIC) {
        if (gx0 <= 0 && 0 < gx0 + TILE_L) lxw = -gx0;
        if (gx0 <= a.nx - 1 && a.nx - 1 < gx0 + TILE_L) lxe = a.nx - 1 - gx0;
        if (gy0 <= 0 && 0 < gy0 + TILE_LH) lys = -gy0;
        if (gy0 <= a.ny - 1 && a.ny - 1 < gy0 + TILE_LH) lyn = a.ny - 1 - gy0;
    }
    const bool wall_tile = BC != LB_BC_PERIODIC && (lxw >= 0 || lxe >= 0 || lys >= 0 || lyn >= 0);

    // The region into LDS -- thread t loads cells t, t + THREADS, ... in linear order c = ly * TILE_L + lx, rows coalesced -- and the
    // cells I step: region index cc, global (gx, gy) wrapped where the box is periodic, ring = how many steps the cell stays
    // in the computed part of the region (-1: never -- no cell, wall cells).  Two cells per thread: the tile and the rings
    // (tile_step_cell) -- except the pipe families with an obstacle mask, whose instantiations then spill 20-24 B per lane
    // within the 64 registers that four workgroups per CU allow and run 3-4 % slower than in the linear order.  Otherwise:
    // the cells I load.
    constexpr bool RINGS = (TILE_CPT == 2 && TILE_THREADS >= TW * TH && !(MASK && (BC == LB_BC_PIPE || BC == LB_BC_PIPE_I)));
    auto load_cell = [&](int c, int lx, int ly) {
        const int gx = gx0 + lx, gy = gy0 + ly;
        int sx, sy;                                     // where the cell's data comes from
        if (BC == LB_BC_PERIODIC) {
            sx = gx < 0 ? gx + a.nx : (gx >= a.nx ? gx - a.nx : gx);
            sy = gy < 0 ? gy + a.ny : (gy >= a.ny ? gy - a.ny : gy);
        } else {
            // cells outside a walled box are computed like any other, from copies of the nearest cells inside:
            // nothing valid consumes them (the boundary rule overwrites every link pulled from outside), and
            // the main pass needs no in-box bookkeeping
            sx = min(max(gx, 0), a.nx - 1);
            sy = min(max(gy, 0), a.ny - 1);
        }
        const long long o = (long long)sy * P + sx;
        if (!flip) {
#pragma unroll
            for (int k = 0; k < 9; ++k) lds[k][c] = a.src[k * S + o];
        }
        const bool inside = BC == LB_BC_PERIODIC || (sx == gx && sy == gy);
        lmask[c] = (MASK && inside) ? a.mask[(long long)sy * a.fpitch + sx] : 0;
    };
    int cc[TILE_CPT], gxs[TILE_CPT], gys[TILE_CPT], ring[TILE_CPT];
    bool mine[TILE_CPT];                                // mine to store: not a periodic image / inside the walled box
#pragma unroll
    for (int i = 0; i < TILE_CPT; ++i) {
        int lx, ly;
        bool have;
        if (RINGS) have = tile_step_cell<TW, TH>(tid, i, lx, ly);
        else {
            const int c = tid + i * TILE_THREADS;
            lx = c % TILE_L; ly = c / TILE_L;
            have = c < TILE_CELLS;
        }
        int gx = gx0 + lx, gy = gy0 + ly;
        if (BC == LB_BC_PERIODIC) {
            gx = gx < 0 ? gx + a.nx : (gx >= a.nx ? gx - a.nx : gx);
            gy = gy < 0 ? gy + a.ny : (gy >= a.ny ? gy - a.ny : gy);
        }
        cc[i] = ly * TILE_L + lx;
        gxs[i] = gx; gys[i] = gy;
        ring[i] = have ? min(min(lx, TILE_L - 1 - lx), min(ly, TILE_LH - 1 - ly)) : -1;
        if (BC != LB_BC_PERIODIC && (lx == lxw || lx == lxe || ly == lys || ly == lyn)) ring[i] = -1;
        mine[i] = BC == LB_BC_PERIODIC ? (gx0 + lx == gx && gy0 + ly == gy) : (gx >= 0 && gx < a.nx && gy >= 0 && gy < a.ny);
        if (!RINGS && have) load_cell(cc[i], lx, ly);
    }
    if (RINGS) {
#pragma unroll
        for (int i = 0; i < TILE_CPT; ++i) {
            const int c = tid + i * TILE_THREADS;
            if (c < TILE_CELLS) load_cell(c, c % TILE_L, c / TILE_L);
        }
    }
    __syncthreads();

    // my wall cell (thread t: [0,LH) west column, [LH,2LH) east, [2LH,2LH+L) south row, [2LH+L,2LH+2L) north).  Stepped by
    // rings, the wall cells go to the LAST threads of the workgroup: their waves have no ring cell in slot 1 (324 ring cells
    // for 512 threads), so the wall pass fills a hole instead of making the first two waves -- and with them, at the
    // barrier, the whole tile -- a body late in every step.
    constexpr int WALL_CELLS = 2 * (TILE_L + TILE_LH);
    constexpr int WALL_T0 = (RINGS && TILE_THREADS - WALL_CELLS >= (TILE_L - 2) * (TILE_LH - 2) - TW * TH) ? TILE_THREADS - WALL_CELLS : 0;
    int wlx = -1, wly = -1;
    if (wall_tile && tid >= WALL_T0) {
        int t = tid - WALL_T0;
        if (t < TILE_LH) { wlx = lxw; wly = t; }
        else if ((t -= TILE_LH) < TILE_LH) { wlx = lxe; wly = t; }
        else if ((t -= TILE_LH) < TILE_L) { wlx = t; wly = lys; if (t == lxw || t == lxe) wlx = -1; }
        else if ((t -= TILE_L) < TILE_L) { wlx = t; wly = lyn; if (t == lxw || t == lxe) wlx = -1; }
        if (wlx < 0 || wly < 0) wlx = wly = -1;
        else {
            const int gx = gx0 + wlx, gy = gy0 + wly;
            if (gx < 0 || gx >= a.nx || gy < 0 || gy >= a.ny) wlx = wly = -1;
        }
    }
    const int wc = wly * TILE_L + wlx;
    const int wring = wlx < 0 ? -1 : min(min(wlx, TILE_L - 1 - wlx), min(wly, TILE_LH - 1 - wly));

    // pull (+ boundary rule) + obstacle swap + relaxation of region cell c at global (gx, gy); in the last step
    // the result goes to global memory (the cells still computed then are exactly the tile)
    auto cell_step = [&](int c, int gx, int gy, auto wall, bool last, bool mine, Cell &q) {
        q.f0 = lds[0][c];
        q.f1 = lds[1][c - 1];
        q.f2 = lds[2][c - TILE_L];
        q.f3 = lds[3][c + 1];
        q.f4 = lds[4][c + TILE_L];
        q.f5 = lds[5][c - TILE_L - 1];
        q.f6 = lds[6][c - TILE_L + 1];
        q.f7 = lds[7][c + TILE_L + 1];
        q.f8 = lds[8][c + TILE_L - 1];
        float rho = 0.f, ux = 0.f, uy = 0.f;
#ifdef LB_DIAG
        if (!(a.diag & 1))                              // timing only: data movement and barriers alone (tools/ablate.py --tile)
#endif
        {
            if (decltype(wall)::value) {
                const bool w = (gx == 0), e = (gx == a.nx - 1), so = (gy == 0), no = (gy == a.ny - 1);
                boundary_rule<BC>(a, q, w, e, so, no);
            }
            finish_cell<BC, MASK>(a, gx, gy - a.y0, q, MASK && lmask[c] != 0, rho, ux, uy);
        }
        if (last && mine) {
            const long long o = (long long)gy * P + gx;
            float *d = a.dst + o;
            d[0] = q.f0; d[S] = q.f1; d[2 * S] = q.f2; d[3 * S] = q.f3; d[4 * S] = q.f4;
            d[5 * S] = q.f5; d[6 * S] = q.f6; d[7 * S] = q.f7; d[8 * S] = q.f8;
            if (MACRO) { const long long m = (long long)gy * a.fpitch + gx; a.rho[m] = rho; a.u[m] = ux; a.v[m] = uy; }
        }
    };
    auto cell_put = [&](int c, const Cell &q) {
        lds[0][c] = q.f0; lds[1][c] = q.f1; lds[2][c] = q.f2; lds[3][c] = q.f3; lds[4][c] = q.f4;
        lds[5][c] = q.f5; lds[6][c] = q.f6; lds[7][c] = q.f7; lds[8][c] = q.f8;
    };

    auto steps = [&](const bool store) {
#pragma unroll 1
        for (int s = 1; s <= TILE_T; ++s) {
#ifdef LB_DIAG
            if ((a.diag & 2) && s < TILE_T) continue;       // timing only: load, ONE step, store (what the three steps in LDS cost)
#endif
            const bool last = (s == TILE_T);
            Cell cs[TILE_CPT], wq;
            bool act[TILE_CPT], wact = false;
            // ---- main pass: every cell that is not on a wall ----------------------------------------------------
#pragma unroll
            for (int i = 0; i < TILE_CPT; ++i) {
                act[i] = ring[i] >= s;
                // (stepped by rings, only slot 0 -- the tile -- is ever stored: slot 1's coordinates are dead values)
                if (act[i]) cell_step(cc[i], gxs[i], gys[i], std::false_type(), last && store && (i == 0 || !RINGS), mine[i], cs[i]);
            }
            // ---- wall pass ------------------------------------------------------------------------------------------
            if (wall_tile) {
                wact = wring >= s;
                if (wact) cell_step(wc, gx0 + wlx, gy0 + wly, std::true_type(), last && store, true, wq);
            }
            if (last) break;
            __syncthreads();
            // ---- post-collision values back in place ------------------------------------------------------------
#pragma unroll
            for (int i = 0; i < TILE_CPT; ++i)
                if (act[i]) cell_put(cc[i], cs[i]);
            if (wact) cell_put(wc, wq);
            __syncthreads();
        }
    };
#ifdef LB_DIAG
    // timing only (tools/ablate.py --tile, bit 3): every other workgroup of an XCD runs its four steps FIRST (on whatever LDS
    // holds) and loads and stores afterwards -- the workgroups of a CU out of phase with each other: what would overlapping
    // one workgroup's loads with another's steps be worth?
    if (flip) {
        steps(false);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TILE_CPT; ++i) {
            const int c = tid + i * TILE_THREADS;
            if (c < TILE_CELLS) {
                const int sx = min(max(gxs[i], 0), a.nx - 1), sy = min(max(gys[i], 0), a.ny - 1);
                const long long o = (long long)sy * P + sx;
#pragma unroll
                for (int k = 0; k < 9; ++k) lds[k][c] = a.src[k * S + o];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TILE_CPT; ++i) {
            if (ring[i] >= TILE_T && mine[i]) {
                float *d = a.dst + (long long)gys[i] * P + gxs[i];
#pragma unroll
                for (int k = 0; k < 9; ++k) d[k * S] = lds[k][cc[i]];
            }
        }
        return;
    }
#endif
    steps(true);
}

// ---- the velocity-inlet family's wall-row bands, D time steps in one launch ---------------------------------------
// A three-, four- or five-step marching pass of that family covers rows [D, ny-D); the 2D rows next to each wall, stacked, are a
// velocity-inlet lattice of 4D rows of their own (lb_hip.cpp vel_band_pass: the north row's pull reaches "row ny-2" = band
// row 4D-2, the south row's "row 1" = band row 1; the seam in the middle spreads one row of garbage per step and reaches
// exactly the rows that are not kept).  Round 2 copied them into a second handle, ran D single steps there and copied
// the outer rows back: a dozen dependent launches.  Here one workgroup holds a column chunk of the band -- TW columns + D
// halo columns on either side, all 4D rows, nine planes -- in LDS, advances it D steps (after step s the outermost s
// columns are stale and no longer computed; rows wrap onto each other inside the band, so there is no halo in y) and
// stores the outer D + D rows.  Same cell functions as k_step on the band handle: same bits.
// band row j <-> grid row j (j < 2D) or ny - 4D + j (j >= 2D)
template <bool MASK, bool MACRO, int D>
__global__ __launch_bounds__(256) void k_vel_band(const StepArgs a)
{
    constexpr int HB = 4 * D, LW = 64, TW = LW - 2 * D, CELLS = LW * HB, CPT = CELLS / 256;
    static_assert(CELLS % 256 == 0, "region cells per thread");
    __shared__ float lds[9][CELLS];
    __shared__ unsigned char lmask[CELLS];
    const int tid = threadIdx.x;
    const int gx0 = blockIdx.x * TW - D;                        // grid column of region column 0
    const long long P = a.pitch, S = a.plane;
    auto grid_row = [&](int j) { return j < 2 * D ? j : a.ny - HB + j; };
    int cxs[CPT], cys[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + i * 256;
        const int cx = c % LW, cy = c / LW;
        cxs[i] = cx; cys[i] = cy;
        // (columns outside the box: copies of the nearest one inside; the rule overwrites every link pulled from them)
        const int sx = min(max(gx0 + cx, 0), a.nx - 1);
        const long long o = (long long)grid_row(cy) * P + sx;
#pragma unroll
        for (int k = 0; k < 9; ++k) lds[k][c] = a.src[k * S + o];
        lmask[c] = (MASK && sx == gx0 + cx) ? a.mask[(long long)grid_row(cy) * a.fpitch + sx] : 0;
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 1; s <= D; ++s) {
        const bool last = (s == D);
        Cell out[CPT];
        bool act[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * 256, cx = cxs[i], cy = cys[i], x = gx0 + cx;
            act[i] = cx >= s && cx < LW - s;
            if (!act[i]) continue;
            // rows the cy = +1 / cy = -1 links come from: the band is a velocity-inlet lattice of HB rows (wrap_y == 2)
            const int rm = (cy == 0 ? HB - 2 : cy - 1) * LW + cx, r0 = c, rp = (cy == HB - 1 ? 1 : cy + 1) * LW + cx;
            Cell q = {lds[0][r0], lds[1][r0 - 1], lds[2][rm], lds[3][r0 + 1], lds[4][rp], lds[5][rm - 1], lds[6][rm + 1],
                      lds[7][rp + 1], lds[8][rp - 1]};
            const bool in_box = x >= 0 && x < a.nx;
            const bool w = (x == 0), e = (x == a.nx - 1), so = (cy == 0), no = (cy == HB - 1);
            if (w || e) bc_vel_cell(q, w, e, so, no, a.u_w, a.u_e, a.corner);
            if (MASK) bounce_cell(q, lmask[c] != 0);
            float rho, ux, uy;
            moments_cell(q, rho, ux, uy);
            const long long m = (long long)grid_row(cy) * a.fpitch + (in_box ? x : 0);
            if (w || e) vel_moments_cell(q, w, so, no, a.u_w, a.u_e, a.u[m], a.v[m], rho, ux, uy);
            equilibrate_cell(q, a.omega, rho, ux, uy);
            out[i] = q;
            if (last && in_box && cx >= D && cx < LW - D && (cy < D || cy >= 3 * D)) {
                float *d = a.dst + (long long)grid_row(cy) * P + x;
                d[0] = q.f0; d[S] = q.f1; d[2 * S] = q.f2; d[3 * S] = q.f3; d[4 * S] = q.f4;
                d[5 * S] = q.f5; d[6 * S] = q.f6; d[7 * S] = q.f7; d[8 * S] = q.f8;
                if (MACRO) { a.rho[m] = rho; a.u[m] = ux; a.v[m] = uy; }
            }
        }
        if (last) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            if (act[i]) {
                const int c = tid + i * 256;
                lds[0][c] = out[i].f0; lds[1][c] = out[i].f1; lds[2][c] = out[i].f2; lds[3][c] = out[i].f3; lds[4][c] = out[i].f4;
                lds[5][c] = out[i].f5; lds[6][c] = out[i].f6; lds[7][c] = out[i].f7; lds[8][c] = out[i].f8;
            }
        __syncthreads();
    }
}

}  // namespace
