// tile.cpp -- instantiates k_tile4 (four time steps inside LDS tiles) and k_vel_band (the wall-side rows of the velocity-inlet
// family as an LDS band); kernels_tile.h.  See launchers.h.
#include "launchers.h"
#include "kernels_tile.h"

namespace {

template <int BC, bool MASK, bool MACRO, int TW, int TH, int CPT>
void tile_shape(int nx, int h, hipStream_t st, const StepArgs &a)
{
    const int tiles_x = (nx + TW - 1) / TW, tiles_y = (h + TH - 1) / TH, n_tiles = tiles_x * tiles_y;
    const dim3 grid((n_tiles + 7) / 8 * 8), block(TileShape<TW, TH, CPT>::THREADS);     // (eight equal shares: xcd_band_tile)
    hipLaunchKernelGGL((k_tile4<BC, MASK, MACRO, TW, TH, CPT>), grid, block, 0, st, a, tiles_x, n_tiles);
}

template <int BC, bool MASK, bool MACRO>
struct LT {
    static void go(int shape, int nx, int h, hipStream_t st, const StepArgs &a)
    {
        if (shape == 0) tile_shape<BC, MASK, MACRO, 32, 16, 2>(nx, h, st, a);
        else if (shape == 1) tile_shape<BC, MASK, MACRO, 32, 16, 1>(nx, h, st, a);
        else tile_shape<BC, MASK, MACRO, 16, 16, 1>(nx, h, st, a);
    }
};

template <bool MASK, bool MACRO>
void vel_band(int d, dim3 grid, dim3 block, hipStream_t st, const StepArgs &a)
{
    if (d == 5) hipLaunchKernelGGL((k_vel_band<MASK, MACRO, 5>), grid, block, 0, st, a);
    else if (d == 4) hipLaunchKernelGGL((k_vel_band<MASK, MACRO, 4>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_vel_band<MASK, MACRO, 3>), grid, block, 0, st, a);
}

}  // namespace

bool lbk_launch_tile4(int bc, bool mask, bool macro, int shape, int nx, int h, hipStream_t st, const StepArgs &a)
{
    return lbk_dispatch<LT, false>(bc, mask, macro, shape, nx, h, st, a);
}

void lbk_launch_vel_band(bool mask, bool macro, int d, dim3 grid, dim3 block, hipStream_t st, const StepArgs &a)
{
    if (mask) { if (macro) vel_band<true, true>(d, grid, block, st, a); else vel_band<true, false>(d, grid, block, st, a); }
    else      { if (macro) vel_band<false, true>(d, grid, block, st, a); else vel_band<false, false>(d, grid, block, st, a); }
}
