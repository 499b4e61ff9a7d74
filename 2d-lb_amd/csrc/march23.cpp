// march23.cpp -- instantiates k_step2 and k_step3 (two / three time steps per pass; kernels_fused.h).  See launchers.h.
#include "launchers.h"

namespace {

template <int BC, bool MASK, bool MACRO>
struct L23 {
    static void go(int depth, const MarchLaunch &g, const StepArgs &a)
    {
        if (depth == 3)
            hipLaunchKernelGGL((k_step3<BC, MASK, MACRO, false>), g.grid, g.block, 0, g.stream, a, g.strips, g.seg_rows, g.nsegs,
                               g.row_end);
        else
            hipLaunchKernelGGL((k_step2<BC, MASK, MACRO, false>), g.grid, g.block, 0, g.stream, a, g.strips, g.seg_rows, g.nsegs,
                               g.row_end);
    }
};

}  // namespace

void lbk_launch_march23(int depth, int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a)
{
    lbk_dispatch<L23, true>(bc, mask, macro, depth, g, a);
}
