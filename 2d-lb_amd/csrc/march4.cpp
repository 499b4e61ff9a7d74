// march4.cpp -- instantiates k_step4 (four time steps per pass; kernels_step4.h).  See launchers.h.
#include "launchers.h"
#include "kernels_step4.h"

namespace {

template <int BC, bool MASK, bool MACRO>
struct L4 {
    static void go(bool prefetch, const MarchLaunch &g, const StepArgs &a)
    {
        // the row-ahead gather where it fits in 256 registers without scratch (step4_prefetch), unless the caller switches it off
        if (step4_prefetch(BC, MASK, MACRO) && prefetch)
            hipLaunchKernelGGL((k_step4<BC, MASK, MACRO, false, step4_prefetch(BC, MASK, MACRO)>), g.grid, g.block, 0, g.stream, a,
                               g.strips, g.seg_rows, g.nsegs, g.row_end);
        else
            hipLaunchKernelGGL((k_step4<BC, MASK, MACRO, false, false>), g.grid, g.block, 0, g.stream, a, g.strips, g.seg_rows,
                               g.nsegs, g.row_end);
    }
};

}  // namespace

void lbk_launch_march4(int bc, bool mask, bool macro, bool prefetch, const MarchLaunch &g, const StepArgs &a)
{
    lbk_dispatch<L4, true>(bc, mask, macro, prefetch, g, a);
}
