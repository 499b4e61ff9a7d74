// march5.cpp -- instantiates k_step5 (five time steps per pass on overlapping strips; kernels_step5.h).  See launchers.h.
#include "launchers.h"
#include "kernels_step4.h"
#include "kernels_step5.h"

namespace {

template <int BC, bool MASK, bool MACRO>
struct L5 {
    static void go(const MarchLaunch &g, const StepArgs &a)
    {
        hipLaunchKernelGGL((k_step5<BC, MASK, MACRO, false>), g.grid, g.block, 0, g.stream, a, g.strips, g.seg_rows, g.nsegs,
                           g.row_end);
    }
};

}  // namespace

void lbk_launch_march5(int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a)
{
    lbk_dispatch<L5, true>(bc, mask, macro, g, a);
}
