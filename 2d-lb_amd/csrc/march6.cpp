// march6.cpp -- instantiates k_step6 (six time steps per pass; kernels_step6.h).  See launchers.h.
#include "launchers.h"
#include "kernels_step4.h"
#include "kernels_step5.h"
#include "kernels_step6.h"

namespace {

template <int BC, bool MASK, bool MACRO>
struct L6 {
    static void go(const MarchLaunch &g, const StepArgs &a)
    {
        hipLaunchKernelGGL((k_step6<BC, MASK, MACRO>), g.grid, g.block, 0, g.stream, a, g.strips, g.seg_rows, g.nsegs, g.row_end);
    }
};

}  // namespace

void lbk_launch_march6(int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a)
{
    lbk_dispatch<L6, false>(bc, mask, macro, g, a);
}
