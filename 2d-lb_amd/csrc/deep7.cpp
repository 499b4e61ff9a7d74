// deep7.cpp -- instantiates k_deep for 7 time steps per pass (kernels_deep.h).  See launchers.h.
#include "launchers.h"
#include "kernels_step4.h"
#include "kernels_step5.h"
#include "kernels_deep.h"

namespace {

template <int BC, bool MASK, bool MACRO>
struct LD {
    static void go(const MarchLaunch &g, const StepArgs &a)
    {
        hipLaunchKernelGGL((k_deep<BC, MASK, MACRO, 7, deep_rw(7), deep_pfd(7)>), g.grid, g.block, 0, g.stream, a, g.strips, g.seg_rows,
                           g.nsegs, g.row_end);
    }
};

}  // namespace

bool lbk_launch_deep7(int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a)
{
    return lbk_dispatch<LD, false>(bc, mask, macro, g, a);
}
