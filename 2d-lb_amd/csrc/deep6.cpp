// deep6.cpp -- instantiates k_deep for 6 time steps per pass (kernels_deep.h).  See launchers.h.
#include "launchers.h"
#include "kernels_step4.h"
#include "kernels_step5.h"
#include "kernels_deep.h"

namespace {

template <int BC, bool MASK, bool MACRO>
struct LD {
    static void go(const MarchLaunch &g, const StepArgs &a)
    {
        // (LB_DEEP6_DEPTH: timing probes only -- another depth behind the six-step launcher; the host still counts six steps)
#ifndef LB_DEEP6_DEPTH
#define LB_DEEP6_DEPTH 6
#endif
        hipLaunchKernelGGL((k_deep<BC, MASK, MACRO, LB_DEEP6_DEPTH, deep_rw(LB_DEEP6_DEPTH), deep_pfd(LB_DEEP6_DEPTH)>), g.grid, g.block, 0,
                           g.stream, a, g.strips, g.seg_rows, g.nsegs, g.row_end);
    }
};

}  // namespace

bool lbk_launch_deep6(int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a)
{
    return lbk_dispatch<LD, false>(bc, mask, macro, g, a);
}
