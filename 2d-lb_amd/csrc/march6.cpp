// march6.cpp -- instantiates k_step6 (six time steps per pass; kernels_step6.h).  See launchers.h.
#include <stdlib.h>
#include "launchers.h"
#include "kernels_step4.h"
#include "kernels_step5.h"
#include "kernels_step6.h"
#include "kernels_deep.h"

namespace {

template <int BC, bool MASK, bool MACRO>
struct L6 {
    static void go(const MarchLaunch &g, const StepArgs &a)
    {
        // PROBE (round 5): rows gathered ahead at one wave per SIMD -- periodic, no mask, no rho/u/v epilogue only
        static const int pfd = getenv("LB_STEP6_PFD") ? atoi(getenv("LB_STEP6_PFD")) : 0;
        static const int deep = getenv("LB_DEEP") ? atoi(getenv("LB_DEEP")) : 0;       // D * 100 + RW * 10 + PFD
        if (BC == LB_BC_PERIODIC && !MASK && !MACRO && deep) {
#define LB_DEEP_CASE(D, RW, PFD)                                                                                              \
    case D * 100 + RW * 10 + PFD:                                                                                             \
        hipLaunchKernelGGL((k_deep<LB_BC_PERIODIC, false, false, D, RW, PFD>), g.grid, g.block, 0, g.stream, a, g.strips, g.seg_rows, \
                           g.nsegs, g.row_end);                                                                               \
        return;
            switch (deep) {
                LB_DEEP_CASE(6, 2, 0) LB_DEEP_CASE(6, 2, 1) LB_DEEP_CASE(6, 1, 1) LB_DEEP_CASE(6, 1, 0) LB_DEEP_CASE(7, 2, 1) LB_DEEP_CASE(7, 1, 1)
                LB_DEEP_CASE(5, 1, 1) LB_DEEP_CASE(5, 2, 0)
            }
#undef LB_DEEP_CASE
        }
        if (BC == LB_BC_PERIODIC && !MASK && !MACRO && pfd == 1)
            hipLaunchKernelGGL((k_step6<LB_BC_PERIODIC, false, false, 1>), g.grid, g.block, 0, g.stream, a, g.strips, g.seg_rows, g.nsegs, g.row_end);
        else if (BC == LB_BC_PERIODIC && !MASK && !MACRO && pfd == 2)
            hipLaunchKernelGGL((k_step6<LB_BC_PERIODIC, false, false, 2>), g.grid, g.block, 0, g.stream, a, g.strips, g.seg_rows, g.nsegs, g.row_end);
        else
        hipLaunchKernelGGL((k_step6<BC, MASK, MACRO>), g.grid, g.block, 0, g.stream, a, g.strips, g.seg_rows, g.nsegs, g.row_end);
    }
};

}  // namespace

void lbk_launch_march6(int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a)
{
    lbk_dispatch<L6, false>(bc, mask, macro, g, a);
}
