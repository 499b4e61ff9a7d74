// deep2.cpp -- instantiates k_deep2: seven time steps per pass, two waves per strip and direction (kernels_deep2.h).  See launchers.h.
#include "launchers.h"
#include "kernels_step4.h"
#include "kernels_step5.h"
#include "kernels_deep.h"
#include "kernels_deep2.h"

namespace {

template <int BC, bool MASK, bool MACRO>
struct LD2 {
    static void go(const MarchLaunch &g, const StepArgs &a)
    {
        hipLaunchKernelGGL((k_deep2<BC, MASK, MACRO, 7>), g.grid, dim3(64, DEEP2_WAVES), 0, g.stream, a, g.strips, g.seg_rows, g.nsegs, g.row_end);
    }
};

}  // namespace

bool lbk_launch_deep2_7(int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a)
{
    return lbk_dispatch<LD2, false>(bc, mask, macro, g, a);
}
