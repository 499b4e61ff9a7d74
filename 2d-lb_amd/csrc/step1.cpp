// step1.cpp -- instantiates k_step (one time step per launch; kernels_fused.h) and k_step_batch.  See launchers.h.
#include "launchers.h"

namespace {

template <int BC, bool MASK, bool MACRO>
struct L1 {
    static void go(int variant, dim3 grid, dim3 block, hipStream_t st, const StepArgs &a)
    {
#define LB_LAUNCH(NTL, NTS, XCD) hipLaunchKernelGGL((k_step<BC, MASK, MACRO, NTL, NTS, XCD>), grid, block, 0, st, a)
        switch (variant & 19) {           // bit 0: NT stores, bit 1: NT loads, bit 4: XCD-aware tile order
        case 0: LB_LAUNCH(false, false, false); break;
        case 1: LB_LAUNCH(false, true, false); break;
        case 2: LB_LAUNCH(true, false, false); break;
        case 3: LB_LAUNCH(true, true, false); break;
        case 16: LB_LAUNCH(false, false, true); break;
        case 17: LB_LAUNCH(false, true, true); break;
        case 18: LB_LAUNCH(true, false, true); break;
        default: LB_LAUNCH(true, true, true); break;
        }
#undef LB_LAUNCH
    }
};

}  // namespace

void lbk_launch_step(int bc, bool mask, bool macro, int variant, dim3 grid, dim3 block, hipStream_t st, const StepArgs &a)
{
    lbk_dispatch<L1, true>(bc, mask, macro, variant, grid, block, st, a);
}

void lbk_launch_step_batch(bool mask, bool macro, dim3 grid, dim3 block, hipStream_t st, const BatchArgs &b)
{
    if (mask) {
        if (macro) hipLaunchKernelGGL((k_step_batch<LB_BC_PERIODIC, true, true>), grid, block, 0, st, b);
        else hipLaunchKernelGGL((k_step_batch<LB_BC_PERIODIC, true, false>), grid, block, 0, st, b);
    } else {
        if (macro) hipLaunchKernelGGL((k_step_batch<LB_BC_PERIODIC, false, true>), grid, block, 0, st, b);
        else hipLaunchKernelGGL((k_step_batch<LB_BC_PERIODIC, false, false>), grid, block, 0, st, b);
    }
}
