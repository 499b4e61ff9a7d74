// launchers.h -- the seams between the library's translation units.
//
// Until round 5 liblbhip.so was ONE translation unit: 368 kernel instantiations compiled one after the other, six and a half
// minutes on eight cores.  Now every kernel family is instantiated in a file of its own (step1.cpp, march23.cpp, march4.cpp,
// march5.cpp, deep6.cpp, deep7.cpp, tile.cpp), which build.py compiles in parallel; lb_hip.cpp keeps the host side, the C ABI and the
// small un-fused kernels.  A family's file exports one plain function -- below -- that picks the instantiation (boundary family,
// obstacle mask, rho/u/v epilogue, ...) and launches it; arguments are the kernels' own (StepArgs, kernels_fused.h) plus the launch
// geometry.  Every kernel stays a template in a header: a translation unit only pays for what it launches.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_fused.h"

// geometry of a marching launch (k_step2 ... k_step5, k_deep; launch_step2 in lb_hip.cpp computes it)
struct MarchLaunch {
    dim3 grid, block;
    hipStream_t stream;
    int strips, seg_rows, nsegs, row_end;
};

// bc: the kernels' template value (LB_BC_PIPE, _PERIODIC, _CAVITY, _VELOCITY_INLET, LB_BC_PIPE_I)
void lbk_launch_step(int bc, bool mask, bool macro, int variant, dim3 grid, dim3 block, hipStream_t st, const StepArgs &a);       // step1.cpp
void lbk_launch_step_batch(bool mask, bool macro, dim3 grid, dim3 block, hipStream_t st, const BatchArgs &b);                     // step1.cpp
void lbk_launch_march23(int depth, int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a);                       // march23.cpp
void lbk_launch_march4(int bc, bool mask, bool macro, bool prefetch, const MarchLaunch &g, const StepArgs &a);                    // march4.cpp
void lbk_launch_march5(int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a);                                   // march5.cpp
bool lbk_launch_deep6(int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a);                                    // deep6.cpp (not VELOCITY_INLET)
bool lbk_launch_deep7(int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a);                                    // deep7.cpp (not VELOCITY_INLET)
bool lbk_launch_deep2_7(int bc, bool mask, bool macro, const MarchLaunch &g, const StepArgs &a);                                  // deep2.cpp: k_deep2, four waves per workgroup (not VELOCITY_INLET)
// k_tile4 over a whole grid of nx x h cells; shape 0: 32 x 16 tiles, two cells per thread; 1: 32 x 16, one; 2: 16 x 16, one
bool lbk_launch_tile4(int bc, bool mask, bool macro, int shape, int nx, int h, hipStream_t st, const StepArgs &a);                // tile.cpp (not VELOCITY_INLET)
void lbk_launch_vel_band(bool mask, bool macro, int d, dim3 grid, dim3 block, hipStream_t st, const StepArgs &a);                 // tile.cpp (d = 3, 4, 5)

// Run-time (bc, mask, macro) -> L<BC, MASK, MACRO>::go(args...).  VEL: the family list includes LB_BC_VELOCITY_INLET.  Returns false --
// and launches nothing -- for a family the unit does not instantiate: the caller reports it (a silent no-op would skip time steps).
template <template <int, bool, bool> class L, bool VEL, typename... A>
inline bool lbk_dispatch(int bc, bool mask, bool macro, const A &...args)
{
#define LBK_MM(BC)                                                            \
    do {                                                                      \
        if (mask) { if (macro) L<BC, true, true>::go(args...); else L<BC, true, false>::go(args...); }    \
        else      { if (macro) L<BC, false, true>::go(args...); else L<BC, false, false>::go(args...); }  \
    } while (0)
    switch (bc) {
    case LB_BC_PIPE_I: LBK_MM(LB_BC_PIPE_I); break;
    case LB_BC_PIPE: LBK_MM(LB_BC_PIPE); break;
    case LB_BC_PERIODIC: LBK_MM(LB_BC_PERIODIC); break;
    case LB_BC_VELOCITY_INLET:
        if constexpr (VEL) { LBK_MM(LB_BC_VELOCITY_INLET); }
        else return false;
        break;
    default: LBK_MM(LB_BC_CAVITY); break;
    }
#undef LBK_MM
    return true;
}
