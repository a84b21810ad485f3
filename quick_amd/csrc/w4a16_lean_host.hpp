// Host-side interface of the lean small-M kernels' translation unit (w4a16_lean.hip) for the planner / dispatcher in w4a16_gemm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "w4a16_args.hpp"

namespace quick_amd {

// waves per workgroup (4, 8, 16), tmax = most k tiles a wave may own (all in flight at once: 2 .. 16), ntw = 16-channel tiles per workgroup
// (1, 2), abl: 0, 32 (in-kernel span stamps) or, tools builds, 64 (phase stamps into a.dbg); grid_x = N / 16 / ntw workgroups along the
// channels, grid_y token blocks of 16.  false: no build for this configuration / group size.
bool lean_launch(int waves, int tmax, int ntw, int abl, const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop);
#ifdef QA_EXP_LEAN_OVERLAP
// experiment builds: what the NEXT lean launch of this thread is told about its neighbours (consumed and cleared by that launch)
struct LeanOverlapExp {
  const unsigned* wait_sig;
  unsigned* my_cnt;
  unsigned wait_per_exec;
  unsigned* signal;
  int any_order;
};
extern thread_local LeanOverlapExp g_lean_overlap;
#endif
// dynamic LDS of one workgroup
unsigned lean_lds_need(int M, int K, int waves, int ntw, bool ln, bool persist = false);

}  // namespace quick_amd
