// Host-side interface of the mid-token kernels' translation unit (w4a16_xm.hip) for the planner / dispatcher in w4a16_gemm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "w4a16_args.hpp"

namespace quick_amd {

// mb: 32-token blocks per workgroup (1, 2), pr: 32-channel pairs per workgroup (1, 2, 3); abl: 0 or 32 (in-kernel span stamps); grid_x = N / (32 pr)
// channel blocks, grid_y token tiles of 32 mb.  false: no build for this configuration / group size (G a power-of-two multiple of 128).
bool xm_launch(int mb, int pr, int abl, const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop);
// dynamic LDS of one workgroup
unsigned xm_lds_need(int mb, int pr);

}  // namespace quick_amd
