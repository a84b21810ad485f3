// Host-side interface of the 128 x 256 four-wave kernels' translation unit (w4a16_xw.hip) for the planner / dispatcher in w4a16_gemm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "w4a16_args.hpp"

namespace quick_amd {

// slices: K slices per tile (1, 2, 4); abl: 0, 32 (in-kernel span stamps) or, tools builds, 64 (phase stamps) / 68 (no exchange).
// false: no build for this configuration / group size.
bool xw_launch(int slices, int abl, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop);

}  // namespace quick_amd
