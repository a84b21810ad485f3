// Host-side interface of the 128 x 256 four-wave kernels' translation unit (w4a16_xw.hip) for the planner / dispatcher in w4a16_gemm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "w4a16_args.hpp"

namespace quick_amd {

// mb x pairs: 32-token blocks x 32-channel pairs per wave, (4, 2) = 128 x 256 tiles, (4, 1) = 128 x 128, (2, 1) = 64 x 128; slices: K slices
// per tile (1, 2, 4; mb % slices == 0); abl: 0, 32 (in-kernel span stamps) or, tools builds, 64 (phase stamps) / 68 (no exchange).
// false: no build for this configuration / group size.
bool xw_launch(int mb, int pairs, int slices, int abl, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop);

}  // namespace quick_amd
