// Host-side interface of the exchange-K kernels' translation unit (w4a16_xk.hip) for the planner / dispatcher in w4a16_gemm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "w4a16_args.hpp"

namespace quick_amd {

struct XkConfig {
  int mb;    // token tiles of 32 per workgroup: 2 or 4
  int s;     // K slices per tile (workgroups that exchange): 1, 2, 4, 8
  int nbuf;  // x ring slots
  int wd;    // weight queue depth (stages)
  int abl;   // tools builds only: timing experiments / phase stamps
  int kq = 2;      // K groups of waves inside the workgroup: 2 (eight waves) or 4 (sixteen waves: 64-token tiles, one slice, K % 256 == 0)
  int loader = 0;  // 1: the twelve-wave flavour (four loader waves issue every vector-memory instruction; nbuf / wd do not apply)
};
constexpr size_t kXkZoneBytesHost = (size_t)16 << 20;  // == kXkZoneBytes (w4a16_xk.hpp)

// the (nbuf, wd) the library ships for a tile size
XkConfig xk_default_config(int mb, int s);
// false: no instantiation for this configuration / group mode
bool xk_launch(const XkConfig& c, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop);

}  // namespace quick_amd
