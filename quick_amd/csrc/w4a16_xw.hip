// 128 x 256 four-wave kernels with the hand-placed K loop: instantiations and launcher (see w4a16_xw.hpp).  Its own translation unit.
#include "w4a16_common.hpp"

#include <hip/hip_ext.h>

#include "w4a16_args.hpp"
#include "w4a16_wide.hpp"
#include "w4a16_xk.hpp"
#include "w4a16_xw.hpp"
#include "w4a16_xw_host.hpp"

namespace quick_amd {

static_assert(kXwZoneBytes == kXkZoneBytes, "exchange zone size");

template <int S, int ABL>
static bool xw_go(const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  constexpr unsigned lds = 128 * 1024;
  auto kfn = w4a16_xw_kernel<S, ABL>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipExtLaunchKernelGGL(kfn, dim3(workgroups), dim3(256), lds, st, start, stop, 0, a);
  return true;
}

template <int ABL>
static bool xw_go_s(int s, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  switch (s) {
    case 1: return xw_go<1, ABL>(a, workgroups, st, start, stop);
    case 2: return xw_go<2, ABL>(a, workgroups, st, start, stop);
    case 4: return xw_go<4, ABL>(a, workgroups, st, start, stop);
    default: return false;
  }
}

bool xw_launch(int slices, int abl, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  if (a.G % 128 != 0 || (a.tpg & (a.tpg - 1)) != 0) return false;   // the loop shifts the k tile by log2(k tiles per group)
  switch (abl) {
    case 0: return xw_go_s<0>(slices, a, workgroups, st, start, stop);
    case 32: return xw_go_s<32>(slices, a, workgroups, st, start, stop);
#ifdef QUICK_AMD_TOOLS
    case 64: return xw_go_s<64>(slices, a, workgroups, st, start, stop);
    case 68: return xw_go_s<68>(slices, a, workgroups, st, start, stop);
#endif
    default: return false;
  }
}

}  // namespace quick_amd
