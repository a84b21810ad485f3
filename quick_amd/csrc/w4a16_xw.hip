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

template <int MB, int PAIRS, int S, int ABL>
static bool xw_go(const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  constexpr unsigned lds = 128 * 1024;
  auto kfn = w4a16_xw_kernel<MB, PAIRS, S, ABL>;
  static std::atomic<unsigned long long> attr_set{0};
  (void)lds_limit_once(attr_set, (const void*)kfn, (int)lds);
  XwRest rest{a.bias, a.residual, a.Y, a.slabs, a.counters, a.dbg, a.span, a.silu_mul, a.G};
  hipExtLaunchKernelGGL(kfn, dim3(workgroups), dim3(256), lds, st, start, stop, 0, a.X, a.QW, a.S, a.M, a.K, a.N, a.tpg, a.ksplit, a.kt_per_split, a.xcd_gm, rest);
  return true;
}

template <int MB, int PAIRS, int ABL>
static bool xw_go_s(int s, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  switch (s) {
    case 1: return xw_go<MB, PAIRS, 1, ABL>(a, workgroups, st, start, stop);
    case 2: return xw_go<MB, PAIRS, 2, ABL>(a, workgroups, st, start, stop);
    case 4:
      if constexpr (MB == 4) return xw_go<MB, PAIRS, 4, ABL>(a, workgroups, st, start, stop);
      return false;
    default: return false;
  }
}

template <int ABL>
static bool xw_go_t(int mb, int pairs, int s, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  if (mb == 8 && pairs == 2) return s == 1 ? xw_go<8, 2, 1, ABL>(a, workgroups, st, start, stop) : false;
  if (mb == 4 && pairs == 2) return xw_go_s<4, 2, ABL>(s, a, workgroups, st, start, stop);
  if (mb == 4 && pairs == 1) return xw_go_s<4, 1, ABL>(s, a, workgroups, st, start, stop);
  if (mb == 2 && pairs == 1) return xw_go_s<2, 1, ABL>(s, a, workgroups, st, start, stop);
  return false;
}

bool xw_launch(int mb, int pairs, int slices, int abl, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  if (a.G % 128 != 0 || (a.tpg & (a.tpg - 1)) != 0) return false;   // the loop shifts the k tile by log2(k tiles per group)
  switch (abl) {
    case 0: return xw_go_t<0>(mb, pairs, slices, a, workgroups, st, start, stop);
    case 32: return xw_go_t<32>(mb, pairs, slices, a, workgroups, st, start, stop);
#ifdef QUICK_AMD_TOOLS
    case 64: return xw_go_t<64>(mb, pairs, slices, a, workgroups, st, start, stop);
    case 68: return xw_go_t<68>(mb, pairs, slices, a, workgroups, st, start, stop);
    // loop experiments (stamps + wrong results): 64 + 256 * {1 no barrier, 2 no vector memory, 4 no dequantisation, 8 no B reads, 15 all}
    case 320: return mb == 2 ? xw_go<2, 1, 1, 320>(a, workgroups, st, start, stop) : (pairs == 1 && slices == 2 ? xw_go<4, 1, 2, 320>(a, workgroups, st, start, stop) : false);
    case 576: return mb == 2 ? xw_go<2, 1, 1, 576>(a, workgroups, st, start, stop) : (pairs == 1 && slices == 2 ? xw_go<4, 1, 2, 576>(a, workgroups, st, start, stop) : false);
    case 1088: return mb == 2 ? xw_go<2, 1, 1, 1088>(a, workgroups, st, start, stop) : (pairs == 1 && slices == 2 ? xw_go<4, 1, 2, 1088>(a, workgroups, st, start, stop) : false);
    case 2112: return mb == 2 ? xw_go<2, 1, 1, 2112>(a, workgroups, st, start, stop) : (pairs == 1 && slices == 2 ? xw_go<4, 1, 2, 2112>(a, workgroups, st, start, stop) : false);
    case 3904: return mb == 2 ? xw_go<2, 1, 1, 3904>(a, workgroups, st, start, stop) : false;
#endif
    default: return false;
  }
}

}  // namespace quick_amd
