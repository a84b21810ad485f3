// Mid-token kernels (w4a16_xm.hpp): instantiations and launcher.  Its own translation unit.
#include "w4a16_common.hpp"

#include <hip/hip_ext.h>

#include "w4a16_args.hpp"
#include "w4a16_xm.hpp"
#include "w4a16_xm_host.hpp"

namespace quick_amd {

unsigned xm_lds_need(int mb, int pr) { return xm_lds_bytes(mb, pr); }

template <int MB, int PR, int ABL>
static bool xm_go(const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  auto kfn = w4a16_xm_kernel<MB, PR, ABL>;
  static std::atomic<unsigned long long> attr_set{0};
  (void)lds_limit_once(attr_set, (const void*)kfn, 160 * 1024);
  XmRest rest{a.bias, a.residual, a.Y, a.span, a.dbg, a.silu_mul};
  const int tpg_log2 = 31 - __builtin_clz((unsigned)a.tpg);
  hipExtLaunchKernelGGL(kfn, dim3(grid_x, grid_y), dim3(512), xm_lds_bytes(MB, PR), st, start, stop, 0, a.X, a.QW, a.S, a.M, a.K, a.N, tpg_log2, grid_x, rest);
  return true;
}

template <int ABL>
static bool xm_go_t(int mb, int pr, const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  switch (mb * 10 + pr) {
    case 11: return xm_go<1, 1, ABL>(a, grid_x, grid_y, st, start, stop);
    case 12: return xm_go<1, 2, ABL>(a, grid_x, grid_y, st, start, stop);
    case 13: return xm_go<1, 3, ABL>(a, grid_x, grid_y, st, start, stop);
    case 21: return xm_go<2, 1, ABL>(a, grid_x, grid_y, st, start, stop);
    case 22: return xm_go<2, 2, ABL>(a, grid_x, grid_y, st, start, stop);
    case 23: return xm_go<2, 3, ABL>(a, grid_x, grid_y, st, start, stop);
    default: return false;
  }
}

bool xm_launch(int mb, int pr, int abl, const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  if (a.G % 128 != 0 || (a.tpg & (a.tpg - 1)) != 0 || a.ln_w != nullptr) return false;
  switch (abl) {
    case 0: return xm_go_t<0>(mb, pr, a, grid_x, grid_y, st, start, stop);
    case 32: return xm_go_t<32>(mb, pr, a, grid_x, grid_y, st, start, stop);
#ifdef QUICK_AMD_TOOLS
    case 64: return xm_go_t<64>(mb, pr, a, grid_x, grid_y, st, start, stop);
#endif
    default: return false;
  }
}

}  // namespace quick_amd
