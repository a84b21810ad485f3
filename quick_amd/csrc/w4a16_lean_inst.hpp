// Lean small-M kernels: the per-build launcher template (group mode, RMSNorm prologue and token-count flavours of one (waves, tiles per wave,
// channel tiles) build).  Explicitly instantiated in w4a16_lean_a/b/c.hip so that the builds compile in parallel; w4a16_lean.hip dispatches.
#pragma once
#include <hip/hip_ext.h>

#include <algorithm>

#include "w4a16_args.hpp"
#include "w4a16_lean.hpp"
#include "w4a16_lean_host.hpp"

namespace quick_amd {

template <int WAVES, int TMAX, int NTW, int GM, int ABL, bool LN, int MR, int NSETS = 1>
static bool lean_go(const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  auto kfn = w4a16_lean_kernel<WAVES, TMAX, NTW, GM, ABL, LN, MR, NSETS>;
  static std::atomic<unsigned long long> attr_set{0};
  (void)lds_limit_once(attr_set, (const void*)kfn, 160 * 1024);
  const unsigned lds = lean_lds_need(a.M, a.K, WAVES, NTW, a.ln_w != nullptr, NSETS > 1);
  LeanRest rest{};
  rest.Y = a.Y; rest.bias = a.bias; rest.residual = a.residual; rest.silu_mul = a.silu_mul; rest.ln_eps = a.ln_eps; rest.span = a.span; rest.dbg = a.dbg;
  unsigned launch_flags = 0;
#ifdef QA_EXP_LEAN_OVERLAP
  rest.wait_sig = g_lean_overlap.wait_sig; rest.my_cnt = g_lean_overlap.my_cnt; rest.wait_per_exec = g_lean_overlap.wait_per_exec; rest.signal = g_lean_overlap.signal;
  rest.my_per_exec = (unsigned)(grid_x * grid_y);
  if (g_lean_overlap.any_order) launch_flags = hipExtAnyOrderLaunch;   // no barrier in front of this launch's packet: it is dispatched BEHIND its predecessor of the same stream and runs beside it
  g_lean_overlap = LeanOverlapExp{};
#endif
  hipExtLaunchKernelGGL(kfn, dim3(grid_x, grid_y), dim3(WAVES * 64), lds, st, start, stop, launch_flags, a.X, a.QW, a.S, a.ln_w, a.K, a.N, a.M, (unsigned)(a.K / a.G), grid_x,
                        (unsigned)a.tpg, rest);
  return true;
}

// token-count flavour: 1 token / up to 4 tokens with the unit sums as scalars (while TMAX * 4 * 2 of them fit the SGPRs), else the matrix-core sums
template <int WAVES, int TMAX, int NTW, int GM, int ABL, bool LN>
static bool lean_go_m(const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  // fewer workgroups than channel blocks: persistent (two register sets, the next block's requests in flight under this block's tiles);
  // builds with up to four tiles per wave, G = 128, no phase stamps
  if (grid_x < (a.N >> 4) / NTW) {
    // eight waves (sixteen would need more registers than a 1024-thread workgroup may have), up to four tiles per wave, G = 128, no stamps;
    // two register sets (four were measured slower, 16 x 4096 x 22016 14.4 -> 14.8 us: with one workgroup per CU the block time is the waves'
    // own dependency chains, not the depth of the requests in flight)
    if constexpr (WAVES == 8 && TMAX <= 4 && GM == 0 && ABL != 64) {
      constexpr int NS = 2;
      if (a.M == 1) return lean_go<WAVES, TMAX, NTW, GM, ABL, LN, 1, NS>(a, grid_x, grid_y, st, start, stop);
      if (a.M <= 4) return lean_go<WAVES, TMAX, NTW, GM, ABL, LN, 4, NS>(a, grid_x, grid_y, st, start, stop);
      return lean_go<WAVES, TMAX, NTW, GM, ABL, LN, 16, NS>(a, grid_x, grid_y, st, start, stop);
    }
    return false;
  }
  if (a.M == 1) return lean_go<WAVES, TMAX, NTW, GM, ABL, LN, 1>(a, grid_x, grid_y, st, start, stop);
  if constexpr (TMAX <= 8) {
    if (a.M <= 4) return lean_go<WAVES, TMAX, NTW, GM, ABL, LN, 4>(a, grid_x, grid_y, st, start, stop);
  }
  return lean_go<WAVES, TMAX, NTW, GM, ABL, LN, 16>(a, grid_x, grid_y, st, start, stop);
}

template <int WAVES, int TMAX, int NTW, int ABL>
bool lean_build(const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  if constexpr (ABL == 0) {
    if (a.ln_w != nullptr) {
      if (a.G == 128) return lean_go_m<WAVES, TMAX, NTW, 0, ABL, true>(a, grid_x, grid_y, st, start, stop);
      return lean_go_m<WAVES, TMAX, NTW, 1, ABL, true>(a, grid_x, grid_y, st, start, stop);
    }
    if (a.G == 128) return lean_go_m<WAVES, TMAX, NTW, 0, ABL, false>(a, grid_x, grid_y, st, start, stop);
    return lean_go_m<WAVES, TMAX, NTW, 1, ABL, false>(a, grid_x, grid_y, st, start, stop);
  } else {  // span / phase stamps: G = 128, the stamps of the RMSNorm prologue on the 8 x 4 builds only
    if (a.G != 128) return false;
    if (a.ln_w != nullptr) {
      if constexpr (ABL == 64 && WAVES == 8 && TMAX == 4) return lean_go_m<WAVES, TMAX, NTW, 0, ABL, true>(a, grid_x, grid_y, st, start, stop);
      return false;
    }
    return lean_go_m<WAVES, TMAX, NTW, 0, ABL, false>(a, grid_x, grid_y, st, start, stop);
  }
}

#ifdef QUICK_AMD_TOOLS
#define QA_LEAN_INSTANTIATE(W, T, C)                                                                                    \
  template bool lean_build<W, T, C, 0>(const GemmArgs&, int, int, hipStream_t, hipEvent_t, hipEvent_t);                 \
  template bool lean_build<W, T, C, 32>(const GemmArgs&, int, int, hipStream_t, hipEvent_t, hipEvent_t);                \
  template bool lean_build<W, T, C, 64>(const GemmArgs&, int, int, hipStream_t, hipEvent_t, hipEvent_t);
#else
#define QA_LEAN_INSTANTIATE(W, T, C)                                                                                    \
  template bool lean_build<W, T, C, 0>(const GemmArgs&, int, int, hipStream_t, hipEvent_t, hipEvent_t);                 \
  template bool lean_build<W, T, C, 32>(const GemmArgs&, int, int, hipStream_t, hipEvent_t, hipEvent_t);
#endif

}  // namespace quick_amd
