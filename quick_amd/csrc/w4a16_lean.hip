// Lean small-M kernels (w4a16_lean.hpp): instantiations and launcher.  Its own translation unit.
#include "w4a16_common.hpp"

#include <hip/hip_ext.h>

#include <algorithm>

#include "w4a16_args.hpp"
#include "w4a16_lean.hpp"
#include "w4a16_lean_host.hpp"

namespace quick_amd {

unsigned lean_lds_need(int M, int K, int waves, int ntw, bool ln) { return lean_lds_bytes(std::min(M, 16), K, waves, ntw, ln); }

template <int WAVES, int TMAX, int NTW, int GM, int ABL>
static bool lean_go(const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  auto kfn = w4a16_lean_kernel<WAVES, TMAX, NTW, GM, ABL>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const unsigned lds = lean_lds_need(a.M, a.K, WAVES, NTW, a.ln_w != nullptr);
  LeanRest rest{};
  rest.Y = a.Y; rest.bias = a.bias; rest.residual = a.residual; rest.silu_mul = a.silu_mul; rest.ln_eps = a.ln_eps; rest.span = a.span; rest.dbg = a.dbg;
  hipExtLaunchKernelGGL(kfn, dim3(grid_x, grid_y), dim3(WAVES * 64), lds, st, start, stop, 0, a.X, a.QW, a.S, a.ln_w, a.K, a.N, a.M, (unsigned)(a.K / a.G), grid_x,
                        (unsigned)a.tpg, rest);
  return true;
}

template <int WAVES, int TMAX, int NTW, int ABL>
static bool lean_go_g(const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  if (a.G == 128) return lean_go<WAVES, TMAX, NTW, 0, ABL>(a, grid_x, grid_y, st, start, stop);
  return lean_go<WAVES, TMAX, NTW, 1, ABL>(a, grid_x, grid_y, st, start, stop);
}

template <int ABL>
static bool lean_go_w(int waves, int tmax, int ntw, const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  switch (waves * 1000 + tmax * 10 + ntw) {
#define QA_LEAN_CASE(W, T, C) \
  case W * 1000 + T * 10 + C: return lean_go_g<W, T, C, ABL>(a, grid_x, grid_y, st, start, stop)
    QA_LEAN_CASE(4, 8, 1);
    QA_LEAN_CASE(4, 16, 1);
    QA_LEAN_CASE(8, 4, 1);
    QA_LEAN_CASE(8, 4, 2);
    QA_LEAN_CASE(8, 8, 1);
    QA_LEAN_CASE(8, 8, 2);
    QA_LEAN_CASE(8, 12, 1);
    QA_LEAN_CASE(16, 2, 1);
    QA_LEAN_CASE(16, 2, 2);
    QA_LEAN_CASE(16, 4, 1);
    QA_LEAN_CASE(16, 4, 2);
    QA_LEAN_CASE(16, 8, 1);
#undef QA_LEAN_CASE
    default: return false;
  }
}

bool lean_launch(int waves, int tmax, int ntw, int abl, const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  if (a.G % 128 != 0 || (a.K >> 7) < waves || ((a.K >> 7) + waves - 1) / waves > tmax || (a.N >> 4) % ntw != 0) return false;
  switch (abl) {
    case 0: return lean_go_w<0>(waves, tmax, ntw, a, grid_x, grid_y, st, start, stop);
    case 32: return lean_go_w<32>(waves, tmax, ntw, a, grid_x, grid_y, st, start, stop);
#ifdef QUICK_AMD_TOOLS
    case 64: return lean_go_w<64>(waves, tmax, ntw, a, grid_x, grid_y, st, start, stop);
#endif
    default: return false;
  }
}

}  // namespace quick_amd
