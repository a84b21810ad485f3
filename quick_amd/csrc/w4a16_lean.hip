// Lean small-M kernels (w4a16_lean.hpp): the launcher.  The kernels themselves are instantiated in w4a16_lean_a/b/c.hip.
#include "w4a16_common.hpp"

#include <hip/hip_ext.h>

#include <algorithm>

#include "w4a16_args.hpp"
#include "w4a16_lean.hpp"
#include "w4a16_lean_host.hpp"

namespace quick_amd {

template <int WAVES, int TMAX, int NTW, int ABL>
bool lean_build(const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop);

#ifdef QA_EXP_LEAN_OVERLAP
thread_local LeanOverlapExp g_lean_overlap{};
}  // namespace quick_amd
// experiment entry point: the next lean launch of this thread waits for `wait_sig` to reach (its own run number + 1) * wait_per_exec before it asks for x,
// counts its own runs in `my_cnt`, and adds one arrival per storing wave to `signal` behind its rows.  Null pointers switch each part off.
extern "C" void quick_amd_exp_lean_overlap(const unsigned* wait_sig, unsigned wait_per_exec, unsigned* my_cnt, unsigned* signal, int any_order) {
  quick_amd::g_lean_overlap = quick_amd::LeanOverlapExp{wait_sig, my_cnt, wait_per_exec, signal, any_order};
}
namespace quick_amd {
#endif
unsigned lean_lds_need(int M, int K, int waves, int ntw, bool ln, bool persist) { return lean_lds_bytes(std::min(M, 16), K, waves, ntw, ln, persist); }

template <int ABL>
static bool lean_go_w(int waves, int tmax, int ntw, const GemmArgs& a, int grid_x, int grid_y, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  switch (waves * 1000 + tmax * 10 + ntw) {
#define QA_LEAN_CASE(W, T, C) \
  case W * 1000 + T * 10 + C: return lean_build<W, T, C, ABL>(a, grid_x, grid_y, st, start, stop)
    QA_LEAN_CASE(8, 4, 1);
    QA_LEAN_CASE(8, 4, 2);
    QA_LEAN_CASE(8, 8, 1);
    QA_LEAN_CASE(8, 12, 1);
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
