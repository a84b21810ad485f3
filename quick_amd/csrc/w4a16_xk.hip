// Exchange-K kernels: instantiations and launcher (see w4a16_xk.hpp).  Its own translation unit so that the library builds in parallel.
#include "w4a16_common.hpp"

#include <hip/hip_ext.h>

#include "w4a16_args.hpp"
#include "w4a16_wide.hpp"
#include "w4a16_xk.hpp"
#include "w4a16_xk_host.hpp"

namespace quick_amd {

static_assert(kXkZoneBytesHost == kXkZoneBytes, "exchange zone size");

XkConfig xk_default_config(int mb, int s) { return XkConfig{mb, s, 5, 4, 0}; }

template <int MB, int NBUF, int WD, int S, int ABL = 0, int KQ = 2>
static bool xk_go(const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  constexpr unsigned lds = NBUF * MB * 8192 * (KQ / 2);
  const dim3 grid(workgroups), block(KQ == 4 ? 1024 : ((ABL & 4096) ? 768 : 512));
  const int gm = a.G == 128 ? 0 : (a.G % 128 == 0 ? 1 : -1);
  if (gm < 0) return false;
  const XwRest rest{a.bias, a.residual, a.Y, a.slabs, a.counters, a.dbg, a.span, a.silu_mul, a.G};
#define QA_XK_K(GMV)                                                                                               \
  do {                                                                                                             \
    auto kfn = w4a16_xk_kernel<MB, GMV, NBUF, WD, S, ABL, KQ>;                                                     \
    static std::atomic<unsigned long long> attr_set{0};                                                                \
    (void)lds_limit_once(attr_set, (const void*)kfn, (int)lds);                                                    \
    hipExtLaunchKernelGGL(kfn, grid, block, lds, st, start, stop, 0, a.X, a.QW, a.S, a.M, a.K, a.N, a.tpg, a.ksplit, a.kt_per_split, a.xcd_gm,    \
                          rest);                                                                                   \
  } while (0)
  if constexpr (ABL != 0) {
    if (gm != 0) return false;
    QA_XK_K(0);
  } else {
    if (gm == 0) QA_XK_K(0);
    else QA_XK_K(1);
  }
#undef QA_XK_K
  return true;
}

template <int MB, int S, int ABL = 0>
static bool xl_go(const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  constexpr unsigned lds = (MB == 2 ? 4 : 3) * MB * 8192 + 5 * (8192 + 512);
  const dim3 grid(workgroups), block(768);
  const int gm = a.G == 128 ? 0 : (a.G % 128 == 0 ? 1 : -1);
  if (gm < 0 || (ABL != 0 && gm != 0)) return false;
  const XwRest rest{a.bias, a.residual, a.Y, a.slabs, a.counters, a.dbg, a.span, a.silu_mul, a.G};
#define QA_XL_K(GMV)                                                                                               \
  do {                                                                                                             \
    auto kfn = w4a16_xl_kernel<MB, GMV, S, ABL>;                                                                   \
    static std::atomic<unsigned long long> attr_set{0};                                                                \
    (void)lds_limit_once(attr_set, (const void*)kfn, (int)lds);                                                    \
    hipExtLaunchKernelGGL(kfn, grid, block, lds, st, start, stop, 0, a.X, a.QW, a.S, a.M, a.K, a.N, a.tpg, a.ksplit, a.kt_per_split, a.xcd_gm,    \
                          rest);                                                                                   \
  } while (0)
  if constexpr (ABL != 0) {
    QA_XL_K(0);
  } else {
    if (gm == 0) QA_XL_K(0);
    else QA_XL_K(1);
  }
#undef QA_XL_K
  return true;
}

template <int MB, int ABL = 0>
static bool xl_go_s(int s, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  switch (s) {
    case 1: return xl_go<MB, 1, ABL>(a, workgroups, st, start, stop);
    case 2: return xl_go<MB, 2, ABL>(a, workgroups, st, start, stop);
    case 4: return xl_go<MB, 4, ABL>(a, workgroups, st, start, stop);
    case 8: return xl_go<MB, 8, ABL>(a, workgroups, st, start, stop);
    default: return false;
  }
}

template <int MB, int NBUF, int WD, int ABL = 0>
static bool xk_go_s(int s, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  switch (s) {
    case 1: return xk_go<MB, NBUF, WD, 1, ABL>(a, workgroups, st, start, stop);
    case 2: return xk_go<MB, NBUF, WD, 2, ABL>(a, workgroups, st, start, stop);
    case 4: return xk_go<MB, NBUF, WD, 4, ABL>(a, workgroups, st, start, stop);
    case 8: return xk_go<MB, NBUF, WD, 8, ABL>(a, workgroups, st, start, stop);
    default: return false;
  }
}

bool xk_launch(const XkConfig& c, const GemmArgs& a, int workgroups, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
  if (c.loader) {  // twelve waves: four loaders + eight compute waves (an experiment the product library does not carry)
#ifdef QUICK_AMD_TOOLS
    if (c.abl == 0) return c.mb == 4 ? xl_go_s<4>(c.s, a, workgroups, st, start, stop) : (c.mb == 2 ? xl_go_s<2>(c.s, a, workgroups, st, start, stop) : false);
    if (c.abl == 32) {
      if (c.mb == 2 && c.s == 1) return xl_go<2, 1, 32>(a, workgroups, st, start, stop);
      if (c.mb == 4 && c.s == 2) return xl_go<4, 2, 32>(a, workgroups, st, start, stop);
      if (c.mb == 4 && c.s == 1) return xl_go<4, 1, 32>(a, workgroups, st, start, stop);
      return false;
    }
    if (c.abl == 64) {
      if (c.mb == 2 && c.s == 1) return xl_go<2, 1, 64>(a, workgroups, st, start, stop);
      if (c.mb == 4 && c.s == 2) return xl_go<4, 2, 64>(a, workgroups, st, start, stop);
      if (c.mb == 4 && c.s == 4) return xl_go<4, 4, 64>(a, workgroups, st, start, stop);
      if (c.mb == 2 && c.s == 8) return xl_go<2, 8, 64>(a, workgroups, st, start, stop);
    }
#endif
    return false;
  }
  if (c.kq == 4) {  // sixteen waves, four per SIMD (an experiment the product library does not carry: level with eight waves)
#ifdef QUICK_AMD_TOOLS
    if (c.mb != 2 || c.s != 1 || c.nbuf != 5 || c.wd != 4) return false;
    if (c.abl == 0) return xk_go<2, 5, 4, 1, 0, 4>(a, workgroups, st, start, stop);
    if (c.abl == 32) return xk_go<2, 5, 4, 1, 32, 4>(a, workgroups, st, start, stop);
    if (c.abl == 64) return xk_go<2, 5, 4, 1, 64, 4>(a, workgroups, st, start, stop);
    if (c.abl == 262208) return xk_go<2, 5, 4, 1, 262208, 4>(a, workgroups, st, start, stop);  // the two halves in step (no skew)
#endif
    return false;
  }
  const int key = c.mb * 100 + c.nbuf * 10 + c.wd;
#ifdef QUICK_AMD_TOOLS
  if (c.abl) {  // timing experiments (tools builds): phase stamps, and the launch without the cross-CU exchange (wrong results)
    if (key == 454 && c.s == 2) switch (c.abl) {   // (all with phase stamps: tools/xk_phases.py reads the K loop's own time)
      case 64: return xk_go<4, 5, 4, 2, 64>(a, workgroups, st, start, stop);
      case 65: return xk_go<4, 5, 4, 2, 65>(a, workgroups, st, start, stop);    // loads only
      case 66: return xk_go<4, 5, 4, 2, 66>(a, workgroups, st, start, stop);    // no loads
      case 82: return xk_go<4, 5, 4, 2, 82>(a, workgroups, st, start, stop);    // no loads, no B-fragment reads
      case 74: return xk_go<4, 5, 4, 2, 74>(a, workgroups, st, start, stop);    // no loads, no dequantisation
      case 90: return xk_go<4, 5, 4, 2, 90>(a, workgroups, st, start, stop);    // MFMAs + barrier
      case 98: return xk_go<4, 5, 4, 2, 98>(a, workgroups, st, start, stop);    // no loads, no barrier
      case 122: return xk_go<4, 5, 4, 2, 122>(a, workgroups, st, start, stop);  // MFMAs only
      case 68: return xk_go<4, 5, 4, 2, 68>(a, workgroups, st, start, stop);    // no cross-CU exchange
      case 192: return xk_go<4, 5, 4, 2, 192>(a, workgroups, st, start, stop);  // no counted wait at the end of a stage
      case 320: return xk_go<4, 5, 4, 2, 320>(a, workgroups, st, start, stop);  // no x pieces in the K loop
      case 576: return xk_go<4, 5, 4, 2, 576>(a, workgroups, st, start, stop);  // no weight loads in the K loop
      case 1088: return xk_go<4, 5, 4, 2, 1088>(a, workgroups, st, start, stop);  // one x piece with every unit
      case 4160: return xk_go<4, 5, 4, 2, 4160>(a, workgroups, st, start, stop);  // four loader waves issue the x pieces
      case 4672: return xk_go<4, 5, 4, 2, 4672>(a, workgroups, st, start, stop);  // ... and the compute waves issue no weight loads either
      case 8256: return xk_go<4, 5, 4, 2, 8256>(a, workgroups, st, start, stop);  // + clocks spent in the counted wait / at the barrier
      case 8258: return xk_go<4, 5, 4, 2, 8258>(a, workgroups, st, start, stop);  // ... without loads
      case 16720: return xk_go<4, 5, 4, 2, 16720>(a, workgroups, st, start, stop);  // weight loads only, always the same (cached) stage, no B-fragment reads
      case 80: return xk_go<4, 5, 4, 2, 80>(a, workgroups, st, start, stop);      // loads, no B-fragment reads
      case 72: return xk_go<4, 5, 4, 2, 72>(a, workgroups, st, start, stop);      // loads, no dequantisation
      case 88: return xk_go<4, 5, 4, 2, 88>(a, workgroups, st, start, stop);      // loads, MFMAs, barrier
      case 336: return xk_go<4, 5, 4, 2, 336>(a, workgroups, st, start, stop);    // weight loads only, no B-fragment reads
      default: return false;
    }
    if (key == 254 && c.s == 1 && c.abl == 65536) return xk_go<2, 5, 4, 1, 65536>(a, workgroups, st, start, stop);  // staggered issue
    if (key == 254 && c.s == 1 && c.abl == 65600) return xk_go<2, 5, 4, 1, 65600>(a, workgroups, st, start, stop);
    if (key == 454 && c.s == 2 && c.abl == 65536) return xk_go<4, 5, 4, 2, 65536>(a, workgroups, st, start, stop);
    if (key == 454 && c.s == 2 && c.abl == 65600) return xk_go<4, 5, 4, 2, 65600>(a, workgroups, st, start, stop);
    if (key == 254 && c.s == 1 && c.abl == 131072) return xk_go<2, 5, 4, 1, 131072>(a, workgroups, st, start, stop);  // <= 128 registers: two workgroups per CU
    if (key == 254 && c.s == 1 && c.abl == 131136) return xk_go<2, 5, 4, 1, 131136>(a, workgroups, st, start, stop);
    if (key == 244 && c.s == 1 && c.abl == 131072) return xk_go<2, 4, 4, 1, 131072>(a, workgroups, st, start, stop);
    if (key == 244 && c.s == 1 && c.abl == 131136) return xk_go<2, 4, 4, 1, 131136>(a, workgroups, st, start, stop);
    if (key == 244 && c.s == 1 && c.abl == 64) return xk_go<2, 4, 4, 1, 64>(a, workgroups, st, start, stop);
    if (key == 254 && c.s == 1 && c.abl == 524352) return xk_go<2, 5, 4, 1, 524352>(a, workgroups, st, start, stop);    // a barrier behind every second stage only
    if (key == 254 && c.s == 1 && c.abl == 1572928) return xk_go<2, 5, 4, 1, 1572928>(a, workgroups, st, start, stop);  // ... every fourth
    if (key == 454 && c.s == 2 && c.abl == 524352) return xk_go<4, 5, 4, 2, 524352>(a, workgroups, st, start, stop);
    if (key == 254 && c.s == 1 && c.abl == 32768) return xk_go<2, 5, 4, 1, 32768>(a, workgroups, st, start, stop);  // write-through y stores
    if (key == 254 && c.s == 1 && c.abl == 64) return xk_go<2, 5, 4, 1, 64>(a, workgroups, st, start, stop);
    if (key == 454 && c.s == 4 && c.abl == 64) return xk_go<4, 5, 4, 4, 64>(a, workgroups, st, start, stop);
    if (key == 254 && c.s == 8 && c.abl == 64) return xk_go<2, 5, 4, 8, 64>(a, workgroups, st, start, stop);
    return false;
  }
  // ring / queue geometries that only the tuning sweeps ask for
  if (key == 444) return xk_go_s<4, 4, 4>(c.s, a, workgroups, st, start, stop);
  if (key == 445) return xk_go_s<4, 4, 5>(c.s, a, workgroups, st, start, stop);
  if (key == 455) return xk_go_s<4, 5, 5>(c.s, a, workgroups, st, start, stop);
  if (key == 456) return xk_go_s<4, 5, 6>(c.s, a, workgroups, st, start, stop);
  if (key == 244) return xk_go_s<2, 4, 4>(c.s, a, workgroups, st, start, stop);
#else
  if (c.abl && c.abl != 32) return false;
#endif
  if (c.abl == 32) {  // in-kernel span stamps (quick_w4a16_gemm_span, bench.py): the shapes of the BASELINE sweep and their neighbours
    if (key == 254 && c.s == 1) return xk_go<2, 5, 4, 1, 32>(a, workgroups, st, start, stop);
    if (key == 254 && c.s == 2) return xk_go<2, 5, 4, 2, 32>(a, workgroups, st, start, stop);
    if (key == 254 && c.s == 4) return xk_go<2, 5, 4, 4, 32>(a, workgroups, st, start, stop);
    if (key == 454 && c.s == 1) return xk_go<4, 5, 4, 1, 32>(a, workgroups, st, start, stop);
    if (key == 454 && c.s == 2) return xk_go<4, 5, 4, 2, 32>(a, workgroups, st, start, stop);
    return false;
  }
  if (key == 454) return xk_go_s<4, 5, 4>(c.s, a, workgroups, st, start, stop);
  if (key == 254) return xk_go_s<2, 5, 4>(c.s, a, workgroups, st, start, stop);
  return false;
}

}  // namespace quick_amd
