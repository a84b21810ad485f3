// "Exchange-K" kernels (r03): 128- / 64-token tiles whose K slices live on DIFFERENT compute units and meet without a
// last arriver -- the co-resident slices of a tile swap parts of their fp32 partial tiles through write-through mailboxes and
// each finishes its own part.  Replaces the reference's `[split_k, M, N]` scratch + torch `.sum(0)`
// (csrc/gemm_cuda_quick.cu:1468, 1515) where a tile count below the CU count forces a K split (M = 64 .. 512 on 4096^2).
//
// Shape: tile = MB*32 tokens x 128 channels, EIGHT waves = 4 along N x 2 along the k16 steps of a stage (two per SIMD: one
// wave alone issues at most ~5 instructions per 32-cycle MFMA, and a 128-token wave needs 4.3 -- 13 VALU of dequantisation + 4
// B-fragment reads per 4 MFMAs -- before any load or wait; two waves issue VALU, LDS, SALU and vector-memory instructions
// side by side).  A wave owns all MB*32 tokens x 32 channels for the k16 steps of its parity: 13 / MB VALU per MFMA.
//
// Pipeline (what r02's ring kernel could not do at 128 tokens -- three 41 KiB slots = vmcnt(0) every stage):
//   x        LDS-DMA into a ring of NBUF slots of MB*8 KiB (x only), NBUF - 3 whole stages in flight behind a counted vmcnt;
//   weights  HBM -> registers directly, WD stages ahead, into a queue of ACCUMULATION registers (a[0 : 5 WD)): the loads are
//            inline asm that names the AGPRs, so the compiler neither sees an asynchronous result it could copy too early nor
//            spends VGPRs on the queue; a landed set is moved to VGPRs (5 v_accvgpr_read) one stage before use;
//   all vector-memory instructions are asm, every wait is an explicit counted s_waitcnt (hipcc's own waitcnt pass only sees
//   the LDS reads).
//
// The way out, S = K slices of a tile on S compute units (S = 1: no exchange):
//   1. the two K parities of a channel quarter swap halves through LDS: wave (wn, wk) ends up with blocks wk*MB/2 .. of 32 tokens;
//   2. each wave's holding (MB/2 * 16 registers) is cut into S parts; part p goes to slice p's mailbox with 16-byte write-through
//      (sc1) stores -- NEGATED, and never -0.0, so that a stored word is never 0x00000000 -- and the receiving wave polls the
//      mailbox words themselves (sc1 loads) until none is zero: no flag, no atomic, no drain on the producer's side.  The
//      consumer zeroes the mailbox again (the workspace's "exchange zone" is all-zero between launches).  Sums are taken in
//      slice order, so a result does not depend on timing.  The poll is bounded and traps: the protocol needs the S slices of
//      a tile co-resident, which the host guarantees by launching at most one workgroup per CU (make_plan).
//   3. S <= 2 with whole 32-token blocks per wave: f16 image of the finished rows in LDS, whole rows to y (as wide_store_tile);
//      otherwise 8-byte stores straight from the registers (the tiles of those launches are small).
#pragma once

#include <utility>

namespace quick_amd {

// ------------------------------------------------------------------------------------------------
// the weight queue in accumulation registers: set J = a[4 J : 4 J + 3] (16 bytes of packed weights) + a[24 + J] ((scale, zero) word)
// ------------------------------------------------------------------------------------------------
template <int J>
struct XkSet;
#define QA_XK_SET(J, A0, A1, A2, A3, AS)                                                                                      \
  template <>                                                                                                                 \
  struct XkSet<J> {                                                                                                           \
    static __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rw, unsigned vw, unsigned sw, __amdgpu_buffer_rsrc_t rs, \
                                                 unsigned vs, unsigned ss) {                                                  \
      asm volatile("buffer_load_dwordx4 a[" #A0 ":" #A3 "], %0, %1, %2 offen\n\tbuffer_load_dword a" #AS ", %3, %4, %5 offen" \
                   :: "v"(vw), "s"(rw), "s"(sw), "v"(vs), "s"(rs), "s"(ss)                                                    \
                   : "memory", "a" #A0, "a" #A1, "a" #A2, "a" #A3, "a" #AS);                                                  \
    }                                                                                                                         \
    static __device__ __forceinline__ void read(u32x4& q, uint32_t& sz) {                                                     \
      uint32_t q0, q1, q2, q3, s;                                                                                             \
      asm volatile("v_accvgpr_read_b32 %0, a" #A0 "\n\tv_accvgpr_read_b32 %1, a" #A1 "\n\tv_accvgpr_read_b32 %2, a" #A2      \
                   "\n\tv_accvgpr_read_b32 %3, a" #A3 "\n\tv_accvgpr_read_b32 %4, a" #AS                                      \
                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(s));                                                        \
      q = u32x4{q0, q1, q2, q3};                                                                                              \
      sz = s;                                                                                                                 \
    }                                                                                                                         \
  };
QA_XK_SET(0, 0, 1, 2, 3, 24)
QA_XK_SET(1, 4, 5, 6, 7, 25)
QA_XK_SET(2, 8, 9, 10, 11, 26)
QA_XK_SET(3, 12, 13, 14, 15, 27)
QA_XK_SET(4, 16, 17, 18, 19, 28)
QA_XK_SET(5, 20, 21, 22, 23, 29)
#undef QA_XK_SET

template <int V>
using xk_ic = std::integral_constant<int, V>;

// ------------------------------------------------------------------------------------------------
// mailbox traffic (exchange zone of the workspace): all asm, see the header comment
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void xk_mail_store(__amdgpu_buffer_rsrc_t r, unsigned off, floatx4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, /*sc1*/ 16);
}
__device__ __forceinline__ void xk_mail_store(__amdgpu_buffer_rsrc_t r, unsigned off, float v0, float v1) {
  __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(uint32_t, v0), __builtin_bit_cast(uint32_t, v1)}, r, off, 0, /*sc1*/ 16);
}
// N mailbox granules (16 or 8 bytes per lane) at voff + so[k] (so: wave-uniform byte offsets), loaded AND waited for inside ONE asm
// statement: the results are valid where the compiler believes they are.
template <int N>
__device__ __forceinline__ void xk_mail_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, const unsigned (&so)[N], u32x4 (&v)[N]) {
  static_assert(N >= 1 && N <= 7, "granules per poll");
  if constexpr (N == 1)
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0])
                 : "v"(voff), "s"(r), "s"(so[0])
                 : "memory");
  else if constexpr (N == 2)
    asm volatile("buffer_load_dwordx4 %0, %2, %3, %4 offen sc1\n\t"
                 "buffer_load_dwordx4 %1, %2, %3, %5 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1])
                 : "memory");
  else if constexpr (N == 3)
    asm volatile("buffer_load_dwordx4 %0, %3, %4, %5 offen sc1\n\t"
                 "buffer_load_dwordx4 %1, %3, %4, %6 offen sc1\n\t"
                 "buffer_load_dwordx4 %2, %3, %4, %7 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1]), "s"(so[2])
                 : "memory");
  else if constexpr (N == 4)
    asm volatile("buffer_load_dwordx4 %0, %4, %5, %6 offen sc1\n\t"
                 "buffer_load_dwordx4 %1, %4, %5, %7 offen sc1\n\t"
                 "buffer_load_dwordx4 %2, %4, %5, %8 offen sc1\n\t"
                 "buffer_load_dwordx4 %3, %4, %5, %9 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3])
                 : "memory");
  else if constexpr (N == 5)
    asm volatile("buffer_load_dwordx4 %0, %5, %6, %7 offen sc1\n\t"
                 "buffer_load_dwordx4 %1, %5, %6, %8 offen sc1\n\t"
                 "buffer_load_dwordx4 %2, %5, %6, %9 offen sc1\n\t"
                 "buffer_load_dwordx4 %3, %5, %6, %10 offen sc1\n\t"
                 "buffer_load_dwordx4 %4, %5, %6, %11 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]), "s"(so[4])
                 : "memory");
  else if constexpr (N == 6)
    asm volatile("buffer_load_dwordx4 %0, %6, %7, %8 offen sc1\n\t"
                 "buffer_load_dwordx4 %1, %6, %7, %9 offen sc1\n\t"
                 "buffer_load_dwordx4 %2, %6, %7, %10 offen sc1\n\t"
                 "buffer_load_dwordx4 %3, %6, %7, %11 offen sc1\n\t"
                 "buffer_load_dwordx4 %4, %6, %7, %12 offen sc1\n\t"
                 "buffer_load_dwordx4 %5, %6, %7, %13 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]), "s"(so[4]), "s"(so[5])
                 : "memory");
  else if constexpr (N == 7)
    asm volatile("buffer_load_dwordx4 %0, %7, %8, %9 offen sc1\n\t"
                 "buffer_load_dwordx4 %1, %7, %8, %10 offen sc1\n\t"
                 "buffer_load_dwordx4 %2, %7, %8, %11 offen sc1\n\t"
                 "buffer_load_dwordx4 %3, %7, %8, %12 offen sc1\n\t"
                 "buffer_load_dwordx4 %4, %7, %8, %13 offen sc1\n\t"
                 "buffer_load_dwordx4 %5, %7, %8, %14 offen sc1\n\t"
                 "buffer_load_dwordx4 %6, %7, %8, %15 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]), "s"(so[4]), "s"(so[5]), "s"(so[6])
                 : "memory");
}
template <int N>
__device__ __forceinline__ void xk_mail_load8(__amdgpu_buffer_rsrc_t r, unsigned voff, const unsigned (&so)[N], u32x2 (&v)[N]) {
  static_assert(N >= 1 && N <= 7, "granules per poll");
  if constexpr (N == 1)
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0])
                 : "v"(voff), "s"(r), "s"(so[0])
                 : "memory");
  else if constexpr (N == 2)
    asm volatile("buffer_load_dwordx2 %0, %2, %3, %4 offen sc1\n\t"
                 "buffer_load_dwordx2 %1, %2, %3, %5 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1])
                 : "memory");
  else if constexpr (N == 3)
    asm volatile("buffer_load_dwordx2 %0, %3, %4, %5 offen sc1\n\t"
                 "buffer_load_dwordx2 %1, %3, %4, %6 offen sc1\n\t"
                 "buffer_load_dwordx2 %2, %3, %4, %7 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1]), "s"(so[2])
                 : "memory");
  else if constexpr (N == 4)
    asm volatile("buffer_load_dwordx2 %0, %4, %5, %6 offen sc1\n\t"
                 "buffer_load_dwordx2 %1, %4, %5, %7 offen sc1\n\t"
                 "buffer_load_dwordx2 %2, %4, %5, %8 offen sc1\n\t"
                 "buffer_load_dwordx2 %3, %4, %5, %9 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3])
                 : "memory");
  else if constexpr (N == 5)
    asm volatile("buffer_load_dwordx2 %0, %5, %6, %7 offen sc1\n\t"
                 "buffer_load_dwordx2 %1, %5, %6, %8 offen sc1\n\t"
                 "buffer_load_dwordx2 %2, %5, %6, %9 offen sc1\n\t"
                 "buffer_load_dwordx2 %3, %5, %6, %10 offen sc1\n\t"
                 "buffer_load_dwordx2 %4, %5, %6, %11 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]), "s"(so[4])
                 : "memory");
  else if constexpr (N == 6)
    asm volatile("buffer_load_dwordx2 %0, %6, %7, %8 offen sc1\n\t"
                 "buffer_load_dwordx2 %1, %6, %7, %9 offen sc1\n\t"
                 "buffer_load_dwordx2 %2, %6, %7, %10 offen sc1\n\t"
                 "buffer_load_dwordx2 %3, %6, %7, %11 offen sc1\n\t"
                 "buffer_load_dwordx2 %4, %6, %7, %12 offen sc1\n\t"
                 "buffer_load_dwordx2 %5, %6, %7, %13 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]), "s"(so[4]), "s"(so[5])
                 : "memory");
  else if constexpr (N == 7)
    asm volatile("buffer_load_dwordx2 %0, %7, %8, %9 offen sc1\n\t"
                 "buffer_load_dwordx2 %1, %7, %8, %10 offen sc1\n\t"
                 "buffer_load_dwordx2 %2, %7, %8, %11 offen sc1\n\t"
                 "buffer_load_dwordx2 %3, %7, %8, %12 offen sc1\n\t"
                 "buffer_load_dwordx2 %4, %7, %8, %13 offen sc1\n\t"
                 "buffer_load_dwordx2 %5, %7, %8, %14 offen sc1\n\t"
                 "buffer_load_dwordx2 %6, %7, %8, %15 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6])
                 : "v"(voff), "s"(r), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]), "s"(so[4]), "s"(so[5]), "s"(so[6])
                 : "memory");
}

// tile / slice coordinates.  XCD-aware order (a.xcd_gm > 0): workgroup b runs on XCD b % 8; XCD x serves K slice x % S only, and the
// 8 / S XCDs of a slice form an xcd_gm x gn grid over the (token, channel) tiles, each taking a compact rectangle -- its L2 fetches
// the x rows and weight columns of that rectangle for ONE K slice.  The S slices of a tile are S consecutive workgroups either way.
struct XkTile {
  int mb, nb, ks, tile, kt_lo, kt_hi, nstage, m0;
};
template <int MB, int S>
__device__ __forceinline__ XkTile xk_tile(const GemmArgs& a) {
  XkTile t;
  const int NB = a.N >> 7, MBk = (a.M + MB * 32 - 1) / (MB * 32);
  const int b = blockIdx.x;
  const int gmrows = a.xcd_gm & 255;   // (bits 8..: log2 of the exchange poll limit)
  if (gmrows > 0) {
    const int xcd = b & 7, idx = b >> 3;
    t.ks = xcd % S;
    const int g = xcd / S, gn = (8 / S) / gmrows;
    const int mcnt = MBk / gmrows, ncnt = NB / gn;
    t.mb = (g / gn) * mcnt + idx / ncnt;
    t.nb = (g % gn) * ncnt + idx % ncnt;
  } else {
    t.ks = b % S;
    t.mb = (b / S) / NB;
    t.nb = (b / S) % NB;
  }
  t.tile = t.mb * NB + t.nb;
  const int KT = a.K >> 7;
  t.kt_lo = t.ks * a.kt_per_split;
  t.kt_hi = min(KT, t.kt_lo + a.kt_per_split);
  t.nstage = t.kt_hi - t.kt_lo;
  t.m0 = t.mb * MB * 32;
  return t;
}

constexpr unsigned kXkZoneBytes = 16u << 20;   // exchange zone of the workspace (all-zero between launches), see workspace_need()

// The way out shared by the exchange-K kernels: K parities swapped through LDS, slices exchanged through the mailboxes, finished rows
// stored (see the header comment).  `acc`: this wave's partial sums over its K parity and its workgroup's K slice.
template <int MB, int S, int ABL>
__device__ __forceinline__ void xk_way_out(const GemmArgs& a, const XkTile& t, floatx16 (&acc)[1][MB], char* smem, int lane, int wave, int wn, int wk,
                                           int rho, int h, unsigned long long (&ph)[6]) {
  // ---- 1. the two K parities of a channel quarter swap halves through LDS ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the replayed loads of the last stages)
  __builtin_amdgcn_s_barrier();                     // the ring is free
  constexpr int HB = MB / 2;                        // 32-token blocks a wave keeps: blocks wk * HB .. + HB - 1
  floatx16 part[HB];
  {
    floatx4* ex = (floatx4*)smem;  // inbox [wave][HB * 4 quads][lane]
    floatx4* out = ex + (size_t)((wn + 4 * (1 - wk)) * (HB * 4)) * 64 + lane;
    const floatx4* in = ex + (size_t)(wave * (HB * 4)) * 64 + lane;
    auto swap = [&](auto wkc) __attribute__((always_inline)) {
      constexpr int WKV = decltype(wkc)::value;
#pragma unroll
      for (int j = 0; j < HB; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const floatx16& v = acc[0][(1 - WKV) * HB + j];
          out[(j * 4 + c) * 64] = floatx4{v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
        }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < HB; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const floatx4 v = in[(j * 4 + c) * 64];
#pragma unroll
          for (int r = 0; r < 4; ++r) part[j][4 * c + r] = acc[0][WKV * HB + j][4 * c + r] + v[r];
        }
    };
    if (wk == 0) swap(xk_ic<0>{});
    else swap(xk_ic<1>{});
  }
  if constexpr (ABL & 64) ph[3] = __builtin_amdgcn_s_memrealtime();

  // ---- 2. the S slices of the tile swap parts through the mailboxes ----
  // [r04] Nobody has to be co-resident any more (the r03 poll spun for ~4 s and trapped), and the fast path pays
  // no atomic for it.  A state word per (tile, owner, wave): bit 0 = the owner GAVE its part UP, bit 1 = somebody CLAIMED it.
  //   owner:  polls `limit` ticks for its partners' shares.  If they do not come: own share to its self box [dst][dst], then (stores
  //           acknowledged) bit 0, then (acknowledged) ONE more look at the boxes -- all there after all: claim and finish; else leave.
  //   sender: once its shares are acknowledged in memory it reads the owners' state words (a plain load, issued behind the polling so that
  //           no poll waits for it, looked at after its own rows have left).  Bit 0 set, bit 1 not: look at ALL S boxes of that part; all
  //           full: claim (compare-and-swap 1 -> 3, the only atomic of the protocol) and, having won, finish the part from the boxes --
  //           slice order, the same sums whoever finishes -- store it straight from the registers, zero the boxes and the word.
  // Whoever's contribution (shares, or self box + flag) is acknowledged LAST looks afterwards and therefore sees everything: exactly one
  // party finishes an abandoned part.  No spinning without bound, no trap, no dependence on dispatch order.
  constexpr int H = HB * 16, P = H / S;       // registers a wave holds / finishes
  constexpr int GR = P >= 4 ? 4 : 2;          // registers per mailbox granule
  constexpr int NGR = P / GR;                 // granules per part
  static_assert(P >= 2 && P % GR == 0, "part size");
  float fin[P];
  bool mine = true;
  auto flat = [&](int f) { return part[f / 16][f % 16]; };
  const __amdgpu_buffer_rsrc_t ry_d = __builtin_amdgcn_make_buffer_rsrc((void*)a.Y, 0, (unsigned)((size_t)a.M * (a.silu_mul ? a.N >> 1 : a.N) * 2), 0x00020000);
  // part `ks_part` of this wave (P registers, flattened register ks_part * P + f = (block j, accumulator register r): token
  // m0 + ((wk * HB + j) * 32 + rho), channels 128 nb + 32 wn + 8 (r / 4) + 4 h + r % 4) straight from the registers, GR channels per store
  auto store_direct = [&](int ks_part, const float (&v)[P]) __attribute__((always_inline)) {
    if (a.silu_mul) {   // (only where a part is whole 32-token blocks: make_plan / run_gemm)
      if constexpr (P % 16 == 0) {
#pragma unroll
        for (int jj = 0; jj < P / 16; ++jj) {
          const int m = t.m0 + ((wk * HB + (ks_part * P) / 16 + jj) * 32 + rho);
          if (m >= a.M) continue;
#pragma unroll
          for (int c = 0; c < 4; c += 2) {
            half4_t o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = silu_mul_f16((half_t)v[jj * 16 + 4 * c + r], (half_t)v[jj * 16 + 4 * c + 4 + r]);
            const unsigned yoff = (unsigned)(((size_t)m * (a.N >> 1) + (t.nb * 64 + wn * 16 + (c >> 1) * 8 + 4 * h)) * 2);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry_d, yoff, 0, /*sc1*/ 16);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int g = 0; g < NGR; ++g) {
      const int f0 = ks_part * P + g * GR;       // flattened register of v[g * GR]
      const int j = f0 / 16, r0 = f0 % 16;
      const int m = t.m0 + ((wk * HB + j) * 32 + rho);
      const int n = t.nb * 128 + wn * 32 + 8 * (r0 / 4) + 4 * h + (r0 % 4);
      if (m < a.M) {
        half_t o[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r) o[r] = (half_t)(v[g * GR + r] + (a.bias ? (float)a.bias[n + r] : 0.f));
        if (a.residual) {
#pragma unroll
          for (int r = 0; r < GR; ++r) o[r] = (half_t)((float)o[r] + (float)a.residual[(size_t)m * a.N + n + r]);
        }
        const unsigned yoff = (unsigned)(((size_t)m * a.N + n) * 2);
        if constexpr (GR == 4) {  // write-through, as the image path
          const half4_t ov = {o[0], o[1], o[2], o[3]};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ov), ry_d, yoff, 0, /*sc1*/ 16);
        } else {
          const half2_t ov = {o[0], o[1]};
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, ov), ry_d, yoff, 0, /*sc1*/ 16);
        }
      }
    }
  };
  // what this slice still owes after its own rows have left: boxes (and, after a give-up, the state word) back to zero, parts their owners gave up
  unsigned flag_lane = 0u;  // lane q: partner q's state word, read after this slice's shares were acknowledged
  if constexpr (S == 1) {
#pragma unroll
    for (int f = 0; f < P; ++f) fin[f] = flat(f);
  }
  // mailbox [tile][dst][src (the self box at src == dst)][wave][granule][lane] x GR * 4 bytes
  constexpr unsigned GBYTES = GR * 4 * 64, WBYTES = NGR * GBYTES, BOX = 8 * WBYTES;
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)a.slabs, 0, kXkZoneBytes, 0x00020000);
  const unsigned tbase = (unsigned)t.tile * (unsigned)(S * S) * BOX + (unsigned)wave * WBYTES + (unsigned)lane * (GR * 4);
  unsigned* state = S > 1 ? a.counters + ((unsigned)t.tile * S) * 8u + (unsigned)wave : nullptr;   // + dst * 8: state word of (tile, dst, wave)
  constexpr int TL = S > 1 ? (S - 1) * NGR : 1;
  unsigned so[TL];
  bool zero_self = false;   // (the owner gave up, found everything there after all and won the claim: self box and state word are its to clear)
  typedef float floatx2 __attribute__((ext_vector_type(2)));
  auto mail_store = [&](unsigned off, const float* v) __attribute__((always_inline)) {
    if constexpr (GR == 4) xk_mail_store(rz, off, floatx4{v[0], v[1], v[2], v[3]});
    else xk_mail_store(rz, off, v[0], v[1]);
  };
  // (16-byte stores with an SGPR offset carry their own wait states: gfx950 corrupts the store's data when the next VALU instruction
  // overwrites the data registers, and hipcc pads that hazard only for stores without an SGPR offset -- see xw_mail_store, w4a16_xw.hpp)
  auto mail_zero = [&](unsigned soff) __attribute__((always_inline)) {
    if constexpr (GR == 4) {
      const u32x4 zero = {0u, 0u, 0u, 0u};
      asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 3" : : "v"(zero), "v"(tbase), "s"(rz), "s"(soff) : "memory");
    } else {
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, rz, tbase, soff, 16);
    }
  };
  if constexpr (S > 1) {
    const unsigned limit = 1u << (((a.xcd_gm >> 8) & 31) ? ((a.xcd_gm >> 8) & 31) : 12);   // ticks of 10 ns; default 41 us
    auto exchange = [&](auto ksc) __attribute__((always_inline)) {
      constexpr int KS = decltype(ksc)::value;
      // send: part p to slice p, negated and never -0.0 (so never the bit pattern 0): w = v + 0.0 is never -0.0, -0.0 - w is -w exactly
#pragma unroll
      for (int p = 0; p < S; ++p) {
        if (p == KS) continue;
        const unsigned box = tbase + (unsigned)(p * S + KS) * BOX;
#pragma unroll
        for (int g = 0; g < NGR; ++g) {
          float v[GR];
#pragma unroll
          for (int r = 0; r < GR; ++r) v[r] = -0.0f - (flat(p * P + g * GR + r) + 0.0f);
          mail_store(box + g * GBYTES, v);
        }
      }
#pragma unroll
      for (int f = 0; f < P; ++f) fin[f] = flat(KS * P + f);
      if constexpr (!(ABL & 4)) {
        // receive: ONE poll fetches the boxes of all S - 1 sources (a round trip to memory each time: polled one after the other, seven
        // sources cost seven of them [r03 phase stamps, S = 8: 3.6 us]) and is repeated until no word of any of them is zero -- or `limit`
#pragma unroll
        for (int k = 0; k < TL; ++k) {
          const int si = k / NGR, src = si < KS ? si : si + 1;
          so[k] = (unsigned)(KS * S + src) * BOX + (unsigned)(k % NGR) * GBYTES;
        }
        float got[S - 1][P];
        auto poll = [&]() {
          bool ok = true;
          if constexpr (GR == 4) {
            u32x4 q[TL];
            xk_mail_load16<TL>(rz, tbase, so, q);
#pragma unroll
            for (int k = 0; k < TL; ++k) {
              const floatx4 fq = __builtin_bit_cast(floatx4, q[k]);  // (whole vector: hipcc's bit_cast of a vector ELEMENT reads element 0)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                ok = ok && q[k][r] != 0u;
                got[k / NGR][(k % NGR) * 4 + r] = -fq[r];
              }
            }
          } else {
            u32x2 q[TL];
            xk_mail_load8<TL>(rz, tbase, so, q);
#pragma unroll
            for (int k = 0; k < TL; ++k) {
              const floatx2 fq = __builtin_bit_cast(floatx2, q[k]);
              ok = ok && q[k][0] != 0u && q[k][1] != 0u;
              got[k][0] = -fq[0];
              got[k][1] = -fq[1];
            }
          }
          return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
        };
        // The first poll's wait also acknowledges the stores above: the shares are in memory.  Right behind it lane q asks whether partner q has
        // given its part up -- a plain load, as quick as the polls that may follow it, long back when it is looked at (after the rows have left).
        // Early is enough: an owner that gives up LATER looks at its boxes once more after raising its flag, and finds these shares.
        bool ok = poll();
        if (lane < S && lane != KS) flag_lane = __hip_atomic_load(state + lane * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (!ok && (unsigned)(__builtin_amdgcn_s_memrealtime() - t0) <= limit) {
          __builtin_amdgcn_s_sleep(4);
          ok = poll();
        }
        bool finish = ok;
        if (!ok) {
          // give the part up: own share (encoded as the ones that travel) to the self box, then the flag, then one more look
          const unsigned box = tbase + (unsigned)(KS * S + KS) * BOX;
#pragma unroll
          for (int g = 0; g < NGR; ++g) {
            float v[GR];
#pragma unroll
            for (int r = 0; r < GR; ++r) v[r] = -0.0f - (fin[g * GR + r] + 0.0f);
            mail_store(box + g * GBYTES, v);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_store(state + KS * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (poll()) {   // every share is there after all: whoever of the partners looks now sees the same, so claim
            unsigned expect = 1u;
            bool won = false;
            if (lane == 0) won = __hip_atomic_compare_exchange_strong(state + KS * 8, &expect, 3u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_readfirstlane((int)won)) {
              finish = true;
              zero_self = true;
            }
          }
        }
        mine = finish;
        if (finish) {
          // the shares are in registers: hand the boxes back zeroed NOW -- these stores are acknowledged under the rows' way out, not behind it
          // (behind it they cost 0.25 us of every launch, profiles/r04_xk_giveup_phases.txt)
#pragma unroll
          for (int k = 0; k < TL; ++k) mail_zero(so[k]);
          // sum in slice order, the own part at position KS
          float sum[P];
#pragma unroll
          for (int src = 0; src < S; ++src) {
#pragma unroll
            for (int f = 0; f < P; ++f) {
              const float v = src == KS ? fin[f] : got[src < KS ? src : src - 1][f];
              sum[f] = src == 0 ? v : sum[f] + v;
            }
          }
#pragma unroll
          for (int f = 0; f < P; ++f) fin[f] = sum[f];
        }
      }
    };
    if constexpr (S == 2) {
      if (t.ks == 0) exchange(xk_ic<0>{});
      else exchange(xk_ic<1>{});
    } else if constexpr (S == 4) {
      if (t.ks == 0) exchange(xk_ic<0>{});
      else if (t.ks == 1) exchange(xk_ic<1>{});
      else if (t.ks == 2) exchange(xk_ic<2>{});
      else exchange(xk_ic<3>{});
    } else {
      if (t.ks == 0) exchange(xk_ic<0>{});
      else if (t.ks == 1) exchange(xk_ic<1 % S>{});
      else if (t.ks == 2) exchange(xk_ic<2 % S>{});
      else if (t.ks == 3) exchange(xk_ic<3 % S>{});
      else if (t.ks == 4) exchange(xk_ic<4 % S>{});
      else if (t.ks == 5) exchange(xk_ic<5 % S>{});
      else if (t.ks == 6) exchange(xk_ic<6 % S>{});
      else exchange(xk_ic<7 % S>{});
    }
  }
  if constexpr (ABL & 64) ph[4] = __builtin_amdgcn_s_memrealtime();

  // ---- 3. the way out.  fin[f], f = 0 .. P - 1, is flattened register ks * P + f = (block j, accumulator register r):
  //         token m0 + ((wk * HB + j) * 32 + rho), channels 128 nb + 32 wn + 8 (r / 4) + 4 h + r % 4 ----
  if constexpr (P % 16 == 0) {
    // whole blocks: f16 image of the finished rows in LDS (row lr, 16-byte chunk q at (lr * CPR + (q ^ lr % 8)) * 16), whole rows out.
    // A wave that gave its part up writes nothing, and its 32 channels of its blocks' rows are left to whoever finishes the part.
    constexpr int NBL = P / 16;               // blocks a wave finishes; local row block index = wk * NBL + jj
    constexpr int ROWS = 2 * NBL * 32;        // rows this workgroup finishes
    const int jb0 = (t.ks * P) / 16;          // first finished block among the wave's HB
    __syncthreads();                          // (the K-parity inboxes have been read)
    unsigned* flags = (unsigned*)(smem + 64 * 1024);   // [wave]: this wave finished its part (the image is at most 32 KiB)
    if constexpr (S > 1)
      if (lane == 0) flags[wave] = mine ? 1u : 0u;
    auto image = [&](auto siluc) __attribute__((always_inline)) {
      constexpr bool SILU = decltype(siluc)::value != 0;
      constexpr int CPR = SILU ? 8 : 16;
      const unsigned r7 = (unsigned)rho & 7u;
      if (mine) {
#pragma unroll
        for (int jj = 0; jj < NBL; ++jj) {
          char* wrow = smem + ((wk * NBL + jj) * 32 + rho) * (CPR * 16) + h * 8;
#pragma unroll
          for (int c = 0; c < 4; c += SILU ? 2 : 1) {
            const unsigned q = (unsigned)wn * (SILU ? 2 : 4) + (SILU ? c / 2 : c);
            half4_t bv = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
            if constexpr (!SILU)
              if (a.bias) bv = *(const half4_t*)(a.bias + t.nb * 128 + wn * 32 + 8 * c + 4 * h);
            half4_t o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if constexpr (SILU) o[r] = silu_mul_f16((half_t)fin[jj * 16 + 4 * c + r], (half_t)fin[jj * 16 + 4 * c + 4 + r]);
              else o[r] = (half_t)(fin[jj * 16 + 4 * c + r] + (float)bv[r]);
            }
            *(half4_t*)(wrow + ((q ^ r7) << 4)) = o;
          }
        }
      }
      __syncthreads();
      const int ldy = SILU ? a.N >> 1 : a.N;
      const int q = (int)threadIdx.x % CPR;
      const unsigned ycol0 = (unsigned)((SILU ? t.nb * 64 : t.nb * 128) + q * 8);
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.Y, 0, (unsigned)((size_t)a.M * ldy * 2), 0x00020000);
      const half_t* rcol = (!SILU && a.residual) ? a.residual + t.nb * 128 + q * 8 : nullptr;
      constexpr int RPI = 512 / CPR;          // rows per pass of the workgroup
      // did the wave that owns this thread's chunk column (wn = the chunk's channel quarter) finish its rows?  One look per K parity, in front
      // of the loop: read inside it, next to the image reads, the two flag words cost every launch 0.5 us (profiles/r04_ab_exchange.txt)
      bool fin_ok[2] = {true, true};
      if constexpr (S > 1) {
        fin_ok[0] = flags[q / (CPR / 4)] != 0u;
        fin_ok[1] = flags[q / (CPR / 4) + 4] != 0u;
      }
#pragma unroll
      for (int it = 0; it < ROWS / RPI; ++it) {
        const int lr = it * RPI + (int)threadIdx.x / CPR;  // local row: block lr / 32 (= wkk * NBL + jj), token lr % 32
        half8_t v = *(const half8_t*)(smem + (lr * CPR + (q ^ (lr & 7))) * 16);
        const int blk = lr >> 5, wkk = blk / NBL, jj = blk % NBL;
        const int m = t.m0 + ((wkk * HB + jb0 + jj) * 32 + (lr & 31));
        const bool owner_ok = fin_ok[wkk];   // wave (wn, wk = wkk) finished these rows (one slice: nobody can have given anything up)
        if (m < a.M && owner_ok) {
          if (rcol) {
            const half8_t res = *(const half8_t*)(rcol + (size_t)m * a.N);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = (half_t)((float)v[r] + (float)res[r]);
          }
          // Write-through (sc1): the end of the launch then finds nothing of y dirty in the XCD's L2 -- the boundary to the next
          // kernel writes a plain-stored 4 MB result back at ~6 TB/s, 0.4-0.5 us of every 512-token step [r03 A/B, one session:
          // 24.4-24.6 against 24.9-25.0 us per step of a 20-launch graph]; ABL bit 32768 (tools builds) keeps the plain stores.
          if constexpr (!(ABL & 32768)) {  // (a buffer store the compiler knows: an asm store leaves its data registers unprotected)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, (unsigned)(((size_t)m * ldy + ycol0) * 2), 0, /*sc1*/ 16);
          } else {
            *(half8_t*)(a.Y + (size_t)m * ldy + ycol0) = v;
          }
        }
      }
    };
    if (a.silu_mul) image(xk_ic<1>{});
    else image(xk_ic<0>{});
  } else {
    // part of a block: straight from the registers, GR channels (8 or 4 bytes) per store.  (No SiLU * mul here: make_plan does not
    // route such launches to these shapes.)
    if (mine) store_direct(t.ks, fin);
  }

  // ---- 4. after the rows: (after a give-up won back) self box and state word to zero; parts whose owners gave up and which this slice is the one to finish ----
  if constexpr (S > 1 && !(ABL & 4)) {
    if (mine) {
      if (zero_self) {
#pragma unroll
        for (int g = 0; g < NGR; ++g) mail_zero((unsigned)(t.ks * S + t.ks) * BOX + (unsigned)g * GBYTES);
        if (lane == 0) __hip_atomic_store(state + t.ks * 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    for (int q = 0; q < S; ++q) {
      if (q == t.ks) continue;
      if ((unsigned)__builtin_amdgcn_readlane((int)flag_lane, q) != 1u) continue;   // not given up (or claimed already)
      // all S boxes of the part (the self box at src == q), one look: complete?
      float val[S][P];
      bool full = true;
#pragma unroll
      for (int src = 0; src < S; ++src) {
        const unsigned boff = (unsigned)(q * S + src) * BOX;
#pragma unroll
        for (int g = 0; g < NGR; ++g) {
          if constexpr (GR == 4) {
            u32x4 v[1];
            const unsigned o1[1] = {boff + (unsigned)g * GBYTES};
            xk_mail_load16<1>(rz, tbase, o1, v);
            const floatx4 fq = __builtin_bit_cast(floatx4, v[0]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              full = full && v[0][r] != 0u;
              val[src][g * 4 + r] = -fq[r];
            }
          } else {
            u32x2 v[1];
            const unsigned o1[1] = {boff + (unsigned)g * GBYTES};
            xk_mail_load8<1>(rz, tbase, o1, v);
            const floatx2 fq = __builtin_bit_cast(floatx2, v[0]);
            full = full && v[0][0] != 0u && v[0][1] != 0u;
            val[src][g * 2] = -fq[0];
            val[src][g * 2 + 1] = -fq[1];
          }
        }
      }
      if (__builtin_amdgcn_ballot_w64(!full) != 0ull) continue;   // somebody's share is still on its way: that somebody will look later
      unsigned expect = 1u;
      bool won = false;
      if (lane == 0) won = __hip_atomic_compare_exchange_strong(state + q * 8, &expect, 3u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!__builtin_amdgcn_readfirstlane((int)won)) continue;
      float sum[P];
#pragma unroll
      for (int src = 0; src < S; ++src) {   // slice order
#pragma unroll
        for (int f = 0; f < P; ++f) sum[f] = src == 0 ? val[0][f] : sum[f] + val[src][f];
      }
#pragma unroll
      for (int src = 0; src < S; ++src) {
#pragma unroll
        for (int g = 0; g < NGR; ++g) mail_zero((unsigned)(q * S + src) * BOX + (unsigned)g * GBYTES);
      }
      if (lane == 0) __hip_atomic_store(state + q * 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      store_direct(q, sum);
    }
  }
}

// Way out of the sixteen-wave kernel (64 x 128 tile, one slice, K quarters wk = 0 .. 3 per channel quarter wn).  A wave holds two
// 32-token blocks x 16 registers of partial sums = 8 quads (block j, quad c); quad (j, c) is FINISHED by wave wk' = 2 j + c / 2 of the
// same channel quarter, which adds the four partials in K order.  Then the f16 image of the tile in LDS and whole rows out, as in
// xk_way_out.  Token of (j, rho): m0 + 32 j + rho; channels of quad c: 128 nb + 32 wn + 8 c + 4 h + (0 .. 3).
template <int ABL>
__device__ __forceinline__ void xk_way_out16(const GemmArgs& a, const XkTile& t, floatx16 (&acc)[1][2], char* smem, int lane, int wave, int wn, int wk,
                                             int rho, int h, unsigned long long (&ph)[6]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the replayed loads of the last stages)
  __builtin_amdgcn_s_barrier();                     // the ring is free
  floatx4* ex = (floatx4*)smem;                     // inbox [wave (wn, owner)][source slot 0 .. 2][quad 0 .. 1][lane]
  auto quad = [&](int j, int c) { const floatx16& v = acc[0][j]; return floatx4{v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]}; };
  float fin[8];
  auto reduce = [&](auto wkc) __attribute__((always_inline)) {
    constexpr int WK = decltype(wkc)::value;
#pragma unroll
    for (int o = 0; o < 4; ++o) {  // send the two quads owner o finishes
      if (o == WK) continue;
      floatx4* out = ex + (size_t)(((wn + 4 * o) * 3 + (WK < o ? WK : WK - 1)) * 2) * 64 + lane;
      out[0] = quad(o >> 1, 2 * (o & 1));
      out[64] = quad(o >> 1, 2 * (o & 1) + 1);
    }
    __syncthreads();
    const floatx4* in = ex + (size_t)((wave * 3) * 2) * 64 + lane;
    const floatx4 own0 = quad(WK >> 1, 2 * (WK & 1)), own1 = quad(WK >> 1, 2 * (WK & 1) + 1);
    floatx4 s0, s1;
#pragma unroll
    for (int src = 0; src < 4; ++src) {  // K order, the own partial at position WK
      const floatx4 v0 = src == WK ? own0 : in[((src < WK ? src : src - 1) * 2) * 64];
      const floatx4 v1 = src == WK ? own1 : in[((src < WK ? src : src - 1) * 2 + 1) * 64];
      if (src == 0) { s0 = v0; s1 = v1; }
      else { s0 += v0; s1 += v1; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { fin[r] = s0[r]; fin[4 + r] = s1[r]; }
  };
  if (wk == 0) reduce(xk_ic<0>{});
  else if (wk == 1) reduce(xk_ic<1>{});
  else if (wk == 2) reduce(xk_ic<2>{});
  else reduce(xk_ic<3>{});
  if constexpr (ABL & 64) ph[3] = ph[4] = __builtin_amdgcn_s_memrealtime();
  const int j = wk >> 1, c0 = 2 * (wk & 1);
  __syncthreads();  // (the inboxes have been read)
  auto image = [&](auto siluc) __attribute__((always_inline)) {
    constexpr bool SILU = decltype(siluc)::value != 0;
    constexpr int CPR = SILU ? 8 : 16;
    const unsigned r7 = (unsigned)rho & 7u;
    char* wrow = smem + (j * 32 + rho) * (CPR * 16) + h * 8;
    if constexpr (SILU) {
      half4_t o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = silu_mul_f16((half_t)fin[r], (half_t)fin[4 + r]);
      const unsigned q = (unsigned)wn * 2 + (unsigned)(c0 >> 1);
      *(half4_t*)(wrow + ((q ^ r7) << 4)) = o;
    } else {
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int c = c0 + cc;
        half4_t bv = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        if (a.bias) bv = *(const half4_t*)(a.bias + t.nb * 128 + wn * 32 + 8 * c + 4 * h);
        half4_t o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)(fin[4 * cc + r] + (float)bv[r]);
        const unsigned q = (unsigned)wn * 4 + (unsigned)c;
        *(half4_t*)(wrow + ((q ^ r7) << 4)) = o;
      }
    }
    __syncthreads();
    const int ldy = SILU ? a.N >> 1 : a.N;
    const int q = (int)threadIdx.x % CPR;
    const unsigned ycol0 = (unsigned)((SILU ? t.nb * 64 : t.nb * 128) + q * 8);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.Y, 0, (unsigned)((size_t)a.M * ldy * 2), 0x00020000);
    const half_t* rcol = (!SILU && a.residual) ? a.residual + t.nb * 128 + q * 8 : nullptr;
    constexpr int RPI = 1024 / CPR;  // rows per pass of the workgroup: 64 (one pass) or 128 (half the threads)
    const int lr = (int)threadIdx.x / CPR;
    if (lr < 64) {
      half8_t v = *(const half8_t*)(smem + (lr * CPR + (q ^ (lr & 7))) * 16);
      const int m = t.m0 + lr;
      if (m < a.M) {
        if (rcol) {
          const half8_t res = *(const half8_t*)(rcol + (size_t)m * a.N);
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = (half_t)((float)v[r] + (float)res[r]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, (unsigned)(((size_t)m * ldy + ycol0) * 2), 0, /*sc1*/ 16);
      }
    }
    (void)RPI;
  };
  if (a.silu_mul) image(xk_ic<1>{});
  else image(xk_ic<0>{});
}

// ABL (tools builds only): 64 = s_memrealtime stamps at the phase boundaries of every wave into a.dbg; 4 = no cross-CU exchange (the
// own part is finished without the other slices: wrong results, the launch minus the exchange); 1 / 2 / 8 / 16 as in wide_compute (no
// compute / no loads in the K loop / no dequantisation / no B-fragment reads); 32 = no barrier in the K loop; 128 = no counted wait at the
// end of a stage; 256 / 512 = no x pieces / no weight loads in the K loop; 1024 = one x piece with every unit instead of two with two.
// KQ = 4 (64-token tiles, one slice): SIXTEEN waves = 4 (N) x 4 (K) -- four waves per SIMD.  A stage is 256 k: waves 0-7 run the
// first 128 k of it exactly as the eight-wave kernel runs a stage (same pieces, same weight loads, same counted waits), waves 8-15
// the second 128 k in the other half of the slot; one barrier per stage for all sixteen; the four K quarters of a channel quarter
// are summed through LDS on the way out.  [r03: two co-resident eight-wave workgroups per CU (tools build, <= 128 registers) run a
// 64 x 128 x 4096 tile in 24.4 k clocks each-equivalent where one alone takes 33.8 k -- scripts/gpu_occ.sh]
template <int MB, int GM, int NBUF, int WD, int S, int ABL = 0, int KQ = 2>
__global__ __launch_bounds__(KQ == 4 ? 1024 : ((ABL & 4096) ? 768 : 512), (KQ == 4 || (ABL & 131072)) ? 4 : 1) void w4a16_xk_kernel(const half_t* __restrict__ aX, const u32x4* __restrict__ aQW, const half_t* __restrict__ aS, int aM, int aK, int aN, int a_tpg, int a_ksplit, int a_kps, int a_xcd_gm, const XwRest rest) {  // (131072, tools: <= 128 registers, two workgroups per CU)
  const GemmArgs a = xw_args(aX, aQW, aS, aM, aK, aN, a_tpg, a_ksplit, a_kps, a_xcd_gm, rest);
  constexpr int KH = KQ / 2;   // 128-k halves of a stage
  constexpr int NW = 4 * KQ, NG = 1;
  static_assert(KQ == 2 || (KQ == 4 && MB == 2 && S == 1 && !(ABL & 4096)), "sixteen waves: 64-token tiles, one slice");
  constexpr bool LD = (ABL & 4096) != 0;  // experiment: four extra LOADER waves issue every x piece, the eight compute waves only their weights
  constexpr int SLOT = MB * 8192 * KH;
  constexpr int XI = MB;       // x LDS-DMA instructions per wave and stage (MB * 32 rows / (8 waves * 4 rows))
  constexpr int L = XI + 2;    // vector-memory instructions per wave and stage (+ weights, + (scale, zero) word)
  constexpr int NU = 4;        // units (k16 steps of this wave's parity) per stage
  constexpr int PEND = (NBUF - 3) * L;
  static_assert(GM <= 1 && (MB == 2 || MB == 4), "G % 128 == 0, 64- or 128-token tiles");
  static_assert(NBUF >= 3 && NBUF * SLOT <= 160 * 1024 && WD >= NBUF - 1 && WD >= 3 && WD <= 6 && PEND <= 63, "ring / queue geometry");
  static_assert(S == 1 || S == 2 || S == 4 || S == 8, "K slices per tile");
  static_assert(MB * 16384 <= NBUF * SLOT, "the K-parity exchange must fit in the ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NBUF * SLOT

  if constexpr (ABL & 32) span_stamp(a.span, 0);  // (32: per-wave start / end stamps of the in-kernel span clock, nothing else changes)
  unsigned long long ph[6], cyc = 0;  // (cyc: shader clocks spent in the K loop, s_memtime)
  if constexpr (ABL & 64) ph[0] = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int wn = wave & 3, wk = wave >> 2;   // wk: K parity (bit 0) and, with sixteen waves, the 128-k half of the stage (bit 1)
  const int wave8 = wave & 7, half = wave >> 3;
  const int rho = lane & 31, h = lane >> 5;
  XkTile t = xk_tile<MB, S>(a);
  const int ct0 = (t.nb * 4 + wn) * 2;
  const WideBufs<XI> b = wide_bufs<MB, 1, 2>(a, t.m0, ct0, lane, wave8);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned xdst = lds_base + (unsigned)wave8 * 1024u + (unsigned)half * (MB * 8192u);  // + slot + i * 8 KiB
  const unsigned xrd = ((lds_base + (unsigned)rho * 256u + (unsigned)((h ^ (rho & 15)) << 4)) ^ ((unsigned)(wk & 1) << 5)) + (unsigned)half * (MB * 8192u);
  // k-tile (128 k) of this wave in stage q of the tile's K range; past the end: a replay of the last one (prefetches nobody uses)
  auto ktof = [&](int q) { return min(t.kt_lo + KH * q + half, t.kt_hi - 1); };
  if constexpr (KQ == 4) t.nstage /= KH;   // (K % 256 == 0: make_plan)

  if constexpr (LD) {
    static_assert(!LD || (NBUF == 5 && MB == 4), "loader experiment: 128-token tiles, five slots");
    if (wave >= 8) {  // loader wave lw: piece i = rows 16 i + 4 lw + lane / 16 of the token tile, i = 0 .. 7
      const unsigned lw = (unsigned)wave - 8u;
      const unsigned row = 4u * lw + ((unsigned)lane >> 4);
      unsigned voff[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        voff[i] = (unsigned)min(t.m0 + 16 * i + (int)row, a.M - 1) * (unsigned)a.K * 2u + 16u * (((unsigned)lane & 15u) ^ (row & 15u));
      const unsigned dst = lds_base + lw * 1024u;
      auto fillx = [&](int q, unsigned slot) {
        const int kt = min(t.kt_lo + q, t.kt_hi - 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) lds_dma16(b.x, voff[i], (unsigned)kt * 256u, dst + slot + i * 4096);
      };
      fillx(0, 0u); fillx(1, SLOT); fillx(2, 2 * SLOT); fillx(3, 3 * SLOT);
      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // x stage 0
      __builtin_amdgcn_s_barrier();                      // A
      fillx(4, 4 * SLOT);
      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // x stage 1
      __builtin_amdgcn_s_barrier();                      // M (stage 0, before the first read of stage 1)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // x stage 2
      __builtin_amdgcn_s_barrier();                      // end of stage 0
      unsigned slot = 0u;                                // slot of stage s + 4 = slot of stage s - 1
      for (int s2 = 1; s2 < t.nstage; ++s2) {
        fillx(s2 + 4, slot);
        slot = slot + SLOT >= (unsigned)(NBUF * SLOT) ? 0u : slot + SLOT;
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // x stage s + 2
        __builtin_amdgcn_s_barrier();                      // end of stage s
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // the ring is free
      return;
    }
  }
  auto issue_x = [&](int i, int kt, unsigned slot) {
    if constexpr (!LD) lds_dma16(b.x, b.x_voff[i], (unsigned)kt * 256u, xdst + slot + i * (8 * 1024));
  };
  auto issue_w = [&](auto jc, int kt) {
    const unsigned g = (unsigned)group_index<GM>(kt, 0, a.tpg, a.G);
    XkSet<decltype(jc)::value>::issue(b.w, b.w_voff, (unsigned)kt * 1024u, b.s, b.s_voff, g * 64u);
  };
  auto read_w = [&](auto jc, WideW<1, GM>& w) {
    XkSet<decltype(jc)::value>::read(w.lo[0], w.sz[0][0]);
    w.hi[0] = w.lo[0];
  };

  floatx16 acc[1][MB];
  wide_zero<MB, 1>(acc);
  const DqConsts dq = make_dq_consts();

  // FAST start (the shipped geometry, five slots / four sets): the prologue asks only for what stage 0 needs -- weight sets 0, 1 and x
  // stage 0 (and 1 where stage 0 reads it from its second unit on: 64-token tiles) -- and stage 0 itself issues the rest of the
  // ring next to its MFMAs.  Every CU's vector-memory path moves 64 B per clock: a prologue that asks for all four stages (164 KB
  // at 128 tokens) spends ~1.5 us ISSUING before the first wait [r03 phase stamps: 2.6-2.8 us to the first MFMA, with or without
  // waiting for stage 1].  Other geometries (tuning sweeps) keep the plain prologue, in the order the steady state would have
  // issued it: [W(0 .. WD - NBUF)], then W(WD - NBUF + 1 + q), X(q) for q = 0 .. NBUF - 2.
  constexpr bool FAST = NBUF == 5 && WD == 4 && !(ABL & 2) && !LD;
  constexpr int DEPTH = wide_bdepth<MB, 1>();
  constexpr int UM = NU - DEPTH;            // the unit of a stage that first reads the NEXT stage's tokens
  constexpr int PX = UM >= 2 ? 1 : 2;       // x stages the FAST prologue asks for
  constexpr int PER = XI / 2;               // steady state: x pieces with units 1 and 2 (nothing with the last unit of the stage)
  auto x_stage = [&](int q) {               // all pieces of x stage q (q < NBUF: its slot is q)
    const int kt = ktof(q);
#pragma unroll
    for (int i = 0; i < XI; ++i) issue_x(i, kt, (unsigned)q * SLOT);
  };
  auto pro_w = [&](auto jc) { issue_w(jc, ktof(decltype(jc)::value)); };
  if constexpr (FAST) {
    pro_w(xk_ic<0>{});
    pro_w(xk_ic<1>{});
    x_stage(0);
    if constexpr (PX == 2) x_stage(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PX - 1) * XI) : "memory");  // weight sets 0, 1 and x stage 0 have landed
  } else {
    constexpr int E = WD - NBUF + 1;  // sets issued ahead of the first x stage
    if constexpr (E > 0) pro_w(xk_ic<0>{});
    if constexpr (E > 1) pro_w(xk_ic<1>{});
    if constexpr (E > 2) pro_w(xk_ic<2>{});
    if constexpr (E > 3) pro_w(xk_ic<3>{});
    pro_w(xk_ic<E>{});
    x_stage(0);
    if constexpr (NBUF > 2) { pro_w(xk_ic<E + 1>{}); x_stage(1); }
    if constexpr (NBUF > 3) { pro_w(xk_ic<E + 2>{}); x_stage(2); }
    if constexpr (NBUF > 4) { pro_w(xk_ic<E + 3>{}); x_stage(3); }
    static_assert(NBUF <= 5, "prologue written out for up to five slots");
    // start as soon as x stage 0 and weight sets 0, 1 are there; stage 0 waits for x stage 1 itself, just before its first read of it
    constexpr int INIT = LD ? 2 * (WD - 2) : (NBUF - 2) * L - (E == 0 ? 2 : 0);  // what the prologue issued behind X(0) (and W(1))
    static_assert(INIT <= 63, "vmcnt field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INIT) : "memory");
  }
  __builtin_amdgcn_s_barrier();
  WideW<1, GM> wc, wnx;
  read_w(xk_ic<0>{}, wc);
  WideCarry<MB, 1, GM> carry;
  wide_prepare<MB, 1, GM, true, 2>(carry, wc, xrd, dq);
  if (wk) __builtin_amdgcn_s_setprio(1);  // the later-dispatched half loses every VALU arbitration otherwise (MI355X_MICROARCH.md, two waves per SIMD)
  if constexpr (ABL & 64) { ph[1] = __builtin_amdgcn_s_memrealtime(); cyc = __builtin_amdgcn_s_memtime(); }

  // vmcnt bookkeeping.  Steady state: stage s issues [W(s + WD), X(s + NBUF - 1)] = L instructions per wave; "x stage s + 2 and weight
  // set s + 2 have landed" at the end of stage s is vmcnt((NBUF - 3) L).  FAST: stage 0 issues W(2) W(3) W(4) X(PX) .. X(4), in that
  // order, so X(2) is followed by 2 XI instructions at the end of stage 0 and X(3) by 2 XI + 2 at the end of stage 1; before its unit
  // UM stage 0 waits for X(1) (followed by X(2) at 128 tokens; by W(2..4) and X(2) at 64, where X(1) was part of the prologue).
  constexpr int END0 = FAST ? 2 * XI : PEND, END1 = FAST ? 2 * XI + 2 : PEND;
  constexpr int MID = FAST ? (PX == 1 ? XI : 6 + XI) : (NBUF - 3) * L + 2 + (UM - 1) * PER;
  static_assert(MID <= 63, "vmcnt field");
  unsigned cur = 0u, nxt = (unsigned)SLOT, fill = (unsigned)(NBUF - 1) * SLOT;  // slots of stage s, s + 1, s + NBUF - 1
  unsigned long long seg_wait = 0, seg_bar = 0;
  auto stage = [&](auto jc, int s) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::value;
    const int ktx = ktof(s + NBUF - 1), ktw = (ABL & 16384) ? t.kt_lo : ktof(s + WD);  // (16384: the same, cache-resident weight stage every time)
    read_w(xk_ic<(J + 1) % WD>{}, wnx);  // W(s + 1): landed since the wait that ended stage s - 1
    wide_compute<MB, 1, GM, (ABL & 27), true, 2>(wc, wnx, xrd + cur, xrd + nxt, dq, acc, carry, [&](int u) {
      if constexpr (KQ == 4 && !(ABL & 262144)) {
        if (u == 2) __builtin_amdgcn_s_barrier();  // sixteen waves: the OTHER half's end of stage (see the skew below)
      }
      if constexpr (!(ABL & 2)) {
        bool first = false;
        if constexpr (J == 0) first = s == 0;
        if (first) {  // stage 0: wait for x stage 1 before the first read of it; FAST: fill the rest of the ring
          if (u == UM) {
            if constexpr (!LD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MID) : "memory");
            __builtin_amdgcn_s_barrier();  // ... in every wave
          }
          if constexpr (FAST) {
            if (u == 0) {
              pro_w(xk_ic<2>{});
              pro_w(xk_ic<3>{});
              issue_w(xk_ic<0>{}, ktw);
              x_stage(PX);
            }
            if (PX == 1 && u == 1) x_stage(2);
            if (u == (PX == 1 ? 2 : 1)) x_stage(3);
            if (u == (PX == 1 ? 3 : 2)) x_stage(4);   // (= this stage's own share: ktx, fill)
            return;
          }
        }
        if constexpr (ABL & 65536) {
          // (experiment: STAGGERED issue.  The barrier aligns the eight waves, so with fixed issue points all of them hand their
          // vector-memory instructions to the CU's one address path in the same few hundred clocks and wait for it together -- both
          // waves of every SIMD at once, nobody left to issue MFMAs.  Here wave (wn, wk) issues its whole share of the stage with
          // unit (wn + 2 wk) % 4: two waves per unit, never the two of one SIMD.)
          if (u == ((wn + 2 * wk) & 3)) {
            issue_w(xk_ic<J>{}, ktw);
#pragma unroll
            for (int i = 0; i < XI; ++i) issue_x(i, ktx, fill);
          }
          return;
        }
        if constexpr (!(ABL & 512))
          if (u == 0) issue_w(xk_ic<J>{}, ktw);  // set J held W(s), which has been in VGPRs since stage s - 1
        if constexpr (!(ABL & 256)) {
          if constexpr ((ABL & 1024) && XI == 4) {
            issue_x(u, ktx, fill);               // (experiment: one piece with every unit)
          } else if (u == 1 || u == 2) {
#pragma unroll
            for (int i = 0; i < PER; ++i) issue_x((u - 1) * PER + i, ktx, fill);
          }
        }
      }
    });
    unsigned long long tq0 = 0, tq1 = 0;
    if constexpr (ABL & 8192) tq0 = __builtin_amdgcn_s_memtime();
    if constexpr (!(ABL & 2) && !(ABL & 128)) {  // x stage s + 2 and weight set s + 2 have landed ...
      bool s0 = false, s1 = false;
      if constexpr (J == 0) s0 = s == 0;
      if constexpr (J == 1) s1 = s == 1;
      if constexpr (LD) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // (only the weights: set s + 2 was issued two stages ago)
      else if (s0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(END0) : "memory");
      else if (s1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(END1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PEND) : "memory");
    }
    if constexpr (ABL & 8192) tq1 = __builtin_amdgcn_s_memtime();
    if constexpr (ABL & 524288) {  // (timing experiment, results may be wrong: a barrier only behind every second / fourth stage)
      if ((s & ((ABL & 1048576) ? 3 : 1)) == 1) __builtin_amdgcn_s_barrier();
    } else if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();  // ... in every wave; everybody is done with x stage s
    if constexpr (ABL & 8192) {  // (experiment: shader clocks a wave spends in the counted wait / at the barrier, summed over the stages)
      const unsigned long long tq2 = __builtin_amdgcn_s_memtime();
      seg_wait += tq1 - tq0;
      seg_bar += tq2 - tq1;
    }
    wc = wnx;
    fill = cur;
    cur = nxt;
    nxt = nxt + SLOT >= (unsigned)(NBUF * SLOT) ? 0u : nxt + SLOT;
  };
  // Sixteen waves: the two halves share nothing in the K loop but the workgroup's one barrier.  Run in step, all four waves of a SIMD
  // reach the end of a stage -- counted wait, barrier, first fragments of the next stage -- together and the matrix pipe idles through
  // it.  So the second half runs HALF A STAGE behind the first: it starts one barrier late, every wave joins a barrier in the middle
  // of its stage (the other half's end of stage) and at its own end, and the first half joins one more at the very end.
  if constexpr (KQ == 4 && !(ABL & 262144)) {
    if (half) __builtin_amdgcn_s_barrier();
  }
  for (int base = 0; base < t.nstage; base += WD) {
    stage(xk_ic<0>{}, base);
    if (base + 1 < t.nstage) stage(xk_ic<1>{}, base + 1);
    if (base + 2 < t.nstage) stage(xk_ic<2>{}, base + 2);
    if constexpr (WD > 3) if (base + 3 < t.nstage) stage(xk_ic<3 % WD>{}, base + 3);
    if constexpr (WD > 4) if (base + 4 < t.nstage) stage(xk_ic<4 % WD>{}, base + 4);
    if constexpr (WD > 5) if (base + 5 < t.nstage) stage(xk_ic<5 % WD>{}, base + 5);
  }
  if constexpr (KQ == 4 && !(ABL & 262144)) {
    if (!half) __builtin_amdgcn_s_barrier();
  }
  if (wk) __builtin_amdgcn_s_setprio(0);
  if constexpr (ABL & 64) { ph[2] = __builtin_amdgcn_s_memrealtime(); cyc = __builtin_amdgcn_s_memtime() - cyc; }

  if constexpr (KQ == 4) xk_way_out16<ABL>(a, t, acc, smem, lane, wave, wn, wk, rho, h, ph);
  else xk_way_out<MB, S, ABL>(a, t, acc, smem, lane, wave, wn, wk, rho, h, ph);
  if constexpr (ABL & 32) span_stamp(a.span, 1);
  if constexpr (ABL & 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ph[5] = __builtin_amdgcn_s_memrealtime();
    if (a.dbg && lane == 0) {
      unsigned long long* o = a.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
#pragma unroll
      for (int i = 0; i < 6; ++i) o[i] = ph[i];
      o[6] = cyc;
      if constexpr (ABL & 8192) o[7] = (seg_wait << 32) | (seg_bar & 0xffffffffull);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Loader-wave flavour (r03, second half): the same tile, compute core and way out, but the eight compute waves issue NO vector-memory
// instruction at all.  Measured on the kernel above [profiles/r03_xk_anatomy.txt]: a vector-memory instruction blocks the wave that
// issues it until the CU's address path takes it, and with the memory pipe kept full by the prefetch that is hundreds of clocks a
// stage in which that wave issues no MFMA -- two weight loads per wave and stage alone cost the 128-token K loop 5 k of its 27 k
// clocks, while x pieces issued by four EXTRA waves cost 1.4 k.  So: twelve waves.  Waves 8, 9 bring x in (LDS-DMA, a ring of three
// slots: stage s + 2 is requested when stage s starts), waves 10, 11 the packed weights and the (scale, zero) words (LDS-DMA, a ring
// of five slots, stage s + 4: HBM-cold, so three stages of slack); each loader waits for what the NEXT stage needs and joins the one
// barrier of the stage.  The compute waves pick their 16 bytes of packed weights and their group word out of the slot with two LDS reads
// a stage (what r02's ring kernel did) and otherwise see only LDS, VALU and the matrix core.  168 registers per wave (three per SIMD).
// ------------------------------------------------------------------------------------------------
template <int MB, int GM, int S, int ABL = 0>
__global__ __launch_bounds__(768) void w4a16_xl_kernel(const half_t* __restrict__ aX, const u32x4* __restrict__ aQW, const half_t* __restrict__ aS, int aM, int aK, int aN, int a_tpg, int a_ksplit, int a_kps, int a_xcd_gm, const XwRest rest) {
  const GemmArgs a = xw_args(aX, aQW, aS, aM, aK, aN, a_tpg, a_ksplit, a_kps, a_xcd_gm, rest);
  constexpr int NXS = MB == 2 ? 4 : 3, NWS = 5;   // (64 tokens: a stage is too short for one stage of lookahead)
  constexpr int SLOTX = MB * 8192, SLOTW = 8192 + 512;
  constexpr int UM = 4 - wide_bdepth<MB, 1>();   // the unit of a stage that first reads the NEXT stage's tokens
  static_assert(GM <= 1 && (MB == 2 || MB == 4), "G % 128 == 0, 64- or 128-token tiles");
  static_assert(NXS * SLOTX + NWS * SLOTW <= 160 * 1024 && MB * 16384 <= NXS * SLOTX + NWS * SLOTW, "rings / K-parity exchange vs LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [x ring][weight ring]

  if constexpr (ABL & 32) span_stamp(a.span, 0);
  unsigned long long ph[6], cyc = 0;
  if constexpr (ABL & 64) ph[0] = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const XkTile t = xk_tile<MB, S>(a);
  const int KT = a.K >> 7, NGRP = a.K / a.G;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned wring = lds_base + NXS * SLOTX;
  auto ktq = [&](int q) { return min(t.kt_lo + q, t.kt_hi - 1); };  // (past the end of the slice: replays of its last stage)

  if (wave >= 8) {
    const unsigned lw = (unsigned)wave - 8u;
    if (lw < 2u) {  // ---- x loader: piece i = rows 8 i + 4 lw + lane / 16 of the token tile
      constexpr int NP = MB * 4;
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (unsigned)a.M * (unsigned)a.K * 2u, 0x00020000);
      unsigned voff[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const unsigned row = 8u * i + 4u * lw + ((unsigned)lane >> 4);
        voff[i] = (unsigned)min(t.m0 + (int)row, a.M - 1) * (unsigned)a.K * 2u + 16u * (((unsigned)lane & 15u) ^ (row & 15u));
      }
      const unsigned dst = lds_base + lw * 1024u;
      auto fillx = [&](int q, unsigned slot) {
        const unsigned so = (unsigned)ktq(q) * 256u;
#pragma unroll
        for (int i = 0; i < NP; ++i) lds_dma16(rx, voff[i], so, dst + slot + i * 2048);
      };
      constexpr int AHEAD = (NXS - 3) * NP;  // pieces that may still be in flight when "x stage s + 2 has landed"
      fillx(0, 0u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // A: the compute waves start on x stage 0 ...
      fillx(1, SLOTX);
      if constexpr (NXS > 3) fillx(2, 2u * SLOTX);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD) : "memory");
      __builtin_amdgcn_s_barrier();  // M: ... and read x stage 1 from unit UM of stage 0 on
      unsigned slot = (unsigned)(NXS - 1) * SLOTX;  // slot of stage s + NXS - 1
      for (int s2 = 0; s2 < t.nstage; ++s2) {
        fillx(s2 + NXS - 1, slot);
        slot = slot + SLOTX >= (unsigned)(NXS * SLOTX) ? 0u : slot + SLOTX;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD) : "memory");  // x stage s + 2 has landed
        __builtin_amdgcn_s_barrier();                                  // end of stage s
      }
    } else {  // ---- weight loader wl: tiles 4 wl .. 4 wl + 3 of the workgroup's eight 16-channel tiles, and their group words
      const unsigned wl = lw - 2u;
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.QW + (size_t)(t.nb * 8) * KT * 64), 0, 8u * (unsigned)KT * 1024u, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const uint32_t*)a.S + (size_t)(t.nb * 8) * NGRP * 16), 0, 8u * (unsigned)NGRP * 64u, 0x00020000);
      const unsigned wv = (unsigned)lane * 16u;
      const unsigned sv = ((unsigned)lane >> 4) * (unsigned)NGRP * 64u + ((unsigned)lane & 15u) * 4u;
      auto fillw = [&](int q, unsigned slot) {
        const int kt = ktq(q);
        const unsigned g = (unsigned)group_index<GM>(kt, 0, a.tpg, a.G);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          lds_dma16(rw, wv, (4u * wl + i) * (unsigned)KT * 1024u + (unsigned)kt * 1024u, wring + slot + (4u * wl + i) * 1024u);
        lds_dma4(rs, sv, 4u * wl * (unsigned)NGRP * 64u + g * 64u, wring + slot + 8192u + wl * 256u);
      };
      fillw(0, 0u);
      fillw(1, SLOTW);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // sets 0 and 1
      __builtin_amdgcn_s_barrier();                      // A
      fillw(2, 2u * SLOTW);
      fillw(3, 3u * SLOTW);
      __builtin_amdgcn_s_barrier();                      // M
      unsigned slot = 4u * SLOTW;                         // slot of stage s + 4
      for (int s2 = 0; s2 < t.nstage; ++s2) {
        fillw(s2 + 4, slot);
        slot = slot + SLOTW >= (unsigned)(NWS * SLOTW) ? 0u : slot + SLOTW;
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // set s + 2 has landed
        __builtin_amdgcn_s_barrier();                       // end of stage s
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the replays behind the end of the slice)
    __builtin_amdgcn_s_barrier();                      // the rings are free (xk_way_out's first barrier)
    return;
  }

  // ---- compute waves
  const int wn = wave & 3, wk = wave >> 2;
  const int rho = lane & 31, h = lane >> 5;
  const unsigned xrd = (lds_base + (unsigned)rho * 256u + (unsigned)((h ^ (rho & 15)) << 4)) ^ ((unsigned)wk << 5);
  typedef const __attribute__((address_space(3))) char* lds_ptr;
  const unsigned wrd = wring + (unsigned)(2 * wn + (rho >> 4)) * 1024u + 16u * ((unsigned)(rho & 15) + 16u * (unsigned)(h + 2 * wk));
  const unsigned srd = wring + 8192u + (unsigned)(2 * wn + (rho >> 4)) * 64u + (unsigned)(rho & 15) * 4u;
  auto read_w = [&](WideW<1, GM>& w, unsigned slot) {
    w.lo[0] = *(const __attribute__((address_space(3))) u32x4*)(lds_ptr)(uintptr_t)(wrd + slot);
    w.hi[0] = w.lo[0];
    w.sz[0][0] = *(const __attribute__((address_space(3))) uint32_t*)(lds_ptr)(uintptr_t)(srd + slot);
  };
  floatx16 acc[1][MB];
  wide_zero<MB, 1>(acc);
  const DqConsts dq = make_dq_consts();
  __builtin_amdgcn_s_barrier();  // A: x stage 0 and weight sets 0, 1 have landed
  WideW<1, GM> wc, wnx;
  read_w(wc, 0u);
  WideCarry<MB, 1, GM> carry;
  wide_prepare<MB, 1, GM, true, 2>(carry, wc, xrd, dq);
  if (wk) __builtin_amdgcn_s_setprio(1);
  if constexpr (ABL & 64) { ph[1] = __builtin_amdgcn_s_memrealtime(); cyc = __builtin_amdgcn_s_memtime(); }
  unsigned cur = 0u, nxt = (unsigned)SLOTX, wnext = (unsigned)SLOTW;
  for (int s = 0; s < t.nstage; ++s) {
    read_w(wnx, wnext);  // W(s + 1): landed since the barrier that ended stage s - 1
    wide_compute<MB, 1, GM, 0, true, 2>(wc, wnx, xrd + cur, xrd + nxt, dq, acc, carry, [&](int u) {
      if (u == UM && s == 0) __builtin_amdgcn_s_barrier();  // M (first stage only): x stage 1 has landed
    });
    __builtin_amdgcn_s_barrier();  // the loaders have x stage s + 2 and weight set s + 2; everybody is done with x stage s
    wc = wnx;
    cur = nxt;
    nxt = nxt + SLOTX >= (unsigned)(NXS * SLOTX) ? 0u : nxt + SLOTX;
    wnext = wnext + SLOTW >= (unsigned)(NWS * SLOTW) ? 0u : wnext + SLOTW;
  }
  if (wk) __builtin_amdgcn_s_setprio(0);
  if constexpr (ABL & 64) { ph[2] = __builtin_amdgcn_s_memrealtime(); cyc = __builtin_amdgcn_s_memtime() - cyc; }
  xk_way_out<MB, S, ABL>(a, t, acc, smem, lane, wave, wn, wk, rho, h, ph);
  if constexpr (ABL & 32) span_stamp(a.span, 1);
  if constexpr (ABL & 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ph[5] = __builtin_amdgcn_s_memrealtime();
    if (a.dbg && lane == 0) {
      unsigned long long* o = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
      for (int i = 0; i < 6; ++i) o[i] = ph[i];
      o[6] = cyc;
    }
  }
}

}  // namespace quick_amd
