// "Lean" small-M kernel (r05): 1..16 tokens, one 16-channel tile per workgroup, the waves split K.
//
// Replaces, for the decode regime, the reference's _m1n128k32 / _m16n128k32 kernels (csrc/gemm_cuda_quick.cu:1199-1290).
// Same arithmetic as the deferred-zero table flavour of the skinny kernel (w4a16_gemm.hip, skinny_compute_dz): x is the A
// operand of v_mfma_f32_16x16x32_f16, the still biased weights (biased8) the B operand, bias / zero point / scale leave once
// per 128-k unit in fp32.  What is new is the ORDER of the launch, built around the fact that vector-memory loads return in
// issue order (vmcnt counts in order):
//
//   * x is requested FIRST and lands in LDS by LDS-DMA (no VGPRs, any count), the weight ring right behind it.  A wave only
//     fetches the K range it multiplies itself, so nothing in the head waits for another wave: no workgroup barrier before the
//     reduction.  (r01-r04's kernel asked for the weights first; its x copy, unit-sum table and barrier then sat BEHIND the
//     weights' trip to HBM instead of under it.)
//   * the unit sums (A = sum x, C = sum b_k x) come out of the matrix core -- one MFMA per k-step against a constant operand,
//     as in the fragment flavour -- and so does the sum of squares of the RMSNorm prologue (x x^T, the diagonal); both run
//     while the weights are in flight.  What is left after a weight tile lands is 5 VALU per dword, 4 MFMAs and 8 FMAs;
//   * the kernel arguments arrive in ONE scalar round trip (a compact 128-byte block fetched by two s_load_dwordx16 at entry);
//   * every wave requests ALL its weight tiles (1 KiB each + their group words) up front -- one workgroup per channel block,
//     the chip's memory-level parallelism comes from several small workgroups per CU, not from a ring inside one;
//   * one barrier, in front of the reduction.
//
// G % 128 == 0.  RMSNorm prologue: x * weight in fp16 on the way (in LDS), 1 / rms on the fp32 result -- the rounding points of the
// fragment flavour (w4a16_gemm.hip, skinny_compute_dzf LN).  Epilogues: bias, residual, SiLU * mul as in skinny_finish.
#pragma once
#include "w4a16_args.hpp"

namespace quick_amd {

#ifndef QA_LEAN_WAUX
#define QA_LEAN_WAUX 2   // cache policy of the weight requests: nt (streaming).  Against the default policy, alternating builds in one session
                         // (profiles/r05_ab_lean_nt.txt): 1 x 4096 x 22016 10.2 -> 9.0 us in-kernel, x 12288 6.05 -> 5.63, 11008 x 4096 6.2 -> 5.8,
                         // 16 x 4096 x 22016 15.6 -> 14.5; level at 4096 x 4096 -- the weights are read once and x stays in L2
#endif
constexpr int kLeanStamps = 16;  // phase stamps per wave (tools builds): u64 s_memrealtime ticks

// LDS of one lean workgroup (host and device agree through these functions)
__host__ __device__ constexpr unsigned lean_red_bytes(int waves, int ntw, bool persist = false) {
  return (persist ? 2u : 1u) * (unsigned)(waves * ntw) * 1024u + (unsigned)waves * 64u;
}
__host__ __device__ inline unsigned lean_lds_bytes(int rows, int K, int waves, int ntw, bool ln, bool persist = false) {
  return lean_red_bytes(waves, ntw, persist) + (ln ? (unsigned)K * 2u : 0u) + (unsigned)rows * ((unsigned)K * 2u + 16u);
}

// Arguments: the ones every request of the head depends on come FIRST and as scalars -- the translation unit is built with
// -amdgpu-kernarg-preload-count=16, which has the command processor hand the first 14 dwords over in SGPRs at wave launch
// instead of through s_load (a round trip to memory the host has just written: L2-cold, ~0.5 us before the first request could
// go out).  The rest travels as an ordinary by-value block, fetched under the memory flight.
struct LeanRest {
  half_t* Y;
  const half_t* bias;
  const half_t* residual;
  int silu_mul;
  float ln_eps;
  unsigned long long* span;
  unsigned long long* dbg;
#ifdef QA_EXP_LEAN_OVERLAP   // (experiment builds only, DESIGN.md 9.2 / profiles/r06_launch_overlap.txt: a launch that starts BEFORE its predecessor has finished)
  const unsigned* wait_sig;  // the predecessor's arrival word: this launch requests its weights at entry and asks for x only once the word says the predecessor's rows are in memory
  unsigned* my_cnt;          // this launch's own exit counter (how many times it has run = which value of the arrival word to wait for)
  unsigned wait_per_exec;    // arrivals the predecessor adds per run
  unsigned my_per_exec;      // exits this launch adds per run
  unsigned* signal;          // this launch's arrival word for ITS successor (every storing wave adds one behind its rows)
#endif
};

// 64 lanes x 16 bytes: global (descriptor base + voff + soff) -> LDS (lds_addr + 16 * lane), exec-masked lanes write nothing.
// s_nop 4: the operands may come straight from a v_readfirstlane (VALU writes SGPR -> VMEM reads it: 5 wait states).
__device__ __forceinline__ void lean_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr, bool coherent = false) {
#ifdef QA_EXP_LEAN_OVERLAP
  if (coherent) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc0 sc1 lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return;
  }
#endif
  (void)coherent;
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc),
               "s"(soff)
               : "memory");
}

// One workgroup = NTW adjacent 16-channel tiles (blockIdx.x, XCD-aware order) x one block of 16 tokens (blockIdx.y); wave w owns the k tiles
// [KT w / WAVES, KT (w + 1) / WAVES), at most TMAX of them, of every channel tile, ALL requested up front.  NTW = 2 shares the x fragments and
// unit sums between two tiles and halves the workgroup count: on wide layers all of them are resident at once and the whole weight
// matrix is in flight after the first microsecond.  Straight-line code: hipcc's s_waitcnt placement is
// exact there (one counted wait per tile), while any loop carrying requests around its back edge made it drain the queue at the loop head.
template <int WAVES, int TMAX, int NTW, int GM, int ABL, bool LN, int MR, int NSETS = 1>
__global__ __launch_bounds__(WAVES * 64) void w4a16_lean_kernel(const half_t* __restrict__ aX, const u32x4* __restrict__ aQW, const half_t* __restrict__ aS,
                                                                const half_t* __restrict__ a_lnw, int aK, int aN, int aM, unsigned groups, int gx, unsigned tpg,
                                                                const LeanRest rest) {
  constexpr bool SPAN = ABL == 32, STAMP = ABL == 64;
  constexpr bool PERSIST = NSETS > 1;   // a workgroup walks the channel blocks nb, nb + gx, ...: x, unit sums and sum of squares made once
  static_assert(!(PERSIST && STAMP), "phase stamps: the one-block launches only");
  constexpr int NSZ = (TMAX + 3) / 4;
  constexpr int LSET = NTW * (TMAX + NSZ);   // requests per channel block and wave
  unsigned long long t_entry = 0;
  if constexpr (SPAN) t_entry = __builtin_amdgcn_s_memrealtime();  // (stored once the arguments are here: the span starts at the wave's first instruction)
  unsigned long long ts[kLeanStamps];
  if constexpr (STAMP) {
#pragma unroll
    for (int i = 0; i < kLeanStamps; ++i) ts[i] = 0;
    ts[0] = __builtin_amdgcn_s_memrealtime();
  }
  (void)tpg;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int KT = aK >> 7;
  const int row0 = blockIdx.y * 16, rows = min(16, aM - row0);
  constexpr bool ln = LN;  // (a_lnw != nullptr: the launcher picks the instantiation)
  // XCD-aware block order (workgroups are dealt to the XCDs round-robin: a contiguous run of channel blocks per XCD shares the
  // 128-byte lines of their group words)
  const int nb = (gx & 7) == 0 ? ((int)blockIdx.x & 7) * (gx >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const unsigned span_slot = ((blockIdx.y * (unsigned)gx + blockIdx.x) * WAVES + (unsigned)wave) & (kSpanWaves - 1);
  if constexpr (SPAN) {
    if (rest.span != nullptr) rest.span[span_slot] = t_entry;
  }
  const int kb = KT * wave / WAVES, ke = KT * (wave + 1) / WAVES, T = ke - kb;  // this wave's k tiles, 1 <= T <= TMAX (the planner's job)

  if constexpr (STAMP) ts[11] = __builtin_amdgcn_s_memrealtime();
  // LDS: [red WAVES x NTW x 1 KiB][ssq WAVES x 16 f32][ln weight K x 2 B][x rows x (2 K + 16) B]
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned ssq_off = (PERSIST ? 2u : 1u) * WAVES * NTW * 1024u, lnw_off = lean_red_bytes(WAVES, NTW, PERSIST);
  const unsigned x_off = lnw_off + (ln ? (unsigned)aK * 2u : 0u);
  const unsigned pitch = (unsigned)aK * 2u + 16u;
  const int nseg = (T + 3) >> 2;

#ifdef QA_EXP_LEAN_OVERLAP
  const bool defer = rest.wait_sig != nullptr;
#else
  constexpr bool defer = false;
#endif
  const auto issue_x = [&]() __attribute__((always_inline)) {
  // ---- 1. x (and the norm weight) of this wave's K range: LDS-DMA, first in the memory queue ----
  {
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)aX, 0, (unsigned)aM * (unsigned)aK * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc((void*)(ln ? a_lnw : aX), 0, (unsigned)aK * 2u, 0x00020000);
    if constexpr (STAMP) ts[12] = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < nseg; ++s) {
      const unsigned kbyte = (unsigned)(kb * 128 + 512 * s) * 2u;
      if (lane < 16 * (T - 4 * s)) {  // lanes past the wave's range write nothing (the neighbour wave owns those bytes)
        for (int tk = 0; tk < rows; ++tk)
          lean_dma16(xr, (unsigned)lane * 16u, (unsigned)(row0 + tk) * (unsigned)aK * 2u + kbyte, lds_base + x_off + (unsigned)tk * pitch + kbyte, defer);
        if (ln) lean_dma16(lr, (unsigned)lane * 16u, kbyte, lds_base + lnw_off + kbyte);
      }
    }
  }

  };
  if (!defer) issue_x();
  if constexpr (STAMP) ts[10] = __builtin_amdgcn_s_memrealtime();

  // ---- 2. the group words of four tiles per request (lane (n16, q): tile kb + 4 i + q -- with G = 128 one contiguous 256 bytes), then
  //         every weight tile of this wave (1 KiB each); tiles past T go out of the descriptor's range: zeros, no traffic ----
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)aQW, 0, (unsigned)aK * (unsigned)aN / 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)aS, 0, groups * (unsigned)aN * 4u, 0x00020000);
  uint32_t szr[NTW][NSZ];
  u32x4 wq[NTW][TMAX];
  // PERSIST: every pass of the block loop requests its block's tiles and consumes them before the next pass -- nothing is in flight around the
  // loop's back edge, which is what hipcc's s_waitcnt placement needs (requests it cannot see, kept in flight across the back edge, were tried:
  // it copies the destination registers at the loop head, before the data has landed).  What a persistent launch saves is the x traffic: at
  // 8..16 tokens a one-block workgroup fetches more bytes of x than of weights.
  const int nblocks = (aN >> 4) / NTW;
  const auto request = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NTW; ++c)
#pragma unroll
      for (int i = 0; i < NSZ; ++i) {
        const unsigned kt = (unsigned)(kb + 4 * i + q);
        const unsigned g = GM == 0 ? kt : kt / tpg;
        szr[c][i] = __builtin_amdgcn_raw_buffer_load_b32(sr, g * 64u + 4u * (unsigned)n16 + ((int)kt < ke ? 0u : 0x80000000u), (unsigned)(blk * NTW + c) * groups * 64u, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < TMAX; ++j)
#pragma unroll
      for (int c = 0; c < NTW; ++c)
        wq[c][j] = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)lane * 16u + (j < T ? 0u : 0x80000000u), ((unsigned)(blk * NTW + c) * (unsigned)KT + (unsigned)(kb + j)) * 1024u, QA_LEAN_WAUX);
    __builtin_amdgcn_sched_barrier(0);
  };
  request(nb);
  if constexpr (STAMP) ts[1] = __builtin_amdgcn_s_memrealtime();
#ifdef QA_EXP_LEAN_OVERLAP
  if (defer) {
    // the weights are on their way; x is the predecessor's output: wait for its arrival word (bounded: a launch that cannot see it goes on and computes
    // on whatever is there -- wrong numbers, never a hang), then make the rows visible (acquire at agent scope: the other XCDs' L2 lines of an x read
    // in an earlier step are stale) and ask for x
    const unsigned run = __hip_atomic_load(rest.my_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / rest.my_per_exec;
    const unsigned target = (run + 1u) * rest.wait_per_exec;
    for (int spin = 0; spin < (1 << 16); ++spin) {
      if ((int)(__hip_atomic_load(rest.wait_sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) break;
      __builtin_amdgcn_s_sleep(2);
    }
    issue_x();   // (sc1 requests: lean_dma16 below)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // x is the YOUNGEST request here: everything has landed
  } else
#endif
  // ---- 3. x has landed (it is older than every other request) ----
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LSET) : "memory");
  if constexpr (STAMP) ts[2] = __builtin_amdgcn_s_memrealtime();
  // the unit sums from the matrix core: B operand with ones in the even columns and b_k (1024 / 64 in biased8's order) in the odd ones,
  // so lane (n16, q) ends up with A (even n16) or C (odd n16) of its tokens 4q .. 4q+3
  const u32x4 bc_bits = (lane & 1) ? u32x4{0x64006400u, 0x54005400u, 0x64006400u, 0x54005400u} : u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
  const half8_t bconst = __builtin_bit_cast(half8_t, bc_bits);
  const char* xl = smem + x_off + (unsigned)min(n16, rows - 1) * pitch + (unsigned)q * 16u;  // A fragment rows: token n16 (rows past M replay the last)
  const char* gl = smem + lnw_off + (unsigned)q * 16u;
  // MR = 16: the unit sums from the matrix core (sm: lane holds A or C of its four tokens); MR = 1, 4 (that many tokens at most): the unit
  // sums as wave-uniform scalars -- computed on the compact copy (lane L = 8 consecutive k of one token), summed over the sixteen lanes of a
  // k tile and read into SGPRs.  Half the MFMAs and LDS fragment reads per tile of the MR = 16 form, which at 1..4 tokens on a wide layer
  // are what the launch runs out of first (both pipes ~50 % busy at the weights' HBM rate; with the RMSNorm prologue on fragments, over it).
  constexpr int SMR = MR == 16 ? 1 : MR;
  floatx4 sm[MR == 16 ? TMAX : 1];
  float sA[MR == 16 ? 1 : TMAX][SMR], sC[MR == 16 ? 1 : TMAX][SMR];
  floatx4 sq = floatx4{0.f, 0.f, 0.f, 0.f};
  if constexpr (MR == 16) {
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
      sm[j] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (j < T) {  // wave-uniform
        const unsigned ko = (unsigned)(kb + j) * 256u;
        half8_t xf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) xf[t] = *(const half8_t*)(xl + ko + 64 * t);
        if constexpr (LN) {
          // RMSNorm: sum of squares of the raw x as the diagonal of x x^T; x * weight in fp16 (nothing is written back: the tile loop below
          // multiplies again -- an in-place update chains every tile's reads behind the previous tile's writes)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const half8_t gf = *(const half8_t*)(gl + ko + 64 * t);
            sq = mfma16(xf[t], xf[t], sq);
            xf[t] = xf[t] * gf;
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) sm[j] = mfma16(xf[t], bconst, sm[j]);
      }
    }
    if constexpr (LN) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n16 == 4 * q + r) ((float*)(smem + ssq_off))[wave * 16 + n16] = sq[r];
    }
  } else {
    float ss[SMR];
#pragma unroll
    for (int tk = 0; tk < SMR; ++tk) ss[tk] = 0.f;
#pragma unroll
    for (int sg = 0; sg < NSZ; ++sg) {
      const bool mine = lane < 16 * (T - 4 * sg);
      const unsigned cb = (unsigned)(kb * 128 + 512 * sg) * 2u + (unsigned)lane * 16u;
      u32x4 gv = u32x4{0u, 0u, 0u, 0u};
      if constexpr (LN) gv = mine ? *(const u32x4*)(smem + lnw_off + cb) : gv;
#pragma unroll
      for (int tk = 0; tk < SMR; ++tk) {
        char* xp = smem + x_off + (unsigned)min(tk, rows - 1) * pitch + cb;  // (tokens past M: the last one again, never stored)
        u32x4 v = mine ? *(const u32x4*)xp : u32x4{0u, 0u, 0u, 0u};
        if constexpr (LN) {
          // RMSNorm: x * weight in fp16 back into the compact copy (lane-linear, before any fragment of it is read), squares summed per lane
#pragma unroll
          for (int i = 0; i < 4; ++i) ss[tk] = __builtin_amdgcn_fdot2(as_h2(v[i]), as_h2(v[i]), ss[tk], false);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = as_u32(as_h2(v[i]) * as_h2(gv[i]));
          if (mine && tk < rows) *(u32x4*)xp = v;
        }
        const half2_t one2 = {(half_t)1.f, (half_t)1.f};
        const float lo = __builtin_amdgcn_fdot2(as_h2(v[0]), one2, __builtin_amdgcn_fdot2(as_h2(v[2]), one2, 0.f, false), false);
        const float hi = __builtin_amdgcn_fdot2(as_h2(v[1]), one2, __builtin_amdgcn_fdot2(as_h2(v[3]), one2, 0.f, false), false);
        const float sa = lanes_sum<16>(lo + hi);
        const float sc = lanes_sum<16>(1024.f * lo + 64.f * hi);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          if (4 * sg + qq < TMAX) {
            sA[4 * sg + qq][tk] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sa), 16 * qq));
            sC[4 * sg + qq][tk] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), 16 * qq));
          }
      }
    }
    if constexpr (LN) {
#pragma unroll
      for (int tk = 0; tk < SMR; ++tk) {
        const float tot = wave_sum(ss[tk]);
        if (lane == 0) ((float*)(smem + ssq_off))[wave * 16 + tk] = tot;
      }
    }
  }
  if constexpr (STAMP) ts[3] = __builtin_amdgcn_s_memrealtime();

  // ---- 4. a channel block's tiles, in the order they land; 5. the waves' partials meet in LDS, NTW waves finish one channel tile each ----
  const bool odd = (lane & 1) != 0;
  (void)odd;
  const auto compute_and_finish = [&](uint32_t (&sz)[NTW][NSZ], u32x4 (&wv)[NTW][TMAX], int blk, int it) __attribute__((always_inline)) {
    // this lane's (scale, zero) words, one per tile, out of the four-tile requests
    uint32_t szj[NTW][TMAX];
#pragma unroll
    for (int c = 0; c < NTW; ++c)
#pragma unroll
      for (int j = 0; j < TMAX; ++j) szj[c][j] = (uint32_t)__builtin_amdgcn_ds_bpermute(((j & 3) * 16 + n16) * 4, (int)sz[c][j >> 2]);
    floatx4 acc[NTW];
#pragma unroll
    for (int c = 0; c < NTW; ++c) acc[c] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
      if (j < T) {  // wave-uniform
        const unsigned ko = (unsigned)(kb + j) * 256u;
        half8_t xf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) xf[t] = *(const half8_t*)(xl + ko + 64 * t);
        if constexpr (LN && MR == 16) {
#pragma unroll
          for (int t = 0; t < 4; ++t) xf[t] = xf[t] * *(const half8_t*)(gl + ko + 64 * t);
        }
        floatx4 xa, nc;
        if constexpr (MR == 16) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float mine = sm[j][r];
            const float other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine), 0xB1, 0xf, 0xf, false));
            xa[r] = odd ? other : mine;
            nc[r] = -(odd ? mine : other);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {  // (lanes q = 0 hold tokens 0..3; the other rows are never stored)
            xa[r] = sA[j][r % SMR];
            nc[r] = -sC[j][r % SMR];
          }
        }
        if constexpr (STAMP) {
          if (j == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NTW * TMAX - 1) : "memory");
            ts[4] = __builtin_amdgcn_s_memrealtime();
          }
          if (j == T - 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ts[5] = __builtin_amdgcn_s_memrealtime();
          }
        }
#pragma unroll
        for (int c = 0; c < NTW; ++c) {
          const u32x4 w = wv[c][j];
          floatx4 g = nc;
#pragma unroll
          for (int t = 0; t < 4; ++t) g = mfma16(xf[t], biased8(w[t]), g);
          const GroupRaw raw{szj[c][j]};
          const float s = group_scale_f32(raw), z = group_zero_f32(raw);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[c][r] = __builtin_fmaf(s, __builtin_fmaf(-z, xa[r], g[r]), acc[c][r]);
        }
      }
    }
    if constexpr (STAMP) ts[6] = __builtin_amdgcn_s_memrealtime();

    // PERSIST: two reduction buffers alternate (a wave that runs ahead writes the other one; one barrier per block is enough), and the
    // finishing waves rotate, so that nobody is late at every barrier
    floatx4* red = (floatx4*)smem + (PERSIST ? (it & 1) * (WAVES * NTW * 64) : 0);
#pragma unroll
    for (int c = 0; c < NTW; ++c) red[(wave * NTW + c) * 64 + lane] = acc[c];
    if constexpr (STAMP) ts[7] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if constexpr (STAMP) ts[8] = __builtin_amdgcn_s_memrealtime();
    const int first = PERSIST ? (it % (WAVES / NTW)) * NTW : 0;
    if (wave >= first && wave < first + NTW) {
      const int ct = wave - first;
      floatx4 sum = red[ct * 64 + lane];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) sum += red[(w * NTW + ct) * 64 + lane];
      // lane (n16, q) holds tokens 4q .. 4q+3 of channel n16
      if constexpr (LN) {
        const float* sqp = (const float*)(smem + ssq_off);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float ss = 0.f;
#pragma unroll
          for (int w = 0; w < WAVES; ++w) ss += sqp[w * 16 + 4 * q + r];
          sum[r] *= rsqrtf(ss / (float)aK + rest.ln_eps);
        }
      }
      const int nt = blk * NTW + ct, n = nt * 16 + n16;
      if (rest.silu_mul) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float up = __shfl_xor(sum[r], 8);  // channels 0..7 gate, 8..15 up
          const int m = row0 + 4 * q + r;
          if (4 * q + r < rows && n16 < 8) rest.Y[(size_t)m * (aN >> 1) + nt * 8 + n16] = silu_mul_f16((half_t)sum[r], (half_t)up);
        }
      } else {
        const float bv = rest.bias ? (float)rest.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = row0 + 4 * q + r;
          if (4 * q + r < rows) {
            float v = sum[r] + bv;
            if (rest.residual) v += (float)rest.residual[(size_t)m * aN + n];
#ifdef QA_EXP_LEAN_OVERLAP
            if (rest.signal != nullptr) {   // rows written THROUGH to memory (sc1): the successor on another XCD asks for them with sc1 loads, no cache-wide fence on either side
              const half_t hv = (half_t)v;
              asm volatile("global_store_short %0, %1, off sc0 sc1" ::"v"(rest.Y + (size_t)m * aN + n), "v"((unsigned)__builtin_bit_cast(unsigned short, hv)) : "memory");
            } else
#endif
            rest.Y[(size_t)m * aN + n] = (half_t)v;
          }
        }
      }
    }
  };
  if constexpr (PERSIST) {
    int it = 0;
    for (int blk = nb; blk < nblocks; blk += gx) {
      if (it > 0) request(blk);
      compute_and_finish(szr, wq, blk, it++);
    }
  } else {
    compute_and_finish(szr, wq, nb, 0);
  }
#ifdef QA_EXP_LEAN_OVERLAP
  if (rest.signal != nullptr || rest.my_cnt != nullptr) {
    if (wave < NTW) {   // (the waves that stored rows: one-block launches)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the written-through rows are acknowledged: in memory before the word
      if (lane == 0 && rest.signal != nullptr) __hip_atomic_fetch_add(rest.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0 && rest.my_cnt != nullptr) __hip_atomic_fetch_add(rest.my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
  if constexpr (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[9] = __builtin_amdgcn_s_memrealtime();
    if (rest.dbg != nullptr) {
      const unsigned wid = ((blockIdx.y * (unsigned)gx + blockIdx.x) * WAVES + (unsigned)wave) & 4095u;
      if (lane < kLeanStamps) {
        unsigned long long mine = 0;
#pragma unroll
        for (int i = 0; i < kLeanStamps; ++i) mine = lane == i ? ts[i] : mine;
        rest.dbg[wid * kLeanStamps + lane] = mine;
      }
    }
  }
  if constexpr (SPAN) {
    if (rest.span != nullptr) rest.span[kSpanWaves + span_slot] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace quick_amd
