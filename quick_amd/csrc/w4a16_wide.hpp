// Large-M W4A16 kernels for MI355X (gfx950): v_mfma_f32_32x32x16_f16, ONE wave per SIMD (accumulators in AGPRs),
// operands brought on chip by LDS-DMA.
//
// Why these kernels (r01's w4a16_tiled_kernel stays for small token counts): the cost of this GEMM on the matrix core's
// side is fixed, the cost on the VALU side -- 13 packed-f16 ops per packed dword -- is paid once per (weight, wave), so
// VALU ops per MFMA fall with the number of TOKENS a wave owns.  A 16-cycle 16x16x32 MFMA hides no VALU work of its own
// wave (tools/mfma_valu_overlap.hip); a 32-cycle 32x32x16 hides ~5 issue slots (MI355X_MICROARCH.md, "one wave per
// SIMD").  So: 4 waves per workgroup, all along N, each wave owning ALL the workgroup's MB*32 tokens x PAIRS*32 channels;
// 13 / MB VALU ops per MFMA, no weight dequantised twice in a workgroup, MB*PAIRS*16 accumulator registers.
//
// Operands of v_mfma_f32_32x32x16_f16 (A = 32 channels x 16 k, B = 16 k x 32 tokens, lane l = (rho = l % 32, h = l / 32)):
//   A: lane holds channel rho, k = 8 h .. 8 h + 7 of the k16 step.  The HBM weight layout is unchanged ("mi355x order":
//      a 16-channel x 128-k tile is 1 KiB, byte 16 * (c + 16 q) holds the dwords t = 0..3 of channel c, k = 32 t + 8 q + j):
//      lane (rho, h) takes 16 bytes at [tile rho / 16][c = rho % 16, q = h] ("lo") and at q = 2 + h ("hi", +512 B); dword t of
//      lo / hi is the A operand of k16 step 2 t / 2 t + 1.  No lane shuffles; every 16-lane group reads 256 contiguous bytes.
//   B: lane holds token rho, k = 8 h .. + 7.  A stage (128 k) of the token tile lives in LDS ROW-MAJOR, 256 B per token,
//      16-byte chunk c of row r stored at chunk c ^ (r % 16): ds_read_b128 of a fragment is conflict-free, and the image is
//      filled by buffer_load_dwordx4 ... lds (LDS-DMA: no staging registers, no ds_write) in full 256-byte row segments,
//      the swizzle applied to the per-lane SOURCE address (the LDS side of an LDS-DMA is lane-linear).
//   C/D: lane holds token rho and channels (r % 4) + 8 (r / 4) + 4 h, r = 0..15, of the 32-channel pair.
//
// Two pipelines around the same compute core:
//   w4a16_wide_kernel  (256-token tiles: a stage is 2-4 k cycles of MFMA, longer than any load): two LDS stage buffers for
//       x, the weights HBM -> VGPR one stage ahead, vmcnt(0) + one barrier per stage.
//   w4a16_ring_kernel  (64- and 128-token tiles: a stage is SHORTER than a trip to L2 / HBM): x, the packed weights and the
//       (scale, zero) words all travel by LDS-DMA into a ring of NBUF stage slots, NBUF - 2 stages in flight behind a
//       COUNTED vmcnt; a wave picks its own packed weights out of the slot again with ds_read_b128 one stage early.  The
//       weights still go to the matrix core straight from registers after dequantisation -- what passes through LDS is the
//       4-bit stream, as a prefetch queue that costs no registers.
// All LDS-DMA is inline asm on purpose: hipcc makes every ds_read wait for every LDS-DMA it knows about, and beside them
// it turns its counted waits for ordinary loads into drains (cdna_hip_programming.md, "Three .s-level traps").
#pragma once

namespace quick_amd {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// 64 lanes x 16 (4) bytes: global (descriptor base + voff + soff) -> LDS (lds_addr + 16 (4) * lane).  M0 carries the LDS
// address and is written in the statement that uses it (hipcc reserves M0 and does not preserve it across asm).
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc),
               "s"(soff)
               : "memory");
}
__device__ __forceinline__ void lds_dma4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc),
               "s"(soff)
               : "memory");
}

template <int PAIRS, int GM>
struct WideW {  // one 128-k tile of this wave's weights: per 32-channel pair the lo / hi dwordx4 and the raw (scale, zero) words
  u32x4 lo[PAIRS], hi[PAIRS];
  uint32_t sz[PAIRS][groups_per_tile<GM>()];
};

template <int NX>
struct WideBufs {
  __amdgpu_buffer_rsrc_t x, w, s;
  unsigned x_voff[NX];  // per LDS-DMA instruction of a stage: clamped row * K * 2 + swizzled chunk * 16
  unsigned w_voff, s_voff;
  unsigned w_pstride;  // bytes between consecutive 32-channel pairs of the weights
  unsigned s_pstride;  // ... of the (scale, zero point) words
};

// Descriptors and per-lane offsets.  x: with NW = 4 WK waves, instruction i of a wave moves rows 4 NW i + 4 wave + lane / 16
// of the token tile, lane % 16 picks the 16-byte chunk (the row's chunk c lands at LDS chunk c ^ (row % 16), so the lane
// writing LDS chunk lane % 16 fetches source chunk (lane % 16) ^ (row % 16)).  Rows past M replay row M - 1 (never stored).
// WK = 2 (eight waves, the two halves of a workgroup split the k16 steps of every stage by parity): wave (wn, wk) takes only
// the "lo" (wk = 0) or "hi" (wk = 1) 16 bytes of its weights -- w_voff points at its half.
template <int MB, int PAIRS, int WK = 1>
__device__ __forceinline__ WideBufs<MB * 2 / WK> wide_bufs(const GemmArgs& a, int m0, int ct0, int lane, int wave) {
  constexpr int NW = 4 * WK, NX = MB * 2 / WK;
  WideBufs<NX> b;
  const int KT = a.K >> 7, NGRP = a.K / a.G;
  const unsigned rho = (unsigned)lane & 31u, h = (unsigned)lane >> 5;
  const unsigned xrow = 4u * (unsigned)wave + ((unsigned)lane >> 4);
  b.x = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (unsigned)a.M * (unsigned)a.K * 2u, 0x00020000);
#pragma unroll
  for (int i = 0; i < NX; ++i)
    b.x_voff[i] = (unsigned)min(m0 + 4 * NW * i + (int)xrow, a.M - 1) * (unsigned)a.K * 2u + 16u * (((unsigned)lane & 15u) ^ (xrow & 15u));
  b.w = __builtin_amdgcn_make_buffer_rsrc((void*)(a.QW + (size_t)ct0 * KT * 64), 0, (unsigned)(2 * PAIRS) * (unsigned)KT * 1024u, 0x00020000);
  b.w_voff = (rho >> 4) * (unsigned)KT * 1024u + 16u * ((rho & 15u) + 16u * h) + (WK == 2 ? ((unsigned)wave >> 2) * 512u : 0u);
  b.w_pstride = 2u * (unsigned)KT * 1024u;
  b.s = __builtin_amdgcn_make_buffer_rsrc((void*)((const uint32_t*)a.S + (size_t)ct0 * NGRP * 16), 0, (unsigned)(2 * PAIRS) * (unsigned)NGRP * 64u, 0x00020000);
  b.s_voff = (rho >> 4) * (unsigned)NGRP * 64u + 4u * (rho & 15u);
  b.s_pstride = 2u * (unsigned)NGRP * 64u;
  return b;
}

// dequant8 (w4a16_common.hpp) without inline asm: the masks live in SGPRs and the magic number in a VGPR the compiler
// cannot see through, so (q & mask) | magic selects v_and_or_b32 on its own -- no asm statement boundaries, hence none of
// the s_nop hipcc pads them with (56 per 128 MFMAs in the first build of this kernel) and free scheduling.
struct DqConsts {
  uint32_t mlo, mhi, magic;
};
__device__ __forceinline__ DqConsts make_dq_consts() {
  DqConsts d{0x000f000fu, 0x00f000f0u, 0x64006400u};
  asm volatile("" : "+s"(d.mlo), "+s"(d.mhi));
  asm volatile("" : "+v"(d.magic));
  return d;
}
__device__ __forceinline__ half8_t dequant8(uint32_t q, const GroupQ& g, const DqConsts& d) {
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const uint32_t q8 = q >> 8;
  const half2_t h0 = (as_h2((q & d.mlo) | d.magic) + g.nzlo) * g.s2;               // k0, k1
  const half2_t h1 = (as_h2((q & d.mhi) | d.magic) * sixteenth + g.nzhi) * g.s2;   // k2, k3
  const half2_t h2 = (as_h2((q8 & d.mlo) | d.magic) + g.nzlo) * g.s2;              // k4, k5
  const half2_t h3 = (as_h2((q8 & d.mhi) | d.magic) * sixteenth + g.nzhi) * g.s2;  // k6, k7
  half8_t r;
  r[0] = h0[0]; r[1] = h0[1]; r[2] = h1[0]; r[3] = h1[1];
  r[4] = h2[0]; r[5] = h2[1]; r[6] = h3[0]; r[7] = h3[1];
  return r;
}

// One stage = 8 k16 steps x PAIRS "units"; unit u = (k16 step kk, pair p) is MB MFMAs sharing one dequantised A fragment.
// Software pipeline: while the MFMAs of unit u run, the VALU dequantises the fragment of unit u + 1 and (p == 0) the LDS
// returns the B fragments of step kk + 1.  The pipeline runs ACROSS the stage boundary: the last unit of a stage prepares
// the first unit of the next one -- group constants and A fragment from the next stage's weights (already in registers),
// and, where the caller guarantees that the next stage's tokens have landed (PRE_B: the ring kernel), its B fragments --
// into `carry`, so that the first MFMA of a stage issues right behind the barrier.  sched_group_barrier spells the
// interleave out -- one MFMA, then its share of the VALU ops and one ds_read -- because hipcc otherwise emits "13 VALU,
// then MB MFMAs back to back": an in-order wave issues the VALU block only after the last MFMA has issued, i.e. in the open.
// `hook(u)` runs at the head of unit u: the callers spread their LDS-DMA issue (and the ring kernel its early weight
// read) over the units with it -- a wave that issues a stage's 16 KiB of LDS-DMA in one go keeps the CU's vector memory
// path busy for ~1000 cycles during which no wave issues an MFMA.
// ABL (timing experiments only, results are wrong): 1 = no compute (hooks only), 2 = the callers issue no loads in the K loop,
// 8 = no dequantisation (the packed dword goes to the matrix core as it is), 16 = no B-fragment reads.
// B fragments are requested DEPTH k16 steps ahead of their MFMAs: a step is MB * PAIRS MFMAs = 32 * MB * PAIRS cycles, an
// LDS round trip under load 130-200, so short steps need a longer lead (first build: lead 1 everywhere -- at MB = 2 every
// MFMA waited ~100 cycles for a fragment requested eight instructions earlier).
template <int MB, int PAIRS>
constexpr int wide_bdepth() { return MB * PAIRS >= 8 ? 1 : (MB * PAIRS >= 4 ? 2 : 3); }

template <int MB, int PAIRS, int GM>
struct WideCarry {
  half8_t af;
  half8_t bf[wide_bdepth<MB, PAIRS>()][MB];
  GroupQ grp[PAIRS][groups_per_tile<GM>()];
};

// fragments of this wave's step i (WK = 2: the wave owns k16 steps 2 i + wk, and xb already carries the ^ (wk << 5))
template <int MB, int WK = 1, int ABL = 0>
__device__ __forceinline__ void wide_read_frags(unsigned xb, int i, half8_t (&f)[MB]) {
  typedef const __attribute__((address_space(3))) char* lds_ptr;
  if constexpr (ABL & 16) {  // timing experiment: no LDS reads, the fragments are whatever the registers hold
#pragma unroll
    for (int mt = 0; mt < MB; ++mt) asm volatile("" : "=v"(f[mt]));
    return;
  }
  const lds_ptr xk = (lds_ptr)(uintptr_t)(xb ^ (unsigned)(i << (WK == 2 ? 6 : 5)));  // chunk (2 kk + h) ^ (rho % 16) of this lane's row
#pragma unroll
  for (int mt = 0; mt < MB; ++mt) f[mt] = *(const __attribute__((address_space(3))) half8_t*)(xk + mt * 8192);
}
template <int PAIRS, int GM, int WK = 1>
__device__ __forceinline__ half8_t wide_frag(const WideW<PAIRS, GM>& w, const GroupQ (&grp)[PAIRS][groups_per_tile<GM>()], int u,
                                             const DqConsts& dq) {
  const int i = u / PAIRS, p = u % PAIRS;
  const int t = WK == 2 ? i : (i >> 1);  // the k32 step, i.e. the dword of the lane's 16 bytes
  const uint32_t q = WK == 2 ? w.lo[p][t] : ((i & 1) ? w.hi[p][t] : w.lo[p][t]);
  return dequant8(q, grp[p][group_slot<GM>(t)], dq);
}
// what the first unit of a stage needs, from that stage's weights (and, PRE_B, its landed tokens at LDS address xb)
template <int MB, int PAIRS, int GM, bool PRE_B, int WK = 1>
__device__ __forceinline__ void wide_prepare(WideCarry<MB, PAIRS, GM>& c, const WideW<PAIRS, GM>& w, unsigned xb, const DqConsts& dq) {
  constexpr int NG = groups_per_tile<GM>();
#pragma unroll
  for (int p = 0; p < PAIRS; ++p)
#pragma unroll
    for (int i = 0; i < NG; ++i) c.grp[p][i] = make_group(GroupRaw{w.sz[p][i]});
  c.af = wide_frag<PAIRS, GM, WK>(w, c.grp, 0, dq);
  if constexpr (PRE_B) {
#pragma unroll
    for (int d = 0; d < wide_bdepth<MB, PAIRS>(); ++d) wide_read_frags<MB, WK>(xb, d, c.bf[d]);
  }
}

template <int MB, int PAIRS, int GM, int ABL = 0, bool PRE_B = false, int WK = 1, class Hook>
__device__ __forceinline__ void wide_compute(const WideW<PAIRS, GM>& w, const WideW<PAIRS, GM>& wnext, unsigned xb, unsigned xb_next,
                                             const DqConsts& dq, floatx16 (&acc)[PAIRS][MB], WideCarry<MB, PAIRS, GM>& carry,
                                             Hook&& hook) {
  constexpr int NG = groups_per_tile<GM>();
  constexpr int NSTEP = 8 / WK;  // k16 steps of the stage this wave computes
  constexpr int NU = NSTEP * PAIRS;
  if constexpr (ABL & 1) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      hook(u);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" ::"v"(w.lo[0]), "v"(w.hi[0]), "v"(w.sz[0][0]), "v"(wnext.lo[0]));
    return;
  }
  constexpr int VPM = (14 + MB - 1) / MB;  // VALU ops placed behind each MFMA (13 per fragment + the address xor)
  GroupQ grp[PAIRS][NG];
#pragma unroll
  for (int p = 0; p < PAIRS; ++p)
#pragma unroll
    for (int i = 0; i < NG; ++i) grp[p][i] = carry.grp[p][i];
  constexpr int DEPTH = wide_bdepth<MB, PAIRS>(), NBF = DEPTH + 1;
  static_assert(DEPTH < NSTEP, "fragment lead longer than the stage");
  half8_t bf[NBF][MB], af[2];
  af[0] = carry.af;
  if constexpr (PRE_B) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int mt = 0; mt < MB; ++mt) bf[d][mt] = carry.bf[d][mt];
  } else {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) wide_read_frags<MB, WK, ABL>(xb, d, bf[d]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int kk = u / PAIRS, p = u % PAIRS;
    const bool last = u + 1 == NU;
    const int ahead = kk + DEPTH;  // the step whose fragments this unit requests (p == 0 units only)
    const bool reads = p == 0 && (ahead < NSTEP || PRE_B);
    hook(u);
    __builtin_amdgcn_sched_barrier(0);
    if (reads) {
      if (ahead < NSTEP) wide_read_frags<MB, WK, ABL>(xb, ahead, bf[ahead % NBF]);
      else wide_read_frags<MB, WK, ABL>(xb_next, ahead - NSTEP, carry.bf[ahead - NSTEP]);  // the next stage's first steps (PRE_B: landed)
    }
    if constexpr (ABL & 8) {  // timing experiment: no dequantisation
      const uint32_t raw = w.lo[0][(u + 1) & 3];
      af[(u + 1) & 1] = __builtin_bit_cast(half8_t, u32x4{raw, raw, raw, raw});
      if (last) carry.af = af[(u + 1) & 1];
    } else if (!last) {
      af[(u + 1) & 1] = wide_frag<PAIRS, GM, WK>(w, grp, u + 1, dq);
    } else {  // the next stage's group constants and first A fragment
#pragma unroll
      for (int pp = 0; pp < PAIRS; ++pp)
#pragma unroll
        for (int i = 0; i < NG; ++i) carry.grp[pp][i] = make_group(GroupRaw{wnext.sz[pp][i]});
      carry.af = wide_frag<PAIRS, GM, WK>(wnext, carry.grp, 0, dq);
    }
#pragma unroll
    for (int mt = 0; mt < MB; ++mt) acc[p][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u & 1], bf[kk % NBF][mt], acc[p][mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MB; ++mt) {
      if (reads && !(ABL & 16)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // MFMA
      if (last) __builtin_amdgcn_sched_group_barrier(0x002, VPM + (4 * PAIRS * NG + MB - 1) / MB, 0);  // VALU (+ the group constants)
      else __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// tile / K-slice coordinates shared by the two kernels
struct WideTile {
  int mb, nb, ks, kt_lo, kt_hi, nstage, m0;
};
template <int MB, int PAIRS>
__device__ __forceinline__ WideTile wide_tile(const GemmArgs& a) {
  WideTile t;
  const int NB = a.N / (PAIRS * 128);
  t.nb = blockIdx.x % NB;
  t.mb = blockIdx.x / NB;
  if (a.xcd_gm > 0) {  // see w4a16_tiled_kernel: every XCD gets a compact rectangle of tiles
    const int MBk = gridDim.x / NB, gn = 8 / a.xcd_gm;
    const int mcnt = MBk / a.xcd_gm, ncnt = NB / gn;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t.mb = (xcd / gn) * mcnt + idx / ncnt;
    t.nb = (xcd % gn) * ncnt + idx % ncnt;
  }
  t.ks = blockIdx.y;
  const int KT = a.K >> 7;
  t.kt_lo = t.ks * a.kt_per_split;
  t.kt_hi = min(KT, t.kt_lo + a.kt_per_split);
  t.nstage = t.kt_hi - t.kt_lo;
  t.m0 = t.mb * MB * 32;
  return t;
}

template <int MB, int PAIRS>
__device__ __forceinline__ void wide_zero(floatx16 (&acc)[PAIRS][MB]) {
#pragma unroll
  for (int p = 0; p < PAIRS; ++p)
#pragma unroll
    for (int mt = 0; mt < MB; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][mt][r] = 0.f;
}

// The tile's way out: accumulators (+ bias, or SiLU(gate) * up) -> f16 image of the tile in LDS -> whole rows to y (+ residual).
// Image: row m of the tile, 16-byte chunk q (8 channels) at byte (m * CPR + (q ^ (m % 8))) * 16 -- a lane group of the
// ds_write_b64 is 16 rows of one chunk column, the XOR spreads them over the eight chunks of a 128-byte bank row (2-way
// instead of 16-way); the read-back of a wave covers 64 / CPR rows that share m % 8, so the XOR is one constant for it.
// The residual is added to the ROUNDED product, which is what the unfused sequence (GEMM, then y + residual) computes.
// SiLU * mul: gate / up interleaved by 8 channels (c = 0, 2 gate of the two 16-channel tiles, c = 1, 3 their up), the
// image and y have half the channels.  All 4 * WK waves store; waves with active == false hold no accumulators.
template <int MB, int PAIRS, int WK, bool SILU>
__device__ __forceinline__ void wide_store_tile(const GemmArgs& a, const WideTile& t, floatx16 (&acc)[PAIRS][MB], char* smem, int lane,
                                                int wave, bool active) {
  constexpr int CPR = SILU ? PAIRS * 8 : PAIRS * 16;  // 16-byte chunks per tile row
  constexpr int R = 64 / CPR;                         // rows per wave-instruction of the read-back
  constexpr int ROWS = MB * 32, NWV = 4 * WK, NIT = ROWS / R / NWV;
  static_assert(ROWS % (8 * R) == 0 && ROWS % (R * NWV) == 0, "tile rows vs the read-back pattern");
  const int rho = lane & 31, h = lane >> 5;
  if (active) {
    const unsigned r7 = (unsigned)rho & 7u;
    char* wrow = smem + rho * (CPR * 16) + h * 8;
    const unsigned q0 = (unsigned)wave * (SILU ? PAIRS * 2 : PAIRS * 4);
#pragma unroll
    for (int p = 0; p < PAIRS; ++p)
#pragma unroll
      for (int c = 0; c < 4; c += SILU ? 2 : 1) {
        const unsigned q = q0 + (SILU ? p * 2 + c / 2 : p * 4 + c);
        char* wp = wrow + ((q ^ r7) << 4);
        half4_t bv = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        if constexpr (!SILU)
          if (a.bias) bv = *(const half4_t*)(a.bias + (t.nb * 4 + wave) * PAIRS * 32 + 32 * p + 8 * c + 4 * h);
#pragma unroll
        for (int mt = 0; mt < MB; ++mt) {
          half4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (SILU) o[r] = silu_mul_f16((half_t)acc[p][mt][4 * c + r], (half_t)acc[p][mt][4 * c + 4 + r]);
            else o[r] = (half_t)(acc[p][mt][4 * c + r] + (float)bv[r]);
          }
          *(half4_t*)(wp + mt * 32 * (CPR * 16)) = o;
        }
      }
  }
  __syncthreads();
  const int wv = uniform((int)(threadIdx.x >> 6));
  const int ldy = SILU ? a.N >> 1 : a.N;
  const int q = lane % CPR, k = lane / CPR;
  half_t* ycol = a.Y + (SILU ? t.nb * PAIRS * 64 : t.nb * PAIRS * 128) + q * 8;
  const half_t* rcol = (!SILU && a.residual) ? a.residual + t.nb * PAIRS * 128 + q * 8 : nullptr;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int base = it * NWV + wv;
    const int row = (base >> 3) * (8 * R) + (base & 7) + 8 * k;
    half8_t v = *(const half8_t*)(smem + (row * CPR + (q ^ (base & 7))) * 16);
    const int m = t.m0 + row;
    if (m < a.M) {
      if (rcol) {
        const half8_t res = *(const half8_t*)(rcol + (size_t)m * a.N);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (half_t)((float)v[r] + (float)res[r]);
      }
      *(half8_t*)(ycol + (size_t)m * ldy) = v;
    }
  }
}

// K split across workgroups (slab = [(p, mt, c)][wave][lane] floatx4) and the fused epilogue.
// `wave` = the wave's index along N (0..3); waves with active == false (the second K half of an eight-wave workgroup, whose
// partial was already added in) only take part in the workgroup barriers.
template <int MB, int PAIRS, int WK>
__device__ __forceinline__ void wide_epilogue(const GemmArgs& a, const WideTile& t, floatx16 (&acc)[PAIRS][MB], char* smem, int lane, int wave,
                                              bool active) {
  // The accumulators hold token rho, channels 8 c + 4 h .. + 3 per lane: stored from here, a wave's store instruction
  // touches 32 rows x 16 bytes.  So the tile goes through LDS (free now) and leaves in whole rows, 16 bytes per lane.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (ring kernels: the replayed LDS-DMA of the last stages has landed)
  __builtin_amdgcn_s_barrier();                     // everybody is done with the ring / the split-K flag
  if (a.silu_mul) wide_store_tile<MB, PAIRS, WK, true>(a, t, acc, smem, lane, wave, active);
  else wide_store_tile<MB, PAIRS, WK, false>(a, t, acc, smem, lane, wave, active);
}

template <int MB, int PAIRS, int WK = 1>
__device__ __forceinline__ void wide_finish(const GemmArgs& a, const WideTile& t, floatx16 (&acc)[PAIRS][MB], char* smem, int ct0,
                                            int lane, int wave, bool active = true) {
  // Two separate ways out on purpose.  With one epilogue behind "if (ksplit > 1) acc = sum of the slabs", the accumulators
  // that arrive from the K loop (AGPRs) and the sums (VGPRs) meet in one set of registers, and hipcc makes that set the
  // scratch memory: 52 scratch_store_dwordx4 + 52 loads per lane on the path that splits nothing -- 14-16 us per 256 x 256
  // tile, 20-30 us of the 133 us 4096^3 launch [r02, tools/wide_phases.py].
  if (a.ksplit > 1) {
    constexpr unsigned SLAB_BYTES = PAIRS * MB * 16384;
    const __amdgpu_buffer_rsrc_t rs = slab_rsrc(a.slabs + (size_t)blockIdx.x * a.ksplit * (SLAB_BYTES / 4), a.ksplit * SLAB_BYTES);
    const unsigned my = ((unsigned)wave * 64u + (unsigned)lane) * 16u;
    if (active) {
#pragma unroll
      for (int p = 0; p < PAIRS; ++p)
#pragma unroll
        for (int mt = 0; mt < MB; ++mt)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            slab_store(rs, t.ks * SLAB_BYTES + ((p * MB + mt) * 4 + c) * 4096 + my,
                       floatx4{acc[p][mt][4 * c], acc[p][mt][4 * c + 1], acc[p][mt][4 * c + 2], acc[p][mt][4 * c + 3]});
    }
    if (!splitk_arrive(a.counters + blockIdx.x, a.ksplit, (unsigned*)smem)) return;
    // every slice is read back from its slab (the own one too) and added in index order: the sum does not depend on
    // who arrived last.  The slabs were written through by other CUs, so every load is a trip to memory (~0.65 us): the loads
    // of D slices are issued together (as many as 128 registers hold), the adds stay in slice order.
    constexpr int NL = PAIRS * MB * 4, D = NL <= 8 ? 4 : (NL <= 16 ? 2 : 1);
    floatx16 sum[PAIRS][MB];
    wide_zero<MB, PAIRS>(sum);
    if (NL <= 32 && a.ksplit == 2) {  // two slices: a + b == b + a exactly -- the own partial stays in registers, one slab is read
      if (active) {
        const unsigned base = (unsigned)(1 - t.ks) * SLAB_BYTES + my;
        floatx4 part[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) part[i] = slab_load(rs, base + i * 4096);
#pragma unroll
        for (int p = 0; p < PAIRS; ++p)
#pragma unroll
          for (int mt = 0; mt < MB; ++mt)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int r = 0; r < 4; ++r) sum[p][mt][4 * c + r] = acc[p][mt][4 * c + r] + part[(p * MB + mt) * 4 + c][r];
      }
      wide_epilogue<MB, PAIRS, WK>(a, t, sum, smem, lane, wave, active);
      return;
    }
    for (int o = 0; active && o < a.ksplit; o += D) {
      floatx4 part[D][NL];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const unsigned base = (unsigned)min(o + d, a.ksplit - 1) * SLAB_BYTES + my;  // (past the end: a load nobody adds)
#pragma unroll
        for (int i = 0; i < NL; ++i) part[d][i] = slab_load(rs, base + i * 4096);
      }
#pragma unroll
      for (int d = 0; d < D; ++d)
        if (o + d < a.ksplit) {
#pragma unroll
          for (int p = 0; p < PAIRS; ++p)
#pragma unroll
            for (int mt = 0; mt < MB; ++mt)
#pragma unroll
              for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[p][mt][4 * c + r] += part[d][(p * MB + mt) * 4 + c][r];
        }
    }
    wide_epilogue<MB, PAIRS, WK>(a, t, sum, smem, lane, wave, active);
    return;
  }
  wide_epilogue<MB, PAIRS, WK>(a, t, acc, smem, lane, wave, active);
}

// ------------------------------------------------------------------------------------------------
// 256-token tiles: x double-buffered in LDS, weights HBM -> VGPR one stage ahead
// ------------------------------------------------------------------------------------------------
template <int MB, int PAIRS, int GM>
__device__ __forceinline__ void wide_load_w(WideW<PAIRS, GM>& w, const WideBufs<MB * 2>& b, int kt, const GemmArgs& a) {
  constexpr int NG = groups_per_tile<GM>();
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const unsigned so = (unsigned)p * b.w_pstride + (unsigned)kt * 1024u;
    w.lo[p] = __builtin_amdgcn_raw_buffer_load_b128(b.w, b.w_voff, so, 0);
    w.hi[p] = __builtin_amdgcn_raw_buffer_load_b128(b.w, b.w_voff + 512u, so, 0);
  }
#pragma unroll
  for (int p = 0; p < PAIRS; ++p)
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const unsigned g = (unsigned)group_index<GM>(kt, i * (4 / NG), a.tpg, a.G);
      w.sz[p][i] = __builtin_amdgcn_raw_buffer_load_b32(b.s, b.s_voff, (unsigned)p * b.s_pstride + g * 64u, 0);
    }
}

// Workgroup tile (MB*32 tokens) x (PAIRS*128 channels), 4 waves along N.  Grid: x = tiles (XCD-aware order), y = K slices.
template <int MB, int PAIRS, int GM, int ABL = 0>
__global__ __launch_bounds__(256) void w4a16_wide_kernel(const GemmArgs a) {
  constexpr int STAGE_BYTES = MB * 32 * 256;
  constexpr int XI = MB * 2, NU = 8 * PAIRS;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 * STAGE_BYTES

  unsigned long long ph[5];  // ABL bit 64: s_memrealtime stamps (100 MHz) at the phase boundaries, per wave, into a.dbg
  if constexpr (ABL & 64) ph[0] = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int rho = lane & 31, h = lane >> 5;
  const WideTile t = wide_tile<MB, PAIRS>(a);
  const int ct0 = (t.nb * 4 + wave) * PAIRS * 2;  // first 16-channel tile of this wave
  const WideBufs<MB * 2> b = wide_bufs<MB, PAIRS>(a, t.m0, ct0, lane, wave);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lds0 = lds_base + (unsigned)wave * 1024u;  // LDS-DMA destination of this wave's first instruction: rows 4 w .. 4 w + 3
  // B-fragment read address of this lane in stage buffer 0, token tile 0, k16 step 0: row rho, chunk h ^ (rho % 16)
  const unsigned xrd = lds_base + (unsigned)rho * 256u + (unsigned)((h ^ (rho & 15)) << 4);

  floatx16 acc[PAIRS][MB];
  wide_zero<MB, PAIRS>(acc);

  // Straight-line loop body on purpose: with control flow around the MFMAs hipcc carries the accumulators through the
  // loop in VGPRs and copies them to AGPRs and back every iteration (and then spills).  So the weight "double buffer" is a
  // register copy at the end of the stage (18 moves per 128-k stage), and the last iteration issues its prefetch anyway --
  // a replay of the last stage into the other LDS buffer, which nobody reads.
  const DqConsts dq = make_dq_consts();
  WideW<PAIRS, GM> wc, wn;
  if (t.nstage > 0) {
#pragma unroll
    for (int i = 0; i < XI; ++i) lds_dma16(b.x, b.x_voff[i], (unsigned)t.kt_lo * 256u, lds0 + i * 4096);
    wide_load_w<MB, PAIRS, GM>(wc, b, t.kt_lo, a);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  WideCarry<MB, PAIRS, GM> carry;
  wide_prepare<MB, PAIRS, GM, false>(carry, wc, 0u, dq);
  if constexpr (ABL & 64) ph[1] = __builtin_amdgcn_s_memrealtime();

  for (int s = 0; s < t.nstage; ++s) {
    const int ktn = min(t.kt_lo + s + 1, t.kt_hi - 1);
    const unsigned par = (unsigned)(s & 1);
    const unsigned dst = lds0 + (par ^ 1u) * STAGE_BYTES;
    if constexpr (!(ABL & 2)) wide_load_w<MB, PAIRS, GM>(wn, b, ktn, a);
    else wn = wc;
    __builtin_amdgcn_sched_barrier(0);
    wide_compute<MB, PAIRS, GM, ABL, false>(wc, wn, xrd + par * STAGE_BYTES, 0u, dq, acc, carry, [&](int u) {
      // this unit's share of the next stage's LDS-DMA (all of it issued two units before the stage ends)
      constexpr int PER = (XI + NU - 3) / (NU - 2);
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int i = u * PER + j;
        if constexpr (!(ABL & 2))
          if (i < XI) lds_dma16(b.x, b.x_voff[i], (unsigned)ktn * 256u, dst + i * 4096);
      }
    });
    __builtin_amdgcn_s_waitcnt(0x0F70);  // stage s + 1 has landed (LDS-DMA and weights) ...
    __builtin_amdgcn_s_barrier();        // ... in every wave, and everybody is done reading stage s
    wc = wn;
  }
  if constexpr (ABL & 64) ph[2] = __builtin_amdgcn_s_memrealtime();
  wide_finish<MB, PAIRS>(a, t, acc, smem, ct0, lane, wave);
  if constexpr (ABL & 64) {
    ph[3] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ph[4] = __builtin_amdgcn_s_memrealtime();
    if (a.dbg && lane == 0) {
      unsigned long long* o = a.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
#pragma unroll
      for (int i = 0; i < 5; ++i) o[i] = ph[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 64- / 128-token tiles: everything by LDS-DMA into a ring of NBUF stage slots, counted vmcnt
// ------------------------------------------------------------------------------------------------
// Slot = [x: MB * 8 KiB][packed weights: PAIRS * 8 KiB, 1 KiB pieces per wave][(scale, zero) words: NW waves x PAIRS x NG x 256 B].
// At the top of iteration s the slots hold stages s .. s + NBUF - 2: s and s + 1 landed and visible to every wave (s + 1
// because a wave reads its stage-(s + 1) weights and first token fragments out of the slot during stage s), the rest in
// flight; slot (s - 1) % NBUF was released by the barrier that ended iteration s - 1 and receives stage s + NBUF - 1,
// issued over the units of stage s.  Every wave issues the same L LDS-DMA instructions per stage, so "stage s + 2 has
// landed" is s_waitcnt vmcnt((NBUF - 3) * L) at the end of iteration s, followed by the one barrier of the stage.
//
// WK = 2: EIGHT waves, two per SIMD -- wave (wn, wk) owns the channels of wave wn and the k16 steps of parity wk of every
// stage (for wk = 0 these are the "lo" dwords of its weights, for wk = 1 the "hi" ones: no weight is fetched or dequantised
// twice).  One wave alone on a SIMD issues ~1 instruction per 6 cycles on this instruction mix (dependent packed-f16 chains,
// waits): at 64 tokens per tile that, not the matrix pipe, bounds the stage.  Two waves interleave.  The two K halves are
// added through LDS after the loop.
template <int MB, int PAIRS, int GM, int NBUF, int ABL = 0, int WK = 1>
__global__ __launch_bounds__(256 * WK) void w4a16_ring_kernel(const GemmArgs a) {
  if constexpr (ABL & 32) span_stamp(a.span, 0);  // (the one "ablation" bit that changes nothing but writes the in-kernel span stamps)
  unsigned long long ph[5];  // ABL bit 64: s_memrealtime stamps (100 MHz) at the phase boundaries, per wave, into a.dbg (tools/wide_phases.py)
  if constexpr (ABL & 64) ph[0] = __builtin_amdgcn_s_memrealtime();
  constexpr int NG = groups_per_tile<GM>();
  constexpr int NW = 4 * WK;
  constexpr int X_BYTES = MB * 8192, W_BYTES = PAIRS * 8192, S_BYTES = NW * PAIRS * NG * 256;
  constexpr int SLOT = X_BYTES + W_BYTES + S_BYTES;
  constexpr int XI = MB * 2 / WK, LW = PAIRS * 2 / WK, LS = PAIRS * NG, L = XI + LW + LS, NU = 8 / WK * PAIRS;
  constexpr int PENDING = (NBUF - 3) * ((ABL & 4) ? LW + LS : ((ABL & 8) ? XI : L));
  static_assert(NBUF >= 3 && NBUF * SLOT <= 160 * 1024 && PENDING <= 63, "ring does not fit LDS / the vmcnt field");
  static_assert(WK == 1 || (WK - 1) * 4 * PAIRS * MB * 4096 <= NBUF * SLOT, "K-half exchange must fit in the ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NBUF * SLOT

  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int wn = wave & 3, wk = wave >> 2;
  const int rho = lane & 31, h = lane >> 5;
  const WideTile t = wide_tile<MB, PAIRS>(a);
  const int ct0 = (t.nb * 4 + wn) * PAIRS * 2;
  const WideBufs<XI> b = wide_bufs<MB, PAIRS, WK>(a, t.m0, ct0, lane, wave);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned xdst = lds_base + (unsigned)wave * 1024u;                           // + slot + i * NW KiB
  const unsigned wdst = lds_base + X_BYTES + (unsigned)wave * (LW * 1024);           // + slot + piece * 1024
  const unsigned sdst = lds_base + X_BYTES + W_BYTES + (unsigned)wave * (LS * 256);  // + slot + (p * NG + i) * 256
  // B-fragment read address of this lane: row rho, chunk (k16 step wk) * 2 + h, swizzled by rho % 16
  const unsigned xrd = (lds_base + (unsigned)rho * 256u + (unsigned)((h ^ (rho & 15)) << 4)) ^ (WK == 2 ? (unsigned)wk << 5 : 0u);
  const unsigned w_voff_hi = b.w_voff + 512u;

  // item j of stage kt -> slot at byte offset `slot`: the x rows first, then the packed weights, then the group words
  auto issue = [&](int j, int kt, unsigned slot) {
    if constexpr (ABL & 4) { if (j < XI) return; }   // (timing experiments: no x / no weight traffic)
    if constexpr (ABL & 8) { if (j >= XI) return; }
    if (j < XI) {
      lds_dma16(b.x, b.x_voff[j], (unsigned)kt * 256u, xdst + slot + j * (NW * 1024));
    } else if (j < XI + LW) {
      const int piece = j - XI;  // WK = 1: (pair, lo / hi); WK = 2: pair (this wave's half only)
      const int p = WK == 2 ? piece : (piece >> 1), hi = WK == 2 ? 0 : (piece & 1);
      lds_dma16(b.w, hi ? w_voff_hi : b.w_voff, (unsigned)p * b.w_pstride + (unsigned)kt * 1024u, wdst + slot + piece * 1024);
    } else {
      const int p = (j - XI - LW) / NG, i = (j - XI - LW) % NG;
      const unsigned g = (unsigned)group_index<GM>(kt, i * (4 / NG), a.tpg, a.G);
      lds_dma4(b.s, b.s_voff, (unsigned)p * b.s_pstride + g * 64u, sdst + slot + (j - XI - LW) * 256);
    }
  };
  typedef const __attribute__((address_space(3))) char* lds_ptr;
  auto read_w = [&](WideW<PAIRS, GM>& w, unsigned slot) {  // this lane's own 16 (+ 16) bytes per pair, and its group words
    const lds_ptr wp = (lds_ptr)(uintptr_t)(wdst + slot + (unsigned)lane * 16u);
    const lds_ptr sp = (lds_ptr)(uintptr_t)(sdst + slot + (unsigned)lane * 4u);
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      if constexpr (WK == 2) {
        w.lo[p] = *(const __attribute__((address_space(3))) u32x4*)(wp + p * 1024);
        w.hi[p] = w.lo[p];
      } else {
        w.lo[p] = *(const __attribute__((address_space(3))) u32x4*)(wp + (2 * p) * 1024);
        w.hi[p] = *(const __attribute__((address_space(3))) u32x4*)(wp + (2 * p + 1) * 1024);
      }
#pragma unroll
      for (int i = 0; i < NG; ++i) w.sz[p][i] = *(const __attribute__((address_space(3))) uint32_t*)(sp + (p * NG + i) * 256);
    }
  };

  floatx16 acc[PAIRS][MB];
  wide_zero<MB, PAIRS>(acc);
  const DqConsts dq = make_dq_consts();
  WideW<PAIRS, GM> wc, wn_;

  // prologue: stages 0 .. NBUF - 2 into slots 0 .. NBUF - 2 (past the end of the K range: replays of the last stage)
#pragma unroll
  for (int q = 0; q < NBUF - 1; ++q) {
    const int kt = min(t.kt_lo + q, t.kt_hi - 1);
#pragma unroll
    for (int j = 0; j < L; ++j) issue(j, kt, (unsigned)q * SLOT);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING) : "memory");  // stages 0 and 1 have landed
  __builtin_amdgcn_s_barrier();
  read_w(wc, 0u);
  WideCarry<MB, PAIRS, GM> carry;
  wide_prepare<MB, PAIRS, GM, true, WK>(carry, wc, xrd, dq);

  if constexpr (ABL & 64) ph[1] = __builtin_amdgcn_s_memrealtime();
  unsigned cur = 0u, nxt = (unsigned)SLOT, fill = (unsigned)(NBUF - 1) * SLOT;  // slot offsets of stage s, s + 1, s + NBUF - 1
  for (int s = 0; s < t.nstage; ++s) {
    const int ktf = min(t.kt_lo + s + NBUF - 1, t.kt_hi - 1);
    wide_compute<MB, PAIRS, GM, ABL, true, WK>(wc, wn_, xrd + cur, xrd + nxt, dq, acc, carry, [&](int u) {
      constexpr int PER = NU > 1 ? (L + NU - 2) / (NU - 1) : L;  // everything issued one unit before the stage ends
#pragma unroll
      for (int j = 0; j < PER; ++j)
        if constexpr (!(ABL & 2))
          if (u * PER + j < L) issue(u * PER + j, ktf, fill);
      if (u == NU / 2) read_w(wn_, nxt);  // next stage's packed weights: LDS -> registers, half a stage early
    });
    if constexpr (!(ABL & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING) : "memory");  // stage s + 2 has landed ...
    __builtin_amdgcn_s_barrier();                                    // ... in every wave; everybody is done with stage s
    wc = wn_;
    fill = cur;
    cur = nxt;
    nxt = nxt + SLOT >= (unsigned)(NBUF * SLOT) ? 0u : nxt + SLOT;
  }
  if constexpr (ABL & 64) ph[2] = __builtin_amdgcn_s_memrealtime();
  if constexpr (WK == 2) {  // add the second K half to the first through LDS (the ring is free: drain the replayed DMAs first)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    floatx4* ex = (floatx4*)smem;  // [wn][p][mt][c][lane]
    if (wk == 1) {
#pragma unroll
      for (int p = 0; p < PAIRS; ++p)
#pragma unroll
        for (int mt = 0; mt < MB; ++mt)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            ex[(((wn * PAIRS + p) * MB + mt) * 4 + c) * 64 + lane] =
                floatx4{acc[p][mt][4 * c], acc[p][mt][4 * c + 1], acc[p][mt][4 * c + 2], acc[p][mt][4 * c + 3]};
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int p = 0; p < PAIRS; ++p)
#pragma unroll
        for (int mt = 0; mt < MB; ++mt)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const floatx4 v = ex[(((wn * PAIRS + p) * MB + mt) * 4 + c) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[p][mt][4 * c + r] += v[r];
          }
    }
    __syncthreads();  // (splitk_arrive writes its flag into the same LDS)
  }
  if constexpr (ABL & 64) ph[3] = __builtin_amdgcn_s_memrealtime();   // (K halves added)
  wide_finish<MB, PAIRS, WK>(a, t, acc, smem, ct0, lane, wn, wk == 0);
  if constexpr (ABL & 32) span_stamp(a.span, 1);
  if constexpr (ABL & 64) {  // (only the workgroups that finish a tile get here when K is split)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ph[4] = __builtin_amdgcn_s_memrealtime();
    if (a.dbg && lane == 0) {
      unsigned long long* o = a.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
#pragma unroll
      for (int i = 0; i < 5; ++i) o[i] = ph[i];
    }
  }
}

}  // namespace quick_amd
