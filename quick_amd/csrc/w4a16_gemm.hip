// W4A16 GEMM for MI355X (gfx950): y[M,N] = x[M,K] @ fp16((w - z) * s).
//
// Replaces the five __global__ kernels + host dispatcher of the reference
// (csrc/gemm_cuda_quick.cu:1199-1517) with a design made for CDNA4:
//   * weights are the MFMA *A* operand (rows = output channels), activations the *B* operand, so
//     the accumulator fragment of a lane is 4 consecutive output channels of one token -> 8-byte
//     row-major stores with no transpose;
//   * the offline interleave ("mi355x order") makes one 16-byte global load per lane
//     deliver the A fragments of 4 consecutive k-steps of v_mfma_f32_16x16x32_f16: dequantised
//     weights go HBM -> VGPR -> matrix core, never through LDS (the QUICK idea, re-derived for the
//     64-lane MFMA operand order instead of mma.sync/ldmatrix);
//   * activations: skinny kernel (M <= 64) -- one 16-channel tile per workgroup, K split over the
//     workgroup's waves, x either broadcast from an LDS copy (tiny M) or loaded as fragments straight
//     from L2; tiled kernel (large M) -- 64 x 128 workgroup tile, x staged in LDS *in fragment order*
//     by global_load_lds, 8 waves = 4 along N x 2 along K.
#include "w4a16_common.hpp"
#include "../../include/quick_amd.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "w4a16_args.hpp"
#include "w4a16_wide.hpp"
#include "w4a16_xk_host.hpp"
#include "w4a16_xw_host.hpp"
#include "w4a16_lean_host.hpp"
#include "w4a16_xm_host.hpp"
namespace quick_amd {

// ------------------------------------------------------------------------------------------------
// skinny kernel: one workgroup owns 16 tokens x (NTW*16) channels for the whole K; its waves split K (and
// blockIdx.z splits it further when the grid would not fill the chip).
// ------------------------------------------------------------------------------------------------
// Every wave walks its k-tiles in chunks of U.  A chunk is LOADED (weights 16 B/lane/tile straight from HBM, raw
// group constants, and -- unless x sits in LDS -- the B fragments from L2, once per k-step for all NTW channel
// tiles) one chunk ahead of being COMPUTED, into the other of two register sets; the sched_barriers keep hipcc
// from sinking the loads next to their uses, which would serialise one HBM round trip per tile.
// NTW trades workgroup count against x traffic: token blocks re-read the weights (L2 hits), channel blocks re-read
// x, so NTW ~ number of token blocks keeps both at N/16 workgroups' worth.
//
// XLDS: the workgroup first copies x[rows, kbegin:kend] into LDS with coalesced 16-byte loads (row pitch padded
// by 16 B so the token rows of a fragment read spread over the bank groups) and every B fragment is a
// ds_read_b128 -- for M == 1 a 4-address broadcast -- instead of a 64-lane global load fetching 64 B per token
// row, 15/16 of them wasted when M == 1.
template <int NTW, int GM, int U, bool XLDS, bool LN = false>
struct SkinnyChunk {
  u32x4 w[U][NTW];
  GroupRaw raw[U][NTW][groups_per_tile<GM>()];
  half8_t xf[XLDS ? 1 : U][XLDS ? 1 : 4];
  half8_t gf[LN ? U : 1][LN ? 4 : 1];  // RMSNorm weight fragments (LN: the norm is applied to the x fragments in registers)
};

// Weights and group constants come in through buffer loads: address = descriptor base + per-lane byte offset (a launch
// constant VGPR) + a wave-uniform byte offset in an SGPR -- no 64-bit address arithmetic per load (it was ~28 scalar and
// ~10 vector instructions per 1 KiB weight tile).  Offsets are 32-bit: tensors up to 4 GiB.
struct SkinnyBufs {
  __amdgpu_buffer_rsrc_t w, s;
  unsigned w_voff, s_voff;  // lane * 16;  4 * n16 (this lane's channel within its 16-channel block)
  unsigned wstride_bytes;   // between consecutive channel tiles of the weights
  unsigned sstride_bytes;   // between consecutive channel tiles of the (scale, zero point) words: groups * 64
};
__device__ __forceinline__ SkinnyBufs skinny_bufs(const GemmArgs& a, int lane) {
  SkinnyBufs b;
  const unsigned groups = (unsigned)(a.K / a.G);
  b.w = __builtin_amdgcn_make_buffer_rsrc((void*)a.QW, 0, (unsigned)((size_t)a.K * a.N / 2), 0x00020000);
  b.s = __builtin_amdgcn_make_buffer_rsrc((void*)a.S, 0, groups * (unsigned)a.N * 4u, 0x00020000);
  b.w_voff = (unsigned)lane * 16u;
  b.s_voff = 4u * (unsigned)(lane & 15);
  b.wstride_bytes = (unsigned)(a.K >> 7) * 1024u;
  b.sstride_bytes = groups * 64u;
  return b;
}

// chunk of U k-tiles starting at kt of channel block `cb` (in units of 16 channels: block index * NTW)
// NT: the weight requests carry the nt (streaming) cache policy -- launches with ONE token block, where every weight byte is read once
// [r05 A/B builds, profiles/r05_ab_skinny_nt.txt: 1 x 28672 x 8192 24.0 -> 21.9 us, 16 x 8192 x 57344 59.3 -> 57.5, Llama-2-70B bs = 16 1358 -> 1379 tok/s;
// with two token blocks the second one's re-read misses L2: 32 x 4096 x 8192 8.6 -> 9.1, so those keep the default policy]
template <int NTW, int GM, int U, bool XLDS, bool LN = false, bool NT = false>
__device__ __forceinline__ void skinny_load(SkinnyChunk<NTW, GM, U, XLDS, LN>& c, int kt, int kt_last, const SkinnyBufs& b,
                                            int cb, const half_t* xp, const GemmArgs& a, const half_t* gp = nullptr) {
  constexpr int NG = groups_per_tile<GM>();
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int j = 0; j < NTW; ++j)  // past the end: replay
      c.w[u][j] = __builtin_amdgcn_raw_buffer_load_b128(b.w, b.w_voff, (unsigned)(cb + j) * b.wstride_bytes + (unsigned)min(kt + u, kt_last) * 1024u, NT ? 2 : 0);
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const unsigned g = (unsigned)group_index<GM>(min(kt + u, kt_last), i * (4 / NG), a.tpg, a.G);
        c.raw[u][j][i].sz = __builtin_amdgcn_raw_buffer_load_b32(b.s, b.s_voff, (unsigned)(cb + j) * b.sstride_bytes + g * 64u, 0);
      }
  if constexpr (!XLDS) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) c.xf[u][t] = *(const half8_t*)(xp + min(kt + u, kt_last) * 128 + 32 * t);
  }
  if constexpr (LN) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) c.gf[u][t] = *(const half8_t*)(gp + min(kt + u, kt_last) * 128 + 32 * t);
  }
}

template <int NTW, int GM, int U, bool XLDS>
__device__ __forceinline__ void skinny_compute(const SkinnyChunk<NTW, GM, U, XLDS>& c, int kt, int kt_end, const char* xl,
                                               floatx4 (&acc)[NTW]) {
  constexpr int NG = groups_per_tile<GM>();
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (kt + u < kt_end) {  // wave-uniform
      GroupQ grp[NTW][NG];
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int i = 0; i < NG; ++i) grp[j][i] = make_group(c.raw[u][j][i]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        half8_t bf;
        if constexpr (XLDS) bf = *(const half8_t*)(xl + ((kt + u) * 128 + 32 * t) * 2);
        else bf = c.xf[u][t];
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[j] = mfma16(dequant8(c.w[u][j][t], grp[j][group_slot<GM>(t)]), bf, acc[j]);
      }
    } else {
      // a skipped (replayed) tile still counts as consumed: otherwise hipcc's s_waitcnt pass carries its loads as
      // pending into the next iteration and stalls the next prefetch on the write-after-write hazard
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        asm volatile("" ::"v"(c.w[u][j]));
#pragma unroll
        for (int i = 0; i < NG; ++i) asm volatile("" ::"v"(c.raw[u][j][i].sz));
      }
      if constexpr (!XLDS) {
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(c.xf[u][t]));
      }
    }
  }
}

// DEFERRED-ZERO compute (M <= 16, x in LDS).  The exact kernel spends 13 VALU ops per packed dword on fp16((w - z) * s)
// and the matrix core hides none of them; at M <= 16 that, not HBM, bounds the kernel.  Here the roles of the MFMA
// operands are swapped -- x is the A operand (rows = tokens), the still-biased weights biased8() the B operand (columns =
// channels) -- so that a lane's accumulators are 4 tokens of ONE channel and the group constants are per-lane scalars:
//     y[m, n] = sum_units s[g, n] * ( sum_{k in unit} x[m, k] * (b_k + w[n, k])  -  C[unit, m]  -  z[g, n] * A[unit, m] )
// with b_k = 1024 or 64 (see biased8), A[unit, m] = sum_k x[m, k] and C[unit, m] = sum_k b_k x[m, k] tabulated once per
// workgroup while x is copied to LDS (`tab`: per unit 16 x A then 16 x -C, fp32).  A unit is min(G, 128) consecutive k.
// The accumulator of a unit starts at -C, so the epilogue of a unit is 2 FMAs per register.  5 VALU per dword plus
// ~12 per (unit, channel tile) instead of 13 + 5; products x * (b + w) are exact in the fp32 accumulator and w - z is
// never rounded to fp16, i.e. this path differs from the exact kernel by less than the latter's own weight rounding
// (2^-11 relative per weight) -- see DESIGN.md for the bound and tests/test_gemm_gpu.py for the comparison.
template <int NTW, int GM, int U>
__device__ __forceinline__ void skinny_compute_dz(const SkinnyChunk<NTW, GM, U, true>& c, int kt, int kt_end,
                                                  const char* xl, const float* tab,
                                                  floatx4 (&acc)[NTW]) {
  constexpr int NG = groups_per_tile<GM>();  // units per 128-k tile
  constexpr int TPU = 4 / NG;                // k-steps per unit
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (kt + u < kt_end) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const float* tp = tab + ((kt + u) * NG + i) * 32;
        const floatx4 xa = *(const floatx4*)tp;
        const floatx4 nc = *(const floatx4*)(tp + 16);
        floatx4 g[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) g[j] = nc;
#pragma unroll
        for (int t = i * TPU; t < (i + 1) * TPU; ++t) {
          const half8_t xf = *(const half8_t*)(xl + ((kt + u) * 128 + 32 * t) * 2);
#pragma unroll
          for (int j = 0; j < NTW; ++j) g[j] = mfma16(xf, biased8(c.w[u][j][t]), g[j]);
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const float s = group_scale_f32(c.raw[u][j][i]), z = group_zero_f32(c.raw[u][j][i]);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[j][r] = __builtin_fmaf(s, __builtin_fmaf(-z, xa[r], g[j][r]), acc[j][r]);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        asm volatile("" ::"v"(c.w[u][j]));
#pragma unroll
        for (int i = 0; i < NG; ++i) asm volatile("" ::"v"(c.raw[u][j][i].sz));
      }
    }
  }
}

// Deferred-zero compute WITHOUT the LDS copy of x (fragments straight from L2, any M <= 16 x NTW tiles): the unit sums
// A = sum_k x and C = sum_k b_k x come out of the matrix core too -- one extra MFMA per k-step, shared by the NTW channel
// tiles, with a CONSTANT B operand whose even columns are all ones and whose odd columns hold b_k (1024 / 64 in
// biased8's order) -- so an accumulator lane ends up with A (even channel lane) or C (odd channel lane) of its 4 tokens
// and gets the other from its neighbour with one DPP move.  ~12 VALU per unit per wave on top of 5 per packed dword and
// ~12 per (unit, channel tile); no table, no LDS, no per-workgroup prologue.
// LN: RMSNorm folded in -- the x fragments are multiplied by the norm weight in registers (fp16, like the weight multiply of
// quick_rmsnorm_f16), the squares of the raw x are summed per lane (`ssq`), and the row scale 1/rms is applied to the
// fp32 result in skinny_finish: gemm(x * w) * rstd instead of gemm(fp16(fp16(x * rstd) * w)).
template <int NTW, int GM, int U, bool LN = false>
__device__ __forceinline__ void skinny_compute_dzf(const SkinnyChunk<NTW, GM, U, false, LN>& c, int kt, int kt_end,
                                                   half8_t bconst, bool odd, floatx4 (&acc)[NTW],
                                                   float* ssq = nullptr) {
  constexpr int NG = groups_per_tile<GM>();  // units per 128-k tile
  constexpr int TPU = 4 / NG;                // k-steps per unit
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (kt + u < kt_end) {  // wave-uniform
      half8_t xs[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (LN) {
          xs[t] = c.xf[u][t] * c.gf[u][t];
          const u32x4 raw = __builtin_bit_cast(u32x4, c.xf[u][t]);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const uint32_t rd = raw[d];
            *ssq = __builtin_amdgcn_fdot2(as_h2(rd), as_h2(rd), *ssq, false);
          }
        } else {
          xs[t] = c.xf[u][t];
        }
      }
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        floatx4 sm = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = i * TPU; t < (i + 1) * TPU; ++t) sm = mfma16(xs[t], bconst, sm);
        floatx4 xa, nc;  // A and -C of this lane's 4 tokens
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mine = sm[r];  // (a copy: __builtin_bit_cast of the vector element itself reads element 0)
          const float other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine), 0xB1, 0xf, 0xf, false));
          xa[r] = odd ? other : mine;
          nc[r] = -(odd ? mine : other);
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          floatx4 g = nc;
#pragma unroll
          for (int t = i * TPU; t < (i + 1) * TPU; ++t) g = mfma16(xs[t], biased8(c.w[u][j][t]), g);
          const float s = group_scale_f32(c.raw[u][j][i]), z = group_zero_f32(c.raw[u][j][i]);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[j][r] = __builtin_fmaf(s, __builtin_fmaf(-z, xa[r], g[r]), acc[j][r]);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        asm volatile("" ::"v"(c.w[u][j]));
#pragma unroll
        for (int i = 0; i < NG; ++i) asm volatile("" ::"v"(c.raw[u][j][i].sz));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(c.xf[u][t]));
      if constexpr (LN) {
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(c.gf[u][t]));
      }
    }
  }
}

// Finishes the channel block `nb`: the waves' K partials meet in LDS (buffer `red`), wave j < NTW adds them for
// channel tile j, joins the other K slices if there are any, applies the epilogue and stores.  Returns with `acc`
// zeroed for the next block.  One workgroup barrier (two more when K is split across workgroups).
// TR: the accumulators come from the deferred-zero path (lane = channel n16, registers = tokens 4q..4q+3) and are
// written to LDS transposed, so that everything after the barrier sees the usual fragment (lane = token, registers =
// channels 4q..4q+3).
template <int NTW, int WAVES, bool TR, bool LN = false>
__device__ __forceinline__ void skinny_finish(const GemmArgs& a, floatx4 (&acc)[NTW], floatx4* red, char* smem, int nb,
                                              int nblocks, int mb, int ks, int lane, int wave, float ssq = 0.f) {
  const int n16 = lane & 15, q = lane >> 4;
  float* ssq_lds = (float*)(red + WAVES * NTW * 64);  // LN: [WAVES][16] behind the reduction buffer
  if constexpr (LN) {  // this wave's sum of squares per token: the 4 k-octet lanes of a token row, then LDS
    ssq += __shfl_xor(ssq, 16);
    ssq += __shfl_xor(ssq, 32);
    if (q == 0) ssq_lds[wave * 16 + n16] = ssq;
  }
  if constexpr (TR && NTW == 1) {
    if (a.ksplit == 1) {
      // Deferred-zero path, K not split across workgroups: every wave finishes its own 16 / WAVES tokens (lane = one
      // output), so no wave does the whole reduction while the others wait for it at the next block's barrier.
      // Partials in LDS row-major [wave][token][channel].
      float* rf = (float*)(red + wave * 64) + 64 * q + n16;
#pragma unroll
      for (int r = 0; r < 4; ++r) rf[16 * r] = acc[0][r];
      acc[0] = floatx4{0.f, 0.f, 0.f, 0.f};
      __syncthreads();
      constexpr int TPW = WAVES >= 16 ? 1 : 16 / WAVES;  // tokens per wave
      const int t = min(wave * TPW + q, 15), m = mb * 16 + t;
      if (wave * TPW >= min(16, a.M - mb * 16)) return;  // whole waves: the shuffle below stays wave-wide
      const float* src = (const float*)red + t * 16 + n16;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) v += src[w * 256];
      const bool live = q < TPW && m < a.M;
      if (a.silu_mul) {
        const float up = __shfl_xor(v, 8);  // channels 0..7 gate, 8..15 up
        if (live && n16 < 8) a.Y[(size_t)m * (a.N >> 1) + nb * 8 + n16] = silu_mul_f16((half_t)v, (half_t)up);
        return;
      }
      if (live) {
        const int n = nb * 16 + n16;
        if (a.bias) v += (float)a.bias[n];
        if (a.residual) v += (float)a.residual[(size_t)m * a.N + n];
        a.Y[(size_t)m * a.N + n] = (half_t)v;
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    if constexpr (TR) {
      float* rf = (float*)(red + (wave * NTW + j) * 64) + 64 * (n16 >> 2) + 16 * q + (n16 & 3);
#pragma unroll
      for (int r = 0; r < 4; ++r) rf[4 * r] = acc[j][r];
    } else {
      red[(wave * NTW + j) * 64 + lane] = acc[j];
    }
    acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  floatx4 sum = floatx4{0.f, 0.f, 0.f, 0.f};
  if (wave < NTW) {
    sum = red[wave * 64 + lane];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) sum += red[(w * NTW + wave) * 64 + lane];
  }
  if (a.ksplit > 1) {  // never combined with the persistent loop: this workgroup owns exactly one block
    const int tile = mb * nblocks + nb;
    constexpr unsigned SLAB_BYTES = NTW * 1024;
    const __amdgpu_buffer_rsrc_t rs = slab_rsrc(a.slabs + (size_t)tile * a.ksplit * (NTW * 256), a.ksplit * SLAB_BYTES);
    if (wave < NTW) slab_store(rs, ks * SLAB_BYTES + (wave * 64 + lane) * 16, sum);
    if (!splitk_arrive(a.counters + tile, a.ksplit, (unsigned*)smem)) return;
    if (wave < NTW) {  // slices are added in index order (own partial from registers at its index): the result does
      const floatx4 own = sum;  // not depend on which workgroup happened to arrive last
      for (int o = 0; o < a.ksplit; ++o) {
        const floatx4 part = o == ks ? own : slab_load(rs, o * SLAB_BYTES + (wave * 64 + lane) * 16);
        sum = o == 0 ? part : sum + part;
      }
    }
  }
  if (wave >= NTW) return;
  const int m = mb * 16 + n16;
  if constexpr (LN) {  // lane = token n16: scale its 4 channels by 1 / rms(x[token])
    float ss = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) ss += ssq_lds[w * 16 + n16];
    const float rstd = rsqrtf(ss / (float)a.K + a.ln_eps);
#pragma unroll
    for (int r = 0; r < 4; ++r) sum[r] *= rstd;
  }
  if (a.silu_mul) {
    floatx4 up;
#pragma unroll
    for (int r = 0; r < 4; ++r) up[r] = __shfl_xor(sum[r], 32);  // lanes q = 0,1 hold gate, their partners q = 2,3 up
    if (m < a.M && q < 2) {
      half4_t o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = silu_mul_f16((half_t)sum[r], (half_t)up[r]);
      *(half4_t*)(a.Y + (size_t)m * (a.N >> 1) + (nb * NTW + wave) * 8 + 4 * q) = o;
    }
    return;
  }
  const int nc = (nb * NTW + wave) * 16 + 4 * q;  // lane holds channels nc..nc+3 of token m
  if (m < a.M) {
    if (a.bias) {
      const half4_t b = *(const half4_t*)(a.bias + nc);
#pragma unroll
      for (int r = 0; r < 4; ++r) sum[r] += (float)b[r];
    }
    if (a.residual) {
      const half4_t b = *(const half4_t*)(a.residual + (size_t)m * a.N + nc);
#pragma unroll
      for (int r = 0; r < 4; ++r) sum[r] += (float)b[r];
    }
    half4_t o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (half_t)sum[r];
    *(half4_t*)(a.Y + (size_t)m * a.N + nc) = o;
  }
}

// PERSISTENT launch (gridDim.x < channel blocks; needs ksplit == 1): a workgroup walks the channel blocks
// blockIdx.x, + gridDim.x, ...  The chunk pipeline runs across block boundaries -- while the last chunk of a block
// is computed and its partials are reduced, the first chunk of the workgroup's next block is already in flight --
// so the HBM stream does not stop for the load-latency / compute / reduce phases of each block, and x is copied to
// LDS once per workgroup instead of once per block.  Consecutive blocks alternate between two reduction buffers,
// which makes one barrier per block enough.
// Only the XLDS variants can be launched persistent.  The fragments-from-L2 variants keep the plain one-block loop: in
// the cross-block form hipcc needs 141 instead of 121 VGPRs for NTW = 1 (one workgroup per CU instead of two: -9 % on
// the Llama-2-70B shapes at M = 16 [r01]).
// (span stamps: written out at the kernel's own two exits -- moving the body into a forceinline device function called between
// two stamps changed hipcc's code for the deferred-zero paths into something that fails the parity tests [r02])
template <int NTW, int WAVES, int GM, bool XLDS, bool DZ, bool LN = false, bool SPAN = false, bool NT = false>
__global__ __launch_bounds__(WAVES * 64) void w4a16_skinny_kernel(const GemmArgs a) {
  if constexpr (SPAN) span_stamp(a.span, 0);
  constexpr bool PERSIST = XLDS;
  static_assert(!LN || (DZ && !XLDS && NTW >= 2), "the register-level RMSNorm lives in the fragment deferred-zero flavour");
  static_assert(WAVES >= NTW, "the final reduction assigns one channel tile per wave");
  constexpr int U = XLDS ? (NTW == 1 ? 4 : 2) : (NTW <= 2 ? 2 : 1);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nblocks = a.N / (16 * NTW);
  const int nred = (int)gridDim.x < nblocks ? 2 : 1;
  floatx4* red = (floatx4*)smem;  // [nred][WAVES][NTW][64]
  char* xlds = smem + nred * (WAVES * NTW * 64 * sizeof(floatx4));

  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int mb = blockIdx.y, ks = blockIdx.z;
  // XCD-aware block order: workgroups are dealt to the 8 XCDs round-robin, and neighbouring channel blocks share the
  // 128-byte lines of their group constants (4 blocks per line of scales, 16 per line of zero points) -- give every XCD
  // a contiguous run of blocks, so that its L2 fetches each of those lines once instead of all eight fetching it.
  const int gx = (int)gridDim.x;
  const int bx = (gx & 7) == 0 ? ((int)blockIdx.x & 7) * (gx >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int KT = a.K >> 7;
  const int wg_begin = ks * a.kt_per_split, wg_end = min(KT, wg_begin + a.kt_per_split);
  const int cnt = wg_end - wg_begin;
  const int kt_begin = wg_begin + cnt * wave / WAVES, kt_end = wg_begin + cnt * (wave + 1) / WAVES;
  const int kt_last = max(kt_end - 1, kt_begin);  // clamp for replayed loads (a wave may own no k-tile at all)

  const SkinnyBufs bufs = skinny_bufs(a, lane);
  const int row = min(mb * 16 + n16, a.M - 1);  // rows >= M replay row M-1; never stored
  const half_t* xp = a.X + (size_t)row * a.K + q * 8;

  floatx4 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};

  // (block, k-tile) of the chunk being computed / of the chunk being loaded; both wave-uniform
  int nb_cur = bx, kt_cur = kt_begin;
  int nb_nxt = nb_cur, kt_nxt = kt_cur;
  int parity = 0;
  // The load of the next chunk is issued in a block that always issues it: a guarded load would make hipcc's
  // s_waitcnt pass assume the smaller in-flight count and wait for most of the prefetch before the compute.
#define QA_SKINNY_LOAD(c) skinny_load<NTW, GM, U, XLDS, LN, NT>(c, kt_nxt, kt_last, bufs, nb_nxt * NTW, xp, a)
#define QA_SKINNY_ADVANCE(nb, kt)                                                                                  \
  do {                                                                                                             \
    kt += U;                                                                                                       \
    if (kt >= kt_end) {                                                                                            \
      kt = kt_begin;                                                                                               \
      nb += gridDim.x;                                                                                             \
    }                                                                                                              \
  } while (0)
#define QA_SKINNY_COMPUTE(ccomp)                                                                                   \
  if constexpr (DZ && XLDS) skinny_compute_dz<NTW, GM, U>(ccomp, kt_cur, kt_end, xl, tab, acc);                    \
  else if constexpr (!DZ) skinny_compute<NTW, GM, U, XLDS>(ccomp, kt_cur, kt_end, xl, acc);                        \
  if (kt_cur + U >= kt_end) {                                                                                      \
    skinny_finish<NTW, WAVES, DZ>(a, acc, red + parity * (WAVES * NTW * 64), smem, nb_cur, nblocks, mb, ks, lane,  \
                                  wave);                                                                           \
    parity = (nred - 1) - parity;                                                                                  \
  }
#define QA_SKINNY_STEP(cload, ccomp)                                                                               \
  if (nb_nxt >= nblocks) { /* the chunk in hand is this workgroup's last */                                        \
    QA_SKINNY_COMPUTE(ccomp);                                                                                      \
    break;                                                                                                         \
  }                                                                                                                \
  QA_SKINNY_LOAD(cload);                                                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  QA_SKINNY_COMPUTE(ccomp);                                                                                        \
  nb_cur = nb_nxt;                                                                                                 \
  kt_cur = kt_nxt;                                                                                                 \
  QA_SKINNY_ADVANCE(nb_nxt, kt_nxt)

  SkinnyChunk<NTW, GM, U, XLDS, LN> cA, cB;
  if constexpr (!PERSIST) {
    float ssq = 0.f;
    const half_t* gp = LN ? a.ln_w + q * 8 : nullptr;
    // one block per workgroup, x fragments straight from L2
    const int cb = bx * NTW;
    // B operand of the sum MFMAs (deferred-zero flavour): column = lane & 15; even columns ones, odd columns b_k
    const u32x4 bc_bits = (lane & 1) ? u32x4{0x64006400u, 0x54005400u, 0x64006400u, 0x54005400u}
                                     : u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
    const half8_t bconst = __builtin_bit_cast(half8_t, bc_bits);
    // [r05, profiles/r05_skinny_variants.txt -- read before touching this loop]  Behind the GUARDED requests below hipcc's s_waitcnt pass
    // assumes the smaller in-flight count: the ISA waits vmcnt(3..0) in front of the first MFMAs of the chunk in hand, i.e. the chunk just
    // requested is drained too.  Issuing them unconditionally (replays past the wave's range) gives counted waits (vmcnt(15..12)) and
    // 3-7 % at 6..16 tokens -- and WRONG results in the four-tile instantiation (tile 0 of every block), as did an eight-tile
    // instantiation of this guarded form (tile 2, varying from run to run; 62 -> 54 us at 16 x 8192 x 57344, its x fragments being half
    // the weights' bytes instead of as many).  Both are cured by -mllvm -amdgpu-waitcnt-forcezero: the wait counts hipcc derives for
    // requests whose results die on the loop's way out cannot be trusted here.  Neither is shipped; the instantiations that are have
    // been through the parity suite in this exact form since r01.
    if (kt_begin < kt_end) skinny_load<NTW, GM, U, XLDS, LN, NT>(cA, kt_begin, kt_end - 1, bufs, cb, xp, a, gp);
    __builtin_amdgcn_sched_barrier(0);
#ifdef QA_EXP_SKINNY_UNCOND   // (A/B builds only: the unconditional requests of profiles/r05_skinny_variants.txt -- wrong results in the four-tile build)
#define QA_SKINNY_IF(c)
#else
#define QA_SKINNY_IF(c) if (c)
#endif
#ifndef QA_EXP_SKINNY_WAIT
#define QA_EXP_SKINNY_WAIT 0   // (A/B builds: full waits at chosen points of the loop, to bisect which request the ISA's own waits do not cover)
#endif
#define QA_SKINNY_FULLWAIT(bit) do { if constexpr ((QA_EXP_SKINNY_WAIT >> (bit)) & 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); } while (0)
    QA_SKINNY_FULLWAIT(2);
    for (int kt = kt_begin; kt < kt_end; kt += 2 * U) {
      QA_SKINNY_IF(kt + U < kt_end) skinny_load<NTW, GM, U, XLDS, LN, NT>(cB, kt + U, kt_end - 1, bufs, cb, xp, a, gp);
      __builtin_amdgcn_sched_barrier(0);
      QA_SKINNY_FULLWAIT(0);
      if constexpr (DZ) skinny_compute_dzf<NTW, GM, U, LN>(cA, kt, kt_end, bconst, (lane & 1) != 0, acc, &ssq);
      else skinny_compute<NTW, GM, U, XLDS>(cA, kt, kt_end, nullptr, acc);
      if (kt + U >= kt_end) break;
      QA_SKINNY_IF(kt + 2 * U < kt_end) skinny_load<NTW, GM, U, XLDS, LN, NT>(cA, kt + 2 * U, kt_end - 1, bufs, cb, xp, a, gp);
      __builtin_amdgcn_sched_barrier(0);
      QA_SKINNY_FULLWAIT(1);
      if constexpr (DZ) skinny_compute_dzf<NTW, GM, U, LN>(cB, kt + U, kt_end, bconst, (lane & 1) != 0, acc, &ssq);
      else skinny_compute<NTW, GM, U, XLDS>(cB, kt + U, kt_end, nullptr, acc);
    }
    QA_SKINNY_FULLWAIT(3);
    skinny_finish<NTW, WAVES, DZ, LN>(a, acc, red, smem, bx, nblocks, mb, ks, lane, wave, ssq);
    if constexpr (SPAN) span_stamp(a.span, 1);
    return;
  }
  QA_SKINNY_LOAD(cA);  // HBM requests first
  QA_SKINNY_ADVANCE(nb_nxt, kt_nxt);
  __builtin_amdgcn_sched_barrier(0);

  const char* xl = nullptr;
  const float* tab = nullptr;
  if constexpr (XLDS) {
    // copy x[mb*16 .. , wg_begin*128 .. wg_end*128) -> LDS [rows][pitch]; DZ: tabulate the unit sums on the way
    const int rows = min(16, a.M - mb * 16);
    const int kc = cnt * 16;  // 16-byte chunks per row
    const int pitch = cnt * 256 + 16;
    xl = xlds + min(n16, rows - 1) * pitch + q * 16 - wg_begin * 256;
    float* tab0 = (float*)(xlds + rows * pitch);
    constexpr int NG = groups_per_tile<GM>();
    constexpr int L = 16 / NG;  // lanes (16-byte chunks) per unit
    if constexpr (DZ) tab = tab0 + 4 * q - wg_begin * NG * 32;
    if constexpr (DZ) {
      if (a.ln_w) {
        // RMSNorm prologue (needs the whole row: ksplit == 1).  Pass 1 copies x raw and sums its squares per row;
        // pass 2, below, finds x in LDS instead of in global memory, scales it in place and tabulates the result.
        float* ssq = (float*)smem;  // [rows][WAVES], in the still unused reduction buffer
        for (int r = 0; r < rows; ++r) {
          const half_t* src = a.X + (size_t)(mb * 16 + r) * a.K;
          float ss = 0.f;
          for (int c = threadIdx.x; c < kc; c += WAVES * 64) {
            const u32x4 v = *(const u32x4*)(src + c * 8);
            *(u32x4*)(xlds + r * pitch + c * 16) = v;
#pragma unroll
            for (int i = 0; i < 4; ++i) ss = __builtin_amdgcn_fdot2(as_h2(v[i]), as_h2(v[i]), ss, false);
          }
          ss = wave_sum(ss);
          if (lane == 0) ssq[r * WAVES + wave] = ss;
        }
        __syncthreads();
      }
    }
    // Rows in batches of RB: the loads of a batch are all issued before the first of them is used.  One row at a time this loop was a
    // chain of `rows` round trips to L2 (load -> LDS store -> unit sums -> table), 1.2-1.5 us of an 8-token launch [r02 stamps]; the
    // deferred-zero table flavour lost to the exact path at 3..16 tokens for that reason alone.
    constexpr int RB = 4;
    for (int r0 = 0; r0 < rows; r0 += RB) {
      float inv[RB];
#pragma unroll
      for (int j = 0; j < RB; ++j) inv[j] = 0.f;
      if constexpr (DZ) {
        if (a.ln_w) {
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const int r = min(r0 + j, rows - 1);
            float ss = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) ss += ((const float*)smem)[r * WAVES + w];
            inv[j] = rsqrtf(ss / (float)a.K + a.ln_eps);
          }
        }
      }
      for (int c = threadIdx.x; c < kc; c += WAVES * 64) {  // kc % 16 == 0: rows of 16 lanes are all in or all out
        u32x4 v[RB];
        bool normed = false;
        if constexpr (DZ) {
          if (a.ln_w) {  // fp16(fp16(x * inv) * weight): the rounding points of quick_rmsnorm_f16 (and of torch)
            normed = true;
            const half8_t gv = *(const half8_t*)(a.ln_w + c * 8);
#pragma unroll
            for (int j = 0; j < RB; ++j) {
              const half8_t xv = *(const half8_t*)(xlds + min(r0 + j, rows - 1) * pitch + c * 16);
              half8_t o;
#pragma unroll
              for (int i = 0; i < 8; ++i) o[i] = (half_t)((half_t)((float)xv[i] * inv[j]) * gv[i]);
              v[j] = __builtin_bit_cast(u32x4, o);
            }
          }
        }
        if (!normed) {
#pragma unroll
          for (int j = 0; j < RB; ++j)  // (past the last row: a replay of it, not stored)
            v[j] = *(const u32x4*)(a.X + (size_t)(mb * 16 + min(r0 + j, rows - 1)) * a.K + wg_begin * 128 + c * 8);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int r = r0 + j;
          if (r < rows) {  // workgroup-uniform
            *(u32x4*)(xlds + r * pitch + c * 16) = v[j];
            if constexpr (DZ) {
              const half2_t one2 = {(half_t)1.f, (half_t)1.f};
              const float lo = __builtin_amdgcn_fdot2(as_h2(v[j][0]), one2, __builtin_amdgcn_fdot2(as_h2(v[j][2]), one2, 0.f, false), false);
              const float hi = __builtin_amdgcn_fdot2(as_h2(v[j][1]), one2, __builtin_amdgcn_fdot2(as_h2(v[j][3]), one2, 0.f, false), false);
              const float sa = lanes_sum<L>(lo + hi);
              const float sc = lanes_sum<L>(1024.f * lo + 64.f * hi);
              if ((lane & (L - 1)) == 0) {
                float* t = tab0 + (c / L) * 32 + r;
                t[0] = sa;
                t[16] = -sc;
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }

  while (true) {
    QA_SKINNY_STEP(cB, cA);
    QA_SKINNY_STEP(cA, cB);
  }
  if constexpr (SPAN) span_stamp(a.span, 1);
#undef QA_SKINNY_STEP
#undef QA_SKINNY_COMPUTE
#undef QA_SKINNY_ADVANCE
#undef QA_SKINNY_LOAD
}

// ------------------------------------------------------------------------------------------------
// tiled kernel
// ------------------------------------------------------------------------------------------------
// per-wave state of the tiled kernel that does not change over the K loop
template <int BMT, int TN, int WK, int WN = 4>
struct TiledCtx {
  static constexpr int XPW = 4 * BMT / WN;  // x fragments staged per wave per stage (4*WK*BMT fragments, WN*WK waves)
  const u32x4* wp[TN];      // this lane's 16 bytes of weight tile (channel tile j, k-tile 0)
  int ncol[TN];             // this lane's output channel in channel tile j
  const half_t* xsrc[XPW];  // global source of the fragments this wave stages (k-tile 0 of a stage)
  int xkt[XPW];             // which of the stage's WK k-tiles fragment i belongs to
  int kt_lo, kt_hi, wave, wk;
};

// x fragments of stage s: global -> registers (ordinary loads: hipcc counts them, so they can stay in flight
// across barriers; global_load_lds cannot -- the compiler drains it at every barrier and at the first use of any
// other load)
template <int BMT, int TN, int WK, int WN>
__device__ __forceinline__ void tiled_load_x(const TiledCtx<BMT, TN, WK, WN>& c, int s, u32x4 (&xr)[4 * BMT / WN]) {
#pragma unroll
  for (int i = 0; i < 4 * BMT / WN; ++i) {
    const int kt = min(c.kt_lo + WK * s + c.xkt[i], c.kt_hi - 1);  // past the end: replay the last tile (unused)
    xr[i] = *(const u32x4*)(c.xsrc[i] + kt * 128);
  }
}
// registers -> LDS in B-fragment order: fragment f = wave*BMT + i is 1 KiB, lane l at byte 16 l
template <int BMT, int TN, int WK, int WN>
__device__ __forceinline__ void tiled_store_x(const TiledCtx<BMT, TN, WK, WN>& c, char* buf, int lane,
                                              const u32x4 (&xr)[4 * BMT / WN]) {
  constexpr int XPW = 4 * BMT / WN;
#pragma unroll
  for (int i = 0; i < XPW; ++i) *(u32x4*)(buf + (c.wave * XPW + i) * 1024 + lane * 16) = xr[i];
}

// weights + raw group constants of this wave's k-tile of stage s (no dependent ALU: see GroupRaw)
template <int BMT, int TN, int WK, int GM, int WN>
__device__ __forceinline__ void tiled_load_w(const TiledCtx<BMT, TN, WK, WN>& c, const GemmArgs& a, int s, u32x4 (&w)[TN],
                                             uint32_t (&gs)[TN][groups_per_tile<GM>()]) {
  constexpr int NG = groups_per_tile<GM>();
  const int kt = min(c.kt_lo + WK * s + c.wk, c.kt_hi - 1);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    w[j] = c.wp[j][(size_t)kt * 64];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int g = group_index<GM>(kt, i * (4 / NG), a.tpg, a.G);
      const GroupRaw r = load_group_raw(a.S, a.QZ, g, c.ncol[j], a.N, a.K / a.G);
      gs[j][i] = r.sz;  // (the zero point travels in the same word)
    }
  }
}

// ABL (ablation bits, timing experiments only -- results are wrong when set): 1 = no global loads in the K loop,
// 2 = no LDS traffic in the K loop, 4 = no dequantisation, 8 = no barrier.
template <int BMT, int TN, int WK, int GM, int ABL, int WN>
__device__ __forceinline__ void tiled_compute(const TiledCtx<BMT, TN, WK, WN>& c, const char* sb, int s, const u32x4 (&w)[TN],
                                              const uint32_t (&gs)[TN][groups_per_tile<GM>()],
                                              floatx4 (&acc)[TN][BMT]) {
  constexpr int NG = groups_per_tile<GM>();
  if (c.kt_lo + WK * s + c.wk >= c.kt_hi) {  // wave-uniform: a ragged last stage has no tile for this wave
#pragma unroll
    for (int j = 0; j < TN; ++j) {  // "use" the loads so that both paths leave the same ones pending (see the K loop)
      asm volatile("" ::"v"(w[j]));
#pragma unroll
      for (int i = 0; i < NG; ++i) asm volatile("" ::"v"(gs[j][i]));
    }
    return;
  }
  GroupQ grp[TN][NG];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < NG; ++i) grp[j][i] = make_group(GroupRaw{gs[j][i]});
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    half8_t bf[BMT];
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) {
      if constexpr (ABL & 2) bf[mt] = __builtin_bit_cast(half8_t, w[0]);
      else bf[mt] = *(const half8_t*)(sb + (t * BMT + mt) * 1024);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      half8_t af;
      if constexpr (ABL & 4) af = __builtin_bit_cast(half8_t, w[j]);
      else af = dequant8(w[j][t], grp[j][group_slot<GM>(t)]);
#pragma unroll
      for (int mt = 0; mt < BMT; ++mt) acc[j][mt] = mfma16(af, bf[mt], acc[j][mt]);
    }
  }
}

// Tiled kernel.  Workgroup tile (BMT*16 tokens) x (4*TN*16 channels); 4*WK waves = 4 along N x WK along K.
// One stage = WK*128 k: wave (wn, wk) owns channel tiles wn*TN..+TN-1 and the wk-th 128-k weight tile of every
// stage, so no weight is dequantised twice inside a workgroup and every LDS fragment is read by 4 waves only.  The
// token tile of a stage lives in LDS *in B-fragment order* (lane l of fragment (kt, t, mt) holds
// x[mt*16 + l%16][kt*128 + 32t + 8*(l/16) ..+7]), so every ds_read_b128 / ds_write_b128 is lane-linear and
// conflict free.  The WK partial sums are added through LDS in the epilogue.
//
// Software pipeline (2 LDS buffers, 2 weight register sets, 1 x register set; loop unrolled by 2 so that every
// slot index is static):  iteration s  =  { x(s+1): regs -> LDS[other] ; issue x(s+2) loads ; compute stage s from
// LDS[cur] with weights[s%2] ; issue weights(s+2) into the set just freed ; barrier }.
// Loads past the last stage are NOT guarded: they replay the last tile (clamped index) and are never consumed -- a
// guard would merge "issued" and "not issued" paths and make hipcc drain the whole load queue at every consumer.
// For the same reason the unrolled pair has no early exit: with an odd stage count the second half replays a stage
// (stores it, skips its compute) -- an `if (s >= nstage) break` gives the waitcnt pass a path from one half-iteration
// straight into the same half again, on which the weights just requested look like the ones about to be used.
template <int BMT, int TN, int WK, int GM, int ABL = 0, int WN = 4>
__global__ __launch_bounds__(64 * WN * WK) void w4a16_tiled_kernel(const GemmArgs a) {
  constexpr int XPW = 4 * BMT / WN;
  constexpr int NG = groups_per_tile<GM>();
  constexpr int FRAGS = 4 * WK * BMT;        // 1 KiB fragments per stage
  constexpr int STAGE_BYTES = FRAGS * 1024;  // 32 KiB at BMT = 4, WK = 2
  static_assert((WK - 1) * WN * TN * BMT * 1024 <= 2 * STAGE_BYTES, "epilogue exchange must fit in the stage buffers");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 * STAGE_BYTES

  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int wn = wave % WN, wk = wave / WN;
  const int n16 = lane & 15, q = lane >> 4;
  const int NB = a.N / (16 * WN * TN);
  int nb = blockIdx.x % NB, mb = blockIdx.x / NB;
  if (a.xcd_gm > 0) {
    // Workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on it).  Give every XCD a compact
    // rectangle of tiles so that its private L2 fetches few distinct x rows AND few distinct weight columns.
    const int MB = gridDim.x / NB, gn = 8 / a.xcd_gm;
    const int mcnt = MB / a.xcd_gm, ncnt = NB / gn;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    mb = (xcd / gn) * mcnt + idx / ncnt;
    nb = (xcd % gn) * ncnt + idx % ncnt;
  }
  const int ks = blockIdx.y;
  const int KT = a.K >> 7;
  TiledCtx<BMT, TN, WK, WN> c;
  c.kt_lo = ks * a.kt_per_split;
  c.kt_hi = min(KT, c.kt_lo + a.kt_per_split);
  c.wave = wave;
  c.wk = wk;
  const int nstage = (c.kt_hi - c.kt_lo + WK - 1) / WK;
  const int m0 = mb * BMT * 16;
  const int nt0 = (nb * WN + wn) * TN;  // first 16-channel tile of this wave

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    c.wp[j] = a.QW + (size_t)(nt0 + j) * KT * 64 + lane;
    c.ncol[j] = (nt0 + j) * 16 + n16;
  }
  // fragments this wave stages: f = wave*BMT + i  ->  (k-tile f / (4 BMT), k-step (f / BMT) % 4, token tile f % BMT)
#pragma unroll
  for (int i = 0; i < XPW; ++i) {
    const int f = wave * XPW + i, t = (f / BMT) & 3, mt = f % BMT;
    c.xkt[i] = f / (4 * BMT);
    const int row = min(m0 + mt * 16 + n16, a.M - 1);
    c.xsrc[i] = a.X + (size_t)row * a.K + 32 * t + 8 * q;
  }

  floatx4 acc[TN][BMT];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) acc[j][mt] = floatx4{0.f, 0.f, 0.f, 0.f};

  unsigned long long t_entry = 0, r_entry = 0;
  if constexpr (ABL & 16) {
    t_entry = __builtin_amdgcn_s_memtime();
    r_entry = __builtin_amdgcn_s_memrealtime();
  }
  u32x4 xr[XPW];
  u32x4 w[2][TN];
  uint32_t gs[2][TN][NG];
  const int rd = wk * (4 * BMT * 1024) + lane * 16;  // this wave reads its k-tile's part of a stage

  if (nstage > 0) {
    // Issue order = the K loop's steady state (x(s+1) then w(s+1) in flight at the top of iteration s), and pinned:
    // hipcc's waitcnt pass merges the prologue's view of "which loads are still in flight" into every iteration, so a
    // prologue that ends on x loads (or lets the scheduler sink a scale load below them) turns the loop's
    // s_waitcnt vmcnt(7..4) into vmcnt(3..0) -- a full drain of the weight prefetch once per stage [r01: -4 %].
    tiled_load_x<BMT, TN, WK, WN>(c, 0, xr);
    __builtin_amdgcn_sched_barrier(0);
    tiled_load_w<BMT, TN, WK, GM, WN>(c, a, 0, w[0], gs[0]);
    __builtin_amdgcn_sched_barrier(0);
    tiled_store_x<BMT, TN, WK, WN>(c, smem, lane, xr);
    __builtin_amdgcn_sched_barrier(0);
    tiled_load_x<BMT, TN, WK, WN>(c, 1, xr);
    __builtin_amdgcn_sched_barrier(0);
    tiled_load_w<BMT, TN, WK, GM, WN>(c, a, 1, w[1], gs[1]);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();

  // ABL bit 16: s_memtime stamps between the phases of every iteration, summed per wave (perturbs the schedule a bit)
  unsigned long long ph[5] = {0, 0, 0, 0, 0};
  unsigned long long t_prev = 0;
#define QA_STAMP(i)                                              \
  if constexpr (ABL & 16) {                                      \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
    ph[i] += t_now - t_prev;                                     \
    t_prev = t_now;                                              \
  }
  if constexpr (ABL & 16) t_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long t_loop = t_prev;
  for (int s0 = 0; s0 < nstage; s0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int s = s0 + u;  // (s == nstage for odd nstage: a replayed stage whose compute is skipped)
      char* const cur = smem + u * STAGE_BYTES;
      char* const nxt = smem + (u ^ 1) * STAGE_BYTES;
      if constexpr (!(ABL & 2)) tiled_store_x<BMT, TN, WK, WN>(c, nxt, lane, xr);  // stage s+1 (a replay at the very end)
      QA_STAMP(0)
      if constexpr (!(ABL & 1)) tiled_load_x<BMT, TN, WK, WN>(c, s + 2, xr);
      QA_STAMP(1)
      tiled_compute<BMT, TN, WK, GM, ABL, WN>(c, cur + rd, s, w[u], gs[u], acc);
      QA_STAMP(2)
      if constexpr (!(ABL & 1)) tiled_load_w<BMT, TN, WK, GM, WN>(c, a, s + 2, w[u], gs[u]);
      QA_STAMP(3)
      if constexpr (!(ABL & 8)) __syncthreads();
      QA_STAMP(4)
    }
  }
  if constexpr (ABL & 16) {
    if (lane == 0 && a.dbg) {
      unsigned long long* o = a.dbg + ((size_t)blockIdx.x * (WN * WK) + wave) * 8;
#pragma unroll
      for (int i = 0; i < 5; ++i) o[i] = ph[i];
      o[5] = (unsigned long long)nstage | (r_entry << 8);  // (+ when the wave started, 100 MHz ticks)
      o[6] = t_loop - t_entry;  // prologue (first loads) in shader cycles
      // shader cycles and 100 MHz ticks from entry to the end of the K loop: the clock the kernel actually ran at
      o[7] = ((__builtin_amdgcn_s_memtime() - t_entry) << 24) | ((__builtin_amdgcn_s_memrealtime() - r_entry) & 0xffffff);
    }
  }
#undef QA_STAMP

  // add the WK partial sums through LDS (the stage buffers are free after the last barrier)
  floatx4* ex = (floatx4*)smem;  // [wk-1][wn][j][mt][lane]
  if (wk > 0) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int mt = 0; mt < BMT; ++mt) ex[((((wk - 1) * WN + wn) * TN + j) * BMT + mt) * 64 + lane] = acc[j][mt];
  }
  __syncthreads();
  if (wk == 0) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int mt = 0; mt < BMT; ++mt)
#pragma unroll
        for (int k = 1; k < WK; ++k) acc[j][mt] += ex[((((k - 1) * WN + wn) * TN + j) * BMT + mt) * 64 + lane];
  }
  if (a.ksplit > 1) {
    constexpr unsigned SLAB_BYTES = WN * TN * BMT * 1024;  // one fp32 partial tile
    const __amdgpu_buffer_rsrc_t rs =
        slab_rsrc(a.slabs + (size_t)blockIdx.x * a.ksplit * (SLAB_BYTES / 4), a.ksplit * SLAB_BYTES);
    const unsigned my = ((wn * TN) * BMT * 64 + lane) * 16;
    if (wk == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int mt = 0; mt < BMT; ++mt) slab_store(rs, ks * SLAB_BYTES + my + (j * BMT + mt) * 1024, acc[j][mt]);
    }
    if (!splitk_arrive(a.counters + blockIdx.x, a.ksplit, (unsigned*)smem)) return;
    if (wk == 0) {  // slices are added in index order (own partial from registers at its index): the result does not
      floatx4 own[TN][BMT];  // depend on which workgroup happened to arrive last
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int mt = 0; mt < BMT; ++mt) own[j][mt] = acc[j][mt];
      for (int o = 0; o < a.ksplit; ++o) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int mt = 0; mt < BMT; ++mt) {
            const floatx4 part = o == ks ? own[j][mt] : slab_load(rs, o * SLAB_BYTES + my + (j * BMT + mt) * 1024);
            acc[j][mt] = o == 0 ? part : acc[j][mt] + part;
          }
      }
    }
  }
  if (a.silu_mul) {
    if (wk == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int mt = 0; mt < BMT; ++mt) {
          floatx4 up;
#pragma unroll
          for (int r = 0; r < 4; ++r) up[r] = __shfl_xor(acc[j][mt][r], 32);
          const int m = m0 + mt * 16 + n16;
          if (m < a.M && q < 2) {
            half4_t o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = silu_mul_f16((half_t)acc[j][mt][r], (half_t)up[r]);
            *(half4_t*)(a.Y + (size_t)m * (a.N >> 1) + (nt0 + j) * 8 + 4 * q) = o;
          }
        }
    }
    return;
  }
  if (wk == 0) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nc = (nt0 + j) * 16 + 4 * q;
      half4_t b = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
      if (a.bias) b = *(const half4_t*)(a.bias + nc);
#pragma unroll
      for (int mt = 0; mt < BMT; ++mt) {
        const int m = m0 + mt * 16 + n16;
        if (m < a.M) {
          half4_t res = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
          if (a.residual) res = *(const half4_t*)(a.residual + (size_t)m * a.N + nc);
          half4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (half_t)(acc[j][mt][r] + (float)b[r] + (float)res[r]);
          *(half4_t*)(a.Y + (size_t)m * a.N + nc) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// tiled kernel, 32x32x16 MFMA flavour.  Same workgroup tile, wave grid, LDS double buffer and pipeline as above,
// but the matrix instruction is v_mfma_f32_32x32x16_f16: 32 cycles per instruction instead of 16, which -- unlike
// the 16-cycle 16x16x32 -- leaves the SIMD free to issue ~4 VALU ops per MFMA (tools/mfma_valu_overlap.hip), so
// part of the dequantisation hides under the matrix pipe inside ONE wave.
//   A operand (32 channels x 16 k; lane l: channel l%32, k-half l/32): built from the raw packed dwords of TWO
//     16-channel tiles -- whose lanes are (channel%16, k-octet l/16) -- with v_permlane16_swap + v_permlane32_swap:
//     [r0.q0 r1.q0 | r0.q1 r1.q1] feeds the first k16 step of a k32 step, [r0.q2 r1.q2 | r0.q3 r1.q3] the second.
//     The weight layout in HBM is unchanged.
//   B operand (16 k x 32 tokens; lane l: token l%32, k-half l/32): the LDS image of a stage is
//     [k-tile][k16 step 0..7][32-token tile][64 lanes x 16 B]; the staging loads keep the 16-row x 64-byte shape and
//     each lane writes its 16 bytes to the slot the 32-wide fragment wants.
//   C/D: lane l holds token l%32 and channels (r%4) + 8*(r/4) + 4*(l/32), r = 0..15, of the 32-channel pair.
// ------------------------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));


// empty kernel with the GEMM's launch shape: what the dispatch-duration clock reads with no work at all
__global__ __launch_bounds__(512) void w4a16_empty_kernel(unsigned* sink) {
  if (sink != nullptr && threadIdx.x == 0xffffffffu) sink[0] = 1;
}

// ------------------------------------------------------------------------------------------------
// workspace check: the counter region and the exchange zone must be all-zero between launches (include/quick_amd.h, "workspace").
// One pass over them; the lowest dirty byte offset lands in *first (0xffffffff = clean).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void w4a16_workspace_scan_kernel(const u32x4* __restrict__ ws, unsigned n16, unsigned* __restrict__ first) {
  unsigned lo = 0xffffffffu;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) {
    const u32x4 v = __builtin_nontemporal_load(ws + i);
    if ((v[0] | v[1] | v[2] | v[3]) != 0u) lo = min(lo, i * 16u);
  }
  if (lo != 0xffffffffu) atomicMin(first, lo);
}

// ------------------------------------------------------------------------------------------------
// dense dequantisation (debug / parity aid): W[k, n] = fp16((w - z) * s), row-major [K, N]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void w4a16_dequant_kernel(const u32x4* __restrict__ QW, const half_t* __restrict__ S,
                                                           const uint32_t* __restrict__ QZ, half_t* __restrict__ W,
                                                           int K, int N, int G) {
  const int lane = threadIdx.x, n16 = lane & 15, q = lane >> 4;
  const int nt = blockIdx.x, kt = blockIdx.y, KT = K >> 7;
  const int n = nt * 16 + n16;
  const u32x4 w = QW[((size_t)nt * KT + kt) * 64 + lane];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int k0 = kt * 128 + 32 * t + 8 * q;
    const GroupQ g = make_group(load_group_raw(S, QZ, k0 / G, n, N, K / G));
    const half8_t af = dequant8(w[t], g);
#pragma unroll
    for (int j = 0; j < 8; ++j) W[(size_t)(k0 + j) * N + n] = af[j];
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
extern "C" int quick_w4a16_workspace_check(const void* workspace, size_t workspace_bytes, void* hip_stream);
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ring slots that fit the 160 KiB of LDS for a (MB, PAIRS) tile at one (scale, zero) word per pair and stage (G % 128 == 0);
// wk = waves along K (1: four waves, 2: eight)
constexpr int wide_ring_nbuf(int mb, int pairs, int wk = 1) {
  const int slot = mb * 8192 + pairs * 8192 + wk * pairs * 1024;
  const int n = (160 * 1024) / slot;
  return n > 6 ? 6 : n;
}

struct Plan {
  int kernel;  // QUICK_KERNEL_SKINNY / QUICK_KERNEL_TILED / QUICK_KERNEL_WIDE
  int wide_mb, wide_pairs;  // wide: token tiles of 32 per workgroup, 32-channel pairs per wave
  int wide_nbuf;            // wide: LDS ring slots (>= 3: w4a16_ring_kernel, everything by LDS-DMA; 0: w4a16_wide_kernel)
  int mt;      // skinny: channel tiles (of 16) per workgroup, NTW; tiled: token tiles per workgroup, BMT
  int waves;   // skinny: waves per workgroup
  bool xlds;   // skinny: x through an LDS copy
  int grid_x;  // skinny: workgroups along the channel blocks; fewer than the blocks = persistent workgroups
  bool dz;     // skinny: deferred-zero compute; with xlds the table flavour (NTW = 1), without the fragment flavour
  int ksplit;  // K slices across workgroups, reduced in-kernel by the last arriver
  int ntiles;  // output tiles (one arrival counter each)
  size_t slab_floats;  // fp32 elements of one partial tile
  int kt_per_split;
  int ablate;  // kernel bits 16-20: ablation variant of the tiled kernel (timing experiments only)
  int xcd_gm;  // tiled: rows of the XCD grid over the tile grid (0 = plain order)
  bool wn2;    // tiled: 2 x 4 wave grid (kernel bit 15)
  int tch;     // tiled: channels per workgroup tile, 128 or 256
  int xk_nbuf, xk_wd;  // exchange-K: x ring slots, weight queue depth (wide_mb = token tiles of 32, ksplit = slices that exchange)
  bool xk_loader;      // exchange-K: the twelve-wave flavour (four loader waves; kernel bit 12)
  int xk_kq;           // exchange-K: K groups of waves per workgroup, 2 (eight waves) or 4 (sixteen; kernel bit 13)
  int poll_log2;       // XW: log2 of the ticks (10 ns) a wave waits for a partner slice before it gives its block up (0 = the kernel's default)
  int lean_tmax;       // LEAN: the most k tiles a wave may own (the build's register ring), 0 = no build for this shape
  int frag8_t;         // SKINNY, mt == 8: k tiles per wave of the straight-line eight-tile fragment kernel (w4a16_frag8_kernel), 0 = not that kernel
  double est_us, est_xw_us;  // launch-time model: the r02 / r03 candidates' minimum, the four-wave kernels' minimum (0 = not evaluated)
};

// Where the main kernel goes: the stream, plus an optional event pair bound to that one dispatch
// (hipExtLaunchKernelGGL) so that a profiler-free caller can read the kernel's own duration.
struct Launch {
  hipStream_t st;
  hipEvent_t start, stop;
  unsigned long long* span = nullptr;  // device [2]: in-kernel span of this launch (see span_stamp)
};

static int check_shapes(int M, int K, int N, int G) {
  if (M <= 0 || K <= 0 || N <= 0 || G <= 0) return fail(QUICK_ERR_INVALID_ARGUMENT, "non-positive dimension");
  // same messages as csrc/gemm_cuda_quick.cu:1479-1484
  if (N % 128 != 0) return fail(QUICK_ERR_INVALID_ARGUMENT, "OC is not multiple of cta_N = 128");
  if (N % 8 != 0) return fail(QUICK_ERR_INVALID_ARGUMENT, "OC is not multiple of pack_num = 8");
  if (G % 32 != 0) return fail(QUICK_ERR_INVALID_ARGUMENT, "Group size should be a multiple of 32");
  if (K % G != 0) return fail(QUICK_ERR_UNSUPPORTED, "in_features (%d) is not a multiple of the group size (%d)", K, G);
  if (K % 128 != 0) return fail(QUICK_ERR_UNSUPPORTED, "in_features (%d) must be a multiple of 128 on MI355X", K);
  return QUICK_OK;
}

static constexpr size_t kLdsPerCu = 160 * 1024;  // gfx950
static thread_local bool g_span_unsupported = false;  // set by a launcher asked for span stamps its kernel does not carry

// [r05] EIGHT channel tiles per workgroup in the fragment deferred-zero flavour, as STRAIGHT-LINE code: 9..16 tokens on layers whose x (16 x K
// fp16) does not fit LDS (K >= 8192: Llama-2-70B).  Why eight: the launch time of this flavour follows the bytes of its x fragments, which come from L2 once per workgroup and k tile
// (64 B per token row and request: 16 cache lines per instruction) -- as many bytes as the weights at four tiles, half as many at eight (16 x 8192 x
// 57344: 62 -> 50 us); queue depth and occupancy do not move it (profiles/r05_skinny_variants.txt).  Why straight-line: the chunk loop of w4a16_skinny_kernel instantiated for eight tiles gave wrong
// results that -amdgpu-waitcnt-forcezero cured (hipcc's wait counts around guarded / replayed requests in a loop).  Here
// every wave owns EXACTLY T k tiles (the planner only picks the kernel when K / 128 == 8 waves x ksplit x T), every request is unconditional,
// nothing is replayed and no request's result dies unread: the form in which hipcc's counts are exact (as in the lean kernels).
// [r06, late] NTW = 7: the same kernel with SEVEN channel tiles per workgroup, for layers whose block count makes whole rounds that way -- 16 x 8192 x 57344 (Llama-2-70B
// gate_up: 57344 = 7 x 8192) is 448 blocks of 128 channels = 1.75 rounds of one workgroup per CU (212 registers), but 512 blocks of 112 = two whole rounds.
template <int T, bool NT, bool SPAN = false, int NTW_ = 8>
__global__ __launch_bounds__(512) void w4a16_frag8_kernel(const GemmArgs a) {
  if constexpr (SPAN) span_stamp(a.span, 0);
  constexpr int NTW = NTW_, WAVES = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  floatx4* red = (floatx4*)smem;  // [WAVES][NTW][64]
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int mb = blockIdx.y, ks = blockIdx.z;
  const int nblocks = a.N / (16 * NTW);
  const int gx = (int)gridDim.x;
  const int bx = (gx & 7) == 0 ? ((int)blockIdx.x & 7) * (gx >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;   // XCD-aware block order (skinny kernel)
  const int kt_begin = (ks * WAVES + wave) * T;   // a.kt_per_split == WAVES * T
  const SkinnyBufs bufs = skinny_bufs(a, lane);
  const int row = min(mb * 16 + n16, a.M - 1);  // rows >= M replay row M-1; never stored
  const half_t* xp = a.X + (size_t)row * a.K + q * 8;
  floatx4 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
  const u32x4 bc_bits = (lane & 1) ? u32x4{0x64006400u, 0x54005400u, 0x64006400u, 0x54005400u}
                                   : u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
  const half8_t bconst = __builtin_bit_cast(half8_t, bc_bits);
  const int cb = bx * NTW;
  const int kt_far = a.K >> 7;   // (no clamp ever bites: every k tile asked for is this wave's own)
  SkinnyChunk<NTW, 0, 1, false, false> c[2];
  skinny_load<NTW, 0, 1, false, false, NT>(c[0], kt_begin, kt_far, bufs, cb, xp, a);
#pragma unroll
  for (int i = 0; i < T; ++i) {
    if (i + 1 < T) skinny_load<NTW, 0, 1, false, false, NT>(c[(i + 1) & 1], kt_begin + i + 1, kt_far, bufs, cb, xp, a);   // (compile-time condition)
    __builtin_amdgcn_sched_barrier(0);
    skinny_compute_dzf<NTW, 0, 1, false>(c[i & 1], 0, 1, bconst, (lane & 1) != 0, acc);   // (kt = 0 < kt_end = 1: the k tile is always this wave's)
  }
  skinny_finish<NTW, WAVES, true, false>(a, acc, red, smem, bx, nblocks, mb, ks, lane, wave);
  if constexpr (SPAN) span_stamp(a.span, 1);
}

// LDS of one skinny workgroup: reduction buffer(s), the x copy, the deferred-zero table
static size_t skinny_lds_bytes(int M, int G, int ntw, int waves, int kt_per_split, bool persistent, bool xlds, bool dz) {
  size_t b = (size_t)(persistent ? 2 : 1) * waves * ntw * 1024;
  if (xlds) b += (size_t)std::min(M, 16) * (kt_per_split * 256 + 16);
  if (dz) b += (size_t)kt_per_split * (G >= 128 ? 1 : (G == 64 ? 2 : 4)) * 128;
  return b;
}

// compute units of the current device (256 on MI355X); 256 when there is no device to ask (plan_describe on a CPU-only host)
static int cu_count();
// CUs the K slices of an exchange-K launch may count on being CO-RESIDENT.  The slices of a tile poll each other's mailboxes: a launch
// whose workgroups cannot all run at once (a CU mask on the queue, CUs reserved by the runtime) would spin until the poll limit traps.
// The device attribute does not see such masks; QUICK_AMD_EXCHANGE_CUS=<n> tells the planner (0: never split K across CUs that way).
static int exchange_cus() {
  static const int env = [] {
    const char* e = getenv("QUICK_AMD_EXCHANGE_CUS");
    return e && *e ? std::max(0, atoi(e)) : -1;
  }();
  return env >= 0 ? std::min(env, cu_count()) : cu_count();
}
static int cu_count() {
  static thread_local int cached_dev = -2, cached = 256;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 256;
  }
  if (dev != cached_dev) {
    int n = 0;
    cached = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    (void)hipGetLastError();
    cached_dev = dev;
  }
  return cached;
}

// `kernel`: low 4 bits = family (QUICK_KERNEL_*), bits 4-7 = token tiles (tiled) / channel tiles per workgroup (skinny),
// 0 = auto; bits 8-11 = skinny waves / 4 (0 = auto).  The rest is for tests and tuning:
//   12 skinny: no LDS copy of x            13 tiled: 32x32x16 MFMA flavour      14 tiled: plain (not XCD-aware) tile order
//   15 tiled: 2 x 4 wave grid              16-20 tiled: ablation / phase stamps 21 skinny: flip the persistence default
//   22-24 skinny: persistent slots per CU  25 skinny: exact dequantisation      26 skinny: force the table deferred-zero path
//   27 tiled: force 128 x 256 four-wave tiles 28 skinny: no fragment deferred-zero   29 / 30 tiled: force / forbid 256-channel tiles
static int group_mode(int G) { return G == 128 ? 0 : (G % 128 == 0 ? 1 : (G == 64 ? 2 : (G == 32 ? 3 : 4))); }

// with_ln: the launch carries an RMSNorm prologue -- AUTO keeps to the kernels that have one (no mid-token, no eight-tile fragment launch)
static Plan make_plan(int M, int K, int N, int G, int kernel, int grid_split_k, bool allow_xk = true, bool with_ln = false) {
  Plan p{};
  const int KT = K / 128;
  const int family = kernel & 15, mt_req = (kernel >> 4) & 15, waves_req = ((kernel >> 8) & 15) * 4;
  const bool no_xlds = (kernel >> 12) & 1;
  p.ablate = (kernel >> 16) & 31;
  p.wn2 = (kernel >> 15) & 1;
  // [r05] lean small-M kernels (w4a16_lean.hpp): one workgroup per 16 tokens x (ntw x 16) channels, `waves` waves split K and request all
  // their k tiles up front.  Forced: family LEAN, bits 8-11 waves / 4 (1, 2, 4; 0 = choose), bits 4-7 channel tiles per workgroup (1, 2;
  // 0 = choose).  AUTO takes them for 1..4 tokens wherever a build exists, and up to 16 tokens on layers of <= 512 channel blocks -- each
  // workgroup fetches its own copy of x, which at 8..16 tokens on a wide layer is more L2 -> LDS traffic than the weights
  // [profiles/r05_lean_sweep.txt: in-kernel spans against the r01-r04 skinny kernels, one session: 1 x 4096 x 4096 4.81 -> 3.65 us, 8 x: 5.28 ->
  // 4.08, 16 x: 6.48 -> 4.58; 1 x 4096 x 12288 7.83 -> 6.77, 1 x 4096 x 22016 11.56 -> 10.64, 1 x 11008 x 4096 8.13 -> 7.00, 1 x 8192 x 8192 9.65 -> 8.36;
  // 8 x 4096 x 22016 13.6 -> 15.1 and 16 x 4096 x 12288 9.9 -> 10.7 stay with the old kernels].  QUICK_AMD_LEAN=0 switches the AUTO rule off (A/B).
  {
    static const int builds[7][3] = {{8, 4, 1}, {8, 4, 2}, {8, 8, 1}, {8, 12, 1}, {16, 4, 1}, {16, 4, 2}, {16, 8, 1}};
    const int mblocks = (M + 15) / 16;
    const bool forced = family == QUICK_KERNEL_LEAN;
    static const bool lean_on = [] {
      const char* e = getenv("QUICK_AMD_LEAN");
      return !(e && *e && atoi(e) == 0);
    }();
    const bool envelope = G % 128 == 0 && (size_t)K * N / 2 < ((size_t)1 << 31) && (size_t)M * K * 2 < ((size_t)1 << 31);
    if (forced || (family == QUICK_KERNEL_AUTO && lean_on && !mt_req && !waves_req && !(kernel >> 12) && grid_split_k <= 1 && envelope && M <= 16)) {
      // [r06 audit, profiles/r06_lean_rule_audit.txt] 5 / 6 tokens on the 768 blocks of 4096 x 12288: 8.5 -> 6.9 / 7.6 us
      const bool one_block_ok = forced || M <= 4 || N / 16 <= 512 || (M <= 6 && N / 16 <= 768 && KT <= 32);
      int bw = 0, bt = 0, bn = 0;
      double bcost = 0;
      for (const auto& b : builds) {
        if (!one_block_ok) break;
        const int waves = b[0], tmax = b[1], ntw = b[2];
        if (forced && ((waves_req && waves != waves_req) || (mt_req && ntw != mt_req))) continue;
        if (KT < waves || (KT + waves - 1) / waves > tmax || (N / 16) % ntw != 0) continue;
        if (lean_lds_need(M, K, waves, ntw, true) > kLdsPerCu) continue;
        // [r06 audit of this rule for K > 4096 (tools/lean_rule_audit.py, profiles/r06_lean_rule_audit.txt)] from five tokens a long K makes x
        // most of a workgroup's LDS (115-140 KB: one workgroup per CU): a launch that needs a second round of such workgroups loses to the
        // r01-r04 fragment kernels (12 x 5120 x 5120 10.1 against 8.1 us, 8 x 8192 x 8192 12.0 / 10.0, 6 x 11008 x 8192 15.5 / 13.5), one that fits
        // a single round wins (5..8 x 8192 x 4096 5.9-6.5 / 7.0-7.6, 5..8 x 5120 x 5120 5.4-5.8 / 7.3-7.6) -- so: all workgroups co-resident
        if (!forced && M >= 5 && KT > 32 && (long)(N / 16 / ntw) * mblocks > (long)std::max(1u, (unsigned)kLdsPerCu / lean_lds_need(M, K, waves, ntw, true)) * cu_count()) continue;
        // what the fullest CU streams: rounds of workgroups x bytes per workgroup; two tiles per workgroup share the head (0.85, measured);
        // sixteen waves with <= 4 tiles each and twelve tiles per wave cost
        const long wgs = (long)(N / 16 / ntw) * mblocks;
        double cost = (double)((wgs + 255) / 256) * ntw * (ntw == 2 ? 0.85 : 1.0);
        if (waves == 16 && tmax <= 4) cost *= 1.13;
        if (tmax >= 12) cost *= 1.12;
        cost *= 1.0 + 0.001 * tmax;  // (ties: the smallest register ring)
        if (bw == 0 || cost < bcost) {
          bcost = cost;
          bw = waves;
          bt = tmax;
          bn = ntw;
        }
      }
      if (forced || bw) {
        p.kernel = QUICK_KERNEL_LEAN;
        p.ksplit = 1;
        p.kt_per_split = KT;
        p.mt = bn ? bn : (mt_req == 2 ? 2 : 1);
        p.waves = bw ? bw : (waves_req ? waves_req : 8);
        p.lean_tmax = envelope ? bt : 0;
        p.ntiles = (N / 16) * mblocks;
        p.grid_x = N / 16 / p.mt;
        // persistent launches (fewer workgroups than channel blocks; builds of <= 4 tiles per wave, G = 128, one token block): forced by
        // kernel bits 22-24 = workgroups per CU
        const int slots_req = (kernel >> 22) & 7;
        if (forced && slots_req && bt && bt <= 4 && p.waves == 8 && G == 128 && mblocks == 1 && lean_lds_need(M, K, p.waves, p.mt, true, true) <= kLdsPerCu)
          p.grid_x = std::min(p.grid_x, cu_count() * slots_req);
        return p;
      }
      // 8..16 tokens on a wide layer: every one-block workgroup would fetch its own copy of x (more L2 -> LDS bytes than weights) -- one
      // persistent workgroup per CU instead, x staged once, the next block's weights requested under this block's tiles [profiles/r05_lean_persist.txt,
      // in-kernel spans against the r01-r04 picks: 16 x 4096 x 22016 18.0 -> 14.4 us, 16 x 4096 x 12288 10.2 -> 9.0, 8 x 4096 x 22016 13.7 -> 12.4]
      if (!forced && family == QUICK_KERNEL_AUTO && lean_on && !mt_req && !waves_req && !(kernel >> 12) && grid_split_k <= 1 && envelope && G == 128 && M >= 8 &&
          M <= 16 && N / 16 > 512 && KT >= 8 && KT <= 32) {
        const int ntw = ((N / 16) % 2 == 0 && M <= 12 && N / 16 >= 1024 && lean_lds_need(M, K, 8, 2, true, true) <= kLdsPerCu) ? 2 : 1;
        if (lean_lds_need(M, K, 8, ntw, true, true) <= kLdsPerCu) {
          p.kernel = QUICK_KERNEL_LEAN;
          p.ksplit = 1;
          p.kt_per_split = KT;
          p.mt = ntw;
          p.waves = 8;
          p.lean_tmax = 4;
          p.ntiles = (N / 16) * mblocks;
          p.grid_x = std::min(N / 16 / ntw, cu_count());
          return p;
        }
      }
    }
  }
  // [r06] mid-token kernels (w4a16_xm.hpp): one workgroup = 32 or 64 tokens x pr x 32 channels for all of K, eight waves splitting K, each with its
  // own x ring and weight queue; x is fetched once per workgroup, in whole cache lines.  Forced: family XM, bits 4-7 = channel pairs per workgroup
  // (1..3; 0 = choose), bits 8-9 = 1 / 2: 32- / 64-token tiles whatever the count.  A launch of these is launch ramp + the workgroup's x and
  // weight streams one behind the other + the dequantisation + the waves' reduction, with little overlap (profiles/r06_xm_anatomy.txt),
  // so what counts is ONE round of workgroups, few bytes of x per CU and every CU busy.  The rule, from the audit of every selection against
  // the other families on 15 layer shapes x 8 token counts, three boxes (profiles/r06_xm_audit.txt: pick / best geomean 1.0006-1.005, worst 1.04 on two boxes; QUICK_AMD_XM=0
  // switches the family off for A/B; tools/xm_rule_eval.py restates the rule and replays it against an audit file):
  //   * K >= 4096 (the audit's layers), and only layers that ONE round of workgroups covers with <= 3 channel pairs each (N = 27648 .. 57344: the
  //     exchange-K kernels stay);
  //   * 17..32 tokens, K <= 8192: the fewest pairs per workgroup that make one round (0.72-0.97 of the others' time; longer K: the fragment kernels,
  //     whose x is 16 tokens deep, stay ahead by 7-35 %);
  //   * 33..64 tokens on layers of <= 128 pairs (N <= 4096): two 32-token tiles x one pair -- every CU busy beats the halved dequantisation -- up to
  //     K = 8192, and up to 48 tokens at K = 11008 (0.83-1.0 of the exchange launch it replaces, whose time follows the box: three boxes,
  //     profiles/r06_xm_audit.txt; 56 / 64 tokens there and K = 14336 were ahead on one box and 4-16 % behind on two: not taken);
  //   * 65..128 tokens: the rule and its audit sit at the branch below;
  //   * 33..64 tokens, wider layers, K <= 8192: two 32-token tiles x <= 2 pairs where that is one round (N = 5120 .. 8192: 0.82-0.93; layers that
  //     leave > 30 % of the CUs idle only from 56 tokens), else 64-token tiles x the fewest pairs that make one round (N = 10240 .. 22016: 0.87-1.0).
  {
    const bool forced = family == QUICK_KERNEL_XM;
    static const bool xm_on = [] {
      const char* e = getenv("QUICK_AMD_XM");
      return !(e && *e && atoi(e) == 0);
    }();
    const int tpg = G / 128;
    const bool envelope = G % 128 == 0 && (tpg & (tpg - 1)) == 0 && (size_t)K * N / 2 < ((size_t)1 << 31) && (size_t)M * K * 2 < ((size_t)1 << 31);   // (a wave without a k tile adds zeros)
    const int pairs = N / 32, cus = cu_count();
    const auto one_round = [&](int mtiles) {   // fewest channel pairs per workgroup (1..3) that cover the layer in one round, or 0
      for (int c = 1; c <= 3; ++c)
        if ((pairs + c - 1) / c * mtiles <= cus) return c;
      return 0;
    };
    int mb = 0, pr = 0;
    if (forced) {
      const int tile_req = (kernel >> 8) & 3;   // 1 = 32-token tiles, 2 = 64-token tiles
      mb = tile_req ? tile_req : (M <= 32 ? 1 : 2);
      pr = mt_req ? std::max(1, std::min(3, mt_req)) : 1;
      if (!mt_req)
        while (pr < 3 && (long)((pairs + pr - 1) / pr) * ((M + mb * 32 - 1) / (mb * 32)) > cus) ++pr;
    } else if (family == QUICK_KERNEL_AUTO && xm_on && !with_ln && !mt_req && !waves_req && !(kernel >> 12) && grid_split_k <= 1 && envelope && M > 16 && M <= 128 && KT >= 32) {   // (K >= 4096: what the audit measured)
      if (M > 64) {
        // [r06, profiles/r06_xm_audit_65_128.txt (three boxes) and r06_planner_audit.txt, third section (two more, every family forced)] 65..128 tokens against the
        // four-slice 128 x 128 four-wave tile:
        //   * three or four 32-token tiles x the fewest pairs that make ONE round where that is <= 2 pairs (N <= 5120; K <= 8192: 80 x 4096 x 4096 10.7-11.8 -> 8.4-8.9 us,
        //     128 x: 11.0 -> 9.0-10.6, 96 x 5120 x 5120 12.7 -> 11.0; K = 11008 up to 80 tokens: 19.4 -> 18.6-19.3) or, up to 96 tokens, 3 pairs (4096 x 6144: 0.92-1.0);
        //   * else two 64-token tiles x <= 2 pairs that fill >= 80 % of the CUs: K = 4096 up to 128 tokens (4096 x 8192: 20.5 -> 13.6-15.0), K <= 8192 up to 95
        //     (8192 x 8192: 24.4 -> 21.9 on three boxes of four; the four-wave tile follows the box there, 18.8-25.4 us);
        //   * three pairs of 64-token tiles (4096 x 12288, 8192 x 10240) were ahead of the 64-token exchange tiles those counts ran until this audit, but are level with / 12 %
        //     behind the 128 x 128 four-wave tile, which the planner runs from 65 tokens since (below): not taken.  Wider layers, longer K: behind -- they stay.
        const int p1 = one_round((M + 31) / 32), p2 = one_round(2);
        if (p1 && p1 <= 2 && (KT <= 64 || (KT <= 86 && M <= 80))) mb = 1, pr = p1;
        else if (p1 == 3 && M <= 96 && KT <= 64) mb = 1, pr = 3;
        // [r06, end of round: 96 / 128 x 8192 x 8192 on three more boxes (profiles/r06_planner_audit.txt)] ... and up to 128 tokens with K <= 8192 as well: the four-slice 128 x 128 launch these
        // counts ran reads 19.6-32.8 us by box there (all 256 CUs in a four-way exchange), the two 64-token tiles 21.2-23.4: ahead by 9-11 % on three boxes of five, level on one, 2-4 % behind on one.
        else if (p2 && 10 * ((pairs + p2 - 1) / p2 * 2) >= 8 * cus && p2 <= 2 && KT <= 64) mb = 2, pr = p2;
      } else if (M <= 32) {
        if (KT <= 64 && (pr = one_round(1))) mb = 1;
      } else if (2 * pairs <= cus) {
        if (KT <= 64 || (KT <= 86 && M <= 48)) mb = 1, pr = 1;   // (K = 11008 at 56 / 64 tokens: 4-7 % behind the exchange launch on two boxes of three, 13 % ahead on the third; K = 14336: ahead on the audit's first box only, 7-16 % behind the eight-slice exchange launch on two more -- not taken)
      } else if (KT <= 64) {
        const int p1 = one_round(2), p2 = one_round(1);
        if (p1 && p1 <= 2) {
          const int wgs = (pairs + p1 - 1) / p1 * 2;
          if (10 * wgs >= 7 * cus || M >= 56) mb = 1, pr = p1;
        } else if (p2) {
          // [r06, three boxes, profiles/r06_xm_audit.txt] ... except 57..64 tokens on a long K (> 4096) where the 64-token tiles leave > 30 % of the CUs idle: 64 x 8192 x 10240
          // (160 workgroups) 21.6-22.0 us against 20.3-21.2 for the 64 x 128 four-wave tile on every box (level at 56 tokens, 3-8 % ahead up to 48)
          if (!(KT > 32 && M > 56 && 10 * ((pairs + p2 - 1) / p2) < 7 * cus)) mb = 2, pr = p2;
        }
      }
      if (!mb) pr = 0;
    }
    if (forced || pr) {
      const int mtiles = (M + mb * 32 - 1) / (mb * 32);
      p.kernel = QUICK_KERNEL_XM;
      p.ksplit = 1;
      p.kt_per_split = KT;
      p.wide_mb = mb;
      p.wide_pairs = pr;
      p.grid_x = (pairs + pr - 1) / pr;
      p.ntiles = p.grid_x * mtiles;
      p.waves = 8;
      p.lean_tmax = envelope ? 1 : 0;   // (0 = no build for this shape: the launch answers UNSUPPORTED)
      return p;
    }
  }
  // skinny: one workgroup per 16 tokens x 16..64 channels for all of K (x re-read per channel block, no cross-workgroup
  // reduction); tiled: 32..64 tokens x 128 channels through LDS.  Measured crossover [r01]: the tiled kernel wins from
  // M = 65, and from M = 17 once there are >= 64 tiles of 128 channels (N >= 8192) so that it needs no K split.
  // [r02] ... and at M = 57..64 on a small layer with a long K: 32-token tiles with a 4-way K split (64 x 11008 x 4096 17.7-18.9 ->
  // 17.0-17.3 us, 64 x 14336 x 4096 22.2 -> 19.8, 64 x 16384 x 4096 24.8 -> 21.0; a tie at M = 49..56, and up to K = 8192 the
  // skinny kernel stays ahead: 64 x 8192 x 4096 13.9 against 14.5 us, 64 x 4096 x 4096 8.3 against 10.9)
  const int tiled_tiles = (N / 128) * ((M + 63) / 64);
  // [r02 audit, profiles/r02_planner_audit*.jsonl] 48 tiles are enough (4096 x 6144, M = 24..64: 10.4-12.8 against the skinny
  // kernel's 11.4-14.2 us); and the long-K rule starts at M = 33 from K = 12288 (48 x 14336 x 4096: 22.2 -> 19.2)
  // (the 48-tile rule with G % 128 == 0 only: at G = 64 / 32 the four-tile exact skinny kernel is ahead there, 32 x 4096 x 6144 9.4 /
  // 10.0 against 11.1 / 11.5 us)
  bool want_tiled = M > 64 || (M > 16 && tiled_tiles >= (G % 128 == 0 ? 48 : 64)) || (M > 56 && K >= 10240) || (M > 32 && K >= 12288);
  // [r02, third audit: 21 layer shapes incl. Llama-2-13B / Qwen2 / Yi ones the rules above were not tuned on, M = 20..64,
  // scripts/gpu_planner_sweep_17_64.sh]  With G % 128 == 0 the rule is the four-tile skinny kernel's own geometry instead: it wins where
  // its workgroups (64 channels x 16 tokens each, K split in powers of two down to 8 stages) fit one round of 256 AND a workgroup's K
  // slice is at most 64 stages (4096 x 6144 M = 20..32: 7.6 against the tiled kernel's 10.3 us, 5120 x 5120 M <= 48: 8.6-9.2 against
  // 10.8-12.8, 8192 x 8192 M <= 32: 13.3 against 14.2); two rounds (5120 x 5120 M = 64: 16.0 against 13.0) or long slices
  // (13824 x 5120 M <= 32: 19.3-21.3 against 16.2, 20480 x 7168: 28.4 against 23.7) go to the tiled kernel.
  if (G % 128 == 0 && M > 16 && M <= 64 && N % 64 == 0) {
    const long wg = (long)(N / 64) * ((M + 15) / 16);
    int sk = 1;
    while (wg * sk * 2 <= 256 && KT / (sk * 2) >= 8) sk *= 2;
    want_tiled = !(wg <= 256 && (KT + sk - 1) / sk <= 64);
  }
  p.kernel = family == QUICK_KERNEL_AUTO ? (want_tiled ? QUICK_KERNEL_TILED : QUICK_KERNEL_SKINNY) : family;
  // Wide kernels (32x32x16 MFMA, one wave per SIMD, LDS-DMA) from 256 tokens, and from 64 tokens once there are enough
  // 64 x 128 tiles to cover the chip (large N): 13 % ahead of the r01 kernels on average over 45 (M, K, N) shapes between
  // 64 x 4096 x 12288 and 8192 x 4096 x 22016, never behind by more than 3 % [r02 probes, profiles/r02_planner_probe*.jsonl].
  // The tile (mb x 32 tokens, pairs x 128 channels) minimises a fitted launch-time model, see below.
  int wide_mb = 0, wide_pairs = 0, xk_auto_mb = 0;
  int xw_auto_mb = 0, xw_auto_pairs = 0, xw_auto_s = 0;   // [r04] the planner's own four-wave pick (w4a16_xw.hpp)
  bool wide_ring = false;
  // K slices of a wide launch: as many as keep the workgroups within one round of 256 and the slices >= 4 stages -- any count,
  // not only powers of two (80 tiles run 3 slices = 240 workgroups; Llama-2-13B's N = 5120 is 40 / 80 tiles wide)
  const auto wide_split = [KT](long tiles) { return (int)std::max<long>(1, std::min<long>(256 / std::max<long>(1, tiles), KT / 4)); };
  // [r02] ... or, with fewer tiles, once the K slices that fill the chip are still >= 32 stages long (64 x 28672 x 8192: 64 tiles
  // x 4 slices of 56 stages, 46.6 us against the tiled kernel's 54.6-58; M = 80 / 96 / 128 / 200 there: 70.6 / 72.4 / 73.8 / 126.9
  // against 78.4 / 80.0 / 79.3 / 150.8; at 16-22 stages per slice the r01 kernels stay ahead)
  const long wide_tiles64 = (long)((M + 63) / 64) * (N / 128);
  // (audit: from 24 stages -- 96 / 128 x 14336 x 4096, 4 slices of 28: 27.3 / 28.0 against 29.0 / 29.3 us; at 21 stages it is a toss-up,
  // 96 x 11008 x 4096 22.8 against 23.6 but 48 x 8192 x 10240 25.2 against 22.3 -- and from M = 33: 48 x 28672 x 8192 54.6 -> 45.5)
  // (after the store-path change [r02, second audit]: from 21 stages above 64 tokens -- 96 / 128 x 11008 x 4096 22.8 / 24.3 against 24.4 / 25.6)
  // (third audit, with any-count K splits: one token tile, M = 33..64, from 16 stages per slice while the tiles fit one round -- 8192 x
  // 8192 18.0 against 19.0 us, 18944 x 3584 20.6 against 22.3, 28672 x 8192 41.6 against 53, 4096 x 22016 21.3 against 22.2; at 4-11
  // stages per slice the reduction makes it lose 2-8 %, and so do two rounds of tiles, 8192 x 57344 82 against 74)
  const bool wide_long_k = M <= 64 ? (wide_tiles64 <= 256 && KT / std::max<long>(1, 256 / wide_tiles64) >= 16)
                                   : (wide_tiles64 <= 128 && KT / std::max<long>(1, 256 / wide_tiles64) >= 21);
  // ... except where 32-token tiles fit the token count exactly, fill one round and the 64- / 128-token tiles would run a quarter
  // empty (M = 65..96 on Llama-2-70B's qkv: 96 x 8192 x 10240 = 240 tiles of 32 x 128, 30.2 us against 38.6 [r02 audit])
  const long tiles32 = (long)((M + 31) / 32) * (N / 128);
  const bool exact32 = M < 128 && ((M + 63) / 64) * 64 - M >= 32 && tiles32 >= 208 && tiles32 <= 256 && !wide_long_k;
  // 65..256 tokens: the r01 tiled kernel's 32 x 128 and 64 x 128 tiles compete in the same model (its own K-split rule, coefficients
  // from all five audits); over the 152 audited shapes of that range the minimum is 1.2 % behind the best measured kernel on average,
  // 12 % at worst, where the separate rules were 2.1 % / 22 % (160 x 4096 x 6144: five exact 32-token tile rows, 17.7 against 21.4 us)
  const bool unified = family == QUICK_KERNEL_AUTO && G % 128 == 0 && M > 64 && M <= 256 && !mt_req && !waves_req;
  int model_tiled_mt = 0;
  if (family == QUICK_KERNEL_AUTO && G % 128 == 0 &&
      (unified || (!exact32 && ((M >= 64 && (M >= 256 || wide_tiles64 >= 96)) || (M > 32 && wide_long_k))))) {
    // (96 tiles of 64 x 128 [second audit]: 64 x 4096 x 12288 17.3 against 18.0 us, 96 / 128 x 4096 x 6144 16.8 / 17.2 against 17.4 / 18.0,
    // 192 x 4096 x 4096 16.9 against 17.9, 192 x 4096 x 6144 22.2 against 24.5; at 64-80 tiles the r01 kernels are level or ahead)
    // Launch-time model per kernel, fitted (relative least squares) to 198 shapes x 10 kernels, 64..8192 tokens on the model
    // layers [r02, scripts/gpu_planner_sweep_large.sh, two sessions averaged; residual 2.5-3.4 % rms = the session noise]:
    //   us = c + a n + stages (b_ceil n + b_frac f) [+ split0 + split1 * slices * tile MB],  n = ceil(f), f = workgroups / 256
    // -- a round of tiles costs what its fullest CU runs (b_ceil), and a partly filled chip runs its tiles faster (b_frac: clocks,
    // HBM share).  Picking the minimum is within 0.6 % of the best measured kernel on average over the 182 shapes routed here,
    // 7.5 % at worst (the r01 cost model after the store-path change: 4.4 % / 26 %).  The ring kernel only at 64 x 128 (one
    // workgroup per CU); 64 x 256 / 128-token ring variants and the 256 x 128 tile never won a shape in the sweep.
    struct WideCand { int mb, pairs; double c, a, b_ceil, b_frac, split0, split1; };
    static const WideCand cand[5] = {{2, 1, 1.31, 2.61, 0.3806, 0.2214, 2.03, 19.6},   // (fitted on four waves, 0.3985 / 0.2318; eight waves: x 0.955, the median over 337 rows)
                                      {2, 2, 6.34, -0.46, 0.4908, 0.5177, 2.13, 13.5},
                                     {4, 1, 6.08, -0.99, 0.4298, 0.5297, 2.54, 13.1}, {4, 2, -1.68, 7.41, 0.6715, 1.0087, 3.53, 10.6},
                                     {8, 2, -7.23, 15.76, 1.2716, 1.7451, 2.48, 58.9}};
    double best = 0;
    for (int c = 0; c < 5; ++c) {
      const int mb = cand[c].mb, pairs = cand[c].pairs;
      if (N % (pairs * 128) != 0) continue;
      const long T = (long)((M + mb * 32 - 1) / (mb * 32)) * (N / (pairs * 128));
      const int s = wide_split(T);
      const double f = (double)(T * s) / 256.0, n = (double)((T * s + 255) / 256), stages = (double)((KT + s - 1) / s);
      // (64 x 128 tiles in more than one round: the double-buffered kernel, 2-4 workgroups per CU, instead of the ring's one --
      // 256 x 4096 x 12288 33.8 against 42.3 us; its own fit over the 219 such rows of the audits, 2.1 % rms)
      const bool rounds64 = mb == 2 && pairs == 1 && T * s > 256;
      double cost = rounds64 ? 5.25 - 0.05 * n + stages * (0.2382 * n + 0.3242 * f)
                                   : cand[c].c + cand[c].a * n + stages * (cand[c].b_ceil * n + cand[c].b_frac * f) +
                                         (s > 1 ? cand[c].split0 + cand[c].split1 * s * (mb * 32.0 * pairs * 128 * 4 / 1e6) : 0.0) -
                                         (s == 2 && mb * pairs <= 8 ? 1.2 : 0.0);   // (two slices: the own partial stays in registers, measured after the fit)
      // [r03 audit, profiles/archive/r03_xk_audit.jsonl] tiles of <= 128 x 128 in SEVERAL rounds run 10-20 % behind this fit on the r03 boxes
      // (measured / predicted, medians: 64 x 128 1.13-1.21, 64 x 256 1.14-1.21, 128 x 128 1.22 against 1.12 in one round; 128 x 256
      // 1.04-1.10 and 256 x 256 1.01-1.05 as fitted): 640 x 5120 x 13824 121 us on 540 tiles against 101 on 162 of 256 x 256
      static const double several_rounds[5] = {1.2, 1.2, 1.2, 1.0, 1.0};  // (replayed on the audit's rows, tools/audit_replay.py: mean gap 0.95 -> 0.37 %, worst 19 -> 7 %)
      if (n >= 2.0) cost *= several_rounds[c];
      if (wide_mb == 0 || cost < best) {
        best = cost;
        wide_mb = mb;
        wide_pairs = pairs;
        wide_ring = mb == 2 && pairs == 1 && !rounds64;
      }
    }
    if (unified) {
      static const double tc[2][6] = {{5.968, -0.491, 0.263, 0.145, 0.697, 0.421}, {4.645, 1.535, 0.432, 0.195, 1.204, 0.711}};  // 32 / 64 tokens
      for (int c = 0; c < 2; ++c) {
        const int mt = c == 0 ? 2 : 4;
        const long T = (long)(N / 128) * ((M + mt * 16 - 1) / (mt * 16));
        const int nstage = (KT + 1) / 2;  // (eight waves: stages of 256 k)
        int s = 1;
        while (T * s * 2 <= 256 && nstage / (s * 2) >= 2) s *= 2;
        if (256 / T > s && nstage / (256 / T) >= 2) s = (int)(256 / T);
        s = std::max(1, std::min(s, nstage));
        const int kps = ((nstage + s - 1) / s) * 2, slices = (KT + kps - 1) / kps;
        const double f = (double)(T * slices) / 256.0, n = (double)((T * slices + 255) / 256);
        const double cost = tc[c][0] + tc[c][1] * n + kps * (tc[c][2] * n + tc[c][3] * f) + (slices > 1 ? tc[c][4] + tc[c][5] * slices : 0.0);
        if (wide_mb == 0 || cost < best) {
          best = cost;
          wide_mb = -1;  // (not a wide tile)
          model_tiled_mt = mt;
        }
      }
      if (wide_mb < 0) wide_mb = 0;
      else model_tiled_mt = 0;
    }
    // [r03] the exchange-K kernels (w4a16_xk.hpp) compete in the same model where their workgroups -- tiles x K slices, slices = the
    // largest power of two that still fits one round -- are co-resident.  Fitted (relative least squares, 246 / 336 rows of
    // scripts/gpu_xk_sweep.sh: 15 layer shapes x 24..1024 tokens, 4.4 / 3.7 % rms) and brought to this model's scale with the same
    // session's measurements of the planner's own wide picks (194 rows, predicted / measured = 0.945):
    //   us = c + stages (b0 + b1 f) + [s > 1] (s0 + s1 s) + d f,   f = workgroups / 256
    // 64-token tiles: the r02 ring tile with the weights in an AGPR queue and five x slots, 3-6 % ahead of it wherever both run
    // (512 x 4096 x 4096 22.6 against 23.8 us); 128-token tiles with 2 / 4 slices: 384 x 11008 x 4096 1.16x, 640 x 4096 x 4096 1.12x.
    if (best > 0 && allow_xk) {
      static const double xc[2][6] = {{1.728, 0.3869, 0.1572, 2.320, 0.1693, 2.064}, {3.422, 0.4881, 0.3847, 1.214, 0.4284, 2.382}};
      const int cus = exchange_cus();
      for (int c = 0; c < 2; ++c) {
        const int mb = c == 0 ? 2 : 4;
        const long T = (long)((M + mb * 32 - 1) / (mb * 32)) * (N / 128);
        int sx = 1;
        while (sx < 8 && T * sx * 2 <= cus && KT / (sx * 2) >= 4) sx *= 2;
        while (sx > 1 && (T * sx > cus || (KT + (KT + sx - 1) / sx - 1) / ((KT + sx - 1) / sx) != sx)) sx /= 2;
        if (T * sx > 256) continue;  // (several rounds: the 128 x 256 / 256 x 256 tiles of the wide family are ahead, 0.8x in the sweep)
        const double f = (double)(T * sx) / 256.0, stages = (double)((KT + sx - 1) / sx);
        // (128-token tiles + 4 %: where the two are close the 64-token tile was the better pick in the audit that followed,
        // 128 x 7168 x 7168 22.2 -> 21.0 us, 160 x 14336 x 4096 35.2 -> 32.7 [scripts/gpu_xk_audit.sh])
        const double cost = (xc[c][0] + stages * (xc[c][1] + xc[c][2] * f) + (sx > 1 ? xc[c][3] + xc[c][4] * sx : 0.0) + xc[c][5] * f) * (c == 1 ? 1.04 : 1.0);
        if (cost < best) {
          best = cost;
          xk_auto_mb = mb;
          wide_mb = 0;
          model_tiled_mt = 0;
        }
      }
    }
    // [r04] the four-wave kernels with generated hand-placed loops (w4a16_xw.hpp) -- 128 x 256, 128 x 128 and 64 x 128 tiles, 1 / 2 / 4 K
    // slices per tile.  Their own launch-time model (same form as above; fitted, relative least squares, to scripts/archive/r04_gpu_xw_sweep.sh:
    // 9 layer shapes x 12 token counts x 6 forced variants, 4.3 / 4.7 / 1.7 % rms) picks the tile and the slice count; WHETHER one of them
    // runs is decided on the sweep's own rows, against the pick of the rules above (profiles/r04_xw_sweep.jsonl, tools/xw_sweep_report.py):
    // they replace the 64- / 128-token wide tiles, the tiled kernel and the exchange-K tiles (6 % faster on the geometric mean of the 108
    // shapes, 13-24 % where the r02 128-token wide tile ran), not the 256 x 256 tile (level within 5 %, behind at 8192 tokens), and below
    // 160 tokens only from 96 tokens on wide layers (N >= 10240: 96 / 128 x 4096 x 11008 20.7 / 21.0 -> 17.8 / 18.6 us; at N = 4096 the 64-token
    // exchange-K tile stays 5-7 % ahead).
    // (where the 256 x 256 tile is the pick they compete only if it runs with a K split: 1024 x 28672 x 8192, 128 tiles x 2 slices, 419 us
    // against 379 on 256 tiles of 128 x 256 -- profiles/r04_xw256.txt)
    if (best > 0 && allow_xk && (wide_mb != 8 || (N % 256 == 0 && wide_split((long)((M + 255) / 256) * (N / 256)) > 1)) && (G / 128 & (G / 128 - 1)) == 0 &&
        M >= 65 &&   // [r06 audit at 80 tokens, profiles/r06_planner_audit.txt: from 65 tokens -- the 128 x 128 tile with its upper rows empty is 10-32 % ahead of the 64-token
                     // exchange tiles these counts fell to: 80 x 4096 x 22016 29.8 -> 25.4 us, x 13824 x 5120 31.6 -> 23.4, x 28672 x 8192 66.0 -> 55.8]   [r05 sweep, profiles/r05_mid_sweep.txt: from 96 tokens on the narrow layers as well -- 96 / 128 x 4096 x 4096 12.4 / 12.6 -> 11.4 / 11.6 us, x 5120 x 5120 14.8 / 16.9 -> 13.6 / 13.9, x 8192 x 8192 27.9 / 29.1 -> 25.0 / 26.0, x 13824 x 5120 29.5 -> 25.4; at 33..64 tokens no four-wave tile beats the r03 picks by more than the session noise]

        (size_t)M * (size_t)K * 2 < ((size_t)1 << 32) && (size_t)M * (size_t)N * 2 < ((size_t)1 << 32)) {
      struct XwCand { int mb, pairs; double c, a, b_ceil, b_frac, s0, s1, d; };
      static const XwCand xwc[3] = {{4, 2, -2.8632, 5.8843, 0.8799, 0.6018, -0.6402, 1.2461, 5.8244},
                                    {4, 1, -0.3015, 1.9468, 0.5116, 0.3473, -0.0715, 1.0316, 4.3280},
                                    {2, 1, 1.8009, 3.7953, 0.3277, 0.2559, 0.1770, 0.3541, 0.0325}};
      double xbest = 0;
      for (int c = 0; c < 3; ++c) {
        const int mb = xwc[c].mb, pairs = xwc[c].pairs;
        if (N % (pairs * 128) != 0) continue;
        const long T = (long)((M + mb * 32 - 1) / (mb * 32)) * (N / (pairs * 128));
        for (int sx = 1; sx <= mb && sx <= 4; sx *= 2) {
          if (sx > 1 && (T * sx > std::min(256, exchange_cus()) || KT / sx < 4 || (KT + (KT + sx - 1) / sx - 1) / ((KT + sx - 1) / sx) != sx)) continue;
          const double f = (double)(T * sx) / 256.0, n = (double)((T * sx + 255) / 256), stages = (double)((KT + sx - 1) / sx);
          // (- 0.5 us with slices: the exchange lost its atomics after the fit -- 512 x 4096 x 4096 22.0 -> 21.5 / 24.2 -> 23.3 us on two / four
          // slices, profiles/r04_ab_exchange.txt)
          const double cost = xwc[c].c + xwc[c].a * n + stages * (xwc[c].b_ceil * n + xwc[c].b_frac * f) + (sx > 1 ? xwc[c].s0 + xwc[c].s1 * sx - 0.5 : 0.0) + xwc[c].d * f;
          if (xw_auto_mb == 0 || cost < xbest) {
            xbest = cost;
            p.est_xw_us = cost;
            xw_auto_mb = mb;
            xw_auto_pairs = pairs;
            xw_auto_s = sx;
          }
        }
      }
      // [r06, three-box audit of the final tree, profiles/r06_planner_audit.txt] 129..256 tokens on layers where FOUR slices of the 128 x 128 tile are exactly one round
      // (N = 4096: 2 x 32 tiles x 4): the fit prefers two slices of the 64 x 128 tile there, and every box measured the 128 x 128 tile ahead -- 160 / 192 / 256 x 4096 x 4096
      // 13.5-15.2 -> 12.6-14.0 us (4-7 %), 160 / 192 x 11008 x 4096 27.2-28.5 -> 24.1-26.7 (7-10 %).  (N = 5120: 320 workgroups, the 64-token tile stays ahead.)
      if (xw_auto_mb == 2 && xw_auto_pairs == 1 && xw_auto_s == 2 && M > 128 && M <= 256 && N % 128 == 0) {
        const long T41 = (long)((M + 127) / 128) * (N / 128);
        if (T41 * 4 <= std::min(256, exchange_cus()) && KT / 4 >= 4 && (KT + (KT + 3) / 4 - 1) / ((KT + 3) / 4) == 4) {
          xw_auto_mb = 4;
          xw_auto_pairs = 1;
          xw_auto_s = 4;
        }
      }
    }
    // [r04] where the r02 model's pick is the 256 x 256 tile with ONE K slice, the four-wave kernel with the generated 256 x 256 loop runs it
    // instead: 0.935-0.965 of the hipcc-scheduled kernel's time on 28 prefill shapes of 1024..8192 tokens in one session, bit-identical
    // results (scripts/archive/r04_gpu_xw82.sh, profiles/r04_xw256_sweep.jsonl; 4096^3 117.3 -> 111.6 us, 8192 x 4096 x 22016 1232 -> 1170).
    // QUICK_AMD_XW256=0 keeps r02's kernel (the A/B switch; since r06 that kernel is only in QUICK_AMD_TOOLS builds -- it spills 92 bytes per lane --
    // and the product answers the switch with 128 x 256 tiles).
    if (best > 0 && wide_mb == 8 && !xw_auto_mb && !xk_auto_mb && N % 256 == 0 && (G / 128 & (G / 128 - 1)) == 0 && KT >= 2 &&
        wide_split((long)((M + 255) / 256) * (N / 256)) == 1 &&
        (size_t)M * (size_t)K * 2 < ((size_t)1 << 32) && (size_t)M * (size_t)N * 2 < ((size_t)1 << 32)) {
      const char* e = getenv("QUICK_AMD_XW256");
      if (!e || atoi(e) != 0) {
        xw_auto_mb = 8;
        xw_auto_pairs = 2;
        xw_auto_s = 1;
        p.est_xw_us = 0.95 * best;
      }
    }
    p.est_us = best;
    // ... unless the models say the r02 / r03 candidate is clearly ahead: on the audit's 133 four-wave picks (scripts/archive/r04_gpu_audit_vs_r03.sh, this
    // tree against r03's library in one session) the r03 model reads 0.95x and the four-wave model 1.03x the measured time, and "four-wave
    // iff its estimate < 1.10 x the other" is the best threshold (geometric mean 0.904 against 0.905 for always; it returns 320 x 4096 x 12288 /
    // 22016 -- 430 tiles of 64 x 256 against 516 of 128 x 128, a nearly empty third round -- to the r02 ring kernel: 43.5 -> 39.6 us, 82.9 -> 75.0)
    if (xw_auto_mb && p.est_xw_us >= 1.10 * best) xw_auto_mb = 0;
    if (xw_auto_mb) p.kernel = QUICK_KERNEL_XW;
    else if (xk_auto_mb) p.kernel = QUICK_KERNEL_XK;
    else if (wide_mb) p.kernel = QUICK_KERNEL_WIDE;
    else if (model_tiled_mt) p.kernel = QUICK_KERNEL_TILED;
  }
  // [r03] 33..64 tokens where the rules above leave the 32-token tiled kernel with a K split: one 64-token exchange-K tile per CU
  // instead, when its slices are long (>= 8 stages) and cover most of the chip -- Mistral's fused GQA qkv, 4096 x 6144, at 48 / 64
  // tokens 12.8 / 13.0 -> 10.6 / 10.8 us, 64 x 5120 x 5120 13.5 -> 11.6; with eight slices, 40..64 x 11008 / 14336 x 4096 (the down
  // projections at batch 40..64): 16.6-19.8 -> 14.6-16.4 us in two sessions of three, 3 % behind in the third
  // [scripts/gpu_xk_sweep.sh, gpu_xk_audit.sh]
  if (allow_xk && family == QUICK_KERNEL_AUTO && p.kernel == QUICK_KERNEL_TILED && G % 128 == 0 && M > 32 && M <= 64 && !mt_req && !waves_req) {
    const long T = N / 128;
    int sx = 1;
    while (sx < 8 && T * sx * 2 <= std::min(256, exchange_cus()) && KT / (sx * 2) >= 4) sx *= 2;
    while (sx > 1 && (KT + (KT + sx - 1) / sx - 1) / ((KT + sx - 1) / sx) != sx) sx /= 2;
    if (T * sx >= 160 && T * sx <= cu_count() && KT / sx >= 8) {
      p.kernel = QUICK_KERNEL_XK;
      xk_auto_mb = 2;
    }
  }
  // [r05 audit, profiles/r05_planner_audit.txt / r05_mid_sweep.txt: AUTO against forced launches of every family, 234 + 65 shapes, two sessions]
  // two corrections where a four-wave tile was consistently ahead of the models' pick:
  //  * 33..95 tokens, 160..256 workgroups of 64 x 128 tiles x TWO slices (N = 12288, 13824, 10240): 33 / 48 / 64 x 4096 x 12288 13.8 / 14.9 / 15.1 ->
  //    12.5 / 13.4 / 13.7 us, 48 / 64 x 8192 x 10240 26.6 / 25.4 -> 21.8 / 22.2, 64 x 5120 x 13824 21.0 -> 18.7;
  //  * 96..128 tokens where ONE row of 128 x 128 tiles x FOUR slices fits a round (N <= 8192): 96 / 128 x 4096 x 4096 12.0 / 12.1 -> 11.0 / 11.0,
  //    x 4096 x 6144 13.7 / 14.0 -> 11.7 / 12.0, x 5120 x 5120 14.8 / 16.9 -> 13.6 / 13.9
  // Final tree, two boxes (tools/audit_auto_vs_forced.py): geometric mean AUTO / best forced 1.0013, worst 1.048 (96 x 8192 x 8192, where two
  // token tiles of 64 x 128 x two slices are ahead of the four-slice 128 x 128 tile; a rule for it was tried and lost 18 % at 96 / 128 x
  // 28672 x 8192 -- the 64-token loop is the slower one per MFMA, only its exchange is cheaper -- so it stays a 5 % gap).
  if (allow_xk && family == QUICK_KERNEL_AUTO && G % 128 == 0 && ((G / 128) & (G / 128 - 1)) == 0 && !mt_req && !waves_req && grid_split_k <= 1 &&
      (size_t)M * (size_t)K * 2 < ((size_t)1 << 32) && (size_t)M * (size_t)N * 2 < ((size_t)1 << 32)) {
    const long t64 = (long)((M + 63) / 64) * (N / 128), t128 = (long)(N / 128);
    const auto splits_evenly = [KT](int sx) { return KT / sx >= 4 && (KT + (KT + sx - 1) / sx - 1) / ((KT + sx - 1) / sx) == sx; };
    const long xcus = std::min(256, exchange_cus());   // (QUICK_AMD_EXCHANGE_CUS: how many CUs the K slices of a tile may count on)
    if (M >= 33 && M <= 64 && t64 * 2 >= 160 && t64 * 2 <= xcus && splits_evenly(2)) {   // (r06: up to 64 tokens; above, the 128 x 128 tile -- see the four-wave rule)
      p.kernel = QUICK_KERNEL_XW;
      xw_auto_mb = 2;
      xw_auto_pairs = 1;
      xw_auto_s = 2;
    } else if (M >= 65 && M <= 128 && t128 * 4 <= xcus && t128 * 4 >= 96 && splits_evenly(4)) {
      p.kernel = QUICK_KERNEL_XW;
      xw_auto_mb = 4;
      xw_auto_pairs = 1;
      xw_auto_s = 4;
    }
  }
  int ks = 1;
  if ((p.kernel == QUICK_KERNEL_WIDE || p.kernel == QUICK_KERNEL_XK) && G % 128 != 0) p.kernel = QUICK_KERNEL_TILED;  // small groups: r01's tiled kernel
  // x travels through a buffer descriptor with 32-bit offsets in the wide / exchange-K kernels: from 4 GiB of activations on, r01's
  // tiled kernel (64-bit pointers) runs instead
  if ((p.kernel == QUICK_KERNEL_WIDE || p.kernel == QUICK_KERNEL_XK) && (size_t)M * (size_t)K * 2 >= ((size_t)1 << 32)) p.kernel = QUICK_KERNEL_TILED;
  if (p.kernel == QUICK_KERNEL_XK && (size_t)M * (size_t)N * 2 >= ((size_t)1 << 32)) p.kernel = QUICK_KERNEL_TILED;  // (y through a buffer descriptor as well)
  // (the tile the id asks for, worked out BEFORE the width check: a forced id with bit 12 or two token blocks and a block count of 8 means
  // 256-channel tiles all the same, and N % 256 == 128 would leave the last 128 channels unwritten -- ADVICE r04)
  const int xw_mb = xw_auto_mb ? xw_auto_mb : (mt_req == 2 ? 2 : (mt_req == 8 ? 8 : 4));
  const int xw_pairs = xw_auto_mb ? xw_auto_pairs : (xw_mb == 8 ? 2 : ((xw_mb == 2 || no_xlds) ? 1 : 2));
  if (p.kernel == QUICK_KERNEL_XW && (G % 128 != 0 || ((G / 128) & (G / 128 - 1)) != 0 || N % (xw_pairs * 128) != 0 || (size_t)M * (size_t)K * 2 >= ((size_t)1 << 32) ||
                                      (size_t)M * (size_t)N * 2 >= ((size_t)1 << 32)))
    p.kernel = QUICK_KERNEL_TILED;  // (the loop shifts the k tile by log2(G / 128); whole channel tiles; 32-bit buffer offsets)
  if (p.kernel == QUICK_KERNEL_XW) {
    // Four waves, one per SIMD, hand-placed K loops (w4a16_xw.hpp): tiles of 128 x 256, 128 x 128 or 64 x 128; S = 1, 2, 4 K slices per
    // tile on S compute units.  Nobody has to be co-resident (a wave that waits too long gives its block up, the last partner finishes it),
    // but the exchange zone holds S * S boxes of (a tile's fp16 image / S) per tile: workgroups <= 256 with S > 1.
    // bits 4-7: 32-token blocks per tile (2, 4; 0 = 4); bit 12: 128-channel tiles (implied by 2 blocks); bits 8-11: S (0 = as many as fit
    // the CUs); bits 22-26: log2 of the poll limit in ticks of 10 ns (tests: 1 = every wave gives up at once).
    // (bits 4-7 = 8: the 256 x 256 tile -- waves of 256 tokens x 64 channels, a ring of two 64 KiB slots, one slice only)
    const int mb = xw_mb, pairs = xw_pairs;
    const int MBk = (M + mb * 32 - 1) / (mb * 32), NBk = N / (pairs * 128);
    p.wide_mb = mb;
    p.wide_pairs = pairs;
    p.tch = pairs * 128;
    p.waves = 4;
    p.ntiles = MBk * NBk;
    const int s_req = xw_auto_s ? xw_auto_s : (grid_split_k > 0 ? grid_split_k : (kernel >> 8) & 15);
    int s = 1;
    if (mb == 8) s = 1;
    else if (s_req == 1 || s_req == 2 || (s_req == 4 && mb == 4)) s = s_req;
    else
      while (s < mb && s < 4 && (long)p.ntiles * s * 2 <= cu_count() && KT / (s * 2) >= 4) s *= 2;
    while (s > 1 && ((long)p.ntiles * s > 256 || (KT + (KT + s - 1) / s - 1) / ((KT + s - 1) / s) != s)) s /= 2;
    p.ksplit = s;
    p.kt_per_split = (KT + s - 1) / s;
    p.poll_log2 = (kernel >> 22) & 31;
    if (const char* e = getenv("QUICK_AMD_EXCHANGE_POLL_LOG2")) p.poll_log2 = std::max(0, std::min(31, atoi(e)));
    const int groups = 8 / s;  // XCDs per K slice: they form a gm x gn grid over the (token, channel) tiles
    long best = -1;
    if (!((kernel >> 14) & 1) && ((long)p.ntiles * s) % 8 == 0)
      for (int gm = 1; gm <= groups; gm *= 2) {
        if (MBk % gm != 0 || NBk % (groups / gm) != 0) continue;
        const long cost = (long)(MBk / gm) * 64 * mb + (long)(NBk * gm / groups) * 64 * pairs;  // bytes per k and XCD: x rows (2 B) + weight columns (1/2 B)
        if (best < 0 || cost < best) {
          best = cost;
          p.xcd_gm = gm;
        }
      }
  } else if (p.kernel == QUICK_KERNEL_XK) {
    // exchange-K kernels (w4a16_xk.hpp): tile = mb * 32 tokens x 128 channels, eight waves; the S slices of a tile run on S compute
    // units at the same time and swap parts of their partial tiles, so S > 1 needs the whole grid co-resident: tiles * S <= CUs.
    // bits 4-7: mb (2, 4; 0 = by M), bits 8-11: S (1, 2, 4, 8; 0 = as many as fit), bits 22-24: x ring slots, bits 26-28: weight queue depth
    const int mb = xk_auto_mb ? xk_auto_mb : ((mt_req == 2 || mt_req == 4) ? mt_req : (M > 64 ? 4 : 2));
    const int MBk = (M + mb * 32 - 1) / (mb * 32), NBk = N / 128;
    const int cus = exchange_cus();
    p.wide_mb = mb;
    p.wide_pairs = 1;
    p.tch = 128;
    p.waves = 8;
    p.ntiles = MBk * NBk;
    const int s_req = grid_split_k > 0 ? grid_split_k : (kernel >> 8) & 15;
    int s = 1;
    if (s_req == 1 || s_req == 2 || s_req == 4 || s_req == 8) s = s_req;
    else {
      while (s < 8 && (long)p.ntiles * s * 2 <= cus && KT / (s * 2) >= 4) s *= 2;
      if (s_req == 15 && s > 1) s /= 2;  // (tuning sweeps: half the count the rule gives)
    }
    // (the slices must fit the chip, and ceil-dividing K must give exactly S non-empty slices)
    // 64-token tiles with K slices need <= 128 registers and 80 KiB of LDS: TWO workgroups are resident per CU (four waves per SIMD)
    // (tools builds, forced slice counts: measured slower than one slice on one CU -- 512 x 4096 x 4096 25.8 against 22.8 us)
#ifdef QUICK_AMD_TOOLS
    const long cap = mb == 2 ? 2L * cus : cus;
#else
    const long cap = cus;
#endif
    while (s > 1 && ((long)p.ntiles * s > ((s_req == 2 || s_req == 4 || s_req == 8) ? cap : (long)cus) || (KT + (KT + s - 1) / s - 1) / ((KT + s - 1) / s) != s)) s /= 2;
    p.ksplit = s;
    p.kt_per_split = (KT + s - 1) / s;
    const int nb_req = (kernel >> 22) & 7, wd_req = (kernel >> 26) & 7;
    p.xk_nbuf = nb_req >= 3 ? nb_req : 5;
    p.xk_wd = wd_req >= 3 ? wd_req : 4;
    p.xk_loader = ((kernel >> 12) & 1) != 0;
    // sixteen waves (bit 13): 64-token tiles with ONE slice whose K range is a whole number of 256-k stages
    p.xk_kq = (((kernel >> 13) & 1) && mb == 2 && s == 1 && KT % 2 == 0 && KT >= 4 && p.xk_nbuf == 5 && p.xk_wd == 4 && !p.xk_loader) ? 4 : 2;
    const int groups = 8 / s;  // XCDs per K slice: they form a gm x gn grid over the (token, channel) tiles
    long best = -1;
    if (!((kernel >> 14) & 1) && ((long)p.ntiles * s) % 8 == 0)
      for (int gm = 1; gm <= groups; gm *= 2) {
        if (MBk % gm != 0 || NBk % (groups / gm) != 0) continue;
        const long cost = (long)(MBk / gm) * 64 * mb + (long)(NBk * gm / groups) * 64;  // bytes per k and XCD: x rows (2 B) + weight columns (1/2 B)
        if (best < 0 || cost < best) {
          best = cost;
          p.xcd_gm = gm;
        }
      }
  } else if (p.kernel == QUICK_KERNEL_WIDE) {
    // bits 4-7: MB (token tiles of 32 per workgroup: 2, 4, 8), bits 8-11: PAIRS (32-channel pairs per wave: 1, 2); 0 = choose
    int mb = wide_mb ? wide_mb : mt_req, pairs = wide_mb ? wide_pairs : (kernel >> 8) & 15;
    if (mb != 2 && mb != 4 && mb != 8) mb = M > 128 ? 8 : (M > 64 ? 4 : 2);
    if (pairs != 1 && pairs != 2) pairs = 2;
    if (N % (pairs * 128) != 0) pairs = 1;
#ifndef QUICK_AMD_TOOLS
    if (mb == 8 && pairs == 2) mb = 4;   // [r06] no register-spilling kernel in the product: where the generated 256 x 256 loop cannot take a launch (K split, 64-bit offsets,
                                         // G not a power-of-two multiple of 128) 128 x 256 tiles run instead of r02's hipcc-scheduled 256 x 256 tile
#endif
    p.wide_mb = mb;
    p.wide_pairs = pairs;
    // bit 12: no ring (the double-buffered kernel at every tile size); bits 22-24: ring slots (0 = as many as fit, up to 6);
    // bit 15: eight waves per workgroup (two per SIMD, k16 steps split by parity) -- ring kernel, tiles up to 128 x 128 / 64 x 256
    // The planner's own ring launches (64 x 128 tiles) run the eight-wave variant with four slots: since the whole workgroup shares
    // the way out (§5.3 of DESIGN.md) it is 2-5 % ahead of four waves / six slots on every shape probed, warm against warm in one
    // session -- 512 x 4096 x 4096 24.04 against 24.68 us, 128 x 4096 x 12288 21.5 / 22.6, 64 x 4096 x 22016 20.8 / 21.7, 512 x 11008 x 4096
    // 57.8 / 59.2, 448 x 4096 x 4096 22.7 / 23.4; four slots are level with or 1 % ahead of six [r02].
    const bool auto_ring = wide_mb && wide_ring;
    const bool eight = (((kernel >> 15) & 1) || auto_ring) && mb * pairs <= 4 && !no_xlds;
    const int nb_req = auto_ring ? 4 : (kernel >> 22) & 7, nb_max = mb == 8 ? 0 : wide_ring_nbuf(mb, pairs, eight ? 2 : 1);
    p.wide_nbuf = (no_xlds || nb_max < 3 || (wide_mb && !wide_ring)) ? 0 : (nb_req >= 3 ? std::min(nb_req, nb_max) : nb_max);
    p.waves = (eight && p.wide_nbuf >= 3) ? 8 : 4;
    p.tch = pairs * 128;
    const int MBk = (M + mb * 32 - 1) / (mb * 32), NBk = N / p.tch;
    p.ntiles = MBk * NBk;
    p.slab_floats = (size_t)mb * 32 * p.tch;
    ks = wide_split(p.ntiles);
    p.ksplit = std::max(1, std::min(grid_split_k > 0 ? grid_split_k : ks, KT));
    p.kt_per_split = (KT + p.ksplit - 1) / p.ksplit;
    long best = -1;
    if (!((kernel >> 14) & 1) && (MBk * NBk) % 8 == 0)
      for (int gm = 1; gm <= 8; gm *= 2) {
        if (MBk % gm != 0 || NBk % (8 / gm) != 0) continue;
        const long cost = (long)(MBk / gm) * 64 * mb + (long)(NBk * gm / 8) * (p.tch / 2);  // bytes per k: x rows (2 B) + weight columns (1/2 B) per XCD
        if (best < 0 || cost < best) {
          best = cost;
          p.xcd_gm = gm;
        }
      }
  } else if (p.kernel == QUICK_KERNEL_SKINNY) {
    const int mblocks = (M + 15) / 16;
    // channel tiles per workgroup ~ token blocks (weights are re-read per token block, x per channel block)
    // (two token blocks: four tiles as well with G % 128 == 0 -- never behind two in the third audit, 32 x 5120 x 5120 15.8 -> 9.1 us)
    int mt_auto = mblocks <= 1 ? 1 : (mblocks <= 2 && G % 128 != 0 ? 2 : 4);
    // M = 9..16 on a large layer: the fragments-from-L2 path reads x once per 16-channel tile -- 4x the weight bytes at
    // M = 16 -- so share every x fragment among 4 channel tiles (8192 x 57344 at M = 16: 119 -> 78 us, 28672 x 8192:
    // 65 -> 36 us [r01]).  Small layers keep NTW = 1 (more workgroups); from 1024 channel blocks the deferred-zero path
    // (x in LDS) is as fast or faster and stays.
    if (!mt_req && M > 8 && M <= 32 && (long)K * N >= 40L * 1000 * 1000) {  // (M = 17..32: 11008 x 4096 16.6 -> 13.5 us)
      // (the table flavour up to M = 12: at M = 16 the 4-tile fragment flavour is ahead, 4096 x 22016 18.3 -> 16.9 us, 4096 x
      // 28672 20.3 -> 19.3 [r02 audit])
      // [r03 audit on the final kernels, scripts/gpu_small_audit2.sh: as a bare GEMM the table flavour is ahead at M = 16 as well up to
      // ~120 M weights -- 4096 x 22016 17.0 -> 14.8 us, 4096 x 28672 18.2 -> 16.8; 8192 x 57344 keeps the four tiles.]
      // Not taken: these layers are the gate_up projections, which carry the RMSNorm prologue in a decode step, and the table flavour
      // normalises its 16 rows one after the other -- Llama-2-7B / Mistral-7B at batch 16 lost 5-8 % tok/s with it [bench_decode.py].
      const bool dz_ok = mblocks == 1 && M <= 12 && !no_xlds && !((kernel >> 25) & 1) && N / 16 >= 1024 &&
                         skinny_lds_bytes(M, G, 1, 8, KT, true, true, true) <= kLdsPerCu;
      if (!dz_ok) mt_auto = 4;
    } else if (!mt_req && M >= 5 && M <= 8 && N >= 8192 && N / 16 < 1024 && (G % 128 == 0 || (M >= 6 && (size_t)M * (K * 2 + 16) > (size_t)64 * 1024))) {
      // (r03: with G % 128 == 0 from M = 5 and at any K -- the four-tile kernel then takes its fragments from L2, below: 4096 x 12288
      // M = 5, 6 8.4-8.6 -> 7.6 us)
      // (only where the 4-tile kernel takes its fragments straight from L2: its LDS-copy flavour is slower, M = 6, 7 at K = 4096)
      mt_auto = 4;  // at M = 8: 4096 x 12288 11.0 -> 10.3 us, 8192 x 10240 18.7 -> 16.6 us, 28672 x 8192 49 -> 34 us; from
    }               // 1024 blocks (4096 x 22016, 8192 x 57344) the deferred-zero path stays ahead, and so it does at M <= 4
    // 257..512 channel blocks (N = 6144, 8192) at M = 4..16: one-tile workgroups come in 1.5 or 2 per CU -- two tiles per
    // workgroup are one round of <= 256 and share every x fragment [r02 audit: 4096 x 6144 M = 4 / 8 / 16 7.6 / 8.2 / 10.9 -> 6.4 /
    // 6.3 / 7.6 us, 8192 x 8192 M = 6 14.7 -> 10.1, 28672 x 8192 M = 3, 4 33.5 -> 25.4]; large layers keep the 4 tiles from M = 9
    if (!mt_req && mblocks == 1 && N / 16 > 256 && N / 16 <= 512 && G % 128 == 0) {
      if (KT >= 128) mt_auto = M >= 3 ? 4 : mt_auto;  // (a very long K: four tiles and a 2-way K split, 28672 x 8192 M = 3..8 24.8-25.1 us)
      // [r03 audit, scripts/gpu_small_audit2.sh] up to 320 blocks (N = 5120) four tiles from seven tokens (5120 x 5120 M = 8 .. 16 8.0-9.5 ->
      // 7.6-8.5 us); above, two tiles up to twelve tokens (8192 x 8192 M = 10 11.8 -> 11.0)
      else if (N / 16 <= 320) mt_auto = M >= 7 ? 4 : (M >= 4 ? 2 : mt_auto);
      // (4096 x 6144 keeps two tiles at 13..16 tokens although four with a two-way K split are 2-5 % ahead: the split would cost the
      // RMSNorm prologue of Mistral's fused qkv a launch)
      else if (M >= 4 && (M <= 12 || (long)K * N < 40L * 1000 * 1000)) mt_auto = 2;
    }
    // M = 7, 8 with a long K (>= 10240: the x copy is far beyond the 64 KiB LDS budget) where the exact path would take its fragments
    // from L2 once per 16 channels: two tiles per workgroup share them (8 x 11008 x 4096 10.0 -> 9.2 us, 14336 x 4096 12.2 -> 11.4,
    // 18944 x 3584 14.7 -> 13.1; at M = 6 it is a tie) [audits]
    // (r03 audit: four tiles, 8 x 11008 x 4096 9.3 -> 8.8 us, 8 x 14336 x 4096 11.4 -> 10.3)
    if (!mt_req && mt_auto == 1 && M >= 7 && M <= 8 && N / 16 <= 256 && G % 128 == 0 && K >= 10240) mt_auto = 4;
    p.mt = mt_req ? mt_req : mt_auto;
    if (p.mt != 1 && p.mt != 2) p.mt = 4;
    while (p.mt > 1 && (N / 16) % p.mt != 0) p.mt /= 2;
    p.waves = (waves_req == 4 || waves_req == 8 || waves_req == 16) ? waves_req : 8;
    p.ntiles = (N / (16 * p.mt)) * mblocks;
    p.slab_floats = (size_t)p.mt * 256;
    // fill the 256 CUs when N is small: every workgroup should still own >= 8 k-tiles
    while (p.ntiles * ks * 2 <= 256 && KT / (ks * 2) >= 8) ks *= 2;
    p.ksplit = std::max(1, std::min(grid_split_k > 0 ? grid_split_k : ks, KT));
    p.kt_per_split = (KT + p.ksplit - 1) / p.ksplit;
  } else {
    p.mt = mt_req ? (mt_req == 2 ? 2 : (mt_req == 8 ? 8 : 4)) : (model_tiled_mt ? model_tiled_mt : (M <= 32 ? 2 : 4));
    p.waves = waves_req == 16 ? 16 : 8;
    int wk = p.wn2 ? 4 : p.waves / 4;
    // 32-token tiles once 64-token tiles would need a 4-way K split to cover the CUs (twice the tiles, half the slices
    // to reduce): M = 65..128 at N = 4096, 15.5 us instead of 16.5 us at M = 128 [r01]
    if (!mt_req && !model_tiled_mt && p.mt == 4 && (N / 128) * ((M + 63) / 64) * 4 <= 256 && (KT + wk - 1) / wk >= 8) p.mt = 2;
    // ... and at K <= 4096 already where they would need a 2-way split: twice the tiles and nothing to reduce (64 x 4096 x 12288
    // 18.1 -> 17.0 us, 192 x 4096 x 4096 18.2 -> 17.1, 96 / 128 x 4096 x 6144 17.9 -> 16.9 [r02 audit]; with a longer K the split
    // tiles stay ahead: 48 x 8192 x 10240 22.3 against 29.4)
    if (!mt_req && !model_tiled_mt && p.mt == 4 && exact32 && !((kernel >> 27) & 1) && !((kernel >> 29) & 1)) p.mt = 2;  // (see exact32 above)
    if (!mt_req && !model_tiled_mt && p.mt == 4 && K <= 4096 && !((kernel >> 27) & 1) && !((kernel >> 29) & 1) && (N / 128) * ((M + 63) / 64) * 2 <= 256 &&
        (N / 128) * ((M + 31) / 32) <= 256)
      p.mt = 2;
    // 256-channel tiles (4 channel tiles per wave: one LDS fragment read per four MFMAs instead of two, 2/3 of the L2 -> CU
    // bytes per MAC): M = 1024 at N = 4096 59 -> 50 us, M = 8192 x 22016 772 -> 845 TFLOP/s [r01].  Whether they pay is a
    // matter of how the tiles quantise onto the 256 CUs.  In units of one 128-channel tile alone on its CU: two
    // co-resident 128-channel tiles take ~1.65, a 256-channel tile ~1.5; W wide tiles run in ceil(W / 256) rounds, the
    // 2W narrow ones in full rounds of 512 plus a remainder that runs one or two per CU.  Up to 128 wide tiles they
    // would need a K split and lose (M = 512 at N = 4096: 33.7 against 31.1 us); 129..256 win by 8-10 % (M = 576..960 at
    // N = 4096, 256 x 12288, 128 x 22016, M = 384..512 at K = N = 8192); just above a multiple of 256 the second, nearly
    // empty round loses (192 x 22016: 80 against 70 us; 512 x 11008: 84 against 75) [r01 sweep, one session].
    // Kernel bit 29 forces them, bit 30 forbids them.
    const bool wide_ok = p.mt == 4 && p.waves == 8 && !p.wn2 && !p.ablate && N % 256 == 0 && !mt_req && !model_tiled_mt;
    const long wtiles = (long)(N / 256) * ((M + 63) / 64);
    const long nfull = 2 * wtiles / 512, nrem = 2 * wtiles % 512;
    const double narrow_cost = 1.65 * nfull + (nrem == 0 ? 0.0 : (nrem <= 256 ? 1.0 : 1.65));
    const double wide_cost = 1.5 * ((wtiles + 255) / 256);
    const bool wide = ((kernel >> 29) & 1) || (!((kernel >> 30) & 1) && wtiles > 128 && wide_cost < narrow_cost);
    p.tch = wide_ok && wide ? 256 : 128;
    // 128 x 256 tiles run by FOUR waves of 128 tokens x 64 channels each (128 accumulators per lane: at 256 threads hipcc
    // hands out AGPRs, 396 registers, one workgroup per CU): 1.6 VALU and 0.25 LDS fragment reads per MFMA, half the
    // L2 -> CU bytes per MAC of the 64 x 128 tile.  A round of them costs ~1.87 rounds of 64 x 256 tiles for twice the
    // work [r01: M = 2048 at N = 4096 91.6 -> 85.6 us, 8192 x 4096 x 22016 1758 -> 1580 us, 4096 x 8192 x 8192 587 -> 534 us
    // = 1.03 PFLOP/s], so they win where their rounds quantise no worse.  Kernel bit 27 forces them, bit 30 forbids them.
    const long btiles = (long)(N / 256) * ((M + 127) / 128);
    const double big_cost = 1.5 * 1.87 * ((btiles + 255) / 256);
    const bool big = ((kernel >> 27) & 1) || (!((kernel >> 30) & 1) && !((kernel >> 29) & 1) && btiles > 128 &&
                                              big_cost < std::min(narrow_cost, wide_cost));
    if (wide_ok && big) {
      p.tch = 256;
      p.mt = 8;
      p.waves = 4;
      wk = 1;  // 4 waves along N, stages of 128 k
    }
    p.ntiles = (N / p.tch) * ((M + p.mt * 16 - 1) / (p.mt * 16));
    p.slab_floats = (size_t)(p.tch / 16) * p.mt * 256;
    // one workgroup per CU: split K until the 256 CUs are covered, keeping >= 2 stages per slice
    const int nstage = (KT + wk - 1) / wk;
    while (p.ntiles * ks * 2 <= 256 && nstage / (ks * 2) >= 2) ks *= 2;
    // ... and not only in powers of two: 80 tiles (Llama-2-70B's qkv, N = 10240) run 3 slices = 240 workgroups,
    // 25.8 us against 29.7 us with 2 at M = 64 [r01]
    if (256 / p.ntiles > ks && nstage / (256 / p.ntiles) >= 2) ks = 256 / p.ntiles;
    p.ksplit = std::max(1, std::min(grid_split_k > 0 ? grid_split_k : ks, nstage));
    p.kt_per_split = ((nstage + p.ksplit - 1) / p.ksplit) * wk;  // whole stages
    // XCD-aware tile order: minimise what each XCD's L2 has to fetch, (MB/gm) token blocks of 4*BMT*K bytes plus
    // (NB*gm/8) channel blocks of 64*K bytes
    const int MBk = (M + p.mt * 16 - 1) / (p.mt * 16), NBk = N / p.tch;
    long best = -1;
    if (!((kernel >> 14) & 1) && (MBk * NBk) % 8 == 0)
      for (int gm = 1; gm <= 8; gm *= 2) {
        if (MBk % gm != 0 || NBk % (8 / gm) != 0) continue;
        const long cost = (long)(MBk / gm) * 4 * p.mt + (long)(NBk * gm / 8) * 64;
        if (best < 0 || cost < best) {
          best = cost;
          p.xcd_gm = gm;
        }
      }
  }
  p.ksplit = (KT + p.kt_per_split - 1) / p.kt_per_split;
  if (p.kernel == QUICK_KERNEL_SKINNY) {
    const int nblocks = N / (16 * p.mt), mblocks = (M + 15) / 16;
    const bool exact = (kernel >> 25) & 1, flip = (kernel >> 21) & 1;
    const int cu_req = (kernel >> 22) & 7;  // workgroup slots per CU for a persistent launch, 0 = choose
    p.grid_x = nblocks;
    // Deferred-zero path (M <= 16, one channel tile per workgroup, x and its unit sums in LDS).  It launches PERSISTENT
    // workgroups -- each walks the channel blocks b, b + grid, ... with the chunk pipeline running across blocks -- so
    // that the x copy and the table are paid once per workgroup.  Slots per CU c in {1, 2} (what LDS allows): two
    // co-resident workgroups finish a block each ~1.6x slower than one alone [r01 sweep], so pick the c with the
    // smaller rounds(c) * (c == 1 ? 1 : 1.8).  With M > 2 the copy + table only pay off from two blocks per workgroup
    // [r01: N = 4096, M = 8: 7.1 us against 5.4 us exact]; below that the exact path runs.
    p.dz = false;
    const size_t lds_dz = skinny_lds_bytes(M, G, 1, p.waves, p.kt_per_split, true, true, true);
    const int fit = (int)std::min<size_t>(2, kLdsPerCu / lds_dz);
    if (!no_xlds && !exact && p.mt == 1 && M <= 16 && fit >= 1) {
      int c = 1;
      if (cu_req) c = std::min(cu_req, 8);
      // (r03 audit, scripts/gpu_tiny_audit.sh: 1.8 -- 1792 blocks, 4096 x 28672, run 7 rounds of one per CU in 13.1 us against 4 rounds of
      // two in 13.8 at M = 1 .. 4; 1376 blocks, 4096 x 22016, stay with 3 rounds of two: 10.7 against 11.2)
      else if (fit >= 2 && ((nblocks + 511) / 512) * 1.8 < (double)((nblocks + 255) / 256)) c = 2;
      int rounds = (nblocks + 256 * c - 1) / (256 * c);
      if (!cu_req && M > 2 && rounds < 2 && c == 2) {
        c = 1;
        rounds = (nblocks + 255) / 256;
      }
      // ... and only when the rounds keep the slots busy (384 blocks on 256 slots = 2 rounds at 75 %: Mistral's qkv
      // at M = 16 lost 2 % to the exact path's 384 one-block workgroups [r01])
      const bool balanced = (double)nblocks >= 0.8 * rounds * 256 * c;
      // (M = 3 from 256 blocks as well [third audit: never behind the exact path there, 13824 x 5120 16.3 -> 11.5 us, 4096 x 6144 6.7 -> 6.2,
      // 5120 x 5120 8.0 -> 7.4, 11008 x 4096 8.2 -> 7.9; with 224 blocks, 18944 x 3584, the exact path is ahead 12.8 against 13.9])
      // (r03: M = 4 as well -- with the rows of the x copy / table prologue batched four at a time the table flavour is ahead there too,
      // 4 x 4096 x 4096 4.88 -> 4.60 us, 4 x 11008 x 4096 9.04 -> 8.28 [scripts/gpu_dz.sh]; from 6 tokens the exact path keeps its lead)
      if (M <= 2 || (M <= 4 && nblocks >= 256 && p.ksplit == 1) || ((kernel >> 26) & 1) || (rounds >= 2 && balanced && p.ksplit == 1)) {  // bit 26: tests force the path
        p.dz = p.xlds = true;
        if (!flip && p.ksplit == 1 && rounds >= 2) p.grid_x = std::min(nblocks, ((nblocks + rounds - 1) / rounds + 7) & ~7);
        // One workgroup per CU and a long K: 16 waves instead of 8 -- twice the bytes in flight per CU and half the chain of
        // k-tiles a wave still has to compute after its last load lands.  M = 1, one session [r02]: 11008 x 4096 8.08 -> 7.12 us,
        // 14336 x 4096 8.80 -> 8.44, 28672 x 8192 25.0 -> 23.4 (M = 2: 8.92 -> 7.64); nothing at K = 4096 (4096 x 12288 7.40 ->
        // 7.36), and slower wherever two 8-wave workgroups share a CU (4096 x 22016 10.4 -> 13.0, 8192 x 8192 8.9 -> 11.6).
        // (G = 32: four units per k-tile, 1 x 11008 x 4096 15.1 us with 16 waves against 9.3 -- not there)
        if (!waves_req && p.waves == 8 && G >= 64 && p.grid_x <= 256 && p.kt_per_split >= 64 &&
            skinny_lds_bytes(M, G, 1, 16, p.kt_per_split, true, true, true) <= kLdsPerCu)
          p.waves = 16;
      }
    }
    // [r03 audit on the final kernels, 5..16 tokens x 12 layer shapes, scripts/gpu_small_audit2.sh] From five tokens the copy of x into
    // LDS loses to fragments straight from L2 wherever the table flavour above does not run: 5 / 6 x 4096 x 4096 5.1 -> 4.4 / 4.7 us,
    // 5 / 6 x 4096 x 6144 6.6 -> 5.7, 10 x 11008 x 4096 11.9 -> 8.9 (a four-way K split with the copy).  G % 128 == 0 (what was measured).
    const bool l2_frag = G % 128 == 0 && (M >= 5 || (M == 4 && p.mt >= 2));   // (4 x 4096 x 6144, two tiles: 6.2 -> 5.8 us)
    if (!p.dz && !exact && !((kernel >> 28) & 1) && p.mt >= 2 && G % 128 == 0 &&  // (G < 128: 4 units per tile, spills)
        (no_xlds || l2_frag || M > 16 || (size_t)std::min(M, 16) * (p.kt_per_split * 256 + 16) > (size_t)64 * 1024)) {
      // fragment flavour of the deferred-zero path: several channel tiles per workgroup, x fragments straight from L2,
      // the unit sums from one extra MFMA per k-step (kernel bit 28 forbids it)
      p.dz = true;
      p.xlds = false;
    } else if (!p.dz) {
      // exact path: x through LDS while the copy is small (64 KiB), else fragments straight from L2; persistent
      // launches measured within +-5 % of one block per workgroup [r01] and are off unless asked for
      p.xlds = !no_xlds && !l2_frag && M <= 16 && (size_t)std::min(M, 16) * (p.kt_per_split * 256 + 16) <= (size_t)64 * 1024;
      // ... and a one-tile launch with a short K runs sixteen waves (twice the loads in flight per CU; K = 4096: M = 5 .. 16 4.4-5.9 ->
      // 4.3-5.9 us, 3-5 % at 6..12 tokens; at K = 11008 eight waves stay ahead)
      if (l2_frag && !waves_req && p.mt == 1 && p.ksplit == 1 && KT <= 32 && mblocks == 1) p.waves = 16;
      const int slots = std::max(1, 256 * (cu_req ? cu_req : 2) / mblocks);
      if (p.xlds && (flip || cu_req) && p.ksplit == 1 && nblocks > slots) {  // the fragments-from-L2 variants cannot
        const int rounds = (nblocks + slots - 1) / slots;
        p.grid_x = (nblocks + rounds - 1) / rounds;
      }
    }
    // [r05] 9..16 tokens where the four-tile fragment flavour runs and K / 128 splits into 8 waves x ksplit x T whole k tiles, T in {2, 4, 7, 8}:
    // eight tiles per workgroup, straight-line (w4a16_frag8_kernel).  ksplit: one slice from 192 blocks, else the smallest of 2, 4 that gives >= 256 workgroups.
    // Forced by kernel id (SKINNY, 8 channel tiles); AUTO takes it where QUICK_AMD_FRAG8 says (default below).
    {
      static const int frag8_env = [] {
        const char* e = getenv("QUICK_AMD_FRAG8");
        return e ? atoi(e) : 1;
      }();
      const bool asked7 = family == QUICK_KERNEL_SKINNY && mt_req == 7;   // (forced: seven tiles per workgroup, T = 8, one slice)
      const bool asked = (family == QUICK_KERNEL_SKINNY && mt_req == 8) || asked7;
      const bool auto_ok = frag8_env != 0 && !with_ln && family == QUICK_KERNEL_AUTO && !mt_req && !waves_req && !(kernel >> 12) && grid_split_k == 0 && p.mt == 4 && p.dz &&
                           !p.xlds && (long)K * N >= 60L * 1000 * 1000;
      if ((asked || auto_ok) && G == 128 && M >= 9 && M <= 16 && N % 128 == 0 && KT % 8 == 0 && (long)M * K * 2 < (1L << 31)) {
        int best_ks = 0, best_t = 0;
        for (int s2 = 1; s2 <= 4; s2 *= 2) {   // the forced slice count, else the first that gives >= 256 workgroups, else the last one that fits
          if (grid_split_k > 0 && s2 != grid_split_k) continue;
          if (asked7 && s2 != 1) continue;
          if ((KT / 8) % s2 != 0) continue;
          const int t = KT / 8 / s2;
          if (t != 2 && t != 4 && t != 7 && t != 8) continue;
          best_ks = s2;
          best_t = t;
          if ((N / 128) * s2 >= (s2 == 1 ? 192 : 256)) break;   // (one slice from 192 blocks: a K split costs more than the idle CUs)
        }
        // (AUTO only where measured ahead: 8192 x 57344 one slice 62 -> 54 us, 28672 x 8192 four slices 29.5 -> 27.9, 8192 x 8192 four slices
        // 12.9 -> 11.9; 8192 x 10240 -- 80 blocks -- two slices 17.2 against 15.3: stays with four tiles)
        // ... and 8192 x 28672 -- 224 blocks, one slice -- 28.7 -> 23.5: one slice from 192 blocks, K slices only on whole multiples of 64 blocks)
        // (128 blocks x two slices, 8192 x 16384: 16.4 -> 18.5 us, behind: K slices only where measured ahead, 64 blocks x four)
        if (best_ks && !asked && !((best_ks == 1 && N / 128 >= 192) || (N / 128 == 64 && best_ks == 4))) best_ks = 0;
        if (asked7 && !(best_ks == 1 && best_t == 8 && N % 112 == 0)) best_ks = 0;
        if (best_ks) {
          p.mt = 8;
          p.waves = 8;
          p.dz = true;
          p.xlds = false;
          p.frag8_t = best_t;
          p.ksplit = best_ks;
          p.kt_per_split = 8 * best_t;
          p.grid_x = N / 128;
          p.ntiles = (N / 128) * ((M + 15) / 16);
          p.slab_floats = (size_t)8 * 256;
          // [r06, late] seven tiles per workgroup where that makes fewer or fuller rounds of one workgroup per CU (one slice, T = 8 only -- K = 8192): a round costs
          // its weights + its x fragments (x: as many bytes as four tiles' weights), so 7 + 4 against 8 + 4 per round.  16 x 8192 x 57344: 448 blocks = 2 rounds (the
          // second three quarters full) -> 512 blocks = 2 full rounds of less: QUICK_AMD_FRAG7=0 keeps eight.
          static const int frag7_env = [] {
            const char* e = getenv("QUICK_AMD_FRAG7");
            return e ? atoi(e) : 1;
          }();
          if (best_ks == 1 && best_t == 8 && N % 112 == 0 && (asked7 || (!asked && frag7_env != 0))) {
            const int cus = cu_count();
            const long cost8 = (long)((N / 128 + cus - 1) / cus) * 12, cost7 = (long)((N / 112 + cus - 1) / cus) * 11;
            if (asked7 || cost7 < cost8) {
              p.mt = 7;
              p.grid_x = N / 112;
              p.ntiles = (N / 112) * ((M + 15) / 16);
              p.slab_floats = (size_t)7 * 256;
            }
          }
        }
      }
    }
  }
  // [r06] sixteen waves (128 registers per lane): one-tile workgroups, and the table flavour only with <= 2 units per k tile -- the other builds
  // spilled registers and were never the planner's own pick; a forced request runs eight waves
  if (p.kernel == QUICK_KERNEL_SKINNY && p.waves == 16 && (p.mt != 1 || (p.dz && group_mode(G) > 2))) p.waves = 8;
  return p;
}

// workspace: [64 KiB of arrival counters, one per output tile][exchange zone][ntiles * ksplit fp32 slabs]
// The counter region has a FIXED size: the slabs are not handed back zeroed, so a region that grew with the tile count
// would lay the counters of one launch over the stale partial sums of an earlier, smaller one.
static constexpr int kMaxSplitTiles = 16384;
static size_t counters_bytes(const Plan&) { return (size_t)kMaxSplitTiles * 4; }
// ... and behind the counters a FIXED 16 MiB "exchange zone": the mailboxes of the exchange-K kernels (w4a16_xk.hpp), all-zero
// between launches (every consumer zeroes what it has read); the slabs of the last-arriver kernels start behind it and never
// touch it.  Layout: [64 KiB counters][16 MiB exchange zone][ntiles * ksplit fp32 slabs].
static size_t slabs_offset(const Plan& p) { return counters_bytes(p) + kXkZoneBytesHost; }
static size_t workspace_need(const Plan& p) {
  if (p.ksplit <= 1) return 0;
  if (p.kernel == QUICK_KERNEL_XK || p.kernel == QUICK_KERNEL_XW) return slabs_offset(p);
  return slabs_offset(p) + (size_t)p.ntiles * p.ksplit * p.slab_floats * sizeof(float);
}

// SiLU * mul in an exchange-K launch needs every wave to finish whole 32-token blocks (w4a16_xk.hpp, the way out)
static bool xk_takes_silu(const Plan& p) { return (p.wide_mb / 2 * 16) % (16 * p.ksplit) == 0; }
// The planner does not see the epilogue; where it picks an exchange-K launch that cannot carry SiLU * mul, the launch falls back to
// the plan without those kernels.  The workspace must serve either.
static Plan plan_for(int M, int K, int N, int G, int kernel, int grid_split_k, bool silu, bool with_ln = false) {
  Plan p = make_plan(M, K, N, G, kernel, grid_split_k, true, with_ln);
  if (silu && (kernel & 15) == QUICK_KERNEL_AUTO && p.kernel == QUICK_KERNEL_XK && !xk_takes_silu(p)) p = make_plan(M, K, N, G, kernel, grid_split_k, false);
  return p;
}


template <int NTW, int WAVES, bool XLDS, bool DZ, bool LN = false>
static void launch_skinny_gm(const Plan& p, const GemmArgs& a, const Launch& L) {
  dim3 grid(p.grid_x, (a.M + 15) / 16, p.ksplit), block(WAVES * 64);
  const size_t lds = skinny_lds_bytes(a.M, a.G, NTW, WAVES, p.kt_per_split, p.grid_x < a.N / (16 * NTW), XLDS, DZ) + (LN ? 1024 : 0);
  if (a.span) {  // in-kernel span stamps: separate instantiations, for the small-M kernels the BASELINE sweep and the decode shapes run
    constexpr bool stamped = !LN && ((WAVES == 8 && ((NTW == 1 && XLDS && DZ) || (NTW == 1 && !XLDS && !DZ) || (NTW == 4 && !XLDS && DZ))) ||
                                     (WAVES == 16 && NTW == 1 && ((XLDS && DZ) || (!XLDS && !DZ))));
    if constexpr (stamped) {
      if (group_mode(a.G) == 0) {   // [r06, ADVICE r05] the stamped build of the kernel the product runs: nt weight requests with one token block
        auto kfn = w4a16_skinny_kernel<NTW, WAVES, 0, XLDS, DZ, LN, true>;
        auto kfn_nt = w4a16_skinny_kernel<NTW, WAVES, 0, XLDS, DZ, LN, true, true>;
        (void)hipFuncSetAttribute((const void*)(a.M <= 16 ? kfn_nt : kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
        if (a.M <= 16) hipExtLaunchKernelGGL(kfn_nt, grid, block, (unsigned)lds, L.st, L.start, L.stop, 0, a);
        else hipExtLaunchKernelGGL(kfn, grid, block, (unsigned)lds, L.st, L.start, L.stop, 0, a);
        return;
      }
    }
    g_span_unsupported = true;
    return;
  }
  if (a.M <= 16 && group_mode(a.G) == 0) {  // one token block: weights streamed once -> nt requests (see skinny_load)
    auto kfn = w4a16_skinny_kernel<NTW, WAVES, 0, XLDS, DZ, LN, false, true>;
    static std::atomic<unsigned long long> attr_set_nt{0};
    (void)lds_limit_once(attr_set_nt, (const void*)kfn, (int)kLdsPerCu);
    hipExtLaunchKernelGGL(kfn, grid, block, (unsigned)lds, L.st, L.start, L.stop, 0, a);
    return;
  }
#define QA_SKINNY(GMV)                                                                                             \
  do {                                                                                                             \
    auto kfn = w4a16_skinny_kernel<NTW, WAVES, GMV, XLDS, DZ, LN>;                                                 \
    static std::atomic<unsigned long long> attr_set{0};                                                                \
    (void)lds_limit_once(attr_set, (const void*)kfn, (int)kLdsPerCu);                                              \
    hipExtLaunchKernelGGL(kfn, grid, block, (unsigned)lds, L.st, L.start, L.stop, 0, a);                           \
  } while (0)
  if constexpr (LN) {  // only reached with G % 128 == 0
    if (group_mode(a.G) == 0) QA_SKINNY(0);
    else QA_SKINNY(1);
  } else {
    switch (group_mode(a.G)) {
      case 0: QA_SKINNY(0); break;
      case 1: QA_SKINNY(1); break;
      case 2: QA_SKINNY(2); break;
      // (sixteen waves of the table flavour with > 2 units per k tile: no build -- it spilled; make_plan hands such a request eight waves)
      case 3: if constexpr (!(WAVES == 16 && DZ)) QA_SKINNY(3); break;
      default: if constexpr (!(WAVES == 16 && DZ)) QA_SKINNY(4); break;
    }
  }
#undef QA_SKINNY
}

template <int NTW>
static void launch_skinny(const Plan& p, const GemmArgs& a, const Launch& L) {
  if constexpr (NTW == 1) {
    if (p.dz) {  // M <= 16: one channel tile per workgroup
      if (p.waves == 4) launch_skinny_gm<1, 4, true, true>(p, a, L);
      else if (p.waves == 16) launch_skinny_gm<1, 16, true, true>(p, a, L);
      else launch_skinny_gm<1, 8, true, true>(p, a, L);
      return;
    }
  }
  if constexpr (NTW >= 2) {
    if (p.dz && !p.xlds) {
      if (a.ln_w) launch_skinny_gm<NTW, 8, false, true, true>(p, a, L);  // RMSNorm on the fragments (run_gemm checked waves == 8)
      else if (p.waves == 4) launch_skinny_gm<NTW, 4, false, true>(p, a, L);
      else launch_skinny_gm<NTW, 8, false, true>(p, a, L);
      return;
    }
  }
  // (sixteen waves: one-tile workgroups only -- make_plan never asks for more, and with 128 registers per lane the two- and four-tile builds spilled)
  if (p.xlds) {
    if (p.waves == 4) launch_skinny_gm<NTW, 4, true, false>(p, a, L);
    else if (NTW == 1 && p.waves == 16) launch_skinny_gm<1, 16, true, false>(p, a, L);
    else launch_skinny_gm<NTW, 8, true, false>(p, a, L);
  } else {
    if (p.waves == 4) launch_skinny_gm<NTW, 4, false, false>(p, a, L);
    else if (NTW == 1 && p.waves == 16) launch_skinny_gm<1, 16, false, false>(p, a, L);
    else launch_skinny_gm<NTW, 8, false, false>(p, a, L);
  }
}

template <int BMT, int WK, int WN = 4, int TCH = 128>
static void launch_tiled(const Plan& p, const GemmArgs& a, const Launch& L) {
  constexpr int TN = TCH / 16 / WN;  // channel tiles per wave
  dim3 grid((a.N / TCH) * ((a.M + BMT * 16 - 1) / (BMT * 16)), p.ksplit), block(64 * WN * WK);
  const unsigned lds = 2 * 4 * WK * BMT * 1024;
  if (a.span) {
    g_span_unsupported = true;
    return;
  }
#define QA_TILED_K(GMV, ABLV)                                                                                      \
  do {                                                                                                             \
    auto kfn = w4a16_tiled_kernel<BMT, TN, WK, GMV, ABLV, WN>;                                                        \
    static std::atomic<unsigned long long> attr_set{0};                                                                \
    (void)lds_limit_once(attr_set, (const void*)kfn, (int)lds);                                                    \
    hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);                                     \
  } while (0)
#ifdef QUICK_AMD_TOOLS
  if constexpr (BMT == 4 && WN == 4 && TCH == 128) if (p.ablate && a.G == 128) {  // timing experiments (tools/): results are wrong on purpose
    switch (p.ablate) {
      case 1: QA_TILED_K(0, 1); return;
      case 2: QA_TILED_K(0, 2); return;
      case 3: QA_TILED_K(0, 3); return;
      case 4: QA_TILED_K(0, 4); return;
      case 7: QA_TILED_K(0, 7); return;
      case 15: QA_TILED_K(0, 15); return;
      case 16: QA_TILED_K(0, 16); return;
      case 17: QA_TILED_K(0, 17); return;
      case 18: QA_TILED_K(0, 18); return;
      case 20: QA_TILED_K(0, 20); return;
      case 22: QA_TILED_K(0, 22); return;
      case 23: QA_TILED_K(0, 23); return;
      default: break;
    }
  }
#endif
  switch (group_mode(a.G)) {
    case 0: QA_TILED_K(0, 0); break;
    case 1: QA_TILED_K(1, 0); break;
    case 2: QA_TILED_K(2, 0); break;
    case 3: QA_TILED_K(3, 0); break;
    default: QA_TILED_K(4, 0); break;
  }
#undef QA_TILED_K
}

template <int MB, int PAIRS, int NBUF, int WK = 1>
static void launch_ring(const Plan& p, const GemmArgs& a, const Launch& L) {
  dim3 grid(p.ntiles, p.ksplit), block(256 * WK);
  constexpr unsigned lds = NBUF * (MB * 8192 + PAIRS * 8192 + WK * PAIRS * 1024);
#define QA_RING_K(GMV)                                                                                             \
  do {                                                                                                             \
    auto kfn = w4a16_ring_kernel<MB, PAIRS, GMV, NBUF, 0, WK>;                                                     \
    static std::atomic<unsigned long long> attr_set{0};                                                                \
    (void)lds_limit_once(attr_set, (const void*)kfn, (int)lds);                                                    \
    hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);                                     \
  } while (0)
#ifdef QUICK_AMD_TOOLS
  if constexpr ((MB == 2 && PAIRS == 1 && NBUF == 4 && WK == 2) || (MB == 4 && PAIRS == 1 && NBUF == 3 && WK == 2))
    if (p.ablate == 16 && a.G == 128) {  // phase stamps into the workspace (tools/wide_phases.py), nothing else changes
      auto kfn = w4a16_ring_kernel<MB, PAIRS, 0, NBUF, 64, WK>;
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);
      return;
    }
  if constexpr (MB == 2 && PAIRS == 1 && NBUF == 6 && WK == 1)
    if (p.ablate && a.G == 128) {  // timing experiments (results are wrong on purpose)
      auto kfn1 = w4a16_ring_kernel<MB, PAIRS, 0, NBUF, 1>;
      auto kfn2 = w4a16_ring_kernel<MB, PAIRS, 0, NBUF, 2>;
      auto kfn5 = w4a16_ring_kernel<MB, PAIRS, 0, NBUF, 5>;
      auto kfn9 = w4a16_ring_kernel<MB, PAIRS, 0, NBUF, 9>;
      auto kfn18 = w4a16_ring_kernel<MB, PAIRS, 0, NBUF, 18>;
      auto kfn = p.ablate == 1 ? kfn1 : (p.ablate == 5 ? kfn5 : (p.ablate == 9 ? kfn9 : (p.ablate == 18 ? kfn18 : kfn2)));
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);
      return;
    }
  if constexpr (MB == 2 && PAIRS == 1 && WK == 2)
    if (p.ablate && a.G == 128) {
      auto kfn1 = w4a16_ring_kernel<MB, PAIRS, 0, NBUF, 1, WK>;
      auto kfn2 = w4a16_ring_kernel<MB, PAIRS, 0, NBUF, 2, WK>;
      auto kfn = p.ablate == 1 ? kfn1 : kfn2;
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);
      return;
    }
#endif
  if (a.span) {
    if constexpr (MB == 2 && PAIRS == 1 && ((NBUF == 6 && WK == 1) || (NBUF == 4 && WK == 2))) {
      if (group_mode(a.G) == 0) {
        auto kfn = w4a16_ring_kernel<MB, PAIRS, 0, NBUF, 32, WK>;   // ABL bit 32 = span stamps, nothing else
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);
        return;
      }
    }
    g_span_unsupported = true;
    return;
  }
  if (group_mode(a.G) == 0) QA_RING_K(0);
  else QA_RING_K(1);
#undef QA_RING_K
}

template <int MB, int PAIRS>
static void launch_wide(const Plan& p, const GemmArgs& a, const Launch& L) {
  if constexpr (MB <= 4 && MB * PAIRS <= 4) {
    if (p.wide_nbuf >= 3 && p.waves == 8) {  // eight waves (two per SIMD), the k16 steps split by parity
      constexpr int NMAX8 = wide_ring_nbuf(MB, PAIRS, 2);
      if (p.wide_nbuf == 3) launch_ring<MB, PAIRS, 3, 2>(p, a, L);
      else if (NMAX8 >= 4 && p.wide_nbuf == 4) launch_ring<MB, PAIRS, (NMAX8 >= 4 ? 4 : 3), 2>(p, a, L);
      else launch_ring<MB, PAIRS, NMAX8, 2>(p, a, L);
      return;
    }
  }
  if constexpr (MB <= 4) {
    if (p.wide_nbuf >= 3) {
      constexpr int NMAX = wide_ring_nbuf(MB, PAIRS);
      if (p.wide_nbuf == 3) launch_ring<MB, PAIRS, 3>(p, a, L);
      else if (NMAX >= 4 && p.wide_nbuf == 4) launch_ring<MB, PAIRS, (NMAX >= 4 ? 4 : 3)>(p, a, L);
      else launch_ring<MB, PAIRS, NMAX>(p, a, L);
      return;
    }
  }
  dim3 grid(p.ntiles, p.ksplit), block(256);
  const unsigned lds = 2 * MB * 32 * 256;
  if (a.span) {
    g_span_unsupported = true;
    return;
  }
#define QA_WIDE_K(GMV)                                                                                             \
  do {                                                                                                             \
    auto kfn = w4a16_wide_kernel<MB, PAIRS, GMV>;                                                                  \
    static std::atomic<unsigned long long> attr_set{0};                                                                \
    (void)lds_limit_once(attr_set, (const void*)kfn, (int)lds);                                                    \
    hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);                                     \
  } while (0)
#ifdef QUICK_AMD_TOOLS
  if constexpr ((MB == 2 && PAIRS == 1) || (MB == 8 && PAIRS == 2) || (MB == 4 && PAIRS == 2))
    if (p.ablate && a.G == 128) {  // timing experiments (results are wrong on purpose)
      auto kfn1 = w4a16_wide_kernel<MB, PAIRS, 0, 1>;
      auto kfn2 = w4a16_wide_kernel<MB, PAIRS, 0, 2>;
      auto kfn64 = w4a16_wide_kernel<MB, PAIRS, 0, 64>;  // phase stamps into the workspace, nothing else changes
      auto kfn = p.ablate == 1 ? kfn1 : (p.ablate == 16 ? kfn64 : kfn2);
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);
      return;
    }
#endif
  if (group_mode(a.G) == 0) QA_WIDE_K(0);
  else QA_WIDE_K(1);
#undef QA_WIDE_K
}

struct Fusion {
  const void* bias = nullptr;
  const void* residual = nullptr;
  const void* ln_w = nullptr;
  float ln_eps = 0.f;
  int silu_mul = 0;
};

static int run_gemm(const void* x, const void* qweight, const void* scales, const void* qzeros, const Fusion& f, void* y,
                    void* workspace, size_t workspace_bytes, int M, int K, int N, int G, int kernel, int grid_split_k,
                    const Launch& L) {
  if (int rc = check_shapes(M, K, N, G)) return rc;
  if ((kernel & 15) > QUICK_KERNEL_XM || kernel < 0) return fail(QUICK_ERR_INVALID_ARGUMENT, "unknown kernel id %d", kernel);
#ifndef QUICK_AMD_TOOLS
  if ((kernel >> 16) & 31)  // the timing-experiment builds (wrong results on purpose, phase stamps) are not in the product library
    return fail(QUICK_ERR_INVALID_ARGUMENT, "kernel id %d: bits 16-20 select timing experiments that only a QUICK_AMD_TOOLS build contains", kernel);
  if ((kernel & 15) == QUICK_KERNEL_XK && ((kernel >> 12) & 3))  // the twelve-wave (loader waves, bit 12) and sixteen-wave (bit 13) flavours: measured level with the eight-wave one, DESIGN.md 5.9
    return fail(QUICK_ERR_INVALID_ARGUMENT, "kernel id %d: the loader-wave / sixteen-wave flavours of the exchange-K kernels are only in a QUICK_AMD_TOOLS build", kernel);
#endif
  if ((kernel & 15) == QUICK_KERNEL_TILED && ((kernel >> 13) & 1))  // r01's 32x32x16 flavour of the tiled kernel (measured behind the 16x16x32 one, spilled registers): retired in r06
    return fail(QUICK_ERR_INVALID_ARGUMENT, "kernel id %d: the 32x32x16 flavour of the tiled kernel (bit 13) was retired in r06", kernel);
  if (!x || !qweight || !scales || !qzeros || !y) return fail(QUICK_ERR_INVALID_ARGUMENT, "null tensor pointer");
  const Plan p = plan_for(M, K, N, G, kernel, grid_split_k, f.silu_mul != 0, f.ln_w != nullptr);
  if (f.silu_mul && (f.bias || f.residual)) return fail(QUICK_ERR_INVALID_ARGUMENT, "silu_mul excludes bias and residual");
  if (f.ln_w && p.kernel != QUICK_KERNEL_LEAN && !(p.kernel == QUICK_KERNEL_SKINNY && p.dz && p.ksplit == 1 && !p.frag8_t && (p.xlds || (p.mt >= 2 && p.waves == 8))))
    return fail(QUICK_ERR_UNSUPPORTED, "RMSNorm prologue: only on the deferred-zero path (see quick_w4a16_can_fuse_rmsnorm)");
  GemmArgs a{(const half_t*)x, (const u32x4*)qweight, (const half_t*)scales, (const uint32_t*)qzeros, (const half_t*)f.bias,
             (const half_t*)f.residual, f.silu_mul, (half_t*)y, nullptr, nullptr, M, K, N, G, std::max(1, G / 128), p.ksplit, p.kt_per_split, p.xcd_gm, nullptr, (const half_t*)f.ln_w, f.ln_eps};
  // (phase stamps: the LAST 2 MiB of the workspace, behind whatever a K split needs)
  if (p.ablate >= 16 && workspace && workspace_bytes >= (size_t)4096 * 8 * 64 + workspace_need(p))
    a.dbg = (unsigned long long*)((char*)workspace + workspace_bytes - (size_t)4096 * 8 * 64);
  a.span = L.span;
  g_span_unsupported = false;
  if (p.ksplit > 1) {
    if (p.ntiles > kMaxSplitTiles) return fail(QUICK_ERR_UNSUPPORTED, "K split over %d output tiles (limit %d)", p.ntiles, kMaxSplitTiles);
    const size_t need = workspace_need(p);
    if (!workspace || workspace_bytes < need)
      return fail(QUICK_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    // QUICK_AMD_CHECK_WORKSPACE=1 (a debugging aid: it synchronises the stream): every launch that meets its partners through the workspace
    // first proves the guarded regions all-zero and answers QUICK_ERR_WORKSPACE otherwise -- a stale granule is indistinguishable from a
    // partner's partial sum, see include/quick_amd.h
    static const bool check_ws = [] {
      const char* e = getenv("QUICK_AMD_CHECK_WORKSPACE");
      return e && *e && atoi(e) != 0;
    }();
    if (check_ws)
      if (int rcw = quick_w4a16_workspace_check(workspace, workspace_bytes, (void*)L.st)) return rcw;
    a.counters = (unsigned*)workspace;  // zero on entry (caller's contract), zero again when the launch completes
    a.slabs = (float*)((char*)workspace + ((p.kernel == QUICK_KERNEL_XK || p.kernel == QUICK_KERNEL_XW) ? counters_bytes(p) : slabs_offset(p)));  // (exchange-K / XW: the zone)
  }
  if (p.kernel == QUICK_KERNEL_LEAN) {
    int abl = a.span ? 32 : 0;
#ifdef QUICK_AMD_TOOLS
    if (p.ablate == 16) abl = 64;  // phase stamps into the workspace (tools/lean_phases.py)
    else if (p.ablate) return fail(QUICK_ERR_INVALID_ARGUMENT, "LEAN: timing-experiment bit 16 (stamps) only");
#endif
    if (!p.lean_tmax || !lean_launch(p.waves, p.lean_tmax, p.mt, abl, a, p.grid_x, (M + 15) / 16, L.st, L.start, L.stop)) {
      if (abl == 32) g_span_unsupported = true;
      return fail(QUICK_ERR_UNSUPPORTED, "no lean build for K=%d G=%d waves=%d (G %% 128 == 0, waves <= K / 128 <= 16 waves, x + table within 160 KiB of LDS)", K, G, p.waves);
    }
  } else if (p.kernel == QUICK_KERNEL_XM) {
    if (f.ln_w) return fail(QUICK_ERR_UNSUPPORTED, "RMSNorm prologue: only on the deferred-zero path (see quick_w4a16_can_fuse_rmsnorm)");
    int abl = a.span ? 32 : 0;
#ifdef QUICK_AMD_TOOLS
    if (p.ablate == 16) abl = 64;  // phase stamps into the workspace (tools/xm_phases.py)
    else if (p.ablate) return fail(QUICK_ERR_INVALID_ARGUMENT, "XM: timing-experiment bit 16 (stamps) only");
#endif
    if (!p.lean_tmax || !xm_launch(p.wide_mb, p.wide_pairs, abl, a, p.grid_x, p.ntiles / p.grid_x, L.st, L.start, L.stop))
      return fail(QUICK_ERR_UNSUPPORTED, "no mid-token build for K=%d G=%d (G a power-of-two multiple of 128, 32-bit offsets)", K, G);
  } else if (p.kernel == QUICK_KERNEL_XW) {
    if (f.ln_w) return fail(QUICK_ERR_UNSUPPORTED, "RMSNorm prologue: only on the deferred-zero path (see quick_w4a16_can_fuse_rmsnorm)");
    int abl = a.span ? 32 : 0;
#ifdef QUICK_AMD_TOOLS
    if (p.ablate == 16) {               // phase stamps (QUICK_XW_EXP: + a loop experiment, tools/gen_xw_loop.py)
      abl = 64;
      if (const char* e = getenv("QUICK_XW_EXP")) abl += 256 * atoi(e);
    }
    else if (p.ablate == 20) abl = 68;  // ... and no exchange (wrong results)
    else if (p.ablate) return fail(QUICK_ERR_INVALID_ARGUMENT, "XW: timing-experiment bits 16 (stamps) and 20 (no exchange) only");
#endif
    a.xcd_gm |= p.poll_log2 << 8;  // (the kernel reads the tile-order rows from the low byte)
    if (!xw_launch(p.wide_mb, p.wide_pairs, p.ksplit, abl, a, p.ntiles * p.ksplit, L.st, L.start, L.stop)) {
      if (abl == 32) g_span_unsupported = true;
      return fail(QUICK_ERR_UNSUPPORTED, "no four-wave build for tokens=%d channels=%d slices=%d abl=%d", p.wide_mb * 32, p.tch, p.ksplit, abl);
    }
  } else if (p.kernel == QUICK_KERNEL_XK) {
    if (f.ln_w) return fail(QUICK_ERR_UNSUPPORTED, "RMSNorm prologue: only on the deferred-zero path (see quick_w4a16_can_fuse_rmsnorm)");
    if (f.silu_mul && !xk_takes_silu(p))
      return fail(QUICK_ERR_UNSUPPORTED, "exchange-K kernels: SiLU * mul only where a wave finishes whole 32-token blocks");
    // (tools builds: kernel bits 16-20 = 16 phase stamps only; 17 loads only; 18 no loads; 19 no loads, no B-fragment reads; 20 no cross-CU
    // exchange; 21 no loads, no dequantisation; 22 MFMAs + barrier; 23 no loads, no barrier; 24 MFMAs only -- all with the stamps)
    static const int xk_abl[9] = {64, 65, 66, 82, 68, 74, 90, 98, 122};
    int abl = p.ablate == 0 ? (a.span ? 32 : 0) : ((p.ablate >= 16 && p.ablate <= 24) ? xk_abl[p.ablate - 16] : -1);  // (32: in-kernel span stamps, a measurement aid)
#ifdef QUICK_AMD_TOOLS
    if (p.ablate)  // (experiments whose bits do not fit the kernel id: the ABL value itself, tools/xk_phases.py --env-abl)
      if (const char* e = getenv("QUICK_XK_ABL")) abl = atoi(e) == 0 && a.span ? 32 : atoi(e);
#endif
    if (const char* e = getenv("QUICK_AMD_EXCHANGE_POLL_LOG2")) a.xcd_gm |= std::max(0, std::min(31, atoi(e))) << 8;   // (tests: 1 = every wave gives its part up at once)
    if (!xk_launch(XkConfig{p.wide_mb, p.ksplit, p.xk_nbuf, p.xk_wd, abl, p.xk_kq, p.xk_loader ? 1 : 0}, a, p.ntiles * p.ksplit, L.st, L.start, L.stop)) {
      if (abl == 32) {
        g_span_unsupported = true;
        return fail(QUICK_ERR_UNSUPPORTED, "no span-stamped build of the kernel this shape runs");
      }
      return fail(QUICK_ERR_UNSUPPORTED, "no exchange-K build for tokens=%d slices=%d ring=%d queue=%d%s", p.wide_mb * 32, p.ksplit, p.xk_nbuf,
                  p.xk_wd, abl ? " (timing-experiment bits need a QUICK_AMD_TOOLS build)" : "");
    }
  } else if (p.kernel == QUICK_KERNEL_WIDE) {
    if (f.ln_w) return fail(QUICK_ERR_UNSUPPORTED, "RMSNorm prologue: only on the deferred-zero path (see quick_w4a16_can_fuse_rmsnorm)");
    const int sel = p.wide_mb * 10 + p.wide_pairs;
    switch (sel) {
      case 21: launch_wide<2, 1>(p, a, L); break;
      case 22: launch_wide<2, 2>(p, a, L); break;
      case 41: launch_wide<4, 1>(p, a, L); break;
      case 42: launch_wide<4, 2>(p, a, L); break;
      case 81: launch_wide<8, 1>(p, a, L); break;
#ifdef QUICK_AMD_TOOLS
      default: launch_wide<8, 2>(p, a, L); break;   // r02's hipcc-scheduled 256 x 256 tile (92 bytes of scratch): the A/B partner of the generated loop, tools builds only
#else
      default: return fail(QUICK_ERR_UNSUPPORTED, "the hipcc-scheduled 256 x 256 tile is only in a QUICK_AMD_TOOLS build (the product runs the generated loop, w4a16_xw.hpp, or 128 x 256 tiles)");
#endif
    }
  } else if (p.kernel == QUICK_KERNEL_SKINNY) {
    switch (p.mt) {
      case 1: launch_skinny<1>(p, a, L); break;
      case 2: launch_skinny<2>(p, a, L); break;
      case 7: {   // seven tiles per workgroup: the straight-line fragment kernel with T = 8, one slice
        if (p.frag8_t != 8 || p.ksplit != 1 || f.ln_w) return fail(QUICK_ERR_UNSUPPORTED, "skinny: seven channel tiles per workgroup only in the straight-line fragment kernel (K = 8192, one slice), no RMSNorm prologue");
        dim3 grid(p.grid_x, (M + 15) / 16, 1), block(512);
        const unsigned lds = 8 * 7 * 1024;
        if (a.span) {
          auto kfn = w4a16_frag8_kernel<8, true, true, 7>;
          static std::atomic<unsigned long long> attr_set_s{0};
          (void)lds_limit_once(attr_set_s, (const void*)kfn, (int)kLdsPerCu);
          hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);
        } else {
          auto kfn = w4a16_frag8_kernel<8, true, false, 7>;
          static std::atomic<unsigned long long> attr_set{0};
          (void)lds_limit_once(attr_set, (const void*)kfn, (int)kLdsPerCu);
          hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);
        }
        break;
      }
      case 8: {
        if (!p.frag8_t || f.ln_w) return fail(QUICK_ERR_UNSUPPORTED, "skinny: eight channel tiles per workgroup only in the straight-line fragment kernel, no RMSNorm prologue");
        dim3 grid(p.grid_x, (M + 15) / 16, p.ksplit), block(512);
        const unsigned lds = 8 * 8 * 1024;
#define QA_FRAG8(TV)                                                                                               \
  do {                                                                                                             \
    if (a.span) { /* in-kernel span stamps (a measurement aid): the same kernel + two stamps */                    \
      auto kfn = w4a16_frag8_kernel<TV, true, true>;                                                               \
      static std::atomic<unsigned long long> attr_set_s{0};                                                        \
      (void)lds_limit_once(attr_set_s, (const void*)kfn, (int)kLdsPerCu);                                          \
      hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);                                   \
    } else {                                                                                                       \
      auto kfn = w4a16_frag8_kernel<TV, true>;                                                                     \
      static std::atomic<unsigned long long> attr_set{0};                                                          \
      (void)lds_limit_once(attr_set, (const void*)kfn, (int)kLdsPerCu);                                            \
      hipExtLaunchKernelGGL(kfn, grid, block, lds, L.st, L.start, L.stop, 0, a);                                   \
    }                                                                                                              \
  } while (0)
        switch (p.frag8_t) {
          case 2: QA_FRAG8(2); break;
          case 4: QA_FRAG8(4); break;
          case 7: QA_FRAG8(7); break;
          default: QA_FRAG8(8); break;
        }
#undef QA_FRAG8
        break;
      }
      default: launch_skinny<4>(p, a, L); break;
    }
  } else {
    if (p.tch == 256 && p.mt == 8) launch_tiled<8, 1, 4, 256>(p, a, L);  // 128 x 256, four waves of 128 tokens x 64 channels
    else if (p.tch == 256) launch_tiled<4, 2, 4, 256>(p, a, L);    // 64 tokens x 256 channels, 4 channel tiles per wave
    else if (p.mt == 4 && p.wn2) launch_tiled<4, 4, 2>(p, a, L);  // 2 waves along N x 4 along K, 64 channels per wave
    else if (p.mt == 2) launch_tiled<2, 2>(p, a, L);
    else if (p.mt == 8) launch_tiled<8, 2>(p, a, L);
    else if (p.waves == 8) launch_tiled<4, 2>(p, a, L);
    else launch_tiled<4, 4>(p, a, L);
  }
  if (g_span_unsupported) return fail(QUICK_ERR_UNSUPPORTED, "no span-stamped build of the kernel this shape runs");
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QUICK_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
  return QUICK_OK;
}

}  // namespace quick_amd

using namespace quick_amd;

extern "C" {

int quick_amd_abi_version(void) { return QUICK_AMD_ABI_VERSION; }
const char* quick_amd_last_error(void) { return g_err; }

size_t quick_w4a16_workspace_bytes_ex(int M, int K, int N, int group_size, int kernel, int grid_split_k) {
  if (check_shapes(M, K, N, group_size) != QUICK_OK) return 0;
  const Plan p = make_plan(M, K, N, group_size, kernel, grid_split_k);
  size_t need = workspace_need(p);
  if ((kernel & 15) == QUICK_KERNEL_AUTO && p.kernel == QUICK_KERNEL_XK && !xk_takes_silu(p))   // (the fallback of a SiLU * mul launch)
    need = std::max(need, workspace_need(make_plan(M, K, N, group_size, kernel, grid_split_k, false)));
  return need;
}

size_t quick_w4a16_workspace_bytes(int M, int K, int N, int group_size, int split_k_iters) {
  (void)split_k_iters;
  return quick_w4a16_workspace_bytes_ex(M, K, N, group_size, QUICK_KERNEL_AUTO, 0);
}

int quick_w4a16_gemm_f16_ex(const void* x, const void* qweight, const void* scales, const void* qzeros,
                            const void* bias, void* y, void* workspace, size_t workspace_bytes, int M, int K, int N,
                            int group_size, int kernel, int grid_split_k, void* hip_stream) {
  const Launch L{(hipStream_t)hip_stream, nullptr, nullptr};
  Fusion f;
  f.bias = bias;
  return run_gemm(x, qweight, scales, qzeros, f, y, workspace, workspace_bytes, M, K, N, group_size, kernel, grid_split_k, L);
}

int quick_w4a16_gemm_f16_fused(const void* x, const void* qweight, const void* scales, const void* qzeros,
                               const quick_gemm_fusion* fusion, void* y, void* workspace, size_t workspace_bytes, int M,
                               int K, int N, int group_size, int kernel, int grid_split_k, void* hip_stream) {
  const Launch L{(hipStream_t)hip_stream, nullptr, nullptr};
  Fusion f;
  if (fusion) {
    f.bias = fusion->bias;
    f.residual = fusion->residual;
    f.ln_w = fusion->rmsnorm_weight;
    f.ln_eps = fusion->rmsnorm_eps;
    f.silu_mul = fusion->silu_mul;
  }
  return run_gemm(x, qweight, scales, qzeros, f, y, workspace, workspace_bytes, M, K, N, group_size, kernel, grid_split_k, L);
}

int quick_w4a16_can_fuse_rmsnorm(int M, int K, int N, int group_size) {
  if (check_shapes(M, K, N, group_size) != QUICK_OK) return 0;
  // the deferred-zero skinny kernel copies (and tabulates) x per workgroup anyway: normalising on the way costs a
  // second pass over LDS, not a launch
  const Plan p = make_plan(M, K, N, group_size, QUICK_KERNEL_AUTO, 0);
  if (p.kernel == QUICK_KERNEL_LEAN) return p.lean_tmax > 0;   // x * weight in LDS on the way in, 1 / rms on the fp32 result
  if (!(p.kernel == QUICK_KERNEL_SKINNY && p.dz && p.ksplit == 1 && !p.frag8_t && (p.xlds || (p.mt >= 2 && p.waves == 8)))) return 0;
  // The fragment flavour pays for the norm in registers (188 against 134, the weight multiply and the squares on every x fragment of every
  // channel block): on a very large layer that is more than a separate launch of quick_rmsnorm_f16.  Llama-2-70B at bs = 16, one-session
  // A/B (profiles/r05_decode70_ab.txt): gate_up (8192 x 57344) un-fused 1412 -> 1456 tok/s, every fragment launch un-fused 1426; Mistral-7B's
  // 4096 x 28672 at bs = 32 LOSES 2 % un-fused.  So: fused up to K * N = 2^28.  QUICK_AMD_LN_FRAGMENT_MAX overrides the bound (A/B; 0 = never fuse).
  static const long ln_fragment_max = [] {
    const char* e = getenv("QUICK_AMD_LN_FRAGMENT_MAX");
    return e && *e ? atol(e) : (1L << 28);
  }();
  if (!p.xlds && (long)K * N > ln_fragment_max) return 0;
  return 1;
}

int quick_w4a16_plan_describe(int M, int K, int N, int group_size, int kernel, int grid_split_k, char* text, size_t text_bytes) {
  const int rc = check_shapes(M, K, N, group_size);
  if (rc != QUICK_OK) return rc;
  if (!text || text_bytes == 0) return fail(QUICK_ERR_INVALID_ARGUMENT, "no text buffer");
  const Plan p = make_plan(M, K, N, group_size, kernel, grid_split_k);
  if (p.kernel == QUICK_KERNEL_LEAN)
    snprintf(text, text_bytes, "lean ntw=%d waves=%d tiles_per_wave<=%d grid=%dx%d lds=%u workspace=0", p.mt, p.waves, p.lean_tmax, p.grid_x, (M + 15) / 16,
             lean_lds_need(M, K, p.waves, p.mt, false));
  else if (p.kernel == QUICK_KERNEL_XM)
    snprintf(text, text_bytes, "xm tokens=%d channels=%d waves=8 grid=%dx%d lds=%u workspace=0", p.wide_mb * 32, p.wide_pairs * 32, p.grid_x, p.ntiles / p.grid_x,
             xm_lds_need(p.wide_mb, p.wide_pairs));
  else if (p.kernel == QUICK_KERNEL_SKINNY)
    snprintf(text, text_bytes, "skinny ntw=%d waves=%d x=%s dequant=%s grid=%dx%dx%d ksplit=%d workspace=%zu", p.mt, p.waves,
             p.xlds ? "lds" : "l2", p.dz ? (p.xlds ? "deferred-zero-table" : "deferred-zero-fragment") : "exact", p.grid_x,
             (M + 15) / 16, p.ksplit, p.ksplit, workspace_need(p));
  else if (p.kernel == QUICK_KERNEL_XW)
    snprintf(text, text_bytes, "xw tokens=%d channels=%d waves=4 ring=%d queue=%d grid=%d slices=%d xcd_rows=%d workspace=%zu", p.wide_mb * 32, p.tch,
             p.wide_mb == 2 ? 8 : (p.wide_mb == 8 ? 2 : 4), p.wide_mb == 2 ? 8 : (p.wide_mb == 8 ? 2 : 4), p.ntiles * p.ksplit, p.ksplit, p.xcd_gm, workspace_need(p));
  else if (p.kernel == QUICK_KERNEL_XK)
    snprintf(text, text_bytes, "xk tokens=%d channels=128 waves=%d ring=%d queue=%d grid=%d slices=%d xcd_rows=%d workspace=%zu", p.wide_mb * 32,
             p.xk_loader ? 12 : (p.xk_kq == 4 ? 16 : 8), p.xk_loader ? 3 : p.xk_nbuf, p.xk_loader ? 5 : p.xk_wd, p.ntiles * p.ksplit, p.ksplit, p.xcd_gm, workspace_need(p));
  else if (p.kernel == QUICK_KERNEL_WIDE)
    snprintf(text, text_bytes, "wide tokens=%d channels=%d waves=%d ring=%d grid=%dx%d ksplit=%d xcd_rows=%d workspace=%zu", p.wide_mb * 32,
             p.tch, p.waves, p.wide_nbuf, p.ntiles, p.ksplit, p.ksplit, p.xcd_gm, workspace_need(p));
  else
    snprintf(text, text_bytes, "tiled tokens=%d channels=%d waves=%d grid=%dx%d ksplit=%d xcd_rows=%d workspace=%zu", p.mt * 16,
             p.tch, p.wn2 ? 8 : p.waves, p.ntiles, p.ksplit, p.ksplit, p.xcd_gm, workspace_need(p));
  if ((p.est_us > 0 || p.est_xw_us > 0) && getenv("QUICK_AMD_PLAN_ESTIMATES")) {   // (the launch-time models' estimates, for planner audits)
    const size_t n = strlen(text);
    if (n + 40 < text_bytes) snprintf(text + n, text_bytes - n, " est=%.1f est_xw=%.1f", p.est_us, p.est_xw_us);
  }
  return QUICK_OK;
}

int quick_w4a16_gemm_profile(const void* x, const void* const* qweights, const void* const* scales,
                             const void* const* qzeros, int n_sets, void* y, void* workspace, size_t workspace_bytes,
                             int M, int K, int N, int group_size, int kernel, int grid_split_k, int iters,
                             float* kernel_us, void* hip_stream) {
  if (n_sets < 1 || iters < 1 || !kernel_us) return fail(QUICK_ERR_INVALID_ARGUMENT, "bad profile arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  std::vector<hipEvent_t> ev(2 * (size_t)iters);
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) return fail(QUICK_ERR_LAUNCH, "hipEventCreate failed");
  int rc = QUICK_OK;
  for (int i = 0; i < iters && rc == QUICK_OK; ++i) {
    const int s = i % n_sets;
    const Launch L{st, ev[2 * i], ev[2 * i + 1]};
    rc = run_gemm(x, qweights[s], scales[s], qzeros[s], Fusion{}, y, workspace, workspace_bytes, M, K, N, group_size,
                  kernel, grid_split_k, L);
  }
  if (rc == QUICK_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(QUICK_ERR_LAUNCH, "stream synchronize failed");
  for (int i = 0; i < iters; ++i) {
    float ms = 0.f;
    if (rc == QUICK_OK && hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess)
      rc = fail(QUICK_ERR_LAUNCH, "hipEventElapsedTime failed");
    kernel_us[i] = ms * 1000.f;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

int quick_w4a16_gemm_span(const void* x, const void* const* qweights, const void* const* scales, const void* const* qzeros,
                          int n_sets, void* y, void* workspace, size_t workspace_bytes, int M, int K, int N, int group_size,
                          int kernel, int grid_split_k, int iters, float* span_us, void* hip_stream) {
  if (n_sets < 1 || iters < 1 || iters > 256 || !span_us) return fail(QUICK_ERR_INVALID_ARGUMENT, "bad span arguments (1..256 launches)");
  hipStream_t st = (hipStream_t)hip_stream;
  const size_t per = 2 * (size_t)kSpanWaves;  // u64 words per launch
  unsigned long long* dev = nullptr;
  if (hipMalloc(&dev, (size_t)iters * per * 8) != hipSuccess) return fail(QUICK_ERR_LAUNCH, "hipMalloc failed");
  int rc = QUICK_OK;
  for (int i = 0; i < iters && rc == QUICK_OK; ++i) {  // starts: all ones, ends: zero
    if (hipMemsetAsync(dev + i * per, 0xff, kSpanWaves * 8, st) != hipSuccess ||
        hipMemsetAsync(dev + i * per + kSpanWaves, 0, kSpanWaves * 8, st) != hipSuccess)
      rc = fail(QUICK_ERR_LAUNCH, "memset failed");
  }
  for (int i = 0; i < iters && rc == QUICK_OK; ++i) {
    const int s = i % n_sets;
    Launch L{st, nullptr, nullptr};
    L.span = dev + i * per;
    rc = run_gemm(x, qweights[s], scales[s], qzeros[s], Fusion{}, y, workspace, workspace_bytes, M, K, N, group_size, kernel,
                  grid_split_k, L);
  }
  std::vector<unsigned long long> host(per);
  for (int i = 0; i < iters; ++i) {
    span_us[i] = 0.f;
    if (rc != QUICK_OK) continue;
    if (hipMemcpyAsync(host.data(), dev + i * per, per * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
      rc = fail(QUICK_ERR_LAUNCH, "reading the span stamps back failed");
      continue;
    }
    unsigned long long lo = ~0ull, hi = 0ull;
    for (unsigned w = 0; w < kSpanWaves; ++w) {
      lo = std::min(lo, host[w]);
      hi = std::max(hi, host[kSpanWaves + w]);
    }
    span_us[i] = hi > lo ? (float)((double)(hi - lo) * 0.01) : 0.f;  // 100 MHz ticks
  }
  (void)hipFree(dev);
  return rc;
}

int quick_amd_dispatch_floor(int iters, float* kernel_us, void* hip_stream) {
  if (iters < 1 || !kernel_us) return fail(QUICK_ERR_INVALID_ARGUMENT, "bad arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  std::vector<hipEvent_t> ev(2 * (size_t)iters);
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) return fail(QUICK_ERR_LAUNCH, "hipEventCreate failed");
  for (int i = 0; i < iters; ++i)
    hipExtLaunchKernelGGL(w4a16_empty_kernel, dim3(256), dim3(512), 0, st, ev[2 * i], ev[2 * i + 1], 0, (unsigned*)nullptr);
  int rc = hipStreamSynchronize(st) == hipSuccess ? QUICK_OK : fail(QUICK_ERR_LAUNCH, "stream synchronize failed");
  for (int i = 0; i < iters; ++i) {
    float ms = 0.f;
    if (rc == QUICK_OK && hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess)
      rc = fail(QUICK_ERR_LAUNCH, "hipEventElapsedTime failed");
    kernel_us[i] = ms * 1000.f;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

int quick_w4a16_workspace_check(const void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!workspace || workspace_bytes == 0) return QUICK_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  // [r06, ADVICE r05] the scan allocates, copies back and synchronises: none of that is legal while the stream is being captured into a graph --
  // say so instead of invalidating the capture (QUICK_AMD_CHECK_WORKSPACE=1 reaches here from every split launch)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return fail(QUICK_ERR_INVALID_ARGUMENT, "quick_w4a16_workspace_check synchronises the stream and cannot run while it is being captured into a hipGraph (unset QUICK_AMD_CHECK_WORKSPACE for captured launches)");
  const size_t guarded = std::min(workspace_bytes, (size_t)kMaxSplitTiles * 4 + kXkZoneBytesHost) & ~(size_t)15;
  unsigned* first = nullptr;
  if (hipMalloc(&first, sizeof(unsigned)) != hipSuccess) return fail(QUICK_ERR_LAUNCH, "hipMalloc failed");
  int rc = QUICK_OK;
  unsigned host = 0xffffffffu;
  if (hipMemsetAsync(first, 0xff, sizeof(unsigned), st) != hipSuccess) rc = fail(QUICK_ERR_LAUNCH, "memset failed");
  if (rc == QUICK_OK) {
    hipLaunchKernelGGL(w4a16_workspace_scan_kernel, dim3(1024), dim3(256), 0, st, (const u32x4*)workspace, (unsigned)(guarded / 16), first);
    if (hipGetLastError() != hipSuccess) rc = fail(QUICK_ERR_LAUNCH, "workspace scan kernel did not launch");
    if (rc == QUICK_OK) if (hipMemcpyAsync(&host, first, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      rc = fail(QUICK_ERR_LAUNCH, "workspace scan failed: %s", hipGetErrorString(hipGetLastError()));
  }
  (void)hipFree(first);
  if (rc == QUICK_OK && host != 0xffffffffu)
    rc = fail(QUICK_ERR_WORKSPACE, "workspace is not zero at byte %u (%s): an aborted launch, a buffer that was never zeroed, or two streams sharing it; zero it again before the next call",
              host, (size_t)host < (size_t)kMaxSplitTiles * 4 ? "arrival counters / part state words" : "exchange zone");
  return rc;
}

int quick_w4a16_gemm_f16(const void* x, const void* qweight, const void* scales, const void* qzeros, void* y,
                         void* workspace, size_t workspace_bytes, int M, int K, int N, int group_size,
                         int split_k_iters, void* hip_stream) {
  if (split_k_iters < 1) return fail(QUICK_ERR_INVALID_ARGUMENT, "split_k_iters must be >= 1");
  return quick_w4a16_gemm_f16_ex(x, qweight, scales, qzeros, nullptr, y, workspace, workspace_bytes, M, K, N,
                                 group_size, QUICK_KERNEL_AUTO, 0, hip_stream);
}

int quick_dequantize_mi355x_f16(const void* qweight, const void* scales, const void* qzeros, void* w_out, int K,
                                int N, int group_size, void* hip_stream) {
  if (int rc = check_shapes(1, K, N, group_size)) return rc;
  hipLaunchKernelGGL(w4a16_dequant_kernel, dim3(N / 16, K / 128), dim3(64), 0, (hipStream_t)hip_stream,
                     (const u32x4*)qweight, (const half_t*)scales, (const uint32_t*)qzeros, (half_t*)w_out, K, N,
                     group_size);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QUICK_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
  return QUICK_OK;
}

}  // extern "C"
