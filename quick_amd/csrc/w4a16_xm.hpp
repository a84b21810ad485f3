// Mid-token kernels (r06): 17..128 tokens (AUTO: 17..64 on layers one round covers, 65..128 on narrow ones), the regime of the reference's _m32n128k32 / _m64n128k32 kernels and its
// compute_gemm_x2 (csrc/gemm_cuda_quick.cu:1293-1397, :458-1196) -- a dequantised weight fragment feeds EVERY token block of the tile.
//
// One workgroup = MB x 32 tokens x PR x 32 channels for the whole K (or a K slice); its eight waves split the k tiles.  Each wave is a
// self-contained stream: its own x ring in LDS (LDS-DMA, 16 rows x 64 B per instruction, a quarter stage = 32 k per slot), its own weight
// queue HBM -> VGPR, v_mfma_f32_32x32x16_f16 on exactly dequantised fragments (13 VALU per dword, every fragment feeds MB MFMAs), fp32
// accumulators for MB x PR tiles of 32 x 32.  No barrier inside the K loop: nobody reads anybody else's ring.  The loop is ONE generated
// inline-asm statement per (MB, PR) (tools/gen_xm_loop.py -> w4a16_xm_loop.inc): requests stay in flight around the back edge with counted
// waits, which hipcc's own wait insertion drains.
//
// Why this shape.  At 33..64 tokens x is as many L2 -> CU bytes as the weights of 256 channels: a workgroup that owns few channels is
// bound by the 64 B / clock its CU takes from L2 (x: tokens x K x 2 B per workgroup).  So x is fetched ONCE per workgroup (the r01-r05
// skinny launch fetched it per 16-token block and re-dequantised the weights per block: 1.41 x the algorithmic traffic, 8.6 VALU per
// MFMA), the channels per workgroup are chosen so that one round of workgroups covers the layer, and the weights never wait for x: both
// streams are in flight from the first instruction.
//
// After the loop the eight partial tiles meet in LDS (fp32, summed in wave order: results do not depend on timing): wave w finishes
// register pair w of every 32 x 32 tile -- channels 8 (w / 2) + 4 h + 2 (w % 2) + {0, 1} of the pair (SiLU * mul: registers j and j + 4,
// gate and up of one output channel).  Epilogues as the lean kernels: bias and residual in fp32 before the one rounding, SiLU * mul.
#pragma once
#include "w4a16_args.hpp"
#ifdef QA_XM_LOOP_INC   // (timing experiments: a loop file written by XM_EXP=<n> tools/gen_xm_loop.py, wrong results)
#include QA_XM_LOOP_INC
#else
#include "w4a16_xm_loop.inc"
#endif

namespace quick_amd {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct XmRest {
  const half_t* bias;
  const half_t* residual;
  half_t* Y;
  unsigned long long* span;
  unsigned long long* dbg;   // tools builds, ABL 64: per-wave phase stamps [workgroup][wave][16]
  int silu_mul;
};

// LDS of one workgroup: the eight rings (stages of MB x 8 KiB); the reduction buffer aliases them (32 KiB per 32 x 32 tile, at most
// four tiles per pass)
#ifndef QA_XM_RS
#define QA_XM_RS 1   // stages of a wave's ring in the 32-token configurations (the generator's XM_RS: A/B builds set both)
#endif
__host__ __device__ constexpr unsigned xm_ring_bytes(int mb) { return (unsigned)mb * 8192u * (mb == 1 ? (unsigned)QA_XM_RS : 1u); }   // per wave: one stage (32 tokens: QA_XM_RS)
__host__ __device__ constexpr unsigned xm_lds_bytes(int mb, int pr) {
  const unsigned ring = 8u * xm_ring_bytes(mb);
  const unsigned units = (unsigned)(mb * pr);
  const unsigned red = (units < 4u ? units : 4u) * 32768u;
  return ring > red ? ring : red;
}

template <int MB, int PR, bool STAMPED = false>
__device__ __forceinline__ void xm_run(floatx16 (&accr)[MB * PR], unsigned long long rsx_lo, unsigned long long rsx_hi, unsigned long long rsw_lo,
                                       unsigned long long rsw_hi, unsigned long long rss_lo, unsigned long long rss_hi, unsigned x_row, const unsigned (&x_chunk)[2], int m_last, int k2,
                                       unsigned w_voff, unsigned s_voff, unsigned xrd, unsigned xdst, int kb_tpg, int ke, unsigned w_pstride,
                                       unsigned s_pstride, unsigned t_voff, [[maybe_unused]] unsigned long long (&xm_t)[3]) {
#ifdef QUICK_AMD_TOOLS
  if constexpr (STAMPED) {
    if constexpr (MB == 2 && PR == 1) QA_XM_RUN_STAMPED_21();
    else if constexpr (MB == 2 && PR == 2) QA_XM_RUN_STAMPED_22();
    else if constexpr (MB == 2 && PR == 3) QA_XM_RUN_STAMPED_23();
    else if constexpr (MB == 1 && PR == 1) QA_XM_RUN_STAMPED_11();
    else if constexpr (MB == 1 && PR == 2) QA_XM_RUN_STAMPED_12();
    else QA_XM_RUN_STAMPED_13();
    return;
  }
#endif
  if constexpr (MB == 2 && PR == 1) QA_XM_RUN_21();
  else if constexpr (MB == 2 && PR == 2) QA_XM_RUN_22();
  else if constexpr (MB == 2 && PR == 3) QA_XM_RUN_23();
  else if constexpr (MB == 1 && PR == 1) QA_XM_RUN_11();
  else if constexpr (MB == 1 && PR == 2) QA_XM_RUN_12();
  else QA_XM_RUN_13();
}

// The waves' partial tiles -> LDS -> wave w sums register pair w of every tile (wave order 0..7) and stores it.
template <int MB, int PR, bool SILU>
__device__ __forceinline__ void xm_finish(const floatx16 (&accr)[MB * PR], char* smem, int lane, int wave, int m0, int nb, int pv, int aM, int aN, const XmRest& rest) {
  constexpr int U = MB * PR, UP = U < 4 ? U : 4;
  const int rho = lane & 31, h = lane >> 5;
  typedef float float2_t __attribute__((ext_vector_type(2)));
  float2_t* red = (float2_t*)smem;   // [unit][dst wave][src wave][lane]
#pragma unroll
  for (int u0 = 0; u0 < U; u0 += UP) {
    __syncthreads();   // the rings (first pass) / the previous pass's sums are done with
#pragma unroll
    for (int u = u0; u < u0 + UP && u < U; ++u)
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const int ra = SILU ? (w & 3) + 8 * (w >> 2) : 2 * w, rb = SILU ? ra + 4 : ra + 1;
        red[(((u - u0) * 8 + w) * 8 + wave) * 64 + lane] = float2_t{accr[u][ra], accr[u][rb]};
      }
    __syncthreads();
#pragma unroll
    for (int u = u0; u < u0 + UP && u < U; ++u) {
      float2_t sum = red[(((u - u0) * 8 + wave) * 8 + 0) * 64 + lane];
#pragma unroll
      for (int v = 1; v < 8; ++v) sum += red[(((u - u0) * 8 + wave) * 8 + v) * 64 + lane];
      const int p = u / MB, blk = u % MB;
      const int m = m0 + 32 * blk + rho;
      if (m < aM && p < pv) {
        if constexpr (SILU) {
          const int col = ((nb * PR + p) * 2 + (wave >> 2)) * 8 + 4 * h + (wave & 3);
          rest.Y[(size_t)m * (aN >> 1) + col] = silu_mul_f16((half_t)sum[0], (half_t)sum[1]);
        } else {
          const int n = (nb * PR + p) * 32 + 8 * (wave >> 1) + 4 * h + 2 * (wave & 1);
          float v0 = sum[0], v1 = sum[1];
          if (rest.bias) {
            const half2_t bv = *(const half2_t*)(rest.bias + n);
            v0 += (float)bv[0];
            v1 += (float)bv[1];
          }
          if (rest.residual) {
            const half2_t rv = *(const half2_t*)(rest.residual + (size_t)m * aN + n);
            v0 += (float)rv[0];
            v1 += (float)rv[1];
          }
          *(half2_t*)(rest.Y + (size_t)m * aN + n) = half2_t{(half_t)v0, (half_t)v1};
        }
      }
    }
  }
}

// grid: x = N / (32 PR) channel blocks (XCD-aware order), y = token tiles of MB * 32.  ABL: 32 = in-kernel span stamps; 64 (tools builds) = phase
// stamps per wave into rest.dbg: [entry, x(0, 0) landed | W(0) landed << 32, end of stage 0 | 1 << 32, end of stage 2 | 3 << 32 (low words), loop left, exit].
template <int MB, int PR, int ABL = 0>
__global__ __launch_bounds__(512) void w4a16_xm_kernel(const half_t* __restrict__ aX, const u32x4* __restrict__ aQW, const half_t* __restrict__ aS, int aM, int aK,
                                                       int aN, int tpg_log2, int gx, const XmRest rest) {
  if constexpr (ABL & 32) span_stamp(rest.span, 0);
  [[maybe_unused]] unsigned long long t_entry = 0, c_entry = 0, c_loop = 0;   // (c_: the shader clock counter, s_memtime -- core clocks against the 100 MHz stamps)
  if constexpr (ABL & 64) { t_entry = __builtin_amdgcn_s_memrealtime(); c_entry = __builtin_amdgcn_s_memtime(); }
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const unsigned rho = (unsigned)lane & 31u, h = (unsigned)lane >> 5;
  // workgroups are dealt to the XCDs round-robin: a contiguous run of channel blocks per XCD shares the lines of their group words
  const int nb = (gx & 7) == 0 ? ((int)blockIdx.x & 7) * (gx >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int m0 = (int)blockIdx.y * (MB * 32);
  const int KT = aK >> 7, NG = KT >> tpg_log2;
  const int kb = KT * wave / 8, ke = KT * (wave + 1) / 8;   // this wave's k tiles (none: its loads go through empty descriptors and it adds zeros)
  const int ct0 = nb * (2 * PR);                            // first 16-channel tile of the workgroup
  const int pv = min(PR, (aN >> 5) - nb * PR);              // channel pairs of this workgroup that exist (the last block of a ragged layer: loads of the
                                                            // others are out of the descriptors' range -- zeros, no traffic -- and nothing of them is stored)

  // buffer descriptors, spelled as two 64-bit halves (the loop keeps private copies whose record count it switches to 0 for stages past the end)
  const unsigned long long flags = 0x00020000ull << 32;
  const int rows = min(aM - m0, MB * 32);
  const unsigned long long rsx_lo = (unsigned long long)(uintptr_t)(aX + (size_t)m0 * aK) & 0xffffffffffffull;
  const unsigned long long rsx_hi = (unsigned long long)((unsigned)rows * (unsigned)aK * 2u) | flags;
  const unsigned long long rsw_lo = (unsigned long long)(uintptr_t)(aQW + (size_t)ct0 * KT * 64) & 0xffffffffffffull;
  const unsigned long long rsw_hi = (unsigned long long)((unsigned)(2 * pv) * (unsigned)KT * 1024u) | flags;
  const unsigned long long rss_lo = (unsigned long long)(uintptr_t)((const uint32_t*)aS + (size_t)ct0 * NG * 16) & 0xffffffffffffull;
  const unsigned long long rss_hi = (unsigned long long)((unsigned)(2 * pv) * (unsigned)NG * 64u) | flags;
  // x pieces: instruction i of a half stage = rows 8 i .. 8 i + 7 of the tile, 128 bytes (64 k) of each: lane p = row p / 8, 16-byte chunk
  // (p % 8) ^ ((row / 2) % 8) -- the swizzle that makes the ds_read_b128 of the 32 x 16 fragments conflict-free (row t reads chunk c at slot
  // c ^ ((t / 2) % 8)).  The loop's prologue makes the piece offsets from the lane's row, its chunk for even / odd pieces, the tile's last row
  // (rows past M replay it; never stored) and the row pitch.
  const unsigned x_row = (unsigned)lane >> 3;
  const unsigned x_chunk[2] = {16u * ((((unsigned)lane & 7u) ^ ((unsigned)lane >> 4)) & 7u), 16u * ((((unsigned)lane & 7u) ^ (4u + ((unsigned)lane >> 4))) & 7u)};
  const int m_last = uniform(rows - 1), k2 = uniform(aK * 2);
  const unsigned w_voff = (rho >> 4) * (unsigned)KT * 1024u + 16u * ((rho & 15u) + 16u * h);
  const unsigned s_voff = (rho >> 4) * (unsigned)NG * 64u + 4u * (rho & 15u);
  const unsigned w_pstride = 2u * (unsigned)KT * 1024u, s_pstride = 2u * (unsigned)NG * 64u;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned xdst = lds_base + (unsigned)wave * xm_ring_bytes(MB);
  const unsigned xrd = xdst + (rho >> 3) * 1024u + (8u * (rho & 7u) + (h ^ ((rho >> 1) & 7u))) * 16u;
  const int kb_tpg = uniform(kb | (tpg_log2 << 24)), ke_u = uniform(ke);
  // line touches (the loop's prologue): lane l asks for one dword of line l of a tile's k range -- 8 lines per k tile, the first 8 k tiles of the wave
  const unsigned t_voff = lane < 8 * min(ke - kb, 8) ? (unsigned)lane * 128u : 0x80000000u;

  floatx16 accr[MB * PR];
  unsigned long long xm_t[3] = {0ull, 0ull, 0ull};
  {
    const int ke = ke_u;
    xm_run<MB, PR, (ABL & 64) != 0>(accr, rsx_lo, rsx_hi, rsw_lo, rsw_hi, rss_lo, rss_hi, x_row, x_chunk, m_last, k2, w_voff, s_voff, xrd, xdst, kb_tpg, ke, w_pstride, s_pstride, t_voff, xm_t);
  }
  [[maybe_unused]] unsigned long long t_loop = 0;
  if constexpr (ABL & 64) { t_loop = __builtin_amdgcn_s_memrealtime(); c_loop = __builtin_amdgcn_s_memtime(); }
  if (rest.silu_mul) xm_finish<MB, PR, true>(accr, smem, lane, wave, m0, nb, pv, aM, aN, rest);
  else xm_finish<MB, PR, false>(accr, smem, lane, wave, m0, nb, pv, aM, aN, rest);
  if constexpr (ABL & 32) span_stamp(rest.span, 1);
  if constexpr (ABL & 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_exit = __builtin_amdgcn_s_memrealtime();
    if (rest.dbg != nullptr && lane == 0) {
      unsigned long long* o = rest.dbg + ((size_t)((blockIdx.y * gridDim.x + blockIdx.x) & 511u) * 8 + wave) * 8;
      o[0] = t_entry; o[1] = xm_t[0]; o[2] = xm_t[1]; o[3] = xm_t[2]; o[4] = t_loop; o[5] = t_exit; o[6] = c_entry; o[7] = c_loop;
    }
  }
}

}  // namespace quick_amd
