// Glue kernels of the decode step around the W4A16 GEMMs (callers of the hot path):
// RMSNorm, RoPE + KV-cache append, single-query attention, SiLU*mul.  The reference gets these from out-of-tree
// extensions (awq_ext.layernorm_forward_cuda, quick/awq/modules/fused/norm.py:18; awq_ft_ext.single_query_attention,
// quick/awq/modules/fused/attn.py:217) or from eager torch (attn.py:166-210); here they are small HBM/latency-bound
// HIP kernels so that one decode layer is 9 launches instead of ~40.  fp16 in/out, fp32 arithmetic.
#include <cstdlib>
#include "w4a16_common.hpp"
#include "../../include/quick_amd.h"

namespace quick_amd {

// y[r, :] = x[r, :] * rsqrt(mean(x[r, :]^2) + eps) * w      one 256-thread workgroup per row, H % 8 == 0
// [r05] NIT > 0 (H <= NIT * 2048): the row and the weight are requested ONCE, together, and stay in registers across the barrier -- one
// trip to memory per launch instead of three dependent ones (x, then x again and w behind the barrier).  Same arithmetic and rounding
// points either way.  Measured: 5.29 -> 5.17 us per launch in the bs = 64 decode trace (two per layer, ~6.5 % of the step) -- the second
// and third trips were cache hits already; what the launch costs is its place in the chain of dependent launches, not its work.
template <int NIT>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                      half_t* __restrict__ y, int H, float eps) {
  __shared__ float part[4];
  const half_t* xr = x + (size_t)blockIdx.x * H;
  half_t* yr = y + (size_t)blockIdx.x * H;
  float ss = 0.f;
  if constexpr (NIT > 0) {
    half8_t v[NIT], g[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = min((int)threadIdx.x * 8 + k * 2048, H - 8);   // (past the row: its last chunk again, not summed, not stored)
      v[k] = *(const half8_t*)(xr + i);
      g[k] = *(const half8_t*)(w + i);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if ((int)threadIdx.x * 8 + k * 2048 < H) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += (float)v[k][j] * (float)v[k][j];
      }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float inv = rsqrtf((part[0] + part[1] + part[2] + part[3]) / H + eps);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = (int)threadIdx.x * 8 + k * 2048;
      if (i < H) {
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)((half_t)((float)v[k][j] * inv) * g[k][j]);  // cast, then scale: torch order
        *(half8_t*)(yr + i) = o;
      }
    }
  } else {
    for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
      const half8_t v = *(const half8_t*)(xr + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float inv = rsqrtf((part[0] + part[1] + part[2] + part[3]) / H + eps);
    for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
      const half8_t v = *(const half8_t*)(xr + i), g = *(const half8_t*)(w + i);
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)((half_t)((float)v[j] * inv) * g[j]);  // cast, then scale: torch order
      *(half8_t*)(yr + i) = o;
    }
  }
}

// Decode step (one new token per sequence).  qkv [B, (nh + 2 nkv) * D] as produced by the fused qkv GEMM; rotates q
// and k by the angle table row `*pos` (HF rotate-half convention), writes q to q_out [B, nh, D] and appends k, v to
// the caches [B, nkv, L, D] at position *pos.  One workgroup per (sequence, head slot), D/2 active lanes.
__global__ __launch_bounds__(64) void rope_kv_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ cos_t,
                                                     const half_t* __restrict__ sin_t, const long* __restrict__ pos,
                                                     half_t* __restrict__ q_out, half_t* __restrict__ k_cache,
                                                     half_t* __restrict__ v_cache, int nh, int nkv, int D, int L) {
  const int b = blockIdx.y, slot = blockIdx.x;  // slot: 0..nh-1 q heads, nh..nh+nkv-1 k heads, then v heads
  const long p = pos[0];
  const half_t* src = qkv + (size_t)b * (nh + 2 * nkv) * D + (size_t)slot * D;
  const int i = threadIdx.x;
  if (i >= D / 2) return;
  if (slot < nh + nkv) {
    const float x0 = (float)src[i], x1 = (float)src[i + D / 2];
    const float c0 = (float)cos_t[p * D + i], s0 = (float)sin_t[p * D + i];
    const float c1 = (float)cos_t[p * D + i + D / 2], s1 = (float)sin_t[p * D + i + D / 2];
    // torch reference: x * cos + rotate_half(x) * sin, every product and the sum rounded to fp16
    const half_t r0 = (half_t)((float)(half_t)(x0 * c0) + (float)(half_t)(-x1 * s0));
    const half_t r1 = (half_t)((float)(half_t)(x1 * c1) + (float)(half_t)(x0 * s1));
    half_t* dst = slot < nh ? q_out + ((size_t)b * nh + slot) * D
                            : k_cache + (((size_t)b * nkv + (slot - nh)) * L + p) * D;
    dst[i] = r0;
    dst[i + D / 2] = r1;
  } else {
    half_t* dst = v_cache + (((size_t)b * nkv + (slot - nh - nkv)) * L + p) * D;
    dst[i] = src[i];
    dst[i + D / 2] = src[i + D / 2];
  }
}

// Prefill form of the kernel above: T consecutive tokens per sequence, positions *pos0 .. *pos0 + T - 1.  qkv [B * T, (nh + 2 nkv) * D]
// (row b * T + t); q goes to q_out [B, nh, T, D] (the layout attention wants), k and v to the caches at their positions.
__global__ __launch_bounds__(64) void rope_kv_prefill_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ cos_t,
                                                             const half_t* __restrict__ sin_t, const long* __restrict__ pos0,
                                                             half_t* __restrict__ q_out, half_t* __restrict__ k_cache,
                                                             half_t* __restrict__ v_cache, int T, int nh, int nkv, int D, int L) {
  const int bt = blockIdx.y, b = bt / T, t = bt % T, slot = blockIdx.x;
  const long p = pos0[0] + t;
  const half_t* src = qkv + (size_t)bt * (nh + 2 * nkv) * D + (size_t)slot * D;
  const int i = threadIdx.x;
  if (i >= D / 2) return;
  if (slot < nh + nkv) {
    const float x0 = (float)src[i], x1 = (float)src[i + D / 2];
    const float c0 = (float)cos_t[p * D + i], s0 = (float)sin_t[p * D + i];
    const float c1 = (float)cos_t[p * D + i + D / 2], s1 = (float)sin_t[p * D + i + D / 2];
    const half_t r0 = (half_t)((float)(half_t)(x0 * c0) + (float)(half_t)(-x1 * s0));
    const half_t r1 = (half_t)((float)(half_t)(x1 * c1) + (float)(half_t)(x0 * s1));
    half_t* dst = slot < nh ? q_out + (((size_t)b * nh + slot) * T + t) * D
                            : k_cache + (((size_t)b * nkv + (slot - nh)) * L + p) * D;
    dst[i] = r0;
    dst[i + D / 2] = r1;
  } else {
    half_t* dst = v_cache + (((size_t)b * nkv + (slot - nh - nkv)) * L + p) * D;
    dst[i] = src[i];
    dst[i + D / 2] = src[i + D / 2];
  }
}

// Single-query attention over positions 0..*pos (inclusive), GQA aware.  One 256-thread workgroup per
// (sequence, query head); D == 128.  Scores in fp32, two passes over K then V from HBM/L2 (the cache of one head at a
// few hundred positions is tens of KB).  out [B, nh * D].
__global__ __launch_bounds__(256) void decode_attention_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k_cache,
                                                               const half_t* __restrict__ v_cache, const long* __restrict__ pos,
                                                               half_t* __restrict__ out, int nh, int nkv, int L, float scale) {
  constexpr int D = 128;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sc = (float*)smem_raw;             // [len] scores / probabilities
  float* red = sc + ((L + 3) & ~3);         // [4] per-wave partials, then [4][D] output partials
  const int b = blockIdx.y, h = blockIdx.x, kvh = h / (nh / nkv);
  const int len = (int)pos[0] + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const half_t* qp = q + ((size_t)b * nh + h) * D;
  const half_t* kp = k_cache + ((size_t)b * nkv + kvh) * L * D;
  const half_t* vp = v_cache + ((size_t)b * nkv + kvh) * L * D;

  // scores: 16 lanes per key row (8 fp16 each), 4 rows per wave per step
  const int sub = lane & 15, rsel = lane >> 4;
  const half8_t qv = *(const half8_t*)(qp + sub * 8);
  // scores: a wave takes 4 key rows per step (16 lanes x 16 B each); the loads of UNR steps are issued together so
  // that one L2/HBM round trip serves 4*UNR rows instead of one per row
  constexpr int UNR = 8;
  float mx = -INFINITY;
  for (int t0 = wave * 4; t0 < len; t0 += 16 * UNR) {
    half8_t kv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = min(t0 + 16 * u + rsel, len - 1);
      kv[u] = *(const half8_t*)(kp + (size_t)t * D + sub * 8);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = t0 + 16 * u + rsel;
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d += (float)qv[j] * (float)kv[u][j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) d += __shfl_xor(d, o);
      d *= scale;
      if (t < len && sub == 0) sc[t] = d;
      if (t < len) mx = fmaxf(mx, d);
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int t = threadIdx.x; t < len; t += 256) {
    const float e = __expf(sc[t] - mx);
    sc[t] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();                          // everyone has read red[] (max) before it is reused
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  __syncthreads();

  // out = sum_t p[t] * v[t, :], same row-per-16-lanes mapping and batched loads; lane accumulates 8 channels
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int t0 = wave * 4; t0 < len; t0 += 16 * UNR) {
    half8_t vv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = min(t0 + 16 * u + rsel, len - 1);
      vv[u] = *(const half8_t*)(vp + (size_t)t * D + sub * 8);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = t0 + 16 * u + rsel;
      const float pt = t < len ? sc[t] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += pt * (float)vv[u][j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[j] += __shfl_xor(acc[j], 16);
    acc[j] += __shfl_xor(acc[j], 32);
  }
  float* part = red;                        // [4][D]
  if (rsel == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) part[wave * D + sub * 8 + j] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x < D) {
    const float v = (part[threadIdx.x] + part[D + threadIdx.x] + part[2 * D + threadIdx.x] + part[3 * D + threadIdx.x]) * inv;
    out[((size_t)b * nh + h) * D + threadIdx.x] = (half_t)v;
  }
}

// RoPE + KV append + single-query attention in ONE launch (decode step, D == 128).  Workgroup (sequence b, query head
// h) rotates its own q and -- redundantly per query head of a GQA group -- the new k of its kv head in registers, scores
// cache positions 0..*pos-1 from HBM/L2 and position *pos from those registers, and the group's first head appends
// the rotated k and the v to the caches.  Nobody reads cache position *pos in this launch, so there is no ordering
// problem between the workgroups of a group.
__device__ __forceinline__ void rope8(const half8_t x, const half_t* __restrict__ cos_row, const half_t* __restrict__ sin_row,
                                      int sub, float (&r)[8]) {
  const half8_t c = *(const half8_t*)(cos_row + sub * 8), sn = *(const half8_t*)(sin_row + sub * 8);
  const float sign = sub < 8 ? -1.f : 1.f;  // rotate_half: dims 0..63 pair with -x[i+64], dims 64..127 with +x[i-64]
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float xj = (float)x[j];
    const float partner = __shfl_xor(xj, 8);  // lane sub^8 of the same 16-lane group holds the paired dims
    r[j] = (float)(half_t)((float)(half_t)(xj * (float)c[j]) + (float)(half_t)(sign * partner * (float)sn[j]));
  }
}

// Online softmax: K and V rows of a batch are requested together and consumed in
// one sweep, every (wave, 16-lane row slot) keeps its own running (max, sum, 8 output dims per lane) and the 16 slots of
// the workgroup are merged once at the end -- one barrier in the whole kernel.  (The first version of this kernel made two
// passes -- all scores into LDS, softmax, then all V -- with five barriers; with every workgroup of a large batch starting at
// once its "all K, then all V" phases left HBM idle about half the time, 3.9 TB/s at bs=64, and the single pass is ahead at
// every batch size: +8 % decode tok/s at bs=1, +1 % at bs=32, +0.5 % at bs=64..128 [r01].)
// [r05] WAVES x 4 row slots, UNR rows per slot and trip.  Few workgroups (batch x heads <= 128: decode at bs <= 4) run eight waves, so
// that 256 positions are ONE trip -- one round of requests instead of two dependent ones -- and the first trip's requests no longer wait
// for *pos: rows are clamped to the cache's end instead of to *pos - 1, and rows at or behind *pos are masked out of scores AND of the
// value sum (their bytes may be anything).  Later trips clamp to *pos - 1 again (re-reading one row instead of fetching rows nobody needs).
template <int WAVES, int UNR>
__global__ __launch_bounds__(WAVES * 64) void decode_rope_attention_flash_kernel(
    const half_t* __restrict__ qkv, const half_t* __restrict__ cos_t, const half_t* __restrict__ sin_t,
    const long* __restrict__ pos, half_t* __restrict__ k_cache, half_t* __restrict__ v_cache, half_t* __restrict__ out,
    int nh, int nkv, int L, float scale) {
  constexpr int D = 128;
  constexpr int SLOT = WAVES * 4;       // rows one load instruction of the workgroup covers
  __shared__ float part[WAVES][D + 2];  // per wave: 128 output dims, running max, running sum
  const int b = blockIdx.y, h = blockIdx.x, group = nh / nkv, kvh = h / group;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 15, rsel = lane >> 4;
  const half_t* row = qkv + (size_t)b * (nh + 2 * nkv) * D;
  const half_t* kp = k_cache + ((size_t)b * nkv + kvh) * L * D + sub * 8;
  const half_t* vp = v_cache + ((size_t)b * nkv + kvh) * L * D + sub * 8;
  const half8_t qraw = *(const half8_t*)(row + (size_t)h * D + sub * 8);
  const half8_t kraw = *(const half8_t*)(row + (size_t)(nh + kvh) * D + sub * 8);
  const half8_t vn = *(const half8_t*)(row + (size_t)(nh + nkv + kvh) * D + sub * 8);
  const int p = (int)pos[0];  // cache rows 0..p-1 are attended from memory, row p (this token) from registers
  __builtin_amdgcn_sched_barrier(0);  // (the request for *pos leaves here, its wait sits behind the first trip's cache requests)

  const float LOG2E = 1.44269504088896f;
  float m = -INFINITY, l = 0.f;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float qr[8], kr[8];
  // [r05] Software-pipelined like the grouped-query kernel below: two register sets of UNR rows of K and of V per slot, the requests for
  // trip t + 1 leave before trip t is consumed.  The FIRST trip's requests leave before *pos is known (rows clamped to the cache's end,
  // rows at or behind *pos masked out of scores and value sum: their bytes may be anything); later trips clamp to *pos - 1 (re-reading
  // one row instead of fetching rows nobody needs).
  auto request = [&](half8_t (&kv)[UNR], half8_t (&vv)[UNR], int tb, int last) __attribute__((always_inline)) {
    const int t0 = tb + rsel;
#pragma unroll
    for (int u = 0; u < UNR; ++u) kv[u] = *(const half8_t*)(kp + (size_t)min(t0 + SLOT * u, last) * D);
#pragma unroll
    for (int u = 0; u < UNR; ++u) vv[u] = *(const half8_t*)(vp + (size_t)min(t0 + SLOT * u, last) * D);
  };
  // scores in the log2 domain (q carries scale * log2 e): UNR rows of this slot
  auto consume = [&](const half8_t (&kv)[UNR], const half8_t (&vv)[UNR], int tb) __attribute__((always_inline)) {
    const int t0 = tb + rsel;
    float d[UNR], mb = m;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float x = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) x += qr[j] * (float)kv[u][j];
      x = lanes_sum<16>(x);
      d[u] = t0 + SLOT * u < p ? x : -INFINITY;
      mb = fmaxf(mb, d[u]);
    }
    const float mref = mb == -INFINITY ? 0.f : mb;
    const float corr = exp2f(m - mref);
    l *= corr;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= corr;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (t0 + SLOT * u < p) {  // (a row behind the sequence holds anything, NaN included: 0 * NaN would poison the sum)
        const float e = exp2f(d[u] - mref);
        l += e;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += e * (float)vv[u][j];
      }
    }
    m = mb;
  };
  half8_t kva[UNR], vva[UNR], kvb[UNR], vvb[UNR];
  int tb = uniform(wave * 4);
  request(kva, vva, tb, L - 1);
  __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise hoists the wait for *pos and the cos / sin rows in front of these requests)
  {  // rotate q and the new k while the cache rows are in flight
    const half8_t cs = *(const half8_t*)(cos_t + (size_t)p * D + sub * 8), sn = *(const half8_t*)(sin_t + (size_t)p * D + sub * 8);
    const float sign = sub < 8 ? -1.f : 1.f;  // rotate_half: dims 0..63 pair with -x[i+64], dims 64..127 with +x[i-64]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float qj = (float)qraw[j], kj = (float)kraw[j];
      const float qp = __shfl_xor(qj, 8), kpn = __shfl_xor(kj, 8);  // lane sub^8 of the same row slot holds the paired dims
      qr[j] = (float)(half_t)((float)(half_t)(qj * (float)cs[j]) + (float)(half_t)(sign * qp * (float)sn[j])) * (scale * LOG2E);
      kr[j] = (float)(half_t)((float)(half_t)(kj * (float)cs[j]) + (float)(half_t)(sign * kpn * (float)sn[j]));
    }
    if (threadIdx.x < 16 && h % group == 0) {  // append to the caches (position p is not used by anyone in this launch)
      half8_t kh;
#pragma unroll
      for (int j = 0; j < 8; ++j) kh[j] = (half_t)kr[j];
      *(half8_t*)(k_cache + (((size_t)b * nkv + kvh) * L + p) * D + sub * 8) = kh;
      *(half8_t*)(v_cache + (((size_t)b * nkv + kvh) * L + p) * D + sub * 8) = vn;
    }
  }
  if (tb < p) {  // (else: nothing for this wave -- first position: nothing in the cache yet)
    constexpr int STEP = SLOT * UNR;
    // (Requests behind a wave-uniform guard: hipcc's s_waitcnt pass then assumes the smaller in-flight count and waits for part of the
    // set just requested in front of the value sum of the set in hand.  The unconditional form -- clamped rows past the sequence -- counts
    // exactly, but the same construction in the skinny GEMM kernel's chunk loop produced wrong results that -amdgpu-waitcnt-forcezero
    // cured: this loop stays with the form the parity tests have seen.)
    while (true) {  // wave-uniform trip counts
      const bool more_b = tb + STEP < p;
      if (more_b) request(kvb, vvb, tb + STEP, p - 1);
      consume(kva, vva, tb);
      if (!more_b) break;
      const bool more_a = tb + 2 * STEP < p;
      if (more_a) request(kva, vva, tb + 2 * STEP, p - 1);
      consume(kvb, vvb, tb + STEP);
      if (!more_a) break;
      tb += 2 * STEP;
    }
  }
  if (wave == 0 && rsel == 0) {  // this token (row p), from registers
    float x = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) x += qr[j] * kr[j];
    x = lanes_sum<16>(x);
    const float mb = fmaxf(m, x), corr = exp2f(m - mb), e = exp2f(x - mb);
    l = l * corr + e;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = acc[j] * corr + e * (float)vn[j];
    m = mb;
  }
  // merge the 4 row slots of the wave, then the waves through LDS
  float mw = fmaxf(m, __shfl_xor(m, 16));
  mw = fmaxf(mw, __shfl_xor(mw, 32));
  const float sc_ = exp2f(m - (mw == -INFINITY ? 0.f : mw));
  l *= sc_;
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[j] *= sc_;
    acc[j] += __shfl_xor(acc[j], 16);
    acc[j] += __shfl_xor(acc[j], 32);
  }
  if (rsel == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) part[wave][sub * 8 + j] = acc[j];
    if (sub == 0) {
      part[wave][D] = mw;
      part[wave][D + 1] = l;
    }
  }
  __syncthreads();
  if (threadIdx.x < D) {
    float M = part[0][D];  // finite: wave 0 holds row p
#pragma unroll
    for (int w = 1; w < WAVES; ++w) M = fmaxf(M, part[w][D]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const float f = exp2f(part[w][D] - M);
      num += part[w][threadIdx.x] * f;
      den += part[w][D + 1] * f;
    }
    out[((size_t)b * nh + h) * D + threadIdx.x] = (half_t)(num / den);
  }
}

// Grouped-query form of the single-pass kernel: one workgroup per (sequence, KV head) serves all GROUP query heads of
// that KV head from ONE sweep over its K and V rows.  With one workgroup per query head every head of a group re-reads
// the same cache rows through L2 and the launch is bound by that traffic, not by HBM: Mistral-7B shapes (32 query / 8 KV
// heads) at bs=64, 190 positions took 33 us per layer for 50 MB of cache [r01, tools/time_attention.py].
template <int GROUP, int UNR>
__global__ __launch_bounds__(256, 2) void decode_rope_attention_gqa_kernel(
    const half_t* __restrict__ qkv, const half_t* __restrict__ cos_t, const half_t* __restrict__ sin_t,
    const long* __restrict__ pos, half_t* __restrict__ k_cache, half_t* __restrict__ v_cache, half_t* __restrict__ out,
    int nh, int nkv, int L, float scale, int gsplit) {
  constexpr int D = 128;
  __shared__ float part[4][GROUP][D + 2];  // per wave and query head: 128 output dims, running max, running sum
  // gsplit workgroups share a KV head, GROUP query heads each (8 heads per KV head = 2 x 4: the 8-head instantiation spills)
  const int b = blockIdx.y, kvh = blockIdx.x / gsplit, hbase = kvh * (GROUP * gsplit) + (blockIdx.x % gsplit) * GROUP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 15, rsel = lane >> 4;
  const half_t* row = qkv + (size_t)b * (nh + 2 * nkv) * D;
  const half_t* kp = k_cache + ((size_t)b * nkv + kvh) * L * D + sub * 8;
  const half_t* vp = v_cache + ((size_t)b * nkv + kvh) * L * D + sub * 8;
  half8_t qraw[GROUP];
#pragma unroll
  for (int g = 0; g < GROUP; ++g) qraw[g] = *(const half8_t*)(row + (size_t)(hbase + g) * D + sub * 8);
  const half8_t kraw = *(const half8_t*)(row + (size_t)(nh + kvh) * D + sub * 8);
  const half8_t vn = *(const half8_t*)(row + (size_t)(nh + nkv + kvh) * D + sub * 8);
  const int p = (int)pos[0];  // cache rows 0..p-1 are attended from memory, row p (this token) from registers
  __builtin_amdgcn_sched_barrier(0);  // (the request for *pos leaves here, its wait sits behind the first trip's cache requests)

  const float LOG2E = 1.44269504088896f;
  float m[GROUP], l[GROUP], acc[GROUP][8], qr[GROUP][8], kr[8];
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
    m[g] = -INFINITY;
    l[g] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
  }
  // [r05] The sweep is software-pipelined: the requests for trip t + 1 leave BEFORE trip t is consumed (two register sets of UNR rows of
  // K and of V each), so that a wave always has cache rows in flight while it computes.  The one-set form (load a trip, wait, consume,
  // load the next) left the memory pipe empty during the arithmetic and the ALUs idle during the flight: 24.6 us per launch at bs = 64,
  // 8 KV heads, ~190 positions -- 50 MB, 2.0 TB/s (profiles/r05_decode_trace.txt).
  const int plast = max(p - 1, 0);
  auto request = [&](half8_t (&kv)[UNR], half8_t (&vv)[UNR], int tb) __attribute__((always_inline)) {
    const int t0 = tb + rsel;
#pragma unroll
    for (int u = 0; u < UNR; ++u) kv[u] = *(const half8_t*)(kp + (size_t)min(t0 + 16 * u, plast) * D);
#pragma unroll
    for (int u = 0; u < UNR; ++u) vv[u] = *(const half8_t*)(vp + (size_t)min(t0 + 16 * u, plast) * D);
  };
  // scores in the log2 domain (q carries scale * log2 e): UNR rows of this slot x GROUP heads; every cache row is converted to fp32 once
  // for all heads
  auto consume = [&](const half8_t (&kv)[UNR], const half8_t (&vv)[UNR], int tb) __attribute__((always_inline)) {
    const int t0 = tb + rsel;
    float d[GROUP][UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float kf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) kf[j] = (float)kv[u][j];
      const bool valid = t0 + 16 * u < p;
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        float x = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) x += qr[g][j] * kf[j];
        x = lanes_sum<16>(x);
        d[g][u] = valid ? x : -INFINITY;
      }
    }
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
      float mb = m[g];
#pragma unroll
      for (int u = 0; u < UNR; ++u) mb = fmaxf(mb, d[g][u]);
      const float mref = mb == -INFINITY ? 0.f : mb;
      const float corr = exp2f(m[g] - mref);
      l[g] *= corr;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[g][j] *= corr;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        d[g][u] = exp2f(d[g][u] - mref);
        l[g] += d[g][u];
      }
      m[g] = mb;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float vf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) vf[j] = (float)vv[u][j];
#pragma unroll
      for (int g = 0; g < GROUP; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] += d[g][u] * vf[j];
    }
  };
  half8_t kva[UNR], vva[UNR], kvb[UNR], vvb[UNR];
  int tb = uniform(wave * 4);
  request(kva, vva, tb);
  {  // rotate the q heads and the new k while the first cache rows are in flight
    const half8_t cs = *(const half8_t*)(cos_t + (size_t)p * D + sub * 8), sn = *(const half8_t*)(sin_t + (size_t)p * D + sub * 8);
    const float sign = sub < 8 ? -1.f : 1.f;  // rotate_half: dims 0..63 pair with -x[i+64], dims 64..127 with +x[i-64]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float kj = (float)kraw[j], kpn = __shfl_xor(kj, 8);  // lane sub^8 of the same row slot holds the paired dims
      kr[j] = (float)(half_t)((float)(half_t)(kj * (float)cs[j]) + (float)(half_t)(sign * kpn * (float)sn[j]));
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        const float qj = (float)qraw[g][j], qp = __shfl_xor(qj, 8);
        qr[g][j] = (float)(half_t)((float)(half_t)(qj * (float)cs[j]) + (float)(half_t)(sign * qp * (float)sn[j])) * (scale * LOG2E);
      }
    }
    if (threadIdx.x < 16 && blockIdx.x % gsplit == 0) {  // append to the caches (position p is not read in this launch)
      half8_t kh;
#pragma unroll
      for (int j = 0; j < 8; ++j) kh[j] = (half_t)kr[j];
      *(half8_t*)(k_cache + (((size_t)b * nkv + kvh) * L + p) * D + sub * 8) = kh;
      *(half8_t*)(v_cache + (((size_t)b * nkv + kvh) * L + p) * D + sub * 8) = vn;
    }
  }
  if (p > 0) {  // (first position: nothing in the cache yet -- row 0 is unwritten, and 0 * NaN would poison the sum)
    constexpr int STEP = 16 * UNR;
    while (true) {  // wave-uniform trip counts
      const bool more_b = tb + STEP < p;
      if (more_b) request(kvb, vvb, tb + STEP);
      consume(kva, vva, tb);
      if (!more_b) break;
      const bool more_a = tb + 2 * STEP < p;
      if (more_a) request(kva, vva, tb + 2 * STEP);
      consume(kvb, vvb, tb + STEP);
      if (!more_a) break;
      tb += 2 * STEP;
    }
  }
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
    if (wave == 0 && rsel == 0) {  // this token (row p), from registers
      float x = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) x += qr[g][j] * kr[j];
      x = lanes_sum<16>(x);
      const float mb = fmaxf(m[g], x), corr = exp2f(m[g] - mb), e = exp2f(x - mb);
      l[g] = l[g] * corr + e;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[g][j] = acc[g][j] * corr + e * (float)vn[j];
      m[g] = mb;
    }
    // merge the 4 row slots of the wave, then the 4 waves through LDS
    float mw = fmaxf(m[g], __shfl_xor(m[g], 16));
    mw = fmaxf(mw, __shfl_xor(mw, 32));
    const float sc_ = exp2f(m[g] - (mw == -INFINITY ? 0.f : mw));
    float lw = l[g] * sc_;
    lw += __shfl_xor(lw, 16);
    lw += __shfl_xor(lw, 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = acc[g][j] * sc_;
      a += __shfl_xor(a, 16);
      a += __shfl_xor(a, 32);
      if (rsel == 0) part[wave][g][sub * 8 + j] = a;
    }
    if (lane == 0) {
      part[wave][g][D] = mw;
      part[wave][g][D + 1] = lw;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GROUP * D; i += 256) {
    const int g = i / D, dd = i % D;
    const float M = fmaxf(fmaxf(part[0][g][D], part[1][g][D]), fmaxf(part[2][g][D], part[3][g][D]));  // finite: row p
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = exp2f(part[w][g][D] - M);
      num += part[w][g][dd] * f;
      den += part[w][g][D + 1] * f;
    }
    out[((size_t)b * nh + hbase + g) * D + dd] = (half_t)(num / den);
  }
}

// [r05] Grouped-query sweep with the SCORES on the matrix core.  The kernel above spends its time in the vector ALUs (18 us per launch at
// bs = 64, 8 KV heads, ~190 positions: 50 MB at 2.8 TB/s): per cache row and lane 8 conversions + 8 FMAs per head + a 16-lane sum per
// head for the score alone.  Here a wave takes 16 cache rows at a time as the A operand of v_mfma_f32_16x16x32_f16 (row = lane % 16,
// dims 32 s + 8 (lane / 16) .. + 7 of k step s: four 16-byte requests per lane, 64 contiguous bytes of a row per request) and the
// rotated query heads as the B operand (head = lane % 16, zero beyond GROUP; the same dims): four MFMAs give the 16 x GROUP scores with no
// conversion, no FMA and no cross-lane sum.  Lane (head, row quarter) then holds the scores of rows 4 (lane / 16) + i of the chunk: the
// running max / sum of a head live in ONE lane per row quarter and the softmax arithmetic of all heads runs in parallel across the
// lanes.  The value sum stays on the vector ALUs (V rows in memory have the dims contiguous, the matrix core would want the positions):
// lane (dims 8 (lane % 16) .., row quarter) requests rows 4 (lane / 16) + i, so that the probabilities it needs sit in its own
// 16-lane row -- one DPP row broadcast (row_newbcast) per (row, head) delivers them.  Row *pos -- this token -- is part of the sweep:
// the lanes that would request it take the rotated k and the new v from registers instead (rows behind it are masked).  Software-
// pipelined like the kernels above (two register sets of UC 16-row chunks per wave).  Rounding: scores are sums of exact fp16 x fp16
// products in fp32 (as before, in another order), scale * log2 e is applied to the fp32 score.
__device__ __forceinline__ float row_bcast(float x, int g) {   // the value of lane g of each 16-lane row, to all 16 (g: a constant after unrolling)
#define QA_BC(G) case G: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + G, 0xf, 0xf, false));
  switch (g) {
    QA_BC(0) QA_BC(1) QA_BC(2) QA_BC(3) QA_BC(4) QA_BC(5) QA_BC(6) QA_BC(7)
    default: return x;
  }
#undef QA_BC
}

template <int GROUP, int UC>
__global__ __launch_bounds__(256, 2) void decode_rope_attention_gqa_mfma_kernel(
    const half_t* __restrict__ qkv, const half_t* __restrict__ cos_t, const half_t* __restrict__ sin_t,
    const long* __restrict__ pos, half_t* __restrict__ k_cache, half_t* __restrict__ v_cache, half_t* __restrict__ out,
    int nh, int nkv, int L, float scale, int gsplit) {
  constexpr int D = 128;
  static_assert(GROUP >= 1 && GROUP <= 8, "query heads per KV head");
  __shared__ float part[4][GROUP][D + 2];  // per wave and query head: 128 output dims, running max, running sum
  // gsplit workgroups share a KV head, GROUP query heads each (8 heads per KV head with few sequences: two workgroups of 4)
  const int b = blockIdx.y, kvh = blockIdx.x / gsplit, hbase = kvh * (GROUP * gsplit) + (blockIdx.x % gsplit) * GROUP;
  const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6));
  const int sub = lane & 15, rsel = lane >> 4;
  const half_t* row = qkv + (size_t)b * (nh + 2 * nkv) * D;
  half_t* krow = k_cache + ((size_t)b * nkv + kvh) * L * D;
  half_t* vrow = v_cache + ((size_t)b * nkv + kvh) * L * D;
  const half_t* kp = krow + rsel * 8;   // + row * D + 32 s
  const half_t* vp = vrow + sub * 8;    // + row * D
  half8_t qraw[4], kraw[4];
  const int hq = min(sub, GROUP - 1);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qraw[s] = *(const half8_t*)(row + (size_t)(hbase + hq) * D + 32 * s + 8 * rsel);
    kraw[s] = *(const half8_t*)(row + (size_t)(nh + kvh) * D + 32 * s + 8 * rsel);
  }
  const half8_t vn = *(const half8_t*)(row + (size_t)(nh + nkv + kvh) * D + sub * 8);

  auto request = [&](half8_t (&ka)[UC][4], half8_t (&va)[UC][4], int tb, int last) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < UC; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) ka[c][s] = *(const half8_t*)(kp + (size_t)min(tb + 64 * c + sub, last) * D + 32 * s);
#pragma unroll
    for (int c = 0; c < UC; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) va[c][i] = *(const half8_t*)(vp + (size_t)min(tb + 64 * c + 4 * rsel + i, last) * D);
  };
  half8_t kva[UC][4], vva[UC][4], kvb[UC][4], vvb[UC][4];
  constexpr int STEP = 64 * UC;
  int tb = wave * 16;   // first row of this wave's first chunk; a trip of the workgroup covers 64 UC rows
  const int p = (int)pos[0];  // cache rows 0..p-1 come from memory, row p (this token) from registers
  const int plast = max(p - 1, 0);
  // (requesting the first two sets before *pos is known -- rows clamped to the cache's end, masked afterwards -- measured no faster:
  // 16.4 against 15.2 us at bs = 64, 8 KV heads, 192 positions on two boxes, the same ratio to the vector-ALU sweep within 4 %)
  request(kva, vva, tb, plast);

  // rotate the q heads and the new k (dims 32 s + 8 rsel + j pair with the same j of k step s ^ 2) while the first rows are in flight
  half8_t qb[4], kh[4];
  {
    half8_t cs[4], sn[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      cs[s] = *(const half8_t*)(cos_t + (size_t)p * D + 32 * s + 8 * rsel);
      sn[s] = *(const half8_t*)(sin_t + (size_t)p * D + 32 * s + 8 * rsel);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float sign = s < 2 ? -1.f : 1.f;  // rotate_half: dims 0..63 pair with -x[i+64], dims 64..127 with +x[i-64]
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c = (float)cs[s][j], sj = (float)sn[s][j];
        qb[s][j] = sub < GROUP ? (half_t)((float)(half_t)((float)qraw[s][j] * c) + (float)(half_t)(sign * (float)qraw[s ^ 2][j] * sj)) : (half_t)0.f;
        kh[s][j] = (half_t)((float)(half_t)((float)kraw[s][j] * c) + (float)(half_t)(sign * (float)kraw[s ^ 2][j] * sj));
      }
    }
    if (wave == 0 && sub == 0 && blockIdx.x % gsplit == 0) {  // append to the caches (position p is not requested by anyone in this launch)
#pragma unroll
      for (int s = 0; s < 4; ++s) *(half8_t*)(krow + (size_t)p * D + 32 * s + 8 * rsel) = kh[s];
    }
    if (wave == 0 && rsel == 0 && blockIdx.x % gsplit == 0) *(half8_t*)(vrow + (size_t)p * D + sub * 8) = vn;
  }

  const float qk_scale = scale * 1.44269504088896f;   // scores in the log2 domain
  float m_ = -INFINITY, l_ = 0.f;   // lane (head sub, row quarter rsel): running max / sum of that head over this quarter's rows
  float acc[GROUP][8];
#pragma unroll
  for (int g = 0; g < GROUP; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;

  auto consume = [&](half8_t (&ka)[UC][4], half8_t (&va)[UC][4], int tb) __attribute__((always_inline)) {
    float sc[UC][4], mb = m_;
#pragma unroll
    for (int c = 0; c < UC; ++c) {
      const int base = tb + 64 * c;
      if (base <= p && p < base + 16) {  // (wave-uniform) this chunk holds row p: from registers
        if (base + sub == p) {
#pragma unroll
          for (int s = 0; s < 4; ++s) ka[c][s] = kh[s];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (base + 4 * rsel + i == p) va[c][i] = vn;
      }
      floatx4 d4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) d4 = mfma16(ka[c][s], qb[s], d4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sc[c][i] = base + 4 * rsel + i <= p ? d4[i] * qk_scale : -INFINITY;
        mb = fmaxf(mb, sc[c][i]);
      }
    }
    const float mref = mb == -INFINITY ? 0.f : mb;
    const float corr = exp2f(m_ - mref);
    l_ *= corr;
#pragma unroll
    for (int c = 0; c < UC; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sc[c][i] = exp2f(sc[c][i] - mref);   // (0 for a masked row)
        l_ += sc[c][i];
      }
    m_ = mb;
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
      const float cg = row_bcast(corr, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[g][j] *= cg;
    }
#pragma unroll
    for (int c = 0; c < UC; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float pg[GROUP];
#pragma unroll
        for (int g = 0; g < GROUP; ++g) pg[g] = row_bcast(sc[c][i], g);   // (DPP: every lane takes part, also for masked rows)
        if (tb + 64 * c + 4 * rsel + i <= p) {  // (a row behind the sequence holds anything, NaN included: 0 * NaN would poison the sum)
          float vf[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) vf[j] = (float)va[c][i][j];
#pragma unroll
          for (int g = 0; g < GROUP; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[g][j] += pg[g] * vf[j];
        }
      }
  };
  if (tb <= p) {
    while (true) {  // wave-uniform trip counts; the requests for the next set leave before this one is consumed
      const bool more_b = tb + STEP <= p;
      if (more_b) request(kvb, vvb, tb + STEP, plast);
      consume(kva, vva, tb);
      if (!more_b) break;
      const bool more_a = tb + 2 * STEP <= p;
      if (more_a) request(kva, vva, tb + 2 * STEP, plast);
      consume(kvb, vvb, tb + STEP);
      if (!more_a) break;
      tb += 2 * STEP;
    }
  }
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
    const float mg = row_bcast(m_, g), lg = row_bcast(l_, g);   // this row quarter's state of head g, in all of its 16 lanes
    // merge the 4 row quarters of the wave, then the 4 waves through LDS
    float mw = fmaxf(mg, __shfl_xor(mg, 16));
    mw = fmaxf(mw, __shfl_xor(mw, 32));
    const float sc_ = exp2f(mg - (mw == -INFINITY ? 0.f : mw));
    float lw = lg * sc_;
    lw += __shfl_xor(lw, 16);
    lw += __shfl_xor(lw, 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = acc[g][j] * sc_;
      a += __shfl_xor(a, 16);
      a += __shfl_xor(a, 32);
      if (rsel == 0) part[wave][g][sub * 8 + j] = a;
    }
    if (lane == 0) {
      part[wave][g][D] = mw;
      part[wave][g][D + 1] = lw;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GROUP * D; i += 256) {
    const int g = i / D, dd = i % D;
    const float M = fmaxf(fmaxf(part[0][g][D], part[1][g][D]), fmaxf(part[2][g][D], part[3][g][D]));  // finite: row p is in somebody's sweep
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = exp2f(part[w][g][D] - M);
      num += part[w][g][dd] * f;
      den += part[w][g][D + 1] * f;
    }
    out[((size_t)b * nh + hbase + g) * D + dd] = (half_t)(num / den);
  }
}

// y[m, 8t + i] = silu(gu[m, 16t + i]) * gu[m, 16t + 8 + i], i < 8: gate and up channels interleaved in blocks of 8, the
// order the fused gate_up GEMM produces (and consumes directly when its silu_mul epilogue is on); I % 8 == 0
__global__ __launch_bounds__(256) void silu_mul_kernel(const half_t* __restrict__ gu, half_t* __restrict__ y, int I, size_t n8) {
  const size_t i8 = (size_t)blockIdx.x * 256 + threadIdx.x;  // one block of 8 outputs
  if (i8 >= n8) return;
  const size_t m = (i8 * 8) / I, t = ((i8 * 8) % I) / 8;
  const half8_t g = *(const half8_t*)(gu + m * 2 * I + 16 * t), u = *(const half8_t*)(gu + m * 2 * I + 16 * t + 8);
  half8_t o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = silu_mul_f16(g[j], u[j]);
  *(half8_t*)(y + i8 * 8) = o;
}

// Touch one dword of every 128-byte line of [p, p + lines * 128): the memory side (Infinity Cache, 256 MiB) keeps what HBM
// delivered, so a later launch streams these bytes from the cache.  The loads are never used; a wave ends when they have landed.
// ------------------------------------------------------------------------------------------------
// lm_head of a decode step: logits[b, v] = h[b, :] . W[v, :] (fp16 weights [V, H], fp32 sums, fp16 logits), h = RMSNorm(x) * nw when
// nw is given, else x; with the greedy arg-max folded in.  The reference leaves this layer in fp16 (AWQ does not quantise lm_head)
// and runs torch's linear + max (examples/benchmark.py:54-57); at one token per sequence that is a 262 MB stream (Llama-2-7B) which
// hipBLASLt moves at 3.1 TB/s -- 85 us of a 1.40 ms step with the arg-max [r01 trace].  Here: every wave streams blocks of four
// vocabulary rows, 16 bytes per lane and 512-element chunk, one chunk ahead; h sits in LDS (normalised there by every workgroup
// for itself); fdot2 into fp32; the four sums of a block reduced across the wave by DPP; the arg-max travels as one 64-bit key
// (order-preserving bits of the fp16 logit, then the complement of the index: the maximum is the largest logit at the lowest
// index) per wave -> per workgroup -> a second one-workgroup launch.  B <= 4 sequences (B * 16 accumulators).
__device__ __forceinline__ unsigned long long argmax_key(half_t logit, unsigned idx) {
  const unsigned bits = __builtin_bit_cast(unsigned, (float)logit);
  const unsigned k = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
  return ((unsigned long long)k << 32) | (0xffffffffu - idx);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {  // sum of the 64 lanes, delivered as a wave-uniform value
  v = lanes_sum<16>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
}

template <int B>
__global__ __launch_bounds__(512) void lm_head_kernel(const half_t* __restrict__ x, const half_t* __restrict__ nw, float eps,
                                                      const half_t* __restrict__ W, int V, int H, half_t* __restrict__ hidden_out,
                                                      half_t* __restrict__ logits, unsigned long long* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* xs = (half_t*)smem;                                             // [B][H]
  __shared__ float part[8];
  __shared__ unsigned long long wbest[8][B];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // h = RMSNorm(x) * nw with quick_rmsnorm_f16's rounding points, or x itself
  for (int b = 0; b < B; ++b) {
    const half_t* xr = x + (size_t)b * H;
    float inv = 1.f;
    if (nw) {
      // (the sum of squares in rmsnorm_kernel's own order -- 256 threads, four wave sums -- so that h is that kernel's, bit for bit)
      float ss = 0.f;
      if (threadIdx.x < 256) {
        for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
          const half8_t v = *(const half8_t*)(xr + i);
#pragma unroll
          for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
        }
      }
      ss = wave_sum(ss);
      __syncthreads();  // (part of the previous row has been read)
      if (lane == 0) part[wave] = ss;
      __syncthreads();
      inv = rsqrtf((part[0] + part[1] + part[2] + part[3]) / H + eps);
    }
    for (int i = threadIdx.x * 8; i < H; i += 512 * 8) {
      half8_t v = *(const half8_t*)(xr + i);
      if (nw) {
        const half8_t g = *(const half8_t*)(nw + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (half_t)((half_t)((float)v[j] * inv) * g[j]);
      }
      *(half8_t*)(xs + (size_t)b * H + i) = v;
      if (hidden_out && blockIdx.x == 0) *(half8_t*)(hidden_out + (size_t)b * H + i) = v;
    }
  }
  __syncthreads();

  const int NC = H / 512, nblk = (V + 3) / 4, nwaves = gridDim.x * 8;
  unsigned long long best[B];
#pragma unroll
  for (int b = 0; b < B; ++b) best[b] = 0ull;
  int blk = blockIdx.x * 8 + wave;
  if (blk < nblk) {
    const half_t* wl = W + lane * 8;
    // weight chunks travel one ahead of the one being used, across the wave's blocks (two ahead measured no faster: more registers);
    // behind the last block the sequence replays that block (loads nobody uses)
    u32x4 cur[4], nxt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cur[r] = *(const u32x4*)(wl + (size_t)min(blk * 4 + r, V - 1) * H);
    while (true) {
      float acc[4][B];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
      const int nblk_next = blk + nwaves < nblk ? blk + nwaves : blk;
      for (int c = 0; c < NC; ++c) {
        const bool last = c + 1 == NC;
        const int lb = last ? nblk_next : blk, lc = last ? 0 : c + 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) nxt[r] = *(const u32x4*)(wl + (size_t)min(lb * 4 + r, V - 1) * H + lc * 512);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const u32x4 xv = *(const u32x4*)(xs + (size_t)b * H + c * 512 + lane * 8);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[r][b] = __builtin_amdgcn_fdot2(as_h2(cur[r][i]), as_h2(xv[i]), acc[r][b], false);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) cur[r] = nxt[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = blk * 4 + r;
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const half_t lg = (half_t)wave_sum_dpp(acc[r][b]);
          if (row < V) {  // wave-uniform
            if (logits && lane == 0) logits[(size_t)b * V + row] = lg;
            const unsigned long long k = argmax_key(lg, (unsigned)row);
            best[b] = k > best[b] ? k : best[b];
          }
        }
      }
      if (blk + nwaves >= nblk) break;
      blk += nwaves;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int b = 0; b < B; ++b) wbest[wave][b] = best[b];
  }
  __syncthreads();
  if (threadIdx.x < B) {
    unsigned long long m = 0ull;
#pragma unroll
    for (int w = 0; w < 8; ++w) m = wbest[w][threadIdx.x] > m ? wbest[w][threadIdx.x] : m;
    partial[(size_t)blockIdx.x * B + threadIdx.x] = m;
  }
}

// the arg-max over the workgroups' keys: one workgroup, thread t takes partial[t], partial[t + 256], ...
__global__ __launch_bounds__(256) void lm_head_argmax_kernel(const unsigned long long* __restrict__ partial, int nwg, int B, long* __restrict__ out) {
  __shared__ unsigned long long red[4];
  for (int b = 0; b < B; ++b) {
    unsigned long long m = 0ull;
    for (int i = threadIdx.x; i < nwg; i += 256) m = partial[(size_t)i * B + b] > m ? partial[(size_t)i * B + b] : m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long t = __shfl_xor(m, o);
      m = t > m ? t : m;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int w = 1; w < 4; ++w) m = red[w] > m ? red[w] : m;
      out[b] = (long)(0xffffffffu - (unsigned)(m & 0xffffffffu));
    }
  }
}

template <int DW>  // dwords touched per 128-byte line: 1, 2 (one per 64 bytes), 4 (one per 32 bytes); 32 = every byte (16-byte loads)
__global__ void __launch_bounds__(256) prefetch_kernel(const unsigned* __restrict__ p, size_t lines, unsigned* sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  if constexpr (DW == 32) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < lines * 8; i += stride) {
      const u32x4 v = ((const u32x4*)p)[i];
      acc |= v[0] | v[1] | v[2] | v[3];
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < lines * DW; i += stride) acc |= p[i * (32 / DW)];
  }
  if (acc == 0x9e3779b9u && sink) *sink = acc;  // (keeps the loads alive; sink is null)
}

}  // namespace quick_amd

using namespace quick_amd;

extern "C" {

int quick_rmsnorm_f16(const void* x, const void* weight, void* y, int rows, int hidden, float eps, void* hip_stream) {
  if (rows <= 0 || hidden <= 0 || hidden % 8 != 0) return QUICK_ERR_INVALID_ARGUMENT;
#define QA_NORM(NIT)                                                                                                  \
  hipLaunchKernelGGL(rmsnorm_kernel<NIT>, dim3(rows), dim3(256), 0, (hipStream_t)hip_stream, (const half_t*)x,        \
                     (const half_t*)weight, (half_t*)y, hidden, eps)
  if (hidden <= 2048) QA_NORM(1);
  else if (hidden <= 4096) QA_NORM(2);
  else if (hidden <= 8192) QA_NORM(4);
  else QA_NORM(0);
#undef QA_NORM
  return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
}

int quick_rope_kv_append_f16(const void* qkv, const void* cos_table, const void* sin_table, const void* pos, void* q_out,
                             void* k_cache, void* v_cache, int batch, int n_heads, int n_kv_heads, int head_dim,
                             int cache_len, void* hip_stream) {
  if (batch <= 0 || head_dim % 2 != 0 || head_dim > 128 || n_heads % n_kv_heads != 0) return QUICK_ERR_INVALID_ARGUMENT;
  if (batch > 65535) return QUICK_ERR_UNSUPPORTED;  // grid.y
  hipLaunchKernelGGL(rope_kv_kernel, dim3(n_heads + 2 * n_kv_heads, batch), dim3(64), 0, (hipStream_t)hip_stream,
                     (const half_t*)qkv, (const half_t*)cos_table, (const half_t*)sin_table, (const long*)pos,
                     (half_t*)q_out, (half_t*)k_cache, (half_t*)v_cache, n_heads, n_kv_heads, head_dim, cache_len);
  return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
}

int quick_rope_kv_write_f16(const void* qkv, const void* cos_table, const void* sin_table, const void* pos0, void* q_out,
                            void* k_cache, void* v_cache, int batch, int tokens, int n_heads, int n_kv_heads, int head_dim,
                            int cache_len, void* hip_stream) {
  if (batch <= 0 || tokens <= 0 || head_dim % 2 != 0 || head_dim > 128 || n_heads % n_kv_heads != 0) return QUICK_ERR_INVALID_ARGUMENT;
  if ((long)batch * tokens > 65535) return QUICK_ERR_UNSUPPORTED;  // grid.y
  hipLaunchKernelGGL(rope_kv_prefill_kernel, dim3(n_heads + 2 * n_kv_heads, batch * tokens), dim3(64), 0, (hipStream_t)hip_stream,
                     (const half_t*)qkv, (const half_t*)cos_table, (const half_t*)sin_table, (const long*)pos0, (half_t*)q_out,
                     (half_t*)k_cache, (half_t*)v_cache, tokens, n_heads, n_kv_heads, head_dim, cache_len);
  return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
}

int quick_decode_attention_f16(const void* q, const void* k_cache, const void* v_cache, const void* pos, void* out,
                               int batch, int n_heads, int n_kv_heads, int head_dim, int cache_len, float scale,
                               void* hip_stream) {
  if (batch <= 0 || batch > 65535 || head_dim != 128 || n_heads % n_kv_heads != 0) return QUICK_ERR_UNSUPPORTED;
  const size_t lds = (((size_t)cache_len + 3) & ~(size_t)3) * 4 + 4 * 128 * 4;
  if (lds > 64 * 1024) return QUICK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(decode_attention_kernel, dim3(n_heads, batch), dim3(256), (unsigned)lds, (hipStream_t)hip_stream,
                     (const half_t*)q, (const half_t*)k_cache, (const half_t*)v_cache, (const long*)pos, (half_t*)out,
                     n_heads, n_kv_heads, cache_len, scale);
  return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
}

int quick_decode_rope_attention_f16(const void* qkv, const void* cos_table, const void* sin_table, const void* pos,
                                    void* k_cache, void* v_cache, void* out, int batch, int n_heads, int n_kv_heads,
                                    int head_dim, int cache_len, float scale, void* hip_stream) {
  if (batch <= 0 || head_dim != 128 || n_heads % n_kv_heads != 0) return QUICK_ERR_UNSUPPORTED;
  if (batch > 65535 || cache_len <= 0) return QUICK_ERR_UNSUPPORTED;  // grid.y; the caller keeps *pos < cache_len (device value)
#define QA_GQA(GROUP, UNR, GSPLIT)                                                                                 \
  hipLaunchKernelGGL((decode_rope_attention_gqa_kernel<GROUP, UNR>), dim3(n_kv_heads * (GSPLIT), batch), dim3(256), 0, \
                     (hipStream_t)hip_stream, (const half_t*)qkv, (const half_t*)cos_table, (const half_t*)sin_table, \
                     (const long*)pos, (half_t*)k_cache, (half_t*)v_cache, (half_t*)out, n_heads, n_kv_heads,       \
                     cache_len, scale, GSPLIT)
  // grouped-query models with enough (sequence, KV head) pairs to fill the chip: one sweep over the cache per KV head
  // (measured [r01]: 32 query / 8 KV heads, bs=64, 190 positions: 21.5 us against 32.6 us with a workgroup per query head;
  // with fewer than ~256 workgroups the per-head kernels are ahead).  8 heads per KV head run as two workgroups of 4.
  const int group = n_heads / n_kv_heads;
  if (group == 2 || group == 4 || group == 8) {
    const int pairs = batch * n_kv_heads;
    bool done = true;
    // (four rows per slot and register set: 244 registers with four heads, two waves per SIMD; eight rows spill)
#define QA_GQA_U(GROUP, GSPLIT) QA_GQA(GROUP, 4, GSPLIT)
    static const int mfma_on = [] {   // (A/B: QUICK_AMD_ATTN_MFMA=0 runs the vector-ALU sweeps)
      const char* e = getenv("QUICK_AMD_ATTN_MFMA");
      return e ? atoi(e) : 1;
    }();
#define QA_GQA_M(GROUP, UC, GSPLIT)                                                                                  \
  hipLaunchKernelGGL((decode_rope_attention_gqa_mfma_kernel<GROUP, UC>), dim3(n_kv_heads * (GSPLIT), batch), dim3(256), 0, \
                     (hipStream_t)hip_stream, (const half_t*)qkv, (const half_t*)cos_table, (const half_t*)sin_table, \
                     (const long*)pos, (half_t*)k_cache, (half_t*)v_cache, (half_t*)out, n_heads, n_kv_heads, cache_len, scale, GSPLIT)
    // (8 heads per KV head from 128 pairs -- Llama-2-70B at bs = 16, one workgroup per pair on half the CUs -- measured 1 % BEHIND the two
    // vector-ALU sweeps of 4 heads, profiles/r05_decode70_ab.txt)
    if (mfma_on && group == 8 && pairs >= 256) QA_GQA_M(8, 1, 1);
    else if (mfma_on && group == 8 && pairs * 2 >= 256) QA_GQA_M(4, 1, 2);   // (fewer sequences: two workgroups of 4 heads per KV head -- Llama-2-70B at
                                                                             // bs = 16: 12.6 -> 11.4 us against the vector-ALU sweeps, 1617 -> 1658 tok/s)
    else if (mfma_on && group == 4 && pairs >= 256) QA_GQA_M(4, 1, 1);   // (two chunks per set: 256 registers and a spill, 5-10 % behind)
    else if (group == 8 && pairs * 2 >= 256) QA_GQA_U(4, 2);
    else if (group == 4 && pairs >= 256) QA_GQA_U(4, 1);
    else if (group == 4 && pairs * 2 >= 256) QA_GQA_U(2, 2);  // fewer sequences: two workgroups of 2 heads per KV head
    else if (group == 2 && pairs >= 256) QA_GQA_U(2, 1);
    else done = false;
#undef QA_GQA_U
#undef QA_GQA_M
    if (done) return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
  }
#undef QA_GQA
  // few workgroups and a cache longer than one four-wave trip: eight waves, 256 positions per trip; else four waves, 128 per trip
  // [r05, profiles/r05_attention.txt: bs = 1, 192 / 512 positions 5.39 / 8.6 -> 5.15 / 7.5 us, 128 positions 4.24 against 5.0; from 256
  // workgroups on the four-wave form is ahead everywhere; sixteen waves x 4 rows and four waves x 16 rows measured behind both]
#define QA_FLASH(W, U)                                                                                                       \
  hipLaunchKernelGGL((decode_rope_attention_flash_kernel<W, U>), dim3(n_heads, batch), dim3((W) * 64), 0, (hipStream_t)hip_stream, \
                     (const half_t*)qkv, (const half_t*)cos_table, (const half_t*)sin_table, (const long*)pos,               \
                     (half_t*)k_cache, (half_t*)v_cache, (half_t*)out, n_heads, n_kv_heads, cache_len, scale)
  static const int forced = [] {   // (A/B runs: 4 or 8 waves)
    const char* e = getenv("QUICK_AMD_ATTN_WAVES");
    return e ? atoi(e) : 0;
  }();
  const int pick = forced ? forced : (((long)n_heads * batch <= 128 && cache_len > 128) ? 8 : 4);
  if (pick == 8) QA_FLASH(8, 4);   // (two register sets of 4 rows per slot: the same 256 / 128 positions in flight as one set of 8)
  else QA_FLASH(4, 4);
#undef QA_FLASH
  return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
}

int quick_silu_mul_f16(const void* gate_up, void* y, int rows, int intermediate, void* hip_stream) {
  if (rows <= 0 || intermediate <= 0 || intermediate % 8 != 0) return QUICK_ERR_INVALID_ARGUMENT;
  const size_t n8 = (size_t)rows * intermediate / 8;
  hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                     (const half_t*)gate_up, (half_t*)y, intermediate, n8);
  return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
}

size_t quick_lm_head_workspace_bytes(int batch) { return (size_t)512 * (batch > 0 ? batch : 1) * 8; }

int quick_lm_head_argmax_f16(const void* x, const void* norm_weight, float eps, const void* weight, void* hidden_out, void* logits,
                             void* next_token, void* workspace, size_t workspace_bytes, int batch, int vocab, int hidden,
                             void* hip_stream) {
  if (!x || !weight || !next_token || !workspace) return QUICK_ERR_INVALID_ARGUMENT;
  if (batch < 1 || batch > 4 || vocab < 1 || hidden < 512 || hidden % 512 != 0) return QUICK_ERR_UNSUPPORTED;
  if (workspace_bytes < quick_lm_head_workspace_bytes(batch)) return QUICK_ERR_WORKSPACE;
  const size_t lds = (size_t)batch * hidden * 2;
  if (lds > 72 * 1024) return QUICK_ERR_UNSUPPORTED;  // two workgroups per CU
  const int nblk = (vocab + 3) / 4;
  const unsigned grid = (unsigned)std::min(512, (nblk + 7) / 8);
  hipStream_t st = (hipStream_t)hip_stream;
#define QA_LM(BV)                                                                                                     \
  do {                                                                                                                \
    auto kfn = lm_head_kernel<BV>;                                                                                    \
    static std::atomic<unsigned long long> attr_set{0};                                                                \
    (void)lds_limit_once(attr_set, (const void*)kfn, 72 * 1024);                                                   \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), (unsigned)lds, st, (const half_t*)x, (const half_t*)norm_weight, eps, \
                       (const half_t*)weight, vocab, hidden, (half_t*)hidden_out, (half_t*)logits,                    \
                       (unsigned long long*)workspace);                                                               \
  } while (0)
  switch (batch) {
    case 1: QA_LM(1); break;
    case 2: QA_LM(2); break;
    case 3: QA_LM(3); break;
    default: QA_LM(4); break;
  }
#undef QA_LM
  hipLaunchKernelGGL(lm_head_argmax_kernel, dim3(1), dim3(256), 0, st, (const unsigned long long*)workspace, (int)grid, batch,
                     (long*)next_token);
  return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
}

#ifdef QUICK_AMD_TOOLS   // (a measurement aid of tools/prefetch_probe.py: not in the product library)
int quick_prefetch(const void* ptr, size_t bytes, int workgroups, void* hip_stream) {
  if (!ptr || bytes < 128) return QUICK_OK;
  const int dw = workgroups >> 16;  // (probe builds: bits 16.. choose the touch density; 0 = one dword per line)
  workgroups &= 0xffff;
  if (workgroups <= 0) workgroups = 64;
  unsigned* sink = nullptr;  // (the kernel's store exists only so that the compiler keeps the loads)
  const size_t lines = bytes / 128;
  const unsigned g = (unsigned)std::min<size_t>((size_t)workgroups, (lines + 255) / 256);
  auto k = dw == 32 ? prefetch_kernel<32> : (dw == 4 ? prefetch_kernel<4> : (dw == 2 ? prefetch_kernel<2> : prefetch_kernel<1>));
  hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, (hipStream_t)hip_stream, (const unsigned*)ptr, lines, sink);
  return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
}
#endif

}  // extern "C"
