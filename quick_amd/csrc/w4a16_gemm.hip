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
//   * activations: straight L2 -> VGPR fragment loads when M is small (skinny kernel), staged
//     through LDS in fragment order by global_load_lds when M is large (tiled kernel).
#include "w4a16_common.hpp"
#include "../../include/quick_amd.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

namespace quick_amd {

// ------------------------------------------------------------------------------------------------
// group constants
// ------------------------------------------------------------------------------------------------
// scales[g, n] (fp16, row pitch 2N) and the zero-point nibble of (g, n) (row pitch N/4 dwords).
__device__ __forceinline__ GroupQ load_group(const half_t* __restrict__ S, const uint32_t* __restrict__ QZ,
                                             int g, int n, int N) {
  const half_t s = S[(size_t)g * (2 * N) + n];
  const uint32_t zq = QZ[(size_t)g * (N >> 2) + (n >> 3)];
  return make_group(s, (zq >> (4 * (n & 7))) & 15u);
}

// ------------------------------------------------------------------------------------------------
// skinny kernel: one 16-channel tile per workgroup, K split over the waves of the workgroup
// (and optionally over blockIdx.z), up to MT token tiles of 16 kept in registers.
// ------------------------------------------------------------------------------------------------
template <int MT, int WAVES, bool G128, int U>
__device__ __forceinline__ void skinny_body(int kt, const u32x4* __restrict__ wp, const half_t* const (&xp)[MT],
                                            const half_t* __restrict__ S, const uint32_t* __restrict__ QZ,
                                            int n, int N, int G, floatx4 (&acc)[MT]) {
  u32x4 w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) w[u] = wp[(size_t)(kt + u) * 64];
  GroupQ grp[U][G128 ? 1 : 4];
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int t = 0; t < (G128 ? 1 : 4); ++t) grp[u][t] = load_group(S, QZ, ((kt + u) * 128 + 32 * t) / G, n, N);
  }
  half8_t xf[U][4][MT];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[u][t][mt] = *(const half8_t*)(xp[mt] + (kt + u) * 128 + 32 * t);
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const half8_t a = dequant8(w[u][t], grp[u][G128 ? 0 : t]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(a, xf[u][t][mt], acc[mt]);
    }
}

template <int MT, int WAVES, bool G128>
__global__ __launch_bounds__(WAVES * 64) void w4a16_skinny_kernel(
    const half_t* __restrict__ X, const u32x4* __restrict__ QW, const half_t* __restrict__ S,
    const uint32_t* __restrict__ QZ, const half_t* __restrict__ bias, half_t* __restrict__ Y,
    float* __restrict__ Yacc, int M, int K, int N, int G, int ksplit) {
  static_assert(WAVES >= MT, "reduction assigns one token tile per wave");
  __shared__ floatx4 red[WAVES][MT][64];

  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int nt = blockIdx.x, mb = blockIdx.y, ks = blockIdx.z;
  const int KT = K >> 7;
  const int wg_begin = (int)((long)KT * ks / ksplit), wg_end = (int)((long)KT * (ks + 1) / ksplit);
  const int cnt = wg_end - wg_begin;
  const int kt_begin = wg_begin + cnt * wave / WAVES, kt_end = wg_begin + cnt * (wave + 1) / WAVES;
  const int n = nt * 16 + n16;

  const u32x4* wp = QW + (size_t)nt * KT * 64 + lane;
  const half_t* xp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = min((mb * MT + mt) * 16 + n16, M - 1);  // rows >= M replay row M-1; never stored
    xp[mt] = X + (size_t)row * K + q * 8;
  }

  floatx4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = floatx4{0.f, 0.f, 0.f, 0.f};

  constexpr int U = (MT == 1) ? 4 : (MT == 2 ? 2 : 1);
  int kt = kt_begin;
  for (; kt + U <= kt_end; kt += U) skinny_body<MT, WAVES, G128, U>(kt, wp, xp, S, QZ, n, N, G, acc);
  for (; kt < kt_end; ++kt) skinny_body<MT, WAVES, G128, 1>(kt, wp, xp, S, QZ, n, N, G, acc);

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) red[wave][mt][lane] = acc[mt];
  __syncthreads();
  if (wave < MT) {
    const int mt = wave;
    floatx4 sum = red[0][mt][lane];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) sum += red[w][mt][lane];
    const int m = (mb * MT + mt) * 16 + n16;
    const int nc = nt * 16 + 4 * q;  // lane holds channels nc..nc+3 of token m
    if (m < M) {
      if (ksplit == 1) {
        if (bias) {
          const half4_t b = *(const half4_t*)(bias + nc);
#pragma unroll
          for (int r = 0; r < 4; ++r) sum[r] += (float)b[r];
        }
        half4_t o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)sum[r];
        *(half4_t*)(Y + (size_t)m * N + nc) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(Yacc + (size_t)m * N + nc + r, sum[r]);
      }
    }
  }
}

// fp32 accumulator (grid split-K) -> fp16 output (+ bias)
__global__ __launch_bounds__(256) void w4a16_finalize_kernel(const float* __restrict__ Yacc,
                                                             const half_t* __restrict__ bias,
                                                             half_t* __restrict__ Y, int M, int N) {
  const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= (size_t)M * N) return;
  floatx4 v = *(const floatx4*)(Yacc + i4);
  if (bias) {
    const half4_t b = *(const half4_t*)(bias + (i4 % N));
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += (float)b[r];
  }
  half4_t o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
  *(half4_t*)(Y + i4) = o;
}

// ------------------------------------------------------------------------------------------------
// tiled kernel: workgroup tile (BMT*16 tokens) x (4*TN*16 channels), 4 waves side by side along N.
// The token tile of one 128-k step is staged in LDS *in B-fragment order* by global_load_lds
// (lane l of fragment (t, mt) sources x[mt*16 + l%16][32t + 8*(l/16) ..+7]), so every ds_read_b128
// is lane-linear and conflict free; weights never touch LDS.
// ------------------------------------------------------------------------------------------------
template <int BMT, int TN, bool G128>
__global__ __launch_bounds__(256) void w4a16_tiled_kernel(
    const half_t* __restrict__ X, const u32x4* __restrict__ QW, const half_t* __restrict__ S,
    const uint32_t* __restrict__ QZ, const half_t* __restrict__ bias, half_t* __restrict__ Y,
    float* __restrict__ Yacc, int M, int K, int N, int G, int ksplit) {
  constexpr int FRAGS = 4 * BMT;  // 1 KiB fragments per 128-k step
  __shared__ __attribute__((aligned(16))) char smem[2][FRAGS * 1024];

  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int NB = N / (64 * TN);
  const int nb = blockIdx.x % NB, mb = blockIdx.x / NB, ks = blockIdx.y;
  const int KT = K >> 7;
  const int kt_begin = (int)((long)KT * ks / ksplit), kt_end = (int)((long)KT * (ks + 1) / ksplit);
  const int m0 = mb * BMT * 16;
  const int nt0 = (nb * 4 + wave) * TN;  // first 16-channel tile of this wave

  const u32x4* wp[TN];
  int ncol[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    wp[j] = QW + (size_t)(nt0 + j) * KT * 64 + lane;
    ncol[j] = (nt0 + j) * 16 + n16;
  }
  // fragments this wave stages: f = wave, wave + 4, ... ; f = t * BMT + mt
  const half_t* xsrc[BMT];
#pragma unroll
  for (int i = 0; i < BMT; ++i) {
    const int f = wave + 4 * i, t = f / BMT, mt = f % BMT;
    const int row = min(m0 + mt * 16 + n16, M - 1);
    xsrc[i] = X + (size_t)row * K + 32 * t + 8 * q;
  }

  floatx4 acc[TN][BMT];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) acc[j][mt] = floatx4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < BMT; ++i)
      __builtin_amdgcn_global_load_lds(QA_GLOBAL_PTR(xsrc[i] + kt * 128), QA_LDS_PTR(smem[buf] + (wave + 4 * i) * 1024),
                                       16, 0, 0);
  };

  u32x4 wcur[TN], wnext[TN];
  GroupQ gcur[TN][G128 ? 1 : 4], gnext[TN][G128 ? 1 : 4];
  auto load_w = [&](int kt, u32x4 (&w)[TN], GroupQ (&g)[TN][G128 ? 1 : 4]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      w[j] = wp[j][(size_t)kt * 64];
#pragma unroll
      for (int t = 0; t < (G128 ? 1 : 4); ++t) g[j][t] = load_group(S, QZ, (kt * 128 + 32 * t) / G, ncol[j], N);
    }
  };

  if (kt_begin < kt_end) {
    stage(0, kt_begin);
    load_w(kt_begin, wcur, gcur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int buf = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = kt + 1 < kt_end;
    if (more) {
      stage(buf ^ 1, kt + 1);
      load_w(kt + 1, wnext, gnext);
    }
    const char* sb = smem[buf] + lane * 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      half8_t bf[BMT];
#pragma unroll
      for (int mt = 0; mt < BMT; ++mt) bf[mt] = *(const half8_t*)(sb + (t * BMT + mt) * 1024);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const half8_t a = dequant8(wcur[j][t], gcur[j][G128 ? 0 : t]);
#pragma unroll
        for (int mt = 0; mt < BMT; ++mt) acc[j][mt] = mfma16(a, bf[mt], acc[j][mt]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next tile landed in LDS (and in wnext)
    __syncthreads();
    if (more) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        wcur[j] = wnext[j];
#pragma unroll
        for (int t = 0; t < (G128 ? 1 : 4); ++t) gcur[j][t] = gnext[j][t];
      }
    }
    buf ^= 1;
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nc = (nt0 + j) * 16 + 4 * q;
    half4_t b = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    if (bias && ksplit == 1) b = *(const half4_t*)(bias + nc);
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) {
      const int m = m0 + mt * 16 + n16;
      if (m < M) {
        if (ksplit == 1) {
          half4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (half_t)(acc[j][mt][r] + (float)b[r]);
          *(half4_t*)(Y + (size_t)m * N + nc) = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(Yacc + (size_t)m * N + nc + r, acc[j][mt][r]);
        }
      }
    }
  }
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
    const GroupQ g = load_group(S, QZ, k0 / G, n, N);
    const half8_t a = dequant8(w[t], g);
#pragma unroll
    for (int j = 0; j < 8; ++j) W[(size_t)(k0 + j) * N + n] = a[j];
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

struct Plan {
  int kernel;  // QUICK_KERNEL_SKINNY / QUICK_KERNEL_TILED
  int mt;      // skinny: token tiles per workgroup; tiled: BMT
  int tn;      // tiled: channel tiles per wave
  int waves;   // skinny: waves per workgroup
  int ksplit;  // K slices across workgroups (fp32 atomics + finalize when > 1)
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

static Plan make_plan(int M, int K, int N, int kernel, int grid_split_k) {
  Plan p{};
  const int KT = K / 128;
  if (kernel == QUICK_KERNEL_AUTO) kernel = (M <= 16) ? QUICK_KERNEL_SKINNY : QUICK_KERNEL_TILED;
  p.kernel = kernel;
  if (kernel == QUICK_KERNEL_SKINNY) {
    p.mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    p.waves = 8;
    const int mblocks = (M + p.mt * 16 - 1) / (p.mt * 16);
    int ks = 1;
    // fill the 256 CUs when N is small: every workgroup should still own >= 8 k-tiles
    while ((N / 16) * mblocks * ks < 256 && KT / (ks * 2) >= 8) ks *= 2;
    p.ksplit = grid_split_k > 0 ? grid_split_k : ks;
  } else {
    p.mt = (M <= 32) ? 2 : (M <= 64 ? 4 : 8);
    p.tn = 2;
    p.ksplit = grid_split_k > 0 ? grid_split_k : 1;
  }
  p.ksplit = std::max(1, std::min(p.ksplit, KT));
  return p;
}

// Where the main kernel goes: the stream, plus an optional event pair bound to that one dispatch
// (hipExtLaunchKernelGGL) so that a profiler-free caller can read the kernel's own duration.
struct Launch {
  hipStream_t st;
  hipEvent_t start, stop;
};

template <int MT, int WAVES>
static void launch_skinny(const Plan& p, const void* x, const void* qw, const void* s, const void* qz, const void* bias,
                          void* y, float* yacc, int M, int K, int N, int G, const Launch& L) {
  dim3 grid(N / 16, (M + MT * 16 - 1) / (MT * 16), p.ksplit), block(WAVES * 64);
  if (G % 128 == 0)
    hipExtLaunchKernelGGL((w4a16_skinny_kernel<MT, WAVES, true>), grid, block, 0, L.st, L.start, L.stop, 0,
                          (const half_t*)x, (const u32x4*)qw, (const half_t*)s, (const uint32_t*)qz, (const half_t*)bias,
                          (half_t*)y, yacc, M, K, N, G, p.ksplit);
  else
    hipExtLaunchKernelGGL((w4a16_skinny_kernel<MT, WAVES, false>), grid, block, 0, L.st, L.start, L.stop, 0,
                          (const half_t*)x, (const u32x4*)qw, (const half_t*)s, (const uint32_t*)qz, (const half_t*)bias,
                          (half_t*)y, yacc, M, K, N, G, p.ksplit);
}

template <int BMT, int TN>
static void launch_tiled(const Plan& p, const void* x, const void* qw, const void* s, const void* qz, const void* bias,
                         void* y, float* yacc, int M, int K, int N, int G, const Launch& L) {
  dim3 grid((N / (64 * TN)) * ((M + BMT * 16 - 1) / (BMT * 16)), p.ksplit), block(256);
  if (G % 128 == 0)
    hipExtLaunchKernelGGL((w4a16_tiled_kernel<BMT, TN, true>), grid, block, 0, L.st, L.start, L.stop, 0,
                          (const half_t*)x, (const u32x4*)qw, (const half_t*)s, (const uint32_t*)qz, (const half_t*)bias,
                          (half_t*)y, yacc, M, K, N, G, p.ksplit);
  else
    hipExtLaunchKernelGGL((w4a16_tiled_kernel<BMT, TN, false>), grid, block, 0, L.st, L.start, L.stop, 0,
                          (const half_t*)x, (const u32x4*)qw, (const half_t*)s, (const uint32_t*)qz, (const half_t*)bias,
                          (half_t*)y, yacc, M, K, N, G, p.ksplit);
}

static int run_gemm(const void* x, const void* qweight, const void* scales, const void* qzeros, const void* bias, void* y,
                    void* workspace, size_t workspace_bytes, int M, int K, int N, int G, int kernel, int grid_split_k,
                    const Launch& L) {
  if (int rc = check_shapes(M, K, N, G)) return rc;
  if (!x || !qweight || !scales || !qzeros || !y) return fail(QUICK_ERR_INVALID_ARGUMENT, "null tensor pointer");
  if (kernel < QUICK_KERNEL_AUTO || kernel > QUICK_KERNEL_TILED)
    return fail(QUICK_ERR_INVALID_ARGUMENT, "unknown kernel id %d", kernel);
  hipStream_t st = L.st;
  const Plan p = make_plan(M, K, N, kernel, grid_split_k);
  float* yacc = nullptr;
  if (p.ksplit > 1) {
    const size_t need = (size_t)M * N * sizeof(float);
    if (!workspace || workspace_bytes < need)
      return fail(QUICK_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    yacc = (float*)workspace;
    if (hipMemsetAsync(yacc, 0, need, st) != hipSuccess) return fail(QUICK_ERR_LAUNCH, "hipMemsetAsync failed");
  }
  if (p.kernel == QUICK_KERNEL_SKINNY) {
    switch (p.mt) {
      case 1: launch_skinny<1, 8>(p, x, qweight, scales, qzeros, bias, y, yacc, M, K, N, G, L); break;
      case 2: launch_skinny<2, 8>(p, x, qweight, scales, qzeros, bias, y, yacc, M, K, N, G, L); break;
      default: launch_skinny<4, 8>(p, x, qweight, scales, qzeros, bias, y, yacc, M, K, N, G, L); break;
    }
  } else {
    switch (p.mt) {
      case 2: launch_tiled<2, 2>(p, x, qweight, scales, qzeros, bias, y, yacc, M, K, N, G, L); break;
      case 4: launch_tiled<4, 2>(p, x, qweight, scales, qzeros, bias, y, yacc, M, K, N, G, L); break;
      default: launch_tiled<8, 2>(p, x, qweight, scales, qzeros, bias, y, yacc, M, K, N, G, L); break;
    }
  }
  if (p.ksplit > 1) {
    const size_t n4 = ((size_t)M * N + 3) / 4;
    hipLaunchKernelGGL(w4a16_finalize_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, yacc,
                       (const half_t*)bias, (half_t*)y, M, N);
  }
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
  const Plan p = make_plan(M, K, N, kernel, grid_split_k);
  return p.ksplit > 1 ? (size_t)M * N * sizeof(float) : 0;
}

size_t quick_w4a16_workspace_bytes(int M, int K, int N, int group_size, int split_k_iters) {
  (void)split_k_iters;
  return quick_w4a16_workspace_bytes_ex(M, K, N, group_size, QUICK_KERNEL_AUTO, 0);
}

int quick_w4a16_gemm_f16_ex(const void* x, const void* qweight, const void* scales, const void* qzeros,
                            const void* bias, void* y, void* workspace, size_t workspace_bytes, int M, int K, int N,
                            int group_size, int kernel, int grid_split_k, void* hip_stream) {
  const Launch L{(hipStream_t)hip_stream, nullptr, nullptr};
  return run_gemm(x, qweight, scales, qzeros, bias, y, workspace, workspace_bytes, M, K, N, group_size, kernel,
                  grid_split_k, L);
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
    rc = run_gemm(x, qweights[s], scales[s], qzeros[s], nullptr, y, workspace, workspace_bytes, M, K, N, group_size,
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
