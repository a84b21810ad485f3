// Format bridge between the reference's packed order ("cuda order": what
// WQLinear_QUICK.from_linear writes, quick/awq/modules/linear/quick.py:88-150, and what the CUDA
// kernels index, csrc/gemm_cuda_quick.cu:1257-1277) and the MFMA-fragment order consumed by
// w4a16_gemm.hip ("mi355x order").  One-time, load-side work: gather kernels, one thread per
// output dword.
#include "w4a16_common.hpp"
#include "../../include/quick_amd.h"

namespace quick_amd {

// dword index + nibble of logical weight (k, n) in the reference layout (closed form)
__device__ __forceinline__ void cuda_order_pos(int k, int n, int N, size_t& idx, int& nib) {
  const int kt = k >> 5, half = (k >> 4) & 1, r = k & 15;
  const int l4 = (r & 7) >> 1, hi = r >> 3, odd = r & 1;
  const int bx = n >> 7, ty = (n >> 6) & 1, chunk = (n & 63) >> 4, t = (n & 15) >> 3, j = n & 7;
  idx = (size_t)kt * (4 * N) + (size_t)((2 * ty + (j >> 2)) * (N >> 3) + 16 * bx + 4 * (j & 3) + l4) * 8 + 4 * half + chunk;
  nib = 4 * odd + hi + 2 * t;
}

// slot x(n) of the reference scale / zero permutation
__device__ __forceinline__ int cuda_order_slot(int n, int N) {
  const int bx = n >> 7, ty = (n >> 6) & 1, chunk = (n & 63) >> 4, t = (n & 15) >> 3, j = n & 7;
  return ((2 * ty + (j >> 2)) * (N >> 5) + 4 * bx + (j & 3)) * 8 + 2 * chunk + t;
}

// (k, n) of nibble p of mi355x-order dword d
__device__ __forceinline__ void mi355x_dword_coord(size_t d, int K, int& k0, int& n) {
  const int KT = K >> 7;
  const int t = (int)(d & 3), lane = (int)((d >> 2) & 63);
  const size_t tile = d >> 8;
  const int kt = (int)(tile % KT), nt = (int)(tile / KT);
  n = nt * 16 + (lane & 15);
  k0 = kt * 128 + t * 32 + 8 * (lane >> 4);
}

// K = rows of the OUTPUT (a multiple of 128); rows >= K_in (a multiple of 32: a dword is all inside or all outside) are padding
__global__ __launch_bounds__(256) void repack_weight_cuda_to_mi355x(const uint32_t* __restrict__ in,
                                                                    uint32_t* __restrict__ out, int K, int N, int K_in) {
  const size_t d = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (d >= (size_t)K * N / 8) return;
  int k0, n;
  mi355x_dword_coord(d, K, k0, n);
  uint32_t v = 0;
  if (k0 >= K_in) {
    out[d] = 0u;
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    size_t idx;
    int nib;
    cuda_order_pos(k0 + j, n, N, idx, nib);
    v |= ((in[idx] >> (4 * nib)) & 15u) << (4 * (4 * (j & 1) + (j >> 1)));
  }
  out[d] = v;
}

__global__ __launch_bounds__(256) void repack_weight_mi355x_to_cuda(const uint32_t* __restrict__ in,
                                                                    uint32_t* __restrict__ out, int K, int N) {
  // one thread per (k, n); scatter with atomicOr into a zeroed buffer
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)K * N) return;
  const int k = (int)(i / N), n = (int)(i % N);
  const int KT = K >> 7;
  const int lane = (n & 15) + 16 * ((k & 31) >> 3), j = k & 7;
  const size_t d = (((size_t)(n >> 4) * KT + (k >> 7)) * 64 + lane) * 4 + ((k & 127) >> 5);
  const uint32_t w = (in[d] >> (4 * (4 * (j & 1) + (j >> 1)))) & 15u;
  size_t idx;
  int nib;
  cuda_order_pos(k, n, N, idx, nib);
  atomicOr(out + idx, w << (4 * nib));
}

// scales / zeros: one thread per (g, n).  MI355X order: the scales tensor holds one 32-bit word per (g, n) -- fp16 scale |
// zero point << 16 -- at word ((n/16) * NG + g) * 16 + n%16 (group_word_index, w4a16_common.hpp); qzeros keeps a plain
// copy of the zero points (nibble n%8 of dword n/8) that the GEMM kernels do not read.
__global__ __launch_bounds__(256) void repack_sz_cuda_to_mi355x(const half_t* __restrict__ s_in,
                                                                const uint32_t* __restrict__ z_in,
                                                                uint32_t* __restrict__ sz_out, uint32_t* __restrict__ z_out,
                                                                int NG, int N, int NG_in) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)NG * N) return;
  const int g = (int)(i / N), n = (int)(i % N);
  if (g >= NG_in) {  // a padding group: scale 0, zero point 0 (z_out was zeroed)
    sz_out[group_word_index(g, n, NG)] = 0u;
    return;
  }
  const int x = cuda_order_slot(n, N);
  const uint32_t z = (z_in[(size_t)g * (N >> 2) + (x >> 2)] >> (4 * (x & 3))) & 15u;
  const uint32_t sbits = __builtin_bit_cast(unsigned short, s_in[(size_t)g * 2 * N + 2 * x]);
  sz_out[group_word_index(g, n, NG)] = sbits | (z << 16);
  atomicOr(z_out + (size_t)g * (N >> 2) + (n >> 3), z << (4 * (n & 7)));
}

__global__ __launch_bounds__(256) void repack_sz_mi355x_to_cuda(const uint32_t* __restrict__ sz_in,
                                                                half_t* __restrict__ s_out, uint32_t* __restrict__ z_out,
                                                                int NG, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)NG * N) return;
  const int g = (int)(i / N), n = (int)(i % N);
  const int x = cuda_order_slot(n, N);
  const uint32_t w = sz_in[group_word_index(g, n, NG)];
  const half_t s = __builtin_bit_cast(half_t, (unsigned short)(w & 0xffffu));
  s_out[(size_t)g * 2 * N + 2 * x] = s;
  s_out[(size_t)g * 2 * N + 2 * x + 1] = s;
  const uint32_t z = (w >> 16) & 15u;
  atomicOr(z_out + (size_t)g * (N >> 2) + (x >> 2), (z << (4 * (x & 3))) | (z << (4 * (x & 3) + 16)));
}

static int check(int K, int N, int G) {
  if (K <= 0 || N <= 0 || G <= 0 || N % 128 != 0 || G % 32 != 0 || K % G != 0) return QUICK_ERR_INVALID_ARGUMENT;
  if (K % 128 != 0) return QUICK_ERR_UNSUPPORTED;
  return QUICK_OK;
}

}  // namespace quick_amd

using namespace quick_amd;

extern "C" {

static int repack_to_mi355x(const void* qweight_in, const void* scales_in, const void* qzeros_in, void* qweight_out,
                            void* scales_out, void* qzeros_out, int K_in, int K, int N, int group_size, hipStream_t st) {
  const int NG = K / group_size;
  const size_t nd = (size_t)K * N / 8;
  if (hipMemsetAsync(qzeros_out, 0, (size_t)NG * (N / 4) * 4, st) != hipSuccess) return QUICK_ERR_LAUNCH;
  hipLaunchKernelGGL(repack_weight_cuda_to_mi355x, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st,
                     (const uint32_t*)qweight_in, (uint32_t*)qweight_out, K, N, K_in);
  hipLaunchKernelGGL(repack_sz_cuda_to_mi355x, dim3((unsigned)(((size_t)NG * N + 255) / 256)), dim3(256), 0, st,
                     (const half_t*)scales_in, (const uint32_t*)qzeros_in, (uint32_t*)scales_out, (uint32_t*)qzeros_out, NG, N,
                     K_in / group_size);
  return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
}

int quick_repack_cuda_to_mi355x(const void* qweight_in, const void* scales_in, const void* qzeros_in, void* qweight_out,
                                void* scales_out, void* qzeros_out, int K, int N, int group_size, void* hip_stream) {
  if (int rc = check(K, N, group_size)) return rc;
  return repack_to_mi355x(qweight_in, scales_in, qzeros_in, qweight_out, scales_out, qzeros_out, K, K, N, group_size, (hipStream_t)hip_stream);
}

int quick_padded_in_features(int K, int group_size) {
  if (K <= 0 || group_size <= 0 || group_size % 32 != 0 || K % 32 != 0 || K % group_size != 0) return 0;
  int a = 128, b = group_size;
  while (b) {
    const int t = a % b;
    a = b;
    b = t;
  }
  const long unit = 128L * group_size / a;  // lcm(128, group_size)
  return (int)(((long)K + unit - 1) / unit * unit);
}

int quick_repack_cuda_to_mi355x_padded(const void* qweight_in, const void* scales_in, const void* qzeros_in, void* qweight_out,
                                       void* scales_out, void* qzeros_out, int K, int N, int group_size, void* hip_stream) {
  const int Kp = quick_padded_in_features(K, group_size);
  if (Kp == 0 || N <= 0 || N % 128 != 0) return QUICK_ERR_INVALID_ARGUMENT;
  return repack_to_mi355x(qweight_in, scales_in, qzeros_in, qweight_out, scales_out, qzeros_out, K, Kp, N, group_size, (hipStream_t)hip_stream);
}

int quick_repack_mi355x_to_cuda(const void* qweight_in, const void* scales_in, const void* qzeros_in, void* qweight_out,
                                void* scales_out, void* qzeros_out, int K, int N, int group_size, void* hip_stream) {
  if (int rc = check(K, N, group_size)) return rc;
  hipStream_t st = (hipStream_t)hip_stream;
  const int NG = K / group_size;
  if (hipMemsetAsync(qweight_out, 0, (size_t)K * N / 2, st) != hipSuccess) return QUICK_ERR_LAUNCH;
  if (hipMemsetAsync(qzeros_out, 0, (size_t)NG * (N / 4) * 4, st) != hipSuccess) return QUICK_ERR_LAUNCH;
  hipLaunchKernelGGL(repack_weight_mi355x_to_cuda, dim3((unsigned)(((size_t)K * N + 255) / 256)), dim3(256), 0, st,
                     (const uint32_t*)qweight_in, (uint32_t*)qweight_out, K, N);
  (void)qzeros_in;  // the zero points travel in the scales tensor's words
  hipLaunchKernelGGL(repack_sz_mi355x_to_cuda, dim3((unsigned)(((size_t)NG * N + 255) / 256)), dim3(256), 0, st,
                     (const uint32_t*)scales_in, (half_t*)scales_out, (uint32_t*)qzeros_out, NG, N);
  return hipGetLastError() == hipSuccess ? QUICK_OK : QUICK_ERR_LAUNCH;
}

}  // extern "C"
