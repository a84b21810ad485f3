// Exploration tool (not product): the tiled kernel's instruction pattern -- a cluster of NV packed-f16 VALU ops (the
// dequantisation of one dword is 13) followed by a cluster of 4 independent 16x16x32 MFMAs -- with W waves per SIMD.
// Prints shader cycles per {NV VALU; 4 MFMA} group as seen by one wave: if the VALU cluster of one wave runs under the
// MFMA cluster of another, W = 2 costs max(2 * 64, 2 * 4.7 NV) per pair of groups instead of their sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NV, int W, bool SKEW>
__global__ __launch_bounds__(256 * W) void k(float* out_f, int iters) {
  unsigned long long* out = (unsigned long long*)out_f;
  half8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  floatx4 c[4] = {};
  half2_t v[16];
  for (int i = 0; i < 16; ++i) v[i] = half2_t{(_Float16)(threadIdx.x + i), (_Float16)1};
  const half2_t m = {(_Float16)1.0009765625f, (_Float16)0.9990234375f};
  if (SKEW && (threadIdx.x >> 8) & 1) __builtin_amdgcn_s_sleep(1);  // every other wave of a SIMD starts 64 cycles late
  const long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j & 15] = v[j & 15] * m;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[r], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  const long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 4; ++r) s += c[r][0];
  for (int i = 0; i < 16; ++i) s += (float)v[i][0];
  // first start -> last end over the waves of one CU (the oldest wave alone would show its own latency-bound pace:
  // the arbiter serves it first)
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
    atomicMin(out + 1, (unsigned long long)t0);
    atomicMax(out + 2, (unsigned long long)t1);
  }
  if (s == 12345.678f) out[0] = (unsigned long long)s;
}

template <int NV, int W, bool SKEW>
static void run(float* out) {
  unsigned long long init[3] = {0, ~0ull, 0};
  hipMemcpy(out, init, sizeof(init), hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k<NV, W, SKEW>), dim3(256), dim3(256 * W), 0, 0, out, 4000);
  unsigned long long h[3]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const double per_group = (double)(h[2] - h[1]) / 4000;
  printf("waves/SIMD %d%s  {%2d VALU; 4 MFMA}: %6.1f cycles per round of groups = %6.1f per group per SIMD (MFMA alone: 64)\n", W,
         SKEW ? " skewed" : "       ", NV, per_group, per_group / W);
}
int main() {
  float* out; hipMalloc(&out, 64);  // [unused, min start, max end] as 64-bit words
  run<0, 1, false>(out); run<13, 1, false>(out); run<26, 1, false>(out);
  run<0, 2, false>(out); run<13, 2, false>(out); run<26, 2, false>(out);
  run<13, 2, true>(out); run<26, 2, true>(out);
  run<0, 4, false>(out); run<13, 4, false>(out); run<26, 4, false>(out);
  return 0;
}
