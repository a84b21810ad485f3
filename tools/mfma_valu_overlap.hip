// Exploration tool (not product): how many independent packed-f16 VALU ops hide under one MFMA on gfx950?
// One wave per SIMD (256-thread block, 1 block per CU), loop of { NM independent MFMAs ; NV independent VALU ops }.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NV, int SHAPE, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD) void k(float* out, int iters) {
  half8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  floatx4 c4[4] = {};
  floatx16 c16[2] = {};
  half2_t v[8];
  for (int i = 0; i < 8; ++i) v[i] = half2_t{(_Float16)(threadIdx.x + i), (_Float16)1};
  const half2_t m = {(_Float16)1.0009765625f, (_Float16)0.9990234375f};
  const long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (SHAPE == 16) c4[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[r], 0, 0, 0);
      else c16[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16[r & 1], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(r * NV + j) & 7] = v[(r * NV + j) & 7] * m;
    }
  }
  const long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 4; ++r) s += c4[r][0];
  s += c16[0][0] + c16[1][0];
  for (int i = 0; i < 8; ++i) s += (float)v[i][0];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0) / (iters * 4); }
  if (s == 12345.678f) out[1] = s;
}

template <int NV, int SHAPE, int W>
static void run(float* out) {
  hipLaunchKernelGGL((k<NV, SHAPE, W>), dim3(256), dim3(256 * W), 0, 0, out, 2000);
  float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
  printf("mfma %dx%d  waves/SIMD %d  VALU per MFMA %d : %6.1f cycles per (MFMA + VALUs)\n", SHAPE, SHAPE, W, NV, h[0]);
}
int main() {
  float* out; hipMalloc(&out, 64);
  run<0, 16, 1>(out); run<1, 16, 1>(out); run<2, 16, 1>(out); run<3, 16, 1>(out); run<4, 16, 1>(out); run<6, 16, 1>(out); run<8, 16, 1>(out);
  run<0, 32, 1>(out); run<2, 32, 1>(out); run<4, 32, 1>(out); run<6, 32, 1>(out); run<8, 32, 1>(out); run<12, 32, 1>(out);
  run<0, 16, 2>(out); run<2, 16, 2>(out); run<4, 16, 2>(out); run<8, 16, 2>(out);
  run<0, 32, 2>(out); run<4, 32, 2>(out); run<8, 32, 2>(out);
  return 0;
}
