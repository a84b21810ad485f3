// Exploration tool (not product): the wide kernels' unit in isolation -- the REAL dequantisation of one packed dword
// (13 VALU: shift, 4 and_or, 2 pk_add, 2 pk_fma, 4 pk_mul) interleaved with NM independent 32x32x16 MFMAs, W waves per
// SIMD, no LDS, no loads, no barrier.  Prints shader cycles per unit per SIMD: NM * 32 = the matrix pipe alone.
//   MODE 0: dequant as in the kernel (packed f16)      MODE 1: the same count of plain v_and_or ops (no packed math)
//   MODE 2: MFMAs only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }

template <int NM, int W, int MODE>
__global__ __launch_bounds__(256 * W) void k(float* out_f, int iters, uint32_t seed) {
  unsigned long long* out = (unsigned long long*)out_f;
  uint32_t mlo = 0x000f000fu, mhi = 0x00f000f0u, magic = 0x64006400u;
  asm volatile("" : "+s"(mlo), "+s"(mhi));
  asm volatile("" : "+v"(magic));
  const half2_t sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  const half2_t s2 = {(_Float16)0.013f, (_Float16)0.013f}, nzlo = as_h2(0xE400E400u | 0x00070007u), nzhi = as_h2(0xD400D400u | 0x00700070u);
  half8_t b = {1, 1, 1, 1, 1, 1, 1, 1};
  floatx16 c[NM] = {};
  uint32_t q = seed + threadIdx.x * 2654435761u;
  half8_t af = {1, 2, 3, 4, 5, 6, 7, 8};
  const long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    half8_t nx = af;
    if constexpr (MODE == 0) {
      const uint32_t q8 = q >> 8;
      const half2_t h0 = (as_h2((q & mlo) | magic) + nzlo) * s2;
      const half2_t h1 = (as_h2((q & mhi) | magic) * sixteenth + nzhi) * s2;
      const half2_t h2 = (as_h2((q8 & mlo) | magic) + nzlo) * s2;
      const half2_t h3 = (as_h2((q8 & mhi) | magic) * sixteenth + nzhi) * s2;
      nx[0] = h0[0]; nx[1] = h0[1]; nx[2] = h1[0]; nx[3] = h1[1]; nx[4] = h2[0]; nx[5] = h2[1]; nx[6] = h3[0]; nx[7] = h3[1];
      q = q * 1664525u + 1013904223u;
    } else if constexpr (MODE == 1) {
      u32x4 r = __builtin_bit_cast(u32x4, af);
      uint32_t t = q;
#pragma unroll
      for (int j = 0; j < 13; ++j) { t = (t & mlo) | (r[j & 3] + j); r[j & 3] = t ^ magic; }
      nx = __builtin_bit_cast(half8_t, r);
      q = t;
    }
#pragma unroll
    for (int r = 0; r < NM; ++r) c[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b, c[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NM; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, (15 + NM - 1) / NM, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    af = nx;
  }
  const long t1 = __builtin_amdgcn_s_memtime();
  float s = (float)af[0];
  for (int r = 0; r < NM; ++r) s += c[r][0];
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
    atomicMin(out + 1, (unsigned long long)t0);
    atomicMax(out + 2, (unsigned long long)t1);
  }
  if (s == 12345.678f) out[0] = (unsigned long long)s;
}

template <int NM, int W, int MODE>
static void run(float* out) {
  unsigned long long init[3] = {0, ~0ull, 0};
  hipMemcpy(out, init, sizeof(init), hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k<NM, W, MODE>), dim3(256), dim3(256 * W), 0, 0, out, 4000, 12345u);
  unsigned long long h[3]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const double per_unit = (double)(h[2] - h[1]) / 4000;
  const char* names[] = {"dequant (packed f16)", "13 integer VALU      ", "MFMAs only          "};
  printf("waves/SIMD %d  %s + %d MFMA32: %6.1f cycles per unit per wave = %6.1f per unit per SIMD (matrix pipe alone %d)\n", W, names[MODE], NM,
         per_unit, per_unit / W, NM * 32);
}
int main() {
  float* out; hipMalloc(&out, 64);
  run<2, 1, 2>(out); run<2, 1, 0>(out); run<2, 1, 1>(out);
  run<2, 2, 2>(out); run<2, 2, 0>(out); run<2, 2, 1>(out);
  run<4, 1, 2>(out); run<4, 1, 0>(out); run<4, 1, 1>(out);
  run<4, 2, 2>(out); run<4, 2, 0>(out); run<4, 2, 1>(out);
  run<8, 1, 0>(out); run<8, 2, 0>(out);
  return 0;
}
