// Exploration tool (not product): which VALU instructions issue in the shadow of a 32x32x16 MFMA of the same wave?
// One wave per SIMD, loop of { MFMA ; NV VALU ops } x NM accumulators; the VALU ops are INDEPENDENT of the MFMA operands.
//   KIND 0: v_pk_mul_f16 (8 independent registers)      KIND 1: v_and_or_b32 (SGPR mask)      KIND 2: v_pk_add_f16
//   KIND 3: the dequantisation chain of the wide kernels (and_or -> pk_add -> pk_mul / and_or -> pk_fma -> pk_mul) on its own registers
//   KIND 4: as 3, and the result IS the next unit's A operand (single register set)
//   KIND 5: as 4 with the A operand double-buffered -- the unit's MFMAs read one set while the chain writes the other (the kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t as_u(half2_t h) { return __builtin_bit_cast(uint32_t, h); }

template <int NV, int NM, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, uint32_t seed) {
  half8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  floatx16 c[NM] = {};
  uint32_t mlo = 0x000f000fu, mhi = 0x00f000f0u;
  asm volatile("" : "+s"(mlo), "+s"(mhi));
  uint32_t magic = 0x64006400u;
  asm volatile("" : "+v"(magic));
  half2_t v[8];
  uint32_t w[8];
  for (int i = 0; i < 8; ++i) { v[i] = half2_t{(_Float16)(threadIdx.x + i), (_Float16)1}; w[i] = seed * (i + 1) + threadIdx.x; }
  const half2_t m = {(_Float16)1.0009765625f, (_Float16)0.9990234375f}, nz = as_h2(0xE407E407u), six = {(_Float16)0.0625f, (_Float16)0.0625f};
  const long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < NM; ++r) {
      c[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[r], 0, 0, 0);
      if constexpr (KIND == 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[(r * NV + j) & 7] = v[(r * NV + j) & 7] * m;
      } else if constexpr (KIND == 1) {
#pragma unroll
        for (int j = 0; j < NV; ++j) w[(r * NV + j) & 7] = (w[(r * NV + j) & 7] & mlo) | magic;
      } else if constexpr (KIND == 2) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[(r * NV + j) & 7] = v[(r * NV + j) & 7] + m;
      } else {  // NV ignored: one packed dword = 13 ops per NM MFMAs, a share after each MFMA (the compiler's order)
        if (r == 0) {
          const uint32_t q = w[0], q8 = q >> 8;
          v[0] = (as_h2((q & mlo) | magic) + nz) * m;
          v[1] = (as_h2((q & mhi) | magic) * six + nz) * m;
          v[2] = (as_h2((q8 & mlo) | magic) + nz) * m;
          v[3] = (as_h2((q8 & mhi) | magic) * six + nz) * m;
          w[0] = q * 1664525u + 1013904223u;
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, KIND >= 3 ? (15 + NM - 1) / NM : NV, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KIND == 4) { a[0] = v[0][0]; a[1] = v[0][1]; a[2] = v[1][0]; a[3] = v[1][1]; a[4] = v[2][0]; a[5] = v[2][1]; a[6] = v[3][0]; a[7] = v[3][1]; }
  }
  const long t1 = __builtin_amdgcn_s_memtime();
  float s = (float)a[0];
  for (int r = 0; r < NM; ++r) s += c[r][0];
  for (int i = 0; i < 8; ++i) s += (float)v[i][0] + (float)w[i];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * NM);
  if (s == 12345.678f) out[1] = s;
}

template <int NM, int W = 1, bool NODQ = false>
__global__ __launch_bounds__(256 * W) void k5(float* out, int iters, uint32_t seed) {
  half8_t a[2] = {{1, 2, 3, 4, 5, 6, 7, 8}, {2, 3, 4, 5, 6, 7, 8, 9}}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  floatx16 c[NM] = {};
  uint32_t mlo = 0x000f000fu, mhi = 0x00f000f0u;
  asm volatile("" : "+s"(mlo), "+s"(mhi));
  uint32_t magic = 0x64006400u;
  asm volatile("" : "+v"(magic));
  uint32_t q = seed + threadIdx.x;
  const half2_t m = {(_Float16)1.0009765625f, (_Float16)0.9990234375f}, nz = as_h2(0xE407E407u), six = {(_Float16)0.0625f, (_Float16)0.0625f};
  const long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t q8 = q >> 8;
      const half2_t h0 = (as_h2((q & mlo) | magic) + nz) * m, h1 = (as_h2((q & mhi) | magic) * six + nz) * m;
      const half2_t h2 = (as_h2((q8 & mlo) | magic) + nz) * m, h3 = (as_h2((q8 & mhi) | magic) * six + nz) * m;
      q = q * 1664525u + 1013904223u;
#pragma unroll
      for (int r = 0; r < NM; ++r) {
        c[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b, c[r], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (15 + NM - 1) / NM, 0);
      }
      if constexpr (!NODQ) a[u ^ 1] = half8_t{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long t1 = __builtin_amdgcn_s_memtime();
  float s = (float)a[0][0] + (float)a[1][0] + (float)q;
  for (int r = 0; r < NM; ++r) s += c[r][0];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * NM);
  if (s == 12345.678f) out[1] = s;
}
template <int NM, int W = 1, bool NODQ = false>
static void run5(float* out) {
  hipLaunchKernelGGL((k5<NM, W, NODQ>), dim3(256), dim3(256 * W), 0, 0, out, 2000, 12345u);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k5<NM, W, NODQ>), dim3(256), dim3(256 * W), 0, 0, out, 20000, 12345u);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double pf = 256.0 * 4 * W * 20000.0 * NM * 32768.0 / (ms * 1e-3) / 1e15;
  float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
  if (NODQ) printf("%d accumulators  MFMAs only, %d wave(s) per SIMD: %6.1f cycles per MFMA per wave = %5.1f per SIMD   (wall clock: %.2f PFLOP/s chip-wide)\n", NM, W, h[0], h[0] / W, pf);
  else if (W > 1) printf("%d accumulators  dequant chain -> next A operand, double-buffered, %d waves per SIMD: %6.1f cycles per MFMA per wave = %5.1f per SIMD\n", NM, W, h[0], h[0] / W);
  else printf("%d accumulators  dequant chain -> next A operand, double-buffered (13 ops per %d MFMAs): %6.1f cycles per MFMA\n", NM, NM, h[0]);
}

// KIND 6: KIND 5 plus the B operand of every MFMA read from LDS three MFMA-pairs ahead (ds_read_b128, XOR-ed address), as in
// the kernels' K loop; LDADD extra SALU + VALU ops per unit stand in for the ring bookkeeping
template <int NM, int EXTRA, int W = 1>
__global__ __launch_bounds__(256 * W) void k6(float* out, int iters, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  for (int i = threadIdx.x; i < 32768 / 4; i += 256 * W) ((uint32_t*)lds)[i] = 0x3c003c00u;
  __syncthreads();
  half8_t a[2] = {{1, 2, 3, 4, 5, 6, 7, 8}, {2, 3, 4, 5, 6, 7, 8, 9}};
  floatx16 c[NM] = {};
  uint32_t mlo = 0x000f000fu, mhi = 0x00f000f0u;
  asm volatile("" : "+s"(mlo), "+s"(mhi));
  uint32_t magic = 0x64006400u;
  asm volatile("" : "+v"(magic));
  uint32_t q = seed + threadIdx.x;
  const half2_t m = {(_Float16)1.0009765625f, (_Float16)0.9990234375f}, nz = as_h2(0xE407E407u), six = {(_Float16)0.0625f, (_Float16)0.0625f};
  const unsigned lane = threadIdx.x & 63, xrd = (lane & 31) * 256u + (((lane >> 5) ^ (lane & 15)) << 4);
  constexpr int D = 3;
  half8_t bf[D + 1][NM];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int r = 0; r < NM; ++r) bf[d][r] = *(const half8_t*)(lds + ((xrd ^ (d << 5)) + r * 8192) % 32768);
  uint32_t extra = seed;
  const long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t q8 = q >> 8;
      const half2_t h0 = (as_h2((q & mlo) | magic) + nz) * m, h1 = (as_h2((q & mhi) | magic) * six + nz) * m;
      const half2_t h2 = (as_h2((q8 & mlo) | magic) + nz) * m, h3 = (as_h2((q8 & mhi) | magic) * six + nz) * m;
      q = q * 1664525u + 1013904223u;
#pragma unroll
      for (int r = 0; r < NM; ++r) bf[(u + D) & 3][r] = *(const half8_t*)(lds + ((xrd ^ (((u + D) & 7) << 5)) + r * 8192) % 32768);
#pragma unroll
      for (int e = 0; e < EXTRA; ++e) extra = extra * 3u + (uint32_t)e;
#pragma unroll
      for (int r = 0; r < NM; ++r) {
        c[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 1], bf[u & 3][r], c[r], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (15 + EXTRA + NM - 1) / NM, 0);
      }
      a[(u & 1) ^ 1] = half8_t{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long t1 = __builtin_amdgcn_s_memtime();
  float s = (float)a[0][0] + (float)a[1][0] + (float)q + (float)extra;
  for (int r = 0; r < NM; ++r) s += c[r][0];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * NM);
  if (s == 12345.678f) out[1] = s;
}
template <int NM, int EXTRA, int W = 1>
static void run6(float* out) {
  hipLaunchKernelGGL((k6<NM, EXTRA, W>), dim3(256), dim3(256 * W), 0, 0, out, 2000, 12345u);
  float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
  printf("%d accumulators  dequant -> A (double-buffered) + B from LDS 3 units ahead + %d extra VALU per unit, %d wave(s) per SIMD: %6.1f cycles per MFMA per wave = %5.1f per SIMD\n",
         NM, EXTRA, W, h[0], h[0] / W);
}

template <int NV, int NM, int KIND>
static void run(float* out) {
  hipLaunchKernelGGL((k<NV, NM, KIND>), dim3(256), dim3(256), 0, 0, out, 2000, 12345u);
  float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
  const char* names[] = {"v_pk_mul_f16 ", "v_and_or_b32 ", "v_pk_add_f16 ", "dequant chain (own registers)", "dequant chain -> next A operand"};
  if (KIND >= 3) printf("%d accumulators  %s (13 ops per %d MFMAs): %6.1f cycles per MFMA\n", NM, names[KIND], NM, h[0]);
  else printf("%d accumulators  %d x %s per MFMA: %6.1f cycles per MFMA\n", NM, NV, names[KIND], h[0]);
}
int main() {
  float* out; hipMalloc(&out, 64);
  run<0, 2, 0>(out); run<0, 4, 0>(out);
  run<3, 2, 0>(out); run<6, 2, 0>(out); run<6, 4, 0>(out); run<8, 4, 0>(out);
  run<3, 2, 1>(out); run<6, 2, 1>(out); run<6, 4, 1>(out);
  run<3, 2, 2>(out); run<6, 2, 2>(out); run<6, 4, 2>(out);
  run<0, 2, 3>(out); run<0, 4, 3>(out); run<0, 8, 3>(out);
  run<0, 2, 4>(out); run<0, 4, 4>(out); run<0, 8, 4>(out);
  run5<2>(out); run5<4>(out); run5<8>(out);
  run5<8, 1, true>(out); run5<8, 2, true>(out);   // (wall clock: per-wave s_memtime spans do not add up to the launch with several waves per SIMD)
  run6<2, 0>(out); run6<2, 4>(out); run6<2, 8>(out); run6<4, 0>(out); run6<8, 0>(out);
  return 0;
}
