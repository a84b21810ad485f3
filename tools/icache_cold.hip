// How fast does a wave run through straight-line code it has never executed (instruction fetch misses), against the
// same code the second time round?  hipcc --offload-arch=gfx950 -O3 -o icache_cold tools/icache_cold.hip && ./icache_cold
// Each launch: every wave runs the block of N 8-byte VALU instructions twice and stamps s_memrealtime (100 MHz) around both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int N>
__global__ void cold(unsigned long long* out, float* sink, int reps) {
  float v = threadIdx.x;
  unsigned long long t[3];
  for (int r = 0; r < 2; ++r) {
    t[r] = __builtin_amdgcn_s_memrealtime();
    asm volatile(".rept %1\n v_add_f32 %0, 0x3f800001, %0\n .endr" : "+v"(v) : "n"(N));
    asm volatile("s_nop 0" ::: "memory");
  }
  t[2] = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    out[w * 2] = t[1] - t[0];
    out[w * 2 + 1] = t[2] - t[1];
  }
  if (v == 12345.f) *sink = v;
}

template <int N>
void run(int grid, int block, char* flush, size_t flush_bytes) {
  unsigned long long* d;
  float* sink;
  hipMalloc(&d, 8 * 2 * 4096);
  hipMalloc(&sink, 4);
  const int waves = grid * block / 64;
  std::vector<unsigned long long> h(2 * waves);
  double c = 0, w = 0;
  const int L = 8;
  for (int i = 0; i < L + 2; ++i) {
    if (flush) hipMemsetAsync(flush, i, flush_bytes, 0);  // (push the code out of L2 / MALL between launches)
    hipLaunchKernelGGL(cold<N>, dim3(grid), dim3(block), 0, 0, d, sink, 1);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 16 * waves, hipMemcpyDeviceToHost);
    if (i < 2) continue;
    double cm = 0, wm = 0;
    for (int k = 0; k < waves; ++k) cm += h[2 * k], wm += h[2 * k + 1];
    c += cm / waves, w += wm / waves;
  }
  printf("N=%5d (%3d KiB) grid %3d x %d waves%s: first pass %7.2f us (%5.1f ns / instruction), second pass %6.2f us (%4.1f ns / instruction)\n", N, N * 8 / 1024,
         grid, block / 64, flush ? " flushed" : "        ", c / L / 100, c / L * 10 / N, w / L / 100, w / L * 10 / N);
  hipFree(d);
  hipFree(sink);
}

int main() {
  char* flush;
  const size_t fb = 1ull << 30;
  hipMalloc(&flush, fb);
  run<256>(1, 64, nullptr, 0);
  run<1024>(1, 64, nullptr, 0);
  run<2048>(1, 64, nullptr, 0);
  run<4096>(1, 64, nullptr, 0);
  run<2048>(1, 256, nullptr, 0);
  run<2048>(256, 256, nullptr, 0);
  run<2048>(1, 64, flush, fb);
  run<2048>(256, 256, flush, fb);
  run<6000>(256, 256, nullptr, 0);
  return 0;
}
