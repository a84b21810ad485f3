// Exploration tool (not part of the product): how long does a grid take to START, as a function of its shape?  Every wave stamps
// s_memrealtime at entry (and after one 16-byte load of a kernel-argument-dependent address, the earliest a weight request can go
// out); the host prints first -> last entry, and first entry -> last "argument known".
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ramp_probe tools/ramp_probe.hip && tools/bin/ramp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int VGPRS>
__global__ void k_ramp(const u32x4* __restrict__ w, unsigned long long* __restrict__ ts, unsigned* sink, int lds_touch) {
  extern __shared__ char smem[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const u32x4 v = w[(size_t)wave * 64 + (threadIdx.x & 63)];   // needs the kernel argument
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();  // request issued
  unsigned acc = v[0] ^ v[1] ^ v[2] ^ v[3];
  if (lds_touch) { ((unsigned*)smem)[threadIdx.x] = acc; __syncthreads(); acc ^= ((unsigned*)smem)[threadIdx.x ^ 1]; }
  // register pressure stand-in: VGPRS live values
  unsigned r[VGPRS];
#pragma unroll
  for (int i = 0; i < VGPRS; ++i) r[i] = acc * (i + 3);
#pragma unroll
  for (int i = 0; i < VGPRS; ++i) asm volatile("" : "+v"(r[i]));
#pragma unroll
  for (int i = 0; i < VGPRS; ++i) acc ^= r[i];
  const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();  // data back
  if (acc == 0x12345678u) sink[wave] = acc;
  if ((threadIdx.x & 63) == 0) { ts[3 * wave] = t0; ts[3 * wave + 1] = t1; ts[3 * wave + 2] = t2; }
}

template <int VGPRS>
static void run(const char* what, int grid, int block, int lds, const u32x4* w, unsigned long long* ts, unsigned* sink, size_t set_bytes, int nsets, int cold) {
  const int waves = grid * block / 64;
  std::vector<unsigned long long> h(3 * (size_t)waves);
  double ramp = 0, req = 0, back = 0; int n = 0;
  for (int i = 0; i < 24; ++i) {
    const u32x4* p = (const u32x4*)((const char*)w + (cold ? (size_t)(i % nsets) * set_bytes : 0));
    hipLaunchKernelGGL((k_ramp<VGPRS>), dim3(grid), dim3(block), lds, 0, p, ts, sink, lds > 0 ? 1 : 0);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, last0 = 0, last1 = 0, last2 = 0;
    for (int k = 0; k < waves; ++k) { lo = std::min(lo, h[3 * k]); last0 = std::max(last0, h[3 * k]); last1 = std::max(last1, h[3 * k + 1]); last2 = std::max(last2, h[3 * k + 2]); }
    if (i >= 4) { ramp += (last0 - lo) * 0.01; req += (last1 - lo) * 0.01; back += (last2 - lo) * 0.01; ++n; }
  }
  printf("%-10s grid=%5d block=%4d lds=%6d vgprs~%3d %s: last entry +%.2f us, last request +%.2f, last data +%.2f\n", what, grid, block, lds, VGPRS, cold ? "cold" : "hot ", ramp / n, req / n, back / n);
}

int main() {
  const size_t set_bytes = 8u << 20; const int nsets = 40;
  char* w; hipMalloc(&w, set_bytes * nsets); hipMemset(w, 1, set_bytes * nsets);
  unsigned long long* ts; hipMalloc(&ts, 3 * 8 * 65536);
  unsigned* sink; hipMalloc(&sink, 4 * 65536);
  for (int cold = 0; cold < 2; ++cold) {
    for (int block : {64, 128, 256, 512, 1024})
      for (int waves : {256, 512, 1024, 2048, 4096})
        if (waves * 64 / block >= 64) run<8>("plain", waves * 64 / block, block, 0, (const u32x4*)w, ts, sink, set_bytes, nsets, cold);
    run<8>("lds32k", 256, 256, 32768, (const u32x4*)w, ts, sink, set_bytes, nsets, cold);
    run<8>("lds32k", 256, 512, 32768, (const u32x4*)w, ts, sink, set_bytes, nsets, cold);
    run<8>("lds32k", 1024, 256, 32768, (const u32x4*)w, ts, sink, set_bytes, nsets, cold);
    run<96>("regs", 256, 256, 0, (const u32x4*)w, ts, sink, set_bytes, nsets, cold);
    run<96>("regs", 256, 512, 0, (const u32x4*)w, ts, sink, set_bytes, nsets, cold);
    run<96>("regs", 1024, 256, 0, (const u32x4*)w, ts, sink, set_bytes, nsets, cold);
  }
  return 0;
}
