// Exploration tool (not product): what plain vector loads (buffer_load_dwordx4 -> VGPR, the way the small-M GEMM kernels fetch weights) reach of
// HBM on a 1 GiB HBM-cold stream, as a function of workgroups per CU, waves per workgroup and 1 KiB requests in flight per wave; nt policy on / off.
// Context for the rooflines of DESIGN.md: the "peak" is 8 TB/s; this is the rate a kernel that does nothing else sustains.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/hbm_stream tools/hbm_stream.hip && tools/bin/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U, int AUX>
__global__ __launch_bounds__(1024) void k_stream(const u32x4* __restrict__ buf, unsigned* __restrict__ sink, unsigned long long total_bytes) {
  const int lane = threadIdx.x & 63;
  const unsigned wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
  const unsigned long long chunks = total_bytes / (1024ull * U);   // a chunk = U consecutive 1 KiB tiles of one wave
  unsigned acc = 0;
  for (unsigned long long c = wave_g; c < chunks; c += nwaves) {
    const u32x4* p = buf + c * (64ull * U) + lane;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = AUX ? __builtin_nontemporal_load(p + 64 * u) : p[64 * u];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int U, int AUX>
static float run(const u32x4* buf, unsigned* sink, unsigned long long bytes, int grid, int block) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_stream<U, AUX>), dim3(grid), dim3(block), 0, 0, buf, sink, bytes);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (r && ms < best) best = ms;
  }
  return best;
}

int main() {
  const unsigned long long bytes = 1ull << 30;
  u32x4* buf; unsigned* sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 64);
  hipMemset(buf, 1, bytes);
  printf("1 GiB HBM-cold stream, plain 16-byte-per-lane loads; TB/s (best of 4 timed launches)\n");
  printf("%-34s %8s %8s %8s %8s\n", "grid x block (waves / CU)", "U=1", "U=2", "U=4", "U=8");
  const int grids[] = {256, 512, 1024, 2048}, blocks[] = {256, 512, 1024};
  for (int aux = 0; aux < 2; ++aux)
    for (int g : grids)
      for (int bl : blocks) {
        if ((long)g * bl / 64 / 256 > 32) continue;
        float t[4];
        if (aux) { t[0] = run<1, 1>(buf, sink, bytes, g, bl); t[1] = run<2, 1>(buf, sink, bytes, g, bl); t[2] = run<4, 1>(buf, sink, bytes, g, bl); t[3] = run<8, 1>(buf, sink, bytes, g, bl); }
        else { t[0] = run<1, 0>(buf, sink, bytes, g, bl); t[1] = run<2, 0>(buf, sink, bytes, g, bl); t[2] = run<4, 0>(buf, sink, bytes, g, bl); t[3] = run<8, 0>(buf, sink, bytes, g, bl); }
        printf("%s %4d x %4d (%2ld waves / CU)        %8.2f %8.2f %8.2f %8.2f\n", aux ? "nt   " : "plain", g, bl, (long)g * bl / 64 / 256, bytes / t[0] * 1e-9, bytes / t[1] * 1e-9,
               bytes / t[2] * 1e-9, bytes / t[3] * 1e-9);
      }
  return 0;
}
