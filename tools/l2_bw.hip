// Exploration tool (not product): sustained load bandwidth per CU as a function of the working set (L2-resident,
// MALL-resident, HBM) and of the number of 16-byte loads each wave keeps in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// each workgroup streams `per_wg` bytes (wrapping inside its XCD-local window of `window` bytes), U loads in flight per wave
template <int U>
__global__ __launch_bounds__(1024) void k_read(const u32x4* __restrict__ buf, unsigned* __restrict__ out, size_t window_vec,
                                              int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // blocks with the same blockIdx % 8 share an XCD: give each XCD its own window
  const size_t base = (size_t)(blockIdx.x % 8) * window_vec;
  size_t off = ((size_t)(blockIdx.x / 8) * (blockDim.x >> 6) + wave) * 64 * U;  // in 16-byte units
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = buf[base + (off + u * 64 + lane) % window_vec];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][3];
    off += 32 * (blockDim.x >> 6) * 64 * U;  // all 32 workgroups of the XCD x their waves advance together
  }
  if (acc == 0x12345u) out[blockIdx.x] = acc;
}

template <int U>
static void run(const u32x4* buf, unsigned* out, size_t window_bytes, int waves_per_wg) {
  const int iters = 400;
  const size_t window_vec = window_bytes / 16;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_read<U>, dim3(256), dim3(64 * waves_per_wg), 0, 0, buf, out, window_vec, 20);
  hipEventRecord(a);
  hipLaunchKernelGGL(k_read<U>, dim3(256), dim3(64 * waves_per_wg), 0, 0, buf, out, window_vec, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = 256.0 * waves_per_wg * 64 * 16 * U * iters;
  printf("window/XCD %7.2f MB  waves/CU %d  loads in flight/wave %2d : %7.2f TB/s  = %6.1f GB/s/CU = %5.1f B/clk/CU@2.4GHz\n",
         window_bytes / 1048576.0, waves_per_wg, U, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4);
}

int main() {
  const size_t total = (size_t)8 * 64 << 20;  // 8 windows of up to 64 MB
  u32x4* buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
  unsigned* out; hipMalloc(&out, 4096);
  for (size_t w : {(size_t)512 << 10, (size_t)2 << 20, (size_t)8 << 20, (size_t)64 << 20})
    for (int waves : {4, 8, 16}) {
      run<1>(buf, out, w, waves); run<2>(buf, out, w, waves); run<4>(buf, out, w, waves); run<8>(buf, out, w, waves); run<16>(buf, out, w, waves);
    }
  return 0;
}
