// Exploration tool (not product): per-CU throughput of LDS-DMA (buffer_load_dwordx4 ... lds) and of plain
// buffer_load_dwordx4 -> VGPR as a function of waves per CU, from an L2-resident window shared by the workgroups of an XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U>  // MODE 0: LDS-DMA, 1: plain loads to VGPRs; U = instructions in flight per wave
__global__ __launch_bounds__(1024) void k_read(const u32x4* __restrict__ buf, unsigned* __restrict__ out, unsigned window_bytes,
                                              int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + (size_t)(blockIdx.x % 8) * (window_bytes / 16)), 0, window_bytes, 0x00020000);
  unsigned off = (((blockIdx.x / 8) * nw + wave) * U) * 1024u;  // bytes; wave-uniform
  const unsigned voff = lane * 16;
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (U * 1024);
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned so = (off + u * 1024u) % window_bytes;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds + u * 1024), "v"(voff), "s"(r), "s"(so) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U / 2) : "memory");
    } else {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, (off + u * 1024u) % window_bytes, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][3];
    }
    off += 32u * nw * U * 1024u;
  }
  if (acc == 0x12345u) out[blockIdx.x] = acc;
}

template <int MODE, int U>
static void run(const u32x4* buf, unsigned* out, unsigned window_bytes, int waves) {
  const int iters = 400;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned lds = waves * U * 1024;
  hipFuncSetAttribute((const void*)k_read<MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_read<MODE, U>), dim3(256), dim3(64 * waves), lds, 0, buf, out, window_bytes, 20);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_read<MODE, U>), dim3(256), dim3(64 * waves), lds, 0, buf, out, window_bytes, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = 256.0 * waves * 1024.0 * U * iters;
  printf("%s window/XCD %5.2f MB waves/CU %2d in flight/wave %2d : %6.2f TB/s = %6.1f GB/s/CU\n", MODE ? "plain  " : "LDS-DMA",
         window_bytes / 1048576.0, waves, U, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
  const size_t total = (size_t)8 * 8 << 20;
  u32x4* buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
  unsigned* out; hipMalloc(&out, 4096);
  for (unsigned w : {512u << 10, 2u << 20})
    for (int waves : {4, 8, 16}) {
      run<0, 4>(buf, out, w, waves); run<0, 8>(buf, out, w, waves);
      run<1, 4>(buf, out, w, waves); run<1, 8>(buf, out, w, waves);
    }
  return 0;
}
