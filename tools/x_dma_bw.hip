// Exploration tool (not product): per-CU throughput of the mid-token kernels' x traffic -- 64 token rows x K fp16, every workgroup
// reads ALL of it (eight waves, each its own k range) by LDS-DMA -- as a function of the piece shape (bytes per row and instruction),
// the requests in flight per wave and whether the workgroups share one x or read private copies.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/x_dma_bw tools/x_dma_bw.hip && /tmp/x_dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// SEG: bytes of one row covered by one instruction (64, 128, 256, 1024); rows per instruction = 1024 / SEG; U = instructions in flight per wave
template <int SEG, int U>
__global__ __launch_bounds__(512) void k_x(const char* __restrict__ x, unsigned* __restrict__ out, int K, int reps, size_t copy_stride, unsigned pad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int RPI = 1024 / SEG;             // rows per instruction
  constexpr int LPR = SEG / 16;               // lanes per row
  const unsigned rowbytes = (unsigned)K * 2u + pad;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)blockIdx.x * copy_stride), 0, 64u * rowbytes, 0x00020000);
  const unsigned voff = (unsigned)(lane / LPR) * rowbytes + (unsigned)(lane % LPR) * 16u;
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (U * 1024);
  const unsigned kbytes = (unsigned)K * 2u / 8u;      // this wave's k range in bytes of a row
  const unsigned k0 = (unsigned)wave * kbytes;
  int u = 0;
  for (int rep = 0; rep < reps; ++rep)
    for (unsigned kb = 0; kb < kbytes; kb += SEG)
      for (int rb = 0; rb < 64; rb += RPI) {
        const unsigned so = (unsigned)rb * rowbytes + k0 + kb;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds + (u % U) * 1024), "v"(voff), "s"(r), "s"(so) : "memory");
        ++u;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U - 1) : "memory");
      }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (smem[lane] == 77 && smem[lane + 64] == 78) out[blockIdx.x] = 1;
}

template <int SEG, int U>
static void run(const char* x, unsigned* out, int K, int wgs, bool priv, unsigned pad = 0) {
  const int reps = 16;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned lds = 8 * U * 1024;
  hipFuncSetAttribute((const void*)k_x<SEG, U>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t stride = priv ? (size_t)64 * (K * 2 + pad) : 0;
  hipLaunchKernelGGL((k_x<SEG, U>), dim3(wgs), dim3(512), lds, 0, x, out, K, 2, stride, pad);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_x<SEG, U>), dim3(wgs), dim3(512), lds, 0, x, out, K, reps, stride, pad);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * 64.0 * K * 2.0 * reps;
  printf("pitch +%3u B: row piece %4d B x %2d rows, %2d in flight/wave, %3d workgroups, %s x (K = %d): %6.2f TB/s = %6.1f GB/s per workgroup; one pass of x %6.2f us\n", pad, SEG,
         1024 / SEG, U, wgs, priv ? "private" : "shared ", K, bytes / ms / 1e9, bytes / ms / 1e6 / wgs, ms * 1e3 / reps);
}

int main() {
  const int K = 4096;
  const size_t total = (size_t)256 * 64 * (K * 2 + 512);
  char* x; hipMalloc(&x, total); hipMemset(x, 1, total);
  unsigned* out; hipMalloc(&out, 4096);
  for (int wgs : {128, 256})
    for (int priv = 0; priv < 2; ++priv) {
      run<64, 4>(x, out, K, wgs, priv); run<64, 12>(x, out, K, wgs, priv);
      run<128, 4>(x, out, K, wgs, priv); run<128, 12>(x, out, K, wgs, priv);
      run<256, 4>(x, out, K, wgs, priv); run<256, 12>(x, out, K, wgs, priv);
      run<1024, 4>(x, out, K, wgs, priv); run<1024, 12>(x, out, K, wgs, priv); run<1024, 16>(x, out, K, wgs, priv);
    }
  // row pitch: 2 K bytes puts the 8 rows of a piece 8 KiB apart (one L2 channel if channels interleave below that); + 128 / 256 / 384 B spreads them
  for (unsigned pad : {128u, 256u, 384u})
    for (int priv = 0; priv < 2; ++priv) {
      run<128, 4>(x, out, K, 256, priv, pad); run<128, 12>(x, out, K, 256, priv, pad); run<256, 12>(x, out, K, 256, priv, pad);
    }
  return 0;
}
