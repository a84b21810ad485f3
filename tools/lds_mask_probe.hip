// Exploration tool (not part of the product): what does a ds_read_b128 cost when only a few lanes are active?
// The small-M kernels read A fragments of 16 token rows of which only M are real; if the LDS pipe skips inactive lanes,
// an exec-masked read is the cheap way to build the fragment.  Prints LDS-pipe clocks per wave-instruction for several
// exec masks and broadcast patterns, 1 / 4 / 8 / 16 waves per CU all reading at once.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_mask_probe tools/lds_mask_probe.hip && tools/bin/lds_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: full exec, lane-linear addresses; 1: full exec, M=1 fragment pattern (4 distinct addresses, broadcast over n16);
// 2: exec = one lane per row of 16 (M = 1); 3: exec = four lanes per row (M = 4); 4: exec = lanes 0..3 only; 5: ds_read_b64 full exec fragment pattern
template <int MODE>
__global__ void k_lds(unsigned long long* out, int iters) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = i;
  __syncthreads();
  unsigned addr = MODE == 0 ? lane * 16 : (lane >> 4) * 16;
  addr += (threadIdx.x >> 6) * 1024;
  unsigned long long mask = ~0ull;
  if (MODE == 2) mask = 0x0001000100010001ull;
  if (MODE == 3) mask = 0x000f000f000f000full;
  if (MODE == 4) mask = 0xfull;
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
    if (MODE == 5) {
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:64\n ds_read_b64 %2, %8 offset:128\n ds_read_b64 %3, %8 offset:192\n"
                   "ds_read_b64 %4, %8 offset:256\n ds_read_b64 %5, %8 offset:320\n ds_read_b64 %6, %8 offset:384\n ds_read_b64 %7, %8 offset:448\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(*(unsigned long long*)&v0), "=&v"(*(unsigned long long*)&v1), "=&v"(*(unsigned long long*)&v2), "=&v"(*(unsigned long long*)&v3),
                     "=&v"(*(unsigned long long*)&v4), "=&v"(*(unsigned long long*)&v5), "=&v"(*(unsigned long long*)&v6), "=&v"(*(unsigned long long*)&v7)
                   : "v"(addr));
    } else {
      asm volatile("s_mov_b64 exec, %9\n ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:64\n ds_read_b128 %2, %8 offset:128\n ds_read_b128 %3, %8 offset:192\n"
                   "ds_read_b128 %4, %8 offset:256\n ds_read_b128 %5, %8 offset:320\n ds_read_b128 %6, %8 offset:384\n ds_read_b128 %7, %8 offset:448\n s_mov_b64 exec, -1\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                   : "v"(addr), "s"(mask));
    }
    acc[0] ^= v0[0] ^ v1[0] ^ v2[0] ^ v3[0] ^ v4[0] ^ v5[0] ^ v6[0] ^ v7[0];
    addr ^= (acc[0] & 0x10);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0 + (acc[0] == 0x12345678u);
}

template <int MODE>
static void run(const char* what, unsigned long long* dev) {
  const int iters = 2000;
  for (int waves : {1, 4, 8, 16}) {
    hipLaunchKernelGGL((k_lds<MODE>), dim3(256), dim3(waves * 64), 65536, 0, dev, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * waves);
    hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    const double per_wave_instr = s / h.size() / iters / 8;              // clocks per instruction as one wave sees them
    printf("%-58s waves/CU=%2d: %6.1f clk per instr per wave = %5.1f clk of the CU's LDS pipe per instr\n", what, waves, per_wave_instr, per_wave_instr / waves);
  }
}

int main() {
  unsigned long long* dev; hipMalloc(&dev, 8 * 256 * 16);
  hipFuncSetAttribute((const void*)k_lds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  run<0>("b128 full exec, lane-linear", dev);
  run<1>("b128 full exec, 4 addresses broadcast (M=1 fragment)", dev);
  run<2>("b128 exec = 1 lane per row of 16 (M=1)", dev);
  run<3>("b128 exec = 4 lanes per row of 16 (M=4)", dev);
  run<4>("b128 exec = lanes 0..3", dev);
  run<5>("b64 full exec, 4 addresses broadcast", dev);
  return 0;
}
