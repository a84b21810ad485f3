// Stand-alone reproducer (r06; not product): do a wave's vector-memory loads complete in issue order ACROSS instruction classes?
// hipcc's wait insertion (and every counted wait in this repository) assumes ONE in-order queue behind vmcnt: "s_waitcnt vmcnt(N)" = everything but
// the youngest N loads has landed.  The four-tile skinny chunk loop mixes MUBUF loads (weights, group words: buffer_load, HBM-cold) with FLAT-global
// loads (x fragments: global_load_dwordx4, L2-hot); with its requests issued unconditionally hipcc waits vmcnt(15..12) in front of the chunk in hand
// and the FIRST weights requested (tiles 0 and 1 of four) come out wrong, varying from run to run; -amdgpu-waitcnt-forcezero cures it
// (profiles/r05_skinny_variants.txt).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/vmcnt_order_probe tools/vmcnt_order_probe.hip && tools/bin/vmcnt_order_probe
// Per round a wave poisons v[20:23], requests a COLD 1 KiB into them (class A), then a HOT 1 KiB into v[24:27] (class B), waits vmcnt(1) -- "the
// older of the two has landed" -- and copies v[20:23] out at once.  A poison (or anything but the cold line) there = the younger load completed
// first and the count let the wave through.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define POISON "v_mov_b32 v20, 0xdeadbeef\n\tv_mov_b32 v21, 0xdeadbeef\n\tv_mov_b32 v22, 0xdeadbeef\n\tv_mov_b32 v23, 0xdeadbeef\n\t"
#define LOAD_A_BUF "buffer_load_dwordx4 v[20:23], %[voff], %[rc], %[coff] offen\n\t"
#define LOAD_A_GLB "global_load_dwordx4 v[20:23], %[cptr], off\n\t"
#define LOAD_B_BUF "buffer_load_dwordx4 v[24:27], %[voff], %[rh], 0 offen\n\t"
#define LOAD_B_GLB "global_load_dwordx4 v[24:27], %[hptr], off\n\t"
#define LOAD_B_DMA "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[voff], %[rh], 0 offen lds\n\t"
#define TAIL "s_waitcnt vmcnt(1)\n\tv_mov_b32 %[o0], v20\n\tv_mov_b32 %[o1], v21\n\tv_mov_b32 %[o2], v22\n\tv_mov_b32 %[o3], v23\n\ts_waitcnt vmcnt(0)\n\t"
#define ROUND(LA, LB)                                                                                                                       \
  asm volatile(POISON LA LB TAIL : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3])                                   \
               : [voff] "v"(lane * 16u), [rc] "s"(rc), [coff] "s"(coff), [rh] "s"(rh), [cptr] "v"(cptr), [hptr] "v"(hptr), [lds] "s"(lds)    \
               : "memory", "m0", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27")

// MODE: 0 buf/buf, 1 buf/global, 2 global/buf, 3 global/global, 4 buf/LDS-DMA
template <int MODE>
__global__ __launch_bounds__(64) void probe(const u32x4* __restrict__ cold, const u32x4* __restrict__ hot, unsigned* __restrict__ bad, int rounds) {
  __shared__ __attribute__((aligned(16))) char smem[2048];
  const unsigned lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cold, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, 1024, 0x00020000);
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned nbad = 0;
  for (int r = 0; r < rounds; ++r) {
    const unsigned line = (unsigned)blockIdx.x * (unsigned)rounds + (unsigned)r;   // a 1 KiB of memory nobody has touched since the fill: HBM latency
    const unsigned coff = __builtin_amdgcn_readfirstlane(line * 1024u);
    const u32x4* cptr = (const u32x4*)((const char*)cold + coff) + lane;
    const u32x4* hptr = hot + lane;
    unsigned o[4];
    if constexpr (MODE == 0) ROUND(LOAD_A_BUF, LOAD_B_BUF);
    else if constexpr (MODE == 1) ROUND(LOAD_A_BUF, LOAD_B_GLB);
    else if constexpr (MODE == 2) ROUND(LOAD_A_GLB, LOAD_B_BUF);
    else if constexpr (MODE == 3) ROUND(LOAD_A_GLB, LOAD_B_GLB);
    else ROUND(LOAD_A_BUF, LOAD_B_DMA);
    const unsigned want = 0x40000000u | (line * 256u + lane * 4u);   // the fill: word w of the buffer holds 0x40000000 | w
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) ok = ok && o[i] == want + i;
    nbad += __builtin_popcountll(__ballot(!ok)) != 0;
  }
  if (lane == 0) bad[blockIdx.x] = nbad;
}

__global__ void fill(unsigned* p, size_t words) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0x40000000u | (unsigned)i;
}

template <int MODE>
static void run(const char* what, const u32x4* cold, const u32x4* hot, unsigned* bad, int wgs, int rounds) {
  (void)hipMemset(bad, 0, wgs * 4);
  hipLaunchKernelGGL((probe<MODE>), dim3(wgs), dim3(64), 0, 0, cold, hot, bad, rounds);
  std::vector<unsigned> h(wgs);
  (void)hipMemcpy(h.data(), bad, wgs * 4, hipMemcpyDeviceToHost);
  unsigned long long n = 0;
  for (auto v : h) n += v;
  printf("%-58s %8llu of %d rounds saw the older (cold) load NOT landed behind s_waitcnt vmcnt(1)\n", what, n, wgs * rounds);
}

int main() {
  const int wgs = 512, rounds = 1024;                   // 512 MiB of cold lines per mode, refilled (and flushed out of the caches by the fill) in between
  const size_t bytes = (size_t)wgs * rounds * 1024;
  unsigned *cold, *hot, *bad;
  (void)hipMalloc(&cold, bytes); (void)hipMalloc(&hot, 1024); (void)hipMalloc(&bad, wgs * 4);
  (void)hipMemset(hot, 1, 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, bytes / 4); run<0>("A = buffer_load (cold), B = buffer_load (hot)", (u32x4*)cold, (u32x4*)hot, bad, wgs, rounds);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, bytes / 4); run<1>("A = buffer_load (cold), B = global_load (hot)", (u32x4*)cold, (u32x4*)hot, bad, wgs, rounds);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, bytes / 4); run<2>("A = global_load (cold), B = buffer_load (hot)", (u32x4*)cold, (u32x4*)hot, bad, wgs, rounds);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, bytes / 4); run<3>("A = global_load (cold), B = global_load (hot)", (u32x4*)cold, (u32x4*)hot, bad, wgs, rounds);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, bytes / 4); run<4>("A = buffer_load (cold), B = buffer_load ... lds (hot)", (u32x4*)cold, (u32x4*)hot, bad, wgs, rounds);
  }
  return 0;
}
