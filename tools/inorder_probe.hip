// Exploration tool (not product): do a CU's vector-memory loads return in issue order ACROSS waves?  One workgroup per CU, two waves:
// wave 0 streams HBM-cold lines (its own region, nt), wave 1 streams L2-hot lines (64 KiB shared by everybody), U requests of 1 KiB in flight
// each.  Modes: hot alone, cold alone, both.  If the hot wave's time per request rises to the cold wave's when both run, hits wait for
// misses of OTHER waves (one in-order return queue per CU); if it stays, the order is per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/inorder_probe tools/inorder_probe.hip && tools/bin/inorder_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool DMA>
__global__ __launch_bounds__(128) void probe(const char* __restrict__ cold, const char* __restrict__ hot, unsigned long long* __restrict__ out, int mode, int nreq_cold, int nreq_hot) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool is_cold = wave == 0;
  if ((is_cold && !(mode & 1)) || (!is_cold && !(mode & 2))) return;
  const size_t region = (size_t)nreq_cold * 1024;
  __amdgpu_buffer_rsrc_t r = is_cold ? __builtin_amdgcn_make_buffer_rsrc((void*)(cold + (size_t)blockIdx.x * region), 0, (unsigned)region, 0x00020000)
                                     : __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, 65536u, 0x00020000);
  const int n = is_cold ? nreq_cold : nreq_hot;
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (U * 1024);
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (is_cold || !DMA) {
    u32x4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = u32x4{0, 0, 0, 0};
    for (int i = 0; i < n; i += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc += q[u];
        const unsigned so = is_cold ? (unsigned)(i + u) * 1024u : ((unsigned)(i + u) * 1024u + blockIdx.x * 2048u) & 65535u;
        q[u] = is_cold ? __builtin_amdgcn_raw_buffer_load_b128(r, voff, so, 2) : __builtin_amdgcn_raw_buffer_load_b128(r, voff, so, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += q[u];
  } else {
    for (int i = 0; i < n; ++i) {
      const unsigned so = ((unsigned)i * 1024u + blockIdx.x * 2048u) & 65535u;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds + (i % U) * 1024), "v"(voff), "s"(r), "s"(so) : "memory");
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) out[(blockIdx.x * 2 + wave) * 2] = t1 - t0;
  if (acc[0] == 0x12345678u && lane == 63) out[(blockIdx.x * 2 + wave) * 2 + 1] = acc[1] + smem[lane];
}

// mixed: every wave of the workgroup (NW of them) issues hot LDS-DMA pieces and cold VGPR loads in ONE queue: HPC hot pieces per cold load, at most
// UH hot pieces / UC cold loads outstanding (counted waits; a hot piece is "consumed" when vmcnt says it landed).  split: waves 0 .. NW/2-1 only cold,
// the others only hot, same totals per workgroup.
template <int NW, int HPC, int UC>
__global__ __launch_bounds__(NW * 64) void mix(const char* __restrict__ cold, const char* __restrict__ hot, unsigned long long* __restrict__ out, int split, int ncold) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t region = (size_t)ncold * 1024 * NW;
  __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(cold + (size_t)blockIdx.x * region), 0, (unsigned)region, 0x00020000);
  __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, 1u << 20, 0x00020000);
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (HPC * UC * 1024);
  u32x4 q[UC];
#pragma unroll
  for (int u = 0; u < UC; ++u) q[u] = u32x4{0, 0, 0, 0};
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const bool do_cold = !split || wave < NW / 2, do_hot = !split || wave >= NW / 2;
  const int rounds = split ? 2 * ncold : ncold;   // a specialised wave does the work of two
  unsigned hs = (unsigned)blockIdx.x * 4096u + wave * 512u * 1024u / NW;
  for (int i = 0; i < rounds; i += UC) {
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      if (do_hot) {
#pragma unroll
        for (int h = 0; h < HPC; ++h) {
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds + (u * HPC + h) * 1024), "v"(voff), "s"(rh), "s"(hs & 0xfffffu) : "memory");
          hs += 1024u;
        }
      }
      if (do_cold) {
        acc += q[u];
        q[u] = __builtin_amdgcn_raw_buffer_load_b128(rc, voff, (unsigned)((split ? wave * 2 * ncold : wave * ncold) + i + u) * 1024u, 2);
      }
    }
    if (!do_cold) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HPC * (UC - 1)) : "memory");
  }
#pragma unroll
  for (int u = 0; u < UC; ++u) acc += q[u];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) out[blockIdx.x * NW + wave] = t1 - t0;
  if (acc[0] == 0x12345678u && lane == 63) out[blockIdx.x] = acc[1] + smem[lane];
}

template <int NW, int HPC, int UC>
static void run_mix(const char* cold, const char* hot, unsigned long long* out, int ncold) {
  const int wgs = 256;
  std::vector<unsigned long long> h(wgs * NW);
  for (int split : {0, 1}) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute((const void*)mix<NW, HPC, UC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((mix<NW, HPC, UC>), dim3(wgs), dim3(NW * 64), NW * HPC * UC * 1024, 0, cold, hot, out, split, ncold);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipMemcpy(h.data(), out, wgs * NW * 8, hipMemcpyDeviceToHost);
    double t = 0, tmax = 0;
    for (auto v : h) { t += v / 100.0; tmax = std::max(tmax, v / 100.0); }
    t /= h.size();
    const double cold_kb = (double)NW * ncold, hot_kb = cold_kb * HPC;
    printf("%d waves, %d hot pieces per cold load, %2d cold loads in flight per wave, %s: launch %7.2f us, waves mean %7.2f max %7.2f us; per CU cold %4.0f KiB (%5.2f TB/s chip) hot %5.0f KiB (%6.1f GB/s per CU)\n", NW, HPC, UC,
           split ? "SPLIT (half the waves cold, half hot)" : "MIXED (every wave both)              ", ms * 1e3, t, tmax, cold_kb, cold_kb * 1.024 * wgs / tmax / 1e3, hot_kb, hot_kb * 1.024 / tmax);
  }
}

template <int U, bool DMA>
static void run(const char* cold, const char* hot, unsigned long long* out, int nc, int nh) {
  const int wgs = 256;
  std::vector<unsigned long long> h(wgs * 4);
  for (int mode : {2, 1, 3}) {
    (void)hipMemset(out, 0, wgs * 4 * 8);
    hipLaunchKernelGGL((probe<U, DMA>), dim3(wgs), dim3(128), 2 * U * 1024, 0, cold, hot, out, mode, nc, nh);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), out, wgs * 4 * 8, hipMemcpyDeviceToHost);
    double tc = 0, th = 0;
    for (int b = 0; b < wgs; ++b) { tc += h[(b * 2 + 0) * 2] / 100.0; th += h[(b * 2 + 1) * 2] / 100.0; }
    tc /= wgs; th /= wgs;
    printf("U = %2d %s  mode %s: cold wave %7.2f us for %d KiB (%6.1f GB/s per CU, %5.2f TB/s chip)   hot wave %7.2f us for %d KiB (%6.1f GB/s per CU)\n", U, DMA ? "hot by LDS-DMA" : "hot to VGPRs  ",
           mode == 1 ? "cold only" : mode == 2 ? "hot only " : "both     ", tc, nc, tc > 0 ? nc * 1.024 / tc : 0.0, tc > 0 ? nc * 1.024 / tc * wgs / 1e3 : 0.0, th, nh, th > 0 ? nh * 1.024 / th : 0.0);
  }
}

int main() {
  const int nc = 512, nh = 2048;   // per workgroup: 512 KiB cold, 2 MiB hot
  char *cold, *hot; unsigned long long* out;
  (void)hipMalloc(&cold, (size_t)256 * nc * 1024 * 2); (void)hipMemset(cold, 1, (size_t)256 * nc * 1024 * 2);
  (void)hipMalloc(&hot, 1 << 20); (void)hipMemset(hot, 1, 1 << 20);
  (void)hipMalloc(&out, 256 * 4 * 8);
  run<4, false>(cold, hot, out, nc, nh); run<8, false>(cold, hot, out, nc, nh); run<16, false>(cold, hot, out, nc, nh);
  run<4, true>(cold, hot, out, nc, nh); run<8, true>(cold, hot, out, nc, nh); run<16, true>(cold, hot, out, nc, nh);
  // the mid-token kernels' mix: per CU 192 KiB of weights (cold) and 512 KiB of x (hot): 8 waves x 24 cold loads, 8 hot pieces for 3 cold loads ~ 3 per cold
  run_mix<8, 3, 2>(cold, hot, out, 24); run_mix<8, 3, 4>(cold, hot, out, 24); run_mix<8, 3, 8>(cold, hot, out, 24);
  run_mix<8, 3, 2>(cold, hot, out, 96); run_mix<8, 3, 4>(cold, hot, out, 96); run_mix<8, 3, 8>(cold, hot, out, 96);
  run_mix<8, 1, 4>(cold, hot, out, 96); run_mix<8, 0, 4>(cold, hot, out, 96);
  return 0;
}
