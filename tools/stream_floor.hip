// Exploration tool (not part of the product): what is the floor, on MI355X, for a kernel that streams
// the 8.4 MB of a 4096x4096 int4 weight once?  Prints per-dispatch durations (event pair bound to the
// dispatch) for an empty kernel and for pure-streaming kernels of several shapes, HBM-cold and cache-hot.
//   hipcc --offload-arch=gfx950 -O3 -o stream_floor tools/stream_floor.hip && ./stream_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_empty(unsigned* out) { if (out == nullptr && threadIdx.x == 12345) out[0] = 1; }

// each wave loads L x 16 B per lane from consecutive 1 KiB blocks, XOR-reduces, one store per wave
template <int L>
__global__ void k_stream(const u32x4* __restrict__ w, unsigned* __restrict__ out, int waves_total) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const u32x4* p = w + (size_t)wave * L * 64 + lane;
  u32x4 v[L];
#pragma unroll
  for (int i = 0; i < L; ++i) v[i] = p[i * 64];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
  if (acc == 0x12345678u) out[wave] = acc;   // practically never: keeps the loads alive without store traffic
}

// same, but with a workgroup-level LDS reduction + barrier + one 8-byte store per 64 lanes of wave 0 (GEMV epilogue shape)
template <int L>
__global__ void k_stream_red(const u32x4* __restrict__ w, unsigned* __restrict__ out) {
  __shared__ unsigned red[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
  const u32x4* p = w + (size_t)wave * L * 64 + lane;
  u32x4 v[L];
#pragma unroll
  for (int i = 0; i < L; ++i) v[i] = p[i * 64];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
  red[wv][lane] = acc;
  __syncthreads();
  if (wv == 0) {
    unsigned s = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s ^= red[i][lane];
    out[blockIdx.x * 64 + lane] = s;
  }
}

// streaming kernel that also records each wave's first/last wall-clock tick (s_memrealtime)
template <int L>
__global__ void k_stream_ts(const u32x4* __restrict__ w, unsigned* __restrict__ out, unsigned long long* __restrict__ ts) {
  const unsigned long long t0 = wall_clock64();
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const u32x4* p = w + (size_t)wave * L * 64 + lane;
  u32x4 v[L];
#pragma unroll
  for (int i = 0; i < L; ++i) v[i] = p[i * 64];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
  if (acc == 0x12345678u) out[wave] = acc;
  const unsigned long long t1 = wall_clock64();
  if (lane == 0) { ts[2 * wave] = t0; ts[2 * wave + 1] = t1 + (acc == 0x12345678u); }
}

template <typename F>
static void timeit(const char* name, F launch, int iters = 60) {
  std::vector<hipEvent_t> ev(2 * iters);
  for (auto& e : ev) hipEventCreate(&e);
  for (int i = 0; i < iters; ++i) launch(i, ev[2 * i], ev[2 * i + 1]);
  hipDeviceSynchronize();
  std::vector<float> us;
  for (int i = 5; i < iters; ++i) { float ms; hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]); us.push_back(ms * 1000.f); }
  std::sort(us.begin(), us.end());
  float mean = 0; for (float x : us) mean += x; mean /= us.size();
  printf("%-44s mean %7.2f us  median %7.2f  min %7.2f\n", name, mean, us[us.size() / 2], us[0]);
  for (auto& e : ev) hipEventDestroy(e);
}

int main() {
  const size_t bytes = 8388608;  // 4096 x 4096 int4
  const int nsets = 40;
  char* w; hipMalloc(&w, bytes * nsets); hipMemset(w, 1, bytes * nsets);
  unsigned* out; hipMalloc(&out, 1 << 22);
  hipStream_t st; hipStreamCreate(&st);
  for (int wg : {64, 256, 512, 1024})
    for (int g : {256, 1024}) {
      char nm[64]; snprintf(nm, 64, "empty grid=%d block=%d", g, wg);
      timeit(nm, [&](int, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(g), dim3(wg), 0, st, a, b, 0, out); });
    }
#define RUN(L, BLOCK, RED)                                                                                          \
  for (int hot = 0; hot < 2; ++hot) {                                                                               \
    const int waves = (int)(bytes / (1024 * L));                                                                    \
    const int grid = waves / (BLOCK / 64);                                                                          \
    char nm[96]; snprintf(nm, 96, "%s L=%d block=%d grid=%d %s", RED ? "stream+red" : "stream", L, BLOCK, grid, hot ? "hot" : "cold"); \
    timeit(nm, [&](int i, hipEvent_t a, hipEvent_t b) {                                                             \
      const u32x4* p = (const u32x4*)(w + (hot ? 0 : (size_t)(i % nsets) * bytes));                                 \
      if (RED) hipExtLaunchKernelGGL((k_stream_red<L>), dim3(grid), dim3(BLOCK), 0, st, a, b, 0, p, out);           \
      else hipExtLaunchKernelGGL((k_stream<L>), dim3(grid), dim3(BLOCK), 0, st, a, b, 0, p, out, waves);            \
    });                                                                                                             \
  }
  RUN(1, 256, 0) RUN(2, 256, 0) RUN(4, 256, 0) RUN(8, 256, 0) RUN(16, 256, 0)
  RUN(2, 512, 0) RUN(4, 512, 0) RUN(8, 512, 0) RUN(4, 1024, 0) RUN(2, 1024, 0)
  RUN(4, 512, 1) RUN(2, 1024, 1) RUN(4, 256, 1) RUN(8, 256, 1)
  // in-kernel wall-clock span of the streaming kernel (first wave start -> last wave end)
  {
    int rate_khz = 0; hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const int L = 4, BLOCK = 512, waves = (int)(bytes / (1024 * L)), grid = waves / (BLOCK / 64);
    unsigned long long* ts; hipMalloc(&ts, waves * 16);
    std::vector<unsigned long long> h(2 * waves);
    for (int hot = 0; hot < 2; ++hot) {
      double acc = 0; int n = 0;
      for (int i = 0; i < 30; ++i) {
        const u32x4* p = (const u32x4*)(w + (hot ? 0 : (size_t)(i % nsets) * bytes));
        hipLaunchKernelGGL((k_stream_ts<L>), dim3(grid), dim3(BLOCK), 0, st, p, out, ts);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), ts, waves * 16, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0, latest_start = 0;
        for (int k = 0; k < waves; ++k) { lo = std::min(lo, h[2 * k]); hi = std::max(hi, h[2 * k + 1]); latest_start = std::max(latest_start, h[2 * k]); }
        if (i >= 5) { acc += (double)(hi - lo) / rate_khz * 1000.0; ++n; if (i == 5) printf("   (last wave started %.2f us after the first)\n", (double)(latest_start - lo) / rate_khz * 1000.0); }
      }
      printf("in-kernel span stream L=4 block=512 %s: %.2f us (wall clock %d kHz)\n", hot ? "hot" : "cold", acc / n, rate_khz);
    }
  }
  return 0;
}
