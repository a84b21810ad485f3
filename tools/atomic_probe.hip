// Exploration tool (not part of the product): what do fire-and-forget fp32 atomic adds on a handful of addresses cost a launch?
// The deferred-RMSNorm epilogue lets every workgroup of a GEMM add its rows' partial sums of squares to ssq[row]:
// 64 rows x (64 .. 1024) adders per row.  Prints the launch time (event pair over a graph of 200 launches) of 256 workgroups that each do
// PER atomics on each of ROWS addresses, with the addresses STRIDE floats apart, next to the same kernel without the atomics.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/atomic_probe tools/atomic_probe.hip && tools/bin/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_atomic(float* ssq, int rows, int stride, int per, float* sink) {
  // a little work in front so that the workgroups do not all arrive in the same clock
  float v = (float)threadIdx.x;
  for (int i = 0; i < 64; ++i) v = v * 1.0001f + 0.5f;
  if (per > 0 && (int)threadIdx.x < rows) {
    for (int p = 0; p < per; ++p) unsafeAtomicAdd(ssq + (size_t)threadIdx.x * stride, v);
  }
  if (v == 12345.f) sink[0] = v;
}

static float time_it(float* ssq, int grid, int rows, int stride, int per, float* sink) {
  hipStream_t st;
  hipStreamCreate(&st);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(256), 0, st, ssq, rows, stride, per, sink);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, st);
  hipGraphLaunch(ge, st);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  hipStreamDestroy(st);
  return ms * 1000.f / 200.f;
}

int main() {
  float *ssq, *sink;
  hipMalloc(&ssq, 64 * 4096 * 4);
  hipMalloc(&sink, 64);
  hipMemset(ssq, 0, 64 * 4096 * 4);
  for (int grid : {256, 1024}) {
    const float base = time_it(ssq, grid, 64, 1, 0, sink);
    printf("grid %4d, no atomics: %.2f us per launch\n", grid, base);
    for (int stride : {1, 16, 64, 1024})
      for (int per : {1, 4})
        printf("grid %4d  rows 64  stride %4d floats  %d atomics per row and workgroup (%5d per address): %.2f us per launch (+%.2f)\n", grid, stride, per, grid * per,
               time_it(ssq, grid, 64, stride, per, sink), time_it(ssq, grid, 64, stride, per, sink) - base);
  }
  return 0;
}
