// Dispatch-duration floor vs launch shape: empty kernels timed with an event pair bound to the dispatch (the clock
// bench.py uses) and back to back in a stream (what a dependent chain pays per launch).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_floor tools/dispatch_floor.hip && /tmp/dispatch_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void empty_kernel(unsigned* sink) {
  if (sink != nullptr && threadIdx.x == 0xffffffffu) sink[0] = 1;
}
int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  const int grids[] = {1, 8, 32, 256, 512, 2048}, blocks[] = {64, 256, 512};
  for (int b : blocks)
    for (int g : grids) {
      std::vector<float> us;
      for (int i = 0; i < 60; ++i) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipExtLaunchKernelGGL(empty_kernel, dim3(g), dim3(b), 0, st, e0, e1, 0, (unsigned*)nullptr);
        hipStreamSynchronize(st);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        us.push_back(ms * 1e3f);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
      }
      std::sort(us.begin() + 5, us.end());
      const int n = 2000;
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(b), 0, st, (unsigned*)nullptr);
      hipStreamSynchronize(st);
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(b), 0, st, (unsigned*)nullptr);
      hipStreamSynchronize(st);
      const double chain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
      printf("grid %5d x %3d threads: event pair median %.2f us (min %.2f), back-to-back stream launches %.2f us each\n", g, b,
             us[5 + (us.size() - 5) / 2], us[5], chain);
    }
  return 0;
}
