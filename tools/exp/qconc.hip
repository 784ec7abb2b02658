// Stand-alone experiment: how many HIP streams of one process really run concurrently on this device/runtime.
// N streams each get one ~2 ms kernel of 8 small workgroups (the GPU has room for all of them at once); wall time
// over N tells how many hardware queues the runtime maps the streams to.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/qconc tools/exp/qconc.hip && tools/exp/qconc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int* sink)
{
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (sink && threadIdx.x == 1000) *sink = 1;
}
int main()
{
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
  const long long cyc = (long long)rate * 2;                                        // 2 ms
  for (int prio = 0; prio < 2; prio++)
    for (int n : { 1, 2, 3, 4, 5, 6, 8, 12 }) {
      std::vector<hipStream_t> st(n);
      int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
      for (int i = 0; i < n; i++) {
        if (prio) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, i % 3 == 0 ? (lo + hi) / 2 : (i % 3 == 1 ? hi : lo));
        else hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
      }
      for (int i = 0; i < n; i++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st[i], 1000, nullptr);   // warm
      hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < n; i++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st[i], cyc, nullptr);
      hipDeviceSynchronize();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("%s %2d streams: %.2f ms  (~%.1f kernels deep)\n", prio ? "priority classes" : "default priority ", n, ms, ms / 2.0);
      for (auto s : st) hipStreamDestroy(s);
    }
  return 0;
}
