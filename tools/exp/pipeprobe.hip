// Stand-alone experiment: which HIP streams share a dispatch pipe?  Stream i gets a kernel of 40 000 short workgroups (its
// dispatch keeps the pipe's dispatcher busy for ~1 ms), stream j a one-workgroup kernel right behind it: the time until the
// small kernel has finished tells whether it had to wait for the big grid's dispatch (same queue / same pipe) or slipped in.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/pipeprobe tools/exp/pipeprobe.hip && tools/exp/pipeprobe [n_streams]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void busy(long long cycles, int* sink)
{
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (sink && threadIdx.x == 1000) *sink = 1;
}
int main(int argc, char** argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 12;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;           // 0: default priority; 1: priorities cycling normal / high / low
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
  int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
  std::vector<hipStream_t> st(n);
  for (int i = 0; i < n; i++) {
    if (mode == 1) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, i % 3 == 0 ? 0 : (i % 3 == 1 ? hi : lo));
    else hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < n; i++) hipLaunchKernelGGL(busy, dim3(8), dim3(64), 0, st[i], 100, nullptr);
  hipDeviceSynchronize();
  printf("rows: stream with the big grid; columns: stream with the small kernel; us until the small kernel has finished\n     ");
  for (int j = 0; j < n; j++) printf("%6d", j);
  printf("\n");
  for (int i = 0; i < n; i++) {
    printf("%3d: ", i);
    for (int j = 0; j < n; j++) {
      if (i == j) { printf("     -"); continue; }
      hipDeviceSynchronize();
      hipLaunchKernelGGL(busy, dim3(40000), dim3(256), 0, st[i], (long long)rate / 50, nullptr);       // 20 us per workgroup
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, st[j], 100, nullptr);
      hipStreamSynchronize(st[j]);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%6.0f", us);
    }
    printf("\n");
  }
  return 0;
}
