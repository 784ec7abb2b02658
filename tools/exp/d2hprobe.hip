// D2H probe: how fast do decoded pictures leave the device?  hipMemcpyAsync (SDMA or blit, as the runtime decides) on 1/2/4
// streams against a copy kernel that stores into pinned host memory.  Build: hipcc --offload-arch=gfx950 -O2 d2hprobe.hip -o d2hprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#include <thread>
#include <atomic>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    uint4 v = src[i];
    dst[i] = v;
  }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
  const size_t bytes = 24883200;                       // a 4K10 4:2:0 picture
  const int NB = 8, REP = 40;
  std::vector<void*> d(NB), h(NB);
  for (int i = 0; i < NB; i++) { CK(hipMalloc(&d[i], bytes)); CK(hipMemset(d[i], i + 1, bytes)); CK(hipHostMalloc(&h[i], bytes, hipHostMallocDefault)); }
  hipStream_t st[4]; for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  for (int ns : {1, 2, 4}) {
    for (int w = 0; w < 2; w++) {
      const double t0 = now();
      for (int r = 0; r < REP; r++) CK(hipMemcpyAsync(h[r % NB], d[r % NB], bytes, hipMemcpyDeviceToHost, st[r % ns]));
      CK(hipDeviceSynchronize());
      const double dt = now() - t0;
      if (w) printf("hipMemcpyAsync D2H, %d stream(s): %.1f GB/s (%.0f pictures/s)\n", ns, REP * bytes / dt / 1e9, REP / dt);
    }
  }
  for (int grid : {16, 32, 64, 128, 256, 512, 2048}) {
    for (int ns : {1, 2}) {
      for (int w = 0; w < 2; w++) {
        const double t0 = now();
        for (int r = 0; r < REP; r++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st[r % ns], (const uint4*)d[r % NB], (uint4*)h[r % NB], bytes / 16);
        CK(hipDeviceSynchronize());
        const double dt = now() - t0;
        if (w) printf("copy kernel grid %4d, %d stream(s): %.1f GB/s (%.0f pictures/s)\n", grid, ns, REP * bytes / dt / 1e9, REP / dt);
      }
    }
  }
  // both directions at once (uploads of the next pictures' records run next to the copy-out): H2D on stream 3
  {
    const double t0 = now();
    for (int r = 0; r < REP; r++) {
      CK(hipMemcpyAsync(h[r % 4], d[r % 4], bytes, hipMemcpyDeviceToHost, st[r % 2]));
      CK(hipMemcpyAsync(d[4 + r % 4], h[4 + r % 4], bytes / 4, hipMemcpyHostToDevice, st[3]));
    }
    CK(hipDeviceSynchronize());
    const double dt = now() - t0;
    printf("D2H on 2 streams + H2D of a quarter as much: D2H %.1f GB/s\n", REP * bytes / dt / 1e9);
  }
  // mixed traffic, as in the product with copy-out: per picture 24.9 MB leave and ~8 MB of records arrive
  for (int mode = 0; mode < 4; mode++) {          // bit 0: D2H by the copy kernel (grid 16) instead of hipMemcpyAsync; bit 1: H2D by a kernel reading pinned memory
    const size_t up = 8u << 20;
    for (int w = 0; w < 2; w++) {
      const double t0 = now();
      for (int r = 0; r < REP; r++) {
        if (mode & 1) hipLaunchKernelGGL(k_copy, dim3(16), dim3(256), 0, st[r % 2], (const uint4*)d[r % 4], (uint4*)h[r % 4], bytes / 16);
        else CK(hipMemcpyAsync(h[r % 4], d[r % 4], bytes, hipMemcpyDeviceToHost, st[r % 2]));
        if (mode & 2) hipLaunchKernelGGL(k_copy, dim3(16), dim3(256), 0, st[3], (const uint4*)h[4 + r % 4], (uint4*)d[4 + r % 4], up / 16);
        else CK(hipMemcpyAsync(d[4 + r % 4], h[4 + r % 4], up, hipMemcpyHostToDevice, st[3]));
      }
      CK(hipDeviceSynchronize());
      const double dt = now() - t0;
      if (w) printf("mixed: D2H %s + H2D %s (8 MB per picture): %.0f pictures/s, D2H %.1f GB/s + H2D %.1f GB/s\n", (mode & 1) ? "kernel" : "memcpy", (mode & 2) ? "kernel" : "memcpy",
                    REP / dt, REP * bytes / dt / 1e9, REP * up / dt / 1e9);
    }
  }
  // the same with the host's cores busy on memory (the build threads of the product stage ~8 MB per picture)
  for (int nthreads : {4, 12}) {
    std::atomic<bool> stop{false}; std::atomic<long> moved{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back([&] {
      std::vector<char> a(16u << 20, 1), b(16u << 20);
      while (!stop.load()) { memcpy(b.data(), a.data(), a.size()); moved += (long)a.size(); }
    });
    const size_t up = 8u << 20;
    for (int w = 0; w < 2; w++) {
      const long m0 = moved.load(); const double t0 = now();
      for (int r = 0; r < REP; r++) {
        hipLaunchKernelGGL(k_copy, dim3(16), dim3(256), 0, st[r % 2], (const uint4*)d[r % 4], (uint4*)h[r % 4], bytes / 16);
        CK(hipMemcpyAsync(d[4 + r % 4], h[4 + r % 4], up, hipMemcpyHostToDevice, st[3]));
      }
      CK(hipDeviceSynchronize());
      const double dt = now() - t0;
      if (w) printf("mixed with %d host threads copying memory (%.1f GB/s of memcpy): %.0f pictures/s, D2H %.1f GB/s\n", nthreads, (moved.load() - m0) / dt / 1e9, REP / dt, REP * bytes / dt / 1e9);
    }
    stop = true; for (auto& t : th) t.join();
  }
  unsigned char* p = (unsigned char*)h[3]; printf("check %d %d\n", p[0], p[bytes - 1]);
  return 0;
}
