// Does a copy-out by the DMA engine (hsa_amd_memory_async_copy, issued next to HIP) disturb running kernels less than a copy-out by
// a shader kernel storing into pinned memory?  A streaming kernel (device-to-device, 64 MB, big grid) runs in a loop on one
// stream; pictures of 24.9 MB leave on other streams by (a) nothing, (b) the copy kernel, (c) hipMemcpyAsync, (d) hsa SDMA.
// Build: hipcc --offload-arch=gfx950 -O2 sdmaprobe.hip -o sdmaprobe.bin -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
#include <thread>
#include <atomic>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(s_, &m); fprintf(stderr, "%s: %s\n", #x, m); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static hsa_agent_t g_gpu, g_cpu; static bool have_gpu = false, have_cpu = false;
static hsa_status_t agent_cb(hsa_agent_t a, void*)
{
  hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}

int main()
{
  const size_t bytes = 24883200, big = 64u << 20;
  const int NB = 4;
  std::vector<void*> d(NB), h(NB);
  for (int i = 0; i < NB; i++) { CK(hipMalloc(&d[i], bytes)); CK(hipMemset(d[i], i + 1, bytes)); CK(hipHostMalloc(&h[i], bytes, hipHostMallocDefault)); memset(h[i], 0, bytes); }
  void *A, *B; CK(hipMalloc(&A, big)); CK(hipMalloc(&B, big)); CK(hipMemset(A, 7, big));
  hipStream_t ks, os[2]; CK(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking)); for (auto& s : os) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  HK(hsa_init());
  HK(hsa_iterate_agents(agent_cb, nullptr));
  if (!have_gpu || !have_cpu) { fprintf(stderr, "agents?\n"); return 1; }
  uint32_t bdf = 0; hsa_agent_get_info(g_gpu, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
  char busid[64] = ""; CK(hipDeviceGetPCIBusId(busid, 64, 0));
  printf("hsa gpu agent bdf 0x%x, hip device 0 %s\n", bdf, busid);
  hsa_signal_t sig[2]; for (auto& s : sig) HK(hsa_signal_create(0, 0, nullptr, &s));
  CK(hipDeviceSynchronize());
  const int KREP = 400;
  for (int mode = 0; mode < 4; mode++) {
    std::atomic<bool> stop{false}; std::atomic<long> pics{0};
    std::thread out([&] {
      int r = 0;
      while (!stop.load()) {
        if (mode == 0) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }
        if (mode == 1) { for (int k = 0; k < 2; k++) hipLaunchKernelGGL(k_copy, dim3(16), dim3(256), 0, os[k], (const uint4*)d[(r + k) % NB], (uint4*)h[(r + k) % NB], bytes / 16); CK(hipStreamSynchronize(os[0])); CK(hipStreamSynchronize(os[1])); }
        if (mode == 2) { for (int k = 0; k < 2; k++) CK(hipMemcpyAsync(h[(r + k) % NB], d[(r + k) % NB], bytes, hipMemcpyDeviceToHost, os[k])); CK(hipStreamSynchronize(os[0])); CK(hipStreamSynchronize(os[1])); }
        if (mode == 3) {
          for (int k = 0; k < 2; k++) { hsa_signal_store_relaxed(sig[k], 1); HK(hsa_amd_memory_async_copy(h[(r + k) % NB], g_cpu, d[(r + k) % NB], g_gpu, bytes, 0, nullptr, sig[k])); }
          for (int k = 0; k < 2; k++) if (hsa_signal_wait_scacquire(sig[k], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0) { fprintf(stderr, "copy failed\n"); exit(1); }
        }
        r += 2; pics += 2;
      }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    const long p0 = pics.load(); const double t0 = now();
    for (int i = 0; i < KREP; i++) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, ks, (const uint4*)A, (uint4*)B, big / 16);
    CK(hipStreamSynchronize(ks));
    const double dt = now() - t0; const long p1 = pics.load();
    stop = true; out.join();
    static const char* names[4] = { "no copy-out", "copy kernel (grid 16) on 2 streams", "hipMemcpyAsync on 2 streams", "hsa_amd_memory_async_copy (SDMA), 2 in flight" };
    printf("%-48s: streaming kernel %.1f us per launch (%.0f GB/s), copy-out %.0f pictures/s = %.1f GB/s\n", names[mode], 1e6 * dt / KREP, 2.0 * big * KREP / dt / 1e9,
           (p1 - p0) / dt, (p1 - p0) * (double)bytes / dt / 1e9);
  }
  // ---- uploads (8 MB, host -> device) through the HSA runtime, no compute queue involved: rate with 1 / 2 / 4 copies in flight, on
  // the default engine and spread over the engines the runtime reports free; alone and next to the streaming kernel
  {
    const size_t up = 8u << 20; const int NS = 4, N = 200;
    hsa_signal_t sg[NS]; for (auto& x : sg) HK(hsa_signal_create(0, 0, nullptr, &x));
    uint32_t mask = 0; hsa_status_t es = hsa_amd_memory_copy_engine_status(g_gpu, g_cpu, &mask);
    printf("copy engines free for host->device: status %d mask 0x%x\n", (int)es, mask);
    uint32_t eng[16]; int ne = 0; for (uint32_t b = 1; b && ne < 16; b <<= 1) if (mask & b) eng[ne++] = b;
    for (int loaded = 0; loaded < 2; loaded++)
      for (int spread = 0; spread < (ne > 1 ? 2 : 1); spread++)
        for (int inflight : {1, 2, 4}) {
          std::atomic<bool> stop{false};
          std::thread bg([&] { while (loaded && !stop.load()) { for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, ks, (const uint4*)A, (uint4*)B, big / 16); CK(hipStreamSynchronize(ks)); } });
          const double t0 = now();
          for (int r = 0; r < N; r++) {
            const int k = r % inflight;
            if (r >= inflight && hsa_signal_wait_scacquire(sg[k], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0) { fprintf(stderr, "copy failed\n"); exit(1); }
            hsa_signal_store_relaxed(sg[k], 1);
            if (spread) HK(hsa_amd_memory_async_copy_on_engine((char*)d[k % NB], g_gpu, (char*)h[k % NB], g_cpu, up, 0, nullptr, sg[k], (hsa_amd_sdma_engine_id_t)eng[r % ne], false));
            else HK(hsa_amd_memory_async_copy((char*)d[k % NB], g_gpu, (char*)h[k % NB], g_cpu, up, 0, nullptr, sg[k]));
          }
          for (int k = 0; k < inflight; k++) hsa_signal_wait_scacquire(sg[k], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
          const double dt = now() - t0;
          stop = true; bg.join();
          printf("H2D 8 MB by %s, %d in flight, %s: %.1f GB/s (%.0f us per copy)\n", spread ? "copy_on_engine (engines in turn)" : "hsa_amd_memory_async_copy", inflight,
                 loaded ? "next to the streaming kernel" : "alone", N * (double)up / dt / 1e9, 1e6 * dt / N);
        }
  }
  // ---- copy-outs (24.9 MB pictures as three planes 16.6 + 4.1 + 4.1 MB, device -> host) through the HSA runtime
  {
    const size_t pl[3] = { 3840u * 2160u * 2u, 1920u * 1080u * 2u, 1920u * 1080u * 2u }; const int N = 60;
    hsa_signal_t sg[2]; for (auto& x : sg) HK(hsa_signal_create(0, 0, nullptr, &x));
    uint32_t mask = 0; (void)hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &mask);
    uint32_t pref = 0; hsa_status_t ps = hsa_amd_memory_get_preferred_copy_engine(g_cpu, g_gpu, &pref);
    printf("copy engines free for device->host: mask 0x%x, preferred (status %d) 0x%x\n", mask, (int)ps, pref);
    uint32_t eng[16]; int ne = 0; for (uint32_t b = 1; b && ne < 16; b <<= 1) if (mask & b) eng[ne++] = b;
    uint32_t pe[16]; int npe = 0; for (uint32_t b = 1; b && npe < 16; b <<= 1) if (pref & b) pe[npe++] = b;
    for (int loaded = 0; loaded < 2; loaded++)
      for (int how = 0; how < 3; how++) {                   // 0: plain call; 1: free engines in turn; 2: preferred engines in turn
        if (how == 1 && ne < 2) continue;
        if (how == 2 && npe < 1) continue;
        std::atomic<bool> stop{false};
        std::thread bg([&] { while (loaded && !stop.load()) { for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, ks, (const uint4*)A, (uint4*)B, big / 16); CK(hipStreamSynchronize(ks)); } });
        const double t0 = now();
        int ecount = 0;
        for (int r = 0; r < N; r++) {                         // two pictures in flight, three copies each on one signal
          const int k = r & 1;
          if (r >= 2 && hsa_signal_wait_scacquire(sg[k], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0) { fprintf(stderr, "copy failed\n"); exit(1); }
          hsa_signal_store_relaxed(sg[k], 3);
          size_t off = 0;
          for (int c = 0; c < 3; c++) {
            char* dst = (char*)h[k] + off; const char* src = (const char*)d[k] + off; off += pl[c];
            if (how == 0) HK(hsa_amd_memory_async_copy(dst, g_cpu, src, g_gpu, pl[c], 0, nullptr, sg[k]));
            else { const uint32_t e = how == 1 ? eng[ecount++ % ne] : pe[ecount++ % npe]; HK(hsa_amd_memory_async_copy_on_engine(dst, g_cpu, src, g_gpu, pl[c], 0, nullptr, sg[k], (hsa_amd_sdma_engine_id_t)e, false)); }
          }
        }
        for (int k = 0; k < 2; k++) hsa_signal_wait_scacquire(sg[k], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
        const double dt = now() - t0;
        stop = true; bg.join();
        static const char* nm[3] = { "hsa_amd_memory_async_copy", "copy_on_engine, free engines in turn", "copy_on_engine, preferred engines in turn" };
        printf("D2H pictures by %s, %s: %.1f GB/s (%.0f pictures/s)\n", nm[how], loaded ? "next to the streaming kernel" : "alone", N * (double)bytes / dt / 1e9, N / dt);
      }
  }
  unsigned char* p = (unsigned char*)h[1]; printf("check %d %d\n", p[0], p[bytes - 1]);
  return 0;
}
