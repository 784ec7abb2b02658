#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned* p, unsigned* out)
{
  unsigned v = 1;
  asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(p) : "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main()
{
  unsigned *p, *out; const int N = 4096;
  hipMalloc(&p, 4); hipMalloc(&out, N * 4); hipMemset(p, 0, 4);
  hipLaunchKernelGGL(k, dim3(N), dim3(64), 0, 0, p, out);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  std::vector<unsigned> h(N); unsigned c;
  hipMemcpy(h.data(), out, N * 4, hipMemcpyDeviceToHost); hipMemcpy(&c, p, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool ok = c == N; for (int i = 0; i < N; i++) ok = ok && h[i] == (unsigned)i;
  printf("counter %u unique-sequence %s\n", c, ok ? "yes" : "NO");
  return 0;
}
