// Stand-alone experiment: what a 4K 10-bit picture pass (read + write every sample once) costs on the device for
// different access shapes, timed like the library times its kernels (hipEvents around each launch on one stream).
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/copybench tools/exp/copybench.hip && tools/exp/copybench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

struct Plane { uint16_t* p; int stride, w, h; };

__global__ void k_empty() {}

// flat: every lane copies 16 bytes, consecutive lanes consecutive addresses, grid covers the plane exactly
__global__ __launch_bounds__(256) void k_flat(Plane s, Plane d)
{
  const int chunks_per_row = s.w / 8;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int y = idx / chunks_per_row, x = (idx - y * chunks_per_row) * 8;
  if (y >= s.h) return;
  *reinterpret_cast<uint4*>(d.p + x + y * d.stride) = *reinterpret_cast<const uint4*>(s.p + x + y * s.stride);
}

// strip: a lane owns an 8-wide strip of ROWS rows and loads ROWS + 2 rows (SAO shape); LANES output lanes per wavefront
template <int ROWS, int LANES>
__global__ __launch_bounds__(256) void k_strip(Plane s, Plane d)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = ((blockIdx.x * 4 + wave) * LANES + lane - (LANES == 64 ? 0 : 1)) * 8;
  const int y0 = blockIdx.y * ROWS;
  const bool inpic = x0 >= 0 && x0 < s.w;
  uint4 v[ROWS + 2];
#pragma unroll
  for (int j = 0; j < ROWS + 2; j++) {
    const int y = y0 - 1 + j;
    v[j] = (inpic && y >= 0 && y < s.h) ? *reinterpret_cast<const uint4*>(s.p + x0 + y * s.stride) : make_uint4(0, 0, 0, 0);
  }
  if (!inpic || (LANES != 64 && (lane == 0 || lane == 63))) return;
  // (something that depends on the halo rows so that they are not optimised away)
  const uint32_t t = (v[0].x ^ v[ROWS + 1].x) & 1u;
#pragma unroll
  for (int r = 0; r < ROWS; r++)
    if (y0 + r < s.h) { uint4 o = v[r + 1]; o.x ^= t & (o.y >> 31); *reinterpret_cast<uint4*>(d.p + x0 + (y0 + r) * d.stride) = o; }
}

// tile: a wavefront owns a 64-sample wide column band: lane = 8-sample chunk c of row r: 8 chunks x 8 rows per load
// instruction (128-byte row segments), ROWS8 such groups + halo rows
template <int GROUPS>
__global__ __launch_bounds__(256) void k_tile(Plane s, Plane d)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 7, r = lane >> 3;
  const int x0 = (blockIdx.x * 4 + wave) * 64 + c * 8;
  const int y0 = blockIdx.y * GROUPS * 8;
  if (x0 >= s.w) return;
  uint4 v[GROUPS + 1];
#pragma unroll
  for (int g = 0; g < GROUPS + 1; g++) {
    const int y = y0 - 1 + g * 8 + r;
    v[g] = (y >= 0 && y < s.h && (g < GROUPS || r < 2)) ? *reinterpret_cast<const uint4*>(s.p + x0 + y * s.stride) : make_uint4(0, 0, 0, 0);
  }
  const uint32_t t = (v[0].x ^ v[GROUPS].x) & 1u;
#pragma unroll
  for (int g = 0; g < GROUPS; g++) {
    const int y = y0 + g * 8 + r;
    if (y < s.h) { uint4 o = v[g]; o.x ^= t & (o.y >> 31); *reinterpret_cast<uint4*>(d.p + x0 + y * d.stride) = o; }
  }
}

int main()
{
  const int W = 3840, H = 2160;
  Plane s[3], d[3];
  for (int c = 0; c < 3; c++) {
    const int w = c ? W / 2 : W, h = c ? H / 2 : H, stride = (w + 63) & ~63;
    s[c] = { nullptr, stride, w, h }; d[c] = s[c];
    hipMalloc(&s[c].p, (size_t)stride * h * 2 + 256); hipMalloc(&d[c].p, (size_t)stride * h * 2 + 256);
    hipMemset(s[c].p, 1 + c, (size_t)stride * h * 2);
  }
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    std::vector<float> t;
    for (int rep = 0; rep < 12; rep++) {
      hipEventRecord(e0, st);
      launch();
      hipEventRecord(e1, st);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    printf("%-34s median %.1f us  min %.1f us\n", name, t[t.size() / 2], t[0]);
  };
  timeit("empty kernel", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); });
  timeit("flat copy, 3 launches", [&] {
    for (int c = 0; c < 3; c++)
      hipLaunchKernelGGL(k_flat, dim3((s[c].w / 8 * s[c].h + 255) / 256), dim3(256), 0, st, s[c], d[c]);
  });
  timeit("flat copy, luma only", [&] { hipLaunchKernelGGL(k_flat, dim3((s[0].w / 8 * s[0].h + 255) / 256), dim3(256), 0, st, s[0], d[0]); });
#define STRIP(ROWS, LANES)                                                                                              \
  timeit("strip rows=" #ROWS " lanes=" #LANES, [&] {                                                                     \
    for (int c = 0; c < 3; c++)                                                                                          \
      hipLaunchKernelGGL((k_strip<ROWS, LANES>), dim3((s[c].w / 8 + (LANES == 64 ? 0 : 2) + 4 * LANES - 1) / (4 * LANES), (s[c].h + ROWS - 1) / ROWS), \
                         dim3(256), 0, st, s[c], d[c]);                                                                  \
  });
  STRIP(4, 62) STRIP(8, 62) STRIP(4, 64) STRIP(8, 64) STRIP(16, 64)
#define TILE(G)                                                                                                          \
  timeit("tile 64 wide, groups=" #G, [&] {                                                                               \
    for (int c = 0; c < 3; c++)                                                                                          \
      hipLaunchKernelGGL((k_tile<G>), dim3((s[c].w + 255) / 256, (s[c].h + 8 * G - 1) / (8 * G)), dim3(256), 0, st, s[c], d[c]); \
  });
  TILE(1) TILE(2) TILE(4) TILE(8)
  timeit("hipMemcpyAsync d2d, 3 planes", [&] {
    for (int c = 0; c < 3; c++) hipMemcpyAsync(d[c].p, s[c].p, (size_t)s[c].stride * s[c].h * 2, hipMemcpyDeviceToDevice, st);
  });
  printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
