// k_rext.hip -- range-extension sample paths for gfx950: the residual tools beyond Main / Main10 and the
// chroma planes of 4:2:2 / 4:4:4 pictures in the loop filters.
//
// Main / Main10 pictures never come here: their kernels (k_tu / k_mc / k_lf.hip) are tuned for 4:2:0 geometry and the
// Main tool set.  A 4:2:2 / 4:4:4 picture sends its LUMA plane through those same kernels and its chroma planes through the
// plain kernels below (one lane per edge segment / per sample, every quantity computed where it is used); a TU that
// uses RDPCM, coefficient rotation, cross-component prediction or transform skip beyond 8x8 takes k_resid_rext.
// Behaviour follows (libde265/):
//   transform.cc:235-251, :353-625   cross_comp_pred, scale_coefficients_internal
//   fallback-dct.cc:80-90, :160-257, :470-512, :694-838   int32 residual forms, RDPCM, rotation
//   deblock.cc:730-871               edge_filtering_chroma_internal
//   sao.cc:29-254                    apply_sao_internal
#include "kernels.h"

namespace d265 {

// (own copies of the tables: device symbols are not shared between translation units without relocatable device code)
static __device__ __constant__ int8_t c_dct_mat[32 * 32] = {
#include "dct_table.inc"
};
static __device__ __constant__ int8_t c_dst_mat[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };
static __device__ __constant__ int8_t c_level_scale[6] = { 40, 45, 51, 57, 64, 72 };

__device__ __forceinline__ int rx_clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }
#define RX_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")          // one wavefront per workgroup

// The int32 residual of one TU into s_r (nT*nT, row-major): dequantisation of the sparse list (transform.cc:452-510),
// rotation, then transform bypass / transform skip (+ RDPCM) / inverse DST / inverse DCT in the int32 forms the reference
// uses on these paths.  dst_unclipped: the second stage of the inverse DST is not clipped (transform_idst_4x4_fallback; the
// fused form clips it - same pixels, but cross-component prediction reads the residual itself).
__device__ void rx_residual(const PicDev& P, int lane, int log2, int c_idx, bool intra, bool bypass, bool tskip, int qp,
                            const int16_t* __restrict__ vals, const uint16_t* __restrict__ pos, int n_coeff,
                            const uint8_t* __restrict__ scaling, int rdpcm, bool rotate, bool dst_unclipped,
                            int16_t* s_c, int16_t* s_g, int32_t* s_r)
{
  const int nT = 1 << log2, nS = nT * nT;
  const int bd = c_idx ? P.bd_chroma : P.bd_luma;
  for (int s = lane; s < nS; s += 64) s_c[s] = 0;
  RX_SYNC();
  if (bypass) {
    for (int i = lane; i < n_coeff; i += 64) s_c[pos[i]] = vals[i];
  } else if (!P.scaling_list) {
    const int bdShift = bd + log2 - 9;
    const int32_t fact = (int32_t)c_level_scale[qp % 6] << (qp / 6);
    for (int i = lane; i < n_coeff; i += 64) {
      const int32_t cc = (int32_t)((uint32_t)(int32_t)vals[i] * (uint32_t)fact + (uint32_t)(1 << (bdShift - 1)));   // 32-bit wrap
      s_c[pos[i]] = (int16_t)rx_clip3(-32768, 32767, cc >> bdShift);
    }
  } else {
    const int bdShift = bd + log2 - 5;
    int matrixID = c_idx;
    if (!intra) matrixID += (nT < 32) ? 3 : 1;
    if (nT == 32 && matrixID > 1) matrixID = intra ? 0 : 1;     // (32x32 chroma: the reference indexes beyond its two matrices - undefined there)
    const uint8_t* scl = scaling + (log2 == 2 ? 0 : (log2 == 3 ? 96 : (log2 == 4 ? 96 + 384 : 96 + 384 + 1536))) + matrixID * nS;
    for (int i = lane; i < n_coeff; i += 64) {
      const int p = pos[i];
      const int fact = ((int)scl[p] * c_level_scale[qp % 6]) << (qp / 6);
      long long cc = ((long long)vals[i] * fact + (1ll << (bdShift - 1))) >> bdShift;
      s_c[p] = (int16_t)(cc < -32768 ? -32768 : (cc > 32767 ? 32767 : cc));
    }
  }
  RX_SYNC();
  if (rotate) {                                                  // 4x4 only: coefficient i <-> 15 - i
    int a = 0, b = 0;
    if (lane < 8) { a = s_c[lane]; b = s_c[15 - lane]; }
    RX_SYNC();
    if (lane < 8) { s_c[lane] = (int16_t)b; s_c[15 - lane] = (int16_t)a; }
    RX_SYNC();
  }
  if (bypass || tskip) {
    const int bdShift = 20 - bd, tsShift = 5 + log2, rnd = 1 << (bdShift - 1);
    for (int s = lane; s < nS; s += 64) {
      const int c = s_c[s];
      s_r[s] = bypass ? c : (((int32_t)((uint32_t)c << tsShift) + rnd) >> bdShift);
    }
    RX_SYNC();
    if (rdpcm == 2 && lane < nT) { int sum = 0; for (int y = 0; y < nT; y++) { sum += s_r[lane + y * nT]; s_r[lane + y * nT] = sum; } }
    if (rdpcm == 1 && lane < nT) { int sum = 0; for (int x = 0; x < nT; x++) { sum += s_r[x + lane * nT]; s_r[x + lane * nT] = sum; } }
    RX_SYNC();
    return;
  }
  const int post = 20 - bd, rnd2 = 1 << (post - 1);
  if (nT == 4 && c_idx == 0 && intra) {                           // inverse DST
    if (lane < 16) {
      const int i = lane >> 2, c = lane & 3;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) sum += c_dst_mat[j * 4 + i] * s_c[c + j * 4];
      s_g[i * 4 + c] = (int16_t)rx_clip3(-32768, 32767, (sum + 64) >> 7);
    }
    RX_SYNC();
    if (lane < 16) {
      const int y = lane >> 2, i = lane & 3;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) sum += c_dst_mat[j * 4 + i] * s_g[y * 4 + j];
      const int out = (sum + rnd2) >> post;
      s_r[lane] = dst_unclipped ? out : rx_clip3(-32768, 32767, out);
    }
    RX_SYNC();
    return;
  }
  const int fact = 32 >> log2;
  for (int s = lane; s < nS; s += 64) {
    const int i = s >> log2, c = s & (nT - 1);
    int sum = 0;
    for (int j = 0; j < nT; j++) sum += c_dct_mat[fact * j * 32 + i] * s_c[c + j * nT];
    s_g[i * nT + c] = (int16_t)rx_clip3(-32768, 32767, (sum + 64) >> 7);
  }
  RX_SYNC();
  for (int s = lane; s < nS; s += 64) {
    const int y = s >> log2, i = s & (nT - 1);
    int sum = 0;
    for (int j = 0; j < nT; j++) sum += c_dct_mat[fact * j * 32 + i] * s_g[y * nT + j];
    s_r[s] = (sum + rnd2) >> post;
  }
  RX_SYNC();
}

// One wavefront per TU: a level-0 task (inter TU: the residual is added into the picture; residual-only copy of an intra
// TU: it goes to the residual buffer for the run kernels) with a range-extension tool in it (TuTask::pad3, D265_RX_*).
template <typename PX>
__global__ __launch_bounds__(64)
void k_resid_rext(PicDev P, PlaneRef pl0, PlaneRef pl1, PlaneRef pl2, const TuTask* __restrict__ tasks, int n,
                  const int16_t* __restrict__ coeff_val, const uint16_t* __restrict__ coeff_pos,
                  const uint8_t* __restrict__ scaling, int16_t* __restrict__ resid)
{
  __shared__ int16_t s_c[32 * 32], s_g[32 * 32];
  __shared__ int32_t s_r[32 * 32], s_ry[32 * 32];
  if ((int)blockIdx.x >= n) return;
  const TuTask t = tasks[blockIdx.x];
  const int lane = threadIdx.x;
  const int c = t.c_idx, log2 = t.log2_size, nT = 1 << log2, nS = nT * nT;
  const int rx = t.pad3;
  const bool intra = t.flags & DE265HIP_TU_INTRA;
  const int rdpcm = (rx & D265_RX_RDPCM_V) ? 2 : ((rx & D265_RX_RDPCM_H) ? 1 : 0);
  rx_residual(P, lane, log2, c, intra, t.flags & DE265HIP_TU_BYPASS, t.flags & DE265HIP_TU_TSKIP, t.qp,
              coeff_val + t.coeff_offset, coeff_pos + t.coeff_offset, t.n_coeff, scaling, rdpcm, rx & D265_RX_ROTATE,
              P.xcc_enabled != 0, s_c, s_g, s_r);
  if (rx & D265_RX_XCC) {
    // the luma TU of the same position and size, once more (its residual is not kept anywhere): transform.cc:235-251
    const uint32_t l_off = (uint32_t)t.avail;
    const int l_n = (int)((t.avail >> 32) & 0xFFFF), l_qp = (int)(int8_t)((t.avail >> 48) & 0xFF), l_fl = (int)(t.avail >> 56);
    const int l_rdpcm = (rx >> D265_RX_LUMA_RDPCM_SHIFT) & 3;
    rx_residual(P, lane, log2, 0, intra, l_fl & DE265HIP_TU_BYPASS, l_fl & DE265HIP_TU_TSKIP, l_qp,
                coeff_val + l_off, coeff_pos + l_off, l_n, scaling, l_rdpcm, rx & D265_RX_LUMA_ROT, true, s_c, s_g, s_ry);
    const int rsv = t.angle;
    for (int s = lane; s < nS; s += 64)
      s_r[s] += (rsv * ((int32_t)((uint32_t)s_ry[s] << P.bd_chroma) >> P.bd_luma)) >> 3;
    RX_SYNC();
  }
  if (t.flags & D265_TU_RESID_ONLY) {
    for (int s = lane; s < nS; s += 64) resid[t.resid_offset + s] = (int16_t)rx_clip3(-32768, 32767, s_r[s]);
    return;
  }
  const PlaneRef pr = c == 0 ? pl0 : (c == 1 ? pl1 : pl2);
  PX* dst = (PX*)pr.ptr + t.x0 + t.y0 * pr.stride;
  const int maxv = (1 << (c ? P.bd_chroma : P.bd_luma)) - 1;
  for (int s = lane; s < nS; s += 64) {
    PX* d = dst + (s & (nT - 1)) + (s >> log2) * pr.stride;
    *d = (PX)rx_clip3(0, maxv, (int)*d + s_r[s]);
  }
}
template __global__ void k_resid_rext<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int, const int16_t*, const uint16_t*, const uint8_t*, int16_t*);
template __global__ void k_resid_rext<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int, const int16_t*, const uint16_t*, const uint8_t*, int16_t*);

// ---------------------------------------------------------------- chroma deblocking for any chroma format
// One lane per edge segment of four chroma samples, both chroma planes (blockIdx.z); vertical edges in one launch,
// horizontal edges in the next (they read the vertically filtered samples).  Edges lie on the 8-sample grid of the
// CHROMA plane (deblock.cc:741-756); only bS == 2 is filtered (:763), i.e. an edge flag and an intra block on either side.
static __device__ __constant__ uint8_t c_rx_tc[54] = {
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 1,1,1,1,1,1,1,1,1, 2,2,2,2, 3,3,3,3, 4,4,4,
  5,5, 6,6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
static __device__ __constant__ uint8_t c_rx_qpc[14] = { 29,30,31,32,33,33,34,34,35,35,36,36,37,37 };

template <typename PX>
__global__ __launch_bounds__(256)
void k_deblock_chroma_any(PicDev P, PlaneRef d1, PlaneRef d2, LfMeta M, int vertical)
{
  const int sw = 1 << P.csw, sh = 1 << P.csh;
  const int xIncr = (vertical ? 2 : 1) * sw, yIncr = (vertical ? 1 : 2) * sh;       // in 4-luma-sample units
  const int nx = (P.w4 + xIncr - 1) / xIncr, ny = (P.h4 + yIncr - 1) / yIncr;
  const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y;
  if (ix >= nx || iy >= ny) return;
  const int x = ix * xIncr, y = iy * yIncr;                                          // deblk unit of the Q side
  if (vertical ? x == 0 : y == 0) return;
  const int idx = x + y * P.w4, pidx = vertical ? idx - 1 : idx - P.w4;
  const int f = M.flags[idx];
  if (!(f & (vertical ? (DE265HIP_BLK_EDGE_TU_V | DE265HIP_BLK_EDGE_PB_V) : (DE265HIP_BLK_EDGE_TU_H | DE265HIP_BLK_EDGE_PB_H)))) return;
  const int fp = M.flags[pidx];
  if (!((f | fp) & DE265HIP_BLK_INTRA)) return;                                      // bS < 2
  const int cp = blockIdx.z;
  const PlaneRef pr = cp ? d2 : d1;
  const int stride = pr.stride;
  const int xDi = (x << 2) >> P.csw, yDi = (y << 2) >> P.csh;                        // chroma samples
  PX* ptr = (PX*)pr.ptr + xDi + yDi * stride;
  const int bd = P.bd_chroma;
  const int qPi = (((int)M.qp[idx] + (int)M.qp[pidx] + 1) >> 1) + (cp ? P.cr_qp_offset : P.cb_qp_offset);
  const int QPc = P.chroma_format == 1 ? (qPi < 30 ? qPi : (qPi >= 43 ? qPi - 6 : c_rx_qpc[qPi - 30])) : min(qPi, 51);
  const de265hip_slice_params& slh = M.slices[M.ctbs[((x << 2) >> P.log2_ctb) + ((y << 2) >> P.log2_ctb) * P.ctbs_w].slice_idx];
  const int tc = c_rx_tc[rx_clip3(0, 53, QPc + 2 + slh.slice_tc_offset)] * (1 << (bd - 8));
  const bool exP = (fp & DE265HIP_BLK_BYPASS) || ((fp & DE265HIP_BLK_PCM) && P.pcm_lf_disable);
  const bool exQ = (f & DE265HIP_BLK_BYPASS) || ((f & DE265HIP_BLK_PCM) && P.pcm_lf_disable);
  const int maxv = (1 << bd) - 1;
  const int step = vertical ? stride : 1, across = vertical ? 1 : stride;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    PX* q = ptr + k * step;
    const int p1 = q[-2 * across], p0 = q[-across], q0 = q[0], q1 = q[across];
    const int delta = rx_clip3(-tc, tc, (((q0 - p0) << 2) + p1 - q1 + 4) >> 3);
    if (!exP) q[-across] = (PX)rx_clip3(0, maxv, p0 + delta);
    if (!exQ) q[0] = (PX)rx_clip3(0, maxv, q0 - delta);
  }
}
template __global__ void k_deblock_chroma_any<uint8_t>(PicDev, PlaneRef, PlaneRef, LfMeta, int);
template __global__ void k_deblock_chroma_any<uint16_t>(PicDev, PlaneRef, PlaneRef, LfMeta, int);

// ---------------------------------------------------------------- SAO of the chroma planes for any chroma format
// One lane per sample, out of place (sao.cc:344: the input is the unmodified deblocked picture); every sample is written.
// The slice / tile permissions of the 3x3 CTB neighbourhood come resolved by the host (SaoCtb::perm, chroma set).
template <typename PX>
__global__ __launch_bounds__(256)
void k_sao_chroma_any(PicDev P, PlaneRef s1, PlaneRef s2, PlaneRef d1, PlaneRef d2, SaoMeta M)
{
  const int W = P.cwidth, H = P.cheight;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W || y >= H) return;
  const int c = 1 + blockIdx.z;
  const PlaneRef sp = c == 1 ? s1 : s2, dp = c == 1 ? d1 : d2;
  const PX* in = (const PX*)sp.ptr;
  const int is = sp.stride;
  const int shW = P.log2_ctb - P.csw, shH = P.log2_ctb - P.csh;
  const int cx = x >> shW, cy = y >> shH;
  const SaoCtb rec = M.sao[cx + cy * P.ctbs_w];
  const int cur = in[x + y * is];
  int out = cur;
  const int type = rec.type[c];
  const int fl = M.flags[((x << P.csw) >> 2) + ((y << P.csh) >> 2) * P.w4];
  const bool exempt = (fl & DE265HIP_BLK_BYPASS) || ((fl & DE265HIP_BLK_PCM) && P.pcm_lf_disable);
  const int bd = P.bd_chroma, maxv = (1 << bd) - 1;
  if (type == 1 && !exempt) {
    const int k = ((cur >> (bd - 5)) - rec.band[c]) & 31;                            // bandTable[(k + pos) & 31] = k + 1 for k = 0..3
    if (k < 4) out = rx_clip3(0, maxv, cur + rec.off[c][k]);
  } else if (type == 2 && !exempt) {
    const int eo = rec.eo[c];
    const int hx = eo == 1 ? 0 : (eo == 3 ? 1 : -1), vy = eo == 0 ? 0 : -1;          // neighbour a; b is the opposite one
    const int xC = cx << shW, yC = cy << shH;
    const int ctbW = min(1 << shW, W - xC), ctbH = min(1 << shH, H - yC);
    const int i = x - xC, j = y - yC;
    bool ok = true;
    if (i == 0 || j == 0 || i == ctbW - 1 || j == ctbH - 1) {                        // sao.cc:119-164, boundary samples only
      const unsigned perm = ((unsigned)rec.perm >> 9) | ((unsigned)rec.perm_c_hi << 7);
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int xs = x + (k ? -hx : hx), ys = y + (k ? -vy : vy);
        if (xs < 0 || ys < 0 || xs >= W || ys >= H) { ok = false; break; }
        const int dx = (xs >> shW) - cx, dy = (ys >> shH) - cy;
        if (!((perm >> ((dy + 1) * 3 + dx + 1)) & 1u)) { ok = false; break; }
      }
    }
    if (ok) {
      const int a = in[x + hx + (y + vy) * is], b = in[x - hx + (y - vy) * is];
      const int e = (cur > a) - (cur < a) + (cur > b) - (cur < b);                   // -2 .. 2
      const int off = e == 0 ? 0 : rec.off[c][e < 0 ? e + 2 : e + 1];                // {o1, o2, 0, o3, o4}
      out = rx_clip3(0, maxv, cur + off);
    }
  }
  ((PX*)dp.ptr)[x + y * dp.stride] = (PX)out;
}
template __global__ void k_sao_chroma_any<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, SaoMeta);
template __global__ void k_sao_chroma_any<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, SaoMeta);

}  // namespace d265
