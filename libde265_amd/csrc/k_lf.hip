// k_lf.hip -- in-loop filter kernels for gfx950: boundary strength, deblocking
// edge filters (luma + chroma, vertical and horizontal pass) and SAO.
//
// HBM-bound streaming kernels: one lane owns one 8x4 (luma) edge segment and
// moves it with 8-byte vector accesses; lanes run along x so that every row a
// wavefront touches is one contiguous span.  Behaviour follows (libde265/):
//   deblock.cc:241-375  derive_boundaryStrength
//   deblock.cc:405-699  edge_filtering_luma_internal
//   deblock.cc:730-871  edge_filtering_chroma_internal
//   sao.cc:29-254       apply_sao_internal
#include "kernels.h"

// wavefronts per SIMD the register allocation aims at (A/B with -DDEBLOCK_WAVES=.. / -DSAO_WAVES=..; measured on a
// 4K Main10 B picture: deblock V 13.5 us at 4 (79 VGPRs) vs 16.7 at 8; SAO 42.7 at 4 (83 VGPRs), 48.4 at 6, 56.7 at 8:
// the spills cost more than the extra wavefronts hide, unlike k_resid_big)
#ifndef DEBLOCK_WAVES
#define DEBLOCK_WAVES 4
#endif
#ifndef SAO_WAVES
#define SAO_WAVES 4
#endif

namespace d265 {

__device__ __constant__ uint8_t c_beta[52] = {
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 6,7,8,9,10,11,12,13,14,15,16,17,18,
  20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64 };
__device__ __constant__ uint8_t c_tc[54] = {
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 1,1,1,1,1,1,1,1,1, 2,2,2,2, 3,3,3,3, 4,4,4,
  5,5, 6,6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
__device__ __constant__ uint8_t c_qpc[14] = { 29,30,31,32,33,33,34,34,35,35,36,36,37,37 };

__device__ __forceinline__ int lf_clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }
__device__ __forceinline__ int lf_qpc(int q) { return q < 30 ? q : (q >= 43 ? q - 6 : c_qpc[q - 30]); }

// ---------------------------------------------------------------- bS
// One lane per 4x4 unit; bits 0-1: bS of the vertical edge on the unit's left
// side (x on the 8-sample grid), bits 2-3: bS of the horizontal edge on top.
__device__ __forceinline__ int bs_motion(const de265hip_motion& mP, const de265hip_motion& mQ)
{
  int rP0 = mP.ref_slot[0], rP1 = mP.ref_slot[1], rQ0 = mQ.ref_slot[0], rQ1 = mQ.ref_slot[1];
  bool same = (rP0 == rQ0 && rP1 == rQ1) || (rP0 == rQ1 && rP1 == rQ0);
  if (!same) return 1;
  int p0x = rP0 >= 0 ? mP.mv[0][0] : 0, p0y = rP0 >= 0 ? mP.mv[0][1] : 0;
  int p1x = rP1 >= 0 ? mP.mv[1][0] : 0, p1y = rP1 >= 0 ? mP.mv[1][1] : 0;
  int q0x = rQ0 >= 0 ? mQ.mv[0][0] : 0, q0y = rQ0 >= 0 ? mQ.mv[0][1] : 0;
  int q1x = rQ1 >= 0 ? mQ.mv[1][0] : 0, q1y = rQ1 >= 0 ? mQ.mv[1][1] : 0;
  bool d00 = abs(p0x - q0x) >= 4 || abs(p0y - q0y) >= 4 || abs(p1x - q1x) >= 4 || abs(p1y - q1y) >= 4;
  bool d01 = abs(p0x - q1x) >= 4 || abs(p0y - q1y) >= 4 || abs(p1x - q0x) >= 4 || abs(p1y - q0y) >= 4;
  if (rP0 != rP1) return (rP0 == rQ0) ? d00 : d01;
  return d00 && d01;
}

// The per-4x4 motion plane from the PU records (what image.h pb_info holds and derive_boundaryStrength reads, deblock.cc:295-304:
// motion vectors and reference pictures as DPB slots): a host that does not flatten its motion into de265hip_picture_desc::
// blk_motion (NULL) leaves it to this kernel - 0.5 MB of PU records cross PCIe instead of a 6.2 MB plane per 4K picture, and
// the read-out on the host goes away.  One wavefront per PU; the plane was set to "no reference" (0xFF) before.
__global__ __launch_bounds__(256)
void k_motion_from_pus(PicDev P, const de265hip_pu* __restrict__ pus, int n_pus, const de265hip_slice_params* __restrict__ slices,
                       int n_slices, de265hip_motion* __restrict__ motion)
{
  // sixteen lanes per PU, sixteen PUs per workgroup (a workgroup per PU was 33 000 one-wavefront workgroups per 4K picture: the
  // dispatcher's time, not the kernel's)
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  if (i >= n_pus) return;
  const de265hip_pu pu = pus[i];
  if (pu.slice_idx >= n_slices) return;
  de265hip_motion m;
  for (int l = 0; l < 2; l++) {
    const bool on = (pu.pred_flag >> l) & 1;
    const int ri = pu.ref_idx[l];
    m.ref_slot[l] = (on && ri >= 0 && ri < DE265HIP_MAX_REFS) ? slices[pu.slice_idx].ref_pic_list[l][ri] : (int8_t)-1;
    m.mv[l][0] = on ? pu.mv[l][0] : (int16_t)0; m.mv[l][1] = on ? pu.mv[l][1] : (int16_t)0;
  }
  m.pad[0] = m.pad[1] = 0;
  const int bw = pu.w >> 2, bh = pu.h >> 2;
  for (int q = sub; q < bw * bh; q += 16) {
    const int x = (pu.x >> 2) + q % bw, y = (pu.y >> 2) + q / bw;
    if (x < P.w4 && y < P.h4) motion[x + y * P.w4] = m;
  }
}

// bS of the edge on the left (VERT) / top side of unit idx (derive_boundaryStrength, deblock.cc:241-375), fused into
// the deblocking kernels: the motion records are only fetched for inter/inter edges without coded residual.
// chroma_only: only bS == 2 matters (deblock.cc:763), no motion needed.
template <bool VERT>
__device__ __forceinline__ int edge_bs(const uint8_t* __restrict__ flags, const de265hip_motion* __restrict__ motion,
                                       int idx, int pidx, bool chroma_only)
{
  const int f = flags[idx];
  if (!(f & (VERT ? (DE265HIP_BLK_EDGE_TU_V | DE265HIP_BLK_EDGE_PB_V) : (DE265HIP_BLK_EDGE_TU_H | DE265HIP_BLK_EDGE_PB_H)))) return 0;
  const int fp = flags[pidx];
  if ((f | fp) & DE265HIP_BLK_INTRA) return 2;
  if (chroma_only) return 0;
  if ((f & (VERT ? DE265HIP_BLK_EDGE_TU_V : DE265HIP_BLK_EDGE_TU_H)) && ((f | fp) & DE265HIP_BLK_NONZERO)) return 1;
  return bs_motion(motion[pidx], motion[idx]);
}

__global__ __launch_bounds__(256)
void k_bs(PicDev P, const uint8_t* __restrict__ flags, const de265hip_motion* __restrict__ motion,
          uint8_t* __restrict__ bs)
{
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= P.w4) return;
  int idx = x + y * P.w4;
  int f = flags[idx];
  int out = 0;
  if (!(x & 1) && (f & (DE265HIP_BLK_EDGE_TU_V | DE265HIP_BLK_EDGE_PB_V)) && x > 0) {
    int fp = flags[idx - 1];
    int b;
    if ((f | fp) & DE265HIP_BLK_INTRA) b = 2;
    else if ((f & DE265HIP_BLK_EDGE_TU_V) && ((f | fp) & DE265HIP_BLK_NONZERO)) b = 1;
    else b = bs_motion(motion[idx - 1], motion[idx]);
    out |= b;
  }
  if (!(y & 1) && (f & (DE265HIP_BLK_EDGE_TU_H | DE265HIP_BLK_EDGE_PB_H)) && y > 0) {
    int fp = flags[idx - P.w4];
    int b;
    if ((f | fp) & DE265HIP_BLK_INTRA) b = 2;
    else if ((f & DE265HIP_BLK_EDGE_TU_H) && ((f | fp) & DE265HIP_BLK_NONZERO)) b = 1;
    else b = bs_motion(motion[idx - P.w4], motion[idx]);
    out |= b << 2;
  }
  bs[idx] = (uint8_t)out;
}

// ---------------------------------------------------------------- vector helpers
template <typename PX> struct Vec4;
template <> struct Vec4<uint16_t> { typedef uint2 T; };
template <> struct Vec4<uint8_t> { typedef uint32_t T; };

template <typename PX>
__device__ __forceinline__ void load4(const PX* p, int v[4])
{
  typename Vec4<PX>::T raw = *reinterpret_cast<const typename Vec4<PX>::T*>(p);
  if constexpr (sizeof(PX) == 2) {
    v[0] = raw.x & 0xFFFF; v[1] = raw.x >> 16; v[2] = raw.y & 0xFFFF; v[3] = raw.y >> 16;
  } else {
    v[0] = raw & 0xFF; v[1] = (raw >> 8) & 0xFF; v[2] = (raw >> 16) & 0xFF; v[3] = raw >> 24;
  }
}
template <typename PX>
__device__ __forceinline__ void store4(PX* p, const int v[4])
{
  typename Vec4<PX>::T raw;
  if constexpr (sizeof(PX) == 2) {
    raw.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); raw.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
  } else {
    raw = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
  }
  *reinterpret_cast<typename Vec4<PX>::T*>(p) = raw;
}

__device__ __forceinline__ bool lf_exempt(const PicDev& P, int f)
{ return ((f & DE265HIP_BLK_PCM) && P.pcm_lf_disable) || (f & DE265HIP_BLK_BYPASS); }

// Luma filter of one segment held as p[line][i], q[line][i] (i = distance from the edge).
__device__ __forceinline__ bool luma_filter_segment(int p[4][4], int q[4][4], int beta, int tc, int bd,
                                                    bool filterP, bool filterQ)
{
  int dp0 = abs(p[0][2] - 2 * p[0][1] + p[0][0]);
  int dp3 = abs(p[3][2] - 2 * p[3][1] + p[3][0]);
  int dq0 = abs(q[0][2] - 2 * q[0][1] + q[0][0]);
  int dq3 = abs(q[3][2] - 2 * q[3][1] + q[3][0]);
  int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3;
  int dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
  if (d >= beta) return false;
  bool dSam0 = 2 * dpq0 < (beta >> 2) && abs(p[0][3] - p[0][0]) + abs(q[0][0] - q[0][3]) < (beta >> 3) &&
               abs(p[0][0] - q[0][0]) < ((5 * tc + 1) >> 1);
  bool dSam3 = 2 * dpq3 < (beta >> 2) && abs(p[3][3] - p[3][0]) + abs(q[3][0] - q[3][3]) < (beta >> 3) &&
               abs(p[3][0] - q[3][0]) < ((5 * tc + 1) >> 1);
  bool strong = dSam0 && dSam3;
  bool dEp = dp < ((beta + (beta >> 1)) >> 3);
  bool dEq = dq < ((beta + (beta >> 1)) >> 3);
  const int maxv = (1 << bd) - 1;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int p0 = p[k][0], p1 = p[k][1], p2 = p[k][2], p3 = p[k][3];
    int q0 = q[k][0], q1 = q[k][1], q2 = q[k][2], q3 = q[k][3];
    if (strong) {
      if (filterP) {
        p[k][0] = lf_clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
        p[k][1] = lf_clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
        p[k][2] = lf_clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
      }
      if (filterQ) {
        q[k][0] = lf_clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
        q[k][1] = lf_clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
        q[k][2] = lf_clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
      }
    } else {
      int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (abs(delta) < tc * 10) {
        delta = lf_clip3(-tc, tc, delta);
        if (filterP) p[k][0] = lf_clip3(0, maxv, p0 + delta);
        if (filterQ) q[k][0] = lf_clip3(0, maxv, q0 - delta);
        if (dEp && filterP) {
          int dP = lf_clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1);
          p[k][1] = lf_clip3(0, maxv, p1 + dP);
        }
        if (dEq && filterQ) {
          int dQ = lf_clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1);
          q[k][1] = lf_clip3(0, maxv, q1 + dQ);
        }
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------- deblock
// blockIdx.z: 0 luma, 1 Cb, 2 Cr.  VERT: vertical edges (filtering across x).
template <typename PX, bool VERT>
__global__ __launch_bounds__(256, DEBLOCK_WAVES)
void k_deblock(PicDev P, PlaneRef pl0, PlaneRef pl1, PlaneRef pl2, LfMeta M)
{
  const int comp = blockIdx.z;
  const int tx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ty = blockIdx.y;
  if (comp == 0) {
    // luma: unit grid; vertical edges at even x, horizontal at even y
    const int x = VERT ? tx * 2 : tx;
    const int y = VERT ? ty : ty * 2;
    if (x >= P.w4 || y >= P.h4) return;
    const int idx = x + y * P.w4;
    if ((VERT ? x : y) == 0) return;                                  // the picture border is no edge
    const int bS = M.bs ? (VERT ? (M.bs[idx] & 3) : ((M.bs[idx] >> 2) & 3))
                        : edge_bs<VERT>(M.flags, M.motion, idx, VERT ? idx - 1 : idx - P.w4, false);
    if (bS == 0) return;
    const int xDi = x << 2, yDi = y << 2;
    const int stride = pl0.stride;
    PX* ptr = (PX*)pl0.ptr + xDi + yDi * stride;
    int p[4][4], q[4][4];
    if (VERT) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int a[4];
        load4<PX>(ptr + k * stride - 4, a);
        p[k][3] = a[0]; p[k][2] = a[1]; p[k][1] = a[2]; p[k][0] = a[3];
        load4<PX>(ptr + k * stride, q[k]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        int a[4], b[4];
        load4<PX>(ptr - (i + 1) * stride, a);
        load4<PX>(ptr + i * stride, b);
#pragma unroll
        for (int k = 0; k < 4; k++) { p[k][i] = a[k]; q[k][i] = b[k]; }
      }
    }
    const int pidx = VERT ? idx - 1 : idx - P.w4;
    const int qP_L = ((int)M.qp[idx] + (int)M.qp[pidx] + 1) >> 1;
    const de265hip_slice_params* sh =
      &M.slices[M.ctbs[(xDi >> P.log2_ctb) + (yDi >> P.log2_ctb) * P.ctbs_w].slice_idx];
    const int bd = P.bd_luma;
    const int beta = c_beta[lf_clip3(0, 51, qP_L + sh->slice_beta_offset)] * (1 << (bd - 8));
    const int tc = c_tc[lf_clip3(0, 53, qP_L + 2 * (bS - 1) + sh->slice_tc_offset)] * (1 << (bd - 8));
    const bool filterP = !lf_exempt(P, M.flags[pidx]);
    const bool filterQ = !lf_exempt(P, M.flags[idx]);
    if (!luma_filter_segment(p, q, beta, tc, bd, filterP, filterQ)) return;
    if (VERT) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int a[4] = { p[k][3], p[k][2], p[k][1], p[k][0] };
        if (filterP) store4<PX>(ptr + k * stride - 4, a);
        if (filterQ) store4<PX>(ptr + k * stride, q[k]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        int a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { a[k] = p[k][i]; b[k] = q[k][i]; }
        if (filterP) store4<PX>(ptr - (i + 1) * stride, a);
        if (filterQ) store4<PX>(ptr + i * stride, b);
      }
    }
  } else {
    // chroma 4:2:0: edges on the 8-sample chroma grid = every 4th unit across the
    // edge, one segment of 4 chroma lines = 2 units along it; only bS == 2.
    const int x = VERT ? tx * 4 : tx * 2;
    const int y = VERT ? ty * 2 : ty * 4;
    if (x >= P.w4 || y >= P.h4) return;
    const int idx = x + y * P.w4;
    if ((VERT ? x : y) == 0) return;
    const int bS = M.bs ? (VERT ? (M.bs[idx] & 3) : ((M.bs[idx] >> 2) & 3))
                        : edge_bs<VERT>(M.flags, M.motion, idx, VERT ? idx - 1 : idx - P.w4, true);
    if (bS < 2) return;
    const PlaneRef pl = comp == 1 ? pl1 : pl2;
    const int stride = pl.stride;
    const int xDi = x << 1, yDi = y << 1;            // chroma samples
    PX* ptr = (PX*)pl.ptr + xDi + yDi * stride;
    const int pidx = VERT ? idx - 1 : idx - P.w4;
    const int cQpPicOffset = comp == 1 ? P.cb_qp_offset : P.cr_qp_offset;
    const int qPi = (((int)M.qp[idx] + (int)M.qp[pidx] + 1) >> 1) + cQpPicOffset;
    const int QPc = lf_qpc(qPi);
    const de265hip_slice_params* sh =
      &M.slices[M.ctbs[((x << 2) >> P.log2_ctb) + ((y << 2) >> P.log2_ctb) * P.ctbs_w].slice_idx];
    const int bd = P.bd_chroma;
    const int maxv = (1 << bd) - 1;
    const int tc = c_tc[lf_clip3(0, 53, QPc + 2 * (bS - 1) + sh->slice_tc_offset)] * (1 << (bd - 8));
    const bool filterP = !lf_exempt(P, M.flags[pidx]);
    const bool filterQ = !lf_exempt(P, M.flags[idx]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int p0, p1, q0, q1;
      PX *pp0, *pq0;
      if (VERT) {
        PX* row = ptr + k * stride;
        p1 = row[-2]; p0 = row[-1]; q0 = row[0]; q1 = row[1];
        pp0 = row - 1; pq0 = row;
      } else {
        p1 = ptr[k - 2 * stride]; p0 = ptr[k - stride]; q0 = ptr[k]; q1 = ptr[k + stride];
        pp0 = ptr + k - stride; pq0 = ptr + k;
      }
      int delta = lf_clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
      if (filterP) *pp0 = (PX)lf_clip3(0, maxv, p0 + delta);
      if (filterQ) *pq0 = (PX)lf_clip3(0, maxv, q0 - delta);
    }
  }
}

template __global__ void k_deblock<uint8_t, true>(PicDev, PlaneRef, PlaneRef, PlaneRef, LfMeta);
template __global__ void k_deblock<uint8_t, false>(PicDev, PlaneRef, PlaneRef, PlaneRef, LfMeta);
template __global__ void k_deblock<uint16_t, true>(PicDev, PlaneRef, PlaneRef, PlaneRef, LfMeta);
template __global__ void k_deblock<uint16_t, false>(PicDev, PlaneRef, PlaneRef, PlaneRef, LfMeta);

// ---------------------------------------------------------------- deblock, both directions in one pass
// Every sample of the picture belongs to exactly one 8x8 block centred on a crossing of the 8-sample edge grid
// (block (bx, by) = columns 8bx-4 .. 8bx+3, rows 8by-4 .. 8by+3).  The vertical edge through the block's centre is the only
// vertical edge whose filter reads or writes samples of the block (a filter reads 4 and writes 3 samples either side), the
// same holds for the horizontal edge, and the horizontal filter's inputs - the vertically filtered samples of rows
// 8by-4 .. 8by+3 - all lie in the block.  So one lane takes one block: load 8x8, filter the two vertical-edge segments,
// then the two horizontal-edge segments on the result (deblock.cc:936-1020 filters all vertical edges of the picture
// before any horizontal edge: same result), store what changed.  One read and one write of the picture instead of two
// each, one launch instead of two.  Loads are unconditional (clamped addresses), see k_sao.
#ifndef DEBLOCK_FUSED_WAVES
#define DEBLOCK_FUSED_WAVES 4     // (114 VGPRs; 5/6/8 wavefronts per SIMD spill 84/164/264 B to scratch: 3-stream bench 5 820 -> 5 660/5 330/4 970 frames/s)
#endif
// One 8x8 deblocking block (bx, by) of component comp in three stages, so that a caller can put other work (and other
// loads) between them: meta1 issues the first round of metadata loads, meta2 derives bS as far as possible and issues the
// second round (slice offsets, motion records), filter reads the block's samples at `base` + y * stride + x - a plane in
// global memory (k_deblock_fused) or a tile of it staged in LDS (k_lf_tile: PX is then the tile's uint16_t whatever the
// picture's depth) - filters and stores what changed.
template <typename PX>
struct DeblockLane {
  bool chroma, live, any, any_m;
  int comp, W, H, x0, y0;
  bool ex[4]; int idx[4], pidx[4];
  int f[4], fp[4], qq[4], qp_[4], sl[4];
  int bs[4]; bool need_m[4];
  int boff[4], toff[4];
  uint32_t rp[4][3], rq[4][3];

  __device__ __forceinline__ void meta1(const PicDev& P, const LfMeta& M, int comp_, int bx, int by)
  {
    comp = comp_; chroma = comp != 0;
    W = chroma ? P.width >> 1 : P.width; H = chroma ? P.height >> 1 : P.height;
    x0 = 8 * bx - 4; y0 = 8 * by - 4;
    live = x0 < W && y0 < H;
    any = false; any_m = false;
    if (!live) return;
    const int ush = chroma ? 1 : 2;                      // a unit is 4 luma = 2 chroma samples
    // ---- the four edge segments of the block: k = 0, 1: vertical edge, rows 4k..4k+3; k = 2, 3: horizontal edge, columns
    // 4(k-2)..  Everything is fetched in two rounds for all four together (not one dependent chain per segment):
    // round 1: flags and QP of both sides, the Q side's slice index; round 2: the slice's offsets, the motion records
    // where bS depends on them, and the block's samples.
    const bool hasV = bx > 0 && 8 * bx < W, hasH = by > 0 && 8 * by < H;
    int ux[4], uy[4];
  #pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool vert = k < 2; const int sgm = k & 1;
      ux[k] = vert ? (8 * bx) >> ush : (x0 + 4 * sgm) >> ush;
      uy[k] = vert ? (y0 + 4 * sgm) >> ush : (8 * by) >> ush;
      ex[k] = vert ? (hasV && y0 + 4 * sgm >= 0 && y0 + 4 * sgm < H) : (hasH && x0 + 4 * sgm >= 0 && x0 + 4 * sgm < W);
      // (clamped: the loads below are unconditional)
      const int cx = min(max(ux[k], vert ? 1 : 0), P.w4 - 1), cy = min(max(uy[k], vert ? 0 : 1), P.h4 - 1);
      idx[k] = cx + cy * P.w4; pidx[k] = vert ? idx[k] - 1 : idx[k] - P.w4;
      ux[k] = cx; uy[k] = cy;
    }
  #pragma unroll
    for (int k = 0; k < 4; k++) {
      f[k] = M.flags[idx[k]]; fp[k] = M.flags[pidx[k]];
      qq[k] = M.qp[idx[k]]; qp_[k] = M.qp[pidx[k]];
      sl[k] = M.ctbs[((ux[k] << 2) >> P.log2_ctb) + ((uy[k] << 2) >> P.log2_ctb) * P.ctbs_w].slice_idx;
    }
  }

  __device__ __forceinline__ void meta2(const LfMeta& M)
  {
    if (!live) return;
  #pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool vert = k < 2;
      const int e_any = vert ? (DE265HIP_BLK_EDGE_TU_V | DE265HIP_BLK_EDGE_PB_V) : (DE265HIP_BLK_EDGE_TU_H | DE265HIP_BLK_EDGE_PB_H);
      const int e_tu = vert ? DE265HIP_BLK_EDGE_TU_V : DE265HIP_BLK_EDGE_TU_H;
      bs[k] = 0; need_m[k] = false;
      if (ex[k] && (f[k] & e_any)) {                     // derive_boundaryStrength (deblock.cc:241-375), as in edge_bs
        if ((f[k] | fp[k]) & DE265HIP_BLK_INTRA) bs[k] = 2;
        else if (chroma) bs[k] = 0;                      // chroma filters bS == 2 only (deblock.cc:763)
        else if ((f[k] & e_tu) && ((f[k] | fp[k]) & DE265HIP_BLK_NONZERO)) bs[k] = 1;
        else need_m[k] = true;
      }
      any = any || bs[k] != 0 || need_m[k]; any_m = any_m || need_m[k];
    }
    if (!any) { live = false; return; }
    // round 2
  #pragma unroll
    for (int k = 0; k < 4; k++) { boff[k] = M.slices[sl[k]].slice_beta_offset; toff[k] = M.slices[sl[k]].slice_tc_offset; }
    // (the motion records as raw dwords, fetched here and looked at behind the sample loads: one wait for all; a struct
    //  copy unpacks the fields inside the branch and waits there)
    static_assert(sizeof(de265hip_motion) == 12, "de265hip_motion layout");
  #pragma unroll
    for (int k = 0; k < 4; k++)
      if (need_m[k]) {
        const uint32_t* a = reinterpret_cast<const uint32_t*>(M.motion + pidx[k]);
        const uint32_t* b = reinterpret_cast<const uint32_t*>(M.motion + idx[k]);
        rp[k][0] = a[0]; rp[k][1] = a[1]; rp[k][2] = a[2]; rq[k][0] = b[0]; rq[k][1] = b[1]; rq[k][2] = b[2];
      }
  }

  __device__ __forceinline__ void filter(const PicDev& P, PX* base, int stride)
  {
    if (!live) return;
    const bool okx[2] = { x0 >= 0, x0 + 4 < W };
    int px[8][8];
    {
      const int xa = okx[0] ? x0 : x0 + 4, xb = okx[1] ? x0 + 4 : x0;
  #pragma unroll
      for (int r = 0; r < 8; r++) {
        const PX* row = base + min(max(y0 + r, 0), H - 1) * stride;
        load4<PX>(row + xa, &px[r][0]);
        load4<PX>(row + xb, &px[r][4]);
      }
    }
    if (any_m) {
  #pragma unroll
      for (int k = 0; k < 4; k++)
        if (need_m[k]) {
          de265hip_motion mp, mq;
          __builtin_memcpy(&mp, rp[k], 12); __builtin_memcpy(&mq, rq[k], 12);
          bs[k] = bs_motion(mp, mq);
        }
    }
    const int cQp = comp == 1 ? P.cb_qp_offset : P.cr_qp_offset;
    const int bd = chroma ? P.bd_chroma : P.bd_luma, maxv = (1 << bd) - 1;
    const int need = chroma ? 2 : 1;
    unsigned mod_rows = 0, mod_cols = 0;                 // bit s: segment s of the vertical / horizontal edge changed samples
  #pragma unroll
    for (int k = 0; k < 4; k++) {
      if (bs[k] < need) continue;
      const bool vert = k < 2; const int sgm = k & 1;
      // edge parameters (deblock.cc:497-528 luma, :809-832 chroma): offsets of the Q side's slice
      const int qavg = (qq[k] + qp_[k] + 1) >> 1;
      const int qp = chroma ? lf_qpc(qavg + cQp) : qavg;
      const int beta = chroma ? 0 : c_beta[lf_clip3(0, 51, qp + boff[k])] * (1 << (bd - 8));
      const int tc = c_tc[lf_clip3(0, 53, qp + 2 * (bs[k] - 1) + toff[k])] * (1 << (bd - 8));
      const bool fP = !lf_exempt(P, fp[k]), fQ = !lf_exempt(P, f[k]);
      if (!chroma) {
        int p[4][4], q[4][4];
  #pragma unroll
        for (int j = 0; j < 4; j++)
  #pragma unroll
          for (int i = 0; i < 4; i++) {
            p[j][i] = vert ? px[4 * sgm + j][3 - i] : px[3 - i][4 * sgm + j];
            q[j][i] = vert ? px[4 * sgm + j][4 + i] : px[4 + i][4 * sgm + j];
          }
        if (luma_filter_segment(p, q, beta, tc, bd, fP, fQ)) {
          if (vert) mod_rows |= 1u << sgm; else mod_cols |= 1u << sgm;
  #pragma unroll
          for (int j = 0; j < 4; j++)
  #pragma unroll
            for (int i = 0; i < 3; i++) {
              if (vert) { px[4 * sgm + j][3 - i] = p[j][i]; px[4 * sgm + j][4 + i] = q[j][i]; }
              else { px[3 - i][4 * sgm + j] = p[j][i]; px[4 + i][4 * sgm + j] = q[j][i]; }
            }
        }
      } else {
        if (vert) mod_rows |= 1u << sgm; else mod_cols |= 1u << sgm;
  #pragma unroll
        for (int j = 0; j < 4; j++) {
          const int c = 4 * sgm + j;
          int &p1 = vert ? px[c][2] : px[2][c], &p0 = vert ? px[c][3] : px[3][c];
          int &q0 = vert ? px[c][4] : px[4][c], &q1 = vert ? px[c][5] : px[5][c];
          const int delta = lf_clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
          if (fP) p0 = lf_clip3(0, maxv, p0 + delta);
          if (fQ) q0 = lf_clip3(0, maxv, q0 - delta);
        }
      }
    }
    // ---- store the halves of rows that a filter touched (vertical segment s: rows 4s..4s+3, both halves; horizontal
    // segment s: rows 1..6 of half s)
  #pragma unroll
    for (int r = 0; r < 8; r++) {
      const int y = y0 + r;
      if (y < 0 || y >= H) continue;
      const bool rowV = (mod_rows >> (r >> 2)) & 1u;
      const bool inH = r >= 1 && r <= 6;
  #pragma unroll
      for (int h = 0; h < 2; h++)
        if (okx[h] && (rowV || (inH && ((mod_cols >> h) & 1u)))) store4<PX>(base + y * stride + x0 + 4 * h, &px[r][4 * h]);
    }
  }
};

template <typename PX>
__device__ __forceinline__ void deblock_block(const PicDev& P, const LfMeta& M, int comp, PX* base, int stride, int bx, int by)
{
  DeblockLane<PX> L;
  L.meta1(P, M, comp, bx, by);
  L.meta2(M);
  L.filter(P, base, stride);
}

template <typename PX>
__global__ __launch_bounds__(256, DEBLOCK_FUSED_WAVES)
void k_deblock_fused(PicDev P, PlaneRef pl0, PlaneRef pl1, PlaneRef pl2, LfMeta M)
{
  const int comp = blockIdx.z;
  const PlaneRef pl = comp == 0 ? pl0 : (comp == 1 ? pl1 : pl2);
  deblock_block<PX>(P, M, comp, (PX*)pl.ptr, pl.stride, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y);
}
template __global__ void k_deblock_fused<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, LfMeta);
template __global__ void k_deblock_fused<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, LfMeta);

// ---------------------------------------------------------------- SAO
// Out of place: src = deblocked picture, dst = output picture (every sample is
// written, so dst needs no initialisation).  One lane per 8 samples of a row.
#define SAO_LANES 62                    // output lanes per wavefront; lanes 0 and 63 are halo lanes

template <typename PX> __device__ __forceinline__ void load8i(const PX* p, int v[8]);
template <> __device__ __forceinline__ void load8i<uint16_t>(const uint16_t* p, int v[8])
{
  const uint4 r = *reinterpret_cast<const uint4*>(p);
  v[0] = r.x & 0xFFFF; v[1] = r.x >> 16; v[2] = r.y & 0xFFFF; v[3] = r.y >> 16;
  v[4] = r.z & 0xFFFF; v[5] = r.z >> 16; v[6] = r.w & 0xFFFF; v[7] = r.w >> 16;
}
template <> __device__ __forceinline__ void load8i<uint8_t>(const uint8_t* p, int v[8])
{
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  v[0] = r.x & 0xFF; v[1] = (r.x >> 8) & 0xFF; v[2] = (r.x >> 16) & 0xFF; v[3] = r.x >> 24;
  v[4] = r.y & 0xFF; v[5] = (r.y >> 8) & 0xFF; v[6] = (r.y >> 16) & 0xFF; v[7] = r.y >> 24;
}
template <typename PX> __device__ __forceinline__ void store8i(PX* p, const int v[8]);
template <> __device__ __forceinline__ void store8i<uint16_t>(uint16_t* p, const int v[8])
{
  uint4 r;
  r.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); r.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
  r.z = (uint32_t)v[4] | ((uint32_t)v[5] << 16); r.w = (uint32_t)v[6] | ((uint32_t)v[7] << 16);
  *reinterpret_cast<uint4*>(p) = r;
}
template <> __device__ __forceinline__ void store8i<uint8_t>(uint8_t* p, const int v[8])
{
  uint2 r;
  r.x = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
  r.y = (uint32_t)v[4] | ((uint32_t)v[5] << 8) | ((uint32_t)v[6] << 16) | ((uint32_t)v[7] << 24);
  *reinterpret_cast<uint2*>(p) = r;
}

template <bool WIDE> struct SaoMaskT { typedef unsigned long long T; };
template <> struct SaoMaskT<false> { typedef uint32_t T; };

// 8 samples as four dwords of two 16-bit samples each (8-bit samples are widened / narrowed)
template <typename PX> __device__ __forceinline__ uint4 load8_pk(const PX* p);
template <> __device__ __forceinline__ uint4 load8_pk<uint16_t>(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
template <> __device__ __forceinline__ uint4 load8_pk<uint8_t>(const uint8_t* p)
{
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  return make_uint4(__builtin_amdgcn_perm(0, r.x, 0x0C010C00u), __builtin_amdgcn_perm(0, r.x, 0x0C030C02u),
                    __builtin_amdgcn_perm(0, r.y, 0x0C010C00u), __builtin_amdgcn_perm(0, r.y, 0x0C030C02u));
}
template <typename PX> __device__ __forceinline__ void store8_pk(PX* p, uint4 v);
template <> __device__ __forceinline__ void store8_pk<uint16_t>(uint16_t* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
template <> __device__ __forceinline__ void store8_pk<uint8_t>(uint8_t* p, uint4 v)
{
  uint2 r;
  r.x = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u);
  r.y = __builtin_amdgcn_perm(v.w, v.z, 0x06040200u);
  *reinterpret_cast<uint2*>(p) = r;
}

// One lane filters an 8 (wide) x 8 (high) strip.  All ten rows it needs are requested up
// front as 16-byte loads (ten loads in flight per lane), the left/right neighbours come from the
// adjacent lanes through DPP wave shifts (lanes 0 and 63 of every wavefront only supply them),
// the CTB parameters and the slice/tile permissions of the 3x3 CTB neighbourhood are evaluated
// once per strip, and every output row is one 16-byte store.
struct SaoRec { uint32_t w[6]; };
struct SaoRows { uint4 r[SAO_ROWS + 2]; };
// The part of a strip behind its loads: w = the CTB's record, R = rows y0-1 .. y0+SAO_ROWS of the strip, packed.
// NB_DPP: the left / right neighbour samples of every row come from the adjacent lanes (k_sao: lanes 0 and 63 of a wavefront are
// halo lanes); otherwise from nbL / nbR (k_lf_tile reads them from its LDS tile).
struct SaoNb { uint32_t l[SAO_ROWS + 2], r[SAO_ROWS + 2]; };
template <typename PX, bool NB_DPP>
__device__ __forceinline__ void sao_strip(const PicDev& P, const SaoMeta& M, PX* dst, int dstride, int comp, int cs,
                                          int width, int height, int lane, int x0, int y0, bool inpic, int ctbshift,
                                          const SaoRec rec, const SaoRows rows, const SaoNb nb)
{
  const uint32_t (&w)[6] = rec.w;
  const uint4 (&R)[SAO_ROWS + 2] = rows.r;
  // (the record's fields first: its loads were issued ahead of the rows', so waiting for them leaves the rows in flight;
  //  behind the rows' first use the compiler would sink the loads themselves there - a second memory latency in sequence.
  //  An empty asm statement that merely "uses" the six words does the same but broke component 2 after a refactoring.)
  const int bd = comp ? P.bd_chroma : P.bd_luma;
  const int maxv = (1 << bd) - 1;
#define SAO_BYTE(k) ((w[(k) >> 2] >> (((k) & 3) * 8)) & 0xFFu)        // byte k of the SaoCtb record (dev_common.h)
  int type, eo, band_pos, o4[4];
  if (comp == 0) {
    type = SAO_BYTE(0); eo = SAO_BYTE(3); band_pos = SAO_BYTE(6);
    o4[0] = (int8_t)SAO_BYTE(9); o4[1] = (int8_t)SAO_BYTE(10); o4[2] = (int8_t)SAO_BYTE(11); o4[3] = (int8_t)SAO_BYTE(12);
  } else if (comp == 1) {
    type = SAO_BYTE(1); eo = SAO_BYTE(4); band_pos = SAO_BYTE(7);
    o4[0] = (int8_t)SAO_BYTE(13); o4[1] = (int8_t)SAO_BYTE(14); o4[2] = (int8_t)SAO_BYTE(15); o4[3] = (int8_t)SAO_BYTE(16);
  } else {
    type = SAO_BYTE(2); eo = SAO_BYTE(5); band_pos = SAO_BYTE(8);
    o4[0] = (int8_t)SAO_BYTE(17); o4[1] = (int8_t)SAO_BYTE(18); o4[2] = (int8_t)SAO_BYTE(19); o4[3] = (int8_t)SAO_BYTE(20);
  }
#undef SAO_BYTE
  // luma: bits 0..8 of SaoCtb::perm; chroma: its bits 9..15 + perm_c_hi (dev_common.h)
  const unsigned perm = comp == 0 ? (w[5] >> 16) & 0x1FFu : ((w[5] >> 25) & 0x7Fu) | (((w[5] >> 8) & 3u) << 7);

  // S[j][m], m = 0..4: the row shifted by one sample: (left neighbour, v0), (v1, v2), (v3, v4), (v5, v6), (v7, right
  // neighbour); the neighbours come from the adjacent lanes (DPP wave shifts; lanes 0 and 63 only supply them)
  uint32_t S[SAO_ROWS + 2][5];
#pragma unroll
  for (int j = 0; j < SAO_ROWS + 2; j++) {
    uint32_t pw, nx;
    if constexpr (NB_DPP) {
      pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)R[j].w, 0x138, 0xf, 0xf, false);   // wave_shr:1 -> lane-1's (v6, v7)
      nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)R[j].x, 0x130, 0xf, 0xf, false);   // wave_shl:1 -> lane+1's (v0, v1)
    } else { pw = nb.l[j] << 16; nx = nb.r[j]; }
    S[j][0] = __builtin_amdgcn_alignbit(R[j].x, pw, 16);
    S[j][1] = __builtin_amdgcn_alignbit(R[j].y, R[j].x, 16);
    S[j][2] = __builtin_amdgcn_alignbit(R[j].z, R[j].y, 16);
    S[j][3] = __builtin_amdgcn_alignbit(R[j].w, R[j].z, 16);
    S[j][4] = __builtin_amdgcn_alignbit(nx, R[j].w, 16);
  }
  if (!inpic || (NB_DPP && (lane == 0 || lane == 63))) return;
  if (P.dbg & 32) {                                           // ablation: pure copy
#pragma unroll
    for (int r = 0; r < SAO_ROWS; r++)
      if (y0 + r < height) store8_pk<PX>(dst + x0 + (y0 + r) * dstride, R[r + 1]);
    return;
  }

  const int nrows = min(SAO_ROWS, height - y0);

  if (type == 0) {                                            // plain copy of the deblocked samples
#pragma unroll
    for (int r = 0; r < SAO_ROWS; r++)
      if (r < nrows) store8_pk<PX>(dst + x0 + (y0 + r) * dstride, R[r + 1]);
    return;
  }

  if (P.dbg & 64) type = 1;                                   // ablation: band offset everywhere

  // pcm / transquant-bypass exemptions of the 4x4 units under the strip (only when the picture has any)
  // luma: 2 units x 2 unit rows; chroma: 4 units x 4 unit rows; bit (unit_row*4 + unit)
  unsigned exm = 0;
  if (P.has_exempt) {
    const int urows = cs ? SAO_ROWS / 2 : (SAO_ROWS + 3) / 4, ucols = cs ? 4 : 2;
    for (int ur = 0; ur < urows; ur++)
      for (int uc = 0; uc < ucols; uc++) {
        const int lx = ((x0 << cs) >> 2) + uc, ly = ((y0 << cs) >> 2) + ur;
        if (lx < P.w4 && ly < P.h4 && lf_exempt(P, M.flags[lx + ly * P.w4])) exm |= 1u << (ur * 4 + uc);
      }
  }

  if (type == 1) {                                            // band offset (sao.cc:182-251)
    const int bandShift = bd - 5, left = band_pos;
#pragma unroll
    for (int r = 0; r < SAO_ROWS; r++) {
      if (r >= nrows) break;
      const uint32_t rw[4] = { R[r + 1].x, R[r + 1].y, R[r + 1].z, R[r + 1].w };
      int out[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int c = (rw[i >> 1] >> (16 * (i & 1))) & 0xFFFF;
        const int k = ((c >> bandShift) - left) & 31;         // bandTable[(k+left)&31] = k+1
        const bool ex = (exm >> ((cs ? r >> 1 : r >> 2) * 4 + (cs ? i >> 1 : i >> 2))) & 1;
        const int off = k == 0 ? o4[0] : (k == 1 ? o4[1] : (k == 2 ? o4[2] : o4[3]));   // no dynamic register indexing
        out[i] = (!ex && k < 4) ? lf_clip3(0, maxv, c + off) : c;
      }
      store8i<PX>(dst + x0 + (y0 + r) * dstride, out);
    }
    return;
  }

  // ---- edge offset (sao.cc:75-178)
  const int mask = (1 << ctbshift) - 1;
  const bool e0 = eo == 0, e1 = eo == 1, e2 = eo == 2;
  const int hx = (eo == 1) ? 0 : (eo == 3 ? 1 : -1), hy = (eo == 0) ? 0 : -1;   // first neighbour; second is the mirror

  // ---- which of the samples may be modified: one 64-bit mask per strip (bit r*8+i) instead of
  // per-sample boundary logic.  A sample is blocked when one of its two neighbours lies outside the
  // picture, or in another CTB whose slice/tile rules forbid filtering across (perm), or when it is
  // pcm/bypass-exempt, or outside the picture itself.
  const int ncols = min(8, width - x0);
  // (SAO_ROWS <= 4: the whole mask is one 32-bit word)
  typedef typename SaoMaskT<(SAO_ROWS > 4)>::T mask_t;
  const mask_t COL0 = (mask_t)0x0101010101010101ull, ROW0 = (mask_t)0xFF;
  const mask_t colL = COL0, colR = COL0 << (ncols - 1);
  const mask_t rowT = ROW0, rowB = ROW0 << (8 * (nrows - 1));
  const bool touchL = (x0 & mask) == 0, touchR = ((x0 + ncols - 1) & mask) == mask || x0 + ncols >= width;
  const bool touchT = (y0 & mask) == 0, touchB = ((y0 + nrows - 1) & mask) == mask || y0 + nrows >= height;
  // permission of the CTB at offset (dx,dy) (0 = this CTB); out-of-picture CTBs have their bit cleared in perm
  auto permitted = [&](int dx, int dy) -> bool { return (perm >> ((dy + 1) * 3 + dx + 1)) & 1; };
  // the samples of the strip on the CTB's outline: only they are tested at all (sao.cc:120 testBoundary)
  const mask_t onEdge = (touchL ? colL : (mask_t)0) | (touchR ? colR : (mask_t)0) | (touchT ? rowT : (mask_t)0) |
                        (touchB ? rowB : (mask_t)0);
  auto blocked = [&](int sx, int sy) -> mask_t {              // samples whose neighbour in direction (sx,sy) is not usable
    const int dx = sx < 0 ? (touchL ? -1 : 0) : (sx > 0 ? (touchR ? 1 : 0) : 0);
    const int dy = sy < 0 ? (touchT ? -1 : 0) : (sy > 0 ? (touchB ? 1 : 0) : 0);
    const mask_t X = dx < 0 ? colL : (dx > 0 ? colR : (mask_t)0);     // samples whose neighbour lies in another CTB column
    const mask_t Y = dy < 0 ? rowT : (dy > 0 ? rowB : (mask_t)0);     // ... row
    mask_t bad = 0;
    if (!permitted(dx, 0)) bad |= X & ~Y;
    if (!permitted(0, dy)) bad |= Y & ~X;
    if (!permitted(dx, dy)) bad |= X & Y;
    if (!permitted(0, 0)) bad |= onEdge & ~(X | Y);           // (chroma only: the CTB itself as "another slice")
    return bad;
  };
  mask_t okmask = ~(blocked(hx, hy) | blocked(-hx, -hy));
  okmask &= (COL0 * (mask_t)((1u << ncols) - 1u)) & (nrows >= (int)sizeof(mask_t) ? ~(mask_t)0 : (((mask_t)1 << (8 * nrows)) - 1));
  if (exm) {
#pragma unroll
    for (int r = 0; r < SAO_ROWS; r++)
#pragma unroll
      for (int i = 0; i < 8; i++)
        if ((exm >> ((cs ? r >> 1 : r >> 2) * 4 + (cs ? i >> 1 : i >> 2))) & 1) okmask &= ~((mask_t)1 << (r * 8 + i));
  }
  const unsigned ok_lo = (unsigned)okmask, ok_hi = (unsigned)((unsigned long long)okmask >> 32);

  // offsets {o1, o2, 0, o3, o4} indexed by edgeIdx + 2 (sao.cc:95-100) as a byte table, biased by 128 (v_perm looks up
  // unsigned bytes): bytes 0..3 in tab_lo, byte 4 in tab_hi
  const uint32_t tab_lo = (uint32_t)(uint8_t)(o4[0] + 128) | ((uint32_t)(uint8_t)(o4[1] + 128) << 8) | (128u << 16) |
                          ((uint32_t)(uint8_t)(o4[2] + 128) << 24);
  const uint32_t tab_hi = (uint32_t)(uint8_t)(o4[3] + 128);
  typedef short sao_s2 __attribute__((ext_vector_type(2)));
  const sao_s2 one = { 1, 1 }, mone = { -1, -1 }, two = { 2, 2 }, zero = { 0, 0 }, bias = { 128, 128 };
  const sao_s2 vmax = { (short)maxv, (short)maxv };
#pragma unroll
  for (int r = 0; r < SAO_ROWS; r++) {
    if (r >= nrows) break;
    const uint32_t cur[4] = { R[r + 1].x, R[r + 1].y, R[r + 1].z, R[r + 1].w };
    const uint32_t up[4] = { R[r].x, R[r].y, R[r].z, R[r].w }, dn[4] = { R[r + 2].x, R[r + 2].y, R[r + 2].z, R[r + 2].w };
    uint32_t outw[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      // the two neighbours of the sample pair (2m, 2m+1) by edge class: 0 left/right, 1 up/down, 2 up-left/down-right,
      // 3 up-right/down-left
      const uint32_t na = e0 ? S[r + 1][m] : (e1 ? up[m] : (e2 ? S[r][m] : S[r][m + 1]));
      const uint32_t nb = e0 ? S[r + 1][m + 1] : (e1 ? dn[m] : (e2 ? S[r + 2][m + 1] : S[r + 2][m]));
      const sao_s2 c = __builtin_bit_cast(sao_s2, cur[m]);
      const sao_s2 sa = __builtin_elementwise_min(__builtin_elementwise_max(c - __builtin_bit_cast(sao_s2, na), mone), one);
      const sao_s2 sb = __builtin_elementwise_min(__builtin_elementwise_max(c - __builtin_bit_cast(sao_s2, nb), mone), one);
      const uint32_t ee = __builtin_bit_cast(uint32_t, (sao_s2)(sa + sb + two));       // 0..4 in each half
      const uint32_t t = __builtin_amdgcn_perm(tab_hi, tab_lo, ee | 0x0C000C00u);      // biased offsets, zero-extended
      sao_s2 o = c + __builtin_bit_cast(sao_s2, t) - bias;
      o = __builtin_elementwise_min(__builtin_elementwise_max(o, zero), vmax);
      const unsigned okw = r < 4 ? ok_lo : ok_hi;
      const int pos = (r & 3) * 8 + 2 * m;
      const uint32_t mlo = (uint32_t)(((int)(okw << (31 - pos))) >> 31), mhi = (uint32_t)(((int)(okw << (30 - pos))) >> 31);
      const uint32_t msk = (mlo & 0xFFFFu) | (mhi & 0xFFFF0000u);
      outw[m] = (__builtin_bit_cast(uint32_t, o) & msk) | (cur[m] & ~msk);
    }
    store8_pk<PX>(dst + x0 + (y0 + r) * dstride, make_uint4(outw[0], outw[1], outw[2], outw[3]));
  }
}

template <typename PX>
__global__ __launch_bounds__(256, SAO_WAVES)
void k_sao(PicDev P, PlaneRef s0, PlaneRef s1, PlaneRef s2, PlaneRef d0, PlaneRef d1, PlaneRef d2,
           SaoMeta M)
{
  const int comp = blockIdx.z;
  const int cs = comp ? 1 : 0;
  const int width = P.width >> cs, height = P.height >> cs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y0 = blockIdx.y * SAO_ROWS;
  if (y0 >= height) return;                                   // uniform per workgroup
  const PlaneRef sp = comp == 0 ? s0 : (comp == 1 ? s1 : s2);
  const PlaneRef dp = comp == 0 ? d0 : (comp == 1 ? d1 : d2);
  const PX* src = (const PX*)sp.ptr;
  PX* dst = (PX*)dp.ptr;
  const int sstride = sp.stride, dstride = dp.stride;
  const int ctbshift = P.log2_ctb - cs;
  const int ctbY = y0 >> ctbshift;
  // A wavefront takes SAO_GROUPS groups of 62 strips one after the other, with the loads of all of them issued up front:
  // the second group's rows arrive while the first is computed, the first group's stores drain while the second is
  // computed (with one group per wavefront all wavefronts of the picture load, compute and store in step).
  // Every load is unconditional (the address is clamped into the picture instead): loads under a condition are
  // compiled into one branch + wait each, i.e. one memory latency per row.  What a clamped load returns for a row or
  // strip outside the picture is never used: such neighbours are masked out (okmask), such strips return.
  // The rows stay packed: 8 samples = four dwords of two 16-bit samples (8-bit pictures are widened on load).  The edge
  // class arithmetic runs on sample pairs (v_pk_* 16-bit instructions, the offset table as a byte lookup with v_perm).
  int x0[SAO_GROUPS]; bool inpic[SAO_GROUPS];
  SaoRec w[SAO_GROUPS];
  SaoRows R[SAO_GROUPS];
#pragma unroll
  for (int g = 0; g < SAO_GROUPS; g++) {
    x0[g] = (((blockIdx.x * 4 + wave) * SAO_GROUPS + g) * SAO_LANES + lane - 1) * 8;
    inpic[g] = x0[g] >= 0 && x0[g] < width;
    // the CTB record is requested together with the samples (one latency, not two), as three 8-byte words: its byte
    // fields are picked out with shifts (indexing the struct by component sent it through LDS)
    const int ctbX = min(max(x0[g], 0) >> ctbshift, P.ctbs_w - 1);
    const uint2* q = reinterpret_cast<const uint2*>(&M.sao[ctbX + ctbY * P.ctbs_w]);
    const uint2 q0 = q[0], q1 = q[1], q2 = q[2];
    w[g].w[0] = q0.x; w[g].w[1] = q0.y; w[g].w[2] = q1.x; w[g].w[3] = q1.y; w[g].w[4] = q2.x; w[g].w[5] = q2.y;
  }
#pragma unroll
  for (int g = 0; g < SAO_GROUPS; g++) {
    const PX* col = src + min(max(x0[g], 0), (width - 1) & ~7);      // (the last strip may be partial: rows are padded)
#pragma unroll
    for (int j = 0; j < SAO_ROWS + 2; j++) R[g].r[j] = load8_pk<PX>(col + min(max(y0 - 1 + j, 0), height - 1) * sstride);
  }
#pragma unroll
  for (int g = 0; g < SAO_GROUPS; g++)
    sao_strip<PX, true>(P, M, dst, dstride, comp, cs, width, height, lane, x0[g], y0, inpic[g], ctbshift, w[g], R[g], SaoNb());
}

// CTB-local lane mapping (default; k_sao above stays as the parity variant DE265HIP_SAO_STRIPS=1).  k_sao lays the 62 useful
// lanes of a wavefront along one row pair, i.e. across eight 64-sample CTBs: SaoTypeIdx / the edge class differ from CTB to CTB,
// so every wavefront executes the copy, the band AND the edge path (377 VALU instructions per wavefront, tools/exp/pmc_insts.sh:
// the kernel is bound by instruction issue, not by HBM).  Here a wavefront is a tile SW strips wide and 64/SW row pairs high with
// SW = min(8, CTB width / 8): 64x16 luma / 32x32 chroma samples of ONE CTB for 64x64 CTBs (one CTB, 32x32, for 32x32 luma CTBs), so the
// type is wavefront-uniform and only one of the three paths is executed.  Left / right neighbours come from the adjacent lanes
// by DPP row shifts (SW divides 16: lane-1 / lane+1 of an inner strip lie in the same DPP row); the strips on the tile's left and
// right edge fetch the one sample beyond it themselves: one divergent block of four 2-byte loads, issued together with the rows.
template <typename PX>
__global__ __launch_bounds__(256, SAO_WAVES)
void k_sao_ctb(PicDev P, PlaneRef s0, PlaneRef s1, PlaneRef s2, PlaneRef d0, PlaneRef d1, PlaneRef d2, SaoMeta M, uint3 G)
{
  // (XCD-aware: the one sample a tile reads beyond its left / right edge sits in the neighbouring tile's cache line; with the
  //  plain round-robin dispatch that line went into a second L2: the kernel fetched 3.3x the picture)
  const XcdBlk B = xcd_block(G.x, G.y, G.z);
  if (!B.ok) return;
  // (4:4:4: the chroma planes have luma's geometry - the same strips, only the parameters are the component's)
  const int comp = B.z, cs = (comp && P.chroma_format != 3) ? 1 : 0;
  const int width = P.width >> cs, height = P.height >> cs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ctbshift = P.log2_ctb - cs;
  const int lsw = min(3, ctbshift - 3);                         // log2 strips per tile row (CTB 16 chroma: one strip)
  const int sx = lane & ((1 << lsw) - 1), sy = lane >> lsw;
  const int tw = 8 << lsw, th = (64 >> lsw) * SAO_ROWS;         // tile size in samples
  const int x0 = B.x * tw + sx * 8;
  const int y0 = (B.y * (int)(blockDim.x >> 6) + wave) * th + sy * SAO_ROWS;      // (a workgroup's wavefronts: tile rows below each other; G.y counts workgroups)
  const PlaneRef sp = comp == 0 ? s0 : (comp == 1 ? s1 : s2);
  const PlaneRef dp = comp == 0 ? d0 : (comp == 1 ? d1 : d2);
  const PX* src = (const PX*)sp.ptr;
  const int sstride = sp.stride;
  const bool inpic = x0 < width && y0 < height;
  const int ctbX = min(x0 >> ctbshift, P.ctbs_w - 1), ctbY = min(y0 >> ctbshift, P.ctbs_h - 1);
  SaoRec w;
  {
    const uint2* q = reinterpret_cast<const uint2*>(&M.sao[ctbX + ctbY * P.ctbs_w]);
    const uint2 q0 = q[0], q1 = q[1], q2 = q[2];
    w.w[0] = q0.x; w.w[1] = q0.y; w.w[2] = q1.x; w.w[3] = q1.y; w.w[4] = q2.x; w.w[5] = q2.y;
  }
  SaoRows R;
  const int xc = min(x0, (width - 1) & ~7);                      // all loads unconditional, addresses clamped (see k_sao)
  int yr[SAO_ROWS + 2];
#pragma unroll
  for (int j = 0; j < SAO_ROWS + 2; j++) { yr[j] = min(max(y0 - 1 + j, 0), height - 1); R.r[j] = load8_pk<PX>(src + xc + yr[j] * sstride); }
  uint32_t halo[SAO_ROWS + 2];
#pragma unroll
  for (int j = 0; j < SAO_ROWS + 2; j++) halo[j] = 0;
  const bool first = sx == 0, last = sx == (1 << lsw) - 1;
  if (first || last) {
    // (a one-strip tile needs both: the right one then goes through a second block; tiles that narrow exist for 16x16 chroma CTBs only)
    const int xh = first ? max(xc - 1, 0) : min(xc + 8, width - 1);
#pragma unroll
    for (int j = 0; j < SAO_ROWS + 2; j++) halo[j] = src[xh + yr[j] * sstride];
  }
  uint32_t halo2[SAO_ROWS + 2];
#pragma unroll
  for (int j = 0; j < SAO_ROWS + 2; j++) halo2[j] = 0;
  if (lsw == 0) {
    const int xh = min(xc + 8, width - 1);
#pragma unroll
    for (int j = 0; j < SAO_ROWS + 2; j++) halo2[j] = src[xh + yr[j] * sstride];
  }
  SaoNb nb;
#pragma unroll
  for (int j = 0; j < SAO_ROWS + 2; j++) {
    const uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)R.r[j].w, 0x111, 0xf, 0xf, false);   // row_shr:1 -> lane-1's (v6, v7)
    const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)R.r[j].x, 0x101, 0xf, 0xf, false);   // row_shl:1 -> lane+1's (v0, v1)
    nb.l[j] = first ? halo[j] : pw >> 16;
    nb.r[j] = last ? (lsw == 0 ? halo2[j] : halo[j]) : nx;
  }
  sao_strip<PX, false>(P, M, (PX*)dp.ptr, dp.stride, comp, cs, width, height, lane, x0, y0, inpic, ctbshift, w, R, nb);
}

// ---------------------------------------------------------------- deblocking + SAO in one pass over the picture
// One workgroup per LF_TW x LF_TH tile of a component: (0) the tile and a ring around it - the 8x8 deblocking blocks that
// overlap it by 4 samples plus what SAO's 3x3 neighbourhood needs - are staged in LDS as 16-bit samples with coalesced
// 16-byte loads; (1) one lane per 8x8 block deblocks it IN LDS (deblock_block: the blocks partition the staged region, so
// nothing is read after a neighbour wrote it); (2) SAO reads the deblocked tile from LDS and writes the OUTPUT picture.
// The input picture is never written, so the ring a neighbouring workgroup recomputes comes out identical.  Against
// k_deblock_fused + k_sao: one launch instead of two, the picture is read once (x1.2 for the ring, mostly L2 hits) and
// written once instead of read twice and written 1.5 times, and SAO's neighbours come from LDS instead of wave shifts.
#define LF_PITCH (LF_TW + 16)                 // tile columns x0-8 .. x0+LF_TW+7: every 8-sample group 16-byte aligned
#define LF_ROWS (LF_TH + 8)                   // tile rows y0-4 .. y0+LF_TH+3
template <typename PX>
__global__ __launch_bounds__(256)
void k_lf_tile(PicDev P, PlaneRef s0, PlaneRef s1, PlaneRef s2, PlaneRef d0, PlaneRef d1, PlaneRef d2, LfMeta LM, SaoMeta SM,
               int do_deblock)
{
  __shared__ __attribute__((aligned(16))) uint16_t tile[LF_ROWS * LF_PITCH];
  const int comp = blockIdx.z, cs = comp ? 1 : 0;
  const int W = P.width >> cs, H = P.height >> cs;
  const int x0 = blockIdx.x * LF_TW, y0 = blockIdx.y * LF_TH;
  if (x0 >= W || y0 >= H) return;                             // uniform per workgroup
  const PlaneRef sp = comp == 0 ? s0 : (comp == 1 ? s1 : s2), dp = comp == 0 ? d0 : (comp == 1 ? d1 : d2);
  const PX* src = (const PX*)sp.ptr;
  const int tid = threadIdx.x;
  // tile_at(x, y) for picture coordinates inside the staged region
  uint16_t* const origin = tile - ((y0 - 4) * LF_PITCH + (x0 - 8));
  // The workgroup's memory round trips are put side by side instead of one behind the other (all ~1 500 workgroups of a
  // 4K picture are resident at once, so the kernel takes as long as ONE workgroup's chain): the tile's loads, the
  // deblocking lanes' first metadata round and the SAO records are issued together; the second metadata round (slice
  // offsets, motion) goes out while the tile is written to LDS.  Loads are unconditional (clamped addresses) - a load
  // under a per-lane condition is compiled into its own branch + wait.
  constexpr int NCH = LF_PITCH / 8, NLD = (LF_ROWS * NCH + 255) / 256;
  const int Wal = (W + 7) & ~7;                                // (rows are padded to 64 samples: a partial last group is readable)
  uint4 tv[NLD];
#pragma unroll
  for (int u = 0; u < NLD; u++) {
    const int i = min(tid + 256 * u, LF_ROWS * NCH - 1);
    const int r = i / NCH, c = i - r * NCH;
    const int gy = min(max(y0 - 4 + r, 0), H - 1), gx = min(max(x0 - 8 + 8 * c, 0), Wal - 8);
    tv[u] = load8_pk<PX>(src + gx + gy * sp.stride);
  }
  constexpr int NBX = LF_TW / 8 + 1, NBY = LF_TH / 8 + 1;
  static_assert(NBX * NBY <= 256, "one deblocking block per lane");
  DeblockLane<uint16_t> DL;
  DL.live = false;
  if (do_deblock && tid < NBX * NBY) DL.meta1(P, LM, comp, (x0 >> 3) + tid % NBX, (y0 >> 3) + tid / NBX);
  const int ctbshift = P.log2_ctb - cs;
  constexpr int SX = LF_TW / 8, SY = LF_TH / SAO_ROWS, NST = (SX * SY + 255) / 256;
  SaoRec rec[NST];
#pragma unroll
  for (int u = 0; u < NST; u++) {
    const int sidx = min(tid + 256 * u, SX * SY - 1);
    const int xs = min(x0 + 8 * (sidx % SX), W - 1), ys = min(y0 + SAO_ROWS * (sidx / SX), H - 1);
    const uint2* q = reinterpret_cast<const uint2*>(&SM.sao[(xs >> ctbshift) + (ys >> ctbshift) * P.ctbs_w]);
    const uint2 q0 = q[0], q1 = q[1], q2 = q[2];
    rec[u].w[0] = q0.x; rec[u].w[1] = q0.y; rec[u].w[2] = q1.x; rec[u].w[3] = q1.y; rec[u].w[4] = q2.x; rec[u].w[5] = q2.y;
  }
  // (0) the tile into LDS (a clamped load's duplicate lands in a cell that is never read unclamped)
#pragma unroll
  for (int u = 0; u < NLD; u++) {
    const int i = tid + 256 * u;
    if (i < LF_ROWS * NCH) {
      const int r = i / NCH, c = i - r * NCH;
      *reinterpret_cast<uint4*>(&tile[r * LF_PITCH + 8 * c]) = tv[u];
    }
  }
  DL.meta2(LM);
  __syncthreads();
  // ---- (1) deblock the blocks that overlap the tile or its ring, in LDS
  if (do_deblock) {
    DL.filter(P, origin, LF_PITCH);
    __syncthreads();
  }
  // ---- (2) SAO from the tile to the output picture, one 8 x SAO_ROWS strip per lane and step
  PX* dst = (PX*)dp.ptr;
#pragma unroll
  for (int u = 0; u < NST; u++) {
    const int sidx = tid + 256 * u;
    const int sy = sidx / SX, sx = sidx - sy * SX;
    const int xs = x0 + 8 * sx, ys = y0 + SAO_ROWS * sy;
    if (sidx >= SX * SY || xs >= W || ys >= H) continue;
    SaoRows R; SaoNb nb;
#pragma unroll
    for (int j = 0; j < SAO_ROWS + 2; j++) {
      const int yy = min(max(ys - 1 + j, 0), H - 1);            // (clamped like k_sao: such neighbours are masked out)
      const uint16_t* row = origin + yy * LF_PITCH;
      R.r[j] = *reinterpret_cast<const uint4*>(row + xs);
      nb.l[j] = row[max(xs - 1, 0)]; nb.r[j] = row[min(xs + 8, Wal - 1)];
    }
    sao_strip<PX, false>(P, SM, dst, dp.stride, comp, cs, W, H, 1, xs, ys, true, ctbshift, rec[u], R, nb);
  }
}
template __global__ void k_lf_tile<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, LfMeta, SaoMeta, int);
template __global__ void k_lf_tile<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, LfMeta, SaoMeta, int);

template __global__ void k_sao<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, SaoMeta);
template __global__ void k_sao<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, SaoMeta);
template __global__ void k_sao_ctb<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, SaoMeta, uint3);
template __global__ void k_sao_ctb<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, SaoMeta, uint3);

}  // namespace d265
