// k_mc.hip -- inter prediction kernel for gfx950: luma 8-tap quarter-pel and
// chroma 4-tap eighth-pel interpolation with picture-border clamping, fused with
// the (un)weighted / bi-predictive sample write.
//
// One 64-lane wavefront (= one workgroup) produces one MC task: a <=16x16 luma
// tile of a PU and its two <=8x8 chroma tiles.  The reference block (+ filter
// margins) is staged through LDS once per list; the horizontal pass result is
// kept in LDS as int16 exactly like the reference's mcbuffer.  Behaviour follows
// (libde265/):
//   motion.cc:50-170 mc_luma, :175-273 mc_chroma, :279-660 generate_inter_prediction_samples
//   fallback-motion.cc:423-648 put_qpel_*, :257-419 put_epel_*, :33-251 put_*_pred
#include "kernels.h"

namespace d265 {

__device__ __constant__ int8_t c_qpel_filt[4][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 },
  { -1, 4, -10, 58, 17, -5, 1, 0 },
  { -1, 4, -11, 40, 40, -11, 4, -1 },
  { 0, 1, -5, 17, 58, -10, 4, -1 } };
__device__ __constant__ int8_t c_epel_filt[8][4] = {
  { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
  { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

#define MC_IWP 24          // LDS pitch of the staged input tile (>= 16+7)

__device__ __forceinline__ int mc_clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// Interpolates a w x h block (w,h <= 16 for NT=8, <= 8 for NT=4) of one reference
// plane into out[] (sample s = lane + 64*k, row-major over w).  Intermediates are
// truncated to int16 after each stage (fallback-motion.cc:346,:377,:508-545).
template <typename PX, int NT>
__device__ void mc_block(const PX* __restrict__ ref, int rstride, int picW, int picH,
                         int xInt, int yInt, int xF, int yF, int w, int h, int bd,
                         uint16_t* in, int16_t* tmp, int lane, int16_t* out)
{
  constexpr int before = NT == 8 ? 3 : 1;
  constexpr int KMAX = NT == 8 ? 4 : 1;     // outputs per lane
  const int IW = w + NT - 1, IH = h + NT - 1;
  for (int idx = lane; idx < IW * IH; idx += 64) {
    int r = idx / IW, c = idx - r * IW;
    int xA = mc_clip3(0, picW - 1, xInt - before + c);
    int yA = mc_clip3(0, picH - 1, yInt - before + r);
    in[r * MC_IWP + c] = ref[xA + yA * rstride];
  }
  __syncthreads();
  const int shift1 = bd - 8;
  const int nOut = w * h;
  if (xF == 0 && yF == 0) {
    const int shift3 = 14 - bd;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
      int s = lane + 64 * k;
      if (s < nOut) {
        int y = s / w, x = s - y * w;
        out[k] = (int16_t)(in[(y + before) * MC_IWP + x + before] << shift3);
      }
    }
    __syncthreads();
    return;
  }
  for (int idx = lane; idx < IH * w; idx += 64) {
    int r = idx / w, x = idx - r * w;
    int v;
    if (xF == 0) v = in[r * MC_IWP + x + before];
    else {
      int sum = 0;
#pragma unroll
      for (int k = 0; k < NT; k++) {
        int tap = NT == 8 ? c_qpel_filt[xF][k] : c_epel_filt[xF][k];
        sum += tap * (int)in[r * MC_IWP + x + k];
      }
      v = sum >> shift1;
    }
    tmp[r * 16 + x] = (int16_t)v;
  }
  __syncthreads();
  const int vshift = (xF == 0) ? shift1 : 6;
#pragma unroll
  for (int k = 0; k < KMAX; k++) {
    int s = lane + 64 * k;
    if (s < nOut) {
      int y = s / w, x = s - y * w;
      int v;
      if (yF == 0) v = tmp[(y + before) * 16 + x];
      else {
        int sum = 0;
#pragma unroll
        for (int j = 0; j < NT; j++) {
          int tap = NT == 8 ? c_qpel_filt[yF][j] : c_epel_filt[yF][j];
          sum += tap * (int)tmp[(y + j) * 16 + x];
        }
        v = sum >> vshift;
      }
      out[k] = (int16_t)v;
    }
  }
  __syncthreads();
}

// weighted sample prediction (8.5.3.2.3; fallback-motion.cc:33-251)
// mode: 0 unweighted, 1 weighted uni, 2 avg, 3 weighted bi
template <typename PX>
__device__ __forceinline__ PX mc_combine(int mode, int a, int b, int bd, int w0, int o0, int w1, int o1,
                                         int log2WD)
{
  const int maxv = (1 << bd) - 1;
  int v;
  switch (mode) {
    case 0: { int sh = 14 - bd; int off = sh > 0 ? 1 << (sh - 1) : 0; v = (a + off) >> sh; break; }
    case 1: v = ((a * w0 + (1 << (log2WD - 1))) >> log2WD) + o0; break;
    case 2: { int sh = 15 - bd; v = (a + b + (1 << (sh - 1))) >> sh; break; }
    default: v = (a * w0 + b * w1 + ((o0 + o1 + 1) << log2WD)) >> (log2WD + 1); break;
  }
  return (PX)mc_clip3(0, maxv, v);
}

template <typename PX>
__global__ __launch_bounds__(64)
void k_mc(PicDev P, DpbTable dpb, PlaneRef d0, PlaneRef d1, PlaneRef d2,
          const McTask* __restrict__ tasks, const de265hip_slice_params* __restrict__ slices)
{
  __shared__ uint16_t s_in[23 * MC_IWP];
  __shared__ int16_t s_tmp[23 * 16];
  const int lane = threadIdx.x;
  const McTask t = tasks[blockIdx.x];
  const de265hip_slice_params* sh = &slices[t.slice_idx];
  const bool use0 = t.slot[0] >= 0, use1 = t.slot[1] >= 0;
  const bool bi = use0 && use1;
  const int l_uni = use0 ? 0 : 1;

  // prediction mode (motion.cc:440-620)
  int mode;
  if (sh->slice_type == 1) mode = P.weighted_pred ? 1 : 0;
  else if (bi) mode = P.weighted_bipred ? 3 : 2;
  else mode = P.weighted_bipred ? 1 : 0;

  const PlaneRef dsts[3] = { d0, d1, d2 };
#pragma unroll
  for (int comp = 0; comp < 3; comp++) {
    const int bd = comp ? P.bd_chroma : P.bd_luma;
    const int w = comp ? t.w >> 1 : t.w, h = comp ? t.h >> 1 : t.h;
    const int x0 = comp ? t.x >> 1 : t.x, y0 = comp ? t.y >> 1 : t.y;
    const int picW = comp ? P.width >> 1 : P.width, picH = comp ? P.height >> 1 : P.height;
    int16_t pr[2][4];
#pragma unroll
    for (int l = 0; l < 2; l++) {
      if (t.slot[l] < 0) continue;
      const PlaneRef r = dpb.p[t.slot[l]][comp];
      const int mvx = t.mv[l][0], mvy = t.mv[l][1];
      if (comp == 0)
        mc_block<PX, 8>((const PX*)r.ptr, r.stride, picW, picH, x0 + (mvx >> 2), y0 + (mvy >> 2),
                        mvx & 3, mvy & 3, w, h, bd, s_in, s_tmp, lane, pr[l]);
      else
        mc_block<PX, 4>((const PX*)r.ptr, r.stride, picW, picH, x0 + (mvx >> 3), y0 + (mvy >> 3),
                        mvx & 7, mvy & 7, w, h, bd, s_in, s_tmp, lane, pr[l]);
    }
    // weights (motion.cc:403-406, :464-473, :522-540)
    int w0 = 0, o0 = 0, w1 = 0, o1 = 0, log2WD = 1;
    if (mode == 1 || mode == 3) {
      const int shift1 = max(2, 14 - bd);
      const int offsh = bd - 8;                              // WpOffsetBdShift (sps.cc:554-563)
      const int la = (mode == 3) ? 0 : l_uni;
      const int ra = t.ref_idx[la];
      if (comp == 0) {
        log2WD = sh->luma_log2_weight_denom + shift1;
        w0 = sh->luma_weight[la][ra]; o0 = sh->luma_offset[la][ra] * (1 << offsh);
        if (mode == 3) { int rb = t.ref_idx[1]; w1 = sh->luma_weight[1][rb]; o1 = sh->luma_offset[1][rb] * (1 << offsh); }
      } else {
        log2WD = sh->chroma_log2_weight_denom + shift1;
        w0 = sh->chroma_weight[la][ra][comp - 1]; o0 = sh->chroma_offset[la][ra][comp - 1] * (1 << offsh);
        if (mode == 3) { int rb = t.ref_idx[1]; w1 = sh->chroma_weight[1][rb][comp - 1]; o1 = sh->chroma_offset[1][rb][comp - 1] * (1 << offsh); }
      }
    }
    PX* dst = (PX*)dsts[comp].ptr;
    const int dstride = dsts[comp].stride;
    const int nOut = w * h;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int s = lane + 64 * k;
      if (s < nOut) {
        int y = s / w, x = s - y * w;
        int a = bi ? pr[0][k] : (use0 ? pr[0][k] : pr[1][k]);
        int b = pr[1][k];
        dst[(x0 + x) + (y0 + y) * dstride] = mc_combine<PX>(mode, a, b, bd, w0, o0, w1, o1, log2WD);
      }
    }
  }
}

template __global__ void k_mc<uint8_t>(PicDev, DpbTable, PlaneRef, PlaneRef, PlaneRef, const McTask*,
                                       const de265hip_slice_params*);
template __global__ void k_mc<uint16_t>(PicDev, DpbTable, PlaneRef, PlaneRef, PlaneRef, const McTask*,
                                        const de265hip_slice_params*);

// ---- PCM sample copy (slice.cc:4143-4183), one workgroup per PCM CU
template <typename PX>
__global__ __launch_bounds__(256)
void k_pcm(PlaneRef d0, PlaneRef d1, PlaneRef d2, const PcmTask* __restrict__ tasks,
           const uint16_t* __restrict__ samples)
{
  const PcmTask t = tasks[blockIdx.x];
  const int n = 1 << t.log2_cb_size;
  const uint16_t* s = samples + t.sample_offset;
  const PlaneRef dsts[3] = { d0, d1, d2 };
  for (int comp = 0; comp < 3; comp++) {
    const int w = comp ? n >> 1 : n;
    const int x0 = comp ? t.x0 >> 1 : t.x0, y0 = comp ? t.y0 >> 1 : t.y0;
    PX* dst = (PX*)dsts[comp].ptr;
    for (int i = threadIdx.x; i < w * w; i += 256) {
      int y = i / w, x = i - y * w;
      dst[(x0 + x) + (y0 + y) * dsts[comp].stride] = (PX)s[i];
    }
    s += w * w;
  }
}
template __global__ void k_pcm<uint8_t>(PlaneRef, PlaneRef, PlaneRef, const PcmTask*, const uint16_t*);
template __global__ void k_pcm<uint16_t>(PlaneRef, PlaneRef, PlaneRef, const PcmTask*, const uint16_t*);

// ---- function-level forms (acceleration.h:31-120 slot semantics) ----
// put_hevc_qpel_* / put_hevc_epel_*: one workgroup per block; blocks wider/higher
// than one tile are walked in <=16x16 (luma) / <=8x8 (chroma) tiles.
template <typename PX>
__global__ __launch_bounds__(64)
void k_fn_interp(int luma, int bit_depth, const PX* __restrict__ plane, int stride, int pw, int ph,
                 int w, int h, int fx, int fy, const int32_t* __restrict__ xy, int16_t* __restrict__ out)
{
  __shared__ uint16_t s_in[23 * MC_IWP];
  __shared__ int16_t s_tmp[23 * 16];
  const int lane = threadIdx.x;
  const int bx = xy[2 * blockIdx.x], by = xy[2 * blockIdx.x + 1];
  int16_t* o = out + (size_t)blockIdx.x * w * h;
  const int T = luma ? 16 : 8;
  for (int ty = 0; ty < h; ty += T)
    for (int tx = 0; tx < w; tx += T) {
      const int tw = min(T, w - tx), th = min(T, h - ty);
      int16_t pr[4];
      if (luma) mc_block<PX, 8>(plane, stride, pw, ph, bx + tx, by + ty, fx, fy, tw, th, bit_depth, s_in, s_tmp, lane, pr);
      else      mc_block<PX, 4>(plane, stride, pw, ph, bx + tx, by + ty, fx, fy, tw, th, bit_depth, s_in, s_tmp, lane, pr);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int s = lane + 64 * k;
        if (s < tw * th) { int y = s / tw, x = s - y * tw; o[(tx + x) + (ty + y) * w] = pr[k]; }
      }
    }
}
template __global__ void k_fn_interp<uint8_t>(int, int, const uint8_t*, int, int, int, int, int, int, int, const int32_t*, int16_t*);
template __global__ void k_fn_interp<uint16_t>(int, int, const uint16_t*, int, int, int, int, int, int, int, const int32_t*, int16_t*);

// put_unweighted_pred / put_weighted_pred / put_weighted_pred_avg / put_weighted_bipred
template <typename PX>
__global__ __launch_bounds__(64)
void k_fn_put(int mode, int bit_depth, PX* plane, int stride, int w, int h,
              const int32_t* __restrict__ xy, const int16_t* __restrict__ s0, const int16_t* __restrict__ s1,
              int w0, int o0, int w1, int o1, int log2wd)
{
  const int bx = xy[2 * blockIdx.x], by = xy[2 * blockIdx.x + 1];
  const size_t base = (size_t)blockIdx.x * w * h;
  for (int s = threadIdx.x; s < w * h; s += 64) {
    int y = s / w, x = s - y * w;
    int a = s0[base + s], b = s1 ? s1[base + s] : 0;
    plane[(bx + x) + (by + y) * stride] = mc_combine<PX>(mode, a, b, bit_depth, w0, o0, w1, o1, log2wd);
  }
}
template __global__ void k_fn_put<uint8_t>(int, int, uint8_t*, int, int, int, const int32_t*, const int16_t*, const int16_t*, int, int, int, int, int);
template __global__ void k_fn_put<uint16_t>(int, int, uint16_t*, int, int, int, const int32_t*, const int16_t*, const int16_t*, int, int, int, int, int);

}  // namespace d265
