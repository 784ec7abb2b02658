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

// the luma taps as int16 pairs (tap[2m], tap[2m+1]) for v_dot2_i32_i16: two multiply-adds per instruction on sample pairs
#define QPK(a, b) ((uint32_t)(uint16_t)(int16_t)(a) | ((uint32_t)(uint16_t)(int16_t)(b) << 16))
__device__ __constant__ uint32_t c_qpel_pk[4][4] = {
  { QPK(0, 0), QPK(0, 64), QPK(0, 0), QPK(0, 0) },
  { QPK(-1, 4), QPK(-10, 58), QPK(17, -5), QPK(1, 0) },
  { QPK(-1, 4), QPK(-11, 40), QPK(40, -11), QPK(4, -1) },
  { QPK(0, 1), QPK(-5, 17), QPK(58, -10), QPK(4, -1) } };
// the chroma taps the same way: (tap0, tap1), (tap2, tap3)
__device__ __constant__ uint32_t c_epel_pk[8][2] = {
  { QPK(0, 64), QPK(0, 0) }, { QPK(-2, 58), QPK(10, -2) }, { QPK(-4, 54), QPK(16, -2) }, { QPK(-6, 46), QPK(28, -4) },
  { QPK(-4, 36), QPK(36, -4) }, { QPK(-4, 28), QPK(46, -6) }, { QPK(-2, 16), QPK(54, -4) }, { QPK(-2, 10), QPK(58, -2) } };
#undef QPK
typedef short mc_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int mc_dot2(uint32_t a, uint32_t b, int acc)
{ return __builtin_amdgcn_sdot2(__builtin_bit_cast(mc_s2, a), __builtin_bit_cast(mc_s2, b), acc, false); }

#define MC_IWP 24          // LDS pitch of the staged input tile (>= 16+7)

__device__ __forceinline__ int mc_clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// Interpolates a w x h block (w,h <= 16 for NT=8, <= 8 for NT=4) of one reference
// plane into out[] (sample s = lane + 64*k, row-major over w).  Intermediates are
// truncated to int16 after each stage (fallback-motion.cc:346,:377,:508-545).
template <typename PX, int NT, int KMAX = (NT == 8 ? 4 : 1)>     // KMAX: outputs per lane (chroma blocks beyond 8x8: 4)
__device__ void mc_block(const PX* __restrict__ ref, int rstride, int picW, int picH,
                         int xInt, int yInt, int xF, int yF, int w, int h, int bd,
                         uint16_t* in, int16_t* tmp, int lane, int16_t* out)
{
  constexpr int before = NT == 8 ? 3 : 1;
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

// (Tried: a second kernel for tiles of at most 8x8 -- 42 % of the tasks, 12 % of the samples of the synthetic B
//  pictures -- with four tasks per wavefront, one per 16-lane group, every task quantity per lane and both filter passes
//  always run (fraction-0 taps {0,0,0,64,..} are bit-exact).  Bit-exact, but 71-79 us instead of 62 us per 4K B picture:
//  without the scalar registers and wave-uniform shortcuts a packed wavefront costs about twice a plain one.)
// ---------------------------------------------------------------- picture-level MC kernel
// One wavefront per MC task.  All reference fetches of the task (2 lists x 3 planes) are
// issued back to back as aligned 4-sample vector loads into LDS (one exposed HBM/L2 latency
// instead of six); tasks that touch the picture border use per-sample clamped loads into the
// same LDS layout.  Each lane then produces 4 horizontally adjacent luma samples (2 chroma)
// so that filters slide over registers and the final store is one 8-byte access per lane.
#define MCL_P 36                       // luma input pitch: 28 used; 72-byte rows are 8-byte aligned and spread over banks
#define MCC_P 20                       // chroma input pitch: 16 used
#define MCT_P 20                       // pitch of the horizontal-pass buffer (16 used)

template <typename PX> __device__ __forceinline__ uint2 ld4_u16(const PX* p);
template <> __device__ __forceinline__ uint2 ld4_u16<uint16_t>(const uint16_t* p) { return *reinterpret_cast<const uint2*>(p); }
template <> __device__ __forceinline__ uint2 ld4_u16<uint8_t>(const uint8_t* p)
{
  uint32_t r = *reinterpret_cast<const uint32_t*>(p);
  return make_uint2((r & 0xFF) | ((r & 0xFF00) << 8), ((r >> 16) & 0xFF) | ((r >> 24) << 16));
}
template <typename PX> __device__ __forceinline__ void st4_px(PX* p, const int v[4]);
template <> __device__ __forceinline__ void st4_px<uint16_t>(uint16_t* p, const int v[4])
{ *reinterpret_cast<uint2*>(p) = make_uint2((uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16)); }
template <> __device__ __forceinline__ void st4_px<uint8_t>(uint8_t* p, const int v[4])
{ *reinterpret_cast<uint32_t*>(p) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24); }
template <typename PX> __device__ __forceinline__ void st2_px(PX* p, int a, int b);
template <> __device__ __forceinline__ void st2_px<uint16_t>(uint16_t* p, int a, int b)
{ *reinterpret_cast<uint32_t*>(p) = (uint32_t)a | ((uint32_t)b << 16); }
template <> __device__ __forceinline__ void st2_px<uint8_t>(uint8_t* p, int a, int b)
{ *reinterpret_cast<uint16_t*>(p) = (uint16_t)(a | (b << 8)); }

#define MC_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")   // one wave per workgroup

template <typename PX>
__global__ __launch_bounds__(64)
void k_mc(PicDev P, DpbTable dpb, PlaneRef d0, PlaneRef d1, PlaneRef d2,
          const McTask* __restrict__ tasks, const de265hip_slice_params* __restrict__ slices, int n_tasks)
{
  __shared__ __attribute__((aligned(16))) uint16_t s_inL[2][23 * MCL_P];
  __shared__ __attribute__((aligned(16))) uint16_t s_inC[2][2][11 * MCC_P];
  __shared__ __attribute__((aligned(16))) int16_t s_tmp[23 * MCT_P];
  const int lane = threadIdx.x;
  // XCD-aware mapping: workgroups b and b+8 share an XCD (and its L2); give every XCD one contiguous
  // eighth of the task list (tasks are in decode order, i.e. spatial neighbours) so that the
  // overlapping filter margins of neighbouring tiles hit in the same L2.  Speed only, never correctness.
  const int per = (n_tasks + 7) >> 3;
  const int tix = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tix >= n_tasks) return;
  McTask t;                                         // (five dwords at a uniform address: scalar loads)
  {
    const uint32_t* tq = reinterpret_cast<const uint32_t*>(tasks + tix);
    uint32_t tw[5];
#pragma unroll
    for (int i = 0; i < 5; i++) tw[i] = __builtin_amdgcn_readfirstlane(tq[i]);
    __builtin_memcpy(&t, tw, sizeof(t));
  }
  const de265hip_slice_params* sh = &slices[t.slice_idx];
  const bool use0 = t.slot[0] >= 0, use1 = t.slot[1] >= 0;
  const bool bi = use0 && use1;
  const int l_uni = use0 ? 0 : 1;
  const int w = t.w, h = t.h, wc = w >> 1, hc = h >> 1;
  const int cW = P.width >> 1, cH = P.height >> 1;
  const bool c420 = P.chroma_format == 1;           // (4:2:2 / 4:4:4: the chroma planes are predicted by k_mc_chroma_any)

  int mode;                                         // motion.cc:440-620
  if (sh->slice_type == 1) mode = P.weighted_pred ? 1 : 0;
  else if (bi) mode = P.weighted_bipred ? 3 : 2;
  else mode = P.weighted_bipred ? 1 : 0;

  // ---------------- fetch.  First every aligned vector load of both lists and all three planes is issued into
  // registers, unconditionally (row / chunk clamped into the block: a load under a per-lane condition is compiled into
  // its own branch + wait, i.e. one memory latency each - measured: the picture's MC pass 62 -> XX us); then they are
  // stored to LDS.  Blocks that touch the picture border (few) are fetched sample by sample with clamped coordinates.
  int oxL[2] = { 0, 0 }, oxC[2] = { 0, 0 };
  bool insL[2] = { false, false }, insC[2] = { false, false };
  uint2 rl[2][3], rc[2][2];
#pragma unroll
  for (int l = 0; l < 2; l++) {
    if (t.slot[l] < 0) continue;
    const int mvx = t.mv[l][0], mvy = t.mv[l][1];
    {   // luma: rows yI-3 .. yI+h+3, columns xI-3 .. xI+w+3
      const PlaneRef r = dpb.p[t.slot[l]][0];
      const PX* ref = (const PX*)r.ptr;
      const int xs = t.x + (mvx >> 2) - 3, ys = t.y + (mvy >> 2) - 3;
      const int nrow = h + 7, ncol = w + 7;
      insL[l] = xs >= 0 && ys >= 0 && xs + ncol <= P.width && ys + nrow <= P.height;
      if (insL[l]) {
        const int ax = xs & ~3;
        oxL[l] = xs & 3;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const int slot = lane + 64 * i, rr = min(slot >> 3, nrow - 1), ch = min(slot & 7, 6);
          rl[l][i] = ld4_u16<PX>(ref + ax + 4 * ch + (ys + rr) * r.stride);
        }
      }
    }
    if (c420) {   // chroma: rows yI-1 .. yI+hc+1, columns xI-1 .. xI+wc+1
      const int xs = (t.x >> 1) + (mvx >> 3) - 1, ys = (t.y >> 1) + (mvy >> 3) - 1;
      const int nrow = hc + 3, ncol = wc + 3;
      insC[l] = xs >= 0 && ys >= 0 && xs + ncol <= cW && ys + nrow <= cH;
      if (insC[l]) {
        const int ax = xs & ~3;
        oxC[l] = xs & 3;
        const int rr = min(lane >> 2, nrow - 1), ch = lane & 3;
#pragma unroll
        for (int cp = 0; cp < 2; cp++) {
          const PlaneRef r = dpb.p[t.slot[l]][cp + 1];
          rc[l][cp] = ld4_u16<PX>((const PX*)r.ptr + ax + 4 * ch + (ys + rr) * r.stride);
        }
      }
    }
  }
#pragma unroll
  for (int l = 0; l < 2; l++) {
    if (t.slot[l] < 0) continue;
    const int mvx = t.mv[l][0], mvy = t.mv[l][1];
    {
      const int nrow = h + 7, ncol = w + 7;
      if (insL[l]) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const int slot = lane + 64 * i, rr = slot >> 3, ch = slot & 7;
          if (rr < nrow && ch < 7) *reinterpret_cast<uint2*>(&s_inL[l][rr * MCL_P + 4 * ch]) = rl[l][i];
        }
      } else {
        const PlaneRef r = dpb.p[t.slot[l]][0];
        const PX* ref = (const PX*)r.ptr;
        const int xs = t.x + (mvx >> 2) - 3, ys = t.y + (mvy >> 2) - 3;
        for (int idx = lane; idx < nrow * 32; idx += 64) {
          int rr = idx >> 5, c = idx & 31;   // 32 > w + 7
          if (c < ncol) {
            int xA = mc_clip3(0, P.width - 1, xs + c), yA = mc_clip3(0, P.height - 1, ys + rr);
            s_inL[l][rr * MCL_P + c] = ref[xA + yA * r.stride];
          }
        }
      }
    }
    if (c420)
#pragma unroll
    for (int cp = 0; cp < 2; cp++) {
      const int nrow = hc + 3, ncol = wc + 3;
      if (insC[l]) {
        const int rr = lane >> 2, ch = lane & 3;
        if (rr < nrow) *reinterpret_cast<uint2*>(&s_inC[l][cp][rr * MCC_P + 4 * ch]) = rc[l][cp];
      } else {
        const PlaneRef r = dpb.p[t.slot[l]][cp + 1];
        const PX* ref = (const PX*)r.ptr;
        const int xs = (t.x >> 1) + (mvx >> 3) - 1, ys = (t.y >> 1) + (mvy >> 3) - 1;
        for (int idx = lane; idx < nrow * 16; idx += 64) {
          int rr = idx >> 4, c = idx & 15;
          if (c < ncol) {
            int xA = mc_clip3(0, cW - 1, xs + c), yA = mc_clip3(0, cH - 1, ys + rr);
            s_inC[l][cp][rr * MCC_P + c] = ref[xA + yA * r.stride];
          }
        }
      }
    }
  }
  MC_LDS_SYNC();

  // ---------------- luma: lane -> row (lane>>2), 4 adjacent columns
  const int ly = lane >> 2, lx4 = (lane & 3) * 4;
  int prL[2][4];
#pragma unroll
  for (int l = 0; l < 2; l++) {
    if (t.slot[l] < 0) continue;
    const int xF = t.mv[l][0] & 3, yF = t.mv[l][1] & 3;
    const int bd = P.bd_luma, shift1 = bd - 8;
    const uint16_t* in = &s_inL[l][oxL[l]];
    if (xF == 0 && yF == 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) prL[l][j] = (int16_t)(in[(ly + 3) * MCL_P + lx4 + j + 3] << (14 - bd));
      continue;
    }
    const int nrow = h + 7;
    {                                              // horizontal pass -> s_tmp[row][16]: lane -> row lane>>1, eight adjacent columns
      // (all <= 23 rows in ONE step of 46 lanes; four columns per lane took two steps, the second with 28 of 64 lanes at work)
      const int rr = lane >> 1, x8 = (lane & 1) * 8;
      if (rr < nrow) {
        int o[8];
        if (xF == 0) {
#pragma unroll
          for (int j = 0; j < 8; j++) o[j] = in[rr * MCL_P + x8 + j + 3];
        } else {
          // the 15 samples sv[0..14] the eight outputs need, as pairs: E[m] = (sv[2m], sv[2m+1]) read as dwords (from the
          // even element below the row's start, shifted by one sample when the start is odd), O[m] = (sv[2m+1], sv[2m+2])
          const int e0 = rr * MCL_P + x8 + (oxL[l] & ~1);
          const uint32_t* rowd = reinterpret_cast<const uint32_t*>(&s_inL[l][e0]);
          uint32_t D[9], E[8], O[7];
#pragma unroll
          for (int m = 0; m < 9; m++) D[m] = rowd[m];
          if (oxL[l] & 1) {
#pragma unroll
            for (int m = 0; m < 8; m++) E[m] = __builtin_amdgcn_alignbit(D[m + 1], D[m], 16);
          } else {
#pragma unroll
            for (int m = 0; m < 8; m++) E[m] = D[m];
          }
#pragma unroll
          for (int m = 0; m < 7; m++) O[m] = __builtin_amdgcn_alignbit(E[m + 1], E[m], 16);
          int sum[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
          for (int m = 0; m < 4; m++) {
            const uint32_t tp = c_qpel_pk[xF][m];
#pragma unroll
            for (int i = 0; i < 4; i++) { sum[2 * i] = mc_dot2(E[i + m], tp, sum[2 * i]); sum[2 * i + 1] = mc_dot2(O[i + m], tp, sum[2 * i + 1]); }
          }
#pragma unroll
          for (int j = 0; j < 8; j++) o[j] = (int16_t)(sum[j] >> shift1);
        }
        uint2* dstp = reinterpret_cast<uint2*>(&s_tmp[rr * MCT_P + x8]);
        dstp[0] = make_uint2((uint32_t)(uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16), (uint32_t)(uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16));
        dstp[1] = make_uint2((uint32_t)(uint16_t)o[4] | ((uint32_t)(uint16_t)o[5] << 16), (uint32_t)(uint16_t)o[6] | ((uint32_t)(uint16_t)o[7] << 16));
      }
    }
    MC_LDS_SYNC();
    if (yF == 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) prL[l][j] = s_tmp[(ly + 3) * MCT_P + lx4 + j];
    } else {
      const int vshift = (xF == 0) ? shift1 : 6;
      int acc[4] = { 0, 0, 0, 0 };
#pragma unroll
      for (int k = 0; k < 8; k += 2) {                 // rows k, k+1: their samples of one column as a pair, one dot2 each
        const uint2 r0 = *reinterpret_cast<const uint2*>(&s_tmp[(ly + k) * MCT_P + lx4]);
        const uint2 r1 = *reinterpret_cast<const uint2*>(&s_tmp[(ly + k + 1) * MCT_P + lx4]);
        const uint32_t tp = c_qpel_pk[yF][k >> 1];
        acc[0] = mc_dot2(__builtin_amdgcn_perm(r1.x, r0.x, 0x05040100u), tp, acc[0]);
        acc[1] = mc_dot2(__builtin_amdgcn_perm(r1.x, r0.x, 0x07060302u), tp, acc[1]);
        acc[2] = mc_dot2(__builtin_amdgcn_perm(r1.y, r0.y, 0x05040100u), tp, acc[2]);
        acc[3] = mc_dot2(__builtin_amdgcn_perm(r1.y, r0.y, 0x07060302u), tp, acc[3]);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) prL[l][j] = (int16_t)(acc[j] >> vshift);
    }
    MC_LDS_SYNC();                                 // s_tmp is reused by the next list / chroma
  }

  // weights (motion.cc:403-406, :464-473, :522-540)
  const int la = (mode == 3) ? 0 : l_uni;
  int w0 = 0, o0 = 0, w1 = 0, o1 = 0, log2WD = 1;
  if (mode == 1 || mode == 3) {
    const int bd = P.bd_luma;
    log2WD = sh->luma_log2_weight_denom + max(2, 14 - bd);
    w0 = sh->luma_weight[la][t.ref_idx[la]]; o0 = sh->luma_offset[la][t.ref_idx[la]] * (1 << P.wp_shift_luma);
    if (mode == 3) { w1 = sh->luma_weight[1][t.ref_idx[1]]; o1 = sh->luma_offset[1][t.ref_idx[1]] * (1 << P.wp_shift_luma); }
  }
  if (ly < h && lx4 < w) {
    int o[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
      o[j] = mc_combine<PX>(mode, bi ? prL[0][j] : (use0 ? prL[0][j] : prL[1][j]), prL[1][j], P.bd_luma, w0, o0, w1, o1, log2WD);
    st4_px<PX>((PX*)d0.ptr + t.x + lx4 + (t.y + ly) * d0.stride, o);
  }

  if (!c420) return;
  // ---------------- chroma: both planes at once, lane -> plane (lane>>5), row, 2 adjacent columns
  const int cp = lane >> 5, cy = (lane & 31) >> 2, cx2 = (lane & 3) * 2;
  int prC[2][2];
#pragma unroll
  for (int l = 0; l < 2; l++) {
    if (t.slot[l] < 0) continue;
    const int xF = t.mv[l][0] & 7, yF = t.mv[l][1] & 7;
    const int bd = P.bd_chroma, shift1 = bd - 8;
    if (xF == 0 && yF == 0) {
      const uint16_t* in = &s_inC[l][cp][oxC[l]];
      prC[l][0] = (int16_t)(in[(cy + 1) * MCC_P + cx2 + 1] << (14 - bd));
      prC[l][1] = (int16_t)(in[(cy + 1) * MCC_P + cx2 + 2] << (14 - bd));
      continue;
    }
    const int nrow = hc + 3;
    // horizontal pass, both planes at once: lane -> plane (lane>>5), row ((lane&31)>>1, 11 rows), 4 columns -> s_tmp[plane*11*MCT_P + row*MCT_P + x]
    {
      const int q = lane >> 5, rr = (lane & 31) >> 1, x4 = (lane & 1) * 4;
      if (rr < nrow) {
        int o[4];
        if (xF == 0) {
          const uint16_t* in = &s_inC[l][q][oxC[l]];
#pragma unroll
          for (int j = 0; j < 4; j++) o[j] = in[rr * MCC_P + x4 + j + 1];
        } else {
          // sv[0..6] as pairs (see the luma pass); 4 taps = two dot2 per output
          const int e0 = rr * MCC_P + x4 + (oxC[l] & ~1);
          const uint32_t* rowd = reinterpret_cast<const uint32_t*>(&s_inC[l][q][e0]);
          uint32_t D[5], E[4], O[3];
#pragma unroll
          for (int m = 0; m < 5; m++) D[m] = rowd[m];
          if (oxC[l] & 1) {
#pragma unroll
            for (int m = 0; m < 4; m++) E[m] = __builtin_amdgcn_alignbit(D[m + 1], D[m], 16);
          } else {
#pragma unroll
            for (int m = 0; m < 4; m++) E[m] = D[m];
          }
#pragma unroll
          for (int m = 0; m < 3; m++) O[m] = __builtin_amdgcn_alignbit(E[m + 1], E[m], 16);
          const uint32_t p0 = c_epel_pk[xF][0], p1 = c_epel_pk[xF][1];
          o[0] = (int16_t)(mc_dot2(E[1], p1, mc_dot2(E[0], p0, 0)) >> shift1);
          o[1] = (int16_t)(mc_dot2(O[1], p1, mc_dot2(O[0], p0, 0)) >> shift1);
          o[2] = (int16_t)(mc_dot2(E[2], p1, mc_dot2(E[1], p0, 0)) >> shift1);
          o[3] = (int16_t)(mc_dot2(O[2], p1, mc_dot2(O[1], p0, 0)) >> shift1);
        }
        *reinterpret_cast<uint2*>(&s_tmp[q * (11 * MCT_P) + rr * MCT_P + x4]) =
          make_uint2((uint32_t)(uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16), (uint32_t)(uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16));
      }
    }
    MC_LDS_SYNC();
    const int16_t* tp = &s_tmp[cp * (11 * MCT_P)];
    if (yF == 0) { prC[l][0] = tp[(cy + 1) * MCT_P + cx2]; prC[l][1] = tp[(cy + 1) * MCT_P + cx2 + 1]; }
    else {
      const int vshift = (xF == 0) ? shift1 : 6;
      uint32_t rv[4];
#pragma unroll
      for (int k = 0; k < 4; k++) rv[k] = *reinterpret_cast<const uint32_t*>(&tp[(cy + k) * MCT_P + cx2]);
      // rows k, k+1 of one column as a pair: one dot2 per two taps
      const uint32_t q0 = c_epel_pk[yF][0], q1 = c_epel_pk[yF][1];
      const int a0 = mc_dot2(__builtin_amdgcn_perm(rv[3], rv[2], 0x05040100u), q1, mc_dot2(__builtin_amdgcn_perm(rv[1], rv[0], 0x05040100u), q0, 0));
      const int a1 = mc_dot2(__builtin_amdgcn_perm(rv[3], rv[2], 0x07060302u), q1, mc_dot2(__builtin_amdgcn_perm(rv[1], rv[0], 0x07060302u), q0, 0));
      prC[l][0] = (int16_t)(a0 >> vshift); prC[l][1] = (int16_t)(a1 >> vshift);
    }
    MC_LDS_SYNC();
  }
  if (mode == 1 || mode == 3) {
    const int bd = P.bd_chroma;
    log2WD = sh->chroma_log2_weight_denom + max(2, 14 - bd);
    w0 = sh->chroma_weight[la][t.ref_idx[la]][cp]; o0 = sh->chroma_offset[la][t.ref_idx[la]][cp] * (1 << P.wp_shift_chroma);
    if (mode == 3) { w1 = sh->chroma_weight[1][t.ref_idx[1]][cp]; o1 = sh->chroma_offset[1][t.ref_idx[1]][cp] * (1 << P.wp_shift_chroma); }
  }
  if (cy < hc && cx2 < wc) {
    const PlaneRef dc = cp ? d2 : d1;
    const int a0 = bi ? prC[0][0] : (use0 ? prC[0][0] : prC[1][0]), a1 = bi ? prC[0][1] : (use0 ? prC[0][1] : prC[1][1]);
    const int r0 = mc_combine<PX>(mode, a0, prC[1][0], P.bd_chroma, w0, o0, w1, o1, log2WD);
    const int r1 = mc_combine<PX>(mode, a1, prC[1][1], P.bd_chroma, w0, o0, w1, o1, log2WD);
    st2_px<PX>((PX*)dc.ptr + (t.x >> 1) + cx2 + ((t.y >> 1) + cy) * dc.stride, r0, r1);
  }
}

template __global__ void k_mc<uint8_t>(PicDev, DpbTable, PlaneRef, PlaneRef, PlaneRef, const McTask*,
                                       const de265hip_slice_params*, int);
template __global__ void k_mc<uint16_t>(PicDev, DpbTable, PlaneRef, PlaneRef, PlaneRef, const McTask*,
                                        const de265hip_slice_params*, int);

// ---- chroma prediction of one MC task for any chroma format (4:2:2 / 4:4:4 pictures; mc_chroma, motion.cc:175-273: the
// vector scaled by 2 / SubWidthC, 2 / SubHeightC, eighth-sample fractions).  One wavefront per task and plane
// (blockIdx.y), the tile of at most 16x16 chroma samples through the function-level block interpolator.
template <typename PX>
__global__ __launch_bounds__(64)
void k_mc_chroma_any(PicDev P, DpbTable dpb, PlaneRef d1, PlaneRef d2, const McTask* __restrict__ tasks,
                     const de265hip_slice_params* __restrict__ slices, int n_tasks)
{
  __shared__ uint16_t s_in[23 * MC_IWP];
  __shared__ int16_t s_tmp[23 * 16];
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= n_tasks) return;
  const McTask t = tasks[blockIdx.x];
  const int cp = blockIdx.y;
  const de265hip_slice_params* sh = &slices[t.slice_idx];
  const bool use0 = t.slot[0] >= 0, use1 = t.slot[1] >= 0, bi = use0 && use1;
  const int wc = t.w >> P.csw, hc = t.h >> P.csh, xc = t.x >> P.csw, yc = t.y >> P.csh;
  int mode;
  if (sh->slice_type == 1) mode = P.weighted_pred ? 1 : 0;
  else if (bi) mode = P.weighted_bipred ? 3 : 2;
  else mode = P.weighted_bipred ? 1 : 0;
  int16_t pr[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#pragma unroll
  for (int l = 0; l < 2; l++) {
    if (t.slot[l] < 0) continue;
    const int mvx = t.mv[l][0] * (2 >> P.csw), mvy = t.mv[l][1] * (2 >> P.csh);
    const PlaneRef r = dpb.p[t.slot[l]][cp + 1];
    mc_block<PX, 4, 4>((const PX*)r.ptr, r.stride, P.cwidth, P.cheight, xc + (mvx >> 3), yc + (mvy >> 3), mvx & 7, mvy & 7,
                    wc, hc, P.bd_chroma, s_in, s_tmp, lane, pr[l]);
  }
  int w0 = 0, o0 = 0, w1 = 0, o1 = 0, log2WD = 1;
  if (mode == 1 || mode == 3) {
    const int la = mode == 3 ? 0 : (use0 ? 0 : 1);
    log2WD = sh->chroma_log2_weight_denom + max(2, 14 - P.bd_chroma);
    w0 = sh->chroma_weight[la][t.ref_idx[la]][cp]; o0 = sh->chroma_offset[la][t.ref_idx[la]][cp] * (1 << P.wp_shift_chroma);
    if (mode == 3) { w1 = sh->chroma_weight[1][t.ref_idx[1]][cp]; o1 = sh->chroma_offset[1][t.ref_idx[1]][cp] * (1 << P.wp_shift_chroma); }
  }
  const PlaneRef dc = cp ? d2 : d1;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int s = lane + 64 * k;
    if (s < wc * hc) {
      const int y = s / wc, x = s - y * wc;
      const int a = bi ? pr[0][k] : (use0 ? pr[0][k] : pr[1][k]);
      ((PX*)dc.ptr)[xc + x + (yc + y) * dc.stride] = mc_combine<PX>(mode, a, pr[1][k], P.bd_chroma, w0, o0, w1, o1, log2WD);
    }
  }
}
template __global__ void k_mc_chroma_any<uint8_t>(PicDev, DpbTable, PlaneRef, PlaneRef, const McTask*, const de265hip_slice_params*, int);
template __global__ void k_mc_chroma_any<uint16_t>(PicDev, DpbTable, PlaneRef, PlaneRef, const McTask*, const de265hip_slice_params*, int);

// ---- PCM sample copy (slice.cc:4143-4183), one workgroup per PCM CU
template <typename PX>
__global__ __launch_bounds__(256)
void k_pcm(PicDev P, PlaneRef d0, PlaneRef d1, PlaneRef d2, const PcmTask* __restrict__ tasks,
           const uint16_t* __restrict__ samples)
{
  const PcmTask t = tasks[blockIdx.x];
  const int n = 1 << t.log2_cb_size;
  const uint16_t* s = samples + t.sample_offset;
  const PlaneRef dsts[3] = { d0, d1, d2 };
  for (int comp = 0; comp < 3; comp++) {
    const int w = comp ? n >> P.csw : n, h = comp ? n >> P.csh : n;
    const int x0 = comp ? t.x0 >> P.csw : t.x0, y0 = comp ? t.y0 >> P.csh : t.y0;
    PX* dst = (PX*)dsts[comp].ptr;
    for (int i = threadIdx.x; i < w * h; i += 256) {
      int y = i / w, x = i - y * w;
      dst[(x0 + x) + (y0 + y) * dsts[comp].stride] = (PX)s[i];
    }
    s += w * h;
  }
}
template __global__ void k_pcm<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const PcmTask*, const uint16_t*);
template __global__ void k_pcm<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const PcmTask*, const uint16_t*);

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
