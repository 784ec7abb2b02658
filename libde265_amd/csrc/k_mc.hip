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
// first term of a chain.  (The compiler picks the accumulate-in-place form v_dot2c and spends a v_mov on every chain's zero;
// the three-address form takes the inline constant.)
__device__ __forceinline__ int mc_dot2_first(uint32_t a, uint32_t b)
{ int r; asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int mc_dot2_first_s(uint32_t a, uint32_t b_uniform)
{ int r; asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(b_uniform)); return r; }

#define MC_IWP 24          // LDS pitch of the staged input tile (>= 16+7)

__device__ __forceinline__ int mc_clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// Interpolates a w x h block (w,h <= 16 for NT=8, <= 8 for NT=4) of one reference
// plane into out[] (sample s = lane + 64*k, row-major over w).  Intermediates are
// truncated to int16 after each stage (fallback-motion.cc:346,:377,:508-545).
template <typename PX, int NT, int KMAX = (NT == 8 ? 4 : 1)>     // KMAX: outputs per lane (chroma blocks beyond 8x8: 4)
__device__ __forceinline__ void mc_block_filter(int xF, int yF, int w, int h, int bd, const uint16_t* in, int16_t* tmp, int lane, int16_t* out);

template <typename PX, int NT, int KMAX = (NT == 8 ? 4 : 1)>
__device__ void mc_block(const PX* __restrict__ ref, int rstride, int picW, int picH,
                         int xInt, int yInt, int xF, int yF, int w, int h, int bd,
                         uint16_t* in, int16_t* tmp, int lane, int16_t* out)
{
  constexpr int before = NT == 8 ? 3 : 1;
  const int IW = w + NT - 1, IH = h + NT - 1;
  // idx / IW and idx / w as multiplications (exact for divisors up to 23 and idx < 600: every block here)
  const uint32_t invIW = 65536u / (uint32_t)IW + 1u;
  for (int idx = lane; idx < IW * IH; idx += 64) {
    int r = (int)(((uint32_t)idx * invIW) >> 16), c = idx - r * IW;
    int xA = mc_clip3(0, picW - 1, xInt - before + c);
    int yA = mc_clip3(0, picH - 1, yInt - before + r);
    in[r * MC_IWP + c] = ref[xA + yA * rstride];
  }
  __syncthreads();
  mc_block_filter<PX, NT, KMAX>(xF, yF, w, h, bd, in, tmp, lane, out);
}

// the two filter stages on a staged input tile (rows of MC_IWP samples, the block's first sample at [before][before])
template <typename PX, int NT, int KMAX>
__device__ __forceinline__ void mc_block_filter(int xF, int yF, int w, int h, int bd, const uint16_t* in, int16_t* tmp, int lane, int16_t* out)
{
  constexpr int before = NT == 8 ? 3 : 1;
  const int IH = h + NT - 1;
  const uint32_t invW = 65536u / (uint32_t)w + 1u;
  const int shift1 = bd - 8;
  const int nOut = w * h;
  if (xF == 0 && yF == 0) {
    const int shift3 = 14 - bd;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
      int s = lane + 64 * k;
      if (s < nOut) {
        int y = (int)(((uint32_t)s * invW) >> 16), x = s - y * w;
        out[k] = (int16_t)(in[(y + before) * MC_IWP + x + before] << shift3);
      }
    }
    __syncthreads();
    return;
  }
  for (int idx = lane; idx < IH * w; idx += 64) {
    int r = (int)(((uint32_t)idx * invW) >> 16), x = idx - r * w;
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
      int y = (int)(((uint32_t)s * invW) >> 16), x = s - y * w;
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

// (Round 2 tried a second KERNEL for tiles of at most 8x8, four per wavefront: 71-79 us instead of 62 us per 4K B picture -
//  a launch of its own pays its own ramp and tail.  Round 3 has that form inside k_mc_all's single launch: mc_micro_body.)
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

#define MC_TILE_LDS (2 * 23 * MCL_P * 2 + 4 * 11 * MCC_P * 2 + 23 * MCT_P * 2)
template <typename PX>
__device__ __forceinline__ void mc_tile_body(const PicDev& P, const DpbTable& dpb, const PlaneRef& d0, const PlaneRef& d1, const PlaneRef& d2,
                                             const McTask* __restrict__ tasks, const de265hip_slice_params* __restrict__ slices,
                                             int tix, char* smem)
{
  uint16_t (*s_inL)[23 * MCL_P] = reinterpret_cast<uint16_t (*)[23 * MCL_P]>(smem);
  uint16_t (*s_inC)[2][11 * MCC_P] = reinterpret_cast<uint16_t (*)[2][11 * MCC_P]>(smem + 2 * 23 * MCL_P * 2);
  int16_t* s_tmp = reinterpret_cast<int16_t*>(smem + 2 * 23 * MCL_P * 2 + 4 * 11 * MCC_P * 2);
  const int lane = threadIdx.x;

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
  const bool c420 = P.chroma_format == 1;           // (4:2:2 / 4:4:4: the chroma planes are predicted by mc_chroma_any_body, k_mc's blockIdx.y == 1)

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

#define MC_CHROMA_ANY_LDS (4 * 19 * MC_IWP * 2 + 23 * 16 * 2)
template <typename PX>
__device__ __forceinline__ void mc_chroma_any_body(const PicDev& P, const DpbTable& dpb, const PlaneRef& d1, const PlaneRef& d2, const McTask* __restrict__ tasks,
                                                   const de265hip_slice_params* __restrict__ slices, int tix, char* smem);

// blockIdx.y == 1 (4:2:2 / 4:4:4 pictures, where the tile body predicts luma only): the task's chroma planes, in the same launch -
// a kernel of its own behind this one paid its own ramp and tail, and took its tasks in list order across the XCDs
template <typename PX>
__global__ __launch_bounds__(64)
void k_mc(PicDev P, DpbTable dpb, PlaneRef d0, PlaneRef d1, PlaneRef d2,
          const McTask* __restrict__ tasks, const de265hip_slice_params* __restrict__ slices, int n_tasks)
{
  __shared__ __attribute__((aligned(16))) char smem[MC_TILE_LDS > MC_CHROMA_ANY_LDS ? MC_TILE_LDS : MC_CHROMA_ANY_LDS];
  // XCD-aware mapping: workgroups b and b+8 share an XCD (and its L2); give every XCD one contiguous
  // eighth of the task list (tasks are in decode order, i.e. spatial neighbours) so that the
  // overlapping filter margins of neighbouring tiles hit in the same L2.  Speed only, never correctness.
  const int per = (n_tasks + 7) >> 3;
  const int tix = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tix >= n_tasks) return;
  if (blockIdx.y == 1) { mc_chroma_any_body<PX>(P, dpb, d1, d2, tasks, slices, tix, smem); return; }
  mc_tile_body<PX>(P, dpb, d0, d1, d2, tasks, slices, tix, smem);
}
template __global__ void k_mc<uint8_t>(PicDev, DpbTable, PlaneRef, PlaneRef, PlaneRef, const McTask*,
                                       const de265hip_slice_params*, int);
template __global__ void k_mc<uint16_t>(PicDev, DpbTable, PlaneRef, PlaneRef, PlaneRef, const McTask*,
                                        const de265hip_slice_params*, int);

// ---------------------------------------------------------------- picture-level MC kernel, second form (4:2:0, interior tasks)
// k_mc above is bound by instruction issue (tools/exp/pmc_mc.sh: 380 VALU + 372 SALU instructions per wavefront, the VALU
// pipes ~90 % busy while its wavefronts are resident), not by HBM.  The chunk form (mc_chunk_body) does the same arithmetic with about half the
// instructions per tile:
//  * a task is a PU chunk of up to 32x32 luma samples (the host makes them 32x16: two 16x16 tiles) walked by ONE wavefront: task decode, mode,
//    weights, filter taps, buffer descriptors and every per-lane address are computed once per chunk, a tile costs a few
//    scalar offsets;
//  * only chunks whose reference blocks (margins included) lie inside the picture come here (the host sorts the others into
//    k_mc's list): the fetch is unconditional raw buffer loads - descriptor in scalar registers, one per-lane offset for all
//    planes of a list, out-of-range rows read as zero instead of faulting - into an LDS tile of fixed geometry: no per-lane
//    conditions, no 64-bit address arithmetic;
//  * horizontal pass: one lane = 8 adjacent outputs of one row from 8 dwords of sample pairs.  Outputs that start on an
//    even sample take 4 v_dot2 with the tap pairs (t0,t1)..(t6,t7), outputs that start on an odd sample 5 v_dot2 with
//    (0,t0)(t1,t2)..(t7,0): no v_alignbit re-pairing of the row, whatever the parity of the block's start;
//  * its int16 results go to LDS as halves of VERTICAL pairs, twice: (row 2q, row 2q+1) and (row 2q-1, row 2q).  The
//    vertical pass of output row y then reads its four tap pairs ready-made - from the first copy for even y, the second
//    for odd y - with four 16-byte LDS reads per lane and 16 v_dot2: no v_perm pair building;
//  * stores are raw buffer stores with a per-lane offset computed once per chunk.
// Truncations as in the reference (fallback-motion.cc:346,:377,:508-545): int16 after each pass.
typedef unsigned int mc_v2u __attribute__((ext_vector_type(2)));
#define M2_LP 28                      // luma input tile: 23 rows x 7 chunks of 4 samples
#define M2_LROWS 23
#define M2_CP 16                      // chroma input tile: 11 rows x 4 chunks
#define M2_CROWS 11
#define M2_PP 20                      // pair buffers: 12 pair rows x 16 columns (pitch 20 dwords: 16-byte aligned rows)
#define M2_PR 12
#define M2_CPP 8                      // chroma pair buffers (inside the luma ones): per plane 6 pair rows x 8 columns
#define M2_CPR 6

#define QPK(a, b) ((uint32_t)(uint16_t)(int16_t)(a) | ((uint32_t)(uint16_t)(int16_t)(b) << 16))
#define QROW(t0, t1, t2, t3, t4, t5, t6, t7) { QPK(t0, t1), QPK(t2, t3), QPK(t4, t5), QPK(t6, t7), \
                                               QPK(0, t0), QPK(t1, t2), QPK(t3, t4), QPK(t5, t6), QPK(t7, 0), 0, 0, 0 }
#define EROW(t0, t1, t2, t3) { QPK(t0, t1), QPK(t2, t3), QPK(0, t0), QPK(t1, t2), QPK(t3, 0), 0, 0, 0 }
// [fraction][0..3: pairs for an even start, 4..8: pairs for an odd start]
__device__ __constant__ uint32_t c_qpel_eo[4][12] = {
  QROW(0, 0, 0, 64, 0, 0, 0, 0), QROW(-1, 4, -10, 58, 17, -5, 1, 0), QROW(-1, 4, -11, 40, 40, -11, 4, -1), QROW(0, 1, -5, 17, 58, -10, 4, -1) };
// [fraction][0..1 even start, 2..4 odd start]
__device__ __constant__ uint32_t c_epel_eo[8][8] = {
  EROW(0, 64, 0, 0), EROW(-2, 58, 10, -2), EROW(-4, 54, 16, -2), EROW(-6, 46, 28, -4),
  EROW(-4, 36, 36, -4), EROW(-4, 28, 46, -6), EROW(-2, 16, 54, -4), EROW(-2, 10, 58, -2) };
#undef QROW
#undef EROW
#undef QPK

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mc_rsrc(const void* base, uint32_t bytes)
{ return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000); }
// four samples at byte offset voff + soff of a plane, as two dwords of two 16-bit samples
template <typename PX> __device__ __forceinline__ uint2 mc_fetch4(__amdgpu_buffer_rsrc_t r, int voff, int soff);
template <> __device__ __forceinline__ uint2 mc_fetch4<uint16_t>(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{ const mc_v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0); return make_uint2(v.x, v.y); }
template <> __device__ __forceinline__ uint2 mc_fetch4<uint8_t>(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
  const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
  return make_uint2(__builtin_amdgcn_perm(0, v, 0x0C010C00u), __builtin_amdgcn_perm(0, v, 0x0C030C02u));
}
template <typename PX> __device__ __forceinline__ void mc_store4(__amdgpu_buffer_rsrc_t r, int voff, int soff, const int v[4]);
template <> __device__ __forceinline__ void mc_store4<uint16_t>(__amdgpu_buffer_rsrc_t r, int voff, int soff, const int v[4])
{
  mc_v2u q; q.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); q.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
  __builtin_amdgcn_raw_buffer_store_b64(q, r, voff, soff, 0);
}
template <> __device__ __forceinline__ void mc_store4<uint8_t>(__amdgpu_buffer_rsrc_t r, int voff, int soff, const int v[4])
{ __builtin_amdgcn_raw_buffer_store_b32((uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24), r, voff, soff, 0); }
template <typename PX> __device__ __forceinline__ void mc_store2(__amdgpu_buffer_rsrc_t r, int voff, int soff, int a, int b);
template <> __device__ __forceinline__ void mc_store2<uint16_t>(__amdgpu_buffer_rsrc_t r, int voff, int soff, int a, int b)
{ __builtin_amdgcn_raw_buffer_store_b32((uint32_t)a | ((uint32_t)b << 16), r, voff, soff, 0); }
template <> __device__ __forceinline__ void mc_store2<uint8_t>(__amdgpu_buffer_rsrc_t r, int voff, int soff, int a, int b)
{ __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(a | (b << 8)), r, voff, soff, 0); }

// the 8 outputs of a horizontal 8-tap pass from the row's dwords D[0..7] = sample pairs (d0,d1)..(d14,d15); output j starts at
// sample j + ODD.  T: 4 even-start + 5 odd-start tap pairs (scalar registers)
template <int ODD, bool UNI>
__device__ __forceinline__ void mc_h8(const uint32_t D[8], const uint32_t T[9], int o[8])
{
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int q = j + ODD, i = q >> 1;
    int a;
    if (q & 1) {
      a = UNI ? mc_dot2_first_s(D[i], T[4]) : mc_dot2_first(D[i], T[4]);
#pragma unroll
      for (int m = 1; m < 5; m++) a = mc_dot2(D[i + m], T[4 + m], a);
    } else {
      a = UNI ? mc_dot2_first_s(D[i], T[0]) : mc_dot2_first(D[i], T[0]);
#pragma unroll
      for (int m = 1; m < 4; m++) a = mc_dot2(D[i + m], T[m], a);
    }
    o[j] = a;
  }
}
// the same for 4 outputs of a 4-tap pass from D[0..3]; T: 2 even-start + 3 odd-start pairs
template <int ODD, bool UNI>
__device__ __forceinline__ void mc_h4(const uint32_t D[4], const uint32_t T[5], int o[4])
{
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int q = j + ODD, i = q >> 1;
    int a;
    if (q & 1) {
      a = UNI ? mc_dot2_first_s(D[i], T[2]) : mc_dot2_first(D[i], T[2]);
#pragma unroll
      for (int m = 1; m < 3; m++) a = mc_dot2(D[i + m], T[2 + m], a);
    } else {
      a = UNI ? mc_dot2_first_s(D[i], T[0]) : mc_dot2_first(D[i], T[0]);
      a = mc_dot2(D[i + 1], T[1], a);
    }
    o[j] = a;
  }
}
// sample n (0..) of a row of pairs
__device__ __forceinline__ int mc_pick(const uint32_t* D, int n) { return (int)((D[n >> 1] >> (16 * (n & 1))) & 0xFFFFu); }

#define MC_CHUNK_LDS (2 * M2_LROWS * M2_LP * 2 + 4 * M2_CROWS * M2_CP * 2 + 2 * M2_PR * M2_PP * 4)
template <typename PX>
__device__ __forceinline__ void mc_chunk_body(const PicDev& P, const DpbTable& dpb, const PlaneRef& d0, const PlaneRef& d1, const PlaneRef& d2,
                                              const McTask* __restrict__ tasks, const de265hip_slice_params* __restrict__ slices,
                                              int tix, char* smem)
{
  uint16_t (*s_inL)[M2_LROWS * M2_LP] = reinterpret_cast<uint16_t (*)[M2_LROWS * M2_LP]>(smem);
  uint16_t (*s_inC)[2][M2_CROWS * M2_CP] = reinterpret_cast<uint16_t (*)[2][M2_CROWS * M2_CP]>(smem + 2 * M2_LROWS * M2_LP * 2);
  uint32_t* s_pair = reinterpret_cast<uint32_t*>(smem + 2 * M2_LROWS * M2_LP * 2 + 4 * M2_CROWS * M2_CP * 2);      // [first copy | second copy]
  const int lane = threadIdx.x;

  McTask t;
  {
    const uint32_t* tq = reinterpret_cast<const uint32_t*>(tasks + tix);
    uint32_t tw[5];
#pragma unroll
    for (int i = 0; i < 5; i++) tw[i] = __builtin_amdgcn_readfirstlane(tq[i]);
    __builtin_memcpy(&t, tw, sizeof(t));
  }
  const de265hip_slice_params* sh = &slices[t.slice_idx];
  const bool use0 = t.slot[0] >= 0, use1 = t.slot[1] >= 0, bi = use0 && use1;
  const int l_uni = use0 ? 0 : 1;
  constexpr int bpp = (int)sizeof(PX);
  int mode;                                           // motion.cc:440-620
  if (sh->slice_type == 1) mode = P.weighted_pred ? 1 : 0;
  else if (bi) mode = P.weighted_bipred ? 3 : 2;
  else mode = P.weighted_bipred ? 1 : 0;
  const int bdL = P.bd_luma, bdC = P.bd_chroma;
  const int cH = P.height >> 1;

  // ---- per list: one descriptor for the slot's three planes (one allocation, luma first: host.hip alloc_slot), per-lane
  // fetch offsets
  __amdgpu_buffer_rsrc_t rsR[2];
  int sbL[2], sbC[2], ofCb[2], ofCr[2];                // row pitch in bytes; where the chroma planes start
  // luma fetch slots lane + 64 i: row (lane >> 3) + 8 i, chunk lane & 7 (the 8th chunk of a row is fetched and dropped); the
  // slots of a picture's DPB share one geometry, hence one pitch (checked at launch): one per-lane offset serves both lists
  int vofL = 0, vofC = 0;
#pragma unroll
  for (int l = 0; l < 2; l++) {
    if (t.slot[l] < 0) continue;
    const PlaneRef pl = dpb.p[t.slot[l]][0], pb = dpb.p[t.slot[l]][1], pr = dpb.p[t.slot[l]][2];
    sbL[l] = pl.stride * bpp; sbC[l] = pb.stride * bpp;
    ofCb[l] = (int)((const char*)pb.ptr - (const char*)pl.ptr); ofCr[l] = (int)((const char*)pr.ptr - (const char*)pl.ptr);
    rsR[l] = mc_rsrc(pl.ptr, (uint32_t)(max(ofCb[l], ofCr[l]) + sbC[l] * cH));
    vofL = (lane >> 3) * sbL[l] + (lane & 7) * 4 * bpp;
    vofC = (lane >> 2) * sbC[l] + (lane & 3) * 4 * bpp;
  }
  const int ldsL0 = (lane >> 3) * M2_LP + 4 * (lane & 7);

  // weights (motion.cc:403-406, :464-473, :522-540), luma and the two chroma planes
  const int la = (mode == 3) ? 0 : l_uni;
  const int cp = lane >> 5;                            // chroma: lane -> plane
  int w0 = 0, o0 = 0, w1 = 0, o1 = 0, log2WD = 1, cw0 = 0, co0 = 0, cw1 = 0, co1 = 0, clog2WD = 1;
  if (mode == 1 || mode == 3) {
    log2WD = sh->luma_log2_weight_denom + max(2, 14 - bdL);
    w0 = sh->luma_weight[la][t.ref_idx[la]]; o0 = sh->luma_offset[la][t.ref_idx[la]] * (1 << P.wp_shift_luma);
    clog2WD = sh->chroma_log2_weight_denom + max(2, 14 - bdC);
    cw0 = sh->chroma_weight[la][t.ref_idx[la]][cp]; co0 = sh->chroma_offset[la][t.ref_idx[la]][cp] * (1 << P.wp_shift_chroma);
    if (mode == 3) {
      w1 = sh->luma_weight[1][t.ref_idx[1]]; o1 = sh->luma_offset[1][t.ref_idx[1]] * (1 << P.wp_shift_luma);
      cw1 = sh->chroma_weight[1][t.ref_idx[1]][cp]; co1 = sh->chroma_offset[1][t.ref_idx[1]][cp] * (1 << P.wp_shift_chroma);
    }
  }

  // destination: descriptors and per-lane offsets.  luma: lane -> row lane>>2, 4 columns; chroma: plane lane>>5, row, 2 columns
  const int dsbL = d0.stride * bpp, dsbC = d1.stride * bpp;
  const int dofCb = (int)((const char*)d1.ptr - (const char*)d0.ptr), dofCr = (int)((const char*)d2.ptr - (const char*)d0.ptr);
  const __amdgpu_buffer_rsrc_t rdD = mc_rsrc(d0.ptr, (uint32_t)(max(dofCb, dofCr) + dsbC * cH));
  const int ly = lane >> 2, lx4 = (lane & 3) * 4;
  const int cy = (lane & 31) >> 2, cx2 = (lane & 3) * 2;
  const int dvL = ly * dsbL + lx4 * bpp, dvC = cy * dsbC + cx2 * bpp + (cp ? dofCr : dofCb);
  // horizontal pass lanes.  luma: row lane>>1 (lanes 0..45), 8 columns; chroma: plane lane>>5, row (lane&31)>>1 (< 11), 4 columns
  const int hr = lane >> 1, hx8 = (lane & 1) * 8;
  const int hq = lane >> 5, hcr = (lane & 31) >> 1, hx4 = (lane & 1) * 4;
  uint16_t* const pair16 = reinterpret_cast<uint16_t*>(s_pair);
  // where a lane's horizontal results go: first copy pair row r>>1, half r&1; second copy pair row (r+1)>>1, half (r+1)&1
  const int peL = ((hr >> 1) * M2_PP + hx8) * 2 + (hr & 1);
  const int poL = (M2_PR * M2_PP + ((hr + 1) >> 1) * M2_PP + hx8) * 2 + ((hr + 1) & 1);
  const int cbase = hq * (2 * M2_CPR * M2_CPP);                         // chroma plane q: [first copy | second copy] of 6 x 8 dwords
  const int peC = (cbase + (hcr >> 1) * M2_CPP + hx4) * 2 + (hcr & 1);
  const int poC = (cbase + M2_CPR * M2_CPP + ((hcr + 1) >> 1) * M2_CPP + hx4) * 2 + ((hcr + 1) & 1);
  // where a lane's vertical pass reads: output row y -> first copy from pair row y>>1 (y even), second from (y+1)>>1 (y odd)
  const int pvL = (ly & 1) * (M2_PR * M2_PP) + ((ly + (ly & 1)) >> 1) * M2_PP + lx4;
  const int pvC = cp * (2 * M2_CPR * M2_CPP) + (cy & 1) * (M2_CPR * M2_CPP) + ((cy + (cy & 1)) >> 1) * M2_CPP + cx2;

  for (int ty = 0; ty < t.h; ty += 16)
    for (int tx = 0; tx < t.w; tx += 16) {
      const int tw = min(16, t.w - tx), th = min(16, t.h - ty);
      const int X = t.x + tx, Y = t.y + ty;
      // ---------------- fetch: every load of both lists and all planes back to back, then into LDS
      uint2 rl[2][3], rc[2][2];
      int oxL[2] = { 0, 0 }, oxC[2] = { 0, 0 };
#pragma unroll
      for (int l = 0; l < 2; l++) {
        if (t.slot[l] < 0) continue;
        const int xs = X + (t.mv[l][0] >> 2) - 3, ys = Y + (t.mv[l][1] >> 2) - 3;
        oxL[l] = xs & 3;
        const int so = ys * sbL[l] + (xs & ~3) * bpp;
        rl[l][0] = mc_fetch4<PX>(rsR[l], vofL, so);
        rl[l][1] = mc_fetch4<PX>(rsR[l], vofL, so + 8 * sbL[l]);
        rl[l][2] = mc_fetch4<PX>(rsR[l], vofL, so + 16 * sbL[l]);
        const int xc = (X >> 1) + (t.mv[l][0] >> 3) - 1, yc = (Y >> 1) + (t.mv[l][1] >> 3) - 1;
        oxC[l] = xc & 3;
        const int sc = yc * sbC[l] + (xc & ~3) * bpp;
        rc[l][0] = mc_fetch4<PX>(rsR[l], vofC, sc + ofCb[l]);
        rc[l][1] = mc_fetch4<PX>(rsR[l], vofC, sc + ofCr[l]);
      }
#pragma unroll
      for (int l = 0; l < 2; l++) {
        if (t.slot[l] < 0) continue;
        if ((lane & 7) != 7) {
          *reinterpret_cast<uint2*>(&s_inL[l][ldsL0]) = rl[l][0];
          *reinterpret_cast<uint2*>(&s_inL[l][ldsL0 + 8 * M2_LP]) = rl[l][1];
          if (lane < 8 * (M2_LROWS - 16)) *reinterpret_cast<uint2*>(&s_inL[l][ldsL0 + 16 * M2_LP]) = rl[l][2];
        }
        if (lane < M2_CROWS * 4) {
          *reinterpret_cast<uint2*>(&s_inC[l][0][lane * 4]) = rc[l][0];
          *reinterpret_cast<uint2*>(&s_inC[l][1][lane * 4]) = rc[l][1];
        }
      }
      MC_LDS_SYNC();

      // ---------------- luma
      int prL[2][4];
#pragma unroll
      for (int l = 0; l < 2; l++) {
        if (t.slot[l] < 0) continue;
        int xF = t.mv[l][0] & 3, yF = t.mv[l][1] & 3;
        const int shift1 = bdL - 8;
        asm volatile("" : "+s"(xF), "+s"(yF));       // (tap pairs are fetched where they are used: hoisted out of the tile loop they do not fit the scalar registers)
        if (xF == 0 && yF == 0) {
          const uint16_t* in = &s_inL[l][(ly + 3) * M2_LP + lx4 + 3 + oxL[l]];
#pragma unroll
          for (int j = 0; j < 4; j++) prL[l][j] = (int16_t)(in[j] << (14 - bdL));
          continue;
        }
        if (lane < 2 * M2_LROWS) {
          const uint32_t* rowd = reinterpret_cast<const uint32_t*>(&s_inL[l][hr * M2_LP + hx8 + (oxL[l] & ~1)]);
          uint32_t D[8];
#pragma unroll
          for (int m = 0; m < 8; m++) D[m] = rowd[m];
          int o[8];
          if (xF == 0) {
            if (oxL[l] & 1) {
#pragma unroll
              for (int j = 0; j < 8; j++) o[j] = mc_pick(D, j + 4);
            } else {
#pragma unroll
              for (int j = 0; j < 8; j++) o[j] = mc_pick(D, j + 3);
            }
          } else {
            uint32_t T[9];
#pragma unroll
            for (int m = 0; m < 9; m++) T[m] = c_qpel_eo[xF][m];
            if (oxL[l] & 1) mc_h8<1, true>(D, T, o); else mc_h8<0, true>(D, T, o);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] >>= shift1;
          }
#pragma unroll
          for (int j = 0; j < 8; j++) { pair16[peL + 2 * j] = (uint16_t)o[j]; pair16[poL + 2 * j] = (uint16_t)o[j]; }
        }
        MC_LDS_SYNC();
        {
          const int vshift = (xF == 0) ? shift1 : 6;
          int acc[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint4 v = *reinterpret_cast<const uint4*>(&s_pair[pvL + j * M2_PP]);
            const uint32_t tp = c_qpel_eo[yF][j];
            if (j == 0) { acc[0] = mc_dot2_first_s(v.x, tp); acc[1] = mc_dot2_first_s(v.y, tp); acc[2] = mc_dot2_first_s(v.z, tp); acc[3] = mc_dot2_first_s(v.w, tp); }
            else {
              acc[0] = mc_dot2(v.x, tp, acc[0]); acc[1] = mc_dot2(v.y, tp, acc[1]);
              acc[2] = mc_dot2(v.z, tp, acc[2]); acc[3] = mc_dot2(v.w, tp, acc[3]);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; j++) prL[l][j] = (int16_t)(acc[j] >> vshift);
        }
        MC_LDS_SYNC();                               // the pair buffers are reused by the next list / chroma
      }
      if (ly < th && lx4 < tw) {
        int o[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
          o[j] = mc_combine<PX>(mode, bi ? prL[0][j] : (use0 ? prL[0][j] : prL[1][j]), prL[1][j], bdL, w0, o0, w1, o1, log2WD);
        mc_store4<PX>(rdD, dvL, Y * dsbL + X * bpp, o);
      }

      // ---------------- chroma, both planes at once
      int prC[2][2];
#pragma unroll
      for (int l = 0; l < 2; l++) {
        if (t.slot[l] < 0) continue;
        int xF = t.mv[l][0] & 7, yF = t.mv[l][1] & 7;
        const int shift1 = bdC - 8;
        asm volatile("" : "+s"(xF), "+s"(yF));
        if (xF == 0 && yF == 0) {
          const uint16_t* in = &s_inC[l][cp][(cy + 1) * M2_CP + cx2 + 1 + oxC[l]];
          prC[l][0] = (int16_t)(in[0] << (14 - bdC)); prC[l][1] = (int16_t)(in[1] << (14 - bdC));
          continue;
        }
        if ((lane & 31) < 2 * M2_CROWS) {
          const uint32_t* rowd = reinterpret_cast<const uint32_t*>(&s_inC[l][hq][hcr * M2_CP + hx4 + (oxC[l] & ~1)]);
          uint32_t D[4];
#pragma unroll
          for (int m = 0; m < 4; m++) D[m] = rowd[m];
          int o[4];
          if (xF == 0) {
            if (oxC[l] & 1) {
#pragma unroll
              for (int j = 0; j < 4; j++) o[j] = mc_pick(D, j + 2);
            } else {
#pragma unroll
              for (int j = 0; j < 4; j++) o[j] = mc_pick(D, j + 1);
            }
          } else {
            uint32_t T[5];
#pragma unroll
            for (int m = 0; m < 5; m++) T[m] = c_epel_eo[xF][m];
            if (oxC[l] & 1) mc_h4<1, true>(D, T, o); else mc_h4<0, true>(D, T, o);
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] >>= shift1;
          }
#pragma unroll
          for (int j = 0; j < 4; j++) { pair16[peC + 2 * j] = (uint16_t)o[j]; pair16[poC + 2 * j] = (uint16_t)o[j]; }
        }
        MC_LDS_SYNC();
        {
          const int vshift = (xF == 0) ? shift1 : 6;
          const uint2 v0 = *reinterpret_cast<const uint2*>(&s_pair[pvC]);
          const uint2 v1 = *reinterpret_cast<const uint2*>(&s_pair[pvC + M2_CPP]);
          const uint32_t q0 = c_epel_eo[yF][0], q1 = c_epel_eo[yF][1];
          const int a0 = mc_dot2(v1.x, q1, mc_dot2_first_s(v0.x, q0));
          const int a1 = mc_dot2(v1.y, q1, mc_dot2_first_s(v0.y, q0));
          prC[l][0] = (int16_t)(a0 >> vshift); prC[l][1] = (int16_t)(a1 >> vshift);
        }
        MC_LDS_SYNC();
      }
      if (cy < (th >> 1) && cx2 < (tw >> 1)) {
        const int a0 = bi ? prC[0][0] : (use0 ? prC[0][0] : prC[1][0]), a1 = bi ? prC[0][1] : (use0 ? prC[0][1] : prC[1][1]);
        const int q0 = mc_combine<PX>(mode, a0, prC[1][0], bdC, cw0, co0, cw1, co1, clog2WD);
        const int q1 = mc_combine<PX>(mode, a1, prC[1][1], bdC, cw0, co0, cw1, co1, clog2WD);
        mc_store2<PX>(rdD, dvC, (Y >> 1) * dsbC + (X >> 1) * bpp, q0, q1);
      }
    }
}

// ---------------------------------------------------------------- picture-level MC kernel for small PUs (4:2:0, interior)
// Half of a B picture's MC tasks are PUs of 8x8, 8x4 or 4x8 luma samples: as tiles of k_mc / the chunk form (mc_chunk_body) they use 4-16 of a
// wavefront's 64 lanes for the price of a full tile.  Here a wavefront takes FOUR blocks of at most 8x8 luma samples, one per
// 16-lane group, through the same steps as the chunk form (mc_chunk_body) (horizontal pass -> vertical pairs in LDS -> vertical pass), with everything
// that is a scalar there - position, vector, fractions, tap pairs, weights - held per lane.  Wavefront-uniform stay: the two
// reference slots (the host sorts the blocks by their slot pair: one buffer descriptor per list, uni / bi known) and the code
// path: both passes always run, fraction 0 through the taps (0,0,0,64,0,..): bit-exact because 64 * s >> shift1 fits the
// int16 intermediate and (64 * v) >> 6 = v; all four weighting modes through the explicit-weight formulas with w = 1, o = 0
// where the slice has none ((a + 2^(s-1)) >> s and (a + b + 2^s) >> (s+1) are those formulas at log2WD = s).
#define MM_LP 20                      // luma input: 15 rows x 5 chunks of 4 samples
#define MM_LROWS 15
#define MM_CP 12                      // chroma input: 2 planes x 7 rows x 3 chunks
#define MM_CROWS 7
#define MC_MICRO_LDS (4 * MM_LROWS * MM_LP * 2 + 8 * MM_CROWS * MM_CP * 2 + 4 * 128 * 4 + (48 + 64) * 4)
template <typename PX>
__device__ __forceinline__ void mc_micro_body(const PicDev& P, const DpbTable& dpb, const PlaneRef& d0, const PlaneRef& d1, const PlaneRef& d2,
                                              const McTask* __restrict__ tasks, const de265hip_slice_params* __restrict__ slices,
                                              int qix, char* smem)
{
  uint16_t (*s_inL)[MM_LROWS * MM_LP] = reinterpret_cast<uint16_t (*)[MM_LROWS * MM_LP]>(smem);                      // 4 x 600 B
  uint16_t (*s_inC)[2][MM_CROWS * MM_CP] = reinterpret_cast<uint16_t (*)[2][MM_CROWS * MM_CP]>(smem + 4 * MM_LROWS * MM_LP * 2);   // 4 x 336 B
  uint32_t (*s_pair)[128] = reinterpret_cast<uint32_t (*)[128]>(smem + 4 * MM_LROWS * MM_LP * 2 + 8 * MM_CROWS * MM_CP * 2);   // per group [first copy | second copy] of 8 pair rows x 8 columns
  // the tap tables, indexed per lane: from LDS (as loads from constant memory they were four dependent memory round trips per list)
  uint32_t* s_tq = reinterpret_cast<uint32_t*>(smem + 4 * MM_LROWS * MM_LP * 2 + 8 * MM_CROWS * MM_CP * 2 + 4 * 128 * 4);      // [4][12]
  uint32_t* s_te = s_tq + 48;                                                                                                  // [8][8]
  const int lane = threadIdx.x, g = lane >> 4, gl = lane & 15;

  constexpr int bpp = (int)sizeof(PX);
  if (lane < 48) s_tq[lane] = (&c_qpel_eo[0][0])[lane];
  s_te[lane] = (&c_epel_eo[0][0])[lane];
  // the group's task, per lane
  uint32_t tw[5];
  {
    const uint32_t* tq = reinterpret_cast<const uint32_t*>(tasks + 4 * qix + g);
#pragma unroll
    for (int i = 0; i < 5; i++) tw[i] = tq[i];
  }
  const int X = tw[0] & 0xFFFF, Y = tw[0] >> 16, W = tw[1] & 0xFF, H = (tw[1] >> 8) & 0xFF;
  // (slots: wavefront-uniform by construction; read from the first group)
  const int slot0 = (int8_t)(__builtin_amdgcn_readfirstlane(tw[1]) >> 16), slot1 = (int8_t)(__builtin_amdgcn_readfirstlane(tw[1]) >> 24);
  const bool use0 = slot0 >= 0, use1 = slot1 >= 0, bi = use0 && use1;
  const int mvx[2] = { (int16_t)(tw[2] & 0xFFFF), (int16_t)(tw[3] & 0xFFFF) }, mvy[2] = { (int16_t)(tw[2] >> 16), (int16_t)(tw[3] >> 16) };
  const int slice_idx = tw[4] & 0xFFFF, ref0 = (int8_t)(tw[4] >> 16), ref1 = (int8_t)(tw[4] >> 24);
  const de265hip_slice_params* sh = &slices[slice_idx];
  const int bdL = P.bd_luma, bdC = P.bd_chroma, cH = P.height >> 1;
  const int cpl = gl >> 3;                             // chroma: lane -> plane

  // weights per lane: luma, and the lane's chroma plane
  const int l_uni = use0 ? 0 : 1;
  int w0 = 1, o0 = 0, w1 = 1, o1 = 0, log2WD = 14 - bdL, cw0 = 1, co0 = 0, cw1 = 1, co1 = 0, clog2WD = 14 - bdC;
  {
    const bool weighted = sh->slice_type == 1 ? P.weighted_pred != 0 : P.weighted_bipred != 0;
    if (weighted) {
      const int la = bi ? 0 : l_uni, ra = la ? ref1 : ref0;
      log2WD = sh->luma_log2_weight_denom + max(2, 14 - bdL);
      clog2WD = sh->chroma_log2_weight_denom + max(2, 14 - bdC);
      w0 = sh->luma_weight[la][ra]; o0 = sh->luma_offset[la][ra] * (1 << P.wp_shift_luma);
      cw0 = sh->chroma_weight[la][ra][cpl]; co0 = sh->chroma_offset[la][ra][cpl] * (1 << P.wp_shift_chroma);
      if (bi) {
        w1 = sh->luma_weight[1][ref1]; o1 = sh->luma_offset[1][ref1] * (1 << P.wp_shift_luma);
        cw1 = sh->chroma_weight[1][ref1][cpl]; co1 = sh->chroma_offset[1][ref1][cpl] * (1 << P.wp_shift_chroma);
      }
    }
  }

  // destination (one descriptor: the planes of a slot are one allocation, host.hip alloc_slot)
  const int dsbL = d0.stride * bpp, dsbC = d1.stride * bpp;
  const int dofCb = (int)((const char*)d1.ptr - (const char*)d0.ptr), dofCr = (int)((const char*)d2.ptr - (const char*)d0.ptr);
  const __amdgpu_buffer_rsrc_t rdD = mc_rsrc(d0.ptr, (uint32_t)(max(dofCb, dofCr) + dsbC * cH));
  const int vr = gl >> 1, vc4 = (gl & 1) * 4;          // luma output: row, 4 columns
  const int cy = (gl >> 1) & 3, cc2 = (gl & 1) * 2;    // chroma output: plane cpl, row, 2 columns
  uint16_t* const pair16 = reinterpret_cast<uint16_t*>(s_pair[g]);
  const uint32_t* const pairw = s_pair[g];

  int prL[2][4], prC[2][2];
  uint2 fl[5], fc[3];
  int oxL = 0, oxC = 0;
  auto issue = [&](int l) {                            // the list's reference block of this lane's group: luma row gl, chroma (plane, row)
    const int slot = l ? slot1 : slot0;
    const PlaneRef pl = dpb.p[slot][0], pb = dpb.p[slot][1], pr = dpb.p[slot][2];
    const int sbL = pl.stride * bpp, sbC = pb.stride * bpp;
    const int ofCb = (int)((const char*)pb.ptr - (const char*)pl.ptr), ofCr = (int)((const char*)pr.ptr - (const char*)pl.ptr);
    const __amdgpu_buffer_rsrc_t rs = mc_rsrc(pl.ptr, (uint32_t)(max(ofCb, ofCr) + sbC * cH));
    const int xs = X + (mvx[l] >> 2) - 3, ys = Y + (mvy[l] >> 2) - 3;
    oxL = xs & 3;
    const int vo = (ys + min(gl, MM_LROWS - 1)) * sbL + (xs & ~3) * bpp;
#pragma unroll
    for (int k = 0; k < 5; k++) fl[k] = mc_fetch4<PX>(rs, vo + 4 * k * bpp, 0);
    const int xc = (X >> 1) + (mvx[l] >> 3) - 1, yc = (Y >> 1) + (mvy[l] >> 3) - 1;
    oxC = xc & 3;
    const int vc = (cpl ? ofCr : ofCb) + (yc + min(gl & 7, MM_CROWS - 1)) * sbC + (xc & ~3) * bpp;
#pragma unroll
    for (int k = 0; k < 3; k++) fc[k] = mc_fetch4<PX>(rs, vc + 4 * k * bpp, 0);
  };
  auto commit = [&]() {
    if (gl < MM_LROWS) {
#pragma unroll
      for (int k = 0; k < 5; k++) *reinterpret_cast<uint2*>(&s_inL[g][gl * MM_LP + 4 * k]) = fl[k];
    }
    if ((gl & 7) < MM_CROWS) {
#pragma unroll
      for (int k = 0; k < 3; k++) *reinterpret_cast<uint2*>(&s_inC[g][cpl][(gl & 7) * MM_CP + 4 * k]) = fc[k];
    }
  };
  if (use0) issue(0); else issue(1);
#pragma unroll
  for (int l = 0; l < 2; l++) {
    if (l == 0 ? !use0 : !use1) continue;
    const int oL = oxL, oC = oxC;                      // (of the block being committed)
    commit();
    if (l == 0 && use1) issue(1);                      // the second list's block travels while the first is filtered
    MC_LDS_SYNC();
    // ---- luma: horizontal pass, lane -> row gl (15 rows), 8 outputs
    {
      const uint32_t* th = s_tq + 12 * (mvx[l] & 3);
      uint32_t T[9];
#pragma unroll
      for (int m = 0; m < 9; m++) T[m] = th[m];
      if (gl < MM_LROWS) {
        const uint32_t* rowd = reinterpret_cast<const uint32_t*>(&s_inL[g][gl * MM_LP + (oL & ~1)]);
        uint32_t D[9], E[8];
#pragma unroll
        for (int m = 0; m < 9; m++) D[m] = rowd[m];
        const int sa = (oL & 1) * 16;
#pragma unroll
        for (int m = 0; m < 8; m++) E[m] = __builtin_amdgcn_alignbit(D[m + 1], D[m], sa);
        int o[8];
        mc_h8<0, false>(E, T, o);
        const int pe = ((gl >> 1) * 8) * 2 + (gl & 1), po = (64 + ((gl + 1) >> 1) * 8) * 2 + ((gl + 1) & 1);
#pragma unroll
        for (int j = 0; j < 8; j++) { const uint16_t v = (uint16_t)(o[j] >> (bdL - 8)); pair16[pe + 2 * j] = v; pair16[po + 2 * j] = v; }
      }
    }
    MC_LDS_SYNC();
    {
      const uint32_t* tv = s_tq + 12 * (mvy[l] & 3);
      const int pv = (vr & 1) * 64 + ((vr + (vr & 1)) >> 1) * 8 + vc4;
      int acc[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint4 v = *reinterpret_cast<const uint4*>(&pairw[pv + j * 8]);
        const uint32_t tp = tv[j];
        if (j == 0) { acc[0] = mc_dot2_first(v.x, tp); acc[1] = mc_dot2_first(v.y, tp); acc[2] = mc_dot2_first(v.z, tp); acc[3] = mc_dot2_first(v.w, tp); }
        else {
          acc[0] = mc_dot2(v.x, tp, acc[0]); acc[1] = mc_dot2(v.y, tp, acc[1]);
          acc[2] = mc_dot2(v.z, tp, acc[2]); acc[3] = mc_dot2(v.w, tp, acc[3]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) prL[l][j] = (int16_t)(acc[j] >> 6);
    }
    MC_LDS_SYNC();
    // ---- chroma: horizontal pass, lane -> plane gl>>3, row gl&7 (7 rows), 4 outputs
    {
      const uint32_t* th = s_te + 8 * (mvx[l] & 7);
      uint32_t T[5];
#pragma unroll
      for (int m = 0; m < 5; m++) T[m] = th[m];
      if ((gl & 7) < MM_CROWS) {
        const int r = gl & 7;
        const uint32_t* rowd = reinterpret_cast<const uint32_t*>(&s_inC[g][cpl][r * MM_CP + (oC & ~1)]);
        uint32_t D[5], E[4];
#pragma unroll
        for (int m = 0; m < 5; m++) D[m] = rowd[m];
        const int sa = (oC & 1) * 16;
#pragma unroll
        for (int m = 0; m < 4; m++) E[m] = __builtin_amdgcn_alignbit(D[m + 1], D[m], sa);
        int o[4];
        mc_h4<0, false>(E, T, o);
        // per plane [first copy | second copy] of 4 pair rows x 4 columns
        const int cb = cpl * 32;
        const int pe = (cb + (r >> 1) * 4) * 2 + (r & 1), po = (cb + 16 + ((r + 1) >> 1) * 4) * 2 + ((r + 1) & 1);
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint16_t v = (uint16_t)(o[j] >> (bdC - 8)); pair16[pe + 2 * j] = v; pair16[po + 2 * j] = v; }
      }
    }
    MC_LDS_SYNC();
    {
      const uint32_t* tv = s_te + 8 * (mvy[l] & 7);
      const int pv = cpl * 32 + (cy & 1) * 16 + ((cy + (cy & 1)) >> 1) * 4 + cc2;
      const uint2 v0 = *reinterpret_cast<const uint2*>(&pairw[pv]);
      const uint2 v1 = *reinterpret_cast<const uint2*>(&pairw[pv + 4]);
      const uint32_t q0 = tv[0], q1 = tv[1];
      prC[l][0] = (int16_t)(mc_dot2(v1.x, q1, mc_dot2_first(v0.x, q0)) >> 6);
      prC[l][1] = (int16_t)(mc_dot2(v1.y, q1, mc_dot2_first(v0.y, q0)) >> 6);
    }
    MC_LDS_SYNC();                                     // the input tile and the pair buffers are reused by the next list
  }
  // ---- weighted sample prediction and store
  const int maxL = (1 << bdL) - 1, maxC = (1 << bdC) - 1;
  int ol[4], oc[2];
  if (bi) {
#pragma unroll
    for (int j = 0; j < 4; j++) ol[j] = mc_clip3(0, maxL, (prL[0][j] * w0 + prL[1][j] * w1 + ((o0 + o1 + 1) << log2WD)) >> (log2WD + 1));
#pragma unroll
    for (int j = 0; j < 2; j++) oc[j] = mc_clip3(0, maxC, (prC[0][j] * cw0 + prC[1][j] * cw1 + ((co0 + co1 + 1) << clog2WD)) >> (clog2WD + 1));
  } else {
    const int lu = l_uni;
#pragma unroll
    for (int j = 0; j < 4; j++) ol[j] = mc_clip3(0, maxL, ((prL[lu][j] * w0 + (1 << (log2WD - 1))) >> log2WD) + o0);
#pragma unroll
    for (int j = 0; j < 2; j++) oc[j] = mc_clip3(0, maxC, ((prC[lu][j] * cw0 + (1 << (clog2WD - 1))) >> clog2WD) + co0);
  }
  if (vr < H && vc4 < W) mc_store4<PX>(rdD, (Y + vr) * dsbL + (X + vc4) * bpp, 0, ol);
  if (cy < (H >> 1) && cc2 < (W >> 1))
    mc_store2<PX>(rdD, (cpl ? dofCr : dofCb) + ((Y >> 1) + cy) * dsbC + ((X >> 1) + cc2) * bpp, 0, oc[0], oc[1]);
}
// ---- the three forms in one launch.  The host sorts a picture's MC tasks into eight horizontal bands of CTB rows, one per
// XCD (workgroups b and b+8 share an XCD and its L2): band x = [k_mc's border tiles | the chunk form's chunks | the quad form's quads] of
// that part of the picture, worked off by the workgroups with blockIdx & 7 == x in the order the host lists (`order`): the few
// border tiles first (per-sample clamped fetch: the longest single tasks), then chunks and quads mixed in DECODE order - one
// sweep over the band's reference area (form after form the band was swept three times: 3x the bytes).  Whatever form a task takes,
// its neighbours' overlapping filter margins and the cache lines they write side by side meet in the same L2 (with the
// tasks of one form dealt to the XCDs in list order the micro blocks - sorted by slot pair, not by position - moved 4x the
// algorithmic bytes through the L2s).  The three populations overlap instead of each launch paying its own ramp and tail.
#define MC_ALL_LDS (MC_TILE_LDS > MC_CHUNK_LDS ? (MC_TILE_LDS > MC_MICRO_LDS ? MC_TILE_LDS : MC_MICRO_LDS) : (MC_CHUNK_LDS > MC_MICRO_LDS ? MC_CHUNK_LDS : MC_MICRO_LDS))
#ifndef MC_ALL_WAVES
#define MC_ALL_WAVES 5
#endif
template <typename PX>
__global__ __launch_bounds__(64, MC_ALL_WAVES)
void k_mc_all(PicDev P, DpbTable dpb, PlaneRef d0, PlaneRef d1, PlaneRef d2,
              const McTask* __restrict__ tasks, const de265hip_slice_params* __restrict__ slices, const uint32_t* __restrict__ order, McBands B)
{
  __shared__ __attribute__((aligned(16))) char smem[MC_ALL_LDS];
  const unsigned x = blockIdx.x & 7u, i = blockIdx.x >> 3;
  if (i >= B.n_entries[x]) return;
  const uint32_t e = __builtin_amdgcn_readfirstlane(order[B.order_first[x] + i]);
  const McTask* t = tasks + B.first[x];
  const int nt = (int)B.n_tiles[x], nc = (int)B.n_chunks[x];
  const int idx = (int)(e & 0x3FFFFFFFu);
  if (e >> 31) mc_micro_body<PX>(P, dpb, d0, d1, d2, t + nt + nc, slices, idx, smem);
  else if (e >> 30) mc_chunk_body<PX>(P, dpb, d0, d1, d2, t + nt, slices, idx, smem);
  else mc_tile_body<PX>(P, dpb, d0, d1, d2, t, slices, idx, smem);
}
template __global__ void k_mc_all<uint8_t>(PicDev, DpbTable, PlaneRef, PlaneRef, PlaneRef, const McTask*, const de265hip_slice_params*, const uint32_t*, McBands);
template __global__ void k_mc_all<uint16_t>(PicDev, DpbTable, PlaneRef, PlaneRef, PlaneRef, const McTask*, const de265hip_slice_params*, const uint32_t*, McBands);

// ---- chroma prediction of one MC task for any chroma format (4:2:2 / 4:4:4 pictures; mc_chroma, motion.cc:175-273: the
// vector scaled by 2 / SubWidthC, 2 / SubHeightC, eighth-sample fractions).  One wavefront per task (k_mc's blockIdx.y == 1),
// the tile of at most 16x16 chroma samples through the filter stages of the function-level block interpolator.
template <typename PX>
__device__ __forceinline__ void mc_chroma_any_body(const PicDev& P, const DpbTable& dpb, const PlaneRef& d1, const PlaneRef& d2, const McTask* __restrict__ tasks,
                                                   const de265hip_slice_params* __restrict__ slices, int tix, char* smem)
{
  uint16_t (*s_in)[19 * MC_IWP] = reinterpret_cast<uint16_t (*)[19 * MC_IWP]>(smem);      // [list * 2 + plane]: the staged input tiles of the task
  int16_t* s_tmp = reinterpret_cast<int16_t*>(smem + 4 * 19 * MC_IWP * 2);
  const int lane = threadIdx.x;
  const McTask t = tasks[tix];
  const de265hip_slice_params* sh = &slices[t.slice_idx];
  const bool use0 = t.slot[0] >= 0, use1 = t.slot[1] >= 0, bi = use0 && use1;
  const int wc = t.w >> P.csw, hc = t.h >> P.csh, xc = t.x >> P.csw, yc = t.y >> P.csh;
  const uint32_t invWc = 65536u / (uint32_t)max(wc, 1) + 1u;             // (s / wc for s < 256, wc <= 16: exact)
  int mode;
  if (sh->slice_type == 1) mode = P.weighted_pred ? 1 : 0;
  else if (bi) mode = P.weighted_bipred ? 3 : 2;
  else mode = P.weighted_bipred ? 1 : 0;
  // Both chroma planes and both lists by the one wavefront that decoded the task, and EVERY reference sample of the task
  // requested before anything is filtered: plane after plane, list after list, each with its own fetch -> LDS -> filter round,
  // a tile was four dependent memory round trips (87 us per 4K10 4:4:4 picture against luma's 34, with half the filter taps)
  constexpr int NLD = 6;                                 // (16 + 3)^2 = 361 input samples of a tile at most: six per lane
  const int IW = wc + 3, IH = hc + 3, nIn = IW * IH;
  const uint32_t invIW = 65536u / (uint32_t)IW + 1u;
  int xI[2] = { 0, 0 }, yI[2] = { 0, 0 }, xF[2] = { 0, 0 }, yF[2] = { 0, 0 };
  PX v[2][2][NLD];                                       // [list][plane][k]
#pragma unroll
  for (int l = 0; l < 2; l++) {
    if (t.slot[l] < 0) continue;
    const int mvx = t.mv[l][0] * (2 >> P.csw), mvy = t.mv[l][1] * (2 >> P.csh);
    xI[l] = xc + (mvx >> 3); yI[l] = yc + (mvy >> 3); xF[l] = mvx & 7; yF[l] = mvy & 7;
    const PlaneRef r1 = dpb.p[t.slot[l]][1], r2 = dpb.p[t.slot[l]][2];
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const int idx = lane + 64 * k;
      if (idx < nIn) {
        const int r = (int)(((uint32_t)idx * invIW) >> 16), c = idx - r * IW;
        const int xA = mc_clip3(0, P.cwidth - 1, xI[l] - 1 + c), yA = mc_clip3(0, P.cheight - 1, yI[l] - 1 + r);
        v[l][0][k] = ((const PX*)r1.ptr)[xA + yA * r1.stride];
        v[l][1][k] = ((const PX*)r2.ptr)[xA + yA * r2.stride];
      }
    }
  }
#pragma unroll
  for (int l = 0; l < 2; l++) {
    if (t.slot[l] < 0) continue;
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const int idx = lane + 64 * k;
      if (idx < nIn) {
        const int r = (int)(((uint32_t)idx * invIW) >> 16), c = idx - r * IW;
        s_in[l * 2 + 0][r * MC_IWP + c] = v[l][0][k];
        s_in[l * 2 + 1][r * MC_IWP + c] = v[l][1][k];
      }
    }
  }
  __syncthreads();
  for (int cp = 0; cp < 2; cp++) {
    int16_t pr[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#pragma unroll
    for (int l = 0; l < 2; l++) {
      if (t.slot[l] < 0) continue;
      mc_block_filter<PX, 4, 4>(xF[l], yF[l], wc, hc, P.bd_chroma, s_in[l * 2 + cp], s_tmp, lane, pr[l]);
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
        const int y = (int)(((uint32_t)s * invWc) >> 16), x = s - y * wc;
        const int a = bi ? pr[0][k] : (use0 ? pr[0][k] : pr[1][k]);
        ((PX*)dc.ptr)[xc + x + (yc + y) * dc.stride] = mc_combine<PX>(mode, a, pr[1][k], P.bd_chroma, w0, o0, w1, o1, log2WD);
      }
    }
  }
}

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
  const int ncomp = P.chroma_format ? 3 : 1;                 // (monochrome: luma samples only, slice.cc:4200)
  for (int comp = 0; comp < ncomp; comp++) {
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
