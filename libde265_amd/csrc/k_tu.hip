// k_tu.hip -- transform-unit kernel for gfx950: intra prediction (border fetch,
// substitution, smoothing, planar/DC/angular) fused with dequantisation,
// inverse DCT/DST / transform-skip / bypass and the residual add.
//
// One 64-lane wavefront (= one workgroup) reconstructs one TU.  Coefficients,
// the intermediate of the separable transform, the neighbour border and the
// prediction live in LDS; the picture is touched once for the border read and
// once for the final store.  Behaviour follows (libde265/):
//   transform.cc:353-625  scale_coefficients_internal (dequant, dispatch)
//   fallback-dct.cc:270-408, :551-692, :80-90, :216-224  (DST, IDCT, skip, bypass)
//   intrapred.cc:395-431, :577-688, :816-1111            (border, filter, predictors)
#include "kernels.h"

namespace d265 {

__device__ __constant__ int8_t c_dct_mat[32 * 32] = {
#include "dct_table.inc"
};
__device__ __constant__ __attribute__((aligned(4))) int8_t c_dst_mat[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };
__device__ __constant__ int8_t c_level_scale[6] = { 40, 45, 51, 57, 64, 72 };
__device__ __constant__ int8_t c_intra_angle[35] = {
  0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
  -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__device__ __constant__ int16_t c_inv_angle[15] = {
  -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// All kernels in this file run ONE wavefront per workgroup, so the phases of a TU only
// need LDS ordering inside the wave.  __syncthreads() would also drain vmcnt (global
// stores of the previous TU, ~1 us each); this waits for LDS only and is a compiler barrier.
#define LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Diagnostic cycle stamps (DE265HIP_DEBUG bit 16): per-phase s_memtime deltas summed into a
// debug buffer that nothing else reads.  stamp == nullptr in normal runs.
// Cycle stamps of the run kernel's phases: a diagnostic that is compiled in only with -DD265_STAMPS
// (DE265HIP_EXTRA_CXXFLAGS) and switched on with DE265HIP_DEBUG=16; otherwise every call is empty.
#ifdef D265_STAMPS
struct Stamper {
  uint32_t* buf; long long t0; int lane;
  uint32_t acc[8];                    // per-phase cycle sums, kept in registers (phase numbers are literals)
  __device__ __forceinline__ void mark(int phase)
  {
    if (buf) {
      LDS_SYNC();
      long long t = clock64();
      acc[phase] += (uint32_t)(t - t0);
      t0 = clock64();
    }
  }
  __device__ __forceinline__ void count() { if (buf) acc[7]++; }
  __device__ __forceinline__ void flush()        // one atomic per phase and run: the stamps must not load the fabric
  {
    if (buf && lane == 0)
      for (int i = 0; i < 8; i++) if (acc[i]) atomicAdd(&buf[i], acc[i]);
  }
};
#else
struct Stamper {
  uint32_t* buf; long long t0; int lane;
  uint32_t acc[8];
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void count() {}
  __device__ __forceinline__ void flush() {}
};
#endif

template <typename PX>
struct TuShared {
  int8_t  mat[32 * 32];        // DCT matrix
  int8_t  dstm[16];            // DST matrix
  int16_t coeff[32 * 32];      // dequantised coefficients
  int16_t g[32 * 32];          // first-stage output
  uint16_t pred[32 * 32];      // intra prediction
  int32_t border[4 * 32 + 4];  // unfiltered neighbours, centre at [64]
  int32_t bfilt[4 * 32 + 4];   // filtered neighbours / angular ref array
  int32_t last_row, last_col;
};

// wave-wide sum over 64 lanes
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename PX, typename SH>
__device__ void intra_predict(const PicDev& P, const TuTask& t, const PX* plane, int stride,
                              SH& S, int lane, Stamper& st)
{
  const int nT = 1 << t.log2_size;
  const int cIdx = t.c_idx;
  const int bd = cIdx ? P.bd_chroma : P.bd_luma;
  const int xB = t.x0, yB = t.y0;
  int* border = &S.border[64];
  int* bf = &S.bfilt[64];
  const uint64_t avail = t.avail;
  const int nLeftUnits = nT >> 1;            // 2nT/4
  const int cornerUnit = nLeftUnits;

  // ---- border fetch + substitution (fill_from_image + reference_sample_substitution)
  for (int p = lane; p <= 4 * nT; p += 64) {
    int i = p - 2 * nT;                      // border index
    int u = (i < 0) ? (p >> 2) : (i == 0 ? cornerUnit : cornerUnit + 1 + ((i - 1) >> 2));
    int val;
    if (avail == 0) val = 1 << (bd - 1);
    else {
      int src = i;
      if (!((avail >> u) & 1)) {
        uint64_t below = avail & ((2ull << u) - 1ull);
        if (below) {
          int su = 63 - __clzll((long long)below);   // nearest available unit before
          // last sample of that unit in scan order
          src = (su < cornerUnit) ? (-2 * nT + 4 * su + 3) : (su == cornerUnit ? 0 : 4 * (su - cornerUnit));
        } else {
          int su = __ffsll((long long)avail) - 1;    // first available unit: its first sample
          src = (su < cornerUnit) ? (-2 * nT + 4 * su) : (su == cornerUnit ? 0 : 4 * (su - cornerUnit - 1) + 1);
        }
      }
      int sx, sy;
      if (src < 0) { sx = xB - 1; sy = yB - src - 1; }
      else if (src == 0) { sx = xB - 1; sy = yB - 1; }
      else { sx = xB + src - 1; sy = yB - 1; }
      val = plane[sx + sy * stride];
    }
    border[i] = val;
  }
  LDS_SYNC();
  st.mark(1);

  // ---- smoothing (intra_prediction_sample_filtering), luma only in 4:2:0
  const int mode = t.intra_mode >= 35 ? 1 : t.intra_mode;
  int filterFlag = 0;
  if (intra_smooth_on(P, cIdx) && mode != 1 && nT != 4) {     // (chroma too in 4:4:4, nobody with intra_smoothing_disabled)
    int minDist = min(abs(mode - 26), abs(mode - 10));
    filterFlag = (nT == 8) ? (minDist > 7) : (nT == 16 ? (minDist > 1) : (minDist > 0));
  }
  if (filterFlag) {
    bool biInt = false;
    if (P.strong_intra && nT == 32 && cIdx == 0) {
      int th = 1 << (P.bd_luma - 5);
      biInt = abs(border[0] + border[64] - 2 * border[32]) < th &&
              abs(border[0] + border[-64] - 2 * border[-32]) < th;
    }
    for (int p = lane; p <= 4 * nT; p += 64) {
      int i = p - 2 * nT;
      int v;
      if (i == -2 * nT || i == 2 * nT) v = border[i];
      else if (biInt) {
        if (i == 0) v = border[0];
        else if (i < 0) v = (PX)(border[0] + (((-i) * (border[-64] - border[0]) + 32) >> 6));
        else v = (PX)(border[0] + ((i * (border[64] - border[0]) + 32) >> 6));
      } else v = (border[i + 1] + 2 * border[i] + border[i - 1] + 2) >> 2;
      bf[i] = v;
    }
    LDS_SYNC();
    for (int p = lane; p <= 4 * nT; p += 64) border[p - 2 * nT] = bf[p - 2 * nT];
    LDS_SYNC();
  }

  st.mark(2);
  const int log2 = t.log2_size;
  const int nS = nT * nT;
  if (mode == 0) {                           // planar
    for (int s = lane; s < nS; s += 64) {
      int y = s >> log2, x = s & (nT - 1);
      S.pred[s] = (uint16_t)(((nT - 1 - x) * border[-1 - y] + (x + 1) * border[1 + nT] +
                              (nT - 1 - y) * border[1 + x] + (y + 1) * border[-1 - nT] + nT) >> (log2 + 1));
    }
  } else if (mode == 1) {                    // DC, wavefront-shuffle reduction
    int v = 0;
    if (lane < nT) v = border[lane + 1] + border[-lane - 1];
    int dc = (wave_sum(v) + nT) >> (log2 + 1);
    bool edge = (cIdx == 0 && nT < 32);
    for (int s = lane; s < nS; s += 64) {
      int y = s >> log2, x = s & (nT - 1);
      int o = dc;
      if (edge) {
        if (x == 0 && y == 0) o = (border[-1] + 2 * dc + border[1] + 2) >> 2;
        else if (y == 0) o = (border[x + 1] + 3 * dc + 2) >> 2;
        else if (x == 0) o = (border[-y - 1] + 3 * dc + 2) >> 2;
      }
      S.pred[s] = (uint16_t)o;
    }
  } else {                                   // angular
    const int angle = c_intra_angle[mode];
    const bool vert = mode >= 18;
    // ref[] in bf (index -nT..2nT)
    for (int x = lane; x <= 2 * nT; x += 64) {
      if (x <= nT || angle >= 0) bf[x] = vert ? border[x] : border[-x];
    }
    if (angle < 0) {
      int inv = c_inv_angle[mode - 11];
      int lo = (nT * angle) >> 5;
      if (lo < -1)
        for (int x = lo + lane; x <= -1; x += 64) {
          int k = (x * inv + 128) >> 8;
          bf[x] = vert ? border[-k] : border[k];
        }
    }
    LDS_SYNC();
    for (int s = lane; s < nS; s += 64) {
      int y = s >> log2, x = s & (nT - 1);
      int a = vert ? y : x, b = vert ? x : y;      // a: along prediction direction
      int iIdx = ((a + 1) * angle) >> 5;
      int iFact = ((a + 1) * angle) & 31;
      int o;
      if (iFact) o = ((32 - iFact) * bf[b + iIdx + 1] + iFact * bf[b + iIdx + 2] + 16) >> 5;
      else o = bf[b + iIdx + 1];
      if (cIdx == 0 && nT < 32 && !(P.implicit_rdpcm && (t.flags & DE265HIP_TU_BYPASS))) {   // intrapred.cc:1102-1104
        if (mode == 26 && x == 0) o = clip3(0, (1 << bd) - 1, border[1] + ((border[-1 - y] - border[0]) >> 1));
        if (mode == 10 && y == 0) o = clip3(0, (1 << bd) - 1, border[-1] + ((border[1 + x] - border[0]) >> 1));
      }
      S.pred[s] = (uint16_t)o;
    }
  }
  LDS_SYNC();
  st.mark(3);
}

// Residual reconstruction from the dense coefficient block in S.coeff.
// kind: 0 inverse DCT, 1 inverse DST 4x4, 2 transform skip, 3 transquant bypass.
// has_pred: prediction comes from S.pred (intra) instead of the picture.
template <typename PX>
__device__ void tu_residual_add(TuShared<PX>& S, int lane, PX* dst, int stride, int log2, int bd,
                                int kind, bool has_pred, int lastRow, int lastCol, int16_t* res_out = nullptr)
{
  const int nT = 1 << log2, nS = nT * nT;
  const int maxv = (1 << bd) - 1;
  if (kind >= 2) {
    const bool bypass = kind == 3;
    const int bdShift = 20 - bd, tsShift = 5 + log2, rnd = 1 << (bdShift - 1);
    for (int s = lane; s < nS; s += 64) {
      int x = s & (nT - 1), y = s >> log2;
      int r = bypass ? (int)S.coeff[s]
                     : (((int32_t)((uint32_t)(int32_t)S.coeff[s] << tsShift) + rnd) >> bdShift);
      if (res_out) { res_out[s] = (int16_t)clip3(-32768, 32767, r); continue; }
      int p = has_pred ? (int)S.pred[s] : (int)dst[x + y * stride];
      dst[x + y * stride] = (PX)clip3(0, maxv, p + r);
    }
    return;
  }

  const int post = 20 - bd, rnd2 = 1 << (post - 1);
  if (kind == 1) {
    if (lane < 16) {
      int i = lane >> 2, c = lane & 3;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) sum += S.dstm[j * 4 + i] * S.coeff[c + j * 4];
      S.g[i * 4 + c] = (int16_t)clip3(-32768, 32767, (sum + 64) >> 7);
    }
    LDS_SYNC();
    if (lane < 16) {
      int y = lane >> 2, i = lane & 3;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) sum += S.dstm[j * 4 + i] * S.g[y * 4 + j];
      int out = clip3(-32768, 32767, (sum + rnd2) >> post);
      if (res_out) res_out[lane] = (int16_t)out;
      else {
        int p = has_pred ? (int)S.pred[lane] : (int)dst[i + y * stride];
        dst[i + y * stride] = (PX)clip3(0, maxv, p + out);
      }
    }
    return;
  }

  // ---- separable inverse DCT; rows/cols beyond the last nonzero are skipped
  const int fact = 32 >> log2;                 // row subsampling of the 32-point matrix
  const int ncols = lastCol + 1;
  for (int tix = lane; tix < nT * ncols; tix += 64) {
    int c = tix % ncols, i = tix / ncols;
    int sum = 0;
    for (int j = 0; j <= lastRow; j++) sum += S.mat[fact * j * 32 + i] * S.coeff[c + j * nT];
    S.g[i * nT + c] = (int16_t)clip3(-32768, 32767, (sum + 64) >> 7);
  }
  LDS_SYNC();
  for (int s = lane; s < nS; s += 64) {
    int y = s >> log2, i = s & (nT - 1);
    int sum = 0;
    for (int j = 0; j <= lastCol; j++) sum += S.mat[fact * j * 32 + i] * S.g[y * nT + j];
    int out = (sum + rnd2) >> post;            // second stage is not clipped (fallback-dct.cc:682)
    // a residual beyond int16 saturates the pixel clip either way, so the stored form may saturate
    if (res_out) { res_out[s] = (int16_t)clip3(-32768, 32767, out); continue; }
    int p = has_pred ? (int)S.pred[s] : (int)dst[i + y * stride];
    dst[i + y * stride] = (PX)clip3(0, maxv, p + out);
  }
}

// One TU on `plane` (a global picture plane or an LDS pixel window; t.x0/t.y0 are
// coordinates in that plane): intra prediction, coefficient scatter + dequantisation,
// residual add.  load_mat: copy the DCT matrix into LDS first (skip when already there).
template <typename PX>
__device__ void tu_reconstruct(const PicDev& P, const TuTask& t, PX* plane, int stride, TuShared<PX>& S,
                               int lane, const int16_t* __restrict__ coeff_val,
                               const uint16_t* __restrict__ coeff_pos, const uint8_t* __restrict__ scaling,
                               bool load_mat, Stamper& st, int16_t* resid = nullptr)
{
  const int cIdx = t.c_idx;
  const int log2 = t.log2_size;
  const int nT = 1 << log2, nS = nT * nT;
  const int bd = cIdx ? P.bd_chroma : P.bd_luma;
  const bool intra = t.flags & DE265HIP_TU_INTRA;
  const bool cbf = t.flags & DE265HIP_TU_CBF;
  PX* dst = plane + t.x0 + t.y0 * stride;

  const bool resid_only = t.flags & D265_TU_RESID_ONLY;
  if (intra && !resid_only) intra_predict<PX, TuShared<PX>>(P, t, plane, stride, S, lane, st);

  if (!cbf) {
    if (intra)
      for (int s = lane; s < nS; s += 64) dst[(s & (nT - 1)) + (s >> log2) * stride] = (PX)S.pred[s];
    return;
  }

  // ---- scatter (dequantised) coefficients into the dense LDS block
  const bool bypass = t.flags & DE265HIP_TU_BYPASS;
  const bool tskip = t.flags & DE265HIP_TU_TSKIP;
  const bool is_dst = (nT == 4 && cIdx == 0 && intra);
  for (int s = lane; s < nS; s += 64) S.coeff[s] = 0;
  if (lane == 0) { S.last_row = 0; S.last_col = 0; }
  if (load_mat && !bypass && !tskip) {
    if (is_dst) { if (lane < 4) ((int32_t*)S.dstm)[lane] = ((const int32_t*)c_dst_mat)[lane]; }
    else for (int s = lane; s < 256; s += 64) ((int32_t*)S.mat)[s] = ((const int32_t*)c_dct_mat)[s];
  }
  LDS_SYNC();
  {
    const int16_t* vals = coeff_val + t.coeff_offset;
    const uint16_t* pos = coeff_pos + t.coeff_offset;
    int lr = 0, lc = 0;
    if (bypass) {
      for (int i = lane; i < t.n_coeff; i += 64) S.coeff[pos[i]] = vals[i];
    } else if (!P.scaling_list) {
      const int bdShift = bd + log2 - 9;
      const int32_t offset = 1 << (bdShift - 1);
      const int32_t fact = (int32_t)c_level_scale[t.qp % 6] << (t.qp / 6);
      for (int i = lane; i < t.n_coeff; i += 64) {
        int p = pos[i];
        int32_t cc = (int32_t)((uint32_t)(int32_t)vals[i] * (uint32_t)fact + (uint32_t)offset);   // 32-bit wrap
        S.coeff[p] = (int16_t)clip3(-32768, 32767, cc >> bdShift);
        lr = max(lr, p >> log2); lc = max(lc, p & (nT - 1));
      }
    } else {
      const int bdShift = bd + log2 - 5;
      const long long offset = 1ll << (bdShift - 1);
      int matrixID = cIdx;
      if (!intra) matrixID += (nT < 32) ? 3 : 1;
      if (nT == 32 && matrixID > 1) matrixID = intra ? 0 : 1;   // (32x32 chroma, 4:4:4: the reference indexes beyond its two 32x32 matrices - undefined there)
      const uint8_t* scl = scaling + (log2 == 2 ? 0 : (log2 == 3 ? 96 : (log2 == 4 ? 96 + 384 : 96 + 384 + 1536))) +
                           matrixID * nS;
      for (int i = lane; i < t.n_coeff; i += 64) {
        int p = pos[i];
        int fact = ((int)scl[p] * c_level_scale[t.qp % 6]) << (t.qp / 6);
        long long cc = ((long long)vals[i] * fact + offset) >> bdShift;
        cc = cc < -32768 ? -32768 : (cc > 32767 ? 32767 : cc);
        S.coeff[p] = (int16_t)cc;
        lr = max(lr, p >> log2); lc = max(lc, p & (nT - 1));
      }
    }
    if (lr) atomicMax(&S.last_row, lr);
    if (lc) atomicMax(&S.last_col, lc);
  }
  LDS_SYNC();
  st.mark(4);

  const int kind = bypass ? 3 : (tskip ? 2 : (is_dst ? 1 : 0));
  tu_residual_add<PX>(S, lane, dst, stride, log2, bd, kind, intra, S.last_row, S.last_col,
                      resid_only ? resid + t.resid_offset : nullptr);
  st.mark(5);
}

// ---------------------------------------------------------------- large-TU residual kernel
// 16x16 and 32x32 TUs are few (~4 % of the TUs) but long: four wavefronts share one TU so that the
// two transform stages (up to 32 k multiply-adds each) finish in a quarter of the time.  Same
// arithmetic as tu_reconstruct's residual path; level-0 work only (inter add / intra residual-only).
template <typename PX>
__device__ __forceinline__ void resid_big_body(const PicDev& P, const PlaneRef& pl0, const PlaneRef& pl1, const PlaneRef& pl2,
                                               const TuTask& t, const int16_t* __restrict__ coeff_val,
                                               const uint16_t* __restrict__ coeff_pos, const uint8_t* __restrict__ scaling,
                                               int16_t* __restrict__ resid, int8_t* s_mat, int16_t* s_c, int16_t* s_g,
                                               int& s_last_row, int& s_last_col)
{
  const int tid = threadIdx.x;
  const int cIdx = t.c_idx, log2 = t.log2_size, nT = 1 << log2, nS = nT * nT;
  const int bd = cIdx ? P.bd_chroma : P.bd_luma, maxv = (1 << bd) - 1;
  const bool intra = t.flags & DE265HIP_TU_INTRA;
  const bool bypass = t.flags & DE265HIP_TU_BYPASS;
  // every global read of the TU goes out first (one memory latency for coefficients and prediction together): the
  // thread's coefficient entry (entries beyond 256 are rare and fetched in the loops below) and its <= 4 prediction samples
  const int16_t* vals = coeff_val + t.coeff_offset;
  const uint16_t* pos = coeff_pos + t.coeff_offset;
  const bool resid_only = t.flags & D265_TU_RESID_ONLY;
  const PlaneRef pr = cIdx == 0 ? pl0 : (cIdx == 1 ? pl1 : pl2);
  PX* dst = (PX*)pr.ptr + t.x0 + t.y0 * pr.stride;
  int v_first = 0, p_first = 0, pred[4] = { 0, 0, 0, 0 };
  if (tid < (int)t.n_coeff) { v_first = vals[tid]; p_first = pos[tid]; }
  if (!resid_only) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int s = tid + 256 * k;
      if (s < nS) pred[k] = dst[(s & (nT - 1)) + (s >> log2) * pr.stride];     // (uniform condition: nS is 256 or 1024)
    }
  }
  for (int s = tid; s < nS; s += 256) s_c[s] = 0;
  ((int32_t*)s_mat)[tid] = ((const int32_t*)c_dct_mat)[tid];
  if (tid == 0) { s_last_row = 0; s_last_col = 0; }
  __syncthreads();
  {
    int lr = 0, lc = 0;
    if (bypass) {
      for (int i = tid; i < t.n_coeff; i += 256) s_c[i == tid ? p_first : pos[i]] = (int16_t)(i == tid ? v_first : vals[i]);
    } else if (!P.scaling_list) {
      const int bdShift = bd + log2 - 9;
      const int32_t fact = (int32_t)c_level_scale[t.qp % 6] << (t.qp / 6);
      for (int i = tid; i < t.n_coeff; i += 256) {
        const int p = i == tid ? p_first : pos[i];
        const int vi = i == tid ? v_first : vals[i];
        const int32_t cc = (int32_t)((uint32_t)(int32_t)vi * (uint32_t)fact + (uint32_t)(1 << (bdShift - 1)));
        s_c[p] = (int16_t)clip3(-32768, 32767, cc >> bdShift);
        lr = max(lr, p >> log2); lc = max(lc, p & (nT - 1));
      }
    } else {
      const int bdShift = bd + log2 - 5;
      int matrixID = cIdx;
      if (!intra) matrixID += (nT < 32) ? 3 : 1;
      if (nT == 32 && matrixID > 1) matrixID = intra ? 0 : 1;
      const uint8_t* scl = scaling + (log2 == 4 ? 96 + 384 : 96 + 384 + 1536) + matrixID * nS;
      for (int i = tid; i < t.n_coeff; i += 256) {
        const int p = i == tid ? p_first : pos[i];
        const int vi = i == tid ? v_first : vals[i];
        const int fact = ((int)scl[p] * c_level_scale[t.qp % 6]) << (t.qp / 6);
        long long cc = ((long long)vi * fact + (1ll << (bdShift - 1))) >> bdShift;
        s_c[p] = (int16_t)(cc < -32768 ? -32768 : (cc > 32767 ? 32767 : cc));
        lr = max(lr, p >> log2); lc = max(lc, p & (nT - 1));
      }
    }
    if (lr) atomicMax(&s_last_row, lr);
    if (lc) atomicMax(&s_last_col, lc);
  }
  __syncthreads();
  int16_t* ro = resid + t.resid_offset;
  if (bypass) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int s = tid + 256 * k;
      if (s >= nS) break;
      const int r = s_c[s];
      if (resid_only) ro[s] = (int16_t)r;
      else dst[(s & (nT - 1)) + (s >> log2) * pr.stride] = (PX)clip3(0, maxv, pred[k] + r);
    }
    return;
  }
  const int fact = 32 >> log2, lastRow = s_last_row, lastCol = s_last_col, ncols = lastCol + 1;
  for (int tix = tid; tix < nT * ncols; tix += 256) {
    const int c = tix % ncols, i = tix / ncols;
    int sum = 0;
    for (int j = 0; j <= lastRow; j++) sum += s_mat[fact * j * 32 + i] * s_c[c + j * nT];
    s_g[i * nT + c] = (int16_t)clip3(-32768, 32767, (sum + 64) >> 7);
  }
  __syncthreads();
  const int post = 20 - bd, rnd2 = 1 << (post - 1);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int s = tid + 256 * k;
    if (s >= nS) break;
    const int y = s >> log2, i = s & (nT - 1);
    int sum = 0;
    for (int j = 0; j <= lastCol; j++) sum += s_mat[fact * j * 32 + i] * s_g[y * nT + j];
    const int out = (sum + rnd2) >> post;
    if (resid_only) ro[s] = (int16_t)clip3(-32768, 32767, out);
    else dst[i + y * pr.stride] = (PX)clip3(0, maxv, pred[k] + out);
  }
}

// ---------------------------------------------------------------- small-TU residual kernel
// 4x4 and 8x8 TUs are ~95 % of all TUs.  One wavefront handles four 4x4 TUs (16 lanes each) or
// one 8x8 TU: every lane owns one coefficient in the scatter and one sample in both transform
// stages, the 4/8-point matrices sit in LDS (64 bytes), nothing else is staged.  Handles the
// level-0 work of run mode: inter TUs (residual added into the picture) and the residual-only
// copies of intra TUs (int16 block into the residual buffer).
// wave-level: `wblock` numbers the wavefronts of this TU size; all LDS pointers are this wavefront's own
// RESID_SPL samples per lane: an 8x8 TU takes 16 lanes, a 4x4 TU 4 lanes, so a wavefront holds 4 / 16 TUs and all small
// TUs of a picture are in flight at once (with one sample per lane a 4K B picture's ~30 000 wavefronts took several rounds).
#ifndef RESID_SPL
#define RESID_SPL 4
#endif
template <typename PX, int LOG2>
__device__ __forceinline__ void resid_small_body(const PicDev& P, const PlaneRef& pl0, const PlaneRef& pl1, const PlaneRef& pl2,
                                                 const TuTask* __restrict__ tasks, int first, int count, int wblock, int lane,
                                                 const int16_t* __restrict__ coeff_val, const uint16_t* __restrict__ coeff_pos,
                                                 const uint8_t* __restrict__ scaling, int16_t* __restrict__ resid,
                                                 int16_t* s_c_w, int16_t* s_g_w, int8_t* s_dct, int8_t* s_dst)
{
  constexpr int nT = 1 << LOG2, nS = nT * nT, SPL = RESID_SPL, LPT = nS / SPL, TPW = 64 / LPT;   // lanes per TU, TUs per wavefront
  int16_t (*s_c)[nS] = reinterpret_cast<int16_t (*)[nS]>(s_c_w);
  int16_t (*s_g)[nS] = reinterpret_cast<int16_t (*)[nS]>(s_g_w);
  const int sub = lane / LPT, sl = lane % LPT;
  const int tix = wblock * TPW + sub;
  const bool live = tix < count;
  if (lane < nS) s_dct[lane] = c_dct_mat[(32 / nT) * (lane / nT) * 32 + (lane % nT)];
  if (LOG2 == 2 && lane < 16) s_dst[lane] = c_dst_mat[lane];
#pragma unroll
  for (int k = 0; k < SPL; k++) s_c[sub][sl + LPT * k] = 0;
  TuTask t;
  if (live) t = tasks[first + tix];
  LDS_SYNC();
  const int cIdx = live ? t.c_idx : 0;
  const int bd = cIdx ? P.bd_chroma : P.bd_luma;
  const bool intra = live && (t.flags & DE265HIP_TU_INTRA);
  const bool bypass = live && (t.flags & DE265HIP_TU_BYPASS);
  const bool tskip = live && (t.flags & DE265HIP_TU_TSKIP);
  const bool to_pic = live && !(t.flags & D265_TU_RESID_ONLY);
  // every global read of the TU goes out as soon as its record is there (one memory latency for coefficients and
  // prediction together, not two in sequence)
  int p[SPL], v[SPL], pred[SPL];
  bool has_c[SPL];
  PX* d = nullptr; int dstride = 0;
  if (to_pic) {
    const PlaneRef pr = cIdx == 0 ? pl0 : (cIdx == 1 ? pl1 : pl2);
    d = (PX*)pr.ptr + t.x0 + t.y0 * pr.stride; dstride = pr.stride;
  }
#pragma unroll
  for (int k = 0; k < SPL; k++) {
    const int s = sl + LPT * k;
    has_c[k] = live && s < (int)t.n_coeff;
    p[k] = 0; v[k] = 0; pred[k] = 0;
    if (has_c[k]) { p[k] = coeff_pos[t.coeff_offset + s]; v[k] = coeff_val[t.coeff_offset + s]; }
    if (to_pic) pred[k] = d[(s % nT) + (s / nT) * dstride];
  }
#pragma unroll
  for (int k = 0; k < SPL; k++) {
    if (!has_c[k]) continue;
    int out;
    if (bypass) out = v[k];
    else if (!P.scaling_list) {
      const int bdShift = bd + LOG2 - 9;
      const int32_t fact = (int32_t)c_level_scale[t.qp % 6] << (t.qp / 6);
      const int32_t cc = (int32_t)((uint32_t)v[k] * (uint32_t)fact + (uint32_t)(1 << (bdShift - 1)));   // 32-bit wrap
      out = clip3(-32768, 32767, cc >> bdShift);
    } else {
      const int bdShift = bd + LOG2 - 5;
      const int matrixID = cIdx + (intra ? 0 : 3);
      const int m = scaling[(LOG2 == 2 ? 0 : 96) + matrixID * nS + p[k]];
      const int fact = (m * c_level_scale[t.qp % 6]) << (t.qp / 6);
      long long cc = ((long long)v[k] * fact + (1ll << (bdShift - 1))) >> bdShift;
      out = (int)(cc < -32768 ? -32768 : (cc > 32767 ? 32767 : cc));
    }
    s_c[sub][p[k]] = (int16_t)out;
  }
  LDS_SYNC();
  if (!live) return;
  int r[SPL];
  if (bypass) {
#pragma unroll
    for (int k = 0; k < SPL; k++) r[k] = s_c[sub][sl + LPT * k];
  } else if (tskip) {
#pragma unroll
    for (int k = 0; k < SPL; k++)
      r[k] = ((int32_t)((uint32_t)(int32_t)s_c[sub][sl + LPT * k] << (5 + LOG2)) + (1 << (19 - bd))) >> (20 - bd);
  } else {
    const bool is_dst = LOG2 == 2 && cIdx == 0 && intra;
    const int8_t* M = is_dst ? s_dst : s_dct;
#pragma unroll
    for (int k = 0; k < SPL; k++) {     // first stage: sample -> (row i, column c)
      const int s = sl + LPT * k, i = s / nT, c = s % nT;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < nT; j++) sum += M[j * nT + i] * s_c[sub][c + j * nT];
      s_g[sub][i * nT + c] = (int16_t)clip3(-32768, 32767, (sum + 64) >> 7);
    }
    LDS_SYNC();
#pragma unroll
    for (int k = 0; k < SPL; k++) {
      const int s = sl + LPT * k, y = s / nT, i = s % nT;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < nT; j++) sum += M[j * nT + i] * s_g[sub][y * nT + j];
      r[k] = (sum + (1 << (19 - bd))) >> (20 - bd);
      if (is_dst) r[k] = clip3(-32768, 32767, r[k]);             // DST clips its second stage, the DCT does not
    }
  }
#pragma unroll
  for (int k = 0; k < SPL; k++) {
    const int s = sl + LPT * k;
    if (!to_pic) resid[t.resid_offset + s] = (int16_t)clip3(-32768, 32767, r[k]);
    else d[(s % nT) + (s / nT) * dstride] = (PX)clip3(0, (1 << bd) - 1, pred[k] + r[k]);
  }
}

__device__ __forceinline__ int wave_max_dpp(int v);      // (defined with wave_sum_dpp below)

// A 16x16 TU by ONE wavefront (four samples per lane): same arithmetic as resid_big_body, but a picture's ~6 000 TUs of
// this size are all in flight at once as single wavefronts instead of queueing as 4-wavefront workgroups.
// s_c / s_g: 256 int16 each, s_m: the 16x16 matrix M[j][i] = mat_dct[2j][i].
template <typename PX>
__device__ __forceinline__ void resid16_body(const PicDev& P, const PlaneRef& pl0, const PlaneRef& pl1, const PlaneRef& pl2,
                                             const TuTask& t, int lane, const int16_t* __restrict__ coeff_val,
                                             const uint16_t* __restrict__ coeff_pos, const uint8_t* __restrict__ scaling,
                                             int16_t* __restrict__ resid, int16_t* s_c, int16_t* s_g, int8_t* s_m)
{
  constexpr int nT = 16, nS = 256, log2 = 4;
  const int cIdx = t.c_idx;
  const int bd = cIdx ? P.bd_chroma : P.bd_luma, maxv = (1 << bd) - 1;
  const bool intra = t.flags & DE265HIP_TU_INTRA;
  const bool bypass = t.flags & DE265HIP_TU_BYPASS;
  const bool resid_only = t.flags & D265_TU_RESID_ONLY;
  const int16_t* vals = coeff_val + t.coeff_offset;
  const uint16_t* pos = coeff_pos + t.coeff_offset;
  const PlaneRef pr = cIdx == 0 ? pl0 : (cIdx == 1 ? pl1 : pl2);
  PX* dst = (PX*)pr.ptr + t.x0 + t.y0 * pr.stride;
  // every global read first: the lane's first coefficient entry and its four prediction samples
  const int n_coeff = t.n_coeff;
  int v_first = 0, p_first = 0, pred[4] = { 0, 0, 0, 0 };
  if (lane < n_coeff) { v_first = vals[lane]; p_first = pos[lane]; }
  if (!resid_only) {
#pragma unroll
    for (int k = 0; k < 4; k++) { const int s = lane + 64 * k; pred[k] = dst[(s & 15) + (s >> 4) * pr.stride]; }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) s_c[lane + 64 * k] = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { const int e = lane + 64 * k; s_m[e] = c_dct_mat[2 * (e >> 4) * 32 + (e & 15)]; }
  LDS_SYNC();
  int lr = 0, lc = 0;
  {
    const int qp = t.qp;
    if (bypass) {
      for (int i = lane; i < n_coeff; i += 64) s_c[i == lane ? p_first : pos[i]] = (int16_t)(i == lane ? v_first : vals[i]);
    } else if (!P.scaling_list) {
      const int bdShift = bd + log2 - 9;
      const int32_t fact = (int32_t)c_level_scale[qp % 6] << (qp / 6);
      for (int i = lane; i < n_coeff; i += 64) {
        const int p = i == lane ? p_first : pos[i];
        const int vi = i == lane ? v_first : vals[i];
        const int32_t cc = (int32_t)((uint32_t)(int32_t)vi * (uint32_t)fact + (uint32_t)(1 << (bdShift - 1)));
        s_c[p] = (int16_t)clip3(-32768, 32767, cc >> bdShift);
        lr = max(lr, p >> log2); lc = max(lc, p & (nT - 1));
      }
    } else {
      const int bdShift = bd + log2 - 5;
      const int matrixID = cIdx + (intra ? 0 : 3);
      const uint8_t* scl = scaling + (96 + 384) + matrixID * nS;
      for (int i = lane; i < n_coeff; i += 64) {
        const int p = i == lane ? p_first : pos[i];
        const int vi = i == lane ? v_first : vals[i];
        const int fact = ((int)scl[p] * c_level_scale[qp % 6]) << (qp / 6);
        long long cc = ((long long)vi * fact + (1ll << (bdShift - 1))) >> bdShift;
        s_c[p] = (int16_t)(cc < -32768 ? -32768 : (cc > 32767 ? 32767 : cc));
        lr = max(lr, p >> log2); lc = max(lc, p & (nT - 1));
      }
    }
  }
  LDS_SYNC();
  int16_t* ro = resid + t.resid_offset;
  if (bypass) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int s = lane + 64 * k, r = s_c[s];
      if (resid_only) ro[s] = (int16_t)r;
      else dst[(s & 15) + (s >> 4) * pr.stride] = (PX)clip3(0, maxv, pred[k] + r);
    }
    return;
  }
  // the largest row / column with a coefficient, over the wavefront (DPP row shifts + row broadcasts, as wave_sum_dpp)
  const int lastRow = wave_max_dpp(lr), lastCol = wave_max_dpp(lc), ncols = lastCol + 1;
  for (int tix = lane; tix < nT * ncols; tix += 64) {
    const int c = tix % ncols, i = tix / ncols;
    int sum = 0;
    for (int j = 0; j <= lastRow; j++) sum += s_m[j * nT + i] * s_c[c + j * nT];
    s_g[i * nT + c] = (int16_t)clip3(-32768, 32767, (sum + 64) >> 7);
  }
  LDS_SYNC();
  const int post = 20 - bd, rnd2 = 1 << (post - 1);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int s = lane + 64 * k;
    const int y = s >> log2, i = s & (nT - 1);
    int sum = 0;
    for (int j = 0; j <= lastCol; j++) sum += s_m[j * nT + i] * s_g[y * nT + j];
    const int out = (sum + rnd2) >> post;
    if (resid_only) ro[s] = (int16_t)clip3(-32768, 32767, out);
    else dst[i + y * pr.stride] = (PX)clip3(0, maxv, pred[k] + out);
  }
}

#ifndef RESID_BIG_WAVES
#define RESID_BIG_WAVES 8         // wavefronts per SIMD the register allocation of k_resid_big aims at (4: 112 VGPRs, 58 us per 4K B picture; 8: 64 VGPRs + 148 B scratch, 48 us)
#endif
// The residuals of a picture in two launches.  k_resid_big: one 4-wavefront workgroup per 16x16/32x32 TU.
// k_resid_small: one wavefront per workgroup, workgroups [0, n8) an 8x8 TU each, the rest four 4x4 TUs each (one launch
// for both small sizes; a single launch for all three made every workgroup pay the big path's 112 VGPRs and 5 KB of LDS:
// 7.0 instead of 5.3 ms per 48 pictures with three GOP streams in flight).  tasks[] is sorted [32x32 | 16x16 | 8x8 | 4x4].
// ---- coefficient positions, checked where they are cheap to check (de265hip_picture_build enqueues this behind the upload,
// on the copy stream, ahead of the event every kernel of the picture waits for): a position beyond its TU's nT x nT block is
// folded into the block and the decoder's error word is raised (de265hip_decoder_sync: DE265HIP_ERROR_DECODING), so that no
// residual kernel ever indexes LDS out of its block.  On the host this loop cost 12 % of the host stage of a 4K B picture.
__global__ __launch_bounds__(256)
void k_check_coeffs(const TuTask* __restrict__ l0, int n_l0, const TuTask* __restrict__ l0x, int n_l0x, uint16_t* cpos, uint32_t* err)
{
  // sixteen lanes per TU, striding its list (a thread per TU walked up to 1024 positions alone: 138 us per 4K picture)
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15;
  if (i >= n_l0 + n_l0x) return;
  const TuTask* t = i < n_l0 ? l0 + i : l0x + (i - n_l0);
  const unsigned nS = 1u << (2 * t->log2_size), n = t->n_coeff;
  uint16_t* p = cpos + t->coeff_offset;
  bool bad = false;
  for (unsigned k = sub; k < n; k += 16) {
    const unsigned v = p[k];
    if (v >= nS) { p[k] = (uint16_t)(v & (nS - 1)); bad = true; }
  }
  if (bad) atomicOr(err, 1u);
}

template <typename PX>
__global__ __launch_bounds__(256, RESID_BIG_WAVES)
void k_resid_big(PicDev P, PlaneRef pl0, PlaneRef pl1, PlaneRef pl2, const TuTask* __restrict__ tasks, int n_wg, int n_wave,
                 int n8, int n4, const int16_t* __restrict__ coeff_val, const uint16_t* __restrict__ coeff_pos,
                 const uint8_t* __restrict__ scaling, int16_t* __restrict__ resid)
{
  __shared__ __attribute__((aligned(16))) int8_t s_mat[32 * 32];
  __shared__ __attribute__((aligned(16))) int16_t s_c[32 * 32];
  __shared__ __attribute__((aligned(16))) int16_t s_g[32 * 32];
  __shared__ int s_last_row, s_last_col;
  // workgroups [0, n_wg): one TU by the whole workgroup (the 32x32 ones); the rest: four 16x16 TUs, one per wavefront
  // (tasks [n_wg, n_wg + n_wave), no workgroup barrier on that path)
  if ((int)blockIdx.x < n_wg) {
    const TuTask t = tasks[blockIdx.x];
    resid_big_body<PX>(P, pl0, pl1, pl2, t, coeff_val, coeff_pos, scaling, resid, s_mat, s_c, s_g, s_last_row, s_last_col);
  } else {
    // wavefront slots behind the workgroup TUs: [0, n_wave) a 16x16 TU each, then ceil(n8 / RESID_SPL) slots of 8x8 TUs and
    // ceil(n4 / 4 RESID_SPL) slots of 4x4 TUs (n8 = n4 = 0: those have their own launch, k_resid_small)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int k = ((int)blockIdx.x - n_wg) * 4 + wave;
    int16_t* wc = s_c + 256 * wave; int16_t* wg = s_g + 256 * wave; int8_t* wm = s_mat + 256 * wave;
    const int w8 = (n8 + RESID_SPL - 1) / RESID_SPL;
    if (k < n_wave) {
      const TuTask t = tasks[n_wg + k];
      resid16_body<PX>(P, pl0, pl1, pl2, t, lane, coeff_val, coeff_pos, scaling, resid, wc, wg, wm);
    } else if (k < n_wave + w8)
      resid_small_body<PX, 3>(P, pl0, pl1, pl2, tasks, n_wg + n_wave, n8, k - n_wave, lane, coeff_val, coeff_pos, scaling, resid,
                              wc, wg, wm, wm + 64);
    else if (k < n_wave + w8 + (n4 + 4 * RESID_SPL - 1) / (4 * RESID_SPL))
      resid_small_body<PX, 2>(P, pl0, pl1, pl2, tasks, n_wg + n_wave + n8, n4, k - n_wave - w8, lane, coeff_val, coeff_pos, scaling,
                              resid, wc, wg, wm, wm + 64);
  }
}
template __global__ void k_resid_big<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int, int, int, int, const int16_t*, const uint16_t*, const uint8_t*, int16_t*);
template __global__ void k_resid_big<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int, int, int, int, const int16_t*, const uint16_t*, const uint8_t*, int16_t*);

template <typename PX>
__global__ __launch_bounds__(64)
void k_resid_small(PicDev P, PlaneRef pl0, PlaneRef pl1, PlaneRef pl2, const TuTask* __restrict__ tasks, int nbig, int n8, int n4,
                   const int16_t* __restrict__ coeff_val, const uint16_t* __restrict__ coeff_pos,
                   const uint8_t* __restrict__ scaling, int16_t* __restrict__ resid)
{
  __shared__ int16_t s_c[64 * RESID_SPL];
  __shared__ int16_t s_g[64 * RESID_SPL];
  __shared__ int8_t s_m[80];            // 64: M[j][i] = mat_dct[(32/nT)*j][i]; 16: the DST matrix
  const int lane = threadIdx.x;
  const int wg8 = (n8 + RESID_SPL - 1) / RESID_SPL;           // wavefronts of 8x8 TUs (RESID_SPL TUs each), then 4x4 (4 RESID_SPL each)
  if ((int)blockIdx.x < wg8)
    resid_small_body<PX, 3>(P, pl0, pl1, pl2, tasks, nbig, n8, blockIdx.x, lane, coeff_val, coeff_pos, scaling, resid,
                            s_c, s_g, s_m, s_m + 64);
  else
    resid_small_body<PX, 2>(P, pl0, pl1, pl2, tasks, nbig + n8, n4, (int)blockIdx.x - wg8, lane, coeff_val, coeff_pos, scaling,
                            resid, s_c, s_g, s_m, s_m + 64);
}
template __global__ void k_resid_small<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int, int, int, const int16_t*, const uint16_t*, const uint8_t*, int16_t*);
template __global__ void k_resid_small<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int, int, int, const int16_t*, const uint16_t*, const uint8_t*, int16_t*);

// One workgroup (one wavefront) per TU of a dependency level.
template <typename PX>
__global__ __launch_bounds__(64)
void k_tu(PicDev P, PlaneRef pl0, PlaneRef pl1, PlaneRef pl2, const TuTask* __restrict__ tasks,
          int first, const int16_t* __restrict__ coeff_val, const uint16_t* __restrict__ coeff_pos,
          const uint8_t* __restrict__ scaling, int16_t* resid)
{
  __shared__ TuShared<PX> S;
  const TuTask t = tasks[first + blockIdx.x];
  const PlaneRef pr = t.c_idx == 0 ? pl0 : (t.c_idx == 1 ? pl1 : pl2);
  Stamper st{ nullptr, 0, (int)threadIdx.x, {} };
  tu_reconstruct<PX>(P, t, (PX*)pr.ptr, pr.stride, S, threadIdx.x, coeff_val, coeff_pos, scaling, true, st, resid);
}

// ---------------------------------------------------------------- run kernel
// All intra TUs of a picture in ONE launch.  Each wavefront takes a ticket (runs are
// stored in dependency order, producers first), waits for the runs it reads from
// (agent-scope flag + acquire), loads the pixel window of its run into LDS, reconstructs
// the run's TUs serially inside LDS (the z-scan dependency chain never leaves the CU),
// writes every finished TU back and publishes its flag (agent-scope release).
// Deadlock-free for any dispatch order: a run only waits on smaller tickets, and a
// ticket is only taken by a wavefront that is already running.  Spins are bounded.
// Pixel window of a run with a bbox of at most B x B samples: 1 + B + 32 rows, and
// 7 (alignment slack) + 1 + B + 32 columns rounded to 8 (16-byte aligned rows).
#define RUN_TILE_H_OF(B) ((B) + 33)
#define RUN_TILE_P_OF(B) ((((B) + 40) + 7) & ~7)
#define RUN_TILE_P_MAX RUN_TILE_P_OF(64)
#ifndef RUN_POLL_FAST
#define RUN_POLL_FAST 8        // s_sleep units (64 cycles) between the first RUN_POLL_FAST_N polls of a producer's flag, then RUN_POLL_SLOW
#define RUN_POLL_FAST_N 4
#define RUN_POLL_SLOW 16
#endif
// (bound of the dependency waits: the kernel argument spin_limit, RUN_SPIN_LIMIT_DEFAULT polls of ~1 us: a couple of
//  seconds, then the picture fails - error word -> DE265_ERROR_UNSPECIFIED_DECODING_ERROR - instead of hanging)

// 8 consecutive samples <-> 8 x uint16 in LDS
__device__ __forceinline__ uint4 load8_as_u16(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 load8_as_u16(const uint8_t* p)
{
  uint2 r = *reinterpret_cast<const uint2*>(p);
  uint4 o;
  o.x = (r.x & 0xFF) | ((r.x & 0xFF00) << 8);           o.y = ((r.x >> 16) & 0xFF) | ((r.x >> 24) << 16);
  o.z = (r.y & 0xFF) | ((r.y & 0xFF00) << 8);           o.w = ((r.y >> 16) & 0xFF) | ((r.y >> 24) << 16);
  return o;
}
// The same 8 samples by an agent-coherent (sc1) load that bypasses this CU's L1: what the run kernel reads other runs'
// samples with.  Producer: sc1 stores, drained (vmcnt(0)) in every storing wavefront, workgroup barrier, sc1 flag store;
// consumer: sc1 poll of the flag, workgroup barrier, sc1 loads - the hand-off form of MI355X_MICROARCH.md that needs
// neither an L2 write-back on the producer nor an L1 invalidate (buffer_inv sc1: 1.7 us alone on a CU, 4x that with four
// workgroups per CU) on the consumer.  RUN_SC1_WINDOW=0: plain loads behind an agent-scope acquire (the older form).
#ifndef RUN_SC1_WINDOW
#define RUN_SC1_WINDOW 1
#endif
typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned int v4u32_ld __attribute__((ext_vector_type(4)));
#define D265_BUF_SC1 16                          // cache-policy operand of the raw buffer loads: bit 4 = sc1 (gfx940+)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void* plane)
{ return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(plane), 0, 0x7FFFFFFF, 0x00020000); }
__device__ __forceinline__ uint4 load8_as_u16_sc1(__amdgpu_buffer_rsrc_t r, const uint16_t*, int sample)
{
  const v4u32_ld v = __builtin_amdgcn_raw_buffer_load_b128(r, sample * 2, 0, D265_BUF_SC1);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 load8_as_u16_sc1(__amdgpu_buffer_rsrc_t r, const uint8_t*, int sample)
{
  const v2u32 q = __builtin_amdgcn_raw_buffer_load_b64(r, sample, 0, D265_BUF_SC1);
  uint4 o;
  o.x = (q.x & 0xFF) | ((q.x & 0xFF00) << 8);           o.y = ((q.x >> 16) & 0xFF) | ((q.x >> 24) << 16);
  o.z = (q.y & 0xFF) | ((q.y & 0xFF00) << 8);           o.w = ((q.y >> 16) & 0xFF) | ((q.y >> 24) << 16);
  return o;
}
// Write-through (sc1) stores: the run kernel hands these bytes to other CUs without an L2
// write-back fence (relaxed agent-scope atomic store == global_store ... sc1).
__device__ __forceinline__ void store4_from_u16(uint16_t* g, uint2 v)
{
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(g), (unsigned long long)v.x | ((unsigned long long)v.y << 32),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store4_from_u16(uint8_t* g, uint2 v)
{
  __hip_atomic_store(reinterpret_cast<uint32_t*>(g),
                     (v.x & 0xFF) | ((v.x >> 8) & 0xFF00) | ((v.y & 0xFF) << 16) | ((v.y >> 16) << 24),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 8 / 4 consecutive window samples (uint16 in LDS) -> picture, one write-through store
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store8_from_u16(uint16_t* g, uint4 v)
{
  v4u32 d = { v.x, v.y, v.z, v.w };
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(g), "v"(d) : "memory");   // 16-B write-through store
}
__device__ __forceinline__ void store8_from_u16(uint8_t* g, uint4 v)
{
  const uint32_t lo = (v.x & 0xFF) | ((v.x >> 8) & 0xFF00) | ((v.y & 0xFF) << 16) | ((v.y >> 16) << 24);
  const uint32_t hi = (v.z & 0xFF) | ((v.z >> 8) & 0xFF00) | ((v.w & 0xFF) << 16) | ((v.w >> 16) << 24);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(g), (unsigned long long)lo | ((unsigned long long)hi << 32),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A TuTask as the run kernel holds it: every field in a scalar register.  The 32-byte record is fetched with
// two 16-byte scalar loads (field-wise access makes the compiler fetch the byte fields with vector loads,
// each followed by a vmcnt(0) that also waits for the previous TU's write-through stores).
struct RunTu {
  int x0, y0, log2_size, c_idx, flags, intra_mode, angle, inv_angle;
  bool smooth_on, edge_off;       // neighbour smoothing applies to this component; mode 10 / 26 edge filter switched off
  uint32_t resid_offset;
  uint64_t avail;
};
// Staged form in LDS (16 bytes per TU, written once per run by run_tu_pack).  The chain reads x and y only:
//   x: tb (LDS byte address of the TU origin in the window, 15 bits) | is4x4 << 15 | in-run level << 16 | kind << 24 |
//      vertical << 26 | smooth << 27 | big (16x16, 32x32) << 28 | cbf << 29
//      (kind 0 planar, 1 DC, 2 angular, 3 angular with the mode 10/26 edge filter; mode >= 35 folded into DC)
//   y: 4 * (sample offset in the run) (14 bits) | mode << 14 | (uint8)angle << 24
//   z: avail bits 0-31
//   w: avail bit 32 | (offset in the run's residual range) << 1 | window x << 13 | window y << 20 | log2 << 27
#define RTU_IS4 (1u << 15)
#define RTU_VERT (1u << 26)
#define RTU_SMOOTH (1u << 27)
#define RTU_BIG (1u << 28)
#define RTU_CBF (1u << 29)
template <int RUN_TILE_P>
__device__ __forceinline__ uint4 run_tu_pack(const TuTask* __restrict__ tp, int k, int ax0, int wy0, uint32_t res_base, int c,
                                              bool smooth_on, bool implicit_rdpcm)
{
  const uint4* q = reinterpret_cast<const uint4*>(tp + k);
  const uint4 a = q[0], b = q[1];
  const uint32_t xw = (a.x & 0xFFFF) - ax0, yw = (a.x >> 16) - wy0;
  const uint32_t log2 = a.y & 0xFF, flags = (a.y >> 16) & 0xFF;
  uint32_t mode = a.y >> 24; if (mode >= 35) mode = 1;
  const uint32_t level = (a.z >> 8) & 0xFF, samp = a.w;
  const int md = min(abs((int)mode - 26), abs((int)mode - 10));
  // (the mode 10 / 26 edge filter: luma only, and not for a transquant-bypass CU under implicit RDPCM, intrapred.cc:1102-1104)
  const bool edge = c == 0 && md == 0 && log2 < 5 && !(implicit_rdpcm && (flags & DE265HIP_TU_BYPASS));
  const uint32_t kind = mode == 0 ? 0u : (mode == 1 ? 1u : (edge ? 3u : 2u));
  uint4 o;
  o.x = ((yw * RUN_TILE_P + xw) * 2) | (log2 == 2 ? RTU_IS4 : 0u) | (level << 16) | (kind << 24) |
        (mode >= 18 ? RTU_VERT : 0u) |
        ((smooth_on && mode != 1 && (log2 == 3 ? md > 7 : (log2 == 4 ? md > 1 : (log2 == 5 && md > 0)))) ? RTU_SMOOTH : 0u) |
        (log2 > 3 ? RTU_BIG : 0u) | ((flags & DE265HIP_TU_CBF) ? RTU_CBF : 0u);
  o.y = (samp << 2) | (mode << 14) | ((b.w & 0xFF) << 24);
  o.z = b.x;
  o.w = (b.y & 1u) | (((b.z - res_base) & 0xFFFu) << 1) | (xw << 13) | (yw << 20) | (log2 << 27);
  return o;
}
// Intra-only scratch of the run kernel (the residuals were computed beforehand).
struct RunShared {
  uint16_t b0[4 * 32 + 4];     // neighbours as fetched (+ substitution), centre at [64]
  uint16_t b1[4 * 32 + 4];     // smoothed neighbours
};

// wave64 sum with DPP row shifts (no LDS traffic); the total is returned to every lane
__device__ __forceinline__ int wave_sum_dpp(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8  -> lane 15 of each row = row sum
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2,3 -> lane 63 = total
  return __builtin_amdgcn_readlane(v, 63);
}

// wave64 maximum of non-negative values with the same DPP pattern; the result is returned to every lane
__device__ __forceinline__ int wave_max_dpp(int v)
{
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));   // row_shr:1
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));   // row_shr:2
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));   // row_shr:4
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));   // row_shr:8
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true));   // row_bcast:15
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true));   // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

// ---- fast path for 4x4 and 8x8 intra TUs (about 95 % of all TUs) ----
// A run is bound by the latency of its TU chain, and a lone wavefront issues one instruction every ~4 cycles,
// so the chain carries only what depends on pixel values.  Everything that depends on the TU records alone is
// computed beforehand by all threads of the workgroup (run_prepare_sample, typically while the run still waits
// for its producers) and left in LDS per sample:
//   s_ctl  32 bits per sample.  TUs without smoothing: the LDS byte addresses (in the pixel window) of the sample's
//          two predictor operands, substitution of unavailable neighbours (intrapred.cc:395-431) already resolved
//          (a constant cell when nothing is available).  Smoothed TUs (8x8 luma): bits 0-15 the border lanes of the
//          two operands (A*4 | B*4 << 8), bits 16-31 the window address border lane `local` gathers from.
//   s_ex   per TU: the window addresses of the two TU-wide operands (planar: top-right | bottom-left << 16; edge
//          filter: corner)
//   s_res  per sample: residual (0 for TUs without coefficients)
// On the chain, TU without smoothing: two ds_read (operands) -> ~10 VALU -> ds_write: ONE LDS round trip.
// Smoothed TU: ds_read (gather) -> DPP [1 2 1] -> two ds_bpermute -> ~10 VALU -> ds_write.
#define WAVE_BARRIER_ONLY() asm volatile("" ::: "memory")
#define RUN_MAX_TUS 256                 // a run lies inside one 64x64 CTB: at most 256 TUs / levels

// WT: write-through (sc1) store - the bytes are handed to another run through a flag; otherwise a plain store (runs of
// the front kernel: nobody reads them before the kernel boundary)
template <typename PX, bool WT> __device__ __forceinline__ void store4_packed(PX* g, int v0123_lo, int v0123_hi);
template <> __device__ __forceinline__ void store4_packed<uint16_t, true>(uint16_t* g, int lo, int hi)
{ store4_from_u16(g, make_uint2((uint32_t)lo, (uint32_t)hi)); }
template <> __device__ __forceinline__ void store4_packed<uint8_t, true>(uint8_t* g, int lo, int hi)
{ store4_from_u16(g, make_uint2((uint32_t)lo, (uint32_t)hi)); }
template <> __device__ __forceinline__ void store4_packed<uint16_t, false>(uint16_t* g, int lo, int hi)
{ *reinterpret_cast<uint2*>(g) = make_uint2((uint32_t)lo, (uint32_t)hi); }
template <> __device__ __forceinline__ void store4_packed<uint8_t, false>(uint8_t* g, int lo, int hi)
{ *reinterpret_cast<uint32_t*>(g) = ((uint32_t)lo & 0xFF) | (((uint32_t)lo >> 8) & 0xFF00) | (((uint32_t)hi & 0xFF) << 16) | (((uint32_t)hi >> 16) << 24); }

// LDS byte address (inside the pixel window) that border entry p of a TU reads; const_addr when no
// neighbour is available at all.  Border order as intrapred.cc:577-688: p = 0 bottom-left ... 2nT corner ... 4nT.
template <int RUN_TILE_P>
__device__ __forceinline__ int run_gather_addr(int p, int nT, int xB, int yB, uint64_t avail, int const_addr)
{
  if (avail == 0) return const_addr;
  const int C = 2 * nT, i = p - C, cornerUnit = nT >> 1;
  int src = i;
  const int u = (i < 0) ? (p >> 2) : (i == 0 ? cornerUnit : cornerUnit + 1 + ((i - 1) >> 2));
  if (!((avail >> u) & 1)) {                          // nearest available unit before, else the first available one
    const uint64_t below = avail & ((2ull << u) - 1ull);
    if (below) {
      const int su = 63 - __clzll((long long)below);
      src = (su < cornerUnit) ? (-C + 4 * su + 3) : (su == cornerUnit ? 0 : 4 * (su - cornerUnit));
    } else {
      const int su = __ffsll((long long)avail) - 1;
      src = (su < cornerUnit) ? (-C + 4 * su) : (su == cornerUnit ? 0 : 4 * (su - cornerUnit - 1) + 1);
    }
  }
  const int sx = src <= 0 ? xB - 1 : xB + src - 1;
  const int sy = src < 0 ? yB - src - 1 : yB - 1;
  return (sx + sy * RUN_TILE_P) * 2;
}

// Off-chain preparation of sample s of the run (one thread per sample).
template <int RUN_TILE_P>
__device__ __forceinline__ void run_prepare_sample(int s, const uint4* s_task, const uint8_t* s_own, int32_t* s_res,
                                                   uint32_t* s_ctl, uint32_t* s_ex, uint32_t* s_mine, int const_addr)
{
  const int k = s_own[s >> 4];
  const uint4 r = s_task[k];
  const int log2 = (r.w >> 27) & 7, nT = 1 << log2;
  const int samp = (r.y & 0x3FFF) >> 2, local = s - samp;
  const int x = local & (nT - 1), y = local >> log2;
  const int xB = (r.w >> 13) & 0x7F, yB = (r.w >> 20) & 0x7F;
  if ((x & 3) == 0) atomicOr(&s_mine[yB + y], 1u << ((xB + x) >> 2));
  const int kind = (r.x >> 24) & 3;
  const bool vert = r.x & RTU_VERT;
  // the sample's residual and, above it, its angular weight ((x+1) or (y+1)) * angle & 31: the chain's common path then
  // needs neither the TU's angle nor the lane's position
  // (the residual is already there, low half: fetched as one vector per run behind the run record, k_run)
  s_res[s] = (s_res[s] & 0xFFFF) | ((__mul24(vert ? y + 1 : x + 1, (int)(int8_t)(r.y >> 24)) & 31) << 16);
  const int C = 2 * nT;
  int A = C - 1 - y, B = C + 1 + x;                   // planar / DC: A = left[y], B = top[x]
  if (kind >= 2) {                                    // angular (intrapred.cc:903-1069), reference array evaluated in place
    const int mode = (r.y >> 14) & 63;
    const int angle = (int)(int8_t)(r.y >> 24);
    const int inv = (mode >= 11 && mode <= 25 && angle < 0) ? (int)c_inv_angle[mode - 11] : 0;
    const int a = vert ? y : x, b = vert ? x : y;
    const int iIdx = ((a + 1) * angle) >> 5;
    const int i0 = b + iIdx + 1, i1 = i0 + 1;
    const int k0 = i0 >= 0 ? i0 : -((i0 * inv + 128) >> 8);
    const int k1 = min(i1 >= 0 ? i1 : -((i1 * inv + 128) >> 8), C);
    A = vert ? C + k0 : C - k0;
    B = (kind == 3) ? (vert ? C - 1 - y : C + 1 + x)  // operand of the mode 26 / 10 edge filter (iFact == 0: B unused)
                    : (vert ? C + k1 : C - k1);
  }
  const uint64_t avail = (uint64_t)r.z | ((uint64_t)(r.w & 1) << 32);
  if (r.x & RTU_SMOOTH) {
    const uint32_t g = local <= 4 * nT ? (uint32_t)run_gather_addr<RUN_TILE_P>(local, nT, xB, yB, avail, const_addr) : 0u;
    // 4x4 / 8x8: border lanes for ds_bpermute (byte offsets); 16x16 / 32x32: indices into the border array in LDS
    s_ctl[s] = (log2 <= 3 ? (uint32_t)((A << 2) | (B << 10)) : (uint32_t)(A | (B << 8))) | (g << 16);
  } else {
    const uint32_t aA = (uint32_t)run_gather_addr<RUN_TILE_P>(A, nT, xB, yB, avail, const_addr);
    const uint32_t aB = (uint32_t)run_gather_addr<RUN_TILE_P>(B, nT, xB, yB, avail, const_addr);
    s_ctl[s] = aA | (aB << 16);
    if (local == 0 && (kind == 0 || kind == 3)) {
      const uint32_t e0 = (uint32_t)run_gather_addr<RUN_TILE_P>(kind == 0 ? C + 1 + nT : C, nT, xB, yB, avail, const_addr);
      const uint32_t e1 = (uint32_t)run_gather_addr<RUN_TILE_P>(C - 1 - nT, nT, xB, yB, avail, const_addr);
      s_ex[k] = e0 | (e1 << 16);
    }
  }
}

struct RunLane { int x, y, toff2; };                  // per-lane constants of one TU size: sample position, byte offset in the window

// The chain step of a 4x4 / 8x8 TU.  w0: packed record word x (uniform), ctl / rs: this lane's operand word and
// residual, tb: LDS byte address of the TU origin in the window.  A0 / B0: the window samples at the two addresses of
// ctl, read by the caller (the first thing behind the barrier, ahead of its read-ahead for the next TU).
template <int LOG2>
__device__ __forceinline__ void run_chain_small(uint32_t w0, int angle, int c, int maxv, int lane, const RunLane& L,
                                                uint32_t ctl, int rs, char* tile_b, int tb, const uint32_t* s_ex, int k,
                                                int A0, int B0)
{
  constexpr int nT = 1 << LOG2, nS = nT * nT, NB = 4 * nT + 1, C = 2 * nT;
  const int kind = (w0 >> 24) & 3;
  const bool vert = w0 & RTU_VERT;
  int A, B, bv = 0;
  const bool smooth = LOG2 == 3 && (w0 & RTU_SMOOTH);
  if (smooth) {
    bv = B0;
    // [1 2 1] smoothing (intrapred.cc:816-889); both ends keep their value
    const int prev = __builtin_amdgcn_update_dpp(bv, bv, 0x138, 0xf, 0xf, false);   // wave_shr:1 -> lane-1
    const int next = __builtin_amdgcn_update_dpp(bv, bv, 0x130, 0xf, 0xf, false);   // wave_shl:1 -> lane+1
    const int f = (prev + 2 * bv + next + 2) >> 2;
    bv = (lane == 0 || lane >= NB - 1) ? bv : f;
    A = __builtin_amdgcn_ds_bpermute(ctl & 0xFF, bv);
    B = __builtin_amdgcn_ds_bpermute((ctl >> 8) & 0xFF, bv);
  } else {
    A = A0; B = B0;
  }
  int pv;
  if (__builtin_expect(kind == 2, 1)) {
    const int f = __mul24(vert ? L.y + 1 : L.x + 1, angle) & 31;
    pv = (__mul24(A, 32 - f) + __mul24(B, f) + 16) >> 5;
  } else if (kind == 3) {
    const int b0 = smooth ? __builtin_amdgcn_readlane(bv, C)
                          : (int)*reinterpret_cast<uint16_t*>(tile_b + (s_ex[k] & 0xFFFF));
    const int e = clip3(0, maxv, A + ((B - b0) >> 1));
    pv = ((vert ? L.x : L.y) == 0) ? e : A;
  } else if (kind == 0) {
    int tr, bl;
    if (smooth) { tr = __builtin_amdgcn_readlane(bv, C + 1 + nT); bl = __builtin_amdgcn_readlane(bv, C - 1 - nT); }
    else {
      const uint32_t e = s_ex[k];
      tr = *reinterpret_cast<uint16_t*>(tile_b + (e & 0xFFFF)); bl = *reinterpret_cast<uint16_t*>(tile_b + (e >> 16));
    }
    pv = ((nT - 1 - L.x) * A + (L.x + 1) * tr + (nT - 1 - L.y) * B + (L.y + 1) * bl + nT) >> (LOG2 + 1);
  } else {
    // DC: the 2nT neighbours are the A operands of column 0 and the B operands of row 0 (never smoothed)
    const int v = (lane < nS) ? ((L.x == 0 ? A : 0) + (L.y == 0 ? B : 0)) : 0;
    const int dc = (wave_sum_dpp(v) + nT) >> (LOG2 + 1);
    pv = dc;
    if (c == 0) pv = (L.x | L.y) == 0 ? (A + 2 * dc + B + 2) >> 2
                                      : (L.y == 0 ? (B + 3 * dc + 2) >> 2 : (L.x == 0 ? (A + 3 * dc + 2) >> 2 : dc));
  }
  const int outv = clip3(0, maxv, pv + rs);
  // into the window only (lanes beyond the TU mirror lanes of it: same value, same address); the picture is
  // written once, at the end of the run, in whole 16-byte chunks
  *reinterpret_cast<uint16_t*>(tile_b + tb + L.toff2) = (uint16_t)outv;
  WAVE_BARRIER_ONLY();
}

// workgroup barrier that orders LDS traffic only (no vmcnt wait: the chain has no global traffic in flight to order)
#ifndef RUN_READS_FIRST
#define RUN_READS_FIRST 0   // (1: the two chain reads ahead of the read-ahead: measured 1.2 % slower)
#endif
#define RUN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// A 16x16 / 32x32 TU is reconstructed by ALL wavefronts of the workgroup together (measured: done by one wavefront,
// these ~10 % of the TUs took half of an all-intra picture's chain): nS / (64 nw) samples per lane, workgroup barriers
// between the phases.  Same prepared operands as the small TUs; a smoothed TU goes through a border array in LDS
// ([1 2 1] or the 32x32 bilinear filter, intrapred.cc:816-889, straight from the window -> operands by index), the
// others read their operands from the window.  Every wavefront calls this with the same (uniform) arguments.
// All LDS reads of a phase are issued before the first is used: the chain pays one LDS latency per phase.
// (A/B on an all-intra 4K picture: 16x16 and 32x32 together 2.07 ms, only 32x32 together 2.18 ms, none 2.38 ms.)
template <int RUN_TILE_P, int CH>
__device__ __forceinline__ void run_big_predict(uint32_t w0, int angle, int c, int maxv, int tid, int nthr, int log2,
                                                const uint32_t* ctl, const int32_t* res, char* tile_b, int tb,
                                                const uint16_t* bord, int tr, int bl, int b0, int dc)
{
  const int nT = 1 << log2, nS = nT * nT;
  const int kind = (w0 >> 24) & 3;
  const bool vert = w0 & RTU_VERT;
  const bool smooth = w0 & RTU_SMOOTH;
  const bool dc_edge = c == 0 && nT < 32;
  for (int base = tid; base < nS; base += CH * nthr) {
    uint32_t cw[CH]; int rs[CH], A[CH], B[CH];
#pragma unroll
    for (int u = 0; u < CH; u++) { cw[u] = ctl[base + u * nthr]; rs[u] = (int)(int16_t)res[base + u * nthr]; }
#pragma unroll
    for (int u = 0; u < CH; u++) {
      if (smooth) { A[u] = bord[cw[u] & 0xFF]; B[u] = bord[(cw[u] >> 8) & 0xFF]; }
      else { A[u] = *reinterpret_cast<uint16_t*>(tile_b + (cw[u] & 0xFFFF)); B[u] = *reinterpret_cast<uint16_t*>(tile_b + (cw[u] >> 16)); }
    }
#pragma unroll
    for (int u = 0; u < CH; u++) {
      const int s = base + u * nthr;
      const int x = s & (nT - 1), y = s >> log2;
      int pv;
      if (kind == 2) {
        const int f = __mul24(vert ? y + 1 : x + 1, angle) & 31;
        pv = (__mul24(f, B[u] - A[u]) + (A[u] << 5) + 16) >> 5;
      } else if (kind == 3) {
        const int e = clip3(0, maxv, A[u] + ((B[u] - b0) >> 1));
        pv = ((vert ? x : y) == 0) ? e : A[u];
      } else if (kind == 0) {
        pv = ((nT - 1 - x) * A[u] + (x + 1) * tr + (nT - 1 - y) * B[u] + (y + 1) * bl + nT) >> (log2 + 1);
      } else {
        pv = dc;
        if (dc_edge) pv = (x | y) == 0 ? (A[u] + 2 * dc + B[u] + 2) >> 2
                                       : (y == 0 ? (B[u] + 3 * dc + 2) >> 2 : (x == 0 ? (A[u] + 3 * dc + 2) >> 2 : dc));
      }
      *reinterpret_cast<uint16_t*>(tile_b + tb + (y * RUN_TILE_P + x) * 2) = (uint16_t)clip3(0, maxv, pv + rs[u]);
    }
  }
}

template <int RUN_TILE_P>
__device__ __forceinline__ void run_chain_big(const PicDev& P, uint32_t w0, int angle, int c, int maxv, int tid, int nthr, int log2,
                                              const uint32_t* ctl, const int32_t* res, char* tile_b, int tb,
                                              const uint32_t* s_ex, int k, RunShared& S, int* s_dc)
{
  const int nT = 1 << log2, nS = nT * nT, NB = 4 * nT + 1, C = 2 * nT;
  const int kind = (w0 >> 24) & 3;
  const bool smooth = w0 & RTU_SMOOTH;
  const uint16_t* bord = S.b1;
  auto win = [&](uint32_t word) { return (int)*reinterpret_cast<uint16_t*>(tile_b + (word >> 16)); };   // border sample p <- ctl[p]
  if (smooth) {
    const bool strong = P.strong_intra && nT == 32 && c == 0;
    for (int p = tid; p < NB; p += nthr) {
      const uint32_t am = ctl[p ? p - 1 : 0], ac = ctl[p], ap = ctl[p < NB - 1 ? p + 1 : p];
      uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
      if (strong) { a0 = ctl[0]; a1 = ctl[nT]; a2 = ctl[C]; a3 = ctl[3 * nT]; a4 = ctl[4 * nT]; }
      const int vm = win(am), vc = win(ac), vp = win(ap);
      int v = (p == 0 || p == NB - 1) ? vc : (vp + 2 * vc + vm + 2) >> 2;
      if (strong) {
        const int e0 = win(a0), e1 = win(a1), e2 = win(a2), e3 = win(a3), e4 = win(a4);
        const int th = 1 << (P.bd_luma - 5);
        if (abs(e2 + e4 - 2 * e3) < th && abs(e2 + e0 - 2 * e1) < th && p != 0 && p != NB - 1)
          v = p < C ? e2 + (((C - p) * (e0 - e2) + 32) >> 6) : e2 + (((p - C) * (e4 - e2) + 32) >> 6);
      }
      S.b1[p] = (uint16_t)v;
    }
    RUN_LDS_BARRIER();
  }
  int tr = 0, bl = 0, b0 = 0, dc = 0;
  if (kind == 0) {
    if (smooth) { tr = bord[C + 1 + nT]; bl = bord[C - 1 - nT]; }
    else { const uint32_t e = s_ex[k]; tr = *reinterpret_cast<uint16_t*>(tile_b + (e & 0xFFFF)); bl = *reinterpret_cast<uint16_t*>(tile_b + (e >> 16)); }
  } else if (kind == 3) {
    b0 = *reinterpret_cast<uint16_t*>(tile_b + (s_ex[k] & 0xFFFF));
  } else if (kind == 1) {
    // DC (never smoothed): left[y] is the A operand of sample (0, y), top[x] the B operand of sample (x, 0); the first
    // wavefront sums the 2 nT <= 64 of them (one per lane) and shares the mean
    if (tid < 64) {
      int v = 0;
      if (tid < nT) v = *reinterpret_cast<uint16_t*>(tile_b + (ctl[tid << log2] & 0xFFFF));
      else if (tid < 2 * nT) v = *reinterpret_cast<uint16_t*>(tile_b + (ctl[tid - nT] >> 16));
      const int d = (wave_sum_dpp(v) + nT) >> (log2 + 1);
      if (tid == 0) *s_dc = d;
    }
    RUN_LDS_BARRIER();
    dc = *s_dc;
  }
  const int per = nS / nthr;                            // (uniform; a power of two >= 1)
  if (per >= 4) run_big_predict<RUN_TILE_P, 4>(w0, angle, c, maxv, tid, nthr, log2, ctl, res, tile_b, tb, bord, tr, bl, b0, dc);
  else if (per == 2) run_big_predict<RUN_TILE_P, 2>(w0, angle, c, maxv, tid, nthr, log2, ctl, res, tile_b, tb, bord, tr, bl, b0, dc);
  else run_big_predict<RUN_TILE_P, 1>(w0, angle, c, maxv, tid, nthr, log2, ctl, res, tile_b, tb, bord, tr, bl, b0, dc);
  RUN_LDS_BARRIER();                                    // (also: the border array and s_dc are free again)
}

// A run record at a wave-uniform index, as eleven dwords: scalar loads, one round trip (field-wise access makes the
// compiler fetch the byte and short fields with vector loads, in several dependent steps).
__device__ __forceinline__ RunTask load_run_task(const RunTask* __restrict__ runs, uint32_t r)
{
  static_assert(sizeof(RunTask) == 44, "RunTask layout");
  const uint32_t* q = reinterpret_cast<const uint32_t*>(runs + r);
  uint32_t w[11];
#pragma unroll
  for (int i = 0; i < 11; i++) w[i] = __builtin_amdgcn_readfirstlane(q[i]);
  RunTask t;
  __builtin_memcpy(&t, w, sizeof(t));
  return t;
}

// ---- micro runs: <= 16 TUs of <= 8x8 inside a 32x32 box (most runs of a picture with inter PUs) ----
// One wavefront reconstructs the whole run on its own: no workgroup barrier, no staging of operands; a workgroup
// works on four of them at a time (one ticket).  The window is at most 41 rows x 48 columns.
#define MICRO_P 56                       // window pitch (samples): 8 slack + 32 + 8 and a spare chunk, rows 16-byte aligned
#define MICRO_H 41                       // 1 + 32 + 8 rows
#define MICRO_BOX 32
#define MICRO_TUS 16
#define MICRO_SLICE 2520                 // uint16 per wavefront inside the workgroup's window array (>= MICRO_H * MICRO_P)
#define MICRO_RES 1024                   // residual samples per wavefront (MICRO_TUS * 64)

// one 4x4 / 8x8 TU, operands computed on the spot (intrapred.cc:395-431 substitution, :816-889 smoothing,
// :903-1069 predictors); t in window coordinates; rs = this lane's residual
template <int LOG2, typename PX, bool WT>
__device__ __forceinline__ void micro_intra_tu(const RunTu& t, uint16_t* tile, int lane, int rs, int bd, PX* gdst, int gstride)
{
  constexpr int nT = 1 << LOG2, nS = nT * nT, NB = 4 * nT + 1, C = 2 * nT;       // C: lane of border[0]
  const int xB = t.x0, yB = t.y0, cIdx = t.c_idx;
  const uint32_t avail = (uint32_t)t.avail;          // 2nT/4*2 + 1 <= 9 units
  const int maxv = (1 << bd) - 1;
  int bv = 1 << (bd - 1);
  if (avail != 0) {
    const int p = min(lane, NB - 1), i = p - C;
    int src = i;
    if (avail != ((2u << nT) - 1u)) {                 // not every unit available: nearest available one before
      constexpr int cornerUnit = nT >> 1;
      const int u = (i < 0) ? (p >> 2) : (i == 0 ? cornerUnit : cornerUnit + 1 + ((i - 1) >> 2));
      if (!((avail >> u) & 1)) {
        const uint32_t below = avail & ((2u << u) - 1u);
        if (below) {
          const int su = 31 - __clz((int)below);
          src = (su < cornerUnit) ? (-C + 4 * su + 3) : (su == cornerUnit ? 0 : 4 * (su - cornerUnit));
        } else {
          const int su = __ffs((int)avail) - 1;
          src = (su < cornerUnit) ? (-C + 4 * su) : (su == cornerUnit ? 0 : 4 * (su - cornerUnit - 1) + 1);
        }
      }
    }
    const int sx = src <= 0 ? xB - 1 : xB + src - 1;
    const int sy = src < 0 ? yB - src - 1 : yB - 1;
    bv = tile[sx + sy * MICRO_P];
  }
  const int mode = t.intra_mode >= 35 ? 1 : t.intra_mode;
  if (LOG2 == 3 && t.smooth_on && mode != 1 && min(abs(mode - 26), abs(mode - 10)) > 7) {
    const int prev = __builtin_amdgcn_update_dpp(bv, bv, 0x138, 0xf, 0xf, false);   // wave_shr:1 -> lane-1
    const int next = __builtin_amdgcn_update_dpp(bv, bv, 0x130, 0xf, 0xf, false);   // wave_shl:1 -> lane+1
    const int f = (prev + 2 * bv + next + 2) >> 2;
    bv = (lane == 0 || lane >= NB - 1) ? bv : f;
  }
#define BORD(idx) __builtin_amdgcn_ds_bpermute(((idx) + C) << 2, bv)
  const int y = (lane >> LOG2) & (nT - 1), x = lane & (nT - 1);
  int pv;
  if (mode == 0) {
    const int l = BORD(-1 - y), tp = BORD(1 + x);
    const int tr = __builtin_amdgcn_readlane(bv, C + 1 + nT), bl = __builtin_amdgcn_readlane(bv, C - 1 - nT);
    pv = ((nT - 1 - x) * l + (x + 1) * tr + (nT - 1 - y) * tp + (y + 1) * bl + nT) >> (LOG2 + 1);
  } else if (mode == 1) {
    const int tp = BORD(1 + x), l = BORD(-1 - y);
    const int v = (lane >= nT && lane <= 3 * nT && lane != C) ? bv : 0;
    const int dc = (wave_sum_dpp(v) + nT) >> (LOG2 + 1);
    const int corner = (__builtin_amdgcn_readlane(bv, C - 1) + 2 * dc + __builtin_amdgcn_readlane(bv, C + 1) + 2) >> 2;
    pv = dc;
    if (cIdx == 0) pv = (x | y) == 0 ? corner : (y == 0 ? (tp + 3 * dc + 2) >> 2 : (x == 0 ? (l + 3 * dc + 2) >> 2 : dc));
  } else {
    const int angle = t.angle;
    const bool vert = mode >= 18;
    const int inv = t.inv_angle;
    const int a = vert ? y : x, b = vert ? x : y;
    const int iIdx = ((a + 1) * angle) >> 5, iFact = ((a + 1) * angle) & 31;
    const int i0 = b + iIdx + 1, i1 = i0 + 1;
    const int k0 = i0 >= 0 ? i0 : -((i0 * inv + 128) >> 8);
    const int k1 = i1 >= 0 ? i1 : -((i1 * inv + 128) >> 8);
    const int r0 = BORD(vert ? k0 : -k0), r1 = BORD(vert ? min(k1, C) : -min(k1, C));
    const int ev = BORD(vert ? -1 - y : 1 + x);
    pv = ((32 - iFact) * r0 + iFact * r1 + 16) >> 5;
    if (cIdx == 0 && (mode == 26 || mode == 10) && !t.edge_off) {
      const int b0 = __builtin_amdgcn_readlane(bv, C);
      const int b1 = vert ? __builtin_amdgcn_readlane(bv, C + 1) : __builtin_amdgcn_readlane(bv, C - 1);
      const int e = clip3(0, maxv, b1 + ((ev - b0) >> 1));
      pv = (vert ? x == 0 : y == 0) ? e : pv;
    }
  }
#undef BORD
  const int outv = clip3(0, maxv, pv + rs);
  if (lane < nS) tile[xB + x + (yB + y) * MICRO_P] = (uint16_t)outv;
  // four adjacent lanes are packed with two DPP row shifts, every fourth lane issues one write-through store
  const int w01 = outv | (__builtin_amdgcn_update_dpp(0, outv, 0x101, 0xf, 0xf, true) << 16);   // row_shl:1
  const int w23 = __builtin_amdgcn_update_dpp(0, w01, 0x102, 0xf, 0xf, true);                   // row_shl:2
  if (lane < nS && (x & 3) == 0) store4_packed<PX, WT>(gdst + x + y * gstride, w01, w23);
  WAVE_BARRIER_ONLY();
}

// A 16x16 TU inside a micro run (one wavefront): 65 border entries - lanes 0..63 hold border[-32..31] as in
// micro_intra_tu, the top-right end border[32] lives in a register of its own (bx, the same in every lane) - and four
// samples per lane (rows y, y+4, y+8, y+12 of column lane & 15).  Availability: 17 units.
template <typename PX, bool WT>
__device__ __forceinline__ void micro_intra_tu16(const RunTu& t, uint16_t* tile, int lane, const int16_t* res, int bd, PX* gdst, int gstride)
{
  constexpr int nT = 16, NB = 65, C = 32, cornerUnit = 8;
  const int xB = t.x0, yB = t.y0, cIdx = t.c_idx;
  const uint32_t avail = (uint32_t)t.avail;
  const int maxv = (1 << bd) - 1;
  int bv = 1 << (bd - 1), bx = bv;
  if (avail != 0) {
    auto fetch = [&](int p) {
      const int i = p - C;
      int src = i;
      if (avail != ((2u << nT) - 1u)) {               // not every unit available: nearest available one before
        const int u = (i < 0) ? (p >> 2) : (i == 0 ? cornerUnit : cornerUnit + 1 + ((i - 1) >> 2));
        if (!((avail >> u) & 1)) {
          const uint32_t below = avail & ((2u << u) - 1u);
          if (below) {
            const int su = 31 - __clz((int)below);
            src = (su < cornerUnit) ? (-C + 4 * su + 3) : (su == cornerUnit ? 0 : 4 * (su - cornerUnit));
          } else {
            const int su = __ffs((int)avail) - 1;
            src = (su < cornerUnit) ? (-C + 4 * su) : (su == cornerUnit ? 0 : 4 * (su - cornerUnit - 1) + 1);
          }
        }
      }
      const int sx = src <= 0 ? xB - 1 : xB + src - 1;
      const int sy = src < 0 ? yB - src - 1 : yB - 1;
      return (int)tile[sx + sy * MICRO_P];
    };
    bv = fetch(lane);
    bx = fetch(NB - 1);
  }
  const int mode = t.intra_mode >= 35 ? 1 : t.intra_mode;
  if (t.smooth_on && mode != 1 && min(abs(mode - 26), abs(mode - 10)) > 1) {          // [1 2 1], both ends keep their value
    const int prev = __builtin_amdgcn_update_dpp(bv, bv, 0x138, 0xf, 0xf, false);   // wave_shr:1 -> lane-1
    int next = __builtin_amdgcn_update_dpp(bv, bv, 0x130, 0xf, 0xf, false);         // wave_shl:1 -> lane+1
    if (lane == 63) next = bx;
    const int f = (prev + 2 * bv + next + 2) >> 2;
    bv = lane == 0 ? bv : f;
  }
  auto bord = [&](int idx) {                           // border[idx], idx = -32 .. 32
    const int v = __builtin_amdgcn_ds_bpermute((idx + C) << 2, bv);
    return idx == 32 ? bx : v;
  };
  const int x = lane & 15, yq = lane >> 4;
  int dc = 0;
  if (mode == 1) {
    const int v = (lane >= nT && lane <= 3 * nT && lane != C) ? bv : 0;
    dc = (wave_sum_dpp(v) + nT) >> 5;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int y = yq + 4 * k;
    int pv;
    if (mode == 0) {
      const int l = bord(-1 - y), tp = bord(1 + x);
      const int tr = bord(1 + nT), bl = bord(-1 - nT);
      pv = ((nT - 1 - x) * l + (x + 1) * tr + (nT - 1 - y) * tp + (y + 1) * bl + nT) >> 5;
    } else if (mode == 1) {
      const int tp = bord(1 + x), l = bord(-1 - y);
      const int corner = (bord(-1) + 2 * dc + bord(1) + 2) >> 2;
      pv = dc;
      if (cIdx == 0) pv = (x | y) == 0 ? corner : (y == 0 ? (tp + 3 * dc + 2) >> 2 : (x == 0 ? (l + 3 * dc + 2) >> 2 : dc));
    } else {
      const int angle = t.angle;
      const bool vert = mode >= 18;
      const int inv = t.inv_angle;
      const int a = vert ? y : x, b = vert ? x : y;
      const int iIdx = ((a + 1) * angle) >> 5, iFact = ((a + 1) * angle) & 31;
      const int i0 = b + iIdx + 1, i1 = i0 + 1;
      const int k0 = i0 >= 0 ? i0 : -((i0 * inv + 128) >> 8);
      const int k1 = i1 >= 0 ? i1 : -((i1 * inv + 128) >> 8);
      const int r0 = bord(vert ? k0 : -k0), r1 = bord(vert ? min(k1, C) : -min(k1, C));
      pv = ((32 - iFact) * r0 + iFact * r1 + 16) >> 5;
      if (cIdx == 0 && (mode == 26 || mode == 10) && !t.edge_off) {
        const int ev = bord(vert ? -1 - y : 1 + x);
        const int b0 = bord(0), b1 = vert ? bord(1) : bord(-1);
        const int e = clip3(0, maxv, b1 + ((ev - b0) >> 1));
        pv = (vert ? x == 0 : y == 0) ? e : pv;
      }
    }
    const int outv = clip3(0, maxv, pv + (int)res[lane + 64 * k]);
    tile[xB + x + (yB + y) * MICRO_P] = (uint16_t)outv;
    // four adjacent lanes are packed with two DPP row shifts, every fourth lane issues one write-through store
    const int w01 = outv | (__builtin_amdgcn_update_dpp(0, outv, 0x101, 0xf, 0xf, true) << 16);   // row_shl:1
    const int w23 = __builtin_amdgcn_update_dpp(0, w01, 0x102, 0xf, 0xf, true);                   // row_shl:2
    if ((x & 3) == 0) store4_packed<PX, WT>(gdst + x + y * gstride, w01, w23);
  }
  WAVE_BARRIER_ONLY();
}

// Timing-only ablation switches of the run kernel (DE265HIP_DEBUG bits 2, 4, 8, 32, 64, 1024) exist only in builds with
// -DD265_ABLATE: in the shipped kernel every one of them is a branch and code the chain has to step over.
#ifdef D265_ABLATE
#define RUN_DBG dbg
#else
#define RUN_DBG 0
#endif
// FRONT: a run without producers inside the picture's intra graph (k_intra_front): no flags, no hand-off - plain loads and
// stores, nothing to publish.
template <typename PX, bool FRONT>
__device__ __forceinline__ void micro_run(const PicDev& P, const PlaneRef& pl0, const PlaneRef& pl1, const PlaneRef& pl2,
                                          const RunTask* __restrict__ runs, const uint32_t* __restrict__ deps,
                                          uint32_t* sync, uint32_t* err, const TuTask* __restrict__ tasks,
                                          const int16_t* __restrict__ resid, uint16_t* mt, int16_t* mres,
                                          uint32_t r, int lane, uint32_t gen, int dbg, uint32_t spin_limit)
{
  const RunTask run = load_run_task(runs, r);
  const int n_tus = min((int)run.n_tus, MICRO_TUS);
  // one round trip: producer ids (a lane each) and the TU records (a lane each, two 16-byte loads)
  uint32_t dep_id = 0;
  const bool has_dep = !FRONT && lane < (int)run.n_deps;
  if (has_dep) dep_id = deps[run.dep_offset + lane];
  uint4 ra = make_uint4(0, 0, 0, 0), rb = ra;
  if (lane < n_tus) {
    const uint4* q = reinterpret_cast<const uint4*>(tasks + run.first_tu + lane);
    ra = q[0]; rb = q[1];
  }
  const int c = run.c_idx;
  const PlaneRef pr = c == 0 ? pl0 : (c == 1 ? pl1 : pl2);
  PX* plane = (PX*)pr.ptr;
  const int stride = pr.stride;
  const int cw = c ? P.cwidth : P.width, ch = c ? P.cheight : P.height;
  const int bd = c ? P.bd_chroma : P.bd_luma;
  const int wx0 = (int)run.x0 - 1, wy0 = (int)run.y0 - 1;
  const int ax0 = wx0 & ~7;
  const int wx1 = min(min((int)run.wx1, cw), ax0 + MICRO_P), wy1 = min(min((int)run.wy1, ch), wy0 + MICRO_H);
  // The run's residuals (one contiguous int16 vector, by sample of the run: two 16-byte pieces per lane cover the 1024
  // samples a micro run can have) and, for a front run, its window are fetched in the SAME round trip as the TU records:
  // neither needs them.  A run with producers takes its first look at their flags instead, the window follows the flags.
  const int n_samp = min((int)run.n_samples, MICRO_RES);
  uint4 rres[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int i = 8 * (64 * u + lane);
    rres[u] = i < n_samp ? *reinterpret_cast<const uint4*>(resid + run.res_offset + i) : make_uint4(0, 0, 0, 0);
  }
  // window: at most 41 rows x 6 chunks of 8 samples.  A dense run (its TUs cover its box: the usual isolated intra CU) writes
  // every sample of the box before it reads it: only row 0 and the first chunk of the other rows are fetched (as in k_run)
  const bool dense = run.micro & 2;
  const int nchx = (wx1 - ax0 + 7) >> 3, nrows = wy1 - wy0, nchunks = dense ? nchx + nrows - 1 : nchx * nrows;
  const __amdgpu_buffer_rsrc_t wrs = plane_rsrc(plane);
  uint4 wv[4]; int woff[4];
  auto window_issue = [&]() {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int idx = u * 64 + lane;
      woff[u] = -1;
      if (idx < nchunks) {
        int rr, cx;
        if (dense) { rr = idx < nchx ? 0 : idx - nchx + 1; cx = idx < nchx ? idx : 0; }
        else { rr = idx / nchx; cx = idx - rr * nchx; }
        const int gx = ax0 + 8 * cx, gy = wy0 + rr;
        if (gx >= 0 && gy >= 0) {
#if RUN_SC1_WINDOW
          wv[u] = FRONT ? load8_as_u16(plane + gx + gy * stride) : load8_as_u16_sc1(wrs, plane, gx + gy * stride);
#else
          wv[u] = load8_as_u16(plane + gx + gy * stride);
#endif
          woff[u] = rr * MICRO_P + 8 * cx;
        }
      }
    }
  };
  if (FRONT) window_issue();
  uint32_t flag0 = gen;
  if (has_dep) flag0 = __hip_atomic_load(&sync[2 + dep_id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // residual pieces into this wavefront's LDS slice; the pieces of TUs without coefficients are zero (a piece lies inside
  // one TU: TUs hold multiples of 16 samples back to back; its TU = the last one whose first sample is <= the piece's)
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int i = 8 * (64 * u + lane);
    int kk = 0;
    for (int k = 1; k < n_tus; k++) kk = ((int)(__builtin_amdgcn_readlane(ra.w, k) & (MICRO_RES - 1)) <= i) ? k : kk;
    const uint32_t w1 = __builtin_amdgcn_ds_bpermute(kk << 2, (int)ra.y);
    const bool cbf = (w1 >> 16) & DE265HIP_TU_CBF;
    if (i < n_samp) *reinterpret_cast<uint4*>(&mres[i]) = cbf ? rres[u] : make_uint4(0, 0, 0, 0);
  }
  // producers (bounded spin, as in the workgroup path)
  if (!FRONT)
  for (int i = lane; i < (int)run.n_deps; i += 64) {
    const uint32_t* flag = &sync[2 + (i == lane ? dep_id : deps[run.dep_offset + i])];
    uint32_t spins = 0;
    uint32_t f = (i == lane) ? flag0 : gen - 1u;
    while (f != gen && !(RUN_DBG & 32)) {
      if (spins) { if (spins < RUN_POLL_FAST_N) __builtin_amdgcn_s_sleep(RUN_POLL_FAST); else __builtin_amdgcn_s_sleep(RUN_POLL_SLOW); }
      if (++spins > spin_limit) { atomicExch(err, 1u); break; }
      f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#if !RUN_SC1_WINDOW
  if (run.n_deps && !(RUN_DBG & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // this wavefront's own loads follow
#endif
  if (!FRONT) window_issue();
#pragma unroll
  for (int u = 0; u < 4; u++)
    if (woff[u] >= 0) *reinterpret_cast<uint4*>(&mt[woff[u]]) = wv[u];
  LDS_SYNC();
  // the chain
  for (int k = 0; k < n_tus; k++) {
    const uint32_t w0 = __builtin_amdgcn_readlane(ra.x, k), w1 = __builtin_amdgcn_readlane(ra.y, k);
    const uint32_t w3 = __builtin_amdgcn_readlane(ra.w, k);
    const uint32_t w4 = __builtin_amdgcn_readlane(rb.x, k), w7 = __builtin_amdgcn_readlane(rb.w, k);
    RunTu t;
    const int gx0 = w0 & 0xFFFF, gy0 = w0 >> 16;
    t.x0 = gx0 - ax0; t.y0 = gy0 - wy0;
    t.log2_size = w1 & 0xFF; t.c_idx = c; t.flags = (w1 >> 16) & 0xFF; t.intra_mode = w1 >> 24;
    t.avail = w4; t.resid_offset = 0;
    t.smooth_on = intra_smooth_on(P, c); t.edge_off = P.implicit_rdpcm && (t.flags & DE265HIP_TU_BYPASS);
    t.angle = (int)(int8_t)(w7 & 0xFF); t.inv_angle = (int)(int16_t)(w7 >> 16);
    PX* gdst = plane + gx0 + gy0 * stride;
    if (t.log2_size == 4) { micro_intra_tu16<PX, !FRONT>(t, mt, lane, mres + (w3 & (MICRO_RES - 1)), bd, gdst, stride); continue; }
    const int rs = mres[(w3 & (MICRO_RES - 1)) + (lane & ((1 << (2 * t.log2_size)) - 1))];
    if (t.log2_size == 2) micro_intra_tu<2, PX, !FRONT>(t, mt, lane, rs, bd, gdst, stride);
    else micro_intra_tu<3, PX, !FRONT>(t, mt, lane, rs, bd, gdst, stride);
  }
  if (FRONT) return;
  // publish: write-through stores drained, then the flag
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(&sync[2 + r], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- front kernel: the micro runs that read from no other intra run (in a picture with inter PUs: nine runs out of ten -
// isolated intra CUs whose neighbours are inter samples, final since the MC and residual launches).  They need neither
// tickets nor flags, so they do not go through the persistent workers of k_run (where a wavefront works its runs off one
// after the other, each a chain of five dependent memory round trips, at 8 wavefronts per CU): one 64-thread workgroup
// with 7 KB of LDS per run, dispatched by the hardware - every CU holds twenty of them, so all of a 4K B picture's ~6 800
// are in flight within two rounds.  k_run, launched behind it, sees them as finished (the host drops them from the
// producer lists of the runs that read from them).
template <typename PX>
__global__ __launch_bounds__(64)
void k_intra_front(PicDev P, PlaneRef pl0, PlaneRef pl1, PlaneRef pl2, const RunTask* __restrict__ runs,
                   const TuTask* __restrict__ tasks, const int16_t* __restrict__ resid, int n, const uint32_t* __restrict__ front_idx)
{
  __shared__ __attribute__((aligned(16))) uint16_t mt[MICRO_SLICE];
  __shared__ __attribute__((aligned(16))) int16_t mres[MICRO_RES];
  if ((int)blockIdx.x >= n) return;
  // (front_idx: the front runs' ids when the run records are not sorted - the device-side scan; else they are runs [0, n))
  const uint32_t r = front_idx ? __builtin_amdgcn_readfirstlane(front_idx[blockIdx.x]) : blockIdx.x;
  micro_run<PX, true>(P, pl0, pl1, pl2, runs, nullptr, nullptr, nullptr, tasks, resid, mt, mres, r, threadIdx.x, 0u, 0, 0u);
}
template __global__ void k_intra_front<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const RunTask*, const TuTask*, const int16_t*, int, const uint32_t*);
template __global__ void k_intra_front<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const RunTask*, const TuTask*, const int16_t*, int, const uint32_t*);


// Ticket draw on the scalar unit (s_atomic_add ... glc returns the old value through lgkmcnt): unlike a vector
// atomic it does not queue behind the wavefront's outstanding write-through stores (vmcnt).
__device__ __forceinline__ uint32_t run_draw_ticket(uint32_t* counter, uint32_t n)
{
  uint32_t v = n;
  asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(counter) : "memory");
  return v;
}
#define RUN_NO_TICKET 0xFFFFFFFFu

template <typename PX, int BOX>
__global__ __launch_bounds__(64 * RUN_WAVES)
void k_run(PicDev P, PlaneRef pl0, PlaneRef pl1, PlaneRef pl2, const RunTask* __restrict__ runs,
           const uint32_t* __restrict__ deps, uint32_t* sync, uint32_t* err, const TuTask* __restrict__ tasks,
           const int16_t* __restrict__ resid, const uint32_t* __restrict__ slots, int n_batches, int batch, uint32_t ticket_base,
           uint32_t gen, int dbg, uint32_t spin_limit, const uint32_t* __restrict__ mbx, const uint32_t* __restrict__ mbsegs,
           unsigned long long* mb)
{
  constexpr int RUN_TILE_H = RUN_TILE_H_OF(BOX), RUN_TILE_P = RUN_TILE_P_OF(BOX);
  constexpr int MAX_TUS = BOX * BOX / 16;
  constexpr int CONST_ADDR = RUN_TILE_H * RUN_TILE_P * 2;               // the "nothing available" cell behind the window
  __shared__ RunShared S;                         // border arrays of the 16x16 / 32x32 TU being reconstructed
  __shared__ int s_dc;
  __shared__ __attribute__((aligned(16))) uint16_t tile[RUN_TILE_H * RUN_TILE_P + 8];
  __shared__ __attribute__((aligned(16))) int32_t s_res[BOX * BOX + 64];   // residual | angular weight << 16 per sample (micro runs: int16 residual slices)
  __shared__ uint32_t s_ctl[BOX * BOX + 64];      // per sample: operand addresses / border lanes (run_prepare_sample)
  __shared__ uint32_t s_ex[MAX_TUS];              // per TU: window addresses of the TU-wide operands (planar, edge filter)
  __shared__ uint4 s_task[MAX_TUS];               // the run's TUs, packed (run_tu_pack)
  __shared__ uint8_t s_own[BOX * BOX / 16];       // TU that owns each group of 16 samples of the run
  __shared__ uint32_t s_mine[RUN_TILE_H];         // per window row: bit g = the 4 samples at columns 4g.. are the run's
  __shared__ uint32_t s_ticket;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthr = blockDim.x;
  char* tile_b = reinterpret_cast<char*>(tile);
  const RunLane L4{ lane & 3, (lane >> 2) & 3, (((lane >> 2) & 3) * RUN_TILE_P + (lane & 3)) * 2 };
  const RunLane L8{ lane & 7, (lane >> 3) & 7, (((lane >> 3) & 7) * RUN_TILE_P + (lane & 7)) * 2 };
  const uint32_t lane_toff = (uint32_t)L4.toff2 | ((uint32_t)L8.toff2 << 16);
  // persistent workgroup: the grid is only as wide as the picture's widest dependency level
  // (waiting workgroups would just occupy LDS), every one pulls tickets until none are left.
  // No deadlock for any dispatch order: a run only waits on smaller tickets, a ticket is only ever held by a
  // running workgroup, and a finished run's flag is published without waiting on any other run.
  // Memory round trips on a run's path: ticket -> run record -> {TU records, producer ids} -> producer flags ->
  // {window, residuals} -> chain -> store drain; what is inside braces is in flight together.
  // (Raising a finished run's flag one step late, after the next run's first round trip had drained the stores,
  //  gained nothing and can deadlock: the workgroup may itself wait on the flag it still holds.)
  Stamper st{ ((dbg & 16) && wave == 0) ? err + 8 : nullptr, 0, lane, {} };
  uint32_t prev = RUN_NO_TICKET;                                       // finished run whose flag is not raised yet
  uint32_t next_ticket = 0, batch_end = 0;
  st.t0 = clock64();
  // (Experiment: only the workgroups of one XCD (s_getreg HW_REG_XCC_ID) take part and store without write-through, so
  //  that every hand-over stays inside one L2: bit-exact, but an all-intra 4K picture is not faster (2.16 vs 2.13 ms) -
  //  the hand-over between runs is not bound by the trip through the memory side.)
  for (;;) {
  // publish the previous run: its payload was stored write-through (sc1); drained in every wavefront, then the flag.
  // (MI355X_MICROARCH.md, valid forms: sc1 payload stores + vmcnt(0) + flag on the producer,
  //  poll + agent acquire + plain loads on the consumer; no L2 write-back fence needed.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                     // also: the previous run's LDS is free
  if (prev != RUN_NO_TICKET) {
    if (tid == 0) __hip_atomic_store(&sync[2 + prev], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = RUN_NO_TICKET;
  }
  // Tickets are drawn `batch` at a time and worked off in increasing order (the no-deadlock argument holds: the
  // smallest unfinished ticket is held by a running workgroup that waits on nothing larger).  One device-scope
  // counter serves about one add per 12 ns: with thousands of small runs the draw rate itself is the limit,
  // so throughput-bound pictures draw several at once.
  // (Two ways of taking the draw's round trip off the path, both measured slower on 4K pictures: drawing between a
  //  run's stores and their drain - the flag then waits for the atomic: all-intra 2.06 -> 2.47 ms, B 100 -> 111 us;
  //  drawing one ticket ahead by the last wavefront while the others start the current run - a held ticket delays its
  //  dependants: B 100 -> 106-114 us.)
  // batch == 0: DIRECT mode for wide pictures (a B picture's thousands of independent small runs): the grid has one
  // workgroup per ticket, ticket = blockIdx.x, no counter, no loop - the hardware's workgroup dispatcher fills the GPU
  // instead of a few hundred persistent workgroups that each work ~6 tickets off in sequence (each a chain of dependent
  // memory round trips).  Still no deadlock: every XCD dispatches its share of the grid in increasing order, so the
  // workgroup of the smallest unfinished ticket is resident or next in line on its XCD, and it waits on nothing larger;
  // spins are bounded anyway (err word).  Deep pictures (an I picture's long chains) keep the persistent mode: there a
  // grid of thousands of waiting workgroups would only occupy LDS.
  if (batch == 0) {
    if (next_ticket) break;                                            // second pass: the flag of this workgroup's run is out
    next_ticket = 1;
    batch_end = 0xFFFFFFFFu;
  } else if (next_ticket == batch_end) {
    if (wave == 0) { const uint32_t t = run_draw_ticket(sync, (uint32_t)batch) - ticket_base; if (lane == 0) s_ticket = t; }
    __syncthreads();
    next_ticket = __builtin_amdgcn_readfirstlane(s_ticket);
    batch_end = next_ticket + batch;
  }
  st.mark(6);
  const uint32_t bticket = batch == 0 ? blockIdx.x : next_ticket++;    // uniform: scalar loads/branches below
  if (bticket >= (uint32_t)n_batches) break;
  // a ticket is RUN_TICKET_SLOTS slots: that many micro runs (a wavefront each, no workgroup barrier; slots q, q+4, ..
  // one after the other by the same wavefront: fewer draws per run; the runs of a ticket follow each other in the run
  // order, so a slot never waits for a later one), or one ordinary run in slot 0
  uint32_t slv[RUN_TICKET_SLOTS];
#pragma unroll
  for (int i = 0; i < RUN_TICKET_SLOTS / 4; i++) {
    const uint4 v = *reinterpret_cast<const uint4*>(slots + RUN_TICKET_SLOTS * bticket + 4 * i);
    slv[4 * i] = v.x; slv[4 * i + 1] = v.y; slv[4 * i + 2] = v.z; slv[4 * i + 3] = v.w;
  }
  if (slv[0] & 0x80000000u) {
    for (int q = wave; q < RUN_TICKET_SLOTS; q += (nthr >> 6)) {
      uint32_t mine = slv[0];
#pragma unroll
      for (int i = 1; i < RUN_TICKET_SLOTS; i++) mine = q == i ? slv[i] : mine;     // (scalar selects: no register array)
      if (mine != 0xFFFFFFFFu)
        micro_run<PX, false>(P, pl0, pl1, pl2, runs, deps, sync, err, tasks, resid, tile + (q & 3) * MICRO_SLICE,
                      reinterpret_cast<int16_t*>(s_res) + (q & 3) * MICRO_RES, mine & 0x7FFFFFFFu, lane, gen, dbg, spin_limit);
    }
    continue;
  }
  const uint32_t ticket = slv[0];
  const RunTask run = load_run_task(runs, ticket);
  st.mark(0);
  // Edge mailboxes (host.hip): a run marked 8 stores its bottom row and right column as packets (two samples | generation << 32)
  // the moment its chain ends; a run marked 4 reads every neighbour sample from such packets instead of waiting for the
  // producers' flags (which follow the drain of ALL their stores) and then fetching its window from the picture: one memory
  // round trip between two dependent runs instead of three.
  const bool mb_cons = (run.micro & 4) && !(RUN_DBG & 2048), mb_pub = (run.micro & 8) != 0;
  uint32_t mb_own = 0, mb_seg0 = 0, mb_rdy0 = 0xFFFFFFFFu;
  if (run.micro & 12) {
    mb_own = __builtin_amdgcn_readfirstlane(mbx[3 * ticket]); mb_seg0 = __builtin_amdgcn_readfirstlane(mbx[3 * ticket + 1]);
    mb_rdy0 = __builtin_amdgcn_readfirstlane(mbx[3 * ticket + 2]);
  }
  // Phased hand-over (host.hip): a publishing run with a table of ready epochs stores each packet behind the barrier epoch
  // that completes the TU under it; a reading run with several sample groups fetches group g > 0 at poll epoch g of its chain
  // (group 0 before the chain), prefetched at the start.
  uint32_t mb_ready = 255;                                   // this thread's packet (tid < 64): the epoch whose barrier completes it
  if (mb_pub && mb_rdy0 != 0xFFFFFFFFu && tid < 64) mb_ready = (mbsegs[mb_rdy0 + (tid >> 2)] >> (8 * (tid & 3))) & 0xFF;
  const bool mb_pub_phased = mb_pub && mb_rdy0 != 0xFFFFFFFFu;
  uint32_t mb_pubs = 0xFFFFFFFFu;                            // its store points (epochs, ascending; 255: none)
  if (mb_pub_phased) mb_pubs = __builtin_amdgcn_readfirstlane(mbsegs[mb_rdy0 + 16]);
  uint32_t mb_ends = 0, mb_polls = 0; int mb_ngroups = 1;
  if (mb_cons) {
    mb_ngroups = (int)((__builtin_amdgcn_readfirstlane(mbsegs[mb_seg0]) >> 8) & 0xFF);
    mb_ends = __builtin_amdgcn_readfirstlane(mbsegs[mb_seg0 + 1]); mb_polls = __builtin_amdgcn_readfirstlane(mbsegs[mb_seg0 + 2]);
  }
  const bool has_dep = tid < (int)run.n_deps && !mb_cons;
  uint32_t dep_id = 0;
  if (has_dep) dep_id = deps[run.dep_offset + tid];
  // neighbour sample s of the run (s counts through its segments): packet address, which half, where it goes in the window
  auto mb_locate = [&](int s_, const unsigned long long*& src_p, int& half, int& dst) {
    const int nseg = (int)(__builtin_amdgcn_readfirstlane(mbsegs[mb_seg0]) & 0xFF);
    int acc = 0;
    src_p = nullptr;
    for (int q = 0; q < nseg; q++) {
      const uint32_t a = __builtin_amdgcn_readfirstlane(mbsegs[mb_seg0 + 3 + 2 * q]), b = __builtin_amdgcn_readfirstlane(mbsegs[mb_seg0 + 4 + 2 * q]);
      const int cnt = (int)((a >> 24) & 63) + 1, off = s_ - acc;
      if (off >= 0 && off < cnt) {
        const bool col = a >> 31;
        const int src = (int)(b & 63) + off;
        src_p = mb + (size_t)(a & 0xFFFFFFu) * 64 + (col ? 32 : 0) + (src >> 1);
        half = src & 1;
        dst = (int)(b >> 8) + off * (col ? RUN_TILE_P : 1);
      }
      acc += cnt;
    }
    return acc;                                        // samples in all segments
  };
  const unsigned long long* mb_src = nullptr; int mb_half = 0, mb_dst = 0, mb_total = 0;
  if (mb_cons) mb_total = mb_locate(tid, mb_src, mb_half, mb_dst);

  const int c = run.c_idx;
  const PlaneRef pr = c == 0 ? pl0 : (c == 1 ? pl1 : pl2);
  PX* plane = (PX*)pr.ptr;
  const int stride = pr.stride;
  const int cw = c ? P.cwidth : P.width, ch = c ? P.cheight : P.height;
  const int bd = c ? P.bd_chroma : P.bd_luma;
  const int maxv = (1 << bd) - 1;
  // pixel window: bbox + 1 left/top + the top-right / bottom-left reach of its TUs (host-computed)
  const int wx0 = (int)run.x0 - 1, wy0 = (int)run.y0 - 1;
  const int wx1 = min((int)run.wx1, cw), wy1 = min((int)run.wy1, ch);
  const int ax0 = wx0 & ~7;                                  // -8 when the run touches the left picture edge
  const int n_tus = min((int)run.n_tus, MAX_TUS), n_samples = min((int)run.n_samples, BOX * BOX);
  const uint32_t res_base = run.res_offset;

  // the run's residuals: one contiguous int16 vector (laid out by sample of the run), the first two 16-byte pieces per
  // thread (= the whole 64x64 run with 256 threads) in flight from here on, together with the TU records
  constexpr int RCH = 2;
  uint4 rres[RCH];
  auto resid_issue = [&](int base) {
#pragma unroll
    for (int u = 0; u < RCH; u++) {
      const int i = base + 8 * (u * nthr + tid);
      rres[u] = i < n_samples ? *reinterpret_cast<const uint4*>(resid + res_base + i) : make_uint4(0, 0, 0, 0);
    }
  };
  auto resid_commit = [&](int base) {                // (behind the barrier that publishes s_task / s_own: blocks of TUs without coefficients are masked)
#pragma unroll
    for (int u = 0; u < RCH; u++) {
      const int i = base + 8 * (u * nthr + tid);
      if (i < n_samples) {
        const bool cbf = s_task[s_own[i >> 4]].x & RTU_CBF;
        const uint4 v = cbf ? rres[u] : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(&s_res[i]) = make_uint4(v.x & 0xFFFF, v.x >> 16, v.y & 0xFFFF, v.y >> 16);
        *reinterpret_cast<uint4*>(&s_res[i + 4]) = make_uint4(v.z & 0xFFFF, v.z >> 16, v.w & 0xFFFF, v.w >> 16);
      }
    }
  };
  resid_issue(0);
  // ---- preparation, independent of the producers: pack the TU records, then one thread per sample
  for (int i = tid; i < RUN_TILE_H; i += nthr) s_mine[i] = 0;
  for (int i = tid; i < n_tus; i += nthr) {
    const uint4 r = run_tu_pack<RUN_TILE_P>(tasks + run.first_tu, i, ax0, wy0, res_base, c, intra_smooth_on(P, c), P.implicit_rdpcm != 0);
    s_task[i] = r;
    const int samp = (r.y & 0x3FFF) >> 2, cells = 1 << (2 * ((r.w >> 27) & 7) - 4);
    for (int q = 0; q < cells; q++) s_own[(samp >> 4) + q] = (uint8_t)i;
  }
  uint32_t flag0 = gen;                                                 // first look at the producers' flags
  if (has_dep) flag0 = __hip_atomic_load(&sync[2 + dep_id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) tile[RUN_TILE_H * RUN_TILE_P] = (uint16_t)(1 << (bd - 1));
  __syncthreads();
  prev = ticket;
  st.mark(1);

  // the window, fetched in aligned 8-sample chunks, up to four loads in flight per lane
  const int nchx = (wx1 - ax0 + 7) >> 3, nrows = wy1 - wy0;
  // a dense run (its TUs cover its whole bounding box, e.g. an all-intra CTB) writes every sample of the box before it
  // reads it: only row 0 and the first chunk of the other rows (left column, below-left reach) are fetched
  const bool dense = run.micro & 2;
  const int nchunks = dense ? nchx + nrows - 1 : nchx * nrows;
  uint4 wv[4]; int woff[4];
  const __amdgpu_buffer_rsrc_t wrs = plane_rsrc(plane);
  auto window_issue = [&](int base, int nth, int id) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int idx = base + u * nth + id;
      woff[u] = -1;
      if (idx < nchunks) {
        int r, cx;
        if (dense) { r = idx < nchx ? 0 : idx - nchx + 1; cx = idx < nchx ? idx : 0; }
        else { r = idx / nchx; cx = idx - r * nchx; }
        const int gx = ax0 + 8 * cx, gy = wy0 + r;
        // the bottom-right 32x32 corner of the window is never read
        if (gx >= 0 && gy >= 0 && !(gx >= (int)run.x1 && gy >= (int)run.y1)) {
#if RUN_SC1_WINDOW
          wv[u] = load8_as_u16_sc1(wrs, plane, gx + gy * stride);
#else
          wv[u] = load8_as_u16(plane + gx + gy * stride);
#endif
          woff[u] = r * RUN_TILE_P + 8 * cx;
        }
      }
    }
  };
  auto window_commit = [&]() {
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (woff[u] >= 0) *reinterpret_cast<uint4*>(&tile[woff[u]]) = wv[u];
  };
  // producers already finished (the usual case when the picture is throughput-bound): fetch the window now,
  // under the preparation of the samples, instead of after it
  const bool early = !(RUN_DBG & 8) && !mb_cons && (run.n_deps == 0 || __syncthreads_and(flag0 == gen));
  if (early) {
#if !RUN_SC1_WINDOW
    if (run.n_deps && !(RUN_DBG & 2)) {
      // one acquire per workgroup (the L1 belongs to the CU): wavefront 0 invalidates and waits for it,
      // the others load behind the barrier
      if (wave == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __syncthreads();
    }
#endif
    // (sc1 window loads: the barrier inside __syncthreads_and above is the one every wavefront's loads must come behind)
    window_issue(0, nthr, tid);
  }
  resid_commit(0);
  for (int base = 8 * RCH * nthr; base < n_samples; base += 8 * RCH * nthr) { resid_issue(base); resid_commit(base); }   // (fewer than 256 threads only)
  __syncthreads();                                   // a sample's residual is written by the thread that fetched its piece
  if (!(RUN_DBG & 64))
  for (int s = tid; s < n_samples; s += nthr)
    run_prepare_sample<RUN_TILE_P>(s, s_task, s_own, s_res, s_ctl, s_ex, s_mine, CONST_ADDR);
  st.mark(2);
  int wbase = 0;
  bool mb_phased = false; int mb_grp = 0; unsigned long long mb_v = 0;
  if (early) { window_commit(); wbase = 4 * nthr; }
  else if (mb_cons) {
    // (several groups only when every sample has a thread of its own: the later groups' packets stay in registers)
    mb_phased = mb_ngroups > 1 && mb_total <= nthr;
    const int g0_end = mb_phased ? (int)(mb_ends & 0xFF) : mb_total;
    if (mb_phased && tid >= g0_end && tid < mb_total && mb_src) {
      mb_grp = 1 + (tid >= (int)((mb_ends >> 8) & 0xFF)) + (tid >= (int)((mb_ends >> 16) & 0xFF));
      mb_v = __hip_atomic_load(mb_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (looked at when its poll epoch comes)
    }
    for (int s_ = tid; s_ < g0_end; s_ += nthr) {       // (one pass with 256 threads: at most 193 samples)
      if (s_ != tid) mb_locate(s_, mb_src, mb_half, mb_dst);
      if (!mb_src) continue;
      uint32_t spins = 0;
      unsigned long long v = __hip_atomic_load(mb_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while ((uint32_t)(v >> 32) != gen) {
        if (spins) { if (spins < RUN_POLL_FAST_N) __builtin_amdgcn_s_sleep(RUN_POLL_FAST); else __builtin_amdgcn_s_sleep(RUN_POLL_SLOW); }
        if (++spins > spin_limit) { atomicExch(err, 1u); break; }               // never hang the grid
        v = __hip_atomic_load(mb_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      tile[mb_dst] = (uint16_t)(v >> (16 * mb_half));
    }
    wbase = nchunks;                                   // (a dense run reads nothing else from outside its box)
  }
  else if (run.n_deps) {
    for (int i = tid; i < run.n_deps; i += nthr) {
      const uint32_t* flag = &sync[2 + (i == tid ? dep_id : deps[run.dep_offset + i])];
      uint32_t spins = 0;
      uint32_t f = (i == tid) ? flag0 : gen - 1u;
      while (f != gen && !(RUN_DBG & 32)) {                                  // (dbg 32: timing-only ablation, ignores producers)
        // back off: hundreds of waiting wavefronts polling at full rate starve the fabric (s_sleep 64 instead of 16 here
        // costs 3 % of an all-intra picture: the flag is seen up to 2 us late)
        if (spins) { if (spins < RUN_POLL_FAST_N) __builtin_amdgcn_s_sleep(RUN_POLL_FAST); else __builtin_amdgcn_s_sleep(RUN_POLL_SLOW); }
        if (++spins > spin_limit) { atomicExch(err, 1u); break; }               // never hang the grid
        f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
#if !RUN_SC1_WINDOW
    if (dense && nchunks <= 256 && !(RUN_DBG & 2)) {
      // the few border chunks of a dense run: the acquiring wavefront fetches them itself, straight behind the invalidate
      // (its own loads need neither the wait for it nor a barrier), instead of acquire -> wait -> barrier -> loads by all
      if (wave == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        window_issue(0, 64, lane); window_commit();
      }
      wbase = nchunks;
    } else if (!(RUN_DBG & 2)) {
      if (wave == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __syncthreads();
    }
#endif
  }
  st.mark(3);
  if (!(RUN_DBG & 8))
  for (int base = wbase; base < nchunks; base += 4 * nthr) { window_issue(base, nthr, tid); window_commit(); }
  __syncthreads();
  st.mark(4);
  auto mb_publish = [&]() {                              // this thread's packet (tid < 64): two samples of the bottom row / right column
    const bool col = tid >= 32;
    const int i = 2 * (tid & 31);
    const int bw = (int)run.x1 - (int)run.x0, bh = (int)run.y1 - (int)run.y0;
    if (i < (col ? bh : bw)) {
      const int r = col ? (int)run.y0 - wy0 + i : (int)run.y1 - 1 - wy0, cx = col ? (int)run.x1 - 1 - ax0 : (int)run.x0 - ax0 + i;
      const uint32_t a = tile[r * RUN_TILE_P + cx], b = col ? tile[(r + 1) * RUN_TILE_P + cx] : tile[r * RUN_TILE_P + cx + 1];
      __hip_atomic_store(mb + (size_t)mb_own * 64 + tid, ((unsigned long long)gen << 32) | (b << 16) | a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  // ---- the dependency chain.  Each wavefront walks its own list of the run's TUs (the host's list schedule keeps
  // a z-scan chain on one wavefront and moves independent branches to the others).  A TU whose producers all come
  // earlier in the own list starts at once (LDS executes a wavefront's operations in order); producers on another
  // wavefront lie a barrier epoch earlier: before a TU of epoch e its wavefront has passed e workgroup barriers.
  // Records are read two TUs ahead and per-lane operands one TU ahead (none of this is waited for on the chain).
  if (!(RUN_DBG & 4)) {
    // this wavefront's list [j0, j1) of the run's TU array, from the four list ends (scalar shifts: no register array)
    const uint64_t we = (uint64_t)run.wave_end[0] | ((uint64_t)run.wave_end[1] << 16) | ((uint64_t)run.wave_end[2] << 32) |
                        ((uint64_t)run.wave_end[3] << 48);
    const int j1 = (int)((we >> (16 * wave)) & 0xFFFF), j0 = wave ? (int)((we >> (16 * wave - 16)) & 0xFFFF) : 0;
    const int n_epochs = run.n_lvls;
    // the 16x16 / 32x32 TUs of the run form a fifth list (level order) that every wavefront walks in lockstep: after the
    // barrier that opens level l, the workgroup reconstructs the level's big TUs together, then every wavefront its own
    // small ones.  (The level of the list's head is kept in a scalar register: the test costs no LDS read.)
    int jc = (int)run.wave_end[3];
    const int jc1 = n_tus;
    uint4 rc = s_task[jc < jc1 ? jc : 0];
    int c_lvl = jc < jc1 ? (int)((__builtin_amdgcn_readfirstlane(rc.x) >> 16) & 0xFF) : -1;
    __builtin_amdgcn_s_setprio(3);                                       // the chain is latency-bound: issue it first
    const char* task_b = reinterpret_cast<const char*>(s_task);
    const char* res_b = reinterpret_cast<const char*>(s_res); const char* ctl_b = reinterpret_cast<const char*>(s_ctl);
    // The record of the wavefront's next TU (~0 behind the last: level 255 never comes), of the one after it, and the
    // per-lane operands of the next one are read ahead (none of this is waited for on the chain).
    const int jl = max(j1 - 1, 0);
    const uint2 r0 = *reinterpret_cast<const uint2*>(task_b + 16 * min(j0, jl));
    uint32_t w0 = j0 < j1 ? __builtin_amdgcn_readfirstlane(r0.x) : ~0u, w1 = __builtin_amdgcn_readfirstlane(r0.y);
    uint2 r_nxt = *reinterpret_cast<const uint2*>(task_b + 16 * min(j0 + 1, jl));
    int sl = lane & ((w0 & RTU_IS4) ? 15 : 63);
    uint32_t ctl = *reinterpret_cast<const uint32_t*>(ctl_b + (w1 & 0x3FFF) + 4 * sl);
    int rs = *reinterpret_cast<const int32_t*>(res_b + (w1 & 0x3FFF) + 4 * sl);      // residual | angular weight << 16
    int j = j0;
#define RUN_CHAIN_HEAD()                                                                                                   \
      /* off the chain: next TU's record -> its per-lane operands; the record after next */                               \
      const uint32_t n0 = j + 1 < j1 ? __builtin_amdgcn_readfirstlane(r_nxt.x) : ~0u, n1 = __builtin_amdgcn_readfirstlane(r_nxt.y); \
      const int nsl = lane & ((n0 & RTU_IS4) ? 15 : 63);                                                                   \
      const uint32_t nctl = *reinterpret_cast<const uint32_t*>(ctl_b + (n1 & 0x3FFF) + 4 * nsl);                           \
      const int nrs = *reinterpret_cast<const int32_t*>(res_b + (n1 & 0x3FFF) + 4 * nsl);                                 \
      r_nxt = *reinterpret_cast<const uint2*>(task_b + 16 * min(j + 2, jl));                                               \
      const int tb = w0 & 0x7FFF;
#define RUN_CHAIN_NEXT() w0 = n0; w1 = n1; ctl = nctl; rs = nrs; j++;
    // One pass per level: the big TUs together, then the own list's TUs of the level, then the barrier that closes it.
    // The own TUs are walked in a tight inner loop over the common kind (angular, not smoothed, 4x4 / 8x8: spelled out
    // with as few instructions as possible, the chain costs about a dozen cycles per instruction) that drops out to the
    // general code for one TU of any other kind: the compiler then lays the common path out contiguously.
    int mb_next = 1;                                                     // next sample group to fetch (phased readers)
    // (one scalar compare per epoch and kind of event: the epoch of the next fetch / the next store point, -1: none)
    int mb_poll_at = (mb_phased && mb_ngroups > 1) ? (int)((mb_polls >> 8) & 0xFF) : -1;
    int mb_pub_at = (mb_pubs & 0xFF) == 255 ? -1 : (int)(mb_pubs & 0xFF);
    for (int epoch = 0;; epoch++) {
      if (epoch == mb_poll_at) {
        // the neighbour samples first read by TUs of this epoch (and later ones): their packets were requested before the chain
        if (mb_grp == mb_next) {
          uint32_t spins = 0;
          while ((uint32_t)(mb_v >> 32) != gen) {
            if (spins) { if (spins < RUN_POLL_FAST_N) __builtin_amdgcn_s_sleep(RUN_POLL_FAST); else __builtin_amdgcn_s_sleep(RUN_POLL_SLOW); }
            if (++spins > spin_limit) { atomicExch(err, 1u); break; }               // never hang the grid
            mb_v = __hip_atomic_load(mb_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          tile[mb_dst] = (uint16_t)(mb_v >> (16 * mb_half));
        }
        mb_next++;
        mb_poll_at = mb_next < mb_ngroups ? (int)((mb_polls >> (8 * mb_next)) & 0xFF) : -1;
        RUN_LDS_BARRIER();
      }
      while (c_lvl == epoch) {
        const uint32_t c0 = __builtin_amdgcn_readfirstlane(rc.x), c1 = __builtin_amdgcn_readfirstlane(rc.y);
        const int log2 = __builtin_amdgcn_readfirstlane((int)((rc.w >> 27) & 7));
        const int k = jc++;
        rc = s_task[jc < jc1 ? jc : 0];                                  // the next head, read under this TU's work
        const int samp = (c1 & 0x3FFF) >> 2;
        run_chain_big<RUN_TILE_P>(P, c0, (int)(int8_t)(c1 >> 24), c, maxv, tid, nthr, log2, s_ctl + samp, s_res + samp, tile_b,
                                  (int)(c0 & 0x7FFF), s_ex, k, S, &s_dc);
        if (RUN_DBG & 256)                                               // (timing-only ablation: the TU twice)
          run_chain_big<RUN_TILE_P>(P, c0, (int)(int8_t)(c1 >> 24), c, maxv, tid, nthr, log2, s_ctl + samp, s_res + samp, tile_b,
                                    (int)(c0 & 0x7FFF), s_ex, k, S, &s_dc);
        c_lvl = jc < jc1 ? (int)((__builtin_amdgcn_readfirstlane(rc.x) >> 16) & 0xFF) : -1;
      }
      // (level and kind of a record in one compare)
      const uint32_t lvl_key = (uint32_t)epoch << 16, fast_key = lvl_key | (2u << 24);
      const uint32_t lvl_mask = 0xFFu << 16, fast_mask = lvl_mask | RTU_BIG | RTU_SMOOTH | (3u << 24);
      while ((w0 & lvl_mask) == lvl_key) {
        while ((w0 & fast_mask) == fast_key) {
          // (the two reads the chain waits for go out first, the read-ahead behind them)
#if RUN_READS_FIRST
          const int A = *reinterpret_cast<uint16_t*>(tile_b + (ctl & 0xFFFF));
          const int B = *reinterpret_cast<uint16_t*>(tile_b + (ctl >> 16));
          RUN_CHAIN_HEAD()
          __builtin_amdgcn_sched_barrier(0);           // (keep the read-ahead in front of the arithmetic that waits for A and B)
#else
          RUN_CHAIN_HEAD()
          const int A = *reinterpret_cast<uint16_t*>(tile_b + (ctl & 0xFFFF));
          const int B = *reinterpret_cast<uint16_t*>(tile_b + (ctl >> 16));
#endif
          // the sample's angular weight comes prepared with its residual; the window offsets of the two TU sizes are the
          // halves of the per-lane constant lane_toff
          const int toff2 = (w0 & RTU_IS4) ? (lane_toff & 0xFFFF) : (lane_toff >> 16);
          const int f = (rs >> 16) & 31;
          const int pv = (__mul24(f, B - A) + (A << 5) + 16) >> 5;
          *reinterpret_cast<uint16_t*>(tile_b + tb + toff2) = (uint16_t)clip3(0, maxv, pv + (int)(int16_t)rs);
          WAVE_BARRIER_ONLY();
          RUN_CHAIN_NEXT()
        }
        if ((w0 & lvl_mask) != lvl_key) break;
#if RUN_READS_FIRST
        const int B0 = *reinterpret_cast<uint16_t*>(tile_b + (ctl >> 16));
        const int A0 = (w0 & RTU_SMOOTH) ? 0 : (int)*reinterpret_cast<uint16_t*>(tile_b + (ctl & 0xFFFF));   // (smoothed: indices, not an address)
        RUN_CHAIN_HEAD()
        __builtin_amdgcn_sched_barrier(0);
#else
        RUN_CHAIN_HEAD()
        const int B0 = *reinterpret_cast<uint16_t*>(tile_b + (ctl >> 16));
        const int A0 = (w0 & RTU_SMOOTH) ? 0 : (int)*reinterpret_cast<uint16_t*>(tile_b + (ctl & 0xFFFF));   // (smoothed: indices, not an address)
#endif
        const int angle = (int)(int8_t)(w1 >> 24);
        if (w0 & RTU_IS4) run_chain_small<2>(w0, angle, c, maxv, lane, L4, ctl, (int)(int16_t)rs, tile_b, tb, s_ex, j, A0, B0);
        else run_chain_small<3>(w0, angle, c, maxv, lane, L8, ctl, (int)(int16_t)rs, tile_b, tb, s_ex, j, A0, B0);
        RUN_CHAIN_NEXT()
      }
      if (epoch >= n_epochs) break;
      RUN_LDS_BARRIER();
      if (RUN_DBG & 128) RUN_LDS_BARRIER();
      if (epoch == mb_pub_at) {                                         // a store point: the packets completed since the last one (tid < 64)
        if (mb_ready == (uint32_t)epoch) mb_publish();
        mb_pubs >>= 8;
        mb_pub_at = (mb_pubs & 0xFF) == 255 ? -1 : (int)(mb_pubs & 0xFF);
      }
    }
#undef RUN_CHAIN_HEAD
#undef RUN_CHAIN_NEXT
    __builtin_amdgcn_s_setprio(0);
  }
  // ---- write the run's samples to the picture: whole 8-sample chunks where both halves are the run's (one
  // 16-byte write-through store), half chunks otherwise; nothing outside the run's own TUs is ever written.
  // The flag is only raised after the run, so nothing is lost by storing here instead of per TU, and a
  // write-through store costs one fabric write whatever its size.
  __syncthreads();
  if (mb_pub && tid < 64 && (!mb_pub_phased || mb_ready == 255 || (RUN_DBG & 4))) mb_publish();      // (the packets no earlier epoch has stored)
  {
    PX* wplane = plane + ax0 + wy0 * stride;                             // picture address of window sample (0, 0)
    const int rows = (int)run.y1 - (int)run.y0, nch = ((int)run.x1 - ax0 + 7) >> 3;
    if (!(RUN_DBG & 1024))
    for (int idx = tid; idx < rows * nch; idx += nthr) {
      const int r = 1 + idx / nch, cx = idx - (r - 1) * nch;
      const uint32_t m = (s_mine[r] >> (2 * cx)) & 3u;
      if (m == 0) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(&tile[r * RUN_TILE_P + 8 * cx]);
      PX* g = wplane + r * stride + 8 * cx;
      if (m == 3) store8_from_u16(g, v);
      else if (m == 1) store4_from_u16(g, make_uint2(v.x, v.y));
      else store4_from_u16(g + 4, make_uint2(v.z, v.w));
    }
  }

  st.mark(5);
  st.count();
  }
  // the last run of this workgroup: drain, then raise its flag
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && prev != RUN_NO_TICKET)
    __hip_atomic_store(&sync[2 + prev], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  st.flush();
}

// Function-level form (acceleration.h:143-178 slot semantics): dense coefficient
// blocks added into a plane.  kind as in tu_residual_add.
template <typename PX>
__global__ __launch_bounds__(64)
void k_fn_residual(int kind, int log2_size, int bit_depth, PX* plane, int stride,
                   const int32_t* __restrict__ xy, const int16_t* __restrict__ coeffs)
{
  __shared__ TuShared<PX> S;
  const int lane = threadIdx.x;
  const int nT = 1 << log2_size, nS = nT * nT;
  const int16_t* c = coeffs + (size_t)blockIdx.x * nS;
  for (int s = lane; s < nS; s += 64) S.coeff[s] = c[s];
  for (int s = lane; s < 256; s += 64) ((int32_t*)S.mat)[s] = ((const int32_t*)c_dct_mat)[s];
  if (lane < 4) ((int32_t*)S.dstm)[lane] = ((const int32_t*)c_dst_mat)[lane];
  __syncthreads();
  PX* dst = plane + xy[2 * blockIdx.x] + xy[2 * blockIdx.x + 1] * stride;
  tu_residual_add<PX>(S, lane, dst, stride, log2_size, bit_depth, kind, false, nT - 1, nT - 1);
}

template __global__ void k_tu<uint8_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int,
                                       const int16_t*, const uint16_t*, const uint8_t*, int16_t*);
template __global__ void k_tu<uint16_t>(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int,
                                        const int16_t*, const uint16_t*, const uint8_t*, int16_t*);
template __global__ void k_run<uint8_t, 64>(PicDev, PlaneRef, PlaneRef, PlaneRef, const RunTask*, const uint32_t*, uint32_t*, uint32_t*, const TuTask*, const int16_t*, const uint32_t*, int, int, uint32_t, uint32_t, int, uint32_t, const uint32_t*, const uint32_t*, unsigned long long*);
template __global__ void k_run<uint16_t, 64>(PicDev, PlaneRef, PlaneRef, PlaneRef, const RunTask*, const uint32_t*, uint32_t*, uint32_t*, const TuTask*, const int16_t*, const uint32_t*, int, int, uint32_t, uint32_t, int, uint32_t, const uint32_t*, const uint32_t*, unsigned long long*);
template __global__ void k_fn_residual<uint8_t>(int, int, int, uint8_t*, int, const int32_t*, const int16_t*);
template __global__ void k_fn_residual<uint16_t>(int, int, int, uint16_t*, int, const int32_t*, const int16_t*);

}  // namespace d265
