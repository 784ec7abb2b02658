// vtable.hip -- Interface 1 of the boundary (include/de265_hip_vtable.h): init_acceleration_functions_hip() fills
// libde265's struct acceleration_functions (acceleration.h:29-201) with slots that run ONE block on the GPU,
// synchronously.  Parity/maintenance boundary, not the product path (that is the frame-level interface in host.hip).
//
// Sample-prediction, interpolation and transform+add slots gather the block (and exactly the filter margins the
// fraction needs: fallback-motion.cc:478-479, :297-300) into a compact host plane and go through the batched
// de265hip_fn_* entry points with n = 1.  The int32-residual family (transform_bypass*, transform_skip_residual,
// rdpcm_*, transform_id{c,s}t_*, add_residual_*, rotate_coefficients: acceleration.h:143-178) has its own small kernel
// below: one workgroup per block, plain loops -- these slots exist for completeness of the vtable (cross-component
// prediction, RDPCM and rotation are RExt; Main/Main10 reach them only through transform_bypass / transform_skip_residual
// + add_residual, transform.cc:399-439, :531-579).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels.h"
#include "../../include/de265_hip_vtable.h"

namespace {

__device__ __constant__ int8_t v_dct_mat[32 * 32] = {
#include "dct_table.inc"
};
__device__ __constant__ int8_t v_dst_mat[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };

enum { OP_BYPASS, OP_BYPASS_RDPCM_V, OP_BYPASS_RDPCM_H, OP_TSKIP, OP_RDPCM_V, OP_RDPCM_H, OP_IDCT, OP_IDST };

__device__ __forceinline__ int vclip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// int16 coefficients -> int32 residual of one nT x nT block.  p0/p1: (tsShift, bdShift) for the skip family,
// (bdShift, max_coeff_bits) for the transforms.
__global__ __launch_bounds__(256)
void k_slot_residual(int op, int nT, int p0, int p1, const int16_t* __restrict__ coeffs, int32_t* __restrict__ out)
{
  __shared__ int16_t s_c[32 * 32];
  __shared__ int16_t s_g[32 * 32];
  const int t = threadIdx.x, n = nT * nT;
  for (int i = t; i < n; i += 256) s_c[i] = coeffs[i];
  __syncthreads();
  if (op == OP_BYPASS) {                                    // fallback-dct.cc:216-224
    for (int i = t; i < n; i += 256) out[i] = s_c[i];
  } else if (op == OP_TSKIP) {                              // :80-90
    const int rnd = 1 << (p1 - 1);
    for (int i = t; i < n; i += 256) out[i] = (((int)s_c[i] << p0) + rnd) >> p1;
  } else if (op == OP_BYPASS_RDPCM_V || op == OP_RDPCM_V) { // :160-183, :186-199: running sum down each column
    const int rnd = op == OP_RDPCM_V ? 1 << (p1 - 1) : 0;
    if (t < nT) {
      int sum = 0;
      for (int y = 0; y < nT; y++) {
        const int c = s_c[t + y * nT];
        sum += op == OP_RDPCM_V ? ((c << p0) + rnd) >> p1 : c;
        out[y * nT + t] = sum;
      }
    }
  } else if (op == OP_BYPASS_RDPCM_H || op == OP_RDPCM_H) { // along each row
    const int rnd = op == OP_RDPCM_H ? 1 << (p1 - 1) : 0;
    if (t < nT) {
      int sum = 0;
      for (int x = 0; x < nT; x++) {
        const int c = s_c[x + t * nT];
        sum += op == OP_RDPCM_H ? ((c << p0) + rnd) >> p1 : c;
        out[t * nT + x] = sum;
      }
    }
  } else {
    // transform_idct_fallback (:696-837) / transform_idst_4x4_fallback (:470-509): first stage clipped to
    // max_coeff_bits and kept as int16, second stage rounded by bdShift, NOT clipped (unlike the DST of the add form)
    const int cmax = (1 << p1) - 1, cmin = -(1 << p1);
    const int fact = 32 / nT;
    for (int i = t; i < n; i += 256) {
      const int c = i % nT, r = i / nT;                     // g[c + r*nT] = sum_j M[j][r] * coeff[c + j*nT]
      int sum = 0;
      for (int j = 0; j < nT; j++)
        sum += (op == OP_IDST ? v_dst_mat[j * 4 + r] : v_dct_mat[fact * j * 32 + r]) * s_c[c + j * nT];
      s_g[c + r * nT] = (int16_t)vclip3(cmin, cmax, (sum + 64) >> 7);
    }
    __syncthreads();
    const int rnd2 = 1 << (p0 - 1);
    for (int i = t; i < n; i += 256) {
      const int x = i % nT, y = i / nT;                     // dst[y*nT + x] = sum_j M[j][x] * g[y*nT + j]
      int sum = 0;
      for (int j = 0; j < nT; j++)
        sum += (op == OP_IDST ? v_dst_mat[j * 4 + x] : v_dct_mat[fact * j * 32 + x]) * s_g[y * nT + j];
      out[i] = (sum + rnd2) >> p0;
    }
  }
}

// add_residual_fallback (fallback-dct.h:66-74) on a compact nT x nT block
template <typename PX>
__global__ __launch_bounds__(256)
void k_slot_add_residual(PX* __restrict__ blk, const int32_t* __restrict__ r, int n, int bit_depth)
{
  const int maxv = (1 << bit_depth) - 1;
  for (int i = threadIdx.x; i < n; i += 256) blk[i] = (PX)vclip3(0, maxv, (int)blk[i] + r[i]);
}

// rotate_coefficients_fallback (fallback-dct.cc:251-257): 180 degree rotation in place
__global__ __launch_bounds__(256)
void k_slot_rotate(int16_t* c, int n)
{
  for (int i = threadIdx.x; i < n / 2; i += 256) { const int16_t a = c[i], b = c[n - 1 - i]; c[i] = b; c[n - 1 - i] = a; }
}

[[noreturn]] void die(const char* what, hipError_t e)
{
  fprintf(stderr, "de265hip vtable slot: %s failed (%s); the vtable has no error channel and there is no CPU fallback\n",
          what, e == hipSuccess ? "de265_error" : hipGetErrorString(e));
  abort();
}
#define VCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) die(#expr, e_); } while (0)
#define FCHK(expr) do { if ((expr) != 0) die(#expr, hipSuccess); } while (0)

// per-thread device scratch (slots are called from up to 32 libde265 worker threads, threads.h:142)
struct Scratch {
  void* p = nullptr; size_t bytes = 0;
  void* get(size_t n) {
    if (n > bytes) { if (p) (void)hipFree(p); VCHK(hipMalloc(&p, n)); bytes = n; }
    return p;
  }
  ~Scratch() { if (p) (void)hipFree(p); }
};
thread_local Scratch t_in, t_out;

void run_residual(int op, int nT, int p0, int p1, const int16_t* coeffs, int32_t* out)
{
  const size_t n = (size_t)nT * nT;
  int16_t* dc = (int16_t*)t_in.get(n * 2);
  int32_t* dout = (int32_t*)t_out.get(n * 4);
  VCHK(hipMemcpy(dc, coeffs, n * 2, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_slot_residual, dim3(1), dim3(256), 0, 0, op, nT, p0, p1, dc, dout);
  VCHK(hipGetLastError());
  VCHK(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));           // (synchronises)
}

// ---- int32-residual family ----
void s_transform_bypass(int32_t* r, const int16_t* c, int nT) { run_residual(OP_BYPASS, nT, 0, 0, c, r); }
void s_transform_bypass_rdpcm_v(int32_t* r, const int16_t* c, int nT) { run_residual(OP_BYPASS_RDPCM_V, nT, 0, 0, c, r); }
void s_transform_bypass_rdpcm_h(int32_t* r, const int16_t* c, int nT) { run_residual(OP_BYPASS_RDPCM_H, nT, 0, 0, c, r); }
void s_transform_skip_residual(int32_t* r, const int16_t* c, int nT, int tsShift, int bdShift) { run_residual(OP_TSKIP, nT, tsShift, bdShift, c, r); }
void s_rdpcm_v(int32_t* r, const int16_t* c, int nT, int tsShift, int bdShift) { run_residual(OP_RDPCM_V, nT, tsShift, bdShift, c, r); }
void s_rdpcm_h(int32_t* r, const int16_t* c, int nT, int tsShift, int bdShift) { run_residual(OP_RDPCM_H, nT, tsShift, bdShift, c, r); }
void s_idst_4x4(int32_t* d, const int16_t* c, int bdShift, int mcb) { run_residual(OP_IDST, 4, bdShift, mcb, c, d); }
void s_idct_4x4(int32_t* d, const int16_t* c, int bdShift, int mcb) { run_residual(OP_IDCT, 4, bdShift, mcb, c, d); }
void s_idct_8x8(int32_t* d, const int16_t* c, int bdShift, int mcb) { run_residual(OP_IDCT, 8, bdShift, mcb, c, d); }
void s_idct_16x16(int32_t* d, const int16_t* c, int bdShift, int mcb) { run_residual(OP_IDCT, 16, bdShift, mcb, c, d); }
void s_idct_32x32(int32_t* d, const int16_t* c, int bdShift, int mcb) { run_residual(OP_IDCT, 32, bdShift, mcb, c, d); }

void s_rotate_coefficients(int16_t* c, int nT)
{
  const size_t n = (size_t)nT * nT;
  int16_t* dc = (int16_t*)t_in.get(n * 2);
  VCHK(hipMemcpy(dc, c, n * 2, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_slot_rotate, dim3(1), dim3(256), 0, 0, dc, (int)n);
  VCHK(hipGetLastError());
  VCHK(hipMemcpy(c, dc, n * 2, hipMemcpyDeviceToHost));
}

template <typename PX>
void s_add_residual(PX* dst, ptrdiff_t stride, const int32_t* r, int nT, int bit_depth)
{
  const size_t n = (size_t)nT * nT;
  std::vector<PX> blk(n);
  for (int y = 0; y < nT; y++) memcpy(&blk[(size_t)y * nT], dst + y * stride, nT * sizeof(PX));
  PX* db = (PX*)t_in.get(n * sizeof(PX));
  int32_t* dr = (int32_t*)t_out.get(n * 4);
  VCHK(hipMemcpy(db, blk.data(), n * sizeof(PX), hipMemcpyHostToDevice));
  VCHK(hipMemcpy(dr, r, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_slot_add_residual<PX>, dim3(1), dim3(256), 0, 0, db, dr, (int)n, bit_depth);
  VCHK(hipGetLastError());
  VCHK(hipMemcpy(blk.data(), db, n * sizeof(PX), hipMemcpyDeviceToHost));
  for (int y = 0; y < nT; y++) memcpy(dst + y * stride, &blk[(size_t)y * nT], nT * sizeof(PX));
}
void s_add_residual_8(uint8_t* d, ptrdiff_t s, const int32_t* r, int nT, int bd) { s_add_residual<uint8_t>(d, s, r, nT, bd); }
void s_add_residual_16(uint16_t* d, ptrdiff_t s, const int32_t* r, int nT, int bd) { s_add_residual<uint16_t>(d, s, r, nT, bd); }

// the 8-bit RDPCM transform-skip slots (fallback-dct.cc:93-130) = rdpcm + add_residual with the fixed 8-bit shifts
void s_transform_skip_rdpcm_v_8(uint8_t* dst, const int16_t* c, int log2nT, ptrdiff_t stride)
{ int32_t r[32 * 32]; run_residual(OP_RDPCM_V, 1 << log2nT, 5 + log2nT, 12, c, r); s_add_residual<uint8_t>(dst, stride, r, 1 << log2nT, 8); }
void s_transform_skip_rdpcm_h_8(uint8_t* dst, const int16_t* c, int log2nT, ptrdiff_t stride)
{ int32_t r[32 * 32]; run_residual(OP_RDPCM_H, 1 << log2nT, 5 + log2nT, 12, c, r); s_add_residual<uint8_t>(dst, stride, r, 1 << log2nT, 8); }

// ---- transform + add: the compact block through de265hip_fn_transform_add ----
template <typename PX>
void transform_add_block(PX* dst, const int16_t* coeffs, ptrdiff_t stride, int log2, int is_dst, int bit_depth)
{
  const int nT = 1 << log2;
  std::vector<PX> blk((size_t)nT * nT);
  for (int y = 0; y < nT; y++) memcpy(&blk[(size_t)y * nT], dst + y * stride, nT * sizeof(PX));
  const int32_t xy[2] = { 0, 0 };
  FCHK(de265hip_fn_transform_add(log2, is_dst, bit_depth, blk.data(), nT, nT, 1, xy, coeffs));
  for (int y = 0; y < nT; y++) memcpy(dst + y * stride, &blk[(size_t)y * nT], nT * sizeof(PX));
}
template <int L2> void s_transform_add_8(uint8_t* d, const int16_t* c, ptrdiff_t s) { transform_add_block<uint8_t>(d, c, s, L2, 0, 8); }
template <int L2> void s_transform_add_16(uint16_t* d, const int16_t* c, ptrdiff_t s, int bd) { transform_add_block<uint16_t>(d, c, s, L2, 0, bd); }
void s_transform_dst_add_8(uint8_t* d, const int16_t* c, ptrdiff_t s) { transform_add_block<uint8_t>(d, c, s, 2, 1, 8); }
void s_transform_dst_add_16(uint16_t* d, const int16_t* c, ptrdiff_t s, int bd) { transform_add_block<uint16_t>(d, c, s, 2, 1, bd); }

// ---- sample prediction writes ----
template <typename PX>
void put_block(int mode, PX* dst, ptrdiff_t ds, const int16_t* s0, const int16_t* s1, ptrdiff_t ss, int w, int h,
               int w0, int o0, int w1, int o1, int log2wd, int bit_depth)
{
  std::vector<PX> blk((size_t)w * h);
  std::vector<int16_t> a((size_t)w * h), b(s1 ? (size_t)w * h : 0);
  for (int y = 0; y < h; y++) {
    memcpy(&blk[(size_t)y * w], dst + y * ds, w * sizeof(PX));
    memcpy(&a[(size_t)y * w], s0 + y * ss, w * 2);
    if (s1) memcpy(&b[(size_t)y * w], s1 + y * ss, w * 2);
  }
  const int32_t xy[2] = { 0, 0 };
  FCHK(de265hip_fn_put_pred(mode, bit_depth, blk.data(), w, h, w, h, 1, xy, a.data(), s1 ? b.data() : nullptr, w0, o0, w1, o1, log2wd));
  for (int y = 0; y < h; y++) memcpy(dst + y * ds, &blk[(size_t)y * w], w * sizeof(PX));
}
void s_put_avg_8(uint8_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h) { put_block<uint8_t>(2, d, ds, a, b, ss, w, h, 0, 0, 0, 0, 1, 8); }
void s_put_uni_8(uint8_t* d, ptrdiff_t ds, const int16_t* a, ptrdiff_t ss, int w, int h) { put_block<uint8_t>(0, d, ds, a, nullptr, ss, w, h, 0, 0, 0, 0, 1, 8); }
void s_put_w_8(uint8_t* d, ptrdiff_t ds, const int16_t* a, ptrdiff_t ss, int w, int h, int w0, int o0, int wd) { put_block<uint8_t>(1, d, ds, a, nullptr, ss, w, h, w0, o0, 0, 0, wd, 8); }
void s_put_bi_8(uint8_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h, int w0, int o0, int w1, int o1, int wd)
{ put_block<uint8_t>(3, d, ds, a, b, ss, w, h, w0, o0, w1, o1, wd, 8); }
void s_put_avg_16(uint16_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h, int bd) { put_block<uint16_t>(2, d, ds, a, b, ss, w, h, 0, 0, 0, 0, 1, bd); }
void s_put_uni_16(uint16_t* d, ptrdiff_t ds, const int16_t* a, ptrdiff_t ss, int w, int h, int bd) { put_block<uint16_t>(0, d, ds, a, nullptr, ss, w, h, 0, 0, 0, 0, 1, bd); }
void s_put_w_16(uint16_t* d, ptrdiff_t ds, const int16_t* a, ptrdiff_t ss, int w, int h, int w0, int o0, int wd, int bd) { put_block<uint16_t>(1, d, ds, a, nullptr, ss, w, h, w0, o0, 0, 0, wd, bd); }
void s_put_bi_16(uint16_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h, int w0, int o0, int w1, int o1, int wd, int bd)
{ put_block<uint16_t>(3, d, ds, a, b, ss, w, h, w0, o0, w1, o1, wd, bd); }

// ---- interpolation: exactly the margins the fraction reads are taken from the caller's plane ----
template <typename PX>
void interp_block(int luma, int16_t* dst, ptrdiff_t dststride, const PX* src, ptrdiff_t srcstride, int w, int h, int fx, int fy, int bit_depth)
{
  static const int qb[4] = { 0, 3, 3, 2 }, qa[4] = { 0, 3, 4, 4 };      // fallback-motion.cc:478-479
  const int bx = luma ? qb[fx] : (fx ? 1 : 0), ax = luma ? qa[fx] : (fx ? 2 : 0);   // epel: 1 before, 2 after (:297-300)
  const int by = luma ? qb[fy] : (fy ? 1 : 0), ay = luma ? qa[fy] : (fy ? 2 : 0);
  const int mb = luma ? 3 : 1, ma = luma ? 4 : 2;                      // the batched entry point's uniform margins
  const int pw = w + mb + ma, ph = h + mb + ma;
  std::vector<PX> plane((size_t)pw * ph, 0);                           // cells outside the needed margins meet zero taps
  for (int y = -by; y < h + ay; y++)
    memcpy(&plane[(size_t)(y + mb) * pw + (mb - bx)], src + y * srcstride - bx, (size_t)(w + bx + ax) * sizeof(PX));
  std::vector<int16_t> out((size_t)w * h);
  const int32_t xy[2] = { mb, mb };
  if (luma) FCHK(de265hip_fn_put_qpel(bit_depth, plane.data(), pw, pw, ph, w, h, fx, fy, 1, xy, out.data()));
  else FCHK(de265hip_fn_put_epel(bit_depth, plane.data(), pw, pw, ph, w, h, fx, fy, 1, xy, out.data()));
  for (int y = 0; y < h; y++) memcpy(dst + y * dststride, &out[(size_t)y * w], (size_t)w * 2);
}
template <int DX, int DY> void s_qpel_8(int16_t* d, ptrdiff_t ds, const uint8_t* s, ptrdiff_t ss, int w, int h, int16_t*) { interp_block<uint8_t>(1, d, ds, s, ss, w, h, DX, DY, 8); }
template <int DX, int DY> void s_qpel_16(int16_t* d, ptrdiff_t ds, const uint16_t* s, ptrdiff_t ss, int w, int h, int16_t*, int bd) { interp_block<uint16_t>(1, d, ds, s, ss, w, h, DX, DY, bd); }
void s_epel_8(int16_t* d, ptrdiff_t ds, const uint8_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*) { interp_block<uint8_t>(0, d, ds, s, ss, w, h, mx, my, 8); }
void s_epel_8b(int16_t* d, ptrdiff_t ds, const uint8_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*, int) { interp_block<uint8_t>(0, d, ds, s, ss, w, h, mx, my, 8); }
void s_epel_16(int16_t* d, ptrdiff_t ds, const uint16_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*, int bd) { interp_block<uint16_t>(0, d, ds, s, ss, w, h, mx, my, bd); }

template <int DX> void fill_qpel_row(de265hip_acceleration_functions* a)
{
  a->put_hevc_qpel_8[DX][0] = s_qpel_8<DX, 0>; a->put_hevc_qpel_8[DX][1] = s_qpel_8<DX, 1>;
  a->put_hevc_qpel_8[DX][2] = s_qpel_8<DX, 2>; a->put_hevc_qpel_8[DX][3] = s_qpel_8<DX, 3>;
  a->put_hevc_qpel_16[DX][0] = s_qpel_16<DX, 0>; a->put_hevc_qpel_16[DX][1] = s_qpel_16<DX, 1>;
  a->put_hevc_qpel_16[DX][2] = s_qpel_16<DX, 2>; a->put_hevc_qpel_16[DX][3] = s_qpel_16<DX, 3>;
}

}  // namespace

// Overrides every slot the decoder calls (fallback.cc:26-127 lists them); leaves the encoder slots and the deprecated
// transform_skip_{8,16} alone -- call it after init_acceleration_functions_fallback, as decctx.cc:430-449 does for SSE/ARM.
extern "C" void init_acceleration_functions_hip(struct acceleration_functions* accel)
{
  de265hip_acceleration_functions* a = reinterpret_cast<de265hip_acceleration_functions*>(accel);   // same layout (tests/native/vtable_layout_check.cc)
  a->put_weighted_pred_avg_8 = s_put_avg_8; a->put_unweighted_pred_8 = s_put_uni_8;
  a->put_weighted_pred_8 = s_put_w_8; a->put_weighted_bipred_8 = s_put_bi_8;
  a->put_weighted_pred_avg_16 = s_put_avg_16; a->put_unweighted_pred_16 = s_put_uni_16;
  a->put_weighted_pred_16 = s_put_w_16; a->put_weighted_bipred_16 = s_put_bi_16;
  a->put_hevc_epel_8 = s_epel_8; a->put_hevc_epel_h_8 = s_epel_8b; a->put_hevc_epel_v_8 = s_epel_8b; a->put_hevc_epel_hv_8 = s_epel_8b;
  a->put_hevc_epel_16 = s_epel_16; a->put_hevc_epel_h_16 = s_epel_16; a->put_hevc_epel_v_16 = s_epel_16; a->put_hevc_epel_hv_16 = s_epel_16;
  fill_qpel_row<0>(a); fill_qpel_row<1>(a); fill_qpel_row<2>(a); fill_qpel_row<3>(a);
  a->transform_bypass = s_transform_bypass; a->transform_bypass_rdpcm_v = s_transform_bypass_rdpcm_v;
  a->transform_bypass_rdpcm_h = s_transform_bypass_rdpcm_h;
  a->transform_skip_rdpcm_v_8 = s_transform_skip_rdpcm_v_8; a->transform_skip_rdpcm_h_8 = s_transform_skip_rdpcm_h_8;
  a->transform_4x4_dst_add_8 = s_transform_dst_add_8; a->transform_4x4_dst_add_16 = s_transform_dst_add_16;
  a->transform_add_8[0] = s_transform_add_8<2>; a->transform_add_8[1] = s_transform_add_8<3>;
  a->transform_add_8[2] = s_transform_add_8<4>; a->transform_add_8[3] = s_transform_add_8<5>;
  a->transform_add_16[0] = s_transform_add_16<2>; a->transform_add_16[1] = s_transform_add_16<3>;
  a->transform_add_16[2] = s_transform_add_16<4>; a->transform_add_16[3] = s_transform_add_16<5>;
  a->rotate_coefficients = s_rotate_coefficients;
  a->transform_idst_4x4 = s_idst_4x4; a->transform_idct_4x4 = s_idct_4x4; a->transform_idct_8x8 = s_idct_8x8;
  a->transform_idct_16x16 = s_idct_16x16; a->transform_idct_32x32 = s_idct_32x32;
  a->add_residual_8 = s_add_residual_8; a->add_residual_16 = s_add_residual_16;
  a->rdpcm_v = s_rdpcm_v; a->rdpcm_h = s_rdpcm_h; a->transform_skip_residual = s_transform_skip_residual;
}
