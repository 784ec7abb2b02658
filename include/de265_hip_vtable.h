/*
 * de265_hip_vtable.h -- Interface 1 of the drop-in boundary: the MI355X back end behind
 * libde265's own DSP vtable, `struct acceleration_functions` (libde265/acceleration.h:29-201).
 *
 *   void init_acceleration_functions_hip(struct acceleration_functions* accel);
 *
 * is exported by libde265_hip.so with exactly the shape of init_acceleration_functions_fallback
 * (libde265/fallback.cc:26), init_acceleration_functions_sse (x86/sse.cc:37) and _arm (arm/arm.cc:102), and
 * is installed the same way: in base_context::set_acceleration_functions (decctx.cc:430-449), after the
 * fallback has filled every slot (INTEGRATION.md shows the four-line patch).  It overrides every slot the
 * DECODER calls; encoder slots (fwd_transform_*, hadamard_*) and the two deprecated transform_skip_{8,16}
 * slots (fallback-dct.cc:51, assert(0)) keep the fallback's pointers.
 *
 * Each slot runs its block on the GPU synchronously (upload block + margins, one kernel, download):
 * functionally a drop-in, bit-exact (tests/test_vtable.py: every slot against the compiled reference's
 * fallback slot, called through the reference's own struct type), but one PCIe round trip per block.  It
 * is the parity/maintenance boundary; the product path is the frame-level interface of de265_hip.h.
 * A slot that cannot reach the GPU calls abort(): the vtable has no error channel (acceleration.h: all
 * slots return void) and there is no CPU fallback.
 *
 * A libde265 build includes its own acceleration.h and declares only the init function.  For code that has
 * no libde265 headers (this repository's tests), the struct below mirrors the DATA layout of
 * acceleration_functions -- function pointers only, in declaration order; the reference's inline helper
 * methods occupy no storage.  tests/test_vtable.py compiles a translation unit against the reference's header
 * (build container) that static_asserts size and every offset to be identical.
 */
#ifndef DE265_HIP_VTABLE_H
#define DE265_HIP_VTABLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct acceleration_functions;     /* libde265/acceleration.h */

/* the drop-in entry point (fallback.cc:26 pattern) */
void init_acceleration_functions_hip(struct acceleration_functions* accel);

/* ---- layout mirror (see above); slot groups cite acceleration.h ---- */
typedef void (*de265hip_put_avg8_fn)(uint8_t*, ptrdiff_t, const int16_t*, const int16_t*, ptrdiff_t, int, int);
typedef void (*de265hip_put_uni8_fn)(uint8_t*, ptrdiff_t, const int16_t*, ptrdiff_t, int, int);
typedef void (*de265hip_put_w8_fn)(uint8_t*, ptrdiff_t, const int16_t*, ptrdiff_t, int, int, int, int, int);
typedef void (*de265hip_put_bi8_fn)(uint8_t*, ptrdiff_t, const int16_t*, const int16_t*, ptrdiff_t, int, int, int, int, int, int, int);
typedef void (*de265hip_put_avg16_fn)(uint16_t*, ptrdiff_t, const int16_t*, const int16_t*, ptrdiff_t, int, int, int);
typedef void (*de265hip_put_uni16_fn)(uint16_t*, ptrdiff_t, const int16_t*, ptrdiff_t, int, int, int);
typedef void (*de265hip_put_w16_fn)(uint16_t*, ptrdiff_t, const int16_t*, ptrdiff_t, int, int, int, int, int, int);
typedef void (*de265hip_put_bi16_fn)(uint16_t*, ptrdiff_t, const int16_t*, const int16_t*, ptrdiff_t, int, int, int, int, int, int, int, int);
typedef void (*de265hip_epel8_fn)(int16_t*, ptrdiff_t, const uint8_t*, ptrdiff_t, int, int, int, int, int16_t*);
typedef void (*de265hip_epel8b_fn)(int16_t*, ptrdiff_t, const uint8_t*, ptrdiff_t, int, int, int, int, int16_t*, int);
typedef void (*de265hip_qpel8_fn)(int16_t*, ptrdiff_t, const uint8_t*, ptrdiff_t, int, int, int16_t*);
typedef void (*de265hip_epel16_fn)(int16_t*, ptrdiff_t, const uint16_t*, ptrdiff_t, int, int, int, int, int16_t*, int);
typedef void (*de265hip_qpel16_fn)(int16_t*, ptrdiff_t, const uint16_t*, ptrdiff_t, int, int, int16_t*, int);
typedef void (*de265hip_resid32_fn)(int32_t*, const int16_t*, int);
typedef void (*de265hip_idct32_fn)(int32_t*, const int16_t*, int, int);

struct de265hip_acceleration_functions {
  /* acceleration.h:31-64  sample prediction writes */
  de265hip_put_avg8_fn  put_weighted_pred_avg_8;
  de265hip_put_uni8_fn  put_unweighted_pred_8;
  de265hip_put_w8_fn    put_weighted_pred_8;
  de265hip_put_bi8_fn   put_weighted_bipred_8;
  de265hip_put_avg16_fn put_weighted_pred_avg_16;
  de265hip_put_uni16_fn put_unweighted_pred_16;
  de265hip_put_w16_fn   put_weighted_pred_16;
  de265hip_put_bi16_fn  put_weighted_bipred_16;
  /* acceleration.h:87-120  interpolation */
  de265hip_epel8_fn     put_hevc_epel_8;
  de265hip_epel8b_fn    put_hevc_epel_h_8, put_hevc_epel_v_8, put_hevc_epel_hv_8;
  de265hip_qpel8_fn     put_hevc_qpel_8[4][4];
  de265hip_epel16_fn    put_hevc_epel_16, put_hevc_epel_h_16, put_hevc_epel_v_16, put_hevc_epel_hv_16;
  de265hip_qpel16_fn    put_hevc_qpel_16[4][4];
  /* acceleration.h:143-178  inverse transforms and residual helpers */
  de265hip_resid32_fn   transform_bypass, transform_bypass_rdpcm_v, transform_bypass_rdpcm_h;
  void (*transform_skip_8)(uint8_t*, const int16_t*, ptrdiff_t);
  void (*transform_skip_rdpcm_v_8)(uint8_t*, const int16_t*, int, ptrdiff_t);
  void (*transform_skip_rdpcm_h_8)(uint8_t*, const int16_t*, int, ptrdiff_t);
  void (*transform_4x4_dst_add_8)(uint8_t*, const int16_t*, ptrdiff_t);
  void (*transform_add_8[4])(uint8_t*, const int16_t*, ptrdiff_t);
  void (*transform_skip_16)(uint16_t*, const int16_t*, ptrdiff_t, int);
  void (*transform_4x4_dst_add_16)(uint16_t*, const int16_t*, ptrdiff_t, int);
  void (*transform_add_16[4])(uint16_t*, const int16_t*, ptrdiff_t, int);
  void (*rotate_coefficients)(int16_t*, int);
  de265hip_idct32_fn    transform_idst_4x4, transform_idct_4x4, transform_idct_8x8, transform_idct_16x16, transform_idct_32x32;
  void (*add_residual_8)(uint8_t*, ptrdiff_t, const int32_t*, int, int);
  void (*add_residual_16)(uint16_t*, ptrdiff_t, const int32_t*, int, int);
  void (*rdpcm_v)(int32_t*, const int16_t*, int, int, int);
  void (*rdpcm_h)(int32_t*, const int16_t*, int, int, int);
  void (*transform_skip_residual)(int32_t*, const int16_t*, int, int, int);
  /* acceleration.h:192-200  encoder only: left to the fallback */
  void (*fwd_transform_4x4_dst_8)(int16_t*, const int16_t*, ptrdiff_t);
  void (*fwd_transform_8[4])(int16_t*, const int16_t*, ptrdiff_t);
  void (*hadamard_transform_8[4])(int16_t*, const int16_t*, ptrdiff_t);
};

#ifdef __cplusplus
}
#endif
#endif /* DE265_HIP_VTABLE_H */
