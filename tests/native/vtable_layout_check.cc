// Build-container check (tests/test_vtable.py): the layout mirror of include/de265_hip_vtable.h against the
// reference's own struct acceleration_functions (libde265/acceleration.h:29-201), and a type check that every
// decoder slot accepts the pointer the mirror holds (assignment in both directions compiles only for identical types).
#include "acceleration.h"
#include "de265_hip_vtable.h"
#include <cstddef>
#include <cstdio>

#define SAME(slot) static_assert(offsetof(acceleration_functions, slot) == offsetof(de265hip_acceleration_functions, slot), #slot)
static_assert(sizeof(acceleration_functions) == sizeof(de265hip_acceleration_functions), "size");
SAME(put_weighted_pred_avg_8); SAME(put_unweighted_pred_8); SAME(put_weighted_pred_8); SAME(put_weighted_bipred_8);
SAME(put_weighted_pred_avg_16); SAME(put_unweighted_pred_16); SAME(put_weighted_pred_16); SAME(put_weighted_bipred_16);
SAME(put_hevc_epel_8); SAME(put_hevc_epel_h_8); SAME(put_hevc_epel_v_8); SAME(put_hevc_epel_hv_8); SAME(put_hevc_qpel_8);
SAME(put_hevc_epel_16); SAME(put_hevc_epel_h_16); SAME(put_hevc_epel_v_16); SAME(put_hevc_epel_hv_16); SAME(put_hevc_qpel_16);
SAME(transform_bypass); SAME(transform_bypass_rdpcm_v); SAME(transform_bypass_rdpcm_h);
SAME(transform_skip_8); SAME(transform_skip_rdpcm_v_8); SAME(transform_skip_rdpcm_h_8);
SAME(transform_4x4_dst_add_8); SAME(transform_add_8); SAME(transform_skip_16); SAME(transform_4x4_dst_add_16); SAME(transform_add_16);
SAME(rotate_coefficients); SAME(transform_idst_4x4); SAME(transform_idct_4x4); SAME(transform_idct_8x8); SAME(transform_idct_16x16);
SAME(transform_idct_32x32); SAME(add_residual_8); SAME(add_residual_16); SAME(rdpcm_v); SAME(rdpcm_h); SAME(transform_skip_residual);
SAME(fwd_transform_4x4_dst_8); SAME(fwd_transform_8); SAME(hadamard_transform_8);

#define TYPES(slot) do { r.slot = m.slot; m.slot = r.slot; } while (0)
int main()
{
  acceleration_functions r = {};
  de265hip_acceleration_functions m = {};
  TYPES(put_weighted_pred_avg_8); TYPES(put_unweighted_pred_8); TYPES(put_weighted_pred_8); TYPES(put_weighted_bipred_8);
  TYPES(put_weighted_pred_avg_16); TYPES(put_unweighted_pred_16); TYPES(put_weighted_pred_16); TYPES(put_weighted_bipred_16);
  TYPES(put_hevc_epel_8); TYPES(put_hevc_epel_h_8); TYPES(put_hevc_epel_v_8); TYPES(put_hevc_epel_hv_8);
  TYPES(put_hevc_epel_16); TYPES(put_hevc_epel_h_16); TYPES(put_hevc_epel_v_16); TYPES(put_hevc_epel_hv_16);
  for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) { TYPES(put_hevc_qpel_8[a][b]); TYPES(put_hevc_qpel_16[a][b]); }
  TYPES(transform_bypass); TYPES(transform_bypass_rdpcm_v); TYPES(transform_bypass_rdpcm_h);
  TYPES(transform_skip_8); TYPES(transform_skip_rdpcm_v_8); TYPES(transform_skip_rdpcm_h_8);
  TYPES(transform_4x4_dst_add_8); TYPES(transform_skip_16); TYPES(transform_4x4_dst_add_16);
  for (int a = 0; a < 4; a++) { TYPES(transform_add_8[a]); TYPES(transform_add_16[a]); TYPES(fwd_transform_8[a]); TYPES(hadamard_transform_8[a]); }
  TYPES(rotate_coefficients); TYPES(transform_idst_4x4); TYPES(transform_idct_4x4); TYPES(transform_idct_8x8);
  TYPES(transform_idct_16x16); TYPES(transform_idct_32x32); TYPES(add_residual_8); TYPES(add_residual_16);
  TYPES(rdpcm_v); TYPES(rdpcm_h); TYPES(transform_skip_residual); TYPES(fwd_transform_4x4_dst_8);
  // the entry point has the shape of init_acceleration_functions_fallback (fallback.cc:26)
  void (*init)(struct acceleration_functions*) = init_acceleration_functions_hip;
  (void)init; (void)r; (void)m;
  puts("layout ok");
  return 0;
}
