// dev_common.h -- structures shared between the host library and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/de265_hip.h"

namespace d265 {

// One device-resident plane.
struct PlaneRef {
  void* ptr;
  int32_t stride;   // samples
  int32_t pad;
};

// Plane table passed by value to kernels: [slot][component].
struct DpbTable {
  PlaneRef p[DE265HIP_MAX_DPB_SLOTS][3];
};

// Picture constants needed on the device.
struct PicDev {
  int32_t width, height;            // luma
  int32_t bd_luma, bd_chroma;
  int32_t log2_ctb, ctbs_w, ctbs_h;
  int32_t w4, h4;                   // ceil(W/4), ceil(H/4)
  int32_t strong_intra, pcm_lf_disable;
  int32_t weighted_pred, weighted_bipred;
  int32_t cb_qp_offset, cr_qp_offset;
  int32_t lf_across_tiles;
  int32_t scaling_list;
  int32_t dbg;                      // DE265HIP_DEBUG ablation bits (timing only)
  int32_t has_exempt;               // some 4x4 unit is pcm (with pcm_loop_filter_disable) or transquant-bypass
  // chroma format and range-extension tools (all "4:2:0, off" for Main / Main10)
  int32_t chroma_format;            // 1, 2, 3
  int32_t csw, csh;                 // log2 SubWidthC / SubHeightC
  int32_t cwidth, cheight;          // chroma plane size
  int32_t smooth_luma, smooth_chroma;   // intra neighbour smoothing applies (intrapred.cc:1085-1089)
  int32_t implicit_rdpcm;           // sps flag: with cu_transquant_bypass it switches the mode 10 / 26 edge filters off (intrapred.cc:1102)
  int32_t xcc_enabled;              // pps cross_component_prediction_enabled_flag
  int32_t wp_shift_luma, wp_shift_chroma;   // WpOffsetBdShift
};

// XCD-aware block index.  Workgroups are dealt round-robin to the 8 XCDs in dispatch order, so blocks L and L+8 share an XCD
// and its L2.  A kernel launched as a 1-D grid of xcd_grid(gx*gy*gz) blocks calls xcd_block(): every XCD then works on ONE
// contiguous eighth of the logical (x fastest, z slowest) grid, i.e. a band of the picture, and the rows / columns that
// neighbouring tiles share are fetched into one L2 once instead of into several.  Speed only, never correctness.
struct XcdBlk { int x, y, z; bool ok; };
static inline unsigned xcd_grid(unsigned n) { return (n + 7u) & ~7u; }
__device__ __forceinline__ XcdBlk xcd_block(unsigned gx, unsigned gy, unsigned gz)
{
  const unsigned n = gx * gy * gz, per = (n + 7u) >> 3;
  const unsigned lg = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  XcdBlk b;
  b.ok = lg < n && (blockIdx.x >> 3) < per;
  const unsigned plane = gx * gy, z = lg / plane, r = lg - z * plane, y = r / gx;
  b.x = (int)(r - y * gx); b.y = (int)y; b.z = (int)z;
  return b;
}

__device__ __forceinline__ bool intra_smooth_on(const PicDev& P, int c_idx) { return c_idx ? P.smooth_chroma != 0 : P.smooth_luma != 0; }

// TU task: de265hip_tu plus the host-derived neighbour availability.
// avail bit u (scan order of intrapred.cc:577-688): u < 2nT/4 left column
// bottom->top in 4-sample units, u == 2nT/4 corner, then top row left->right.
struct TuTask {
  uint16_t x0, y0;
  uint8_t  log2_size, c_idx, flags, intra_mode;
  int8_t   qp;
  uint8_t  run_level;      // run mode (run-ordered array): workgroup barriers the TU's wavefront passes before it (k_run)
  uint16_t n_coeff;
  uint32_t coeff_offset;   // run mode (run-ordered array only): sample offset of the TU inside its run
  uint64_t avail;
  uint32_t resid_offset;   // run mode: where the precomputed int16 residual block of an intra TU lives
  int8_t   angle;          // run mode: intraPredAngle of the mode (intrapred.cc:742-760), host-resolved so that
  uint8_t  pad3;           //           no table load sits on the z-scan chain
  int16_t  inv_angle;      //           invAngle for the negative angles, else 0
};
static_assert(sizeof(TuTask) == 32, "TuTask layout");
// internal TuTask flag: compute the residual only (into the residual buffer), no prediction, no picture access
#define D265_TU_RESID_ONLY 0x80

// Range-extension residual tasks (level-0 tasks only: TuTask::pad3 and friends are free there).  A TU takes the generic
// k_resid_rext path when any of these applies; everything else stays on the tuned residual kernels.
#define D265_RX_RDPCM_H   0x01      // residual DPCM, horizontal / vertical (fallback-dct.cc:160-213)
#define D265_RX_RDPCM_V   0x02
#define D265_RX_ROTATE    0x04      // rotate_coefficients (fallback-dct.cc:251-257)
#define D265_RX_XCC       0x08      // cross-component prediction: TuTask::angle = ResScaleVal, TuTask::avail = the luma TU's
                                    // coeff_offset | n_coeff << 32 | (uint8)qp << 48 | flags << 56 (transform.cc:235-251)
#define D265_RX_LUMA_ROT  0x10      // ... and the luma TU's coefficients are rotated
#define D265_RX_LUMA_RDPCM_SHIFT 5  // ... bits 5-6: the luma TU's RDPCM mode

// MC task: a <=16x16 luma tile of one PU (plus its two chroma tiles).
struct __attribute__((aligned(4))) McTask {   // (4-byte aligned: the kernel fetches its task with scalar loads)
  uint16_t x, y;          // luma position of the tile
  uint8_t  w, h;          // luma size (multiples of 4, <=16)
  int8_t   slot[2];       // DPB slot per list, -1 = list unused
  int16_t  mv[2][2];
  uint16_t slice_idx;
  int8_t   ref_idx[2];
};
static_assert(sizeof(McTask) == 20, "McTask layout");
// k_mc_all: the MC tasks of band x (a range of CTB rows; XCD x) are tasks[first[x] ..): n_tiles[x] 16x16 tiles, then n_chunks[x]
// chunks of up to 32x32 (the host: 32x16), then 4 * n_quads[x] blocks of up to 8x8 (every four of one slot pair)
// and the order its wavefronts take them in: n_entries[x] dwords from order_first[x] on: bit 31 a quad, bit 30 a chunk, else a tile | index in the band's list of that form
struct McBands { uint32_t first[8], n_tiles[8], n_chunks[8], n_quads[8], order_first[8], n_entries[8]; };

// Run task: a run of consecutive (decode order) intra TUs of ONE colour component that
// one wavefront reconstructs serially with its pixel window resident in LDS.
// Runs are ordered so that every producer run has a smaller index (ticket).
struct RunTask {
  uint16_t x0, y0, x1, y1;   // bounding box of the run's TUs, component samples, x1/y1 exclusive
  uint16_t wx1, wy1;         // end of the pixel window: furthest neighbour any TU of the run may read
  uint8_t  c_idx, micro;     // bit 1: dense (the run's TUs cover its whole bounding box); bit 0 micro: <= 16 TUs of <= 8x8 in a <= 32x32 box: reconstructed by ONE wavefront (k_run)
  uint16_t n_tus;
  uint32_t first_tu;         // into the run-ordered TuTask array
  uint32_t dep_offset;       // into the producer-run id array
  uint16_t n_deps, n_lvls;  // n_lvls: workgroup barriers of the run's chain (barrier epochs, see host.hip)
  uint32_t res_offset;       // the run's residual blocks (TUs with coefficients only): one contiguous int16 range
  uint32_t n_samples;        // samples of all TUs of the run
  uint16_t wave_end[4];      // the run's TUs are stored as one list per wavefront of the workgroup (each list in
                             // level order): list w is [wave_end[w-1], wave_end[w]) relative to first_tu
};
static_assert(sizeof(RunTask) == 44, "RunTask layout");

// Per-CTB SAO record, fully resolved on the host (slice flags applied, slice/tile permissions of the
// 3x3 CTB neighbourhood evaluated): one 24-byte load per lane instead of a chain of dependent loads.
struct __attribute__((aligned(8))) SaoCtb {
  uint8_t type[3];          // 0 off, 1 band, 2 edge (0 when the slice's sao flag for that component is off)
  uint8_t eo[3];
  uint8_t band[3];
  int8_t  off[3][4];
  uint8_t perm_c_hi;        // bits 7..8 of the chroma permissions
  uint16_t perm;            // bits 0..8: luma, bit (dy+1)*3+(dx+1): samples of this CTB may use neighbours in CTB
                            // (x+dx, y+dy); bits 9..15: bits 0..6 of the chroma permissions (same numbering).  Chroma has
                            // its own set, and its bit 4 (the CTB itself) can be 0: sao.cc:55 compares against the slice
                            // of CTB (x/2, y/2) for chroma (component coordinates passed as luma coordinates)
};
static_assert(sizeof(SaoCtb) == 24, "SaoCtb layout");

struct PcmTask {
  uint16_t x0, y0;
  uint32_t log2_cb_size;
  uint32_t sample_offset;
};

}  // namespace d265
