// kernels.h -- declarations of the gfx950 kernels (defined in k_*.hip).
#pragma once
#include "dev_common.h"

namespace d265 {

struct LfMeta {
  const uint8_t* flags;
  const int8_t* qp;
  const uint8_t* bs;                    // from k_bs, or null: the deblocking kernels derive bS themselves (default)
  const de265hip_motion* motion;
  const de265hip_ctb_info* ctbs;
  const de265hip_slice_params* slices;
};
#ifndef SAO_ROWS
#define SAO_ROWS 2                      // rows per lane of k_sao (never crosses a CTB).  Per 4K picture, packed kernel: 8 -> 32.3 us, 4 -> 27.6 us,
                                        // 2 -> 26.6 us (twice the wavefronts and 2x instead of 1.5x the rows read, but their load / arithmetic /
                                        // store phases overlap better; 3-stream bench +1.1 %)
#endif
#ifndef SAO_GROUPS
#define SAO_GROUPS 1                    // groups of 62 strips per wavefront of k_sao, loaded together and finished one after the other
                                        // (2: the second group's rows arrive under the first's arithmetic, but 105 VGPRs + scratch: 34 vs 27.6 us)
#endif
struct SaoMeta {
  const uint8_t* flags;
  const SaoCtb* sao;
};

template <typename PX>
__global__ void k_tu(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int,
                     const int16_t*, const uint16_t*, const uint8_t*, int16_t*);
template <typename PX>
__global__ void k_resid_big(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int, int, int, int, const int16_t*, const uint16_t*,
                            const uint8_t*, int16_t*);
template <typename PX>
__global__ void k_resid_small(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int, int, int, const int16_t*,
                              const uint16_t*, const uint8_t*, int16_t*);
#ifndef RUN_TICKET_SLOTS
#define RUN_TICKET_SLOTS 8   // slots of a k_run ticket: that many micro runs (slot q and q+4 .. by the same wavefront, one after
                             // the other), or one ordinary run in slot 0
#endif
#ifndef RESID_SPL
#define RESID_SPL 4          // samples per lane of k_resid_small: RESID_SPL 8x8 TUs / 4 RESID_SPL 4x4 TUs per wavefront
#endif
#define RUN_WAVES 4          // wavefronts per run workgroup (one per SIMD of a CU); blockDim.x = 64..64*RUN_WAVES
__global__ void k_check_coeffs(const TuTask*, int, const TuTask*, int, uint16_t*, uint32_t*);
template <typename PX, int BOX>
__global__ void k_run(PicDev, PlaneRef, PlaneRef, PlaneRef, const RunTask*, const uint32_t*, uint32_t*, uint32_t*,
                      const TuTask*, const int16_t*, const uint32_t*, int, int, uint32_t, uint32_t, int, uint32_t, const uint32_t*, const uint32_t*, unsigned long long*);
#define RUN_SPIN_LIMIT_DEFAULT (1u << 21)   // polls (~1 us each) a k_run wavefront waits for a producer's flag before the picture fails instead of hanging
template <typename PX>
__global__ void k_intra_front(PicDev, PlaneRef, PlaneRef, PlaneRef, const RunTask*, const TuTask*, const int16_t*, int, const uint32_t*);
template <typename PX>
__global__ void k_mc(PicDev, DpbTable, PlaneRef, PlaneRef, PlaneRef, const McTask*,
                     const de265hip_slice_params*, int);
template <typename PX>
__global__ void k_mc_all(PicDev, DpbTable, PlaneRef, PlaneRef, PlaneRef, const McTask*,
                         const de265hip_slice_params*, const uint32_t*, McBands);
template <typename PX>
__global__ void k_pcm(PicDev, PlaneRef, PlaneRef, PlaneRef, const PcmTask*, const uint16_t*);
__global__ void k_bs(PicDev, const uint8_t*, const de265hip_motion*, uint8_t*);
__global__ void k_motion_from_pus(PicDev, const de265hip_pu*, int, const de265hip_slice_params*, int, de265hip_motion*);
template <typename PX, bool VERT>
__global__ void k_deblock(PicDev, PlaneRef, PlaneRef, PlaneRef, LfMeta);
template <typename PX>
__global__ void k_deblock_fused(PicDev, PlaneRef, PlaneRef, PlaneRef, LfMeta);
template <typename PX>
__global__ void k_sao(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, SaoMeta);
template <typename PX>
__global__ void k_sao_ctb(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, SaoMeta, uint3);
#ifndef LF_TW
#define LF_TW 128                        // tile of k_lf_tile (deblocking + SAO in one pass), luma / chroma samples alike
#endif
#ifndef LF_TH
#define LF_TH 64
#endif
#define LF_TILE_W LF_TW
#define LF_TILE_H LF_TH
template <typename PX>
__global__ void k_lf_tile(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, PlaneRef, LfMeta, SaoMeta, int);

// range-extension paths (k_rext.hip; the chroma planes of 4:2:2 / 4:4:4 MC tasks: k_mc's blockIdx.y == 1, k_mc.hip)
template <typename PX>
__global__ void k_resid_rext(PicDev, PlaneRef, PlaneRef, PlaneRef, const TuTask*, int, const int16_t*, const uint16_t*, const uint8_t*, int16_t*);
template <typename PX>
__global__ void k_deblock_chroma_any(PicDev, PlaneRef, PlaneRef, LfMeta, int);
template <typename PX>
__global__ void k_sao_chroma_any(PicDev, PlaneRef, PlaneRef, PlaneRef, PlaneRef, SaoMeta);

// function-level kernels (k_fn.hip)
template <typename PX>
__global__ void k_fn_residual(int kind, int log2_size, int bit_depth, PX* plane, int stride,
                              const int32_t* xy, const int16_t* coeffs);
template <typename PX>
__global__ void k_fn_interp(int luma, int bit_depth, const PX* plane, int stride, int pw, int ph,
                            int w, int h, int fx, int fy, const int32_t* xy, int16_t* out);
template <typename PX>
__global__ void k_fn_put(int mode, int bit_depth, PX* plane, int stride, int w, int h,
                         const int32_t* xy, const int16_t* s0, const int16_t* s1,
                         int w0, int o0, int w1, int o1, int log2wd);

}  // namespace d265
