/*
 * de265_hip.h -- C ABI of the MI355X (gfx950) HEVC pixel-reconstruction back end.
 *
 * This is the drop-in boundary for libde265's reconstruction path.  All entry
 * points are extern "C", take plain pointers and sizes, and return
 * de265_error-compatible ints (0 == DE265_OK, libde265/de265.h:82-139).
 *
 * Two interfaces are declared here:
 *
 *  Interface 2 (frame level, this file, part A): the host parser records one
 *    picture's worth of reconstruction commands (what decode_TU,
 *    generate_inter_prediction_samples, read_pcm_samples, apply_deblocking_filter
 *    and apply_sample_adaptive_offset_sequential would have consumed) into a
 *    de265hip_picture_desc and the device reconstructs the whole picture.
 *    Replaces, per picture:
 *      libde265/slice.cc:3424      decode_TU            -> de265hip_tu records
 *      libde265/motion.cc:279      generate_inter_prediction_samples -> de265hip_pu
 *      libde265/slice.cc:4143      read_pcm_samples_internal -> de265hip_pcm
 *      libde265/deblock.cc:1020    apply_deblocking_filter
 *      libde265/sao.cc:318         apply_sample_adaptive_offset_sequential
 *    Hook site for submit: libde265/decctx.cc:757-766 (decode_some, after
 *    mark_all_CTB_progress(PREFILTER)).
 *
 *  Interface 1 (function level, part B): batched forms of the
 *    acceleration_functions slots (libde265/acceleration.h:29-201) that work
 *    on host pointers, for per-function parity tests and for a maintainer who
 *    wants to call single DSP functions.
 *
 * Units: all positions/sizes of TUs are in samples of their own component,
 * PUs/PCM/metadata in luma samples.  ChromaArrayType 1 (4:2:0), 2 (4:2:2) and 3 (4:4:4) are supported, with the
 * range-extension sample tools the reference implements: cross-component prediction, implicit and
 * explicit RDPCM, transform-skip rotation, transform skip beyond 4x4, intra smoothing switched off, chroma smoothing
 * in 4:4:4.  Monochrome (0) is supported for pictures WITHOUT prediction units (all-intra): the two chroma planes of such
 * a picture are empty (uploads / downloads of them are no-ops, a TU record with c_idx > 0 is out of range, a PCM block
 * carries n*n samples).  Monochrome with prediction units and extended_precision_processing return
 * DE265_ERROR_NOT_IMPLEMENTED_YET (the reference's inter path addresses chroma planes whatever the format,
 * motion.cc:296-305, which a monochrome picture does not have, and its transform path hard-codes
 * extended_precision_processing_flag = 0, transform.cc:535: nothing defined to match).
 */
#ifndef DE265_HIP_H
#define DE265_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: numeric values of de265_error (de265.h:82-139) ---- */
#define DE265HIP_OK                          0
#define DE265HIP_ERROR_OUT_OF_MEMORY         7
#define DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE 8   /* DE265_ERROR_CODED_PARAMETER_OUT_OF_RANGE */
#define DE265HIP_ERROR_INIT_FAILED           11  /* DE265_ERROR_LIBRARY_INITIALIZATION_FAILED */
#define DE265HIP_ERROR_DECODING              18  /* DE265_ERROR_UNSPECIFIED_DECODING_ERROR (device fault) */
#define DE265HIP_ERROR_NOT_IMPLEMENTED       502 /* DE265_ERROR_NOT_IMPLEMENTED_YET */

/* Proposed value for enum de265_acceleration (de265.h:391-402), above NEON=80. */
#define DE265HIP_ACCELERATION_LEVEL 90

#define DE265HIP_MAX_DPB_SLOTS 20
#define DE265HIP_MAX_REFS      16   /* MAX_NUM_REF_PICS */

/* ------------------------------------------------------------------ */
/* Part A: frame-level interface                                        */
/* ------------------------------------------------------------------ */

/* Picture constants: the SPS/PPS-derived values the reconstruction reads
 * (sps.h:188-236, pps.h:150-157). */
typedef struct de265hip_pic_params {
  int32_t width, height;              /* pic_{width,height}_in_luma_samples */
  int32_t bit_depth_luma;             /* BitDepth_Y */
  int32_t bit_depth_chroma;           /* BitDepth_C */
  int32_t chroma_format_idc;          /* 1, 2 or 3 (ChromaArrayType; separate_colour_plane is not supported) */
  int32_t log2_ctb_size;              /* Log2CtbSizeY 4..6 */
  int32_t log2_min_cb_size;           /* Log2MinCbSizeY >=3 */
  int32_t log2_min_tb_size;           /* Log2MinTrafoSize 2..5 */
  int32_t pcm_loop_filter_disable_flag;
  int32_t strong_intra_smoothing_enable_flag;
  int32_t constrained_intra_pred_flag;
  int32_t sample_adaptive_offset_enabled_flag;
  int32_t scaling_list_enable_flag;
  int32_t weighted_pred_flag;
  int32_t weighted_bipred_flag;
  int32_t pic_cb_qp_offset;
  int32_t pic_cr_qp_offset;
  int32_t loop_filter_across_tiles_enabled_flag;
  int32_t num_tile_columns;           /* >=1 */
  int32_t num_tile_rows;              /* >=1 */
  uint16_t col_bd[24];                /* colBd[0..num_tile_columns], CTB units */
  uint16_t row_bd[24];                /* rowBd[0..num_tile_rows],    CTB units */
  int32_t disable_deblocking;         /* DE265_DECODER_PARAM_DISABLE_DEBLOCKING */
  int32_t disable_sao;                /* DE265_DECODER_PARAM_DISABLE_SAO */
  /* range extensions (sps.h:66-84 sps_range_extension, pps.h pps_range_extension): all 0 for Main / Main10 */
  int32_t implicit_rdpcm_enabled_flag;            /* slice.cc:3456-3461, intrapred.cc:1102-1104 */
  int32_t transform_skip_rotation_enabled_flag;   /* transform.cc:393-395 */
  int32_t intra_smoothing_disabled_flag;          /* intrapred.cc:1085 */
  int32_t cross_component_prediction_enabled_flag;/* pps; transform.cc:592-598: residuals go through the int32 transform forms */
  int32_t extended_precision_processing_flag;     /* must be 0 */
  int32_t high_precision_offsets_enabled_flag;    /* WpOffsetBdShift = 0 instead of BitDepth - 8 (sps.cc:554-563) */
} de265hip_pic_params;

/* Size of the flat scaling-factor blob: ScalingFactor_Size0[6][4][4],
 * Size1[6][8][8], Size2[6][16][16], Size3[2][32][32] back to back (sps.h:51-58).
 * Limitation: a 32x32 CHROMA transform unit (4:4:4 only) with scaling lists makes the reference index Size3 beyond its two
 * matrices (undefined behaviour upstream); this library takes matrix 0 for intra and 1 for inter units there.  No fixture of
 * the reference defines the case: parity unpinned. */
#define DE265HIP_SCALING_BLOB_BYTES (6*16 + 6*64 + 6*256 + 2*1024)

/* Per slice segment values read by MC weighting, deblocking and SAO
 * (slice.h:147-248). */
typedef struct de265hip_slice_params {
  int32_t slice_type;                 /* 0=B 1=P 2=I (slice.h SLICE_TYPE_*) */
  int32_t slice_addr_rs;              /* SliceAddrRS */
  int32_t slice_deblocking_filter_disabled_flag;
  int32_t slice_beta_offset;          /* already *2 */
  int32_t slice_tc_offset;            /* already *2 */
  int32_t slice_loop_filter_across_slices_enabled_flag;
  int32_t slice_sao_luma_flag;
  int32_t slice_sao_chroma_flag;
  int32_t luma_log2_weight_denom;
  int32_t chroma_log2_weight_denom;   /* ChromaLog2WeightDenom */
  int16_t luma_weight[2][16];
  int16_t luma_offset[2][16];
  int16_t chroma_weight[2][16][2];
  int16_t chroma_offset[2][16][2];
  int8_t  ref_pic_list[2][16];        /* RefPicList -> DPB slot id */
} de265hip_slice_params;

/* Per CTB (image.h:180-190 CTB_info + slice.h:267-275 sao_info). */
typedef struct de265hip_ctb_info {
  uint16_t slice_addr_rs;             /* SliceAddrRS of the slice covering the CTB */
  uint16_t slice_idx;                 /* SliceHeaderIndex into the slice table.  Every CTB needs one: a CTB that no slice covers (a
                                         damaged stream; the reference's SAO skips it, sao.cc:140, decode_some marks it decoded,
                                         decctx.cc:751-757) has no representation here - de265hip_picture_build refuses the picture
                                         with DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE and the host conceals it as it sees fit */
  uint8_t  sao_type_idx;              /* (>>2*cIdx)&3: 0 off, 1 band, 2 edge */
  uint8_t  sao_eo_class;              /* (>>2*cIdx)&3 */
  uint8_t  sao_band_position[3];
  int8_t   sao_offset_val[3][4];      /* already sign-applied and << log2OffsetScale */
  uint8_t  pad[3];
} de265hip_ctb_info;

/* TU flags */
#define DE265HIP_TU_INTRA      0x01   /* cuPredMode == MODE_INTRA: predict, DST for 4x4 luma */
#define DE265HIP_TU_CBF        0x02   /* residual present */
#define DE265HIP_TU_TSKIP      0x04   /* transform_skip_flag[cIdx] */
#define DE265HIP_TU_BYPASS     0x08   /* cu_transquant_bypass_flag */
#define DE265HIP_TU_EXPLICIT_RDPCM      0x10   /* explicit_rdpcm_flag (inter CUs, transform skip / bypass TUs; slice.cc:3464-3468) */
#define DE265HIP_TU_EXPLICIT_RDPCM_VERT 0x20   /* explicit_rdpcm_dir */

/* One transform unit of one colour component, in decode order
 * (arguments of decode_TU, slice.cc:3424, plus the thread_context values it
 * reads: qP*Prime, transform_skip_flag, cu_transquant_bypass_flag, coeffList). */
typedef struct de265hip_tu {
  uint16_t x0, y0;                    /* component samples */
  uint8_t  log2_size;                 /* 2..5 */
  uint8_t  c_idx;                     /* 0..2 */
  uint8_t  flags;                     /* DE265HIP_TU_* */
  uint8_t  intra_mode;                /* 0..34 (only if INTRA) */
  int8_t   qp;                        /* qP{Y,Cb,Cr}Prime (transform.cc:362-368) */
  int8_t   res_scale_val;             /* ResScaleVal of cross-component prediction (chroma TUs of 4:4:4, slice.cc:3522-3541): 0, +-1, +-2, +-4, +-8.
                                         A chroma TU with res_scale_val != 0 is recorded even without CBF; it follows the luma TU of
                                         the same position and size in the TU array (slice.cc:3699-3750) */
  uint16_t n_coeff;                   /* nCoeff[cIdx] */
  uint32_t coeff_offset;              /* first entry in coeff_val/coeff_pos */
} de265hip_tu;

/* One prediction unit (arguments of generate_inter_prediction_samples,
 * motion.cc:279, and PBMotion motion.h:36-44). */
typedef struct de265hip_pu {
  uint16_t x, y;                      /* xP,yP luma samples */
  uint8_t  w, h;                      /* nPbW,nPbH */
  uint8_t  pred_flag;                 /* bit0 L0, bit1 L1 */
  uint8_t  pad;
  uint16_t slice_idx;                 /* slice table index (weights, RefPicList) */
  int8_t   ref_idx[2];
  int16_t  mv[2][2];                  /* [list][x,y] quarter-pel */
} de265hip_pu;

/* One PCM coding unit (slice.cc:4143-4183); samples already << (bitDepth-pcmBits),
 * stored Y (size^2) then Cb, Cr ((size/SubWidthC) x (size/SubHeightC) each) as uint16. */
typedef struct de265hip_pcm {
  uint16_t x0, y0;
  uint8_t  log2_cb_size;
  uint8_t  pad[3];
  uint32_t sample_offset;
} de265hip_pcm;

/* Per 4x4 luma unit flags: flattened view of deblk_info (image.h:70-74),
 * cb_info.PredMode/pcm_flag/cu_transquant_bypass (image.h:193-216) and
 * tu_info bit 7 (image.h:67). */
#define DE265HIP_BLK_INTRA       0x01
#define DE265HIP_BLK_NONZERO     0x02  /* TU_FLAG_NONZERO_COEFF of the covering TU */
#define DE265HIP_BLK_PCM         0x04  /* pcm_flag */
#define DE265HIP_BLK_BYPASS      0x08  /* cu_transquant_bypass */
#define DE265HIP_BLK_EDGE_TU_V   0x10  /* DEBLOCK_FLAG_VERTI */
#define DE265HIP_BLK_EDGE_TU_H   0x20  /* DEBLOCK_FLAG_HORIZ */
#define DE265HIP_BLK_EDGE_PB_V   0x40  /* DEBLOCK_PB_EDGE_VERTI */
#define DE265HIP_BLK_EDGE_PB_H   0x80  /* DEBLOCK_PB_EDGE_HORIZ */

/* Per 4x4 luma unit motion (PBMotion with RefPicList already resolved to DPB
 * slot ids, as derive_boundaryStrength compares them, deblock.cc:295-304). */
typedef struct de265hip_motion {
  int16_t mv[2][2];
  int8_t  ref_slot[2];                /* -1 when predFlag[l]==0 */
  uint8_t pad[2];
} de265hip_motion;

/* Everything the device needs for one picture.  Arrays are caller-owned and
 * only read during de265hip_picture_build(). */
typedef struct de265hip_picture_desc {
  de265hip_pic_params params;
  const uint8_t* scaling_factors;     /* DE265HIP_SCALING_BLOB_BYTES or NULL */
  int32_t n_slices;  const de265hip_slice_params* slices;
  int32_t n_ctbs;    const de265hip_ctb_info* ctbs;      /* raster, PicSizeInCtbsY */
  int32_t n_tus;     const de265hip_tu* tus;             /* decode order */
  int32_t n_coeffs;  const int16_t* coeff_val; const uint16_t* coeff_pos;   /* positions x + y * nT inside the TU's block (coeffPos, slice.cc:3408-3410).
                                   * Everything else of a descriptor is validated by de265hip_picture_build (PARAMETER_OUT_OF_RANGE); a
                                   * position beyond its block is caught on the device behind the upload: the picture still decodes
                                   * memory-safely and de265hip_decoder_sync returns DE265HIP_ERROR_DECODING */
  int32_t n_pus;     const de265hip_pu* pus;
  int32_t n_pcms;    const de265hip_pcm* pcms;
  int32_t n_pcm_samples; const uint16_t* pcm_samples;
  /* metadata planes, ceil(W/4) x ceil(H/4), row-major */
  const uint8_t* blk_flags;
  const int8_t*  blk_qp_y;
  const de265hip_motion* blk_motion;  /* may be NULL: the plane is then made on the device from the PU records (mv, RefPicList entry of
                                         the PU's slice as DPB slot; units no PU covers: no reference) - 12x fewer bytes to hand over */
} de265hip_picture_desc;

typedef struct de265hip_decoder de265hip_decoder;
typedef struct de265hip_picture de265hip_picture;

/* stage selector for de265hip_picture_run / get planes */
#define DE265HIP_STAGE_PREFILTER 0   /* after MC + residual + intra (pre-lf) */
#define DE265HIP_STAGE_DEBLOCKED 1
#define DE265HIP_STAGE_FINAL     2   /* after SAO */

/* Library / device */
const char* de265hip_version(void);
int  de265hip_device_count(void);

/* Decoder context: owns a HIP stream (kernels), a copy stream (command-buffer uploads), the device-resident
 * DPB, and the device side of every picture built on it (pooled arenas + pinned staging buffers).
 * device < 0 selects the current device.
 *
 * LIFETIME: a de265hip_picture belongs to the decoder it was built on.  The normal order is
 * picture_free() for every picture, then decoder_free().  decoder_free() with pictures still alive is
 * allowed: it waits for the decoder's streams, releases those pictures' device memory and ORPHANS their
 * handles; an orphaned handle may only be passed to de265hip_picture_free() (which then just releases the
 * handle) and to de265hip_picture_get_stats(); de265hip_picture_run() refuses it with
 * DE265_ERROR_CODED_PARAMETER_OUT_OF_RANGE.  de265hip_picture_free() never touches a freed decoder.
 *
 * GEOMETRY: a picture is reconstructed with the size and bit depths of ITS de265hip_pic_params.  de265hip_picture_build()
 * (re)allocates the destination slot (and the decoder's internal SAO target) when it is empty or when no picture of the
 * decoder is waiting for its first launch; otherwise a slot that holds planes of another geometry is re-allocated by
 * de265hip_picture_run(), i.e. in launch order: pictures of the old size that are built but not launched yet
 * (de265hip_pipeline_*, builds on several threads) keep their planes, and the re-allocation waits for the device.  That every reference slot holds a picture of the picture's own geometry is checked by de265hip_picture_run()
 * (DE265_ERROR_CODED_PARAMETER_OUT_OF_RANGE, nothing is launched): at build time a reference may still be under construction
 * on another thread.  de265hip_dpb_alloc() of a slot that queued pictures still use is the caller's error.
 *
 * THREADS: de265hip_picture_build() and de265hip_picture_free() may be called from several host threads
 * on the same decoder at once (the host stage of picture n+1 overlaps the device work of picture n:
 * decctx.cc:976-1178 is the reference's parallel host side); de265hip_picture_run(), the dpb_* calls and
 * decoder_sync() of one decoder belong to one thread at a time. */
int  de265hip_decoder_new(de265hip_decoder** out, int device);
void de265hip_decoder_free(de265hip_decoder*);
/* Lanes: picture-level concurrency inside one decoder (the reference has none: decctx.cc:904-910 "TODO ... frame-parallel
 * decoding").  With n_lanes > 1 (at most 4) pictures that do not depend on each other - hierarchical-B pictures of one
 * layer, the first pictures of the next closed GOP - are enqueued on different HIP streams; the calls stay the same
 * and stay in decode order (de265hip_picture_run / the pipeline), the device work is ordered by what the pictures' DPB
 * slots say: a picture waits for the pictures it references, and for every picture that still reads or writes the slot
 * it is decoded into.  Default 1 (DE265HIP_LANES overrides it when the decoder is created); each extra lane holds one more
 * spare picture in device memory.  Call it between pictures (it synchronises the decoder). */
int  de265hip_decoder_set_lanes(de265hip_decoder*, int n_lanes);
/* Allocate (or re-use) DPB slot `slot` for a picture of this geometry. */
int  de265hip_dpb_alloc(de265hip_decoder*, int slot, int width, int height,
                        int bit_depth_luma, int bit_depth_chroma);
/* The same for any chroma format (chroma_format_idc 1, 2, 3); de265hip_dpb_alloc is the 4:2:0 form.  A picture is always
 * reconstructed with the format of ITS de265hip_pic_params (de265hip_picture_build / _run re-allocate the slot if need be). */
int  de265hip_dpb_alloc_ex(de265hip_decoder*, int slot, int width, int height,
                           int bit_depth_luma, int bit_depth_chroma, int chroma_format_idc);
/* chroma_format_idc of the picture a slot holds; -1 for an unallocated slot */
int  de265hip_dpb_chroma_format(de265hip_decoder*, int slot);
/* Copy host planes into / out of a DPB slot.  stride_bytes as in
 * de265_get_image_plane (de265.h:173-174).  Sample type is uint8_t when the
 * component's bit depth is <=8, else uint16_t. */
int  de265hip_dpb_upload(de265hip_decoder*, int slot, int c_idx,
                         const void* src, ptrdiff_t stride_bytes);
int  de265hip_dpb_download(de265hip_decoder*, int slot, int c_idx,
                           void* dst, ptrdiff_t stride_bytes);
/* Set every sample of the picture in `slot` (allocated before: de265hip_dpb_alloc) to one value per component.  Replaces
 * de265_image::fill_image (image.cc) where libde265 synthesises a reference picture that the stream does not contain -
 * generate_unavailable_reference_picture, decctx.cc:1408-1434: 1 << (bitDepth - 1) in all three planes - so that pictures
 * which predict from it find it in the device-resident DPB (a stream joined at a CRA picture, a lost picture).
 * DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE for an unallocated slot or a value beyond the component's bit depth. */
int  de265hip_dpb_fill(de265hip_decoder*, int slot, int y, int cb, int cr);
/* Picture-level pipelining (the reference's parallel host side is decctx.cc:976-1178): copy a plane out
 * WITHOUT waiting on the host.  The copy is ordered behind everything enqueued on the decoder's stream so far (the
 * picture's kernels) and runs on the decoder's output stream, so the kernels of the pictures enqueued after it overlap
 * it; de265hip_dpb_wait() blocks until every copy-out of the slot has landed (and reports a failed picture like
 * decoder_sync).  A picture run into the slot later, an upload or a re-allocation waits for the copy-out by itself.
 * dst should be pinned memory (de265hip_host_alloc) - with pageable memory the call degrades to a staged copy.
 * de265hip_dpb_wait() may be called from another thread than the one that enqueues pictures. */
int  de265hip_dpb_download_async(de265hip_decoder*, int slot, int c_idx,
                                 void* dst, ptrdiff_t stride_bytes);
/* The same for all three planes of the picture in ONE call (what de265_get_image_plane hands out for c = 0, 1, 2,
 * de265.h:173-178): dst[c] == NULL skips plane c.  Planes in pinned memory leave by one kernel that stores into them (one launch
 * per picture, a few dozen workgroups: the link's rate without the runtime's per-plane blit kernels next to the decoder's own
 * kernels); planes in pageable memory, or with strides / addresses that are not multiples of 16, leave by hipMemcpy2DAsync as with
 * the call above, which is this one for a single plane. */
int  de265hip_dpb_download_planes_async(de265hip_decoder*, int slot, void* const dst[3], const ptrdiff_t stride_bytes[3],
                                        uint64_t* copy_out_id /* may be NULL */);
/* Wait for ONE copy-out: the one de265hip_dpb_download_planes_async numbered `copy_out_id` (1, 2, .. per slot since its
 * allocation).  de265hip_dpb_wait waits for the slot's LATEST copy-out - the wrong picture for an output queue that is
 * deeper than the DPB's cycle, where the slot has been decoded into and copied out again by the time a picture is collected.
 * Reports the picture's failure like de265hip_dpb_wait.  A number from before the slot's last re-allocation is not an error:
 * that copy-out landed when the slot was re-allocated. */
int  de265hip_dpb_wait_copy_out(de265hip_decoder*, int slot, uint64_t copy_out_id);
int  de265hip_dpb_wait(de265hip_decoder*, int slot);
/* Pinned host memory for the planes pictures are copied out to: what a libde265 host installs as its
 * de265_image_allocation (de265.h:325-343).  NULL when the allocation fails. */
void* de265hip_host_alloc(size_t bytes);
void  de265hip_host_free(void*);
/* Geometry of the picture a DPB slot currently holds (what upload / download copy): the counterpart of
 * de265_get_image_width / _height / de265_get_bits_per_pixel (de265.h:160-171) for a device-resident picture.
 * DE265_ERROR_PARAMETER_OUT_OF_RANGE for an unallocated slot; any out pointer may be NULL. */
int  de265hip_dpb_info(de265hip_decoder*, int slot, int* width, int* height,
                       int* bit_depth_luma, int* bit_depth_chroma);
/* Device pointer + stride (bytes) of a DPB plane (for zero-copy consumers). */
int  de265hip_dpb_plane(de265hip_decoder*, int slot, int c_idx,
                        void** dev_ptr, ptrdiff_t* stride_bytes);

/* Reference-picture exchange (open GOPs / inter-GOP references): copy the three planes of
 * src_dec's slot src_slot into dst_dec's slot dst_slot, device to device (hipMemcpyPeerAsync when the
 * decoders sit on different GPUs of this process, one xGMI hop; a plain device copy otherwise), ordered
 * behind everything enqueued on src_dec's stream so far AND on dst_dec's stream so far (pictures already queued on dst_dec
 * that read the slot's old content finish first; so does a pending copy-out of it); dst_dec's stream waits for the copy.  The
 * destination slot is (re)allocated to the source's geometry.  Between PROCESSES (one rank per GPU) the same
 * planes are sent with RCCL on zero-copy tensor views of de265hip_dpb_plane (libde265_amd/farm.py). */
int  de265hip_dpb_copy(de265hip_decoder* src_dec, int src_slot, de265hip_decoder* dst_dec, int dst_slot);

/* Build: host-side preprocessing (intra availability + dependency levels,
 * level sort, MC task split) into a pinned staging buffer and ASYNCHRONOUS upload of the command
 * buffers on the decoder's copy stream into a pooled device arena (no allocation and no host-side wait
 * in the steady state).  The desc's arrays may be reused as soon as the call returns; the first
 * de265hip_picture_run() of the picture waits for the upload on the device. */
int  de265hip_picture_build(de265hip_decoder*, int dst_slot,
                            const de265hip_picture_desc*, de265hip_picture** out);
/* Run: enqueue all reconstruction kernels of the picture on the decoder's
 * stream (asynchronous).  last_stage: DE265HIP_STAGE_*. */
int  de265hip_picture_run(de265hip_decoder*, de265hip_picture*, int last_stage);
/* Wait for everything enqueued on the decoder's stream. */
int  de265hip_decoder_sync(de265hip_decoder*);
void de265hip_picture_free(de265hip_picture*);
/* Convenience: build + run(FINAL) + sync + free. */
int  de265hip_decode_picture(de265hip_decoder*, int dst_slot,
                             const de265hip_picture_desc*);

/* ---- recorder: the same interface as incremental calls, for a host parser that produces TUs,
 * PUs and PCM blocks one at a time (what decode_TU / generate_inter_prediction_samples /
 * read_pcm_samples receive, slice.cc:3424, motion.cc:279, slice.cc:4185).  The recorder only
 * accumulates a de265hip_picture_desc on the host; de265hip_recorder_submit == picture_build. ---- */
typedef struct de265hip_recorder de265hip_recorder;
int  de265hip_recorder_new(de265hip_recorder** out, const de265hip_pic_params* params,
                           const uint8_t* scaling_factors /* NULL unless scaling lists are on */);
void de265hip_recorder_free(de265hip_recorder*);
/* TU in decode order; vals/pos hold tu->n_coeff entries (coeff_offset is filled in here) */
int  de265hip_record_tu(de265hip_recorder*, const de265hip_tu* tu, const int16_t* vals, const uint16_t* pos);
int  de265hip_record_pu(de265hip_recorder*, const de265hip_pu* pu);
/* samples: Y (size^2) then Cb, Cr ((size/SubWidthC) x (size/SubHeightC) each), already << (bitDepth - pcmBits) */
int  de265hip_record_pcm(de265hip_recorder*, int x0, int y0, int log2_cb_size, const uint16_t* samples);
int  de265hip_record_slice(de265hip_recorder*, const de265hip_slice_params* slice);      /* returns its index via n_slices-1 */
int  de265hip_record_ctb(de265hip_recorder*, int ctb_addr_rs, const de265hip_ctb_info* info);
/* metadata planes (ceil(W/4) x ceil(H/4)); copied */
int  de265hip_record_blk_planes(de265hip_recorder*, const uint8_t* blk_flags, const int8_t* blk_qp_y,
                                const de265hip_motion* blk_motion /* may be NULL */);
/* the accumulated description (valid until the recorder is modified or freed) */
const de265hip_picture_desc* de265hip_recorder_desc(de265hip_recorder*);
int  de265hip_recorder_submit(de265hip_decoder*, int dst_slot, de265hip_recorder*, de265hip_picture** out);

/* ---- pipeline: picture-level pipelining (the reference's parallel host side is decctx.cc:976-1178).
 * While the host parser works on picture n+1, `n_workers` threads owned by the pipeline run this library's host stage for the
 * pictures before it - `prepare` (the integration fills a recorder from what its parser left behind: it is called on a worker
 * thread, several pictures at once) and de265hip_recorder_submit (= de265hip_picture_build) - and the device reconstructs the
 * pictures before those.  Pictures are LAUNCHED in submission order (a picture's kernels read the DPB slots its references were
 * launched into): de265hip_picture_run(STAGE_FINAL) + de265hip_dpb_download_async of every non-NULL `planes[c]` (pinned memory,
 * de265hip_host_alloc; strides as in de265hip_dpb_download).  submit() returns as soon as there is room (at most 4 n_workers + 4
 * pictures between parser and device); nobody waits for a picture until de265hip_pipeline_wait(ticket) - what a decoder calls
 * when the picture is about to be output or read (de265.cc:392 de265_peek_next_picture).  While a pipeline exists, run / dpb_* /
 * sync of its decoder belong to the pipeline; dpb_alloc of slots no queued picture uses is allowed.  An error of prepare, build
 * or run is returned by wait() of that ticket (or by drain()).  Free the pipeline before its decoder.  oracle/f1_recorder.cc
 * (test infrastructure: the patched reference decoder) is the libde265-side user. ---- */
typedef struct de265hip_pipeline de265hip_pipeline;
typedef int (*de265hip_prepare_fn)(void* user, de265hip_recorder** out);   /* 0 and a filled recorder (the pipeline frees it) */
int  de265hip_pipeline_new(de265hip_pipeline** out, de265hip_decoder*, int n_workers /* 1..16 */);
int  de265hip_pipeline_submit(de265hip_pipeline*, int dst_slot, de265hip_prepare_fn prepare, void* user,
                              void* const planes[3], const ptrdiff_t stride_bytes[3], uint64_t* ticket);
/* The same for a host that already holds the picture's description (no prepare step): `desc` and the arrays it points to
 * stay valid and unchanged until the ticket has been waited for (or the pipeline drained). */
int  de265hip_pipeline_submit_desc(de265hip_pipeline*, int dst_slot, const de265hip_picture_desc* desc,
                                   void* const planes[3], const ptrdiff_t stride_bytes[3], uint64_t* ticket);
int  de265hip_pipeline_wait(de265hip_pipeline*, uint64_t ticket);
int  de265hip_pipeline_drain(de265hip_pipeline*);          /* every submitted picture launched, finished and copied out */
void de265hip_pipeline_free(de265hip_pipeline*);           /* drains first */

/* de265hip_picture_build in two steps, for a host that wants every HIP call of a decoder issued by ONE of its threads (the
 * pipeline does: many threads calling into the HIP runtime for one device queue up behind its locks).  _build_host: the host
 * stage only - validation, MC tasks, staging into pinned memory -, on any thread; _enqueue: the upload and the kernels that
 * prepare the picture on the device (the scan of the TU records), before de265hip_picture_run (which calls it itself if nobody
 * has).  A picture that was built but never enqueued may be freed. */
int  de265hip_picture_build_host(de265hip_decoder*, int dst_slot, const de265hip_picture_desc*, de265hip_picture** out);
int  de265hip_picture_enqueue(de265hip_picture*);
int  de265hip_picture_enqueue_batch(de265hip_picture** pics, int n);   /* pictures of ONE decoder: their scans share their kernel launches */
int  de265hip_picture_ready(de265hip_picture*);     /* 1: de265hip_picture_run will not wait for the device side of the build; 0: not yet */

/* Profiling aid: the host stage of de265hip_picture_build `reps` times, without a GPU and without any HIP call. */
int  de265hip_debug_build_host_only(const de265hip_picture_desc*, int reps);
/* FNV-1a hash over everything the last de265hip_debug_build_host_only of this thread would have uploaded (regression
 * net for changes to the host stage: tools/exp/build_hash.py). */
uint64_t de265hip_debug_last_build_hash(void);
/* mode 0: the round-3 host scan of the TU records; 1: the host's part of a build with the device-side scan (scan_core.h); 2: the
 * same plus the CPU rehearsal of the scan's passes.  keep: the last picture, its arena in host memory (de265hip_debug_picture_*,
 * de265hip_picture_free). */
int  de265hip_debug_build_host_only_ex(const de265hip_picture_desc*, int reps, int mode, de265hip_picture** keep);
/* Test entry points: fault injection (a run other runs wait for is left out of the pictures built from now on; spin_limit
 * bounds k_run's dependency waits, 0 = default), and read-back of a picture's run-side structures (offsets and counts, then
 * bytes of its arena) for tests/test_scan_equivalence.py. */
int  de265hip_debug_fault_injection(de265hip_decoder*, int drop_producer, uint32_t spin_limit);
int  de265hip_debug_picture_layout(de265hip_picture*, int64_t out[32]);
int  de265hip_debug_picture_read(de265hip_picture*, int64_t offset, int64_t bytes, void* dst);

/* Introspection used by bench/tests */
typedef struct de265hip_picture_stats {
  int32_t n_levels;          /* intra dependency levels at TU granularity */
  int32_t n_tu_tasks;
  int32_t n_mc_tasks;
  int32_t n_runs;            /* intra runs (wavefront tasks of the single-launch run kernel) */
  int32_t n_run_levels;      /* longest producer->consumer chain of runs */
  int32_t n_in_run_levels;   /* sum over runs of their in-run dependency levels (barrier steps of the run kernel) */
  int64_t device_bytes;      /* command-buffer bytes resident in HBM */
  int64_t alg_bytes_mc;      /* algorithmic bytes, SURVEY 8d definitions */
  int64_t alg_bytes_resid;
  int64_t alg_bytes_intra;       /* run kernel only; see alg_bytes_intra_front */
  int64_t alg_bytes_deblock;
  int64_t alg_bytes_sao;
  int64_t alg_bytes_intra_front; /* the part of the intra work done by the front kernel (alg_bytes_intra: the run kernel's part) */
  int32_t n_front_runs;          /* of n_runs: runs without producers, reconstructed ahead of the run kernel */
  int32_t pad;
} de265hip_picture_stats;
int  de265hip_picture_get_stats(const de265hip_picture*, de265hip_picture_stats*);

/* Per-kernel device time of the last N runs, measured with hipEvents on the
 * decoder's stream.  kernel ids: */
#define DE265HIP_K_MC        0
#define DE265HIP_K_RESID     1   /* level-0 TU kernel (inter residual) */
#define DE265HIP_K_INTRA     2   /* sum of intra level launches */
#define DE265HIP_K_BS        3
#define DE265HIP_K_DEBLOCK_V 4
#define DE265HIP_K_DEBLOCK_H 5
#define DE265HIP_K_SAO       6
#define DE265HIP_K_PCM       7
#define DE265HIP_K_INTRA_FRONT 8 /* intra runs without producers (k_intra_front), ahead of the run kernel */
#define DE265HIP_K_COUNT     9
/* enable: 0 = off, 1 = every kernel, otherwise a mask with bit (id + 1) set for each kernel id to be timed
 * (DE265HIP_PROFILE_ONLY(id)): every timed launch costs two event records on the stream. */
#define DE265HIP_PROFILE_ONLY(id) (2 << (id))
int  de265hip_set_profiling(de265hip_decoder*, int enable);
/* Accumulated ms and launch counts since the last reset (sync first). */
int  de265hip_get_kernel_times(de265hip_decoder*, double ms[DE265HIP_K_COUNT],
                               int64_t launches[DE265HIP_K_COUNT], int reset);

/* Diagnostic: the neighbour units (bit u as in the availability masks of intrapred.cc:437-527: 4-sample units of the
 * left column bottom-up, the corner, the top row left to right) whose samples an intra TU of this size and mode can
 * read - prediction (intrapred.cc:903-1069), edge filters and smoothing incl. the strong-smoothing decision (:816-889).
 * The build uses it to order intra TUs only behind the producers they really depend on. */
int  de265hip_intra_used_units(int log2_size, int intra_mode, int luma, uint64_t* units);

/* Host helper: edge-flag derivation (deblock.cc:31-225 derive_edgeFlags) from
 * CU/TU structure, for hosts that do not already run it.  cb_log2_size /
 * cb_part_mode are per MinCb unit (top-left only, 0 elsewhere), tu_split per
 * MinTb unit with bit d set when split_transform_flag at depth d
 * (image.h:67-68).  ORs DE265HIP_BLK_EDGE_* into blk_flags. */
int  de265hip_derive_edge_flags(const de265hip_pic_params*,
                                const de265hip_slice_params* slices, int n_slices,
                                const de265hip_ctb_info* ctbs,
                                const uint8_t* cb_log2_size, const uint8_t* cb_part_mode,
                                const uint8_t* tu_split, uint8_t* blk_flags);

/* ------------------------------------------------------------------ */
/* Part B: function-level (vtable-shaped) interface, batched            */
/* ------------------------------------------------------------------ */
/* Each call processes `n` independent blocks that live in ONE host plane.
 * xy[2*i],xy[2*i+1] is the block origin in the plane.  Semantics per block
 * are those of the acceleration_functions slot named in the comment. */

/* transform_add_{8,16}[log2-2] / transform_4x4_dst_add_{8,16}
 * (acceleration.h:152-159).  coeffs: n * nT*nT dense row-major int16. */
int de265hip_fn_transform_add(int log2_size, int dst_type /*0 DCT,1 DST*/, int bit_depth,
                              void* plane, ptrdiff_t stride_samples, int plane_h,
                              int n, const int32_t* xy, const int16_t* coeffs);
/* transform_skip_residual + add_residual (acceleration.h:169-178) */
int de265hip_fn_transform_skip_add(int log2_size, int bit_depth,
                                   void* plane, ptrdiff_t stride_samples, int plane_h,
                                   int n, const int32_t* xy, const int16_t* coeffs);
/* transform_bypass + add_residual (acceleration.h:143,169) */
int de265hip_fn_transform_bypass_add(int log2_size, int bit_depth,
                                     void* plane, ptrdiff_t stride_samples, int plane_h,
                                     int n, const int32_t* xy, const int16_t* coeffs);
/* put_hevc_qpel_{8,16}[dX][dY] (acceleration.h:100,118): src plane -> int16
 * out blocks (n * w*h, stride w).  Source blocks must lie inside the plane
 * including their filter margins (the vtable contract). */
int de265hip_fn_put_qpel(int bit_depth, const void* src_plane, ptrdiff_t stride_samples,
                         int plane_w, int plane_h, int w, int h, int dx, int dy,
                         int n, const int32_t* xy, int16_t* out);
/* put_hevc_epel{,_h,_v,_hv}_{8,16} (acceleration.h:87-116) */
int de265hip_fn_put_epel(int bit_depth, const void* src_plane, ptrdiff_t stride_samples,
                         int plane_w, int plane_h, int w, int h, int mx, int my,
                         int n, const int32_t* xy, int16_t* out);
/* put_unweighted_pred / put_weighted_pred / put_weighted_bipred /
 * put_weighted_pred_avg (acceleration.h:31-64).  mode: 0 unweighted,
 * 1 weighted uni, 2 avg, 3 weighted bi.  src blocks: n * w*h int16. */
int de265hip_fn_put_pred(int mode, int bit_depth, void* plane, ptrdiff_t stride_samples,
                         int plane_h, int w, int h, int n, const int32_t* xy,
                         const int16_t* src0, const int16_t* src1,
                         int w0, int o0, int w1, int o1, int log2wd);

#ifdef __cplusplus
}
#endif
#endif /* DE265_HIP_H */
