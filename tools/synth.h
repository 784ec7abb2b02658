/*
 * synth.h -- synthetic "conformance-style" command-buffer generator
 * (SURVEY.md 8d, configs 2-5).  Stands in for libde265's CABAC/slice parser:
 * it emits, for one picture, exactly what the host parser would hand to the
 * reconstruction back end (a de265hip_picture_desc) plus the true decode
 * order and the CU/TU structure arrays used to cross-check edge-flag
 * derivation.  Test/bench infrastructure; deterministic for a given config.
 */
#ifndef DE265_SYNTH_H
#define DE265_SYNTH_H
#include <stdint.h>
#include "../include/de265_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct synth_config {
  int32_t width, height;        /* multiples of 8 */
  int32_t bit_depth;            /* 8..12, luma == chroma */
  int32_t log2_ctb_size;        /* 4..6 */
  int32_t log2_min_tb_size;     /* 2..5 (min CB is fixed at 8) */
  int32_t log2_max_tb_size;     /* <=5 */
  uint64_t seed;
  int32_t slice_type;           /* 0 B, 1 P, 2 I */
  int32_t intra_pct;            /* % of CUs intra in P/B pictures */
  int32_t n_ref_slots;          /* number of DPB slots usable as reference (>=1 for P/B) */
  int8_t  ref_slots[16];        /* DPB slot ids */
  int32_t bi_pct;               /* % bi-predicted PUs in B pictures */
  int32_t mv_sigma_qpel;        /* std-dev of MVs in quarter-pel */
  int32_t weighted_pred;        /* sets weighted_pred_flag and weighted_bipred_flag */
  int32_t n_slices;             /* >=1, ignored when tiles are used with slice_per_tile */
  int32_t tile_cols, tile_rows; /* >=1 */
  int32_t slice_per_tile;
  int32_t cbf_pct;              /* % of TUs (per component) with residual */
  int32_t tskip_pct;            /* % of 4x4 TUs using transform skip */
  int32_t bypass_pct;           /* % of CUs with cu_transquant_bypass */
  int32_t pcm_pct;              /* % of intra CUs (8..32) coded PCM */
  int32_t pcm_loop_filter_disable;
  int32_t scaling_list;
  int32_t constrained_intra_pred;
  int32_t strong_intra_smoothing;
  int32_t deblocking;           /* 0: slice_deblocking_filter_disabled everywhere */
  int32_t sao;                  /* 0: off */
  int32_t lf_across_slices_pct; /* % slices with loop filter across slices enabled */
  int32_t lf_across_tiles;
  int32_t big_coeff_pct;        /* % coefficients drawn near +-32767 (clip paths) */
  int32_t qp_min, qp_max;       /* QP_Y range */
  int32_t amp;                  /* allow asymmetric partitions */
  int32_t split_bias;           /* 0..100, higher = smaller blocks */
  /* range extensions; all 0 = Main / Main10 */
  int32_t chroma_format;        /* 0 -> 1 (4:2:0); 2 = 4:2:2, 3 = 4:4:4 */
  int32_t cross_component_pct;  /* 4:4:4: % of eligible TUs (luma cbf, inter CU or chroma mode 4) with ResScaleVal != 0 per chroma component */
  int32_t implicit_rdpcm;       /* sps flag */
  int32_t explicit_rdpcm_pct;   /* % of inter transform-skip / bypass TUs with explicit_rdpcm_flag (sets the sps flag when > 0) */
  int32_t rotation;             /* transform_skip_rotation_enabled_flag */
  int32_t intra_smoothing_disabled;
  int32_t log2_max_tskip_size;  /* 0 -> 2; transform skip on TUs up to this size (tskip_pct applies) */
  int32_t high_precision_offsets;
  int32_t monochrome;           /* chroma_format_idc 0 (intra pictures only: the reference's inter path has no defined behaviour without chroma planes) */
} synth_config;

typedef struct synth_picture synth_picture;

void synth_default_config(synth_config*, int width, int height, int bit_depth, int slice_type,
                          uint64_t seed);
synth_picture* synth_generate(const synth_config*);
void synth_free(synth_picture*);
const de265hip_picture_desc* synth_desc(const synth_picture*);
/* true decode order: entries ORACLE_ORD_* | index (see oracle/hevc_oracle.h) */
const uint32_t* synth_order(const synth_picture*, int32_t* n);
/* CU/TU structure for edge-flag cross-checks (per MinCb / per MinTb unit) */
const uint8_t* synth_cb_log2_size(const synth_picture*);
const uint8_t* synth_cb_part_mode(const synth_picture*);
const uint8_t* synth_tu_split(const synth_picture*);
/* blk_flags without the edge bits (input of derive_edge_flags) */
const uint8_t* synth_blk_flags_noedge(const synth_picture*);

/* Seeded reference/initial picture: smooth gradient + sinusoid + noise,
 * samples in [0, 2^bd-1].  plane = uint8_t* or uint16_t*, stride in samples. */
void synth_fill_plane(void* plane, int stride, int w, int h, int bit_depth, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
