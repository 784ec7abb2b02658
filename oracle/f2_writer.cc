/*
 * f2_writer.cc -- SURVEY.md 8(f2): a synthetic, conformance-style HEVC bitstream WRITER (build container only).
 *
 * TEST INFRASTRUCTURE, never part of the product.  The reference ships no conformance streams and its encoder CLI emits
 * all-intra pictures only (its P path asserts/crashes, encoder-syntax.cc:1397-1437 is a stub), so no real bitstream
 * could reach the inter, weighted-prediction, PCM, cu_qp_delta, AMP, multi-slice, deblocking-override and SAO-merge
 * paths of the hot path end to end.  This tool writes such streams: a seeded random walk over the HEVC syntax
 * (ITU-T H.265 7.3: VPS/SPS/PPS, slice_segment_header with explicit short-term RPS and pred_weight_table, SAO, coding
 * quadtree, CU, PU, transform tree, transform unit, residual_coding, PCM), every element binarised and
 * context-selected as the reference DECODER parses it (slice.cc read_* / decode_* cited at each function).  The
 * pictures are noise-like but every stream is legal syntax, and WHAT they decode to is defined by the reference
 * decoder alone: oracle/_ref/f1_dec (the recording libde265) turns the stream into the authoritative
 * de265hip_picture_desc + picture MD5 fixtures (tools/make_stream_golden.py), which the oracle, the HIP path and the
 * HIP-backed libde265 must then reproduce bit-exactly (tests/test_stream_golden.py).
 *
 * Linked against the compiled reference for exactly three things it would make no sense to restate: the CABAC
 * arithmetic ENCODER + context initialisation tables (cabac.cc, contextmodel.cc), the coefficient scan tables (scan.cc)
 * and the parameter-set writers (vps.cc/sps.cc/pps.cc write()).  The slice-level syntax below is this file's own: the
 * reference's encoder cannot write it.
 *
 *   f2_writer out=stream.bin w=416 h=240 pics=6 gop=B seed=1 [key=value ...]      (keys: see struct Cfg)
 */
#include "libde265/cabac.h"
#include "libde265/de265.h"
#include "libde265/md5.h"
#include "libde265/contextmodel.h"
#include "libde265/decctx.h"
#include "libde265/nal.h"
#include "libde265/pps.h"
#include "libde265/scan.h"
#include "libde265/slice.h"
#include "libde265/sps.h"
#include "libde265/vps.h"

#include <algorithm>
#include <memory>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace {

/* ---------------------------------------------------------------- configuration ---------------------------------------------------------------- */
struct Cfg {
  std::string out = "f2.bin", gop = "I";
  int w = 416, h = 240, bits = 8, pics = 3, slices = 1, seed = 1;
  int log2ctb = 5, log2mincb = 3, log2mintb = 2, log2maxtb = 5, depth_intra = 2, depth_inter = 2;
  int amp = 1, sao = 1, pcm = 1, pcm_bits = 0 /*0: bit depth*/, pcm_lf_off = 0, strong = 1, tmvp = 1, cip = 0;
  int qp = 30, cuqpd = 1, qg_depth = 1, cb_off = 1, cr_off = -2, slice_cqp = 1;
  int wp = 0, sdh = 0, tskip = 0, tqbypass = 0;
  int deblock = 1, lf_slices = 1, cabac_init = 1, lists_mod = 0, merge_cand = 5, par_mrg = 2;
  int nref = 2, max_level = 24, big_mv = 1;
  int wpp = 0, tile_cols = 1, tile_rows = 1, tile_uniform = 1, lf_tiles = 1, md5 = 1;
  int idr_period = 0;                 /* gop=I/P/LDB: an IDR picture every so many pictures (POC restarts, the DPB is flushed) */
  int dep = 0;                        /* percent: a cut point inside a slice starts a DEPENDENT slice segment (7.3.6.1, 9.3.1) */
  int scaling = 0;                    /* 1: scaling lists on, default lists (sps); 2: explicit lists in the PPS (7.3.4 scaling_list_data) */
  int dens = 50;                      /* percent: how often cbf flags are set */
  /* range extensions: chroma format and the sample tools of sps / pps_range_extension the reference implements */
  int chroma = 1;                     /* chroma_format_idc: 1, 2 (4:2:2), 3 (4:4:4) */
  int xcc = 0;                        /* cross_component_prediction_enabled_flag (4:4:4) */
  int irdpcm = 0, erdpcm = 0;         /* implicit / explicit_rdpcm_enabled_flag */
  int rot = 0;                        /* transform_skip_rotation_enabled_flag */
  int tskip_log2 = 2;                 /* log2_max_transform_skip_block_size */
  int nosmooth = 0;                   /* intra_smoothing_disabled_flag */
  int hpo = 0;                        /* high_precision_offsets_enabled_flag */
};

struct Kv { const char* k; int Cfg::* p; };
const Kv KV[] = {
  {"w",&Cfg::w},{"h",&Cfg::h},{"bits",&Cfg::bits},{"pics",&Cfg::pics},{"slices",&Cfg::slices},{"seed",&Cfg::seed},
  {"log2ctb",&Cfg::log2ctb},{"log2mincb",&Cfg::log2mincb},{"log2mintb",&Cfg::log2mintb},{"log2maxtb",&Cfg::log2maxtb},
  {"depth_intra",&Cfg::depth_intra},{"depth_inter",&Cfg::depth_inter},{"amp",&Cfg::amp},{"sao",&Cfg::sao},{"pcm",&Cfg::pcm},
  {"pcm_bits",&Cfg::pcm_bits},{"pcm_lf_off",&Cfg::pcm_lf_off},{"strong",&Cfg::strong},{"tmvp",&Cfg::tmvp},{"cip",&Cfg::cip},
  {"qp",&Cfg::qp},{"cuqpd",&Cfg::cuqpd},{"qg_depth",&Cfg::qg_depth},{"cb_off",&Cfg::cb_off},{"cr_off",&Cfg::cr_off},
  {"slice_cqp",&Cfg::slice_cqp},{"wp",&Cfg::wp},{"sdh",&Cfg::sdh},{"tskip",&Cfg::tskip},{"tqbypass",&Cfg::tqbypass},
  {"deblock",&Cfg::deblock},{"lf_slices",&Cfg::lf_slices},{"cabac_init",&Cfg::cabac_init},{"lists_mod",&Cfg::lists_mod},
  {"merge_cand",&Cfg::merge_cand},{"par_mrg",&Cfg::par_mrg},{"nref",&Cfg::nref},{"max_level",&Cfg::max_level},
  {"big_mv",&Cfg::big_mv},{"dens",&Cfg::dens},{"wpp",&Cfg::wpp},{"tile_cols",&Cfg::tile_cols},{"tile_rows",&Cfg::tile_rows},
  {"tile_uniform",&Cfg::tile_uniform},{"lf_tiles",&Cfg::lf_tiles},{"md5",&Cfg::md5},{"scaling",&Cfg::scaling},{"dep",&Cfg::dep},{"idr_period",&Cfg::idr_period},
  {"chroma",&Cfg::chroma},{"xcc",&Cfg::xcc},{"irdpcm",&Cfg::irdpcm},{"erdpcm",&Cfg::erdpcm},{"rot",&Cfg::rot},{"tskip_log2",&Cfg::tskip_log2},
  {"nosmooth",&Cfg::nosmooth},{"hpo",&Cfg::hpo},
};

[[noreturn]] void die(const char* msg) { fprintf(stderr, "f2_writer: %s\n", msg); exit(2); }

/* ---------------------------------------------------------------- random source ---------------------------------------------------------------- */
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull) { for (int i=0;i<8;i++) next(); }
  uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
  int below(int n) { return n <= 1 ? 0 : (int)((next() >> 33) % (uint64_t)n); }
  int range(int lo, int hi) { return lo + below(hi - lo + 1); }
  bool pct(int p) { return below(100) < p; }
};

/* ---------------------------------------------------------------- coding structure ---------------------------------------------------------------- */
struct PicPlan {
  int poc = 0, type = SLICE_TYPE_I;
  bool idr = false;
  std::vector<int> refs;              /* POCs this picture may predict from */
};

std::vector<PicPlan> plan_gop(const Cfg& c)
{
  std::vector<PicPlan> v;
  auto add = [&](int poc, int type, std::vector<int> refs) { PicPlan p; p.poc = poc; p.type = type; p.refs = refs; p.idr = v.empty(); v.push_back(p); };
  const int period = c.idr_period > 0 ? c.idr_period : 1 << 30;
  if (c.gop == "I") {
    for (int i=0;i<c.pics;i++) { add(i % period, SLICE_TYPE_I, {}); v.back().idr = i % period == 0; }
  } else if (c.gop == "P" || c.gop == "LDB") {              /* low delay: the nref previous pictures (of the same coded video sequence) */
    for (int i=0;i<c.pics;i++) {
      const int j = i % period;                               /* POC = position behind the last IDR */
      std::vector<int> r;
      for (int k=1;k<=c.nref && j-k>=0;k++) r.push_back(j-k);
      add(j, j==0 ? SLICE_TYPE_I : (c.gop == "P" ? SLICE_TYPE_P : SLICE_TYPE_B), r);
      v.back().idr = j == 0;
    }
  } else if (c.gop == "B") {                                /* random access, hierarchical GOP of 4 */
    add(0, SLICE_TYPE_I, {});
    for (int base=0; (int)v.size() < c.pics; base+=4) {
      std::vector<int> back = {base}; if (base>=4) back.push_back(base-4);
      add(base+4, SLICE_TYPE_P, back);
      add(base+2, SLICE_TYPE_B, {base, base+4});
      add(base+1, SLICE_TYPE_B, {base, base+2, base+4});
      add(base+3, SLICE_TYPE_B, {base+2, base, base+4});
    }
    v.resize(c.pics);
  } else die("gop must be I, P, LDB or B");
  return v;
}

/* ---------------------------------------------------------------- the writer ---------------------------------------------------------------- */
enum { PM_NONE=0, PM_INTRA=1, PM_INTER=2, PM_SKIP=3 };

struct SliceCtx {
  int type, addr /*SliceAddrRS*/, qp, n_l0, n_l1, max_merge, mvd_l1_zero, init_type;
  bool sao_luma, sao_chroma;
};

struct Writer {
  Cfg c;
  Rng rng;
  CABAC_encoder_bitstream hdr;                                  /* NAL header, parameter sets, slice segment headers */
  CABAC_encoder_bitstream cab;                                  /* slice segment data */
  context_model_table models;
  struct Nal { std::vector<uint8_t> bytes; int pic; };          /* without start code; pic: index in decode order, -1 for parameter sets */
  std::vector<Nal> nals;
  error_queue errq;
  std::shared_ptr<video_parameter_set> vps;
  std::shared_ptr<seq_parameter_set> sps;
  std::shared_ptr<pic_parameter_set> pps;

  int W4, H4, ctbW, ctbH, nCtb;
  std::vector<uint8_t> ct_depth, skipf, pmode, pcmf, ipm;       /* per 4x4 block */
  std::vector<uint8_t> ipmc, ipmc4;                             /* IntraPredModeC (4:2:2: after Table 8-3), intra_chroma_pred_mode == 4 */
  bool rext() const { return c.chroma != 1 || c.xcc || c.irdpcm || c.erdpcm || c.rot || c.tskip_log2 > 2 || c.nosmooth || c.hpo; }
  std::vector<int> ctb_slice;                                   /* SliceAddrRS per CTB of the current picture, -1: not yet coded */
  SliceCtx S;
  struct Stats { long n_coeffs=0, abs_sum=0, n_pus=0, n_pcms=0, n_cus=0, n_resid=0; } st;   /* per picture, for the .chk sidecar */
  FILE* fchk = nullptr;
  bool qpd_coded = false;

  explicit Writer(const Cfg& cfg) : c(cfg), rng(cfg.seed) {}

  /* ---------- NAL plumbing: one CABAC_encoder_bitstream per NAL (it inserts the emulation prevention bytes) ---------- */
  void nal_begin(int type) { hdr.reset(); hdr.write_bits(0,1); hdr.write_bits(type,6); hdr.write_bits(0,6); hdr.write_bits(1,3); }
  void nal_end(int pic, bool with_data) {
    Nal n; n.pic = pic;
    n.bytes.assign(hdr.data(), hdr.data()+hdr.size());
    /* the slice segment header ends in a non-zero byte (byte_alignment()), so the emulation prevention of the data does not depend on it */
    if (with_data) n.bytes.insert(n.bytes.end(), cab.data(), cab.data()+cab.size());
    nals.push_back(n);
  }

  /* ---------- parameter sets (vps.cc/sps.cc/pps.cc write()) ---------- */
  void write_parameter_sets()
  {
    vps = std::make_shared<video_parameter_set>();
    vps->set_defaults(c.bits > 8 ? Profile_Main10 : Profile_Main, 6, 2);
    vps->layer[0].vps_max_dec_pic_buffering = 6; vps->layer[0].vps_max_num_reorder_pics = 4;

    sps = std::make_shared<seq_parameter_set>();
    sps->set_defaults();
    sps->profile_tier_level_.general.set_defaults(c.bits > 8 ? Profile_Main10 : Profile_Main, 6, 2);
    sps->set_resolution(c.w, c.h);
    sps->bit_depth_luma = sps->bit_depth_chroma = c.bits;
    sps->chroma_format_idc = c.chroma;
    if (rext()) {
      sps->sps_extension_present_flag = 1; sps->sps_range_extension_flag = 1;
      sps->range_extension.transform_skip_rotation_enabled_flag = c.rot;
      sps->range_extension.implicit_rdpcm_enabled_flag = c.irdpcm;
      sps->range_extension.explicit_rdpcm_enabled_flag = c.erdpcm;
      sps->range_extension.intra_smoothing_disabled_flag = c.nosmooth;
      sps->range_extension.high_precision_offsets_enabled_flag = c.hpo;
    }
    sps->sps_max_dec_pic_buffering[0] = 6; sps->sps_max_num_reorder_pics[0] = 4;
    sps->set_CB_log2size_range(c.log2mincb, c.log2ctb);
    sps->set_TB_log2size_range(c.log2mintb, c.log2maxtb);
    sps->max_transform_hierarchy_depth_inter = c.depth_inter;
    sps->max_transform_hierarchy_depth_intra = c.depth_intra;
    sps->amp_enabled_flag = c.amp; sps->sample_adaptive_offset_enabled_flag = c.sao;
    sps->pcm_enabled_flag = c.pcm;
    if (c.pcm) {
      sps->pcm_sample_bit_depth_luma = sps->pcm_sample_bit_depth_chroma = c.pcm_bits ? c.pcm_bits : c.bits;
      sps->log2_min_pcm_luma_coding_block_size = std::max(3, c.log2mincb);
      sps->log2_diff_max_min_pcm_luma_coding_block_size = std::min(5, c.log2ctb) - sps->log2_min_pcm_luma_coding_block_size;
      sps->pcm_loop_filter_disable_flag = c.pcm_lf_off;
    }
    sps->sps_temporal_mvp_enabled_flag = c.tmvp; sps->strong_intra_smoothing_enable_flag = c.strong;
    sps->scaling_list_enable_flag = c.scaling ? 1 : 0; sps->sps_scaling_list_data_present_flag = 0;   /* default lists unless the PPS sends its own */
    if (sps->compute_derived_values() != DE265_OK) die("sps: invalid parameters");

    pps = std::make_shared<pic_parameter_set>();
    pps->set_defaults();
    pps->sign_data_hiding_flag = c.sdh; pps->cabac_init_present_flag = c.cabac_init;
    pps->num_ref_idx_l0_default_active = 1; pps->num_ref_idx_l1_default_active = 1;
    pps->pic_init_qp = c.qp; pps->constrained_intra_pred_flag = c.cip; pps->transform_skip_enabled_flag = c.tskip;
    pps->cu_qp_delta_enabled_flag = c.cuqpd; pps->diff_cu_qp_delta_depth = c.cuqpd ? std::min(c.qg_depth, c.log2ctb - c.log2mincb) : 0;
    pps->pic_cb_qp_offset = c.cb_off; pps->pic_cr_qp_offset = c.cr_off; pps->pps_slice_chroma_qp_offsets_present_flag = c.slice_cqp;
    pps->weighted_pred_flag = c.wp; pps->weighted_bipred_flag = c.wp; pps->transquant_bypass_enable_flag = c.tqbypass;
    pps->pps_loop_filter_across_slices_enabled_flag = c.lf_slices;
    pps->deblocking_filter_control_present_flag = 1; pps->deblocking_filter_override_enabled_flag = c.deblock ? 1 : 0;
    pps->pic_disable_deblocking_filter_flag = c.deblock ? 0 : 1;
    pps->beta_offset = c.deblock ? 2*rng.range(-3,3) : 0; pps->tc_offset = c.deblock ? 2*rng.range(-3,3) : 0;
    pps->lists_modification_present_flag = c.lists_mod; pps->log2_parallel_merge_level = c.par_mrg;
    pps->entropy_coding_sync_enabled_flag = c.wpp;
    pps->dependent_slice_segments_enabled_flag = c.dep ? 1 : 0;
    if (rext()) {
      pps->pps_extension_flag = 1; pps->pps_range_extension_flag = 1;
      pps->range_extension.log2_max_transform_skip_block_size = c.tskip_log2;
      pps->range_extension.cross_component_prediction_enabled_flag = c.xcc;
    }
    W4 = (c.w+3)/4; H4 = (c.h+3)/4;
    ctbW = (c.w + (1<<c.log2ctb) - 1) >> c.log2ctb; ctbH = (c.h + (1<<c.log2ctb) - 1) >> c.log2ctb; nCtb = ctbW*ctbH;
    if (c.tile_cols > 1 || c.tile_rows > 1) {
      if (c.tile_cols > ctbW || c.tile_rows > ctbH || c.tile_cols > DE265_MAX_TILE_COLUMNS || c.tile_rows > DE265_MAX_TILE_ROWS) die("more tiles than CTBs");
      pps->tiles_enabled_flag = 1; pps->num_tile_columns = c.tile_cols; pps->num_tile_rows = c.tile_rows;
      pps->uniform_spacing_flag = c.tile_uniform; pps->loop_filter_across_tiles_enabled_flag = c.lf_tiles;
      if (!c.tile_uniform) {                                        /* random boundaries, every tile at least one CTB */
        for (int d=0; d<2; d++) {
          const int n = d ? c.tile_rows : c.tile_cols, tot = d ? ctbH : ctbW;
          int left = tot;
          for (int i=0;i<n;i++) { const int sz = i==n-1 ? left : rng.range(1, left-(n-1-i)); (d ? pps->rowHeight : pps->colWidth)[i] = sz; left -= sz; }
        }
      }
    }
    pps->set_derived_values(sps.get());

    nal_begin(NAL_UNIT_VPS_NUT); vps->write(&errq, hdr); hdr.add_trailing_bits(); hdr.flush_VLC(); nal_end(-1,false);
    nal_begin(NAL_UNIT_SPS_NUT); sps->write(&errq, hdr);
    if (rext()) {                                                   /* sps.cc:421-435 + sps_range_extension::read (:1254-1264): the reference's write() stops at sps_extension_present_flag */
      hdr.write_bit(1); hdr.write_bit(0); hdr.write_bits(0,6);       /* sps_range_extension_flag, sps_multilayer_extension_flag, sps_extension_6bits */
      const sps_range_extension& r = sps->range_extension;
      hdr.write_bit(r.transform_skip_rotation_enabled_flag); hdr.write_bit(0 /*transform_skip_context*/); hdr.write_bit(r.implicit_rdpcm_enabled_flag);
      hdr.write_bit(r.explicit_rdpcm_enabled_flag); hdr.write_bit(0 /*extended_precision*/); hdr.write_bit(r.intra_smoothing_disabled_flag);
      hdr.write_bit(r.high_precision_offsets_enabled_flag); hdr.write_bit(0 /*persistent_rice*/); hdr.write_bit(0 /*cabac_bypass_alignment*/);
    }
    hdr.add_trailing_bits(); hdr.flush_VLC(); nal_end(-1,false);
    nal_begin(NAL_UNIT_PPS_NUT);
    if (c.scaling == 2) write_pps_with_scaling_lists(); else pps->write(&errq, hdr, sps.get());
    if (rext()) {                                                   /* pps.cc:503-512 + pps_range_extension::read (:47-143) */
      hdr.write_bit(1); hdr.write_bit(0); hdr.write_bits(0,6);       /* pps_range_extension_flag, pps_multilayer_extension_flag, pps_extension_6bits */
      if (pps->transform_skip_enabled_flag) hdr.write_uvlc(c.tskip_log2 - 2);
      hdr.write_bit(c.xcc); hdr.write_bit(0 /*chroma_qp_offset_list_enabled_flag*/);
      hdr.write_uvlc(0); hdr.write_uvlc(0);                          /* log2_sao_offset_scale_luma / _chroma */
    }
    hdr.add_trailing_bits(); hdr.flush_VLC(); nal_end(-1,false);

  }

  /* pps.cc:276 pic_parameter_set::read, field by field (the reference's write() cannot emit scaling_list_data: sps.cc:979 is a stub),
     with pic_scaling_list_data_present_flag = 1 and random lists: explicit coefficients, copies of earlier lists and default lists */
  void write_pps_with_scaling_lists()
  {
    const pic_parameter_set& p = *pps;
    hdr.write_uvlc(p.pic_parameter_set_id); hdr.write_uvlc(p.seq_parameter_set_id);
    hdr.write_bit(p.dependent_slice_segments_enabled_flag); hdr.write_bit(p.output_flag_present_flag); hdr.write_bits(p.num_extra_slice_header_bits,3);
    hdr.write_bit(p.sign_data_hiding_flag); hdr.write_bit(p.cabac_init_present_flag);
    hdr.write_uvlc(p.num_ref_idx_l0_default_active-1); hdr.write_uvlc(p.num_ref_idx_l1_default_active-1);
    hdr.write_svlc(p.pic_init_qp-26); hdr.write_bit(p.constrained_intra_pred_flag); hdr.write_bit(p.transform_skip_enabled_flag);
    hdr.write_bit(p.cu_qp_delta_enabled_flag); if (p.cu_qp_delta_enabled_flag) hdr.write_uvlc(p.diff_cu_qp_delta_depth);
    hdr.write_svlc(p.pic_cb_qp_offset); hdr.write_svlc(p.pic_cr_qp_offset); hdr.write_bit(p.pps_slice_chroma_qp_offsets_present_flag);
    hdr.write_bit(p.weighted_pred_flag); hdr.write_bit(p.weighted_bipred_flag); hdr.write_bit(p.transquant_bypass_enable_flag);
    hdr.write_bit(p.tiles_enabled_flag); hdr.write_bit(p.entropy_coding_sync_enabled_flag);
    if (p.tiles_enabled_flag) {
      hdr.write_uvlc(p.num_tile_columns-1); hdr.write_uvlc(p.num_tile_rows-1); hdr.write_bit(p.uniform_spacing_flag);
      if (!p.uniform_spacing_flag) { for (int i=0;i<p.num_tile_columns-1;i++) hdr.write_uvlc(p.colWidth[i]-1); for (int i=0;i<p.num_tile_rows-1;i++) hdr.write_uvlc(p.rowHeight[i]-1); }
      hdr.write_bit(p.loop_filter_across_tiles_enabled_flag);
    }
    hdr.write_bit(p.pps_loop_filter_across_slices_enabled_flag);
    hdr.write_bit(p.deblocking_filter_control_present_flag);
    if (p.deblocking_filter_control_present_flag) {
      hdr.write_bit(p.deblocking_filter_override_enabled_flag); hdr.write_bit(p.pic_disable_deblocking_filter_flag);
      if (!p.pic_disable_deblocking_filter_flag) { hdr.write_svlc(p.beta_offset/2); hdr.write_svlc(p.tc_offset/2); }
    }
    hdr.write_bit(1);                                               /* pic_scaling_list_data_present_flag */
    for (int sizeId=0;sizeId<4;sizeId++)                            /* sps.cc:770 read_scaling_list */
      for (int matrixId=0; matrixId<(sizeId==3 ? 2 : 6); matrixId++) {
        const bool explicit_list = rng.pct(60);
        hdr.write_bit(explicit_list);                               /* scaling_list_pred_mode_flag */
        if (!explicit_list) { hdr.write_uvlc(rng.below(matrixId+1)); continue; }   /* pred_matrix_id_delta: 0 = default list, else copy an earlier one */
        int next = 8;
        if (sizeId > 1) { const int dc = rng.range(1,200); hdr.write_svlc(dc-8); next = dc; }
        for (int i=0;i<(sizeId==0 ? 16 : 64);i++) {
          int v = std::min(255, std::max(1, next + rng.range(-12,14)));       /* the list entries must stay in 1..255 */
          if (rng.pct(3)) v = rng.range(1,255);
          int d = v - next; if (d > 127) d -= 256; if (d < -128) d += 256;
          hdr.write_svlc(d); next = v;
        }
      }
    hdr.write_bit(p.lists_modification_present_flag); hdr.write_uvlc(p.log2_parallel_merge_level-2);
    hdr.write_bit(p.slice_segment_header_extension_present_flag); hdr.write_bit(p.pps_extension_flag);
  }

  /* ---------- block state ---------- */
  void fill(std::vector<uint8_t>& a, int x0, int y0, int w, int h, int v) {
    for (int y=y0/4; y<std::min(H4,(y0+h+3)/4); y++) for (int x=x0/4; x<std::min(W4,(x0+w+3)/4); x++) a[x+y*W4] = (uint8_t)v;
  }
  int at(const std::vector<uint8_t>& a, int x, int y) const { return a[(x>>2)+(y>>2)*W4]; }
  /* slice.cc:2873 check_CTB_available (one tile) */
  bool available(int xC, int yC, int xN, int yN) const {
    if (xN<0 || yN<0 || xN>=c.w || yN>=c.h) return false;
    const int cur = (xC>>c.log2ctb) + (yC>>c.log2ctb)*ctbW, nb = (xN>>c.log2ctb) + (yN>>c.log2ctb)*ctbW;
    return ctb_slice[nb] == ctb_slice[cur] && pps->TileIdRS[nb] == pps->TileIdRS[cur];
  }

  /* ---------- slice header (slice.cc:352 slice_segment_header::read, refpic.cc:85, slice.cc:215 read_pred_weight_table) ---------- */
  struct PicState { std::vector<int> dpb; };                 /* POCs of decoded pictures still held */
  PicState dpbs;

  void write_slice_header(const PicPlan& p, const std::vector<int>& neg, const std::vector<int>& pos,
                          const std::vector<bool>& neg_used, const std::vector<bool>& pos_used,
                          int slice_idx, int addr, int slice_type, int nal_type)
  {
    const int n_curr = (int)std::count(neg_used.begin(),neg_used.end(),true) + (int)std::count(pos_used.begin(),pos_used.end(),true);
    hdr.write_bit(slice_idx==0);                                    /* first_slice_segment_in_pic_flag */
    if (nal_type >= 16 && nal_type <= 23) hdr.write_bit(0);         /* no_output_of_prior_pics_flag */
    hdr.write_uvlc(0);                                              /* slice_pic_parameter_set_id */
    if (slice_idx) { if (c.dep) hdr.write_bit(0); int nb = 0; while ((1<<nb) < nCtb) nb++; hdr.write_bits(addr, nb); }   /* dependent_slice_segment_flag = 0 */
    hdr.write_uvlc(slice_type);
    bool tmvp = false;
    if (!p.idr) {
      hdr.write_bits(p.poc & 255, 8);                               /* slice_pic_order_cnt_lsb, log2_max_poc_lsb = 8 */
      hdr.write_bit(0);                                             /* short_term_ref_pic_set_sps_flag: explicit set, no prediction (idx 0) */
      hdr.write_uvlc((int)neg.size()); hdr.write_uvlc((int)pos.size());
      int last = p.poc;
      for (size_t i=0;i<neg.size();i++) { hdr.write_uvlc(last-neg[i]-1); hdr.write_bit(neg_used[i]); last = neg[i]; }
      last = p.poc;
      for (size_t i=0;i<pos.size();i++) { hdr.write_uvlc(pos[i]-last-1); hdr.write_bit(pos_used[i]); last = pos[i]; }
      if (c.tmvp) { tmvp = pic_tmvp; hdr.write_bit(tmvp); }
    }
    S.sao_luma = S.sao_chroma = false;
    if (c.sao) { S.sao_luma = rng.pct(80); S.sao_chroma = rng.pct(70); hdr.write_bit(S.sao_luma); hdr.write_bit(S.sao_chroma); }
    S.n_l0 = S.n_l1 = 0; S.max_merge = 5; S.mvd_l1_zero = 0;
    int cabac_init_flag = 0;
    if (slice_type != SLICE_TYPE_I) {
      S.n_l0 = rng.range(1, std::min(4, n_curr+1)); S.n_l1 = slice_type==SLICE_TYPE_B ? rng.range(1, std::min(3, n_curr+1)) : 0;
      if (tmvp) { S.n_l0 = std::max(S.n_l0, pic_col_idx+1); if (slice_type==SLICE_TYPE_B) S.n_l1 = std::max(S.n_l1, pic_col_idx+1); }
      const bool ovr = S.n_l0 != 1 || (slice_type==SLICE_TYPE_B && S.n_l1 != 1) || rng.pct(20);
      hdr.write_bit(ovr);
      if (ovr) { hdr.write_uvlc(S.n_l0-1); if (slice_type==SLICE_TYPE_B) hdr.write_uvlc(S.n_l1-1); }
      if (c.lists_mod && n_curr > 1) {
        int nb = 0; while ((1<<nb) < n_curr) nb++;
        for (int l=0; l<(slice_type==SLICE_TYPE_B ? 2 : 1); l++) {
          const bool m = rng.pct(50); hdr.write_bit(m);
          if (m) for (int i=0;i<(l?S.n_l1:S.n_l0);i++) hdr.write_bits(rng.below(n_curr), nb);
        }
      }
      if (slice_type==SLICE_TYPE_B) { S.mvd_l1_zero = rng.pct(25); hdr.write_bit(S.mvd_l1_zero); }
      if (c.cabac_init) { cabac_init_flag = rng.pct(50); hdr.write_bit(cabac_init_flag); }
      if (tmvp) {
        /* one collocated picture per picture (7.4.7.1): with lists_mod the same POC cannot be guaranteed -> idx 0 of L0 only then */
        bool from_l0 = true;
        if (slice_type==SLICE_TYPE_B) { from_l0 = pic_col_from_l0; hdr.write_bit(from_l0); }
        if ((from_l0 && S.n_l0>1) || (!from_l0 && S.n_l1>1)) hdr.write_uvlc(pic_col_idx);
      }
      if (c.wp) write_pred_weight_table(slice_type);
      S.max_merge = c.merge_cand > 0 ? c.merge_cand : rng.range(1,5);
      hdr.write_uvlc(5 - S.max_merge);
    }
    const int qpd = rng.range(-4,4);
    S.qp = c.qp + qpd; hdr.write_svlc(qpd);
    if (c.slice_cqp) { hdr.write_svlc(rng.range(-3,3)); hdr.write_svlc(rng.range(-3,3)); }
    bool dbk_off = !c.deblock;
    if (c.deblock) {
      const bool ovr = rng.pct(50); hdr.write_bit(ovr);
      if (ovr) { dbk_off = rng.pct(20); hdr.write_bit(dbk_off); if (!dbk_off) { hdr.write_svlc(rng.range(-4,4)); hdr.write_svlc(rng.range(-4,4)); } }
    }
    if (c.lf_slices && (S.sao_luma || S.sao_chroma || !dbk_off)) hdr.write_bit(rng.pct(60));   /* slice_loop_filter_across_slices_enabled_flag */
    /* entry points and byte_alignment() follow once the data is coded: finish_slice_header() */
    S.type = slice_type; S.addr = addr;
    S.init_type = slice_type==SLICE_TYPE_I ? 0 : slice_type==SLICE_TYPE_P ? (cabac_init_flag ? 2 : 1) : (cabac_init_flag ? 1 : 2);
    models.init(S.init_type, S.qp);
    cab.reset();
    cab.set_context_models(&models);
    cab.init_CABAC();
  }

  /* a dependent slice segment: everything but the address is inherited from the slice's first segment (slice.cc:372-395) */
  void write_dependent_segment_header(int addr, int nal_type)
  {
    hdr.write_bit(0);                                               /* first_slice_segment_in_pic_flag */
    if (nal_type >= 16 && nal_type <= 23) hdr.write_bit(0);         /* no_output_of_prior_pics_flag */
    hdr.write_uvlc(0);                                              /* slice_pic_parameter_set_id */
    hdr.write_bit(1);                                               /* dependent_slice_segment_flag */
    int nb = 0; while ((1<<nb) < nCtb) nb++;
    hdr.write_bits(addr, nb);
    cab.reset();
    cab.set_context_models(&models);
    cab.init_CABAC();
  }

  /* slice.cc:660-700: entry_point_offset_minus1[] = bytes of every substream but the last, emulation prevention bytes included */
  void finish_slice_header(const std::vector<int>& sub_end)
  {
    if (c.wpp || pps->tiles_enabled_flag) {
      const int n = (int)sub_end.size()-1;
      hdr.write_uvlc(n);
      if (n > 0) {
        int mx = 0;
        for (int i=0;i<n;i++) mx = std::max(mx, sub_end[i]-(i?sub_end[i-1]:0)-1);
        int len = 1; while ((mx >> len) != 0) len++;
        hdr.write_uvlc(len-1);
        for (int i=0;i<n;i++) {
          const uint32_t v = (uint32_t)(sub_end[i]-(i?sub_end[i-1]:0)-1);
          if (len > 16) { hdr.write_bits(v >> 16, len-16); hdr.write_bits(v & 0xFFFF, 16); } else hdr.write_bits(v, len);
        }
      }
    }
    hdr.add_trailing_bits(); hdr.flush_VLC();                       /* byte_alignment() */
  }

  void write_pred_weight_table(int slice_type)
  {
    const int ld = rng.range(0,7); hdr.write_uvlc(ld);
    const int dc = rng.range(std::max(-ld,-2), std::min(7-ld,2)); hdr.write_svlc(dc);       /* delta_chroma_log2_weight_denom */
    for (int l=0; l<(slice_type==SLICE_TYPE_B ? 2 : 1); l++) {
      const int n = l ? S.n_l1 : S.n_l0;
      std::vector<int> lf(n), cf(n);
      for (int i=0;i<n;i++) { lf[i] = rng.pct(60); hdr.write_bit(lf[i]); }
      for (int i=0;i<n;i++) { cf[i] = rng.pct(50); hdr.write_bit(cf[i]); }
      for (int i=0;i<n;i++) {
        if (lf[i]) { hdr.write_svlc(rng.range(-20,20)); hdr.write_svlc(rng.range(-40,40)); }
        if (cf[i]) for (int j=0;j<2;j++) { hdr.write_svlc(rng.range(-20,20)); hdr.write_svlc(rng.range(-100,100)); }
      }
    }
  }

  /* ---------- SAO (slice.cc:2695 read_sao) ---------- */
  void code_sao(int xCtb, int yCtb)
  {
    const int addr = xCtb + yCtb*ctbW;
    bool left = false, up = false;
    if (xCtb>0 && addr > S.addr && pps->TileIdRS[addr] == pps->TileIdRS[addr-1]) { left = rng.pct(20); cab.write_CABAC_bit(CONTEXT_MODEL_SAO_MERGE_FLAG, left); }
    if (yCtb>0 && !left && addr-ctbW >= S.addr && pps->TileIdRS[addr] == pps->TileIdRS[addr-ctbW]) { up = rng.pct(20); cab.write_CABAC_bit(CONTEXT_MODEL_SAO_MERGE_FLAG, up); }
    if (left || up) return;
    int type_c = 0;
    for (int cIdx=0;cIdx<3;cIdx++) {
      if (!((cIdx==0 && S.sao_luma) || (cIdx>0 && S.sao_chroma))) continue;
      int type;
      if (cIdx<2) {
        type = rng.pct(30) ? 0 : rng.range(1,2);
        cab.write_CABAC_bit(CONTEXT_MODEL_SAO_TYPE_IDX, type!=0);
        if (type) cab.write_CABAC_bypass(type==2);
        if (cIdx==1) type_c = type;
      } else type = type_c;
      if (!type) continue;
      const int cMax = (1 << (std::min(c.bits,10)-5)) - 1;
      int off[4];
      for (int i=0;i<4;i++) { off[i] = rng.pct(25) ? 0 : rng.range(0,cMax); cab.write_CABAC_TU_bypass(off[i], cMax); }
      if (type==1) {
        for (int i=0;i<4;i++) if (off[i]) cab.write_CABAC_bypass(rng.below(2));
        cab.write_CABAC_FL_bypass(rng.below(32), 5);
      } else if (cIdx<2) cab.write_CABAC_FL_bypass(rng.below(4), 2);
    }
  }

  /* ---------- coding quadtree (slice.cc:4582) ---------- */
  void code_quadtree(int x0, int y0, int log2, int depth)
  {
    bool split;
    if (x0+(1<<log2) <= c.w && y0+(1<<log2) <= c.h && log2 > c.log2mincb) {
      split = rng.pct(log2 >= 6 ? 75 : log2 == 5 ? 55 : 40);
      const int condL = available(x0,y0,x0-1,y0) && at(ct_depth,x0-1,y0) > depth;
      const int condA = available(x0,y0,x0,y0-1) && at(ct_depth,x0,y0-1) > depth;
      cab.write_CABAC_bit(CONTEXT_MODEL_SPLIT_CU_FLAG + condL + condA, split);
    } else split = log2 > c.log2mincb;
    if (c.cuqpd && log2 >= c.log2ctb - pps->diff_cu_qp_delta_depth) qpd_coded = false;
    if (split) {
      const int x1 = x0 + (1<<(log2-1)), y1 = y0 + (1<<(log2-1));
      code_quadtree(x0,y0,log2-1,depth+1);
      if (x1<c.w) code_quadtree(x1,y0,log2-1,depth+1);
      if (y1<c.h) code_quadtree(x0,y1,log2-1,depth+1);
      if (x1<c.w && y1<c.h) code_quadtree(x1,y1,log2-1,depth+1);
    } else {
      fill(ct_depth, x0,y0, 1<<log2,1<<log2, depth);
      code_cu(x0,y0,log2,depth);
    }
  }

  /* ---------- coding unit (slice.cc:4245) ---------- */
  bool cu_bypass = false;

  void code_cu(int x0, int y0, int log2, int depth)
  {
    const int n = 1<<log2;
    cu_bypass = false;
    if (c.tqbypass) { cu_bypass = rng.pct(12); cab.write_CABAC_bit(CONTEXT_MODEL_CU_TRANSQUANT_BYPASS_FLAG, cu_bypass); }
    bool skip = false;
    if (S.type != SLICE_TYPE_I) {
      skip = rng.pct(22);
      const int condL = available(x0,y0,x0-1,y0) && at(skipf,x0-1,y0);
      const int condA = available(x0,y0,x0,y0-1) && at(skipf,x0,y0-1);
      cab.write_CABAC_bit(CONTEXT_MODEL_CU_SKIP_FLAG + condL + condA, skip);
    }
    fill(skipf, x0,y0,n,n, skip); fill(pcmf, x0,y0,n,n, 0);
    st.n_cus++;
    if (skip) { code_merge_idx(); st.n_pus++; fill(pmode, x0,y0,n,n, PM_SKIP); return; }

    bool intra = true;
    if (S.type != SLICE_TYPE_I) { intra = rng.pct(18); cab.write_CABAC_bit(CONTEXT_MODEL_PRED_MODE_FLAG, intra); }
    fill(pmode, x0,y0,n,n, intra ? PM_INTRA : PM_INTER);

    int part = PART_2Nx2N;
    if (intra) {
      if (log2 == c.log2mincb) { part = rng.pct(45) ? PART_NxN : PART_2Nx2N; cab.write_CABAC_bit(CONTEXT_MODEL_PART_MODE, part==PART_2Nx2N); }
    } else {
      part = choose_inter_part(log2);
      code_inter_part_mode(part, log2);
    }

    bool merge_2Nx2N = false;
    if (intra) {
      bool pcm = false;
      if (part==PART_2Nx2N && c.pcm && log2 >= sps->Log2MinIpcmCbSizeY && log2 <= sps->Log2MaxIpcmCbSizeY) {
        pcm = rng.pct(log2 >= 5 ? 3 : 8);
        cab.write_CABAC_term_bit(pcm);
      }
      if (pcm) { st.n_pcms++; fill(pcmf, x0,y0,n,n, 1); fill(ipm, x0,y0,n,n, 1); code_pcm_samples(log2); return; }
      code_intra_modes(x0,y0,log2,part);
    } else {
      static const int geo[8][2][4] = {        /* per PartMode: PUs as x,y,w,h in quarters of the CB */
        {{0,0,4,4},{0,0,0,0}}, {{0,0,4,2},{0,2,4,2}}, {{0,0,2,4},{2,0,2,4}}, {{0,0,2,2},{2,0,2,2}},
        {{0,0,4,1},{0,1,4,3}}, {{0,0,4,3},{0,3,4,1}}, {{0,0,1,4},{1,0,3,4}}, {{0,0,3,4},{3,0,1,4}} };
      if (part == PART_NxN) {
        for (int k=0;k<4;k++) code_pu(n/2, n/2, depth);
      } else {
        const int npu = part==PART_2Nx2N ? 1 : 2;
        for (int k=0;k<npu;k++) { const bool m = code_pu(geo[part][k][2]*n/4, geo[part][k][3]*n/4, depth); if (part==PART_2Nx2N) merge_2Nx2N = m; }
      }
    }

    bool rqt = true;
    if (!intra && !merge_2Nx2N) { rqt = rng.pct(65); cab.write_CABAC_bit(CONTEXT_MODEL_RQT_ROOT_CBF, rqt); }
    if (rqt) {
      const int intraSplit = intra && part==PART_NxN;
      const int maxDepth = intra ? c.depth_intra + intraSplit : c.depth_inter;
      code_transform_tree(x0,y0,x0,y0,log2,0,0,maxDepth,intraSplit,intra,part,1,1);
    }
  }

  int choose_inter_part(int log2)
  {
    if (rng.pct(45)) return PART_2Nx2N;
    std::vector<int> opt = {PART_2NxN, PART_Nx2N};
    if (log2 > c.log2mincb) { if (c.amp) { opt.push_back(PART_2NxnU); opt.push_back(PART_2NxnD); opt.push_back(PART_nLx2N); opt.push_back(PART_nRx2N); } }
    else if (log2 > 3) opt.push_back(PART_NxN);
    return opt[rng.below((int)opt.size())];
  }

  /* slice.cc:1689 decode_part_mode, inter branch */
  void code_inter_part_mode(int part, int log2)
  {
    cab.write_CABAC_bit(CONTEXT_MODEL_PART_MODE+0, part==PART_2Nx2N);
    if (part==PART_2Nx2N) return;
    const bool horiz = part==PART_2NxN || part==PART_2NxnU || part==PART_2NxnD;
    if (log2 > c.log2mincb) {
      cab.write_CABAC_bit(CONTEXT_MODEL_PART_MODE+1, horiz);
      if (!c.amp) return;
      const bool sym = part==PART_2NxN || part==PART_Nx2N;
      cab.write_CABAC_bit(CONTEXT_MODEL_PART_MODE+3, sym);
      if (!sym) cab.write_CABAC_bypass(part==PART_2NxnD || part==PART_nRx2N);
    } else {
      cab.write_CABAC_bit(CONTEXT_MODEL_PART_MODE+1, part==PART_2NxN);
      if (part==PART_2NxN || log2==3) return;
      cab.write_CABAC_bit(CONTEXT_MODEL_PART_MODE+2, part==PART_Nx2N);
    }
  }

  /* slice.cc:4144-4243: pcm_flag was the terminating bin -> CABAC flush, stop bit, pcm_alignment_zero_bits, raw samples, CABAC restart */
  void code_pcm_samples(int log2)
  {
    cab.flush_CABAC(); cab.write_bit(1); cab.write_bits(0, cab.number_free_bits_in_byte()); cab.flush_VLC();
    const int nb = sps->pcm_sample_bit_depth_luma, n = 1<<log2;
    const int mode = rng.below(3), base = rng.below(1<<nb);
    for (int comp=0;comp<3;comp++) {
      const int w = comp ? n/(c.chroma==3 ? 1 : 2) : n, h = comp ? n/(c.chroma==1 ? 2 : 1) : n;      /* slice.cc:4166-4170: nCS / SubWidthC x nCS / SubHeightC */
      for (int i=0;i<w*h;i++) {
        int v = mode==0 ? rng.below(1<<nb) : mode==1 ? base : std::min((1<<nb)-1, std::max(0, base + rng.range(-3,3)));
        cab.write_bits(v, nb);
      }
    }
    cab.flush_VLC();
    cab.init_CABAC();
  }

  /* slice.cc:4336-4440 + intrapred.cc:62-152 */
  void code_intra_modes(int x0, int y0, int log2, int part)
  {
    const int n = 1<<log2, npu = part==PART_NxN ? 4 : 1, pb = part==PART_NxN ? n/2 : n;
    int prev[4], val[4];
    for (int k=0;k<npu;k++) { prev[k] = rng.pct(55); val[k] = prev[k] ? rng.below(3) : rng.below(32); cab.write_CABAC_bit(CONTEXT_MODEL_PREV_INTRA_LUMA_PRED_FLAG, prev[k]); }
    const bool availA0 = available(x0,y0,x0-1,y0), availB0 = available(x0,y0,x0,y0-1);
    int mode0 = 0;
    for (int k=0;k<npu;k++) {
      if (prev[k]) cab.write_CABAC_TU_bypass(val[k], 2); else cab.write_CABAC_FL_bypass(val[k], 5);
      const int i = (k&1)*pb, j = (k>>1)*pb, x = x0+i, y = y0+j;
      const bool availA = availA0 || i>0, availB = availB0 || j>0;
      int A = 1, B = 1;                                                              /* INTRA_DC */
      if (availA && at(pmode,x-1,y)==PM_INTRA && !at(pcmf,x-1,y)) A = at(ipm,x-1,y);
      if (availB && at(pmode,x,y-1)==PM_INTRA && !at(pcmf,x,y-1) && y-1 >= ((y>>c.log2ctb)<<c.log2ctb)) B = at(ipm,x,y-1);
      int cand[3];
      if (A==B) {
        if (A<2) { cand[0]=0; cand[1]=1; cand[2]=26; }
        else { cand[0]=A; cand[1]=2+((A-2-1+32)%32); cand[2]=2+((A-2+1)%32); }
      } else {
        cand[0]=A; cand[1]=B;
        cand[2] = (A!=0 && B!=0) ? 0 : (A!=1 && B!=1) ? 1 : 26;
      }
      int mode;
      if (prev[k]) mode = cand[val[k]];
      else {
        std::sort(cand, cand+3);
        mode = val[k];
        for (int t=0;t<3;t++) if (mode >= cand[t]) mode++;
      }
      fill(ipm, x,y,pb,pb, mode);
      if (k==0) mode0 = mode;
    }
    /* intra_chroma_pred_mode: one per CU; in 4:4:4 one per PU (slice.cc:4444-4485).  IntraPredModeC of 4:2:2 goes through
       Table 8-3 (slice.cc:4240-4243, :4478-4480) */
    static const int tab[4] = {0,26,10,1};
    static const uint8_t map422[35] = { 0,1,2, 2, 2, 2, 3, 5, 7, 8,10,12,13,15,17,18,19,20, 21,22,23,23,24,24,25,25,26,27,27,28,28,29,29,30,31 };
    for (int k=0; k<(c.chroma==3 ? npu : 1); k++) {
      const int x = c.chroma==3 ? x0 + (k&1)*pb : x0, y = c.chroma==3 ? y0 + (k>>1)*pb : y0, sz = c.chroma==3 ? pb : n;
      const int cpm = rng.below(5);
      cab.write_CABAC_bit(CONTEXT_MODEL_INTRA_CHROMA_PRED_MODE, cpm!=4);
      if (cpm!=4) cab.write_CABAC_FL_bypass(cpm, 2);
      const int lm = at(ipm,x,y);
      int m = cpm==4 ? lm : (tab[cpm]==lm ? 34 : tab[cpm]);
      if (c.chroma==2) m = map422[m];
      fill(ipmc, x,y,sz,sz, m); fill(ipmc4, x,y,sz,sz, cpm==4);
    }
    (void)mode0;
  }

  /* slice.cc:2503 */
  void code_merge_idx()
  {
    if (S.max_merge <= 1) return;
    const int idx = rng.below(S.max_merge);
    cab.write_CABAC_bit(CONTEXT_MODEL_MERGE_IDX, idx!=0);
    if (idx) for (int i=1; i<S.max_merge-1; i++) { const bool more = i < idx; cab.write_CABAC_bypass(more); if (!more) break; }
  }

  /* slice.cc:2574 */
  void code_ref_idx(int n_active)
  {
    const int cMax = n_active-1;
    if (!cMax) return;
    const int idx = rng.below(n_active);
    for (int i=0;i<cMax;i++) {
      const bool bit = i < idx;
      if (i<2) cab.write_CABAC_bit(CONTEXT_MODEL_REF_IDX_LX + i, bit); else cab.write_CABAC_bypass(bit);
      if (!bit) break;
    }
  }

  /* slice.cc:3986 */
  void code_mvd()
  {
    int v[2];
    for (int k=0;k<2;k++) {
      const int r = rng.below(100);
      int a = r<30 ? 0 : r<50 ? 1 : r<85 ? rng.range(2,24) : r<98 ? rng.range(25,160) : (c.big_mv ? rng.range(161,3000) : 2);
      v[k] = rng.below(2) ? -a : a;
    }
    const int a0 = abs(v[0]), a1 = abs(v[1]);
    cab.write_CABAC_bit(CONTEXT_MODEL_ABS_MVD_GREATER01_FLAG+0, a0>0);
    cab.write_CABAC_bit(CONTEXT_MODEL_ABS_MVD_GREATER01_FLAG+0, a1>0);
    if (a0) cab.write_CABAC_bit(CONTEXT_MODEL_ABS_MVD_GREATER01_FLAG+1, a0>1);
    if (a1) cab.write_CABAC_bit(CONTEXT_MODEL_ABS_MVD_GREATER01_FLAG+1, a1>1);
    if (a0) { if (a0>1) cab.write_CABAC_EGk(a0-2,1); cab.write_CABAC_bypass(v[0]<0); }
    if (a1) { if (a1>1) cab.write_CABAC_EGk(a1-2,1); cab.write_CABAC_bypass(v[1]<0); }
  }

  /* slice.cc:4062 read_prediction_unit; returns merge_flag */
  bool code_pu(int w, int h, int ctDepth)
  {
    const bool merge = rng.pct(40);
    st.n_pus++;
    cab.write_CABAC_bit(CONTEXT_MODEL_MERGE_FLAG, merge);
    if (merge) { code_merge_idx(); return true; }
    int idc = 0;                                                           /* 0: L0, 1: L1, 2: BI */
    if (S.type == SLICE_TYPE_B) {
      if (w+h==12) { idc = rng.below(2); cab.write_CABAC_bit(CONTEXT_MODEL_INTER_PRED_IDC+4, idc); }
      else {
        idc = rng.pct(40) ? 2 : rng.below(2);
        cab.write_CABAC_bit(CONTEXT_MODEL_INTER_PRED_IDC+ctDepth, idc==2);
        if (idc!=2) cab.write_CABAC_bit(CONTEXT_MODEL_INTER_PRED_IDC+4, idc);
      }
    }
    if (idc != 1) { code_ref_idx(S.n_l0); code_mvd(); cab.write_CABAC_bit(CONTEXT_MODEL_MVP_LX_FLAG, rng.below(2)); }
    if (idc != 0) {
      code_ref_idx(S.n_l1);
      if (!(S.mvd_l1_zero && idc==2)) code_mvd();
      cab.write_CABAC_bit(CONTEXT_MODEL_MVP_LX_FLAG, rng.below(2));
    }
    return false;
  }

  /* ---------- transform tree (slice.cc:3821) and unit (slice.cc:3549) ---------- */
  void code_transform_tree(int x0, int y0, int xBase, int yBase, int log2, int depth, int blkIdx, int maxDepth, int intraSplit,
                           bool intra, int part, int parent_cb, int parent_cr)
  {
    bool split;
    if (log2 <= c.log2maxtb && log2 > c.log2mintb && depth < maxDepth && !(intraSplit && depth==0)) {
      split = rng.pct(log2 >= 5 ? 60 : 40);
      cab.write_CABAC_bit(CONTEXT_MODEL_SPLIT_TRANSFORM_FLAG + 5-log2, split);
    } else {
      const bool interSplit = c.depth_inter==0 && depth==0 && !intra && part != PART_2Nx2N;
      split = log2 > c.log2maxtb || (intraSplit && depth==0) || interSplit;
    }
    /* cbf_cb / cbf_cr (slice.cc:3885-3906): bit 0 the chroma TU (4:2:2: the upper one), bit 1 the lower one of 4:2:2 */
    int cbf_cb = -1, cbf_cr = -1;
    if (log2 > 2 || c.chroma == 3) {
      const bool second = c.chroma == 2 && (!split || log2 == 3);
      if (parent_cb) {
        cbf_cb = rng.pct(c.dens*7/10); cab.write_CABAC_bit(CONTEXT_MODEL_CBF_CHROMA + depth, cbf_cb);
        if (second) { const int b = rng.pct(c.dens*7/10); cab.write_CABAC_bit(CONTEXT_MODEL_CBF_CHROMA + depth, b); cbf_cb |= b<<1; }
      }
      if (parent_cr) {
        cbf_cr = rng.pct(c.dens*7/10); cab.write_CABAC_bit(CONTEXT_MODEL_CBF_CHROMA + depth, cbf_cr);
        if (second) { const int b = rng.pct(c.dens*7/10); cab.write_CABAC_bit(CONTEXT_MODEL_CBF_CHROMA + depth, b); cbf_cr |= b<<1; }
      }
    }
    if (cbf_cb < 0) cbf_cb = (depth>0 && log2==2) ? parent_cb : 0;
    if (cbf_cr < 0) cbf_cr = (depth>0 && log2==2) ? parent_cr : 0;
    if (split) {
      const int x1 = x0 + (1<<(log2-1)), y1 = y0 + (1<<(log2-1));
      code_transform_tree(x0,y0,x0,y0,log2-1,depth+1,0,maxDepth,intraSplit,intra,part,cbf_cb,cbf_cr);
      code_transform_tree(x1,y0,x0,y0,log2-1,depth+1,1,maxDepth,intraSplit,intra,part,cbf_cb,cbf_cr);
      code_transform_tree(x0,y1,x0,y0,log2-1,depth+1,2,maxDepth,intraSplit,intra,part,cbf_cb,cbf_cr);
      code_transform_tree(x1,y1,x0,y0,log2-1,depth+1,3,maxDepth,intraSplit,intra,part,cbf_cb,cbf_cr);
      return;
    }
    int cbf_luma = 1;
    if (intra || depth != 0 || cbf_cb || cbf_cr) { cbf_luma = rng.pct(c.dens); cab.write_CABAC_bit(CONTEXT_MODEL_CBF_LUMA + (depth==0), cbf_luma); }

    if (cbf_luma || cbf_cb || cbf_cr) {
      if (c.cuqpd && !qpd_coded) { code_cu_qp_delta(); qpd_coded = true; }
    }
    /* transform unit (slice.cc:3549-3800): luma, then per chroma component [cross_comp_pred] + one or (4:2:2) two blocks */
    if (cbf_luma) code_residual(log2, 0, intra ? scan_idx(log2, at(ipm,x0,y0), 0) : 0, intra, at(ipm,x0,y0));
    const int cm = at(ipmc,x0,y0);
    if (log2 > 2 || c.chroma == 3) {
      const int lc = c.chroma == 3 ? log2 : log2-1;
      const bool do_xcc = c.xcc && cbf_luma && (!intra || at(ipmc4,x0,y0));
      for (int cIdx=1; cIdx<=2; cIdx++) {
        const int cbf = cIdx==1 ? cbf_cb : cbf_cr;
        if (do_xcc) code_cross_comp_pred(cIdx-1);
        if (cbf & 1) code_residual(lc, cIdx, intra ? scan_idx(lc, cm, cIdx) : 0, intra, cm);
        if (c.chroma == 2 && (cbf & 2)) code_residual(lc, cIdx, intra ? scan_idx(lc, cm, cIdx) : 0, intra, cm);
      }
    } else if (blkIdx == 3) {
      const int cmb = at(ipmc,xBase,yBase);
      for (int cIdx=1; cIdx<=2; cIdx++) {
        const int cbf = cIdx==1 ? cbf_cb : cbf_cr;
        if (cbf & 1) code_residual(2, cIdx, intra ? scan_idx(2, cmb, cIdx) : 0, intra, cmb);
        if (cbf & 2) code_residual(2, cIdx, intra ? scan_idx(2, cmb, cIdx) : 0, intra, cmb);
      }
    }
  }

  /* slice.cc:3490-3541 read_cross_comp_pred: log2_res_scale_abs_plus1 (TU, cMax 4, one context per bin and component), sign */
  void code_cross_comp_pred(int cIdxMinus1)
  {
    const int v = rng.pct(45) ? 0 : rng.range(1,4);
    for (int b=0;b<v;b++) cab.write_CABAC_bit(CONTEXT_MODEL_LOG2_RES_SCALE_ABS_PLUS1 + 4*cIdxMinus1 + b, 1);
    if (v<4) cab.write_CABAC_bit(CONTEXT_MODEL_LOG2_RES_SCALE_ABS_PLUS1 + 4*cIdxMinus1 + v, 0);
    if (v) cab.write_CABAC_bit(CONTEXT_MODEL_RES_SCALE_SIGN_FLAG + cIdxMinus1, rng.below(2));
  }

  /* intrapred.cc:279 get_intra_scan_idx */
  int scan_idx(int log2, int mode, int cIdx) const
  {
    if (log2==2 || (log2==3 && (cIdx==0 || c.chroma==3))) { if (mode>=6 && mode<=14) return 2; if (mode>=22 && mode<=30) return 1; }
    return 0;
  }

  /* slice.cc:1884 */
  void code_cu_qp_delta()
  {
    const int r = rng.below(100);
    const int a = r<55 ? 0 : r<90 ? rng.range(1,3) : rng.range(4,9);
    cab.write_CABAC_bit(CONTEXT_MODEL_CU_QP_DELTA_ABS+0, a>0);
    if (!a) return;
    for (int i=1;i<std::min(a,5);i++) cab.write_CABAC_bit(CONTEXT_MODEL_CU_QP_DELTA_ABS+1, 1);
    if (a<5) cab.write_CABAC_bit(CONTEXT_MODEL_CU_QP_DELTA_ABS+1, 0); else cab.write_CABAC_EGk(a-5, 0);
    cab.write_CABAC_bypass(rng.below(2));
  }

  /* ---------- residual_coding (slice.cc:2905-3420 and the context helpers at 1861-2490) ---------- */
  void random_block(int16_t* co, int n, int cIdx)
  {
    memset(co, 0, sizeof(int16_t)*n*n);
    const int r = rng.below(100);
    int k = r<30 ? 1 : r<70 ? rng.range(2,5) : r<93 ? rng.range(6, std::min(n*n, 24)) : rng.range(1, std::min(n*n, n*n/2+1));
    const bool dc_only = k==1 && rng.pct(60);
    for (int i=0;i<k;i++) {
      int x, y;
      if (dc_only) x = y = 0;
      else if (rng.pct(70)) { x = std::min(n-1, rng.below(n)*rng.below(n)/n); y = std::min(n-1, rng.below(n)*rng.below(n)/n); }
      else { x = rng.below(n); y = rng.below(n); }
      const int m = rng.below(100);
      int a = m<50 ? 1 : m<72 ? 2 : m<84 ? 3 : m<97 ? rng.range(4, std::max(4,c.max_level)) : rng.range(c.max_level, 40*c.max_level);
      if (cIdx && a > 3) a = 1 + a/2;
      co[x+y*n] = (int16_t)(rng.below(2) ? -a : a);
    }
    if (!dc_only && rng.pct(4)) co[n*n-1] = (int16_t)(rng.below(2) ? -1 : 1);      /* last position of the block */
  }

  void code_last_prefix(int prefix, int log2, int cIdx, int model)
  {
    const int cMax = (log2<<1)-1;
    int ctxOffset, ctxShift;
    if (cIdx==0) { ctxOffset = 3*(log2-2) + ((log2-1)>>2); ctxShift = (log2+1)>>2; }
    else { ctxOffset = 15; ctxShift = log2-2; }
    for (int b=0;b<prefix;b++) cab.write_CABAC_bit(model + ctxOffset + (b>>ctxShift), 1);
    if (prefix<cMax) cab.write_CABAC_bit(model + ctxOffset + (prefix>>ctxShift), 0);
  }

  static void split_last(int pos, int* prefix, int* suffix, int* nbits)
  {
    if (pos<4) { *prefix = pos; *suffix = 0; *nbits = 0; return; }
    for (int p=4;;p++) {
      const int nb = (p>>1)-1, base = (2+(p&1))<<nb;
      if (pos < base + (1<<nb)) { *prefix = p; *suffix = pos-base; *nbits = nb; return; }
    }
  }

  void code_abs_remaining(int v, int k)
  {
    if ((v>>k) <= 3) { for (int i=0;i<(v>>k);i++) cab.write_CABAC_bypass(1); cab.write_CABAC_bypass(0); cab.write_CABAC_FL_bypass(v & ((1<<k)-1), k); return; }
    for (int p=4;;p++) {
      const int base = ((1<<(p-3))+2)<<k, nb = p-3+k;
      if (v < base + (1<<nb)) { for (int i=0;i<p;i++) cab.write_CABAC_bypass(1); cab.write_CABAC_bypass(0); cab.write_CABAC_FL_bypass(v-base, nb); return; }
    }
  }

  void code_residual(int log2, int cIdx, int scanIdx, bool intra, int predModeIntra)
  {
    const int n = 1<<log2;
    int16_t co[32*32];
    random_block(co, n, cIdx);
    st.n_resid++;
    for (int i=0;i<n*n;i++) if (co[i]) { st.n_coeffs++; st.abs_sum += abs(co[i]); }
    bool tskip = false;
    if (c.tskip && !cu_bypass && log2 <= c.tskip_log2) { tskip = rng.pct(35); cab.write_CABAC_bit(CONTEXT_MODEL_TRANSFORM_SKIP_FLAG + (cIdx?1:0), tskip); }
    bool explicit_rdpcm = false;                                    /* slice.cc:2937-2952 */
    if (!intra && c.erdpcm && (tskip || cu_bypass)) {
      explicit_rdpcm = rng.pct(50);
      cab.write_CABAC_bit(CONTEXT_MODEL_RDPCM_FLAG + (cIdx?1:0), explicit_rdpcm);
      if (explicit_rdpcm) cab.write_CABAC_bit(CONTEXT_MODEL_RDPCM_DIR + (cIdx?1:0), rng.below(2));
    }
    /* sign data hiding is off where the residual goes through RDPCM or the CU bypasses the transform (slice.cc:3294-3307) */
    const bool no_hiding = cu_bypass || explicit_rdpcm || (intra && c.irdpcm && tskip && (predModeIntra == 10 || predModeIntra == 26));

    const position* subScan = get_scan_order(log2-2, scanIdx);
    const position* posScan = get_scan_order(2, scanIdx);
    const int nSub = 1<<(2*(log2-2));
    int lastSub = -1, lastPos = -1;
    for (int i=nSub-1;i>=0 && lastSub<0;i--)
      for (int p=15;p>=0;p--) {
        const int x = (subScan[i].x<<2)+posScan[p].x, y = (subScan[i].y<<2)+posScan[p].y;
        if (co[x+y*n]) { lastSub = i; lastPos = p; break; }
      }
    int lx = (subScan[lastSub].x<<2)+posScan[lastPos].x, ly = (subScan[lastSub].y<<2)+posScan[lastPos].y;
    if (scanIdx==2) std::swap(lx,ly);
    int px,sx,nx, py,sy,ny;
    split_last(lx,&px,&sx,&nx); split_last(ly,&py,&sy,&ny);
    code_last_prefix(px, log2, cIdx, CONTEXT_MODEL_LAST_SIGNIFICANT_COEFFICIENT_X_PREFIX);
    code_last_prefix(py, log2, cIdx, CONTEXT_MODEL_LAST_SIGNIFICANT_COEFFICIENT_Y_PREFIX);
    if (px>3) cab.write_CABAC_FL_bypass(sx,nx);
    if (py>3) cab.write_CABAC_FL_bypass(sy,ny);

    const int sbW = 1<<(log2-2);
    uint8_t nbr[64]; memset(nbr,0,sizeof(nbr));
    int c1 = 1;
    for (int i=lastSub;i>=0;i--) {
      const position Sb = subScan[i];
      const int bx = Sb.x<<2, by = Sb.y<<2;
      bool has = false;
      for (int p=0;p<16;p++) if (co[bx+posScan[p].x + (by+posScan[p].y)*n]) has = true;
      bool coded, inferDc = false;
      if (i<lastSub && i>0) {
        const int nb = nbr[Sb.x+Sb.y*sbW];
        cab.write_CABAC_bit(CONTEXT_MODEL_CODED_SUB_BLOCK_FLAG + ((nb&1)|(nb>>1)) + (cIdx?2:0), has);
        coded = has; inferDc = true;
      } else coded = true;
      if (!coded) continue;
      if (Sb.x>0) nbr[Sb.x-1+Sb.y*sbW] |= 1;
      if (Sb.y>0) nbr[Sb.x+(Sb.y-1)*sbW] |= 2;
      const int prevCsbf = nbr[Sb.x+Sb.y*sbW];

      int lev[16], nc = 0, firstP = 0, lastP = 0;                   /* levels in coding order; scan position of the first / last one */
      const int start = i==lastSub ? lastPos-1 : 15;
      if (i==lastSub) { lev[nc++] = co[bx+posScan[lastPos].x + (by+posScan[lastPos].y)*n]; firstP = lastP = lastPos; }
      for (int p=start;p>0;p--) {
        const int xC = bx+posScan[p].x, yC = by+posScan[p].y, v = co[xC+yC*n];
        cab.write_CABAC_bit(CONTEXT_MODEL_SIGNIFICANT_COEFF_FLAG + sig_ctx(xC,yC,log2,cIdx,scanIdx,prevCsbf), v!=0);
        if (v) { if (!nc) firstP = p; lastP = p; lev[nc++] = v; inferDc = false; }
      }
      if (start>=0) {
        const int v = co[bx+by*n];
        if (!inferDc) { cab.write_CABAC_bit(CONTEXT_MODEL_SIGNIFICANT_COEFF_FLAG + sig_ctx(bx,by,log2,cIdx,scanIdx,prevCsbf), v!=0); if (v) { if (!nc) firstP = 0; lastP = 0; lev[nc++] = v; } }
        else { firstP = lastP = 0; lev[nc++] = v; }                  /* inferred significant: the only coefficient of the sub-block */
      }
      if (!nc) continue;

      int ctxSet = (i==0 || cIdx>0) ? 0 : 2;
      if (c1==0) ctxSet++;
      c1 = 1;
      int g1ctx = 1, firstG1 = -1, lastFlag = 0;
      const int n8 = std::min(8,nc);
      for (int k=0;k<n8;k++) {
        if (k>0 && g1ctx>0) { if (lastFlag) g1ctx = 0; else g1ctx++; }
        const int flag = abs(lev[k])>1;
        cab.write_CABAC_bit(CONTEXT_MODEL_COEFF_ABS_LEVEL_GREATER1_FLAG + ctxSet*4 + std::min(3,g1ctx) + (cIdx?16:0), flag);
        lastFlag = flag;
        if (flag) { c1 = 0; if (firstG1<0) firstG1 = k; }
        else if (c1<3 && c1>0) c1++;
      }
      if (firstG1>=0) cab.write_CABAC_bit(CONTEXT_MODEL_COEFF_ABS_LEVEL_GREATER2_FLAG + ctxSet + (cIdx?4:0), abs(lev[firstG1])>2);
      /* sign_data_hiding: the sign of the last (lowest-frequency) level is the parity of the sum; noise content, so it is simply left out */
      const bool hidden = c.sdh && !no_hiding && firstP-lastP > 3;
      for (int k=0;k<nc-(hidden?1:0);k++) cab.write_CABAC_bypass(lev[k]<0);
      int rice = 0;
      for (int k=0;k<nc;k++) {
        const int a = abs(lev[k]);
        int base; bool more;
        if (k<8) {
          if (a==1) { base = 1; more = false; }
          else if (k==firstG1) { base = a>2 ? 3 : 2; more = a>2; }
          else { base = 2; more = true; }
        } else { base = 1; more = true; }
        if (!more) continue;
        code_abs_remaining(a-base, rice);
        if (a > 3*(1<<rice)) rice = std::min(rice+1, 4);
      }
    }
  }

  /* slice.cc:2140-2240: sig_coeff_flag context increment */
  static int sig_ctx(int xC, int yC, int log2, int cIdx, int scanIdx, int prevCsbf)
  {
    static const uint8_t map4[16] = {0,1,4,5, 2,3,4,5, 6,6,8,8, 7,7,8,8};
    int sigCtx;
    const int sbW = 1<<(log2-2);
    if (sbW==1) sigCtx = map4[(yC<<2)+xC];
    else if (xC+yC==0) sigCtx = 0;
    else {
      const int xS = xC>>2, yS = yC>>2, xP = xC&3, yP = yC&3;
      switch (prevCsbf) {
      case 0: sigCtx = (xP+yP>=3) ? 0 : (xP+yP>0) ? 1 : 2; break;
      case 1: sigCtx = (yP==0) ? 2 : (yP==1) ? 1 : 0; break;
      case 2: sigCtx = (xP==0) ? 2 : (xP==1) ? 1 : 0; break;
      default: sigCtx = 2;
      }
      if (cIdx==0) { if (xS+yS>0) sigCtx += 3; sigCtx += sbW==2 ? (scanIdx==0 ? 9 : 15) : 21; }
      else sigCtx += sbW==2 ? 9 : 12;
    }
    return cIdx==0 ? sigCtx : 27+sigCtx;
  }

  /* ---------- pictures ---------- */
  bool pic_tmvp = false, pic_col_from_l0 = true; int pic_col_idx = 0;

  void write_picture(const std::vector<PicPlan>& plan, size_t k)
  {
    const PicPlan& p = plan[k];
    /* reference picture set: what this picture uses + what later pictures still need (8.3.2) */
    std::vector<int> keep;
    for (int poc : dpbs.dpb) {
      bool need = std::find(p.refs.begin(),p.refs.end(),poc) != p.refs.end();
      for (size_t j=k+1;j<plan.size() && !need && !plan[j].idr;j++) need = std::find(plan[j].refs.begin(),plan[j].refs.end(),poc) != plan[j].refs.end();
      if (need && !p.idr) keep.push_back(poc);
    }
    std::vector<int> neg, pos;
    for (int poc : keep) (poc < p.poc ? neg : pos).push_back(poc);
    std::sort(neg.begin(),neg.end(),[](int a,int b){return a>b;}); std::sort(pos.begin(),pos.end());
    std::vector<bool> nu, pu;
    for (int poc : neg) nu.push_back(std::find(p.refs.begin(),p.refs.end(),poc) != p.refs.end());
    for (int poc : pos) pu.push_back(std::find(p.refs.begin(),p.refs.end(),poc) != p.refs.end());
    if (p.idr) keep.clear();
    dpbs.dpb = keep; dpbs.dpb.push_back(p.poc);

    std::fill(ct_depth.begin(),ct_depth.end(),0); std::fill(skipf.begin(),skipf.end(),0); std::fill(pmode.begin(),pmode.end(),PM_NONE);
    std::fill(pcmf.begin(),pcmf.end(),0); std::fill(ipm.begin(),ipm.end(),1); std::fill(ctb_slice.begin(),ctb_slice.end(),-1);

    pic_tmvp = c.tmvp && p.type != SLICE_TYPE_I && rng.pct(75);
    pic_col_from_l0 = true; pic_col_idx = 0;                  /* idx 0 of L0 in every slice: the same picture whatever n_l0 is (no list modification then) */
    if (c.lists_mod) pic_tmvp = false;

    const int nal_type = p.idr ? NAL_UNIT_IDR_W_RADL : NAL_UNIT_TRAIL_R;
    /* slices are runs of CTBs in TILE SCAN; with tiles they start at tile starts, with WPP at CTB row starts (7.4.7.1) */
    std::vector<int> cand;                                    /* tile-scan addresses a slice may start at */
    for (int ts=1; ts<nCtb; ts++) {
      const int rs = pps->CtbAddrTStoRS[ts];
      if (pps->tiles_enabled_flag) { if (pps->TileId[ts] != pps->TileId[ts-1]) cand.push_back(ts); }
      else if (!c.wpp || rs % ctbW == 0) cand.push_back(ts);
    }
    std::vector<int> start = {0};
    for (int s=1; s<c.slices && !cand.empty(); s++) { const int k = rng.below((int)cand.size()); start.push_back(cand[k]); cand.erase(cand.begin()+k); }
    std::sort(start.begin(), start.end());
    const int ns = (int)start.size();
    std::vector<context_model_table> wpp_saved(ctbH);
    context_model_table seg_end;                              /* 9.3.2.2 TableStateIdxDs: the contexts at the end of the previous slice segment */
    int n_nal = 0;
    for (int s=0;s<ns;s++) {
      const int end = s+1<ns ? start[s+1] : nCtb, addr0 = pps->CtbAddrTStoRS[start[s]];
      int type = p.type;
      if (type != SLICE_TYPE_I && ns>1 && rng.pct(15)) type = SLICE_TYPE_I;      /* an intra slice inside an inter picture */
      if (type == SLICE_TYPE_B && ns>1 && rng.pct(20)) type = SLICE_TYPE_P;
      /* the slice's segments: the first is independent, every further one (cut at the same kind of place slices may start) dependent */
      std::vector<int> seg = {start[s]};
      if (c.dep)
        for (int ts=start[s]+1; ts<end; ts++) {
          const int rs = pps->CtbAddrTStoRS[ts];
          const bool ok = pps->tiles_enabled_flag ? pps->TileId[ts] != pps->TileId[ts-1] : (!c.wpp || rs % ctbW == 0);
          if (ok && rng.pct(c.wpp || pps->tiles_enabled_flag ? c.dep : std::max(1, c.dep/8))) seg.push_back(ts);
        }
      for (size_t g=0; g<seg.size(); g++) {
        const int s0 = seg[g], s1 = g+1<seg.size() ? seg[g+1] : end;
        nal_begin(nal_type);
        if (g == 0) write_slice_header(p, neg, pos, nu, pu, n_nal, addr0, type, nal_type);
        else {
          write_dependent_segment_header(pps->CtbAddrTStoRS[s0], nal_type);
          /* slice.cc:4834-4889: a tile start initialises, otherwise the contexts of the previous segment's end carry on */
          if (pps->tiles_enabled_flag && pps->TileId[s0] != pps->TileId[s0-1]) models.init(S.init_type, S.qp);
          else models = seg_end.copy();
        }
        n_nal++;
        std::vector<int> sub_end;
        for (int ts=s0; ts<s1; ts++) {
          const int a = pps->CtbAddrTStoRS[ts], cx = a % ctbW, cy = a / ctbW;
          if (ts > s0 || g > 0) {                             /* slice.cc:4664-4690 / 5050-5075: what a new substream starts from */
            if (ts > s0 && pps->tiles_enabled_flag && pps->TileId[ts] != pps->TileId[ts-1]) models.init(S.init_type, S.qp);
            else if (c.wpp && cx == 0 && cy > 0) { if (ctbW > 1) models = wpp_saved[cy-1].copy(); else models.init(S.init_type, S.qp); }
          }
          ctb_slice[a] = addr0;
          if (S.sao_luma || S.sao_chroma) code_sao(cx,cy);
          code_quadtree(cx<<c.log2ctb, cy<<c.log2ctb, c.log2ctb, 0);
          if (c.wpp && cx == 1) wpp_saved[cy] = models.copy();  /* 9.3.2.2: storage after the second CTB of a row */
          const bool last = ts == s1-1;
          cab.write_CABAC_term_bit(last);                     /* end_of_slice_segment_flag */
          if (!last) {
            const int an = pps->CtbAddrTStoRS[ts+1];
            const bool sub = (pps->tiles_enabled_flag && pps->TileId[ts+1] != pps->TileId[ts]) || (c.wpp && an / ctbW != cy);
            if (sub) {                                        /* end_of_subset_one_bit, byte_alignment(), CABAC restart */
              cab.write_CABAC_term_bit(1);
              cab.flush_CABAC(); cab.write_bit(1); cab.write_bits(0, cab.number_free_bits_in_byte()); cab.flush_VLC();
              sub_end.push_back(cab.size());
              cab.init_CABAC();
            }
          }
        }
        seg_end = models.copy();
        cab.flush_CABAC(); cab.add_trailing_bits(); cab.flush_VLC();
        sub_end.push_back(cab.size());
        finish_slice_header(sub_end);
        nal_end((int)k, true);
      }
    }
    /* what the decoder must find in this picture if it stayed in sync with every bin (tools/f2_check.py) */
    fprintf(fchk, "pic %zu poc %d cus %ld pus %ld pcms %ld resid %ld coeffs %ld abs_sum %ld\n", k, p.poc, st.n_cus, st.n_pus, st.n_pcms, st.n_resid, st.n_coeffs, st.abs_sum);
    st = Stats();
  }

  /* ---------- decoded picture hash SEI (sei.cc:251-330 checks it): the stream is decoded once with the linked reference ---------- */
  static void plane_md5(const de265_image* im, int cIdx, uint8_t out[16])
  {
    int stride = 0;
    const uint8_t* p = de265_get_image_plane(im, cIdx, &stride);
    const int w = de265_get_image_width(im, cIdx), h = de265_get_image_height(im, cIdx), bpp = (de265_get_bits_per_pixel(im, cIdx)+7)/8;
    MD5_CTX m; MD5_Init(&m);
    for (int y=0;y<h;y++) MD5_Update(&m, (void*)(p + (size_t)y*stride), (unsigned long)w*bpp);     /* little-endian samples, as sei.cc:131-157 on this host */
    MD5_Final(out, &m);
  }

  std::vector<std::vector<uint8_t>> decode_md5s(int n_pics)
  {
    std::vector<std::vector<uint8_t>> md5(n_pics);
    de265_decoder_context* ctx = de265_new_decoder();
    de265_set_parameter_int(ctx, DE265_DECODER_PARAM_ACCELERATION_CODE, de265_acceleration_SCALAR);
    auto drain = [&]() {
      int more = 1;
      while (more) {
        de265_error e = de265_decode(ctx, &more);
        while (const de265_image* im = de265_get_next_picture(ctx)) {
          const int k = (int)de265_get_image_PTS(im);
          md5[k].resize(48);
          for (int cI=0;cI<3;cI++) plane_md5(im, cI, &md5[k][16*cI]);
        }
        if (e != DE265_OK) break;
      }
    };
    for (const Nal& n : nals) { de265_push_NAL(ctx, n.bytes.data(), (int)n.bytes.size(), n.pic < 0 ? 0 : n.pic, NULL); drain(); }
    de265_flush_data(ctx); drain();
    for (;;) { de265_error w = de265_get_warning(ctx); if (w == DE265_OK) break; fprintf(stderr, "f2_writer: the reference decoder warns: %s\n", de265_get_error_text(w)); }
    de265_free_decoder(ctx);
    for (int k=0;k<n_pics;k++) if (md5[k].size() != 48) die("md5 pass: a picture was not output by the reference decoder");
    return md5;
  }

  void run()
  {
    init_scan_orders();
    write_parameter_sets();
    ct_depth.assign(W4*H4,0); skipf = pmode = pcmf = ipm = ipmc = ipmc4 = ct_depth; ctb_slice.assign(nCtb,-1);
    fchk = fopen((c.out + ".chk").c_str(), "w");
    if (!fchk) die("cannot open .chk output");
    const std::vector<PicPlan> plan = plan_gop(c);
    for (size_t k=0;k<plan.size();k++) write_picture(plan, k);
    fclose(fchk);

    std::vector<std::vector<uint8_t>> md5;
    if (c.md5) md5 = decode_md5s((int)plan.size());
    FILE* fout = fopen(c.out.c_str(), "wb");
    if (!fout) die("cannot open output");
    static const uint8_t sc[4] = {0,0,0,1};
    for (size_t i=0;i<nals.size();i++) {
      fwrite(sc,1,4,fout); fwrite(nals[i].bytes.data(),1,nals[i].bytes.size(),fout);
      const int k = nals[i].pic;
      if (c.md5 && k >= 0 && (i+1 == nals.size() || nals[i+1].pic != k)) {        /* suffix SEI after the picture's last slice segment */
        nal_begin(NAL_UNIT_SUFFIX_SEI_NUT);
        hdr.write_bits(132,8); hdr.write_bits(49,8); hdr.write_bits(0,8);            /* decoded_picture_hash, 1+3*16 bytes, MD5 */
        for (int b=0;b<48;b++) hdr.write_bits(md5[k][b],8);
        hdr.add_trailing_bits(); hdr.flush_VLC();
        fwrite(sc,1,4,fout); fwrite(hdr.data(),1,hdr.size(),fout);
      }
    }
    fclose(fout);
  }
};

} // namespace

int main(int argc, char** argv)
{
  Cfg c;
  for (int i=1;i<argc;i++) {
    const char* eq = strchr(argv[i], '=');
    if (!eq) die("arguments are key=value");
    const std::string k(argv[i], eq-argv[i]), v(eq+1);
    if (k=="out") { c.out = v; continue; }
    if (k=="gop") { c.gop = v; continue; }
    bool ok = false;
    for (const Kv& kv : KV) if (k==kv.k) { c.*(kv.p) = atoi(v.c_str()); ok = true; }
    if (!ok) { fprintf(stderr, "f2_writer: unknown key %s\n", k.c_str()); return 2; }
  }
  if (c.w % (1<<c.log2mincb) || c.h % (1<<c.log2mincb)) die("w and h must be multiples of the minimum coding block size");
  if (c.log2mintb >= c.log2mincb || c.log2maxtb > std::min(5,c.log2ctb) || c.bits < 8 || c.bits > 12) die("inconsistent block sizes / bit depth");
  if (c.log2ctb < 4 || c.log2ctb > 6 || c.log2mincb < 3 || c.log2mincb > c.log2ctb || c.log2mintb < 2 || c.log2maxtb < c.log2mintb) die("block sizes outside 7.4.3.2.1");
  if (c.wpp && (c.tile_cols > 1 || c.tile_rows > 1)) die("wpp together with tiles is not written (the reference's parser does not combine them either, slice.cc:4664)");
  Writer w(c);
  w.run();
  return 0;
}
