/*
 * ref_shim.cc -- drives the COMPILED REFERENCE (libde265, built from /root/reference by
 * oracle/Makefile into oracle/_ref/) through the same C entry points as the CPU restatement
 * (oracle/hevc_oracle.h), so that tests can compare restatement, HIP path and reference on the
 * same inputs, and tools/make_ref_golden.py can write reference-generated fixtures.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes the reference's
 * headers where they lie (-I/root/reference/libde265) and calls its functions:
 *   init_acceleration_functions_fallback        fallback.cc:26        (function-level slots)
 *   decode_intra_prediction                     intrapred.cc:1115
 *   scale_coefficients                          transform.cc:628
 *   generate_inter_prediction_samples           motion.cc:279
 *   derive_edgeFlags / derive_boundaryStrength  deblock.cc:228 / :241
 *   apply_deblocking_filter                     deblock.cc:1020
 *   apply_sample_adaptive_offset_sequential     sao.cc:318
 * The picture-level entry builds a seq_parameter_set / pic_parameter_set / de265_image with
 * metadata the way SURVEY.md section 4 ("metadata-driven oracle") and Appendix C describe, from the
 * de265hip_picture_desc the product consumes plus the CU/TU structure arrays of the generator,
 * and then does what decode_TU (slice.cc:3424-3488), decode_prediction_unit (motion.cc:2142-2165)
 * and decode_some's post-processing (decctx.cc:757-766) do, in true decode order.
 *
 * It exists only in the build container (the reference cannot travel as source); the built
 * oracle/_ref/libde265_ref.so does travel to the GPU box like any other built .so.
 */
#include "libde265/de265.h"
#include "decctx.h"
#include "image.h"
#include "slice.h"
#include "sps.h"
#include "pps.h"
#include "motion.h"
#include "intrapred.h"
#include "transform.h"
#include "deblock.h"
#include "sao.h"
#include "fallback.h"
#include "acceleration.h"

#include <memory>
#include <vector>
#include <string.h>
#include <stdio.h>

#include "hevc_oracle.h"    /* oracle_image, ORACLE_ORD_*, the POD structs of include/de265_hip.h */

// deblock.cc exports these without a header declaration
void derive_boundaryStrength(de265_image* img, bool vertical, int yStart,int yEnd, int xStart,int xEnd);
bool derive_edgeFlags(de265_image* img);

namespace {

struct shim_ctx : public base_context {
  const de265_image* slots[DE265HIP_MAX_DPB_SLOTS];
  shim_ctx() { for (auto& s : slots) s = NULL; set_acceleration_functions(de265_acceleration_SCALAR); }
  virtual const de265_image* get_image(int id) const { return (id>=0 && id<DE265HIP_MAX_DPB_SLOTS) ? slots[id] : NULL; }
  virtual bool has_image(int id) const { return get_image(id)!=NULL; }
};

static acceleration_functions& accel()
{
  static acceleration_functions a;
  static bool init = false;
  if (!init) { memset(&a,0,sizeof(a)); init_acceleration_functions_fallback(&a); init = true; }
  return a;
}

static void copy_in(de265_image* img, int c, const void* src, int src_stride, int w, int h, int bytes)
{
  uint8_t* dst = img->get_image_plane(c);
  int ds = img->get_image_stride(c);
  for (int y=0;y<h;y++)
    memcpy(dst + (size_t)y*ds*bytes, (const uint8_t*)src + (size_t)y*src_stride*bytes, (size_t)w*bytes);
}
static void copy_out(const de265_image* img, int c, void* dst, int dst_stride, int w, int h, int bytes)
{
  const uint8_t* src = img->get_image_plane(c);
  int ss = img->get_image_stride(c);
  for (int y=0;y<h;y++)
    memcpy((uint8_t*)dst + (size_t)y*dst_stride*bytes, src + (size_t)y*ss*bytes, (size_t)w*bytes);
}

/* the reference state for one picture */
struct ref_picture {
  std::shared_ptr<video_parameter_set> vps;
  std::shared_ptr<seq_parameter_set> sps;
  std::shared_ptr<pic_parameter_set> pps;
  decoder_context* dctx;
  shim_ctx* sctx;
  de265_image* img;
  std::vector<de265_image*> refs;
  std::vector<slice_segment_header*> shdrs;
  thread_context* tctx;
  int w, h, bytes, cw, ch, ncomp = 3;

  ref_picture() : dctx(NULL), sctx(NULL), img(NULL), tctx(NULL) {}
  ~ref_picture() {
    delete tctx;
    if (!img) for (auto s : shdrs) delete s;
    delete img;                          // de265_image::release() deletes its slice headers (image.cc:500-506)
    for (auto r : refs) delete r;
    delete sctx;
    delete dctx;
  }
};

static int build_headers(ref_picture& R, const de265hip_pic_params& P, const uint8_t* scaling)
{
  R.vps = std::make_shared<video_parameter_set>();
  R.sps = std::make_shared<seq_parameter_set>();
  R.pps = std::make_shared<pic_parameter_set>();
  seq_parameter_set* sps = R.sps.get();
  pic_parameter_set* pps = R.pps.get();

  // encoder/encoder-context.cc:137-168 recipe
  sps->set_defaults();
  sps->set_CB_log2size_range(P.log2_min_cb_size, P.log2_ctb_size);
  sps->set_TB_log2size_range(P.log2_min_tb_size, P.log2_ctb_size < 5 ? P.log2_ctb_size : 5);
  sps->max_transform_hierarchy_depth_intra = 4;
  sps->max_transform_hierarchy_depth_inter = 4;
  sps->chroma_format_idc = P.chroma_format_idc;
  sps->bit_depth_luma = P.bit_depth_luma;
  sps->bit_depth_chroma = P.bit_depth_chroma;
  sps->set_resolution(P.width, P.height);
  sps->amp_enabled_flag = 1;
  sps->sample_adaptive_offset_enabled_flag = P.sample_adaptive_offset_enabled_flag;
  sps->pcm_enabled_flag = 1;
  sps->pcm_sample_bit_depth_luma = P.bit_depth_luma;
  sps->pcm_sample_bit_depth_chroma = P.bit_depth_chroma;
  sps->log2_min_pcm_luma_coding_block_size = 3;
  sps->log2_diff_max_min_pcm_luma_coding_block_size = 2;
  sps->pcm_loop_filter_disable_flag = P.pcm_loop_filter_disable_flag;
  sps->strong_intra_smoothing_enable_flag = P.strong_intra_smoothing_enable_flag;
  sps->scaling_list_enable_flag = P.scaling_list_enable_flag;
  // range extensions (sps.cc:1254-1264): only what the sample paths read
  if (P.implicit_rdpcm_enabled_flag || P.transform_skip_rotation_enabled_flag || P.intra_smoothing_disabled_flag ||
      P.high_precision_offsets_enabled_flag || P.cross_component_prediction_enabled_flag) sps->sps_range_extension_flag = 1;
  sps->range_extension.implicit_rdpcm_enabled_flag = P.implicit_rdpcm_enabled_flag;
  sps->range_extension.explicit_rdpcm_enabled_flag = 1;
  sps->range_extension.transform_skip_rotation_enabled_flag = P.transform_skip_rotation_enabled_flag;
  sps->range_extension.intra_smoothing_disabled_flag = P.intra_smoothing_disabled_flag;
  sps->range_extension.high_precision_offsets_enabled_flag = P.high_precision_offsets_enabled_flag;
  if (sps->compute_derived_values(true) != DE265_OK) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  // monochrome: sps.cc:40-41 tabulate SubWidthC/SubHeightC as -1 for chroma_format_idc 0 and image.cc:314-317
  // asserts they equal 1, so an assert-enabled build of the reference (this one) aborts on such a stream while a
  // release build goes on.  No luma path reads the two fields; give them the value alloc_image expects.
  if (P.chroma_format_idc == 0) { sps->SubWidthC = 1; sps->SubHeightC = 1; }

  pps->set_defaults();
  pps->sps = sps;
  pps->constrained_intra_pred_flag = P.constrained_intra_pred_flag;
  pps->transform_skip_enabled_flag = 1;
  pps->transquant_bypass_enable_flag = 1;
  pps->weighted_pred_flag = P.weighted_pred_flag;
  pps->weighted_bipred_flag = P.weighted_bipred_flag;
  pps->pic_cb_qp_offset = P.pic_cb_qp_offset;
  pps->pic_cr_qp_offset = P.pic_cr_qp_offset;
  pps->loop_filter_across_tiles_enabled_flag = P.loop_filter_across_tiles_enabled_flag;
  pps->range_extension.cross_component_prediction_enabled_flag = P.cross_component_prediction_enabled_flag;
  pps->range_extension.log2_max_transform_skip_block_size = 5;
  pps->num_tile_columns = P.num_tile_columns;
  pps->num_tile_rows = P.num_tile_rows;
  pps->tiles_enabled_flag = (P.num_tile_columns>1 || P.num_tile_rows>1);
  pps->uniform_spacing_flag = 0;       // explicit column widths / row heights from the boundary arrays
  for (int i=0;i<P.num_tile_columns;i++) pps->colWidth[i] = P.col_bd[i+1]-P.col_bd[i];
  for (int i=0;i<P.num_tile_rows;i++)    pps->rowHeight[i] = P.row_bd[i+1]-P.row_bd[i];
  if (P.scaling_list_enable_flag) {
    if (!scaling) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
    // transform.cc:487-493 reads pps.scaling_list; same flat layout as the boundary's blob
    static_assert(sizeof(scaling_list_data)==DE265HIP_SCALING_BLOB_BYTES, "scaling blob layout");
    memcpy(&pps->scaling_list, scaling, DE265HIP_SCALING_BLOB_BYTES);
    memcpy(&sps->scaling_list, scaling, DE265HIP_SCALING_BLOB_BYTES);
  }
  pps->set_derived_values(sps);
  return 0;
}

static slice_segment_header* make_shdr(const ref_picture& R, const de265hip_slice_params& s)
{
  slice_segment_header* h = new slice_segment_header;
  h->pps = R.pps.get();
  h->slice_type = s.slice_type;
  h->SliceAddrRS = s.slice_addr_rs;
  h->slice_segment_address = s.slice_addr_rs;
  h->slice_deblocking_filter_disabled_flag = s.slice_deblocking_filter_disabled_flag;
  h->slice_beta_offset = s.slice_beta_offset;
  h->slice_tc_offset = s.slice_tc_offset;
  h->slice_loop_filter_across_slices_enabled_flag = s.slice_loop_filter_across_slices_enabled_flag;
  h->slice_sao_luma_flag = s.slice_sao_luma_flag;
  h->slice_sao_chroma_flag = s.slice_sao_chroma_flag;
  h->luma_log2_weight_denom = s.luma_log2_weight_denom;
  h->ChromaLog2WeightDenom = s.chroma_log2_weight_denom;
  h->num_ref_idx_l0_active = 16;
  h->num_ref_idx_l1_active = 16;
  for (int l=0;l<2;l++)
    for (int i=0;i<16;i++) {
      h->LumaWeight[l][i] = s.luma_weight[l][i];
      h->luma_offset[l][i] = (int8_t)s.luma_offset[l][i];
      for (int c=0;c<2;c++) {
        h->ChromaWeight[l][i][c] = s.chroma_weight[l][i][c];
        h->ChromaOffset[l][i][c] = (int8_t)s.chroma_offset[l][i][c];
      }
      h->RefPicList[l][i] = s.ref_pic_list[l][i];
      h->RefPicList_PicState[l][i] = UsedForShortTermReference;
      h->LongTermRefPic[l][i] = 0;
    }
  return h;
}

/* Fill the de265_image metadata the reconstruction reads. */
static int fill_metadata(ref_picture& R, const de265hip_picture_desc* d,
                         const uint8_t* cb_log2_size, const uint8_t* cb_part_mode, const uint8_t* tu_split)
{
  const de265hip_pic_params& P = d->params;
  de265_image* img = R.img;
  const seq_parameter_set& sps = *R.sps;
  const int w4 = (P.width+3)/4;
  img->clear_metadata();

  for (int i=0;i<d->n_slices;i++) {
    slice_segment_header* h = make_shdr(R, d->slices[i]);
    R.shdrs.push_back(h);
    img->add_slice_segment_header(h);
  }
  for (int cy=0;cy<sps.PicHeightInCtbsY;cy++)
    for (int cx=0;cx<sps.PicWidthInCtbsY;cx++) {
      const de265hip_ctb_info& ci = d->ctbs[cx + cy*sps.PicWidthInCtbsY];
      img->set_SliceAddrRS(cx,cy, ci.slice_addr_rs);
      img->set_SliceHeaderIndex(cx<<sps.Log2CtbSizeY, cy<<sps.Log2CtbSizeY, ci.slice_idx);
      sao_info sao;
      sao.SaoTypeIdx = ci.sao_type_idx;
      sao.SaoEoClass = ci.sao_eo_class;
      for (int c=0;c<3;c++) {
        sao.sao_band_position[c] = ci.sao_band_position[c];
        for (int k=0;k<4;k++) sao.saoOffsetVal[c][k] = ci.sao_offset_val[c][k];
      }
      img->set_sao_info(cx,cy,&sao);
    }

  // coding blocks: what read_coding_unit stores (slice.cc:4245-4580)
  const int cbw = sps.PicWidthInMinCbsY, cbh = sps.PicHeightInMinCbsY;
  const int cbshift = sps.Log2MinCbSizeY;
  for (int by=0;by<cbh;by++)
    for (int bx=0;bx<cbw;bx++) {
      int l2 = cb_log2_size[bx + by*cbw];
      if (!l2) continue;
      int x0 = bx<<cbshift, y0 = by<<cbshift;
      uint8_t f = d->blk_flags[(x0>>2) + (y0>>2)*w4];
      img->set_log2CbSize(x0,y0,l2,true);
      img->set_PartMode(x0,y0,(enum PartMode)cb_part_mode[bx + by*cbw]);
      img->set_pred_mode(x0,y0,l2, (f & DE265HIP_BLK_INTRA) ? MODE_INTRA : MODE_INTER);
      if (f & DE265HIP_BLK_PCM) img->set_pcm_flag(x0,y0,l2);                       // slice.cc:4354
      if (f & DE265HIP_BLK_BYPASS) img->set_cu_transquant_bypass(x0,y0,l2);        // slice.cc:4280
      img->set_QPY(x0,y0,l2, d->blk_qp_y[(x0>>2) + (y0>>2)*w4]);
      img->clear_split_transform_flags(x0,y0,l2);
    }

  // transform tree split flags (slice.cc:3861) and nonzero-coefficient marks (slice.cc:2920-2922)
  const int tbw = sps.PicWidthInTbsY, tbh = sps.PicHeightInTbsY, tbshift = sps.Log2MinTrafoSize;
  for (int ty=0;ty<tbh;ty++)
    for (int tx=0;tx<tbw;tx++) {
      uint8_t s = tu_split[tx + ty*tbw];
      for (int dpt=0;dpt<5;dpt++)
        if (s & (1<<dpt)) img->set_split_transform_flag(tx<<tbshift, ty<<tbshift, dpt);
    }
  for (int i=0;i<d->n_tus;i++) {
    const de265hip_tu& t = d->tus[i];
    if (t.c_idx==0 && (t.flags & DE265HIP_TU_CBF)) img->set_nonzero_coefficient(t.x0,t.y0,t.log2_size);
  }
  // prediction blocks (motion.cc:2164)
  for (int i=0;i<d->n_pus;i++) {
    const de265hip_pu& p = d->pus[i];
    PBMotion m; memset(&m,0,sizeof(m));
    for (int l=0;l<2;l++) {
      m.predFlag[l] = (p.pred_flag>>l)&1;
      m.refIdx[l] = p.ref_idx[l];
      m.mv[l].x = p.mv[l][0]; m.mv[l].y = p.mv[l][1];
    }
    img->set_mv_info(p.x,p.y,p.w,p.h,m);
  }
  // the boundary's flattened per-4x4 views must say the same as the metadata just built
  if (d->blk_motion) {
    const int h4 = (P.height+3)/4;
    for (int y=0;y<h4;y++)
      for (int x=0;x<w4;x++) {
        if (d->blk_flags[x+y*w4] & DE265HIP_BLK_INTRA) continue;
        const PBMotion& m = img->get_mv_info(x<<2,y<<2);
        const de265hip_motion& bm = d->blk_motion[x+y*w4];
        const slice_segment_header* sh = img->get_SliceHeader(x<<2,y<<2);
        for (int l=0;l<2;l++) {
          int slot = m.predFlag[l] ? sh->RefPicList[l][m.refIdx[l]] : -1;
          if (slot != bm.ref_slot[l]) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
          if (m.predFlag[l] && (m.mv[l].x!=bm.mv[l][0] || m.mv[l].y!=bm.mv[l][1])) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
        }
      }
  }
  return 0;
}

static int setup(ref_picture& R, const de265hip_picture_desc* d, const oracle_image* dpb, const oracle_image* cur,
                 const uint8_t* cb_log2_size, const uint8_t* cb_part_mode, const uint8_t* tu_split)
{
  const de265hip_pic_params& P = d->params;
  if (P.chroma_format_idc < 0 || P.chroma_format_idc > 3 || P.extended_precision_processing_flag) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  // monochrome: the reference's motion compensation addresses the chroma planes whatever the format
  // (motion.cc:296-305), so only pictures without prediction units have a defined result there
  if (P.chroma_format_idc == 0 && d->n_pus) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  const enum de265_chroma chroma = P.chroma_format_idc == 0 ? de265_chroma_mono : P.chroma_format_idc == 1 ? de265_chroma_420 :
                                   (P.chroma_format_idc == 2 ? de265_chroma_422 : de265_chroma_444);
  const int cw = P.chroma_format_idc == 0 ? 0 : P.width / (P.chroma_format_idc == 3 ? 1 : 2);
  const int ch = P.chroma_format_idc == 0 ? 0 : P.height / (P.chroma_format_idc == 1 ? 2 : 1);
  R.ncomp = P.chroma_format_idc == 0 ? 1 : 3;
  int rc = build_headers(R, P, d->scaling_factors);
  if (rc) return rc;
  R.w = P.width; R.h = P.height; R.bytes = P.bit_depth_luma>8 ? 2 : 1;
  R.dctx = new decoder_context;
  R.dctx->set_acceleration_functions(de265_acceleration_SCALAR);
  R.sctx = new shim_ctx;
  R.img = new de265_image;
  R.cw = cw; R.ch = ch;
  if (R.img->alloc_image(P.width,P.height,chroma,R.sps,true,R.dctx,NULL,0,NULL,false) != DE265_OK)
    return DE265HIP_ERROR_OUT_OF_MEMORY;
  R.img->set_headers(R.vps,R.sps,R.pps);
  R.img->integrity = INTEGRITY_CORRECT;      // dpb.cc new_image(); any reference-side complaint lowers it
  if (cur)
    for (int c=0;c<R.ncomp;c++)
      copy_in(R.img,c,cur->plane[c],cur->stride[c], c?cw:P.width, c?ch:P.height, R.bytes);
  if (dpb)
    for (int s=0;s<DE265HIP_MAX_DPB_SLOTS;s++) {
      if (!dpb[s].plane[0]) continue;
      de265_image* r = new de265_image;
      if (r->alloc_image(P.width,P.height,chroma,R.sps,false,R.dctx,NULL,0,NULL,false) != DE265_OK)
        { delete r; return DE265HIP_ERROR_OUT_OF_MEMORY; }
      r->PicState = UsedForShortTermReference;
      for (int c=0;c<R.ncomp;c++)
        copy_in(r,c,dpb[s].plane[c],dpb[s].stride[c], c?cw:P.width, c?ch:P.height, R.bytes);
      R.refs.push_back(r);
      R.sctx->slots[s] = r;
    }
  rc = fill_metadata(R,d,cb_log2_size,cb_part_mode,tu_split);
  if (rc) return rc;
  R.tctx = new thread_context;
  thread_context* t = R.tctx;
  t->img = R.img; t->decctx = R.dctx; t->shdr = R.shdrs.empty() ? NULL : R.shdrs[0];
  // decctx.cc:2192-2195: 16-byte aligned scratch inside _coeffBuf, zero between TUs
  t->coeffBuf = (int16_t*)(((uintptr_t)t->_coeffBuf + 15) & ~(uintptr_t)15);
  memset(t->coeffBuf,0,32*32*sizeof(int16_t));
  t->ResScaleVal = 0; t->explicit_rdpcm_flag = 0; t->explicit_rdpcm_dir = 0;
  t->cu_transquant_bypass_flag = 0; memset(t->transform_skip_flag,0,3);
  return 0;
}

/* decode_TU (slice.cc:3424-3488) for one recorded TU: the same few lines, on the values the parser leaves in the
 * thread context (transform_skip_flag, cu_transquant_bypass_flag, explicit_rdpcm_*, ResScaleVal, qP*Prime, coeffList) */
static void do_tu(ref_picture& R, const de265hip_picture_desc* d, int i)
{
  const de265hip_tu& tu = d->tus[i];
  thread_context* t = R.tctx;
  const seq_parameter_set& sps = *R.sps;
  const int nT = 1<<tu.log2_size, c = tu.c_idx;
  const bool intra = tu.flags & DE265HIP_TU_INTRA;
  const bool cbf = tu.flags & DE265HIP_TU_CBF;
  t->cu_transquant_bypass_flag = (tu.flags & DE265HIP_TU_BYPASS) ? 1 : 0;
  memset(t->transform_skip_flag,0,3);
  t->transform_skip_flag[c] = (tu.flags & DE265HIP_TU_TSKIP) ? 1 : 0;
  t->explicit_rdpcm_flag = (tu.flags & DE265HIP_TU_EXPLICIT_RDPCM) ? 1 : 0;
  t->explicit_rdpcm_dir = (tu.flags & DE265HIP_TU_EXPLICIT_RDPCM_VERT) ? 1 : 0;
  t->ResScaleVal = c ? tu.res_scale_val : 0;
  t->qPYPrime = t->qPCbPrime = t->qPCrPrime = tu.qp;
  int residualDpcm = 0;
  if (intra) {
    decode_intra_prediction(R.img, tu.x0,tu.y0, (enum IntraPredMode)tu.intra_mode, nT, c);
    residualDpcm = sps.range_extension.implicit_rdpcm_enabled_flag &&
      (t->cu_transquant_bypass_flag || t->transform_skip_flag[c]) && (tu.intra_mode == 10 || tu.intra_mode == 26);
    if (residualDpcm && tu.intra_mode == 26) residualDpcm = 2;
  } else if (t->explicit_rdpcm_flag) residualDpcm = t->explicit_rdpcm_dir ? 2 : 1;
  if (cbf) {
    t->nCoeff[c] = tu.n_coeff;
    for (int k=0;k<tu.n_coeff;k++) {
      t->coeffList[c][k] = d->coeff_val[tu.coeff_offset+k];
      t->coeffPos[c][k]  = (int16_t)d->coeff_pos[tu.coeff_offset+k];
    }
    scale_coefficients(t, tu.x0,tu.y0, tu.x0,tu.y0, nT, c, t->transform_skip_flag[c], intra, residualDpcm);
  } else if (c != 0 && t->ResScaleVal) {            // cross-component prediction with CBF == 0 (slice.cc:3478-3487)
    t->nCoeff[c] = 0;
    scale_coefficients(t, tu.x0,tu.y0, tu.x0,tu.y0, nT, c, t->transform_skip_flag[c], intra, 0);
  }
}

static void do_pu(ref_picture& R, const de265hip_picture_desc* d, int i)
{
  const de265hip_pu& p = d->pus[i];
  PBMotion m; memset(&m,0,sizeof(m));
  for (int l=0;l<2;l++) {
    m.predFlag[l] = (p.pred_flag>>l)&1;
    m.refIdx[l] = p.ref_idx[l];
    m.mv[l].x = p.mv[l][0]; m.mv[l].y = p.mv[l][1];
  }
  generate_inter_prediction_samples(R.sctx, R.shdrs[p.slice_idx], R.img, p.x,p.y, 0,0, 64, p.w,p.h, &m);
}

/* read_pcm_samples_internal (slice.cc:4143-4183) reads the bits itself; what reaches the picture is
 * value << (bitDepth - pcmBits), which the boundary already carries.  Plain stores. */
static void do_pcm(ref_picture& R, const de265hip_picture_desc* d, int i)
{
  const de265hip_pcm& p = d->pcms[i];
  const uint16_t* s = d->pcm_samples + p.sample_offset;
  const int sw = R.sps->SubWidthC, sh = R.sps->SubHeightC;
  for (int c=0;c<R.ncomp;c++) {
    int nw = (1<<p.log2_cb_size) / (c?sw:1), nh = (1<<p.log2_cb_size) / (c?sh:1);
    int x0 = p.x0 / (c?sw:1), y0 = p.y0 / (c?sh:1);
    int stride = R.img->get_image_stride(c);
    for (int y=0;y<nh;y++)
      for (int x=0;x<nw;x++) {
        if (R.bytes==2) ((uint16_t*)R.img->get_image_plane(c))[x0+x + (y0+y)*stride] = *s++;
        else            R.img->get_image_plane(c)[x0+x + (y0+y)*stride] = (uint8_t)*s++;
      }
  }
}

} // namespace

extern "C" {

const char* ref_version(void) { return de265_get_version(); }

/* Whole picture, same contract as oracle_reconstruct() plus the generator's CU/TU structure
 * (the inputs of de265hip_derive_edge_flags): the reference derives its edge flags and boundary
 * strengths itself.  out_deblk (may be NULL, ceil(W/4)*ceil(H/4)) receives deblk_info afterwards:
 * edge bits 4-7 as derived by derive_edgeFlags, bS of the horizontal pass in bits 0-1. */
int ref_reconstruct(const de265hip_picture_desc* d, const uint32_t* order, int n_order,
                    const oracle_image* dpb, oracle_image* img, int last_stage,
                    const uint8_t* cb_log2_size, const uint8_t* cb_part_mode, const uint8_t* tu_split,
                    uint8_t* out_deblk)
{
  ref_picture R;
  int rc = setup(R,d,dpb,img,cb_log2_size,cb_part_mode,tu_split);
  if (rc) return rc;
  if (order) {
    for (int k=0;k<n_order;k++) {
      uint32_t e = order[k]; int idx = ORACLE_ORD_IDX(e);
      switch (e & 0xF0000000u) {
        case ORACLE_ORD_PU:  do_pu(R,d,idx); break;
        case ORACLE_ORD_PCM: do_pcm(R,d,idx); break;
        case ORACLE_ORD_TU:  do_tu(R,d,idx); break;
        default: return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
      }
    }
  } else {
    for (int i=0;i<d->n_pus;i++) do_pu(R,d,i);
    for (int i=0;i<d->n_pcms;i++) do_pcm(R,d,i);
    for (int i=0;i<d->n_tus;i++) do_tu(R,d,i);
  }
  // decctx.cc:1859-1883 run_postprocessing_filters_sequential
  if (last_stage >= DE265HIP_STAGE_DEBLOCKED && !d->params.disable_deblocking) apply_deblocking_filter(R.img);
  if (last_stage >= DE265HIP_STAGE_FINAL && !d->params.disable_sao) apply_sample_adaptive_offset_sequential(R.img);
  for (int c=0;c<R.ncomp;c++)
    copy_out(R.img,c,img->plane[c],img->stride[c], c?R.cw:R.w, c?R.ch:R.h, R.bytes);
  if (out_deblk) {
    int w4 = (R.w+3)/4, h4 = (R.h+3)/4;
    for (int y=0;y<h4;y++) for (int x=0;x<w4;x++) out_deblk[x+y*w4] = R.img->get_deblk_flags(x<<2,y<<2);
  }
  if (R.img->integrity != INTEGRITY_CORRECT) return DE265HIP_ERROR_DECODING;
  return 0;
}

/* a11: derive_edgeFlags (deblock.cc:228) on metadata only; ORs bits 4-7 into blk_flags like
 * de265hip_derive_edge_flags / oracle_derive_edge_flags. */
int ref_derive_edge_flags(const de265hip_picture_desc* d,
                          const uint8_t* cb_log2_size, const uint8_t* cb_part_mode, const uint8_t* tu_split,
                          uint8_t* blk_flags)
{
  ref_picture R;
  int rc = setup(R,d,NULL,NULL,cb_log2_size,cb_part_mode,tu_split);
  if (rc) return rc;
  derive_edgeFlags(R.img);
  int w4 = (R.w+3)/4, h4 = (R.h+3)/4;
  for (int y=0;y<h4;y++) for (int x=0;x<w4;x++) blk_flags[x+y*w4] |= R.img->get_deblk_flags(x<<2,y<<2) & 0xF0;
  return 0;
}

/* a12: derive_boundaryStrength (deblock.cc:241) for one direction, after derive_edgeFlags. */
int ref_derive_bs(const de265hip_picture_desc* d,
                  const uint8_t* cb_log2_size, const uint8_t* cb_part_mode, const uint8_t* tu_split,
                  int vertical, uint8_t* bs)
{
  ref_picture R;
  int rc = setup(R,d,NULL,NULL,cb_log2_size,cb_part_mode,tu_split);
  if (rc) return rc;
  derive_edgeFlags(R.img);
  derive_boundaryStrength(R.img, vertical!=0, 0,R.img->get_deblk_height(), 0,R.img->get_deblk_width());
  int w4 = (R.w+3)/4, h4 = (R.h+3)/4;
  for (int y=0;y<h4;y++) for (int x=0;x<w4;x++) bs[x+y*w4] = R.img->get_deblk_bS(x<<2,y<<2);
  return 0;
}

/* ---- function level: the fallback vtable slots (fallback.cc:26-127) ---- */

void ref_transform_add(int log2_size, int is_dst, int bit_depth, void* dst, ptrdiff_t stride, const int16_t* coeffs)
{
  acceleration_functions& a = accel();
  if (bit_depth<=8) {
    if (is_dst) a.transform_4x4_dst_add_8((uint8_t*)dst,coeffs,stride);
    else a.transform_add_8[log2_size-2]((uint8_t*)dst,coeffs,stride);
  } else {
    if (is_dst) a.transform_4x4_dst_add_16((uint16_t*)dst,coeffs,stride,bit_depth);
    else a.transform_add_16[log2_size-2]((uint16_t*)dst,coeffs,stride,bit_depth);
  }
}

/* transform_skip_residual + add_residual with the shifts of transform.cc:533-537 */
void ref_transform_skip_add(int log2_size, int bit_depth, void* dst, ptrdiff_t stride, const int16_t* coeffs)
{
  acceleration_functions& a = accel();
  int nT = 1<<log2_size;
  int32_t r[32*32];
  int bdShift = 20-bit_depth, tsShift = 5+log2_size;
  a.transform_skip_residual(r,coeffs,nT,tsShift,bdShift);
  if (bit_depth<=8) a.add_residual_8((uint8_t*)dst,stride,r,nT,bit_depth);
  else a.add_residual_16((uint16_t*)dst,stride,r,nT,bit_depth);
}

void ref_transform_bypass_add(int log2_size, int bit_depth, void* dst, ptrdiff_t stride, const int16_t* coeffs)
{
  acceleration_functions& a = accel();
  int nT = 1<<log2_size;
  int32_t r[32*32];
  a.transform_bypass(r,coeffs,nT);
  if (bit_depth<=8) a.add_residual_8((uint8_t*)dst,stride,r,nT,bit_depth);
  else a.add_residual_16((uint16_t*)dst,stride,r,nT,bit_depth);
}

/* int32-residual forms (acceleration.h:164-168); kind 0 = DCT, 1 = DST (4x4) */
void ref_transform_residual(int log2_size, int is_dst, int bit_depth, int32_t* residual, const int16_t* coeffs)
{
  acceleration_functions& a = accel();
  int bdShift = 20-bit_depth, max_coeff_bits = 15;
  if (is_dst) a.transform_idst_4x4(residual,coeffs,bdShift,max_coeff_bits);
  else switch (log2_size) {
    case 2: a.transform_idct_4x4(residual,coeffs,bdShift,max_coeff_bits); break;
    case 3: a.transform_idct_8x8(residual,coeffs,bdShift,max_coeff_bits); break;
    case 4: a.transform_idct_16x16(residual,coeffs,bdShift,max_coeff_bits); break;
    default: a.transform_idct_32x32(residual,coeffs,bdShift,max_coeff_bits); break;
  }
}

void ref_put_qpel(int bit_depth, int16_t* out, ptrdiff_t out_stride, const void* src, ptrdiff_t src_stride,
                  int w, int h, int dx, int dy)
{
  ALIGNED_16(int16_t) mcbuffer[64*(64+8)];
  accel().put_hevc_qpel(out,out_stride,src,src_stride,w,h,mcbuffer,dx,dy,bit_depth);
}

void ref_put_epel(int bit_depth, int16_t* out, ptrdiff_t out_stride, const void* src, ptrdiff_t src_stride,
                  int w, int h, int mx, int my)
{
  ALIGNED_16(int16_t) mcbuffer[64*(64+8)];
  acceleration_functions& a = accel();
  // slot choice of mc_chroma (motion.cc:206-267)
  if (mx==0 && my==0) a.put_hevc_epel(out,out_stride,src,src_stride,w,h,mx,my,mcbuffer,bit_depth);
  else if (my==0)     a.put_hevc_epel_h(out,out_stride,src,src_stride,w,h,mx,my,mcbuffer,bit_depth);
  else if (mx==0)     a.put_hevc_epel_v(out,out_stride,src,src_stride,w,h,mx,my,mcbuffer,bit_depth);
  else                a.put_hevc_epel_hv(out,out_stride,src,src_stride,w,h,mx,my,mcbuffer,bit_depth);
}

void ref_put_pred(int mode, int bit_depth, void* dst, ptrdiff_t ds, const int16_t* s0, const int16_t* s1,
                  ptrdiff_t ss, int w, int h, int w0, int o0, int w1, int o1, int log2wd)
{
  acceleration_functions& a = accel();
  switch (mode) {
    case 0: a.put_unweighted_pred(dst,ds,s0,ss,w,h,bit_depth); break;
    case 1: a.put_weighted_pred(dst,ds,s0,ss,w,h,w0,o0,log2wd,bit_depth); break;
    case 2: a.put_weighted_pred_avg(dst,ds,s0,s1,ss,w,h,bit_depth); break;
    default: a.put_weighted_bipred(dst,ds,s0,s1,ss,w,h,w0,o0,w1,o1,log2wd,bit_depth); break;
  }
}

/* ---- Interface 1 check: the product's init_acceleration_functions_hip (passed in as a pointer; this library never
 * links the product) fills a REAL struct acceleration_functions after the fallback, as decctx.cc:430-449 would, and
 * every decoder slot is then called through the reference's own struct type next to the fallback's slot on the same
 * seeded inputs.  counts[k] receives the number of mismatching calls of family k:
 *   0 put_*_pred (8 slots)  1 put_hevc_epel* (8)  2 put_hevc_qpel (32)  3 transform_add / dst_add (10)
 *   4 transform_bypass*, transform_skip_residual, rdpcm_*, transform_id{c,s}t_* (11)  5 add_residual (2)
 *   6 rotate_coefficients  7 transform_skip_rdpcm_{v,h}_8  8 slots that must keep the fallback's pointer
 * Returns the number of slot calls made. */
int ref_vtable_compare(void (*init_hip)(struct acceleration_functions*), uint64_t seed, int iters, int* counts)
{
  acceleration_functions R, H;
  memset(&R,0,sizeof(R)); memset(&H,0,sizeof(H));
  init_acceleration_functions_fallback(&R);
  init_acceleration_functions_fallback(&H);
  init_hip(&H);
  for (int k=0;k<9;k++) counts[k]=0;
  uint64_t st = seed*6364136223846793005ull + 1442695040888963407ull;
  auto rnd = [&](int lo,int hi)->int { st = st*6364136223846793005ull + 1442695040888963407ull; return lo + (int)((st>>33) % (uint64_t)(hi-lo+1)); };
  int calls = 0;
  // encoder slots and the deprecated transform_skip_{8,16} stay with the fallback
  if (H.fwd_transform_4x4_dst_8 != R.fwd_transform_4x4_dst_8 || H.transform_skip_8 != R.transform_skip_8 || H.transform_skip_16 != R.transform_skip_16) counts[8]++;
  for (int k=0;k<4;k++) if (H.fwd_transform_8[k] != R.fwd_transform_8[k] || H.hadamard_transform_8[k] != R.hadamard_transform_8[k]) counts[8]++;
  // and every decoder slot was replaced
  if (H.put_weighted_pred_avg_8 == R.put_weighted_pred_avg_8 || H.transform_add_16[3] == R.transform_add_16[3] ||
      H.put_hevc_qpel_16[3][3] == R.put_hevc_qpel_16[3][3] || H.transform_idct_32x32 == R.transform_idct_32x32 ||
      H.add_residual_16 == R.add_residual_16 || H.rdpcm_h == R.rdpcm_h) counts[8]++;

  static const int bds[4] = { 8, 9, 10, 12 };
  for (int it=0; it<iters; it++) {
    const int bd = bds[it & 3];
    const int maxv = (1<<bd)-1;
    const bool hi = bd > 8;
    // ---- sample prediction writes
    {
      int w = 2*rnd(1,32), h = rnd(1,32)*2, ss = 64, ds = 80;
      std::vector<int16_t> a(ss*64), b(ss*64);
      for (auto& v : a) v = (int16_t)rnd(-9000,16383);
      for (auto& v : b) v = (int16_t)rnd(-9000,16383);
      std::vector<uint16_t> d0(ds*64), d1, d2;
      for (auto& v : d0) v = (uint16_t)rnd(0,maxv);
      int w0 = rnd(-128,127), w1 = rnd(-128,127), o0 = rnd(-128,127)*(1<<(bd-8)), o1 = rnd(-128,127)*(1<<(bd-8));
      int wd = rnd(0,7) + (14-bd > 2 ? 14-bd : 2);
      for (int mode=0; mode<4; mode++) {
        if (!hi) {
          std::vector<uint8_t> x(ds*64), y;
          for (size_t i=0;i<x.size();i++) x[i] = (uint8_t)d0[i];
          y = x;
          switch (mode) {
            case 0: R.put_unweighted_pred_8(x.data(),ds,a.data(),ss,w,h); H.put_unweighted_pred_8(y.data(),ds,a.data(),ss,w,h); break;
            case 1: R.put_weighted_pred_8(x.data(),ds,a.data(),ss,w,h,w0,o0,wd); H.put_weighted_pred_8(y.data(),ds,a.data(),ss,w,h,w0,o0,wd); break;
            case 2: R.put_weighted_pred_avg_8(x.data(),ds,a.data(),b.data(),ss,w,h); H.put_weighted_pred_avg_8(y.data(),ds,a.data(),b.data(),ss,w,h); break;
            default: R.put_weighted_bipred_8(x.data(),ds,a.data(),b.data(),ss,w,h,w0,o0,w1,o1,wd); H.put_weighted_bipred_8(y.data(),ds,a.data(),b.data(),ss,w,h,w0,o0,w1,o1,wd); break;
          }
          if (x != y) counts[0]++;
        } else {
          d1 = d0; d2 = d0;
          switch (mode) {
            case 0: R.put_unweighted_pred_16(d1.data(),ds,a.data(),ss,w,h,bd); H.put_unweighted_pred_16(d2.data(),ds,a.data(),ss,w,h,bd); break;
            case 1: R.put_weighted_pred_16(d1.data(),ds,a.data(),ss,w,h,w0,o0,wd,bd); H.put_weighted_pred_16(d2.data(),ds,a.data(),ss,w,h,w0,o0,wd,bd); break;
            case 2: R.put_weighted_pred_avg_16(d1.data(),ds,a.data(),b.data(),ss,w,h,bd); H.put_weighted_pred_avg_16(d2.data(),ds,a.data(),b.data(),ss,w,h,bd); break;
            default: R.put_weighted_bipred_16(d1.data(),ds,a.data(),b.data(),ss,w,h,w0,o0,w1,o1,wd,bd); H.put_weighted_bipred_16(d2.data(),ds,a.data(),b.data(),ss,w,h,w0,o0,w1,o1,wd,bd); break;
          }
          if (d1 != d2) counts[0]++;
        }
        calls++;
      }
    }
    // ---- interpolation: the block sits in a plane with exactly the margins its fraction may read (guard cells differ
    // between the two runs: a slot that reads beyond its margins shows up as a mismatch)
    {
      const int PW = 96, PH = 96;
      std::vector<uint16_t> p16(PW*PH); std::vector<uint8_t> p8(PW*PH);
      for (int i=0;i<PW*PH;i++) { p16[i] = (uint16_t)rnd(0,maxv); p8[i] = (uint8_t)p16[i]; }
      ALIGNED_16(int16_t) mcb[64*72];
      for (int k=0;k<2;k++) {
        const bool luma = k==0;
        int w = luma ? 4*rnd(1,16) : 2*rnd(1,16), h = luma ? 4*rnd(1,16) : 2*rnd(1,16);
        int fx = rnd(0, luma?3:7), fy = rnd(0, luma?3:7);
        int x0 = rnd(4, PW-w-5), y0 = rnd(4, PH-h-5);
        std::vector<int16_t> o1(64*64, 0), o2(64*64, 0);
        if (luma) {
          if (hi) { R.put_hevc_qpel_16[fx][fy](o1.data(),64,&p16[x0+y0*PW],PW,w,h,mcb,bd); H.put_hevc_qpel_16[fx][fy](o2.data(),64,&p16[x0+y0*PW],PW,w,h,mcb,bd); }
          else    { R.put_hevc_qpel_8[fx][fy](o1.data(),64,&p8[x0+y0*PW],PW,w,h,mcb);      H.put_hevc_qpel_8[fx][fy](o2.data(),64,&p8[x0+y0*PW],PW,w,h,mcb); }
          if (o1 != o2) counts[2]++;
        } else {
          // slot choice of mc_chroma (motion.cc:206-267)
          if (hi) {
            auto fr = (fx==0&&fy==0) ? R.put_hevc_epel_16 : (fy==0 ? R.put_hevc_epel_h_16 : (fx==0 ? R.put_hevc_epel_v_16 : R.put_hevc_epel_hv_16));
            auto fh = (fx==0&&fy==0) ? H.put_hevc_epel_16 : (fy==0 ? H.put_hevc_epel_h_16 : (fx==0 ? H.put_hevc_epel_v_16 : H.put_hevc_epel_hv_16));
            fr(o1.data(),64,&p16[x0+y0*PW],PW,w,h,fx,fy,mcb,bd); fh(o2.data(),64,&p16[x0+y0*PW],PW,w,h,fx,fy,mcb,bd);
          } else if (fx==0 && fy==0) {
            R.put_hevc_epel_8(o1.data(),64,&p8[x0+y0*PW],PW,w,h,fx,fy,mcb); H.put_hevc_epel_8(o2.data(),64,&p8[x0+y0*PW],PW,w,h,fx,fy,mcb);
          } else {
            auto fr = fy==0 ? R.put_hevc_epel_h_8 : (fx==0 ? R.put_hevc_epel_v_8 : R.put_hevc_epel_hv_8);
            auto fh = fy==0 ? H.put_hevc_epel_h_8 : (fx==0 ? H.put_hevc_epel_v_8 : H.put_hevc_epel_hv_8);
            fr(o1.data(),64,&p8[x0+y0*PW],PW,w,h,fx,fy,mcb,8); fh(o2.data(),64,&p8[x0+y0*PW],PW,w,h,fx,fy,mcb,8);
          }
          if (o1 != o2) counts[1]++;
        }
        calls++;
      }
    }
    // ---- transforms
    {
      const int log2 = 2 + (it>>2)%4, nT = 1<<log2, n = nT*nT, ds = 48;
      std::vector<int16_t> c(n, 0);
      const int kind = rnd(0,4);
      for (int i=0;i<n;i++) {
        if (kind==0) c[i] = (int16_t)((i%nT<4 && i/nT<4) ? rnd(-600,600) : 0);
        else if (kind==1) c[i] = (int16_t)rnd(-300,300);
        else if (kind==2) { int q = rnd(0,4); c[i] = q==0 ? -32768 : (q==1 ? 32767 : 0); }
        else if (kind==3) c[i] = (int16_t)rnd(-32768,32767);
      }
      if (kind==4) c[rnd(0,n-1)] = (int16_t)rnd(-32768,32767);
      std::vector<uint16_t> d0(ds*32);
      for (auto& v : d0) v = (uint16_t)rnd(0,maxv);
      for (int dst=0; dst < (log2==2 ? 2 : 1); dst++) {
        if (hi) {
          std::vector<uint16_t> x = d0, y = d0;
          if (dst) { R.transform_4x4_dst_add_16(x.data(),c.data(),ds,bd); H.transform_4x4_dst_add_16(y.data(),c.data(),ds,bd); }
          else     { R.transform_add_16[log2-2](x.data(),c.data(),ds,bd); H.transform_add_16[log2-2](y.data(),c.data(),ds,bd); }
          if (x != y) counts[3]++;
        } else {
          std::vector<uint8_t> x(d0.size()), y;
          for (size_t i=0;i<x.size();i++) x[i] = (uint8_t)d0[i];
          y = x;
          if (dst) { R.transform_4x4_dst_add_8(x.data(),c.data(),ds); H.transform_4x4_dst_add_8(y.data(),c.data(),ds); }
          else     { R.transform_add_8[log2-2](x.data(),c.data(),ds); H.transform_add_8[log2-2](y.data(),c.data(),ds); }
          if (x != y) counts[3]++;
        }
        calls++;
      }
      // int32-residual family
      std::vector<int32_t> r1(n), r2(n);
      const int bdShift = 20-bd, tsShift = 5+log2;
      auto cmp4 = [&]() { if (r1 != r2) counts[4]++; calls++; };
      R.transform_bypass(r1.data(),c.data(),nT); H.transform_bypass(r2.data(),c.data(),nT); cmp4();
      R.transform_bypass_rdpcm_v(r1.data(),c.data(),nT); H.transform_bypass_rdpcm_v(r2.data(),c.data(),nT); cmp4();
      R.transform_bypass_rdpcm_h(r1.data(),c.data(),nT); H.transform_bypass_rdpcm_h(r2.data(),c.data(),nT); cmp4();
      R.transform_skip_residual(r1.data(),c.data(),nT,tsShift,bdShift); H.transform_skip_residual(r2.data(),c.data(),nT,tsShift,bdShift); cmp4();
      R.rdpcm_v(r1.data(),c.data(),nT,tsShift,bdShift); H.rdpcm_v(r2.data(),c.data(),nT,tsShift,bdShift); cmp4();
      R.rdpcm_h(r1.data(),c.data(),nT,tsShift,bdShift); H.rdpcm_h(r2.data(),c.data(),nT,tsShift,bdShift); cmp4();
      switch (log2) {
        case 2: R.transform_idct_4x4(r1.data(),c.data(),bdShift,15); H.transform_idct_4x4(r2.data(),c.data(),bdShift,15); cmp4();
                R.transform_idst_4x4(r1.data(),c.data(),bdShift,15); H.transform_idst_4x4(r2.data(),c.data(),bdShift,15); cmp4(); break;
        case 3: R.transform_idct_8x8(r1.data(),c.data(),bdShift,15); H.transform_idct_8x8(r2.data(),c.data(),bdShift,15); cmp4(); break;
        case 4: R.transform_idct_16x16(r1.data(),c.data(),bdShift,15); H.transform_idct_16x16(r2.data(),c.data(),bdShift,15); cmp4(); break;
        default: R.transform_idct_32x32(r1.data(),c.data(),bdShift,15); H.transform_idct_32x32(r2.data(),c.data(),bdShift,15); cmp4(); break;
      }
      // add_residual on whatever the last transform left in r1 (wide range incl. clipping)
      for (auto& v : r1) v = v / (1 + rnd(0,3));
      if (hi) { std::vector<uint16_t> x = d0, y = d0; R.add_residual_16(x.data(),ds,r1.data(),nT,bd); H.add_residual_16(y.data(),ds,r1.data(),nT,bd); if (x != y) counts[5]++; }
      else { std::vector<uint8_t> x(d0.size()), y; for (size_t i=0;i<x.size();i++) x[i] = (uint8_t)d0[i]; y = x;
             R.add_residual_8(x.data(),ds,r1.data(),nT,bd); H.add_residual_8(y.data(),ds,r1.data(),nT,bd); if (x != y) counts[5]++; }
      calls++;
      std::vector<int16_t> c1 = c, c2 = c;
      R.rotate_coefficients(c1.data(),nT); H.rotate_coefficients(c2.data(),nT);
      if (c1 != c2) counts[6]++;
      calls++;
      if (!hi) {
        std::vector<uint8_t> x(d0.size()), y; for (size_t i=0;i<x.size();i++) x[i] = (uint8_t)d0[i]; y = x;
        R.transform_skip_rdpcm_v_8(x.data(),c.data(),log2,ds); H.transform_skip_rdpcm_v_8(y.data(),c.data(),log2,ds);
        if (x != y) counts[7]++;
        R.transform_skip_rdpcm_h_8(x.data(),c.data(),log2,ds); H.transform_skip_rdpcm_h_8(y.data(),c.data(),log2,ds);
        if (x != y) counts[7]++;
        calls += 2;
      }
    }
  }
  return calls;
}

/* One intra TU on a small single-slice picture: decode_intra_prediction (intrapred.cc:1115) reads
 * its neighbours from `plane` (w x h luma samples of geometry, the component plane is passed) and
 * writes the nT x nT block at (x0,y0) (component samples).  All earlier z-order neighbours inside
 * the picture count as available (single slice, single tile, everything intra).  */
int ref_intra_tu(int bit_depth, int strong_smoothing, int log2_ctb, int pic_w, int pic_h,
                 void* plane, ptrdiff_t stride, int c_idx, int x0, int y0, int log2_size, int mode)
{
  de265hip_pic_params P; memset(&P,0,sizeof(P));
  P.width = pic_w; P.height = pic_h; P.bit_depth_luma = P.bit_depth_chroma = bit_depth; P.chroma_format_idc = 1;
  P.log2_ctb_size = log2_ctb; P.log2_min_cb_size = 3; P.log2_min_tb_size = 2;
  P.strong_intra_smoothing_enable_flag = strong_smoothing;
  P.num_tile_columns = P.num_tile_rows = 1;
  int ctb = 1<<log2_ctb;
  P.col_bd[1] = (pic_w+ctb-1)/ctb; P.row_bd[1] = (pic_h+ctb-1)/ctb;
  ref_picture R;
  int rc = build_headers(R,P,NULL);
  if (rc) return rc;
  R.dctx = new decoder_context;
  R.img = new de265_image;
  if (R.img->alloc_image(pic_w,pic_h,de265_chroma_420,R.sps,true,R.dctx,NULL,0,NULL,false) != DE265_OK)
    return DE265HIP_ERROR_OUT_OF_MEMORY;
  R.img->set_headers(R.vps,R.sps,R.pps);
  R.img->clear_metadata();
  R.img->fill_pred_mode(MODE_INTRA);
  int bytes = bit_depth>8 ? 2 : 1;
  int cw = c_idx ? pic_w/2 : pic_w, ch = c_idx ? pic_h/2 : pic_h;
  copy_in(R.img,c_idx,plane,stride,cw,ch,bytes);
  decode_intra_prediction(R.img,x0,y0,(enum IntraPredMode)mode,1<<log2_size,c_idx);
  copy_out(R.img,c_idx,plane,stride,cw,ch,bytes);
  return 0;
}

} // extern "C"
