/*
 * f1_recorder.cc -- SURVEY.md 8(f1) as code: the recorder behind the hooks that oracle/f1_recorder.patch adds to a
 * scratch copy of the reference (oracle/Makefile `make f1`, build container only).  While the patched libde265 decodes a
 * real bitstream on the CPU, the hooks collect -- per picture -- exactly what the product's frame-level interface
 * consumes (include/de265_hip.h: de265hip_picture_desc) and write it, together with the reference's own picture before
 * and after its post-filters, to $F1_OUT/pic_NNN.f1.  tools/make_stream_golden.py turns those dumps into the
 * tests/golden/stream_* fixtures; tests/test_stream_golden.py replays them through the oracle and, on the GPU,
 * through the product's incremental recorder API (de265hip_record_* / de265hip_recorder_submit).
 *
 * TEST INFRASTRUCTURE.  This is also the template of the libde265-side integration: each hook body
 * is what the real recorder does, with de265hip_record_* in place of the vectors below.
 */
#include "decctx.h"
#include "image.h"
#include "slice.h"
#include "sps.h"
#include "pps.h"
#include "motion.h"
#include "f1_hooks.h"
#include "../include/de265_hip.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

bool derive_edgeFlags(de265_image* img);          // deblock.cc:228

namespace {

struct State {
  std::vector<de265hip_tu> tus;
  std::vector<int16_t> cval;
  std::vector<uint16_t> cpos;
  std::vector<de265hip_pu> pus;
  std::vector<de265hip_pcm> pcms;
  std::vector<uint16_t> pcm_samples;
  std::vector<uint8_t> file;          // the dump being assembled between f1_submit and f1_picture_done
  int n_pictures = 0;
};
State S;

/* ---- offload mode (F1_MODE=hip): the product library, loaded at run time (this decoder never links it) ---- */
struct Hip {
  bool tried = false, on = false;
  void* lib = nullptr;
  de265hip_decoder* dec = nullptr;
  int  (*decoder_new)(de265hip_decoder**, int);
  int  (*dpb_alloc)(de265hip_decoder*, int, int, int, int, int);
  int  (*dpb_download)(de265hip_decoder*, int, int, void*, ptrdiff_t);
  int  (*recorder_new)(de265hip_recorder**, const de265hip_pic_params*, const uint8_t*);
  void (*recorder_free)(de265hip_recorder*);
  int  (*record_tu)(de265hip_recorder*, const de265hip_tu*, const int16_t*, const uint16_t*);
  int  (*record_pu)(de265hip_recorder*, const de265hip_pu*);
  int  (*record_pcm)(de265hip_recorder*, int, int, int, const uint16_t*);
  int  (*record_slice)(de265hip_recorder*, const de265hip_slice_params*);
  int  (*record_ctb)(de265hip_recorder*, int, const de265hip_ctb_info*);
  int  (*record_blk_planes)(de265hip_recorder*, const uint8_t*, const int8_t*, const de265hip_motion*);
  int  (*recorder_submit)(de265hip_decoder*, int, de265hip_recorder*, de265hip_picture**);
  int  (*picture_run)(de265hip_decoder*, de265hip_picture*, int);
  int  (*decoder_sync)(de265hip_decoder*);
  void (*picture_free)(de265hip_picture*);
};
Hip H;

[[noreturn]] void hip_die(const char* what, int rc) { fprintf(stderr, "f1_recorder (hip mode): %s failed (%d)\n", what, rc); exit(6); }

bool hip_mode()
{
  if (H.tried) return H.on;
  H.tried = true;
  const char* m = getenv("F1_MODE");
  if (!m || strcmp(m, "hip")) return false;
  const char* path = getenv("F1_HIP_LIB");
  H.lib = dlopen(path ? path : "libde265_hip.so", RTLD_NOW);
  if (!H.lib) { fprintf(stderr, "f1_recorder: %s\n", dlerror()); exit(6); }
#define SYM(f) do { *(void**)&H.f = dlsym(H.lib, "de265hip_" #f); if (!H.f) hip_die("dlsym de265hip_" #f, 0); } while (0)
  SYM(decoder_new); SYM(dpb_alloc); SYM(dpb_download); SYM(recorder_new); SYM(recorder_free); SYM(record_tu); SYM(record_pu);
  SYM(record_pcm); SYM(record_slice); SYM(record_ctb); SYM(record_blk_planes); SYM(recorder_submit); SYM(picture_run);
  SYM(decoder_sync); SYM(picture_free);
#undef SYM
  int rc = H.decoder_new(&H.dec, -1);
  if (rc) hip_die("de265hip_decoder_new", rc);
  H.on = true;
  return true;
}

template <class T> void put(const T* p, size_t n) { const uint8_t* b = (const uint8_t*)p; S.file.insert(S.file.end(), b, b + n*sizeof(T)); }
void put_i32(int32_t v) { put(&v, 1); }

void put_planes(const de265_image* img)
{
  for (int c=0;c<3;c++) {
    const int w = img->get_width(c), h = img->get_height(c), bpp = img->get_bytes_per_pixel(c);
    const uint8_t* p = img->get_image_plane(c);
    for (int y=0;y<h;y++) put(p + (size_t)y*img->get_image_stride(c)*bpp, (size_t)w*bpp);
  }
}

int dpb_index_of(const de265_image* img)
{
  decoder_context* ctx = img->decctx;
  for (int i=0; ctx && ctx->has_image(i); i++) if (ctx->get_image(i)==img) return i;
  return 0;
}

} // namespace

bool f1_record_tu(thread_context* tctx, int x0, int y0, int nT, int cIdx, int cuPredMode, bool cbf)
{
  const bool intra = cuPredMode == MODE_INTRA;
  if (!intra && !cbf) return hip_mode();                          // decode_TU does nothing for it (slice.cc:3424-3488)
  de265_image* img = tctx->img;
  const seq_parameter_set& sps = img->get_sps();
  de265hip_tu t; memset(&t,0,sizeof(t));
  t.x0 = (uint16_t)x0; t.y0 = (uint16_t)y0; t.c_idx = (uint8_t)cIdx;
  int l2 = 0; while ((1<<l2) < nT) l2++;
  t.log2_size = (uint8_t)l2;
  t.flags = (uint8_t)((intra ? DE265HIP_TU_INTRA : 0) | (cbf ? DE265HIP_TU_CBF : 0) |
                      (tctx->transform_skip_flag[cIdx] && cbf ? DE265HIP_TU_TSKIP : 0) |
                      (tctx->cu_transquant_bypass_flag ? DE265HIP_TU_BYPASS : 0));
  if (intra) {                                         // slice.cc:3436-3451
    int m = cIdx==0 ? img->get_IntraPredMode(x0,y0) : img->get_IntraPredModeC(x0*sps.SubWidthC, y0*sps.SubHeightC);
    if (m<0 || m>=35) m = INTRA_DC;
    t.intra_mode = (uint8_t)m;
  }
  t.qp = (int8_t)(cIdx==0 ? tctx->qPYPrime : (cIdx==1 ? tctx->qPCbPrime : tctx->qPCrPrime));   // transform.cc:362-368
  if (cbf) {
    t.n_coeff = (uint16_t)tctx->nCoeff[cIdx];
    t.coeff_offset = (uint32_t)S.cval.size();
    for (int i=0;i<tctx->nCoeff[cIdx];i++) {
      S.cval.push_back(tctx->coeffList[cIdx][i]);
      S.cpos.push_back((uint16_t)tctx->coeffPos[cIdx][i]);
    }
  }
  S.tus.push_back(t);
  return hip_mode();
}

bool f1_record_pu(const slice_segment_header* shdr, de265_image*, int xP, int yP, int nPbW, int nPbH, const PBMotion* vi)
{
  de265hip_pu p; memset(&p,0,sizeof(p));
  p.x = (uint16_t)xP; p.y = (uint16_t)yP; p.w = (uint8_t)nPbW; p.h = (uint8_t)nPbH;
  p.pred_flag = (uint8_t)((vi->predFlag[0]?1:0) | (vi->predFlag[1]?2:0));
  p.slice_idx = (uint16_t)shdr->slice_index;
  for (int l=0;l<2;l++)                                 // libde265 leaves the unused list's refIdx / mv undefined: record zeros
    if (vi->predFlag[l]) { p.ref_idx[l] = vi->refIdx[l]; p.mv[l][0] = vi->mv[l].x; p.mv[l][1] = vi->mv[l].y; }
  S.pus.push_back(p);
  return hip_mode();
}

void f1_record_pcm(thread_context* tctx, int x0, int y0, int log2CbSize)
{
  de265_image* img = tctx->img;
  de265hip_pcm p; memset(&p,0,sizeof(p));
  p.x0 = (uint16_t)x0; p.y0 = (uint16_t)y0; p.log2_cb_size = (uint8_t)log2CbSize;
  p.sample_offset = (uint32_t)S.pcm_samples.size();
  for (int c=0;c<3;c++) {                              // the samples as read_pcm_samples_internal stored them (already << shift)
    const int n = (1<<log2CbSize) >> (c?1:0), xx = x0 >> (c?1:0), yy = y0 >> (c?1:0), stride = img->get_image_stride(c);
    for (int y=0;y<n;y++) for (int x=0;x<n;x++)
      S.pcm_samples.push_back(img->high_bit_depth(c) ? ((const uint16_t*)img->get_image_plane(c))[xx+x+(yy+y)*stride]
                                                      : img->get_image_plane(c)[xx+x+(yy+y)*stride]);
  }
  S.pcms.push_back(p);
}

bool f1_submit(de265_image* img)
{
  const seq_parameter_set& sps = img->get_sps();
  const pic_parameter_set& pps = img->get_pps();
  const bool hip = hip_mode();
  de265hip_pic_params P; memset(&P,0,sizeof(P));
  P.width = sps.pic_width_in_luma_samples; P.height = sps.pic_height_in_luma_samples;
  P.bit_depth_luma = sps.BitDepth_Y; P.bit_depth_chroma = sps.BitDepth_C; P.chroma_format_idc = sps.chroma_format_idc;
  P.log2_ctb_size = sps.Log2CtbSizeY; P.log2_min_cb_size = sps.Log2MinCbSizeY; P.log2_min_tb_size = sps.Log2MinTrafoSize;
  P.pcm_loop_filter_disable_flag = sps.pcm_loop_filter_disable_flag;
  P.strong_intra_smoothing_enable_flag = sps.strong_intra_smoothing_enable_flag;
  P.constrained_intra_pred_flag = pps.constrained_intra_pred_flag;
  P.sample_adaptive_offset_enabled_flag = sps.sample_adaptive_offset_enabled_flag;
  P.scaling_list_enable_flag = sps.scaling_list_enable_flag;
  P.weighted_pred_flag = pps.weighted_pred_flag; P.weighted_bipred_flag = pps.weighted_bipred_flag;
  P.pic_cb_qp_offset = pps.pic_cb_qp_offset; P.pic_cr_qp_offset = pps.pic_cr_qp_offset;
  P.loop_filter_across_tiles_enabled_flag = pps.loop_filter_across_tiles_enabled_flag;
  P.num_tile_columns = pps.num_tile_columns; P.num_tile_rows = pps.num_tile_rows;
  for (int i=0;i<=pps.num_tile_columns && i<24;i++) P.col_bd[i] = (uint16_t)pps.colBd[i];
  for (int i=0;i<=pps.num_tile_rows && i<24;i++) P.row_bd[i] = (uint16_t)pps.rowBd[i];
  P.disable_deblocking = img->decctx->param_disable_deblocking;
  P.disable_sao = img->decctx->param_disable_sao;

  const int w4 = (P.width+3)/4, h4 = (P.height+3)/4, nctb = sps.PicSizeInCtbsY;
  const int cbw = sps.PicWidthInMinCbsY, cbh = sps.PicHeightInMinCbsY, tbw = sps.PicWidthInTbsY, tbh = sps.PicHeightInTbsY;
  const uint8_t* scaling = sps.scaling_list_enable_flag ? (const uint8_t*)&pps.scaling_list : NULL;      // transform.cc:487-493
  std::vector<de265hip_slice_params> slices;
  for (slice_segment_header* h : img->slices) {
    de265hip_slice_params s; memset(&s,0,sizeof(s));
    s.slice_type = h->slice_type; s.slice_addr_rs = h->SliceAddrRS;
    s.slice_deblocking_filter_disabled_flag = h->slice_deblocking_filter_disabled_flag;
    s.slice_beta_offset = h->slice_beta_offset; s.slice_tc_offset = h->slice_tc_offset;
    s.slice_loop_filter_across_slices_enabled_flag = h->slice_loop_filter_across_slices_enabled_flag;
    s.slice_sao_luma_flag = h->slice_sao_luma_flag; s.slice_sao_chroma_flag = h->slice_sao_chroma_flag;
    s.luma_log2_weight_denom = h->luma_log2_weight_denom; s.chroma_log2_weight_denom = h->ChromaLog2WeightDenom;
    for (int l=0;l<2;l++) for (int i=0;i<16;i++) {
      s.luma_weight[l][i] = h->LumaWeight[l][i]; s.luma_offset[l][i] = h->luma_offset[l][i];
      for (int c=0;c<2;c++) { s.chroma_weight[l][i][c] = h->ChromaWeight[l][i][c]; s.chroma_offset[l][i][c] = h->ChromaOffset[l][i][c]; }
      s.ref_pic_list[l][i] = (int8_t)h->RefPicList[l][i];
    }
    slices.push_back(s);
  }
  std::vector<de265hip_ctb_info> ctbs(nctb);
  for (int a=0;a<nctb;a++) {
    const int cx = a % sps.PicWidthInCtbsY, cy = a / sps.PicWidthInCtbsY;
    de265hip_ctb_info ci; memset(&ci,0,sizeof(ci));
    ci.slice_addr_rs = (uint16_t)img->get_SliceAddrRS(cx,cy);
    ci.slice_idx = (uint16_t)img->get_SliceHeaderIndexCtb(cx,cy);
    const sao_info* sao = img->get_sao_info(cx,cy);
    ci.sao_type_idx = sao->SaoTypeIdx; ci.sao_eo_class = sao->SaoEoClass;
    for (int c=0;c<3;c++) { ci.sao_band_position[c] = sao->sao_band_position[c]; for (int k=0;k<4;k++) ci.sao_offset_val[c][k] = sao->saoOffsetVal[c][k]; }
    ctbs[a] = ci;
  }
  // flattened per-4x4 views (include/de265_hip.h DE265HIP_BLK_*), without the edge bits
  std::vector<uint8_t> flags((size_t)w4*h4); std::vector<int8_t> qp((size_t)w4*h4); std::vector<de265hip_motion> mot((size_t)w4*h4);
  for (int y=0;y<h4;y++) for (int x=0;x<w4;x++) {
    const int xl = x<<2, yl = y<<2;
    const bool intra = img->get_pred_mode(xl,yl)==MODE_INTRA;
    flags[x+y*w4] = (uint8_t)((intra ? DE265HIP_BLK_INTRA : 0) | (img->get_nonzero_coefficient(xl,yl) ? DE265HIP_BLK_NONZERO : 0) |
                              (img->get_pcm_flag(xl,yl) ? DE265HIP_BLK_PCM : 0) | (img->get_cu_transquant_bypass(xl,yl) ? DE265HIP_BLK_BYPASS : 0));
    qp[x+y*w4] = (int8_t)img->get_QPY(xl,yl);
    de265hip_motion m; memset(&m,0,sizeof(m)); m.ref_slot[0] = m.ref_slot[1] = -1;
    if (!intra) {
      const PBMotion& pb = img->get_mv_info(xl,yl);
      const slice_segment_header* sh = img->get_SliceHeader(xl,yl);
      for (int l=0;l<2;l++) if (pb.predFlag[l] && sh) {           // deblock.cc:295-304 compares these
        m.ref_slot[l] = (int8_t)sh->RefPicList[l][pb.refIdx[l]]; m.mv[l][0] = pb.mv[l].x; m.mv[l][1] = pb.mv[l].y;
      }
    }
    mot[x+y*w4] = m;
  }

  if (hip) {
    // ---- OFFLOAD: what an integrated libde265 does at decctx.cc:757-766 instead of run_postprocessing_filters_*
    if (!P.disable_deblocking) derive_edgeFlags(img);               // cheap host code; or de265hip_derive_edge_flags
    for (int y=0;y<h4;y++) for (int x=0;x<w4;x++) flags[x+y*w4] |= img->get_deblk_flags(x<<2,y<<2) & 0xF0;
    const int slot = dpb_index_of(img) % DE265HIP_MAX_DPB_SLOTS;
    int rc;
    de265hip_recorder* rec = NULL;
    if ((rc = H.recorder_new(&rec, &P, scaling))) hip_die("recorder_new", rc);
    for (const auto& sl : slices) if ((rc = H.record_slice(rec, &sl))) hip_die("record_slice", rc);
    for (int a=0;a<nctb;a++) if ((rc = H.record_ctb(rec, a, &ctbs[a]))) hip_die("record_ctb", rc);
    for (const auto& t : S.tus) if ((rc = H.record_tu(rec, &t, S.cval.data()+t.coeff_offset, S.cpos.data()+t.coeff_offset))) hip_die("record_tu", rc);
    for (const auto& pu : S.pus) if ((rc = H.record_pu(rec, &pu))) hip_die("record_pu", rc);
    for (const auto& pc : S.pcms) if ((rc = H.record_pcm(rec, pc.x0, pc.y0, pc.log2_cb_size, S.pcm_samples.data()+pc.sample_offset))) hip_die("record_pcm", rc);
    if ((rc = H.record_blk_planes(rec, flags.data(), qp.data(), mot.data()))) hip_die("record_blk_planes", rc);
    de265hip_picture* pic = NULL;
    if ((rc = H.dpb_alloc(H.dec, slot, P.width, P.height, P.bit_depth_luma, P.bit_depth_chroma))) hip_die("dpb_alloc", rc);
    if ((rc = H.recorder_submit(H.dec, slot, rec, &pic))) hip_die("recorder_submit", rc);
    if ((rc = H.picture_run(H.dec, pic, DE265HIP_STAGE_FINAL))) hip_die("picture_run", rc);
    if ((rc = H.decoder_sync(H.dec))) hip_die("decoder_sync", rc);
    for (int c=0;c<3;c++)                                           // the GPU's picture becomes the decoder's picture
      if ((rc = H.dpb_download(H.dec, slot, c, img->get_image_plane(c), (ptrdiff_t)img->get_image_stride(c)*img->get_bytes_per_pixel(c))))
        hip_die("dpb_download", rc);
    H.picture_free(pic);
    H.recorder_free(rec);
    S.tus.clear(); S.cval.clear(); S.cpos.clear(); S.pus.clear(); S.pcms.clear(); S.pcm_samples.clear();
    S.n_pictures++;
    return true;
  }

  // ---- DUMP: the description + the reference's own pictures to $F1_OUT/pic_NNN.f1
  S.file.clear();
  put("F1DESC02", 8);
  put(&P,1);
  put_i32((int)slices.size()); put_i32(nctb); put_i32((int)S.tus.size()); put_i32((int)S.cval.size());
  put_i32((int)S.pus.size()); put_i32((int)S.pcms.size()); put_i32((int)S.pcm_samples.size());
  put_i32(w4); put_i32(h4); put_i32(cbw*cbh); put_i32(tbw*tbh);
  put_i32(dpb_index_of(img)); put_i32(img->PicOrderCntVal); put_i32(scaling ? 1 : 0);
  if (scaling) put(scaling, DE265HIP_SCALING_BLOB_BYTES);
  put(slices.data(), slices.size()); put(ctbs.data(), ctbs.size());
  put(S.tus.data(), S.tus.size()); put(S.cval.data(), S.cval.size()); put(S.cpos.data(), S.cpos.size());
  put(S.pus.data(), S.pus.size()); put(S.pcms.data(), S.pcms.size()); put(S.pcm_samples.data(), S.pcm_samples.size());
  put(flags.data(), flags.size()); put(qp.data(), qp.size()); put(mot.data(), mot.size());
  // CU/TU structure: the inputs of de265hip_derive_edge_flags
  std::vector<uint8_t> cb_log2((size_t)cbw*cbh), cb_part((size_t)cbw*cbh), tu_split((size_t)tbw*tbh);
  for (int y=0;y<cbh;y++) for (int x=0;x<cbw;x++) {
    const int l2 = img->get_log2CbSize_cbUnits(x,y);
    cb_log2[x+y*cbw] = (uint8_t)l2;
    cb_part[x+y*cbw] = l2 ? (uint8_t)img->get_PartMode(x<<sps.Log2MinCbSizeY, y<<sps.Log2MinCbSizeY) : 0;
  }
  for (int y=0;y<tbh;y++) for (int x=0;x<tbw;x++) {
    uint8_t b = 0;
    if ((x<<sps.Log2MinTrafoSize) < P.width && (y<<sps.Log2MinTrafoSize) < P.height)
      for (int d=0;d<5;d++) if (img->get_split_transform_flag(x<<sps.Log2MinTrafoSize, y<<sps.Log2MinTrafoSize, d)) b |= (uint8_t)(1<<d);
    tu_split[x+y*tbw] = b;
  }
  put(cb_log2.data(), cb_log2.size()); put(cb_part.data(), cb_part.size()); put(tu_split.data(), tu_split.size());
  put_planes(img);                                     // the reference's picture before its post-filters
  S.tus.clear(); S.cval.clear(); S.cpos.clear(); S.pus.clear(); S.pcms.clear(); S.pcm_samples.clear();
  return false;
}

void f1_picture_done(de265_image* img)
{
  if (hip_mode()) return;
  const int w4 = (img->get_width(0)+3)/4, h4 = (img->get_height(0)+3)/4;
  std::vector<uint8_t> edges((size_t)w4*h4);
  for (int y=0;y<h4;y++) for (int x=0;x<w4;x++) edges[x+y*w4] = img->get_deblk_flags(x<<2,y<<2) & 0xF0;   // as derive_edgeFlags marked them
  put(edges.data(), edges.size());
  put_planes(img);                                     // ... and after deblocking + SAO
  const char* dir = getenv("F1_OUT");
  if (!dir) { S.file.clear(); S.n_pictures++; return; }   // plain CPU decode (f1_dec stream.bin out.yuv): nothing to dump
  char name[1024];
  snprintf(name, sizeof(name), "%s/pic_%03d.f1", dir, S.n_pictures++);
  FILE* f = fopen(name, "wb");
  if (!f || fwrite(S.file.data(), 1, S.file.size(), f) != S.file.size()) { fprintf(stderr, "f1_recorder: cannot write %s\n", name); exit(5); }
  fclose(f);
  S.file.clear();
}
