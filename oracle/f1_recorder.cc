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

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

bool derive_edgeFlags(de265_image* img);          // deblock.cc:228

namespace {

/* What the hooks collect for the picture being parsed.  With libde265's worker threads (WPP rows / tiles,
 * decctx.cc:976-1178) the hooks fire on several threads at once: every thread appends to a buffer of its own, each record
 * tagged with its CTB's address in tile scan; f1_submit merges the buffers by that address.  A CTB is parsed by one
 * thread, so the merged order is exactly the order of a sequential decode. */
struct Rec {
  std::vector<de265hip_tu> tus;   std::vector<uint32_t> tu_ts;
  std::vector<int16_t> cval;      std::vector<uint16_t> cpos;
  std::vector<de265hip_pu> pus;   std::vector<uint32_t> pu_ts;
  std::vector<de265hip_pcm> pcms; std::vector<uint32_t> pcm_ts;
  std::vector<uint16_t> pcm_samples;
  void clear() { tus.clear(); tu_ts.clear(); cval.clear(); cpos.clear(); pus.clear(); pu_ts.clear(); pcms.clear(); pcm_ts.clear(); pcm_samples.clear(); }
};
std::mutex reg_mu;
std::vector<Rec*> all_recs;
thread_local Rec* my_rec = nullptr;
Rec& mine()
{
  if (!my_rec) { my_rec = new Rec; std::lock_guard<std::mutex> lk(reg_mu); all_recs.push_back(my_rec); }
  return *my_rec;
}

struct PicRec {                        // one picture's merged records (decode order)
  std::vector<de265hip_tu> tus;
  std::vector<int16_t> cval;
  std::vector<uint16_t> cpos;
  std::vector<de265hip_pu> pus;
  std::vector<de265hip_pcm> pcms;
  std::vector<uint16_t> pcm_samples;
};
struct State {
  std::vector<uint8_t> file;          // the dump being assembled between f1_submit and f1_picture_done
  int n_pictures = 0;
};
State S;

/* the main thread, picture finished, no hook running: take the per-thread buffers of the picture away (moves, no copies) */
std::vector<Rec> spare_recs;            // emptied buffers that kept their capacity (guarded by reg_mu)
void take_records(std::vector<Rec>& out)
{
  std::lock_guard<std::mutex> lk(reg_mu);
  out.resize(all_recs.size());
  for (size_t r=0;r<all_recs.size();r++) {
    out[r] = std::move(*all_recs[r]);
    // the parse thread gets a buffer that has been through a picture already: no re-growing, no fresh pages to fault in
    if (!spare_recs.empty()) { *all_recs[r] = std::move(spare_recs.back()); spare_recs.pop_back(); }
    all_recs[r]->clear();
  }
}

void merge_records(std::vector<Rec>& recs, PicRec& M)
{
  std::vector<Rec*> all_recs; for (Rec& r : recs) all_recs.push_back(&r);
  struct Ref { uint32_t ts; uint16_t rec; uint32_t idx; };
  std::vector<Ref> order;
  auto sorted = [&](std::vector<uint32_t> Rec::* key) {
    order.clear();
    for (size_t r=0;r<all_recs.size();r++) { const auto& k = all_recs[r]->*key; for (size_t i=0;i<k.size();i++) order.push_back(Ref{k[i],(uint16_t)r,(uint32_t)i}); }
    std::stable_sort(order.begin(), order.end(), [](const Ref& a, const Ref& b) { return a.ts < b.ts; });
  };
  sorted(&Rec::tu_ts);
  M.tus.reserve(order.size());
  for (const Ref& o : order) {
    const Rec& R = *all_recs[o.rec];
    de265hip_tu t = R.tus[o.idx];
    if (t.n_coeff) {
      const uint32_t src = t.coeff_offset;
      t.coeff_offset = (uint32_t)M.cval.size();
      M.cval.insert(M.cval.end(), R.cval.begin()+src, R.cval.begin()+src+t.n_coeff);
      M.cpos.insert(M.cpos.end(), R.cpos.begin()+src, R.cpos.begin()+src+t.n_coeff);
    }
    M.tus.push_back(t);
  }
  sorted(&Rec::pu_ts);
  for (const Ref& o : order) M.pus.push_back(all_recs[o.rec]->pus[o.idx]);
  sorted(&Rec::pcm_ts);
  for (const Ref& o : order) {
    const Rec& R = *all_recs[o.rec];
    de265hip_pcm pc = R.pcms[o.idx];
    const uint32_t src = pc.sample_offset;                 // luma + both chroma blocks, sized by the chroma format
    const uint32_t cnt = (o.idx+1 < R.pcms.size() ? R.pcms[o.idx+1].sample_offset : (uint32_t)R.pcm_samples.size()) - src;
    pc.sample_offset = (uint32_t)M.pcm_samples.size();
    M.pcm_samples.insert(M.pcm_samples.end(), R.pcm_samples.begin()+src, R.pcm_samples.begin()+src+cnt);
    M.pcms.push_back(pc);
  }
  {
    std::lock_guard<std::mutex> lk(reg_mu);
    for (Rec& r : recs) if (spare_recs.size() < 64) { r.clear(); spare_recs.push_back(std::move(r)); }
  }
  recs.clear();
}

/* ---- offload mode (F1_MODE=hip): the product library, loaded at run time (this decoder never links it) ---- */
struct Hip {
  bool tried = false, on = false;
  void* lib = nullptr;
  de265hip_decoder* dec = nullptr;
  int  (*decoder_new)(de265hip_decoder**, int);
  int  (*dpb_alloc)(de265hip_decoder*, int, int, int, int, int, int);      // de265hip_dpb_alloc_ex
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
  int  (*dpb_download_async)(de265hip_decoder*, int, int, void*, ptrdiff_t);
  int  (*dpb_download_planes_async)(de265hip_decoder*, int, void* const*, const ptrdiff_t*, uint64_t*);
  int  (*dpb_wait)(de265hip_decoder*, int);
  int  (*dpb_fill)(de265hip_decoder*, int, int, int, int);
  int  (*dpb_upload)(de265hip_decoder*, int, int, const void*, ptrdiff_t);
  void* (*host_alloc)(size_t);
  void (*host_free)(void*);
  int  (*pipeline_new)(de265hip_pipeline**, de265hip_decoder*, int);
  int  (*pipeline_submit)(de265hip_pipeline*, int, de265hip_prepare_fn, void*, void* const*, const ptrdiff_t*, uint64_t*);
  int  (*pipeline_wait)(de265hip_pipeline*, uint64_t);
  int  (*pipeline_drain)(de265hip_pipeline*);
  void (*pipeline_free)(de265hip_pipeline*);
  de265hip_pipeline* pipe = nullptr;
  bool pipeline = false;               // F1_PIPELINE=n: SURVEY 8(f3), n worker threads between parser and device, see f1_submit
  int n_workers = 1;
};
Hip H;

[[noreturn]] void hip_die(const char* what, int rc) { fprintf(stderr, "f1_recorder (hip mode): %s failed (%d)\n", what, rc); exit(6); }   // (start-up only: no library, no device)
/* A back-end call failed while decoding: the first such error is kept and becomes the result of de265_decode (the hook behind
 * the picture, f1_picture_done, returns it; de265.h:82-139 - the back end's codes are de265_error numbers), the picture is
 * marked (image.h:57-61 integrity), and the hooks stop handing pictures over.  The decoder stays usable for de265_free_decoder. */
std::mutex err_mu;
int hip_error = 0;
int hip_fail(const char* what, int rc)
{
  std::lock_guard<std::mutex> lk(err_mu);
  if (!hip_error) { hip_error = rc ? rc : DE265HIP_ERROR_DECODING; fprintf(stderr, "f1_recorder (hip mode): %s failed (%d)\n", what, rc); }
  return hip_error;
}
int hip_failed() { std::lock_guard<std::mutex> lk(err_mu); return hip_error; }

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
  SYM(decoder_new); *(void**)&H.dpb_alloc = dlsym(H.lib, "de265hip_dpb_alloc_ex"); if (!H.dpb_alloc) hip_die("dlsym de265hip_dpb_alloc_ex", 0); SYM(dpb_download); SYM(recorder_new); SYM(recorder_free); SYM(record_tu); SYM(record_pu);
  SYM(record_pcm); SYM(record_slice); SYM(record_ctb); SYM(record_blk_planes); SYM(recorder_submit); SYM(picture_run);
  SYM(decoder_sync); SYM(picture_free); SYM(dpb_fill); SYM(dpb_upload); SYM(dpb_download_async); SYM(dpb_download_planes_async); SYM(dpb_wait); SYM(host_alloc); SYM(host_free);
  SYM(pipeline_new); SYM(pipeline_submit); SYM(pipeline_wait); SYM(pipeline_drain); SYM(pipeline_free);
#undef SYM
  const char* pl = getenv("F1_PIPELINE");
  H.pipeline = pl && atoi(pl) != 0;
  H.n_workers = H.pipeline ? std::min(8, std::max(1, atoi(pl))) : 1;
  int rc = H.decoder_new(&H.dec, -1);
  if (rc) hip_die("de265hip_decoder_new", rc);
  if (const char* fi = getenv("F1_FAULT")) {             // tests: the one failure the device side admits (a dependency wait that expires)
    int (*inject)(de265hip_decoder*, int, uint32_t) = nullptr;
    *(void**)&inject = dlsym(H.lib, "de265hip_debug_fault_injection");
    if (!inject || inject(H.dec, 1, (uint32_t)std::max(1, atoi(fi)))) hip_die("de265hip_debug_fault_injection", 0);
  }
  H.on = true;
  return true;
}

// plain CPU decode (neither offload nor dump): the hooks stay out of the way, so that `f1_dec stream.bin` times the reference itself
bool passive()
{
  static const bool p = !hip_mode() && !getenv("F1_OUT");
  return p;
}

// F1_PROFILE=1: where the host side of a picture goes (printed by f1_drain / at the last picture)
struct Prof { double merge=0, flatten=0, edges=0, record=0, submit=0, run=0, out=0, free_=0, w_out=0, w_queue=0, w_turn=0, t_first=0, t_last=0, in_submit=0; int n=0; bool on=false, init=false; };
Prof PR;
bool prof_on() { if (!PR.init) { PR.init = true; const char* e = getenv("F1_PROFILE"); PR.on = e && atoi(e); } return PR.on; }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void prof_print()
{
  if (!prof_on() || !PR.n) return;
  fprintf(stderr, "f1 profile, ms per picture over %d pictures: merge %.2f  flatten %.2f  edge flags %.2f | record_* %.2f  recorder_submit(build) %.2f  picture_run %.2f  copy-out %.2f\n",
          PR.n, 1e3*PR.merge/PR.n, 1e3*PR.flatten/PR.n, 1e3*PR.edges/PR.n, 1e3*PR.record/PR.n, 1e3*PR.submit/PR.n, 1e3*PR.run/PR.n, 1e3*PR.out/PR.n);
  fprintf(stderr, "            frees %.2f  worker waits for its launch turn %.2f | parser thread: %.2f ms between two f1_submit calls, of which inside f1_submit %.2f (waiting for a free pipeline place %.2f), waiting for copy-outs %.2f\n",
          1e3*PR.free_/PR.n, 1e3*PR.w_turn/PR.n, PR.n > 1 ? 1e3*(PR.t_last-PR.t_first)/(PR.n-1) : 0.0, 1e3*PR.in_submit/PR.n, 1e3*PR.w_queue/PR.n, 1e3*PR.w_out/PR.n);
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

/* offload mode: the same merge straight into the product's recorder (de265hip_record_*), without the intermediate copy:
 * the per-thread buffers are walked in tile-scan order of their CTBs and every record is handed over where it lies */
int merge_into_recorder(std::vector<Rec>& recs, de265hip_recorder* rec,
                        int (*record_tu)(de265hip_recorder*, const de265hip_tu*, const int16_t*, const uint16_t*),
                        int (*record_pu)(de265hip_recorder*, const de265hip_pu*),
                        int (*record_pcm)(de265hip_recorder*, int, int, int, const uint16_t*))
{
  struct Ref { uint32_t ts; uint16_t rec; uint32_t idx; };
  std::vector<Ref> order;
  auto sorted = [&](std::vector<uint32_t> Rec::* key) {
    order.clear();
    for (size_t r=0;r<recs.size();r++) { const auto& k = recs[r].*key; for (size_t i=0;i<k.size();i++) order.push_back(Ref{k[i],(uint16_t)r,(uint32_t)i}); }
    std::stable_sort(order.begin(), order.end(), [](const Ref& a, const Ref& b) { return a.ts < b.ts; });
  };
  int rc;
  // (one thread's records are in decode order already: nothing to sort)
  if (recs.size() == 1) {
    const Rec& R = recs[0];
    for (const auto& t : R.tus) if ((rc = record_tu(rec, &t, R.cval.data()+t.coeff_offset, R.cpos.data()+t.coeff_offset))) return rc;
    for (const auto& pu : R.pus) if ((rc = record_pu(rec, &pu))) return rc;
    for (size_t i=0;i<R.pcms.size();i++) if ((rc = record_pcm(rec, R.pcms[i].x0, R.pcms[i].y0, R.pcms[i].log2_cb_size, R.pcm_samples.data()+R.pcms[i].sample_offset))) return rc;
    return 0;
  }
  sorted(&Rec::tu_ts);
  for (const Ref& o : order) { const Rec& R = recs[o.rec]; const de265hip_tu& t = R.tus[o.idx]; if ((rc = record_tu(rec, &t, R.cval.data()+t.coeff_offset, R.cpos.data()+t.coeff_offset))) return rc; }
  sorted(&Rec::pu_ts);
  for (const Ref& o : order) if ((rc = record_pu(rec, &recs[o.rec].pus[o.idx]))) return rc;
  sorted(&Rec::pcm_ts);
  for (const Ref& o : order) { const Rec& R = recs[o.rec]; const de265hip_pcm& pc = R.pcms[o.idx]; if ((rc = record_pcm(rec, pc.x0, pc.y0, pc.log2_cb_size, R.pcm_samples.data()+pc.sample_offset))) return rc; }
  return 0;
}

int dpb_index_of(const de265_image* img)
{
  decoder_context* ctx = img->decctx;
  for (int i=0; ctx && ctx->has_image(i); i++) if (ctx->get_image(i)==img) return i;
  return 0;
}

/* ---- one picture on its way to the device (offload mode) ---- */
struct Job {
  de265_image* img = nullptr;          // its metadata stays untouched until libde265 re-allocates the DPB entry (pin_release_buffer waits)
  std::vector<Rec> recs;               // the hooks' per-thread buffers of this picture
  uint64_t seq = 0;                    // decode order: pictures are launched on the device in this order
  de265hip_pic_params P;
  std::vector<uint8_t> scaling;
  std::vector<de265hip_slice_params> slices;
  std::vector<de265hip_ctb_info> ctbs;
  PicRec M;
  std::vector<uint8_t> flags; std::vector<int8_t> qp; std::vector<de265hip_motion> mot;
  int slot = 0;
  void* plane[3] = {nullptr,nullptr,nullptr}; ptrdiff_t stride_bytes[3] = {0,0,0};   // the decoder's own picture memory
  de265hip_recorder* rec = nullptr; de265hip_picture* pic = nullptr;
  bool enqueued = false;               // pipeline mode: a worker has put it on the device's streams
};

/* Everything the product's frame-level interface wants to know about a parsed picture, read out of libde265's own picture
 * metadata (image.h).  Runs on the calling thread in dump / synchronous mode and on a worker thread in pipelined mode. */
void prepare_job(Job* job, bool hip)
{
  de265_image* img = job->img;
  const seq_parameter_set& sps = img->get_sps();
  const pic_parameter_set& pps = img->get_pps();
  const double tp0 = now_s();
  if (!hip) merge_records(job->recs, job->M);         // (offload mode: merged straight into the recorder, merge_into_recorder)
  const double tp1 = now_s();
  de265hip_pic_params& P = job->P; memset(&P,0,sizeof(P));
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
  // range extensions: what the sample paths read of sps / pps_range_extension
  P.implicit_rdpcm_enabled_flag = sps.range_extension.implicit_rdpcm_enabled_flag;
  P.transform_skip_rotation_enabled_flag = sps.range_extension.transform_skip_rotation_enabled_flag;
  P.intra_smoothing_disabled_flag = sps.range_extension.intra_smoothing_disabled_flag;
  P.cross_component_prediction_enabled_flag = pps.range_extension.cross_component_prediction_enabled_flag;
  P.extended_precision_processing_flag = sps.range_extension.extended_precision_processing_flag;
  P.high_precision_offsets_enabled_flag = sps.range_extension.high_precision_offsets_enabled_flag;

  const int w4 = (P.width+3)/4, h4 = (P.height+3)/4, nctb = sps.PicSizeInCtbsY;
  const uint8_t* scaling = sps.scaling_list_enable_flag ? (const uint8_t*)&pps.scaling_list : NULL;      // transform.cc:487-493
  std::vector<de265hip_slice_params>& slices = job->slices;
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
  std::vector<de265hip_ctb_info>& ctbs = job->ctbs; ctbs.resize(nctb);
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
  std::vector<uint8_t>& flags = job->flags; std::vector<int8_t>& qp = job->qp; std::vector<de265hip_motion>& mot = job->mot;
  // (offload mode: no motion plane - de265hip_picture_desc::blk_motion = NULL, the device derives it from the PU records;
  //  flattening 518 000 PBMotion records per 4K picture was half of this function)
  flags.resize((size_t)w4*h4); qp.resize((size_t)w4*h4); if (!hip) mot.resize((size_t)w4*h4);
  for (int y=0;y<h4;y++) for (int x=0;x<w4;x++) {
    const int xl = x<<2, yl = y<<2;
    const bool intra = img->get_pred_mode(xl,yl)==MODE_INTRA;
    flags[x+y*w4] = (uint8_t)((intra ? DE265HIP_BLK_INTRA : 0) | (img->get_nonzero_coefficient(xl,yl) ? DE265HIP_BLK_NONZERO : 0) |
                              (img->get_pcm_flag(xl,yl) ? DE265HIP_BLK_PCM : 0) | (img->get_cu_transquant_bypass(xl,yl) ? DE265HIP_BLK_BYPASS : 0));
    qp[x+y*w4] = (int8_t)img->get_QPY(xl,yl);
    if (hip) continue;
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

  const double tp2 = now_s();
  if (hip) {
    if (!P.disable_deblocking) derive_edgeFlags(img);               // cheap host code; or de265hip_derive_edge_flags
    for (int y=0;y<h4;y++) for (int x=0;x<w4;x++) flags[x+y*w4] |= img->get_deblk_flags(x<<2,y<<2) & 0xF0;
    if (scaling) job->scaling.assign(scaling, scaling + DE265HIP_SCALING_BLOB_BYTES);
  }
  static std::mutex pm; std::lock_guard<std::mutex> lk(pm);
  PR.merge += tp1-tp0; PR.flatten += tp2-tp1; PR.edges += now_s()-tp2; PR.n++;
}

/* What an integrated libde265 does at decctx.cc:757-766 instead of run_postprocessing_filters_*: hand the recorded picture
 * to the MI355X.  build_job: the product's recorder + host stage (any thread, several pictures at once: de265_hip.h THREADS);
 * launch_job: kernels + copy-out, in decode order.  wait = true: ... and wait (the picture is in the decoder's planes on
 * return); wait = false: the copy-out into the (pinned) planes is only enqueued; f1_before_output waits. */
bool build_job(Job& j)
{
  int rc;
  const double t0 = now_s();
#define TRY(call, what) do { if ((rc = (call))) { hip_fail(what, rc); return false; } } while (0)
  TRY(H.recorder_new(&j.rec, &j.P, j.scaling.empty() ? NULL : j.scaling.data()), "recorder_new");
  de265hip_recorder* rec = j.rec;
  for (const auto& sl : j.slices) TRY(H.record_slice(rec, &sl), "record_slice");
  for (size_t a=0;a<j.ctbs.size();a++) TRY(H.record_ctb(rec, (int)a, &j.ctbs[a]), "record_ctb");
  TRY(merge_into_recorder(j.recs, rec, H.record_tu, H.record_pu, H.record_pcm), "record_tu / _pu / _pcm");
  TRY(H.record_blk_planes(rec, j.flags.data(), j.qp.data(), j.mot.empty() ? NULL : j.mot.data()), "record_blk_planes");
  const double t1 = now_s();
  {                                                                 // slot allocation belongs to one thread at a time
    static std::mutex am; std::lock_guard<std::mutex> lk(am);
    TRY(H.dpb_alloc(H.dec, j.slot, j.P.width, j.P.height, j.P.bit_depth_luma, j.P.bit_depth_chroma, j.P.chroma_format_idc), "dpb_alloc");
  }
  TRY(H.recorder_submit(H.dec, j.slot, rec, &j.pic), "recorder_submit");
  static std::mutex pm; std::lock_guard<std::mutex> lk(pm);
  PR.record += t1-t0; PR.submit += now_s()-t1;
  return true;
}

bool launch_job(Job& j, bool wait)
{
  int rc;
  const double t2 = now_s();
  TRY(H.picture_run(H.dec, j.pic, DE265HIP_STAGE_FINAL), "picture_run");
  const double t3 = now_s();
  if (wait) {
    TRY(H.decoder_sync(H.dec), "decoder_sync");
    for (int c=0;c<3;c++)                                           // the GPU's picture becomes the decoder's picture
      TRY(H.dpb_download(H.dec, j.slot, c, j.plane[c], j.stride_bytes[c]), "dpb_download");
  } else {
    TRY(H.dpb_download_planes_async(H.dec, j.slot, j.plane, j.stride_bytes, NULL), "dpb_download_planes_async");   // the three planes in one call
  }
#undef TRY
  static std::mutex pm; std::lock_guard<std::mutex> lk(pm);
  PR.run += t3-t2; PR.out += now_s()-t3;
  return true;
}

void free_job(Job& j)
{
  const double t0 = now_s();
  if (j.pic) H.picture_free(j.pic);                                 // never waits: the decoder owns the device side (de265_hip.h LIFETIME)
  if (j.rec) H.recorder_free(j.rec);
  j.pic = NULL; j.rec = NULL;
  j.M = PicRec(); j.flags = std::vector<uint8_t>(); j.qp = std::vector<int8_t>(); j.mot = std::vector<de265hip_motion>();
  static std::mutex pm; std::lock_guard<std::mutex> lk(pm);
  PR.free_ += now_s()-t0;
}

/* SURVEY 8(f3) picture-level pipelining = the product's own pipeline (include/de265_hip.h de265hip_pipeline_*): libde265's
 * thread(s) parse picture n+1 while F1_PIPELINE worker threads of the library call prepare_cb (metadata read-out + de265hip_record_*)
 * and build the pictures before it, and the device reconstructs the pictures before those; launches happen in decode order. */
struct Pend { std::shared_ptr<Job> job; uint64_t ticket; };
std::mutex pend_mu;
std::map<const de265_image*, Pend> pending;            // submitted, not yet known to have landed in the image's planes

int prepare_cb(void* user, de265hip_recorder** out)
{
  Job& j = *(Job*)user;
  prepare_job(&j, true);
  int rc;
  const double t0 = now_s();
  de265hip_recorder* rec = NULL;
  if ((rc = H.recorder_new(&rec, &j.P, j.scaling.empty() ? NULL : j.scaling.data()))) return rc;
  for (const auto& sl : j.slices) if ((rc = H.record_slice(rec, &sl))) return rc;
  for (size_t a=0;a<j.ctbs.size();a++) if ((rc = H.record_ctb(rec, (int)a, &j.ctbs[a]))) return rc;
  if ((rc = merge_into_recorder(j.recs, rec, H.record_tu, H.record_pu, H.record_pcm))) return rc;
  { std::lock_guard<std::mutex> lk(reg_mu); for (Rec& r : j.recs) { r.clear(); spare_recs.push_back(std::move(r)); } j.recs.clear(); }      // (the buffers keep their capacity for the parse threads)
  if ((rc = H.record_blk_planes(rec, j.flags.data(), j.qp.data(), j.mot.empty() ? NULL : j.mot.data()))) return rc;
  j.M = PicRec(); j.flags = std::vector<uint8_t>(); j.qp = std::vector<int8_t>(); j.mot = std::vector<de265hip_motion>();
  *out = rec;
  static std::mutex pm; std::lock_guard<std::mutex> lk(pm);
  PR.record += now_s()-t0;
  return 0;
}

void pipe_submit(const de265_image* img, std::shared_ptr<Job> j)
{
  int rc;
  if (!H.pipe && (rc = H.pipeline_new(&H.pipe, H.dec, H.n_workers))) { hip_fail("pipeline_new", rc); return; }
  uint64_t ticket = 0;
  const double t0 = now_s();
  if ((rc = H.pipeline_submit(H.pipe, j->slot, prepare_cb, j.get(), j->plane, j->stride_bytes, &ticket))) { hip_fail("pipeline_submit", rc); return; }
  PR.w_queue += now_s()-t0;
  std::lock_guard<std::mutex> lk(pend_mu);
  pending[img] = Pend{ j, ticket };
}

void pipe_wait(const de265_image* img)
{
  Pend p;
  {
    std::lock_guard<std::mutex> lk(pend_mu);
    auto it = pending.find(img);
    if (it == pending.end()) return;
    p = it->second;
    pending.erase(it);
  }
  const double t0 = now_s();
  int rc = H.pipeline_wait(H.pipe, p.ticket);
  if (rc) { hip_fail("pipeline_wait", rc); const_cast<de265_image*>(img)->integrity = INTEGRITY_DECODING_ERRORS; }      // (image.h:57-61)
  PR.w_out += now_s()-t0;
}

/* pinned picture memory for libde265 (de265.h:325-343), pooled: libde265 releases and re-requests the planes of a DPB entry
 * for every picture (image.cc:244) */
struct PinPool { std::mutex mu; std::multimap<size_t, void*> free; std::map<void*, size_t> size; };
PinPool PP;

void* pin_get(size_t bytes)
{
  { std::lock_guard<std::mutex> lk(PP.mu); auto it = PP.free.find(bytes); if (it != PP.free.end()) { void* p = it->second; PP.free.erase(it); return p; } }
  void* p = H.host_alloc(bytes);
  if (p) { std::lock_guard<std::mutex> lk(PP.mu); PP.size[p] = bytes; }
  return p;
}

int pin_get_buffer(de265_decoder_context*, de265_image_spec* spec, de265_image* img, void*)
{                                                                   // geometry as the default allocator, image.cc:106-156
  const int cw = spec->width / img->SubWidthC, ch = spec->height / img->SubHeightC;
  const int ls = (spec->width + spec->alignment-1) / spec->alignment * spec->alignment, cs = (cw + spec->alignment-1) / spec->alignment * spec->alignment;
  void* y = pin_get((size_t)spec->height * ls * ((img->BitDepth_Y+7)/8) + 64);
  void* u = pin_get((size_t)ch * cs * ((img->BitDepth_C+7)/8) + 64);
  void* v = pin_get((size_t)ch * cs * ((img->BitDepth_C+7)/8) + 64);
  if (!y || !u || !v) return 0;
  img->set_image_plane(0, (uint8_t*)y, ls, NULL); img->set_image_plane(1, (uint8_t*)u, cs, NULL); img->set_image_plane(2, (uint8_t*)v, cs, NULL);
  return 1;
}

void pin_release_buffer(de265_decoder_context*, de265_image* img, void*)
{
  pipe_wait(img);                                                   // a copy-out still on its way into these planes
  std::lock_guard<std::mutex> lk(PP.mu);
  for (int c=0;c<3;c++) { void* p = img->get_image_plane(c); auto it = PP.size.find(p); if (p && it != PP.size.end()) PP.free.insert({it->second, p}); }
}

} // namespace

void f1_install_pinned_allocator(void* ctx)
{
  if (!hip_mode()) return;
  static de265_image_allocation a = { pin_get_buffer, pin_release_buffer };
  de265_set_image_allocation_functions(ctx, &a, NULL);
}

bool f1_offloading() { return hip_mode(); }

void f1_before_output(const de265_image* img) { if (H.on && H.pipeline) pipe_wait(img); }

int f1_drain()
{
  if (!(H.on && H.pipeline)) { prof_print(); return H.on ? hip_failed() : 0; }
  if (H.pipe) {
    int rc = H.pipeline_drain(H.pipe);
    if (rc) hip_fail("pipeline_drain", rc);
    H.pipeline_free(H.pipe);                             // joins the workers before the process tears its statics down
    H.pipe = nullptr;
  }
  { std::lock_guard<std::mutex> lk(pend_mu); pending.clear(); }
  prof_print();
  return hip_failed();
}

/* decctx.cc:1408-1434 generate_unavailable_reference_picture: libde265 has just filled DPB entry `idx` with mid-grey on the host
 * (a reference picture the stream does not contain: a lost picture, a stream joined at a CRA picture).  The pictures that
 * predict from it are reconstructed on the device: the same picture goes into the device-resident DPB (the slot IS the
 * index).  Pictures still on their way through the pipeline may read what the slot held before: they go first. */
void f1_unavailable_reference(de265_image* img, int idx)
{
  if (passive() || !hip_mode() || hip_failed()) return;
  if (idx < 0 || idx >= DE265HIP_MAX_DPB_SLOTS) { hip_fail("libde265 holds more pictures than the back end has DPB slots", idx); return; }
  int rc;
  if (H.pipe && (rc = H.pipeline_drain(H.pipe))) { hip_fail("pipeline_drain", rc); return; }
  const seq_parameter_set& sps = img->get_sps();
  if ((rc = H.dpb_alloc(H.dec, idx, img->get_width(0), img->get_height(0), sps.BitDepth_Y, sps.BitDepth_C, sps.chroma_format_idc))) { hip_fail("dpb_alloc", rc); return; }
  if (sps.BitDepth_Y <= 8 && sps.BitDepth_C <= 8) {
    if ((rc = H.dpb_fill(H.dec, idx, 1 << (sps.BitDepth_Y - 1), 1 << (sps.BitDepth_C - 1), 1 << (sps.BitDepth_C - 1)))) hip_fail("dpb_fill", rc);
    return;
  }
  // Samples wider than 8 bits: the reference's fill_image is a BYTE memset over stride x height bytes (image.cc:510-523) - it
  // writes the low byte of the value (0 for 1 << 8 and up) into the first half of each plane and leaves the second half as
  // the allocator handed it out.  What libde265 predicts from is that picture, not the one 8.3.3.2 describes: it is
  // mirrored as it is.
  for (int c = 0; c < (sps.chroma_format_idc ? 3 : 1); c++)
    if ((rc = H.dpb_upload(H.dec, idx, c, img->get_image_plane(c), (ptrdiff_t)img->get_image_stride(c) * img->get_bytes_per_pixel(c)))) { hip_fail("dpb_upload", rc); return; }
}

bool f1_record_tu(thread_context* tctx, int x0, int y0, int nT, int cIdx, int cuPredMode, bool cbf)
{
  if (passive()) return false;
  const bool intra = cuPredMode == MODE_INTRA;
  const int rsv = cIdx ? tctx->ResScaleVal : 0;                   // cross-component prediction acts on a chroma TU without coefficients too (slice.cc:3478-3487)
  if (!intra && !cbf && !rsv) return hip_mode();                  // decode_TU does nothing for it (slice.cc:3424-3488)
  de265_image* img = tctx->img;
  const seq_parameter_set& sps = img->get_sps();
  de265hip_tu t; memset(&t,0,sizeof(t));
  t.x0 = (uint16_t)x0; t.y0 = (uint16_t)y0; t.c_idx = (uint8_t)cIdx;
  int l2 = 0; while ((1<<l2) < nT) l2++;
  t.log2_size = (uint8_t)l2;
  t.flags = (uint8_t)((intra ? DE265HIP_TU_INTRA : 0) | (cbf ? DE265HIP_TU_CBF : 0) |
                      (tctx->transform_skip_flag[cIdx] && cbf ? DE265HIP_TU_TSKIP : 0) |
                      (tctx->cu_transquant_bypass_flag ? DE265HIP_TU_BYPASS : 0) |
                      (!intra && cbf && tctx->explicit_rdpcm_flag ? DE265HIP_TU_EXPLICIT_RDPCM | (tctx->explicit_rdpcm_dir ? DE265HIP_TU_EXPLICIT_RDPCM_VERT : 0) : 0));
  t.res_scale_val = (int8_t)rsv;
  if (intra) {                                         // slice.cc:3436-3451
    int m = cIdx==0 ? img->get_IntraPredMode(x0,y0) : img->get_IntraPredModeC(x0*sps.SubWidthC, y0*sps.SubHeightC);
    if (m<0 || m>=35) m = INTRA_DC;
    t.intra_mode = (uint8_t)m;
  }
  t.qp = (int8_t)(cIdx==0 ? tctx->qPYPrime : (cIdx==1 ? tctx->qPCbPrime : tctx->qPCrPrime));   // transform.cc:362-368
  Rec& R = mine();
  if (cbf) {
    t.n_coeff = (uint16_t)tctx->nCoeff[cIdx];
    t.coeff_offset = (uint32_t)R.cval.size();
    for (int i=0;i<tctx->nCoeff[cIdx];i++) {
      R.cval.push_back(tctx->coeffList[cIdx][i]);
      R.cpos.push_back((uint16_t)tctx->coeffPos[cIdx][i]);
    }
  }
  R.tus.push_back(t); R.tu_ts.push_back((uint32_t)tctx->CtbAddrInTS);
  return hip_mode();
}

bool f1_record_pu(const slice_segment_header* shdr, de265_image* img, int xP, int yP, int nPbW, int nPbH, const PBMotion* vi)
{
  if (passive()) return false;
  de265hip_pu p; memset(&p,0,sizeof(p));
  p.x = (uint16_t)xP; p.y = (uint16_t)yP; p.w = (uint8_t)nPbW; p.h = (uint8_t)nPbH;
  p.pred_flag = (uint8_t)((vi->predFlag[0]?1:0) | (vi->predFlag[1]?2:0));
  p.slice_idx = (uint16_t)shdr->slice_index;
  for (int l=0;l<2;l++)                                 // libde265 leaves the unused list's refIdx / mv undefined: record zeros
    if (vi->predFlag[l]) { p.ref_idx[l] = vi->refIdx[l]; p.mv[l][0] = vi->mv[l].x; p.mv[l][1] = vi->mv[l].y; }
  const seq_parameter_set& sps = img->get_sps();
  Rec& R = mine();
  R.pus.push_back(p);
  R.pu_ts.push_back((uint32_t)img->get_pps().CtbAddrRStoTS[(xP>>sps.Log2CtbSizeY) + (yP>>sps.Log2CtbSizeY)*sps.PicWidthInCtbsY]);
  return hip_mode();
}

void f1_record_pcm(thread_context* tctx, int x0, int y0, int log2CbSize)
{
  if (passive()) return;
  de265_image* img = tctx->img;
  de265hip_pcm p; memset(&p,0,sizeof(p));
  p.x0 = (uint16_t)x0; p.y0 = (uint16_t)y0; p.log2_cb_size = (uint8_t)log2CbSize;
  Rec& R = mine();
  p.sample_offset = (uint32_t)R.pcm_samples.size();
  for (int c=0;c<3;c++) {                              // the samples as read_pcm_samples_internal stored them (already << shift)
    const seq_parameter_set& sps = img->get_sps();
    const int sw = c ? sps.SubWidthC : 1, sh = c ? sps.SubHeightC : 1;
    const int n = (1<<log2CbSize) / sw, nh = (1<<log2CbSize) / sh, xx = x0 / sw, yy = y0 / sh, stride = img->get_image_stride(c);
    for (int y=0;y<nh;y++) for (int x=0;x<n;x++)
      R.pcm_samples.push_back(img->high_bit_depth(c) ? ((const uint16_t*)img->get_image_plane(c))[xx+x+(yy+y)*stride]
                                                      : img->get_image_plane(c)[xx+x+(yy+y)*stride]);
  }
  R.pcms.push_back(p); R.pcm_ts.push_back((uint32_t)tctx->CtbAddrInTS);
}

bool f1_submit(de265_image* img)
{
  if (passive()) return false;
  const bool hip = hip_mode();
  const double ts0 = now_s();
  if (!PR.t_first) PR.t_first = ts0;
  PR.t_last = ts0;
  struct InSubmit { double t0; ~InSubmit() { PR.in_submit += now_s()-t0; } } in_submit{ts0};
  std::shared_ptr<Job> job = std::make_shared<Job>();
  job->img = img;
  take_records(job->recs);
  if (hip) {
    // ---- OFFLOAD
    if (hip_failed()) { img->integrity = INTEGRITY_DECODING_ERRORS; return true; }      // (an earlier picture failed: de265_decode is about to say so)
    job->slot = dpb_index_of(img);                                  // reference lists name libde265's DPB indices: the slot IS the index
    if (job->slot >= DE265HIP_MAX_DPB_SLOTS) { hip_fail("libde265 holds more pictures than the back end has DPB slots", job->slot); return true; }
    for (int c=0;c<3;c++) { job->plane[c] = img->get_image_plane(c); job->stride_bytes[c] = (ptrdiff_t)img->get_image_stride(c)*img->get_bytes_per_pixel(c); }
    if (H.pipeline) pipe_submit(img, job);                          // the workers prepare, build and launch; libde265 goes on parsing
    else {
      prepare_job(job.get(), true);
      if (!(build_job(*job) && launch_job(*job, true))) img->integrity = INTEGRITY_DECODING_ERRORS;
      free_job(*job);
    }
    S.n_pictures++;
    return true;
  }

  prepare_job(job.get(), false);
  const seq_parameter_set& sps = img->get_sps();
  const PicRec& M = job->M;
  const de265hip_pic_params& P = job->P;
  const std::vector<de265hip_slice_params>& slices = job->slices;
  const std::vector<de265hip_ctb_info>& ctbs = job->ctbs;
  const std::vector<uint8_t>& flags = job->flags; const std::vector<int8_t>& qp = job->qp; const std::vector<de265hip_motion>& mot = job->mot;
  const int w4 = (P.width+3)/4, h4 = (P.height+3)/4, nctb = sps.PicSizeInCtbsY;
  const int cbw = sps.PicWidthInMinCbsY, cbh = sps.PicHeightInMinCbsY, tbw = sps.PicWidthInTbsY, tbh = sps.PicHeightInTbsY;
  const uint8_t* scaling = sps.scaling_list_enable_flag ? (const uint8_t*)&img->get_pps().scaling_list : NULL;
  // ---- DUMP: the description + the reference's own pictures to $F1_OUT/pic_NNN.f1
  S.file.clear();
  put("F1DESC02", 8);
  put(&P,1);
  put_i32((int)slices.size()); put_i32(nctb); put_i32((int)M.tus.size()); put_i32((int)M.cval.size());
  put_i32((int)M.pus.size()); put_i32((int)M.pcms.size()); put_i32((int)M.pcm_samples.size());
  put_i32(w4); put_i32(h4); put_i32(cbw*cbh); put_i32(tbw*tbh);
  put_i32(dpb_index_of(img)); put_i32(img->PicOrderCntVal); put_i32(scaling ? 1 : 0);
  if (scaling) put(scaling, DE265HIP_SCALING_BLOB_BYTES);
  put(slices.data(), slices.size()); put(ctbs.data(), ctbs.size());
  put(M.tus.data(), M.tus.size()); put(M.cval.data(), M.cval.size()); put(M.cpos.data(), M.cpos.size());
  put(M.pus.data(), M.pus.size()); put(M.pcms.data(), M.pcms.size()); put(M.pcm_samples.data(), M.pcm_samples.size());
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
  return false;
}

int f1_picture_done(de265_image* img)
{
  if (passive()) return 0;
  if (hip_mode()) {                                    // libde265 checks the picture hash SEI right after this hook (decctx.cc:768-778)
    if (H.pipeline && img->decctx->param_sei_check_hash && !hip_failed()) pipe_wait(img);
    return hip_failed();
  }
  const int w4 = (img->get_width(0)+3)/4, h4 = (img->get_height(0)+3)/4;
  std::vector<uint8_t> edges((size_t)w4*h4);
  for (int y=0;y<h4;y++) for (int x=0;x<w4;x++) edges[x+y*w4] = img->get_deblk_flags(x<<2,y<<2) & 0xF0;   // as derive_edgeFlags marked them
  put(edges.data(), edges.size());
  put_planes(img);                                     // ... and after deblocking + SAO
  const char* dir = getenv("F1_OUT");
  if (!dir) return 0;                                  // passive()
  char name[1024];
  snprintf(name, sizeof(name), "%s/pic_%03d.f1", dir, S.n_pictures++);
  FILE* f = fopen(name, "wb");
  if (!f || fwrite(S.file.data(), 1, S.file.size(), f) != S.file.size()) { fprintf(stderr, "f1_recorder: cannot write %s\n", name); exit(5); }
  fclose(f);
  S.file.clear();
  return 0;
}
