/*
 * f1_hooks.h -- the recorder hooks of SURVEY.md 8(f1), declared for the PATCHED scratch copy of the reference that
 * oracle/Makefile `make f1` builds in oracle/_ref/f1/ (oracle/f1_recorder.patch adds one call at each hook site):
 *   libde265/slice.cc:3424    decode_TU                          -> f1_record_tu
 *   libde265/motion.cc:279    generate_inter_prediction_samples  -> f1_record_pu
 *   libde265/slice.cc:4185    read_pcm_samples                   -> f1_record_pcm
 *   libde265/decctx.cc:757    decode_some, after mark_all_CTB_progress(PREFILTER)  -> f1_submit  (what the product's
 *                             de265hip_recorder_submit would receive), and behind the post-filters -> f1_picture_done
 * TEST INFRASTRUCTURE (build container only): the hooks dump what the product's frame-level interface consumes
 * (de265hip_picture_desc) for real bitstreams, next to the reference's own decoded output.
 */
#ifndef F1_HOOKS_H
#define F1_HOOKS_H
class thread_context;
class slice_segment_header;
struct de265_image;
class PBMotion;

void f1_record_tu(thread_context* tctx, int x0, int y0, int nT, int cIdx, int cuPredMode, bool cbf);
void f1_record_pu(const slice_segment_header* shdr, de265_image* img, int xP, int yP, int nPbW, int nPbH, const PBMotion* vi);
void f1_record_pcm(thread_context* tctx, int x0, int y0, int log2CbSize);
void f1_submit(de265_image* img);
void f1_picture_done(de265_image* img);
#endif
