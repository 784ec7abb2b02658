/*
 * f1_hooks.h -- the recorder hooks of SURVEY.md 8(f1), declared for the PATCHED scratch copy of the reference that
 * oracle/Makefile `make f1` builds in oracle/_ref/f1/ (oracle/f1_recorder.patch adds one call at each hook site):
 *   libde265/slice.cc:3424    decode_TU                          -> f1_record_tu
 *   libde265/motion.cc:279    generate_inter_prediction_samples  -> f1_record_pu
 *   libde265/slice.cc:4185    read_pcm_samples                   -> f1_record_pcm
 *   libde265/decctx.cc:757    decode_some, after mark_all_CTB_progress(PREFILTER)  -> f1_submit  (what the product's
 *                             de265hip_recorder_submit would receive), and behind the post-filters -> f1_picture_done
 * TEST INFRASTRUCTURE.  Two modes (oracle/f1_recorder.cc): dump (default) - the hooks write what the product's frame-level
 * interface consumes (de265hip_picture_desc) for real bitstreams, next to the reference's own decoded output; F1_MODE=hip -
 * the hooks feed libde265_hip.so (de265hip_record_* / recorder_submit / picture_run) INSTEAD of the CPU reconstruction:
 * libde265 parses, the MI355X reconstructs, de265_get_next_picture hands out the GPU's pictures.
 */
#ifndef F1_HOOKS_H
#define F1_HOOKS_H
class thread_context;
class slice_segment_header;
struct de265_image;
class PBMotion;

/* The three recording hooks return true when the picture is being OFFLOADED (F1_MODE=hip: the caller then skips its own
 * reconstruction of that block), false when they only record next to the reference's own reconstruction (dump mode). */
bool f1_record_tu(thread_context* tctx, int x0, int y0, int nT, int cIdx, int cuPredMode, bool cbf);
bool f1_record_pu(const slice_segment_header* shdr, de265_image* img, int xP, int yP, int nPbW, int nPbH, const PBMotion* vi);
void f1_record_pcm(thread_context* tctx, int x0, int y0, int log2CbSize);
/* true: the picture was reconstructed, deblocked and SAO-filtered by the HIP back end and copied into img's planes (the
 * caller skips run_postprocessing_filters_*); false: dump mode, the caller carries on */
bool f1_submit(de265_image* img);
/* returns 0, or - offload mode - the error of the back end (a de265_error number, de265.h:82-139): decode_some returns it, so
 * de265_decode does (errors reach the application through the decoder's own channel, not through exit) */
int f1_picture_done(de265_image* img);
/*   libde265/decctx.cc:1408  generate_unavailable_reference_picture -> f1_unavailable_reference: the grey picture libde265
 *                             synthesises for a reference the stream does not contain goes into the device-resident DPB too */
void f1_unavailable_reference(de265_image* img, int idx);
/* SURVEY 8(f3), F1_MODE=hip F1_PIPELINE=1: f1_submit only ENQUEUES the picture (a submit thread builds the command buffers
 * and launches; the copy-out into libde265's pinned planes is asynchronous) and libde265 carries on parsing; the picture is
 * waited for where somebody is about to look at it: de265_peek_next_picture (de265.cc:392) -> f1_before_output. */
void f1_before_output(const de265_image* img);
/* decctx.cc:1999: with SAO on, libde265 decodes into an internal picture and lets its SAO write the output picture; when the
 * device runs SAO the decoded picture IS the output picture (and is allocated with the application's allocator) */
bool f1_offloading();
/* for the application (oracle/f1_dec.cc): install the pinned-memory image allocator (de265.h:325-343) before decoding, and
 * drain the pipeline before the decoder is freed */
void f1_install_pinned_allocator(void* de265_decoder_ctx);
int f1_drain();                        /* 0, or the back end's error (see f1_picture_done) */
#endif
