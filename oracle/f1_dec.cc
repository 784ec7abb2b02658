/*
 * f1_dec.cc -- minimal driver of the PATCHED reference (oracle/Makefile `make f1`): pushes a bitstream through
 * libde265's public API (de265.h: de265_new_decoder, de265_push_data, de265_decode, de265_get_next_picture; scalar
 * DSP path, no worker threads = decode_slice_unit_sequential + run_postprocessing_filters_sequential) so that the
 * hooks of oracle/f1_recorder.cc see every picture.
 *   F1_OUT=dir f1_dec stream.bin                          dump mode: per-picture description + the reference's pictures
 *   F1_MODE=hip F1_HIP_LIB=.../libde265_hip.so f1_dec stream.bin out.yuv
 *                                                         offload mode: libde265 parses, the MI355X reconstructs; out.yuv receives
 *                                                         what de265_get_next_picture hands out (planes, little-endian samples)
 *   F1_THREADS=n        libde265's own worker threads (de265_start_worker_threads: WPP rows / tiles parse in parallel, decctx.cc:976-1178)
 *   F1_PIPELINE=1       (offload mode) SURVEY 8(f3): pictures are only ENQUEUED on the device, pinned picture memory, copy-out waited
 *                       for at output time - libde265 parses picture n+1 while the MI355X reconstructs picture n
 *   F1_CHECK_HASH=0     do not verify decoded-picture-hash SEIs (the check reads every picture on the host right after its decode)
 *   F1_TIMING=1         print pictures/s of the decode loop (file read and output writing included) on stderr... as the last stdout line
 */
#include "libde265/de265.h"
#include "f1_hooks.h"
#include <chrono>
#include <stdio.h>
#include <stdlib.h>

static void write_picture(FILE* out, const de265_image* im)
{
  if (!out) return;
  for (int c = 0; c < 3; c++) {
    int stride = 0;
    const uint8_t* p = de265_get_image_plane(im, c, &stride);
    const int w = de265_get_image_width(im, c), h = de265_get_image_height(im, c), bpp = (de265_get_bits_per_pixel(im, c) + 7) / 8;
    for (int y = 0; y < h; y++) fwrite(p + (size_t)y * stride, 1, (size_t)w * bpp, out);
  }
}

int main(int argc, char** argv)
{
  if (argc < 2) { fprintf(stderr, "usage: F1_OUT=dir %s stream.bin\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  FILE* out = argc > 2 ? fopen(argv[2], "wb") : NULL;
  de265_decoder_context* ctx = de265_new_decoder();
  de265_set_parameter_int(ctx, DE265_DECODER_PARAM_ACCELERATION_CODE, de265_acceleration_SCALAR);
  const char* e;
  de265_set_parameter_bool(ctx, DE265_DECODER_PARAM_BOOL_SEI_CHECK_HASH, (e = getenv("F1_CHECK_HASH")) ? atoi(e) : 1);
  if ((e = getenv("F1_THREADS")) && atoi(e) > 0 && de265_start_worker_threads(ctx, atoi(e)) != DE265_OK) return 3;
  f1_install_pinned_allocator(ctx);                        // offload mode only
  const auto t0 = std::chrono::steady_clock::now();
  unsigned char buf[65536];
  int n_out = 0, more = 1, failed = 0;
  size_t n;
  // a decode error (the reference's own, or - offload mode - the back end's, which de265_decode hands through) ends the loop;
  // the decoder is still drained and freed, the exit code says that the decode failed
  while (!failed && (n = fread(buf, 1, sizeof(buf), f)) > 0) {
    if (de265_push_data(ctx, buf, (int)n, 0, NULL) != DE265_OK) { failed = 3; break; }
    for (;;) {
      de265_error e = de265_decode(ctx, &more);
      if (e != DE265_OK && e != DE265_ERROR_WAITING_FOR_INPUT_DATA) { fprintf(stderr, "decode error %d: %s\n", (int)e, de265_get_error_text(e)); failed = 4; break; }
      while (const de265_image* im = de265_get_next_picture(ctx)) { n_out++; write_picture(out, im); }
      if (e == DE265_ERROR_WAITING_FOR_INPUT_DATA || !more) break;
    }
  }
  if (!failed) {
    de265_flush_data(ctx);
    more = 1;
    while (more) {
      de265_error e = de265_decode(ctx, &more);
      if (e != DE265_OK && e != DE265_ERROR_WAITING_FOR_INPUT_DATA) { fprintf(stderr, "decode error %d: %s\n", (int)e, de265_get_error_text(e)); failed = 4; break; }
      while (const de265_image* im = de265_get_next_picture(ctx)) { n_out++; write_picture(out, im); }
      if (e != DE265_OK) break;
    }
  }
  if (const int frc = f1_drain()) { if (!failed) fprintf(stderr, "decode error %d (reported when the pipeline was drained)\n", frc); failed = 4; }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (;;) {
    de265_error w = de265_get_warning(ctx);
    if (w == DE265_OK) break;
    fprintf(stderr, "warning: %s\n", de265_get_error_text(w));
  }
  de265_free_decoder(ctx);
  fclose(f);
  if (out) fclose(out);
  printf("%d pictures\n", n_out);
  if ((e = getenv("F1_TIMING")) && atoi(e)) printf("%.3f s  %.2f pictures/s\n", secs, n_out / secs);
  return failed;
}
