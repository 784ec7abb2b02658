// host.hip -- host side of the MI355X reconstruction back end: C ABI of
// include/de265_hip.h, device-resident DPB, per-picture command-buffer build
// (intra availability + dependency levels, MC tile split) and kernel sequencing.
//
// Order of device work for one picture (last paragraph):
//   MC (all PUs) -> PCM copy -> TU level 0 (inter residual) -> TU levels 1..N
//   (intra prediction + residual, dependency-ordered) -> bS -> deblock V ->
//   deblock H -> SAO (out of place, then the slot's plane pointers are swapped).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <chrono>

#include <deque>
#include <array>
#include <map>

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include "kernels.h"
#include "env.h"
#include "scan.h"

using namespace d265;

#define HIPCHK(expr, code)                                                      \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "de265hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return (code);                                                            \
    }                                                                           \
  } while (0)

namespace {

struct Slot {
  PlaneRef pl[3] = {};
  int w = 0, h = 0, bdY = 0, bdC = 0, cf = 1;      // cf: chroma_format_idc (0: monochrome, empty chroma planes)
  bool valid = false;
  int cw() const { return cf == 0 ? 0 : cf == 3 ? w : w / 2; }      // chroma plane size (SubWidthC / SubHeightC, sps.cc:540-552)
  int ch() const { return cf == 0 ? 0 : cf == 1 ? h / 2 : h; }
  // de265hip_dpb_download_async: recorded on the output stream behind the slot's latest copy-out; whoever writes the
  // slot next (a new picture, an upload) or frees it waits for it.  dl_seq counts the copy-outs (dpb_wait compares it).
  hipEvent_t dl_done = nullptr;                    // (= the event of the latest copy-out, one of dl_ev)
  uint64_t dl_seq = 0, dl_waited = 0;
  // the last four copy-outs of the slot, each with its own event and the error-ring entry of the picture it delivers: a
  // pipeline's ticket is waited for long after the slot's NEXT picture has been copied out (de265hip_dpb_wait_copy_out)
  // deferred copy-outs (OutThread below): how many of the slot's copy-outs have been asked for / handed to the runtime.  Whoever
  // is about to use dl_done - wait for it, make a stream wait for it - first waits until the two are equal (out_settle).
  uint64_t out_queued = 0, out_issued = 0;         // (guarded by g_out_mu)
  static constexpr int kDlRing = 4;
  hipEvent_t dl_ev[kDlRing] = {};
  uint64_t dl_ev_seq[kDlRing] = {};
  int dl_ev_err_idx[kDlRing] = { -1, -1, -1, -1 }; uint64_t dl_ev_err_seq[kDlRing] = {};
  int err_idx = -1; uint64_t err_seq = 0;          // error-ring entry of the picture last decoded into the slot
  // Lanes (de265hip_decoder_set_lanes > 1): who wrote the slot's picture and who reads it, as events on the lanes' streams.
  // written: recorded behind the picture that was decoded into the slot (writer_lane >= 0) or behind a copy into it from
  // another decoder (kForeignWriter: every lane waits for it); read_done[l]: behind the latest picture of lane l that
  // references the slot's present content.  A writer waits for all of them, a reader for `written`.
  hipEvent_t written = nullptr;
  int writer_lane = -1;                            // -1: nobody the device has to wait for (empty, or written by a synchronous upload)
  uint64_t written_seq = 0;                        // launch number of the writing picture (lane choice)
  hipEvent_t read_done[4] = { nullptr, nullptr, nullptr, nullptr };
  bool rd_valid[4] = { false, false, false, false };
};
constexpr int kMaxLanes = 4;
constexpr int kForeignWriter = -2;

struct PendingEvent { int kid; hipEvent_t a, b; };

// Pooled device arena of one picture's command buffers.  last_use: recorded on the decoder's stream when the
// picture that used the arena is freed; the next upload into it waits for that event on the copy stream (no
// host-side synchronisation, no hipMalloc / hipFree per picture -- both stall every stream of the process).
struct ArenaBuf { void* ptr = nullptr; size_t bytes = 0; hipEvent_t last_use = nullptr; bool used = false;
                  int64_t epoch = -1;          // epoch: the decoder's generation epoch the arena was last cleared in (-1: never)
                  uint64_t release_seq = 0; }; // the decoder's count of released arenas when this one came back to the pool
// Pinned staging buffer the host stage assembles the command buffers in.  copied: recorded on the copy stream
// behind the upload; the buffer is handed out again once it has completed.
struct StageBuf { void* ptr = nullptr; size_t bytes = 0; hipEvent_t copied = nullptr; int state = 0; uint64_t owner = 0;
                  uint64_t sig = 0; };      // sig: (state 2) the completion signal of an upload issued through the HSA runtime (0: `copied` tells)   // state: 0 idle, 1 being filled by a host thread, 2 upload in flight

}  // namespace

struct de265hip_decoder {
  int device = 0;
  bool dry = false;                   // de265hip_debug_build_host_only: the host stage without any HIP call (profiling on a CPU box)
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // uploads of command buffers (de265hip_picture_build), overlapping the kernels of earlier pictures
  // Round 4: a picture's upload is followed by the kernels of its scan (k_scan.hip), latency chains of ~0.5 ms that hardly
  // occupy the device: the builds of one decoder take turns on a few copy streams, so that the upload and the scan of
  // consecutive pictures overlap (DE265HIP_COPY_STREAMS, default 3; [0] is copy_stream)
  static constexpr int kMaxCopyStreams = 8;
  hipStream_t copy_streams[kMaxCopyStreams] = {};
  int n_copy_streams = 1;
  int prio_low = 0, prio_high = 0;                  // stream priorities (de265hip_decoder_new)
  uint64_t copy_turn = 0;
  // ... and the uploads of a pipeline's builds go out from the worker that staged them, on upload streams of their own:
  // hipMemcpyAsync of a few megabytes of pinned memory holds its caller for about as long as the copy takes (140 us, at times
  // milliseconds): issued by the launcher it was most of what the launcher did
  hipStream_t upload_streams[2] = {};
  uint64_t upload_turn = 0;
  hipStream_t out_stream = nullptr;   // de265hip_dpb_download_async: decoded pictures leave on their own stream, behind an event of `stream`
  bool out_pooled = false;            // ... one of the process's two (DeviceStreams)
  hipEvent_t out_fence = nullptr;
  // deferred copy-outs: jobs for the output thread (guarded by g_out_mu), the events their pictures are waited for by
  struct OutJobRec { Slot* s; uint64_t id; int ring; hipEvent_t picture_done; void* dst[3]; const void* src[3]; size_t dpitch[3], spitch[3], row_bytes[3], rows[3]; };
  std::deque<OutJobRec> out_jobs;
  std::thread out_thread; bool out_thread_started = false, out_stop = false, out_failed = false;
  std::vector<hipEvent_t> out_events;           // (guarded by mu) blocking-sync events, re-used
  std::vector<uint64_t> free_sigs;              // (guarded by mu) completion signals of uploads through the HSA runtime, re-used
  std::mutex mu;                      // guards live, the two pools and slot allocation: build()/free() may come from several host threads
  std::vector<de265hip_picture*> live;        // pictures built on this decoder and not yet freed (decoder_free orphans them)
  std::vector<ArenaBuf> free_arenas;
  std::deque<ArenaBuf> cooling;                 // released, their last user's kernels possibly still running (reap_arenas)
  uint64_t arena_releases = 0;                  // arenas handed back so far (acquire_arena: how long ago was this one released?)
  std::vector<StageBuf> stage_pool;
  std::vector<hipEvent_t> free_events;          // `uploaded` events of freed pictures (no create / destroy per picture)
  size_t pooled_bytes = 0;
  Slot slots[DE265HIP_MAX_DPB_SLOTS];
  Slot spare;                         // SAO output target, swapped with the decoded slot (lane 0's)
  // Lanes: pictures of ONE decoder that do not depend on each other run on different HIP streams (hierarchical-B pictures of
  // one layer, the next closed GOP's first pictures); the reference decodes its pictures strictly one after the other
  // (decctx.cc:904-910).  Lane 0 is `stream` / `spare`; lanes 1.. have a stream and a spare picture of their own.
  // Launch order stays decode order; what orders the device work are the slots' events (Slot::written / read_done).
  int n_lanes = 1;
  hipStream_t lane_stream[kMaxLanes] = { nullptr, nullptr, nullptr, nullptr };
  bool lane_pooled[kMaxLanes] = { false, false, false, false };
  int kstream_index = -1;             // which of the process's pooled kernel streams `stream` is (-1: the decoder's own)
  bool streams_pooled = false;        // copy and upload streams from the process's pools as well
  Slot lane_spare[kMaxLanes];         // [0] unused (lane 0's is `spare`)
  hipEvent_t lane_fence[kMaxLanes] = { nullptr, nullptr, nullptr, nullptr };   // scratch events ("everything lane l has queued so far")
  uint64_t lane_tail_seq[kMaxLanes] = { 0, 0, 0, 0 };   // launch number of the lane's latest picture
  uint64_t launch_seq = 0;
  hipStream_t cur_stream = nullptr;   // the stream of the picture being launched (KTimer)
  uint32_t* d_err = nullptr;          // diagnostic builds only (phase stamps of k_run)
  // Per-picture error words (a k_run dependency wait that expired, a coefficient position beyond its block): a ring of device
  // words, and beside it a ring of pinned host records the device-side scan of a picture reports into (ScanCounts).  A picture
  // takes an entry at build and hands it back when it is freed; entries are re-used first in, first out, so the word of a
  // picture that was freed right behind its launch (the pipeline does) still stands when its ticket is waited for.
  static constexpr int kRing = 4096;
  uint32_t* d_err_ring = nullptr;
  ScanCounts* h_ring = nullptr;
  std::deque<int> ring_free;
  uint64_t ring_owner[kRing] = {};    // sequence number of the picture that holds / last held the entry
  uint64_t ring_seq = 0;
  std::vector<std::pair<int, uint64_t>> launched_err;      // (entry, owner) of the pictures launched since the last de265hip_decoder_sync
  // k_run's flags and mailbox packets carry the generation number of their launch; the decoder never hands a number out twice
  // (arena_epoch counts the wrap-arounds: an arena of an older epoch is cleared before its next use)
  uint32_t gen_tag = 0;
  int64_t arena_epoch = 0;
  bool dev_scan = true;               // the TU scan as kernels behind the upload (scan_core.h); DE265HIP_HOST_SCAN=1: the round-3 host scan
  bool dry_scan = false;              // dry decoders only: run the passes on the host (the CPU rehearsal of tests/test_scan_equivalence.py)
  uint64_t* d_used_units = nullptr;   // g_used_units on the device
  bool drop_producer = false;         // fault injection (de265hip_debug_fault_injection; tests only)
  int building = 0;                   // de265hip_picture_build calls in progress (their pictures are not in `live` yet)
  double t_scan_wait = 0, t_run = 0, t_enq = 0; long n_scan_wait = 0;
  double t_sec[8] = {}, t_sec_max[8] = {}; long n_sec = 0;      // DE265HIP_PIPE_TIMING: sections of de265hip_picture_enqueue_batch      // DE265HIP_PIPE_TIMING: seconds the launches waited for scans / spent enqueueing
  int run_waves = RUN_WAVES;          // DE265HIP_RUN_WAVES: wavefronts per run workgroup (experiments)
  bool resid_one_launch = true;       // DE265HIP_RESID_ONE_LAUNCH=0: 8x8 / 4x4 residual TUs in their own launch (k_resid_small)
  bool resid16_big = false;           // DE265HIP_RESID16_BIG: 16x16 residual TUs by 4-wavefront workgroups (k_resid_big) instead of one wavefront
  bool two_pass_deblock = false;      // DE265HIP_TWO_PASS_DEBLOCK: k_deblock<V> then k_deblock<H> instead of k_deblock_fused
  bool lf_tile = false;               // DE265HIP_LF_TILE=1: the one-pass k_lf_tile instead of k_deblock_fused + k_sao.  Measured (4K Main10): 45-47 us
                                      // against 22 + 27.5 us alone, but no gain in the bench (3 streams 6 320 vs 6 400 frames/s, 1 stream equal):
                                      // both filters are bound by VALU issue (811 / 377 VALU instructions per wavefront, tools/exp/pmc_insts.sh),
                                      // not by the passes over memory the fusion removes, and the tile kernel needs 137 VGPRs + 20 KB of LDS
  bool sao_strips = false;            // DE265HIP_SAO_STRIPS=1: k_sao (a wavefront along one row pair) instead of the CTB-local k_sao_ctb
  bool separate_bs = false;           // DE265HIP_SEPARATE_BS: bS by its own kernel instead of inside the deblocking kernels
  int dbg = 0;                        // DE265HIP_DEBUG: timing-only ablations of k_run (results invalid)
  bool intra_levels = false;          // DE265HIP_INTRA_MODE=levels: one launch per dependency level
  // Fault injection for the one device-side failure mode the design admits (tests/test_gpu_picture_parity.py), through an
  // explicit test entry point only (de265hip_debug_fault_injection; no environment switch makes a decoder fail): a picture
  // built while drop_producer is set leaves one run that other runs wait for out of its ticket list, so that run's flag is
  // never raised; spin_limit bounds the waits so that they expire within milliseconds instead of seconds.
  uint32_t spin_limit = RUN_SPIN_LIMIT_DEFAULT;
  uint32_t profiling = 0;             // bit k: launches of kernel id k are bracketed by hipEvents
  std::vector<PendingEvent> pending;
  double ms[DE265HIP_K_COUNT] = {};
  int64_t launches[DE265HIP_K_COUNT] = {};
};

struct de265hip_picture {
  de265hip_decoder* dec = nullptr;
  int dst_slot = 0;
  de265hip_pic_params params;
  PicDev P;
  void* arena = nullptr;              // one (pooled) device allocation for all command buffers
  size_t arena_bytes = 0;
  ArenaBuf arena_buf;                 // the pool entry behind `arena`
  hipEvent_t uploaded = nullptr;      // recorded on the copy stream behind the upload; the first run on a lane waits for it on that lane's stream
  uint32_t upload_waited = 0;         // bit l: lane l has waited
  int lane = 0;                       // the lane of its latest launch (its arena's last use is recorded there)
  // device pointers into the arena
  TuTask* d_tus = nullptr;
  int16_t* d_cval = nullptr; uint16_t* d_cpos = nullptr;
  uint8_t* d_scaling = nullptr;
  McTask* d_mc = nullptr; uint32_t* d_mc_order = nullptr;
  PcmTask* d_pcm = nullptr; uint16_t* d_pcm_samples = nullptr;
  de265hip_slice_params* d_slices = nullptr;
  de265hip_ctb_info* d_ctbs = nullptr;
  uint16_t* d_tile_id = nullptr;
  uint8_t* d_flags = nullptr; int8_t* d_qp = nullptr; de265hip_motion* d_motion = nullptr;
  uint8_t* d_bs = nullptr;
  SaoCtb* d_sao = nullptr;
  RunTask* d_runs = nullptr; uint32_t* d_deps = nullptr; uint32_t* d_sync = nullptr;
  uint32_t* d_mbx = nullptr; uint32_t* d_mbsegs = nullptr; unsigned long long* d_mb = nullptr; int n_mailboxes = 0;   // k_run edge mailboxes
  TuTask* d_run_tus = nullptr;
  int n_l0_size[4] = { 0, 0, 0, 0 };          // TU count per size in d_l0 (sorted 32,16,8,4)
  TuTask* d_l0 = nullptr; int n_l0 = 0;       // run mode: inter residual TUs + residual-only copies of intra TUs
  TuTask* d_l0_rext = nullptr; int n_l0_rext = 0;   // ... those with a range-extension tool (k_resid_rext)
  int16_t* d_resid = nullptr;                 // precomputed residual blocks of intra TUs
  int n_runs = 0, n_batches = 0, n_workers = 0, run_box = 64, ticket_batch = 1; size_t sync_bytes = 0, clear_bytes = 0;
  int n_front = 0;                            // runs [0, n_front): micro runs without producers, reconstructed by k_intra_front ahead of k_run
  bool run_direct = false;                    // k_run with one workgroup per ticket instead of persistent workers (wide pictures)
  uint32_t* d_slots = nullptr;
  uint32_t gen = 0;                           // runs of this picture so far (k_run flag generation)
  std::vector<int> level_start;       // level_start[l] .. level_start[l+1] in d_tus
  int n_mc = 0, n_mc2 = 0, n_mc_quads = 0, n_pcm = 0, n_tus = 0;   // MC tasks in all; of them n_mc2 chunks and 4 * n_mc_quads small blocks
  McBands mc_bands; bool mc_all = false;                            // k_mc_all: where band x's tiles / chunks / quads are
  bool any_edges = false;
  uint32_t ref_mask = 0;              // DPB slots the picture's MC tasks read (validated when the picture is launched)
  int n_launched = 0;                 // de265hip_picture_run calls so far
  de265hip_picture_stats stats = {};
  // device-side scan (scan_core.h): its parameters, buffers, and the counts the passes report (pinned ring entry)
  bool dev_scan = false, scan_pending = false;
  int scan_rc = 0;
  ScanParams SP; ScanLayout SL; ScanBufs SB;
  uint32_t cap_resid = 0;
  uint32_t* d_front_idx = nullptr;    // the front runs' ids (device-side scan: run records are not sorted)
  int ring_idx = -1; uint64_t ring_seq = 0;
  // what de265hip_picture_enqueue needs of the build (the device work of a build - upload, scan - may be issued later and by
  // another thread than the host stage: the pipeline issues every HIP call of a decoder from one thread)
  struct Enq {
    bool pending = false, uploaded_by_builder = false;
    uint64_t up_sig = 0;                      // upload issued through the HSA runtime: its completion signal (hsa_signal_t::handle)
    uint8_t* host_base = nullptr; size_t upload_bytes = 0; hipEvent_t stage_event = nullptr;
    size_t o_sync = 0, clear_bytes = 0, o_mot = 0, o_pus = 0, o_sl = 0, o_l0 = 0, o_l0x = 0, o_cpos = 0, nblk = 0;
    bool mot_given = true, check_on_device = false;
    int n_pus = 0, n_slices = 0, n_l0chk = 0, n_l0xchk = 0;
  } enq;
  ScanCounts h_counts_dry;            // (dry decoders: no ring)
  std::vector<uint8_t> dry_arena;     // (dry decoders: the arena in host memory)
  int64_t o_layout[16] = {};          // de265hip_debug_picture_layout
};

namespace {

size_t px_bytes(int bd) { return bd > 8 ? 2 : 1; }

// Deferred copy-outs.  The runtime performs a device-to-pinned-host hipMemcpyAsync with its DMA engines only when the stream it
// is enqueued on has nothing pending; behind a hipStreamWaitEvent (the picture's kernels on the decoder's stream) it launches a
// blit kernel instead - 256 workgroups x 512 threads per plane that store across PCIe through the same L2 write paths the
// reconstruction kernels use: a streaming kernel next to such a copy runs 3x longer, next to a DMA copy unchanged
// (tools/exp/sdmaprobe.hip; in bench.py's with_copy_out leg k_mc_all took 115-330 us instead of 59).  So a thread of the decoder
// waits for the picture ON THE HOST and then enqueues the copy on the idle output stream.
// (never destroyed: a process that exits without freeing its decoders leaves output threads waiting on the condition variable,
//  and destroying one that has waiters blocks for ever)
std::mutex& g_out_mu = *new std::mutex;
std::condition_variable& g_out_cv = *new std::condition_variable;
void out_settle(Slot& s, uint64_t upto = ~0ull)
{
  std::unique_lock<std::mutex> lk(g_out_mu);
  g_out_cv.wait(lk, [&] { return s.out_issued >= std::min(upto, s.out_queued); });
}

int free_slot(Slot& s)
{
  out_settle(s);
  { std::lock_guard<std::mutex> lk(g_out_mu); s.out_queued = s.out_issued = 0; }
  if (s.dl_done) (void)hipEventSynchronize(s.dl_done);
  for (int r = 0; r < Slot::kDlRing; r++) { if (s.dl_ev[r]) (void)hipEventDestroy(s.dl_ev[r]); s.dl_ev[r] = nullptr; s.dl_ev_seq[r] = 0; s.dl_ev_err_idx[r] = -1; }
  s.dl_done = nullptr;
  s.dl_seq = s.dl_waited = 0;
  // (hipFree below waits for the device: nobody is left to wait for)
  if (s.written) { (void)hipEventDestroy(s.written); s.written = nullptr; }
  for (int l = 0; l < kMaxLanes; l++) { if (s.read_done[l]) { (void)hipEventDestroy(s.read_done[l]); s.read_done[l] = nullptr; } s.rd_valid[l] = false; }
  s.writer_lane = -1; s.written_seq = 0;
  if (s.pl[0].ptr) (void)hipFree(s.pl[0].ptr);            // (one allocation: the chroma planes follow the luma plane)
  for (int c = 0; c < 3; c++) s.pl[c].ptr = nullptr;
  s.valid = false;
  return 0;
}

int alloc_slot(Slot& s, int w, int h, int bdY, int bdC, int cf = 1)
{
  if (s.valid && s.w == w && s.h == h && s.bdY == bdY && s.bdC == bdC && s.cf == cf) return 0;
  free_slot(s);
  // The three planes of a picture are ONE allocation, luma first: a kernel reaches all of them through one buffer descriptor
  // (the chunk form (mc_chunk_body)), and the planes of a slot always change hands together (the SAO output swap).
  size_t off[4] = { 0, 0, 0, 0 };
  for (int c = 0; c < 3; c++) {
    int cw = c ? (cf == 3 ? w : w / 2) : w, ch = c ? (cf == 1 ? h / 2 : h) : h;
    if (c && cf == 0) cw = ch = 0;                        // monochrome: the two chroma planes are empty (image.cc:301-306)
    int stride = (cw + 63) & ~63;                         // samples; rows start 128/64-byte aligned
    size_t bytes = (size_t)stride * ch * px_bytes(c ? bdC : bdY) + 256;
    off[c + 1] = off[c] + ((bytes + 255) & ~(size_t)255);
    s.pl[c].stride = stride;
  }
  void* base = nullptr;
  HIPCHK(hipMalloc(&base, off[3]), DE265HIP_ERROR_OUT_OF_MEMORY);
  for (int c = 0; c < 3; c++) s.pl[c].ptr = (char*)base + off[c];
  s.w = w; s.h = h; s.bdY = bdY; s.bdC = bdC; s.cf = cf; s.valid = true;
  return 0;
}

// ---- lanes
hipStream_t lane_st(de265hip_decoder* d, int l) { return l == 0 ? d->stream : d->lane_stream[l]; }
Slot& lane_sp(de265hip_decoder* d, int l) { return l == 0 ? d->spare : d->lane_spare[l]; }

hipError_t sync_all_lanes(de265hip_decoder* d)
{
  hipError_t rc = hipSuccess;
  for (int l = 0; l < d->n_lanes; l++) { const hipError_t e = hipStreamSynchronize(lane_st(d, l)); if (e != hipSuccess) rc = e; }
  return rc;
}

// `st` continues behind everything the decoder's lanes have queued so far (st may be a lane of this decoder or any other stream)
bool wait_for_all_lanes(de265hip_decoder* d, hipStream_t st)
{
  for (int l = 0; l < d->n_lanes; l++) {
    hipStream_t ls = lane_st(d, l);
    if (ls == st) continue;
    if (!d->lane_fence[l] && hipEventCreateWithFlags(&d->lane_fence[l], hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventRecord(d->lane_fence[l], ls) != hipSuccess || hipStreamWaitEvent(st, d->lane_fence[l], 0) != hipSuccess) return false;
  }
  return true;
}

// the host (a synchronous upload) or a full synchronisation has made the slot's content final: nobody is left to wait for
void slot_settled(Slot& s)
{
  s.writer_lane = -1;
  for (int l = 0; l < kMaxLanes; l++) s.rd_valid[l] = false;
}

struct Geometry {
  int ctbs_w, ctbs_h, w4, h4, tbs_w, tbs_h;
  std::vector<int> rs2ts, ts2rs, diag_order;      // diag_order: the CTBs by (x + 2y, y) (scan_core.h "tickets")
  std::vector<uint16_t> tile_id;
  std::vector<int> min_tb_zs;
};

// CtbAddrRStoTS, TileIdRS and MinTbAddrZS (6.5.1 / 6.5.2; pps.cc:560-690)
int make_geometry(const de265hip_pic_params& p, Geometry& g)
{
  const int ctb = 1 << p.log2_ctb_size;
  g.ctbs_w = (p.width + ctb - 1) >> p.log2_ctb_size;
  g.ctbs_h = (p.height + ctb - 1) >> p.log2_ctb_size;
  g.w4 = (p.width + 3) / 4; g.h4 = (p.height + 3) / 4;
  const int dl = p.log2_ctb_size - p.log2_min_tb_size;
  g.tbs_w = g.ctbs_w << dl; g.tbs_h = g.ctbs_h << dl;
  const int n = g.ctbs_w * g.ctbs_h;
  g.rs2ts.assign(n, 0); g.ts2rs.assign(n, 0); g.tile_id.assign(n, 0);
  const int nc = p.num_tile_columns, nr = p.num_tile_rows;
  if (p.col_bd[0] != 0 || p.row_bd[0] != 0 || p.col_bd[nc] != g.ctbs_w || p.row_bd[nr] != g.ctbs_h)
    return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  int ts = 0, tid = 0;
  for (int j = 0; j < nr; j++)
    for (int i = 0; i < nc; i++, tid++) {
      if (p.col_bd[i + 1] <= p.col_bd[i] || p.row_bd[j + 1] <= p.row_bd[j]) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
      for (int y = p.row_bd[j]; y < p.row_bd[j + 1]; y++)
        for (int x = p.col_bd[i]; x < p.col_bd[i + 1]; x++) {
          g.ts2rs[ts] = y * g.ctbs_w + x;
          g.rs2ts[y * g.ctbs_w + x] = ts++;                // tiles in raster order, CTBs raster inside a tile
          g.tile_id[y * g.ctbs_w + x] = (uint16_t)tid;
        }
    }
  g.diag_order.resize(n);
  for (int i = 0; i < n; i++) g.diag_order[i] = i;
  { const int cw = g.ctbs_w; std::stable_sort(g.diag_order.begin(), g.diag_order.end(), [cw](int a, int b) { return a % cw + 2 * (a / cw) < b % cw + 2 * (b / cw); }); }
  g.min_tb_zs.assign((size_t)g.tbs_w * g.tbs_h, 0);
  for (int y = 0; y < g.tbs_h; y++)
    for (int x = 0; x < g.tbs_w; x++) {
      int cx = x >> dl, cy = y >> dl;
      int v = g.rs2ts[cy * g.ctbs_w + cx] << (2 * dl);
      for (int i = 0; i < dl; i++) v |= (((x >> i) & 1) << (2 * i)) | (((y >> i) & 1) << (2 * i + 1));   // Morton interleave
      g.min_tb_zs[x + (size_t)y * g.tbs_w] = v;
    }
  return 0;
}

// Which neighbour units a TU's prediction actually reads, by size, mode and luma/chroma: the border is fetched as a whole
// (intrapred.cc:532-542) but a vertical mode never looks at the left column, a mode-2 TU never at the top row, planar only
// at the first sample beyond each side.  Bit u as in the availability mask.  Built once by running the very index
// arithmetic of the predictors (intrapred.cc:903-1069; the same code as run_prepare_sample in k_tu.hip) over every
// sample, plus the [1 2 1] / bilinear smoothing (:816-889) and the mode 10/26 and DC edge filters where they apply.
// Dependencies derived from it (instead of from every available unit) shorten the intra dependency chains.
static uint64_t g_used_units[4][35][2];
static void build_used_units()
{
  static const int8_t k_angle[35] = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                                      -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
  static const int16_t k_inv[15] = { -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };
  for (int l2 = 2; l2 <= 5; l2++)
    for (int mode = 0; mode < 35; mode++)
      for (int luma = 0; luma < 2; luma++) {
        const int nT = 1 << l2, C = 2 * nT, NB = 4 * nT + 1;
        std::vector<uint8_t> e(NB, 0);                    // border entries read
        if (mode == 0) {
          for (int i = 0; i < nT; i++) { e[C - 1 - i] = 1; e[C + 1 + i] = 1; }
          e[C + 1 + nT] = 1; e[C - 1 - nT] = 1;
        } else if (mode == 1) {
          for (int i = 0; i < nT; i++) { e[C - 1 - i] = 1; e[C + 1 + i] = 1; }
        } else {
          const int angle = k_angle[mode], inv = (mode >= 11 && mode <= 25 && angle < 0) ? k_inv[mode - 11] : 0;
          const bool vert = mode >= 18;
          for (int y = 0; y < nT; y++)
            for (int x = 0; x < nT; x++) {
              const int a = vert ? y : x, b = vert ? x : y;
              const int iIdx = ((a + 1) * angle) >> 5;
              const int i0 = b + iIdx + 1, i1 = i0 + 1;
              const int k0 = i0 >= 0 ? i0 : -((i0 * inv + 128) >> 8);
              const int k1 = std::min(i1 >= 0 ? i1 : -((i1 * inv + 128) >> 8), C);
              e[vert ? C + k0 : C - k0] = 1; e[vert ? C + k1 : C - k1] = 1;
            }
          if (luma && nT < 32 && (mode == 26 || mode == 10)) {          // edge filter: the other side and the corner
            for (int i = 0; i < nT; i++) e[mode == 26 ? C - 1 - i : C + 1 + i] = 1;
            e[C] = 1;
          }
        }
        const int md = std::min(std::abs(mode - 26), std::abs(mode - 10));
        const bool smooth = luma && mode != 1 && (l2 == 3 ? md > 7 : (l2 == 4 ? md > 1 : (l2 == 5 && md > 0)));
        if (smooth) {
          std::vector<uint8_t> f(e);
          for (int q = 0; q < NB; q++) if (e[q]) { if (q > 0) f[q - 1] = 1; if (q + 1 < NB) f[q + 1] = 1; }
          // bilinear variant (strong_intra_smoothing): every filtered sample is made from p[0] and p[+-64], and the
          // decision for the variant also reads p[+-32] (intrapred.cc:847-852)
          if (l2 == 5) { f[0] = 1; f[nT] = 1; f[C] = 1; f[3 * nT] = 1; f[4 * nT] = 1; }
          e.swap(f);
        }
        uint64_t u = 0;
        const int corner = nT >> 1;
        for (int q = 0; q < NB; q++) if (e[q]) u |= 1ull << (q < C ? (q >> 2) : (q == C ? corner : corner + 1 + ((q - C - 1) >> 2)));
        g_used_units[l2 - 2][mode][luma] = u;
      }
}
static void ensure_used_units() { static std::once_flag once; std::call_once(once, build_used_units); }

// the units whose samples a TU with availability `avail` really reads: the used ones that are available, plus, for
// every used but unavailable one, the unit its samples are substituted from (intrapred.cc:395-431)
static uint64_t needed_units(uint64_t used, uint64_t avail)
{
  if (avail == 0) return 0;
  uint64_t need = used & avail, miss = used & ~avail;
  while (miss) {
    const int u = __builtin_ctzll(miss); miss &= miss - 1;
    const uint64_t below = avail & ((2ull << u) - 1ull);
    need |= below ? 1ull << (63 - __builtin_clzll(below)) : avail & (~avail + 1ull);
  }
  return need;
}

// ---- flat, reused working set of the host stage (one per host thread): a picture's build allocates nothing in the
// steady state and touches no fresh page
struct Cell { int32_t run; uint16_t lvl, llvl; };        // per 4x4 unit of a component: run of the intra TU covering it (-1: none), its TU level, its in-run level
struct RunB {                                            // a run under construction
  int32_t c, ctu, x0, y0, x1, y1, wx1, wy1, level, est;  // est: run level as far as known during the scan
  int32_t n_tus, head, tail;                             // its TUs: list through TuTask::resid_offset of BuildScratch::it, decode order
  int32_t n_deps, dep_head, dep_tail;                    // its producer runs: list through BuildScratch::dep_next, order of discovery
  int64_t alg;
  int32_t foreign;                                       // some neighbour its TUs read is not written by an intra run of this picture (inter / PCM samples)
};
struct BuildScratch {
  std::vector<Cell> cells[3];
  int32_t epoch_base = 0;
  std::vector<uint32_t> ctb_group;
  // intra TUs in decode order.  Until the runs are laid out three fields of a record carry scan state: resid_offset = the next
  // TU of its run (-1: last), run_level = its in-run level - 1 (what the field finally holds), pad3 = its range-extension
  // bits (D265_RX_*; cleared in the run-ordered copy)
  std::vector<TuTask> it;
  struct ItXcc { int32_t ti; int8_t rsv; uint64_t luma; };
  std::vector<ItXcc> it_xcc;                                                                 // cross-component prediction: ResScaleVal and the luma TU, by ascending ti
  std::vector<TuTask> l0_rext;                                                              // level-0 tasks of k_resid_rext
  std::vector<RunB> rb;
  std::vector<int32_t> dep_val, dep_next;
  // Cr mirrors Cb (tu_scan): what the scan decided for a Cb TU, waiting for the Cr TU of the same place; Cb run -> its Cr run
  struct CbScan { uint16_t x0, y0; uint8_t log2, mode, kind, foreign; int32_t run, level, llev, n_prod; uint64_t mask; int32_t prod[40]; };
  CbScan cbq[4];
  std::vector<int32_t> mirror;
  std::vector<uint64_t> avail_memo; int avail_memo_key = -1;                                   // availability masks by (chroma, size, position in the CTB)
  std::vector<int> level_hist;
  std::vector<TuTask> all_tasks; std::vector<int> all_levels;                              // DE265HIP_INTRA_MODE=levels only
  std::vector<TuTask> l0, run_tus;
  // level-0 tasks of the plain inter TUs by size: raw arrays with room for every TU record of the picture (no growth test per TU)
  struct TaskBuf { TuTask* p = nullptr; size_t cap = 0, n = 0; ~TaskBuf() { free(p); }
                   bool ensure(size_t c) { if (c <= cap) return true; free(p); p = (TuTask*)malloc(c * sizeof(TuTask)); cap = p ? c : 0; return p != nullptr; } };
  TaskBuf l0_inter[4];
  std::vector<RunTask> runs; std::vector<uint32_t> run_deps, slots, mbx, mb_segs, mb_owner; std::vector<uint8_t> rdy_tab;
  std::vector<int> order, newidx, count2, width; std::vector<uint8_t> micro;
  struct QuadPend { McTask t[4]; int n = 0; };                                               // a slot pair's open quad (k_mc_all)
  std::vector<McTask> mcs, mc_tiles[8], mc_chunks[8], mc_quads[8]; QuadPend mc_pend[8 * 17 * 17]; std::vector<int> micro_keys;
  std::vector<uint8_t> band_row; std::vector<uint32_t> roww, mc_order[8], mc_order_all;
  std::vector<PcmTask> pcms; std::vector<SaoCtb> saos;
};
static thread_local BuildScratch g_scratch;
static thread_local bool g_no_cr_mirror = false;         // set while a build is repeated without the Cb -> Cr shortcut of its TU scan
static const int8_t k_intra_angle[35] = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                                          -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const int16_t k_inv_angle[15] = { -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };

// ---- pools (caller holds dec->mu)
constexpr size_t kPoolLimitBytes = (size_t)24 << 30;      // free arenas kept for reuse (of 288 GB of HBM)

// Arenas whose last user has finished on the device move from `cooling` to the free list (caller holds dec->mu).  The only
// place that asks the runtime (hipEventQuery): called by whoever issues the decoder's HIP calls anyway (enqueue, run), and by
// acquire_arena when the free list has nothing to offer.
void reap_arenas(de265hip_decoder* dec)
{
  for (size_t i = 0; i < dec->cooling.size();) {
    if (hipEventQuery(dec->cooling[i].last_use) == hipSuccess) {
      ArenaBuf a = dec->cooling[i];
      a.used = false;                                    // (nothing left to wait for: its next upload starts at once)
      dec->free_arenas.push_back(a);
      dec->cooling.erase(dec->cooling.begin() + i);
    } else if (++i >= 4) break;                          // (they finish roughly in order: no need to ask about all of them)
  }
}

int acquire_arena(de265hip_decoder* dec, size_t bytes, ArenaBuf* out)
{
  // Only arenas that are KNOWN to be free (reap_arenas): an arena taken back while its last picture's kernels were still
  // queued made the new picture's upload - and every upload and scan queued behind it on those streams - wait on the device
  // for that picture's whole reconstruction (round 4: uploads of 0.14 ms completed 1-2 ms after they were issued, and a
  // decoder's scans came in at 800 per second whatever was done to them).  The pool grows to what is in flight plus what cools.
  auto pick = [&]() {
    int best = -1;
    for (size_t i = 0; i < dec->free_arenas.size(); i++) {
      const ArenaBuf& a = dec->free_arenas[i];
      if (a.bytes >= bytes && a.bytes <= 4 * bytes + ((size_t)8 << 20) && (best < 0 || a.bytes < dec->free_arenas[best].bytes)) best = (int)i;
    }
    return best;
  };
  int best = pick();
  if (best < 0 && !dec->cooling.empty()) { reap_arenas(dec); best = pick(); }
  if (best >= 0) {
    *out = dec->free_arenas[best];
    dec->free_arenas.erase(dec->free_arenas.begin() + best);
    dec->pooled_bytes -= out->bytes;
    return 0;
  }
  ArenaBuf a;
  a.bytes = (bytes + ((size_t)4 << 20) - 1) & ~(((size_t)4 << 20) - 1);        // 4 MB size classes
  HIPCHK(hipMalloc(&a.ptr, a.bytes), DE265HIP_ERROR_OUT_OF_MEMORY);
  if (hipEventCreateWithFlags(&a.last_use, hipEventDisableTiming) != hipSuccess) { (void)hipFree(a.ptr); return DE265HIP_ERROR_OUT_OF_MEMORY; }
  *out = a;
  return 0;
}

void destroy_arena(ArenaBuf& a)
{
  if (a.ptr) (void)hipFree(a.ptr);                        // (synchronises with the device: whatever used it has finished)
  if (a.last_use) (void)hipEventDestroy(a.last_use);
  a = ArenaBuf();
}

// hand an arena back: reusable once everything enqueued on the decoder's stream so far has run
void release_arena(de265hip_decoder* dec, ArenaBuf a, hipStream_t last_stream = nullptr)
{
  if (!a.ptr) return;
  if (hipEventRecord(a.last_use, last_stream ? last_stream : dec->stream) != hipSuccess || dec->pooled_bytes + a.bytes > kPoolLimitBytes) { destroy_arena(a); return; }
  a.used = true; a.release_seq = ++dec->arena_releases;
  dec->pooled_bytes += a.bytes;
  dec->cooling.push_back(a);
  reap_arenas(dec);
}

int acquire_stage(de265hip_decoder* dec, size_t bytes, int* index)
{
  // Called by the threads that run the host stage of a build (the pipeline's workers): in the steady state WITHOUT a HIP call -
  // whoever learns that an upload has completed (de265hip_picture_run, when the picture's scan has reported) marks its buffer
  // idle.  Only when no idle buffer is large enough are the buffers with an upload in flight asked through their events
  // (round 4: every worker queried every buffer's event on every build, two dozen calls into the HIP runtime that queued up
  //  behind its locks with the launcher's).
  auto pick = [&]() {
    int best = -1;
    for (size_t i = 0; i < dec->stage_pool.size(); i++) {
      StageBuf& b = dec->stage_pool[i];
      if (b.state == 0 && b.bytes >= bytes && (best < 0 || b.bytes < dec->stage_pool[best].bytes)) best = (int)i;
    }
    return best;
  };
  int best = pick();
  if (best < 0) {
    for (auto& b : dec->stage_pool)
      if (b.state == 2 && (b.sig ? hsa_signal_load_scacquire(hsa_signal_t{ b.sig }) < 1 : hipEventQuery(b.copied) == hipSuccess)) b.state = 0;
    best = pick();
  }
  if (best < 0) {
    // recycle an idle buffer that is too small rather than growing without bound
    for (size_t i = 0; i < dec->stage_pool.size(); i++)
      if (dec->stage_pool[i].state == 0 && dec->stage_pool[i].ptr) {
        (void)hipHostFree(dec->stage_pool[i].ptr); (void)hipEventDestroy(dec->stage_pool[i].copied);
        dec->stage_pool.erase(dec->stage_pool.begin() + i);
        break;
      }
    StageBuf b;
    b.bytes = (bytes + ((size_t)4 << 20) - 1) & ~(((size_t)4 << 20) - 1);
    HIPCHK(hipHostMalloc(&b.ptr, b.bytes, hipHostMallocDefault), DE265HIP_ERROR_OUT_OF_MEMORY);
    if (hipEventCreateWithFlags(&b.copied, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(b.ptr); return DE265HIP_ERROR_OUT_OF_MEMORY; }
    dec->stage_pool.push_back(b);
    best = (int)dec->stage_pool.size() - 1;
  }
  dec->stage_pool[best].state = 1; dec->stage_pool[best].owner = 0;
  *index = best;
  return 0;
}

// DE265HIP_BUILD_TIMING=1: where the host stage of a picture goes (stderr, one line per build)
struct PhaseTimer {
  bool on; std::chrono::steady_clock::time_point t0; char buf[512]; int len = 0;
  PhaseTimer() : on(d265_env("DE265HIP_BUILD_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* name) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    len += snprintf(buf + len, sizeof(buf) - len, " %s=%.2fms", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
  void done() { if (on) fprintf(stderr, "de265hip build:%s\n", buf); }
};

// Kernel streams of the process, four per device, created BACK TO BACK the first time a decoder asks: the device has four
// dispatch pipes, a hardware queue is bound to pipe (number of the queue in order of creation) % 4 whatever its priority, and a
// pipe dispatches one kernel at a time - a kernel of many workgroups keeps it for as long as it runs (tools/exp/pipeprobe.hip:
// a one-workgroup kernel behind a 40 000-workgroup grid finishes after 30 us from another pipe, after 415 us from the same).
// Two decoders whose kernel streams met on one pipe cost a third of the device: replay of three GOP streams 5 200 instead of
// 7 050 pictures/s, decided by how many other streams had been created in between (round 4).  Four streams created in a row
// sit on four different pipes; decoder k of the process works on stream k % 4 (more than four decoders share: they would
// share a pipe anyway).  DE265HIP_OWN_STREAMS=1: every decoder creates its own, as before.
constexpr int kKernelStreams = 4;
// The copy streams (upload of a picture's records, the scan kernels behind it: priority high) and the upload streams (priority
// low) come from pools of the process too, four and two per device: every HIP stream may cost a hardware queue, and beyond
// some 16 hardware queues in the process the device's scheduler time-slices them - three decoders with three copy streams each
// and the priority pools of their own (4 + 9 + 6 + 3 streams) ran at 5 to 300 pictures/s instead of 2 500 (round 4,
// tools/exp/r4b_window.sh).  Copy-out streams stay per decoder (created when first used).
constexpr int kScanStreams = 4, kUploadStreams = 2;
struct DeviceStreams {
  hipStream_t kernel[kKernelStreams] = {}, scan[kScanStreams] = {}, upload[kUploadStreams] = {}, out[2] = {}, spare = nullptr, pad[3] = {};
  int n_scan = kScanStreams, next_out = 0;
  int next_kernel = 0, next_scan = 0, next_upload = 0;
  bool ok = false;
};
static std::mutex g_streams_mu;
static std::map<int, DeviceStreams> g_streams;
static bool own_streams()
{
  static const bool own = d265_env("DE265HIP_OWN_STREAMS") && atoi(d265_env("DE265HIP_OWN_STREAMS"));
  return own;
}
// (caller holds g_streams_mu)
static DeviceStreams* device_streams(int device)
{
  auto it = g_streams.find(device);
  if (it == g_streams.end()) {
    DeviceStreams D;
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return nullptr;      // (numerically: lowest, greatest priority)
    if (const char* e = d265_env("DE265HIP_FLAT_PRIORITIES")) if (atoi(e)) lo = hi = 0;
    bool ok = true;
    // The order of creation decides the dispatch pipe of each stream's hardware queue (pipe = number of the queue % 4).  Three
    // kernel streams on three pipes, and the fourth pipe for the scan streams: a kernel of the reconstruction with a grid of
    // tens of thousands of workgroups keeps its pipe for as long as it runs, and a scan kernel queued on that pipe - seven
    // dependent ones per scan - waited behind it each time (and the other way round).  Every fourth queue created is a scan
    // stream; in between come the two upload streams, the fourth kernel stream (a fourth decoder shares a pipe with the third:
    // there are only four) and two streams for copy-outs.  At most four new queues per priority (GPU_MAX_HW_QUEUES defaults to
    // 4 per priority pool; a stream beyond that re-uses a queue and the count would slip).  DE265HIP_SCAN_PIPES=0: the round's
    // first order (kernel x 4, scan x 4, upload x 2: a scan stream on every pipe).
    const bool own_pipe = !(d265_env("DE265HIP_SCAN_PIPES") && atoi(d265_env("DE265HIP_SCAN_PIPES")) == 0);
    auto mk = [&](hipStream_t* st_, int prio) { if (ok) ok = hipStreamCreateWithPriority(st_, hipStreamNonBlocking, prio) == hipSuccess; };
    if (own_pipe) {
      mk(&D.kernel[0], 0); mk(&D.kernel[1], 0); mk(&D.kernel[2], 0); mk(&D.scan[0], hi);
      mk(&D.upload[0], lo); mk(&D.upload[1], lo); mk(&D.kernel[3], 0); mk(&D.scan[1], hi);
      mk(&D.out[0], lo); mk(&D.out[1], lo); mk(&D.spare, hi); mk(&D.scan[2], hi);
      D.n_scan = 3;
      // A fourth scan stream on the scan pipe - three more queues in between - where the runtime may create that many
      // (GPU_MAX_HW_QUEUES >= 8 per priority; with the default of 4 the new streams would re-use queues and land elsewhere):
      // 3 640-3 810 -> 3 910-3 960 pictures/s, three decoders.  Sixteen hardware queues in all; the device's scheduler began to
      // time-slice somewhere beyond that: DE265HIP_SCAN_STREAMS=4 asks for it (bench.py does for a single rank).
      // (opt-in: a process that creates queues of its own - RCCL does - may be closer to that edge than this library can see)
      const int want_scan = d265_env("DE265HIP_SCAN_STREAMS") ? atoi(d265_env("DE265HIP_SCAN_STREAMS")) : 3;
      if (want_scan >= 4) {
        mk(&D.pad[0], 0); mk(&D.pad[1], lo); mk(&D.pad[2], lo); mk(&D.scan[3], hi);
        D.n_scan = 4;
      }
    } else {
      for (int i = 0; i < kKernelStreams; i++) mk(&D.kernel[i], 0);
      for (int i = 0; i < kScanStreams; i++) mk(&D.scan[i], hi);
      for (int i = 0; i < kUploadStreams; i++) mk(&D.upload[i], lo);
      D.n_scan = kScanStreams;
    }
    if (!ok) return nullptr;                             // (leaves what was created: the process is about to fail anyway)
    D.ok = true;
    it = g_streams.emplace(device, D).first;
  }
  return &it->second;
}
static hipStream_t pooled_kernel_stream(int device, int* index, bool fixed = false)
{
  if (own_streams()) return nullptr;
  std::lock_guard<std::mutex> lk(g_streams_mu);
  DeviceStreams* D = device_streams(device);
  if (!D) return nullptr;
  if (!fixed) *index = D->next_kernel++ % kKernelStreams;
  return D->kernel[*index];
}
static hipStream_t pooled_scan_stream(int device)
{
  if (own_streams()) return nullptr;
  std::lock_guard<std::mutex> lk(g_streams_mu);
  DeviceStreams* D = device_streams(device);
  if (!D) return nullptr;
  static const bool io_on_scan_pipe = d265_env("DE265HIP_UPLOAD_PIPE") && atoi(d265_env("DE265HIP_UPLOAD_PIPE")) == 3;
  const int n = (io_on_scan_pipe && D->n_scan == 4) ? 3 : D->n_scan;
  return D->scan[D->next_scan++ % n];
}
static hipStream_t pooled_out_stream(int device)
{
  if (own_streams()) return nullptr;
  std::lock_guard<std::mutex> lk(g_streams_mu);
  DeviceStreams* D = device_streams(device);
  if (!D || !D->out[0]) return nullptr;
  // two copy-out streams; with the sixteen-queue layout (DE265HIP_SCAN_STREAMS=4) its two low-priority fillers as well: a
  // decoder's copy-outs wait for ITS pictures, and two decoders on one stream wait for each other's
  hipStream_t all[4] = { D->out[0], D->out[1], D->pad[1], D->pad[2] };
  static const int want = d265_env("DE265HIP_OUT_STREAMS") ? std::max(1, std::min(4, atoi(d265_env("DE265HIP_OUT_STREAMS")))) : 4;
  const int n = std::min(want, (D->pad[1] && D->pad[2]) ? 4 : 2);
  return all[D->next_out++ % n];
}
static hipStream_t pooled_upload_stream(int device)
{
  if (own_streams()) return nullptr;
  std::lock_guard<std::mutex> lk(g_streams_mu);
  DeviceStreams* D = device_streams(device);
  if (!D) return nullptr;
  // (experiments, DESIGN §10 "pipes": DE265HIP_UPLOAD_PIPE=2: both upload streams on the third kernel stream's pipe; =3: the fourth
  //  queue of the scan pipe carries the uploads, three scan streams)
  static const int up_pipe = d265_env("DE265HIP_UPLOAD_PIPE") ? atoi(d265_env("DE265HIP_UPLOAD_PIPE")) : -1;
  if (up_pipe == 2 && D->spare && D->pad[2]) return (D->next_upload++ & 1) ? D->pad[2] : D->spare;
  if (up_pipe == 3 && D->n_scan == 4) return D->scan[3];
  return D->upload[D->next_upload++ % kUploadStreams];
}

// ---- uploads through the HSA runtime (the DMA engines, no compute queue).  hipMemcpyAsync on an upload stream puts barrier
// packets around every copy into that stream's hardware queue, and a kernel stream whose queue shares the dispatch pipe with it
// runs its kernels a third slower.  hsa_amd_memory_async_copy has no queue: a copy and its completion signal.
struct HsaDev { bool tried = false, ok = false; hsa_agent_t gpu{}, cpu{}; };
static std::map<int, HsaDev> g_hsa;
struct HsaFind { uint32_t domain, bdf; HsaDev* out; bool have_cpu; };
static hsa_status_t hsa_agent_cb(hsa_agent_t a, void* p)
{
  HsaFind* f = static_cast<HsaFind*>(p);
  hsa_device_type_t t;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->out->cpu = a; f->have_cpu = true; }
  if (t == HSA_DEVICE_TYPE_GPU) {
    uint32_t bdf = 0, dom = 0;
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
    if (bdf == f->bdf && dom == f->domain) { f->out->gpu = a; f->out->ok = true; }
  }
  return HSA_STATUS_SUCCESS;
}
// (the HIP device's agent by its PCI address; nullptr: no such agent, and the uploads go through hipMemcpyAsync as before)
static HsaDev* hsa_dev(int device)
{
  std::lock_guard<std::mutex> lk(g_streams_mu);
  HsaDev& H = g_hsa[device];
  if (!H.tried) {
    H.tried = true;
    char bus[64] = "";
    unsigned dom = 0, b = 0, dv = 0, fn = 0;
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) == hipSuccess && sscanf(bus, "%x:%x:%x.%x", &dom, &b, &dv, &fn) == 4 &&
        hsa_init() == HSA_STATUS_SUCCESS) {
      HsaFind f{ dom, (b << 8) | (dv << 3) | fn, &H, false };
      (void)hsa_iterate_agents(hsa_agent_cb, &f);
      H.ok = H.ok && f.have_cpu;
    }
    (void)hipGetLastError();
  }
  return H.ok ? &H : nullptr;
}
static bool uploads_by_hsa()
{
  static const bool off = d265_env("DE265HIP_UPLOAD") && !strcmp(d265_env("DE265HIP_UPLOAD"), "stream");      // the round's earlier form
  return !off;
}
// wait for an upload's signal; false: the copy failed (the runtime sets the signal negative)
static bool hsa_upload_wait(uint64_t sig)
{
  if (!sig) return true;
  // (a copy of a few megabytes takes a fraction of a millisecond; one that has not arrived after half a minute never will)
  hsa_signal_value_t v = 1;
  for (int tries = 0; tries < 15 && v >= 1; tries++)
    v = hsa_signal_wait_scacquire(hsa_signal_t{ sig }, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED);
  return v == 0;
}

struct ArenaLayout {
  size_t total = 0;
  size_t add(size_t bytes) { size_t o = total; total = (total + bytes + 255) & ~(size_t)255; return o; }
};

}  // namespace

extern "C" {

const char* de265hip_version(void) { return "libde265-hip 0.1 (gfx950)"; }

int de265hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int de265hip_decoder_new(de265hip_decoder** out, int device)
{
  if (!out) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  *out = nullptr;
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n), DE265HIP_ERROR_INIT_FAILED);
  if (n <= 0) return DE265HIP_ERROR_INIT_FAILED;
  if (device >= 0) HIPCHK(hipSetDevice(device), DE265HIP_ERROR_INIT_FAILED);
  de265hip_decoder* d = new (std::nothrow) de265hip_decoder();
  if (!d) return DE265HIP_ERROR_OUT_OF_MEMORY;
  HIPCHK(hipGetDevice(&d->device), DE265HIP_ERROR_INIT_FAILED);
  // Three kinds of streams, three priorities - the runtime keeps a pool of hardware queues PER priority (GPU_MAX_HW_QUEUES
  // each), and hands a new stream the least used queue of its pool.  With every stream at the default priority a decoder's
  // kernel stream shared its hardware queue - or its dispatch pipe - with some other decoder's, or not, depending on how many
  // copy streams had been created before it: device replay of three GOP streams 4 550 / 5 190 / 7 080 pictures/s for 4 / 2 / 1
  // copy streams per decoder (round 4, tools/exp/r4b_replay_streams.sh).  Now the streams come from the process's pools
  // (DeviceStreams above): kernel streams (and lanes) at the default priority, the copy streams (scan of the TU records:
  // short, latency-bound kernels the launches wait for) high, the upload and copy-out streams low.
  int prio_lo = 0, prio_hi = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi), DE265HIP_ERROR_INIT_FAILED);      // (numerically: lowest, greatest)
  d->prio_low = prio_lo; d->prio_high = prio_hi;
  if (const char* e = d265_env("DE265HIP_FLAT_PRIORITIES")) if (atoi(e)) d->prio_low = d->prio_high = 0;
  d->stream = pooled_kernel_stream(d->device, &d->kstream_index);
  if (!d->stream) { d->kstream_index = -1; HIPCHK(hipStreamCreateWithPriority(&d->stream, hipStreamNonBlocking, 0), DE265HIP_ERROR_INIT_FAILED); }
  d->n_copy_streams = 4;               // (pooled: every decoder takes turns on all four; 2 / 3 / 4: 2 570 / 2 770 / 2 900 pictures/s, three decoders)
  if (const char* e = d265_env("DE265HIP_COPY_STREAMS")) d->n_copy_streams = std::min((int)de265hip_decoder::kMaxCopyStreams, std::max(1, atoi(e)));
  d->streams_pooled = d->kstream_index >= 0;
  if (d->streams_pooled) d->n_copy_streams = std::min(d->n_copy_streams, 4);
  for (int i = 0; i < d->n_copy_streams; i++) {
    if (d->streams_pooled) d->copy_streams[i] = pooled_scan_stream(d->device);
    else HIPCHK(hipStreamCreateWithPriority(&d->copy_streams[i], hipStreamNonBlocking, d->prio_high), DE265HIP_ERROR_INIT_FAILED);
  }
  d->copy_stream = d->copy_streams[0];
  for (int i = 0; i < 2; i++) {
    if (d->streams_pooled) d->upload_streams[i] = pooled_upload_stream(d->device);
    else HIPCHK(hipStreamCreateWithPriority(&d->upload_streams[i], hipStreamNonBlocking, d->prio_low), DE265HIP_ERROR_INIT_FAILED);
  }
  HIPCHK(hipMalloc((void**)&d->d_err, 256), DE265HIP_ERROR_OUT_OF_MEMORY);
  HIPCHK(hipMemset(d->d_err, 0, 256), DE265HIP_ERROR_INIT_FAILED);
  HIPCHK(hipMalloc((void**)&d->d_err_ring, de265hip_decoder::kRing * 4), DE265HIP_ERROR_OUT_OF_MEMORY);
  HIPCHK(hipMemset(d->d_err_ring, 0, de265hip_decoder::kRing * 4), DE265HIP_ERROR_INIT_FAILED);
  HIPCHK(hipHostMalloc((void**)&d->h_ring, de265hip_decoder::kRing * sizeof(ScanCounts), hipHostMallocMapped | hipHostMallocCoherent), DE265HIP_ERROR_OUT_OF_MEMORY);
  memset(d->h_ring, 0, de265hip_decoder::kRing * sizeof(ScanCounts));
  for (int i = 0; i < de265hip_decoder::kRing; i++) d->ring_free.push_back(i);
  ensure_used_units();
  HIPCHK(hipMalloc((void**)&d->d_used_units, sizeof(g_used_units)), DE265HIP_ERROR_OUT_OF_MEMORY);
  HIPCHK(hipMemcpy(d->d_used_units, g_used_units, sizeof(g_used_units), hipMemcpyHostToDevice), DE265HIP_ERROR_INIT_FAILED);
  if (const char* e = d265_env("DE265HIP_HOST_SCAN")) d->dev_scan = atoi(e) == 0;
  const char* mode = d265_env("DE265HIP_INTRA_MODE");
  d->intra_levels = mode && !strcmp(mode, "levels");
  if (const char* dbg = d265_env("DE265HIP_DEBUG")) d->dbg = atoi(dbg);
  if (const char* e = d265_env("DE265HIP_SEPARATE_BS")) d->separate_bs = atoi(e) != 0;
  if (const char* e = d265_env("DE265HIP_SAO_STRIPS")) d->sao_strips = atoi(e) != 0;
  if (const char* e = d265_env("DE265HIP_TWO_PASS_DEBLOCK")) d->two_pass_deblock = atoi(e) != 0;
  if (const char* e = d265_env("DE265HIP_LF_TILE")) d->lf_tile = atoi(e) != 0;
  if (const char* e = d265_env("DE265HIP_RESID16_BIG")) d->resid16_big = atoi(e) != 0;
  if (const char* e = d265_env("DE265HIP_RESID_ONE_LAUNCH")) d->resid_one_launch = atoi(e) != 0;
  if (const char* rw = d265_env("DE265HIP_RUN_WAVES")) d->run_waves = std::min(RUN_WAVES, std::max(1, atoi(rw)));
  if (const char* e = d265_env("DE265HIP_LANES")) {
    const int rc = de265hip_decoder_set_lanes(d, std::min(kMaxLanes, std::max(1, atoi(e))));
    if (rc) { de265hip_decoder_free(d); return rc; }
  }
  *out = d;
  return DE265HIP_OK;
}

int de265hip_decoder_set_lanes(de265hip_decoder* d, int n_lanes)
{
  if (!d || n_lanes < 1 || n_lanes > kMaxLanes) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  std::lock_guard<std::mutex> lk(d->mu);
  HIPCHK(sync_all_lanes(d), DE265HIP_ERROR_DECODING);                 // (a change takes effect between pictures)
  for (auto& s : d->slots) slot_settled(s);
  for (int l = 1; l < n_lanes; l++)
    if (!d->lane_stream[l]) {
      // (a lane takes the pool's next kernel stream - another dispatch pipe - when the decoder's own came from the pool)
      if (d->kstream_index >= 0) { int ix = (d->kstream_index + l) % kKernelStreams; d->lane_stream[l] = pooled_kernel_stream(d->device, &ix, true); d->lane_pooled[l] = d->lane_stream[l] != nullptr; }
      if (!d->lane_stream[l]) HIPCHK(hipStreamCreateWithPriority(&d->lane_stream[l], hipStreamNonBlocking, 0), DE265HIP_ERROR_INIT_FAILED);
    }
  d->n_lanes = n_lanes;
  return DE265HIP_OK;
}

// Lifetime rule (include/de265_hip.h): the decoder owns the device side of its pictures.  Freeing the decoder first
// releases their device memory and ORPHANS the handles (dec = nullptr): an orphan can only be freed; picture_free
// never touches a dead decoder.
void de265hip_decoder_free(de265hip_decoder* d)
{
  if (!d) return;
  if (d265_env("DE265HIP_PIPE_TIMING") && d->n_scan_wait)
    fprintf(stderr, "de265hip decoder: %ld launches; ms per picture: waiting for the scan %.3f, run_picture (kernel launches) %.3f\n",
            d->n_scan_wait, 1e3 * d->t_scan_wait / d->n_scan_wait, 1e3 * d->t_run / d->n_scan_wait);
  if (d265_env("DE265HIP_PIPE_TIMING") && d->n_sec) {
    static const char* nm[8] = { "wait-arena-event", "event-pool", "memset", "memcpy-h2d", "stage-event", "motion+clear", "scan-launches", "uploaded-events" };
    fprintf(stderr, "de265hip enqueue sections, us per picture (max of one call):");
    for (int k = 0; k < 8; k++) fprintf(stderr, " %s %.1f (%.0f)", nm[k], 1e6 * d->t_sec[k] / d->n_sec, 1e6 * d->t_sec_max[k]);
    fprintf(stderr, "\n");
  }
  if (d->out_thread_started) {                       // (the output thread hands over what it still holds, then ends)
    { std::lock_guard<std::mutex> lo(g_out_mu); d->out_stop = true; }
    g_out_cv.notify_all();
    d->out_thread.join();
  }
  for (hipEvent_t e : d->out_events) (void)hipEventDestroy(e);
  for (uint64_t sg : d->free_sigs) (void)hsa_signal_destroy(hsa_signal_t{ sg });
  for (int i = 0; i < 2; i++) if (d->upload_streams[i]) (void)hipStreamSynchronize(d->upload_streams[i]);
  for (int i = 0; i < d->n_copy_streams; i++) if (d->copy_streams[i]) (void)hipStreamSynchronize(d->copy_streams[i]);
  (void)sync_all_lanes(d);
  if (d->out_stream) { (void)hipStreamSynchronize(d->out_stream); if (!d->out_pooled) (void)hipStreamDestroy(d->out_stream); }
  if (d->out_fence) (void)hipEventDestroy(d->out_fence);
  {
    std::lock_guard<std::mutex> lk(d->mu);
    for (de265hip_picture* p : d->live) {
      if (p->enq.up_sig) { (void)hsa_upload_wait(p->enq.up_sig); (void)hsa_signal_destroy(hsa_signal_t{ p->enq.up_sig }); p->enq.up_sig = 0; }      // (an upload still on its way into the arena)
      destroy_arena(p->arena_buf);
      p->arena = nullptr;
      if (p->uploaded) { (void)hipEventDestroy(p->uploaded); p->uploaded = nullptr; }
      p->dec = nullptr;
    }
    d->live.clear();
    for (auto& a : d->free_arenas) destroy_arena(a);
    d->free_arenas.clear();
    for (auto& a : d->cooling) destroy_arena(a);
    d->cooling.clear();
    for (auto& b : d->stage_pool) { if (b.ptr) (void)hipHostFree(b.ptr); if (b.copied) (void)hipEventDestroy(b.copied); }
    d->stage_pool.clear();
    for (hipEvent_t e : d->free_events) (void)hipEventDestroy(e);
    d->free_events.clear();
  }
  for (auto& e : d->pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& s : d->slots) free_slot(s);
  free_slot(d->spare);
  for (int l = 1; l < kMaxLanes; l++) { free_slot(d->lane_spare[l]); if (d->lane_stream[l] && !d->lane_pooled[l]) (void)hipStreamDestroy(d->lane_stream[l]); }
  for (int l = 0; l < kMaxLanes; l++) if (d->lane_fence[l]) (void)hipEventDestroy(d->lane_fence[l]);
  if (d->d_err) (void)hipFree(d->d_err);
  if (d->d_err_ring) (void)hipFree(d->d_err_ring);
  if (d->h_ring) (void)hipHostFree(d->h_ring);
  if (d->d_used_units) (void)hipFree(d->d_used_units);
  if (!d->streams_pooled) {
    for (int i = 0; i < d->n_copy_streams; i++) if (d->copy_streams[i]) (void)hipStreamDestroy(d->copy_streams[i]);
    for (int i = 0; i < 2; i++) if (d->upload_streams[i]) (void)hipStreamDestroy(d->upload_streams[i]);
  }
  if (d->kstream_index < 0) (void)hipStreamDestroy(d->stream);      // (a pooled kernel stream lives as long as the process)
  delete d;
}

int de265hip_dpb_alloc(de265hip_decoder* d, int slot, int width, int height, int bdY, int bdC)
{ return de265hip_dpb_alloc_ex(d, slot, width, height, bdY, bdC, 1); }

int de265hip_dpb_alloc_ex(de265hip_decoder* d, int slot, int width, int height, int bdY, int bdC, int chroma_format_idc)
{
  if (!d || slot < 0 || slot >= DE265HIP_MAX_DPB_SLOTS) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  if (chroma_format_idc < 0 || chroma_format_idc > 3) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  if (width <= 0 || height <= 0 || (width & 7) || (height & 7) || bdY < 8 || bdY > 12 || bdC < 8 || bdC > 12)
    return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  if ((bdY > 8) != (bdC > 8)) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  std::lock_guard<std::mutex> lk(d->mu);          // (run_picture snapshots the slot table under the same lock)
  return alloc_slot(d->slots[slot], width, height, bdY, bdC, chroma_format_idc);
}

static int plane_geom(de265hip_decoder* d, int slot, int c, Slot** s, int* w, int* h, size_t* bpp)
{
  if (!d || slot < 0 || slot >= DE265HIP_MAX_DPB_SLOTS || c < 0 || c > 2 || !d->slots[slot].valid)
    return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  *s = &d->slots[slot];
  *w = c ? (*s)->cw() : (*s)->w; *h = c ? (*s)->ch() : (*s)->h;
  *bpp = px_bytes(c ? (*s)->bdC : (*s)->bdY);
  return 0;
}

int de265hip_dpb_upload(de265hip_decoder* d, int slot, int c, const void* src, ptrdiff_t stride_bytes)
{
  Slot* s; int w, h; size_t bpp;
  int rc = plane_geom(d, slot, c, &s, &w, &h, &bpp); if (rc) return rc;
  if (w == 0 || h == 0) return DE265HIP_OK;               // (a chroma plane of a monochrome picture)
  out_settle(*s);
  if (s->dl_done && s->dl_waited != s->dl_seq) HIPCHK(hipEventSynchronize(s->dl_done), DE265HIP_ERROR_DECODING);
  HIPCHK(sync_all_lanes(d), DE265HIP_ERROR_DECODING);
  { std::lock_guard<std::mutex> lk(d->mu); slot_settled(*s); }
  HIPCHK(hipMemcpy2D(s->pl[c].ptr, s->pl[c].stride * bpp, src, (size_t)stride_bytes, w * bpp, h, hipMemcpyHostToDevice),
         DE265HIP_ERROR_DECODING);
  return 0;
}

// de265hip_dpb_fill: every sample of the slot's three planes set to one value per component - what libde265 does on the host for a
// reference picture the stream does not contain (generate_unavailable_reference_picture, decctx.cc:1408-1434:
// fill_image(1 << (bitDepth - 1)), image.cc fill_image) before it decodes pictures that predict from it.  Device memsets on the
// decoder's stream; like an upload, a synchronisation point of the decoder's lanes.
int de265hip_dpb_fill(de265hip_decoder* d, int slot, int y, int cb, int cr)
{
  const int val[3] = { y, cb, cr };
  for (int c = 0; c < 3; c++) {
    Slot* s; int w, h; size_t bpp;
    const int rc = plane_geom(d, slot, c, &s, &w, &h, &bpp); if (rc) return rc;
    if (val[c] < 0 || val[c] >= (1 << (c ? s->bdC : s->bdY))) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
    if (w == 0 || h == 0) continue;                      // (a chroma plane of a monochrome picture)
    if (c == 0) {
      out_settle(*s);
      if (s->dl_done && s->dl_waited != s->dl_seq) HIPCHK(hipEventSynchronize(s->dl_done), DE265HIP_ERROR_DECODING);
      HIPCHK(sync_all_lanes(d), DE265HIP_ERROR_DECODING);
      std::lock_guard<std::mutex> lk(d->mu); slot_settled(*s);
    }
    // (the rows with their padding: the plane is one allocation of stride x height samples)
    const size_t n = (size_t)s->pl[c].stride * (size_t)h;
    if (bpp == 2) HIPCHK(hipMemsetD16Async((hipDeviceptr_t)s->pl[c].ptr, (unsigned short)val[c], n, d->stream), DE265HIP_ERROR_DECODING);
    else HIPCHK(hipMemsetD8Async((hipDeviceptr_t)s->pl[c].ptr, (unsigned char)val[c], n, d->stream), DE265HIP_ERROR_DECODING);
  }
  HIPCHK(hipStreamSynchronize(d->stream), DE265HIP_ERROR_DECODING);
  return 0;
}

int de265hip_dpb_download(de265hip_decoder* d, int slot, int c, void* dst, ptrdiff_t stride_bytes)
{
  Slot* s; int w, h; size_t bpp;
  int rc = plane_geom(d, slot, c, &s, &w, &h, &bpp); if (rc) return rc;
  HIPCHK(sync_all_lanes(d), DE265HIP_ERROR_DECODING);
  if (w == 0 || h == 0) return DE265HIP_OK;
  HIPCHK(hipMemcpy2D(dst, (size_t)stride_bytes, s->pl[c].ptr, s->pl[c].stride * bpp, w * bpp, h, hipMemcpyDeviceToHost),
         DE265HIP_ERROR_DECODING);
  return 0;
}

/* Pinned host memory for the planes decoded pictures are copied into (de265.h:325-343: a libde265 host installs it as its
 * de265_image_allocation), so that de265hip_dpb_download_async is a true DMA that overlaps the next pictures' kernels. */
void* de265hip_host_alloc(size_t bytes)
{
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void de265hip_host_free(void* p) { if (p) (void)hipHostFree(p); }

// Copy-out by a kernel that stores into the pinned planes (device-accessible host memory).  Why not hipMemcpyAsync: with the
// uploads of the next pictures keeping the DMA engines busy the runtime performs a device-to-pinned-host copy as a blit kernel
// of 256 workgroups x 512 threads PER PLANE (rocprofv3 kernel trace of bench.py's with_copy_out leg: __amd_rocclr_copyBuffer,
// 3 per picture, 2 048 wavefronts each parked on PCIe writes next to the reconstruction kernels); 40 workgroups of 256 threads
// carry the link as well (tools/exp/d2hprobe.hip: 16 workgroups 47 GB/s, 32: 52, 64: 54, the runtime's copy 54-55), and one
// launch takes all three planes.
struct OutJob {
  const uint8_t* src[3]; uint8_t* dst[3];
  uint32_t row_bytes[3], rows[3], spitch[3], dpitch[3];      // a plane whose rows are contiguous on both sides: ONE row of all its bytes
};

__global__ __launch_bounds__(256) void k_copy_out(OutJob J)
{
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x, nthr = gridDim.x * 256;
  __builtin_amdgcn_s_setprio(3);                                 // (a handful of wavefronts among thousands: let them issue)
  for (int c = 0; c < 3; c++) {
    const uint32_t rb = J.row_bytes[c], rows = J.rows[c];
    if (!rb || !rows) continue;
    const uint32_t upr = rb >> 4, tail = rb & 15;              // 16-byte units per row; a tail only where rows == 1 (checked by the host)
    const uint64_t n = (uint64_t)upr * rows;
    if (rows == 1) {
      const uint4* sp = reinterpret_cast<const uint4*>(J.src[c]); uint4* dp = reinterpret_cast<uint4*>(J.dst[c]);
      uint64_t i = tid;
      for (; i + 3ull * nthr < n; i += 4ull * nthr) {            // four loads in flight per lane: few wavefronts carry the link
        const uint4 a = sp[i], b = sp[i + nthr], c4 = sp[i + 2ull * nthr], e = sp[i + 3ull * nthr];
        dp[i] = a; dp[i + nthr] = b; dp[i + 2ull * nthr] = c4; dp[i + 3ull * nthr] = e;
      }
      for (; i < n; i += nthr) dp[i] = sp[i];
      if (tid < tail) J.dst[c][(size_t)upr * 16 + tid] = J.src[c][(size_t)upr * 16 + tid];
    } else {
      for (uint64_t i = tid; i < n; i += nthr) {
        const uint32_t r = (uint32_t)(i / upr), k = (uint32_t)(i - (uint64_t)r * upr);
        *reinterpret_cast<uint4*>(J.dst[c] + (size_t)r * J.dpitch[c] + (size_t)k * 16) =
            *reinterpret_cast<const uint4*>(J.src[c] + (size_t)r * J.spitch[c] + (size_t)k * 16);
      }
    }
  }
}

// The upload of a picture's staged records by a kernel that READS the pinned staging buffer (DE265HIP_UPLOAD=kernel): leaves the
// DMA engines to the copy-outs, and returns to the build thread at once (hipMemcpyAsync of pinned memory holds its caller)
static hipError_t upload_by_kernel(void* dev_dst, const void* pinned_src, size_t bytes, hipStream_t st, int grid)
{
  OutJob J; memset(&J, 0, sizeof(J));
  J.src[0] = static_cast<const uint8_t*>(pinned_src); J.dst[0] = static_cast<uint8_t*>(dev_dst);
  J.row_bytes[0] = (uint32_t)bytes; J.rows[0] = 1;
  hipLaunchKernelGGL(k_copy_out, dim3(grid), dim3(256), 0, st, J);
  return hipGetLastError();
}
static int upload_kernel_grid()
{
  static const int g = [] { const char* e = d265_env("DE265HIP_UPLOAD"); if (!e || strncmp(e, "kernel", 6)) return 0; const int n = e[6] == ':' ? atoi(e + 7) : 16; return std::max(1, std::min(512, n)); }();
  return g;
}

// can the device store into [dst, dst + bytes)?  (pinned by de265hip_host_alloc / hipHostMalloc / hipHostRegister: yes, at the
// device pointer returned; pageable memory: no - and the runtime says so with an error that must not stay behind)
static uint8_t* device_view_of_host(void* dst)
{
  hipPointerAttribute_t a; memset(&a, 0, sizeof(a));
  if (hipPointerGetAttributes(&a, dst) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
  return static_cast<uint8_t*>(a.devicePointer);
}

static int out_stream_behind_picture(de265hip_decoder* d, Slot* s)
{
  if (!d->out_stream) {
    if (d->streams_pooled && (d->out_stream = pooled_out_stream(d->device)) != nullptr) d->out_pooled = true;
    else HIPCHK(hipStreamCreateWithPriority(&d->out_stream, hipStreamNonBlocking, d->prio_low), DE265HIP_ERROR_DECODING);
    HIPCHK(hipEventCreateWithFlags(&d->out_fence, hipEventDisableTiming), DE265HIP_ERROR_DECODING);
  }
  // behind everything enqueued on the decoder's stream so far (the picture's kernels), but not in FRONT of what comes next
  if (d->n_lanes > 1) {                                   // (lanes: behind the picture that was decoded into the slot)
    if (s->writer_lane != -1 && s->written) HIPCHK(hipStreamWaitEvent(d->out_stream, s->written, 0), DE265HIP_ERROR_DECODING);
  } else {
    HIPCHK(hipEventRecord(d->out_fence, d->stream), DE265HIP_ERROR_DECODING);
    HIPCHK(hipStreamWaitEvent(d->out_stream, d->out_fence, 0), DE265HIP_ERROR_DECODING);
  }
  return 0;
}

// the decoder's output thread (deferred copy-outs; comment at g_out_mu)
static void out_thread_main(de265hip_decoder* d)
{
  (void)hipSetDevice(d->device);
  for (;;) {
    de265hip_decoder::OutJobRec job;
    {
      std::unique_lock<std::mutex> lk(g_out_mu);
      g_out_cv.wait(lk, [&] { return d->out_stop || !d->out_jobs.empty(); });
      if (d->out_jobs.empty()) return;
      job = d->out_jobs.front(); d->out_jobs.pop_front();
    }
    bool ok = hipEventSynchronize(job.picture_done) == hipSuccess;        // the picture is in its slot (blocking-sync event: no spinning)
    for (int c = 0; c < 3 && ok; c++) {
      if (!job.dst[c]) continue;
      // rows that are contiguous on both sides leave as ONE linear DMA: enqueueing a pitched copy costs the host about 2 us per row
      // (4.3 ms for the three planes of a 4K picture, tools/exp/e2e_profile.sh), more than the whole host stage can afford
      if (job.dpitch[c] == job.row_bytes[c] && job.spitch[c] == job.row_bytes[c])
        ok = hipMemcpyAsync(job.dst[c], job.src[c], job.row_bytes[c] * job.rows[c], hipMemcpyDeviceToHost, d->out_stream) == hipSuccess;
      else
        ok = hipMemcpy2DAsync(job.dst[c], job.dpitch[c], job.src[c], job.spitch[c], job.row_bytes[c], job.rows[c], hipMemcpyDeviceToHost, d->out_stream) == hipSuccess;
    }
    // (recorded also when a call failed: whoever waits for this copy-out must not wait for ever; the failure is kept)
    if (hipEventRecord(job.s->dl_ev[job.ring], d->out_stream) != hipSuccess) ok = false;
    {
      std::lock_guard<std::mutex> lk(g_out_mu);
      if (!ok) d->out_failed = true;
      job.s->out_issued++;
      d->out_events.push_back(job.picture_done);
    }
    g_out_cv.notify_all();
  }
}

int de265hip_dpb_download_planes_async(de265hip_decoder* d, int slot, void* const dst[3], const ptrdiff_t stride_bytes[3], uint64_t* copy_out_id)
{
  if (copy_out_id) *copy_out_id = 0;
  if (!dst || !stride_bytes) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  Slot* s = nullptr; int w[3], h[3]; size_t bpp[3];
  for (int c = 0; c < 3; c++) { const int rc = plane_geom(d, slot, c, &s, &w[c], &h[c], &bpp[c]); if (rc) return rc; }
  // DE265HIP_OUT_COPY: "deferred" (default): the output thread enqueues a DMA copy once the picture is done; "dma": hipMemcpyAsync
  // behind a stream wait, at once (the round-3 form: the runtime makes blit kernels of it); "kernel": k_copy_out behind a stream
  // wait; "none": the events without the bytes (experiments; results invalid)
  static const char* mode_env = d265_env("DE265HIP_OUT_COPY");
  static const int out_grid = [] { const char* e = d265_env("DE265HIP_OUT_GRID"); const int g = e ? atoi(e) : 16; return std::max(1, std::min(1024, g)); }();
  const bool use_kernel = mode_env && !strcmp(mode_env, "kernel");
  const bool deferred = !mode_env || !strcmp(mode_env, "deferred");
  static const bool no_copy = mode_env && !strcmp(mode_env, "none");
  OutJob J; memset(&J, 0, sizeof(J));
  de265hip_decoder::OutJobRec R; memset(&R, 0, sizeof(R));
  bool by_kernel[3] = { false, false, false }; bool any_kernel = false, any = false;
  for (int c = 0; c < 3; c++) {
    if (!dst[c] || w[c] == 0 || h[c] == 0) continue;       // (not wanted; a chroma plane of a monochrome picture)
    any = true;
    const size_t rb = (size_t)w[c] * bpp[c], sp = s->pl[c].stride * bpp[c], dp = (size_t)stride_bytes[c];
    if (stride_bytes[c] < (ptrdiff_t)rb) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
    R.dst[c] = dst[c]; R.src[c] = s->pl[c].ptr; R.dpitch[c] = dp; R.spitch[c] = sp; R.row_bytes[c] = rb; R.rows[c] = (size_t)h[c];
    if (!use_kernel) continue;
    uint8_t* dv = device_view_of_host(dst[c]);
    if (!dv) continue;
    const uint8_t* sv = reinterpret_cast<const uint8_t*>(s->pl[c].ptr);
    const bool flat = sp == rb && dp == rb;
    if (((uintptr_t)dv | (uintptr_t)sv) & 15) continue;
    if (!flat && ((rb | sp | dp) & 15)) continue;
    if ((flat ? rb * (size_t)h[c] : std::max(sp, dp)) > 0xFFFFFFFFu) continue;
    J.src[c] = sv; J.dst[c] = dv;
    if (flat) { J.row_bytes[c] = (uint32_t)(rb * (size_t)h[c]); J.rows[c] = 1; }
    else { J.row_bytes[c] = (uint32_t)rb; J.rows[c] = (uint32_t)h[c]; J.spitch[c] = (uint32_t)sp; J.dpitch[c] = (uint32_t)dp; }
    by_kernel[c] = any_kernel = true;
  }
  if (!any) return DE265HIP_OK;
  std::lock_guard<std::mutex> lk(d->mu);
  if (!d->out_stream) {
    if (d->streams_pooled && (d->out_stream = pooled_out_stream(d->device)) != nullptr) d->out_pooled = true;
    else HIPCHK(hipStreamCreateWithPriority(&d->out_stream, hipStreamNonBlocking, d->prio_low), DE265HIP_ERROR_DECODING);
    HIPCHK(hipEventCreateWithFlags(&d->out_fence, hipEventDisableTiming), DE265HIP_ERROR_DECODING);
  }
  const int r = (int)(s->dl_seq % Slot::kDlRing);
  if (!s->dl_ev[r]) HIPCHK(hipEventCreateWithFlags(&s->dl_ev[r], hipEventDisableTiming), DE265HIP_ERROR_DECODING);
  if (deferred) {
    // an event behind the picture's kernels for the output thread to wait for on the host
    hipEvent_t pd = nullptr;
    { std::lock_guard<std::mutex> lo(g_out_mu); if (!d->out_events.empty()) { pd = d->out_events.back(); d->out_events.pop_back(); } }
    if (!pd) HIPCHK(hipEventCreateWithFlags(&pd, hipEventDisableTiming | hipEventBlockingSync), DE265HIP_ERROR_DECODING);
    hipStream_t ws = d->stream;
    if (d->n_lanes > 1) {
      if (s->writer_lane >= 0) ws = lane_st(d, s->writer_lane);
      else if (s->writer_lane == kForeignWriter && s->written) HIPCHK(hipStreamWaitEvent(ws, s->written, 0), DE265HIP_ERROR_DECODING);
    }
    HIPCHK(hipEventRecord(pd, ws), DE265HIP_ERROR_DECODING);
    R.s = s; R.ring = r; R.picture_done = pd;
    s->dl_done = s->dl_ev[r];                             // (recorded by the output thread: out_settle before anybody uses it)
    s->dl_seq++;
    R.id = s->dl_seq;
    s->dl_ev_seq[r] = s->dl_seq; s->dl_ev_err_idx[r] = s->err_idx; s->dl_ev_err_seq[r] = s->err_seq;
    if (copy_out_id) *copy_out_id = s->dl_seq;
    {
      std::lock_guard<std::mutex> lo(g_out_mu);
      s->out_queued++;
      d->out_jobs.push_back(R);
      if (!d->out_thread_started) { d->out_thread_started = true; d->out_thread = std::thread(out_thread_main, d); }
    }
    g_out_cv.notify_all();
    return 0;
  }
  { const int rc = out_stream_behind_picture(d, s); if (rc) return rc; }
  if (any_kernel && !no_copy) {
    hipLaunchKernelGGL(k_copy_out, dim3(out_grid), dim3(256), 0, d->out_stream, J);
    HIPCHK(hipGetLastError(), DE265HIP_ERROR_DECODING);
  }
  for (int c = 0; c < 3 && !no_copy; c++) {
    if (!R.dst[c] || by_kernel[c]) continue;
    if (R.dpitch[c] == R.row_bytes[c] && R.spitch[c] == R.row_bytes[c])
      HIPCHK(hipMemcpyAsync(R.dst[c], R.src[c], R.row_bytes[c] * R.rows[c], hipMemcpyDeviceToHost, d->out_stream), DE265HIP_ERROR_DECODING);
    else
      HIPCHK(hipMemcpy2DAsync(R.dst[c], R.dpitch[c], R.src[c], R.spitch[c], R.row_bytes[c], R.rows[c], hipMemcpyDeviceToHost, d->out_stream), DE265HIP_ERROR_DECODING);
  }
  HIPCHK(hipEventRecord(s->dl_ev[r], d->out_stream), DE265HIP_ERROR_DECODING);
  s->dl_done = s->dl_ev[r];
  s->dl_seq++;
  s->dl_ev_seq[r] = s->dl_seq; s->dl_ev_err_idx[r] = s->err_idx; s->dl_ev_err_seq[r] = s->err_seq;
  if (copy_out_id) *copy_out_id = s->dl_seq;
  return 0;
}

int de265hip_dpb_download_async(de265hip_decoder* d, int slot, int c, void* dst, ptrdiff_t stride_bytes)
{
  if (c < 0 || c > 2) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  void* planes[3] = { nullptr, nullptr, nullptr }; ptrdiff_t strides[3] = { 0, 0, 0 };
  planes[c] = dst; strides[c] = stride_bytes;
  if (!dst) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  return de265hip_dpb_download_planes_async(d, slot, planes, strides, nullptr);
}

int de265hip_dpb_wait(de265hip_decoder* d, int slot)
{
  if (!d || slot < 0 || slot >= DE265HIP_MAX_DPB_SLOTS || !d->slots[slot].valid) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  hipEvent_t ev; uint64_t seq;
  out_settle(d->slots[slot]);
  { std::lock_guard<std::mutex> lk(d->mu); Slot& s = d->slots[slot]; ev = s.dl_done; seq = s.dl_seq; if (!ev || s.dl_waited == seq) return 0; }
  HIPCHK(hipEventSynchronize(ev), DE265HIP_ERROR_DECODING);        // outside the lock: another thread may be enqueueing the next picture
  // the error word of the picture that was decoded into the slot (a k_run dependency wait that expired): its own word, so a
  // failure of a picture built ahead - or launched behind it - never lands on this one
  uint32_t err = 0;
  int eidx = -1;
  { std::lock_guard<std::mutex> lk(d->mu); Slot& s = d->slots[slot]; if (s.err_idx >= 0 && d->ring_owner[s.err_idx] == s.err_seq) eidx = s.err_idx; }
  if (eidx >= 0) HIPCHK(hipMemcpy(&err, d->d_err_ring + eidx, 4, hipMemcpyDeviceToHost), DE265HIP_ERROR_DECODING);
  { std::lock_guard<std::mutex> lk(d->mu); Slot& s = d->slots[slot]; if (s.dl_seq == seq) s.dl_waited = seq; }
  { std::lock_guard<std::mutex> lo(g_out_mu); if (d->out_failed) err = 1; }
  return err ? DE265HIP_ERROR_DECODING : 0;
}

int de265hip_dpb_wait_copy_out(de265hip_decoder* d, int slot, uint64_t copy_out_id)
{
  if (!d || slot < 0 || slot >= DE265HIP_MAX_DPB_SLOTS || !d->slots[slot].valid || copy_out_id == 0) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  hipEvent_t ev = nullptr; int eidx = -1;
  out_settle(d->slots[slot], copy_out_id);
  {
    std::lock_guard<std::mutex> lk(d->mu);
    Slot& s = d->slots[slot];
    if (copy_out_id > s.dl_seq) return 0;                    // (from before the slot's re-allocation, which waited for it: free_slot)
    const int r = (int)((copy_out_id - 1) % Slot::kDlRing);
    // its own event; one that has been recorded again since belongs to a LATER copy-out on the same stream: waiting for it waits
    // for this one too (and the picture's error word has long been handed on: decoder_sync reports it)
    ev = s.dl_ev[r];
    if (s.dl_ev_seq[r] == copy_out_id && s.dl_ev_err_idx[r] >= 0 && d->ring_owner[s.dl_ev_err_idx[r]] == s.dl_ev_err_seq[r]) eidx = s.dl_ev_err_idx[r];
    if (!ev) return 0;
  }
  HIPCHK(hipEventSynchronize(ev), DE265HIP_ERROR_DECODING);
  uint32_t err = 0;
  if (eidx >= 0) HIPCHK(hipMemcpy(&err, d->d_err_ring + eidx, 4, hipMemcpyDeviceToHost), DE265HIP_ERROR_DECODING);
  { std::lock_guard<std::mutex> lk(d->mu); Slot& s = d->slots[slot]; if (s.dl_seq == copy_out_id) s.dl_waited = copy_out_id; }
  { std::lock_guard<std::mutex> lo(g_out_mu); if (d->out_failed) err = 1; }
  return err ? DE265HIP_ERROR_DECODING : 0;
}

int de265hip_dpb_info(de265hip_decoder* d, int slot, int* width, int* height, int* bit_depth_luma, int* bit_depth_chroma)
{
  if (!d || slot < 0 || slot >= DE265HIP_MAX_DPB_SLOTS || !d->slots[slot].valid) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  const Slot& s = d->slots[slot];
  if (width) *width = s.w;
  if (height) *height = s.h;
  if (bit_depth_luma) *bit_depth_luma = s.bdY;
  if (bit_depth_chroma) *bit_depth_chroma = s.bdC;
  return 0;
}

int de265hip_dpb_chroma_format(de265hip_decoder* d, int slot)
{
  if (!d || slot < 0 || slot >= DE265HIP_MAX_DPB_SLOTS || !d->slots[slot].valid) return -1;
  return d->slots[slot].cf;
}

int de265hip_dpb_plane(de265hip_decoder* d, int slot, int c, void** dev_ptr, ptrdiff_t* stride_bytes)
{
  Slot* s; int w, h; size_t bpp;
  int rc = plane_geom(d, slot, c, &s, &w, &h, &bpp); if (rc) return rc;
  if (dev_ptr) *dev_ptr = s->pl[c].ptr;
  if (stride_bytes) *stride_bytes = (ptrdiff_t)(s->pl[c].stride * bpp);
  return 0;
}

int de265hip_dpb_copy(de265hip_decoder* sd, int ss, de265hip_decoder* dd, int ds)
{
  if (!sd || !dd || ss < 0 || ss >= DE265HIP_MAX_DPB_SLOTS || ds < 0 || ds >= DE265HIP_MAX_DPB_SLOTS || !sd->slots[ss].valid ||
      (sd == dd && ss == ds))
    return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  const Slot& S = sd->slots[ss];
  {
    std::lock_guard<std::mutex> lk(dd->mu);
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(dd->device) != hipSuccess) return DE265HIP_ERROR_DECODING;
    int rc = alloc_slot(dd->slots[ds], S.w, S.h, S.bdY, S.bdC, S.cf);
    (void)hipSetDevice(prev);
    if (rc) return rc;
  }
  Slot& D = dd->slots[ds];
  hipEvent_t ev = nullptr;
  HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming), DE265HIP_ERROR_OUT_OF_MEMORY);
  int rc = 0;
  // The copy runs on the SOURCE decoder's stream.  It must also come behind what the destination decoder has already
  // enqueued - pictures that still read the slot's old content as a reference, or write it - and behind a copy-out of
  // that old content (de265hip_dpb_download_async): overwriting a live reference slot is what an open-GOP hand-over does.
  // (lanes: behind everything EVERY lane of either decoder has queued - the copy is a synchronisation point of both)
  if (sd != dd) {
    std::lock_guard<std::mutex> lk(dd->mu);
    if (!wait_for_all_lanes(dd, sd->stream)) rc = DE265HIP_ERROR_DECODING;
  }
  if (sd->n_lanes > 1) {
    std::lock_guard<std::mutex> lk(sd->mu);
    if (!wait_for_all_lanes(sd, sd->stream)) rc = DE265HIP_ERROR_DECODING;
  }
  {
    std::lock_guard<std::mutex> lk(dd->mu);
    out_settle(D);
    if (!rc && D.dl_done && D.dl_waited != D.dl_seq && hipStreamWaitEvent(sd->stream, D.dl_done, 0) != hipSuccess) rc = DE265HIP_ERROR_DECODING;
  }
  for (int c = 0; c < 3 && !rc; c++) {
    const int h = c ? S.ch() : S.h;
    const size_t bytes = (size_t)S.pl[c].stride * h * px_bytes(c ? S.bdC : S.bdY);      // same pitch on both sides (alloc_slot)
    if (!bytes) continue;
    hipError_t e = sd->device == dd->device
      ? hipMemcpyAsync(D.pl[c].ptr, S.pl[c].ptr, bytes, hipMemcpyDeviceToDevice, sd->stream)
      : hipMemcpyPeerAsync(D.pl[c].ptr, dd->device, S.pl[c].ptr, sd->device, bytes, sd->stream);
    if (e != hipSuccess) rc = DE265HIP_ERROR_DECODING;
  }
  if (!rc && (hipEventRecord(ev, sd->stream) != hipSuccess || hipStreamWaitEvent(dd->stream, ev, 0) != hipSuccess))
    rc = DE265HIP_ERROR_DECODING;
  (void)hipEventDestroy(ev);            // (released once it has completed)
  if (!rc && dd->n_lanes > 1) {         // the destination's other lanes learn of the copy through the slot's `written` event
    std::lock_guard<std::mutex> lk(dd->mu);
    if (!D.written && hipEventCreateWithFlags(&D.written, hipEventDisableTiming) != hipSuccess) rc = DE265HIP_ERROR_OUT_OF_MEMORY;
    if (!rc && hipEventRecord(D.written, sd->stream) != hipSuccess) rc = DE265HIP_ERROR_DECODING;
    D.writer_lane = kForeignWriter; D.written_seq = ++dd->launch_seq;
    for (int l = 0; l < kMaxLanes; l++) D.rd_valid[l] = false;
  }
  if (!rc && sd->n_lanes > 1) {         // and the source's lanes that the copy (on its lane 0) reads the slot
    std::lock_guard<std::mutex> lk(sd->mu);
    Slot& S2 = sd->slots[ss];
    if (!S2.read_done[0] && hipEventCreateWithFlags(&S2.read_done[0], hipEventDisableTiming) != hipSuccess) rc = DE265HIP_ERROR_OUT_OF_MEMORY;
    if (!rc && hipEventRecord(S2.read_done[0], sd->stream) != hipSuccess) rc = DE265HIP_ERROR_DECODING;
    if (!rc) S2.rd_valid[0] = true;
  }
  return rc;
}

void de265hip_picture_free(de265hip_picture* p)
{
  if (!p) return;
  if (p->enq.up_sig) (void)hsa_upload_wait(p->enq.up_sig);      // (freed before it was enqueued: the DMA engine may still be writing its arena)
  if (de265hip_decoder* dec = p->dec) {                   // (an orphan has no device side left: only the handle goes)
    std::lock_guard<std::mutex> lk(dec->mu);
    if (p->enq.up_sig) {
      for (auto& b : dec->stage_pool) if (b.sig == p->enq.up_sig) { b.sig = 0; if (b.state == 2) b.state = 0; }      // (it has arrived: the staging buffer is free)
      dec->free_sigs.push_back(p->enq.up_sig); p->enq.up_sig = 0;
    }
    dec->live.erase(std::remove(dec->live.begin(), dec->live.end(), p), dec->live.end());
    // no synchronisation: the arena goes back to the pool behind an event on the decoder's stream, and its next
    // upload waits for that event on the copy stream
    release_arena(dec, p->arena_buf, lane_st(dec, p->lane));     // (its launches on other lanes precede the latest one: they wrote the same slot)
    if (p->ring_idx >= 0) dec->ring_free.push_back(p->ring_idx);   // (first in, first out: its error word stands for thousands of pictures to come)
    if (p->enq.pending && !p->enq.uploaded_by_builder) for (auto& b : dec->stage_pool) if (b.ptr == p->enq.host_base && b.owner == p->ring_seq) b.state = 0;      // (built, never enqueued)
    if (p->uploaded) { if (dec->free_events.size() < 256) dec->free_events.push_back(p->uploaded); else (void)hipEventDestroy(p->uploaded); }
  }
  delete p;
}

static int finish_scan(de265hip_picture* pic);
static thread_local bool g_defer_enqueue = false;        // de265hip_picture_build_host: the build stops before its HIP calls
static int picture_build_impl(de265hip_decoder* dec, int dst_slot, const de265hip_picture_desc* d, de265hip_picture** out);

int de265hip_picture_build(de265hip_decoder* dec, int dst_slot, const de265hip_picture_desc* d,
                           de265hip_picture** out)
{
  if (!dec || !d || !out) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  // builds in progress are counted: a picture joins `live` only at the end of its build, and until then the geometry guard
  // of another worker's build must still see it (a change of picture size between consecutive submissions)
  { std::lock_guard<std::mutex> lk(dec->mu); dec->building++; }
  const int rc = picture_build_impl(dec, dst_slot, d, out);
  { std::lock_guard<std::mutex> lk(dec->mu); dec->building--; }
  return rc;
}

static int picture_build_impl(de265hip_decoder* dec, int dst_slot, const de265hip_picture_desc* d, de265hip_picture** out)
{
  *out = nullptr;
  const de265hip_pic_params& p = d->params;
  // extended precision: the reference's transform path hard-codes extended_precision_processing_flag = 0 (transform.cc:535)
  if (p.chroma_format_idc < 0 || p.chroma_format_idc > 3 || p.extended_precision_processing_flag) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  // monochrome: pictures without prediction units only - the reference's inter path addresses the chroma planes whatever
  // the format (motion.cc:296-305), which a monochrome picture does not have: no defined result to match
  if (p.chroma_format_idc == 0 && d->n_pus) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  if (p.cross_component_prediction_enabled_flag && p.chroma_format_idc != 3) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  if ((p.bit_depth_luma > 8) != (p.bit_depth_chroma > 8)) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  if (p.bit_depth_luma < 8 || p.bit_depth_luma > 12 || p.bit_depth_chroma < 8 || p.bit_depth_chroma > 12 ||
      p.width <= 0 || p.height <= 0 || (p.width & 7) || (p.height & 7) ||
      p.log2_ctb_size < 4 || p.log2_ctb_size > 6 || p.log2_min_tb_size < 2 || p.log2_min_tb_size > 5 ||
      p.log2_min_tb_size > p.log2_ctb_size || p.num_tile_columns < 1 || p.num_tile_columns > 20 ||
      p.num_tile_rows < 1 || p.num_tile_rows > 22 || d->n_slices < 1 || !d->slices || !d->ctbs ||
      !d->blk_flags || !d->blk_qp_y)
    return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  if (p.scaling_list_enable_flag && !d->scaling_factors) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  if (dst_slot < 0 || dst_slot >= DE265HIP_MAX_DPB_SLOTS) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  int rc = 0;
  PhaseTimer pt;
  if (!dec->dry) {
    // Builds run ahead of launches (de265hip_pipeline_*, several host threads): a slot or the spare that already holds
    // planes of ANOTHER geometry may still be read or written by pictures that are built but not launched yet, so it is
    // not touched here - de265hip_picture_run re-allocates it when this picture's turn comes (launches are in decode
    // order, and the re-allocation waits for the device).  An empty slot is allocated right away, so that the planes of
    // a picture exist once it is built (de265hip_dpb_plane / upload of its initial content).
    // With no picture waiting for its first launch nothing can be disturbed either (the synchronous build -> upload -> run use).
    std::lock_guard<std::mutex> lk(dec->mu);
    // (a picture joins `live` only at the END of its build: builds in progress on other pipeline workers count as pending too)
    bool pending = dec->building > 1;                  // (this build is one of them)
    for (const de265hip_picture* q : dec->live) pending = pending || q->n_launched == 0;
    if (!dec->slots[dst_slot].valid || !pending) rc = alloc_slot(dec->slots[dst_slot], p.width, p.height, p.bit_depth_luma, p.bit_depth_chroma, p.chroma_format_idc);
    if (!rc && (!dec->spare.valid || !pending)) rc = alloc_slot(dec->spare, p.width, p.height, p.bit_depth_luma, p.bit_depth_chroma, p.chroma_format_idc);
  }
  if (rc) return rc;
  pt.mark("slots");
  // CtbAddrRStoTS / TileIdRS / MinTbAddrZS depend on the picture size, CTB / min TB size and the tile grid only: the same for
  // every picture of a sequence.  One cached copy per host thread (4K: 522 240 z-scan addresses, 4 ms to compute).
  struct GeoKey { int32_t w, h, lc, lt, nc, nr; uint16_t cb[24], rb[24]; };
  static thread_local GeoKey g_key = {};
  static thread_local Geometry g_cached;
  static thread_local bool g_valid = false;
  GeoKey key; memset(&key, 0, sizeof(key));
  key.w = p.width; key.h = p.height; key.lc = p.log2_ctb_size; key.lt = p.log2_min_tb_size; key.nc = p.num_tile_columns; key.nr = p.num_tile_rows;
  memcpy(key.cb, p.col_bd, sizeof(key.cb)); memcpy(key.rb, p.row_bd, sizeof(key.rb));
  if (!g_valid || memcmp(&key, &g_key, sizeof(key)) != 0) {
    g_valid = false;
    rc = make_geometry(p, g_cached);
    if (rc) return rc;
    g_key = key; g_valid = true;
  }
  const Geometry& g = g_cached;
  if (d->n_ctbs != g.ctbs_w * g.ctbs_h) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  for (int i = 0; i < d->n_ctbs; i++)
    if (d->ctbs[i].slice_idx >= d->n_slices) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;

  de265hip_picture* pic = new (std::nothrow) de265hip_picture();
  if (!pic) return DE265HIP_ERROR_OUT_OF_MEMORY;
  pic->dec = dec; pic->dst_slot = dst_slot; pic->params = p;
  PicDev& P = pic->P;
  P.width = p.width; P.height = p.height; P.bd_luma = p.bit_depth_luma; P.bd_chroma = p.bit_depth_chroma;
  P.log2_ctb = p.log2_ctb_size; P.ctbs_w = g.ctbs_w; P.ctbs_h = g.ctbs_h; P.w4 = g.w4; P.h4 = g.h4;
  P.strong_intra = p.strong_intra_smoothing_enable_flag; P.pcm_lf_disable = p.pcm_loop_filter_disable_flag;
  P.weighted_pred = p.weighted_pred_flag; P.weighted_bipred = p.weighted_bipred_flag;
  P.cb_qp_offset = p.pic_cb_qp_offset; P.cr_qp_offset = p.pic_cr_qp_offset;
  P.lf_across_tiles = p.loop_filter_across_tiles_enabled_flag; P.scaling_list = p.scaling_list_enable_flag;
  const int cf = p.chroma_format_idc, subw = (cf == 3 || cf == 0) ? 1 : 2, subh = cf == 1 ? 2 : 1;      // SubWidthC, SubHeightC
  const int cwid = cf ? p.width / subw : 0, chei = cf ? p.height / subh : 0;          // (monochrome: no chroma samples; a TU with c_idx > 0 is out of range)
  P.chroma_format = cf; P.csw = subw - 1; P.csh = subh - 1; P.cwidth = cwid; P.cheight = chei;
  P.smooth_luma = !p.intra_smoothing_disabled_flag; P.smooth_chroma = !p.intra_smoothing_disabled_flag && cf == 3;
  P.implicit_rdpcm = p.implicit_rdpcm_enabled_flag; P.xcc_enabled = p.cross_component_prediction_enabled_flag;
  P.wp_shift_luma = p.high_precision_offsets_enabled_flag ? 0 : p.bit_depth_luma - 8;
  P.wp_shift_chroma = p.high_precision_offsets_enabled_flag ? 0 : p.bit_depth_chroma - 8;
  const bool rext_tools = p.implicit_rdpcm_enabled_flag || p.transform_skip_rotation_enabled_flag || p.cross_component_prediction_enabled_flag;
  if (dec->intra_levels && rext_tools) { delete pic; return DE265HIP_ERROR_NOT_IMPLEMENTED; }   // (the level-launch schedule knows the Main tools only)

  pt.mark("geometry");
  // ---- Where the TU records are scanned.  Default: on the DEVICE, behind the upload (scan_core.h / k_scan.hip: availability,
  // needed units, runs, in-run levels, run edges, task lists as kernels with one thread per TU / CTB / run) - the host copies
  // the raw records into the staging buffer and is done.  The round-3 host scan below stays for the level-launch parity
  // schedule (DE265HIP_INTRA_MODE=levels), as the A/B switch DE265HIP_HOST_SCAN=1, and as the reference the equivalence tests
  // hold the passes to (tests/test_scan_equivalence.py).
  BuildScratch& SC = g_scratch;
  const bool dev_scan = dec->dev_scan && !dec->intra_levels;
  int max_level = 0, max_rl = 0, n_front = 0, n_mailboxes = 0;
  int64_t alg_resid = 0, alg_intra = 0, alg_intra_front = 0, sum_lvls = 0;
  size_t n_resid = 0;
  std::vector<TuTask> sorted;
  std::vector<RunTask>& runs = SC.runs; std::vector<uint32_t>& run_deps = SC.run_deps; std::vector<uint32_t>& slots = SC.slots;
  std::vector<TuTask>& run_tus = SC.run_tus;
  std::vector<uint32_t>& mbx = SC.mbx; std::vector<uint32_t>& mb_segs = SC.mb_segs;
  runs.clear(); run_deps.clear(); slots.clear(); run_tus.clear(); mbx.clear(); mb_segs.clear(); SC.l0.clear(); SC.l0_rext.clear();
  const int lc = p.log2_ctb_size, lt = p.log2_min_tb_size;
  const bool host_checks_positions = dec->dry || dec->intra_levels;
  pic->level_start.assign(2, 0);
  if (!dev_scan) {
  // dependencies between intra TUs from the units each mode reads (DE265HIP_NO_MODE_DEPS: from every available unit)
  const bool mode_deps = d265_env("DE265HIP_NO_MODE_DEPS") == nullptr;
  const bool merge_runs = d265_env("DE265HIP_NO_MERGE") == nullptr;
  ensure_used_units();
  // ---- TU scan: tasks, intra availability, dependency levels, runs.  One linear pass over the TU records on flat,
  // reused arrays (BuildScratch): no allocation and no page fault in the steady state.
  // (SC: hoisted above)
  const int map_w[3] = { g.w4, (cwid + 3) / 4, (cwid + 3) / 4 };
  const int map_h[3] = { g.h4, (chei + 3) / 4, (chei + 3) / 4 };
  // The cell maps are not cleared per picture (6 MB at 4K): a cell counts only if its run id is of THIS build - ids start at
  // an epoch base that grows from build to build (cleared when the ids would wrap or the geometry changes).
  {
    bool fresh = SC.epoch_base > 0x70000000 || SC.epoch_base < 0;
    for (int c = 0; c < 3; c++) fresh = fresh || SC.cells[c].size() != (size_t)map_w[c] * map_h[c];
    if (fresh) { for (int c = 0; c < 3; c++) SC.cells[c].assign((size_t)map_w[c] * map_h[c], Cell{ -1, 0, 0 }); SC.epoch_base = 0; }
  }
  const int32_t E = SC.epoch_base;                     // a cell's run id r (>= 0) is stored as E + r
  SC.epoch_base = E + d->n_tus + 1;                    // (the next build's ids lie above everything this one can write, even if it fails half way)
  max_level = 0;
  alg_resid = 0; alg_intra = 0; alg_intra_front = 0;
  // runs: maximal intervals of the per-component intra TU sequence inside one CTB in which
  // every TU reads from the run so far (see k_run); independent TUs start a new run
  std::vector<RunB>& rb = SC.rb; rb.clear();
  SC.it.clear(); SC.it_xcc.clear(); SC.dep_val.clear(); SC.dep_next.clear();
  SC.level_hist.assign(2, 0);
  SC.all_tasks.clear(); SC.all_levels.clear();
  int cur_run[3] = { -1, -1, -1 };
  // dense intra (no inter PUs at all): one run per CTB and component, fewest hand-offs on the z-scan chain.
  // (32x32 run boxes: more runs in flight but 13-15 instead of 8 run levels on a 4K B picture: slower; every picture uses 64x64)
  const int run_box = 64;
  pic->run_box = run_box;
  // CTBs of one slice and tile share a group word (the availability tests of intrapred.cc:486-508 compare exactly these two)
  SC.ctb_group.resize((size_t)d->n_ctbs);
  for (int a = 0; a < d->n_ctbs; a++) SC.ctb_group[a] = (uint32_t)d->ctbs[a].slice_addr_rs | ((uint32_t)g.tile_id[a] << 16);
  const uint32_t* ctb_group = SC.ctb_group.data();
  // (lc, lt: hoisted above)
  const int* zs = g.min_tb_zs.data();
  const bool cip = p.constrained_intra_pred_flag != 0;
  const bool one_tile = p.num_tile_columns == 1 && p.num_tile_rows == 1;
  const int ctb_mask = (1 << lc) - 1;
  {
    const int mkey = lc | (lt << 4) | (cf << 8);
    if (SC.avail_memo.size() != 2 * 4 * 16 * 16 || SC.avail_memo_key != mkey) { SC.avail_memo.assign(2 * 4 * 16 * 16, 0); SC.avail_memo_key = mkey; }
  }
  // pre-pass: how many level-0 tasks of each size there will be - inter TUs with residual, then the residual-only copies of the
  // intra TUs: [32x32 | 16x16 | 8x8 | 4x4], inside a size the inter TUs first - so that both are written to their final place
  // Range-extension tools of a TU (D265_RX_* bits; 0 for every TU of a Main / Main10 picture): such a TU's residual is
  // computed by k_resid_rext instead of the tuned residual kernels.
  //  RDPCM: implicit for intra TUs (mode 10 / 26 with transform skip or bypass, slice.cc:3456-3461), explicit for inter TUs;
  //  rotation: 4x4 transform-skip / bypass TUs of intra CUs - the reference looks the CU up at the TU's position in samples
  //    of ITS COMPONENT through an accessor that takes luma samples (transform.cc:393-395): reproduced;
  //  cross-component prediction: res_scale_val of a chroma TU;  transform skip beyond 8x8 (log2_max_transform_skip_block_size).
  auto rx_bits = [&](const de265hip_tu& tu) -> int {
    if (!(tu.flags & (DE265HIP_TU_TSKIP | DE265HIP_TU_BYPASS | DE265HIP_TU_EXPLICIT_RDPCM)) && !tu.res_scale_val) return 0;
    const bool cbf = (tu.flags & DE265HIP_TU_CBF) && tu.n_coeff;
    const bool ts_or_bp = tu.flags & (DE265HIP_TU_TSKIP | DE265HIP_TU_BYPASS);
    int rx = 0;
    if (cbf && ts_or_bp) {
      if (tu.flags & DE265HIP_TU_INTRA) {
        if (p.implicit_rdpcm_enabled_flag && (tu.intra_mode == 10 || tu.intra_mode == 26)) rx |= tu.intra_mode == 26 ? D265_RX_RDPCM_V : D265_RX_RDPCM_H;
      } else if (tu.flags & DE265HIP_TU_EXPLICIT_RDPCM) rx |= (tu.flags & DE265HIP_TU_EXPLICIT_RDPCM_VERT) ? D265_RX_RDPCM_V : D265_RX_RDPCM_H;
      if (p.transform_skip_rotation_enabled_flag && tu.log2_size == 2 &&
          (d->blk_flags[(tu.x0 >> 2) + (tu.y0 >> 2) * g.w4] & DE265HIP_BLK_INTRA)) rx |= D265_RX_ROTATE;
      if ((tu.flags & DE265HIP_TU_TSKIP) && !(tu.flags & DE265HIP_TU_BYPASS) && tu.log2_size > 3) rx |= 0x80;      // (big transform skip: no tool bit of its own)
    }
    if (tu.c_idx && tu.res_scale_val) rx |= D265_RX_XCC;
    return rx;
  };
  // (the inter TUs of each size are collected as they come and copied to their place behind the scan, when the number of
  //  residual-only intra copies of each size is known: one pass over the TU records instead of two)
  int n_ro_size[4] = { 0, 0, 0, 0 };
  for (auto& v : SC.l0_inter) { v.n = 0; if (!v.ensure((size_t)d->n_tus + 1)) { delete pic; return DE265HIP_ERROR_OUT_OF_MEMORY; } }
  static_assert(sizeof(de265hip_tu) == 16 && offsetof(de265hip_tu, n_coeff) == offsetof(TuTask, n_coeff) && offsetof(de265hip_tu, coeff_offset) == offsetof(TuTask, coeff_offset) &&
                offsetof(de265hip_tu, qp) == offsetof(TuTask, qp) && offsetof(de265hip_tu, res_scale_val) == offsetof(TuTask, run_level) && offsetof(de265hip_tu, flags) == offsetof(TuTask, flags),
                "a TU record is the first half of its task");
  int64_t n_plain = 0;                                     // level-0 tasks of plain inter TUs (added to the level histogram behind the loop)
  SC.it.reserve((size_t)d->n_tus);
  SC.l0_rext.clear();
  // (host_checks_positions: hoisted above)
  int last_luma_tu = -1;                                   // most recent luma TU record (cross-component prediction reads its residual)
  // Cr mirrors Cb: the Cr TU of an intra CU sits at its Cb TU's place with its size and mode, and - by induction over the
  // decode order - among Cr neighbours that are the images of the Cb TU's neighbours: availability, needed units, levels and
  // the run decision of the Cb TU hold for it (run ids through SC.mirror), and the Cr cell map is never consulted.  A
  // descriptor that breaks the pattern (a Cr intra TU without its Cb twin right before it, or the other way round) makes the
  // build start over without the shortcut.  DE265HIP_NO_CR_MIRROR=1: always the long way (the arenas are identical).
  const bool cr_mirror = cf != 0 && !g_no_cr_mirror && d265_env("DE265HIP_NO_CR_MIRROR") == nullptr;
  int cbq_head = 0, cbq_n = 0;
  bool mirror_broken = false;
  SC.mirror.clear();
  int n_tasks = 0;
  int prod[40];
  for (int i = 0; i < d->n_tus; i++) {
    const de265hip_tu& tu = d->tus[i];
    if (tu.log2_size < 2 || tu.log2_size > 5) { delete pic; return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE; }
    const int nT = 1 << tu.log2_size;
    const int cw = tu.c_idx ? cwid : p.width, ch = tu.c_idx ? chei : p.height;
    // (one test of the OR of the conditions instead of a branch per condition)
    if ((unsigned)(tu.c_idx > 2) | (unsigned)((tu.x0 | tu.y0) & (nT - 1)) | (unsigned)(tu.x0 + nT > cw) | (unsigned)(tu.y0 + nT > ch) | (unsigned)(tu.qp < 0) |
        (unsigned)(((nT * (tu.c_idx ? subw : 1)) >> lc) > 1) | (unsigned)(((nT * (tu.c_idx ? subh : 1)) >> lc) > 1) |      // (a quadtree leaf: aligned to its size, inside one CTB)
        (unsigned)((tu.flags & DE265HIP_TU_CBF) && (((int64_t)tu.coeff_offset + tu.n_coeff > d->n_coeffs) | (tu.n_coeff > nT * nT)))) {
      delete pic; return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
    }
    // (positions inside the TU's block: checked - and folded into the block - on the device, k_check_coeffs behind the upload;
    //  the level-launch schedule reads the descriptor's order, not the level-0 lists that kernel walks: checked here)
    if ((tu.flags & DE265HIP_TU_CBF) && host_checks_positions) {
      const uint16_t* cp = d->coeff_pos + tu.coeff_offset;
      unsigned worst = 0;
      const int nc = tu.n_coeff;
      if (nc <= 4) { for (int k = 0; k < nc; k++) worst |= cp[k]; }                        // (an OR bounds the maximum: nT*nT is a power of two)
      else for (int k = 0; k < nc; k++) worst = std::max<unsigned>(worst, cp[k]);        // (vectorises)
      if (worst >= (unsigned)(nT * nT)) { delete pic; return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE; }
    }
    if (tu.c_idx == 0) last_luma_tu = i;
    if (!(tu.flags & (DE265HIP_TU_INTRA | DE265HIP_TU_TSKIP | DE265HIP_TU_BYPASS | DE265HIP_TU_EXPLICIT_RDPCM)) && !tu.res_scale_val) {
      // a plain inter TU (most TUs of a picture with inter PUs): its level-0 task is the record itself plus sixteen zero bytes
      if (!(tu.flags & DE265HIP_TU_CBF)) continue;                                   // nothing to reconstruct
      BuildScratch::TaskBuf& v = SC.l0_inter[tu.log2_size - 2];
      TuTask& t = v.p[v.n++];
      memcpy(&t, &tu, 16); memset((uint8_t*)&t + 16, 0, 16);
      if (tu.n_coeff == 0) t.flags &= (uint8_t)~DE265HIP_TU_CBF;
      else alg_resid += std::min<int64_t>(4 * (int64_t)tu.n_coeff, 2 * (int64_t)nT * nT) + 2 * (int64_t)px_bytes(tu.c_idx ? p.bit_depth_chroma : p.bit_depth_luma) * nT * nT;
      n_plain++; n_tasks++;
      if (dec->intra_levels) { SC.all_tasks.push_back(t); SC.all_levels.push_back(0); }
      continue;
    }
    const int rx = rx_bits(tu);
    if (!(tu.flags & (DE265HIP_TU_INTRA | DE265HIP_TU_CBF)) && !(rx & D265_RX_XCC)) continue;       // nothing to reconstruct
    // cross-component prediction: the luma TU of the same position and size comes right before the chroma TUs (4:4:4,
    // slice.cc:3699-3750); its coefficient list is what k_resid_rext recomputes the luma residual from
    uint64_t luma_info = 0; int rx_luma = 0;
    if (rx & D265_RX_XCC) {
      if (cf != 3 || last_luma_tu < 0) { delete pic; return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE; }
      const de265hip_tu& lt = d->tus[last_luma_tu];
      if (lt.x0 != tu.x0 || lt.y0 != tu.y0 || lt.log2_size != tu.log2_size) { delete pic; return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE; }
      const int lrx = rx_bits(lt);
      const bool lcbf = (lt.flags & DE265HIP_TU_CBF) && lt.n_coeff;
      luma_info = (uint64_t)lt.coeff_offset | ((uint64_t)(lcbf ? lt.n_coeff : 0) << 32) | ((uint64_t)(uint8_t)lt.qp << 48) | ((uint64_t)lt.flags << 56);
      rx_luma = ((lrx & D265_RX_ROTATE) ? D265_RX_LUMA_ROT : 0) |
                (((lrx & D265_RX_RDPCM_V) ? 2 : ((lrx & D265_RX_RDPCM_H) ? 1 : 0)) << D265_RX_LUMA_RDPCM_SHIFT);
    }
    TuTask t; memset(&t, 0, sizeof(t));
    t.x0 = tu.x0; t.y0 = tu.y0; t.log2_size = tu.log2_size; t.c_idx = tu.c_idx; t.flags = tu.flags;
    t.intra_mode = tu.intra_mode; t.qp = tu.qp; t.n_coeff = (tu.flags & DE265HIP_TU_CBF) ? tu.n_coeff : 0;
    t.coeff_offset = tu.coeff_offset;
    if (t.n_coeff == 0) t.flags &= (uint8_t)~DE265HIP_TU_CBF;
    int level = 0;
    const size_t bpp = px_bytes(tu.c_idx ? p.bit_depth_chroma : p.bit_depth_luma);
    if (tu.flags & DE265HIP_TU_INTRA) {
      const int c = tu.c_idx, sbw = c ? subw : 1, sbh = c ? subh : 1;
      const int m = tu.intra_mode < 35 ? tu.intra_mode : 1;
      t.angle = k_intra_angle[m];
      t.inv_angle = (m >= 11 && m <= 25 && k_intra_angle[m] < 0) ? k_inv_angle[m - 11] : 0;
      // -- neighbour availability (8.4.4.2.2; intrapred.cc:437-527 preproc, :577-688 fill_from_image) as a unit bit mask.
      // The left column beside the TU, the row above it and the corner precede the TU in z-scan order whenever they lie in
      // the same slice and tile (Morton order: the highest differing coordinate bit is set in the TU's own position), so
      // only the below-left and above-right units need the MinTbAddrZS comparison.
      const int xB = tu.x0, yB = tu.y0, xL = xB * sbw, yL = yB * sbh;
      // (a 4:2:2 chroma TU covers a 2:1 luma area: the shortcut for the left / top neighbours is not taken there)
      const bool full_z = c != 0 && cf == 2;
      const int cx = xL >> lc, cy = yL >> lc, ctu = cx + cy * g.ctbs_w;
      const int mw = map_w[c], corner = nT >> 1;
      uint64_t mask = 0;
      int lev = 0, llev = 0, n_prod = 0;
      bool foreign = false;
      int r = -1, kind = 2;                                 // kind: 0 extends the current run of its component, 1 joins run r, 2 starts a run
      const Cell* cells = SC.cells[c].data();
      bool mirrored = false;
      if (cr_mirror && c != 1 && (c == 0 ? cbq_n != 0 : cbq_n == 0)) { mirror_broken = true; break; }      // a Cb TU left without its Cr TU / a Cr TU without its Cb TU
      if (cr_mirror && c == 2) {
        const BuildScratch::CbScan& q = SC.cbq[cbq_head];
        if (q.x0 != tu.x0 || q.y0 != tu.y0 || q.log2 != tu.log2_size || q.mode != tu.intra_mode) { mirror_broken = true; break; }
        mask = q.mask; lev = q.level - 1; llev = q.llev; foreign = q.foreign != 0; kind = q.kind; n_prod = q.n_prod;
        for (int k2 = 0; k2 < n_prod; k2++) { prod[k2] = SC.mirror[q.prod[k2]]; if (prod[k2] < 0) mirror_broken = true; }
        r = kind == 0 ? cur_run[2] : (kind == 1 ? SC.mirror[q.run] : -1);
        if ((kind != 2 && r < 0) || mirror_broken) { mirror_broken = true; break; }
        cbq_head = (cbq_head + 1) & 3; cbq_n--;
        mirrored = true;
      }
      if (!mirrored) {
      const uint32_t own = ctb_group[ctu];
      const bool aL = xL > 0 && ctb_group[((xL - 1) >> lc) + cy * g.ctbs_w] == own;
      const bool aT = yL > 0 && ctb_group[cx + ((yL - 1) >> lc) * g.ctbs_w] == own;
      const bool aTL = xL > 0 && yL > 0 && ctb_group[((xL - 1) >> lc) + ((yL - 1) >> lc) * g.ctbs_w] == own;
      const bool aTR = yL > 0 && (xL + nT * sbw < p.width) && ctb_group[((xL + nT * sbw) >> lc) + ((yL - 1) >> lc) * g.ctbs_w] == own;
      int nBottom = (p.height - yL + sbh - 1) >> (sbh - 1); if (nBottom > 2 * nT) nBottom = 2 * nT;      // (sbw, sbh are 1 or 2)
      int nRight = (p.width - xL + sbw - 1) >> (sbw - 1);   if (nRight > 2 * nT) nRight = 2 * nT;
      const int cur = zs[(xL >> lt) + (size_t)(yL >> lt) * g.tbs_w];
      auto intra_ok = [&](int xs, int ys) {                 // constrained_intra_pred: only samples of intra CUs (intrapred.cc:612-615)
        return !cip || (d->blk_flags[((xs * sbw) >> 2) + ((ys * sbh) >> 2) * g.w4] & DE265HIP_BLK_INTRA);
      };
      auto z_ok = [&](int xs, int ys) { return zs[((xs * sbw) >> lt) + (size_t)((ys * sbh) >> lt) * g.tbs_w] <= cur; };
      auto take = [&](int u, int, int) { mask |= 1ull << u; };
      // 4x4 map cell of neighbour unit u (left column bottom -> top, corner, top row left -> right)
      const int cell_l = ((xB - 1) >> 2) + ((yB >> 2) + corner - 1) * mw, cell_t = (xB >> 2) + ((yB >> 2) - 1) * mw;
      auto cell_of = [&](int u) { return u < corner ? cell_l - u * mw : (u == corner ? cell_t - 1 : cell_t + (u - corner - 1)); };
      // One tile, no constrained intra prediction, the four neighbouring CTBs of the TU's slice and nothing clipped by the picture:
      // what is left is the z-scan order, a function of the TU's size and position inside its CTB - remembered per thread
      // (filled by the general code below the first time a position is seen; an entry is never 0: the left column is there)
      uint64_t* memo = nullptr;
      if (one_tile && !cip && !full_z && aL && aT && aTL && aTR && yL + 2 * nT * sbh <= p.height && xL + 2 * nT * sbw <= p.width) {
        memo = &SC.avail_memo[((((c ? 1 : 0) * 4 + (tu.log2_size - 2)) * 16 + ((yL & ctb_mask) >> 2)) * 16) + ((xL & ctb_mask) >> 2)];
        mask = *memo;
      }
      if (!mask) {
        if (aL) {
          for (int y = nT - 1; y >= 0; y -= 4) if ((!full_z || z_ok(xB - 1, yB + y)) && intra_ok(xB - 1, yB + y)) take((2 * nT - 1 - y) >> 2, xB - 1, yB + y);
          for (int y = nBottom - 1; y >= nT; y -= 4)
            if (z_ok(xB - 1, yB + y) && intra_ok(xB - 1, yB + y)) take((2 * nT - 1 - y) >> 2, xB - 1, yB + y);
        }
        if (aTL && (!full_z || z_ok(xB - 1, yB - 1)) && intra_ok(xB - 1, yB - 1)) take(corner, xB - 1, yB - 1);
        if (aT) for (int x = 0; x < nT; x += 4) if ((!full_z || z_ok(xB + x, yB - 1)) && intra_ok(xB + x, yB - 1)) take(corner + 1 + (x >> 2), xB + x, yB - 1);
        if (aTR)
          for (int x = nT; x < nRight; x += 4)
            if (z_ok(xB + x, yB - 1) && intra_ok(xB + x, yB - 1)) take(corner + 1 + (x >> 2), xB + x, yB - 1);
        if (memo) *memo = mask;
      }
      // -- dependencies: only the units the mode reads (mode_deps), or every available unit
      // (4:4:4 chroma is smoothed like luma: it takes luma's table, a superset of what it reads)
      uint64_t need = mode_deps ? needed_units(g_used_units[tu.log2_size - 2][m][c == 0 || cf == 3], mask) : mask;
      bool reads_cur = false;
      const int crun = cur_run[c];
      const uint64_t need0 = need;
      for (; need; need &= need - 1) {
        const Cell C = cells[cell_of(__builtin_ctzll(need))];
        const int cr = C.run - E;                           // (< 0: no intra TU of this picture covers the cell)
        foreign = foreign || cr < 0;
        if (cr >= 0) {
          lev = std::max(lev, (int)C.lvl);
          if (cr == crun) { llev = std::max(llev, (int)C.llvl); reads_cur = true; }
          bool seen = false;
          for (int q = 0; q < n_prod; q++) seen = seen || prod[q] == cr;
          if (!seen) prod[n_prod++] = cr;
        }
      }
      llev += 1;
      // -- which run: the current one of its component if the TU lies in the same CTB and reads from it (the run structure is
      // decided on the FULL neighbourhood, so that an all-intra CTB stays one run per component) ...
      r = crun;
      bool extends = r >= 0 && rb[r].ctu == ctu && rb[r].n_tus < 255;       /* RUN_MAX_TUS of k_run; positions + 1 fit a byte */
      if (extends && !reads_cur) {                          // (a needed unit of the current run settles it: the needed units are available ones)
        extends = false;
        for (uint64_t mm = mask & ~need0; mm && !extends; mm &= mm - 1) extends = cells[cell_of(__builtin_ctzll(mm))].run == E + r;
      }
      // A TU that cannot extend the current run but reads from exactly ONE run joins that run instead of starting its
      // own (e.g. an intra CU next to an intra CU of the neighbouring CTB, or below one decoded long ago): a hand-over
      // between two runs costs ~12 us of dependent memory round trips, an in-run level 0.4 us, and a B picture's intra time
      // is its run-level depth times that hand-over (measured: 44 / 48 / 98 / 156 us at 2 / 3 / 7 / 11 levels).  The run
      // graph stays acyclic: the joined run gains no producer, and every other edge still points from a later-created
      // run to an earlier-created one.  The run is then no interval of the decode order any more; its TUs keep their
      // order by in-run level.  DE265HIP_NO_MERGE=1 switches it off.
      // (Measured, 4K Main10: joining only single-producer TUs takes a B picture from 8 813 to 8 086 runs and 7 to 6 levels,
      //  99 -> 90 us, I pictures unchanged; with several producers allowed the B pictures gain nothing more (8 067 runs) and
      //  the I picture loses - chroma runs merge across CTBs, 6 120 -> 4 142 runs, 1.94 -> 2.04 ms.  So: one producer only.)
      bool merged = false;
      if (!extends && merge_runs && n_prod == 1) {
        const int x = prod[0];
        const RunB& X = rb[x];
        const int bw = std::max(X.x1, tu.x0 + nT) - std::min(X.x0, (int)tu.x0);
        const int bh = std::max(X.y1, tu.y0 + nT) - std::min(X.y0, (int)tu.y0);
        // (round 4: only a run of the TU's own CTB - a run then lives inside one CTB, and what a CTB's TUs join is decided
        //  from that CTB's records alone: the device-side scan works the CTBs off independently)
        if (X.c == c && X.ctu == ctu && X.n_tus < 255 && bw <= run_box && bh <= run_box) {
          // in-run level: behind everything of X it reads; the needed cells are not kept: every cell of X the TU's
          // neighbourhood touches is a safe upper bound
          int lx = 0;
          const int ux0 = std::max(0, (int)tu.x0 - 4) >> 2, uy0 = std::max(0, (int)tu.y0 - 4) >> 2;
          const int ux1 = std::min(cw - 1, (int)tu.x0 + 2 * nT + 3) >> 2, uy1 = std::min(ch - 1, (int)tu.y0 + 2 * nT + 3) >> 2;
          if (uy0 < (tu.y0 >> 2)) for (int x4 = ux0; x4 <= ux1; x4++) { const Cell C = cells[x4 + (size_t)uy0 * mw]; if (C.run == E + x) lx = std::max(lx, (int)C.llvl); }
          if (ux0 < (tu.x0 >> 2)) for (int y4 = uy0; y4 <= uy1; y4++) { const Cell C = cells[ux0 + (size_t)y4 * mw]; if (C.run == E + x) lx = std::max(lx, (int)C.llvl); }
          if (lx + 1 <= 250) { r = x; llev = lx + 1; merged = true; }
        }
      }
      kind = extends ? 0 : (merged ? 1 : 2);
      }                                                     // (!mirrored)
      t.avail = mask;
      level = lev + 1;
      if (kind == 2) {
        r = (int)rb.size();
        RunB nr; memset(&nr, 0, sizeof(nr));
        nr.c = c; nr.ctu = ctu; nr.x0 = tu.x0; nr.y0 = tu.y0; nr.x1 = tu.x0 + nT; nr.y1 = tu.y0 + nT; nr.est = 1;
        nr.head = nr.tail = -1; nr.dep_head = nr.dep_tail = -1;
        rb.push_back(nr);
        SC.mirror.push_back(-1);
        cur_run[c] = r;
        llev = 1;
        if (mirrored) SC.mirror[SC.cbq[(cbq_head + 3) & 3].run] = r;      // (the entry just taken)
      }
      if (cr_mirror && c == 1) {                            // remember what was decided, for the Cr TU of this place
        if (cbq_n == 4) { mirror_broken = true; break; }
        BuildScratch::CbScan& q = SC.cbq[(cbq_head + cbq_n) & 3]; cbq_n++;
        q.x0 = tu.x0; q.y0 = tu.y0; q.log2 = tu.log2_size; q.mode = tu.intra_mode; q.kind = (uint8_t)kind; q.foreign = foreign;
        q.run = r; q.level = level; q.llev = llev; q.n_prod = n_prod; q.mask = mask;
        for (int k2 = 0; k2 < n_prod; k2++) q.prod[k2] = prod[k2];
      }
      RunB& R = rb[r];
      if (foreign) R.foreign = 1;
      R.x0 = std::min(R.x0, (int)tu.x0); R.y0 = std::min(R.y0, (int)tu.y0);
      R.x1 = std::max(R.x1, tu.x0 + nT); R.y1 = std::max(R.y1, tu.y0 + nT);
      R.wx1 = std::max(R.wx1, tu.x0 + 2 * nT); R.wy1 = std::max(R.wy1, tu.y0 + 2 * nT);    // top-right / bottom-left reach
      for (int q = 0; q < n_prod; q++) {
        const int pr = prod[q];
        if (pr == r) continue;
        bool seen = false;
        for (int e = R.dep_head; e >= 0 && !seen; e = SC.dep_next[e]) seen = SC.dep_val[e] == pr;
        if (seen) continue;
        const int e = (int)SC.dep_val.size();
        SC.dep_val.push_back(pr); SC.dep_next.push_back(-1);
        if (R.dep_tail >= 0) SC.dep_next[R.dep_tail] = e; else R.dep_head = e;
        R.dep_tail = e; R.n_deps++;
        R.est = std::max(R.est, rb[pr].est + 1);
      }
      {                                                     // the run's TUs: a list in decode order
        const int ti = (int)SC.it.size();
        t.resid_offset = 0xFFFFFFFFu; t.run_level = (uint8_t)(llev - 1); t.pad3 = (uint8_t)(rx | rx_luma);
        SC.it.push_back(t);
        t.resid_offset = 0; t.run_level = 0; t.pad3 = 0;     // (the level-launch schedule keeps its own copy of t below)
        if (rx & D265_RX_XCC) SC.it_xcc.push_back(BuildScratch::ItXcc{ ti, tu.res_scale_val, luma_info });
        if (R.tail >= 0) SC.it[R.tail].resid_offset = (uint32_t)ti; else R.head = ti;
        R.tail = ti; R.n_tus++;
      }
      if (!(cr_mirror && c == 2)) {                         // (nothing reads the Cr map while Cr mirrors Cb)
        Cell* wc = SC.cells[c].data();
        const Cell v{ E + r, (uint16_t)level, (uint16_t)llev };
        for (int y = tu.y0 >> 2; y < (tu.y0 + nT) >> 2; y++)
          for (int x = tu.x0 >> 2; x < (tu.x0 + nT) >> 2; x++) wc[x + (size_t)y * mw] = v;
      }
      alg_intra += (int64_t)bpp * (4 * nT + 1) + (int64_t)bpp * nT * nT;
      R.alg += (int64_t)bpp * (4 * nT + 1) + (int64_t)bpp * nT * nT;
      if (level >= 65535) { delete pic; return DE265HIP_ERROR_NOT_IMPLEMENTED; }
      if (!rx && (t.flags & DE265HIP_TU_CBF)) n_ro_size[tu.log2_size - 2]++;      // its residual-only copy: a level-0 task of that size
    } else if (rx) {                                        // level 0, a range-extension tool: k_resid_rext's list
      TuTask rt = t;
      rt.pad3 = (uint8_t)(rx | rx_luma); rt.angle = tu.res_scale_val; rt.avail = luma_info;
      SC.l0_rext.push_back(rt);
    } else
      { BuildScratch::TaskBuf& v = SC.l0_inter[tu.log2_size - 2]; v.p[v.n++] = t; }      // level 0: residual added into the (inter-predicted) picture
    if (t.flags & DE265HIP_TU_CBF)
      alg_resid += std::min<int64_t>(4 * (int64_t)t.n_coeff, 2 * (int64_t)nT * nT) +
                   ((tu.flags & DE265HIP_TU_INTRA) ? 0 : 2 * (int64_t)bpp * nT * nT);
    if (level > max_level) { max_level = level; SC.level_hist.resize(max_level + 2, 0); }
    SC.level_hist[level + 1]++;
    n_tasks++;
    if (dec->intra_levels) { SC.all_tasks.push_back(t); SC.all_levels.push_back(level); }
  }
  if (cr_mirror && (mirror_broken || cbq_n != 0)) {         // not the pattern the shortcut relies on: once more, the long way
    delete pic;
    g_no_cr_mirror = true;
    const int rc2 = picture_build_impl(dec, dst_slot, d, out);
    g_no_cr_mirror = false;
    return rc2;
  }
  SC.level_hist[1] += (int)n_plain;
  size_t ro_cur[4];
  {
    size_t at = 0, inter_at[4];
    for (int k = 3; k >= 0; k--) { inter_at[k] = at; at += SC.l0_inter[k].n; ro_cur[k] = at; at += n_ro_size[k]; pic->n_l0_size[k] = (int)SC.l0_inter[k].n + n_ro_size[k]; }
    SC.l0.resize(at);
    for (int k = 0; k < 4; k++) if (SC.l0_inter[k].n) memcpy(SC.l0.data() + inter_at[k], SC.l0_inter[k].p, SC.l0_inter[k].n * sizeof(TuTask));
  }
  TuTask* l0p = SC.l0.data();
  pt.mark("tu_scan");
  pic->level_start.assign(SC.level_hist.begin(), SC.level_hist.begin() + max_level + 2);
  for (int l = 0; l <= max_level; l++) pic->level_start[l + 1] += pic->level_start[l];
  // the level-sorted task array is only needed by the level-launch schedule (DE265HIP_INTRA_MODE=levels)
  // (sorted: hoisted above)
  if (dec->intra_levels) {
    sorted.resize(SC.all_tasks.size());
    std::vector<int> cursor(pic->level_start.begin(), pic->level_start.end() - 1);
    for (size_t i = 0; i < SC.all_tasks.size(); i++) sorted[cursor[SC.all_levels[i]]++] = SC.all_tasks[i];
  }
  pic->n_tus = n_tasks;

  // ---- runs in dependency (ticket) order: producers first
  // (runs, run_deps, slots, run_tus: hoisted above)

  run_tus.reserve(SC.it.size());
  const bool micro_off = d265_env("DE265HIP_NO_MICRO") != nullptr;
  const bool no_dense = d265_env("DE265HIP_NO_DENSE") != nullptr;    // (not per run: getenv walks the whole environment)
  const int micro_tus = d265_env("DE265HIP_MICRO_TUS") ? std::min(16, atoi(d265_env("DE265HIP_MICRO_TUS"))) : 16;   // MICRO_TUS of k_run
  int64_t dbg_foreign = 0, dbg_w[4] = { 0, 0, 0, 0 };


  const bool mailbox_on = d265_env("DE265HIP_NO_MAILBOX") == nullptr;
  // (mbx, mb_segs: hoisted above)  per run: (own mailbox, first dword of its segments); the segments
  mbx.assign(3 * rb.size(), 0xFFFFFFFFu); mb_segs.clear(); SC.mb_owner.clear();      // (+ first dword of its packets' ready epochs, phased hand-over)
  // Phased hand-over between luma runs (DE265HIP_NO_MB_PHASES=1: off): a publishing run stores each packet behind the barrier
  // epoch that completes the TU under it (its table of ready epochs), and a reading run fetches each neighbour sample at the
  // latest of at most four points of its chain that still precedes the first TU needing it - the right column's upper half
  // of the CTB to the left is there when that CTB is half done, its own lower half needs the lower half only later.
  const bool mb_phases = mailbox_on && d265_env("DE265HIP_NO_MB_PHASES") == nullptr;
  {
    for (auto& R : rb) { int l = 0; for (int e = R.dep_head; e >= 0; e = SC.dep_next[e]) l = std::max(l, rb[SC.dep_val[e]].level); R.level = l + 1; max_rl = std::max(max_rl, R.level); }
    std::vector<int>& order = SC.order; std::vector<int>& newidx = SC.newidx;
    order.resize(rb.size()); newidx.resize(rb.size());
    // micro runs (<= 16 TUs of <= 8x8 inside a 32x32 box: most runs of a picture with inter PUs) are reconstructed by
    // one wavefront each, four per workgroup and ticket; inside a level they come first (same-level runs are independent)
    // (up to 32 TUs per micro run instead of 16 - the records live one per lane - measured slower: a long run is better
    //  off with four wavefronts and in-run levels; B picture 99 -> 109 us)
    // micro runs: <= 16 TUs of <= 8x8 in a 32x32 box; 16x16 TUs too (DE265HIP_MICRO16=0: not) when the run's samples
    // fit the wavefront's residual slice (1024) and its window the wavefront's slice of the window array (k_tu.hip:
    // MICRO_P 56 columns from the 8-aligned left edge, MICRO_H 41 rows, 256 chunks of 8 samples)
    const bool micro16 = !d265_env("DE265HIP_MICRO16") || atoi(d265_env("DE265HIP_MICRO16")) != 0;
    auto is_micro = [&](const RunB& R) {
      if (micro_off || R.n_tus > micro_tus || R.x1 - R.x0 > 32 || R.y1 - R.y0 > 32) return false;
      int samples = 0; bool big = false;
      for (int ti = R.head; ti >= 0; ti = (int32_t)SC.it[ti].resid_offset) {
        const TuTask& t = SC.it[ti];
        if (t.log2_size > 4) return false;
        big = big || t.log2_size == 4; samples += 1 << (2 * t.log2_size);
      }
      if (!big) return true;
      if (!micro16 || samples > 1024) return false;
      const int ax0 = (R.x0 - 1) & ~7, wx1 = std::min(R.wx1, R.x1 + 32), wy1 = std::min(R.wy1, R.y1 + 32);
      const int cols = wx1 - ax0, rows = wy1 - (R.y0 - 1);
      return cols <= 56 && rows <= 41 && ((cols + 7) >> 3) * rows <= 256;
    };
    std::vector<uint8_t>& micro = SC.micro; micro.resize(rb.size());
    for (size_t i = 0; i < rb.size(); i++) micro[i] = is_micro(rb[i]);
    std::vector<int>& count2 = SC.count2; count2.assign(2 * (max_rl + 2) + 1, 0);
    for (size_t i = 0; i < rb.size(); i++) count2[2 * rb[i].level + (micro[i] ? 0 : 1) + 1]++;
    for (size_t l = 0; l + 1 < count2.size(); l++) count2[l + 1] += count2[l];
    for (size_t i = 0; i < rb.size(); i++) { int k = count2[2 * rb[i].level + (micro[i] ? 0 : 1)]++; order[k] = (int)i; newidx[i] = k; }
    runs.resize(rb.size());
    // Front runs: the micro runs of level 1 (no producer among the intra runs) - the first block of the run order.  They are
    // reconstructed by k_intra_front, one small workgroup each, ahead of k_run: no ticket, no flag, and the runs that read
    // from them do not list them as producers (the kernel boundary orders them).  DE265HIP_NO_FRONT=1: through k_run as all others.
    const bool front_off = d265_env("DE265HIP_NO_FRONT") != nullptr;
    n_front = 0;
    if (!front_off && !dec->intra_levels)
      while ((size_t)n_front < rb.size() && micro[order[n_front]] && rb[order[n_front]].level == 1) n_front++;
    // ticket slots: a ticket is a batch of RUN_TICKET_SLOTS slots: that many micro runs, or one ordinary run (+ empty slots)
    for (int k = 0; k < n_front; k++) alg_intra_front += rb[order[k]].alg;
    for (size_t k = (size_t)n_front; k < rb.size(); k++) {
      if (micro[order[k]]) slots.push_back((uint32_t)k | 0x80000000u);
      else {
        while (slots.size() % RUN_TICKET_SLOTS) slots.push_back(0xFFFFFFFFu);
        slots.push_back((uint32_t)k);
        for (int q = 1; q < RUN_TICKET_SLOTS; q++) slots.push_back(0xFFFFFFFFu);
      }
    }
    while (slots.size() % RUN_TICKET_SLOTS) slots.push_back(0xFFFFFFFFu);
    if (dec->drop_producer) {                            // fault injection (de265hip_debug_fault_injection): the first run somebody depends on is never executed
      int victim = -1;
      for (size_t i = 0; i < rb.size() && victim < 0; i++)
        for (int e = rb[i].dep_head; e >= 0; e = SC.dep_next[e]) if (victim < 0 && newidx[SC.dep_val[e]] >= n_front) victim = newidx[SC.dep_val[e]];
      for (uint32_t& v : slots) if (v != 0xFFFFFFFFu && (int)(v & 0x7FFFFFFFu) == victim) v = 0xFFFFFFFFu;
    }
    int tix[256];                                        // the run's TUs (indices into SC.it) in decode order
    for (size_t k = 0; k < rb.size(); k++) {
      const RunB& R = rb[order[k]];
      RunTask& o = runs[k]; memset(&o, 0, sizeof(o));
      o.x0 = (uint16_t)R.x0; o.y0 = (uint16_t)R.y0; o.x1 = (uint16_t)R.x1; o.y1 = (uint16_t)R.y1;
      o.wx1 = (uint16_t)std::min(R.wx1, R.x1 + 32); o.wy1 = (uint16_t)std::min(R.wy1, R.y1 + 32);
      int n = 0, own_samples = 0, nl = 0;
      for (int ti = R.head; ti >= 0; ti = (int32_t)SC.it[ti].resid_offset) { tix[n++] = ti; own_samples += 1 << (2 * SC.it[ti].log2_size); nl = std::max(nl, (int)SC.it[ti].run_level + 1); }
      // dense: the run's TUs cover its whole bounding box AND every available neighbour outside the box lies on the row
      // above it or the column left of it (two stacked CUs with an inter CU beside the upper one do not qualify: the
      // lower CU reads above-right samples from inside the box's row range)
      bool dense = own_samples == (R.x1 - R.x0) * (R.y1 - R.y0) && !no_dense;
      // (with the box covered, the only neighbour units that can lie outside "box + row above + column to the left" are
      //  above-right units beyond the box's right edge of TUs below its first row, and below-left units beyond its bottom
      //  edge of TUs right of its first column: two mask tests per TU)
      for (int i = 0; dense && i < n; i++) {
        const TuTask& tt = SC.it[tix[i]];
        const int nT = 1 << tt.log2_size, xB = tt.x0, yB = tt.y0, corner = nT >> 1;
        if (xB > R.x0 && yB + 2 * nT > R.y1) {                 // left column, unit u = rows yB+2nT-4u-4 .. -1 (bottom -> top)
          const int umax = std::min(corner - 1, (yB + 2 * nT - 1 - R.y1) >> 2);
          if (tt.avail & ((2ull << umax) - 1ull)) dense = false;
        }
        if (yB > R.y0 && xB + 2 * nT > R.x1) {                 // top row, unit k = columns xB+4k .. +3
          const int kmin = std::max(0, (R.x1 - xB) >> 2);
          if (kmin < corner && (tt.avail >> (corner + 1 + kmin)) & ((1ull << (corner - kmin)) - 1ull)) dense = false;
        }
      }
      // Phased hand-over: every dense ordinary luma run leaves the ready epochs of its edge packets behind (the epoch of the TU
      // under each pair of samples of its bottom row and right column), for the runs that will read its mailbox.
      // (Tried on top of it: moving the TUs that read the column left of the box to their as-late-as-possible level, from the
      //  run's TU-to-TU edges, so that the run asks for the left neighbour's last samples later in its chain: 4K all-intra
      //  picture 1.62 -> 1.56 ms, but 5 ms more host time per such picture - the edges cost as much as the scan; not kept.)
      if (mb_phases && dense && !micro[order[k]] && R.c == 0 && R.x1 - R.x0 <= 64 && R.y1 - R.y0 <= 64) {
        if (SC.rdy_tab.size() < 64 * rb.size()) SC.rdy_tab.resize(64 * rb.size());
        uint8_t* rdy = &SC.rdy_tab[64 * k];
        memset(rdy, 255, 64);                              // (255: no packet there - or, in a malformed description, no TU under it)
        for (int i = 0; i < n; i++) {
          const TuTask& tt = SC.it[tix[i]];
          const int nT = 1 << tt.log2_size;
          if (tt.y0 + nT == R.y1) memset(rdy + ((tt.x0 - R.x0) >> 1), tt.run_level, (size_t)(nT >> 1));             // packets of the bottom row under this TU
          if (tt.x0 + nT == R.x1) memset(rdy + 32 + ((tt.y0 - R.y0) >> 1), tt.run_level, (size_t)(nT >> 1));        // ... of the right column beside it
        }
      }
      o.c_idx = (uint8_t)R.c; o.micro = (uint8_t)(micro[order[k]] | (dense ? 2 : 0)); o.n_tus = (uint16_t)n;
      o.first_tu = (uint32_t)run_tus.size(); o.dep_offset = (uint32_t)run_deps.size();
      { int nd = 0; for (int e = R.dep_head; e >= 0; e = SC.dep_next[e]) nd += newidx[SC.dep_val[e]] >= n_front; o.n_deps = (uint16_t)nd; }
      o.res_offset = (uint32_t)n_resid;
      if (nl > 256) { delete pic; return DE265HIP_ERROR_NOT_IMPLEMENTED; }      // RUN_MAX_TUS of k_run
      sum_lvls += nl;
      // Order of the run's TUs in its record: one list per wavefront (list w = the TUs dealt to wavefront w), each in
      // in-run level order, then the collective list (16x16 / 32x32 TUs, reconstructed by all wavefronts together).
      // The TUs of one in-run level are independent of each other and are dealt round-robin to the wavefronts; the chain
      // passes one workgroup barrier per level.  One sort by (list, level, decode index).
      // (measured on a 4K all-intra picture: this 3.46 ms; list scheduling that keeps z-scan chains on one wavefront,
      //  with barriers only where a producer sits on another wavefront, 4.0-4.2 ms: a wavefront that runs ahead
      //  arrives late at the barrier the others need; progress counters in LDS polled by the waiting wavefronts
      //  3.9 ms: the pollers take issue slots from the working wavefronts of the other workgroups on their SIMDs;
      //  round-robin lists with barriers only at cross-wavefront edges plus early arrival of the producing wavefront:
      //  54 % fewer barriers, 3.33 instead of 3.19 ms -- the workgroup barrier is not what the chain waits for)
      uint32_t keys[256]; uint8_t rank[260];
      memset(rank, 0, (size_t)nl + 1);
      {
        const int nwv = micro[order[k]] ? 1 : dec->run_waves;
        const int n_epochs = nl > 0 ? nl - 1 : 0;
        for (int i = 0; i < n; i++) {
          const int lev = (int)SC.it[tix[i]].run_level + 1;
          const int list = (SC.it[tix[i]].log2_size > 3 && !micro[order[k]]) ? 4 : rank[lev]++ % nwv;
          keys[i] = ((uint32_t)list << 20) | ((uint32_t)lev << 8) | (uint32_t)i;
        }
        if (n > 12) {
          // counting sort by (list, level), stable in the decode index: the keys are unique and (list, level) has at most
          // 5 (nl + 1) values (an all-intra 4K picture: 5 700 runs of ~40 TUs; std::sort took 2 ms of its host stage)
          uint16_t cnt[5 * 257 + 2]; uint32_t tmp[256];
          const int nb = 5 * (nl + 1);
          memset(cnt, 0, (size_t)(nb + 1) * sizeof(cnt[0]));
          for (int i = 0; i < n; i++) cnt[(keys[i] >> 20) * (nl + 1) + ((keys[i] >> 8) & 0xFFF) + 1]++;
          for (int b = 0; b < nb; b++) cnt[b + 1] = (uint16_t)(cnt[b + 1] + cnt[b]);
          for (int i = 0; i < n; i++) tmp[cnt[(keys[i] >> 20) * (nl + 1) + ((keys[i] >> 8) & 0xFFF)]++] = keys[i];
          memcpy(keys, tmp, (size_t)n * sizeof(keys[0]));
        } else if (n > 1) std::sort(keys, keys + n);
        int pos = 0;
        for (int w = 0; w < 4; w++) {
          while (pos < n && (int)(keys[pos] >> 20) <= w) { dbg_w[w]++; pos++; }
          o.wave_end[w] = (uint16_t)pos;
        }
        if (n_epochs > 255) { delete pic; return DE265HIP_ERROR_NOT_IMPLEMENTED; }
        o.n_lvls = (uint16_t)n_epochs;                   // workgroup barriers of the run's chain
        dbg_foreign += n_epochs;
      }
      uint32_t samp = 0;
      for (int oi = 0; oi < n; oi++) {
        const int ti = tix[keys[oi] & 0xFFu];
        TuTask tt = SC.it[ti];
        const uint32_t coeff_offset = tt.coeff_offset;
        // the run's residual range is laid out by SAMPLE of the run (blocks of TUs without coefficients stay unwritten and are
        // masked by the kernels): a run's residuals are one contiguous, 16-byte aligned vector the run kernels fetch right
        // behind the run record, before they have seen a single TU record
        tt.resid_offset = (uint32_t)n_resid + samp;
        tt.coeff_offset = samp; samp += 1u << (2 * tt.log2_size);
        const int trx = tt.pad3; tt.pad3 = 0;
        if ((tt.flags & DE265HIP_TU_CBF) || (trx & D265_RX_XCC)) {
          TuTask ro = tt; ro.flags |= D265_TU_RESID_ONLY; ro.coeff_offset = coeff_offset; ro.run_level = 0;
          if (trx) {
            ro.pad3 = (uint8_t)trx; ro.angle = 0; ro.avail = 0;
            if (trx & D265_RX_XCC) {
              const auto e = std::lower_bound(SC.it_xcc.begin(), SC.it_xcc.end(), ti, [](const BuildScratch::ItXcc& a, int v) { return a.ti < v; });
              ro.angle = e->rsv; ro.avail = e->luma;
            }
            SC.l0_rext.push_back(ro);
          }
          else l0p[ro_cur[ro.log2_size - 2]++] = ro;
          tt.flags |= DE265HIP_TU_CBF;                     // (the run kernels read the residual block whenever there is one)
        }
        // (tt.run_level: the run-ordered copy carries the TU's barrier epoch = its in-run level - 1, there since the scan)
        run_tus.push_back(tt);
      }
      o.n_samples = samp;
      n_resid += (samp + 7u) & ~7u;
      for (int e = R.dep_head; e >= 0; e = SC.dep_next[e]) if (newidx[SC.dep_val[e]] >= n_front) run_deps.push_back((uint32_t)newidx[SC.dep_val[e]]);
      // Edge mailboxes (k_run): a dense ordinary run whose every neighbour sample comes from the bottom row / right column of
      // dense ordinary runs takes them from those runs' mailboxes - tagged packets the producer stores the moment its chain
      // ends - instead of waiting for the flag (store drain -> flag -> poll) and then fetching its window from the picture.
      // Its part of the deps array: [producer ids | own mailbox, once somebody reads it | segment count, segments].
      if (mailbox_on && !micro[order[k]] && dense && !R.foreign && o.n_deps > 0 && o.n_deps == R.n_deps && o.n_deps <= 8) {
        const int ax0 = ((int)o.x0 - 1) & ~7, wy0 = (int)o.y0 - 1, tile_p = (64 + 40 + 7) & ~7;      // RUN_TILE_P_OF(64) of k_run
        const int cw_ = R.c ? cwid : p.width, ch_ = R.c ? chei : p.height;
        const int wx1c = std::min((int)o.wx1, cw_), wy1c = std::min((int)o.wy1, ch_);
        uint32_t seg[2 * 16]; int nseg = 0; bool ok = true;
        for (int e = R.dep_head; e >= 0 && ok; e = SC.dep_next[e]) {
          const int pk = newidx[SC.dep_val[e]];
          const RunTask& Pq = runs[pk];
          if ((Pq.micro & 3) != 2) { ok = false; break; }                 // producer: ordinary and dense
          bool any = false;
          if ((int)Pq.y0 <= wy0 && wy0 < (int)Pq.y1) {                    // the row above the box
            const int xs = std::max((int)Pq.x0, (int)o.x0 - 1), xe = std::min((int)Pq.x1, wx1c);
            if (xs < xe) {
              if ((int)Pq.y1 - 1 != wy0) { ok = false; break; }
              seg[2 * nseg] = (uint32_t)pk | ((uint32_t)(xe - xs - 1) << 24); seg[2 * nseg + 1] = (uint32_t)(xs - Pq.x0) | ((uint32_t)(xs - ax0) << 8);
              nseg++; any = true;
            }
          }
          if ((int)Pq.x0 <= (int)o.x0 - 1 && (int)o.x0 - 1 < (int)Pq.x1) {  // the column left of it
            const int ys = std::max((int)Pq.y0, (int)o.y0), ye = std::min((int)Pq.y1, wy1c);
            if (ys < ye) {
              if ((int)Pq.x1 != (int)o.x0) { ok = false; break; }
              seg[2 * nseg] = (uint32_t)pk | ((uint32_t)(ye - ys - 1) << 24) | 0x80000000u;
              seg[2 * nseg + 1] = (uint32_t)(ys - Pq.y0) | ((uint32_t)((ys - wy0) * tile_p + ((int)o.x0 - 1 - ax0)) << 8);
              nseg++; any = true;
            }
          }
          if (!any) ok = false;
        }
        if (ok && nseg > 0) {
          o.micro |= 4;
          for (int q = 0; q < nseg; q++) {                  // producer run index -> its mailbox
            const uint32_t pk = seg[2 * q] & 0xFFFFFFu;
            if (!(runs[pk].micro & 8)) {
              runs[pk].micro |= 8; mbx[3 * pk] = (uint32_t)n_mailboxes++; SC.mb_owner.push_back(pk);
              if (mb_phases && runs[pk].c_idx == 0) {       // its packets' ready epochs: 32 of the bottom row, 32 of the right column
                const RunTask& Pq = runs[pk];
                uint8_t rdy[64];
                memcpy(rdy, &SC.rdy_tab[64 * (size_t)pk], 64);   // (left behind when the run was laid out, above)
                // at most three store points before the end of the chain (quantiles of the distinct ready epochs): a packet
                // goes out at the first of them that is not before its ready epoch, the rest when the chain ends (255)
                uint64_t seen_r[4] = { 0, 0, 0, 0 };
                for (int i = 0; i < 64; i++) seen_r[rdy[i] >> 6] |= 1ull << (rdy[i] & 63);
                uint8_t rv[256]; int nrv = 0;
                for (int wd = 0; wd < 4; wd++) for (uint64_t mm = seen_r[wd]; mm; mm &= mm - 1) {
                  const int v = 64 * wd + __builtin_ctzll(mm);
                  if (v < (int)Pq.n_lvls && v < 255) rv[nrv++] = (uint8_t)v;                                     // (epoch n_lvls is the end)
                }
                uint8_t pubs[4] = { 255, 255, 255, 255 };
                const int n_pub = std::min(3, nrv);
                for (int j = 0; j < n_pub; j++) pubs[j] = rv[((j + 1) * nrv) / n_pub - 1];
                for (int i = 0; i < 64; i++) {
                  uint8_t qv = 255;
                  for (int j = n_pub - 1; j >= 0; j--) if (rdy[i] <= pubs[j]) qv = pubs[j];
                  rdy[i] = qv;
                }
                mbx[3 * pk + 2] = (uint32_t)mb_segs.size();
                mb_segs.resize(mb_segs.size() + 17);
                memcpy(&mb_segs[mb_segs.size() - 17], rdy, 64);
                mb_segs[mb_segs.size() - 1] = (uint32_t)pubs[0] | ((uint32_t)pubs[1] << 8) | ((uint32_t)pubs[2] << 16) | (255u << 24);
              }
            }
            seg[2 * q] = (seg[2 * q] & 0xFF000000u) | mbx[3 * pk];
          }
          // -- when is each neighbour sample first needed?  Only TUs on the box's left column / top row read outside it (dense run)
          uint8_t need_row[256], need_col[256];               // by x - (x0 - 1) / y - y0; 255: never read
          uint32_t sub[2 * 48]; uint8_t sub_g[48]; int nsub = 0;
          uint8_t polls[4] = { 0, 0, 0, 0 }; int n_groups = 1;
          bool phased = mb_phases && R.c == 0 && nl >= 4;
          if (phased) {
            // (kept per 4-sample unit - a unit's samples are read together - and spread over the sample arrays at the end;
            //  row unit j: x0 + 4j .. + 3, the corner x0 - 1 on its own; column unit j: y0 + 4j .. + 3)
            uint8_t nru[40], ncu[40], ncorner = 255;
            memset(nru, 255, sizeof(nru)); memset(ncu, 255, sizeof(ncu));
            for (int i = 0; i < n; i++) {
              const TuTask& tt = SC.it[tix[i]];
              const int xB = tt.x0, yB = tt.y0;
              if (xB != R.x0 && yB != R.y0) continue;
              const int nT = 1 << tt.log2_size, corner = nT >> 1, m = tt.intra_mode < 35 ? tt.intra_mode : 1;
              const uint8_t ep = tt.run_level;
              uint64_t need = mode_deps ? needed_units(g_used_units[tt.log2_size - 2][m][1], tt.avail) : tt.avail;
              for (; need; need &= need - 1) {
                const int u = __builtin_ctzll(need);
                if (u < corner) {
                  if (xB != R.x0) continue;
                  const int j = (yB + 2 * nT - 4 * u - 4 - R.y0) >> 2;
                  if (j >= 0 && j < 40) ncu[j] = std::min(ncu[j], ep);
                } else if (u == corner) {
                  if (yB == R.y0) { if (xB == R.x0) ncorner = std::min(ncorner, ep); else { const int j = (xB - 1 - R.x0) >> 2; if (j < 40) nru[j] = std::min(nru[j], ep); } }
                  else if (xB == R.x0) { const int j = (yB - 1 - R.y0) >> 2; if (j >= 0 && j < 40) ncu[j] = std::min(ncu[j], ep); }
                } else {
                  if (yB != R.y0) continue;
                  const int j = (xB + 4 * (u - corner - 1) - R.x0) >> 2;
                  if (j >= 0 && j < 40) nru[j] = std::min(nru[j], ep);
                }
              }
            }
            need_row[0] = ncorner;
            for (int j = 0; j < 40; j++) { memset(need_row + 1 + 4 * j, nru[j], 4); memset(need_col + 4 * j, ncu[j], 4); }
            // the samples' need epochs -> at most four poll points (quantiles of the distinct values)
            const uint8_t* nbase[16];                          // per segment: need epoch of its first sample
            uint64_t seen[4] = { 0, 0, 0, 0 };
            for (int q = 0; q < nseg; q++) {
              const RunTask& Pq = runs[SC.mb_owner[seg[2 * q] & 0xFFFFFFu]];
              const int src = (int)(seg[2 * q + 1] & 63), cnt = (int)((seg[2 * q] >> 24) & 63) + 1;
              nbase[q] = (seg[2 * q] >> 31) ? need_col + ((int)Pq.y0 + src - R.y0) : need_row + ((int)Pq.x0 + src - (R.x0 - 1));
              for (int off = 0; off < cnt; off++) { const uint8_t v = nbase[q][off]; seen[v >> 6] |= 1ull << (v & 63); }
            }
            seen[3] &= ~(1ull << 63);                          // (255: never read)
            uint8_t vals[256]; int nv = 0;
            for (int wd = 0; wd < 4; wd++) for (uint64_t mm = seen[wd]; mm; mm &= mm - 1) vals[nv++] = (uint8_t)(64 * wd + __builtin_ctzll(mm));
            if (nv < 2) phased = false;
            else {
              n_groups = std::min(4, nv);
              for (int g2 = 0; g2 < n_groups; g2++) polls[g2] = vals[(g2 * nv) / n_groups];
              const int p1 = n_groups > 1 ? polls[1] : 256, p2 = n_groups > 2 ? polls[2] : 256, p3 = n_groups > 3 ? polls[3] : 256;
              auto grp_of_v = [&](int v) { return v == 255 ? 255 : (v >= p1) + (v >= p2) + (v >= p3); };      // need epoch -> group (255: nobody reads it)
              // sub-segments of one group each (samples nobody reads are left out)
              for (int q = 0; q < nseg && phased; q++) {
                const int cnt = (int)((seg[2 * q] >> 24) & 63) + 1;
                const bool col = seg[2 * q] >> 31;
                int start = 0, g_cur = grp_of_v(nbase[q][0]);
                for (int off = 1; off <= cnt; off++) {
                  const int g2 = off < cnt ? grp_of_v(nbase[q][off]) : 254;
                  if (g2 == g_cur) continue;
                  if (g_cur != 255) {
                    if (nsub == 48) { phased = false; break; }
                    sub[2 * nsub] = (seg[2 * q] & 0x80FFFFFFu) | ((uint32_t)(off - start - 1) << 24);
                    sub[2 * nsub + 1] = ((seg[2 * q + 1] & 63u) + (uint32_t)start) | (((seg[2 * q + 1] >> 8) + (uint32_t)(start * (col ? tile_p : 1))) << 8);
                    sub_g[nsub++] = (uint8_t)g_cur;
                  }
                  start = off; g_cur = g2;
                }
              }
            }
          }
          mbx[3 * k + 1] = (uint32_t)mb_segs.size();
          if (phased && nsub > 0) {
            int ends[4] = { 0, 0, 0, 0 }, tot = 0;
            const size_t at = mb_segs.size();
            mb_segs.resize(at + 3 + 2 * (size_t)nsub);
            size_t w = at + 3;
            for (int g2 = 0; g2 < n_groups; g2++) {
              for (int q = 0; q < nsub; q++) if (sub_g[q] == g2) { mb_segs[w++] = sub[2 * q]; mb_segs[w++] = sub[2 * q + 1]; tot += (int)((sub[2 * q] >> 24) & 63) + 1; }
              ends[g2] = tot;
            }
            if (tot > 255) {                               // (cannot happen with 64x64 boxes: <= 193 neighbour samples)
              mb_segs.resize(at); phased = false;
            } else {
              for (int g2 = n_groups; g2 < 4; g2++) { ends[g2] = tot; polls[g2] = 255; }
              mb_segs[at] = (uint32_t)nsub | ((uint32_t)n_groups << 8);
              mb_segs[at + 1] = (uint32_t)ends[0] | ((uint32_t)ends[1] << 8) | ((uint32_t)ends[2] << 16) | ((uint32_t)ends[3] << 24);
              mb_segs[at + 2] = (uint32_t)polls[0] | ((uint32_t)polls[1] << 8) | ((uint32_t)polls[2] << 16) | ((uint32_t)polls[3] << 24);
            }
          }
          if (d265_env("DE265HIP_PRINT_PHASES") && R.c == 0 && k % 97 == 0) {
            fprintf(stderr, "run %zu (%d,%d) nl %d phased %d groups %d polls %d %d %d %d nsub %d | need_col:", k, (int)o.x0, (int)o.y0, nl, (int)phased, n_groups, polls[0], polls[1], polls[2], polls[3], nsub);
            if (phased) { for (int q = 0; q < 96; q += 4) fprintf(stderr, " %d", need_col[q]); fprintf(stderr, " | need_row:"); for (int q = 0; q < 100; q += 4) fprintf(stderr, " %d", need_row[q]); }
            fprintf(stderr, "\n");
            for (int q = 0; q < nseg; q++) { const uint32_t mid = seg[2 * q] & 0xFFFFFFu; const uint32_t ro = mbx[3 * SC.mb_owner[mid] + 2];
              if (ro != 0xFFFFFFFFu) { const uint8_t* rd = (const uint8_t*)&mb_segs[ro]; fprintf(stderr, "   producer %u nl %d ready row:", SC.mb_owner[mid], (int)runs[SC.mb_owner[mid]].n_lvls + 1); for (int i = 0; i < 32; i += 2) fprintf(stderr, " %d", rd[i]); fprintf(stderr, " col:"); for (int i = 32; i < 64; i += 2) fprintf(stderr, " %d", rd[i]); fprintf(stderr, "\n"); } }
          }
          if (!(phased && nsub > 0)) {
            mb_segs.push_back((uint32_t)nseg | (1u << 8));
            mb_segs.push_back(0); mb_segs.push_back(0);      // (one group: everything at the start)
            mb_segs.insert(mb_segs.end(), seg, seg + 2 * nseg);
          }
        }
      }
    }
  }
  pt.mark("runs");
  pic->n_runs = (int)runs.size();
  pic->n_front = n_front;
  {
    // worker count = widest dependency level (more workers would only wait), within [64, 4 per CU]
    std::vector<int>& width = SC.width; width.assign(max_rl + 2, 0);
    for (auto& R : rb) width[R.level]++;
    if (max_rl >= 1) width[1] -= n_front;
    int widest = 0; for (int wv : width) widest = std::max(widest, wv);
    const char* wenv = d265_env("DE265HIP_RUN_WORKERS");
    // LDS-limited residency is 3 workgroups per CU (768); 2 per CU leave LDS for the kernels of the other GOP streams:
    // bench with 3 streams 5304 vs 5173 frames/s, one stream alone 2317 vs 2353
    int cap = wenv ? atoi(wenv) : 512;
    pic->n_batches = (int)(slots.size() / RUN_TICKET_SLOTS);
    const char* menv = d265_env("DE265HIP_RUN_WORKER_PCT");     // workers as a percentage of the widest level (experiments)
    const int pct = menv ? atoi(menv) : 125;
    pic->n_workers = std::min(pic->n_batches, std::max(64, std::min(cap, (int)((int64_t)widest * pct / 100))));
    // direct mode (see k_run): when the picture is wide rather than deep - most of its runs sit in its widest level
    // (a B picture: isolated intra CUs, 8 levels; an I picture: 126+ levels of ~100 runs)
    const char* denv = d265_env("DE265HIP_RUN_DIRECT");
    // Measured (4K Main10, tools/exp/ab_env.sh): a B picture alone 96-104 us either way (its time is the 7-level chain of
    // run hand-overs, not the ticket loop), but with three GOP streams in flight direct mode for B pictures costs 6 % and
    // for all pictures 23 % (6 390 -> 6 010 -> 4 890 frames/s): thousands of resident workgroups hold LDS the other
    // streams' kernels need.  Off unless asked for.
    pic->run_direct = denv ? atoi(denv) != 0 : false;
    if (d265_env("DE265HIP_PRINT_CRIT")) {               // diagnostic: run counts per level, longest path through the run DAG
      std::vector<int> wm(max_rl + 2, 0), wo(max_rl + 2, 0);
      for (size_t i = 0; i < rb.size(); i++) (SC.micro[i] ? wm : wo)[rb[i].level]++;
      fprintf(stderr, "de265hip runs %zu (front %d); per level (micro/ordinary):", rb.size(), n_front);
      for (int l = 1; l <= std::min(max_rl, 12); l++) fprintf(stderr, " %d/%d", wm[l], wo[l]);
      fprintf(stderr, "\n");
      // cost model of one run (us; fitted to ablation timings): fixed + per barrier level + per TU a wavefront has to do
      // in sequence inside a level + per 16x16 / 32x32 TU (collective)
      struct Path { double t = 0, lv = 0, slots = 0, n16 = 0, n32 = 0; int runs = 0; };
      std::vector<Path> fin(rb.size()); Path worst;
      for (size_t i = 0; i < rb.size(); i++) {           // rb is in creation order: producers precede consumers
        int nl = 0; for (int ti = rb[i].head; ti >= 0; ti = (int32_t)SC.it[ti].resid_offset) nl = std::max(nl, (int)SC.it[ti].run_level + 1);
        std::vector<int> per(nl + 1, 0); int n16 = 0, n32 = 0;
        for (int ti = rb[i].head; ti >= 0; ti = (int32_t)SC.it[ti].resid_offset) {
          const int l2 = SC.it[ti].log2_size;
          if (l2 == 4) n16++; else if (l2 == 5) n32++; else per[(int)SC.it[ti].run_level + 1]++;
        }
        int nslots = 0; for (int l = 1; l <= nl; l++) nslots += (per[l] + 3) / 4;
        Path st;
        for (int e = rb[i].dep_head; e >= 0; e = SC.dep_next[e]) if (fin[SC.dep_val[e]].t > st.t) st = fin[SC.dep_val[e]];
        st.t += 4.5 + 0.13 * nl + 0.10 * nslots + 0.55 * n16 + 0.65 * n32;
        st.lv += nl; st.slots += nslots; st.n16 += n16; st.n32 += n32; st.runs++;
        fin[i] = st;
        if (st.t > worst.t) worst = st;
      }
      fprintf(stderr, "de265hip chain: %lld barrier epochs in all runs; TUs per wavefront %lld %lld %lld %lld\n",
              (long long)dbg_foreign, (long long)dbg_w[0], (long long)dbg_w[1], (long long)dbg_w[2], (long long)dbg_w[3]);
      fprintf(stderr, "de265hip crit: est %.0f us; on the longest path %d runs, %.0f in-run levels, %.0f TU slots, %.0f 16x16 and %.0f 32x32 TUs; "
              "longest TU-to-TU dependency chain of the picture: %d TUs\n",
              worst.t, worst.runs, worst.lv, worst.slots, worst.n16, worst.n32, max_level);
    }
    // tickets per draw: 1 (DE265HIP_TICKET_BATCH for experiments: several per draw relieve the single device-scope
    // counter, ~12 ns per add, but serialise dependants: +46 % at 4 on a 4K B picture)
    const char* benv = d265_env("DE265HIP_TICKET_BATCH");
    pic->ticket_batch = benv ? std::max(1, std::min(64, atoi(benv))) : 1;
  }
  }                                                     // (!dev_scan)
  // level-0 launch of run mode: inter TUs with residual, then the residual-only intra TUs, largest first
  // [32x32 | 16x16 | 8x8 | 4x4]: both were written to their place above
  std::vector<TuTask>& l0 = SC.l0;
  pic->n_l0 = (int)l0.size();

  pt.mark("l0");
  // ---- MC tasks: resolve references, bi->uni shortcut, split into <=16x16 tiles
  std::vector<McTask>& mcs = SC.mcs; mcs.clear();
  const char* mc_env = d265_env("DE265HIP_MC_PATHS");
  const int mc_paths = mc_env ? atoi(mc_env) : 3;   // bit 0: the quad form (mc_micro_body), bit 1: the chunk form (mc_chunk_body) (experiments)
  const bool mc_all = cf == 1 && mc_paths != 0;        // k_mc_all (tiles, chunks and quads in bands, one launch); else k_mc over 16x16 tiles
  int64_t alg_mc = 0;
  // k_mc_all's bands: ranges of CTB rows with about an eighth of the picture's MC work each (a wavefront per 16x16 tile of a
  // larger PU and list, a third of that per block of a small PU): band_row[CTB row]
  std::vector<uint32_t>& mc_order_all = SC.mc_order_all;
  // (a build that failed half way through its PUs may have left lists of its own behind)
  for (int key : SC.micro_keys) SC.mc_pend[key].n = 0;
  SC.micro_keys.clear();
  for (int b = 0; b < 8; b++) { SC.mc_tiles[b].clear(); SC.mc_chunks[b].clear(); SC.mc_quads[b].clear(); SC.mc_order[b].clear(); }
  constexpr int chunk_h = 16;                             // chunks of 32x16: two tiles (see the order's comment below)
  std::vector<uint8_t>& band_row = SC.band_row; band_row.assign((size_t)g.ctbs_h, 0);
  if (mc_all) {
    std::vector<uint32_t>& roww = SC.roww; roww.assign((size_t)g.ctbs_h, 0);
    uint64_t total = 0;
    for (int i = 0; i < d->n_pus; i++) {
      const de265hip_pu& pu = d->pus[i];
      if (pu.y >= p.height) continue;                  // (rejected below)
      const int nl = (pu.pred_flag & 1) + ((pu.pred_flag >> 1) & 1);
      const uint32_t wgt = (pu.w < 16 || pu.h < 16) ? (uint32_t)nl * ((pu.w + 7) >> 3) * ((pu.h + 7) >> 3)
                                                    : 3u * nl * ((pu.w + 15) >> 4) * ((pu.h + 15) >> 4);
      roww[pu.y >> lc] += wgt; total += wgt;
    }
    uint64_t acc = 0; int b = 0;
    for (int r = 0; r < g.ctbs_h; r++) {
      while (b < 7 && acc * 8 >= total * (uint64_t)(b + 1)) b++;
      band_row[r] = (uint8_t)b;
      acc += roww[r];
    }
  }
  pt.mark("mc_bands");
  for (int i = 0; i < d->n_pus; i++) {
    const de265hip_pu& pu = d->pus[i];
    if (pu.slice_idx >= d->n_slices || pu.w == 0 || pu.h == 0 || (pu.w & 3) || (pu.h & 3) || pu.w > 64 || pu.h > 64 ||
        (pu.x & 3) || (pu.y & 3) || pu.x + pu.w > p.width || pu.y + pu.h > p.height) {
      delete pic; return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
    }
    const de265hip_slice_params& sh = d->slices[pu.slice_idx];
    int pf[2] = { pu.pred_flag & 1, (pu.pred_flag >> 1) & 1 };
    bool bad_ref = false;
    for (int l = 0; l < 2; l++)
      if (pf[l] && (pu.ref_idx[l] < 0 || pu.ref_idx[l] >= DE265HIP_MAX_REFS)) bad_ref = true;
    if (bad_ref) continue;                       // motion.cc:344-348: warning, PU left untouched
    // motion.cc:326-335 (tests weighted_pred_flag also in B slices)
    if (p.weighted_pred_flag == 0 && pf[0] && pf[1] && pu.mv[0][0] == pu.mv[1][0] && pu.mv[0][1] == pu.mv[1][1] &&
        sh.ref_pic_list[0][pu.ref_idx[0]] == sh.ref_pic_list[1][pu.ref_idx[1]])
      pf[1] = 0;
    if (sh.slice_type == 1) { if (!(pf[0] && !pf[1])) continue; }      // motion.cc:440-500
    else if (sh.slice_type == 0) { if (!pf[0] && !pf[1]) continue; }
    else continue;                                                       // I slice carries no PUs
    McTask t; memset(&t, 0, sizeof(t));
    t.slice_idx = pu.slice_idx;
    for (int l = 0; l < 2; l++) {
      t.slot[l] = -1; t.ref_idx[l] = pu.ref_idx[l];
      if (!pf[l]) continue;
      int slot = sh.ref_pic_list[l][pu.ref_idx[l]];
      // (that the slot holds a picture of this geometry is checked when the picture is launched: its reference may be a
      //  picture that is being built on another thread right now)
      if (slot < 0 || slot >= DE265HIP_MAX_DPB_SLOTS || slot == dst_slot) { delete pic; return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE; }
      pic->ref_mask |= 1u << slot;
      t.slot[l] = (int8_t)slot; t.mv[l][0] = pu.mv[l][0]; t.mv[l][1] = pu.mv[l][1];
    }
    const int nref = (t.slot[0] >= 0) + (t.slot[1] >= 0);
    const int64_t bppY = px_bytes(p.bit_depth_luma);
    alg_mc += ((int64_t)pu.w * pu.h + 2 * (int64_t)(pu.w >> (subw - 1)) * (pu.h >> (subh - 1))) * bppY * (nref + 1);      // (SubWidthC, SubHeightC are 1 or 2: two divisions per PU were a tenth of this loop)
    // 4:2:0, reference blocks (filter margins included) inside the picture for every list:
    //   PUs narrower or lower than 16 -> blocks of at most 8x8 for the quad form (mc_micro_body) (four blocks per wavefront, sorted by slot pair);
    //   other PUs -> chunks of up to 32x16 for the chunk form (mc_chunk_body) (one wavefront walks the chunk's 16x16 tiles).
    // What touches the picture border, and every other chroma format, goes to k_mc as 16x16 tiles (clamped per-sample fetch).
    auto interior = [&](int X, int Y, int w_, int h_) {
      for (int l = 0; l < 2; l++) {
        if (t.slot[l] < 0) continue;
        const int xs = X + (t.mv[l][0] >> 2) - 3, ys = Y + (t.mv[l][1] >> 2) - 3;
        const int xc = (X >> 1) + (t.mv[l][0] >> 3) - 1, yc = (Y >> 1) + (t.mv[l][1] >> 3) - 1;
        if (!(xs >= 0 && ys >= 0 && xs + w_ + 7 <= p.width && ys + h_ + 7 <= p.height &&
              xc >= 0 && yc >= 0 && xc + (w_ >> 1) + 3 <= (p.width >> 1) && yc + (h_ >> 1) + 3 <= (p.height >> 1))) return false;
      }
      return true;
    };
    // band of a task: eight ranges of CTB rows, one per XCD (k_mc_all)
    auto band_of = [&](int Y) { return (int)band_row[Y >> lc]; };
    auto tiles16 = [&](int X, int Y, int w_, int h_) {
      for (int ty = 0; ty < h_; ty += 16)
        for (int tx = 0; tx < w_; tx += 16) {
          McTask q = t;
          q.x = (uint16_t)(X + tx); q.y = (uint16_t)(Y + ty);
          q.w = (uint8_t)std::min(16, w_ - tx); q.h = (uint8_t)std::min(16, h_ - ty);
          SC.mc_tiles[band_of(Y + ty)].push_back(q);
        }
    };
    // (a PU whose whole reference block lies inside the picture: so does every part of it)
    const bool pu_interior = mc_all && interior(pu.x, pu.y, pu.w, pu.h);
    if (mc_all && (mc_paths & 1) && (pu.w < 16 || pu.h < 16)) {
      const int key0 = (t.slot[0] + 1) * 17 + (t.slot[1] + 1);
      for (int by = 0; by < pu.h; by += 8)
        for (int bx = 0; bx < pu.w; bx += 8) {
          const int w_ = std::min(8, pu.w - bx), h_ = std::min(8, pu.h - by), X = pu.x + bx, Y = pu.y + by;
          if (!pu_interior && !interior(X, Y, w_, h_)) { tiles16(X, Y, w_, h_); continue; }
          McTask q = t;
          q.x = (uint16_t)X; q.y = (uint16_t)Y; q.w = (uint8_t)w_; q.h = (uint8_t)h_;
          // quads in decode order: a slot pair's open quad is emitted the moment its fourth block arrives (quads sorted by
          // slot pair swept the band once per pair: 3.3x the algorithmic bytes through the L2 with two reference slots)
          const int bnd = band_of(Y), key = bnd * 289 + key0;
          BuildScratch::QuadPend& v = SC.mc_pend[key];
          if (v.n == 0) SC.micro_keys.push_back(key);
          v.t[v.n++] = q;
          if (v.n == 4) {
            SC.mc_order[bnd].push_back(0x80000000u | (uint32_t)(SC.mc_quads[bnd].size() / 4));
            SC.mc_quads[bnd].insert(SC.mc_quads[bnd].end(), v.t, v.t + 4); v.n = 0;
          }
        }
      continue;
    }
    for (int cy0 = 0; cy0 < pu.h; cy0 += chunk_h)
      for (int cx0 = 0; cx0 < pu.w; cx0 += 32) {
        const int cw_ = std::min(32, pu.w - cx0), ch_ = std::min(chunk_h, pu.h - cy0), X = pu.x + cx0, Y = pu.y + cy0;
        if (mc_all && (mc_paths & 2) && (pu_interior || interior(X, Y, cw_, ch_))) {
          McTask q = t;
          q.x = (uint16_t)X; q.y = (uint16_t)Y; q.w = (uint8_t)cw_; q.h = (uint8_t)ch_;
          { const int bc = band_of(Y); SC.mc_order[bc].push_back(0x40000000u | (uint32_t)SC.mc_chunks[bc].size()); SC.mc_chunks[bc].push_back(q); }
        } else tiles16(X, Y, cw_, ch_);
      }
  }
  pt.mark("mc_loop");
  // band by band: [k_mc's tiles | the chunk form's chunks | the quad form's blocks, four per wavefront, every four of one slot pair]
  std::sort(SC.micro_keys.begin(), SC.micro_keys.end());
  {
    McBands& B = pic->mc_bands; memset(&B, 0, sizeof(B));
    mc_order_all.clear();
    size_t mk = 0;
    pic->n_mc2 = pic->n_mc_quads = 0;
    for (int b = 0; b < 8; b++) {
      B.first[b] = (uint32_t)mcs.size();
      B.n_tiles[b] = (uint32_t)SC.mc_tiles[b].size(); B.n_chunks[b] = (uint32_t)SC.mc_chunks[b].size();
      mcs.insert(mcs.end(), SC.mc_tiles[b].begin(), SC.mc_tiles[b].end()); SC.mc_tiles[b].clear();
      mcs.insert(mcs.end(), SC.mc_chunks[b].begin(), SC.mc_chunks[b].end()); SC.mc_chunks[b].clear();
      const size_t q0 = mcs.size();
      mcs.insert(mcs.end(), SC.mc_quads[b].begin(), SC.mc_quads[b].end()); SC.mc_quads[b].clear();
      for (; mk < SC.micro_keys.size() && SC.micro_keys[mk] / 289 == b; mk++) {             // the quads left open
        BuildScratch::QuadPend& v = SC.mc_pend[SC.micro_keys[mk]];
        if (v.n == 0) continue;
        while (v.n < 4) { v.t[v.n] = v.t[v.n - 1]; v.t[v.n].w = v.t[v.n].h = 0; v.n++; }  // (a block that stores nothing)
        SC.mc_order[b].push_back(0x80000000u | (uint32_t)((mcs.size() - q0) / 4));
        mcs.insert(mcs.end(), v.t, v.t + 4);
        v.n = 0;
      }
      B.n_quads[b] = (uint32_t)((mcs.size() - q0) / 4);
      // the band's wavefronts take the border tiles first (few, and the longest single tasks: per-sample clamped fetch), then
      // chunks and quads mixed, in decode order: ONE sweep over the band's reference area.  Measured (4K10 B picture, two
      // reference slots, rocprofv3 + PMC): chunks of 32x32 first, then the quads: 43.6-44.6 us but 2.7x the algorithmic bytes
      // through the L2s (the band is swept twice and its reference area, 6 MB, does not fit the 4 MB L2); mixed in decode
      // order: 1.75x but 48.2 us (four-tile chunks at the end of the list make the tail); chunks of 32x16 mixed: 44.2 us
      // and 1.75x.  Hence chunks of at most two tiles.
      B.order_first[b] = (uint32_t)mc_order_all.size();
      for (uint32_t q = 0; q < B.n_tiles[b]; q++) mc_order_all.push_back(q);
      mc_order_all.insert(mc_order_all.end(), SC.mc_order[b].begin(), SC.mc_order[b].end()); SC.mc_order[b].clear();
      B.n_entries[b] = (uint32_t)mc_order_all.size() - B.order_first[b];
      pic->n_mc2 += (int)B.n_chunks[b]; pic->n_mc_quads += (int)B.n_quads[b];
    }
    SC.micro_keys.clear();
  }
  pic->mc_all = mc_all;
  pic->n_mc = (int)mcs.size();

  pt.mark("mc");
  // ---- PCM tasks
  std::vector<PcmTask>& pcms = SC.pcms; pcms.clear();
  for (int i = 0; i < d->n_pcms; i++) {
    const de265hip_pcm& pc = d->pcms[i];
    const int n = 1 << pc.log2_cb_size;
    if (pc.log2_cb_size < 3 || pc.log2_cb_size > 5 || (pc.x0 & 7) || (pc.y0 & 7) || pc.x0 + n > p.width ||
        pc.y0 + n > p.height || (int64_t)pc.sample_offset + n * n + (cf ? 2 * (n / subw) * (n / subh) : 0) > d->n_pcm_samples) {
      delete pic; return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
    }
    PcmTask t; t.x0 = pc.x0; t.y0 = pc.y0; t.log2_cb_size = pc.log2_cb_size; t.sample_offset = pc.sample_offset;
    pcms.push_back(t);
  }
  pic->n_pcm = (int)pcms.size();

  const size_t nblk = (size_t)g.w4 * g.h4;
  {
    // which flags occur anywhere in the picture: one OR over the plane (vectorised; two loops with an early exit each read all
    // 518 000 bytes of a 4K picture one by one whenever the answer was "none", 0.15 ms of a B picture's 1.6)
    uint8_t any = 0;
    const uint8_t* bf = d->blk_flags;
    for (size_t i = 0; i < nblk; i++) any |= bf[i];
    if (any & 0xF0) pic->any_edges = true;
    P.has_exempt = ((any & DE265HIP_BLK_BYPASS) || ((any & DE265HIP_BLK_PCM) && p.pcm_loop_filter_disable_flag)) ? 1 : 0;
  }

  // ---- per-CTB SAO records: slice flags applied, slice / tile permissions of the 3x3 neighbourhood
  // (sao.cc:127-163) evaluated once here instead of per sample on the device
  std::vector<SaoCtb>& saos = SC.saos; saos.resize((size_t)d->n_ctbs);
  for (int cy = 0; cy < g.ctbs_h; cy++)
    for (int cx = 0; cx < g.ctbs_w; cx++) {
      const int a = cx + cy * g.ctbs_w;
      const de265hip_ctb_info& ci = d->ctbs[a];
      const de265hip_slice_params& sh = d->slices[ci.slice_idx];
      SaoCtb& o = saos[a]; memset(&o, 0, sizeof(o));
      for (int c = 0; c < 3; c++) {
        const bool on = c == 0 ? sh.slice_sao_luma_flag != 0 : sh.slice_sao_chroma_flag != 0;
        o.type[c] = on ? (ci.sao_type_idx >> (2 * c)) & 3 : 0;
        o.eo[c] = (ci.sao_eo_class >> (2 * c)) & 3;
        o.band[c] = ci.sao_band_position[c];
        for (int k = 0; k < 4; k++) o.off[c][k] = ci.sao_offset_val[c][k];
      }
      // The reference looks the CTB's own slice address up with get_SliceHeader(xC,yC) where xC,yC are in samples of
      // the component (sao.cc:55) although the accessor takes luma samples: for chroma the address every neighbour is
      // compared with is that of CTB (cx/2, cy/2).  Reproduced (the oracle is pinned to the compiled reference on it):
      // chroma gets its own permission set, in which even the CTB itself can be "another slice".
      unsigned perm[2] = { 0, 0 };
      for (int ch = 0; ch < 2; ch++) {
        const int own_addr = ch ? d->ctbs[(cx / subw) + (cy / subh) * g.ctbs_w].slice_addr_rs : ci.slice_addr_rs;
        for (int dy = -1; dy <= 1; dy++)
          for (int dx = -1; dx <= 1; dx++) {
            const int nx = cx + dx, ny = cy + dy;
            bool ok = nx >= 0 && ny >= 0 && nx < g.ctbs_w && ny < g.ctbs_h;
            if (ok) {
              const int nb = nx + ny * g.ctbs_w;
              const de265hip_ctb_info& ni = d->ctbs[nb];
              if (ni.slice_addr_rs < own_addr && !sh.slice_loop_filter_across_slices_enabled_flag) ok = false;
              if (ni.slice_addr_rs > own_addr &&
                  !d->slices[ni.slice_idx].slice_loop_filter_across_slices_enabled_flag) ok = false;
              if (!p.loop_filter_across_tiles_enabled_flag && g.tile_id[nb] != g.tile_id[a]) ok = false;
            }
            if (ok) perm[ch] |= 1u << ((dy + 1) * 3 + dx + 1);
          }
      }
      o.perm = (uint16_t)(perm[0] | ((perm[1] & 0x7Fu) << 9));
      o.perm_c_hi = (uint8_t)(perm[1] >> 7);
    }

  pt.mark("pcm_sao_scan");
  // ---- one arena, one upload
  ArenaLayout L;
  // (device-side scan: the raw TU records travel instead of the level-0 / run-ordered task arrays the host used to derive)
  const size_t o_tus = L.add(dev_scan ? (size_t)d->n_tus * sizeof(de265hip_tu) : sorted.size() * sizeof(TuTask));
  const size_t o_cval = L.add((size_t)d->n_coeffs * 2), o_cpos = L.add((size_t)d->n_coeffs * 2);
  const size_t o_scal = L.add(DE265HIP_SCALING_BLOB_BYTES);
  const size_t o_mc = L.add(mcs.size() * sizeof(McTask)), o_mco = L.add(SC.mc_order_all.size() * 4);
  const size_t o_pcm = L.add(pcms.size() * sizeof(PcmTask)), o_pcms = L.add((size_t)d->n_pcm_samples * 2);
  const size_t o_sl = L.add((size_t)d->n_slices * sizeof(de265hip_slice_params));
  const size_t o_ctb = L.add((size_t)d->n_ctbs * sizeof(de265hip_ctb_info));
  const size_t o_tile = L.add((size_t)d->n_ctbs * 2);
  const size_t o_sao = L.add((size_t)d->n_ctbs * sizeof(SaoCtb));
  // the motion plane travels when the host hands one over; without one (blk_motion == NULL) it is made on the device from the
  // PU records, which travel instead (k_motion_from_pus; no PUs: "no reference" everywhere, a device memset)
  const bool mot_given = d->blk_motion != nullptr;
  const size_t o_flags = L.add(nblk), o_qp = L.add(nblk), o_mot_up = L.add(mot_given ? nblk * sizeof(de265hip_motion) : 0);
  const size_t o_pus = L.add(mot_given ? 0 : (size_t)d->n_pus * sizeof(de265hip_pu));
  const size_t o_runs = L.add(runs.size() * sizeof(RunTask)), o_rdeps = L.add(run_deps.size() * 4);
  const size_t o_rtus = L.add(run_tus.size() * sizeof(TuTask));
  const size_t o_slots = L.add(slots.size() * 4);
  const size_t o_l0 = L.add(l0.size() * sizeof(TuTask));
  const size_t o_l0x = L.add(SC.l0_rext.size() * sizeof(TuTask));
  const size_t o_mbx = L.add(n_mailboxes ? mbx.size() * 4 : 0), o_mbs = L.add(n_mailboxes ? mb_segs.size() * 4 : 0);
  const size_t o_grp = L.add(dev_scan ? (size_t)d->n_ctbs * 4 : 0), o_rs2ts = L.add(dev_scan ? (size_t)d->n_ctbs * 4 : 0), o_ts2rs = L.add(dev_scan ? (size_t)d->n_ctbs * 4 : 0);
  const size_t o_dorder = L.add(dev_scan ? (size_t)d->n_ctbs * 4 : 0);
  const size_t upload_bytes = L.total;                 // everything above is written by the host
  // device-only scratch: boundary strengths, residual blocks, run flags (no staging, no upload)
  const size_t o_bs = L.add(nblk);
  const size_t o_mot = mot_given ? o_mot_up : L.add(nblk * sizeof(de265hip_motion));
  // Device-side scan: what the passes will find is not known here, so its lists get room for the most a picture of this size
  // and this many TU records can ask for (address space in 288 GB of HBM; none of it is uploaded or cleared): a residual
  // sample per picture sample, a run per TU record, a producer entry per needed neighbour unit (<= nT + 1 per TU: 5 / 16 per
  // sample of a 4x4 TU), eight slots per ticket.  Mailboxes are an optimisation: a run beyond their number does without.
  ScanParams& SP = pic->SP; memset(&SP, 0, sizeof(SP));
  const int64_t total_samples = (int64_t)p.width * p.height + 2 * (int64_t)cwid * chei;
  const uint32_t cap_resid = (uint32_t)std::min<int64_t>(total_samples, 0x7FFFFF00);
  if (dev_scan) {
    SP.width = p.width; SP.height = p.height; SP.cwid = cwid; SP.chei = chei; SP.subw = subw; SP.subh = subh; SP.lc = lc; SP.lt = lt; SP.cf = cf;
    SP.ctbs_w = g.ctbs_w; SP.ctbs_h = g.ctbs_h; SP.n_ctbs = d->n_ctbs;
    for (int c = 0; c < 3; c++) { SP.map_w[c] = c ? (cwid + 3) / 4 : g.w4; SP.map_h[c] = c ? (chei + 3) / 4 : g.h4; }
    SP.n_tus = d->n_tus; SP.n_coeffs = d->n_coeffs; SP.bppY = (int)px_bytes(p.bit_depth_luma); SP.bppC = (int)px_bytes(p.bit_depth_chroma);
    const bool mailbox_on = d265_env("DE265HIP_NO_MAILBOX") == nullptr;
    SP.flags = (p.constrained_intra_pred_flag ? SCANF_CIP : 0) | (d265_env("DE265HIP_NO_MODE_DEPS") ? 0 : SCANF_MODE_DEPS) | (d265_env("DE265HIP_NO_MERGE") ? 0 : SCANF_MERGE) |
               (mailbox_on ? SCANF_MAILBOX : 0) | ((mailbox_on && !d265_env("DE265HIP_NO_MB_PHASES")) ? SCANF_MB_PHASES : 0) | (d265_env("DE265HIP_NO_MICRO") ? SCANF_MICRO_OFF : 0) |
               (d265_env("DE265HIP_NO_DENSE") ? SCANF_NO_DENSE : 0) | ((!d265_env("DE265HIP_MICRO16") || atoi(d265_env("DE265HIP_MICRO16")) != 0) ? SCANF_MICRO16 : 0) |
               (d265_env("DE265HIP_NO_FRONT") ? SCANF_FRONT_OFF : 0) | (p.implicit_rdpcm_enabled_flag ? SCANF_IMPLICIT_RDPCM : 0) |
               (p.transform_skip_rotation_enabled_flag ? SCANF_ROTATION : 0) | (dec->drop_producer ? SCANF_DROP_PRODUCER : 0) | SCANF_CHECK_POS;
    SP.micro_tus = d265_env("DE265HIP_MICRO_TUS") ? std::min(16, atoi(d265_env("DE265HIP_MICRO_TUS"))) : 16;
    SP.run_waves = dec->run_waves;
    SP.cap_runs = (uint32_t)d->n_tus;
    SP.cap_deps = (uint32_t)std::min<int64_t>(33 * (int64_t)d->n_tus, 5 * total_samples / 16 + d->n_tus) + 64;
    SP.cap_mb = (uint32_t)std::min<int64_t>(d->n_tus, 6 * (int64_t)d->n_ctbs + 64);
    SP.cap_segs = SP.cap_mb * 40 + 64;
    SP.cap_slots = 16u * (uint32_t)d->n_tus + 64;
  }
  const size_t o_resid = L.add((dev_scan ? (size_t)cap_resid : n_resid) * 2 + 64);
  pic->sync_bytes = (2 + (dev_scan ? (size_t)d->n_tus : runs.size())) * 4;
  const size_t o_sync = L.add(pic->sync_bytes);
  // edge mailboxes of k_run: 64 packets of (two samples, generation) per publishing run - its bottom row, then its right
  // column; cleared with the flags at build (a packet counts when it carries its launch's generation)
  const size_t mb_bytes = (size_t)(dev_scan ? SP.cap_mb : (uint32_t)n_mailboxes) * 64 * 8;
  const size_t o_mb = L.add(mb_bytes);
  const size_t clear_bytes = L.total - o_sync;           // flags + mailboxes (experiments that clear them per launch)
  pic->clear_bytes = clear_bytes;
  ScanLayout& SL = pic->SL;
  if (dev_scan) L.total = SL.plan(SP, L.total);
  // pinned staging + pooled arena: no allocation, no host-side wait in the steady state
  int stage_idx = -1;
  uint8_t* host_base = nullptr;
  hipEvent_t stage_event = nullptr;
  if (dec->dry) { pic->dry_arena.assign((dev_scan && dec->dry_scan) ? L.total : upload_bytes, 0); host_base = pic->dry_arena.data(); }
  else {
    std::lock_guard<std::mutex> lk(dec->mu);
    rc = acquire_stage(dec, upload_bytes, &stage_idx);
    if (!rc) {
      host_base = (uint8_t*)dec->stage_pool[stage_idx].ptr; stage_event = dec->stage_pool[stage_idx].copied;
      rc = acquire_arena(dec, L.total, &pic->arena_buf);
      if (rc) dec->stage_pool[stage_idx].state = 0;
    }
    if (!rc) {
      if (dec->ring_free.empty()) { rc = DE265HIP_ERROR_OUT_OF_MEMORY; dec->stage_pool[stage_idx].state = 0; release_arena(dec, pic->arena_buf); }
      else { pic->ring_idx = dec->ring_free.front(); dec->ring_free.pop_front(); pic->ring_seq = ++dec->ring_seq; dec->ring_owner[pic->ring_idx] = pic->ring_seq;
             dec->stage_pool[stage_idx].owner = pic->ring_seq; }
    }
  }
  if (rc) { delete pic; return rc; }
  // (a failure below hands both back)
  auto fail = [&](int code) {
    std::lock_guard<std::mutex> lk(dec->mu);
    for (auto& b : dec->stage_pool) if (b.ptr == host_base) b.state = 0;
    release_arena(dec, pic->arena_buf);
    if (pic->ring_idx >= 0) dec->ring_free.push_back(pic->ring_idx);
    if (pic->uploaded) (void)hipEventDestroy(pic->uploaded);
    delete pic;
    return code;
  };
  struct HostView { uint8_t* p; uint8_t* data() const { return p; } } host{ host_base };
  auto put = [&](size_t off, const void* src, size_t bytes) { if (bytes && src) memcpy(host.data() + off, src, bytes); };
  if (dev_scan) put(o_tus, d->tus, (size_t)d->n_tus * sizeof(de265hip_tu));
  else put(o_tus, sorted.data(), sorted.size() * sizeof(TuTask));
  put(o_cval, d->coeff_val, (size_t)d->n_coeffs * 2); put(o_cpos, d->coeff_pos, (size_t)d->n_coeffs * 2);
  if (p.scaling_list_enable_flag) put(o_scal, d->scaling_factors, DE265HIP_SCALING_BLOB_BYTES);
  else memset(host.data() + o_scal, 0, DE265HIP_SCALING_BLOB_BYTES);
  put(o_mc, mcs.data(), mcs.size() * sizeof(McTask)); put(o_mco, SC.mc_order_all.data(), SC.mc_order_all.size() * 4);
  put(o_pcm, pcms.data(), pcms.size() * sizeof(PcmTask)); put(o_pcms, d->pcm_samples, (size_t)d->n_pcm_samples * 2);
  put(o_sl, d->slices, (size_t)d->n_slices * sizeof(de265hip_slice_params));
  put(o_ctb, d->ctbs, (size_t)d->n_ctbs * sizeof(de265hip_ctb_info));
  put(o_tile, g.tile_id.data(), (size_t)d->n_ctbs * 2);
  put(o_sao, saos.data(), saos.size() * sizeof(SaoCtb));
  put(o_runs, runs.data(), runs.size() * sizeof(RunTask)); put(o_rdeps, run_deps.data(), run_deps.size() * 4);
  put(o_rtus, run_tus.data(), run_tus.size() * sizeof(TuTask));
  put(o_slots, slots.data(), slots.size() * 4);
  put(o_l0, l0.data(), l0.size() * sizeof(TuTask));
  put(o_l0x, SC.l0_rext.data(), SC.l0_rext.size() * sizeof(TuTask));
  if (n_mailboxes) { put(o_mbx, mbx.data(), mbx.size() * 4); put(o_mbs, mb_segs.data(), mb_segs.size() * 4); }
  put(o_flags, d->blk_flags, nblk); put(o_qp, d->blk_qp_y, nblk);
  if (mot_given) put(o_mot, d->blk_motion, nblk * sizeof(de265hip_motion));
  else put(o_pus, d->pus, (size_t)d->n_pus * sizeof(de265hip_pu));
  if (dev_scan) {
    // CTBs of one slice and tile share a group word (the availability tests of intrapred.cc:486-508 compare exactly these two)
    uint32_t* grp = (uint32_t*)(host.data() + o_grp);
    for (int a = 0; a < d->n_ctbs; a++) grp[a] = (uint32_t)d->ctbs[a].slice_addr_rs | ((uint32_t)g.tile_id[a] << 16);
    put(o_rs2ts, g.rs2ts.data(), (size_t)d->n_ctbs * 4); put(o_ts2rs, g.ts2rs.data(), (size_t)d->n_ctbs * 4);
    put(o_dorder, g.diag_order.data(), (size_t)d->n_ctbs * 4);
  }

  pt.mark("staging");
  pic->arena = dec->dry ? (void*)host_base : pic->arena_buf.ptr;
  pic->arena_bytes = L.total;
  uint8_t* base = (uint8_t*)pic->arena;
  ScanBufs& SB = pic->SB; memset(&SB, 0, sizeof(SB));
  if (dev_scan) {
    SB.tus = (const de265hip_tu*)(base + o_tus); SB.ctb_group = (const uint32_t*)(base + o_grp); SB.rs2ts = (const int32_t*)(base + o_rs2ts);
    SB.ts2rs = (const int32_t*)(base + o_ts2rs); SB.ctb_order = (const int32_t*)(base + o_dorder); SB.blk_flags = base + o_flags; SB.coeff_pos = (uint16_t*)(base + o_cpos);
    SB.used_units = dec->dry ? &g_used_units[0][0][0] : dec->d_used_units;
    SL.bind(base, SB);
    pic->cap_resid = cap_resid;
    if (dec->dry && dec->dry_scan) scan_host_run(SP, SB, SL, base, cap_resid);      // the CPU rehearsal of the passes (tests only)
  }
  if (dec->dry && !d265_env("DE265HIP_DRY_NO_HASH") && !dev_scan) { // FNV-1a over everything the device would receive (tools/exp/build_hash.py)
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](const void* ptr, size_t n) { const uint8_t* b = (const uint8_t*)ptr; for (size_t i = 0; i < n; i++) { hsh ^= b[i]; hsh *= 1099511628211ull; } };
    const size_t offs[] = { o_tus, o_cval, o_cpos, o_scal, o_mc, o_pcm, o_pcms, o_sl, o_ctb, o_tile, o_sao, o_flags, o_qp, o_mot, o_runs, o_rdeps, o_rtus, o_slots, o_l0, upload_bytes };
    const size_t lens[] = { sorted.size() * sizeof(TuTask), (size_t)d->n_coeffs * 2, (size_t)d->n_coeffs * 2, (size_t)DE265HIP_SCALING_BLOB_BYTES, mcs.size() * sizeof(McTask),
                            pcms.size() * sizeof(PcmTask), (size_t)d->n_pcm_samples * 2, (size_t)d->n_slices * sizeof(de265hip_slice_params), (size_t)d->n_ctbs * sizeof(de265hip_ctb_info),
                            (size_t)d->n_ctbs * 2, (size_t)d->n_ctbs * sizeof(SaoCtb), nblk, nblk, mot_given ? nblk * sizeof(de265hip_motion) : (size_t)0, runs.size() * sizeof(RunTask), run_deps.size() * 4,
                            run_tus.size() * sizeof(TuTask), slots.size() * 4, l0.size() * sizeof(TuTask), 0 };
    for (int i = 0; i < 19; i++) mix(host.data() + offs[i], lens[i]);      // (only the written bytes: padding between sections is undefined)
    const int64_t scal[] = { pic->n_workers, pic->n_batches, pic->n_l0, pic->n_l0_size[0], pic->n_l0_size[1], pic->n_l0_size[2], pic->n_l0_size[3], pic->n_mc, pic->n_mc2, pic->n_mc_quads, pic->n_pcm,
                             pic->n_tus, pic->n_runs, (int64_t)n_resid, (int64_t)L.total, (int64_t)pic->any_edges, (int64_t)P.has_exempt, (int64_t)pic->run_direct, max_level, max_rl, (int64_t)sum_lvls, (int64_t)pic->n_front };
    mix(scal, sizeof(scal));
    if (!SC.l0_rext.empty()) mix(host.data() + o_l0x, SC.l0_rext.size() * sizeof(TuTask));
    if (n_mailboxes) { mix(host.data() + o_mbx, mbx.size() * 4); mix(host.data() + o_mbs, mb_segs.size() * 4); }
    mix(host.data() + o_mco, SC.mc_order_all.size() * 4);
    mix(pic->level_start.data(), pic->level_start.size() * sizeof(int));
    dec->pooled_bytes = (size_t)hsh;
  }
  if (!dec->dry) {
    de265hip_picture::Enq& E = pic->enq;
    E.pending = true; E.host_base = host_base; E.upload_bytes = upload_bytes; E.stage_event = stage_event;
    E.o_sync = o_sync; E.clear_bytes = clear_bytes; E.o_mot = o_mot; E.o_pus = o_pus; E.o_sl = o_sl; E.o_l0 = o_l0; E.o_l0x = o_l0x; E.o_cpos = o_cpos;
    E.nblk = nblk; E.mot_given = mot_given; E.check_on_device = !dev_scan && !host_checks_positions;
    E.n_pus = d->n_pus; E.n_slices = d->n_slices; E.n_l0chk = (int)l0.size(); E.n_l0xchk = (int)SC.l0_rext.size();
    pic->cap_resid = cap_resid;
  }
  pic->dev_scan = dev_scan;
  pic->d_tus = (TuTask*)(base + o_tus);
  pic->d_cval = (int16_t*)(base + o_cval); pic->d_cpos = (uint16_t*)(base + o_cpos);
  pic->d_scaling = base + o_scal;
  pic->d_mc = (McTask*)(base + o_mc); pic->d_mc_order = (uint32_t*)(base + o_mco);
  pic->d_pcm = (PcmTask*)(base + o_pcm); pic->d_pcm_samples = (uint16_t*)(base + o_pcms);
  pic->d_slices = (de265hip_slice_params*)(base + o_sl);
  pic->d_ctbs = (de265hip_ctb_info*)(base + o_ctb);
  pic->d_tile_id = (uint16_t*)(base + o_tile);
  pic->d_sao = (SaoCtb*)(base + o_sao);
  pic->d_flags = base + o_flags; pic->d_qp = (int8_t*)(base + o_qp);
  pic->d_motion = (de265hip_motion*)(base + o_mot);
  pic->d_bs = base + o_bs;
  pic->d_sync = (uint32_t*)(base + o_sync); pic->d_resid = (int16_t*)(base + o_resid);
  pic->d_mb = (unsigned long long*)(base + o_mb);
  if (dev_scan) {
    pic->d_runs = SB.runs; pic->d_deps = SB.deps; pic->d_run_tus = SB.run_tus; pic->d_slots = SB.slots;
    pic->d_l0 = SB.l0; pic->d_l0_rext = SB.l0x; pic->d_mbx = SB.mbx; pic->d_mbsegs = SB.mb_segs; pic->d_front_idx = SB.front_idx;
    pic->n_mailboxes = (int)SP.cap_mb;
    pic->o_layout[0] = (int64_t)SL.o_runs; pic->o_layout[1] = (int64_t)SL.o_run_tus; pic->o_layout[2] = (int64_t)SL.o_deps; pic->o_layout[3] = (int64_t)SL.o_slots;
    pic->o_layout[4] = (int64_t)SL.o_l0; pic->o_layout[5] = (int64_t)SL.o_l0x; pic->o_layout[6] = (int64_t)SL.o_mbx; pic->o_layout[7] = (int64_t)SL.o_mb_segs;
    pic->o_layout[8] = (int64_t)SL.o_front; pic->o_layout[9] = (int64_t)SL.o_run_ntus; pic->o_layout[10] = (int64_t)SL.o_run_nall; pic->o_layout[11] = (int64_t)SL.o_run_level;
    pic->o_layout[12] = (int64_t)SL.o_counts;
  } else {
    pic->d_runs = (RunTask*)(base + o_runs); pic->d_deps = (uint32_t*)(base + o_rdeps);
    pic->d_run_tus = (TuTask*)(base + o_rtus); pic->d_slots = (uint32_t*)(base + o_slots);
    pic->d_l0 = (TuTask*)(base + o_l0);
    pic->d_l0_rext = (TuTask*)(base + o_l0x); pic->n_l0_rext = (int)SC.l0_rext.size();
    pic->d_mbx = n_mailboxes ? (uint32_t*)(base + o_mbx) : nullptr; pic->d_mbsegs = (uint32_t*)(base + o_mbs);
    pic->n_mailboxes = n_mailboxes;
    pic->o_layout[0] = (int64_t)o_runs; pic->o_layout[1] = (int64_t)o_rtus; pic->o_layout[2] = (int64_t)o_rdeps; pic->o_layout[3] = (int64_t)o_slots;
    pic->o_layout[4] = (int64_t)o_l0; pic->o_layout[5] = (int64_t)o_l0x; pic->o_layout[6] = n_mailboxes ? (int64_t)o_mbx : -1; pic->o_layout[7] = (int64_t)o_mbs;
    for (int q = 8; q < 13; q++) pic->o_layout[q] = -1;
  }

  const int64_t Pbytes = ((int64_t)p.width * p.height + 2 * (int64_t)cwid * chei) * px_bytes(p.bit_depth_luma);
  pic->stats.n_levels = max_level + (pic->level_start[1] > 0 ? 1 : 0);
  pic->stats.n_tu_tasks = pic->n_tus; pic->stats.n_mc_tasks = pic->n_mc;
  pic->stats.n_runs = pic->n_runs; pic->stats.n_run_levels = max_rl;
  pic->stats.n_in_run_levels = (int32_t)sum_lvls;
  pic->stats.device_bytes = (int64_t)L.total;
  pic->stats.alg_bytes_mc = alg_mc; pic->stats.alg_bytes_resid = alg_resid; pic->stats.alg_bytes_intra = alg_intra - alg_intra_front;
  pic->stats.alg_bytes_intra_front = alg_intra_front; pic->stats.n_front_runs = pic->n_front;
  pic->stats.alg_bytes_deblock = pic->any_edges ? 2 * Pbytes : 0;       // one read + one write
  pic->stats.alg_bytes_sao = p.sample_adaptive_offset_enabled_flag ? 2 * Pbytes + 16 * (int64_t)d->n_ctbs : 0;
  if (dec->dry && dev_scan && dec->dry_scan) { pic->h_counts_dry = *SB.counts; pic->scan_pending = true; (void)finish_scan(pic); }
  if (!dec->dry) {
    std::lock_guard<std::mutex> lk(dec->mu);
    dec->live.push_back(pic);
  }
  pt.mark("host");
  if (!dec->dry && g_defer_enqueue) {
    // (a pipeline's build: the upload goes out from this thread, on an upload stream; the launcher's copy stream waits for it)
    de265hip_picture::Enq& E = pic->enq;
    if (HsaDev* H = uploads_by_hsa() ? hsa_dev(dec->device) : nullptr) {
      // (the arena: recycled only when known free - its last user's event, if there is one, has normally long completed)
      bool ok = !pic->arena_buf.used || hipEventQuery(pic->arena_buf.last_use) == hipSuccess || hipEventSynchronize(pic->arena_buf.last_use) == hipSuccess;
      (void)hipGetLastError();
      hsa_signal_t sg{ 0 };
      { std::lock_guard<std::mutex> lk(dec->mu); if (!dec->free_sigs.empty()) { sg.handle = dec->free_sigs.back(); dec->free_sigs.pop_back(); } }
      if (ok && !sg.handle) ok = hsa_signal_create(1, 0, nullptr, &sg) == HSA_STATUS_SUCCESS;
      if (ok) {
        hsa_signal_store_relaxed(sg, 1);
        ok = hsa_amd_memory_async_copy(pic->arena, H->gpu, E.host_base, H->cpu, E.upload_bytes, 0, nullptr, sg) == HSA_STATUS_SUCCESS;
      }
      if (ok) {
        E.uploaded_by_builder = true; E.up_sig = sg.handle;
        {
          std::lock_guard<std::mutex> lk(dec->mu);
          for (auto& b : dec->stage_pool) if (b.ptr == E.host_base && b.owner == pic->ring_seq) { b.state = 2; b.sig = sg.handle; }
        }
        pt.mark("enqueue");
        pt.done();
        *out = pic;
        return DE265HIP_OK;
      }
      // the runtime refused the copy (an agent pair it does not take, a machine this was not tried on): this picture and all later
      // ones go the other way, through an upload stream - slower, never wrong
      if (sg.handle) { std::lock_guard<std::mutex> lk(dec->mu); dec->free_sigs.push_back(sg.handle); }
      { std::lock_guard<std::mutex> lk(g_streams_mu); H->ok = false; }
      static bool told = false;
      if (!told) { told = true; fprintf(stderr, "de265hip: uploads through the HSA runtime are not available here; using upload streams\n"); }
    }
    hipStream_t us;
    { std::lock_guard<std::mutex> lk(dec->mu); us = dec->upload_streams[dec->upload_turn++ & 1]; }
    bool ok = !pic->arena_buf.used || hipStreamWaitEvent(us, pic->arena_buf.last_use, 0) == hipSuccess;
    if (upload_kernel_grid() && E.upload_bytes < 0xFFFFFFFFu && !(((uintptr_t)pic->arena | (uintptr_t)E.host_base) & 15))
      ok = ok && upload_by_kernel(pic->arena, E.host_base, E.upload_bytes, us, upload_kernel_grid()) == hipSuccess;
    else
      ok = ok && hipMemcpyAsync(pic->arena, E.host_base, E.upload_bytes, hipMemcpyHostToDevice, us) == hipSuccess;
    ok = ok && hipEventRecord(E.stage_event, us) == hipSuccess;
    if (!ok) { de265hip_picture_free(pic); return DE265HIP_ERROR_DECODING; }
    E.uploaded_by_builder = true;
    std::lock_guard<std::mutex> lk(dec->mu);
    for (auto& b : dec->stage_pool) if (b.ptr == E.host_base && b.owner == pic->ring_seq) { b.state = 2; b.sig = 0; }
  }
  if (!dec->dry && !g_defer_enqueue) {
    const int erc = de265hip_picture_enqueue(pic);
    if (erc) { de265hip_picture_free(pic); return erc; }
  }
  pt.mark("enqueue");
  pt.done();
  *out = pic;
  return DE265HIP_OK;
}

// The device side of a build: the upload of the staged command buffers, the kernels that work on the raw records behind it (the
// scan of the TU records, the motion plane), on one of the decoder's copy streams.  de265hip_picture_build calls it itself;
// de265hip_picture_build_host leaves it to the caller, who may be another thread: the pipeline issues ALL HIP calls of a
// decoder from one thread (fifteen workers calling into the HIP runtime for the same device spent three quarters of their
// builds waiting for its locks: builds of 1.7 ms took 6.5 ms, round 4).
int de265hip_picture_enqueue(de265hip_picture* pic) { return de265hip_picture_enqueue_batch(&pic, 1); }

// The same for several pictures of ONE decoder at once (at most SCAN_BATCH are scanned per set of launches; more are taken in
// turns): their uploads go out one after the other on one copy stream, the passes of their scans run as one batch behind them.
int de265hip_picture_enqueue_batch(de265hip_picture** pics, int n)
{
  if (!pics || n < 1) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  for (int i = 0; i < n; i++) if (!pics[i] || !pics[i]->dec || pics[i]->dec != pics[0]->dec) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  de265hip_decoder* dec = pics[0]->dec;
  for (int i0 = 0; i0 < n; i0 += SCAN_BATCH) {
    const int m = std::min(SCAN_BATCH, n - i0);
    hipStream_t cs;
    { std::lock_guard<std::mutex> lk(dec->mu); cs = dec->copy_streams[dec->copy_turn++ % (uint64_t)dec->n_copy_streams]; reap_arenas(dec); }
    static const int own_prep = d265_env("DE265HIP_OWN_PREP") ? atoi(d265_env("DE265HIP_OWN_PREP")) : 1;
    static const int run2_lane0 = d265_env("DE265HIP_SCAN_RUN2_LANE0") ? atoi(d265_env("DE265HIP_SCAN_RUN2_LANE0")) : 0;
    ScanBatch J; memset(&J, 0, sizeof(J)); J.pad = run2_lane0 ? 1 : 0;      // (pad bit 0: the mailbox readers' pass on one lane, scan_core.h's loop: the parity variant)
    PrepBatch PJ; PJ.n = 0; PJ.pad = 0;
    de265hip_picture* done[SCAN_BATCH]; int n_done = 0;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tl = tnow();
    auto sec = [&](int k) { const double t = tnow(); dec->t_sec[k] += t - tl; if (t - tl > dec->t_sec_max[k]) dec->t_sec_max[k] = t - tl; tl = t; };
    for (int i = i0; i < i0 + m; i++) {
      de265hip_picture* pic = pics[i];
      de265hip_picture::Enq& E = pic->enq;
      if (!E.pending) continue;
      uint8_t* base = (uint8_t*)pic->arena;
      // a recycled arena may still be read by kernels of the picture that had it before
      tl = tnow();
      if (pic->arena_buf.used) HIPCHK(hipStreamWaitEvent(cs, pic->arena_buf.last_use, 0), DE265HIP_ERROR_DECODING);
      sec(0);
      {
        std::lock_guard<std::mutex> lk(dec->mu);
        if (!dec->free_events.empty()) { pic->uploaded = dec->free_events.back(); dec->free_events.pop_back(); }
      }
      if (!pic->uploaded) HIPCHK(hipEventCreateWithFlags(&pic->uploaded, hipEventDisableTiming), DE265HIP_ERROR_OUT_OF_MEMORY);
      // k_run's ticket counter, run flags and mailbox packets start from zero: a pooled arena held another picture's buffers
      // at these offsets before, and ANY bit pattern there could pass for a raised flag of some launch generation (round 4
      // tried to do without this clear, the generations being unique per decoder: a mid-size picture in a recycled arena came
      // out wrong).  With the device-side scan its own cleared buffers follow the mailboxes: one memset for both.
      const size_t clear_to = pic->dev_scan ? pic->SL.clear_end : E.o_sync + E.clear_bytes;
      sec(1);
      sec(2);
      if (E.uploaded_by_builder && E.up_sig) {             // (through the HSA runtime: on the host - the builds run ahead, it has normally long arrived)
        if (!hsa_upload_wait(E.up_sig)) return DE265HIP_ERROR_DECODING;
      } else if (E.uploaded_by_builder) HIPCHK(hipStreamWaitEvent(cs, E.stage_event, 0), DE265HIP_ERROR_DECODING);      // (the builder's thread sent it)
      else {
        HIPCHK(hipMemcpyAsync(base, E.host_base, E.upload_bytes, hipMemcpyHostToDevice, cs), DE265HIP_ERROR_DECODING);
        sec(3);
        HIPCHK(hipEventRecord(E.stage_event, cs), DE265HIP_ERROR_DECODING);
        std::lock_guard<std::mutex> lk(dec->mu);
        for (auto& b : dec->stage_pool) if (b.ptr == E.host_base && b.owner == pic->ring_seq) { b.state = 2; b.sig = 0; }      // reusable once `copied` has completed
      }
      E.pending = false;
      sec(4);
      // the cleared region and the motion plane: one launch each for the whole batch, behind the waits of all its pictures
      {
        PrepJob& q = PJ.job[PJ.n++];
        q.zero = base + E.o_sync; q.zero_bytes = clear_to - E.o_sync;
        q.ff = E.mot_given ? nullptr : base + E.o_mot; q.ff_bytes = E.mot_given ? 0 : E.nblk * sizeof(de265hip_motion);
        q.pus = (const de265hip_pu*)(base + E.o_pus); q.slices = (const de265hip_slice_params*)(base + E.o_sl); q.motion = (de265hip_motion*)(base + E.o_mot);
        q.n_pus = E.mot_given ? 0 : E.n_pus; q.n_slices = E.n_slices; q.w4 = pic->P.w4; q.h4 = pic->P.h4;
      }
      if (pic->dev_scan) {
        // the last pass of the scan writes the counts into the picture's pinned ring entry and raises its ready word (system-
        // scope stores: no copy, no event for the host to wait on - de265hip_picture_run polls the word)
        ScanCounts* hc = dec->h_ring + pic->ring_idx;
        __atomic_store_n(&hc->ready, 0u, __ATOMIC_RELEASE);
        pic->SB.host_counts = hc; pic->SB.err_word = dec->d_err_ring + pic->ring_idx; pic->SB.ready_tag = (uint32_t)pic->ring_seq | 0x80000000u;
        ScanJob& jb = J.job[J.n++];
        jb.P = pic->SP; jb.B = pic->SB; jb.cap_resid = pic->cap_resid; jb.cap_levels = pic->SL.cap_levels;
        // the motion plane by the per-TU pass's launch instead of a launch of its own (behind k_build_prep, which sets it to "no reference")
        static const bool motion_merged = !(d265_env("DE265HIP_MOTION_LAUNCH") && atoi(d265_env("DE265HIP_MOTION_LAUNCH")));
        PrepJob& q = PJ.job[PJ.n - 1];
        if (motion_merged && own_prep && q.ff && q.n_pus > 0) {
          jb.mo_pus = q.pus; jb.mo_slices = q.slices; jb.mo_plane = q.motion; jb.mo_n_pus = q.n_pus; jb.mo_n_slices = q.n_slices; jb.mo_w4 = q.w4; jb.mo_h4 = q.h4;
          q.n_pus = 0;
        }
        pic->scan_pending = true;
      } else {
        HIPCHK(hipMemsetAsync(dec->d_err_ring + pic->ring_idx, 0, 4, cs), DE265HIP_ERROR_DECODING);
        if (E.check_on_device && E.n_l0chk + E.n_l0xchk > 0)
          hipLaunchKernelGGL(k_check_coeffs, dim3((E.n_l0chk + E.n_l0xchk + 15) / 16), dim3(256), 0, cs, (const TuTask*)(base + E.o_l0), E.n_l0chk,
                             (const TuTask*)(base + E.o_l0x), E.n_l0xchk, (uint16_t*)(base + E.o_cpos), dec->d_err_ring + pic->ring_idx);
      }
      done[n_done++] = pic;
      sec(5);
    }
    tl = tnow();
    if (own_prep) HIPCHK(prep_enqueue_batch(cs, PJ), DE265HIP_ERROR_DECODING);
    else for (int i = 0; i < PJ.n; i++) {
      const PrepJob& q = PJ.job[i];
      HIPCHK(hipMemsetAsync(q.zero, 0, q.zero_bytes, cs), DE265HIP_ERROR_DECODING);
      if (q.ff) {
        HIPCHK(hipMemsetAsync(q.ff, 0xFF, q.ff_bytes, cs), DE265HIP_ERROR_DECODING);
        PicDev Pm; memset(&Pm, 0, sizeof(Pm)); Pm.w4 = q.w4; Pm.h4 = q.h4;
        if (q.n_pus > 0) hipLaunchKernelGGL(k_motion_from_pus, dim3((q.n_pus + 15) / 16), dim3(256), 0, cs, Pm, q.pus, q.n_pus, q.slices, q.n_slices, q.motion);
      }
    }
    if (J.n) HIPCHK(scan_enqueue_batch(cs, J), DE265HIP_ERROR_DECODING);
    sec(6);
    for (int i = 0; i < n_done; i++) HIPCHK(hipEventRecord(done[i]->uploaded, cs), DE265HIP_ERROR_DECODING);
    sec(7); dec->n_sec += n_done;
  }
  return DE265HIP_OK;
}

// Has the device side of the picture's build reported (the scan of its records: de265hip_picture_run would not wait)?  1 yes,
// 0 not yet, < 0 never will (not enqueued).  For a launcher that has other work to do meanwhile.
int de265hip_picture_ready(de265hip_picture* pic)
{
  if (!pic || !pic->dec) return -1;
  if (pic->enq.pending) return -1;
  if (!pic->scan_pending || pic->dec->dry) return 1;
  const ScanCounts* K = pic->dec->h_ring + pic->ring_idx;
  return __atomic_load_n(&K->ready, __ATOMIC_ACQUIRE) == ((uint32_t)pic->ring_seq | 0x80000000u) ? 1 : 0;
}

// de265hip_picture_build without its device side (see de265hip_picture_enqueue)
int de265hip_picture_build_host(de265hip_decoder* dec, int dst_slot, const de265hip_picture_desc* d, de265hip_picture** out)
{
  g_defer_enqueue = true;
  const int rc = de265hip_picture_build(dec, dst_slot, d, out);
  g_defer_enqueue = false;
  return rc;
}

// Profiling aid (tools/time_build.py --host-only): the host stage of de265hip_picture_build `reps` times without a GPU
// and without any HIP call (staging into plain memory, nothing uploaded).  Returns the build's return code.
static thread_local uint64_t g_last_build_hash = 0;
uint64_t de265hip_debug_last_build_hash(void) { return g_last_build_hash; }
int de265hip_debug_build_host_only(const de265hip_picture_desc* d, int reps) { return de265hip_debug_build_host_only_ex(d, reps, 0, nullptr); }

// mode 0: the round-3 host scan; 1: the host's part of a build with the device-side scan (what the product spends on a host
// core: validation, MC tasks, staging); 2: the same plus the CPU rehearsal of the passes (the equivalence tests).
// keep != nullptr: the last picture is handed out (an orphan of a dry decoder: de265hip_debug_picture_layout / _read work on
// its host-memory arena; de265hip_picture_free).
int de265hip_debug_build_host_only_ex(const de265hip_picture_desc* d, int reps, int mode, de265hip_picture** keep)
{
  if (!d || mode < 0 || mode > 2) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  if (keep) *keep = nullptr;
  de265hip_decoder* dec = new de265hip_decoder();
  dec->dry = true; dec->dev_scan = mode != 0; dec->dry_scan = mode == 2;
  if (d265_env("DE265HIP_TEST_DROP_PRODUCER")) dec->drop_producer = true;      // (dry decoders only: the equivalence test of the fault injection)
  for (auto& sl : dec->slots) { sl.valid = true; sl.w = d->params.width; sl.h = d->params.height; sl.bdY = d->params.bit_depth_luma; sl.bdC = d->params.bit_depth_chroma; sl.cf = d->params.chroma_format_idc; }
  int rc = 0;
  for (int i = 0; i < reps && !rc; i++) {
    de265hip_picture* pic = nullptr;
    rc = de265hip_picture_build(dec, DE265HIP_MAX_DPB_SLOTS - 1, d, &pic);
    g_last_build_hash = (uint64_t)dec->pooled_bytes;
    if (pic && !rc && mode == 2) rc = pic->scan_rc;
    if (keep && pic && i == reps - 1 && !rc) { pic->dec = nullptr; *keep = pic; }
    else delete pic;
  }
  for (auto& sl : dec->slots) sl.valid = false;
  delete dec;
  return rc;
}

static int finish_scan(de265hip_picture* pic)
{
  if (!pic->scan_pending) return pic->scan_rc;
  de265hip_decoder* dec = pic->dec;
  if (!dec) return pic->scan_rc = DE265HIP_ERROR_DECODING;       // (an orphan: its device side is gone)
  const ScanCounts* K = &pic->h_counts_dry;
  if (!dec->dry) {
    if (pic->enq.pending) { const int erc = de265hip_picture_enqueue(pic); if (erc) return pic->scan_rc = erc; }
    K = dec->h_ring + pic->ring_idx;
    // the last pass stores the counts into this pinned record and then its ready word (system scope): normally long since there
    const uint32_t tag = (uint32_t)pic->ring_seq | 0x80000000u;
    const auto t0 = std::chrono::steady_clock::now();
    struct Acc { de265hip_decoder* d; std::chrono::steady_clock::time_point t; ~Acc() { d->t_scan_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); d->n_scan_wait++; } } acc{ dec, t0 };
    for (uint32_t spins = 0; __atomic_load_n(&K->ready, __ATOMIC_ACQUIRE) != tag; spins++) {
      if ((spins & 255) == 255) {
        if (hipEventQuery(pic->uploaded) == hipSuccess && __atomic_load_n(&K->ready, __ATOMIC_ACQUIRE) != tag) return pic->scan_rc = DE265HIP_ERROR_DECODING;   // (passes done, no word: a fault)
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) return pic->scan_rc = DE265HIP_ERROR_DECODING;
      }
      __builtin_ia32_pause();
    }
    // (the scan ran behind the upload on the same stream: the staging buffer is free)
    if (pic->enq.host_base) {
      std::lock_guard<std::mutex> lk(dec->mu);
      // (only while the buffer is still this picture's: the event path of acquire_stage may have handed it to another one since)
      for (auto& b : dec->stage_pool) if (b.ptr == pic->enq.host_base && b.state == 2 && b.owner == pic->ring_seq) b.state = 0;
      pic->enq.host_base = nullptr;
    }
  }
  pic->scan_pending = false;
  if (K->status) return pic->scan_rc = (int)K->status;
  pic->n_l0 = 0;
  for (int k = 0; k < 4; k++) { pic->n_l0_size[k] = (int)K->n_l0_size[k]; pic->n_l0 += (int)K->n_l0_size[k]; }
  pic->n_l0_rext = (int)K->n_l0_rext;
  pic->n_runs = (int)K->n_runs; pic->n_front = (int)K->n_front; pic->n_batches = (int)K->n_batches; pic->n_tus = (int)K->n_tasks;
  {
    // worker count = widest dependency level (more workers would only wait), within [64, 2 per CU] (see the host scan)
    const char* wenv = d265_env("DE265HIP_RUN_WORKERS");
    const int cap = wenv ? atoi(wenv) : 512;
    const char* menv = d265_env("DE265HIP_RUN_WORKER_PCT");
    const int pct = menv ? atoi(menv) : 125;
    pic->n_workers = std::min(pic->n_batches, std::max(64, std::min(cap, (int)((int64_t)K->widest * pct / 100))));
    const char* denv = d265_env("DE265HIP_RUN_DIRECT");
    pic->run_direct = denv ? atoi(denv) != 0 : false;
    const char* benv = d265_env("DE265HIP_TICKET_BATCH");
    pic->ticket_batch = benv ? std::max(1, std::min(64, atoi(benv))) : 1;
  }
  pic->stats.n_tu_tasks = pic->n_tus; pic->stats.n_runs = pic->n_runs; pic->stats.n_run_levels = (int)K->max_rl;
  pic->stats.n_in_run_levels = (int32_t)K->sum_lvls; pic->stats.n_levels = 0;
  pic->stats.alg_bytes_resid = (int64_t)K->alg_resid; pic->stats.alg_bytes_intra = (int64_t)(K->alg_intra - K->alg_intra_front);
  pic->stats.alg_bytes_intra_front = (int64_t)K->alg_intra_front; pic->stats.n_front_runs = pic->n_front;
  return pic->scan_rc = 0;
}

/* ---- debug / test entry points (not part of the decoding interface) ---- */
// fault injection for the one device-side failure mode the design admits (tests/test_gpu_picture_parity.py): pictures built
// while drop_producer is set leave one run that other runs wait for out of their ticket lists; spin_limit bounds k_run's
// dependency waits so that they expire within milliseconds (0: the default)
int de265hip_debug_fault_injection(de265hip_decoder* dec, int drop_producer, uint32_t spin_limit)
{
  if (!dec) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  std::lock_guard<std::mutex> lk(dec->mu);
  dec->drop_producer = drop_producer != 0;
  dec->spin_limit = spin_limit ? spin_limit : RUN_SPIN_LIMIT_DEFAULT;
  return 0;
}

// where a picture's run-side structures lie in its arena, and the counts that go with them (tests/test_scan_equivalence.py
// reads both scans' arenas back and compares them run by run).  out[0..12]: offsets of runs, run_tus, deps, slots, l0, l0x, mbx,
// mb_segs, front_idx, run_ntus, run_nall, run_level, counts (-1: not there); out[13..]: dev_scan, n_runs, n_front, n_batches,
// n_l0_size[0..3], n_l0_rext, number of run records to look at, arena bytes
int de265hip_debug_picture_layout(de265hip_picture* pic, int64_t out[32])
{
  if (!pic || !out) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  if (const int rc = finish_scan(pic)) return rc;
  for (int i = 0; i < 13; i++) out[i] = pic->o_layout[i];
  out[13] = pic->dev_scan; out[14] = pic->n_runs; out[15] = pic->n_front; out[16] = pic->n_batches;
  for (int k = 0; k < 4; k++) out[17 + k] = pic->n_l0_size[k];
  out[21] = pic->n_l0_rext;
  out[22] = pic->dev_scan ? (int64_t)pic->SP.cap_runs : pic->n_runs;
  out[23] = (int64_t)pic->arena_bytes;
  out[24] = pic->n_workers;
  return 0;
}

// bytes [offset, offset + bytes) of the picture's arena into dst (a device picture: after its passes have completed)
int de265hip_debug_picture_read(de265hip_picture* pic, int64_t offset, int64_t bytes, void* dst)
{
  if (!pic || !dst || offset < 0 || bytes < 0 || (size_t)(offset + bytes) > pic->arena_bytes) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  if (!pic->dry_arena.empty()) { memcpy(dst, pic->dry_arena.data() + offset, (size_t)bytes); return 0; }
  if (!pic->dec || !pic->arena) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  if (pic->enq.pending) { const int erc = de265hip_picture_enqueue(pic); if (erc) return erc; }
  if (pic->uploaded) HIPCHK(hipEventSynchronize(pic->uploaded), DE265HIP_ERROR_DECODING);
  HIPCHK(hipMemcpy(dst, (const uint8_t*)pic->arena + offset, (size_t)bytes, hipMemcpyDeviceToHost), DE265HIP_ERROR_DECODING);
  return 0;
}

int de265hip_picture_get_stats(const de265hip_picture* p, de265hip_picture_stats* s)
{
  if (!p || !s) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  (void)finish_scan(const_cast<de265hip_picture*>(p));      // (device-side scan: the counts behind some of the figures)
  *s = p->stats;
  return 0;
}

}  // extern "C"

namespace {

struct KTimer {
  de265hip_decoder* d; int kid; hipEvent_t a = nullptr, b = nullptr; bool on;
  KTimer(de265hip_decoder* dec, int k, int n_launches) : d(dec), kid(k), on((dec->profiling >> k) & 1u)
  {
    d->launches[kid] += n_launches;
    if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, d->cur_stream); }
  }
  ~KTimer() { if (on) { (void)hipEventRecord(b, d->cur_stream); d->pending.push_back({ kid, a, b }); } }
};

template <typename PX>
int run_picture(de265hip_decoder* dec, de265hip_picture* pic, int last_stage)
{
  // The slot table is read (plane pointers of the references, the destination and the spare) and the destination's planes
  // are swapped with the spare's below: all under the decoder's lock, which the build threads take for slot / pool work.
  std::lock_guard<std::mutex> lk(dec->mu);
  Slot& dst = dec->slots[pic->dst_slot];
  // ---- lane of this launch: the lane whose latest picture is the reference this picture was decoded after (it simply follows
  // it in stream order); a picture without such a predecessor - no references at all, or references that other pictures have
  // followed already - goes to the lane that was given work longest ago and overlaps with what the other lanes are doing.
  int lane = 0;
  if (dec->n_lanes > 1) {
    uint64_t newest = 0; int wl = -1;
    for (int s = 0; s < DE265HIP_MAX_DPB_SLOTS; s++)
      if (((pic->ref_mask >> s) & 1u) && dec->slots[s].writer_lane >= 0 && dec->slots[s].written_seq >= newest) { newest = dec->slots[s].written_seq; wl = dec->slots[s].writer_lane; }
    if (wl >= 0 && wl < dec->n_lanes && dec->lane_tail_seq[wl] == newest) lane = wl;
    else for (int l = 1; l < dec->n_lanes; l++) if (dec->lane_tail_seq[l] < dec->lane_tail_seq[lane]) lane = l;
  }
  Slot& spare = lane_sp(dec, lane);
  {
    // Geometry is the PICTURE's (recorded at build): the destination and the spare are (re)allocated now, in launch order -
    // a change of picture size with pictures of the old size still queued is therefore safe (everything launched before has
    // the old planes, hipFree waits for it) - and every reference must hold a picture of this geometry by now.
    const de265hip_pic_params& pp = pic->params;
    int rc = alloc_slot(dst, pp.width, pp.height, pp.bit_depth_luma, pp.bit_depth_chroma, pp.chroma_format_idc);
    if (!rc) rc = alloc_slot(spare, pp.width, pp.height, pp.bit_depth_luma, pp.bit_depth_chroma, pp.chroma_format_idc);
    if (rc) return rc;
    for (int s = 0; s < DE265HIP_MAX_DPB_SLOTS; s++)
      if ((pic->ref_mask >> s) & 1u) {
        const Slot& r = dec->slots[s];
        if (!r.valid || r.w != pp.width || r.h != pp.height || r.bdY != pp.bit_depth_luma || r.bdC != pp.bit_depth_chroma || r.cf != pp.chroma_format_idc)
          return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
      }
  }
  pic->n_launched++;
  dst.err_idx = pic->ring_idx; dst.err_seq = pic->ring_seq;
  dec->launched_err.emplace_back(pic->ring_idx, pic->ring_seq);
  hipStream_t st = lane_st(dec, lane);
  dec->cur_stream = st;
  pic->lane = lane;
  // a copy-out of the picture this slot held before must have left (de265hip_dpb_download_async)
  // (a deferred one has to be handed to the runtime first: on the host, until the slot's previous picture is done)
  out_settle(dst);
  if (dst.dl_done && dst.dl_waited != dst.dl_seq && hipStreamWaitEvent(st, dst.dl_done, 0) != hipSuccess) return DE265HIP_ERROR_DECODING;
  if (!((pic->upload_waited >> lane) & 1u)) {            // the command buffers arrive on the copy stream
    if (hipStreamWaitEvent(st, pic->uploaded, 0) != hipSuccess) return DE265HIP_ERROR_DECODING;
    pic->upload_waited |= 1u << lane;
  }
  if (dec->n_lanes > 1) {
    // the pictures this one reads (on other lanes), and - for the slot it overwrites - the picture that wrote it and every
    // picture that still reads it
    for (int s = 0; s < DE265HIP_MAX_DPB_SLOTS; s++)
      if ((pic->ref_mask >> s) & 1u) {
        const Slot& r = dec->slots[s];
        if ((r.writer_lane >= 0 && r.writer_lane != lane) || r.writer_lane == kForeignWriter)
          if (hipStreamWaitEvent(st, r.written, 0) != hipSuccess) return DE265HIP_ERROR_DECODING;
      }
    if ((dst.writer_lane >= 0 && dst.writer_lane != lane) || dst.writer_lane == kForeignWriter)
      if (hipStreamWaitEvent(st, dst.written, 0) != hipSuccess) return DE265HIP_ERROR_DECODING;
    for (int l = 0; l < dec->n_lanes; l++)
      if (l != lane && dst.rd_valid[l] && hipStreamWaitEvent(st, dst.read_done[l], 0) != hipSuccess) return DE265HIP_ERROR_DECODING;
  }
  // behind the picture's last kernel: it has written its slot and read its references
  auto lanes_epilogue = [&]() -> bool {
    if (dec->n_lanes == 1) return true;
    const uint64_t seq = ++dec->launch_seq;
    dec->lane_tail_seq[lane] = seq;
    if (!dst.written && hipEventCreateWithFlags(&dst.written, hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventRecord(dst.written, st) != hipSuccess) return false;
    dst.writer_lane = lane; dst.written_seq = seq;
    for (int l = 0; l < kMaxLanes; l++) dst.rd_valid[l] = false;
    for (int s = 0; s < DE265HIP_MAX_DPB_SLOTS; s++)
      if ((pic->ref_mask >> s) & 1u) {
        Slot& r = dec->slots[s];
        if (!r.read_done[lane] && hipEventCreateWithFlags(&r.read_done[lane], hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventRecord(r.read_done[lane], st) != hipSuccess) return false;
        r.rd_valid[lane] = true;
      }
    return true;
  };
  PicDev P = pic->P;
  P.dbg = dec->dbg;
  const PlaneRef d0 = dst.pl[0], d1 = dst.pl[1], d2 = dst.pl[2];

  if (pic->n_mc) {
    DpbTable tab; memset(&tab, 0, sizeof(tab));
    for (int s = 0; s < DE265HIP_MAX_DPB_SLOTS; s++)
      if (dec->slots[s].valid) for (int c = 0; c < 3; c++) tab.p[s][c] = dec->slots[s].pl[c];
    KTimer t(dec, DE265HIP_K_MC, 1);
    if (pic->mc_all) {
      const McBands& B = pic->mc_bands;
      unsigned most = 0;
      for (int b = 0; b < 8; b++) most = std::max(most, B.n_entries[b]);
      hipLaunchKernelGGL(k_mc_all<PX>, dim3(8 * most), dim3(64), 0, st, P, tab, d0, d1, d2, pic->d_mc, pic->d_slices, pic->d_mc_order, B);
    } else
      // (4:2:2 / 4:4:4: the tile body predicts luma only; the chroma planes of a task by a second wavefront, blockIdx.y == 1)
      hipLaunchKernelGGL(k_mc<PX>, dim3(xcd_grid((unsigned)pic->n_mc), (P.chroma_format == 2 || P.chroma_format == 3) ? 2 : 1), dim3(64), 0, st, P, tab, d0, d1, d2, pic->d_mc, pic->d_slices, pic->n_mc);
  }
  if (pic->n_pcm) {
    KTimer t(dec, DE265HIP_K_PCM, 1);
    hipLaunchKernelGGL(k_pcm<PX>, dim3(pic->n_pcm), dim3(256), 0, st, P, d0, d1, d2, pic->d_pcm, pic->d_pcm_samples);
  }
  const int nlev = (int)pic->level_start.size() - 1;
  if (!dec->intra_levels) {
    // run mode: the residuals (inter TUs add into the picture, intra TUs fill the residual buffer), then one launch for
    // the whole intra dependency graph.  (Tried: the intra TUs' residuals and the bS derivation, which depend on nothing
    // before them, on a second HIP stream per decoder: +1 % with one GOP stream, -8 % with three (six streams on four
    // hardware queues; -20 % with GPU_MAX_HW_QUEUES=8), so everything stays on the decoder's one stream.)
    if (pic->n_l0 > 0) {
      // k_resid_big: a 4-wavefront workgroup per 32x32 TU, then four 16x16 TUs per workgroup, a wavefront each (all of a
      // picture's 16x16 TUs are then in flight at once; DE265HIP_RESID16_BIG=1: a workgroup per 16x16 TU as well);
      // k_resid_small: a wavefront per 8x8 TU / per four 4x4 TUs
      const int n32 = pic->n_l0_size[3], n16 = pic->n_l0_size[2], n8 = pic->n_l0_size[1], n4 = pic->n_l0_size[0];
      const int n_wg = dec->resid16_big ? n32 + n16 : n32, n_wave = dec->resid16_big ? 0 : n16;
      // the small TUs ride in k_resid_big's launch too, as wavefront slots behind the 16x16 ones (one launch less per picture:
      // one GOP stream 2 870 -> 2 950 frames/s, three streams unchanged; DE265HIP_RESID_ONE_LAUNCH=0: k_resid_small launch)
      const bool one = dec->resid_one_launch && !dec->resid16_big;
      const int w_small = (n8 + RESID_SPL - 1) / RESID_SPL + (n4 + 4 * RESID_SPL - 1) / (4 * RESID_SPL);
      KTimer t(dec, DE265HIP_K_RESID, one ? 1 : (n32 + n16 > 0) + (n8 + n4 > 0));
      if (one)
        hipLaunchKernelGGL(k_resid_big<PX>, dim3(n_wg + ((n_wave + w_small + 3) >> 2)), dim3(256), 0, st, P, d0, d1, d2, pic->d_l0, n_wg, n_wave,
                           n8, n4, pic->d_cval, pic->d_cpos, pic->d_scaling, pic->d_resid);
      else {
        if (n32 + n16 > 0)
          hipLaunchKernelGGL(k_resid_big<PX>, dim3(n_wg + ((n_wave + 3) >> 2)), dim3(256), 0, st, P, d0, d1, d2, pic->d_l0, n_wg, n_wave,
                             0, 0, pic->d_cval, pic->d_cpos, pic->d_scaling, pic->d_resid);
        if (n8 + n4 > 0)
          hipLaunchKernelGGL(k_resid_small<PX>, dim3(w_small), dim3(64), 0, st, P, d0, d1, d2, pic->d_l0, n32 + n16, n8, n4,
                             pic->d_cval, pic->d_cpos, pic->d_scaling, pic->d_resid);
      }
    }
    if (pic->n_l0_rext > 0) {                                       // TUs with a range-extension tool (RDPCM, rotation, cross-component prediction, big transform skip)
      KTimer t(dec, DE265HIP_K_RESID, 1);
      hipLaunchKernelGGL(k_resid_rext<PX>, dim3(pic->n_l0_rext), dim3(64), 0, st, P, d0, d1, d2, pic->d_l0_rext, pic->n_l0_rext,
                         pic->d_cval, pic->d_cpos, pic->d_scaling, pic->d_resid);
    }
    if (pic->n_front > 0) {                                         // the runs nobody has to wait for: one small workgroup each
      KTimer t(dec, DE265HIP_K_INTRA_FRONT, 1);
      hipLaunchKernelGGL(k_intra_front<PX>, dim3(pic->n_front), dim3(64), 0, st, P, d0, d1, d2, pic->d_runs, pic->d_run_tus, pic->d_resid,
                         pic->n_front, pic->d_front_idx);
    }
    if (pic->n_batches > 0) {
      KTimer t(dec, DE265HIP_K_INTRA, 1);
      // ticket counter and run flags are not cleared between runs of a picture: every workgroup draws exactly one
      // ticket beyond the last batch, so run g starts at ticket g * (n_batches + n_workers), and a flag is "raised"
      // when it holds the run's generation number (cleared once, at build).  Several tickets per draw (experiment)
      // make the count depend on the schedule: clear instead.
      // The generation number of this launch: decoder-wide, never handed out twice - flags and mailbox packets left in the arena
      // by earlier launches, of this picture or of the one that had the arena before, carry smaller numbers and are never
      // cleared.  (On wrap-around every arena is cleared before its next use: arena_epoch.)
      uint32_t base = 0;
      if (++dec->gen_tag == 0) { dec->gen_tag = 1; dec->arena_epoch++; }
      if (pic->arena_buf.epoch != dec->arena_epoch) { (void)hipMemsetAsync(pic->d_sync, 0, pic->clear_bytes, st); pic->arena_buf.epoch = dec->arena_epoch; pic->gen = 0; }
      const uint32_t gen = dec->gen_tag;
      if (pic->run_direct) pic->gen++;                              // (no ticket counter at all)
      else if (pic->ticket_batch == 1) base = (pic->gen++) * (uint32_t)(pic->n_batches + pic->n_workers);
      else { (void)hipMemsetAsync(pic->d_sync, 0, 8, st); }
      hipLaunchKernelGGL((k_run<PX, 64>), dim3(pic->run_direct ? pic->n_batches : pic->n_workers), dim3(64 * dec->run_waves), 0, st, P, d0, d1, d2,
                         pic->d_runs, pic->d_deps, pic->d_sync, (dec->dbg & 16) ? dec->d_err : dec->d_err_ring + pic->ring_idx, pic->d_run_tus, pic->d_resid, pic->d_slots, pic->n_batches,
                         pic->run_direct ? 0 : pic->ticket_batch, base, gen, dec->dbg, dec->spin_limit, pic->d_mbx, pic->d_mbsegs, pic->d_mb);
    }
  } else {
    if (nlev > 0 && pic->level_start[1] > pic->level_start[0]) {
      KTimer t(dec, DE265HIP_K_RESID, 1);
      hipLaunchKernelGGL(k_tu<PX>, dim3(pic->level_start[1] - pic->level_start[0]), dim3(64), 0, st, P, d0, d1, d2,
                         pic->d_tus, pic->level_start[0], pic->d_cval, pic->d_cpos, pic->d_scaling, (int16_t*)nullptr);
    }
    if (nlev > 1) {
      KTimer t(dec, DE265HIP_K_INTRA, nlev - 1);
      for (int l = 1; l < nlev; l++) {
        int cnt = pic->level_start[l + 1] - pic->level_start[l];
        if (cnt <= 0) continue;
        hipLaunchKernelGGL(k_tu<PX>, dim3(cnt), dim3(64), 0, st, P, d0, d1, d2, pic->d_tus, pic->level_start[l],
                           pic->d_cval, pic->d_cpos, pic->d_scaling, (int16_t*)nullptr);
      }
    }
  }
  // DE265HIP_LF_TILE=1: deblocking + SAO in one tiled pass (k_lf_tile) when the picture goes all the way to SAO (parity
  // variant, off by default: see de265hip_decoder::lf_tile)
  const bool want_sao = last_stage >= DE265HIP_STAGE_FINAL && !pic->params.disable_sao && pic->params.sample_adaptive_offset_enabled_flag;
  const bool want_deblock = last_stage >= DE265HIP_STAGE_DEBLOCKED && !pic->params.disable_deblocking && pic->any_edges;
  const bool c420 = P.chroma_format == 1;           // (4:2:2 / 4:4:4: luma through the tuned kernels, the chroma planes through k_*_chroma_any)
  const bool lf_tile = want_sao && dec->lf_tile && !dec->separate_bs && !dec->two_pass_deblock && c420;
  if (lf_tile) {
    Slot& sp = spare;
    LfMeta LM{ pic->d_flags, pic->d_qp, nullptr, pic->d_motion, pic->d_ctbs, pic->d_slices };
    SaoMeta SM{ pic->d_flags, pic->d_sao };
    {
      KTimer t(dec, DE265HIP_K_SAO, 1);
      hipLaunchKernelGGL(k_lf_tile<PX>, dim3((P.width + LF_TILE_W - 1) / LF_TILE_W, (P.height + LF_TILE_H - 1) / LF_TILE_H, 3), dim3(256), 0, st,
                         P, d0, d1, d2, sp.pl[0], sp.pl[1], sp.pl[2], LM, SM, want_deblock ? 1 : 0);
    }
    for (int c = 0; c < 3; c++) std::swap(dst.pl[c], sp.pl[c]);      // output picture now lives in the slot
    { const bool launched_ok = hipGetLastError() == hipSuccess, epilogue_ok = lanes_epilogue();      // (the slot's events are recorded whatever the launches said: later pictures wait for what IS queued)
    if (!launched_ok || !epilogue_ok) return DE265HIP_ERROR_DECODING; }
    return DE265HIP_OK;
  }
  if (want_deblock) {
    // bS (a12) is derived inside the deblocking kernels (one launch and one pass over the unit grid less: 9 us of a 4K
    // picture); DE265HIP_SEPARATE_BS=1 keeps the separate k_bs launch that writes the bS plane first
    if (dec->separate_bs && c420) {
      KTimer t(dec, DE265HIP_K_BS, 1);
      hipLaunchKernelGGL(k_bs, dim3((P.w4 + 255) / 256, P.h4), dim3(256), 0, st, P, pic->d_flags, pic->d_motion, pic->d_bs);
    }
    LfMeta M{ pic->d_flags, pic->d_qp, (dec->separate_bs && c420) ? pic->d_bs : nullptr, pic->d_motion, pic->d_ctbs, pic->d_slices };
    if ((!dec->two_pass_deblock && !dec->separate_bs) || !c420) {
      // both directions in one pass over 8x8 blocks centred on the edge crossings (k_deblock_fused); reported under
      // the "deblock_v" kernel id
      KTimer t(dec, DE265HIP_K_DEBLOCK_V, 1);
      const int nbx = (P.width + 3) / 8 + 1, nby = (P.height + 3) / 8 + 1;
      static const int lf_wg = d265_env("DE265HIP_LF_WG") ? std::max(64, std::min(256, atoi(d265_env("DE265HIP_LF_WG")) & ~63)) : 64;       // (a thread per 8x8 block, no LDS: any workgroup size; one-wavefront workgroups find their place sooner next to the other streams' kernels: 34-35 us against 40 in the product path, 22 alone either way)
      hipLaunchKernelGGL((k_deblock_fused<PX>), dim3((nbx + lf_wg - 1) / lf_wg, nby, c420 ? 3 : 1), dim3(lf_wg), 0, st, P, d0, d1, d2, M);
      if (!c420 && P.chroma_format)
        for (int vertical = 1; vertical >= 0; vertical--) {       // vertical edges of the whole plane before any horizontal one
          const int xi = (vertical ? 2 : 1) << P.csw, yi = (vertical ? 1 : 2) << P.csh;
          hipLaunchKernelGGL((k_deblock_chroma_any<PX>), dim3(((P.w4 + xi - 1) / xi + 255) / 256, (P.h4 + yi - 1) / yi, 2), dim3(256), 0, st,
                             P, d1, d2, M, vertical);
        }
    } else {
      {
        KTimer t(dec, DE265HIP_K_DEBLOCK_V, 1);
        hipLaunchKernelGGL((k_deblock<PX, true>), dim3(((P.w4 + 1) / 2 + 255) / 256, P.h4, 3), dim3(256), 0, st, P, d0, d1, d2, M);
      }
      {
        KTimer t(dec, DE265HIP_K_DEBLOCK_H, 1);
        hipLaunchKernelGGL((k_deblock<PX, false>), dim3((P.w4 + 255) / 256, (P.h4 + 1) / 2, 3), dim3(256), 0, st, P, d0, d1, d2, M);
      }
    }
  }
  if (last_stage >= DE265HIP_STAGE_FINAL && !pic->params.disable_sao && pic->params.sample_adaptive_offset_enabled_flag) {
    Slot& sp = spare;
    SaoMeta M{ pic->d_flags, pic->d_sao };
    {
      KTimer t(dec, DE265HIP_K_SAO, 1);
      if (dec->sao_strips && c420)
        hipLaunchKernelGGL(k_sao<PX>, dim3((P.width / 8 + 4 * SAO_GROUPS * 62 - 1) / (4 * SAO_GROUPS * 62), (P.height + SAO_ROWS - 1) / SAO_ROWS, 3), dim3(256), 0, st, P, d0, d1, d2,
                           sp.pl[0], sp.pl[1], sp.pl[2], M);
      else {
        // luma tile geometry (chroma tiles are half as wide and twice as high: the grid covers both, surplus tiles return)
        const int lsw = std::min(3, P.log2_ctb - 3), tw = 8 << lsw, th = (64 >> lsw) * SAO_ROWS;
        const int lswc = std::min(3, P.log2_ctb - 4), twc = 8 << lswc, thc = (64 >> lswc) * SAO_ROWS;
        // (4:4:4: the chroma planes through the same kernel with luma's tile geometry - one sample per lane and scalar loads in
        //  k_sao_chroma_any took 101 us per 4K10 picture against 11 for the luma plane; 4:2:2 keeps the plain kernel)
        const bool c444 = P.chroma_format == 3;
        const int gx = c444 ? (P.width + tw - 1) / tw : std::max((P.width + tw - 1) / tw, (P.width / 2 + twc - 1) / twc);
        // (wavefronts per workgroup: no LDS, no barrier - any number; DE265HIP_SAO_WG)
        static const int sw = d265_env("DE265HIP_SAO_WG") ? std::max(1, std::min(4, atoi(d265_env("DE265HIP_SAO_WG")) / 64)) : 1;      // (one: 38 us against 44-48 next to the other streams' kernels, 24 alone either way)
        const int gy = c444 ? (P.height + sw * th - 1) / (sw * th) : std::max((P.height + sw * th - 1) / (sw * th), (P.height / 2 + sw * thc - 1) / (sw * thc));
        const uint3 G = make_uint3((unsigned)gx, (unsigned)gy, (c420 || c444) ? 3u : 1u);
        hipLaunchKernelGGL(k_sao_ctb<PX>, dim3(xcd_grid(G.x * G.y * G.z)), dim3(64 * sw), 0, st, P, d0, d1, d2, sp.pl[0], sp.pl[1], sp.pl[2], M, G);
        if (!c420 && !c444 && P.chroma_format)
          hipLaunchKernelGGL(k_sao_chroma_any<PX>, dim3((P.cwidth + 255) / 256, P.cheight, 2), dim3(256), 0, st, P, d1, d2, sp.pl[1], sp.pl[2], M);
      }
    }
    for (int c = 0; c < 3; c++) std::swap(dst.pl[c], sp.pl[c]);      // output picture now lives in the slot
  }
  { const bool launched_ok = hipGetLastError() == hipSuccess, epilogue_ok = lanes_epilogue();      // (the slot's events are recorded whatever the launches said: later pictures wait for what IS queued)
    if (!launched_ok || !epilogue_ok) return DE265HIP_ERROR_DECODING; }
  return DE265HIP_OK;
}

}  // namespace

extern "C" {

int de265hip_picture_run(de265hip_decoder* dec, de265hip_picture* pic, int last_stage)
{
  if (!dec || !pic || pic->dec != dec || last_stage < 0 || last_stage > 2) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  // device-side scan: its counts (task list sizes, tickets, front runs) and its verdict on the records arrive with the event
  // behind the passes - long since complete when the builds run ahead of the launches (the pipeline); outside the decoder's lock
  if (pic->enq.pending) { const int erc = de265hip_picture_enqueue(pic); if (erc) return erc; }
  if (const int rc = finish_scan(pic)) return rc;
  const auto t0 = std::chrono::steady_clock::now();
  const int rrc = pic->P.bd_luma > 8 ? run_picture<uint16_t>(dec, pic, last_stage) : run_picture<uint8_t>(dec, pic, last_stage);
  dec->t_run += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rrc;
}

int de265hip_decoder_sync(de265hip_decoder* dec)
{
  if (!dec) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  HIPCHK(sync_all_lanes(dec), DE265HIP_ERROR_DECODING);
  // a dependency wait that expired (k_run) in any picture launched since the last call: results are not trustworthy
  uint32_t err = 0;
  {
    std::vector<std::pair<int, uint64_t>> chk;
    { std::lock_guard<std::mutex> lk(dec->mu); chk.swap(dec->launched_err); }
    if (!chk.empty()) {
      std::vector<uint32_t> ring(de265hip_decoder::kRing);
      HIPCHK(hipMemcpy(ring.data(), dec->d_err_ring, ring.size() * 4, hipMemcpyDeviceToHost), DE265HIP_ERROR_DECODING);
      for (auto& e : chk)
        if (ring[e.first]) {
          err = 1;
          std::lock_guard<std::mutex> lk(dec->mu);
          if (dec->ring_owner[e.first] == e.second) (void)hipMemset(dec->d_err_ring + e.first, 0, 4);      // reported once
        }
    }
  }
  if (dec->dbg & 16) {                  // diagnostic build switch: dump and clear the phase stamps
    uint32_t st[16];
    HIPCHK(hipMemcpy(st, dec->d_err + 8, sizeof(st), hipMemcpyDeviceToHost), DE265HIP_ERROR_DECODING);
    const double n = st[7] ? (double)st[7] : 1.0;
    fprintf(stderr, "de265hip stamps: runs=%u cycles/run: ticket=%.0f record=%.0f tasks=%.0f prepare(+early window)=%.0f wait=%.0f window=%.0f chain=%.0f\n",
            st[7], st[6] / n, st[0] / n, st[1] / n, st[2] / n, st[3] / n, st[4] / n, st[5] / n);
    (void)hipMemset(dec->d_err + 8, 0, sizeof(st));
  }
  if (err) return DE265HIP_ERROR_DECODING;
  return 0;
}

int de265hip_decode_picture(de265hip_decoder* dec, int dst_slot, const de265hip_picture_desc* d)
{
  de265hip_picture* pic = nullptr;
  int rc = de265hip_picture_build(dec, dst_slot, d, &pic);
  if (rc) return rc;
  rc = de265hip_picture_run(dec, pic, DE265HIP_STAGE_FINAL);
  int rc2 = de265hip_decoder_sync(dec);
  de265hip_picture_free(pic);
  return rc ? rc : rc2;
}

int de265hip_set_profiling(de265hip_decoder* dec, int enable)
{
  if (!dec) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  dec->profiling = enable == 1 ? 0xFFFFFFFFu : (uint32_t)enable >> 1;
  return 0;
}

int de265hip_get_kernel_times(de265hip_decoder* dec, double ms[DE265HIP_K_COUNT],
                              int64_t launches[DE265HIP_K_COUNT], int reset)
{
  if (!dec) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  HIPCHK(sync_all_lanes(dec), DE265HIP_ERROR_DECODING);
  for (auto& e : dec->pending) {
    float t = 0;
    if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) dec->ms[e.kid] += t;
    (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
  }
  dec->pending.clear();
  for (int k = 0; k < DE265HIP_K_COUNT; k++) {
    if (ms) ms[k] = dec->ms[k];
    if (launches) launches[k] = dec->launches[k];
    if (reset) { dec->ms[k] = 0; dec->launches[k] = 0; }
  }
  return 0;
}

// ---- diagnostic: the neighbour units an intra TU of this size and mode reads (the table behind the mode-aware
// dependencies; tests check it against the oracle's predictors by perturbing border samples)
int de265hip_intra_used_units(int log2_size, int intra_mode, int luma, uint64_t* units)
{
  if (log2_size < 2 || log2_size > 5 || intra_mode < 0 || intra_mode > 34 || !units) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  ensure_used_units();
  *units = g_used_units[log2_size - 2][intra_mode][luma ? 1 : 0];
  return 0;
}

// ---- a11 host helper: derive_edgeFlags (deblock.cc:31-225)
int de265hip_derive_edge_flags(const de265hip_pic_params* pp, const de265hip_slice_params* slices, int n_slices,
                               const de265hip_ctb_info* ctbs, const uint8_t* cb_log2_size,
                               const uint8_t* cb_part_mode, const uint8_t* tu_split, uint8_t* blk_flags)
{
  if (!pp || !slices || !ctbs || !cb_log2_size || !cb_part_mode || !tu_split || !blk_flags)
    return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  const de265hip_pic_params& p = *pp;
  Geometry g;
  int rc = make_geometry(p, g);
  if (rc) return rc;
  const int w4 = g.w4, h4 = g.h4;
  const int mincb = 1 << p.log2_min_cb_size;
  const int cbs_w = (p.width + mincb - 1) / mincb, cbs_h = (p.height + mincb - 1) / mincb;
  auto mark = [&](int x, int y, int bits) {
    int xd = x >> 2, yd = y >> 2;
    if (bits && xd < w4 && yd < h4) blk_flags[xd + yd * w4] |= (uint8_t)bits;
  };
  struct Node { int x, y, log2, depth, left, top; };
  std::vector<Node> stack;
  for (int cy = 0; cy < cbs_h; cy++)
    for (int cx = 0; cx < cbs_w; cx++) {
      const int lcb = cb_log2_size[cx + cy * cbs_w];
      if (!lcb) continue;
      const int x0 = cx * mincb, y0 = cy * mincb;
      const int ctb_a = (x0 >> p.log2_ctb_size) + (y0 >> p.log2_ctb_size) * g.ctbs_w;
      if (ctbs[ctb_a].slice_idx >= n_slices) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
      const de265hip_slice_params& sh = slices[ctbs[ctb_a].slice_idx];
      int left = x0 ? DE265HIP_BLK_EDGE_TU_V : 0, top = y0 ? DE265HIP_BLK_EDGE_TU_H : 0;
      const int mask = (1 << p.log2_ctb_size) - 1;
      if (x0 && !(x0 & mask)) {
        const int nb = ctb_a - 1;
        if (!sh.slice_loop_filter_across_slices_enabled_flag && ctbs[nb].slice_addr_rs != ctbs[ctb_a].slice_addr_rs) left = 0;
        else if (!p.loop_filter_across_tiles_enabled_flag && g.tile_id[nb] != g.tile_id[ctb_a]) left = 0;
      }
      if (y0 && !(y0 & mask)) {
        const int nb = ctb_a - g.ctbs_w;
        if (!sh.slice_loop_filter_across_slices_enabled_flag && ctbs[nb].slice_addr_rs != ctbs[ctb_a].slice_addr_rs) top = 0;
        else if (!p.loop_filter_across_tiles_enabled_flag && g.tile_id[nb] != g.tile_id[ctb_a]) top = 0;
      }
      if (sh.slice_deblocking_filter_disabled_flag) continue;
      // transform tree walk (markTransformBlockBoundary)
      stack.clear(); stack.push_back({ x0, y0, lcb, 0, left, top });
      while (!stack.empty()) {
        Node n = stack.back(); stack.pop_back();
        const int split = (tu_split[(n.x >> p.log2_min_tb_size) + (n.y >> p.log2_min_tb_size) * g.tbs_w] >> n.depth) & 1;
        if (split) {
          const int hh = 1 << (n.log2 - 1);
          stack.push_back({ n.x, n.y, n.log2 - 1, n.depth + 1, n.left, n.top });
          stack.push_back({ n.x + hh, n.y, n.log2 - 1, n.depth + 1, DE265HIP_BLK_EDGE_TU_V, n.top });
          stack.push_back({ n.x, n.y + hh, n.log2 - 1, n.depth + 1, n.left, DE265HIP_BLK_EDGE_TU_H });
          stack.push_back({ n.x + hh, n.y + hh, n.log2 - 1, n.depth + 1, DE265HIP_BLK_EDGE_TU_V, DE265HIP_BLK_EDGE_TU_H });
        } else {
          for (int k = 0; k < (1 << n.log2); k += 4) { mark(n.x, n.y + k, n.left); mark(n.x + k, n.y, n.top); }
        }
      }
      // prediction block boundaries (markPredictionBlockBoundary)
      const int cb = 1 << lcb, h2 = cb >> 1, q4 = cb >> 2;
      int vx = -1, hy = -1, vx2 = -1;
      switch (cb_part_mode[cx + cy * cbs_w]) {
        case 1: hy = h2; break;            // PART_2NxN
        case 2: vx = h2; break;            // PART_Nx2N
        case 3: vx = h2; hy = h2; break;   // PART_NxN
        case 4: hy = q4; break;            // PART_2NxnU
        case 5: hy = h2 + q4; break;       // PART_2NxnD
        case 6: vx = q4; break;            // PART_nLx2N
        case 7: vx = h2 + q4; break;       // PART_nRx2N
        default: break;
      }
      (void)vx2;
      for (int k = 0; k < cb; k += 4) {
        if (vx >= 0) mark(x0 + vx, y0 + k, DE265HIP_BLK_EDGE_PB_V);
        if (hy >= 0) mark(x0 + k, y0 + hy, DE265HIP_BLK_EDGE_PB_H);
      }
    }
  return 0;
}

// ------------------------------------------------------------------ recorder (host only)
struct de265hip_recorder {
  de265hip_picture_desc d;
  std::vector<uint8_t> scaling;
  std::vector<de265hip_slice_params> slices;
  std::vector<de265hip_ctb_info> ctbs;
  std::vector<de265hip_tu> tus;
  std::vector<int16_t> cval; std::vector<uint16_t> cpos;
  std::vector<de265hip_pu> pus;
  std::vector<de265hip_pcm> pcms; std::vector<uint16_t> pcm_samples;
  std::vector<uint8_t> flags; std::vector<int8_t> qp; std::vector<de265hip_motion> motion;
  bool have_motion = false;
};

int de265hip_recorder_new(de265hip_recorder** out, const de265hip_pic_params* params, const uint8_t* scaling_factors)
{
  if (!out || !params) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  *out = nullptr;
  if (params->width <= 0 || params->height <= 0 || params->log2_ctb_size < 4 || params->log2_ctb_size > 6)
    return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  if (params->scaling_list_enable_flag && !scaling_factors) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  de265hip_recorder* r = new (std::nothrow) de265hip_recorder();
  if (!r) return DE265HIP_ERROR_OUT_OF_MEMORY;
  memset(&r->d, 0, sizeof(r->d));
  r->d.params = *params;
  if (scaling_factors) r->scaling.assign(scaling_factors, scaling_factors + DE265HIP_SCALING_BLOB_BYTES);
  const int ctb = 1 << params->log2_ctb_size;
  const int nctb = ((params->width + ctb - 1) / ctb) * ((params->height + ctb - 1) / ctb);
  r->ctbs.assign((size_t)nctb, de265hip_ctb_info{});
  const size_t nblk = (size_t)((params->width + 3) / 4) * ((params->height + 3) / 4);
  r->flags.assign(nblk, 0); r->qp.assign(nblk, 0);
  *out = r;
  return DE265HIP_OK;
}

void de265hip_recorder_free(de265hip_recorder* r) { delete r; }

int de265hip_record_tu(de265hip_recorder* r, const de265hip_tu* tu, const int16_t* vals, const uint16_t* pos)
{
  if (!r || !tu || (tu->n_coeff && (!vals || !pos))) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  de265hip_tu t = *tu;
  t.coeff_offset = (uint32_t)r->cval.size();
  r->cval.insert(r->cval.end(), vals, vals + t.n_coeff);
  r->cpos.insert(r->cpos.end(), pos, pos + t.n_coeff);
  r->tus.push_back(t);
  return DE265HIP_OK;
}

int de265hip_record_pu(de265hip_recorder* r, const de265hip_pu* pu)
{
  if (!r || !pu) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  r->pus.push_back(*pu);
  return DE265HIP_OK;
}

int de265hip_record_pcm(de265hip_recorder* r, int x0, int y0, int log2_cb_size, const uint16_t* samples)
{
  if (!r || !samples || log2_cb_size < 3 || log2_cb_size > 5 || x0 < 0 || y0 < 0) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  de265hip_pcm p; memset(&p, 0, sizeof(p));
  p.x0 = (uint16_t)x0; p.y0 = (uint16_t)y0; p.log2_cb_size = (uint8_t)log2_cb_size;
  p.sample_offset = (uint32_t)r->pcm_samples.size();
  const int n = 1 << log2_cb_size;
  const int rcf = r->d.params.chroma_format_idc;
  r->pcm_samples.insert(r->pcm_samples.end(), samples, samples + n * n + (rcf ? 2 * (n / (rcf == 3 ? 1 : 2)) * (n / (rcf == 1 ? 2 : 1)) : 0));
  r->pcms.push_back(p);
  return DE265HIP_OK;
}

int de265hip_record_slice(de265hip_recorder* r, const de265hip_slice_params* s)
{
  if (!r || !s) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  r->slices.push_back(*s);
  return DE265HIP_OK;
}

int de265hip_record_ctb(de265hip_recorder* r, int addr, const de265hip_ctb_info* info)
{
  if (!r || !info || addr < 0 || addr >= (int)r->ctbs.size()) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  r->ctbs[addr] = *info;
  return DE265HIP_OK;
}

int de265hip_record_blk_planes(de265hip_recorder* r, const uint8_t* f, const int8_t* q, const de265hip_motion* m)
{
  if (!r || !f || !q) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  memcpy(r->flags.data(), f, r->flags.size());
  memcpy(r->qp.data(), q, r->qp.size());
  r->have_motion = m != nullptr;
  if (m) r->motion.assign(m, m + r->flags.size());
  return DE265HIP_OK;
}

const de265hip_picture_desc* de265hip_recorder_desc(de265hip_recorder* r)
{
  if (!r) return nullptr;
  de265hip_picture_desc& d = r->d;
  d.scaling_factors = r->scaling.empty() ? nullptr : r->scaling.data();
  d.n_slices = (int32_t)r->slices.size(); d.slices = r->slices.data();
  d.n_ctbs = (int32_t)r->ctbs.size(); d.ctbs = r->ctbs.data();
  d.n_tus = (int32_t)r->tus.size(); d.tus = r->tus.data();
  d.n_coeffs = (int32_t)r->cval.size(); d.coeff_val = r->cval.data(); d.coeff_pos = r->cpos.data();
  d.n_pus = (int32_t)r->pus.size(); d.pus = r->pus.data();
  d.n_pcms = (int32_t)r->pcms.size(); d.pcms = r->pcms.data();
  d.n_pcm_samples = (int32_t)r->pcm_samples.size(); d.pcm_samples = r->pcm_samples.data();
  d.blk_flags = r->flags.data(); d.blk_qp_y = r->qp.data();
  d.blk_motion = r->have_motion ? r->motion.data() : nullptr;
  return &d;
}

int de265hip_recorder_submit(de265hip_decoder* dec, int dst_slot, de265hip_recorder* r, de265hip_picture** out)
{
  if (!r) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  return de265hip_picture_build(dec, dst_slot, de265hip_recorder_desc(r), out);
}

// ------------------------------------------------------------------ Part B
namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t n) { return hipMalloc(&p, n ? n : 1) == hipSuccess ? 0 : DE265HIP_ERROR_OUT_OF_MEMORY; }
};
int fn_check_blocks(int n, const int32_t* xy, int w, int h, int pw, int ph, int mx0, int my0, int mx1, int my1)
{
  if (n < 0 || (n && !xy)) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  for (int i = 0; i < n; i++) {
    int x = xy[2 * i], y = xy[2 * i + 1];
    if (x - mx0 < 0 || y - my0 < 0 || x + w + mx1 > pw || y + h + my1 > ph) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  }
  return 0;
}
}  // namespace

static int fn_residual(int kind, int log2_size, int bit_depth, void* plane, ptrdiff_t stride, int plane_h, int n,
                       const int32_t* xy, const int16_t* coeffs)
{
  if (!plane || !coeffs || log2_size < 2 || log2_size > 5 || bit_depth < 8 || bit_depth > 12 || stride <= 0 ||
      (kind == 1 && log2_size != 2))
    return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  const int nT = 1 << log2_size;
  int rc = fn_check_blocks(n, xy, nT, nT, (int)stride, plane_h, 0, 0, 0, 0);
  if (rc || n == 0) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DE265HIP_ERROR_INIT_FAILED;
  const size_t bpp = px_bytes(bit_depth), pbytes = (size_t)stride * plane_h * bpp;
  DevBuf dp, dxy, dc;
  if (dp.alloc(pbytes) || dxy.alloc((size_t)n * 8) || dc.alloc((size_t)n * nT * nT * 2)) return DE265HIP_ERROR_OUT_OF_MEMORY;
  HIPCHK(hipMemcpy(dp.p, plane, pbytes, hipMemcpyHostToDevice), DE265HIP_ERROR_DECODING);
  HIPCHK(hipMemcpy(dxy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice), DE265HIP_ERROR_DECODING);
  HIPCHK(hipMemcpy(dc.p, coeffs, (size_t)n * nT * nT * 2, hipMemcpyHostToDevice), DE265HIP_ERROR_DECODING);
  if (bit_depth > 8)
    hipLaunchKernelGGL(k_fn_residual<uint16_t>, dim3(n), dim3(64), 0, 0, kind, log2_size, bit_depth, (uint16_t*)dp.p,
                       (int)stride, (const int32_t*)dxy.p, (const int16_t*)dc.p);
  else
    hipLaunchKernelGGL(k_fn_residual<uint8_t>, dim3(n), dim3(64), 0, 0, kind, log2_size, bit_depth, (uint8_t*)dp.p,
                       (int)stride, (const int32_t*)dxy.p, (const int16_t*)dc.p);
  HIPCHK(hipDeviceSynchronize(), DE265HIP_ERROR_DECODING);
  HIPCHK(hipMemcpy(plane, dp.p, pbytes, hipMemcpyDeviceToHost), DE265HIP_ERROR_DECODING);
  return 0;
}

int de265hip_fn_transform_add(int log2_size, int dst_type, int bit_depth, void* plane, ptrdiff_t stride, int plane_h,
                              int n, const int32_t* xy, const int16_t* coeffs)
{ return fn_residual(dst_type ? 1 : 0, log2_size, bit_depth, plane, stride, plane_h, n, xy, coeffs); }
int de265hip_fn_transform_skip_add(int log2_size, int bit_depth, void* plane, ptrdiff_t stride, int plane_h, int n,
                                   const int32_t* xy, const int16_t* coeffs)
{ return fn_residual(2, log2_size, bit_depth, plane, stride, plane_h, n, xy, coeffs); }
int de265hip_fn_transform_bypass_add(int log2_size, int bit_depth, void* plane, ptrdiff_t stride, int plane_h, int n,
                                     const int32_t* xy, const int16_t* coeffs)
{ return fn_residual(3, log2_size, bit_depth, plane, stride, plane_h, n, xy, coeffs); }

static int fn_interp(int luma, int bit_depth, const void* plane, ptrdiff_t stride, int pw, int ph, int w, int h,
                     int fx, int fy, int n, const int32_t* xy, int16_t* out)
{
  if (!plane || !out || bit_depth < 8 || bit_depth > 12 || w <= 0 || h <= 0 || w > 64 || h > 64 || stride < pw ||
      fx < 0 || fy < 0 || fx > (luma ? 3 : 7) || fy > (luma ? 3 : 7))
    return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  // the vtable contract: source block including filter margins lies inside the plane
  const int b = luma ? 3 : 1, a = luma ? 4 : 2;
  int rc = fn_check_blocks(n, xy, w, h, pw, ph, fx ? b : 0, fy ? b : 0, fx ? a : 0, fy ? a : 0);
  if (rc || n == 0) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DE265HIP_ERROR_INIT_FAILED;
  const size_t bpp = px_bytes(bit_depth), pbytes = (size_t)stride * ph * bpp, obytes = (size_t)n * w * h * 2;
  DevBuf dp, dxy, dout;
  if (dp.alloc(pbytes) || dxy.alloc((size_t)n * 8) || dout.alloc(obytes)) return DE265HIP_ERROR_OUT_OF_MEMORY;
  HIPCHK(hipMemcpy(dp.p, plane, pbytes, hipMemcpyHostToDevice), DE265HIP_ERROR_DECODING);
  HIPCHK(hipMemcpy(dxy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice), DE265HIP_ERROR_DECODING);
  if (bit_depth > 8)
    hipLaunchKernelGGL(k_fn_interp<uint16_t>, dim3(n), dim3(64), 0, 0, luma, bit_depth, (const uint16_t*)dp.p, (int)stride,
                       pw, ph, w, h, fx, fy, (const int32_t*)dxy.p, (int16_t*)dout.p);
  else
    hipLaunchKernelGGL(k_fn_interp<uint8_t>, dim3(n), dim3(64), 0, 0, luma, bit_depth, (const uint8_t*)dp.p, (int)stride,
                       pw, ph, w, h, fx, fy, (const int32_t*)dxy.p, (int16_t*)dout.p);
  HIPCHK(hipDeviceSynchronize(), DE265HIP_ERROR_DECODING);
  HIPCHK(hipMemcpy(out, dout.p, obytes, hipMemcpyDeviceToHost), DE265HIP_ERROR_DECODING);
  return 0;
}

int de265hip_fn_put_qpel(int bit_depth, const void* src_plane, ptrdiff_t stride, int pw, int ph, int w, int h,
                         int dx, int dy, int n, const int32_t* xy, int16_t* out)
{ return fn_interp(1, bit_depth, src_plane, stride, pw, ph, w, h, dx, dy, n, xy, out); }
int de265hip_fn_put_epel(int bit_depth, const void* src_plane, ptrdiff_t stride, int pw, int ph, int w, int h,
                         int mx, int my, int n, const int32_t* xy, int16_t* out)
{ return fn_interp(0, bit_depth, src_plane, stride, pw, ph, w, h, mx, my, n, xy, out); }

int de265hip_fn_put_pred(int mode, int bit_depth, void* plane, ptrdiff_t stride, int plane_h, int w, int h, int n,
                         const int32_t* xy, const int16_t* src0, const int16_t* src1, int w0, int o0, int w1, int o1,
                         int log2wd)
{
  if (!plane || !src0 || mode < 0 || mode > 3 || ((mode == 2 || mode == 3) && !src1) || bit_depth < 8 ||
      bit_depth > 12 || w <= 0 || h <= 0 || stride <= 0 || ((mode == 1 || mode == 3) && log2wd < 1))
    return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  int rc = fn_check_blocks(n, xy, w, h, (int)stride, plane_h, 0, 0, 0, 0);
  if (rc || n == 0) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DE265HIP_ERROR_INIT_FAILED;
  const size_t bpp = px_bytes(bit_depth), pbytes = (size_t)stride * plane_h * bpp, sbytes = (size_t)n * w * h * 2;
  DevBuf dp, dxy, d0, d1;
  if (dp.alloc(pbytes) || dxy.alloc((size_t)n * 8) || d0.alloc(sbytes) || d1.alloc(sbytes)) return DE265HIP_ERROR_OUT_OF_MEMORY;
  HIPCHK(hipMemcpy(dp.p, plane, pbytes, hipMemcpyHostToDevice), DE265HIP_ERROR_DECODING);
  HIPCHK(hipMemcpy(dxy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice), DE265HIP_ERROR_DECODING);
  HIPCHK(hipMemcpy(d0.p, src0, sbytes, hipMemcpyHostToDevice), DE265HIP_ERROR_DECODING);
  if (src1) HIPCHK(hipMemcpy(d1.p, src1, sbytes, hipMemcpyHostToDevice), DE265HIP_ERROR_DECODING);
  if (bit_depth > 8)
    hipLaunchKernelGGL(k_fn_put<uint16_t>, dim3(n), dim3(64), 0, 0, mode, bit_depth, (uint16_t*)dp.p, (int)stride, w, h,
                       (const int32_t*)dxy.p, (const int16_t*)d0.p, src1 ? (const int16_t*)d1.p : nullptr, w0, o0, w1, o1, log2wd);
  else
    hipLaunchKernelGGL(k_fn_put<uint8_t>, dim3(n), dim3(64), 0, 0, mode, bit_depth, (uint8_t*)dp.p, (int)stride, w, h,
                       (const int32_t*)dxy.p, (const int16_t*)d0.p, src1 ? (const int16_t*)d1.p : nullptr, w0, o0, w1, o1, log2wd);
  HIPCHK(hipDeviceSynchronize(), DE265HIP_ERROR_DECODING);
  HIPCHK(hipMemcpy(plane, dp.p, pbytes, hipMemcpyDeviceToHost), DE265HIP_ERROR_DECODING);
  return 0;
}

}  // extern "C"
