// pipeline.hip -- SURVEY.md 8(f3): picture-level pipelining behind the C ABI (include/de265_hip.h, "pipeline").
//
// The reference overlaps parsing and reconstruction with a thread pool inside one picture (decctx.cc:976-1178: WPP rows / tiles /
// slices as tasks).  With the reconstruction on the device, what has to overlap is: the host parser on picture n+1, the host stage
// of this library (recorder -> de265hip_picture_build_host: validation, MC tasks, staging) for pictures
// n, n-1, .. on worker threads, and the kernels + copy-out of the pictures before those on the device.  A pipeline owns the worker
// threads and the ordering rule: pictures are BUILT concurrently and LAUNCHED in submission order (a picture's kernels read the
// DPB slots its references were launched into).  Host-only code: no kernel lives here.
#include "../../include/de265_hip.h"
#include "env.h"

#include <array>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

struct PipeJob {
  uint64_t ticket = 0;
  int slot = 0;
  de265hip_prepare_fn prepare = nullptr;
  void* user = nullptr;
  const de265hip_picture_desc* desc = nullptr;  // de265hip_pipeline_submit_desc: a ready-made description instead of prepare()
  void* plane[3] = { nullptr, nullptr, nullptr };
  ptrdiff_t stride[3] = { 0, 0, 0 };
  double t_sub = 0;
};

}  // namespace

struct de265hip_pipeline {
  de265hip_decoder* dec = nullptr;
  int n_workers = 1;
  std::vector<std::thread> th;
  std::thread launcher_th;
  std::mutex mu;
  std::condition_variable cv;                   // submit / wait / drain
  std::condition_variable cv_launch;            // the launcher: a picture has been built
  std::deque<PipeJob> q;                        // submitted, not yet taken by a worker
  std::map<uint64_t, std::pair<int, uint64_t>> slot_of;   // launched, copy-out possibly still in flight: ticket -> slot, number of its copy-out (0: none)
  std::map<uint64_t, int> failed;               // ticket -> error of prepare / build / run
  struct Built { PipeJob job; de265hip_picture* pic; int rc; bool enqueued; double t_sub, t_b0, t_b1, t_enq, t_ready; };
  std::vector<std::array<double, 4>> chain_trace;   // DE265HIP_PIPE_TRACE: per scan chain: enqueued, reported, pictures built / in flight then
  std::vector<std::array<double, 8>> trace;     // DE265HIP_PIPE_TRACE: per picture: ticket, submitted, build start / end, enqueued, launch start / end
  std::map<uint64_t, Built> ready;              // built (or failed), waiting for their turn to be launched
  uint64_t next_ticket = 0, next_launch = 0;    // tickets are handed out and launched in submission order
  int in_flight = 0;                            // queued or being built, not yet launched
  int window = 4;                               // bound of in_flight (submit blocks)
  int batch = 8;                                // pictures enqueued together at most (their scans share their launches, four pictures per set)
  int chains = 1;                               // enqueues in flight at most (each up to `batch` pictures: two sets of launches on two scan streams)
  bool stop = false;
  // DE265HIP_PIPE_TIMING=1: where the threads' time goes (seconds, summed; printed when the pipeline is freed)
  bool timing = false, tracing = false;
  double t_idle = 0, t_build = 0, t_enqueue = 0, t_launch = 0, t_lidle = 0, t_chain = 0; long n_jobs = 0, n_chains = 0;
};

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// A worker: the HOST stage of its pictures, concurrently with the other workers' (prepare -> recorder, de265hip_picture_build_host:
// validation, MC tasks, staging into pinned memory).  No HIP call that touches a stream: those are the launcher's.
void worker(de265hip_pipeline* p)
{
  for (;;) {
    PipeJob j;
    const double t0 = now();
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv.wait(lk, [&] { return p->stop || !p->q.empty(); });
      if (p->q.empty()) return;
      j = p->q.front(); p->q.pop_front();
    }
    const double t1 = now();
    de265hip_recorder* rec = nullptr;
    de265hip_picture* pic = nullptr;
    int rc = 0;
    if (j.desc) rc = de265hip_picture_build_host(p->dec, j.slot, j.desc, &pic);
    else {
      rc = j.prepare(j.user, &rec);
      if (!rc && !rec) rc = DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
      if (!rc) rc = de265hip_picture_build_host(p->dec, j.slot, de265hip_recorder_desc(rec), &pic);
      if (rec) de265hip_recorder_free(rec);
    }
    const double t2 = now();
    {
      std::lock_guard<std::mutex> lk(p->mu);
      p->ready[j.ticket] = de265hip_pipeline::Built{ j, pic, rc, false, j.t_sub, t1, t2, 0, 0 };
      if (p->timing) { p->t_idle += t1 - t0; p->t_build += t2 - t1; p->n_jobs++; }
    }
    p->cv_launch.notify_one();
  }
}

// The launcher: the ONE thread that talks to the HIP runtime for this pipeline's decoder.  Round 3 let whichever worker finished
// a build enqueue its upload and launch the pictures whose turn had come; with the device-side scan a build issues a dozen HIP
// calls and fifteen workers calling into the runtime for one device spent three quarters of their builds waiting for its
// locks (builds of 1.7 ms took 6.5 ms).  Pictures are ENQUEUED (upload + scan of their records) as soon as they are built, in
// any order, and LAUNCHED in submission order (also when a picture failed: the turn must pass on).
void launcher(de265hip_pipeline* p)
{
  // pictures of the chains (upload + scan launched together) whose scans have not reported yet: the last picture of each
  std::vector<de265hip_picture*> open_chains;
  std::vector<double> open_since;
  static const bool in_order = !d265_env("DE265HIP_PIPE_ANY_ORDER");
  static const int poll_us = d265_env("DE265HIP_PIPE_POLL_US") ? std::max(1, atoi(d265_env("DE265HIP_PIPE_POLL_US"))) : 50;
  for (;;) {
    de265hip_pipeline::Built todo; bool have_enq = false, have_launch = false;
    de265hip_picture* enq_pic[8]; uint64_t enq_tk[8]; int n_enq = 0;
    const double t0 = now();
    {
      std::unique_lock<std::mutex> lk(p->mu);
      for (;;) {
        if (p->tracing) for (auto& kv : p->ready) if (kv.second.enqueued && kv.second.pic && kv.second.t_ready == 0 && de265hip_picture_ready(kv.second.pic) == 1) kv.second.t_ready = now();
        for (size_t i = 0; i < open_chains.size();)
          if (de265hip_picture_ready(open_chains[i]) != 0) {
            if (p->timing) p->t_chain += now() - open_since[i];
            if (p->tracing) p->chain_trace.push_back({ open_since[i], now(), (double)p->ready.size(), (double)p->in_flight });
            open_chains.erase(open_chains.begin() + i); open_since.erase(open_since.begin() + i);
          } else i++;
        // Uploads and scans first: they run ahead of the launches on the copy streams.  The scan of a picture is a chain of
        // six dependent kernels, ~0.6 ms alone and 1-2 ms next to the reconstruction kernels, whatever the number of pictures
        // it works on (grid.y = picture), and the device runs only so many chains at once (three scan streams per device): what
        // counts is pictures per chain.  So a decoder keeps at most `chains` (DE265HIP_PIPE_CHAINS, default 1) in flight and
        // hands the next one every picture that has been built meanwhile (up to DE265HIP_PIPE_BATCH): one picture at once when
        // the device is idle, full batches when the scans are what everybody waits for.  (Round 4 before: every built picture
        // enqueued at once, 1.5 pictures per chain on average, the scan streams saturated at 3 900 pictures/s without any
        // reconstruction and 2 600 with it.)
        // ... and only pictures in decode order without a gap: a chain that takes pictures 6-8 while picture 5 is still being
        // built finishes for nothing - they are launched behind picture 5, whose chain comes later - and holds a chain slot
        n_enq = 0;
        if ((int)open_chains.size() < p->chains && !in_order) {
          for (auto& kv : p->ready) if (!kv.second.enqueued && !kv.second.rc && kv.second.pic && n_enq < p->batch) { enq_pic[n_enq] = kv.second.pic; enq_tk[n_enq++] = kv.first; }
        } else if ((int)open_chains.size() < p->chains) {
          for (uint64_t tk = p->next_launch; n_enq < p->batch; tk++) {
            auto e = p->ready.find(tk);
            if (e == p->ready.end()) break;                              // (not built yet: the chain waits for it)
            if (e->second.enqueued || e->second.rc || !e->second.pic) continue;
            enq_pic[n_enq] = e->second.pic; enq_tk[n_enq++] = tk;
          }
        }
        if (n_enq) { have_enq = true; break; }
        auto it = p->ready.find(p->next_launch);
        if (it != p->ready.end() && (it->second.enqueued || it->second.rc || !it->second.pic)) {
          // its turn - if its scan has reported (or it failed: the turn passes on).  The launcher never blocks on a scan while
          // pictures may arrive that want enqueueing
          if (it->second.rc || !it->second.pic || de265hip_picture_ready(it->second.pic) != 0) { todo = it->second; p->ready.erase(it); have_launch = true; break; }
          p->cv_launch.wait_for(lk, std::chrono::microseconds(poll_us));
          continue;
        }
        if (!open_chains.empty() || it != p->ready.end()) { p->cv_launch.wait_for(lk, std::chrono::microseconds(poll_us)); continue; }
        if (p->stop) return;
        p->cv_launch.wait(lk);
      }
    }
    const double t1 = now();
    if (have_enq) {
      const int rc = de265hip_picture_enqueue_batch(enq_pic, n_enq);
      if (!rc) { open_chains.push_back(enq_pic[n_enq - 1]); open_since.push_back(now()); p->n_chains++; }
      std::lock_guard<std::mutex> lk(p->mu);
      for (int i = 0; i < n_enq; i++) {
        auto it = p->ready.find(enq_tk[i]);
        if (it != p->ready.end()) { it->second.enqueued = true; it->second.t_enq = now(); if (rc) it->second.rc = rc; }
      }
      if (p->timing) { p->t_lidle += t1 - t0; p->t_enqueue += now() - t1; }
      continue;
    }
    if (have_launch) {
      int r = todo.rc;
      uint64_t copy_out_id = 0;
      static const bool no_run = d265_env("DE265HIP_PIPE_NO_RUN") != nullptr;      // (experiment: builds, uploads and scans only)
      if (!r && !no_run) r = de265hip_picture_run(p->dec, todo.pic, DE265HIP_STAGE_FINAL);
      if (!r && (todo.job.plane[0] || todo.job.plane[1] || todo.job.plane[2]))
        r = de265hip_dpb_download_planes_async(p->dec, todo.job.slot, todo.job.plane, todo.job.stride, &copy_out_id);
      if (todo.pic) de265hip_picture_free(todo.pic);          // never waits (de265_hip.h LIFETIME)
      {
        std::lock_guard<std::mutex> lk(p->mu);
        if (r) p->failed[todo.job.ticket] = r; else p->slot_of[todo.job.ticket] = { todo.job.slot, copy_out_id };
        p->next_launch++; p->in_flight--;
        if (p->timing) { p->t_lidle += t1 - t0; p->t_launch += now() - t1; }
        if (p->tracing) p->trace.push_back({ (double)todo.job.ticket, todo.t_sub, todo.t_b0, todo.t_b1, todo.t_enq, t1, now(), todo.t_ready });
      }
      p->cv.notify_all();
    }
  }
}

}  // namespace

extern "C" {

int de265hip_pipeline_new(de265hip_pipeline** out, de265hip_decoder* dec, int n_workers)
{
  if (!out || !dec || n_workers < 1 || n_workers > 16) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  de265hip_pipeline* p = new (std::nothrow) de265hip_pipeline();
  if (!p) return DE265HIP_ERROR_OUT_OF_MEMORY;
  p->dec = dec; p->n_workers = n_workers; p->timing = d265_env("DE265HIP_PIPE_TIMING") != nullptr; p->tracing = d265_env("DE265HIP_PIPE_TRACE") != nullptr;
  p->window = 4 * n_workers + 4;
  if (const char* w = d265_env("DE265HIP_PIPE_WINDOW")) p->window = std::max(1, atoi(w));
  if (const char* b = d265_env("DE265HIP_PIPE_BATCH")) p->batch = std::min(8, std::max(1, atoi(b)));
  if (const char* c = d265_env("DE265HIP_PIPE_CHAINS")) p->chains = std::min(8, std::max(1, atoi(c)));
  p->window = std::max(p->window, p->chains * p->batch + 2 * n_workers + 4);       // (the chains full, every worker busy, a few to spare)
  for (int i = 0; i < n_workers; i++) p->th.emplace_back(worker, p);
  p->launcher_th = std::thread(launcher, p);
  *out = p;
  return 0;
}

static int pipeline_submit_job(de265hip_pipeline* p, int dst_slot, de265hip_prepare_fn prepare, void* user, const de265hip_picture_desc* desc,
                               void* const planes[3], const ptrdiff_t stride_bytes[3], uint64_t* ticket)
{
  if (!p || (!prepare && !desc) || dst_slot < 0 || dst_slot >= DE265HIP_MAX_DPB_SLOTS) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  PipeJob j;
  j.slot = dst_slot; j.prepare = prepare; j.user = user; j.desc = desc;
  for (int c = 0; c < 3; c++) { j.plane[c] = planes ? planes[c] : nullptr; j.stride[c] = (planes && stride_bytes) ? stride_bytes[c] : 0; }
  {
    std::unique_lock<std::mutex> lk(p->mu);
    // bounded: a few pictures between parser and device.  Four times the workers: pictures are launched in submission order, so
    // while one worker is busy with an expensive picture (an all-intra picture's host stage takes 3.6x a B picture's) the
    // others need that many cheaper ones behind it to stay busy (measured, 3 x 5 workers, 4K10 GOPs of 1 I + 15 B: window 12
    // -> workers idle 2.7-3.4 ms per picture, 1 040-1 070 pictures/s; 24 -> 0.5-0.9 ms, 1 307; 40 -> 1 291)
    p->cv.wait(lk, [&] { return p->in_flight < p->window; });
    j.ticket = p->next_ticket++; j.t_sub = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    p->in_flight++;
    p->q.push_back(j);
  }
  p->cv.notify_all();
  if (ticket) *ticket = j.ticket;
  return 0;
}

int de265hip_pipeline_submit(de265hip_pipeline* p, int dst_slot, de265hip_prepare_fn prepare, void* user,
                             void* const planes[3], const ptrdiff_t stride_bytes[3], uint64_t* ticket)
{
  if (!prepare) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  return pipeline_submit_job(p, dst_slot, prepare, user, nullptr, planes, stride_bytes, ticket);
}

int de265hip_pipeline_submit_desc(de265hip_pipeline* p, int dst_slot, const de265hip_picture_desc* desc,
                                  void* const planes[3], const ptrdiff_t stride_bytes[3], uint64_t* ticket)
{
  if (!desc) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  return pipeline_submit_job(p, dst_slot, nullptr, nullptr, desc, planes, stride_bytes, ticket);
}

int de265hip_pipeline_wait(de265hip_pipeline* p, uint64_t ticket)
{
  if (!p) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  std::pair<int, uint64_t> so;
  {
    std::unique_lock<std::mutex> lk(p->mu);
    if (ticket >= p->next_ticket) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
    p->cv.wait(lk, [&] { return p->next_launch > ticket; });             // launched (or failed) by its worker
    auto f = p->failed.find(ticket);
    if (f != p->failed.end()) { const int rc = f->second; p->failed.erase(f); return rc; }
    auto s = p->slot_of.find(ticket);
    if (s == p->slot_of.end()) return 0;                                  // waited for before
    so = s->second;
    p->slot_of.erase(s);
  }
  // outside the lock: the workers go on launching.  THIS picture's copy-out, not the slot's latest: the slot may have been
  // decoded into and copied out again since (an output queue deeper than the DPB's cycle)
  return so.second ? de265hip_dpb_wait_copy_out(p->dec, so.first, so.second) : de265hip_dpb_wait(p->dec, so.first);
}

int de265hip_pipeline_drain(de265hip_pipeline* p)
{
  if (!p) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv.wait(lk, [&] { return p->next_launch == p->next_ticket; });
  }
  int rc = de265hip_decoder_sync(p->dec);
  std::vector<int> slots;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    for (auto& kv : p->slot_of) slots.push_back(kv.second.first);
    p->slot_of.clear();
    if (!rc && !p->failed.empty()) rc = p->failed.begin()->second;
    p->failed.clear();
  }
  for (int s : slots) { const int r = de265hip_dpb_wait(p->dec, s); if (!rc) rc = r; }
  return rc;
}

void de265hip_pipeline_free(de265hip_pipeline* p)
{
  if (!p) return;
  (void)de265hip_pipeline_drain(p);
  { std::lock_guard<std::mutex> lk(p->mu); p->stop = true; }
  p->cv.notify_all(); p->cv_launch.notify_all();
  for (auto& t : p->th) t.join();
  p->launcher_th.join();
  if (p->timing && p->n_jobs)
    fprintf(stderr, "de265hip pipeline: %ld pictures, %d workers + 1 launcher; ms per picture: workers idle %.2f build (host stage) %.2f | launcher idle %.2f enqueue (upload + scan) %.2f launch %.2f | %.2f pictures per scan chain, a chain reports after %.2f ms\n",
            p->n_jobs, p->n_workers, 1e3 * p->t_idle / p->n_jobs, 1e3 * p->t_build / p->n_jobs, 1e3 * p->t_lidle / p->n_jobs,
            1e3 * p->t_enqueue / p->n_jobs, 1e3 * p->t_launch / p->n_jobs, (double)p->n_jobs / std::max(1L, p->n_chains), 1e3 * p->t_chain / std::max(1L, p->n_chains));
  if (p->tracing) for (auto& r : p->chain_trace) fprintf(stderr, "chaintrace %p %.6f %.6f %.0f %.0f\n", (void*)p, r[0], r[1], r[2], r[3]);
  if (p->tracing) for (auto& r : p->trace) fprintf(stderr, "pipetrace %p %.0f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n", (void*)p, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
  delete p;
}

}  // extern "C"
