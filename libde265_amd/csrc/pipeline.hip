// pipeline.hip -- SURVEY.md 8(f3): picture-level pipelining behind the C ABI (include/de265_hip.h, "pipeline").
//
// The reference overlaps parsing and reconstruction with a thread pool inside one picture (decctx.cc:976-1178: WPP rows / tiles /
// slices as tasks).  With the reconstruction on the device, what has to overlap is: the host parser on picture n+1, the host stage
// of this library (recorder -> de265hip_picture_build: availability replay, run construction, staging, upload) for pictures
// n, n-1, .. on worker threads, and the kernels + copy-out of the pictures before those on the device.  A pipeline owns the worker
// threads and the ordering rule: pictures are BUILT concurrently and LAUNCHED in submission order (a picture's kernels read the
// DPB slots its references were launched into).  Host-only code: no kernel lives here.
#include "../../include/de265_hip.h"

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
};

}  // namespace

struct de265hip_pipeline {
  de265hip_decoder* dec = nullptr;
  int n_workers = 1;
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<PipeJob> q;                        // submitted, not yet taken by a worker
  std::map<uint64_t, int> slot_of;              // launched, copy-out possibly still in flight: ticket -> slot
  std::map<uint64_t, int> failed;               // ticket -> error of prepare / build / run
  struct Built { PipeJob job; de265hip_picture* pic; int rc; };
  std::map<uint64_t, Built> ready;              // built (or failed), waiting for their turn to be launched
  bool launching = false;                       // a worker is launching the ready pictures, in order
  uint64_t next_ticket = 0, next_launch = 0;    // tickets are handed out and launched in submission order
  int in_flight = 0;                            // queued or being built, not yet launched
  int window = 4;                               // bound of in_flight (submit blocks)
  bool stop = false;
  // DE265HIP_PIPE_TIMING=1: where the workers' time goes (seconds, summed over workers; printed when the pipeline is freed)
  bool timing = false;
  double t_idle = 0, t_build = 0, t_turn = 0, t_launch = 0, t_free = 0; long n_jobs = 0;
};

namespace {

void worker(de265hip_pipeline* p)
{
  for (;;) {
    PipeJob j;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv.wait(lk, [&] { return p->stop || !p->q.empty(); });
      if (p->q.empty()) return;
      j = p->q.front(); p->q.pop_front();
    }
    const double t1 = now();
    // host stage, concurrently with the other workers' pictures
    de265hip_recorder* rec = nullptr;
    de265hip_picture* pic = nullptr;
    int rc = 0;
    if (j.desc) rc = de265hip_picture_build(p->dec, j.slot, j.desc, &pic);
    else {
      rc = j.prepare(j.user, &rec);
      if (!rc && !rec) rc = DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
      if (!rc) rc = de265hip_recorder_submit(p->dec, j.slot, rec, &pic);
      if (rec) de265hip_recorder_free(rec);
    }
    const double t2 = now();
    // Device stage, in submission order (also when the picture failed: the turn must pass on).  The worker leaves its picture
    // with the built ones and goes back to work; whoever finds the next picture to be launched among them - and nobody
    // launching - launches every picture whose turn has come.  (Workers that WAITED for their turn held one picture each:
    // with an expensive picture at the head, every other worker idled behind it after building a single picture.)
    double t_l = 0;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->ready[j.ticket] = de265hip_pipeline::Built{ j, pic, rc };
      if (!p->launching) {
        p->launching = true;
        for (auto it = p->ready.find(p->next_launch); it != p->ready.end(); it = p->ready.find(p->next_launch)) {
          de265hip_pipeline::Built b = it->second;
          p->ready.erase(it);
          lk.unlock();
          const double ta = now();
          int r = b.rc;
          if (!r) r = de265hip_picture_run(p->dec, b.pic, DE265HIP_STAGE_FINAL);
          for (int c = 0; c < 3 && !r; c++)
            if (b.job.plane[c]) r = de265hip_dpb_download_async(p->dec, b.job.slot, c, b.job.plane[c], b.job.stride[c]);
          if (b.pic) de265hip_picture_free(b.pic);          // never waits (de265_hip.h LIFETIME)
          t_l += now() - ta;
          lk.lock();
          if (r) p->failed[b.job.ticket] = r; else p->slot_of[b.job.ticket] = b.job.slot;
          p->next_launch++; p->in_flight--;
          p->cv.notify_all();
        }
        p->launching = false;
      }
    }
    const double t3 = now(), t4 = t3;
    if (p->timing) {
      const double t5 = now();
      std::lock_guard<std::mutex> lk(p->mu);
      p->t_idle += t1 - t0; p->t_build += t2 - t1; p->t_turn += t3 - t2 - t_l; p->t_launch += t_l; p->t_free += t5 - t4; p->n_jobs++;
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
  p->dec = dec; p->n_workers = n_workers; p->timing = getenv("DE265HIP_PIPE_TIMING") != nullptr;
  p->window = 4 * n_workers + 4;
  if (const char* w = getenv("DE265HIP_PIPE_WINDOW")) p->window = std::max(1, atoi(w));
  for (int i = 0; i < n_workers; i++) p->th.emplace_back(worker, p);
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
    j.ticket = p->next_ticket++;
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
  int slot = -1;
  {
    std::unique_lock<std::mutex> lk(p->mu);
    if (ticket >= p->next_ticket) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
    p->cv.wait(lk, [&] { return p->next_launch > ticket; });             // launched (or failed) by its worker
    auto f = p->failed.find(ticket);
    if (f != p->failed.end()) { const int rc = f->second; p->failed.erase(f); return rc; }
    auto s = p->slot_of.find(ticket);
    if (s == p->slot_of.end()) return 0;                                  // waited for before
    slot = s->second;
    p->slot_of.erase(s);
  }
  return de265hip_dpb_wait(p->dec, slot);                                 // outside the lock: the workers go on launching
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
    for (auto& kv : p->slot_of) slots.push_back(kv.second);
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
  p->cv.notify_all();
  for (auto& t : p->th) t.join();
  if (p->timing && p->n_jobs)
    fprintf(stderr, "de265hip pipeline: %ld pictures, %d workers; ms per picture: idle %.2f build %.2f wait-for-turn %.2f launch %.2f free %.2f\n",
            p->n_jobs, p->n_workers, 1e3 * p->t_idle / p->n_jobs, 1e3 * p->t_build / p->n_jobs, 1e3 * p->t_turn / p->n_jobs,
            1e3 * p->t_launch / p->n_jobs, 1e3 * p->t_free / p->n_jobs);
  delete p;
}

}  // extern "C"
