// pipeline.hip -- SURVEY.md 8(f3): picture-level pipelining behind the C ABI (include/de265_hip.h, "pipeline").
//
// The reference overlaps parsing and reconstruction with a thread pool inside one picture (decctx.cc:976-1178: WPP rows / tiles /
// slices as tasks).  With the reconstruction on the device, what has to overlap is: the host parser on picture n+1, the host stage
// of this library (recorder -> de265hip_picture_build: availability replay, run construction, staging, upload) for pictures
// n, n-1, .. on worker threads, and the kernels + copy-out of the pictures before those on the device.  A pipeline owns the worker
// threads and the ordering rule: pictures are BUILT concurrently and LAUNCHED in submission order (a picture's kernels read the
// DPB slots its references were launched into).  Host-only code: no kernel lives here.
#include "../../include/de265_hip.h"

#include <condition_variable>
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
  uint64_t next_ticket = 0, next_launch = 0;    // tickets are handed out and launched in submission order
  int in_flight = 0;                            // queued or being built, not yet launched
  bool stop = false;
};

namespace {

void worker(de265hip_pipeline* p)
{
  for (;;) {
    PipeJob j;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv.wait(lk, [&] { return p->stop || !p->q.empty(); });
      if (p->q.empty()) return;
      j = p->q.front(); p->q.pop_front();
    }
    // host stage, concurrently with the other workers' pictures
    de265hip_recorder* rec = nullptr;
    de265hip_picture* pic = nullptr;
    int rc = j.prepare(j.user, &rec);
    if (!rc && !rec) rc = DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
    if (!rc) rc = de265hip_recorder_submit(p->dec, j.slot, rec, &pic);
    if (rec) de265hip_recorder_free(rec);
    // device stage, in submission order (also when the picture failed: the turn must pass on)
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv.wait(lk, [&] { return p->next_launch == j.ticket; });
    }
    if (!rc) rc = de265hip_picture_run(p->dec, pic, DE265HIP_STAGE_FINAL);
    for (int c = 0; c < 3 && !rc; c++)
      if (j.plane[c]) rc = de265hip_dpb_download_async(p->dec, j.slot, c, j.plane[c], j.stride[c]);
    {
      std::lock_guard<std::mutex> lk(p->mu);
      if (rc) p->failed[j.ticket] = rc; else p->slot_of[j.ticket] = j.slot;
      p->next_launch++; p->in_flight--;
    }
    p->cv.notify_all();
    if (pic) de265hip_picture_free(pic);          // never waits (de265_hip.h LIFETIME)
  }
}

}  // namespace

extern "C" {

int de265hip_pipeline_new(de265hip_pipeline** out, de265hip_decoder* dec, int n_workers)
{
  if (!out || !dec || n_workers < 1 || n_workers > 16) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  de265hip_pipeline* p = new (std::nothrow) de265hip_pipeline();
  if (!p) return DE265HIP_ERROR_OUT_OF_MEMORY;
  p->dec = dec; p->n_workers = n_workers;
  for (int i = 0; i < n_workers; i++) p->th.emplace_back(worker, p);
  *out = p;
  return 0;
}

int de265hip_pipeline_submit(de265hip_pipeline* p, int dst_slot, de265hip_prepare_fn prepare, void* user,
                             void* const planes[3], const ptrdiff_t stride_bytes[3], uint64_t* ticket)
{
  if (!p || !prepare || dst_slot < 0 || dst_slot >= DE265HIP_MAX_DPB_SLOTS) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  PipeJob j;
  j.slot = dst_slot; j.prepare = prepare; j.user = user;
  for (int c = 0; c < 3; c++) { j.plane[c] = planes ? planes[c] : nullptr; j.stride[c] = (planes && stride_bytes) ? stride_bytes[c] : 0; }
  {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv.wait(lk, [&] { return p->in_flight < p->n_workers + 2; });     // bounded: a few pictures between parser and device
    j.ticket = p->next_ticket++;
    p->in_flight++;
    p->q.push_back(j);
  }
  p->cv.notify_all();
  if (ticket) *ticket = j.ticket;
  return 0;
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
  delete p;
}

}  // extern "C"
