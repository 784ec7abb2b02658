// scan.h -- host-side handle of the device-side intra scan (scan_core.h passes, k_scan.hip kernels).
#pragma once
#include <algorithm>
#include <cstring>
#include "kernels.h"
#include "scan_core.h"

namespace d265 {

// where the scan's buffers lie inside a picture's arena (device-only memory behind the uploaded part)
struct ScanLayout {
  size_t clear_begin = 0, clear_end = 0;             // cleared at every build: CTB counters, cell maps, counts, run sizes
  size_t o_ctb = 0, o_cell[3] = { 0, 0, 0 }, o_counts = 0, o_run_ntus = 0;
  size_t o_tu_avail = 0, o_tu_need = 0, o_tu_info = 0, o_tu_run = 0, o_run_rs = 0, o_run_nall = 0, o_run_level = 0, o_run_list = 0, o_pub_flag = 0;
  size_t o_rdy_tab = 0, o_lvl_cnt = 0, o_l0 = 0, o_l0x = 0, o_runs = 0, o_run_tus = 0, o_deps = 0, o_slots = 0, o_front = 0, o_mbx = 0, o_mb_segs = 0;
  uint32_t cap_levels = 0;
  size_t plan(const ScanParams& P, size_t at);        // lays the buffers out from offset `at`; returns the end
  void bind(uint8_t* base, ScanBufs& B) const;        // the pointers of the scratch / output buffers (inputs are the caller's)
};

// up to SCAN_BATCH pictures whose passes share their launches (kernel arguments by value: < 4 KB)
// (2 / 4 / 8 pictures per set of launches: 3 840-3 940 / 4 450-4 530 / 4 510-4 660 pictures/s in the product path of the bench, same
//  box - every launch of the chain costs its place in the queue of a busy device; 8 jobs are 3.7 KB of kernel arguments)
#ifndef SCAN_BATCH
#define SCAN_BATCH 8
#endif
// mo_*: the picture's motion plane from its PU records, by extra workgroups of the per-TU pass's launch (nullptr / 0: not asked
// for).  A launch of its own - 9 us alone - shows as 90 us in the kernel trace of the product path, like every small kernel of the
// chain; one launch fewer per batch, but no measurable change of the product's rate (4 250-4 290 against 4 310-4 370 pictures/s with
// DE265HIP_MOTION_LAUNCH=1, box noise): what those 90 us are is time in the stream's queue, not work.
struct ScanJob { ScanParams P; ScanBufs B; uint32_t cap_resid, cap_levels;
                 const de265hip_pu* mo_pus; const de265hip_slice_params* mo_slices; de265hip_motion* mo_plane; int mo_n_pus, mo_n_slices, mo_w4, mo_h4, mo_pad; };
struct ScanBatch { int n; int pad; int tus_blocks; int pad2; ScanJob job[SCAN_BATCH]; };      // tus_blocks: the per-TU pass's workgroups per picture (the rest of its grid: the motion plane)
hipError_t scan_enqueue_batch(hipStream_t st, const ScanBatch& J);

// what has to be set before the passes (and the picture's kernels) may run, for the pictures of a batch in ONE launch each:
// the cleared region of the arena (k_run's flags and mailboxes, the scan's counters and cell maps), the motion plane set to
// "no reference" and then filled from the PU records.  (Round 4 first issued these per picture - two memsets and a kernel,
// twelve launches ahead of a batch of four scans, each 50-150 us next to the reconstruction kernels of the other streams:
// a third of the time a batch held its stream.)
struct PrepJob {
  uint8_t* zero; unsigned long long zero_bytes;      // multiples of 16 bytes, 16-byte aligned
  uint8_t* ff; unsigned long long ff_bytes;          // the motion plane (nullptr: the host supplied it)
  const de265hip_pu* pus; const de265hip_slice_params* slices; de265hip_motion* motion;
  int n_pus, n_slices, w4, h4;
};
struct PrepBatch { int n; int pad; PrepJob job[SCAN_BATCH]; };
hipError_t prep_enqueue_batch(hipStream_t st, const PrepBatch& J);
hipError_t scan_enqueue(hipStream_t st, const ScanParams& P, const ScanBufs& B, const ScanLayout& L, uint8_t* base, uint32_t cap_resid);
void scan_host_run(const ScanParams& P, const ScanBufs& B, const ScanLayout& L, uint8_t* base, uint32_t cap_resid);

}  // namespace d265
