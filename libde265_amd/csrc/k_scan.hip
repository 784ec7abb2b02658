// k_scan.hip -- the passes of scan_core.h as gfx950 kernels (one thread per TU record / CTB / run, behind the upload of the raw
// records on the decoder's copy stream) and, from the same functions compiled for the host, the CPU rehearsal the equivalence
// tests run without a GPU.  Integer / byte work on a few megabytes of records: no LDS tiling to speak of, no MFMA; the passes
// are latency chains of a lone thread per unit, and there are thousands of units.
#include "scan.h"

namespace d265 {

// ------------------------------------------------------------------------------------------------ device kernels
__global__ __launch_bounds__(256)
void k_scan_tus(ScanParams P, ScanBufs B)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  ScanTuSums S = { 0, 0, 0, 0, 0 };
  if (i < P.n_tus) scan_tu(P, B, i, S);
  // one atomic per wavefront and sum
  unsigned long long v[3] = { S.alg_resid, S.alg_intra, S.n_isamp };
  uint32_t w[2] = { S.n_tasks, S.n_intra };
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) v[k] += __shfl_down(v[k], off, 64);
#pragma unroll
    for (int k = 0; k < 2; k++) w[k] += __shfl_down(w[k], off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    scan_add64(&B.counts->alg_resid, v[0]); scan_add64(&B.counts->alg_intra, v[1]); scan_add64(&B.counts->n_isamp, v[2]);
    if (w[0]) atomicAdd(&B.counts->n_tasks, w[0]);
  }
}

// exclusive prefix over the CTBs in tile-scan (decode) order of the seven per-CTB counts, by one workgroup: every thread sums
// a contiguous chunk of CTBs, the workgroup scans the 1024 chunk sums in LDS, every thread writes its chunk's bases
__global__ __launch_bounds__(1024)
void k_scan_prefix(ScanParams P, ScanBufs B, uint32_t cap_resid)
{
  __shared__ uint32_t sums[7][1024];
  __shared__ uint32_t tot[7];
  const int tid = threadIdx.x, n = P.n_ctbs, chunk = (n + 1023) / 1024;
  const int t0 = tid * chunk, t1 = min(n, t0 + chunk);
  uint32_t acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
  for (int t = t0; t < t1; t++) {
    const ScanCtb& C = B.ctb[B.ts2rs[t]];
    for (int k = 0; k < 4; k++) acc[k] += C.n_inter[k] + C.n_ro[k];
    acc[4] += C.n_rext_inter + C.n_rext_ro; acc[5] += C.n_intra; acc[6] += C.n_isamp;
  }
  for (int k = 0; k < 7; k++) sums[k][tid] = acc[k];
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {                 // inclusive Hillis-Steele scan of the chunk sums
    uint32_t v[7];
    for (int k = 0; k < 7; k++) v[k] = tid >= off ? sums[k][tid - off] : 0u;
    __syncthreads();
    for (int k = 0; k < 7; k++) sums[k][tid] += v[k];
    __syncthreads();
  }
  if (tid == 1023) for (int k = 0; k < 7; k++) tot[k] = sums[k][1023];
  uint32_t base[7];
  for (int k = 0; k < 7; k++) base[k] = sums[k][tid] - acc[k];
  for (int t = t0; t < t1; t++) {
    ScanCtb& C = B.ctb[B.ts2rs[t]];
    for (int k = 0; k < 4; k++) { C.l0_base[k] = base[k]; base[k] += C.n_inter[k] + C.n_ro[k]; }
    C.rext_base = base[4]; base[4] += C.n_rext_inter + C.n_rext_ro;
    C.intra_base = base[5]; base[5] += C.n_intra;
    C.isamp_base = base[6]; base[6] += C.n_isamp;
  }
  __syncthreads();
  if (tid == 0) {
    scan_prefix_finish_totals(B, tot);
    B.counts->victim = 0xFFFFFFFFu;
    // (overlapping intra TUs - a malformed description - could ask for more residual samples than the picture has)
    if (tot[6] > cap_resid || tot[5] > P.cap_runs) scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE);
  }
}

__global__ __launch_bounds__(64)
void k_scan_ctbs(ScanParams P, ScanBufs B)
{
  const int rs = blockIdx.x * 64 + threadIdx.x;
  if (rs < P.n_ctbs) scan_ctb(P, B, rs);
}

template <int PASS>
__global__ __launch_bounds__(64)
void k_scan_runs(ScanParams P, ScanBufs B)
{
  const uint32_t s = blockIdx.x * 64 + threadIdx.x;
  if (s >= B.counts->n_intra) return;                        // (sparse run ids lie below the number of intra TUs)
  if (PASS == 1) scan_run(P, B, s);
  else if (PASS == 2) scan_run2(P, B, s);
  else scan_run3(P, B, s);
}

// run levels (longest producer chain), ticket slots in level order - one workgroup
__global__ __launch_bounds__(1024)
void k_scan_order(ScanParams P, ScanBufs B, uint32_t cap_levels)
{
  __shared__ int s_changed;
  __shared__ uint32_t s_max;
  const int tid = threadIdx.x;
  ScanCounts& K = *B.counts;
  if (K.status) return;
  const uint32_t n = K.n_listed;
  for (uint32_t q = tid; q < n; q += 1024) B.run_level[B.run_list[q]] = 1;
  if (tid == 0) s_max = 1;
  __syncthreads();
  for (;;) {                                                  // monotone relaxation: one more level is final after every round
    if (tid == 0) s_changed = 0;
    __syncthreads();
    bool ch = false;
    for (uint32_t q = tid; q < n; q += 1024) {
      const uint32_t s = B.run_list[q];
      const uint32_t na = B.run_nall[s] & 0x7FFFFFFFu;
      const uint32_t* dl = B.deps + B.runs[s].dep_offset;
      uint32_t l = 1;
      for (uint32_t d = 0; d < na; d++) {
        const uint32_t pl = __hip_atomic_load(&B.run_level[dl[d]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
        l = pl > l ? pl : l;
      }
      if (l != B.run_level[s]) { __hip_atomic_store(&B.run_level[s], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); ch = true; }
    }
    if (ch) s_changed = 1;
    __syncthreads();
    const int again = s_changed;
    __syncthreads();
    if (!again) break;
  }
  uint32_t mx = 1;
  for (uint32_t q = tid; q < n; q += 1024) mx = max(mx, B.run_level[B.run_list[q]]);
  atomicMax(&s_max, mx);
  __syncthreads();
  const uint32_t max_rl = n ? s_max : 0;
  if (max_rl + 2 > cap_levels) { if (tid == 0) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
  // per level: micro runs, ordinary runs (front runs and the fault-injection victim take no ticket)
  uint32_t* nm = B.lvl_cnt; uint32_t* no = nm + cap_levels; uint32_t* cm = no + cap_levels; uint32_t* co = cm + cap_levels; uint32_t* tb = co + cap_levels;
  for (uint32_t l = tid; l < max_rl + 2; l += 1024) { nm[l] = no[l] = cm[l] = co[l] = 0; }
  __syncthreads();
  for (uint32_t q = tid; q < n; q += 1024) {
    const uint32_t s = B.run_list[q];
    const uint32_t mic = B.runs[s].micro;
    if ((mic & RUN_MICRO_FRONT) || s == K.victim) continue;
    atomicAdd((mic & 1) ? &nm[B.run_level[s]] : &no[B.run_level[s]], 1u);
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t at = 0, widest = 0;
    for (uint32_t l = 0; l < max_rl + 2; l++) { tb[l] = at; at += (nm[l] + RUN_TICKET_SLOTS - 1) / RUN_TICKET_SLOTS + no[l]; widest = max(widest, nm[l] + no[l]); }
    K.n_batches = at; K.widest = widest; K.max_rl = max_rl;
    if ((unsigned long long)at * RUN_TICKET_SLOTS > P.cap_slots) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED);
  }
  __syncthreads();
  if (K.status) return;
  const uint32_t n_slots = K.n_batches * RUN_TICKET_SLOTS;
  for (uint32_t q = tid; q < n_slots; q += 1024) B.slots[q] = 0xFFFFFFFFu;
  __syncthreads();
  for (uint32_t q = tid; q < n; q += 1024) {
    const uint32_t s = B.run_list[q];
    const uint32_t mic = B.runs[s].micro, l = B.run_level[s];
    if ((mic & RUN_MICRO_FRONT) || s == K.victim) continue;
    if (mic & 1) B.slots[tb[l] * RUN_TICKET_SLOTS + atomicAdd(&cm[l], 1u)] = s | 0x80000000u;
    else B.slots[(tb[l] + (nm[l] + RUN_TICKET_SLOTS - 1) / RUN_TICKET_SLOTS + atomicAdd(&co[l], 1u)) * RUN_TICKET_SLOTS] = s;
  }
}

// ------------------------------------------------------------------------------------------------ layout
size_t ScanLayout::plan(const ScanParams& P, size_t at)
{
  auto add = [&](size_t bytes) { size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return o; };
  const size_t nt = (size_t)P.n_tus, nr = P.cap_runs;
  // cleared at every build: [ctb | cells | counts | run_ntus]
  clear_begin = at;
  o_ctb = add((size_t)P.n_ctbs * sizeof(ScanCtb));
  for (int c = 0; c < 3; c++) o_cell[c] = add((size_t)P.map_w[c] * P.map_h[c] * sizeof(ScanCell));
  o_counts = add(sizeof(ScanCounts));
  o_run_ntus = add(nr);
  clear_end = at;
  o_tu_avail = add(nt * 8); o_tu_need = add(nt * 8); o_tu_info = add(nt * 4);
  o_run_rs = add(nr * 4); o_run_nall = add(nr * 4); o_run_level = add(nr * 4); o_run_list = add(nr * 4); o_pub_flag = add(nr);
  o_rdy_tab = add((size_t)P.cap_mb * 64);
  cap_levels = (uint32_t)nr + 2;
  o_lvl_cnt = add((size_t)cap_levels * 5 * 4);
  o_l0 = add(nt * sizeof(TuTask)); o_l0x = add(nt * sizeof(TuTask));
  o_runs = add(nr * sizeof(RunTask)); o_run_tus = add(nt * sizeof(TuTask));
  o_deps = add((size_t)P.cap_deps * 4); o_slots = add((size_t)P.cap_slots * 4); o_front = add(nr * 4);
  o_mbx = add(nr * 12); o_mb_segs = add((size_t)P.cap_segs * 4);
  return at;
}

void ScanLayout::bind(uint8_t* base, ScanBufs& B) const
{
  B.ctb = (ScanCtb*)(base + o_ctb);
  for (int c = 0; c < 3; c++) B.cell[c] = (ScanCell*)(base + o_cell[c]);
  B.counts = (ScanCounts*)(base + o_counts);
  B.run_ntus = base + o_run_ntus;
  B.tu_avail = (uint64_t*)(base + o_tu_avail); B.tu_need = (uint64_t*)(base + o_tu_need); B.tu_info = (uint32_t*)(base + o_tu_info);
  B.run_rs = (uint32_t*)(base + o_run_rs); B.run_nall = (uint32_t*)(base + o_run_nall); B.run_level = (uint32_t*)(base + o_run_level);
  B.run_list = (uint32_t*)(base + o_run_list); B.pub_flag = base + o_pub_flag; B.rdy_tab = base + o_rdy_tab;
  B.lvl_cnt = (uint32_t*)(base + o_lvl_cnt);
  B.l0 = (TuTask*)(base + o_l0); B.l0x = (TuTask*)(base + o_l0x); B.runs = (RunTask*)(base + o_runs); B.run_tus = (TuTask*)(base + o_run_tus);
  B.deps = (uint32_t*)(base + o_deps); B.slots = (uint32_t*)(base + o_slots); B.front_idx = (uint32_t*)(base + o_front);
  B.mbx = (uint32_t*)(base + o_mbx); B.mb_segs = (uint32_t*)(base + o_mb_segs);
}

// ------------------------------------------------------------------------------------------------ enqueue
hipError_t scan_enqueue(hipStream_t st, const ScanParams& P, const ScanBufs& B, const ScanLayout& L, uint8_t* base, uint32_t cap_resid)
{
  hipError_t e = hipMemsetAsync(base + L.clear_begin, 0, L.clear_end - L.clear_begin, st);
  if (e != hipSuccess) return e;
  if (P.n_tus > 0) hipLaunchKernelGGL(k_scan_tus, dim3((P.n_tus + 255) / 256), dim3(256), 0, st, P, B);
  hipLaunchKernelGGL(k_scan_prefix, dim3(1), dim3(1024), 0, st, P, B, cap_resid);
  if (P.n_tus > 0) {
    hipLaunchKernelGGL(k_scan_ctbs, dim3((P.n_ctbs + 63) / 64), dim3(64), 0, st, P, B);
    const unsigned g = (unsigned)((P.cap_runs + 63) / 64);
    hipLaunchKernelGGL(k_scan_runs<1>, dim3(g), dim3(64), 0, st, P, B);
    hipLaunchKernelGGL(k_scan_runs<2>, dim3(g), dim3(64), 0, st, P, B);
    if (P.flags & SCANF_MAILBOX) hipLaunchKernelGGL(k_scan_runs<3>, dim3(g), dim3(64), 0, st, P, B);
    hipLaunchKernelGGL(k_scan_order, dim3(1), dim3(1024), 0, st, P, B, L.cap_levels);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ the CPU rehearsal
// The same passes as plain loops on host memory laid out like the arena (tests/test_scan_equivalence.py: against the round-3
// host scan on pictures drawn at random; de265hip_debug_build_host_only with DE265HIP_SCAN=core).  Not a product path: a
// decoder without a GPU does not exist.
void scan_host_run(const ScanParams& P, const ScanBufs& B, const ScanLayout& L, uint8_t* base, uint32_t cap_resid)
{
  memset(base + L.clear_begin, 0, L.clear_end - L.clear_begin);
  ScanCounts& K = *B.counts;
  ScanTuSums S = { 0, 0, 0, 0, 0 };
  for (int i = 0; i < P.n_tus; i++) scan_tu(P, B, i, S);
  K.alg_resid = S.alg_resid; K.alg_intra = S.alg_intra; K.n_isamp = S.n_isamp; K.n_tasks = S.n_tasks;
  {
    uint32_t base7[7] = { 0, 0, 0, 0, 0, 0, 0 };
    for (int t = 0; t < P.n_ctbs; t++) {
      ScanCtb& C = B.ctb[B.ts2rs[t]];
      for (int k = 0; k < 4; k++) { C.l0_base[k] = base7[k]; base7[k] += C.n_inter[k] + C.n_ro[k]; }
      C.rext_base = base7[4]; base7[4] += C.n_rext_inter + C.n_rext_ro;
      C.intra_base = base7[5]; base7[5] += C.n_intra;
      C.isamp_base = base7[6]; base7[6] += C.n_isamp;
    }
    scan_prefix_finish_totals(B, base7);
    K.victim = 0xFFFFFFFFu;
    if (base7[6] > cap_resid || base7[5] > P.cap_runs) scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE);
  }
  if (P.n_tus == 0) return;
  for (int rs = 0; rs < P.n_ctbs; rs++) scan_ctb(P, B, rs);
  for (uint32_t s = 0; s < K.n_intra; s++) scan_run(P, B, s);
  for (uint32_t s = 0; s < K.n_intra; s++) scan_run2(P, B, s);
  if (P.flags & SCANF_MAILBOX) for (uint32_t s = 0; s < K.n_intra; s++) scan_run3(P, B, s);
  if (K.status) return;
  // scan_order, serially
  const uint32_t n = K.n_listed;
  for (uint32_t q = 0; q < n; q++) B.run_level[B.run_list[q]] = 1;
  for (bool again = true; again;) {
    again = false;
    for (uint32_t q = 0; q < n; q++) {
      const uint32_t s = B.run_list[q], na = B.run_nall[s] & 0x7FFFFFFFu;
      const uint32_t* dl = B.deps + B.runs[s].dep_offset;
      uint32_t l = 1;
      for (uint32_t d = 0; d < na; d++) l = std::max(l, B.run_level[dl[d]] + 1);
      if (l != B.run_level[s]) { B.run_level[s] = l; again = true; }
    }
  }
  uint32_t max_rl = 0;
  for (uint32_t q = 0; q < n; q++) max_rl = std::max(max_rl, B.run_level[B.run_list[q]]);
  const uint32_t cap_levels = L.cap_levels;
  if (max_rl + 2 > cap_levels) { scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
  uint32_t* nm = B.lvl_cnt; uint32_t* no = nm + cap_levels; uint32_t* cm = no + cap_levels; uint32_t* co = cm + cap_levels; uint32_t* tb = co + cap_levels;
  for (uint32_t l = 0; l < max_rl + 2; l++) nm[l] = no[l] = cm[l] = co[l] = 0;
  for (uint32_t q = 0; q < n; q++) {
    const uint32_t s = B.run_list[q], mic = B.runs[s].micro;
    if ((mic & RUN_MICRO_FRONT) || s == K.victim) continue;
    ((mic & 1) ? nm : no)[B.run_level[s]]++;
  }
  uint32_t at = 0, widest = 0;
  for (uint32_t l = 0; l < max_rl + 2; l++) { tb[l] = at; at += (nm[l] + RUN_TICKET_SLOTS - 1) / RUN_TICKET_SLOTS + no[l]; widest = std::max(widest, nm[l] + no[l]); }
  K.n_batches = at; K.widest = widest; K.max_rl = max_rl;
  if ((unsigned long long)at * RUN_TICKET_SLOTS > P.cap_slots) { scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
  for (uint32_t q = 0; q < at * RUN_TICKET_SLOTS; q++) B.slots[q] = 0xFFFFFFFFu;
  for (uint32_t q = 0; q < n; q++) {
    const uint32_t s = B.run_list[q], mic = B.runs[s].micro, l = B.run_level[s];
    if ((mic & RUN_MICRO_FRONT) || s == K.victim) continue;
    if (mic & 1) B.slots[tb[l] * RUN_TICKET_SLOTS + cm[l]++] = s | 0x80000000u;
    else B.slots[(tb[l] + (nm[l] + RUN_TICKET_SLOTS - 1) / RUN_TICKET_SLOTS + co[l]++) * RUN_TICKET_SLOTS] = s;
  }
}

}  // namespace d265
