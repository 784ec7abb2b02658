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

// ---- wave-level helpers (wave64)
__device__ __forceinline__ int wave_max_i(int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ int wave_min_i(int v) { for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ uint32_t wave_sum_u(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ uint64_t lanes_below(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }

// The CTB pass, one WAVEFRONT per CTB (scan_core.h scan_ctb is the same pass as one thread's loop: the CPU rehearsal; the
// equivalence test holds this kernel to it).  What is sequential - which run a TU joins depends on the TUs before it - runs
// on a window of the CTB's cells in LDS (its own 16 x 16 cells, the row above with its above-right reach, the column to the
// left): per intra TU one LDS round trip (a lane per neighbour unit), a handful of cross-lane reductions, the decision, one
// LDS write (a lane per covered cell).  The records are fetched 64 at a time, classified a lane each; the level-0 tasks of
// the inter TUs are written by their lanes (positions by ballot prefix).
#define SCW_W 25                         // window columns: cell x in [-1, 23] relative to the CTB
#define SCW_H 17                         // window rows:    cell y in [-1, 15]
#define SCW_NONLOCAL 0x40000000u         // an intra TU of another CTB covers the cell
__global__ __launch_bounds__(64)
void k_scan_ctbs(ScanParams P, ScanBufs B)
{
  __shared__ uint32_t win[3][SCW_W * SCW_H];
  __shared__ uint8_t s_ntus[768];
  const int rs = blockIdx.x, lane = threadIdx.x;
  if (B.counts->status) return;
  ScanCtb& C = B.ctb[rs];
  const uint32_t first = C.first_tu, end = C.end_tu, seen = C.seen, n_intra = C.n_intra, ibase = C.intra_base;
  if (seen == 0) { if (lane == 0) C.n_runs = 0; return; }
  if (seen != 1 || end <= first || end > (uint32_t)P.n_tus || n_intra > 768) {
    if (lane == 0) { C.n_runs = 0; scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); }
    return;
  }
  const int cx0 = (rs % P.ctbs_w) << P.lc, cy0 = (rs / P.ctbs_w) << P.lc;      // luma origin of the CTB
  // ---- the window: own cells empty, the halo from the cell map of the per-TU pass
  for (int q = lane; q < 3 * SCW_W * SCW_H; q += 64) (&win[0][0])[q] = 0;
  __syncthreads();
  for (int c = 0; c < (P.cf ? 3 : 1); c++) {
    const int sw = c ? P.subw : 1, sh = c ? P.subh : 1, mw = P.map_w[c], mh = P.map_h[c];
    const int ox4 = (cx0 / sw) >> 2, oy4 = (cy0 / sh) >> 2;
    if (lane < SCW_W + SCW_H - 1) {
      const int wx = lane < SCW_W ? lane - 1 : -1, wy = lane < SCW_W ? -1 : lane - SCW_W;
      const int gx = ox4 + wx, gy = oy4 + wy;
      if (gx >= 0 && gy >= 0 && gx < mw && gy < mh && (uint32_t)B.cell[c][gx + (size_t)gy * mw] != 0)
        win[c][(wy + 1) * SCW_W + (wx + 1)] = SCW_NONLOCAL;
    }
  }
  for (int q = lane; q < 768; q += 64) s_ntus[q] = 0;
  __syncthreads();
  uint32_t cls_start[4];
  for (int k = 0; k < 4; k++) cls_start[k] = scan_l0_class_start(B.counts->n_l0_size, k);
  uint32_t inter_at[4] = { 0, 0, 0, 0 }, rext_at = 0;
  int cur_run[3] = { -1, -1, -1 };
  int n_local = 0;
  for (uint32_t base = first; base < end; base += 64) {
    const uint32_t i = base + lane;
    const bool have = i < end;
    de265hip_tu tu; memset(&tu, 0, sizeof(tu));
    uint64_t av = 0, nd = 0;
    int cls = 0, rx = 0;
    if (have) {
      tu = B.tus[i];
      cls = scan_tu_class(P, B, tu, &rx);
      if (cls == 3) { av = B.tu_avail[i]; nd = B.tu_need[i]; }
    }
    // -- level-0 tasks of the inter TUs: a lane each, positions by ballot prefix per size class
    for (int k = 0; k < 4; k++) {
      const uint64_t m = __ballot(cls == 1 && tu.log2_size == k + 2);
      if (cls == 1 && tu.log2_size == k + 2) B.l0[cls_start[k] + C.l0_base[k] + inter_at[k] + __popcll(m & lanes_below(lane))] = scan_task_of(tu);
      inter_at[k] += __popcll(m);
    }
    {
      const uint64_t m = __ballot(cls == 2);
      if (cls == 2) {
        TuTask t = scan_task_of(tu);
        uint64_t luma_info = 0; int rx_luma = 0;
        if (rx & D265_RX_XCC) scan_xcc_luma(P, B, (int)i, &luma_info, &rx_luma);
        t.pad3 = (uint8_t)(rx | rx_luma); t.angle = tu.res_scale_val; t.avail = luma_info;
        B.l0x[C.rext_base + rext_at + __popcll(m & lanes_below(lane))] = t;
      }
      rext_at += __popcll(m);
    }
    // -- the intra TUs of the chunk, one after the other
    const uint32_t pos = (uint32_t)tu.x0 | ((uint32_t)tu.y0 << 16), shape = (uint32_t)tu.log2_size | ((uint32_t)tu.c_idx << 8);
    for (uint64_t im = __ballot(cls == 3); im; im &= im - 1) {
      const int src = __builtin_ctzll(im);
      const uint32_t pos_u = __shfl(pos, src, 64), shape_u = __shfl(shape, src, 64);
      const uint64_t mask = ((uint64_t)(uint32_t)__shfl((int)(av >> 32), src, 64) << 32) | (uint32_t)__shfl((int)av, src, 64);
      const uint64_t need0 = ((uint64_t)(uint32_t)__shfl((int)(nd >> 32), src, 64) << 32) | (uint32_t)__shfl((int)nd, src, 64);
      const int xB = pos_u & 0xFFFF, yB = pos_u >> 16, log2 = shape_u & 0xFF, c = shape_u >> 8, nT = 1 << log2, corner = nT >> 1;
      const int sw = c ? P.subw : 1, sh = c ? P.subh : 1;
      const int ox4 = (cx0 / sw) >> 2, oy4 = (cy0 / sh) >> 2;
      const int cw = c ? P.cwid : P.width, ch = c ? P.chei : P.height;
      const int wx0 = (xB >> 2) - ox4, wy0 = (yB >> 2) - oy4;                 // the TU's first cell in window coordinates
      uint32_t* W = win[c];
      auto look = [&](int wx, int wy) -> uint32_t {
        return (wx >= -1 && wx < SCW_W - 1 && wy >= -1 && wy < SCW_H - 1) ? W[(wy + 1) * SCW_W + (wx + 1)] : 0u;
      };
      // a lane per neighbour unit: left column bottom -> top, corner, top row left -> right
      const bool in_mask = (mask >> lane) & 1, in_need = (need0 >> lane) & 1;
      uint32_t v = 0;
      if (in_mask) {
        const int wx = lane < corner ? wx0 - 1 : (lane == corner ? wx0 - 1 : wx0 + (lane - corner - 1));
        const int wy = lane < corner ? wy0 + corner - 1 - lane : wy0 - 1;
        v = look(wx, wy);
      }
      const bool local = v >> 31, nonlocal = v == SCW_NONLOCAL;
      const int vrun = (int)(v & 0xFFFF), vlev = (int)((v >> 16) & 0xFF);
      const int crun = cur_run[c];
      const bool foreign = __ballot(in_need && !local && !nonlocal) != 0;
      const bool any_nonlocal = __ballot(in_need && nonlocal) != 0;
      const uint64_t loc_m = __ballot(in_need && local);
      const bool reads_cur = __ballot(in_need && local && vrun == crun) != 0;
      int llev = wave_max_i((in_need && local && vrun == crun) ? vlev : 0) + 1;
      const int p0 = loc_m ? __shfl(vrun, __builtin_ctzll(loc_m), 64) : -1;
      const bool multi = __ballot(in_need && local && vrun != p0) != 0;
      int r = crun;
      bool extends = r >= 0 && s_ntus[r] < 255;
      if (extends && !reads_cur) extends = __ballot(in_mask && !in_need && local && vrun == r) != 0;
      bool merged = false;
      if (!extends && (P.flags & SCANF_MERGE) && p0 >= 0 && !multi && !any_nonlocal && s_ntus[p0] < 255) {
        // in-run level: behind everything of that run in the row above and the column to the left of the TU's neighbourhood
        const int ux0 = (xB - 4 > 0 ? xB - 4 : 0) >> 2, uy0 = (yB - 4 > 0 ? yB - 4 : 0) >> 2;
        const int ux1 = (cw - 1 < xB + 2 * nT + 3 ? cw - 1 : xB + 2 * nT + 3) >> 2, uy1 = (ch - 1 < yB + 2 * nT + 3 ? ch - 1 : yB + 2 * nT + 3) >> 2;
        int lv = 0;
        if (lane < 32) { const int x4 = ux0 + lane; if (uy0 < (yB >> 2) && x4 <= ux1) { const uint32_t q = look(x4 - ox4, uy0 - oy4); if ((q >> 31) && (int)(q & 0xFFFF) == p0) lv = (int)((q >> 16) & 0xFF); } }
        else { const int y4 = uy0 + lane - 32; if (ux0 < (xB >> 2) && y4 <= uy1) { const uint32_t q = look(ux0 - ox4, y4 - oy4); if ((q >> 31) && (int)(q & 0xFFFF) == p0) lv = (int)((q >> 16) & 0xFF); } }
        const int lx = wave_max_i(lv);
        if (lx + 1 <= 250) { r = p0; llev = lx + 1; merged = true; }
      }
      if (!extends && !merged) { r = n_local++; cur_run[c] = r; llev = 1; }
      __syncthreads();                                   // (every lane has read s_ntus and the window)
      if (lane == 0) {
        s_ntus[r]++;
        B.tu_info[base + src] = (uint32_t)r | ((uint32_t)llev << 16) | (foreign ? SCAN_TI_FOREIGN : 0u) | SCAN_TI_INTRA;
      }
      const int n4 = nT >> 2;
      if (lane < n4 * n4) W[(wy0 + lane / n4 + 1) * SCW_W + (wx0 + lane % n4 + 1)] = (uint32_t)r | ((uint32_t)llev << 16) | (1u << 31);
      __syncthreads();
    }
  }
  // ---- the CTB's runs: sizes, CTB, a place in the run list
  uint32_t at = 0;
  if (lane == 0) { C.n_runs = (uint32_t)n_local; atomicAdd(&B.counts->n_runs, (uint32_t)n_local); at = atomicAdd(&B.counts->n_listed, (uint32_t)n_local); }
  at = __shfl(at, 0, 64);
  for (int q = lane; q < n_local; q += 64) { B.run_ntus[ibase + q] = s_ntus[q]; B.run_rs[ibase + q] = (uint32_t)rs; B.run_list[at + q] = ibase + (uint32_t)q; }
}

// The run pass, one WAVEFRONT per run (persistent: a fixed grid walks the run list).  scan_core.h scan_run is the same pass as
// one thread's loop.
#define SCR_MAX 256
__global__ __launch_bounds__(64)
void k_scan_runs1(ScanParams P, ScanBufs B)
{
  __shared__ int tix[SCR_MAX];
  __shared__ uint32_t keys[SCR_MAX], sorted[SCR_MAX], s_samp[SCR_MAX + 1];
  __shared__ uint8_t s_lev[SCR_MAX], s_coll[SCR_MAX], s_l2[SCR_MAX], s_rdy[64];
  __shared__ uint32_t s_tab[512];
  __shared__ uint32_t s_nd;
  const int lane = threadIdx.x;
  if (B.counts->status) return;
  const uint32_t n_listed = B.counts->n_listed;
  uint32_t cls_start[4];
  for (int k = 0; k < 4; k++) cls_start[k] = scan_l0_class_start(B.counts->n_l0_size, k);
  for (uint32_t qrun = blockIdx.x; qrun < n_listed; qrun += gridDim.x) {
    const uint32_t s = B.run_list[qrun];
    const int rs = (int)B.run_rs[s];
    const ScanCtb& C = B.ctb[rs];
    const int r = (int)(s - C.intra_base);
    // ---- its TUs (decode order), and what the runs before it in this CTB take of the CTB's lists
    int n = 0;
    uint32_t n_before = 0, samp_before = 0, ro_before[4] = { 0, 0, 0, 0 }, rext_ro_before = 0;
    for (uint32_t base = C.first_tu; base < C.end_tu; base += 64) {
      const uint32_t i = base + lane;
      uint32_t ti = 0;
      if (i < C.end_tu) ti = B.tu_info[i];
      const bool intra = ti & SCAN_TI_INTRA;
      const int rr = (int)SCAN_TI_RUN(ti);
      if (intra && rr < r) {
        const de265hip_tu tu = B.tus[i];
        const int rx = scan_rx_bits(P, B, tu);
        const bool cbf = (tu.flags & DE265HIP_TU_CBF) && tu.n_coeff;
        n_before++; samp_before += 1u << (2 * tu.log2_size);
        if (cbf || (rx & D265_RX_XCC)) { if (rx) rext_ro_before++; else ro_before[tu.log2_size - 2]++; }
      }
      const uint64_t m = __ballot(intra && rr == r);
      if (intra && rr == r) { const int k = n + __popcll(m & lanes_below(lane)); if (k < SCR_MAX) tix[k] = (int)i; }
      n += __popcll(m);
    }
    n_before = wave_sum_u(n_before); samp_before = wave_sum_u(samp_before); rext_ro_before = wave_sum_u(rext_ro_before);
    for (int k = 0; k < 4; k++) ro_before[k] = wave_sum_u(ro_before[k]);
    __syncthreads();
    if (n == 0 || n > 255 || n != (int)B.run_ntus[s]) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); return; }
    // ---- box, window reach, samples, levels: every lane its TUs (k = lane, lane + 64, ..), then across the lanes
    int x0 = 1 << 30, y0 = 1 << 30, x1 = 0, y1 = 0, wx1 = 0, wy1 = 0, nl = 0, c = 0;
    uint32_t own_samples = 0, cand = 0;
    bool foreign = false, big = false, too_big = false;
    for (int k = lane; k < n; k += 64) {
      const de265hip_tu tu = B.tus[tix[k]];
      const uint32_t ti = B.tu_info[tix[k]];
      const int nT = 1 << tu.log2_size;
      c = tu.c_idx;
      x0 = min(x0, (int)tu.x0); y0 = min(y0, (int)tu.y0); x1 = max(x1, tu.x0 + nT); y1 = max(y1, tu.y0 + nT);
      wx1 = max(wx1, tu.x0 + 2 * nT); wy1 = max(wy1, tu.y0 + 2 * nT);
      own_samples += (uint32_t)(nT * nT);
      nl = max(nl, (int)SCAN_TI_LLEV(ti));
      foreign = foreign || (ti & SCAN_TI_FOREIGN);
      big = big || tu.log2_size == 4; too_big = too_big || tu.log2_size > 4;
      s_lev[k] = (uint8_t)SCAN_TI_LLEV(ti); s_l2[k] = tu.log2_size;
      cand += (uint32_t)__popcll(B.tu_need[tix[k]]);
    }
    x0 = wave_min_i(x0); y0 = wave_min_i(y0); x1 = wave_max_i(x1); y1 = wave_max_i(y1); wx1 = wave_max_i(wx1); wy1 = wave_max_i(wy1);
    nl = wave_max_i(nl); c = wave_max_i(c);
    own_samples = wave_sum_u(own_samples);
    foreign = __ballot(foreign) != 0; big = __ballot(big) != 0; too_big = __ballot(too_big) != 0;
    if (nl > 256 || nl - 1 > 255) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
    bool micro = !(P.flags & SCANF_MICRO_OFF) && n <= P.micro_tus && x1 - x0 <= 32 && y1 - y0 <= 32 && !too_big;
    if (micro && big) {
      if (!(P.flags & SCANF_MICRO16) || own_samples > 1024) micro = false;
      else {
        const int ax0 = (x0 - 1) & ~7, wxc = wx1 < x1 + 32 ? wx1 : x1 + 32, wyc = wy1 < y1 + 32 ? wy1 : y1 + 32;
        const int cols = wxc - ax0, rows = wyc - (y0 - 1);
        micro = cols <= 56 && rows <= 41 && ((cols + 7) >> 3) * rows <= 256;
      }
    }
    bool dense = (int)own_samples == (x1 - x0) * (y1 - y0) && !(P.flags & SCANF_NO_DENSE);
    if (dense) {
      bool bad = false;
      for (int k = lane; k < n; k += 64) {
        const de265hip_tu tu = B.tus[tix[k]];
        const uint64_t avail = B.tu_avail[tix[k]];
        const int nT = 1 << tu.log2_size, xB = tu.x0, yB = tu.y0, corner = nT >> 1;
        if (xB > x0 && yB + 2 * nT > y1) {
          int umax = (yB + 2 * nT - 1 - y1) >> 2; if (umax > corner - 1) umax = corner - 1;
          if (avail & ((2ull << umax) - 1ull)) bad = true;
        }
        if (yB > y0 && xB + 2 * nT > x1) {
          int kmin = (x1 - xB) >> 2; if (kmin < 0) kmin = 0;
          if (kmin < corner && ((avail >> (corner + 1 + kmin)) & ((1ull << (corner - kmin)) - 1ull))) bad = true;
        }
      }
      dense = __ballot(bad) == 0;
    }
    RunTask o;
    o.x0 = (uint16_t)x0; o.y0 = (uint16_t)y0; o.x1 = (uint16_t)x1; o.y1 = (uint16_t)y1;
    o.wx1 = (uint16_t)(wx1 < x1 + 32 ? wx1 : x1 + 32); o.wy1 = (uint16_t)(wy1 < y1 + 32 ? wy1 : y1 + 32);
    o.c_idx = (uint8_t)c; o.micro = (uint8_t)((micro ? 1 : 0) | (dense ? 2 : 0)); o.n_tus = (uint16_t)n;
    o.first_tu = C.intra_base + n_before;
    o.res_offset = C.isamp_base + samp_before;
    o.dep_offset = 0; o.n_deps = 0;
    // ---- mailbox of an ordinary dense run; ready epochs of its edge packets
    uint32_t mb_id = 0xFFFFFFFFu;
    if ((P.flags & SCANF_MAILBOX) && !micro && dense) {
      if (lane == 0) mb_id = atomicAdd(&B.counts->n_mailboxes, 1u);
      mb_id = __shfl(mb_id, 0, 64);
      if (mb_id >= P.cap_mb) mb_id = 0xFFFFFFFFu;            // (beyond the mailboxes there are: the run does without)
      else if ((P.flags & SCANF_MB_PHASES) && c == 0 && x1 - x0 <= 64 && y1 - y0 <= 64) {
        s_rdy[lane] = 255;
        __syncthreads();
        for (int k = lane; k < n; k += 64) {
          const de265hip_tu tu = B.tus[tix[k]];
          const int nT = 1 << tu.log2_size;
          const uint8_t ep = (uint8_t)(s_lev[k] - 1);
          if (tu.y0 + nT == y1) for (int q = 0; q < (nT >> 1); q++) s_rdy[((tu.x0 - x0) >> 1) + q] = ep;
          if (tu.x0 + nT == x1) for (int q = 0; q < (nT >> 1); q++) s_rdy[32 + ((tu.y0 - y0) >> 1) + q] = ep;
        }
        __syncthreads();
        B.rdy_tab[64 * (size_t)mb_id + lane] = s_rdy[lane];
      }
    }
    if (lane < 3) B.mbx[3 * (size_t)s + lane] = lane == 0 ? mb_id : 0xFFFFFFFFu;
    if (lane == 0) B.pub_flag[s] = 0;
    // ---- chain order: list (wavefront, or 4 = collective) | in-run level | decode index; the rank of a TU inside its level
    // counts the non-collective TUs of that level before it
    const int nwv = micro ? 1 : P.run_waves;
    for (int k = lane; k < n; k += 64) s_coll[k] = (B.tus[tix[k]].log2_size > 3 && !micro) ? 1 : 0;
    __syncthreads();
    for (int k = lane; k < n; k += 64) {
      int rank = 0;
      for (int q = 0; q < k; q++) rank += (s_lev[q] == s_lev[k] && !s_coll[q]) ? 1 : 0;
      const int list = s_coll[k] ? 4 : rank % nwv;
      keys[k] = ((uint32_t)list << 20) | ((uint32_t)s_lev[k] << 8) | (uint32_t)k;
    }
    __syncthreads();
    int we[4] = { 0, 0, 0, 0 };
    for (int k = lane; k < n; k += 64) {
      int posn = 0;
      for (int q = 0; q < n; q++) posn += keys[q] < keys[k] ? 1 : 0;
      sorted[posn] = keys[k];
      for (int w = 0; w < 4; w++) we[w] += (int)(keys[k] >> 20) <= w ? 1 : 0;
    }
    for (int w = 0; w < 4; w++) o.wave_end[w] = (uint16_t)wave_sum_u((uint32_t)we[w]);
    o.n_lvls = (uint16_t)(nl > 0 ? nl - 1 : 0);
    __syncthreads();
    // ---- sample offsets in chain order (exclusive prefix of the TU sizes)
    if (lane == 0) { uint32_t acc = 0; for (int oi = 0; oi < n; oi++) { s_samp[oi] = acc; acc += 1u << (2 * s_l2[sorted[oi] & 0xFFu]); } s_samp[n] = acc; }
    __syncthreads();
    o.n_samples = s_samp[n];
    // ---- the run-ordered TU records + the residual-only copies (level-0 tasks), a lane per TU, 64 at a time
    uint32_t ro_at[4] = { 0, 0, 0, 0 }, rext_at = 0;
    for (int ob = 0; ob < n; ob += 64) {
      const int oi = ob + lane;
      const bool have = oi < n;
      TuTask tt; memset(&tt, 0, sizeof(tt));
      de265hip_tu tu; memset(&tu, 0, sizeof(tu));
      int trx = 0, i = 0; bool ro_on = false;
      uint32_t coeff_offset = 0;
      if (have) {
        i = tix[sorted[oi] & 0xFFu];
        tu = B.tus[i];
        tt = scan_task_of(tu);
        const int m = tu.intra_mode < 35 ? tu.intra_mode : 1;
        tt.angle = (int8_t)scan_intra_angle(m); tt.inv_angle = (int16_t)scan_inv_angle(m);
        tt.avail = B.tu_avail[i];
        tt.run_level = (uint8_t)(SCAN_TI_LLEV(B.tu_info[i]) - 1);
        coeff_offset = tt.coeff_offset;
        tt.resid_offset = o.res_offset + s_samp[oi];
        tt.coeff_offset = s_samp[oi];
        trx = scan_rx_bits(P, B, tu);
        ro_on = (tt.flags & DE265HIP_TU_CBF) || (trx & D265_RX_XCC);
      }
      for (int k = 0; k < 4; k++) {
        const bool mine = have && ro_on && !trx && tt.log2_size == k + 2;
        const uint64_t m = __ballot(mine);
        if (mine) {
          TuTask ro = tt; ro.flags |= D265_TU_RESID_ONLY; ro.coeff_offset = coeff_offset; ro.run_level = 0;
          B.l0[cls_start[k] + C.l0_base[k] + C.n_inter[k] + ro_before[k] + ro_at[k] + __popcll(m & lanes_below(lane))] = ro;
        }
        ro_at[k] += __popcll(m);
      }
      {
        const bool mine = have && ro_on && trx;
        const uint64_t m = __ballot(mine);
        if (mine) {
          TuTask ro = tt; ro.flags |= D265_TU_RESID_ONLY; ro.coeff_offset = coeff_offset; ro.run_level = 0;
          uint64_t luma_info = 0; int rx_luma = 0;
          if (trx & D265_RX_XCC) scan_xcc_luma(P, B, i, &luma_info, &rx_luma);
          ro.pad3 = (uint8_t)(trx | rx_luma); ro.angle = 0; ro.avail = 0;
          if (trx & D265_RX_XCC) { ro.angle = tu.res_scale_val; ro.avail = luma_info; }
          B.l0x[C.rext_base + C.n_rext_inter + rext_ro_before + rext_at + __popcll(m & lanes_below(lane))] = ro;
        }
        rext_at += __popcll(m);
      }
      if (have) {
        if (ro_on) tt.flags |= DE265HIP_TU_CBF;              // (the run kernels read the residual block whenever there is one)
        B.run_tus[o.first_tu + (uint32_t)oi] = tt;
      }
    }
    // ---- producers: every needed unit of every TU -> the run behind its cell; each run once (a hash set in LDS: the cells a
    // run inside one CTB can need number fewer than its slots)
    for (int q = lane; q < 512; q += 64) s_tab[q] = 0xFFFFFFFFu;
    if (lane == 0) s_nd = 0;
    __syncthreads();
    {
      const int mw = P.map_w[c];
      for (int k = lane; k < n; k += 64) {
        const de265hip_tu tu = B.tus[tix[k]];
        for (uint64_t need = B.tu_need[tix[k]]; need; need &= need - 1) {
          const ScanCell v = B.cell[c][scan_cell_of(__builtin_ctzll(need), tu.x0, tu.y0, 1 << tu.log2_size, mw)];
          if ((uint32_t)v == 0) continue;
          const uint32_t j = (uint32_t)v - 1;
          const uint32_t tj = B.tu_info[j];
          if (!(tj & SCAN_TI_INTRA)) continue;
          const uint32_t ps = B.ctb[scan_tu_ctb(P, B.tus[j])].intra_base + SCAN_TI_RUN(tj);
          if (ps == s) continue;
          uint32_t hsh = (ps * 2654435761u) >> 23;
          for (int probe = 0; probe < 512; probe++, hsh = (hsh + 1) & 511) {
            const uint32_t old = atomicCAS(&s_tab[hsh], 0xFFFFFFFFu, ps);
            if (old == 0xFFFFFFFFu) { atomicAdd(&s_nd, 1u); break; }
            if (old == ps) break;
          }
        }
      }
    }
    __syncthreads();
    const uint32_t nd = s_nd;
    (void)cand;
    if (nd > 500) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
    if (nd) {
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(&B.counts->n_deps_alloc, nd);
      at = __shfl(at, 0, 64);
      if (at + nd > P.cap_deps) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
      o.dep_offset = at;
      uint32_t w = 0;
      for (int qb = 0; qb < 512; qb += 64) {
        const uint32_t e = s_tab[qb + lane];
        const uint64_t m = __ballot(e != 0xFFFFFFFFu);
        if (e != 0xFFFFFFFFu) B.deps[at + w + __popcll(m & lanes_below(lane))] = e;
        w += __popcll(m);
      }
    }
    o.n_deps = (uint16_t)nd;
    const bool front = micro && nd == 0 && !(P.flags & SCANF_FRONT_OFF);
    if (front) o.micro |= RUN_MICRO_FRONT;
    if (lane == 0) {
      B.run_nall[s] = nd | (foreign ? 0x80000000u : 0u);
      atomicAdd(&B.counts->sum_lvls, (uint32_t)nl);
      if (front) B.front_idx[atomicAdd(&B.counts->n_front, 1u)] = s;
      B.runs[s] = o;
    }
    if (front) {
      // algorithmic bytes of the front runs (bench: roofline of k_intra_front)
      unsigned long long alg = 0;
      const unsigned long long bpp = (unsigned long long)(c ? P.bppC : P.bppY);
      for (int k = lane; k < n; k += 64) { const unsigned long long nT = 1ull << B.tus[tix[k]].log2_size; alg += bpp * (4 * nT + 1) + bpp * nT * nT; }
      for (int off = 32; off > 0; off >>= 1) alg += __shfl_xor(alg, off, 64);
      if (lane == 0) scan_add64(&B.counts->alg_intra_front, alg);
    }
    __syncthreads();
  }
}

// the two later run passes: a thread per listed run (scan_core.h scan_run2 / scan_run3)
template <int PASS>
__global__ __launch_bounds__(64)
void k_scan_runs(ScanParams P, ScanBufs B)
{
  const uint32_t q = blockIdx.x * 64 + threadIdx.x;
  if (B.counts->status || q >= B.counts->n_listed) return;
  const uint32_t s = B.run_list[q];
  if (PASS == 2) scan_run2(P, B, s);
  else scan_run3(P, B, s);
}

// run levels (longest producer chain), ticket slots in level order - one workgroup
#define SCO_NMAX 12288                   // runs / producer entries the LDS fast path of the level rounds holds
#define SCO_DMAX 40960
#define SCO_LDS_BYTES ((3 * SCO_NMAX + SCO_DMAX) * 2)
__global__ __launch_bounds__(1024)
void k_scan_order(ScanParams P, ScanBufs B, uint32_t cap_levels)
{
  __shared__ int s_changed;
  __shared__ uint32_t s_max, s_cnt;
  const int tid = threadIdx.x;
  ScanCounts& K = *B.counts;
  if (K.status) return;
  const uint32_t n = K.n_listed;
  if (tid == 0) { s_max = 1; s_cnt = 0; }
  __syncthreads();
  // Levels by monotone relaxation: one more level is final after every round.  Fast path: everything a round touches in LDS -
  // the levels by position in the run list, every run's producers as list positions (a picture's ~6 000 runs with their
  // ~20 000 producer entries: the rounds of an all-intra 4K picture, 126 of them, took 10 us each on the arrays in global
  // memory, dependent L2 round trips; 0.3 us on LDS).  A picture beyond the LDS arrays takes the global-memory rounds.
  extern __shared__ uint16_t dyn[];
  uint16_t* lev = dyn; uint16_t* doff = lev + SCO_NMAX; uint16_t* dna = doff + SCO_NMAX; uint16_t* dpos = dna + SCO_NMAX;
  bool fast = n <= SCO_NMAX;
  if (fast) {
    for (uint32_t q = tid; q < n; q += 1024) B.run_level[B.run_list[q]] = q;        // (for now: a run's position in the list)
    __syncthreads();
    uint32_t tot = 0;
    for (uint32_t q = tid; q < n; q += 1024) tot += B.run_nall[B.run_list[q]] & 0x7FFFFFFFu;
    atomicAdd(&s_cnt, tot);
    __syncthreads();
    fast = s_cnt <= SCO_DMAX;
    __syncthreads();
    if (tid == 0) s_cnt = 0;
    __syncthreads();
  }
  if (fast) {
    for (uint32_t q = tid; q < n; q += 1024) {
      const uint32_t s = B.run_list[q];
      const uint32_t na = B.run_nall[s] & 0x7FFFFFFFu;
      const uint32_t* dl = B.deps + B.runs[s].dep_offset;
      const uint32_t o = atomicAdd(&s_cnt, na);
      lev[q] = 1; doff[q] = (uint16_t)o; dna[q] = (uint16_t)na;
      for (uint32_t d = 0; d < na; d++) dpos[o + d] = (uint16_t)B.run_level[dl[d]];
    }
    __syncthreads();
    for (;;) {
      if (tid == 0) s_changed = 0;
      __syncthreads();
      bool ch = false;
      for (uint32_t q = tid; q < n; q += 1024) {
        const uint32_t o = doff[q], na = dna[q];
        uint32_t l = 1;
        for (uint32_t d = 0; d < na; d++) { const uint32_t pl = (uint32_t)lev[dpos[o + d]] + 1; l = pl > l ? pl : l; }
        if (l != lev[q]) { lev[q] = (uint16_t)l; ch = true; }
      }
      if (ch) s_changed = 1;
      __syncthreads();
      const int again = s_changed;
      __syncthreads();
      if (!again) break;
    }
    for (uint32_t q = tid; q < n; q += 1024) B.run_level[B.run_list[q]] = lev[q];
    __syncthreads();
  } else {
  for (uint32_t q = tid; q < n; q += 1024) B.run_level[B.run_list[q]] = 1;
  __syncthreads();
  for (;;) {
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
    hipLaunchKernelGGL(k_scan_ctbs, dim3(P.n_ctbs), dim3(64), 0, st, P, B);
    // (the number of runs is only known on the device: a fixed grid of wavefronts walks the run list; the thread-per-run passes
    //  are launched for the most runs the CTB grid has seen ... which the host does not know either: for one run per intra TU)
    hipLaunchKernelGGL(k_scan_runs1, dim3(4096), dim3(64), 0, st, P, B);
    const unsigned g = (unsigned)((P.cap_runs + 63) / 64);
    hipLaunchKernelGGL(k_scan_runs<2>, dim3(g), dim3(64), 0, st, P, B);
    if (P.flags & SCANF_MAILBOX) hipLaunchKernelGGL(k_scan_runs<3>, dim3(g), dim3(64), 0, st, P, B);
    static const hipError_t lds_ok = hipFuncSetAttribute((const void*)k_scan_order, hipFuncAttributeMaxDynamicSharedMemorySize, SCO_LDS_BYTES);
    if (lds_ok != hipSuccess) return lds_ok;
    hipLaunchKernelGGL(k_scan_order, dim3(1), dim3(1024), SCO_LDS_BYTES, st, P, B, L.cap_levels);
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
