// k_scan.hip -- the passes of scan_core.h as gfx950 kernels (one thread per TU record / CTB / run, behind the upload of the raw
// records on the decoder's copy stream) and, from the same functions compiled for the host, the CPU rehearsal the equivalence
// tests run without a GPU.  Integer / byte work on a few megabytes of records: no LDS tiling to speak of, no MFMA; the passes
// are latency chains of a lone thread per unit, and there are thousands of units.
#include <vector>

#include "scan.h"

namespace d265 {

// ------------------------------------------------------------------------------------------------ device kernels
__global__ __launch_bounds__(256)
void k_scan_tus(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && B.err_word) *B.err_word = 0;                  // (the picture's kernels that may raise it come behind the scan)
  ScanTuSums S = { 0, 0, 0, 0, 0 };
  ScanParams Pt = P; Pt.flags &= ~SCANF_CHECK_POS;            // (the positions: below, sixteen lanes per TU)
  if (i < P.n_tus) scan_tu(Pt, B, i, S);
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
  // coefficient positions inside their TU's block (de265hip_picture_build's job on the host until round 3): the workgroup's 256
  // records sixteen at a time, sixteen lanes striding each list (a thread walking its own list alone took 7x as long); a
  // position beyond the block is folded into it and the picture fails
  if (P.flags & SCANF_CHECK_POS) {
    const int sub = threadIdx.x & 15;
    bool bad = false;
    for (int g = 0; g < 16; g++) {
      const int j = blockIdx.x * 256 + g * 16 + (threadIdx.x >> 4);
      if (j >= P.n_tus) break;
      const de265hip_tu tu = B.tus[j];
      if (!(tu.flags & DE265HIP_TU_CBF) || !scan_tu_valid(P, tu)) continue;
      const unsigned nS = 1u << (2 * tu.log2_size), n = tu.n_coeff;
      uint16_t* p = B.coeff_pos + tu.coeff_offset;
      for (unsigned k = sub; k < n; k += 16) {
        const unsigned q = p[k];
        if (q >= nS) { p[k] = (uint16_t)(q & (nS - 1)); bad = true; }
      }
    }
    if (bad) scan_fail(B, DE265HIP_ERROR_DECODING);
  }
}

// exclusive prefix over the CTBs in tile-scan (decode) order of the seven per-CTB counts, by one workgroup: every thread sums
// a contiguous chunk of CTBs, the workgroup scans the 1024 chunk sums in LDS, every thread writes its chunk's bases
__global__ __launch_bounds__(1024)
void k_scan_prefix(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  const uint32_t cap_resid = J.job[blockIdx.y].cap_resid;
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
    if (P.n_tus == 0 && B.err_word) *B.err_word = 0;
    scan_prefix_finish_totals(B, tot);
    B.counts->victim = 0xFFFFFFFFu;
    // (overlapping intra TUs - a malformed description - could ask for more residual samples than the picture has)
    if (tot[6] > cap_resid || tot[5] > P.cap_runs) scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE);
  }
}

// ---- wave-level helpers (wave64).  Reductions by DPP row shifts + row broadcasts (no LDS round trip: a __shfl is a
// ds_bpermute, ~100 cycles, and six of them in a row make a reduction cost more than everything else in a step of the CTB pass)
__device__ __forceinline__ int wave_max_i(int v)          // (of non-negative values; the result in every lane)
{
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));   // row_shr:1
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));   // row_shr:2
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));   // row_shr:4
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));   // row_shr:8
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true));   // row_bcast:15
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true));   // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_sum_u(uint32_t x)
{
  int v = (int)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
  return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i(int v) { return 0x7FFFFFFF - wave_max_i(0x7FFFFFFF - v); }      // (v >= 0)
__device__ __forceinline__ uint64_t lanes_below(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }
#define WAVE_ORDER() asm volatile("" ::: "memory")        // single-wavefront workgroups: LDS executes a wavefront's operations in order; only the compiler must not move them

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
void k_scan_ctbs(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  __shared__ uint32_t win[3][SCW_W * SCW_H];
  __shared__ uint8_t s_ntus[768];
  const int rs = blockIdx.x, lane = threadIdx.x;
  if (rs >= P.n_ctbs || B.counts->status) return;
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
      const uint32_t pos_u = __builtin_amdgcn_readlane(pos, src), shape_u = __builtin_amdgcn_readlane(shape, src);
      const uint64_t mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(av >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)av, src);
      const uint64_t need0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(nd >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)nd, src);
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
      const int p0 = loc_m ? __builtin_amdgcn_readlane(vrun, __builtin_ctzll(loc_m)) : -1;
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
      WAVE_ORDER();                                      // (every lane has read s_ntus and the window)
      if (lane == 0) {
        s_ntus[r]++;
        B.tu_info[base + src] = (uint32_t)r | ((uint32_t)llev << 16) | (foreign ? SCAN_TI_FOREIGN : 0u) | SCAN_TI_INTRA;
        B.tu_run[base + src] = ibase + (uint32_t)r;
      }
      const int n4 = nT >> 2;
      if (lane < n4 * n4) W[(wy0 + lane / n4 + 1) * SCW_W + (wx0 + lane % n4 + 1)] = (uint32_t)r | ((uint32_t)llev << 16) | (1u << 31);
      WAVE_ORDER();
    }
  }
  // ---- the CTB's runs: sizes, CTB, a place in the run list
  uint32_t at = 0;
  if (lane == 0) { C.n_runs = (uint32_t)n_local; atomicAdd(&B.counts->n_runs, (uint32_t)n_local); at = atomicAdd(&B.counts->n_listed, (uint32_t)n_local); }
  at = __shfl(at, 0, 64);
  for (int q = lane; q < n_local; q += 64) { B.run_ntus[ibase + q] = s_ntus[q]; B.run_rs[ibase + q] = (uint32_t)rs; B.run_list[at + q] = ibase + (uint32_t)q; }
}

// The run pass, one WAVEFRONT per run (persistent: a fixed grid walks the run list).  scan_core.h scan_run is the same pass as
// one thread's loop.  A run is a chain of dependent global round trips if written naively (record -> CTB -> TU words -> TU
// records -> cells -> their TUs' runs ..., ~50 of them: 100 us per run); here the run's TU records, masks and words are fetched
// ONCE into LDS (a lane each), every later step works on LDS, and the producers are resolved in two batched round trips
// (all needed cells, then the run ids behind them) into a hash set in LDS.
#define SCR_MAX 256
#define SCR_CAND (16 * 33)
__global__ __launch_bounds__(64)
void k_scan_runs1(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  __shared__ int tix[SCR_MAX];
  __shared__ uint4 s_rec[SCR_MAX];                     // the TU records (de265hip_tu, 16 bytes)
  __shared__ uint64_t s_need[SCR_MAX], s_av[SCR_MAX];
  __shared__ uint32_t keys[SCR_MAX], sorted[SCR_MAX], s_samp[SCR_MAX + 1];
  __shared__ uint8_t s_lev[SCR_MAX], s_coll[SCR_MAX], s_rdy[64];
  __shared__ uint32_t s_tab[512], s_cand[SCR_CAND];
  __shared__ uint32_t s_nd, s_ncand;
  const int lane = threadIdx.x;
  if (B.counts->status) return;
  const uint32_t n_listed = B.counts->n_listed;
  uint32_t cls_start[4];
  for (int k = 0; k < 4; k++) cls_start[k] = scan_l0_class_start(B.counts->n_l0_size, k);
  auto rec_of = [&](int k) -> de265hip_tu { de265hip_tu t; const uint4 v = s_rec[k]; __builtin_memcpy(&t, &v, 16); return t; };
  for (uint32_t qrun = blockIdx.x; qrun < n_listed; qrun += gridDim.x) {
    const uint32_t s = B.run_list[qrun];
    const int rs = (int)B.run_rs[s];
    const ScanCtb& C = B.ctb[rs];
    const int r = (int)(s - C.intra_base);
    const uint32_t c_first = C.first_tu, c_end = C.end_tu;
    // ---- its TUs (decode order), and what the runs before it in this CTB take of the CTB's lists
    int n = 0;
    uint32_t n_before = 0, samp_before = 0, ro_before[4] = { 0, 0, 0, 0 }, rext_ro_before = 0;
    for (uint32_t base = c_first; base < c_end; base += 64) {
      const uint32_t i = base + lane;
      uint32_t ti = 0;
      if (i < c_end) ti = B.tu_info[i];
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
      if (intra && rr == r) { const int k = n + __popcll(m & lanes_below(lane)); if (k < SCR_MAX) { tix[k] = (int)i; s_lev[k] = (uint8_t)SCAN_TI_LLEV(ti); s_coll[k] = (ti & SCAN_TI_FOREIGN) ? 1 : 0; } }
      n += __popcll(m);
    }
    n_before = wave_sum_u(n_before); samp_before = wave_sum_u(samp_before); rext_ro_before = wave_sum_u(rext_ro_before);
    for (int k = 0; k < 4; k++) ro_before[k] = wave_sum_u(ro_before[k]);
    WAVE_ORDER();
    if (n == 0 || n > 255 || n != (int)B.run_ntus[s]) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); return; }
    // ---- the run's records into LDS, a lane each (the only time they are read from memory)
    bool foreign = false;
    for (int k = lane; k < n; k += 64) {
      const int i = tix[k];
      s_rec[k] = *reinterpret_cast<const uint4*>(B.tus + i);
      s_need[k] = B.tu_need[i]; s_av[k] = B.tu_avail[i];
      foreign = foreign || s_coll[k];
    }
    WAVE_ORDER();
    // ---- box, window reach, samples, levels: every lane its TUs (k = lane, lane + 64, ..), then across the lanes
    int x0 = 1 << 30, y0 = 1 << 30, x1 = 0, y1 = 0, wx1 = 0, wy1 = 0, nl = 0, c = 0;
    uint32_t own_samples = 0;
    bool big = false, too_big = false;
    for (int k = lane; k < n; k += 64) {
      const de265hip_tu tu = rec_of(k);
      const int nT = 1 << tu.log2_size;
      c = tu.c_idx;
      x0 = min(x0, (int)tu.x0); y0 = min(y0, (int)tu.y0); x1 = max(x1, tu.x0 + nT); y1 = max(y1, tu.y0 + nT);
      wx1 = max(wx1, tu.x0 + 2 * nT); wy1 = max(wy1, tu.y0 + 2 * nT);
      own_samples += (uint32_t)(nT * nT);
      nl = max(nl, (int)s_lev[k]);
      big = big || tu.log2_size == 4; too_big = too_big || tu.log2_size > 4;
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
        const de265hip_tu tu = rec_of(k);
        const uint64_t avail = s_av[k];
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
      mb_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb_id);
      if (mb_id >= P.cap_mb) mb_id = 0xFFFFFFFFu;            // (beyond the mailboxes there are: the run does without)
      else if ((P.flags & SCANF_MB_PHASES) && c == 0 && x1 - x0 <= 64 && y1 - y0 <= 64) {
        s_rdy[lane] = 255;
        WAVE_ORDER();
        for (int k = lane; k < n; k += 64) {
          const de265hip_tu tu = rec_of(k);
          const int nT = 1 << tu.log2_size;
          const uint8_t ep = (uint8_t)(s_lev[k] - 1);
          if (tu.y0 + nT == y1) for (int q = 0; q < (nT >> 1); q++) s_rdy[((tu.x0 - x0) >> 1) + q] = ep;
          if (tu.x0 + nT == x1) for (int q = 0; q < (nT >> 1); q++) s_rdy[32 + ((tu.y0 - y0) >> 1) + q] = ep;
        }
        WAVE_ORDER();
        B.rdy_tab[64 * (size_t)mb_id + lane] = s_rdy[lane];
      }
    }
    if (lane < 3) B.mbx[3 * (size_t)s + lane] = lane == 0 ? mb_id : 0xFFFFFFFFu;
    if (lane == 0) B.pub_flag[s] = 0;
    // ---- chain order: list (wavefront, or 4 = collective) | in-run level | decode index; the rank of a TU inside its level
    // counts the non-collective TUs of that level before it
    const int nwv = micro ? 1 : P.run_waves;
    WAVE_ORDER();
    for (int k = lane; k < n; k += 64) s_coll[k] = (rec_of(k).log2_size > 3 && !micro) ? 1 : 0;
    WAVE_ORDER();
    for (int k = lane; k < n; k += 64) {
      int rank = 0;
      for (int q = 0; q < k; q++) rank += (s_lev[q] == s_lev[k] && !s_coll[q]) ? 1 : 0;
      const int list = s_coll[k] ? 4 : rank % nwv;
      keys[k] = ((uint32_t)list << 20) | ((uint32_t)s_lev[k] << 8) | (uint32_t)k;
    }
    WAVE_ORDER();
    int we[4] = { 0, 0, 0, 0 };
    for (int k = lane; k < n; k += 64) {
      int posn = 0;
      for (int q = 0; q < n; q++) posn += keys[q] < keys[k] ? 1 : 0;
      sorted[posn] = keys[k];
      for (int w = 0; w < 4; w++) we[w] += (int)(keys[k] >> 20) <= w ? 1 : 0;
    }
    for (int w = 0; w < 4; w++) o.wave_end[w] = (uint16_t)wave_sum_u((uint32_t)we[w]);
    o.n_lvls = (uint16_t)(nl > 0 ? nl - 1 : 0);
    WAVE_ORDER();
    // ---- sample offsets in chain order (exclusive prefix of the TU sizes): a lane's four entries, then across the lanes
    {
      uint32_t sz[4], mine = 0;
      for (int j = 0; j < 4; j++) { const int oi = 4 * lane + j; sz[j] = oi < n ? 1u << (2 * rec_of((int)(sorted[oi] & 0xFFu)).log2_size) : 0u; mine += sz[j]; }
      uint32_t incl = mine;                                    // inclusive scan over the lanes (Hillis-Steele on ds_bpermute: six steps, once per run)
      for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64); if (lane >= off) incl += v; }
      uint32_t acc = incl - mine;
      for (int j = 0; j < 4; j++) { const int oi = 4 * lane + j; if (oi <= n && oi < SCR_MAX + 1) s_samp[oi] = acc; acc += sz[j]; }
      o.n_samples = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    WAVE_ORDER();
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
        const int k = (int)(sorted[oi] & 0xFFu);
        i = tix[k];
        tu = rec_of(k);
        tt = scan_task_of(tu);
        const int m = tu.intra_mode < 35 ? tu.intra_mode : 1;
        tt.angle = (int8_t)scan_intra_angle(m); tt.inv_angle = (int16_t)scan_inv_angle(m);
        tt.avail = s_av[k];
        tt.run_level = (uint8_t)(s_lev[k] - 1);
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
    // ---- producers: the cells of every needed unit of every TU (64 TUs at a time, a lane each, into a list), then - a lane
    // per list entry, all loads of a step in flight together - the TU behind the cell and the run behind the TU; each run
    // once: a hash set in LDS (the cells a run inside one CTB can need number fewer than its slots)
    for (int q = lane; q < 512; q += 64) s_tab[q] = 0xFFFFFFFFu;
    if (lane == 0) s_nd = 0;
    {
      const int mw = P.map_w[c];
      const ScanCell* cells = B.cell[c];
      for (int kb = 0; kb < n; kb += 16) {
        if (lane == 0) s_ncand = 0;
        WAVE_ORDER();
        const int k = kb + lane;
        if (lane < 16 && k < n) {
          const de265hip_tu tu = rec_of(k);
          for (uint64_t need = s_need[k]; need; need &= need - 1)
            s_cand[atomicAdd(&s_ncand, 1u)] = (uint32_t)scan_cell_of(__builtin_ctzll(need), tu.x0, tu.y0, 1 << tu.log2_size, mw);
        }
        WAVE_ORDER();
        const uint32_t nc = s_ncand;
        for (uint32_t q0 = 0; q0 < nc; q0 += 256) {
          uint32_t j[4], ps[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const uint32_t q = q0 + 64 * u + lane; j[u] = q < nc ? (uint32_t)cells[s_cand[q]] : 0u; }
#pragma unroll
          for (int u = 0; u < 4; u++) ps[u] = j[u] ? B.tu_run[j[u] - 1] : s;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (ps[u] == s) continue;
            uint32_t hsh = (ps[u] * 2654435761u) >> 23;
            for (int probe = 0; probe < 512; probe++, hsh = (hsh + 1) & 511) {
              const uint32_t old = atomicCAS(&s_tab[hsh], 0xFFFFFFFFu, ps[u]);
              if (old == 0xFFFFFFFFu) { atomicAdd(&s_nd, 1u); break; }
              if (old == ps[u]) break;
            }
          }
        }
        WAVE_ORDER();
      }
    }
    WAVE_ORDER();
    const uint32_t nd = s_nd;
    if (nd > 500) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
    if (nd) {
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(&B.counts->n_deps_alloc, nd);
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
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
      uint32_t alg = 0;
      const uint32_t bpp = (uint32_t)(c ? P.bppC : P.bppY);
      for (int k = lane; k < n; k += 64) { const uint32_t nT = 1u << rec_of(k).log2_size; alg += bpp * (4 * nT + 1) + bpp * nT * nT; }
      alg = wave_sum_u(alg);
      if (lane == 0) scan_add64(&B.counts->alg_intra_front, alg);
    }
    WAVE_ORDER();
  }
}

// The second run pass, a wavefront per run: the run's chain-ordered TU records are staged in LDS by all lanes, then ONE lane
// runs scan_run2 on them (producers that are front runs leave the list; mailbox segments and need epochs of a reader).  The
// logic is a few thousand scalar steps on ~40 records: not worth spreading over lanes, but on records in LDS it takes ~15 us
// instead of the ~500 us a thread took that fetched them one by one from memory next to 63 others doing the same.
__global__ __launch_bounds__(64)
void k_scan_runs2(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  __shared__ TuTask s_tus[SCR_MAX];
  const int lane = threadIdx.x;
  if (B.counts->status) return;
  const uint32_t n_listed = B.counts->n_listed;
  for (uint32_t qrun = blockIdx.x; qrun < n_listed; qrun += gridDim.x) {
    const uint32_t s = B.run_list[qrun];
    const RunTask* R = B.runs + s;
    const uint32_t mic = R->micro, n = R->n_tus, first = R->first_tu;
    const uint32_t n_all = B.run_nall[s] & 0x7FFFFFFFu;
    const bool foreign = B.run_nall[s] >> 31;
    // (only a run that can become a mailbox reader looks at its TU records)
    const bool cand = (P.flags & SCANF_MAILBOX) && !(mic & 1) && (mic & 2) && !foreign && n_all > 0 && n_all <= 8 && n <= SCR_MAX;
    if (cand) {
      const uint4* src = reinterpret_cast<const uint4*>(B.run_tus + first);
      uint4* dst = reinterpret_cast<uint4*>(s_tus);
      for (uint32_t q = lane; q < 2 * n; q += 64) dst[q] = src[q];
    }
    WAVE_ORDER();
    if (lane == 0) scan_run2(P, B, s, cand ? s_tus : nullptr);
    WAVE_ORDER();
  }
}

// Ticket slots (scan_core.h "tickets"), one workgroup: the CTBs in ctb_order are dealt to the threads in contiguous chunks; how
// a chunk's runs fill tickets depends on how full the open ticket is when the chunk begins, so every thread first computes its
// chunk's effect for each of the eight possible fill states (a table), the tables are composed by a prefix scan (composition
// of such tables is associative), and every thread then walks its chunk again from its true start state and writes the slots.
#define SCO_THREADS 1024
__device__ void scan_order_body(const ScanParams& P, const ScanBufs& B, uint32_t cap_levels);
__global__ __launch_bounds__(SCO_THREADS)
void k_scan_order(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  const uint32_t cap_levels = J.job[blockIdx.y].cap_levels;
  // the third run pass first (a thread per listed run, scan_core.h scan_run3: the runs somebody reads through their mailbox)
  if (!B.counts->status && (P.flags & SCANF_MAILBOX)) {
    const uint32_t n = B.counts->n_listed;
    for (uint32_t q = threadIdx.x; q < n; q += SCO_THREADS) scan_run3(P, B, B.run_list[q]);
  }
  __threadfence();
  __syncthreads();
  scan_order_body(P, B, cap_levels);
  // ---- the scan's verdict and counts to the host: into the picture's pinned record, then its ready word (system scope)
  __threadfence();
  __syncthreads();
  if (threadIdx.x < sizeof(ScanCounts) / 4 - 2 && B.host_counts)
    reinterpret_cast<volatile uint32_t*>(B.host_counts)[threadIdx.x] = reinterpret_cast<volatile uint32_t*>(B.counts)[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && B.host_counts) __hip_atomic_store(&B.host_counts->ready, B.ready_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ void scan_order_body(const ScanParams& P, const ScanBufs& B, uint32_t cap_levels)
{
  __shared__ uint32_t tab[SCO_THREADS][RUN_TICKET_SLOTS];          // per chunk and fill state at its start: tickets it opens | fill state at its end << 28
  __shared__ uint32_t s_diag[4096];
  __shared__ uint32_t s_widest, s_ready, s_ndiag;
  const int tid = threadIdx.x;
  ScanCounts& K = *B.counts;
  if (K.status) return;
  const int n = P.n_ctbs, chunk = (n + SCO_THREADS - 1) / SCO_THREADS;
  const int t0 = min(n, tid * chunk), t1 = min(n, t0 + chunk);
  const uint32_t victim = K.victim;
  const int n_diag = P.ctbs_w + 2 * P.ctbs_h;
  uint32_t* diag = n_diag <= 4096 ? s_diag : B.lvl_cnt;          // ticketed runs per anti-diagonal (for the worker count)
  for (int q = tid; q < n_diag; q += SCO_THREADS) diag[q] = 0;
  if (tid == 0) { s_widest = 0; s_ready = 0; s_ndiag = 0; }
  __syncthreads();
  // ---- phase 1: the chunk's table
  ScanTicketState st[RUN_TICKET_SLOTS];
  for (int o = 0; o < RUN_TICKET_SLOTS; o++) { st[o].tickets = 0; st[o].fill = (uint32_t)o; }
  uint32_t ready = 0;
  for (int t = t0; t < t1; t++) {
    const int rs = B.ctb_order[t];
    const ScanCtb& C = B.ctb[rs];
    const uint32_t nr = C.n_runs, ib = C.intra_base;
    uint32_t cnt = 0;
    for (uint32_t r = 0; r < nr; r++) {
      const uint32_t s = ib + r;
      const uint32_t mic = B.runs[s].micro;
      if ((mic & RUN_MICRO_FRONT) || s == victim) continue;
      uint32_t tk, sl;
      for (int o = 0; o < RUN_TICKET_SLOTS; o++) scan_ticket_step(st[o], mic & 1, &tk, &sl);
      cnt++;
      if (B.runs[s].n_deps == 0) ready++;
    }
    if (cnt) atomicAdd(&diag[rs % P.ctbs_w + 2 * (rs / P.ctbs_w)], cnt);
  }
  if (ready) atomicAdd(&s_ready, ready);
  for (int o = 0; o < RUN_TICKET_SLOTS; o++) tab[tid][o] = st[o].tickets | (st[o].fill << 28);
  __syncthreads();
  // ---- phase 2: inclusive prefix composition (Hillis-Steele): tab[i] := tab[i - off] then tab[i]
  for (int off = 1; off < SCO_THREADS; off <<= 1) {
    uint32_t v[RUN_TICKET_SLOTS];
    for (int o = 0; o < RUN_TICKET_SLOTS; o++) {
      v[o] = tab[tid][o];
      if (tid >= off) {
        const uint32_t a = tab[tid - off][o];                    // the earlier chunks from state o ...
        const uint32_t bb = tab[tid][a >> 28];                   // ... then this one from where they end
        v[o] = ((a & 0x0FFFFFFFu) + (bb & 0x0FFFFFFFu)) | (bb & 0xF0000000u);
      }
    }
    __syncthreads();
    for (int o = 0; o < RUN_TICKET_SLOTS; o++) tab[tid][o] = v[o];
    __syncthreads();
  }
  // the state this chunk starts from: what all chunks before it make of (0 tickets, closed)
  ScanTicketState me = { 0, 0 };
  if (tid > 0) { const uint32_t v = tab[tid - 1][0]; me.tickets = v & 0x0FFFFFFFu; me.fill = v >> 28; }
  const uint32_t total = tab[SCO_THREADS - 1][0] & 0x0FFFFFFFu;
  if ((unsigned long long)total * RUN_TICKET_SLOTS > P.cap_slots) { if (tid == 0) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
  for (uint32_t q = tid; q < total * RUN_TICKET_SLOTS; q += SCO_THREADS) B.slots[q] = 0xFFFFFFFFu;
  uint32_t w = 0, nd = 0;
  for (int q = tid; q < n_diag; q += SCO_THREADS) { w = max(w, diag[q]); nd += diag[q] ? 1 : 0; }
  atomicMax(&s_widest, w); atomicAdd(&s_ndiag, nd);
  __syncthreads();
  // ---- phase 3: the slots
  for (int t = t0; t < t1; t++) {
    const int rs = B.ctb_order[t];
    const ScanCtb& C = B.ctb[rs];
    const uint32_t nr = C.n_runs, ib = C.intra_base;
    for (uint32_t r = 0; r < nr; r++) {
      const uint32_t s = ib + r;
      const uint32_t mic = B.runs[s].micro;
      if ((mic & RUN_MICRO_FRONT) || s == victim) continue;
      uint32_t tk, sl;
      scan_ticket_step(me, mic & 1, &tk, &sl);
      B.slots[tk * RUN_TICKET_SLOTS + sl] = (mic & 1) ? (s | 0x80000000u) : s;
    }
  }
  if (tid == 0) {
    K.n_batches = total;
    // workers: as many runs as can be in flight together - an anti-diagonal of the picture, or the runs that wait for nothing
    K.widest = max(s_widest, s_ready);
    K.max_rl = s_ndiag;
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
  o_tu_avail = add(nt * 8); o_tu_need = add(nt * 8); o_tu_info = add(nt * 4); o_tu_run = add(nt * 4);
  o_run_rs = add(nr * 4); o_run_nall = add(nr * 4); o_run_level = add(nr * 4); o_run_list = add(nr * 4); o_pub_flag = add(nr);
  o_rdy_tab = add((size_t)P.cap_mb * 64);
  cap_levels = (uint32_t)nr + 2;
  o_lvl_cnt = add(std::max((size_t)cap_levels * 5, (size_t)(P.ctbs_w + 2 * P.ctbs_h)) * 4);      // (scan_order: a counter per CTB anti-diagonal)
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
  B.tu_avail = (uint64_t*)(base + o_tu_avail); B.tu_need = (uint64_t*)(base + o_tu_need); B.tu_info = (uint32_t*)(base + o_tu_info); B.tu_run = (uint32_t*)(base + o_tu_run);
  B.run_rs = (uint32_t*)(base + o_run_rs); B.run_nall = (uint32_t*)(base + o_run_nall); B.run_level = (uint32_t*)(base + o_run_level);
  B.run_list = (uint32_t*)(base + o_run_list); B.pub_flag = base + o_pub_flag; B.rdy_tab = base + o_rdy_tab;
  B.lvl_cnt = (uint32_t*)(base + o_lvl_cnt);
  B.l0 = (TuTask*)(base + o_l0); B.l0x = (TuTask*)(base + o_l0x); B.runs = (RunTask*)(base + o_runs); B.run_tus = (TuTask*)(base + o_run_tus);
  B.deps = (uint32_t*)(base + o_deps); B.slots = (uint32_t*)(base + o_slots); B.front_idx = (uint32_t*)(base + o_front);
  B.mbx = (uint32_t*)(base + o_mbx); B.mb_segs = (uint32_t*)(base + o_mb_segs);
}

// ------------------------------------------------------------------------------------------------ enqueue
// The passes for a BATCH of pictures in one set of launches (grid.y = picture): the scan is a chain of six dependent kernels,
// each a latency chain that leaves most of the device idle, two of them single workgroups; launched per picture, a decoder's
// scans and the reconstruction kernels of the other streams kept queueing behind each other (round 4: 2 100 pictures/s whatever
// the number of copy streams).  A batch pays the chain once.
hipError_t scan_enqueue_batch(hipStream_t st, const ScanBatch& J)
{
  int max_tus = 0, max_ctbs = 0;
  for (int i = 0; i < J.n; i++) { max_tus = std::max(max_tus, J.job[i].P.n_tus); max_ctbs = std::max(max_ctbs, J.job[i].P.n_ctbs); }
  const unsigned ny = (unsigned)J.n;
  if (max_tus > 0) hipLaunchKernelGGL(k_scan_tus, dim3((max_tus + 255) / 256, ny), dim3(256), 0, st, J);
  hipLaunchKernelGGL(k_scan_prefix, dim3(1, ny), dim3(1024), 0, st, J);
  if (max_tus > 0) {
    hipLaunchKernelGGL(k_scan_ctbs, dim3(max_ctbs, ny), dim3(64), 0, st, J);
    // (the number of runs is only known on the device: fixed grids of wavefronts walk the run lists)
    // (512 wavefronts: with 128 / 256 / 512 / 1024 the product path of the bench made 1 860 / 2 310 / 2 630 / 2 380 pictures/s -
    //  fewer leave the run passes' latency chains too long, more crowd the reconstruction kernels of the other streams)
    static const int run_grid = getenv("DE265HIP_SCAN_RUN_GRID") ? atoi(getenv("DE265HIP_SCAN_RUN_GRID")) : 512;
    hipLaunchKernelGGL(k_scan_runs1, dim3(run_grid, ny), dim3(64), 0, st, J);
    hipLaunchKernelGGL(k_scan_runs2, dim3(run_grid, ny), dim3(64), 0, st, J);
  }
  hipLaunchKernelGGL(k_scan_order, dim3(1, ny), dim3(SCO_THREADS), 0, st, J);      // (always: it reports to the host)
  return hipGetLastError();
}

hipError_t scan_enqueue(hipStream_t st, const ScanParams& P, const ScanBufs& B, const ScanLayout& L, uint8_t* base, uint32_t cap_resid)
{
  (void)base;
  ScanBatch J; J.n = 1;
  J.job[0].P = P; J.job[0].B = B; J.job[0].cap_resid = cap_resid; J.job[0].cap_levels = L.cap_levels;
  return scan_enqueue_batch(st, J);
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
  // scan_order, serially: the CTBs by (anti-diagonal, row), their runs in order of creation
  {
    ScanTicketState st = { 0, 0 };
    const int n_diag = P.ctbs_w + 2 * P.ctbs_h;
    std::vector<uint32_t> diag((size_t)n_diag, 0);
    uint32_t ready = 0;
    struct Placed { uint32_t s, ticket, slot; };
    std::vector<Placed> placed;
    for (int t = 0; t < P.n_ctbs; t++) {
      const int rs = B.ctb_order[t];
      const ScanCtb& C = B.ctb[rs];
      for (uint32_t r = 0; r < C.n_runs; r++) {
        const uint32_t s = C.intra_base + r, mic = B.runs[s].micro;
        if ((mic & RUN_MICRO_FRONT) || s == K.victim) continue;
        uint32_t tk, sl;
        scan_ticket_step(st, mic & 1, &tk, &sl);
        placed.push_back({ (mic & 1) ? (s | 0x80000000u) : s, tk, sl });
        diag[rs % P.ctbs_w + 2 * (rs / P.ctbs_w)]++;
        if (B.runs[s].n_deps == 0) ready++;
      }
    }
    uint32_t widest = ready, ndiag = 0;
    for (uint32_t v : diag) { widest = std::max(widest, v); ndiag += v ? 1 : 0; }
    K.n_batches = st.tickets; K.widest = widest; K.max_rl = ndiag;
    if ((unsigned long long)st.tickets * RUN_TICKET_SLOTS > P.cap_slots) { scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
    for (uint32_t q = 0; q < st.tickets * RUN_TICKET_SLOTS; q++) B.slots[q] = 0xFFFFFFFFu;
    for (const Placed& pl : placed) B.slots[pl.ticket * RUN_TICKET_SLOTS + pl.slot] = pl.s;
  }
}

}  // namespace d265
