// k_scan.hip -- the passes of scan_core.h as gfx950 kernels (one thread per TU record / CTB / run, behind the upload of the raw
// records on the decoder's copy stream) and, from the same functions compiled for the host, the CPU rehearsal the equivalence
// tests run without a GPU.  Integer / byte work on a few megabytes of records: no LDS tiling to speak of, no MFMA; the passes
// are latency chains of a lone thread per unit, and there are thousands of units.
#include <vector>

#include "scan.h"
#include "env.h"

namespace d265 {

// ------------------------------------------------------------------------------------------------ device kernels
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
// The scan's kernels are chains of dependent steps in a few wavefronts, and what the decoder's launches wait for; next to them
// on the CUs run the reconstruction kernels of the other decoders, thousands of wavefronts that fill every issue slot.  At the
// default wave priority a scan wavefront gets its turn once per round of the resident wavefronts: its chains ran 2-4x slower
// than alone.  (DE265HIP_SCAN_PRIO compiled as a constant: s_setprio takes an immediate.)
#ifndef D265_SCAN_PRIO
#define D265_SCAN_PRIO 3
#endif
#define SCAN_PRIO() __builtin_amdgcn_s_setprio(D265_SCAN_PRIO)
#define WAVE_ORDER() asm volatile("" ::: "memory")        // single-wavefront workgroups: LDS executes a wavefront's operations in order; only the compiler must not move them

// The per-TU pass on the device.  scan_core.h scan_tu is the same pass as plain code (the CPU rehearsal); what differs here:
// - the availability mask in closed form.  scan_tu asks MinTbAddrZS for every below-left and above-right unit (a dependent
//   load each, up to 33 in a row: 90 us for a kernel of 100 000 threads).  The below-left units of a TU all lie in ONE block
//   of the TU's size, aligned like it, and so do the above-right ones; aligned blocks of equal size cover disjoint ranges of
//   the z-scan order, so one comparison settles a whole block (4:2:2 chroma blocks are not square in luma: those TUs take
//   scan_tu's loops);
// - the per-CTB counts by one atomic per counter, CTB and wavefront (the records of a CTB are contiguous: a wavefront sees one
//   or two CTBs) instead of one to three per TU.
__device__ __forceinline__ void scan_tu_dev(const ScanParams& P, const ScanBufs& B, int i, bool have, uint32_t& alg_resid, uint32_t& alg_intra, uint32_t& n_tasks, uint32_t* s_tot)
{
  const int lane = threadIdx.x & 63;
  int cls = 0, rx = 0, ctu = -1;
  de265hip_tu tu; memset(&tu, 0, sizeof(tu));
  bool fail_range = false;
  if (have) {
    tu = B.tus[i];
    B.tu_info[i] = 0;                                    // (scan_ctb fills the words of the intra TUs)
    if (!scan_tu_valid(P, tu)) { fail_range = true; have = false; }
  }
  if (have) {
    ctu = scan_tu_ctb(P, tu);
    // -- where the record array enters a CTB
    int prev = -1;
    if (i > 0) { const de265hip_tu q = B.tus[i - 1]; if (scan_tu_valid(P, q)) prev = scan_tu_ctb(P, q); }
    if (i == 0 || prev != ctu) {
      atomicAdd(&B.ctb[ctu].seen, 1u);
      B.ctb[ctu].first_tu = (uint32_t)i;
      if (prev >= 0) B.ctb[prev].end_tu = (uint32_t)i;
    }
    if (i == P.n_tus - 1) B.ctb[ctu].end_tu = (uint32_t)P.n_tus;
    cls = scan_tu_class(P, B, tu, &rx);
    if ((rx & D265_RX_XCC) && cls != 0) { uint64_t li; int rl; if (!scan_xcc_luma(P, B, i, &li, &rl)) { fail_range = true; cls = 0; } }
  }
  if (__ballot(fail_range) != 0 && fail_range) scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE);
  const int nT = 1 << tu.log2_size, c = tu.c_idx;
  const bool cbf = (tu.flags & DE265HIP_TU_CBF) && tu.n_coeff;
  const bool ro = cls == 3 && (cbf || (rx & D265_RX_XCC));
  if (cls != 0) {
    const uint32_t bpp = (uint32_t)(c ? P.bppC : P.bppY);
    const uint32_t coef_bytes = cbf ? (4u * tu.n_coeff < 2u * nT * nT ? 4u * tu.n_coeff : 2u * nT * nT) : 0u;
    n_tasks++;
    if (cls != 3) { if (cbf) alg_resid += coef_bytes + 2 * bpp * nT * nT; }
    else { alg_resid += coef_bytes; alg_intra += bpp * (4 * nT + 1) + bpp * nT * nT; }
  }
  // -- the per-CTB counts: the lanes of one CTB together
  for (uint64_t todo = __ballot(cls != 0); todo;) {
    const int lead = __builtin_ctzll(todo);
    const int lctu = __builtin_amdgcn_readlane(ctu, lead);
    const bool same = cls != 0 && ctu == lctu;
    todo &= ~__ballot(same);
    uint32_t n_inter[4], n_intra_k[4], n_ro[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      n_inter[k] = (uint32_t)__popcll(__ballot(same && cls == 1 && tu.log2_size == k + 2));
      n_intra_k[k] = (uint32_t)__popcll(__ballot(same && cls == 3 && tu.log2_size == k + 2));
      n_ro[k] = (uint32_t)__popcll(__ballot(same && ro && !rx && tu.log2_size == k + 2));
    }
    const uint32_t n_rext_inter = (uint32_t)__popcll(__ballot(same && cls == 2)), n_rext_ro = (uint32_t)__popcll(__ballot(same && ro && rx));
    if (lane == lead) {
      ScanCtb& C = B.ctb[lctu];
      const uint32_t ni = n_intra_k[0] + n_intra_k[1] + n_intra_k[2] + n_intra_k[3];
      // (the picture's totals: per workgroup first, in LDS)
#pragma unroll
      for (int k = 0; k < 4; k++) if (n_inter[k] + n_ro[k]) atomicAdd(&s_tot[k], n_inter[k] + n_ro[k]);
      if (n_rext_inter + n_rext_ro) atomicAdd(&s_tot[4], n_rext_inter + n_rext_ro);
      if (ni) { atomicAdd(&s_tot[5], ni); atomicAdd(&s_tot[6], 16 * n_intra_k[0] + 64 * n_intra_k[1] + 256 * n_intra_k[2] + 1024 * n_intra_k[3]); }
#pragma unroll
      for (int k = 0; k < 4; k++) { if (n_inter[k]) atomicAdd(&C.n_inter[k], n_inter[k]); if (n_ro[k]) atomicAdd(&C.n_ro[k], n_ro[k]); }
      if (n_rext_inter) atomicAdd(&C.n_rext_inter, n_rext_inter);
      if (n_rext_ro) atomicAdd(&C.n_rext_ro, n_rext_ro);
      if (ni) { atomicAdd(&C.n_intra, ni); atomicAdd(&C.n_isamp, 16 * n_intra_k[0] + 64 * n_intra_k[1] + 256 * n_intra_k[2] + 1024 * n_intra_k[3]); }
    }
  }
  if (cls != 3) return;
  // -- an intra TU: neighbour availability (8.4.4.2.2; intrapred.cc:437-527 preproc, :577-688 fill_from_image) as a unit bit mask
  const int sbw = c ? P.subw : 1, sbh = c ? P.subh : 1;
  const int xB = tu.x0, yB = tu.y0, xL = xB * sbw, yL = yB * sbh;
  uint64_t mask = 0;
  if (c != 0 && P.cf == 2) {
    // (4:2:2 chroma: scan_tu's loops, every unit its own comparison)
    const int cx = xL >> P.lc, cy = yL >> P.lc;
    const int corner = nT >> 1, w4 = (P.width + 3) >> 2;
    const uint32_t own = B.ctb_group[ctu];
    const bool aL = xL > 0 && B.ctb_group[((xL - 1) >> P.lc) + cy * P.ctbs_w] == own;
    const bool aT = yL > 0 && B.ctb_group[cx + ((yL - 1) >> P.lc) * P.ctbs_w] == own;
    const bool aTL = xL > 0 && yL > 0 && B.ctb_group[((xL - 1) >> P.lc) + ((yL - 1) >> P.lc) * P.ctbs_w] == own;
    const bool aTR = yL > 0 && (xL + nT * sbw < P.width) && B.ctb_group[((xL + nT * sbw) >> P.lc) + ((yL - 1) >> P.lc) * P.ctbs_w] == own;
    int nBottom = (P.height - yL + sbh - 1) >> (sbh - 1); if (nBottom > 2 * nT) nBottom = 2 * nT;
    int nRight = (P.width - xL + sbw - 1) >> (sbw - 1);   if (nRight > 2 * nT) nRight = 2 * nT;
    const int cur = scan_zs(P, B, xL >> P.lt, yL >> P.lt);
    const bool cip = P.flags & SCANF_CIP;
    auto intra_ok = [&](int xs, int ys) { return !cip || (B.blk_flags[((xs * sbw) >> 2) + ((ys * sbh) >> 2) * w4] & DE265HIP_BLK_INTRA); };
    auto z_ok = [&](int xs, int ys) { return scan_zs(P, B, (xs * sbw) >> P.lt, (ys * sbh) >> P.lt) <= cur; };
    if (aL) for (int y = nBottom - 1; y >= 0; y -= 4) if (z_ok(xB - 1, yB + y) && intra_ok(xB - 1, yB + y)) mask |= 1ull << ((2 * nT - 1 - y) >> 2);
    if (aTL && z_ok(xB - 1, yB - 1) && intra_ok(xB - 1, yB - 1)) mask |= 1ull << corner;
    if (aT || aTR) for (int x = 0; x < nRight; x += 4) if ((x < nT ? aT : aTR) && z_ok(xB + x, yB - 1) && intra_ok(xB + x, yB - 1)) mask |= 1ull << (corner + 1 + (x >> 2));
  } else {
    const int S = nT * sbw;                              // the TU in luma samples (square)
    const int cx = xL >> P.lc, cy = yL >> P.lc, lc = P.lc, cw = P.ctbs_w;
    const int corner = nT >> 1, q4 = nT >> 2;
    const bool hasL = xL > 0, hasT = yL > 0, hasR = xL + S < P.width;
    // the neighbouring CTBs' groups (slice address + tile) and - for the two blocks that may come later in z-scan order - their
    // place in it: all loads up front, none depends on another
    const int ctbL = ((xL - 1) >> lc) + cy * cw, ctbT = cx + ((yL - 1) >> lc) * cw, ctbTL = ((xL - 1) >> lc) + ((yL - 1) >> lc) * cw, ctbTR = ((xL + S) >> lc) + ((yL - 1) >> lc) * cw;
    const int ctbBL = ((xL - 1) >> lc) + ((yL + S) >> lc) * cw;
    int nBottom = (P.height - yL + sbh - 1) >> (sbh - 1); if (nBottom > 2 * nT) nBottom = 2 * nT;      // (sbw, sbh are 1 or 2)
    int nRight = (P.width - xL + sbw - 1) >> (sbw - 1);   if (nRight > 2 * nT) nRight = 2 * nT;
    const bool wantBL = hasL && nBottom > nT;
    const uint32_t own = B.ctb_group[ctu];
    const uint32_t gL = hasL ? B.ctb_group[ctbL] : ~own, gT = hasT ? B.ctb_group[ctbT] : ~own, gTL = (hasL && hasT) ? B.ctb_group[ctbTL] : ~own;
    const uint32_t gTR = (hasT && hasR) ? B.ctb_group[ctbTR] : ~own;
    const int dl = lc - P.lt;
    const int tsC = B.rs2ts[ctu], tsBL = wantBL ? B.rs2ts[ctbBL] : 0, tsTR = (hasT && hasR) ? B.rs2ts[ctbTR] : 0;
    auto morton = [&](int xtb, int ytb) { int v = 0; for (int b2 = 0; b2 < dl; b2++) v |= (((xtb >> b2) & 1) << (2 * b2)) | (((ytb >> b2) & 1) << (2 * b2 + 1)); return v; };
    const int cur = (tsC << (2 * dl)) | morton(xL >> P.lt, yL >> P.lt);
    const bool aL = hasL && gL == own, aT = hasT && gT == own, aTL = hasL && hasT && gTL == own, aTR = hasT && hasR && gTR == own;
    const bool zBL = wantBL && ((tsBL << (2 * dl)) | morton((xL - sbw) >> P.lt, (yL + S) >> P.lt)) <= cur;
    const bool zTR = aTR && ((tsTR << (2 * dl)) | morton((xL + S) >> P.lt, (yL - sbh) >> P.lt)) <= cur;
    auto bits = [](int lo, int hi) -> uint64_t { return hi > lo ? ((hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull)) : 0ull; };      // [lo, hi)
    if (aL) {
      mask |= bits(q4, corner);                                              // beside the TU
      if (zBL) mask |= bits((2 * nT - nBottom) >> 2, q4);                    // below-left, as far as the picture goes
    }
    if (aTL) mask |= 1ull << corner;
    if (aT) mask |= bits(corner + 1, corner + 1 + q4);
    if (zTR && nRight > nT) mask |= bits(corner + 1 + q4, corner + 1 + (nRight >> 2));
    if ((P.flags & SCANF_CIP) && mask) {
      // constrained_intra_pred: only samples of intra CUs (intrapred.cc:612-615)
      const int w4 = (P.width + 3) >> 2;
      uint64_t keep = 0;
      for (uint64_t mm = mask; mm; mm &= mm - 1) {
        const int u = __builtin_ctzll(mm);
        const int xs = u < corner ? xB - 1 : (u == corner ? xB - 1 : xB + 4 * (u - corner - 1));
        const int ys = u < corner ? yB + 2 * nT - 1 - 4 * u : yB - 1;
        if (B.blk_flags[((xs * sbw) >> 2) + ((ys * sbh) >> 2) * w4] & DE265HIP_BLK_INTRA) keep |= 1ull << u;
      }
      mask = keep;
    }
  }
  // -- dependencies: only the units the mode reads (4:4:4 chroma is smoothed like luma: it takes luma's table, a superset)
  const int m = tu.intra_mode < 35 ? tu.intra_mode : 1;
  const uint64_t need = (P.flags & SCANF_MODE_DEPS) ? scan_needed_units(B.used_units[((tu.log2_size - 2) * 35 + m) * 2 + ((c == 0 || P.cf == 3) ? 1 : 0)], mask) : mask;
  B.tu_avail[i] = mask; B.tu_need[i] = need;
  // -- the cells it covers
  const int mw = P.map_w[c];
  ScanCell* cells = B.cell[c];
  for (int y = yB >> 2; y < (yB + nT) >> 2; y++)
    for (int x = xB >> 2; x < (xB + nT) >> 2; x++) cells[x + (size_t)y * mw] = (ScanCell)(uint32_t)(i + 1);
}

// the motion plane of a picture from its PU records (k_lf.hip k_motion_from_pus, the same body): sixteen lanes per PU, sixteen
// PUs per workgroup and step; workgroup `wg` of `n_wg`
__device__ __forceinline__ void motion_from_pus_body(const de265hip_pu* __restrict__ pus, int n_pus, const de265hip_slice_params* __restrict__ slices, int n_slices,
                                                     de265hip_motion* __restrict__ motion, int w4, int h4, int wg, int n_wg)
{
  const int sub = threadIdx.x & 15;
  for (int i = wg * 16 + (threadIdx.x >> 4); i < n_pus; i += n_wg * 16) {
    const de265hip_pu pu = pus[i];
    if (pu.slice_idx >= n_slices) continue;
    de265hip_motion m;
    for (int l = 0; l < 2; l++) {
      const bool on = (pu.pred_flag >> l) & 1;
      const int ri = pu.ref_idx[l];
      m.ref_slot[l] = (on && ri >= 0 && ri < DE265HIP_MAX_REFS) ? slices[pu.slice_idx].ref_pic_list[l][ri] : (int8_t)-1;
      m.mv[l][0] = on ? pu.mv[l][0] : (int16_t)0; m.mv[l][1] = on ? pu.mv[l][1] : (int16_t)0;
    }
    m.pad[0] = m.pad[1] = 0;
    const int bw = pu.w >> 2, bh = pu.h >> 2;
    for (int q = sub; q < bw * bh; q += 16) {
      const int x = (pu.x >> 2) + q % bw, y = (pu.y >> 2) + q / bw;
      if (x < w4 && y < h4) motion[x + y * w4] = m;
    }
  }
}

__global__ __launch_bounds__(256)
void k_scan_tus(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  SCAN_PRIO();
  if ((int)blockIdx.x >= J.tus_blocks) {                    // the workgroups behind the pass's own: the picture's motion plane
    const ScanJob& Q = J.job[blockIdx.y];
    if (Q.mo_plane) motion_from_pus_body(Q.mo_pus, Q.mo_n_pus, Q.mo_slices, Q.mo_n_slices, Q.mo_plane, Q.mo_w4, Q.mo_h4, (int)blockIdx.x - J.tus_blocks, (int)gridDim.x - J.tus_blocks);
    return;
  }
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  if (blockIdx.x == 0 && threadIdx.x == 0) { if (B.err_word) *B.err_word = 0; B.counts->victim = 0xFFFFFFFFu; }      // (the picture's kernels that may raise the word come behind the scan)
  __shared__ uint32_t s_tot[7];                               // this workgroup's share of the list totals (scan_prefix's job until round 4)
  if (threadIdx.x < 7) s_tot[threadIdx.x] = 0;
  __syncthreads();
  uint32_t alg_resid = 0, alg_intra = 0, n_tasks = 0;
  bool bad = false;
  // (a bounded grid that walks the records: see scan_enqueue_batch)
  for (int blk = blockIdx.x; blk * 256 < P.n_tus; blk += J.tus_blocks) {
    const int i = blk * 256 + threadIdx.x;
    scan_tu_dev(P, B, i, i < P.n_tus, alg_resid, alg_intra, n_tasks, s_tot);
    // coefficient positions inside their TU's block (de265hip_picture_build's job on the host until round 3): the workgroup's 256
    // records sixteen at a time, sixteen lanes striding each list (a thread walking its own list alone took 7x as long); a
    // position beyond the block is folded into it and the picture fails
    if (P.flags & SCANF_CHECK_POS) {
      const int sub = threadIdx.x & 15;
      for (int g = 0; g < 16; g++) {
        const int j = blk * 256 + g * 16 + (threadIdx.x >> 4);
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
    }
  }
  if (bad) scan_fail(B, DE265HIP_ERROR_DECODING);
  // one atomic per wavefront and sum
  alg_resid = wave_sum_u(alg_resid); alg_intra = wave_sum_u(alg_intra); n_tasks = wave_sum_u(n_tasks);
  if ((threadIdx.x & 63) == 0) {
    scan_add64(&B.counts->alg_resid, alg_resid); scan_add64(&B.counts->alg_intra, alg_intra);
    if (n_tasks) atomicAdd(&B.counts->n_tasks, n_tasks);
  }
  __syncthreads();
  if (threadIdx.x < 7 && s_tot[threadIdx.x]) {
    const uint32_t v = s_tot[threadIdx.x];
    if (threadIdx.x < 4) atomicAdd(&B.counts->n_l0_size[threadIdx.x], v);
    else if (threadIdx.x == 4) atomicAdd(&B.counts->n_l0_rext, v);
    else if (threadIdx.x == 5) atomicAdd(&B.counts->n_intra, v);
    else scan_add64(&B.counts->n_isamp, v);
  }
}

// The CTB pass, one WAVEFRONT per CTB (scan_core.h scan_ctb is the same pass as one thread's loop: the CPU rehearsal; the
// equivalence test holds this kernel to it).  What is sequential - which run a TU joins depends on the TUs before it - runs
// on a window of the CTB's cells in LDS (its own 16 x 16 cells, the row above with its above-right reach, the column to the
// left): per intra TU one LDS round trip (a lane per neighbour unit), a handful of cross-lane reductions, the decision, one
// LDS write (a lane per covered cell).  The records are fetched 64 at a time, classified a lane each; the level-0 tasks of
// the inter TUs are written by their lanes (positions by ballot prefix).
#define SCW_W 25                         // window columns: cell x in [-1, 23] relative to the CTB
#define SCW_H 17                         // window rows:    cell y in [-1, 15]
#define SCW_NONLOCAL 0x40000000u         // an intra TU of another CTB covers the cell
struct ScanCtbHead { uint32_t first, end, seen, n_intra, take; };      // what a wavefront needs of a CTB's record before anything else (take: lane k's list)
__device__ __forceinline__ void scan_ctb_wave(const ScanParams& P, const ScanBufs& B, const int rs, const int lane, const ScanCtbHead& H, const uint32_t (&cls_start)[4]);
__global__ __launch_bounds__(64)
void k_scan_ctbs(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  SCAN_PRIO();
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  if (B.counts->status) return;
  const int lane = threadIdx.x;
  uint32_t cls_start[4];
  for (int k = 0; k < 4; k++) cls_start[k] = scan_l0_class_start(B.counts->n_l0_size, k);
  // (a bounded grid: every wavefront takes CTBs in turn, see scan_enqueue_batch.  A CTB is a chain of dependent round trips -
  //  its record, its share of the lists, its TU records, their masks -, a few microseconds each next to the other kernels: the
  //  record of the NEXT CTB is fetched while this one is worked on, and inside a CTB everything that depends on the record
  //  alone is requested before anything is waited for)
  auto load_head = [&](int rs) -> ScanCtbHead {
    ScanCtbHead h; const ScanCtb& C = B.ctb[rs];
    h.first = C.first_tu; h.end = C.end_tu; h.seen = C.seen; h.n_intra = C.n_intra;
    h.take = 0;
    if (lane < 4) h.take = C.n_inter[lane] + C.n_ro[lane];
    else if (lane == 4) h.take = C.n_rext_inter + C.n_rext_ro;
    else if (lane == 5) h.take = C.n_intra;
    else if (lane == 6) h.take = C.n_isamp;
    return h;
  };
  ScanCtbHead h = {};
  if ((int)blockIdx.x < P.n_ctbs) h = load_head(blockIdx.x);
  for (int rs = blockIdx.x; rs < P.n_ctbs; rs += gridDim.x) {
    ScanCtbHead hn = {};
    if (rs + (int)gridDim.x < P.n_ctbs) hn = load_head(rs + gridDim.x);
    scan_ctb_wave(P, B, rs, lane, h, cls_start);
    h = hn;
    __syncthreads();
  }
}
__device__ __forceinline__ void scan_ctb_wave(const ScanParams& P, const ScanBufs& B, const int rs, const int lane, const ScanCtbHead& H, const uint32_t (&cls_start)[4])
{
  __shared__ uint32_t win[3][SCW_W * SCW_H];
  __shared__ uint8_t s_ntus[768];
  ScanCtb& C = B.ctb[rs];
  const uint32_t first = H.first, end = H.end, seen = H.seen, n_intra = H.n_intra;
  if (seen == 0) { if (lane == 0) C.n_runs = 0; return; }
  if (seen != 1 || end <= first || end > (uint32_t)P.n_tus || n_intra > 768) {
    if (lane == 0) { C.n_runs = 0; scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); }
    return;
  }
  // ---- this CTB's share of the task lists, the run-ordered TU array (= its sparse run ids) and the residual samples: one
  // atomic per list, a lane each (scan_core.h computes these bases as a prefix over the CTBs in decode order - a launch of
  // its own on the device, 16 us alone and 60-150 us next to the reconstruction kernels, on a chain everything behind it waits
  // for; nothing needs the lists in decode order)
  // (requested now, waited for where they are first used: the list shares, the halo cells, the first 64 TU records with their masks)
  uint32_t got = 0;
  if (lane < 7 && H.take) got = atomicAdd(&B.counts->alloc[lane], H.take);
  const int cx0 = (rs % P.ctbs_w) << P.lc, cy0 = (rs / P.ctbs_w) << P.lc;      // luma origin of the CTB
  bool halo_nonlocal[3] = { false, false, false };
  for (int c = 0; c < (P.cf ? 3 : 1); c++) {
    const int sw = c ? P.subw : 1, sh = c ? P.subh : 1, mw = P.map_w[c], mh = P.map_h[c];
    const int ox4 = (cx0 / sw) >> 2, oy4 = (cy0 / sh) >> 2;
    if (lane < SCW_W + SCW_H - 1) {
      const int wx = lane < SCW_W ? lane - 1 : -1, wy = lane < SCW_W ? -1 : lane - SCW_W;
      const int gx = ox4 + wx, gy = oy4 + wy;
      halo_nonlocal[c] = gx >= 0 && gy >= 0 && gx < mw && gy < mh && (uint32_t)B.cell[c][gx + (size_t)gy * mw] != 0;
    }
  }
  de265hip_tu tu0; memset(&tu0, 0, sizeof(tu0));
  uint64_t av0 = 0, nd0 = 0;
  if (first + lane < end) { tu0 = B.tus[first + lane]; av0 = B.tu_avail[first + lane]; nd0 = B.tu_need[first + lane]; }      // (the masks of a TU that is not intra are never looked at)
  // ---- the window: own cells empty, the halo from the cell map of the per-TU pass
  for (int q = lane; q < 3 * SCW_W * SCW_H; q += 64) (&win[0][0])[q] = 0;
  for (int q = lane; q < 768; q += 64) s_ntus[q] = 0;
  __syncthreads();
  if (lane < SCW_W + SCW_H - 1) {
    const int wx = lane < SCW_W ? lane - 1 : -1, wy = lane < SCW_W ? -1 : lane - SCW_W;
    for (int c = 0; c < 3; c++) if (halo_nonlocal[c]) win[c][(wy + 1) * SCW_W + (wx + 1)] = SCW_NONLOCAL;
  }
  __syncthreads();
  // ---- this CTB's share of the task lists, the run-ordered TU array (= its sparse run ids) and the residual samples: one
  // atomic per list, a lane each (scan_core.h computes these bases as a prefix over the CTBs in decode order - a launch of
  // its own on the device, 16 us alone and 60-150 us next to the reconstruction kernels, on a chain everything behind it waits
  // for; nothing needs the lists in decode order)
  if (lane < 4) C.l0_base[lane] = got;
  else if (lane == 4) C.rext_base = got;
  else if (lane == 5) C.intra_base = got;
  else if (lane == 6) C.isamp_base = got;
  const uint32_t ibase = (uint32_t)__builtin_amdgcn_readlane((int)got, 5), my_rext_base = (uint32_t)__builtin_amdgcn_readlane((int)got, 4);
  uint32_t my_l0_base[4];
  for (int k = 0; k < 4; k++) my_l0_base[k] = (uint32_t)__builtin_amdgcn_readlane((int)got, k);
  uint32_t inter_at[4] = { 0, 0, 0, 0 }, rext_at = 0;
  int cur_run0 = -1, cur_run1 = -1, cur_run2 = -1;      // the current run of each colour component
  int n_local = 0;
  // (kernel arguments the per-TU loop needs, once)
  const int sws = P.subw == 2 ? 1 : 0, shs = P.subh == 2 ? 1 : 0;
  const int ox4_y = cx0 >> 2, oy4_y = cy0 >> 2, ox4_c = (cx0 >> sws) >> 2, oy4_c = (cy0 >> shs) >> 2;
  const int cwid = P.cwid, chei = P.chei, width = P.width, height = P.height;
  const bool merge_on = (P.flags & SCANF_MERGE) != 0;
  uint32_t* const tu_info_out = B.tu_info;
  for (uint32_t base = first; base < end; base += 64) {
    const uint32_t i = base + lane;
    const bool have = i < end;
    de265hip_tu tu = tu0;
    uint64_t av = av0, nd = nd0;
    int cls = 0, rx = 0;
    if (base != first) {
      memset(&tu, 0, sizeof(tu)); av = 0; nd = 0;
      if (have) { tu = B.tus[i]; av = B.tu_avail[i]; nd = B.tu_need[i]; }
    }
    if (have) cls = scan_tu_class(P, B, tu, &rx);
    // -- level-0 tasks of the inter TUs: a lane each, positions by ballot prefix per size class
    for (int k = 0; k < 4; k++) {
      const uint64_t m = __ballot(cls == 1 && tu.log2_size == k + 2);
      if (cls == 1 && tu.log2_size == k + 2) B.l0[cls_start[k] + my_l0_base[k] + inter_at[k] + __popcll(m & lanes_below(lane))] = scan_task_of(tu);
      inter_at[k] += __popcll(m);
    }
    {
      const uint64_t m = __ballot(cls == 2);
      if (cls == 2) {
        TuTask t = scan_task_of(tu);
        uint64_t luma_info = 0; int rx_luma = 0;
        if (rx & D265_RX_XCC) scan_xcc_luma(P, B, (int)i, &luma_info, &rx_luma);
        t.pad3 = (uint8_t)(rx | rx_luma); t.angle = tu.res_scale_val; t.avail = luma_info;
        B.l0x[my_rext_base + rext_at + __popcll(m & lanes_below(lane))] = t;
      }
      rext_at += __popcll(m);
    }
    // -- the intra TUs of the chunk, one after the other (everything that does not change from TU to TU is in registers by now:
    // a lone wavefront pays a dozen cycles per dependent instruction, a scalar load from the kernel's arguments a few hundred)
    const uint32_t pos = (uint32_t)tu.x0 | ((uint32_t)tu.y0 << 16), shape = (uint32_t)tu.log2_size | ((uint32_t)tu.c_idx << 8);
    for (uint64_t im = __ballot(cls == 3); im; im &= im - 1) {
      const int src = __builtin_ctzll(im);
      const uint32_t pos_u = __builtin_amdgcn_readlane(pos, src), shape_u = __builtin_amdgcn_readlane(shape, src);
      const uint64_t mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(av >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)av, src);
      const uint64_t need0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(nd >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)nd, src);
      const int xB = pos_u & 0xFFFF, yB = pos_u >> 16, log2 = shape_u & 0xFF, c = shape_u >> 8, nT = 1 << log2, corner = nT >> 1;
      const int ox4 = c ? ox4_c : ox4_y, oy4 = c ? oy4_c : oy4_y;
      const int wx0 = (xB >> 2) - ox4, wy0 = (yB >> 2) - oy4;                 // the TU's first cell in window coordinates
      uint32_t* W = win[c];
      auto look = [&](int wx, int wy) -> uint32_t {
        return ((unsigned)(wx + 1) < (unsigned)SCW_W && (unsigned)(wy + 1) < (unsigned)SCW_H) ? W[(wy + 1) * SCW_W + (wx + 1)] : 0u;
      };
      // a lane per neighbour unit: left column bottom -> top, corner, top row left -> right
      const bool in_mask = (mask >> lane) & 1, in_need = (need0 >> lane) & 1;
      uint32_t v = 0;
      if (in_mask) {
        const int wx = lane <= corner ? wx0 - 1 : wx0 + (lane - corner - 1);
        const int wy = lane < corner ? wy0 + corner - 1 - lane : wy0 - 1;
        v = look(wx, wy);
      }
      const bool local = v >> 31, nonlocal = v == SCW_NONLOCAL;
      const int vrun = (int)(v & 0xFFFF), vlev = (int)((v >> 16) & 0xFF);
      const int crun = c == 0 ? cur_run0 : (c == 1 ? cur_run1 : cur_run2);
      const bool foreign = __ballot(in_need && !local && !nonlocal) != 0;
      const bool any_nonlocal = __ballot(in_need && nonlocal) != 0;
      const uint64_t loc_m = __ballot(in_need && local);
      const bool reads_cur = __ballot(in_need && local && vrun == crun) != 0;
      int llev = 1;
      if (reads_cur) llev = wave_max_i((in_need && local && vrun == crun) ? vlev : 0) + 1;
      const int p0 = loc_m ? __builtin_amdgcn_readlane(vrun, __builtin_ctzll(loc_m)) : -1;
      int r = crun;
      bool extends = r >= 0 && s_ntus[r] < 255;
      if (extends && !reads_cur) extends = __ballot(in_mask && !in_need && local && vrun == r) != 0;
      bool merged = false;
      if (!extends && merge_on && p0 >= 0 && !any_nonlocal && __ballot(in_need && local && vrun != p0) == 0 && s_ntus[p0] < 255) {
        // in-run level: behind everything of that run in the row above and the column to the left of the TU's neighbourhood
        const int cw = c ? cwid : width, ch = c ? chei : height;
        const int ux0 = (xB - 4 > 0 ? xB - 4 : 0) >> 2, uy0 = (yB - 4 > 0 ? yB - 4 : 0) >> 2;
        const int ux1 = (cw - 1 < xB + 2 * nT + 3 ? cw - 1 : xB + 2 * nT + 3) >> 2, uy1 = (ch - 1 < yB + 2 * nT + 3 ? ch - 1 : yB + 2 * nT + 3) >> 2;
        int lv = 0;
        if (lane < 32) { const int x4 = ux0 + lane; if (uy0 < (yB >> 2) && x4 <= ux1) { const uint32_t q = look(x4 - ox4, uy0 - oy4); if ((q >> 31) && (int)(q & 0xFFFF) == p0) lv = (int)((q >> 16) & 0xFF); } }
        else { const int y4 = uy0 + lane - 32; if (ux0 < (xB >> 2) && y4 <= uy1) { const uint32_t q = look(ux0 - ox4, y4 - oy4); if ((q >> 31) && (int)(q & 0xFFFF) == p0) lv = (int)((q >> 16) & 0xFF); } }
        const int lx = wave_max_i(lv);
        if (lx + 1 <= 250) { r = p0; llev = lx + 1; merged = true; }
      }
      if (!extends && !merged) {
        r = n_local++; llev = 1;
        if (c == 0) cur_run0 = r; else if (c == 1) cur_run1 = r; else cur_run2 = r;
      }
      WAVE_ORDER();                                      // (every lane has read s_ntus and the window)
      if (lane == 0) {
        s_ntus[r]++;
        tu_info_out[base + src] = (uint32_t)r | ((uint32_t)llev << 16) | (foreign ? SCAN_TI_FOREIGN : 0u) | SCAN_TI_INTRA;
      }
      const int l4 = log2 - 2;
      if (lane < (1 << (2 * l4))) W[(wy0 + (lane >> l4) + 1) * SCW_W + (wx0 + (lane & ((1 << l4) - 1)) + 1)] = (uint32_t)r | ((uint32_t)llev << 16) | (1u << 31);
      WAVE_ORDER();
    }
  }
  // ---- the run behind every cell of this CTB that an intra TU covers: sparse id + 1 into the cell's high word (what the run
  // pass resolves producers from: one load per needed unit)
  WAVE_ORDER();
  for (int c = 0; c < (P.cf ? 3 : 1); c++) {
    const int sw = c ? P.subw : 1, sh = c ? P.subh : 1, mw = P.map_w[c], mh = P.map_h[c];
    const int ox4 = (cx0 / sw) >> 2, oy4 = (cy0 / sh) >> 2;
    const int cw4 = ((1 << P.lc) / sw) >> 2, ch4 = ((1 << P.lc) / sh) >> 2;          // the CTB in cells of this component
    for (int q = lane; q < cw4 * ch4; q += 64) {
      const int wx = q % cw4, wy = q / cw4;
      const uint32_t v = win[c][(wy + 1) * SCW_W + (wx + 1)];
      if ((v >> 31) && ox4 + wx < mw && oy4 + wy < mh)
        reinterpret_cast<uint32_t*>(&B.cell[c][(ox4 + wx) + (size_t)(oy4 + wy) * mw])[1] = ibase + (v & 0xFFFFu) + 1u;
    }
  }
  // ---- the CTB's runs: sizes, CTB, a place in the run list
  uint32_t at = 0;
  if (lane == 0) { C.n_runs = (uint32_t)n_local; atomicAdd(&B.counts->n_runs, (uint32_t)n_local); at = atomicAdd(&B.counts->n_listed, (uint32_t)n_local); }
  at = __shfl(at, 0, 64);
  for (int q = lane; q < n_local; q += 64) { B.run_ntus[ibase + q] = s_ntus[q]; B.run_rs[ibase + q] = (uint32_t)rs; B.run_list[at + q] = ibase + (uint32_t)q; }
}

// inclusive prefix sum over the lanes of the wavefront (Hillis-Steele inside each row of 16 by DPP row shifts, then the row
// totals by row broadcasts: lanes a row mask leaves out add the `old` operand, zero)
__device__ __forceinline__ uint32_t wave_scan_incl_u(uint32_t x)
{
  int v = (int)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
  return (uint32_t)v;
}

// The run pass, one WAVEFRONT per CTB: all runs of the CTB together (scan_core.h scan_run is the same pass as one thread's loop
// over one run).  Round 4's first form gave every run a wavefront of its own: ~2 200 instructions of fixed cost per run -
// reductions, sorts, a hash table - for runs of two or three TUs, 19 M wavefront instructions per 4K B picture, as much as the
// picture's motion compensation.  Here the work follows the CTB's intra TUs (a lane each, 64 at a time): per-run sums by LDS
// atomics into a table of the CTB's runs (64 runs at a time: a tile), a TU's place in its run's chain order by counting the
// TUs of its run that sort before it (two loops over the CTB's TUs in LDS), the producers by one load per needed unit (the
// cell's high word: the sparse id of the run that covers it, left there by the CTB pass) into one hash set per tile.
#define SR_TMAX 768                      // intra TUs of a CTB (k_scan_ctbs refuses more; 4:2:0 and monochrome: at most 384)
#define SR_HASH 512
#define SR_TILE 32                       // runs per tile.  LDS per workgroup: 10 KB (4:2:0) / 13 KB - what decides how many CTBs
                                         // are worked on at once next to k_run's 53 KB workgroups (with 21 KB: two per CU)
enum { RF_FOREIGN = 1, RF_BIG = 2, RF_TOO_BIG = 4, RF_BAD = 8, RF_DENSE0 = 16, RF_MICRO = 32, RF_DENSE = 64, RF_MB = 128, RF_PHASED = 256 };
struct ScanRunsHead { uint32_t n_runs, first, end, ibase, isamp_base, rext_base, n_rext_inter, l0_at[4]; };
template <int TMAX> __device__ void scan_runs1_wave(const ScanParams& P, const ScanBufs& B, const int rs, const int lane, const ScanRunsHead& H, const uint32_t (&cls_start)[4]);
template <int TMAX>
__global__ __launch_bounds__(64)
void k_scan_runs1(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  SCAN_PRIO();
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  uint32_t cls_start[4];
  for (int k = 0; k < 4; k++) cls_start[k] = scan_l0_class_start(B.counts->n_l0_size, k);
  // (the next CTB's record is fetched while this one is worked on: see k_scan_ctbs)
  auto load_head = [&](int rs) -> ScanRunsHead {
    ScanRunsHead h; const ScanCtb& C = B.ctb[rs];
    h.n_runs = C.n_runs; h.first = C.first_tu; h.end = C.end_tu; h.ibase = C.intra_base; h.isamp_base = C.isamp_base; h.rext_base = C.rext_base; h.n_rext_inter = C.n_rext_inter;
    for (int k = 0; k < 4; k++) h.l0_at[k] = C.l0_base[k] + C.n_inter[k];
    return h;
  };
  ScanRunsHead h = {};
  if ((int)blockIdx.x < P.n_ctbs) h = load_head(blockIdx.x);
  for (int rs = blockIdx.x; rs < P.n_ctbs; rs += gridDim.x) {
    ScanRunsHead hn = {};
    if (rs + (int)gridDim.x < P.n_ctbs) hn = load_head(rs + gridDim.x);
    if (B.counts->status) return;
    if (h.n_runs) { scan_runs1_wave<TMAX>(P, B, rs, threadIdx.x, h, cls_start); __syncthreads(); }
    h = hn;
  }
}
template <int TMAX>
__device__ void scan_runs1_wave(const ScanParams& P, const ScanBufs& B, const int rs, const int lane, const ScanRunsHead& H, const uint32_t (&cls_start)[4])
{
  __shared__ uint32_t s_w[TMAX];                       // per intra TU of the CTB (decode order): run | level << 10 | (log2 - 2) << 18 | residual-only task << 20 | rext << 21 | foreign << 22
  __shared__ uint16_t s_key[TMAX];                     // list << 8 | level (ties inside a run: decode order)
  __shared__ uint16_t s_ix[TMAX];                      // its record, relative to the CTB's first
  // the tile's runs
  __shared__ uint32_t r_x0[SR_TILE], r_y0[SR_TILE], r_x1[SR_TILE], r_y1[SR_TILE], r_wx1[SR_TILE], r_wy1[SR_TILE], r_n[SR_TILE], r_samp[SR_TILE], r_nl[SR_TILE], r_fl[SR_TILE], r_c[SR_TILE];
  __shared__ uint32_t r_ro[4][SR_TILE], r_rx[SR_TILE], r_alg[SR_TILE], r_cnt[5][SR_TILE], r_nd[SR_TILE], r_fill[SR_TILE];
  __shared__ uint32_t r_first[SR_TILE], r_res[SR_TILE], r_robase[4][SR_TILE], r_rxbase[SR_TILE], r_depoff[SR_TILE], r_mb[SR_TILE];
  __shared__ uint32_t s_tab[SR_HASH];
  __shared__ uint8_t s_rdy[64];
  const uint32_t n_runs = H.n_runs;
  const uint32_t first = H.first, end = H.end, ibase = H.ibase, isamp_base = H.isamp_base, rext_base = H.rext_base, n_rext_inter = H.n_rext_inter;
  uint32_t l0_at[4];                                   // where the residual-only copies of this CTB's intra TUs start, per size class
  for (int k = 0; k < 4; k++) l0_at[k] = cls_start[k] + H.l0_at[k];
  const uint32_t pflags = P.flags;
  const int micro_tus = P.micro_tus, run_waves = P.run_waves;
  if (end - first > 65535u || n_runs > (uint32_t)TMAX) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); return; }
  // ---- the CTB's intra TUs, compacted (decode order)
  int T = 0;
  for (uint32_t base = first; base < end; base += 64) {
    const uint32_t i = base + lane;
    uint32_t ti = 0; de265hip_tu tu; memset(&tu, 0, sizeof(tu));
    if (i < end) { ti = B.tu_info[i]; tu = B.tus[i]; }       // (the record with its info word: one round trip, not two)
    const bool intra = ti & SCAN_TI_INTRA;
    const uint64_t m = __ballot(intra);
    if (intra) {
      const int t = T + __popcll(m & lanes_below(lane));
      if (t < TMAX) {
        const int trx = scan_rx_bits(P, B, tu);
        const bool ro = ((tu.flags & DE265HIP_TU_CBF) && tu.n_coeff) || (trx & D265_RX_XCC);
        s_ix[t] = (uint16_t)(i - first);
        s_w[t] = SCAN_TI_RUN(ti) | (SCAN_TI_LLEV(ti) << 10) | ((uint32_t)(tu.log2_size - 2) << 18) | (ro ? 1u << 20 : 0u) | (trx ? 1u << 21 : 0u) | ((ti & SCAN_TI_FOREIGN) ? 1u << 22 : 0u);
      }
    }
    T += __popcll(m);
  }
  if (T > TMAX) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); return; }
  WAVE_ORDER();
  uint32_t carry_n = 0, carry_samp = 0, carry_ro[4] = { 0, 0, 0, 0 }, carry_rx = 0;
  uint64_t tick_acc = 0, mic_acc = 0;
  for (uint32_t r0 = 0; r0 < n_runs; r0 += SR_TILE) {
    const uint32_t n_tile = n_runs - r0 < (uint32_t)SR_TILE ? n_runs - r0 : (uint32_t)SR_TILE;
    const bool have_run = (uint32_t)lane < n_tile;
    const int rl = lane & (SR_TILE - 1);                                   // (the lanes beyond the tile mirror its entries: they only read)
    const uint32_t s = ibase + r0 + (uint32_t)lane;                      // the sparse id of this lane's run
    // ---- per-run sums
    if (lane < SR_TILE) {
      r_x0[lane] = 0xFFFFu; r_y0[lane] = 0xFFFFu; r_x1[lane] = 0; r_y1[lane] = 0; r_wx1[lane] = 0; r_wy1[lane] = 0; r_n[lane] = 0; r_samp[lane] = 0;
      r_nl[lane] = 0; r_fl[lane] = 0; r_c[lane] = 0; r_rx[lane] = 0; r_alg[lane] = 0; r_nd[lane] = 0; r_fill[lane] = 0;
      for (int k = 0; k < 4; k++) r_ro[k][lane] = 0;
      for (int k = 0; k < 5; k++) r_cnt[k][lane] = 0;
    }
    for (int q = lane; q < SR_HASH; q += 64) s_tab[q] = 0xFFFFFFFFu;
    WAVE_ORDER();
    for (int cb = 0; cb < T; cb += 64) {
      const int t = cb + lane;
      if (t >= T) continue;
      const uint32_t w = s_w[t], jj = (w & 1023u) - r0;
      if (jj >= (uint32_t)SR_TILE) continue;
      const de265hip_tu tu = B.tus[first + s_ix[t]];
      const uint32_t nT = 1u << tu.log2_size, bpp = (uint32_t)(tu.c_idx ? P.bppC : P.bppY);
      atomicMin(&r_x0[jj], (uint32_t)tu.x0); atomicMin(&r_y0[jj], (uint32_t)tu.y0);
      atomicMax(&r_x1[jj], tu.x0 + nT); atomicMax(&r_y1[jj], tu.y0 + nT); atomicMax(&r_wx1[jj], tu.x0 + 2 * nT); atomicMax(&r_wy1[jj], tu.y0 + 2 * nT);
      atomicAdd(&r_n[jj], 1u); atomicAdd(&r_samp[jj], nT * nT); atomicMax(&r_nl[jj], (w >> 10) & 0xFFu); atomicMax(&r_c[jj], (uint32_t)tu.c_idx);
      atomicAdd(&r_alg[jj], bpp * (4 * nT + 1) + bpp * nT * nT);
      const uint32_t fl = ((w >> 22) & 1u ? RF_FOREIGN : 0u) | (tu.log2_size == 4 ? RF_BIG : 0u) | (tu.log2_size > 4 ? RF_TOO_BIG : 0u);
      if (fl) atomicOr(&r_fl[jj], fl);
      if ((w >> 20) & 1u) { if ((w >> 21) & 1u) atomicAdd(&r_rx[jj], 1u); else atomicAdd(&r_ro[tu.log2_size - 2][jj], 1u); }
    }
    WAVE_ORDER();
    // ---- a lane per run: class
    int x0 = (int)r_x0[rl], y0 = (int)r_y0[rl], x1 = (int)r_x1[rl], y1 = (int)r_y1[rl], wx1 = (int)r_wx1[rl], wy1 = (int)r_wy1[rl];
    const int n = (int)r_n[rl], own_samples = (int)r_samp[rl], nl = (int)r_nl[rl], c = (int)r_c[rl];
    uint32_t fl = r_fl[rl];
    if (__ballot(have_run && (n == 0 || n > 255 || n != (int)B.run_ntus[have_run ? s : ibase])) != 0) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); return; }
    if (__ballot(have_run && nl > 256) != 0) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
    bool micro = have_run && !(pflags & SCANF_MICRO_OFF) && n <= micro_tus && x1 - x0 <= 32 && y1 - y0 <= 32 && !(fl & RF_TOO_BIG);
    if (micro && (fl & RF_BIG)) {
      if (!(pflags & SCANF_MICRO16) || own_samples > 1024) micro = false;
      else {
        const int ax0 = (x0 - 1) & ~7, wxc = wx1 < x1 + 32 ? wx1 : x1 + 32, wyc = wy1 < y1 + 32 ? wy1 : y1 + 32;
        const int cols = wxc - ax0, rows = wyc - (y0 - 1);
        micro = cols <= 56 && rows <= 41 && ((cols + 7) >> 3) * rows <= 256;
      }
    }
    const bool dense0 = have_run && own_samples == (x1 - x0) * (y1 - y0) && !(pflags & SCANF_NO_DENSE);
    if (lane < SR_TILE) r_fl[lane] = fl | (micro ? RF_MICRO : 0u) | (dense0 ? RF_DENSE0 : 0u);
    WAVE_ORDER();
    // dense: every available neighbour outside the box lies on the row above it or the column left of it (a lane per TU)
    if (__ballot(dense0) != 0) {
      for (int cb = 0; cb < T; cb += 64) {
        const int t = cb + lane;
        if (t >= T) continue;
        const uint32_t w = s_w[t], jj = (w & 1023u) - r0;
        if (jj >= (uint32_t)SR_TILE || !(r_fl[jj] & RF_DENSE0)) continue;
        const uint32_t i = first + s_ix[t];
        const de265hip_tu tu = B.tus[i];
        const uint64_t avail = B.tu_avail[i];
        const int nT = 1 << tu.log2_size, xB = tu.x0, yB = tu.y0, corner = nT >> 1, bx0 = (int)r_x0[jj], by0 = (int)r_y0[jj], bx1 = (int)r_x1[jj], by1 = (int)r_y1[jj];
        bool bad = false;
        if (xB > bx0 && yB + 2 * nT > by1) {
          int umax = (yB + 2 * nT - 1 - by1) >> 2; if (umax > corner - 1) umax = corner - 1;
          if (avail & ((2ull << umax) - 1ull)) bad = true;
        }
        if (yB > by0 && xB + 2 * nT > bx1) {
          int kmin = (bx1 - xB) >> 2; if (kmin < 0) kmin = 0;
          if (kmin < corner && ((avail >> (corner + 1 + kmin)) & ((1ull << (corner - kmin)) - 1ull))) bad = true;
        }
        if (bad) atomicOr(&r_fl[jj], (uint32_t)RF_BAD);
      }
      WAVE_ORDER();
    }
    fl = r_fl[rl];
    const bool dense = dense0 && !(fl & RF_BAD);
    // ---- what the runs before it take of the CTB's lists: prefix over the runs (three packed sums), carried across tiles
    const uint32_t pa = have_run ? ((uint32_t)n | ((uint32_t)own_samples << 16)) : 0u;
    const uint32_t pb = have_run ? (r_ro[0][rl] | (r_ro[1][rl] << 10) | (r_ro[2][rl] << 20)) : 0u;
    const uint32_t pc = have_run ? (r_ro[3][rl] | (r_rx[rl] << 16)) : 0u;
    const uint32_t ia = wave_scan_incl_u(pa), ib = wave_scan_incl_u(pb), ic = wave_scan_incl_u(pc);
    const uint32_t ea = ia - pa, eb = ib - pb, ec = ic - pc;
    const uint32_t first_tu = ibase + carry_n + (ea & 0xFFFFu), res_offset = isamp_base + carry_samp + (ea >> 16);
    if (lane < SR_TILE) {
      r_first[lane] = first_tu; r_res[lane] = res_offset;
      r_robase[0][lane] = l0_at[0] + carry_ro[0] + (eb & 1023u); r_robase[1][lane] = l0_at[1] + carry_ro[1] + ((eb >> 10) & 1023u);
      r_robase[2][lane] = l0_at[2] + carry_ro[2] + ((eb >> 20) & 1023u); r_robase[3][lane] = l0_at[3] + carry_ro[3] + (ec & 0xFFFFu);
      r_rxbase[lane] = rext_base + n_rext_inter + carry_rx + (ec >> 16);
    }
    {
      const uint32_t ta = (uint32_t)__builtin_amdgcn_readlane((int)ia, 63), tb = (uint32_t)__builtin_amdgcn_readlane((int)ib, 63), tc = (uint32_t)__builtin_amdgcn_readlane((int)ic, 63);
      carry_n += ta & 0xFFFFu; carry_samp += ta >> 16; carry_ro[0] += tb & 1023u; carry_ro[1] += (tb >> 10) & 1023u; carry_ro[2] += (tb >> 20) & 1023u;
      carry_ro[3] += tc & 0xFFFFu; carry_rx += tc >> 16;
    }
    // ---- mailbox of an ordinary dense run
    uint32_t mb_id = 0xFFFFFFFFu;
    bool phased = false;
    {
      const bool want = have_run && (pflags & SCANF_MAILBOX) && !micro && dense;
      const uint64_t wm = __ballot(want);
      if (wm) {
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(&B.counts->n_mailboxes, (uint32_t)__popcll(wm));
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (want) {
          mb_id = at + (uint32_t)__popcll(wm & lanes_below(lane));
          if (mb_id >= P.cap_mb) mb_id = 0xFFFFFFFFu;        // (beyond the mailboxes there are: the run does without)
          else phased = (pflags & SCANF_MB_PHASES) && c == 0 && x1 - x0 <= 64 && y1 - y0 <= 64;
        }
      }
    }
    if (lane < SR_TILE) { r_mb[lane] = mb_id; r_fl[lane] = fl | (dense ? RF_DENSE : 0u) | (phased ? RF_PHASED : 0u); }
    WAVE_ORDER();
    // ---- chain order.  First loop: the TU's number inside its run (decode order) and its rank among the non-collective TUs of
    // its in-run level before it -> its list; second loop: how many TUs of its run sort before it, their samples, and how many
    // residual-only copies of its class
    for (int cb = 0; cb < T; cb += 64) {
      const int t = cb + lane;
      uint32_t key = 0xFFFFu;
      const uint32_t w = t < T ? s_w[t] : 0xFFFFFFFFu, jj = (w & 1023u) - r0;
      const bool mine = t < T && jj < (uint32_t)SR_TILE;
      const bool rmicro = mine && (r_fl[mine ? jj : 0] & RF_MICRO);
      if (__ballot(mine) != 0) {
        uint32_t rank = 0;
        const int u_end = cb + 64 < T ? cb + 64 : T;
        for (int u = 0; u < u_end; u++) {
          const uint32_t wu = s_w[u];
          if (u < t && ((wu ^ w) & 1023u) == 0) {
            const bool coll_u = ((wu >> 18) & 3u) > 1u && !rmicro;
            if (((wu ^ w) & (0xFFu << 10)) == 0 && !coll_u) rank++;
          }
        }
        if (mine) {
          const bool coll = ((w >> 18) & 3u) > 1u && !rmicro;
          const uint32_t list = coll ? 4u : rank % (uint32_t)(rmicro ? 1 : run_waves);
          key = (list << 8) | ((w >> 10) & 0xFFu);
          atomicAdd(&r_cnt[list][jj], 1u);
        }
      }
      if (t < T && mine) s_key[t] = (uint16_t)key;       // (the entries of other tiles' TUs are never compared: the run number differs)
    }
    WAVE_ORDER();
    for (int cb = 0; cb < T; cb += 64) {
      const int t = cb + lane;
      const uint32_t w = t < T ? s_w[t] : 0xFFFFFFFFu, jj = (w & 1023u) - r0;
      const bool mine = t < T && jj < (uint32_t)SR_TILE;
      if (__ballot(mine) == 0) continue;
      const uint32_t key = mine ? s_key[t] : 0u;
      uint32_t pos = 0, samp = 0, ro_rank = 0;
      for (int u = 0; u < T; u++) {
        const uint32_t wu = s_w[u], ku = s_key[u];
        if (((wu ^ w) & 1023u) == 0 && (ku < key || (ku == key && u < t))) {
          pos++; samp += 16u << (2 * ((wu >> 18) & 3u));
          // the same list of residual-only tasks: both with a range-extension tool, or both without and of one size
          if (((wu >> 20) & 1u) && ((wu ^ w) & (1u << 21)) == 0 && (((w >> 21) & 1u) || ((wu ^ w) & (3u << 18)) == 0)) ro_rank++;
        }
      }
      if (!mine) continue;
      // ---- the run-ordered TU record + the residual-only copy (level-0 task)
      const uint32_t i = first + s_ix[t];
      const de265hip_tu tu = B.tus[i];
      TuTask tt = scan_task_of(tu);
      const int m = tu.intra_mode < 35 ? tu.intra_mode : 1;
      tt.angle = (int8_t)scan_intra_angle(m); tt.inv_angle = (int16_t)scan_inv_angle(m);
      tt.avail = B.tu_avail[i];
      tt.run_level = (uint8_t)(((w >> 10) & 0xFFu) - 1);
      const uint32_t coeff_offset = tt.coeff_offset;
      tt.resid_offset = r_res[jj] + samp;
      tt.coeff_offset = samp;
      if ((w >> 20) & 1u) {
        TuTask ro = tt; ro.flags |= D265_TU_RESID_ONLY; ro.coeff_offset = coeff_offset; ro.run_level = 0;
        if ((w >> 21) & 1u) {
          const int trx = scan_rx_bits(P, B, tu);
          uint64_t luma_info = 0; int rx_luma = 0;
          if (trx & D265_RX_XCC) scan_xcc_luma(P, B, (int)i, &luma_info, &rx_luma);
          ro.pad3 = (uint8_t)(trx | rx_luma); ro.angle = 0; ro.avail = 0;
          if (trx & D265_RX_XCC) { ro.angle = tu.res_scale_val; ro.avail = luma_info; }
          B.l0x[r_rxbase[jj] + ro_rank] = ro;
        } else B.l0[r_robase[tu.log2_size - 2][jj] + ro_rank] = ro;
        tt.flags |= DE265HIP_TU_CBF;                      // (the run kernels read the residual block whenever there is one)
      }
      B.run_tus[r_first[jj] + pos] = tt;
      // ---- its producers: the run behind every needed unit (the high word of the cell: sparse id + 1, k_scan_ctbs), each
      // (run, producer) pair once
      const ScanCell* cells = B.cell[tu.c_idx];
      const int mw = P.map_w[tu.c_idx], nT = 1 << tu.log2_size;
      const uint32_t own = ibase + (w & 1023u);
      for (uint64_t need = B.tu_need[i]; need;) {
        uint32_t ps[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          ps[q] = 0;
          if (need) { ps[q] = (uint32_t)(cells[scan_cell_of(__builtin_ctzll(need), tu.x0, tu.y0, nT, mw)] >> 32); need &= need - 1; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (ps[q] == 0 || ps[q] - 1 == own) continue;
          const uint32_t e = (jj << 24) | (ps[q] - 1);
          uint32_t hsh = (e * 2654435761u) >> 23;
          int probe = 0;
          for (; probe < SR_HASH; probe++, hsh = (hsh + 1) & (SR_HASH - 1)) {
            const uint32_t old = atomicCAS(&s_tab[hsh], 0xFFFFFFFFu, e);
            if (old == 0xFFFFFFFFu) { atomicAdd(&r_nd[jj], 1u); break; }
            if (old == e) break;
          }
          if (probe == SR_HASH) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED);
        }
      }
    }
    WAVE_ORDER();
    if (B.counts->status) return;
    // ---- the producer lists: room from the pool (one request per tile), the table's entries to their runs
    const uint32_t nd = have_run ? r_nd[rl] : 0u;
    {
      const uint32_t incl = wave_scan_incl_u(nd), total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      uint32_t at = 0;
      if (total) {
        if (lane == 0) at = atomicAdd(&B.counts->n_deps_alloc, total);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (at + total > P.cap_deps) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
      }
      if (lane < SR_TILE) r_depoff[lane] = at + incl - nd;
      WAVE_ORDER();
      if (total)
        for (int q = lane; q < SR_HASH; q += 64) {
          const uint32_t e = s_tab[q];
          if (e == 0xFFFFFFFFu) continue;
          const uint32_t jj = e >> 24;
          B.deps[r_depoff[jj] + atomicAdd(&r_fill[jj], 1u)] = e & 0xFFFFFFu;
        }
    }
    // ---- ready epochs of the edge packets of a run that publishes in phases (a luma run of at most 64 x 64, dense)
    for (uint64_t pm = __ballot(phased); pm; pm &= pm - 1) {
      const int jj = __builtin_ctzll(pm);
      const int bx0 = (int)r_x0[jj], by0 = (int)r_y0[jj], bx1 = (int)r_x1[jj], by1 = (int)r_y1[jj];
      s_rdy[lane] = 255;
      WAVE_ORDER();
      for (int cb = 0; cb < T; cb += 64) {
        const int t = cb + lane;
        if (t >= T) continue;
        const uint32_t w = s_w[t];
        if ((w & 1023u) - r0 != (uint32_t)jj) continue;
        const de265hip_tu tu = B.tus[first + s_ix[t]];
        const int nT = 1 << tu.log2_size;
        const uint8_t ep = (uint8_t)(((w >> 10) & 0xFFu) - 1);
        if (tu.y0 + nT == by1) for (int q = 0; q < (nT >> 1); q++) s_rdy[((tu.x0 - bx0) >> 1) + q] = ep;
        if (tu.x0 + nT == bx1) for (int q = 0; q < (nT >> 1); q++) s_rdy[32 + ((tu.y0 - by0) >> 1) + q] = ep;
      }
      WAVE_ORDER();
      B.rdy_tab[64 * (size_t)r_mb[jj] + lane] = s_rdy[lane];
      WAVE_ORDER();
    }
    // ---- the run records
    const bool front = have_run && micro && nd == 0 && !(pflags & SCANF_FRONT_OFF);
    if (have_run) {
      RunTask o;
      o.x0 = (uint16_t)x0; o.y0 = (uint16_t)y0; o.x1 = (uint16_t)x1; o.y1 = (uint16_t)y1;
      o.wx1 = (uint16_t)(wx1 < x1 + 32 ? wx1 : x1 + 32); o.wy1 = (uint16_t)(wy1 < y1 + 32 ? wy1 : y1 + 32);
      o.c_idx = (uint8_t)c; o.micro = (uint8_t)((micro ? 1 : 0) | (dense ? 2 : 0) | (front ? RUN_MICRO_FRONT : 0)); o.n_tus = (uint16_t)n;
      o.first_tu = first_tu; o.res_offset = res_offset;
      o.dep_offset = nd ? r_depoff[rl] : 0u; o.n_deps = (uint16_t)nd;
      uint32_t acc = 0;
      for (int wv = 0; wv < 4; wv++) { acc += r_cnt[wv][rl]; o.wave_end[wv] = (uint16_t)acc; }
      o.n_lvls = (uint16_t)(nl > 0 ? nl - 1 : 0);
      o.n_samples = (uint32_t)own_samples;
      B.runs[s] = o;
      B.mbx[3 * (size_t)s] = mb_id; B.mbx[3 * (size_t)s + 1] = 0xFFFFFFFFu; B.mbx[3 * (size_t)s + 2] = 0xFFFFFFFFu;
      B.pub_flag[s] = 0;
      B.run_nall[s] = nd | ((fl & RF_FOREIGN) ? 0x80000000u : 0u);
    }
    {
      const uint64_t fm = __ballot(front);
      uint32_t at = 0;
      if (fm) {
        if (lane == 0) at = atomicAdd(&B.counts->n_front, (uint32_t)__popcll(fm));
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (front) B.front_idx[at + (uint32_t)__popcll(fm & lanes_below(lane))] = s;
      }
      // what the ticket pass needs of this CTB's runs, in two words (more than 64 runs: it reads the run records)
      if (r0 < 64) { tick_acc |= __ballot(have_run && !front) << r0; mic_acc |= __ballot(have_run && micro) << r0; }
      const uint32_t alg = wave_sum_u(front ? r_alg[rl] : 0u), lv = wave_sum_u(have_run ? (uint32_t)nl : 0u);
      if (lane == 0) { if (alg) scan_add64(&B.counts->alg_intra_front, alg); atomicAdd(&B.counts->sum_lvls, lv); }
    }
    WAVE_ORDER();
  }
  if (lane == 0) { B.ctb[rs].tick_mask = tick_acc; B.ctb[rs].micro_mask = mic_acc; B.ctb[rs].slow = n_runs > 64 ? 1u : 0u; }
}

// The second run pass, a wavefront per run: the run's chain-ordered TU records are staged in LDS by all lanes, then ONE lane
// runs scan_run2 on them (producers that are front runs leave the list; mailbox segments and need epochs of a reader).  The
// logic is a few thousand scalar steps on ~40 records: not worth spreading over lanes, but on records in LDS it takes ~15 us
// instead of the ~500 us a thread took that fetched them one by one from memory next to 63 others doing the same.
#define SCR_MAX 256
__global__ __launch_bounds__(64)
void k_scan_runs2(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  SCAN_PRIO();
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  const int lane = threadIdx.x;
  if (B.counts->status) return;
  const uint32_t n_listed = B.counts->n_listed;
  uint32_t n_ready = 0;
  // A lane per run for what every run needs (front runs leave its producer list: a handful of dependent loads - 25 runs in a row
  // per wavefront took 460 us next to the other kernels); the runs that may become mailbox readers (dense, ordinary, few
  // producers: the CTB runs of an all-intra picture) one after the other with the wavefront staging their TU records
  for (uint32_t base = blockIdx.x * 64; base < n_listed; base += gridDim.x * 64) {
    const uint32_t q = base + lane;
    const bool have = q < n_listed;
    uint32_t s = 0, mic = 0, n = 0, n_all = 0; bool foreign = false;
    if (have) {
      s = B.run_list[q];
      const RunTask* R = B.runs + s;
      mic = R->micro; n = R->n_tus;
      const uint32_t na = B.run_nall[s];
      n_all = na & 0x7FFFFFFFu; foreign = na >> 31;
    }
    // (only a run that can become a mailbox reader looks at its TU records)
    const bool cand = have && (P.flags & SCANF_MAILBOX) && !(mic & 1) && (mic & 2) && !foreign && n_all > 0 && n_all <= 8 && n <= SCR_MAX;
    if (have && !cand) {
      scan_run2(P, B, s, nullptr);
      if (!(mic & RUN_MICRO_FRONT) && B.runs[s].n_deps == 0) n_ready++;      // (a ticketed run that waits for nothing: scan_order's worker count)
    }
    // (the candidates: a wavefront each, in k_scan_runs2b - here, one after the other behind a lane, an all-intra picture's
    //  6 000 of them took 2.8 ms)
    const uint64_t cm = __ballot(cand);
    if (cm) {
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(&B.counts->n_cand, (uint32_t)__popcll(cm));
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
      if (cand) B.run_level[at + (uint32_t)__popcll(cm & lanes_below(lane))] = s;
    }
  }
  n_ready = wave_sum_u(n_ready);
  if (lane == 0 && n_ready) atomicAdd(&B.counts->n_ready, n_ready);
}

// ... and the runs that may become mailbox readers, a wavefront per run: its chain-ordered TU records staged in LDS by all
// lanes, then ONE lane runs scan_run2 on them (a few thousand scalar steps on ~40 records)
__global__ __launch_bounds__(64)
void k_scan_runs2b(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  SCAN_PRIO();
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  __shared__ TuTask s_tus[SCR_MAX];
  __shared__ uint32_t s_need[81], s_dl[8], s_seg[32], s_segrun[16], s_nb[16], s_seen[8], s_sub[96];
  __shared__ uint8_t s_nrow[168], s_ncol[168], s_subg[48];
  const int lane = threadIdx.x;
  const bool by_one_lane = J.pad & 1;
  if (B.counts->status) return;
  const uint32_t n_cand = B.counts->n_cand;
  uint32_t n_ready = 0;
  for (uint32_t q = blockIdx.x; q < n_cand; q += gridDim.x) {
    const uint32_t s = B.run_level[q];
    const uint32_t n = B.runs[s].n_tus, first = B.runs[s].first_tu;
    const uint4* srcp = reinterpret_cast<const uint4*>(B.run_tus + first);
    uint4* dst = reinterpret_cast<uint4*>(s_tus);
    for (uint32_t k = lane; k < 2 * n && k < 2 * SCR_MAX; k += 64) dst[k] = srcp[k];
    // when is each neighbour unit of the box first needed?  Only the TUs on the box's left column and top row read outside it
    // (a dense run): a lane per TU (scan_run2's loop over them, half of its ~15 000 scalar steps, as one thread's work)
    if (lane < 40) { s_need[lane] = 255; s_need[40 + lane] = 255; }
    if (lane == 0) s_need[80] = 255;
    WAVE_ORDER();
    {
      const RunTask o = B.runs[s];
      if ((P.flags & SCANF_MB_PHASES) && o.c_idx == 0 && (int)o.n_lvls + 1 >= 4)
        for (uint32_t k = lane; k < n && k < SCR_MAX; k += 64) {
          const TuTask tt = s_tus[k];
          const int xB = tt.x0, yB = tt.y0;
          if (xB != (int)o.x0 && yB != (int)o.y0) continue;
          const int nT = 1 << tt.log2_size, corner = nT >> 1, m = tt.intra_mode < 35 ? tt.intra_mode : 1;
          const uint32_t ep = tt.run_level;
          uint64_t need = (P.flags & SCANF_MODE_DEPS) ? scan_needed_units(B.used_units[((tt.log2_size - 2) * 35 + m) * 2 + 1], tt.avail) : tt.avail;
          for (; need; need &= need - 1) {
            const int u = __builtin_ctzll(need);
            if (u < corner) {
              if (xB != (int)o.x0) continue;
              const int j = (yB + 2 * nT - 4 * u - 4 - (int)o.y0) >> 2;
              if (j >= 0 && j < 40) atomicMin(&s_need[40 + j], ep);
            } else if (u == corner) {
              if (yB == (int)o.y0) { if (xB == (int)o.x0) atomicMin(&s_need[80], ep); else { const int j = (xB - 1 - (int)o.x0) >> 2; if (j < 40) atomicMin(&s_need[j], ep); } }
              else if (xB == (int)o.x0) { const int j = (yB - 1 - (int)o.y0) >> 2; if (j >= 0 && j < 40) atomicMin(&s_need[40 + j], ep); }
            } else {
              if (yB != (int)o.y0) continue;
              const int j = (xB + 4 * (u - corner - 1) - (int)o.x0) >> 2;
              if (j >= 0 && j < 40) atomicMin(&s_need[j], ep);
            }
          }
        }
    }
    WAVE_ORDER();
    if (by_one_lane) {                                   // (DE265HIP_SCAN_RUN2_LANE0=1: scan_core.h's loop on one lane, the parity variant)
      if (lane == 0) { scan_run2(P, B, s, s_tus, s_need); if (B.runs[s].n_deps == 0) n_ready++; }
      WAVE_ORDER();
      continue;
    }
    // ---- scan_core.h scan_run2 for a candidate, by the wavefront (a lane per producer, per segment, per neighbour sample):
    // on one lane its ~8 000 remaining scalar steps on private arrays in scratch memory were 190 us per run, 570 us of an
    // all-intra 4K picture's scan
    const RunTask o = B.runs[s];
    const uint32_t n_all = B.run_nall[s] & 0x7FFFFFFFu;                 // (a candidate: 1..8, not foreign, dense, ordinary)
    uint32_t* dl = B.deps + o.dep_offset;
    uint32_t p = 0; bool front = false;
    if ((uint32_t)lane < n_all) { p = dl[lane]; front = B.runs[p].micro & RUN_MICRO_FRONT; }
    const uint64_t keep_m = __ballot((uint32_t)lane < n_all && !front), front_m = __ballot((uint32_t)lane < n_all && front);
    const uint32_t nd = (uint32_t)__popcll(keep_m);
    if ((uint32_t)lane < n_all) {
      const uint32_t pos = front ? nd + (uint32_t)__popcll(front_m & lanes_below(lane)) : (uint32_t)__popcll(keep_m & lanes_below(lane));
      dl[pos] = p; s_dl[pos] = p;
    }
    if ((P.flags & SCANF_DROP_PRODUCER) && nd) {           // fault injection: the smallest run somebody waits for is never executed
      const int v = wave_min_i(((uint32_t)lane < n_all && !front) ? (int)p : 0x7FFFFFFF);
      if (lane == 0) atomicMin(&B.counts->victim, (uint32_t)v);
    }
    WAVE_ORDER();
    bool reader = false;
    uint32_t new_micro = o.micro;
    if ((P.flags & SCANF_MAILBOX) && nd > 0 && nd == n_all) {
      const int c = o.c_idx;
      const int ax0 = ((int)o.x0 - 1) & ~7, wy0 = (int)o.y0 - 1, tile_p = (64 + 40 + 7) & ~7;      // RUN_TILE_P_OF(64) of k_run
      const int cw_ = c ? P.cwid : P.width, ch_ = c ? P.chei : P.height;
      const int wx1c = (int)o.wx1 < cw_ ? (int)o.wx1 : cw_, wy1c = (int)o.wy1 < ch_ ? (int)o.wy1 : ch_;
      // -- a lane per producer: its segment of the row above the box, its segment of the column left of it
      bool bad = false; uint32_t sg[4] = { 0, 0, 0, 0 }; int have_row = 0, have_col = 0; uint32_t pk = 0; int px0 = 0, py0 = 0;
      if ((uint32_t)lane < nd) {
        pk = s_dl[lane];
        const RunTask Pq = B.runs[pk];
        px0 = Pq.x0; py0 = Pq.y0;
        if ((Pq.micro & 3) != 2) bad = true;                          // producer: ordinary and dense
        if ((int)Pq.y0 <= wy0 && wy0 < (int)Pq.y1) {
          const int xs = (int)Pq.x0 > (int)o.x0 - 1 ? (int)Pq.x0 : (int)o.x0 - 1, xe = (int)Pq.x1 < wx1c ? (int)Pq.x1 : wx1c;
          if (xs < xe) {
            if ((int)Pq.y1 - 1 != wy0) bad = true;
            sg[0] = ((uint32_t)(xe - xs - 1) << 24); sg[1] = (uint32_t)(xs - Pq.x0) | ((uint32_t)(xs - ax0) << 8); have_row = 1;
          }
        }
        if ((int)Pq.x0 <= (int)o.x0 - 1 && (int)o.x0 - 1 < (int)Pq.x1) {
          const int ys = (int)Pq.y0 > (int)o.y0 ? (int)Pq.y0 : (int)o.y0, ye = (int)Pq.y1 < wy1c ? (int)Pq.y1 : wy1c;
          if (ys < ye) {
            if ((int)Pq.x1 != (int)o.x0) bad = true;
            sg[2] = ((uint32_t)(ye - ys - 1) << 24) | 0x80000000u;
            sg[3] = (uint32_t)(ys - Pq.y0) | ((uint32_t)((ys - wy0) * tile_p + ((int)o.x0 - 1 - ax0)) << 8); have_col = 1;
          }
        }
        if (!have_row && !have_col) bad = true;
      }
      const bool ok = __ballot(bad) == 0;
      const uint32_t cnt = (uint32_t)(have_row + have_col), incl = wave_scan_incl_u(cnt);
      const int nseg = (int)__builtin_amdgcn_readlane((int)incl, 63);
      if (ok && nseg > 0) {
        reader = true; new_micro |= 4;
        uint32_t at_seg = incl - cnt;
        // (what each neighbour sample of a segment is: an entry of the row above the box (corner first) or of the column beside it)
        if (have_row) { s_seg[2 * at_seg] = sg[0]; s_seg[2 * at_seg + 1] = sg[1]; s_segrun[at_seg] = pk; s_nb[at_seg] = (uint32_t)(px0 + (int)(sg[1] & 63) - ((int)o.x0 - 1)); at_seg++; }
        if (have_col) { s_seg[2 * at_seg] = sg[2]; s_seg[2 * at_seg + 1] = sg[3]; s_segrun[at_seg] = pk; s_nb[at_seg] = 0x100u + (uint32_t)(py0 + (int)(sg[3] & 63) - (int)o.y0); }
        WAVE_ORDER();
        if (lane < nseg) {                                 // producer run -> its mailbox; it learns that it is read (scan_run3)
          const uint32_t pr = s_segrun[lane];
          uint32_t* word = reinterpret_cast<uint32_t*>(B.pub_flag + (pr & ~3u));
          const uint32_t bit = 1u << (8 * (pr & 3u));
          if (!(atomicOr(word, bit) & bit)) B.lvl_cnt[atomicAdd(&B.counts->n_pub, 1u)] = pr;
          s_seg[2 * lane] |= B.mbx[3 * (size_t)pr] & 0xFFFFFFu;
        }
        WAVE_ORDER();
        // -- when is each neighbour sample first needed?  need epochs -> at most four poll points (quantiles of the distinct values)
        const int nl = (int)o.n_lvls + 1;
        bool phased = (P.flags & SCANF_MB_PHASES) && c == 0 && nl >= 4;
        int n_groups = 1, nsub = 0;
        uint32_t polls[4] = { 0, 0, 0, 0 }, ends[4] = { 0, 0, 0, 0 }, tot = 0;
        if (phased) {
          for (int q2 = lane; q2 < 161; q2 += 64) s_nrow[q2] = (uint8_t)(q2 == 0 ? s_need[80] : s_need[(q2 - 1) >> 2]);
          for (int q2 = lane; q2 < 160; q2 += 64) s_ncol[q2] = (uint8_t)s_need[40 + (q2 >> 2)];
          if (lane < 8) s_seen[lane] = 0;
          WAVE_ORDER();
          auto need_of = [&](int sgi, int off) -> uint32_t { const uint32_t nb = s_nb[sgi]; return (nb & 0x100u) ? s_ncol[(nb & 0xFFu) + off] : s_nrow[nb + off]; };
          for (int sgi = 0; sgi < nseg; sgi++) {
            const int cnt_s = (int)((s_seg[2 * sgi] >> 24) & 63) + 1;
            if (lane < cnt_s) { const uint32_t v = need_of(sgi, lane); atomicOr(&s_seen[v >> 5], 1u << (v & 31)); }
          }
          WAVE_ORDER();
          uint32_t seen[8];
          for (int k = 0; k < 8; k++) seen[k] = s_seen[k];
          seen[7] &= 0x7FFFFFFFu;                          // (255: never read)
          int nv = 0;
          for (int k = 0; k < 8; k++) nv += __popc(seen[k]);
          if (nv < 2) phased = false;
          else {
            n_groups = nv < 4 ? nv : 4;
            for (int g2 = 0; g2 < n_groups; g2++) {        // polls[g] = the (g * nv / n_groups)-th distinct value
              int k = (g2 * nv) / n_groups, wd = 0;
              while (k >= __popc(seen[wd])) { k -= __popc(seen[wd]); wd++; }
              uint32_t m = seen[wd];
              for (; k > 0; k--) m &= m - 1;
              polls[g2] = (uint32_t)(32 * wd + __builtin_ctz(m));
            }
            const int p1 = n_groups > 1 ? (int)polls[1] : 256, p2 = n_groups > 2 ? (int)polls[2] : 256, p3 = n_groups > 3 ? (int)polls[3] : 256;
            auto grp_of_v = [&](int v) { return v == 255 ? 255 : (v >= p1) + (v >= p2) + (v >= p3); };
            // every segment cut where the group of its samples changes (samples nobody reads, group 255, are left out)
            for (int sgi = 0; sgi < nseg && phased; sgi++) {
              const uint32_t s0 = s_seg[2 * sgi], s1 = s_seg[2 * sgi + 1];
              const int cnt_s = (int)((s0 >> 24) & 63) + 1;
              const bool col = s0 >> 31;
              const int g = lane < cnt_s ? grp_of_v((int)need_of(sgi, lane)) : 254;
              const int gprev = __shfl_up(g, 1, 64);
              const bool boundary = lane < cnt_s && (lane == 0 || g != gprev);
              const uint64_t bm = __ballot(boundary), sm = __ballot(boundary && g != 255);
              if (nsub + __popcll(sm) > 48) { phased = false; break; }
              if (boundary && g != 255) {
                const uint64_t above = lane < 63 ? (bm >> (lane + 1)) : 0ull;
                const int len = above ? __builtin_ctzll(above) + 1 : cnt_s - lane;
                const int ix = nsub + __popcll(sm & lanes_below(lane));
                s_sub[2 * ix] = (s0 & 0x80FFFFFFu) | ((uint32_t)(len - 1) << 24);
                s_sub[2 * ix + 1] = ((s1 & 63u) + (uint32_t)lane) | (((s1 >> 8) + (uint32_t)(lane * (col ? tile_p : 1))) << 8);
                s_subg[ix] = (uint8_t)g;
              }
              nsub += __popcll(sm);
              WAVE_ORDER();
            }
          }
        }
        if (phased && nsub > 0) {
          const int mylen = lane < nsub ? (int)((s_sub[2 * lane] >> 24) & 63) + 1 : 0, myg = lane < nsub ? (int)s_subg[lane] : 255;
          for (int g2 = 0; g2 < n_groups; g2++) { tot += wave_sum_u(myg == g2 ? (uint32_t)mylen : 0u); ends[g2] = tot; }
          if (tot > 255) phased = false;                   // (cannot happen with 64x64 boxes: <= 193 neighbour samples)
        }
        const bool use_sub = phased && nsub > 0;
        const uint32_t words = 3u + 2u * (uint32_t)(use_sub ? nsub : nseg);
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(&B.counts->n_segs_alloc, words);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (at + words > P.cap_segs) { if (lane == 0) scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
        uint32_t* ms = B.mb_segs + at;
        if (use_sub) {
          const int myg = lane < nsub ? (int)s_subg[lane] : 255;
          uint32_t pos = 0;
          for (int g2 = 0; g2 < n_groups; g2++) {
            const uint64_t gm = __ballot(myg == g2);
            if (myg > g2) pos += (uint32_t)__popcll(gm);
            else if (myg == g2) pos += (uint32_t)__popcll(gm & lanes_below(lane));
          }
          if (lane < nsub) { ms[3 + 2 * pos] = s_sub[2 * lane]; ms[3 + 2 * pos + 1] = s_sub[2 * lane + 1]; }
          if (lane == 0) {
            for (int g2 = n_groups; g2 < 4; g2++) { ends[g2] = tot; polls[g2] = 255; }
            ms[0] = (uint32_t)nsub | ((uint32_t)n_groups << 8);
            ms[1] = ends[0] | (ends[1] << 8) | (ends[2] << 16) | (ends[3] << 24);
            ms[2] = polls[0] | (polls[1] << 8) | (polls[2] << 16) | (polls[3] << 24);
          }
        } else {
          if (lane == 0) { ms[0] = (uint32_t)nseg | (1u << 8); ms[1] = 0; ms[2] = 0; }      // (one group: everything at the start)
          if (lane < 2 * nseg) ms[3 + lane] = s_seg[lane];
        }
        if (lane == 0) B.mbx[3 * (size_t)s + 1] = at;
      }
    }
    // (only the fields that changed: other runs read this record's box and class bits in the same pass)
    if (lane == 0) {
      B.runs[s].n_deps = (uint16_t)nd;
      if (reader) B.runs[s].micro = (uint8_t)new_micro;
      if (nd == 0) n_ready++;
    }
    WAVE_ORDER();
  }
  if (lane == 0 && n_ready) atomicAdd(&B.counts->n_ready, n_ready);
}

// Ticket slots (scan_core.h "tickets"), one workgroup: the CTBs in ctb_order are dealt to the threads in contiguous chunks; how
// a chunk's runs fill tickets depends on how full the open ticket is when the chunk begins, so every thread first computes its
// chunk's effect for each of the eight possible fill states (a table), the tables are composed by a prefix scan (composition
// of such tables is associative), and every thread then walks its chunk again from its true start state and writes the slots.
#define SCO_THREADS 1024
__device__ void scan_order_body(const ScanParams& P, const ScanBufs& B, uint32_t cap_levels, uint32_t cap_resid);
__global__ __launch_bounds__(SCO_THREADS)
void k_scan_order(ScanBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  SCAN_PRIO();
  const ScanParams& P = J.job[blockIdx.y].P; const ScanBufs& B = J.job[blockIdx.y].B;
  const uint32_t cap_levels = J.job[blockIdx.y].cap_levels;
  // the third run pass first (a thread per listed run, scan_core.h scan_run3: the runs somebody reads through their mailbox)
  if (!B.counts->status && (P.flags & SCANF_MAILBOX)) {
    const uint32_t n = B.counts->n_pub;
    for (uint32_t q = threadIdx.x; q < n; q += SCO_THREADS) scan_run3(P, B, B.lvl_cnt[q]);
  }
  // (one workgroup, one CU: what its threads have stored is visible to each other behind a barrier.  An agent-scope fence here -
  //  __threadfence() - writes the whole L2 back, the reconstruction kernels' dirty lines included, once per wavefront: round 4)
  __syncthreads();
  scan_order_body(P, B, cap_levels, J.job[blockIdx.y].cap_resid);
  // ---- the scan's verdict and counts to the host: into the picture's pinned record, then its ready word (system scope)
  __syncthreads();
  if (threadIdx.x < sizeof(ScanCounts) / 4 - 2 && B.host_counts)
    __hip_atomic_store(reinterpret_cast<uint32_t*>(B.host_counts) + threadIdx.x, reinterpret_cast<const uint32_t*>(B.counts)[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (threadIdx.x == 0 && B.host_counts) __hip_atomic_store(&B.host_counts->ready, B.ready_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ void scan_order_body(const ScanParams& P, const ScanBufs& B, uint32_t cap_levels, uint32_t cap_resid)
{
  __shared__ uint32_t tab[SCO_THREADS][RUN_TICKET_SLOTS];          // per chunk and fill state at its start: tickets it opens | fill state at its end << 28
  __shared__ uint32_t s_diag[4096];
  __shared__ uint32_t s_widest, s_ndiag;
  const int tid = threadIdx.x;
  ScanCounts& K = *B.counts;
  if (K.status) return;
  const int n = P.n_ctbs, chunk = (n + SCO_THREADS - 1) / SCO_THREADS;
  const int t0 = min(n, tid * chunk), t1 = min(n, t0 + chunk);
  const uint32_t victim = K.victim;
  const bool by_records = (P.flags & SCANF_DROP_PRODUCER) != 0;       // (the fault injection's victim is only known by its id)
  const int n_diag = P.ctbs_w + 2 * P.ctbs_h;
  uint32_t* diag = n_diag <= 4096 ? s_diag : B.lvl_cnt;          // ticketed runs per anti-diagonal (for the worker count)
  for (int q = tid; q < n_diag; q += SCO_THREADS) diag[q] = 0;
  if (tid == 0) { s_widest = 0; s_ndiag = 0; }
  __syncthreads();
  // ---- phase 1: the chunk's table
  ScanTicketState st[RUN_TICKET_SLOTS];
  for (int o = 0; o < RUN_TICKET_SLOTS; o++) { st[o].tickets = 0; st[o].fill = (uint32_t)o; }
  for (int t = t0; t < t1; t++) {
    const int rs = B.ctb_order[t];
    const ScanCtb& C = B.ctb[rs];
    const uint32_t nr = C.n_runs, ib = C.intra_base;
    uint32_t cnt = 0;
    if (!by_records && !C.slow) {
      // (the run pass left which runs take a ticket and which are micro runs in two words of the CTB's record: no run record
      //  is read here - 2 x 8 dependent loads per thread were this kernel's 100 us)
      const uint64_t mm = C.micro_mask;
      for (uint64_t tm = C.tick_mask; tm; tm &= tm - 1) {
        const bool micro = (mm >> __builtin_ctzll(tm)) & 1;
        uint32_t tk, sl;
        for (int o = 0; o < RUN_TICKET_SLOTS; o++) scan_ticket_step(st[o], micro, &tk, &sl);
        cnt++;
      }
    } else
    for (uint32_t r = 0; r < nr; r++) {
      const uint32_t s = ib + r;
      const uint32_t mic = B.runs[s].micro;
      if ((mic & RUN_MICRO_FRONT) || s == victim) continue;
      uint32_t tk, sl;
      for (int o = 0; o < RUN_TICKET_SLOTS; o++) scan_ticket_step(st[o], mic & 1, &tk, &sl);
      cnt++;
    }
    if (cnt) atomicAdd(&diag[rs % P.ctbs_w + 2 * (rs / P.ctbs_w)], cnt);
  }
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
    if (!by_records && !C.slow) {
      const uint64_t mm = C.micro_mask;
      for (uint64_t tm = C.tick_mask; tm; tm &= tm - 1) {
        const uint32_t r = (uint32_t)__builtin_ctzll(tm), s = ib + r;
        const bool micro = (mm >> r) & 1;
        uint32_t tk, sl;
        scan_ticket_step(me, micro, &tk, &sl);
        B.slots[tk * RUN_TICKET_SLOTS + sl] = micro ? (s | 0x80000000u) : s;
      }
    } else
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
    K.widest = max(s_widest, K.n_ready);
    if (K.n_isamp > cap_resid) scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE);      // (overlapping intra TUs - a malformed description - could ask for more residual samples than the picture has)
    K.max_rl = s_ndiag;
  }
}

// ------------------------------------------------------------------------------------------------ before the passes
__global__ __launch_bounds__(256)
void k_build_prep(PrepBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  SCAN_PRIO();
  const PrepJob& Q = J.job[blockIdx.y];
  const unsigned long long step = (unsigned long long)gridDim.x * 256 * 16, first = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 16;
  {
    uint4* z = reinterpret_cast<uint4*>(Q.zero);
    const uint4 zero = make_uint4(0, 0, 0, 0);
    for (unsigned long long at = first; at < Q.zero_bytes; at += step) z[at >> 4] = zero;
  }
  if (Q.ff) {
    uint4* f = reinterpret_cast<uint4*>(Q.ff);
    const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
    const unsigned long long whole = Q.ff_bytes & ~15ull;
    for (unsigned long long at = first; at < whole; at += step) f[at >> 4] = ones;
    if (blockIdx.x == 0 && threadIdx.x < (Q.ff_bytes & 15ull)) Q.ff[whole + threadIdx.x] = 0xFF;
  }
}

// the motion planes of the batch's pictures from their PU records (k_lf.hip k_motion_from_pus, the same body): sixteen lanes per
// PU, sixteen PUs per workgroup
__global__ __launch_bounds__(256)
void k_motion_batch(PrepBatch J)
{
  if (blockIdx.y >= (unsigned)J.n) return;
  SCAN_PRIO();
  const PrepJob& Q = J.job[blockIdx.y];
  motion_from_pus_body(Q.pus, Q.n_pus, Q.slices, Q.n_slices, Q.motion, Q.w4, Q.h4, (int)blockIdx.x, (int)gridDim.x);
}

hipError_t prep_enqueue_batch(hipStream_t st, const PrepBatch& J)
{
  if (J.n <= 0) return hipSuccess;
  unsigned long long most = 0; int most_pus = 0;
  for (int i = 0; i < J.n; i++) { most = std::max(most, std::max(J.job[i].zero_bytes, J.job[i].ff ? J.job[i].ff_bytes : 0ull)); most_pus = std::max(most_pus, J.job[i].ff ? J.job[i].n_pus : 0); }
  // (a thread stores 16 bytes per step; 2048 workgroups at most: a grid that is resident at once)
  // (bounded grids: see scan_enqueue_batch)
  const unsigned cap = std::max(64u, 1024u / (unsigned)J.n);
  const unsigned blocks = (unsigned)std::min<unsigned long long>(cap, (most + 256 * 16 - 1) / (256 * 16));
  if (blocks) hipLaunchKernelGGL(k_build_prep, dim3(blocks, (unsigned)J.n), dim3(256), 0, st, J);
  if (most_pus > 0) hipLaunchKernelGGL(k_motion_batch, dim3(std::min<unsigned>((most_pus + 15) / 16, cap), (unsigned)J.n), dim3(256), 0, st, J);
  return hipGetLastError();
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
  // Bounded grids that walk their units (records, CTBs, runs): a dispatch pipe hands out ONE kernel's workgroups at a time, and a
  // grid that does not fit the device at once keeps its pipe until its last workgroup has found a place - a batch of four
  // pictures' per-CTB passes (8 160 one-wavefront workgroups) for most of its run time.  The kernel stream of some decoder
  // lives on the same pipe (there are four pipes): its next kernel waited behind the scan's.  (DE265HIP_SCAN_GRID: the
  // wavefronts of a launch, all pictures of the batch together.)
  static const int scan_grid = d265_env("DE265HIP_SCAN_GRID") ? std::max(64, atoi(d265_env("DE265HIP_SCAN_GRID"))) : 1024;
  const unsigned per_pic = (unsigned)std::max(32, scan_grid / (int)ny);
  // (the per-TU pass's launch also carries the motion planes, behind its own workgroups: ScanJob::mo_*)
  int most_pus = 0;
  for (int i = 0; i < J.n; i++) if (J.job[i].mo_plane) most_pus = std::max(most_pus, J.job[i].mo_n_pus);
  ScanBatch K = J;
  K.tus_blocks = max_tus > 0 ? (int)std::min<unsigned>((max_tus + 255) / 256, per_pic) : 0;
  const unsigned mo_blocks = most_pus > 0 ? std::min<unsigned>((most_pus + 15) / 16, std::max(16u, per_pic / 2)) : 0u;
  if (K.tus_blocks + mo_blocks > 0) hipLaunchKernelGGL(k_scan_tus, dim3((unsigned)K.tus_blocks + mo_blocks, ny), dim3(256), 0, st, K);
  if (max_tus > 0) {
    hipLaunchKernelGGL(k_scan_ctbs, dim3(std::min<unsigned>(max_ctbs, per_pic), ny), dim3(64), 0, st, J);
    // (the number of runs is only known on the device: fixed grids of wavefronts walk the run lists)
    // (512 wavefronts: with 128 / 256 / 512 / 1024 the product path of the bench made 1 860 / 2 310 / 2 630 / 2 380 pictures/s -
    //  fewer leave the run passes' latency chains too long, more crowd the reconstruction kernels of the other streams)
    // (a wavefront per CTB; 4:2:0 and monochrome pictures have at most 384 intra TUs in a CTB: the smaller LDS arrays)
    bool small_ctbs = true;
    for (int i = 0; i < J.n; i++) small_ctbs = small_ctbs && J.job[i].P.cf <= 1;
    if (small_ctbs) hipLaunchKernelGGL(k_scan_runs1<384>, dim3(std::min<unsigned>(max_ctbs, per_pic), ny), dim3(64), 0, st, J);
    else hipLaunchKernelGGL(k_scan_runs1<SR_TMAX>, dim3(std::min<unsigned>(max_ctbs, per_pic), ny), dim3(64), 0, st, J);
    hipLaunchKernelGGL(k_scan_runs2, dim3(std::min<unsigned>(per_pic, 256u), ny), dim3(64), 0, st, J);
    hipLaunchKernelGGL(k_scan_runs2b, dim3(2 * per_pic, ny), dim3(64), 0, st, J);      // (a B picture has a few dozen candidates, an all-intra picture some 6 000 at ~130 us each)
  }
  hipLaunchKernelGGL(k_scan_order, dim3(1, ny), dim3(SCO_THREADS), 0, st, J);      // (always: it reports to the host)
  return hipGetLastError();
}

hipError_t scan_enqueue(hipStream_t st, const ScanParams& P, const ScanBufs& B, const ScanLayout& L, uint8_t* base, uint32_t cap_resid)
{
  (void)base;
  ScanBatch J; memset(&J, 0, sizeof(J)); J.n = 1;
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
