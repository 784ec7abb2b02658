// scan_core.h -- the intra scan of a picture's TU records as data-parallel passes: neighbour availability
// (intrapred.cc:437-527 preproc, :577-688 fill_from_image), the units each mode reads, runs of intra TUs, in-run levels,
// run -> run edges, run records with their TU lists, ready / need epochs and mailbox segments of the phased hand-over,
// level-0 (residual) task lists.  Round 3 did all of this in ONE sequential loop on a host core (host.hip, "TU scan" and
// "runs": 19 of an all-intra 4K picture's 20 ms, 4.5 of a B picture's 5.7 ms); here every pass is a function of ONE unit of
// work - a TU record, a CTB, a run - that touches nothing another unit of the same pass writes, so a pass is a kernel with
// one thread per unit (k_scan.hip) behind the upload of the raw records, and, compiled for the host, a plain loop
// (scan_host.hip: the CPU rehearsal the equivalence tests run against the round-3 scan without a GPU).
//
// Passes (k_scan.hip launches them in this order on the decoder's copy stream):
//   scan_tu      per TU record   validation, class counts per CTB, availability and needed-unit masks, cell map
//   scan_prefix  (one workgroup) per-CTB bases of the task lists, the run-ordered TU array and the sparse run ids
//   scan_ctb     per CTB         level-0 tasks of the inter TUs; the sequential part that is left: which run a TU joins
//                                and its in-run level (a CTB's TUs in decode order; CTBs are independent of each other)
//   scan_run     per run         run record, its TUs in chain order, residual-only tasks, producer list
//   scan_run2    per run         producers that are front runs leave the list; mailbox segments and need epochs of a reader
//   scan_run3    per run         a run somebody reads through its mailbox: ready epochs of its packets
//   scan_order   (one workgroup) ticket slots: the runs in a topological order that needs no graph walk (below)
//
// A run is identified by its SPARSE id: (first intra TU of its CTB in the compacted intra order) + (its number inside the
// CTB, in order of creation).  Ids grow in decode order, so a producer's id is smaller than its readers'.  Run records,
// flags, mailbox tables are indexed by it; only the ticket slots are ordered by dependency level.
#pragma once
#include "dev_common.h"

namespace d265 {

#if defined(__HIP_DEVICE_COMPILE__)
#define SCAN_DEVICE 1
#else
#define SCAN_DEVICE 0
#endif
#define SCAN_FN __host__ __device__ inline

// ---- what a build asks for (by value to every pass)
enum : uint32_t {
  SCANF_CIP = 1u << 0, SCANF_MODE_DEPS = 1u << 1, SCANF_MERGE = 1u << 2, SCANF_MAILBOX = 1u << 3, SCANF_MB_PHASES = 1u << 4,
  SCANF_MICRO_OFF = 1u << 5, SCANF_NO_DENSE = 1u << 6, SCANF_MICRO16 = 1u << 7, SCANF_FRONT_OFF = 1u << 8,
  SCANF_IMPLICIT_RDPCM = 1u << 9, SCANF_ROTATION = 1u << 10, SCANF_DROP_PRODUCER = 1u << 11, SCANF_CHECK_POS = 1u << 12,
};
struct ScanParams {
  int32_t width, height, cwid, chei, subw, subh, lc, lt, cf, ctbs_w, ctbs_h, n_ctbs;
  int32_t map_w[3], map_h[3];          // the 4x4 cell maps of the three components
  int32_t n_tus, n_coeffs;
  int32_t bppY, bppC;                  // bytes per sample
  uint32_t flags;
  int32_t micro_tus, run_waves;
  uint32_t cap_runs, cap_deps, cap_segs, cap_mb, cap_slots;
};

// per CTB (raster address)
struct ScanCtb {
  uint32_t first_tu, end_tu;           // its TU records (decode order: a CTB's records are contiguous)
  uint32_t seen;                       // times the record array entered this CTB (1, or the description is malformed)
  uint32_t n_inter[4], n_ro[4];        // level-0 tasks by size class (log2 - 2): TUs added into the predicted picture / residual-only copies of intra TUs
  uint32_t n_rext_inter, n_rext_ro;    // level-0 tasks of k_resid_rext
  uint32_t n_intra, n_isamp;           // intra TUs and their samples
  uint32_t l0_base[4], rext_base, intra_base, isamp_base;      // scan_prefix: where this CTB's share of each list starts
  uint32_t n_runs;                     // scan_ctb
  uint32_t slow;                       // (device) more than 64 runs: scan_order walks the run records instead of the masks
  unsigned long long tick_mask, micro_mask;      // (device, the run pass) bit r: run r of the CTB takes a ticket (it is not a front run) / is a micro run
};
static_assert(sizeof(ScanCtb) % 8 == 0, "ScanCtb layout");

// device -> host (one asynchronous copy behind the passes)
struct ScanCounts {
  uint32_t status;                     // 0, or the DE265HIP_ERROR_* the build would have returned
  uint32_t n_l0_size[4], n_l0_rext;
  uint32_t n_runs, n_front, n_batches, widest, max_rl, n_mailboxes;
  uint32_t n_deps_alloc, n_segs_alloc, n_tasks, victim, sum_lvls, n_intra;
  uint32_t n_listed;                   // runs in run_list
  uint32_t alloc[7];                   // (device) how far the task lists, the run-ordered TU array and the residual samples are given out (scan_ctb: a CTB takes its share with one atomic per list)
  uint32_t n_ready;                    // (device) ticketed runs that wait for nothing
  uint32_t n_cand;                     // (device) runs that may become mailbox readers (their list: run_level)
  uint32_t n_pub;                      // (device) runs somebody reads through their mailbox (their list: lvl_cnt)
  unsigned long long alg_resid, alg_intra, alg_intra_front, n_isamp;
  uint32_t ready, pad1;                // (the host's copy only) the tag of the build, stored after everything else
};

// a 4x4 cell of a component: low word = index + 1 of the intra TU record covering it (0: none; scan_tu),
// high word = run number inside its CTB | in-run level << 16 | 1 << 31 once scan_ctb has passed that TU
typedef unsigned long long ScanCell;

struct ScanBufs {
  // inputs (uploaded)
  const de265hip_tu* tus; const uint32_t* ctb_group; const int32_t* rs2ts; const int32_t* ts2rs; const int32_t* ctb_order; const uint8_t* blk_flags;
  const uint64_t* used_units;          // [4][35][2]: neighbour units a TU of that size / mode / (luma-like smoothing) reads
  uint16_t* coeff_pos;                 // (positions beyond a TU's block are folded into it, k_check_coeffs' job in round 3)
  // scratch
  ScanCtb* ctb; ScanCell* cell[3]; uint64_t* tu_avail; uint64_t* tu_need; uint32_t* tu_info;
  uint32_t* tu_run;                    // sparse id of the run of an intra TU (scan_ctb)
  uint8_t* run_ntus; uint32_t* run_rs; uint32_t* run_nall; uint32_t* run_level; uint32_t* run_list; uint8_t* pub_flag; uint8_t* rdy_tab;
  uint32_t* lvl_cnt;                   // scan_order: 4 x (levels + 2) counters
  // outputs (what the reconstruction kernels read)
  TuTask* l0; TuTask* l0x; RunTask* runs; TuTask* run_tus; uint32_t* deps; uint32_t* slots; uint32_t* front_idx;
  uint32_t* mbx; uint32_t* mb_segs;
  ScanCounts* counts;
  ScanCounts* host_counts;             // the picture's pinned record on the host (device-visible): written by the last pass
  uint32_t* err_word;                  // the picture's error word (k_run, coefficient positions): cleared by the first pass
  uint32_t ready_tag;
};

// tu_info word of an intra TU
#define SCAN_TI_RUN(w)   ((w) & 0xFFFFu)
#define SCAN_TI_LLEV(w)  (((w) >> 16) & 0xFFu)
#define SCAN_TI_FOREIGN  (1u << 26)
#define SCAN_TI_INTRA    (1u << 27)
// RunTask::micro bit 4 (16): a front run (reconstructed by k_intra_front ahead of k_run; no ticket, no flag)
#define RUN_MICRO_FRONT 16

// ---- atomics: the device's, or plain arithmetic in the single-threaded host rehearsal
SCAN_FN uint32_t scan_add(uint32_t* p, uint32_t v)
{
#if SCAN_DEVICE
  return atomicAdd(p, v);
#else
  const uint32_t o = *p; *p = o + v; return o;
#endif
}
SCAN_FN void scan_add64(unsigned long long* p, unsigned long long v)
{
#if SCAN_DEVICE
  if (v) atomicAdd(p, v);
#else
  *p += v;
#endif
}
SCAN_FN void scan_min(uint32_t* p, uint32_t v)
{
#if SCAN_DEVICE
  atomicMin(p, v);
#else
  if (v < *p) *p = v;
#endif
}
SCAN_FN void scan_fail(const ScanBufs& B, uint32_t code)
{
  // (the first error wins; PARAMETER_OUT_OF_RANGE = 8 is the smallest code, so a malformed record beats a later refusal)
#if SCAN_DEVICE
  atomicCAS(&B.counts->status, 0u, code);
#else
  if (!B.counts->status) B.counts->status = code;
#endif
}

SCAN_FN int scan_ctz64(uint64_t v)
{
#if SCAN_DEVICE
  return __ffsll((long long)v) - 1;
#else
  return __builtin_ctzll(v);
#endif
}
SCAN_FN int scan_clz64(uint64_t v)
{
#if SCAN_DEVICE
  return __clzll((long long)v);
#else
  return __builtin_clzll(v);
#endif
}
SCAN_FN int scan_popc64(uint64_t v)
{
#if SCAN_DEVICE
  return __popcll(v);
#else
  return __builtin_popcountll(v);
#endif
}

// the units whose samples a TU with availability `avail` really reads: the used ones that are available, plus, for every
// used but unavailable one, the unit its samples are substituted from (intrapred.cc:395-431)
SCAN_FN uint64_t scan_needed_units(uint64_t used, uint64_t avail)
{
  if (avail == 0) return 0;
  uint64_t need = used & avail, miss = used & ~avail;
  while (miss) {
    const int u = scan_ctz64(miss); miss &= miss - 1;
    const uint64_t below = avail & ((2ull << u) - 1ull);
    need |= below ? 1ull << (63 - scan_clz64(below)) : avail & (~avail + 1ull);
  }
  return need;
}

SCAN_FN int scan_intra_angle(int m)
{
  // intraPredAngle (intrapred.cc:742-760)
  const int a[35] = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                      -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
  return a[m];
}
SCAN_FN int scan_inv_angle(int m)
{
  const int v[15] = { -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };
  return (m >= 11 && m <= 25 && scan_intra_angle(m) < 0) ? v[m - 11] : 0;
}

// MinTbAddrZS of the minimum transform block at (xtb, ytb) (6.5.2; pps.cc:671-690): tile-scan address of its CTB, then the
// Morton interleave of its position inside the CTB
SCAN_FN int scan_zs(const ScanParams& P, const ScanBufs& B, int xtb, int ytb)
{
  const int dl = P.lc - P.lt;
  const int cx = xtb >> dl, cy = ytb >> dl;
  int v = B.rs2ts[cy * P.ctbs_w + cx] << (2 * dl);
  for (int i = 0; i < dl; i++) v |= (((xtb >> i) & 1) << (2 * i)) | (((ytb >> i) & 1) << (2 * i + 1));
  return v;
}

// the static checks of one TU record (what de265hip_picture_build refused in its scan loop)
SCAN_FN bool scan_tu_valid(const ScanParams& P, const de265hip_tu& tu)
{
  if (tu.log2_size < 2 || tu.log2_size > 5) return false;
  const int nT = 1 << tu.log2_size;
  const int cw = tu.c_idx ? P.cwid : P.width, ch = tu.c_idx ? P.chei : P.height;
  if (tu.c_idx > 2 || ((tu.x0 | tu.y0) & 3) || tu.x0 + nT > cw || tu.y0 + nT > ch || tu.qp < 0) return false;
  // a transform block is a leaf of a quadtree: aligned to its size, inside one CTB (the CTB pass keeps a CTB's cells in LDS)
  if (((tu.x0 | tu.y0) & (nT - 1)) || ((nT * (tu.c_idx ? P.subw : 1)) >> P.lc) > 1 || ((nT * (tu.c_idx ? P.subh : 1)) >> P.lc) > 1) return false;
  if ((tu.flags & DE265HIP_TU_CBF) && ((int64_t)tu.coeff_offset + tu.n_coeff > P.n_coeffs || tu.n_coeff > nT * nT)) return false;
  return true;
}
SCAN_FN int scan_tu_ctb(const ScanParams& P, const de265hip_tu& tu)
{
  const int xL = tu.x0 * (tu.c_idx ? P.subw : 1), yL = tu.y0 * (tu.c_idx ? P.subh : 1);
  return (xL >> P.lc) + (yL >> P.lc) * P.ctbs_w;
}

// Range-extension tools of a TU (D265_RX_* bits; 0 for every TU of a Main / Main10 picture), see host.hip rx_bits
SCAN_FN int scan_rx_bits(const ScanParams& P, const ScanBufs& B, const de265hip_tu& tu)
{
  if (!(tu.flags & (DE265HIP_TU_TSKIP | DE265HIP_TU_BYPASS | DE265HIP_TU_EXPLICIT_RDPCM)) && !tu.res_scale_val) return 0;
  const bool cbf = (tu.flags & DE265HIP_TU_CBF) && tu.n_coeff;
  const bool ts_or_bp = tu.flags & (DE265HIP_TU_TSKIP | DE265HIP_TU_BYPASS);
  int rx = 0;
  if (cbf && ts_or_bp) {
    if (tu.flags & DE265HIP_TU_INTRA) {
      if ((P.flags & SCANF_IMPLICIT_RDPCM) && (tu.intra_mode == 10 || tu.intra_mode == 26)) rx |= tu.intra_mode == 26 ? D265_RX_RDPCM_V : D265_RX_RDPCM_H;
    } else if (tu.flags & DE265HIP_TU_EXPLICIT_RDPCM) rx |= (tu.flags & DE265HIP_TU_EXPLICIT_RDPCM_VERT) ? D265_RX_RDPCM_V : D265_RX_RDPCM_H;
    // (the reference looks the CU up at the TU's position in samples of ITS component through an accessor that takes luma
    //  samples, transform.cc:393-395: reproduced)
    if ((P.flags & SCANF_ROTATION) && tu.log2_size == 2 &&
        (B.blk_flags[(tu.x0 >> 2) + (tu.y0 >> 2) * ((P.width + 3) >> 2)] & DE265HIP_BLK_INTRA)) rx |= D265_RX_ROTATE;
    if ((tu.flags & DE265HIP_TU_TSKIP) && !(tu.flags & DE265HIP_TU_BYPASS) && tu.log2_size > 3) rx |= 0x80;
  }
  if (tu.c_idx && tu.res_scale_val) rx |= D265_RX_XCC;
  return rx;
}

// what becomes of a TU record: 0 nothing to reconstruct, 1 level-0 task of its size class (residual added into the predicted
// picture), 2 level-0 task of k_resid_rext, 3 an intra TU
SCAN_FN int scan_tu_class(const ScanParams& P, const ScanBufs& B, const de265hip_tu& tu, int* rx_out)
{
  *rx_out = 0;
  if (!(tu.flags & (DE265HIP_TU_INTRA | DE265HIP_TU_TSKIP | DE265HIP_TU_BYPASS | DE265HIP_TU_EXPLICIT_RDPCM)) && !tu.res_scale_val)
    return (tu.flags & DE265HIP_TU_CBF) ? 1 : 0;
  const int rx = scan_rx_bits(P, B, tu);
  *rx_out = rx;
  if (!(tu.flags & (DE265HIP_TU_INTRA | DE265HIP_TU_CBF)) && !(rx & D265_RX_XCC)) return 0;
  if (tu.flags & DE265HIP_TU_INTRA) return 3;
  return rx ? 2 : 1;
}

// cross-component prediction: the luma TU of the same position and size comes right before the chroma TUs (4:4:4,
// slice.cc:3699-3750); its coefficient list is what k_resid_rext recomputes the luma residual from
SCAN_FN bool scan_xcc_luma(const ScanParams& P, const ScanBufs& B, int i, uint64_t* luma_info, int* rx_luma)
{
  const de265hip_tu& tu = B.tus[i];
  int j = i - 1;
  for (int back = 0; j >= 0 && back < 4 && B.tus[j].c_idx != 0; back++) j--;
  if (P.cf != 3 || j < 0 || B.tus[j].c_idx != 0) return false;
  const de265hip_tu& lt = B.tus[j];
  if (lt.x0 != tu.x0 || lt.y0 != tu.y0 || lt.log2_size != tu.log2_size || !scan_tu_valid(P, lt)) return false;
  const int lrx = scan_rx_bits(P, B, lt);
  const bool lcbf = (lt.flags & DE265HIP_TU_CBF) && lt.n_coeff;
  *luma_info = (uint64_t)lt.coeff_offset | ((uint64_t)(lcbf ? lt.n_coeff : 0) << 32) | ((uint64_t)(uint8_t)lt.qp << 48) | ((uint64_t)lt.flags << 56);
  *rx_luma = ((lrx & D265_RX_ROTATE) ? D265_RX_LUMA_ROT : 0) |
             (((lrx & D265_RX_RDPCM_V) ? 2 : ((lrx & D265_RX_RDPCM_H) ? 1 : 0)) << D265_RX_LUMA_RDPCM_SHIFT);
  return true;
}

// the task of a TU record: its fields, the rest zero (what every list starts from)
SCAN_FN TuTask scan_task_of(const de265hip_tu& tu)
{
  TuTask t;
  t.x0 = tu.x0; t.y0 = tu.y0; t.log2_size = tu.log2_size; t.c_idx = tu.c_idx; t.flags = tu.flags; t.intra_mode = tu.intra_mode;
  t.qp = tu.qp; t.run_level = 0; t.n_coeff = (tu.flags & DE265HIP_TU_CBF) ? tu.n_coeff : 0; t.coeff_offset = tu.coeff_offset;
  t.avail = 0; t.resid_offset = 0; t.angle = 0; t.pad3 = 0; t.inv_angle = 0;
  if (t.n_coeff == 0) t.flags &= (uint8_t)~DE265HIP_TU_CBF;
  return t;
}

// ------------------------------------------------------------------------------------------------ pass 1: per TU record
struct ScanTuSums { unsigned long long alg_resid, alg_intra, n_isamp; uint32_t n_tasks, n_intra; };

// 4x4 cell (of the TU's component map) of neighbour unit u: left column bottom -> top, corner, top row left -> right
SCAN_FN int scan_cell_of(int u, int xB, int yB, int nT, int mw)
{
  const int corner = nT >> 1;
  const int cell_l = ((xB - 1) >> 2) + ((yB >> 2) + corner - 1) * mw, cell_t = (xB >> 2) + ((yB >> 2) - 1) * mw;
  return u < corner ? cell_l - u * mw : (u == corner ? cell_t - 1 : cell_t + (u - corner - 1));
}

SCAN_FN void scan_tu(const ScanParams& P, const ScanBufs& B, int i, ScanTuSums& S)
{
  const de265hip_tu tu = B.tus[i];
  B.tu_info[i] = 0;                                    // (scan_ctb fills the words of the intra TUs)
  if (!scan_tu_valid(P, tu)) { scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); return; }
  const int nT = 1 << tu.log2_size, c = tu.c_idx;
  const int ctu = scan_tu_ctb(P, tu);
  // -- where the record array enters a CTB
  int prev = -1;
  if (i > 0) { const de265hip_tu q = B.tus[i - 1]; if (scan_tu_valid(P, q)) prev = scan_tu_ctb(P, q); }
  if (i == 0 || prev != ctu) {
    scan_add(&B.ctb[ctu].seen, 1u);
    B.ctb[ctu].first_tu = (uint32_t)i;
    if (prev >= 0) B.ctb[prev].end_tu = (uint32_t)i;
  }
  if (i == P.n_tus - 1) B.ctb[ctu].end_tu = (uint32_t)P.n_tus;
  // -- coefficient positions inside the TU's block: a position beyond it is folded into the block and the picture fails
  if ((P.flags & SCANF_CHECK_POS) && (tu.flags & DE265HIP_TU_CBF)) {
    uint16_t* cp = B.coeff_pos + tu.coeff_offset;
    const unsigned nS = (unsigned)(nT * nT);
    bool bad = false;
    for (int k = 0; k < tu.n_coeff; k++) if (cp[k] >= nS) { cp[k] = (uint16_t)(cp[k] & (nS - 1)); bad = true; }
    if (bad) scan_fail(B, DE265HIP_ERROR_DECODING);
  }
  int rx = 0;
  const int cls = scan_tu_class(P, B, tu, &rx);
  if (cls == 0) return;
  const int64_t bpp = c ? P.bppC : P.bppY;
  const bool cbf = (tu.flags & DE265HIP_TU_CBF) && tu.n_coeff;
  const int64_t coef_bytes = cbf ? (4 * (int64_t)tu.n_coeff < 2 * (int64_t)nT * nT ? 4 * (int64_t)tu.n_coeff : 2 * (int64_t)nT * nT) : 0;
  S.n_tasks++;
  ScanCtb& C = B.ctb[ctu];
  if (cls != 3) {
    if (rx & D265_RX_XCC) { uint64_t li; int rl; if (!scan_xcc_luma(P, B, i, &li, &rl)) { scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); return; } }
    if (cls == 1) scan_add(&C.n_inter[tu.log2_size - 2], 1u); else scan_add(&C.n_rext_inter, 1u);
    if (cbf) S.alg_resid += coef_bytes + 2 * bpp * nT * nT;
    return;
  }
  // -- an intra TU
  if ((rx & D265_RX_XCC)) { uint64_t li; int rl; if (!scan_xcc_luma(P, B, i, &li, &rl)) { scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); return; } }
  scan_add(&C.n_intra, 1u); scan_add(&C.n_isamp, (uint32_t)(nT * nT));
  S.n_intra++; S.n_isamp += (unsigned long long)(nT * nT);
  if (cbf || (rx & D265_RX_XCC)) { if (rx) scan_add(&C.n_rext_ro, 1u); else scan_add(&C.n_ro[tu.log2_size - 2], 1u); }
  S.alg_resid += coef_bytes;
  S.alg_intra += bpp * (4 * nT + 1) + bpp * nT * nT;
  // -- neighbour availability (8.4.4.2.2; intrapred.cc:437-527 preproc, :577-688 fill_from_image) as a unit bit mask.
  // The left column beside the TU, the row above it and the corner precede the TU in z-scan order whenever they lie in the
  // same slice and tile (Morton order), so only the below-left and above-right units need the MinTbAddrZS comparison
  // (a 4:2:2 chroma TU covers a 2:1 luma area: there every unit takes the comparison).
  const int sbw = c ? P.subw : 1, sbh = c ? P.subh : 1;
  const int xB = tu.x0, yB = tu.y0, xL = xB * sbw, yL = yB * sbh;
  const bool full_z = c != 0 && P.cf == 2;
  const int cx = xL >> P.lc, cy = yL >> P.lc;
  const int corner = nT >> 1, w4 = (P.width + 3) >> 2;
  const uint32_t own = B.ctb_group[ctu];
  const bool aL = xL > 0 && B.ctb_group[((xL - 1) >> P.lc) + cy * P.ctbs_w] == own;
  const bool aT = yL > 0 && B.ctb_group[cx + ((yL - 1) >> P.lc) * P.ctbs_w] == own;
  const bool aTL = xL > 0 && yL > 0 && B.ctb_group[((xL - 1) >> P.lc) + ((yL - 1) >> P.lc) * P.ctbs_w] == own;
  const bool aTR = yL > 0 && (xL + nT * sbw < P.width) && B.ctb_group[((xL + nT * sbw) >> P.lc) + ((yL - 1) >> P.lc) * P.ctbs_w] == own;
  int nBottom = (P.height - yL + sbh - 1) >> (sbh - 1); if (nBottom > 2 * nT) nBottom = 2 * nT;      // (sbw, sbh are 1 or 2)
  int nRight = (P.width - xL + sbw - 1) >> (sbw - 1);   if (nRight > 2 * nT) nRight = 2 * nT;
  const int cur = scan_zs(P, B, xL >> P.lt, yL >> P.lt);
  const bool cip = P.flags & SCANF_CIP;
  auto intra_ok = [&](int xs, int ys) {                 // constrained_intra_pred: only samples of intra CUs (intrapred.cc:612-615)
    return !cip || (B.blk_flags[((xs * sbw) >> 2) + ((ys * sbh) >> 2) * w4] & DE265HIP_BLK_INTRA);
  };
  auto z_ok = [&](int xs, int ys) { return scan_zs(P, B, (xs * sbw) >> P.lt, (ys * sbh) >> P.lt) <= cur; };
  uint64_t mask = 0;
  if (aL) {
    for (int y = nT - 1; y >= 0; y -= 4) if ((!full_z || z_ok(xB - 1, yB + y)) && intra_ok(xB - 1, yB + y)) mask |= 1ull << ((2 * nT - 1 - y) >> 2);
    for (int y = nBottom - 1; y >= nT; y -= 4) if (z_ok(xB - 1, yB + y) && intra_ok(xB - 1, yB + y)) mask |= 1ull << ((2 * nT - 1 - y) >> 2);
  }
  if (aTL && (!full_z || z_ok(xB - 1, yB - 1)) && intra_ok(xB - 1, yB - 1)) mask |= 1ull << corner;
  if (aT) for (int x = 0; x < nT; x += 4) if ((!full_z || z_ok(xB + x, yB - 1)) && intra_ok(xB + x, yB - 1)) mask |= 1ull << (corner + 1 + (x >> 2));
  if (aTR) for (int x = nT; x < nRight; x += 4) if (z_ok(xB + x, yB - 1) && intra_ok(xB + x, yB - 1)) mask |= 1ull << (corner + 1 + (x >> 2));
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

// ------------------------------------------------------------------------------------------------ pass 2: bases
// (host form: the serial loop; the device form is a chunked workgroup scan in k_scan.hip over the same arithmetic)
SCAN_FN void scan_prefix_finish_totals(const ScanBufs& B, const uint32_t tot[7])
{
  // level-0 list sorted [32x32 | 16x16 | 8x8 | 4x4]: the class starts are known once the class totals are
  ScanCounts& K = *B.counts;
  for (int k = 0; k < 4; k++) K.n_l0_size[k] = tot[k];
  K.n_l0_rext = tot[4]; K.n_intra = tot[5];
}
SCAN_FN uint32_t scan_l0_class_start(const uint32_t n_l0_size[4], int k)
{
  uint32_t at = 0;
  for (int q = 3; q > k; q--) at += n_l0_size[q];
  return at;
}

// ------------------------------------------------------------------------------------------------ pass 3: per CTB
SCAN_FN void scan_ctb(const ScanParams& P, const ScanBufs& B, int rs)
{
  if (B.counts->status) return;
  ScanCtb& C = B.ctb[rs];
  C.n_runs = 0;
  if (C.seen == 0) return;
  if (C.seen != 1 || C.end_tu <= C.first_tu || C.end_tu > (uint32_t)P.n_tus) { scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); return; }
  uint32_t cls_start[4];
  for (int k = 0; k < 4; k++) cls_start[k] = scan_l0_class_start(B.counts->n_l0_size, k);
  uint32_t inter_at[4] = { 0, 0, 0, 0 }, rext_at = 0;
  int cur_run[3] = { -1, -1, -1 };
  int n_local = 0;
  const int cx0 = (rs % P.ctbs_w) << P.lc, cy0 = (rs / P.ctbs_w) << P.lc;      // luma origin of the CTB
  for (uint32_t i = C.first_tu; i < C.end_tu; i++) {
    const de265hip_tu tu = B.tus[i];
    int rx = 0;
    const int cls = scan_tu_class(P, B, tu, &rx);
    if (cls == 0) continue;
    if (cls == 1) { B.l0[cls_start[tu.log2_size - 2] + C.l0_base[tu.log2_size - 2] + inter_at[tu.log2_size - 2]++] = scan_task_of(tu); continue; }
    if (cls == 2) {
      TuTask t = scan_task_of(tu);
      uint64_t luma_info = 0; int rx_luma = 0;
      if (rx & D265_RX_XCC) scan_xcc_luma(P, B, (int)i, &luma_info, &rx_luma);
      t.pad3 = (uint8_t)(rx | rx_luma); t.angle = tu.res_scale_val; t.avail = luma_info;
      B.l0x[C.rext_base + rext_at++] = t;
      continue;
    }
    // ---- an intra TU: which run, which in-run level (host.hip, round 3: "which run")
    const int c = tu.c_idx, nT = 1 << tu.log2_size, xB = tu.x0, yB = tu.y0;
    const int sbw = c ? P.subw : 1, sbh = c ? P.subh : 1;
    const int mw = P.map_w[c];
    const int cw = c ? P.cwid : P.width, ch = c ? P.chei : P.height;
    ScanCell* cells = B.cell[c];
    const uint64_t mask = B.tu_avail[i], need0 = B.tu_need[i];
    // a cell of this CTB that scan_ctb has passed: its run number and in-run level; -1: no intra TU of this picture there
    // (or one of another CTB: -2)
    auto cell_run = [&](int cell, int* llvl) -> int {
      const ScanCell v = cells[cell];
      if ((uint32_t)v == 0) return -1;
      const int x4 = cell % mw, y4 = cell / mw;
      const bool mine = (((x4 * 4 * sbw) >> P.lc) << P.lc) == cx0 && (((y4 * 4 * sbh) >> P.lc) << P.lc) == cy0;
      if (!mine) return -2;
      const uint32_t hi = (uint32_t)(v >> 32);
      if (!(hi >> 31)) return -1;                        // (a TU later in decode order: malformed overlap; treated as absent)
      *llvl = (int)((hi >> 16) & 0xFF);
      return (int)(hi & 0xFFFF);
    };
    const int crun = cur_run[c];
    bool reads_cur = false, foreign = false, multi = false;
    int llev = 0, p0 = -1, n_nonlocal = 0;
    for (uint64_t need = need0; need; need &= need - 1) {
      int lv = 0;
      const int cr = cell_run(scan_cell_of(scan_ctz64(need), xB, yB, nT, mw), &lv);
      if (cr == -1) { foreign = true; continue; }
      if (cr == -2) { n_nonlocal++; continue; }
      if (cr == crun) { if (lv > llev) llev = lv; reads_cur = true; }
      if (p0 < 0) p0 = cr; else if (cr != p0) multi = true;
    }
    llev += 1;
    // -- the current run of its component if the TU reads from it (decided on the FULL neighbourhood, so that an all-intra
    // CTB stays one run per component) ...
    int r = crun;
    bool extends = r >= 0 && B.run_ntus[C.intra_base + r] < 255;       /* RUN_MAX_TUS of k_run; positions + 1 fit a byte */
    if (extends && !reads_cur) {
      extends = false;
      for (uint64_t mm = mask & ~need0; mm && !extends; mm &= mm - 1) { int lv; extends = cell_run(scan_cell_of(scan_ctz64(mm), xB, yB, nT, mw), &lv) == r; }
    }
    // ... else, reading from exactly ONE run - of this CTB - it joins that run instead of starting its own (a hand-over
    // between two runs costs ~12 us of dependent memory round trips, an in-run level 0.4 us)
    bool merged = false;
    if (!extends && (P.flags & SCANF_MERGE) && p0 >= 0 && !multi && n_nonlocal == 0 && B.run_ntus[C.intra_base + p0] < 255) {
      const int x = p0;
      int lx = 0;
      const int ux0 = (xB - 4 > 0 ? xB - 4 : 0) >> 2, uy0 = (yB - 4 > 0 ? yB - 4 : 0) >> 2;
      const int ux1 = (cw - 1 < xB + 2 * nT + 3 ? cw - 1 : xB + 2 * nT + 3) >> 2, uy1 = (ch - 1 < yB + 2 * nT + 3 ? ch - 1 : yB + 2 * nT + 3) >> 2;
      if (uy0 < (yB >> 2)) for (int x4 = ux0; x4 <= ux1; x4++) { int lv = 0; if (cell_run(x4 + uy0 * mw, &lv) == x && lv > lx) lx = lv; }
      if (ux0 < (xB >> 2)) for (int y4 = uy0; y4 <= uy1; y4++) { int lv = 0; if (cell_run(ux0 + y4 * mw, &lv) == x && lv > lx) lx = lv; }
      if (lx + 1 <= 250) { r = x; llev = lx + 1; merged = true; }
    }
    if (!extends && !merged) {                         // a new run
      r = n_local++;
      B.run_ntus[C.intra_base + r] = 0;
      B.run_rs[C.intra_base + r] = (uint32_t)rs;
      cur_run[c] = r;
      llev = 1;
    }
    B.run_ntus[C.intra_base + r]++;
    B.tu_info[i] = (uint32_t)r | ((uint32_t)llev << 16) | (foreign ? SCAN_TI_FOREIGN : 0u) | SCAN_TI_INTRA;
    B.tu_run[i] = C.intra_base + (uint32_t)r;
    const ScanCell hi = (ScanCell)((uint32_t)r | ((uint32_t)llev << 16) | (1u << 31)) << 32;
    for (int y = yB >> 2; y < (yB + nT) >> 2; y++)
      for (int x = xB >> 2; x < (xB + nT) >> 2; x++) cells[x + (size_t)y * mw] = hi | (uint32_t)(i + 1);
  }
  C.n_runs = (uint32_t)n_local;
  scan_add(&B.counts->n_runs, (uint32_t)n_local);
}

// ------------------------------------------------------------------------------------------------ tickets
// k_run's workers draw tickets in slot order and a run may only wait for runs in earlier tickets (or in lower slots of its own
// ticket: the wavefront of slot q works slot q + 4 off after it).  Round 3 ordered the runs by their level in the run graph
// (longest producer chain: a propagation over the whole graph).  No walk is needed: a run reads from runs of its own CTB
// created before it and from runs of the CTBs to the left, above-left, above and above-right - all of which lie on smaller
// anti-diagonals x + 2y - so (anti-diagonal of the CTB, CTB, number inside the CTB) is a topological order, and the runs of
// one anti-diagonal are exactly the ones the wavefront of an all-intra picture has ready together.  ctb_order lists the CTBs
// by (x + 2y, y); walking it, micro runs fill the open ticket slot by slot, an ordinary run closes it and takes one alone.
struct ScanTicketState { uint32_t tickets, fill; };    // tickets opened so far; slots taken in the last one (0: it is closed / full)
SCAN_FN void scan_ticket_step(ScanTicketState& st, bool micro, uint32_t* ticket, uint32_t* slot)
{
  if (micro) {
    if (st.fill == 0) st.tickets++;
    *ticket = st.tickets - 1; *slot = st.fill;
    st.fill = (st.fill + 1) % RUN_TICKET_SLOTS;
  } else {
    st.tickets++; st.fill = 0;
    *ticket = st.tickets - 1; *slot = 0;
  }
}

// ------------------------------------------------------------------------------------------------ pass 4: per run
// the run with sparse id s: its record, its TUs in chain order, the residual-only copies of its TUs, its producers
SCAN_FN void scan_run(const ScanParams& P, const ScanBufs& B, uint32_t s)
{
  if (B.counts->status || B.run_ntus[s] == 0) return;
  const int rs = (int)B.run_rs[s];
  const ScanCtb& C = B.ctb[rs];
  const int r = (int)(s - C.intra_base);
  // ---- its TUs (decode order), and what the runs before it in this CTB take of the CTB's lists
  int tix[256];
  int n = 0, n_before = 0, c = 0;
  uint32_t samp_before = 0, ro_before[4] = { 0, 0, 0, 0 }, rext_ro_before = 0;
  int x0 = 1 << 30, y0 = 1 << 30, x1 = 0, y1 = 0, wx1 = 0, wy1 = 0, own_samples = 0, nl = 0;
  bool foreign = false, big = false, too_big = false;
  for (uint32_t i = C.first_tu; i < C.end_tu; i++) {
    const uint32_t ti = B.tu_info[i];
    if (!(ti & SCAN_TI_INTRA)) continue;
    const de265hip_tu tu = B.tus[i];
    const int rx = scan_rx_bits(P, B, tu);
    const int rr = (int)SCAN_TI_RUN(ti), nT = 1 << tu.log2_size;
    const bool cbf = (tu.flags & DE265HIP_TU_CBF) && tu.n_coeff;
    const bool ro = cbf || (rx & D265_RX_XCC);
    if (rr < r) {
      n_before++; samp_before += (uint32_t)(nT * nT);
      if (ro) { if (rx) rext_ro_before++; else ro_before[tu.log2_size - 2]++; }
      continue;
    }
    if (rr != r) continue;
    if (n < 256) tix[n] = (int)i;
    n++;
    c = tu.c_idx;
    x0 = x0 < tu.x0 ? x0 : tu.x0; y0 = y0 < tu.y0 ? y0 : tu.y0;
    x1 = x1 > tu.x0 + nT ? x1 : tu.x0 + nT; y1 = y1 > tu.y0 + nT ? y1 : tu.y0 + nT;
    wx1 = wx1 > tu.x0 + 2 * nT ? wx1 : tu.x0 + 2 * nT; wy1 = wy1 > tu.y0 + 2 * nT ? wy1 : tu.y0 + 2 * nT;
    own_samples += nT * nT;
    if ((int)SCAN_TI_LLEV(ti) > nl) nl = (int)SCAN_TI_LLEV(ti);
    foreign = foreign || (ti & SCAN_TI_FOREIGN);
    big = big || tu.log2_size == 4; too_big = too_big || tu.log2_size > 4;
  }
  if (n == 0 || n > 255 || n != (int)B.run_ntus[s]) { scan_fail(B, DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE); return; }
  if (nl > 256 || nl - 1 > 255) { scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
  // ---- micro run: <= 16 TUs of <= 8x8 in a 32x32 box; 16x16 TUs too when the run's samples fit the wavefront's residual
  // slice (1024) and its window the wavefront's slice of the window array (k_tu.hip: MICRO_P, MICRO_H)
  bool micro = !(P.flags & SCANF_MICRO_OFF) && n <= P.micro_tus && x1 - x0 <= 32 && y1 - y0 <= 32 && !too_big;
  if (micro && big) {
    if (!(P.flags & SCANF_MICRO16) || own_samples > 1024) micro = false;
    else {
      const int ax0 = (x0 - 1) & ~7, wxc = wx1 < x1 + 32 ? wx1 : x1 + 32, wyc = wy1 < y1 + 32 ? wy1 : y1 + 32;
      const int cols = wxc - ax0, rows = wyc - (y0 - 1);
      micro = cols <= 56 && rows <= 41 && ((cols + 7) >> 3) * rows <= 256;
    }
  }
  // ---- dense: the run's TUs cover its whole bounding box AND every available neighbour outside the box lies on the row
  // above it or the column left of it
  bool dense = own_samples == (x1 - x0) * (y1 - y0) && !(P.flags & SCANF_NO_DENSE);
  for (int k = 0; dense && k < n; k++) {
    const de265hip_tu tu = B.tus[tix[k]];
    const uint64_t avail = B.tu_avail[tix[k]];
    const int nT = 1 << tu.log2_size, xB = tu.x0, yB = tu.y0, corner = nT >> 1;
    if (xB > x0 && yB + 2 * nT > y1) {
      int umax = (yB + 2 * nT - 1 - y1) >> 2; if (umax > corner - 1) umax = corner - 1;
      if (avail & ((2ull << umax) - 1ull)) dense = false;
    }
    if (yB > y0 && xB + 2 * nT > x1) {
      int kmin = (x1 - xB) >> 2; if (kmin < 0) kmin = 0;
      if (kmin < corner && ((avail >> (corner + 1 + kmin)) & ((1ull << (corner - kmin)) - 1ull))) dense = false;
    }
  }
  RunTask o;
  o.x0 = (uint16_t)x0; o.y0 = (uint16_t)y0; o.x1 = (uint16_t)x1; o.y1 = (uint16_t)y1;
  o.wx1 = (uint16_t)(wx1 < x1 + 32 ? wx1 : x1 + 32); o.wy1 = (uint16_t)(wy1 < y1 + 32 ? wy1 : y1 + 32);
  o.c_idx = (uint8_t)c; o.micro = (uint8_t)((micro ? 1 : 0) | (dense ? 2 : 0)); o.n_tus = (uint16_t)n;
  o.first_tu = C.intra_base + (uint32_t)n_before;
  o.res_offset = C.isamp_base + samp_before;
  o.dep_offset = 0; o.n_deps = 0;
  // ---- the mailbox of an ordinary dense run (whether anybody reads it is settled in scan_run2 / scan_run3), and - phased
  // hand-over - the ready epochs of its 64 edge packets: the barrier epoch of the TU under each pair of samples of its bottom
  // row and right column
  uint32_t mb_id = 0xFFFFFFFFu;
  if ((P.flags & SCANF_MAILBOX) && !micro && dense) {
    mb_id = scan_add(&B.counts->n_mailboxes, 1u);
    if (mb_id >= P.cap_mb) { scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
    if ((P.flags & SCANF_MB_PHASES) && c == 0 && x1 - x0 <= 64 && y1 - y0 <= 64) {
      uint8_t* rdy = B.rdy_tab + 64 * (size_t)mb_id;
      for (int q = 0; q < 64; q++) rdy[q] = 255;        // (255: no packet there - or, in a malformed description, no TU under it)
      for (int k = 0; k < n; k++) {
        const de265hip_tu tu = B.tus[tix[k]];
        const int nT = 1 << tu.log2_size;
        const uint8_t ep = (uint8_t)(SCAN_TI_LLEV(B.tu_info[tix[k]]) - 1);
        if (tu.y0 + nT == y1) for (int q = 0; q < (nT >> 1); q++) rdy[((tu.x0 - x0) >> 1) + q] = ep;
        if (tu.x0 + nT == x1) for (int q = 0; q < (nT >> 1); q++) rdy[32 + ((tu.y0 - y0) >> 1) + q] = ep;
      }
    }
  }
  B.mbx[3 * (size_t)s] = mb_id; B.mbx[3 * (size_t)s + 1] = 0xFFFFFFFFu; B.mbx[3 * (size_t)s + 2] = 0xFFFFFFFFu;
  B.pub_flag[s] = 0;
  // ---- order of the run's TUs in its record: one list per wavefront (list w = the TUs dealt to wavefront w), each in in-run
  // level order, then the collective list (16x16 / 32x32 TUs, reconstructed by all wavefronts together).  The TUs of one
  // in-run level are independent of each other and are dealt round-robin to the wavefronts; the chain passes one workgroup
  // barrier per level.  One sort by (list, level, decode index).
  uint32_t keys[256]; uint8_t rank[260];
  for (int q = 0; q <= nl; q++) rank[q] = 0;
  const int nwv = micro ? 1 : P.run_waves;
  const int n_epochs = nl > 0 ? nl - 1 : 0;
  for (int k = 0; k < n; k++) {
    const int lev = (int)SCAN_TI_LLEV(B.tu_info[tix[k]]);
    const int list = (B.tus[tix[k]].log2_size > 3 && !micro) ? 4 : rank[lev]++ % nwv;
    keys[k] = ((uint32_t)list << 20) | ((uint32_t)lev << 8) | (uint32_t)k;
  }
  // (insertion sort: the keys come nearly sorted - decode order follows the levels - and a run has few TUs)
  for (int a = 1; a < n; a++) { const uint32_t v = keys[a]; int b = a - 1; while (b >= 0 && keys[b] > v) { keys[b + 1] = keys[b]; b--; } keys[b + 1] = v; }
  {
    int pos = 0;
    for (int w = 0; w < 4; w++) { while (pos < n && (int)(keys[pos] >> 20) <= w) pos++; o.wave_end[w] = (uint16_t)pos; }
  }
  o.n_lvls = (uint16_t)n_epochs;
  scan_add(&B.counts->sum_lvls, (uint32_t)nl);
  // ---- the run-ordered TU records + the residual-only copies (level-0 tasks)
  uint32_t cls_start[4];
  for (int k = 0; k < 4; k++) cls_start[k] = scan_l0_class_start(B.counts->n_l0_size, k);
  uint32_t samp = 0, ro_at[4] = { 0, 0, 0, 0 }, rext_at = 0;
  for (int oi = 0; oi < n; oi++) {
    const int i = tix[keys[oi] & 0xFFu];
    const de265hip_tu tu = B.tus[i];
    TuTask tt = scan_task_of(tu);
    const int m = tu.intra_mode < 35 ? tu.intra_mode : 1;
    tt.angle = (int8_t)scan_intra_angle(m); tt.inv_angle = (int16_t)scan_inv_angle(m);
    tt.avail = B.tu_avail[i];
    tt.run_level = (uint8_t)(SCAN_TI_LLEV(B.tu_info[i]) - 1);      // barrier epoch = in-run level - 1
    const uint32_t coeff_offset = tt.coeff_offset;
    tt.resid_offset = o.res_offset + samp;
    tt.coeff_offset = samp; samp += 1u << (2 * tt.log2_size);
    int trx = scan_rx_bits(P, B, tu);
    if ((tt.flags & DE265HIP_TU_CBF) || (trx & D265_RX_XCC)) {
      TuTask ro = tt; ro.flags |= D265_TU_RESID_ONLY; ro.coeff_offset = coeff_offset; ro.run_level = 0;
      if (trx) {
        uint64_t luma_info = 0; int rx_luma = 0;
        if (trx & D265_RX_XCC) scan_xcc_luma(P, B, i, &luma_info, &rx_luma);
        ro.pad3 = (uint8_t)(trx | rx_luma); ro.angle = 0; ro.avail = 0;
        if (trx & D265_RX_XCC) { ro.angle = tu.res_scale_val; ro.avail = luma_info; }
        B.l0x[C.rext_base + C.n_rext_inter + rext_ro_before + rext_at++] = ro;
      } else {
        const int k = ro.log2_size - 2;
        B.l0[cls_start[k] + C.l0_base[k] + C.n_inter[k] + ro_before[k] + ro_at[k]++] = ro;
      }
      tt.flags |= DE265HIP_TU_CBF;                     // (the run kernels read the residual block whenever there is one)
    }
    B.run_tus[o.first_tu + (uint32_t)oi] = tt;
  }
  o.n_samples = samp;
  // ---- producers: the runs whose samples its TUs need, each once (sparse ids).  Room for every needed unit of every TU is
  // taken from the pool; the list is deduplicated in place.
  uint32_t cand = 0;
  for (int k = 0; k < n; k++) cand += (uint32_t)scan_popc64(B.tu_need[tix[k]]);
  uint32_t nd = 0;
  if (cand) {
    const uint32_t at = scan_add(&B.counts->n_deps_alloc, cand);
    if (at + cand > P.cap_deps) { scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
    o.dep_offset = at;
    uint32_t* dl = B.deps + at;
    const int mw = P.map_w[c];
    for (int k = 0; k < n; k++) {
      const de265hip_tu tu = B.tus[tix[k]];
      for (uint64_t need = B.tu_need[tix[k]]; need; need &= need - 1) {
        const ScanCell v = B.cell[c][scan_cell_of(scan_ctz64(need), tu.x0, tu.y0, 1 << tu.log2_size, mw)];
        if ((uint32_t)v == 0) continue;
        const uint32_t j = (uint32_t)v - 1;
        const uint32_t tj = B.tu_info[j];
        if (!(tj & SCAN_TI_INTRA)) continue;
        const uint32_t ps = B.tu_run[j];
        if (ps == s) continue;
        bool seen = false;
        for (uint32_t q = 0; q < nd && !seen; q++) seen = dl[q] == ps;
        if (!seen) dl[nd++] = ps;
      }
    }
  }
  if (nd > 0xFFFFu) { scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
  o.n_deps = (uint16_t)nd;
  B.run_nall[s] = nd | (foreign ? 0x80000000u : 0u);
  // ---- a front run: a micro run without producers among the intra runs (92 % of a B picture's runs): reconstructed by
  // k_intra_front ahead of k_run, no ticket, no flag
  const bool front = micro && nd == 0 && !(P.flags & SCANF_FRONT_OFF);
  unsigned long long alg = 0;
  {
    const unsigned long long bpp = (unsigned long long)(c ? P.bppC : P.bppY);
    for (int k = 0; k < n; k++) { const unsigned long long nT = 1ull << B.tus[tix[k]].log2_size; alg += bpp * (4 * nT + 1) + bpp * nT * nT; }
  }
  if (front) {
    o.micro |= RUN_MICRO_FRONT;
    B.front_idx[scan_add(&B.counts->n_front, 1u)] = s;
    scan_add64(&B.counts->alg_intra_front, alg);
  }
  B.run_list[scan_add(&B.counts->n_listed, 1u)] = s;
  B.runs[s] = o;
}

// ------------------------------------------------------------------------------------------------ pass 5: per run
// producers that are front runs leave the list (the kernel boundary orders them); a dense ordinary run whose every neighbour
// sample comes from the bottom row / right column of dense ordinary runs takes them from those runs' mailboxes: its segment
// list and - phased hand-over - when each neighbour sample is first needed (host.hip, round 3: "Edge mailboxes")
// (pre_need: 40 + 40 + 1 need epochs of the units on the row above, the column beside and the corner of the run's box, where
//  the caller has computed them already - the device does, a lane per TU -; else the loop below does)
SCAN_FN void scan_run2(const ScanParams& P, const ScanBufs& B, uint32_t s, const TuTask* own_tus = nullptr, const uint32_t* pre_need = nullptr)
{
  // (own_tus: the run's TU records where the caller has staged them - LDS on the device -, else they are read from run_tus)
  if (B.counts->status || B.run_ntus[s] == 0) return;
  RunTask o = B.runs[s];
  const uint32_t n_all = B.run_nall[s] & 0x7FFFFFFFu;
  const bool foreign = B.run_nall[s] >> 31;
  uint32_t* dl = B.deps + o.dep_offset;
  // partition: [producers k_run waits for | front runs]; the levels still count every producer
  uint32_t nd = 0;
  for (uint32_t q = 0; q < n_all; q++) {
    const uint32_t p = dl[q];
    if (!(B.runs[p].micro & RUN_MICRO_FRONT)) { dl[q] = dl[nd]; dl[nd++] = p; }
  }
  o.n_deps = (uint16_t)nd;
  if ((P.flags & SCANF_DROP_PRODUCER) && nd) {           // fault injection: the smallest run somebody waits for is never executed
    uint32_t v = 0xFFFFFFFFu;
    for (uint32_t q = 0; q < nd; q++) if (dl[q] < v) v = dl[q];
    scan_min(&B.counts->victim, v);
  }
  const bool micro = o.micro & 1, dense = o.micro & 2;
  const int n = o.n_tus, nl = (int)o.n_lvls + 1;
  if ((P.flags & SCANF_MAILBOX) && !micro && dense && !foreign && nd > 0 && nd == n_all && nd <= 8) {
    const int c = o.c_idx;
    const int ax0 = ((int)o.x0 - 1) & ~7, wy0 = (int)o.y0 - 1, tile_p = (64 + 40 + 7) & ~7;      // RUN_TILE_P_OF(64) of k_run
    const int cw_ = c ? P.cwid : P.width, ch_ = c ? P.chei : P.height;
    const int wx1c = (int)o.wx1 < cw_ ? (int)o.wx1 : cw_, wy1c = (int)o.wy1 < ch_ ? (int)o.wy1 : ch_;
    uint32_t seg[2 * 16], seg_run[16]; int nseg = 0; bool ok = true;
    for (uint32_t q = 0; q < nd && ok; q++) {
      const uint32_t pk = dl[q];
      const RunTask Pq = B.runs[pk];
      if ((Pq.micro & 3) != 2) { ok = false; break; }                 // producer: ordinary and dense
      bool any = false;
      if ((int)Pq.y0 <= wy0 && wy0 < (int)Pq.y1) {                    // the row above the box
        const int xs = (int)Pq.x0 > (int)o.x0 - 1 ? (int)Pq.x0 : (int)o.x0 - 1, xe = (int)Pq.x1 < wx1c ? (int)Pq.x1 : wx1c;
        if (xs < xe) {
          if ((int)Pq.y1 - 1 != wy0 || nseg == 16) { ok = false; break; }
          seg[2 * nseg] = ((uint32_t)(xe - xs - 1) << 24); seg[2 * nseg + 1] = (uint32_t)(xs - Pq.x0) | ((uint32_t)(xs - ax0) << 8);
          seg_run[nseg++] = pk; any = true;
        }
      }
      if ((int)Pq.x0 <= (int)o.x0 - 1 && (int)o.x0 - 1 < (int)Pq.x1) {  // the column left of it
        const int ys = (int)Pq.y0 > (int)o.y0 ? (int)Pq.y0 : (int)o.y0, ye = (int)Pq.y1 < wy1c ? (int)Pq.y1 : wy1c;
        if (ys < ye) {
          if ((int)Pq.x1 != (int)o.x0 || nseg == 16) { ok = false; break; }
          seg[2 * nseg] = ((uint32_t)(ye - ys - 1) << 24) | 0x80000000u;
          seg[2 * nseg + 1] = (uint32_t)(ys - Pq.y0) | ((uint32_t)((ys - wy0) * tile_p + ((int)o.x0 - 1 - ax0)) << 8);
          seg_run[nseg++] = pk; any = true;
        }
      }
      if (!any) ok = false;
    }
    if (ok && nseg > 0) {
      o.micro |= 4;
      for (int q = 0; q < nseg; q++) {                    // producer run -> its mailbox; it learns that it is read (scan_run3)
#if SCAN_DEVICE
        {
          // (the first reader that flags a run puts it on the publishers' list - lvl_cnt's memory -, which is all the ticket pass
          //  then walks: looking at every listed run's flag was eight pairs of dependent loads per thread of its one workgroup)
          const uint32_t pr = seg_run[q];
          uint32_t* word = reinterpret_cast<uint32_t*>(B.pub_flag + (pr & ~3u));
          const uint32_t bit = 1u << (8 * (pr & 3u));
          if (!(atomicOr(word, bit) & bit)) B.lvl_cnt[atomicAdd(&B.counts->n_pub, 1u)] = pr;
        }
#else
        B.pub_flag[seg_run[q]] = 1;
#endif
        seg[2 * q] |= B.mbx[3 * (size_t)seg_run[q]] & 0xFFFFFFu;
      }
      // -- when is each neighbour sample first needed?  Only TUs on the box's left column / top row read outside it (dense run)
      uint8_t need_row[256], need_col[256];               // by x - (x0 - 1) / y - y0; 255: never read
      uint32_t sub[2 * 48]; uint8_t sub_g[48]; int nsub = 0;
      uint8_t polls[4] = { 0, 0, 0, 0 }; int n_groups = 1;
      bool phased = (P.flags & SCANF_MB_PHASES) && c == 0 && nl >= 4;
      if (phased) {
        uint8_t nru[40], ncu[40], ncorner = 255;
        for (int q = 0; q < 40; q++) { nru[q] = 255; ncu[q] = 255; }
        if (pre_need) { for (int q = 0; q < 40; q++) { nru[q] = (uint8_t)pre_need[q]; ncu[q] = (uint8_t)pre_need[40 + q]; } ncorner = (uint8_t)pre_need[80]; }
        else
        for (int k = 0; k < n; k++) {
          const TuTask tt = own_tus ? own_tus[k] : B.run_tus[o.first_tu + (uint32_t)k];
          const int xB = tt.x0, yB = tt.y0;
          if (xB != (int)o.x0 && yB != (int)o.y0) continue;
          const int nT = 1 << tt.log2_size, corner = nT >> 1, m = tt.intra_mode < 35 ? tt.intra_mode : 1;
          const uint8_t ep = tt.run_level;
          uint64_t need = (P.flags & SCANF_MODE_DEPS) ? scan_needed_units(B.used_units[((tt.log2_size - 2) * 35 + m) * 2 + 1], tt.avail) : tt.avail;
          for (; need; need &= need - 1) {
            const int u = scan_ctz64(need);
            if (u < corner) {
              if (xB != (int)o.x0) continue;
              const int j = (yB + 2 * nT - 4 * u - 4 - (int)o.y0) >> 2;
              if (j >= 0 && j < 40 && ep < ncu[j]) ncu[j] = ep;
            } else if (u == corner) {
              if (yB == (int)o.y0) { if (xB == (int)o.x0) { if (ep < ncorner) ncorner = ep; } else { const int j = (xB - 1 - (int)o.x0) >> 2; if (j < 40 && ep < nru[j]) nru[j] = ep; } }
              else if (xB == (int)o.x0) { const int j = (yB - 1 - (int)o.y0) >> 2; if (j >= 0 && j < 40 && ep < ncu[j]) ncu[j] = ep; }
            } else {
              if (yB != (int)o.y0) continue;
              const int j = (xB + 4 * (u - corner - 1) - (int)o.x0) >> 2;
              if (j >= 0 && j < 40 && ep < nru[j]) nru[j] = ep;
            }
          }
        }
        need_row[0] = ncorner;
        for (int j = 0; j < 40; j++) for (int q = 0; q < 4; q++) { need_row[1 + 4 * j + q] = nru[j]; need_col[4 * j + q] = ncu[j]; }
        // the samples' need epochs -> at most four poll points (quantiles of the distinct values)
        const uint8_t* nbase[16];
        uint64_t seen[4] = { 0, 0, 0, 0 };
        for (int q = 0; q < nseg; q++) {
          const RunTask Pq = B.runs[seg_run[q]];
          const int src = (int)(seg[2 * q + 1] & 63), cnt = (int)((seg[2 * q] >> 24) & 63) + 1;
          nbase[q] = (seg[2 * q] >> 31) ? need_col + ((int)Pq.y0 + src - (int)o.y0) : need_row + ((int)Pq.x0 + src - ((int)o.x0 - 1));
          for (int off = 0; off < cnt; off++) { const uint8_t v = nbase[q][off]; seen[v >> 6] |= 1ull << (v & 63); }
        }
        seen[3] &= ~(1ull << 63);                          // (255: never read)
        uint8_t vals[256]; int nv = 0;
        for (int wd = 0; wd < 4; wd++) for (uint64_t mm = seen[wd]; mm; mm &= mm - 1) vals[nv++] = (uint8_t)(64 * wd + scan_ctz64(mm));
        if (nv < 2) phased = false;
        else {
          n_groups = nv < 4 ? nv : 4;
          for (int g2 = 0; g2 < n_groups; g2++) polls[g2] = vals[(g2 * nv) / n_groups];
          const int p1 = n_groups > 1 ? polls[1] : 256, p2 = n_groups > 2 ? polls[2] : 256, p3 = n_groups > 3 ? polls[3] : 256;
          auto grp_of_v = [&](int v) { return v == 255 ? 255 : (v >= p1) + (v >= p2) + (v >= p3); };
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
      int tot = 0, ends[4] = { 0, 0, 0, 0 };
      if (phased && nsub > 0) {
        for (int g2 = 0; g2 < n_groups; g2++) { for (int q = 0; q < nsub; q++) if (sub_g[q] == g2) tot += (int)((sub[2 * q] >> 24) & 63) + 1; ends[g2] = tot; }
        if (tot > 255) phased = false;                     // (cannot happen with 64x64 boxes: <= 193 neighbour samples)
      }
      const bool use_sub = phased && nsub > 0;
      const uint32_t words = 3u + 2u * (uint32_t)(use_sub ? nsub : nseg);
      const uint32_t at = scan_add(&B.counts->n_segs_alloc, words);
      if (at + words > P.cap_segs) { scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
      B.mbx[3 * (size_t)s + 1] = at;
      uint32_t* ms = B.mb_segs + at;
      if (use_sub) {
        uint32_t w = 3;
        for (int g2 = 0; g2 < n_groups; g2++)
          for (int q = 0; q < nsub; q++) if (sub_g[q] == g2) { ms[w++] = sub[2 * q]; ms[w++] = sub[2 * q + 1]; }
        for (int g2 = n_groups; g2 < 4; g2++) { ends[g2] = tot; polls[g2] = 255; }
        ms[0] = (uint32_t)nsub | ((uint32_t)n_groups << 8);
        ms[1] = (uint32_t)ends[0] | ((uint32_t)ends[1] << 8) | ((uint32_t)ends[2] << 16) | ((uint32_t)ends[3] << 24);
        ms[2] = (uint32_t)polls[0] | ((uint32_t)polls[1] << 8) | ((uint32_t)polls[2] << 16) | ((uint32_t)polls[3] << 24);
      } else {
        ms[0] = (uint32_t)nseg | (1u << 8); ms[1] = 0; ms[2] = 0;      // (one group: everything at the start)
        for (int q = 0; q < 2 * nseg; q++) ms[3 + q] = seg[q];
      }
    }
  }
  // (only the fields that changed: other runs read this record's box and class bits in the same pass)
  B.runs[s].n_deps = o.n_deps;
  if (o.micro & 4) B.runs[s].micro = o.micro;
}

// ------------------------------------------------------------------------------------------------ pass 6: per run
// a run somebody reads through its mailbox publishes (micro bit 8); phased hand-over: at most three store points before the
// end of its chain (quantiles of the distinct ready epochs): a packet goes out at the first of them that is not before its
// ready epoch, the rest when the chain ends (255)
SCAN_FN void scan_run3(const ScanParams& P, const ScanBufs& B, uint32_t s)
{
  if (B.counts->status || B.run_ntus[s] == 0 || !B.pub_flag[s]) return;
  RunTask o = B.runs[s];
  o.micro |= 8;
  if ((P.flags & SCANF_MB_PHASES) && o.c_idx == 0 && (int)o.x1 - (int)o.x0 <= 64 && (int)o.y1 - (int)o.y0 <= 64) {
    uint8_t rdy[64];
    const uint8_t* src = B.rdy_tab + 64 * (size_t)B.mbx[3 * (size_t)s];
    for (int i = 0; i < 64; i++) rdy[i] = src[i];
    uint64_t seen_r[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 64; i++) seen_r[rdy[i] >> 6] |= 1ull << (rdy[i] & 63);
    uint8_t rv[256]; int nrv = 0;
    for (int wd = 0; wd < 4; wd++) for (uint64_t mm = seen_r[wd]; mm; mm &= mm - 1) {
      const int v = 64 * wd + scan_ctz64(mm);
      if (v < (int)o.n_lvls && v < 255) rv[nrv++] = (uint8_t)v;                                     // (epoch n_lvls is the end)
    }
    uint8_t pubs[4] = { 255, 255, 255, 255 };
    const int n_pub = nrv < 3 ? nrv : 3;
    for (int j = 0; j < n_pub; j++) pubs[j] = rv[((j + 1) * nrv) / n_pub - 1];
    for (int i = 0; i < 64; i++) {
      uint8_t qv = 255;
      for (int j = n_pub - 1; j >= 0; j--) if (rdy[i] <= pubs[j]) qv = pubs[j];
      rdy[i] = qv;
    }
    const uint32_t at = scan_add(&B.counts->n_segs_alloc, 17u);
    if (at + 17 > P.cap_segs) { scan_fail(B, DE265HIP_ERROR_NOT_IMPLEMENTED); return; }
    B.mbx[3 * (size_t)s + 2] = at;
    uint32_t* ms = B.mb_segs + at;
    for (int i = 0; i < 16; i++) ms[i] = (uint32_t)rdy[4 * i] | ((uint32_t)rdy[4 * i + 1] << 8) | ((uint32_t)rdy[4 * i + 2] << 16) | ((uint32_t)rdy[4 * i + 3] << 24);
    ms[16] = (uint32_t)pubs[0] | ((uint32_t)pubs[1] << 8) | ((uint32_t)pubs[2] << 16) | (255u << 24);
  }
  B.runs[s].micro = o.micro;
}

}  // namespace d265
