/*
 * synth.c -- synthetic command-buffer generator (see synth.h).
 * Walks a random CTU quadtree in decode (tile-scan) order like
 * read_coding_quadtree / read_coding_unit / read_transform_tree would
 * (slice.cc:4582, :4245, :3821) and records what the reconstruction consumes.
 */
#include "synth.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORD_PU  (1u<<28)
#define ORD_PCM (2u<<28)
#define ORD_TU  (3u<<28)

/* ---------- PRNG: splitmix64 ---------- */
typedef struct { uint64_t s; } rng_t;
static uint64_t rnd64(rng_t* r)
{
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static int rnd_int(rng_t* r, int lo, int hi) { return lo + (int)(rnd64(r) % (uint64_t)(hi - lo + 1)); }
static int rnd_pct(rng_t* r, int pct) { return (int)(rnd64(r) % 100) < pct; }
static double rnd_u(rng_t* r) { return (double)(rnd64(r) >> 11) * (1.0 / 9007199254740992.0); }
static double rnd_gauss(rng_t* r)
{
  double u1 = rnd_u(r) + 1e-12, u2 = rnd_u(r);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static int rnd_laplace(rng_t* r, double b)
{
  double u = rnd_u(r) - 0.5;
  double v = -b * (u < 0 ? -1.0 : 1.0) * log(1.0 - 2.0 * fabs(u) + 1e-12);
  return (int)lrint(v);
}

/* ---------- growable arrays ---------- */
#define VEC(T) struct { T* p; int n, cap; }
#define VPUSH(v, val) do { if ((v).n == (v).cap) { (v).cap = (v).cap ? (v).cap*2 : 1024; \
  (v).p = realloc((v).p, sizeof(*(v).p) * (size_t)(v).cap); } (v).p[(v).n++] = (val); } while (0)

struct synth_picture {
  synth_config cfg;
  de265hip_picture_desc desc;
  rng_t rng;
  int ctbs_w, ctbs_h, w4, h4, cbs_w, cbs_h, tbs_w, tbs_h;
  int* tile_id;
  int* ts2rs;
  de265hip_slice_params* slices; int n_slices;
  de265hip_ctb_info* ctbs;
  VEC(de265hip_tu) tus;
  VEC(int16_t) cval;
  VEC(uint16_t) cpos;
  VEC(de265hip_pu) pus;
  VEC(de265hip_pcm) pcms;
  VEC(uint16_t) pcm_samples;
  VEC(uint32_t) order;
  uint8_t* blk_flags; uint8_t* blk_flags_noedge; int8_t* blk_qp; de265hip_motion* blk_motion;
  uint8_t* cb_log2; uint8_t* cb_part; uint8_t* tu_split;
  uint8_t* scaling;
  /* per-CU state */
  int cur_slice, cur_qp, cur_bypass, cur_intra, cur_deblk_off;
  int cu_luma_mode[4], cu_chroma_mode[4], cu_chroma_derived[4];   /* chroma modes per PU quadrant (4:4:4 NxN: four; else [0]); derived: intra_chroma_pred_mode == 4 */
  int cf, sw, sh;                       /* chroma_format_idc, SubWidthC, SubHeightC */
};

static int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static const uint8_t tab_qpc[14] = { 29,30,31,32,33,33,34,34,35,35,36,36,37,37 };
static int table8_22(int q) { if (q < 30) return q; if (q >= 43) return q - 6; return tab_qpc[q - 30]; }

void synth_default_config(synth_config* c, int w, int h, int bd, int slice_type, uint64_t seed)
{
  memset(c, 0, sizeof(*c));
  c->width = w; c->height = h; c->bit_depth = bd;
  c->log2_ctb_size = 6; c->log2_min_tb_size = 2; c->log2_max_tb_size = 5;
  c->seed = seed; c->slice_type = slice_type;
  c->intra_pct = 15; c->n_ref_slots = 2; c->ref_slots[0] = 0; c->ref_slots[1] = 1;
  c->bi_pct = 60; c->mv_sigma_qpel = 12; c->weighted_pred = 0;
  c->n_slices = 1; c->tile_cols = 1; c->tile_rows = 1; c->slice_per_tile = 0;
  c->cbf_pct = 60; c->tskip_pct = 0; c->bypass_pct = 0; c->pcm_pct = 0;
  c->pcm_loop_filter_disable = 0; c->scaling_list = 0; c->constrained_intra_pred = 0;
  c->strong_intra_smoothing = 1; c->deblocking = 1; c->sao = 1;
  c->lf_across_slices_pct = 100; c->lf_across_tiles = 1; c->big_coeff_pct = 0;
  c->qp_min = 22; c->qp_max = 37; c->amp = 1; c->split_bias = 50;
}

/* ---------- metadata helpers ---------- */
static void fill_blk(synth_picture* s, int x0, int y0, int w, int h, int flags_or, int qp)
{
  for (int y = y0 >> 2; y < ((y0 + h + 3) >> 2) && y < s->h4; y++)
    for (int x = x0 >> 2; x < ((x0 + w + 3) >> 2) && x < s->w4; x++) {
      s->blk_flags[x + y*s->w4] |= (uint8_t)flags_or;
      if (qp > -100) s->blk_qp[x + y*s->w4] = (int8_t)qp;
    }
}
static void set_edge(synth_picture* s, int x0, int y0, int bits)
{
  int xd = x0 >> 2, yd = y0 >> 2;
  if (xd < s->w4 && yd < s->h4) s->blk_flags[xd + yd*s->w4] |= (uint8_t)bits;
}

/* ---------- coefficients ---------- */
static void gen_coeffs(synth_picture* s, de265hip_tu* tu)
{
  rng_t* r = &s->rng;
  int nT = 1 << tu->log2_size, area = nT*nT;
  int nnz = 1;
  while (nnz < area && rnd_u(r) > 0.15) nnz++;               /* Geometric(0.15) */
  if (rnd_pct(r, 2)) nnz = area;                             /* occasionally dense */
  uint8_t used[32*32];
  memset(used, 0, (size_t)area);
  tu->coeff_offset = (uint32_t)s->cval.n;
  int n = 0;
  for (int i = 0; i < nnz; i++) {
    /* low-frequency bias: exponential rank along the anti-diagonals */
    int pos;
    if (nnz == area) pos = i;
    else {
      double e = -log(1.0 - rnd_u(r) + 1e-12) * (nT * 0.35);
      int dsum = (int)e; if (dsum > 2*nT - 2) dsum = 2*nT - 2;
      int lo = dsum - (nT - 1) > 0 ? dsum - (nT - 1) : 0;
      int hi = dsum < nT - 1 ? dsum : nT - 1;
      int x = rnd_int(r, lo, hi), y = dsum - x;
      pos = x + y*nT;
      if (used[pos]) continue;
    }
    used[pos] = 1;
    int v;
    if (tu->flags & DE265HIP_TU_BYPASS) v = rnd_laplace(r, 3.0);
    else if (rnd_pct(r, s->cfg.big_coeff_pct)) v = rnd_pct(r, 50) ? 32767 - rnd_int(r, 0, 40) : -32768 + rnd_int(r, 0, 40);
    else v = rnd_laplace(r, 8.0);
    if (v == 0) v = rnd_pct(r, 50) ? 1 : -1;
    v = clip3i(-32768, 32767, v);
    VPUSH(s->cval, (int16_t)v);
    VPUSH(s->cpos, (uint16_t)pos);
    n++;
  }
  tu->n_coeff = (uint16_t)n;
}

/* ---------- TU emission ---------- */
static void emit_tu(synth_picture* s, int x0, int y0, int log2, int cIdx, int cbf, int mode, int res_scale)
{
  const synth_config* g = &s->cfg;
  if (!s->cur_intra && !cbf && !res_scale) return;       /* nothing to do for inter TU without residual */
  de265hip_tu tu; memset(&tu, 0, sizeof(tu));
  tu.x0 = (uint16_t)x0; tu.y0 = (uint16_t)y0; tu.log2_size = (uint8_t)log2; tu.c_idx = (uint8_t)cIdx;
  tu.flags = (uint8_t)((s->cur_intra ? DE265HIP_TU_INTRA : 0) | (cbf ? DE265HIP_TU_CBF : 0) |
                       (s->cur_bypass ? DE265HIP_TU_BYPASS : 0));
  tu.intra_mode = (uint8_t)mode;
  tu.res_scale_val = (int8_t)res_scale;
  int qpbd = 6 * (g->bit_depth - 8);
  if (cIdx == 0) tu.qp = (int8_t)(s->cur_qp + qpbd);
  else {
    int off = cIdx == 1 ? s->desc.params.pic_cb_qp_offset : s->desc.params.pic_cr_qp_offset;
    int qpi = clip3i(-qpbd, 57, s->cur_qp + off);       /* transform.cc:149-172: the table only in 4:2:0 */
    tu.qp = (int8_t)((s->cf == 1 ? table8_22(qpi) : qpi) + qpbd);
  }
  if (cbf) {
    int max_ts = g->log2_max_tskip_size > 2 ? g->log2_max_tskip_size : 2;
    if (log2 <= max_ts && !s->cur_bypass && rnd_pct(&s->rng, g->tskip_pct)) tu.flags |= DE265HIP_TU_TSKIP;
    /* explicit RDPCM: inter CU, transform skip or bypass (slice.cc:3313-3330) */
    if (g->explicit_rdpcm_pct > 0 && !s->cur_intra && (s->cur_bypass || (tu.flags & DE265HIP_TU_TSKIP)) && rnd_pct(&s->rng, g->explicit_rdpcm_pct))
      tu.flags |= (uint8_t)(DE265HIP_TU_EXPLICIT_RDPCM | (rnd_pct(&s->rng, 50) ? DE265HIP_TU_EXPLICIT_RDPCM_VERT : 0));
    gen_coeffs(s, &tu);
    if (tu.n_coeff == 0) tu.flags &= (uint8_t)~(DE265HIP_TU_CBF | DE265HIP_TU_TSKIP | DE265HIP_TU_EXPLICIT_RDPCM | DE265HIP_TU_EXPLICIT_RDPCM_VERT);
  }
  VPUSH(s->order, ORD_TU | (uint32_t)s->tus.n);
  VPUSH(s->tus, tu);
  if (cIdx == 0 && (tu.flags & DE265HIP_TU_CBF))
    fill_blk(s, x0, y0, 1 << log2, 1 << log2, DE265HIP_BLK_NONZERO, -1000);
}

/* transform tree (slice.cc:3821 read_transform_tree + :3549 read_transform_unit).
 * blkIdx/parent carry the 4:2:0 rule that a split 8x8 emits its chroma once,
 * with the 4th luma block, at the parent position (slice.cc:3752-3800). */
static void gen_ttree(synth_picture* s, int x0, int y0, int xBase, int yBase, int log2, int depth,
                      int blkIdx, int force_split_depth0, int left, int top, int cuX, int cuY, int log2Cb)
{
  const synth_config* g = &s->cfg;
  int split;
  if (log2 > g->log2_max_tb_size) split = 1;
  else if (log2 <= g->log2_min_tb_size || log2 <= 2) split = 0;
  else if (depth == 0 && force_split_depth0) split = 1;
  else if (depth >= 3) split = 0;
  else split = rnd_pct(&s->rng, 20 + g->split_bias / 2);
  if (split) {
    s->tu_split[(x0 >> g->log2_min_tb_size) + (y0 >> g->log2_min_tb_size)*s->tbs_w] |= (uint8_t)(1 << depth);
    int h = 1 << (log2 - 1);
    gen_ttree(s, x0,   y0,   x0, y0, log2-1, depth+1, 0, 0, left, top, cuX, cuY, log2Cb);
    gen_ttree(s, x0+h, y0,   x0, y0, log2-1, depth+1, 1, 0, DE265HIP_BLK_EDGE_TU_V, top, cuX, cuY, log2Cb);
    gen_ttree(s, x0,   y0+h, x0, y0, log2-1, depth+1, 2, 0, left, DE265HIP_BLK_EDGE_TU_H, cuX, cuY, log2Cb);
    gen_ttree(s, x0+h, y0+h, x0, y0, log2-1, depth+1, 3, 0, DE265HIP_BLK_EDGE_TU_V, DE265HIP_BLK_EDGE_TU_H, cuX, cuY, log2Cb);
    return;
  }
  /* leaf: edge marks (markTransformBlockBoundary, deblock.cc:31-61) */
  if (!s->cur_deblk_off) {
    for (int k = 0; k < (1 << log2); k += 4) set_edge(s, x0, y0 + k, left);
    for (int k = 0; k < (1 << log2); k += 4) set_edge(s, x0 + k, y0, top);
  }
  /* luma / chroma intra mode of the PU covering this TU */
  int lmode = 0, cmode = 0, pidx = 0;
  if (s->cur_intra) {
    int h2 = 1 << (log2Cb - 1);
    pidx = ((x0 - cuX) >= h2 ? 1 : 0) + ((y0 - cuY) >= h2 ? 2 : 0);
    lmode = s->cu_luma_mode[pidx];
    cmode = s->cu_chroma_mode[s->cf == 3 ? pidx : 0];
  }
  int cbfY = rnd_pct(&s->rng, g->cbf_pct);
  /* chroma cbf: bit 0 the (upper) chroma TU, bit 1 the lower one of 4:2:2 (slice.cc:3699-3750) */
  int cbfCb = rnd_pct(&s->rng, g->cbf_pct * 2 / 3), cbfCr = rnd_pct(&s->rng, g->cbf_pct * 2 / 3);
  int cbfCb2 = 0, cbfCr2 = 0;                            /* (drawn for 4:2:2 only: the 4:2:0 pictures of a seed stay what they were) */
  if (s->cf == 2) { cbfCb2 = rnd_pct(&s->rng, g->cbf_pct * 2 / 3); cbfCr2 = rnd_pct(&s->rng, g->cbf_pct * 2 / 3); }
  int n_before = s->tus.n;
  emit_tu(s, x0, y0, log2, 0, cbfY, lmode, 0);
  int luma_cbf = s->tus.n > n_before && (s->tus.p[s->tus.n - 1].flags & DE265HIP_TU_CBF) && s->tus.p[s->tus.n - 1].c_idx == 0;
  if (s->cf == 0) {
    /* monochrome: no chroma TUs */
  } else if (s->cf == 3) {
    /* 4:4:4: chroma TUs of the luma TU's size, cross-component prediction when the luma TU has coefficients and the CU is
     * inter or its chroma mode is derived (intra_chroma_pred_mode 4), slice.cc:3672-3684 */
    int elig = g->cross_component_pct > 0 && luma_cbf && (!s->cur_intra || s->cu_chroma_derived[pidx]);
    static const int rsv[8] = { 1, 2, 4, 8, -1, -2, -4, -8 };
    int rCb = (elig && rnd_pct(&s->rng, g->cross_component_pct)) ? rsv[rnd_int(&s->rng, 0, 7)] : 0;
    int rCr = (elig && rnd_pct(&s->rng, g->cross_component_pct)) ? rsv[rnd_int(&s->rng, 0, 7)] : 0;
    emit_tu(s, x0, y0, log2, 1, cbfCb, cmode, rCb);
    emit_tu(s, x0, y0, log2, 2, cbfCr, cmode, rCr);
  } else if (log2 > 2) {
    int xc = x0 / s->sw, yc = y0 / s->sh, nC = 1 << (log2 - 1);
    emit_tu(s, xc, yc, log2-1, 1, cbfCb, cmode, 0);
    if (s->cf == 2) emit_tu(s, xc, yc + nC, log2-1, 1, cbfCb2, cmode, 0);
    emit_tu(s, xc, yc, log2-1, 2, cbfCr, cmode, 0);
    if (s->cf == 2) emit_tu(s, xc, yc + nC, log2-1, 2, cbfCr2, cmode, 0);
  } else if (blkIdx == 3) {
    int xc = xBase / s->sw, yc = yBase / s->sh;
    emit_tu(s, xc, yc, 2, 1, cbfCb, cmode, 0);
    if (s->cf == 2) emit_tu(s, xc, yc + 4, 2, 1, cbfCb2, cmode, 0);
    emit_tu(s, xc, yc, 2, 2, cbfCr, cmode, 0);
    if (s->cf == 2) emit_tu(s, xc, yc + 4, 2, 2, cbfCr2, cmode, 0);
  }
}

/* ---------- PU emission ---------- */
static void emit_pu(synth_picture* s, int x, int y, int w, int h)
{
  const synth_config* g = &s->cfg;
  rng_t* r = &s->rng;
  const de265hip_slice_params* sh = &s->slices[s->cur_slice];
  de265hip_pu pu; memset(&pu, 0, sizeof(pu));
  pu.x = (uint16_t)x; pu.y = (uint16_t)y; pu.w = (uint8_t)w; pu.h = (uint8_t)h;
  pu.slice_idx = (uint16_t)s->cur_slice;
  int bi = (sh->slice_type == 0) && rnd_pct(r, g->bi_pct) && !((w == 8 && h == 4) || (w == 4 && h == 8));
  int lists = bi ? 3 : ((sh->slice_type == 0 && rnd_pct(r, 50)) ? 2 : 1);
  pu.pred_flag = (uint8_t)lists;
  de265hip_motion m; memset(&m, 0, sizeof(m)); m.ref_slot[0] = m.ref_slot[1] = -1;
  for (int l = 0; l < 2; l++) {
    if (!(lists & (1 << l))) { pu.ref_idx[l] = -1; continue; }
    pu.ref_idx[l] = (int8_t)rnd_int(r, 0, g->n_ref_slots - 1);
    int mvx = (int)lrint(rnd_gauss(r) * g->mv_sigma_qpel), mvy = (int)lrint(rnd_gauss(r) * g->mv_sigma_qpel);
    if (rnd_pct(r, 3)) { mvx *= 8; mvy *= 8; }                 /* some long vectors */
    if (rnd_pct(r, 10)) { mvx &= ~3; }                          /* some full-pel components */
    if (rnd_pct(r, 10)) { mvy &= ~3; }
    /* keep the referenced block within 80 samples of the picture */
    mvx = clip3i(-(x + 80) * 4, (g->width - x + 80 - w) * 4, mvx);
    mvy = clip3i(-(y + 80) * 4, (g->height - y + 80 - h) * 4, mvy);
    pu.mv[l][0] = (int16_t)mvx; pu.mv[l][1] = (int16_t)mvy;
    m.mv[l][0] = (int16_t)mvx; m.mv[l][1] = (int16_t)mvy;
    m.ref_slot[l] = sh->ref_pic_list[l][pu.ref_idx[l]];
  }
  if (bi && rnd_pct(r, 5)) {                                    /* identical bi -> uni shortcut case */
    pu.mv[1][0] = pu.mv[0][0]; pu.mv[1][1] = pu.mv[0][1];
    m.mv[1][0] = m.mv[0][0]; m.mv[1][1] = m.mv[0][1];
  }
  VPUSH(s->order, ORD_PU | (uint32_t)s->pus.n);
  VPUSH(s->pus, pu);
  for (int yy = y >> 2; yy < (y + h) >> 2; yy++)
    for (int xx = x >> 2; xx < (x + w) >> 2; xx++) s->blk_motion[xx + yy*s->w4] = m;
}

/* ---------- coding unit ---------- */
static void gen_cu(synth_picture* s, int x0, int y0, int log2Cb)
{
  const synth_config* g = &s->cfg;
  rng_t* r = &s->rng;
  int cb = 1 << log2Cb;
  const de265hip_pic_params* P = &s->desc.params;
  int ctbx = x0 >> P->log2_ctb_size, ctby = y0 >> P->log2_ctb_size;
  const de265hip_ctb_info* ci = &s->ctbs[ctbx + ctby*s->ctbs_w];
  s->cur_slice = ci->slice_idx;
  const de265hip_slice_params* sh = &s->slices[s->cur_slice];
  s->cur_deblk_off = sh->slice_deblocking_filter_disabled_flag;
  s->cur_qp = rnd_int(r, g->qp_min, g->qp_max);
  s->cur_bypass = rnd_pct(r, g->bypass_pct);
  s->cur_intra = (sh->slice_type == 2) || rnd_pct(r, g->intra_pct);
  int pcm = s->cur_intra && log2Cb >= 3 && log2Cb <= 5 && rnd_pct(r, g->pcm_pct);

  int cbx = x0 >> 3, cby = y0 >> 3;
  s->cb_log2[cbx + cby*s->cbs_w] = (uint8_t)log2Cb;
  fill_blk(s, x0, y0, cb, cb,
           (s->cur_intra ? DE265HIP_BLK_INTRA : 0) | (s->cur_bypass ? DE265HIP_BLK_BYPASS : 0) |
           (pcm ? DE265HIP_BLK_PCM : 0), s->cur_qp);

  /* CB edge flags (derive_edgeFlags_CTBRow, deblock.cc:166-210) */
  int left = DE265HIP_BLK_EDGE_TU_V, top = DE265HIP_BLK_EDGE_TU_H;
  int mask = (1 << P->log2_ctb_size) - 1;
  if (x0 == 0) left = 0;
  if (y0 == 0) top = 0;
  if (x0 && (x0 & mask) == 0) {
    const de265hip_ctb_info* li = &s->ctbs[(ctbx - 1) + ctby*s->ctbs_w];
    if (!sh->slice_loop_filter_across_slices_enabled_flag && li->slice_addr_rs != ci->slice_addr_rs) left = 0;
    else if (!P->loop_filter_across_tiles_enabled_flag &&
             s->tile_id[ctbx + ctby*s->ctbs_w] != s->tile_id[ctbx - 1 + ctby*s->ctbs_w]) left = 0;
  }
  if (y0 && (y0 & mask) == 0) {
    const de265hip_ctb_info* ti = &s->ctbs[ctbx + (ctby - 1)*s->ctbs_w];
    if (!sh->slice_loop_filter_across_slices_enabled_flag && ti->slice_addr_rs != ci->slice_addr_rs) top = 0;
    else if (!P->loop_filter_across_tiles_enabled_flag &&
             s->tile_id[ctbx + ctby*s->ctbs_w] != s->tile_id[ctbx + (ctby - 1)*s->ctbs_w]) top = 0;
  }

  if (pcm) {
    de265hip_pcm pc; memset(&pc, 0, sizeof(pc));
    pc.x0 = (uint16_t)x0; pc.y0 = (uint16_t)y0; pc.log2_cb_size = (uint8_t)log2Cb;
    pc.sample_offset = (uint32_t)s->pcm_samples.n;
    int pcm_bits = g->bit_depth - rnd_int(r, 0, 2);
    int n = cb*cb + (s->cf ? 2*(cb/s->sw)*(cb/s->sh) : 0);
    for (int i = 0; i < n; i++)
      VPUSH(s->pcm_samples, (uint16_t)(rnd_int(r, 0, (1 << pcm_bits) - 1) << (g->bit_depth - pcm_bits)));
    VPUSH(s->order, ORD_PCM | (uint32_t)s->pcms.n);
    VPUSH(s->pcms, pc);
    /* PCM CU: one TU-less block; its boundary is still a transform edge */
    if (!s->cur_deblk_off) {
      for (int k = 0; k < cb; k += 4) set_edge(s, x0, y0 + k, left);
      for (int k = 0; k < cb; k += 4) set_edge(s, x0 + k, y0, top);
    }
    return;
  }

  if (s->cur_intra) {
    int nxn = (log2Cb == 3) && (g->log2_min_tb_size == 2) && rnd_pct(r, 40);
    s->cb_part[cbx + cby*s->cbs_w] = nxn ? 3 : 0;
    for (int i = 0; i < 4; i++) s->cu_luma_mode[i] = rnd_int(r, 0, 34);
    if (!nxn) s->cu_luma_mode[1] = s->cu_luma_mode[2] = s->cu_luma_mode[3] = s->cu_luma_mode[0];
    /* favour the modes with special filters */
    if (rnd_pct(r, 25)) { static const int fav[5] = { 0, 1, 10, 26, 34 }; s->cu_luma_mode[0] = fav[rnd_int(r, 0, 4)];
      if (!nxn) s->cu_luma_mode[1] = s->cu_luma_mode[2] = s->cu_luma_mode[3] = s->cu_luma_mode[0]; }
    /* one chroma mode per CU, except NxN in 4:4:4: one per PU (slice.cc:4440-4470) */
    for (int i = 0; i < ((nxn && s->cf == 3) ? 4 : 1); i++) {
      s->cu_chroma_derived[i] = rnd_pct(r, 40);
      s->cu_chroma_mode[i] = s->cu_chroma_derived[i] ? s->cu_luma_mode[i] : rnd_int(r, 0, 34);
    }
    if (!(nxn && s->cf == 3))
      for (int i = 1; i < 4; i++) { s->cu_chroma_mode[i] = s->cu_chroma_mode[0]; s->cu_chroma_derived[i] = s->cu_chroma_derived[0]; }
    if (nxn && !s->cur_deblk_off)
      for (int k = 0; k < cb; k++) {
        set_edge(s, x0 + cb/2, y0 + k, DE265HIP_BLK_EDGE_PB_V);
        set_edge(s, x0 + k, y0 + cb/2, DE265HIP_BLK_EDGE_PB_H);
      }
    gen_ttree(s, x0, y0, x0, y0, log2Cb, 0, 0, nxn, left, top, x0, y0, log2Cb);
    return;
  }

  /* inter CU */
  int pm = 0;
  if (rnd_pct(r, 45)) {
    if (log2Cb >= 4 && g->amp && rnd_pct(r, 30)) pm = rnd_int(r, 4, 7);
    else pm = rnd_int(r, 1, 2);
  }
  s->cb_part[cbx + cby*s->cbs_w] = (uint8_t)pm;
  int h2 = cb/2, q4 = cb/4;
  switch (pm) {
    case 0: emit_pu(s, x0, y0, cb, cb); break;
    case 1: emit_pu(s, x0, y0, cb, h2); emit_pu(s, x0, y0+h2, cb, h2); break;
    case 2: emit_pu(s, x0, y0, h2, cb); emit_pu(s, x0+h2, y0, h2, cb); break;
    case 4: emit_pu(s, x0, y0, cb, q4); emit_pu(s, x0, y0+q4, cb, cb-q4); break;
    case 5: emit_pu(s, x0, y0, cb, cb-q4); emit_pu(s, x0, y0+cb-q4, cb, q4); break;
    case 6: emit_pu(s, x0, y0, q4, cb); emit_pu(s, x0+q4, y0, cb-q4, cb); break;
    default: emit_pu(s, x0, y0, cb-q4, cb); emit_pu(s, x0+cb-q4, y0, q4, cb); break;
  }
  if (!s->cur_deblk_off)
    for (int k = 0; k < cb; k++)
      switch (pm) {                                       /* markPredictionBlockBoundary :66-127 */
        case 1: set_edge(s, x0 + k, y0 + h2, DE265HIP_BLK_EDGE_PB_H); break;
        case 2: set_edge(s, x0 + h2, y0 + k, DE265HIP_BLK_EDGE_PB_V); break;
        case 4: set_edge(s, x0 + k, y0 + q4, DE265HIP_BLK_EDGE_PB_H); break;
        case 5: set_edge(s, x0 + k, y0 + h2 + q4, DE265HIP_BLK_EDGE_PB_H); break;
        case 6: set_edge(s, x0 + q4, y0 + k, DE265HIP_BLK_EDGE_PB_V); break;
        case 7: set_edge(s, x0 + h2 + q4, y0 + k, DE265HIP_BLK_EDGE_PB_V); break;
        default: break;
      }
  int root_cbf = rnd_pct(r, 70);
  if (root_cbf) {
    memset(s->cu_chroma_mode, 0, sizeof(s->cu_chroma_mode)); memset(s->cu_luma_mode, 0, sizeof(s->cu_luma_mode));
    gen_ttree(s, x0, y0, x0, y0, log2Cb, 0, 0, 0, left, top, x0, y0, log2Cb);
  } else if (!s->cur_deblk_off) {
    /* rqt_root_cbf==0: no transform tree is read (slice.cc:4551-4574), all
     * split flags stay 0 and the CB outline is the only transform edge */
    for (int k = 0; k < cb; k += 4) { set_edge(s, x0, y0 + k, left); set_edge(s, x0 + k, y0, top); }
  }
}

static void gen_cqt(synth_picture* s, int x0, int y0, int log2)
{
  const synth_config* g = &s->cfg;
  int sz = 1 << log2;
  if (x0 >= g->width || y0 >= g->height) return;
  int split;
  if (x0 + sz > g->width || y0 + sz > g->height) split = 1;   /* implicit split at picture edge */
  else if (log2 <= 3) split = 0;
  else split = rnd_pct(&s->rng, log2 == 6 ? 55 + g->split_bias/3 : (log2 == 5 ? 25 + g->split_bias/2 : g->split_bias/2 + 10));
  if (split) {
    int h = sz/2;
    gen_cqt(s, x0, y0, log2-1); gen_cqt(s, x0+h, y0, log2-1);
    gen_cqt(s, x0, y0+h, log2-1); gen_cqt(s, x0+h, y0+h, log2-1);
  } else gen_cu(s, x0, y0, log2);
}

synth_picture* synth_generate(const synth_config* cfg)
{
  synth_picture* s = calloc(1, sizeof(*s));
  s->cfg = *cfg;
  const synth_config* g = &s->cfg;
  { rng_t h; h.s = cfg->seed ^ 0x5DEECE66Dull; (void)rnd64(&h); s->rng.s = rnd64(&h); }   /* hashed: nearby seeds give unrelated streams */
  rng_t* r = &s->rng;
  de265hip_pic_params* P = &s->desc.params;
  P->width = g->width; P->height = g->height;
  P->bit_depth_luma = P->bit_depth_chroma = g->bit_depth;
  s->cf = g->monochrome ? 0 : (g->chroma_format ? g->chroma_format : 1);
  /* 4:4:4 with scaling lists: no 32x32 TUs.  The reference indexes ScalingFactor_Size3[matrixID] with matrixID up to 5 for a
   * 32x32 chroma TU although the array holds two matrices (transform.cc:487-493, sps.h:57): undefined, nothing to be pinned to. */
  if (s->cf == 3 && g->scaling_list && s->cfg.log2_max_tb_size > 4) s->cfg.log2_max_tb_size = 4;
  s->sw = (s->cf == 3 || s->cf == 0) ? 1 : 2; s->sh = s->cf == 1 ? 2 : 1;
  P->chroma_format_idc = s->cf;
  P->implicit_rdpcm_enabled_flag = g->implicit_rdpcm;
  P->transform_skip_rotation_enabled_flag = g->rotation;
  P->intra_smoothing_disabled_flag = g->intra_smoothing_disabled;
  P->cross_component_prediction_enabled_flag = (s->cf == 3 && g->cross_component_pct > 0);
  P->high_precision_offsets_enabled_flag = g->high_precision_offsets;
  P->log2_ctb_size = g->log2_ctb_size; P->log2_min_cb_size = 3; P->log2_min_tb_size = g->log2_min_tb_size;
  P->pcm_loop_filter_disable_flag = g->pcm_loop_filter_disable;
  P->strong_intra_smoothing_enable_flag = g->strong_intra_smoothing;
  P->constrained_intra_pred_flag = g->constrained_intra_pred;
  P->sample_adaptive_offset_enabled_flag = g->sao;
  P->scaling_list_enable_flag = g->scaling_list;
  P->weighted_pred_flag = g->weighted_pred; P->weighted_bipred_flag = g->weighted_pred;
  P->pic_cb_qp_offset = rnd_int(r, -4, 4); P->pic_cr_qp_offset = rnd_int(r, -4, 4);
  P->loop_filter_across_tiles_enabled_flag = g->lf_across_tiles;
  P->num_tile_columns = g->tile_cols; P->num_tile_rows = g->tile_rows;
  int ctb = 1 << g->log2_ctb_size;
  s->ctbs_w = (g->width + ctb - 1) / ctb; s->ctbs_h = (g->height + ctb - 1) / ctb;
  s->w4 = (g->width + 3)/4; s->h4 = (g->height + 3)/4;
  s->cbs_w = (g->width + 7)/8; s->cbs_h = (g->height + 7)/8;
  s->tbs_w = s->ctbs_w << (g->log2_ctb_size - g->log2_min_tb_size);
  s->tbs_h = s->ctbs_h << (g->log2_ctb_size - g->log2_min_tb_size);
  for (int i = 0; i <= g->tile_cols; i++) P->col_bd[i] = (uint16_t)(i * s->ctbs_w / g->tile_cols);   /* uniform spacing */
  for (int j = 0; j <= g->tile_rows; j++) P->row_bd[j] = (uint16_t)(j * s->ctbs_h / g->tile_rows);
  int nctb = s->ctbs_w * s->ctbs_h;
  s->tile_id = malloc(sizeof(int) * nctb);
  s->ts2rs = malloc(sizeof(int) * nctb);
  { int t = 0, ts = 0;
    for (int j = 0; j < g->tile_rows; j++)
      for (int i = 0; i < g->tile_cols; i++, t++)
        for (int y = P->row_bd[j]; y < P->row_bd[j+1]; y++)
          for (int x = P->col_bd[i]; x < P->col_bd[i+1]; x++) { s->tile_id[x + y*s->ctbs_w] = t; s->ts2rs[ts++] = x + y*s->ctbs_w; } }

  /* slices */
  int ntiles = g->tile_cols * g->tile_rows;
  s->n_slices = (ntiles > 1) ? (g->slice_per_tile ? ntiles : 1) : (g->n_slices < 1 ? 1 : g->n_slices);
  if (s->n_slices > nctb) s->n_slices = nctb;
  s->slices = calloc((size_t)s->n_slices, sizeof(de265hip_slice_params));
  s->ctbs = calloc((size_t)nctb, sizeof(de265hip_ctb_info));
  for (int k = 0; k < s->n_slices; k++) {
    de265hip_slice_params* sh = &s->slices[k];
    sh->slice_type = g->slice_type;
    sh->slice_deblocking_filter_disabled_flag = !g->deblocking || (s->n_slices > 2 && k == 1);
    sh->slice_beta_offset = 2 * rnd_int(r, -3, 3);
    sh->slice_tc_offset = 2 * rnd_int(r, -3, 3);
    sh->slice_loop_filter_across_slices_enabled_flag = rnd_pct(r, g->lf_across_slices_pct);
    sh->slice_sao_luma_flag = g->sao ? rnd_pct(r, 90) : 0;
    sh->slice_sao_chroma_flag = g->sao ? rnd_pct(r, 90) : 0;
    sh->luma_log2_weight_denom = rnd_int(r, 0, 7);
    sh->chroma_log2_weight_denom = clip3i(0, 7, sh->luma_log2_weight_denom + rnd_int(r, -2, 2));
    for (int l = 0; l < 2; l++)
      for (int i = 0; i < 16; i++) {
        int on = rnd_pct(r, 70);
        sh->luma_weight[l][i] = (int16_t)(on ? rnd_int(r, -32, 95) : (1 << sh->luma_log2_weight_denom));
        sh->luma_offset[l][i] = (int16_t)(on ? rnd_int(r, -64, 63) : 0);
        for (int c = 0; c < 2; c++) {
          sh->chroma_weight[l][i][c] = (int16_t)(on ? rnd_int(r, -32, 95) : (1 << sh->chroma_log2_weight_denom));
          sh->chroma_offset[l][i][c] = (int16_t)(on ? rnd_int(r, -64, 63) : 0);
        }
        int nr = g->n_ref_slots > 0 ? g->n_ref_slots : 1;
        sh->ref_pic_list[l][i] = g->ref_slots[(i + l * (k + 1)) % nr];
      }
  }
  /* CTB -> slice assignment in decode (tile-scan) order */
  for (int ts = 0; ts < nctb; ts++) {
    int rs = s->ts2rs[ts];
    int k;
    if (ntiles > 1) k = g->slice_per_tile ? s->tile_id[rs] : 0;
    else k = (int)((int64_t)ts * s->n_slices / nctb);
    s->ctbs[rs].slice_idx = (uint16_t)k;
  }
  { int* first = malloc(sizeof(int) * (size_t)s->n_slices);
    for (int k = 0; k < s->n_slices; k++) first[k] = -1;
    for (int ts = 0; ts < nctb; ts++) { int rs = s->ts2rs[ts]; int k = s->ctbs[rs].slice_idx; if (first[k] < 0) first[k] = rs; }
    for (int k = 0; k < s->n_slices; k++) s->slices[k].slice_addr_rs = first[k];
    for (int rs = 0; rs < nctb; rs++) s->ctbs[rs].slice_addr_rs = (uint16_t)first[s->ctbs[rs].slice_idx];
    free(first); }
  /* SAO per CTB: off 40 %, band 20 %, edge 40 % per component group */
  int maxoff = (1 << ((g->bit_depth < 10 ? g->bit_depth : 10) - 5)) - 1;     /* 7 for 8 bit, 31 for 10 bit */
  for (int rs = 0; rs < nctb; rs++) {
    de265hip_ctb_info* ci = &s->ctbs[rs];
    const de265hip_slice_params* sh = &s->slices[ci->slice_idx];
    int tl = 0, tcv = 0, u;
    if (g->sao && sh->slice_sao_luma_flag) { u = rnd_int(r, 0, 99); tl = u < 40 ? 0 : (u < 60 ? 1 : 2); }
    if (g->sao && sh->slice_sao_chroma_flag) { u = rnd_int(r, 0, 99); tcv = u < 40 ? 0 : (u < 60 ? 1 : 2); }
    ci->sao_type_idx = (uint8_t)(tl | (tcv << 2) | (tcv << 4));
    int el = rnd_int(r, 0, 3), ec = rnd_int(r, 0, 3);
    ci->sao_eo_class = (uint8_t)(el | (ec << 2) | (ec << 4));
    for (int c = 0; c < 3; c++) {
      ci->sao_band_position[c] = (uint8_t)rnd_int(r, 0, 31);
      int type = (ci->sao_type_idx >> (2*c)) & 3;
      for (int i = 0; i < 4; i++) {
        int v = rnd_int(r, 0, maxoff);
        if (type == 2) ci->sao_offset_val[c][i] = (int8_t)(i < 2 ? v : -v);    /* edge: sign by category (slice.cc:2795-2815) */
        else ci->sao_offset_val[c][i] = (int8_t)(rnd_pct(r, 50) ? v : -v);
      }
    }
  }

  size_t nblk = (size_t)s->w4 * s->h4;
  s->blk_flags = calloc(nblk, 1); s->blk_qp = calloc(nblk, 1);
  s->blk_motion = calloc(nblk, sizeof(de265hip_motion));
  for (size_t i = 0; i < nblk; i++) s->blk_motion[i].ref_slot[0] = s->blk_motion[i].ref_slot[1] = -1;
  s->cb_log2 = calloc((size_t)s->cbs_w * s->cbs_h, 1); s->cb_part = calloc((size_t)s->cbs_w * s->cbs_h, 1);
  s->tu_split = calloc((size_t)s->tbs_w * s->tbs_h, 1);
  if (g->scaling_list) {
    s->scaling = malloc(DE265HIP_SCALING_BLOB_BYTES);
    for (int i = 0; i < DE265HIP_SCALING_BLOB_BYTES; i++) s->scaling[i] = (uint8_t)clip3i(1, 255, 16 + rnd_laplace(r, 10.0));
  }

  for (int ts = 0; ts < nctb; ts++) {
    int rs = s->ts2rs[ts];
    gen_cqt(s, (rs % s->ctbs_w) * ctb, (rs / s->ctbs_w) * ctb, g->log2_ctb_size);
  }

  s->blk_flags_noedge = malloc(nblk);
  for (size_t i = 0; i < nblk; i++) s->blk_flags_noedge[i] = s->blk_flags[i] & 0x0F;

  de265hip_picture_desc* d = &s->desc;
  d->scaling_factors = s->scaling;
  d->n_slices = s->n_slices; d->slices = s->slices;
  d->n_ctbs = nctb; d->ctbs = s->ctbs;
  d->n_tus = s->tus.n; d->tus = s->tus.p;
  d->n_coeffs = s->cval.n; d->coeff_val = s->cval.p; d->coeff_pos = s->cpos.p;
  d->n_pus = s->pus.n; d->pus = s->pus.p;
  d->n_pcms = s->pcms.n; d->pcms = s->pcms.p;
  d->n_pcm_samples = s->pcm_samples.n; d->pcm_samples = s->pcm_samples.p;
  d->blk_flags = s->blk_flags; d->blk_qp_y = s->blk_qp; d->blk_motion = s->blk_motion;
  return s;
}

void synth_free(synth_picture* s)
{
  if (!s) return;
  free(s->tile_id); free(s->ts2rs); free(s->slices); free(s->ctbs);
  free(s->tus.p); free(s->cval.p); free(s->cpos.p); free(s->pus.p); free(s->pcms.p);
  free(s->pcm_samples.p); free(s->order.p);
  free(s->blk_flags); free(s->blk_flags_noedge); free(s->blk_qp); free(s->blk_motion);
  free(s->cb_log2); free(s->cb_part); free(s->tu_split); free(s->scaling);
  free(s);
}
const de265hip_picture_desc* synth_desc(const synth_picture* s) { return &s->desc; }
const uint32_t* synth_order(const synth_picture* s, int32_t* n) { *n = s->order.n; return s->order.p; }
const uint8_t* synth_cb_log2_size(const synth_picture* s) { return s->cb_log2; }
const uint8_t* synth_cb_part_mode(const synth_picture* s) { return s->cb_part; }
const uint8_t* synth_tu_split(const synth_picture* s) { return s->tu_split; }
const uint8_t* synth_blk_flags_noedge(const synth_picture* s) { return s->blk_flags_noedge; }

void synth_fill_plane(void* plane, int stride, int w, int h, int bd, uint64_t seed)
{
  rng_t r; { rng_t h; h.s = seed ^ 0xA5A5A5A5ull; (void)rnd64(&h); r.s = rnd64(&h); }
  int maxv = (1 << bd) - 1;
  double amp = maxv * 0.35, mid = maxv * 0.5;
  double fx = 1.0 / (17.0 + (double)(seed % 13)), fy = 1.0 / (29.0 + (double)(seed % 7));
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      double v = mid + amp * sin(x * fx) * cos(y * fy) + (x + y) * (maxv * 0.05 / (w + h));
      int noise = (int)(rnd64(&r) % (uint64_t)(maxv / 12 + 1)) - maxv / 24;
      int iv = clip3i(0, maxv, (int)lrint(v) + noise);
      if (bd > 8) ((uint16_t*)plane)[x + (size_t)y*stride] = (uint16_t)iv;
      else ((uint8_t*)plane)[x + (size_t)y*stride] = (uint8_t)iv;
    }
}
