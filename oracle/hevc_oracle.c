/*
 * hevc_oracle.c -- CPU restatement of libde265's pixel-reconstruction path.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker.  The product (libde265_amd/)
 * never includes, links or calls it.
 *
 * PARITY UNPINNED (see hevc_oracle.h): the reference is not buildable under
 * this round's rules and holds no golden vectors for this path.
 *
 * Every function cites the reference file:line (relative to
 * /root/reference/libde265/) whose behaviour it restates.
 */
#include "hevc_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---------------- small helpers (util.h:107-112) ---------------- */
static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int clip_bd(int v, int bd) { int m = (1 << bd) - 1; return v < 0 ? 0 : (v > m ? m : v); }
static inline int iabs(int a) { return a < 0 ? -a : a; }
static inline int isign(int a) { return a < 0 ? -1 : (a > 0 ? 1 : 0); }
static inline int ilog2(int v) { int n = 0; while (v > 1) { n++; v >>= 1; } return n; }

/* ---------------- constant tables ---------------- */
/* 64*cos-like magnitudes of the HEVC core transform, index m = angle in units
 * of pi/64.  mat_dct[k][n] (fallback-dct.cc:513-546) == dct_c(k,n). */
static const int8_t dct_mag[33] = {
  64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
  61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
static inline int dct_c(int k, int n)
{
  int m = (k * (2*n + 1)) & 127;
  if (m > 64) m = 128 - m;
  return m > 32 ? -dct_mag[64 - m] : dct_mag[m];
}
/* mat_8_357, fallback-dct.cc:261-266 */
static const int8_t dst_mat[4][4] = {
  { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };
/* levelScale, transform.cc:349 */
static const int level_scale[6] = { 40, 45, 51, 57, 64, 72 };
/* intraPredAngle_table / invAngle_table, intrapred.cc:892-898 */
static const int intra_pred_angle[35] = {
  0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
  -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const int inv_angle[15] = {
  -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };
/* extra_before / extra_after, fallback-motion.cc:478-479, motion.cc:43-44 */
static const int qpel_before[4] = { 0, 3, 3, 2 };
static const int qpel_after[4]  = { 0, 3, 4, 4 };
/* table_8_23_beta / table_8_23_tc, deblock.cc:389-399 */
static const uint8_t tab_beta[52] = {
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 6,7,8,9,10,11,12,13,14,15,16,17,18,
  20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64 };
static const uint8_t tab_tc[54] = {
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 1,1,1,1,1,1,1,1,1, 2,2,2,2, 3,3,3,3, 4,4,4,
  5,5, 6,6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
/* tab8_22 / table8_22(), transform.cc:28 + transform.h:29-34 */
static const uint8_t tab_qpc[14] = { 29,30,31,32,33,33,34,34,35,35,36,36,37,37 };
static inline int table8_22(int qPi)
{
  if (qPi < 30) return qPi;
  if (qPi >= 43) return qPi - 6;
  return tab_qpc[qPi - 30];
}

int oracle_dct_coeff(int row, int col) { return dct_c(row, col); }
int oracle_table(const char* name, int idx)
{
  if (!strcmp(name, "beta")) return tab_beta[idx];
  if (!strcmp(name, "tc")) return tab_tc[idx];
  if (!strcmp(name, "angle")) return intra_pred_angle[idx];
  if (!strcmp(name, "invangle")) return inv_angle[idx];
  if (!strcmp(name, "qpc")) return table8_22(idx);
  if (!strcmp(name, "lscale")) return level_scale[idx];
  if (!strcmp(name, "dst")) return dst_mat[idx / 4][idx % 4];
  return -99999;
}

/* ---------------- picture context ---------------- */
typedef struct octx {
  const de265hip_picture_desc* d;
  oracle_image* img;
  const oracle_image* dpb;
  int ctbs_w, ctbs_h;          /* PicWidthInCtbsY / PicHeightInCtbsY */
  int tbs_w, tbs_h;            /* PicWidthInTbsY  / PicHeightInTbsY  */
  int w4, h4;                  /* deblk_width/height = ceil(W/4), ceil(H/4) (image.cc:432-433) */
  int* min_tb_addr_zs;         /* pps.cc:671-690 */
  int* tile_id;                /* TileIdRS, pps.cc:646-660 */
  int sw, sh;                  /* SubWidthC, SubHeightC (sps.cc:540-552) */
  int cw, chh;                 /* chroma plane width / height */
  int ncomp;                   /* 1: monochrome */
  int32_t* residual_luma;      /* thread_context::residual_luma (decctx.h): the last luma TU's residual, read by cross_comp_pred */
} octx;

static inline int blk_flags_at(const octx* c, int xL, int yL)
{ return c->d->blk_flags[(xL >> 2) + (yL >> 2) * c->w4]; }
static inline int blk_is_intra(const octx* c, int xL, int yL)
{ return blk_flags_at(c, xL, yL) & DE265HIP_BLK_INTRA; }
static inline int qp_at(const octx* c, int xL, int yL)
{ return c->d->blk_qp_y[(xL >> 2) + (yL >> 2) * c->w4]; }
/* (pcm_loop_filter_disable_flag && pcm_flag) || cu_transquant_bypass:
 * deblock.cc:577-590, :837-860; sao.cc:112-117 */
static inline int lf_exempt(const octx* c, int xL, int yL)
{
  int f = blk_flags_at(c, xL, yL);
  return ((f & DE265HIP_BLK_PCM) && c->d->params.pcm_loop_filter_disable_flag) ||
         (f & DE265HIP_BLK_BYPASS);
}
static inline const de265hip_slice_params* slice_at(const octx* c, int xL, int yL)
{
  int l = c->d->params.log2_ctb_size;
  return &c->d->slices[c->d->ctbs[(xL >> l) + (yL >> l) * c->ctbs_w].slice_idx];
}

/* pic_parameter_set::set_derived_values, pps.cc:560-690: CtbAddrRStoTS,
 * TileIdRS and MinTbAddrZS from the tile grid. */
static int octx_init(octx* c, const de265hip_picture_desc* d, oracle_image* img, const oracle_image* dpb)
{
  const de265hip_pic_params* P = &d->params;
  memset(c, 0, sizeof(*c));
  /* monochrome: intra pictures only (the reference's inter path reads chroma planes a monochrome picture does not have, motion.cc:302-305) */
  if (P->chroma_format_idc < 0 || P->chroma_format_idc > 3 || P->extended_precision_processing_flag) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  if (P->chroma_format_idc == 0 && d->n_pus > 0) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  if (P->num_tile_columns < 1 || P->num_tile_rows < 1 ||
      P->num_tile_columns > 20 || P->num_tile_rows > 22) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  c->d = d; c->img = img; c->dpb = dpb;
  c->sw = (P->chroma_format_idc == 3 || P->chroma_format_idc == 0) ? 1 : 2; c->sh = P->chroma_format_idc == 1 ? 2 : 1;   /* SubWidthC, SubHeightC (sps.cc:540-552) */
  c->ncomp = P->chroma_format_idc == 0 ? 1 : 3;
  c->cw = c->ncomp == 1 ? 0 : P->width / c->sw; c->chh = c->ncomp == 1 ? 0 : P->height / c->sh;
  c->residual_luma = (int32_t*)calloc(32 * 32, sizeof(int32_t));
  int ctb = 1 << P->log2_ctb_size;
  c->ctbs_w = (P->width + ctb - 1) >> P->log2_ctb_size;
  c->ctbs_h = (P->height + ctb - 1) >> P->log2_ctb_size;
  int tb = 1 << P->log2_min_tb_size;
  c->tbs_w = c->ctbs_w << (P->log2_ctb_size - P->log2_min_tb_size);
  c->tbs_h = c->ctbs_h << (P->log2_ctb_size - P->log2_min_tb_size);
  (void)tb;
  c->w4 = (P->width + 3) / 4;
  c->h4 = (P->height + 3) / 4;
  if (d->n_ctbs != c->ctbs_w * c->ctbs_h) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  int nctb = c->ctbs_w * c->ctbs_h;
  int* rs2ts = (int*)malloc(sizeof(int) * nctb);
  c->tile_id = (int*)malloc(sizeof(int) * nctb);
  c->min_tb_addr_zs = (int*)malloc(sizeof(int) * c->tbs_w * c->tbs_h);
  if (!rs2ts || !c->tile_id || !c->min_tb_addr_zs) { free(rs2ts); return DE265HIP_ERROR_OUT_OF_MEMORY; }
  const uint16_t* colBd = P->col_bd; const uint16_t* rowBd = P->row_bd;
  for (int addr = 0; addr < nctb; addr++) {             /* pps.cc:592-627 */
    int tbX = addr % c->ctbs_w, tbY = addr / c->ctbs_w;
    int tileX = -1, tileY = -1;
    for (int i = 0; i < P->num_tile_columns; i++) if (tbX >= colBd[i]) tileX = i;
    for (int j = 0; j < P->num_tile_rows; j++) if (tbY >= rowBd[j]) tileY = j;
    int v = 0;
    for (int i = 0; i < tileX; i++) v += (rowBd[tileY+1] - rowBd[tileY]) * (colBd[i+1] - colBd[i]);
    for (int j = 0; j < tileY; j++) v += c->ctbs_w * (rowBd[j+1] - rowBd[j]);
    v += (tbY - rowBd[tileY]) * (colBd[tileX+1] - colBd[tileX]);
    v += tbX - colBd[tileX];
    rs2ts[addr] = v;
  }
  for (int j = 0, t = 0; j < P->num_tile_rows; j++)      /* pps.cc:646-660 */
    for (int i = 0; i < P->num_tile_columns; i++, t++)
      for (int y = rowBd[j]; y < rowBd[j+1]; y++)
        for (int x = colBd[i]; x < colBd[i+1]; x++) c->tile_id[y*c->ctbs_w + x] = t;
  int dl = P->log2_ctb_size - P->log2_min_tb_size;
  for (int y = 0; y < c->tbs_h; y++)                     /* pps.cc:671-690 */
    for (int x = 0; x < c->tbs_w; x++) {
      int tbX = (x << P->log2_min_tb_size) >> P->log2_ctb_size;
      int tbY = (y << P->log2_min_tb_size) >> P->log2_ctb_size;
      int v = rs2ts[c->ctbs_w*tbY + tbX] << (dl*2);
      int p = 0;
      for (int i = 0; i < dl; i++) { int m = 1 << i; p += (m & x ? m*m : 0) + (m & y ? 2*m*m : 0); }
      c->min_tb_addr_zs[x + y*c->tbs_w] = v + p;
    }
  free(rs2ts);
  return 0;
}
static void octx_free(octx* c) { free(c->min_tb_addr_zs); free(c->tile_id); free(c->residual_luma); }

/* ---------------- a1: dequantisation, transform.cc:452-510 ---------------- */
void oracle_dequant(int16_t* coeff_buf, int log2_size, int c_idx, int intra, int qp,
                    int bit_depth, const int16_t* vals, const uint16_t* pos, int n,
                    const uint8_t* sf)
{
  int nT = 1 << log2_size;
  int bdShift = bit_depth + log2_size - 5;
  if (!sf) {
    bdShift -= 4;                                       /* m_x_y = 1 instead of 16 (:457-459) */
    int32_t offset = 1 << (bdShift - 1);
    int32_t fact = level_scale[qp % 6] << (qp / 6);
    for (int i = 0; i < n; i++) {
      /* int32 multiply with wraparound (:464-470) */
      int32_t cc = (int32_t)((uint32_t)(int32_t)vals[i] * (uint32_t)fact + (uint32_t)offset);
      cc = clip3(-32768, 32767, cc >> bdShift);
      coeff_buf[pos[i]] = (int16_t)cc;
    }
  } else {
    int64_t offset = 1 << (bdShift - 1);
    int matrixID = c_idx;
    if (!intra) { if (nT < 32) matrixID += 3; else matrixID++; }    /* :480-484 */
    const uint8_t* scl;
    switch (nT) {
      case 4:  scl = sf + matrixID*16; break;
      case 8:  scl = sf + 6*16 + matrixID*64; break;
      case 16: scl = sf + 6*16 + 6*64 + matrixID*256; break;
      default: scl = sf + 6*16 + 6*64 + 6*256 + matrixID*1024; break;
    }
    for (int i = 0; i < n; i++) {
      int p = pos[i];
      int x = p % nT, y = p / nT;
      int m = scl[x + y*nT];
      int fact = (m * level_scale[qp % 6]) << (qp / 6);
      int64_t cc = vals[i];
      cc = (cc * fact + offset) >> bdShift;
      if (cc < -32768) cc = -32768;
      if (cc > 32767) cc = 32767;
      coeff_buf[p] = (int16_t)cc;
    }
  }
}

/* ---------------- pixel-generic part, twice ---------------- */
#define PX uint8_t
#define FN(x) x##_8
#include "oracle_px.inc"
#undef PX
#undef FN
#define PX uint16_t
#define FN(x) x##_16
#include "oracle_px.inc"
#undef PX
#undef FN

/* ---------------- a12: derive_boundaryStrength, deblock.cc:241-375 ---------------- */
static void derive_bs(const octx* c, int vertical, uint8_t* bs)
{
  const de265hip_picture_desc* d = c->d;
  int xIncr = vertical ? 2 : 1, yIncr = vertical ? 1 : 2;
  int xOffs = vertical ? 1 : 0, yOffs = vertical ? 0 : 1;
  int edgeMask = vertical ? (DE265HIP_BLK_EDGE_TU_V | DE265HIP_BLK_EDGE_PB_V)
                          : (DE265HIP_BLK_EDGE_TU_H | DE265HIP_BLK_EDGE_PB_H);
  int tuMask = vertical ? DE265HIP_BLK_EDGE_TU_V : DE265HIP_BLK_EDGE_TU_H;
  memset(bs, 0, (size_t)c->w4 * c->h4);
  for (int y = 0; y < c->h4; y += yIncr)
    for (int x = 0; x < c->w4; x += xIncr) {
      int f = d->blk_flags[x + y*c->w4];
      int b = 0;
      if (f & edgeMask) {
        int xo = x - xOffs, yo = y - yOffs;            /* opposing (P) side unit */
        int fp = d->blk_flags[xo + yo*c->w4];
        if ((fp & DE265HIP_BLK_INTRA) || (f & DE265HIP_BLK_INTRA)) b = 2;
        else if ((f & tuMask) && ((f & DE265HIP_BLK_NONZERO) || (fp & DE265HIP_BLK_NONZERO))) b = 1;
        else {
          const de265hip_motion* mP = &d->blk_motion[xo + yo*c->w4];
          const de265hip_motion* mQ = &d->blk_motion[x + y*c->w4];
          int rP0 = mP->ref_slot[0], rP1 = mP->ref_slot[1];
          int rQ0 = mQ->ref_slot[0], rQ1 = mQ->ref_slot[1];
          int same = (rP0 == rQ0 && rP1 == rQ1) || (rP0 == rQ1 && rP1 == rQ0);
          if (!same) b = 1;
          else {
            int p0x = rP0 >= 0 ? mP->mv[0][0] : 0, p0y = rP0 >= 0 ? mP->mv[0][1] : 0;
            int p1x = rP1 >= 0 ? mP->mv[1][0] : 0, p1y = rP1 >= 0 ? mP->mv[1][1] : 0;
            int q0x = rQ0 >= 0 ? mQ->mv[0][0] : 0, q0y = rQ0 >= 0 ? mQ->mv[0][1] : 0;
            int q1x = rQ1 >= 0 ? mQ->mv[1][0] : 0, q1y = rQ1 >= 0 ? mQ->mv[1][1] : 0;
            if (rP0 != rP1) {
              if (rP0 == rQ0) {
                if (iabs(p0x-q0x) >= 4 || iabs(p0y-q0y) >= 4 || iabs(p1x-q1x) >= 4 || iabs(p1y-q1y) >= 4) b = 1;
              } else {
                if (iabs(p0x-q1x) >= 4 || iabs(p0y-q1y) >= 4 || iabs(p1x-q0x) >= 4 || iabs(p1y-q0y) >= 4) b = 1;
              }
            } else {
              if ((iabs(p0x-q0x) >= 4 || iabs(p0y-q0y) >= 4 || iabs(p1x-q1x) >= 4 || iabs(p1y-q1y) >= 4) &&
                  (iabs(p0x-q1x) >= 4 || iabs(p0y-q1y) >= 4 || iabs(p1x-q0x) >= 4 || iabs(p1y-q0y) >= 4)) b = 1;
            }
          }
        }
      }
      bs[x + y*c->w4] = (uint8_t)b;
    }
}

void oracle_derive_bs(const de265hip_picture_desc* d, int vertical, uint8_t* bs)
{
  octx c;
  if (octx_init(&c, d, NULL, NULL)) return;
  derive_bs(&c, vertical, bs);
  octx_free(&c);
}

/* ---------------- a11: derive_edgeFlags, deblock.cc:31-225 ---------------- */
static void set_flag(uint8_t* f, int w4, int h4, int x0, int y0, int bits)
{ int xd = x0/4, yd = y0/4; if (xd < w4 && yd < h4) f[xd + yd*w4] |= (uint8_t)bits; }

static void mark_tb(const de265hip_pic_params* P, const uint8_t* tu_split, int tbs_w,
                    uint8_t* f, int w4, int h4, int x0, int y0, int log2, int depth,
                    int left, int top)
{
  int split = (tu_split[(x0 >> P->log2_min_tb_size) + (y0 >> P->log2_min_tb_size)*tbs_w] >> depth) & 1;
  if (split) {
    int x1 = x0 + ((1 << log2) >> 1), y1 = y0 + ((1 << log2) >> 1);
    mark_tb(P, tu_split, tbs_w, f, w4, h4, x0, y0, log2-1, depth+1, left, top);
    mark_tb(P, tu_split, tbs_w, f, w4, h4, x1, y0, log2-1, depth+1, DE265HIP_BLK_EDGE_TU_V, top);
    mark_tb(P, tu_split, tbs_w, f, w4, h4, x0, y1, log2-1, depth+1, left, DE265HIP_BLK_EDGE_TU_H);
    mark_tb(P, tu_split, tbs_w, f, w4, h4, x1, y1, log2-1, depth+1, DE265HIP_BLK_EDGE_TU_V, DE265HIP_BLK_EDGE_TU_H);
  } else {
    for (int k = 0; k < (1 << log2); k += 4) set_flag(f, w4, h4, x0, y0 + k, left);
    for (int k = 0; k < (1 << log2); k += 4) set_flag(f, w4, h4, x0 + k, y0, top);
  }
}

int oracle_derive_edge_flags(const de265hip_pic_params* P,
                             const de265hip_slice_params* slices, int n_slices,
                             const de265hip_ctb_info* ctbs,
                             const uint8_t* cb_log2_size, const uint8_t* cb_part_mode,
                             const uint8_t* tu_split, uint8_t* f)
{
  (void)n_slices;
  octx c; de265hip_picture_desc dd; memset(&dd, 0, sizeof(dd));
  dd.params = *P;
  int ctb = 1 << P->log2_ctb_size;
  dd.n_ctbs = ((P->width + ctb - 1) >> P->log2_ctb_size) * ((P->height + ctb - 1) >> P->log2_ctb_size);
  int rc = octx_init(&c, &dd, NULL, NULL);
  if (rc) return rc;
  int minCb = 1 << P->log2_min_cb_size;
  int cbs_w = (P->width + minCb - 1) / minCb, cbs_h = (P->height + minCb - 1) / minCb;
  int mask = ctb - 1, sh = P->log2_ctb_size;
  for (int cy = 0; cy < cbs_h; cy++)
    for (int cx = 0; cx < cbs_w; cx++) {
      int log2Cb = cb_log2_size[cx + cy*cbs_w];
      if (!log2Cb) continue;
      int x0 = cx*minCb, y0 = cy*minCb;
      int x0c = x0 >> sh, y0c = y0 >> sh;
      const de265hip_ctb_info* ci = &ctbs[x0c + y0c*c.ctbs_w];
      const de265hip_slice_params* shdr = &slices[ci->slice_idx];
      int left = DE265HIP_BLK_EDGE_TU_V, top = DE265HIP_BLK_EDGE_TU_H;
      if (x0 == 0) left = 0;
      if (y0 == 0) top = 0;
      if (x0 && (x0 & mask) == 0) {                     /* :182-195 */
        const de265hip_ctb_info* li = &ctbs[((x0-1) >> sh) + y0c*c.ctbs_w];
        if (shdr->slice_loop_filter_across_slices_enabled_flag == 0 &&
            ci->slice_addr_rs != li->slice_addr_rs) left = 0;
        else if (P->loop_filter_across_tiles_enabled_flag == 0 &&
                 c.tile_id[x0c + y0c*c.ctbs_w] != c.tile_id[((x0-1) >> sh) + y0c*c.ctbs_w]) left = 0;
      }
      if (y0 && (y0 & mask) == 0) {                     /* :197-210 */
        const de265hip_ctb_info* ti = &ctbs[x0c + ((y0-1) >> sh)*c.ctbs_w];
        if (shdr->slice_loop_filter_across_slices_enabled_flag == 0 &&
            ci->slice_addr_rs != ti->slice_addr_rs) top = 0;
        else if (P->loop_filter_across_tiles_enabled_flag == 0 &&
                 c.tile_id[x0c + y0c*c.ctbs_w] != c.tile_id[x0c + ((y0-1) >> sh)*c.ctbs_w]) top = 0;
      }
      if (shdr->slice_deblocking_filter_disabled_flag) continue;
      mark_tb(P, tu_split, c.tbs_w, f, c.w4, c.h4, x0, y0, log2Cb, 0, left, top);
      int cb = 1 << log2Cb, h2 = cb >> 1, q4 = cb >> 2;   /* markPredictionBlockBoundary :66-127 */
      int pm = cb_part_mode[cx + cy*cbs_w];
      for (int k = 0; k < cb; k++) {
        switch (pm) {                                      /* enum PartMode, slice.h */
          case 3: set_flag(f, c.w4, c.h4, x0+h2, y0+k, DE265HIP_BLK_EDGE_PB_V);    /* PART_NxN */
                  set_flag(f, c.w4, c.h4, x0+k, y0+h2, DE265HIP_BLK_EDGE_PB_H); break;
          case 2: set_flag(f, c.w4, c.h4, x0+h2, y0+k, DE265HIP_BLK_EDGE_PB_V); break;       /* Nx2N */
          case 1: set_flag(f, c.w4, c.h4, x0+k, y0+h2, DE265HIP_BLK_EDGE_PB_H); break;       /* 2NxN */
          case 6: set_flag(f, c.w4, c.h4, x0+q4, y0+k, DE265HIP_BLK_EDGE_PB_V); break;       /* nLx2N */
          case 7: set_flag(f, c.w4, c.h4, x0+h2+q4, y0+k, DE265HIP_BLK_EDGE_PB_V); break;    /* nRx2N */
          case 4: set_flag(f, c.w4, c.h4, x0+k, y0+q4, DE265HIP_BLK_EDGE_PB_H); break;       /* 2NxnU */
          case 5: set_flag(f, c.w4, c.h4, x0+k, y0+h2+q4, DE265HIP_BLK_EDGE_PB_H); break;    /* 2NxnD */
          default: break;                                                                     /* 2Nx2N */
        }
      }
    }
  octx_free(&c);
  return 0;
}

/* ---------------- picture-level drivers ---------------- */
static int hi_depth(const de265hip_pic_params* P) { return P->bit_depth_luma > 8; }

static int check_params(const de265hip_pic_params* P)
{
  if (P->chroma_format_idc < 0 || P->chroma_format_idc > 3 || P->extended_precision_processing_flag) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  if ((P->bit_depth_luma > 8) != (P->bit_depth_chroma > 8)) return DE265HIP_ERROR_NOT_IMPLEMENTED;
  if (P->bit_depth_luma < 8 || P->bit_depth_luma > 12 || P->bit_depth_chroma < 8 ||
      P->bit_depth_chroma > 12) return DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
  return 0;
}

static int do_pu(const octx* c, int i)
{ return hi_depth(&c->d->params) ? inter_pred_pu_16(c, &c->d->pus[i]) : inter_pred_pu_8(c, &c->d->pus[i]); }
static void do_pcm(const octx* c, int i)
{ if (hi_depth(&c->d->params)) pcm_copy_16(c, &c->d->pcms[i]); else pcm_copy_8(c, &c->d->pcms[i]); }
static void do_tu(const octx* c, int i)
{ if (hi_depth(&c->d->params)) decode_tu_16(c, &c->d->tus[i]); else decode_tu_8(c, &c->d->tus[i]); }

/* apply_deblocking_filter, deblock.cc:1020-1058 (edge flags come with the desc) */
static void do_deblock(const octx* c)
{
  uint8_t* bs = (uint8_t*)malloc((size_t)c->w4 * c->h4);
  int any = 0;
  for (int i = 0; i < c->w4 * c->h4 && !any; i++) any = c->d->blk_flags[i] & 0xF0;
  if (any)
    for (int pass = 0; pass < 2; pass++) {
      int vertical = pass == 0;
      derive_bs(c, vertical, bs);
      if (hi_depth(&c->d->params)) { deblock_luma_16(c, vertical, bs); if (c->ncomp == 3) deblock_chroma_16(c, vertical, bs); }      /* deblock.cc:977, :1034 */
      else { deblock_luma_8(c, vertical, bs); if (c->ncomp == 3) deblock_chroma_8(c, vertical, bs); }
    }
  free(bs);
}

int oracle_reconstruct(const de265hip_picture_desc* d, const uint32_t* order, int n_order,
                       const oracle_image* dpb, oracle_image* img, oracle_image* scratch,
                       int last_stage)
{
  octx c;
  int rc = check_params(&d->params);
  if (rc) return rc;
  rc = octx_init(&c, d, img, dpb);
  if (rc) return rc;
  if (order) {
    for (int k = 0; k < n_order && !rc; k++) {
      uint32_t e = order[k]; int idx = (int)ORACLE_ORD_IDX(e);
      switch (e & 0xF0000000u) {
        case ORACLE_ORD_PU:  rc = do_pu(&c, idx); break;
        case ORACLE_ORD_PCM: do_pcm(&c, idx); break;
        case ORACLE_ORD_TU:  do_tu(&c, idx); break;
        default: rc = DE265HIP_ERROR_PARAMETER_OUT_OF_RANGE;
      }
    }
  } else {
    for (int i = 0; i < d->n_pus && !rc; i++) rc = do_pu(&c, i);
    for (int i = 0; i < d->n_pcms; i++) do_pcm(&c, i);
    for (int i = 0; i < d->n_tus; i++) do_tu(&c, i);
  }
  /* run_postprocessing_filters_sequential, decctx.cc:1859-1885 */
  if (!rc && last_stage >= DE265HIP_STAGE_DEBLOCKED && !d->params.disable_deblocking) do_deblock(&c);
  if (!rc && last_stage >= DE265HIP_STAGE_FINAL && !d->params.disable_sao &&
      d->params.sample_adaptive_offset_enabled_flag) {
    if (hi_depth(&d->params)) sao_picture_16(&c, scratch); else sao_picture_8(&c, scratch);
  }
  octx_free(&c);
  return rc;
}

int oracle_stage_mc(const de265hip_picture_desc* d, const oracle_image* dpb, oracle_image* img)
{
  octx c; int rc = octx_init(&c, d, img, dpb); if (rc) return rc;
  for (int i = 0; i < d->n_pus && !rc; i++) rc = do_pu(&c, i);
  octx_free(&c); return rc;
}
int oracle_stage_pcm(const de265hip_picture_desc* d, oracle_image* img)
{
  octx c; int rc = octx_init(&c, d, img, NULL); if (rc) return rc;
  for (int i = 0; i < d->n_pcms; i++) do_pcm(&c, i);
  octx_free(&c); return 0;
}
int oracle_stage_tus(const de265hip_picture_desc* d, oracle_image* img)
{
  octx c; int rc = octx_init(&c, d, img, NULL); if (rc) return rc;
  for (int i = 0; i < d->n_tus; i++) do_tu(&c, i);
  octx_free(&c); return 0;
}
int oracle_stage_deblock(const de265hip_picture_desc* d, oracle_image* img)
{
  octx c; int rc = octx_init(&c, d, img, NULL); if (rc) return rc;
  do_deblock(&c);
  octx_free(&c); return 0;
}
int oracle_stage_sao(const de265hip_picture_desc* d, oracle_image* img, oracle_image* scratch)
{
  octx c; int rc = octx_init(&c, d, img, NULL); if (rc) return rc;
  if (d->params.sample_adaptive_offset_enabled_flag) {
    if (hi_depth(&d->params)) sao_picture_16(&c, scratch); else sao_picture_8(&c, scratch);
  }
  octx_free(&c); return 0;
}

/* ---------------- function-level exports ---------------- */
void oracle_transform_add(int log2_size, int is_dst, int bit_depth, void* dst, ptrdiff_t stride,
                          const int16_t* coeffs)
{
  int nT = 1 << log2_size;
  if (bit_depth > 8) {
    if (is_dst) dst_add_16((uint16_t*)dst, stride, coeffs, bit_depth);
    else idct_add_16((uint16_t*)dst, stride, nT, coeffs, bit_depth);
  } else {
    if (is_dst) dst_add_8((uint8_t*)dst, stride, coeffs, bit_depth);
    else idct_add_8((uint8_t*)dst, stride, nT, coeffs, bit_depth);
  }
}
void oracle_transform_skip_add(int log2_size, int bit_depth, void* dst, ptrdiff_t stride,
                               const int16_t* coeffs)
{
  if (bit_depth > 8) tskip_add_16((uint16_t*)dst, stride, 1 << log2_size, coeffs, bit_depth);
  else tskip_add_8((uint8_t*)dst, stride, 1 << log2_size, coeffs, bit_depth);
}
void oracle_transform_bypass_add(int log2_size, int bit_depth, void* dst, ptrdiff_t stride,
                                 const int16_t* coeffs)
{
  if (bit_depth > 8) bypass_add_16((uint16_t*)dst, stride, 1 << log2_size, coeffs, bit_depth);
  else bypass_add_8((uint8_t*)dst, stride, 1 << log2_size, coeffs, bit_depth);
}
void oracle_put_qpel(int bit_depth, int16_t* out, ptrdiff_t out_stride, const void* src,
                     ptrdiff_t src_stride, int w, int h, int dx, int dy)
{
  if (bit_depth > 8) put_qpel_16(out, out_stride, (const uint16_t*)src, src_stride, w, h, dx, dy, bit_depth);
  else put_qpel_8(out, out_stride, (const uint8_t*)src, src_stride, w, h, dx, dy, bit_depth);
}
void oracle_put_epel(int bit_depth, int16_t* out, ptrdiff_t out_stride, const void* src,
                     ptrdiff_t src_stride, int w, int h, int mx, int my)
{
  if (bit_depth > 8) put_epel_16(out, out_stride, (const uint16_t*)src, src_stride, w, h, mx, my, bit_depth);
  else put_epel_8(out, out_stride, (const uint8_t*)src, src_stride, w, h, mx, my, bit_depth);
}
void oracle_put_pred(int mode, int bit_depth, void* dst, ptrdiff_t ds, const int16_t* s0,
                     const int16_t* s1, ptrdiff_t ss, int w, int h, int w0, int o0, int w1, int o1,
                     int log2wd)
{
  if (bit_depth > 8) {
    uint16_t* d = (uint16_t*)dst;
    switch (mode) {
      case 0: put_unweighted_16(d, ds, s0, ss, w, h, bit_depth); break;
      case 1: put_weighted_16(d, ds, s0, ss, w, h, w0, o0, log2wd, bit_depth); break;
      case 2: put_avg_16(d, ds, s0, s1, ss, w, h, bit_depth); break;
      default: put_weighted_bi_16(d, ds, s0, s1, ss, w, h, w0, o0, w1, o1, log2wd, bit_depth); break;
    }
  } else {
    uint8_t* d = (uint8_t*)dst;
    switch (mode) {
      case 0: put_unweighted_8(d, ds, s0, ss, w, h, bit_depth); break;
      case 1: put_weighted_8(d, ds, s0, ss, w, h, w0, o0, log2wd, bit_depth); break;
      case 2: put_avg_8(d, ds, s0, s1, ss, w, h, bit_depth); break;
      default: put_weighted_bi_8(d, ds, s0, s1, ss, w, h, w0, o0, w1, o1, log2wd, bit_depth); break;
    }
  }
}
void oracle_intra_predict(int bit_depth, int strong, void* dst, ptrdiff_t stride, int nT, int c_idx,
                          int mode, const void* border_centre)
{
  if (bit_depth > 8) {
    uint16_t mem[4*32 + 1]; uint16_t* b = &mem[2*32];
    memcpy(b - 2*nT, (const uint16_t*)border_centre - 2*nT, (4*nT + 1)*sizeof(uint16_t));
    intra_from_border_16((uint16_t*)dst, stride, nT, c_idx, mode, b, strong, bit_depth, bit_depth, c_idx == 0, 0);
  } else {
    uint8_t mem[4*32 + 1]; uint8_t* b = &mem[2*32];
    memcpy(b - 2*nT, (const uint8_t*)border_centre - 2*nT, 4*nT + 1);
    intra_from_border_8((uint8_t*)dst, stride, nT, c_idx, mode, b, strong, bit_depth, bit_depth, c_idx == 0, 0);
  }
}
