/* lk_oracle.c - CPU ORACLE (test infrastructure only; see lk_oracle.h for the status
 * header: "parity unpinned" for sampling / pyramid / LM driver / 6x6 solve).
 *
 * Every function cites the reference file:line it restates.  Nothing here is used by
 * the product path.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off).
 */
#include "lk_oracle.h"
#include "lk_bicubic_matrix.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* images and pyramid                                                                   */
/* ------------------------------------------------------------------------------------ */

typedef struct {
  uint8_t *lvl[LKO_MAX_LEVELS]; /* owned; each padded by 2 zero rows so the reference's
                                   one-past-the-edge reads (nearest, last row) are
                                   defined here */
  int rows, cols;               /* level 0 */
  int n_levels;                 /* py_stop + 1 when set, 0 when empty */
} lko_image;

struct lko_engine {
  lko_config cfg;
  int n_params;
  lko_image img[3];
  uint8_t *cache_flag[LKO_MAX_LEVELS]; /* cache_mode 1: 0 unbuilt, 1 built, 2 built-while-in-error */
  int cache_rows, cache_cols;
  /* members of CorrelationClass that survive between calls (correlation_class.hpp:83-84) */
  int reached_iterations;
  float last_good_chi;
};

int lko_n_params(int model) { /* model_class.cpp:216-231 */
  switch (model) {
  case LKO_FM_U: return 1;
  case LKO_FM_UV: return 2;
  case LKO_FM_UVQ: return 3;
  case LKO_FM_UVUXUYVXVY: return 6;
  default: return -1;
  }
}

static void image_clear(lko_image *im) {
  for (int l = 0; l < LKO_MAX_LEVELS; ++l) {
    free(im->lvl[l]);
    im->lvl[l] = NULL;
  }
  im->rows = im->cols = im->n_levels = 0;
}

/* pyramid_class.cpp:83-122: 5x5 kernel = products of floats, float accumulation in
 * (dj outer, di inner) order, truncation to u8, untouched 1-pixel border stays 0. */
void lko_pyramid_level(const uint8_t *src, int rows, int cols, uint8_t *dst) {
  const float km[5] = {0.05f, 0.25f, 0.4f, 0.25f, 0.05f};
  float kernel[25];
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j)
      kernel[5 * j + i] = km[i] * km[j];

  int tcols = cols / 2, trows = rows / 2;
  memset(dst, 0, (size_t)tcols * (size_t)trows);
  for (int tj = 1; tj < trows - 1; ++tj) {
    for (int ti = 1; ti < tcols - 1; ++ti) {
      int si = ti * 2, sj = tj * 2;
      float addition = 0.f;
      for (int dj = -2; dj <= 2; ++dj)
        for (int di = -2; di <= 2; ++di) {
          unsigned char s = src[(size_t)cols * (size_t)(sj + dj) + (size_t)(si + di)];
          float ker = kernel[(2 + dj) * 5 + (2 + di)];
          addition += (float)s * ker;
        }
      dst[(size_t)tcols * tj + ti] = (unsigned char)addition;
    }
  }
}

static uint8_t *alloc_level(int rows, int cols) {
  return (uint8_t *)calloc((size_t)(rows + 2) * (size_t)cols + 16, 1);
}

static void cache_reset(lko_engine *e) { /* pyramid_class.cpp:388-414 */
  if (e->cfg.cache_mode != 1)
    return;
  const lko_image *d = &e->img[LKO_IMG_DEF];
  if (d->rows != e->cache_rows || d->cols != e->cache_cols) {
    for (int l = 0; l < LKO_MAX_LEVELS; ++l) {
      free(e->cache_flag[l]);
      e->cache_flag[l] = NULL;
    }
    e->cache_rows = d->rows;
    e->cache_cols = d->cols;
  }
  for (int l = e->cfg.py_start; l <= e->cfg.py_stop; l += e->cfg.py_step) {
    size_t n = (size_t)(d->rows >> l) * (size_t)(d->cols >> l);
    if (!e->cache_flag[l])
      e->cache_flag[l] = (uint8_t *)malloc(n ? n : 1);
    memset(e->cache_flag[l], 0, n);
  }
}

lko_engine *lko_create(const lko_config *cfg) {
  if (!cfg || lko_n_params(cfg->model) < 0 || cfg->interp < 0 || cfg->interp > 2)
    return NULL;
  if (cfg->py_step < 1 || cfg->py_start < 0 || cfg->py_stop < cfg->py_start ||
      cfg->py_stop >= LKO_MAX_LEVELS || (cfg->py_stop - cfg->py_start) % cfg->py_step)
    return NULL;
  lko_engine *e = (lko_engine *)calloc(1, sizeof(*e));
  e->cfg = *cfg;
  e->n_params = lko_n_params(cfg->model);
  e->reached_iterations = 0;
  e->last_good_chi = FLT_MAX;
  return e;
}

void lko_destroy(lko_engine *e) {
  if (!e)
    return;
  for (int i = 0; i < 3; ++i)
    image_clear(&e->img[i]);
  for (int l = 0; l < LKO_MAX_LEVELS; ++l)
    free(e->cache_flag[l]);
  free(e);
}

int lko_set_image(lko_engine *e, int which, const uint8_t *px, int rows, int cols) {
  if (!e || which < 0 || which > 2 || !px || rows < 1 || cols < 1)
    return LKO_ERR_BAD_DOMAIN;
  lko_image *im = &e->img[which];
  image_clear(im);
  im->rows = rows;
  im->cols = cols;
  im->n_levels = e->cfg.py_stop + 1;
  im->lvl[0] = alloc_level(rows, cols);
  memcpy(im->lvl[0], px, (size_t)rows * (size_t)cols);
  int r = rows, c = cols;
  for (int l = 1; l <= e->cfg.py_stop; ++l) { /* all levels 1..stop, pyramid_class.cpp:92 */
    im->lvl[l] = alloc_level(r / 2, c / 2);
    lko_pyramid_level(im->lvl[l - 1], r, c, im->lvl[l]);
    r /= 2;
    c /= 2;
  }
  if (which == LKO_IMG_DEF)
    cache_reset(e);
  return LKO_ERR_NONE;
}

void lko_und_from_def(lko_engine *e) { /* pyramid_class.cpp:211-226: moves, def emptied */
  image_clear(&e->img[LKO_IMG_UND]);
  e->img[LKO_IMG_UND] = e->img[LKO_IMG_DEF];
  memset(&e->img[LKO_IMG_DEF], 0, sizeof(lko_image));
}

void lko_def_from_nxt(lko_engine *e) { /* pyramid_class.cpp:228-258 */
  image_clear(&e->img[LKO_IMG_DEF]);
  e->img[LKO_IMG_DEF] = e->img[LKO_IMG_NXT];
  memset(&e->img[LKO_IMG_NXT], 0, sizeof(lko_image));
  cache_reset(e);
}

const uint8_t *lko_get_level(lko_engine *e, int which, int level, int *rows, int *cols) {
  if (!e || which < 0 || which > 2 || level < 0 || level >= e->img[which].n_levels)
    return NULL;
  if (rows)
    *rows = e->img[which].rows >> level; /* pyramid_class.cpp:447-477 */
  if (cols)
    *cols = e->img[which].cols >> level;
  return e->img[which].lvl[level];
}

/* ------------------------------------------------------------------------------------ */
/* interpolation                                                                        */
/* ------------------------------------------------------------------------------------ */

/* interpolation_class.cpp:243-336 (monochrome: colour index bug at :268-273 is inert) */
void lko_bicubic_coeffs(const uint8_t *img, int step, int ix, int iy, float a[16]) {
  int ix0 = ix - 1, iy0 = iy - 1, ix1 = ix, iy1 = iy;
  int ix2 = ix0 + 2, iy2 = iy0 + 2, ix3 = ix0 + 3, iy3 = iy0 + 3;
  const uint8_t *r0 = img + (size_t)step * iy0, *r1 = img + (size_t)step * iy1;
  const uint8_t *r2 = img + (size_t)step * iy2, *r3 = img + (size_t)step * iy3;
  /* wXY: X = column offset, Y = row offset */
  float w00 = (float)r0[ix0], w01 = (float)r1[ix0], w02 = (float)r2[ix0], w03 = (float)r3[ix0];
  float w10 = (float)r0[ix1], w11 = (float)r1[ix1], w12 = (float)r2[ix1], w13 = (float)r3[ix1];
  float w20 = (float)r0[ix2], w21 = (float)r1[ix2], w22 = (float)r2[ix2], w23 = (float)r3[ix2];
  float w30 = (float)r0[ix3], w31 = (float)r1[ix3], w32 = (float)r2[ix3], w33 = (float)r3[ix3];
  float v[16];
  v[0] = w11;
  v[1] = w21;
  v[2] = w12;
  v[3] = w22;
  v[4] = (w21 - w01) / 2.f;
  v[5] = (w31 - w11) / 2.f;
  v[6] = (w22 - w02) / 2.f;
  v[7] = (w32 - w12) / 2.f;
  v[8] = (w12 - w10) / 2.f;
  v[9] = (w22 - w20) / 2.f;
  v[10] = (w13 - w11) / 2.f;
  v[11] = (w23 - w21) / 2.f;
  v[12] = (w22 + w00 - w20 - w02) / 4.f;
  v[13] = (w32 + w10 - w30 - w12) / 4.f;
  v[14] = (w23 + w01 - w21 - w03) / 4.f;
  v[15] = (w33 + w11 - w31 - w13) / 4.f;
  for (int i = 0; i < 16; ++i) {
    float t = 0.f;
    for (int j = 0; j < 16; ++j)
      t += LKO_BICUBIC_M[i * 16 + j] * v[j];
    a[i] = t;
  }
}

/* evaluation part of interpolation_class.cpp:94-126 */
static void bicubic_eval(const float a[16], float dx, float dy, float *w, float *wx, float *wy) {
  float px[4] = {1.f, dx, dx * dx, dx * dx * dx};
  float py[4] = {1.f, dy, dy * dy, dy * dy * dy};
  float W = 0.f, Wx = 0.f, Wy = 0.f;
  for (int jk = 0; jk < 4; ++jk)
    for (int ik = 0; ik < 4; ++ik) {
      float c = a[jk * 4 + ik];
      W += c * py[jk] * px[ik];
      if (ik > 0)
        Wx += ik * c * py[jk] * px[ik - 1];
      if (jk > 0)
        Wy += jk * c * py[jk - 1] * px[ik];
    }
  *w = W;
  *wx = Wx;
  *wy = Wy;
}

/* state of one interpolator during one evaluation (error flag is reset per evaluation,
 * interpolation_class.cpp:652-653) */
typedef struct {
  int interp;
  const uint8_t *def;
  int rows, cols;
  uint8_t *flags; /* NULL unless cache_mode 1 */
  int error;
} interp_ctx;

/* returns 1 if this pixel's cached coefficients are the all-zero "poisoned" ones
 * (interpolation_class.cpp:228-250: built while error_status was set) */
static int cache_poisoned(interp_ctx *c, int ix, int iy) {
  if (!c->flags)
    return 0;
  uint8_t *f = &c->flags[(size_t)ix + (size_t)iy * (size_t)c->cols];
  if (*f == 0)
    *f = c->error ? 2 : 1;
  return *f == 2;
}

/* interpolation_class.cpp:79-226 */
static void interp_sample(interp_ctx *c, float xdef, float ydef, float *w, float *wx, float *wy) {
  const uint8_t *img = c->def;
  int cols = c->cols, rows = c->rows;
  *w = *wx = *wy = 0.f;
  switch (c->interp) {
  case LKO_IM_BICUBIC:
    if (xdef > 1.f && ydef > 1.f && xdef < cols - 2.f && ydef < rows - 2.f) {
      int ix = (int)xdef, iy = (int)ydef;
      if (cache_poisoned(c, ix, iy))
        return;
      float a[16];
      lko_bicubic_coeffs(img, cols, ix, iy, a);
      float dx = xdef - ix + 1.f, dy = ydef - iy + 1.f;
      bicubic_eval(a, dx, dy, w, wx, wy);
    } else {
      c->error = 1;
    }
    break;
  case LKO_IM_BILINEAR:
    if (xdef > 0 && ydef > 0 && xdef < cols - 1 && ydef < rows - 1) {
      int ix = (int)xdef, iy = (int)ydef;
      if (cache_poisoned(c, ix, iy))
        return;
      /* interpolation_class.cpp:338-374 */
      float w00 = (float)img[(size_t)cols * iy + ix], w01 = (float)img[(size_t)cols * (iy + 1) + ix];
      float w10 = (float)img[(size_t)cols * iy + ix + 1], w11 = (float)img[(size_t)cols * (iy + 1) + ix + 1];
      float a[4] = {w00, w10 - w00, w01 - w00, w11 - w10 - w01 + w00};
      float dx = xdef - ix, dy = ydef - iy;
      float px[2] = {1.f, dx}, py[2] = {1.f, dy};
      float W = 0.f, Wx = 0.f, Wy = 0.f;
      for (int jk = 0; jk < 2; ++jk)
        for (int ik = 0; ik < 2; ++ik) {
          float k = a[jk * 2 + ik];
          W += k * py[jk] * px[ik];
          if (ik > 0)
            Wx += k * py[jk];
          if (jk > 0)
            Wy += k * px[ik];
        }
      *w = W;
      *wx = Wx;
      *wy = Wy;
    } else {
      c->error = 1;
    }
    break;
  default: /* nearest, interpolation_class.cpp:197-226, :376-406 */
    if (xdef > 0 && ydef > 0 && xdef < cols - 1 && ydef < rows - 1) {
      int ix = (int)(xdef + 0.5f), iy = (int)(ydef + 0.5f);
      if (cache_poisoned(c, ix, iy))
        return;
      float w00 = (float)img[(size_t)cols * iy + ix];
      float w01 = (float)img[(size_t)cols * (iy + 1) + ix];
      float w10 = (float)img[(size_t)cols * iy + ix + 1];
      *w = w00;
      *wx = w10 - w00;
      *wy = w01 - w00;
    } else {
      c->error = 1;
    }
    break;
  }
}

int lko_interpolate(int interp, const uint8_t *img, int rows, int cols, float x, float y,
                    float *w, float *wx, float *wy) {
  interp_ctx c = {interp, img, rows, cols, NULL, 0};
  interp_sample(&c, x, y, w, wx, wy);
  return c.error;
}

/* ------------------------------------------------------------------------------------ */
/* warp model                                                                           */
/* ------------------------------------------------------------------------------------ */

/* model_class.cpp:48-202 */
void lko_model_point(int model, float x, float y, float cx, float cy, const float *p,
                     float *xd, float *yd, float dTx[6], float dTy[6]) {
  for (int i = 0; i < 6; ++i)
    dTx[i] = dTy[i] = 0.f;
  switch (model) {
  case LKO_FM_U:
    *xd = x + p[0];
    *yd = y;
    dTx[0] = 1;
    dTy[0] = 0;
    break;
  case LKO_FM_UV:
    *xd = x + p[0];
    *yd = y + p[1];
    dTx[0] = 1;
    dTy[1] = 1;
    break;
  case LKO_FM_UVQ: {
    float dx = x - cx, dy = y - cy;
    *xd = x + p[0] - p[2] * dy;
    *yd = y + p[1] + p[2] * dx;
    dTx[0] = 1;
    dTx[2] = -dy;
    dTy[1] = 1;
    dTy[2] = dx;
    break;
  }
  default: {
    float dx = x - cx, dy = y - cy;
    *xd = x + p[0] + p[2] * dx + p[3] * dy;
    *yd = y + p[1] + p[4] * dx + p[5] * dy;
    dTx[0] = 1;
    dTx[2] = dx;
    dTx[3] = dy;
    dTy[1] = 1;
    dTy[4] = dx;
    dTy[5] = dy;
    break;
  }
  }
}

/* ------------------------------------------------------------------------------------ */
/* one evaluation: warp -> sample -> residual -> A, b, chi                              */
/* ------------------------------------------------------------------------------------ */

/* model_class.cpp compute_model + interpolation_class.cpp:671-764, one thread
 * (correlation_class.cpp:131-300 with number_of_threads = 1). A: row-major P x P, upper. */
static int evaluate_chunk(interp_ctx *ic, int model, int P, const uint8_t *und, int ustep,
                          const float *xy, int n, float cx, float cy, const float *p, float *A,
                          float *b, float *chi_out) {
  float chi = 0.f;
  for (int i = 0; i < P; ++i) {
    b[i] = 0.f;
    for (int j = 0; j < P; ++j)
      A[i * P + j] = 0.f;
  }
  float H[6], dTx[6], dTy[6];
  for (int i = 0; i < n; ++i) {
    float x = xy[2 * i], y = xy[2 * i + 1];
    float xd, yd;
    lko_model_point(model, x, y, cx, cy, p, &xd, &yd, dTx, dTy);
    int und_ix = (int)(x + 0.5f), und_iy = (int)(y + 0.5f);
    float W, Wx, Wy;
    interp_sample(ic, xd, yd, &W, &Wx, &Wy);
    float und_w = (float)und[(size_t)ustep * und_iy + und_ix];
    float V = und_w - W;
    chi += V * V;
    for (int q = 0; q < P; ++q)
      H[q] = Wx * dTx[q] + Wy * dTy[q];
    for (int p1 = 0; p1 < P; ++p1) {
      b[p1] += H[p1] * V;
      for (int p2 = p1; p2 < P; ++p2)
        A[p1 * P + p2] += H[p1] * H[p2];
    }
  }
  *chi_out = chi;
  return ic->error;
}

/* apply_model_and_interpolate (correlation_class.cpp:131-300): T contiguous chunks
 * (n/T samples, the first n%T get one more, :169-186), each summed by its own
 * interpolator from zero, totals accumulated in thread order into flushed A, b, chi
 * (:253-275); the error flag is the OR over chunks (:277-279). */
static _Thread_local int g_eval_threads = 1; /* set per engine call; the oracle is single-caller */
static int evaluate(interp_ctx *ic, int model, int P, const uint8_t *und, int ustep,
                    const float *xy, int n, float cx, float cy, const float *p, float *A,
                    float *b, float *chi_out) {
  int T = g_eval_threads < 1 ? 1 : g_eval_threads;
  float chi = 0.f;
  for (int i = 0; i < P; ++i) {
    b[i] = 0.f;
    for (int j = 0; j < P; ++j)
      A[i * P + j] = 0.f;
  }
  int err = 0, first = 0;
  for (int t = 0; t < T; ++t) {
    int cnt = n / T + (t < n % T ? 1 : 0);
    float At[36], bt[6], chit;
    ic->error = 0; /* every interpolator object starts an evaluation clean (:652-653) */
    err |= evaluate_chunk(ic, model, P, und, ustep, xy + 2 * (size_t)first, cnt, cx, cy, p, At, bt, &chit);
    chi += chit;
    for (int p1 = 0; p1 < P; ++p1) {
      b[p1] += bt[p1];
      for (int p2 = p1; p2 < P; ++p2)
        A[p1 * P + p2] += At[p1 * P + p2];
    }
    first += cnt;
  }
  ic->error = err;
  *chi_out = chi;
  return err;
}

int lko_evaluate(int interp, int model, const uint8_t *und, int urows, int ucols,
                 const uint8_t *def, int drows, int dcols, const float *xy, int n,
                 float cx, float cy, const float *p, float A[36], float b[6], float *chi) {
  (void)urows;
  interp_ctx ic = {interp, def, drows, dcols, NULL, 0};
  int P = lko_n_params(model);
  float At[36], bt[6];
  g_eval_threads = 1;
  int err = evaluate(&ic, model, P, und, ucols, xy, n, cx, cy, p, At, bt, chi);
  memset(A, 0, 36 * sizeof(float));
  memset(b, 0, 6 * sizeof(float));
  for (int i = 0; i < P; ++i) {
    b[i] = bt[i];
    for (int j = 0; j < P; ++j)
      A[i * 6 + j] = At[i * P + j];
  }
  return err ? LKO_ERR_INTERP_OUT_OF_IMAGE : LKO_ERR_NONE;
}

/* ------------------------------------------------------------------------------------ */
/* 6x6 solve: Eigen 3.4.0 ColPivHouseholderQR, restated                                  */
/* ------------------------------------------------------------------------------------ */
/* Third-party algorithm absent from /root/reference (README.md:22 pins Eigen 3.4.0; call
 * site correlation_class.cpp:742-747).  Restated from Eigen's published algorithm:
 *   ColPivHouseholderQR::computeInPlace  (column norms with the LAPACK xGEQPF norm
 *   down-dating of LAWN 176, largest-remaining-column pivoting, makeHouseholderInPlace,
 *   applyHouseholderOnTheLeft) and ::_solve_impl (c = Q^T b, upper back-substitution in
 *   column-axpy form, un-permute).
 * Eigen evaluates its dot products / norms with SIMD packets in an order that depends on
 * the build's vector width; here they are plain ascending scalar sums.  This is the
 * "parity unpinned" boundary: results agree with any Eigen build to rounding, not bit
 * for bit.  M is column-major n x n. */
void lko_colpiv_qr_solve(int n, const float *Ain, const float *bin, float *x) {
  float M[36], hc[6], normU[6], normD[6], c[6];
  int trans[6], perm[6];
  const float eps = FLT_EPSILON;
  for (int i = 0; i < n * n; ++i)
    M[i] = Ain[i];
#define QR(r, cc) M[(cc)*n + (r)]
  float maxn = 0.f;
  for (int k = 0; k < n; ++k) {
    float s = 0.f;
    for (int r = 0; r < n; ++r)
      s += QR(r, k) * QR(r, k);
    normD[k] = sqrtf(s);
    normU[k] = normD[k];
    if (normU[k] > maxn)
      maxn = normU[k];
  }
  float threshold_helper = (maxn * eps) * (maxn * eps) / (float)n;
  float norm_downdate_threshold = sqrtf(eps);
  int nonzero_pivots = n;
  for (int k = 0; k < n; ++k) {
    int big = k;
    float bigv = normU[k];
    for (int j = k + 1; j < n; ++j)
      if (normU[j] > bigv) {
        bigv = normU[j];
        big = j;
      }
    float big_sq = bigv * bigv;
    if (nonzero_pivots == n && big_sq < threshold_helper * (float)(n - k))
      nonzero_pivots = k;
    trans[k] = big;
    if (k != big) {
      for (int r = 0; r < n; ++r) {
        float t = QR(r, k);
        QR(r, k) = QR(r, big);
        QR(r, big) = t;
      }
      float t = normU[k];
      normU[k] = normU[big];
      normU[big] = t;
      t = normD[k];
      normD[k] = normD[big];
      normD[big] = t;
    }
    /* makeHouseholderInPlace on col k, rows k..n-1 */
    float tailSq = 0.f;
    for (int r = k + 1; r < n; ++r)
      tailSq += QR(r, k) * QR(r, k);
    float c0 = QR(k, k), beta, tau;
    if (tailSq <= FLT_MIN) {
      tau = 0.f;
      beta = c0;
      for (int r = k + 1; r < n; ++r)
        QR(r, k) = 0.f;
    } else {
      beta = sqrtf(c0 * c0 + tailSq);
      if (c0 >= 0.f)
        beta = -beta;
      float den = c0 - beta;
      for (int r = k + 1; r < n; ++r)
        QR(r, k) = QR(r, k) / den;
      tau = (beta - c0) / beta;
    }
    hc[k] = tau;
    QR(k, k) = beta;
    /* applyHouseholderOnTheLeft to the trailing block (rows k.., cols k+1..) */
    if (n - k == 1) {
      /* block has one row and zero columns: nothing to do */
    } else if (tau != 0.f) {
      for (int j = k + 1; j < n; ++j) {
        float tmp = 0.f;
        for (int r = k + 1; r < n; ++r)
          tmp += QR(r, k) * QR(r, j);
        tmp += QR(k, j);
        QR(k, j) -= tau * tmp;
        for (int r = k + 1; r < n; ++r)
          QR(r, j) -= tmp * (tau * QR(r, k));
      }
    }
    /* norm down-dating */
    for (int j = k + 1; j < n; ++j) {
      if (normU[j] != 0.f) {
        float temp = fabsf(QR(k, j)) / normU[j];
        temp = (1.f + temp) * (1.f - temp);
        temp = temp < 0.f ? 0.f : temp;
        float ratio = normU[j] / normD[j];
        float temp2 = temp * (ratio * ratio);
        if (temp2 <= norm_downdate_threshold) {
          float s = 0.f;
          for (int r = k + 1; r < n; ++r)
            s += QR(r, j) * QR(r, j);
          normD[j] = sqrtf(s);
          normU[j] = normD[j];
        } else {
          normU[j] *= sqrtf(temp);
        }
      }
    }
  }
  for (int k = 0; k < n; ++k)
    perm[k] = k;
  for (int k = 0; k < n; ++k) { /* applyTranspositionOnTheRight */
    int t = perm[k];
    perm[k] = perm[trans[k]];
    perm[trans[k]] = t;
  }
  /* solve */
  if (nonzero_pivots == 0) {
    for (int i = 0; i < n; ++i)
      x[i] = 0.f;
    return;
  }
  for (int i = 0; i < n; ++i)
    c[i] = bin[i];
  for (int k = 0; k < nonzero_pivots; ++k) { /* c = H_k c, k ascending */
    int len = n - k;
    if (len == 1) {
      c[k] *= 1.f - hc[k];
    } else if (hc[k] != 0.f) {
      float tmp = 0.f;
      for (int r = k + 1; r < n; ++r)
        tmp += QR(r, k) * c[r];
      tmp += c[k];
      c[k] -= hc[k] * tmp;
      for (int r = k + 1; r < n; ++r)
        c[r] -= tmp * (hc[k] * QR(r, k));
    }
  }
  for (int i = nonzero_pivots - 1; i >= 0; --i) { /* upper triangular, column-axpy form */
    c[i] = c[i] / QR(i, i);
    for (int r = 0; r < i; ++r)
      c[r] -= c[i] * QR(r, i);
  }
  for (int i = 0; i < nonzero_pivots; ++i)
    x[perm[i]] = c[i];
  for (int i = nonzero_pivots; i < n; ++i)
    x[perm[i]] = 0.f;
#undef QR
}

/* lko_config.solver: 0 = the reference's solver (default), 1 = float32 root-free Cholesky,
 * 2 = float64 Gaussian elimination.  1 and 2 are yardsticks: they measure how much of the
 * engine-vs-reference deviation is explained by using a different backward-stable solver. */
static _Thread_local int g_solver = 0;

static void solve_ldlt_f32(int n, const float *A, const float *b, float *x) {
  float U[6][6], d[6], inv[6], y[6], dmax = 0.f;
  for (int i = 0; i < n; ++i) {
    for (int j = i; j < n; ++j)
      U[i][j] = A[i * n + j];
    if (U[i][i] > dmax)
      dmax = U[i][i];
    y[i] = b[i];
  }
  float tiny = dmax * 1e-7f;
  for (int j = 0; j < n; ++j) {
    float w[6], dj = U[j][j];
    for (int k = 0; k < j; ++k) {
      w[k] = U[k][j] * d[k];
      dj = fmaf(-U[k][j], w[k], dj);
    }
    int ok = dj > tiny;
    d[j] = ok ? dj : 0.f;
    inv[j] = ok ? 1.f / dj : 0.f;
    for (int i = j + 1; i < n; ++i) {
      float t = U[j][i];
      for (int k = 0; k < j; ++k)
        t = fmaf(-w[k], U[k][i], t);
      U[j][i] = t * inv[j];
    }
  }
  for (int j = 0; j < n; ++j)
    for (int k = 0; k < j; ++k)
      y[j] = fmaf(-U[k][j], y[k], y[j]);
  for (int j = n - 1; j >= 0; --j) {
    float t = y[j] * inv[j];
    for (int i = j + 1; i < n; ++i)
      t = fmaf(-U[j][i], x[i], t);
    x[j] = t;
  }
}

static void solve_f64(int n, const float *A, const float *b, float *x) {
  double M[6][7];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j)
      M[i][j] = A[i * n + j];
    M[i][n] = b[i];
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (fabs(M[r][c]) > fabs(M[piv][c]))
        piv = r;
    for (int j = 0; j <= n; ++j) {
      double t = M[c][j];
      M[c][j] = M[piv][j];
      M[piv][j] = t;
    }
    for (int r = c + 1; r < n; ++r) {
      double f = M[r][c] / M[c][c];
      for (int j = c; j <= n; ++j)
        M[r][j] -= f * M[c][j];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double t = M[i][n];
    for (int j = i + 1; j < n; ++j)
      t -= M[i][j] * x[j];
    x[i] = (float)(t / M[i][i]);
  }
}

/* correlation_class.cpp:642-704 (scale, mirror, damp) + :719-768 (solve) */
void lko_damped_solve(int n, float *A, float *b, float lambda, float scaling, float *dp) {
  for (int p1 = 0; p1 < n; ++p1) {
    b[p1] *= scaling;
    for (int p2 = p1; p2 < n; ++p2)
      A[p1 * n + p2] *= scaling;
  }
  for (int p1 = 0; p1 < n; ++p1) {
    for (int p2 = 0; p2 < p1; ++p2)
      A[p1 * n + p2] = A[p2 * n + p1];
    A[p1 * n + p1] *= (1.f + lambda);
  }
  if (g_solver == 1)
    solve_ldlt_f32(n, A, b, dp);
  else if (g_solver == 2)
    solve_f64(n, A, b, dp);
  else
    lko_colpiv_qr_solve(n, A, b, dp); /* symmetric: row-major == column-major view */
}

/* ------------------------------------------------------------------------------------ */
/* per-level sample lists, centres, parameter rescale                                   */
/* ------------------------------------------------------------------------------------ */

/* pyramid_class.cpp:301-322 for one level transition */
int lko_decimate(const float *xy_prev, int n_prev, int level_delta, float *out) {
  int magnification = 1 << level_delta;
  float inv = 1.f / (float)magnification;
  int m = 0;
  for (int i = 0; i < n_prev; ++i) {
    int ix = (int)(xy_prev[2 * i] + 0.5f), iy = (int)(xy_prev[2 * i + 1] + 0.5f);
    if (ix % magnification == 0 && iy % magnification == 0) {
      if (out) {
        out[2 * m] = xy_prev[2 * i] * inv;
        out[2 * m + 1] = xy_prev[2 * i + 1] * inv;
      }
      ++m;
    }
  }
  return m;
}

/* pyramid_class.cpp:260-287 */
void lko_translate_parameters(int n_params, float *p, int level_src, int level_dst) {
  float mag;
  if (level_dst - level_src > 0)
    mag = 1.f / (float)(1 << (level_dst - level_src));
  else
    mag = (float)(1 << (-level_dst + level_src));
  for (int i = 0; i < (n_params < 2 ? n_params : 2); ++i)
    p[i] *= mag;
}

/* ------------------------------------------------------------------------------------ */
/* Newton_Raphson                                                                       */
/* ------------------------------------------------------------------------------------ */

static void trace_push(lko_trace_rec *trace, int cap, int *cnt, int level, int kind, int iteration,
                       int P, const float *p_in, float chi, float lambda, const float *A,
                       const float *b, const float *dp, int error) {
  if (trace && *cnt < cap) {
    lko_trace_rec *t = &trace[*cnt];
    memset(t, 0, sizeof(*t));
    t->level = level;
    t->kind = kind;
    t->iteration = iteration;
    t->chi = chi;
    t->lambda = lambda;
    t->error = error;
    for (int i = 0; i < P; ++i) {
      t->p_in[i] = p_in[i];
      t->b[i] = b[i];
      if (dp)
        t->dp[i] = dp[i];
      for (int j = 0; j < P; ++j)
        t->A[i * 6 + j] = A[i * P + j];
    }
  }
  ++*cnt;
}

int lko_newton_raphson(lko_engine *e, float *p, int n0, const float *xy, int use_center,
                       float cx_in, float cy_in, lko_result *out, lko_trace_rec *trace,
                       int trace_cap, int *n_trace) {
  const lko_config *cfg = &e->cfg;
  const int P = e->n_params;
  const lko_image *und = &e->img[LKO_IMG_UND], *def = &e->img[LKO_IMG_DEF];
  int tcount = 0;
  g_eval_threads = cfg->n_threads;
  g_solver = cfg->solver;
  if (n_trace)
    *n_trace = 0;
  if (und->n_levels == 0 || def->n_levels == 0 || n0 < 0)
    return LKO_ERR_BAD_DOMAIN;

  /* set_xy_positions, pyramid_class.cpp:289-323 */
  float *lxy[LKO_MAX_LEVELS] = {0};
  int ln[LKO_MAX_LEVELS] = {0};
  lxy[0] = (float *)xy;
  ln[0] = n0;
  int prev = 0, first = (cfg->py_start == 0 ? cfg->py_step : cfg->py_start);
  for (int l = first; l <= cfg->py_stop; l += cfg->py_step) {
    lxy[l] = (float *)malloc(sizeof(float) * 2 * (size_t)(ln[prev] > 0 ? ln[prev] : 1));
    ln[l] = lko_decimate(lxy[prev], ln[prev], l - prev, lxy[l]);
    prev = l;
  }
  /* centre: pyramid_class.cpp:325-362 */
  float cx0, cy0;
  if (use_center) {
    cx0 = cx_in;
    cy0 = cy_in;
  } else {
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < n0; ++i) {
      sx += xy[2 * i];
      sy += xy[2 * i + 1];
    }
    cx0 = sx / (float)n0;
    cy0 = sy / (float)n0;
  }
  float lcx[LKO_MAX_LEVELS], lcy[LKO_MAX_LEVELS];
  lcx[0] = cx0;
  lcy[0] = cy0;
  for (int l = first; l <= cfg->py_stop; l += cfg->py_step) {
    float inv = 1.f / (float)(1 << l);
    lcx[l] = cx0 * inv;
    lcy[l] = cy0 * inv;
  }

  float A[36], b[6], dp[6], chi = 0.f;
  float last_good_p[6], tentative_p[6], saved_p[6], p_eval[6];
  int error_status = 0, error_code = LKO_ERR_NONE;
  int level_old = 0;
  int early_return = 0;

  for (int level = cfg->py_stop; level >= cfg->py_start; level -= cfg->py_step) {
    lko_translate_parameters(P, p, level_old, level);
    error_status = 0;
    error_code = LKO_ERR_NONE;
    float lambda = 0.0001f;
    const float min_lambda = 1e-9f, max_lambda = 1e9f;
    e->last_good_chi = FLT_MAX;
    int n = ln[level];
    float scaling = 1.f / ((float)n);
    const float *sxy = lxy[level];
    const uint8_t *uimg = und->lvl[level], *dimg = def->lvl[level];
    int ucols = und->cols >> level;
    int drows = def->rows >> level, dcols = def->cols >> level;
    interp_ctx ic = {cfg->interp, dimg, drows, dcols,
                     cfg->cache_mode == 1 ? e->cache_flag[level] : NULL, 0};
    float cx = lcx[level], cy = lcy[level];

    for (int q = 0; q < P; ++q)
      last_good_p[q] = p[q];

    /* evaluation #0, correlation_class.cpp:410-437 */
    memcpy(p_eval, p, sizeof(float) * P);
    error_status = evaluate(&ic, cfg->model, P, uimg, ucols, sxy, n, cx, cy, p, A, b, &chi);
    if (error_status) {
      error_code = LKO_ERR_INTERP_OUT_OF_IMAGE;
      trace_push(trace, trace_cap, &tcount, level, 0, 0, P, p_eval, chi, lambda, A, b, NULL, 1);
      lko_translate_parameters(P, p, level, 0);
      early_return = 1;
      break;
    }
    chi *= scaling;
    e->last_good_chi = chi;
    {
      float Ac[36], bc[6];
      memcpy(Ac, A, sizeof(Ac));
      memcpy(bc, b, sizeof(bc));
      lko_damped_solve(P, A, b, lambda, scaling, dp);
      for (int q = 0; q < P; ++q)
        p[q] += dp[q];
      trace_push(trace, trace_cap, &tcount, level, 0, 0, P, p_eval, chi, lambda, Ac, bc, dp, 0);
    }
    for (int q = 0; q < P; ++q)
      saved_p[q] = p[q];
    int use_saved = 1;

    for (int iteration = 1; iteration <= cfg->max_iters + 1; ++iteration) {
      if (iteration > cfg->max_iters || lambda >= max_lambda) {
        error_status = 1;
        error_code = LKO_ERR_MAX_ITERS;
        break;
      } else {
        e->reached_iterations = iteration;
      }
      if (use_saved) {
        for (int q = 0; q < P; ++q)
          tentative_p[q] = saved_p[q];
      } else {
        for (int q = 0; q < P; ++q)
          p[q] = last_good_p[q];
        memcpy(p_eval, p, sizeof(float) * P);
        error_status = evaluate(&ic, cfg->model, P, uimg, ucols, sxy, n, cx, cy, p, A, b, &chi);
        chi *= scaling;
        if (error_status) {
          error_code = LKO_ERR_INTERP_OUT_OF_IMAGE;
          trace_push(trace, trace_cap, &tcount, level, 1, iteration, P, p_eval, chi, lambda, A, b, NULL, 1);
          break;
        }
        float Ac[36], bc[6];
        memcpy(Ac, A, sizeof(Ac));
        memcpy(bc, b, sizeof(bc));
        lko_damped_solve(P, A, b, lambda, scaling, dp);
        for (int q = 0; q < P; ++q)
          p[q] += dp[q];
        trace_push(trace, trace_cap, &tcount, level, 1, iteration, P, p_eval, chi, lambda, Ac, bc, dp, 0);
        for (int q = 0; q < P; ++q)
          tentative_p[q] = p[q];
      }
      for (int q = 0; q < P; ++q)
        p[q] = tentative_p[q];
      memcpy(p_eval, p, sizeof(float) * P);
      error_status = evaluate(&ic, cfg->model, P, uimg, ucols, sxy, n, cx, cy, p, A, b, &chi);
      chi *= scaling;
      if (error_status) {
        error_code = LKO_ERR_INTERP_OUT_OF_IMAGE;
        trace_push(trace, trace_cap, &tcount, level, 2, iteration, P, p_eval, chi, lambda, A, b, NULL, 1);
        break;
      }
      {
        float la = lambda * 0.4f;
        if (!(la > min_lambda)) /* std::max(lambda*0.4f, min_lambda) */
          la = min_lambda;
        float Ac[36], bc[6];
        memcpy(Ac, A, sizeof(Ac));
        memcpy(bc, b, sizeof(bc));
        lko_damped_solve(P, A, b, la, scaling, dp);
        for (int q = 0; q < P; ++q)
          p[q] += dp[q];
        trace_push(trace, trace_cap, &tcount, level, 2, iteration, P, p_eval, chi, la, Ac, bc, dp, 0);
      }
      for (int q = 0; q < P; ++q)
        saved_p[q] = p[q];

      float lg = e->last_good_chi;
      float mx = lg < chi ? chi : lg; /* std::max(last_good_chi, chi) */
      float delta_chi = fabsf((lg - chi) / (mx + cfg->precision));

      if (chi <= lg) {
        e->last_good_chi = chi;
        float la = lambda * 0.4f;
        lambda = la > min_lambda ? la : min_lambda;
        for (int q = 0; q < P; ++q)
          last_good_p[q] = tentative_p[q];
        use_saved = 1;
      } else {
        float la = lambda * 10.0f;
        lambda = la < max_lambda ? la : max_lambda;
        use_saved = 0;
      }
      if (delta_chi < cfg->precision)
        break;
    }
    level_old = level;
  }
  if (!early_return)
    lko_translate_parameters(P, p, level_old, 0);

  for (int l = first; l <= cfg->py_stop; l += cfg->py_step)
    free(lxy[l]);

  if (out) {
    memset(out, 0, sizeof(*out));
    for (int q = 0; q < P; ++q)
      out->p[q] = p[q];
    out->chi = e->last_good_chi;             /* correlation_class.cpp:848 */
    out->n_points = n0;                      /* :850-852 */
    out->iterations = e->reached_iterations; /* :870, last level that ran a trip */
    out->error_code = error_status ? error_code : LKO_ERR_NONE;
    out->und_cx = cx0;
    out->und_cy = cy0;
  }
  if (n_trace)
    *n_trace = tcount;
  return error_status ? error_code : LKO_ERR_NONE;
}

int lko_correlate_sectors(lko_engine *e, int S, const int64_t *off, const int *cnt,
                          const float *xy, int use_center, const float *centers,
                          const float *guesses, lko_result *results, int nthreads) {
  if (nthreads <= 1) {
    for (int s = 0; s < S; ++s) {
      float p[6];
      memcpy(p, guesses + 6 * (size_t)s, sizeof(p));
      lko_newton_raphson(e, p, cnt[s], xy + 2 * off[s], use_center,
                         use_center ? centers[2 * s] : 0.f, use_center ? centers[2 * s + 1] : 0.f,
                         &results[s], NULL, 0, NULL);
    }
    return 0;
  }
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
  {
    lko_engine local = *e; /* shares the read-only pyramids */
    local.cfg.cache_mode = 0;
#pragma omp for schedule(dynamic, 16)
    for (int s = 0; s < S; ++s) {
      float p[6];
      memcpy(p, guesses + 6 * (size_t)s, sizeof(p));
      local.reached_iterations = 0;
      lko_newton_raphson(&local, p, cnt[s], xy + 2 * off[s], use_center,
                         use_center ? centers[2 * s] : 0.f, use_center ? centers[2 * s + 1] : 0.f,
                         &results[s], NULL, 0, NULL);
    }
  }
  return 0;
#else
  (void)nthreads;
  return lko_correlate_sectors(e, S, off, cnt, xy, use_center, centers, guesses, results, 1);
#endif
}

/* ------------------------------------------------------------------------------------ */
/* ROI -> sample lists                                                                  */
/* ------------------------------------------------------------------------------------ */

/* manager_class.cpp:276-310.  (fabs on a float: evaluated in float here.) */
void lko_rect_sector_geometry(float x_begin, float y_begin, float x_end, float y_end, int hs,
                              int vs, int *xdim_out, int *ydim_out, int *centers) {
  int x1 = (int)x_end, x0 = (int)x_begin, y1 = (int)y_end, y0 = (int)y_begin;
  int xdim = (abs(x1 - x0) / hs - 1) / 2;
  int ydim = (abs(y1 - y0) / vs - 1) / 2;
  float fx1 = x_end, fx0 = x_begin, fhs = (float)hs;
  float fy1 = y_end, fy0 = y_begin, fvs = (float)vs;
  float fxdim = (fabsf(fx1 - fx0) / fhs - 1.f) / 2.f;
  float fydim = (fabsf(fy1 - fy0) / fvs - 1.f) / 2.f;
  for (int i = 0; i < hs; ++i) {
    int center_x = (int)(0.5f + fx0 + fxdim + (2.f * fxdim + 1.f) * (float)i);
    for (int j = 0; j < vs; ++j) {
      int center_y = (int)(0.5f + fy0 + fydim + (2.f * fydim + 1.f) * (float)j);
      centers[2 * (i * vs + j)] = center_x;
      centers[2 * (i * vs + j) + 1] = center_y;
    }
  }
  *xdim_out = xdim;
  *ydim_out = ydim;
}

/* manager_class.cpp:1596-1614 */
int lko_rect_points(int x0, int y0, int x1, int y1, float *xy, int cap) {
  int m = 0;
  for (int ix = x0; ix <= x1; ++ix)
    for (int iy = y0; iy <= y1; ++iy) {
      if (xy && m < cap) {
        xy[2 * m] = (float)ix;
        xy[2 * m + 1] = (float)iy;
      }
      ++m;
    }
  return m;
}

static float fmin2(float a, float b) { return b < a ? b : a; } /* std::min */
static float fmax2(float a, float b) { return a < b ? b : a; } /* std::max */

/* manager_class.cpp:816-940, single OpenMP thread (the only deterministic order) */
int64_t lko_annular_points(float r, float dr, float a, float da, float cx, float cy, int as,
                           float *xy, int64_t cap) {
  int x0, y0, x1, y1;
  float c00x = 0, c01x = 0, c10x = 0, c11x = 0, c00y = 0, c01y = 0, c10y = 0, c11y = 0;
  if (as <= 0)
    return -1;
  if (as == 1) {
    x0 = (int)(cx - (r + dr));
    x1 = (int)(cx + (r + dr));
    y0 = (int)(cy - (r + dr));
    y1 = (int)(cy + (r + dr));
  } else {
    float sin0 = (float)sin(a), cos0 = (float)cos(a);
    float sin1 = (float)sin(a + da), cos1 = (float)cos(a + da);
    float sin2 = (float)sin(a + da / 2.f), cos2 = (float)cos(a + da / 2.f);
    c00x = cx + (r)*cos0;
    c01x = cx + (r)*cos1;
    c10x = cx + (r + dr) * cos0 * 1.2f;
    c11x = cx + (r + dr) * cos1 * 1.2f;
    c00y = cy + (r)*sin0;
    c01y = cy + (r)*sin1;
    c10y = cy + (r + dr) * sin0 * 1.2f;
    c11y = cy + (r + dr) * sin1 * 1.2f;
    float arc_x = cx + (r + dr) * cos2, arc_y = cy + (r + dr) * sin2;
    x0 = (int)fmin2(arc_x, fmin2(fmin2(c00x, c01x), fmin2(c10x, c11x)));
    x1 = (int)fmax2(arc_x, fmax2(fmax2(c00x, c01x), fmax2(c10x, c11x)));
    y0 = (int)fmin2(arc_y, fmin2(fmin2(c00y, c01y), fmin2(c10y, c11y)));
    y1 = (int)fmax2(arc_y, fmax2(fmax2(c00y, c01y), fmax2(c10y, c11y)));
  }
  float ro2 = (r + dr) * (r + dr), ri2 = r * r;
  int64_t m = 0;
  for (float i = (float)x0; i < x1; ++i) {
    for (int j = y0; j < y1; ++j) {
      float r2 = (i - cx) * (i - cx) + (j - cy) * (j - cy);
      if (r2 > ri2 && r2 < ro2) {
        float cross1 = (c11x - i) * (c01y - c11y) - (c11y - j) * (c01x - c11x);
        float cross2 = (c00x - i) * (c10y - c00y) - (c00y - j) * (c10x - c00x);
        if (cross1 * cross2 > 0 || as == 1) {
          if (xy && m < cap) {
            xy[2 * m] = i;
            xy[2 * m + 1] = (float)j;
          }
          ++m;
        }
      }
    }
  }
  return m;
}

/* ---- blob polygon: polygon_class.cpp, doubly linked ring restated with indices ---- */
typedef struct {
  float x, y;
  int ear, next, prev;
} pvert;
typedef struct {
  pvert *v;
  int head, n;
} poly;

static float area2(const poly *P, int a, int b, int c) { /* polygon_class.cpp:55-63 */
  return (P->v[b].x - P->v[a].x) * (P->v[c].y - P->v[a].y) -
         (P->v[c].x - P->v[a].x) * (P->v[b].y - P->v[a].y);
}
static int p_left(const poly *P, int a, int b, int c) { return area2(P, a, b, c) > 0.f; }
static int p_lefton(const poly *P, int a, int b, int c) { return area2(P, a, b, c) >= 0.f; }
static int p_coll(const poly *P, int a, int b, int c) { return area2(P, a, b, c) == 0.f; }
static int p_isectprop(const poly *P, int a, int b, int c, int d) { /* :112-122 */
  if (p_coll(P, a, b, c) || p_coll(P, a, b, d) || p_coll(P, b, d, a) || p_coll(P, c, d, b))
    return 0;
  return (!p_left(P, a, b, c) ^ !p_left(P, a, b, d)) && (!p_left(P, c, d, a) ^ !p_left(P, c, d, b));
}
static int p_between(const poly *P, int a, int b, int c) { /* :124-143 */
  if (!p_coll(P, a, b, c))
    return 0;
  const pvert *v = P->v;
  if (v[a].x != v[b].x)
    return ((v[a].x <= v[c].x) && (v[c].x <= v[b].x)) || ((v[a].x >= v[c].x) && (v[c].x >= v[b].x));
  return ((v[a].y <= v[c].y) && (v[c].y <= v[b].y)) || ((v[a].y >= v[c].y) && (v[c].y >= v[b].y));
}
static int p_intersect(const poly *P, int a, int b, int c, int d) { /* :145-156 */
  if (p_isectprop(P, a, b, c, d))
    return 1;
  return p_between(P, a, b, c) || p_between(P, a, b, d) || p_between(P, c, d, a) ||
         p_between(P, c, d, b);
}
static int p_diagonalie(const poly *P, int a, int b) { /* :158-176 */
  int c = P->head;
  do {
    int c1 = P->v[c].next;
    if (c != a && c1 != a && c != b && c1 != b && p_intersect(P, a, b, c, c1))
      return 0;
    c = P->v[c].next;
  } while (c != P->head);
  return 1;
}
static int p_incone(const poly *P, int a, int b) { /* :178-190 */
  int a1 = P->v[a].next, a0 = P->v[a].prev;
  if (p_lefton(P, a, a1, a0))
    return p_left(P, a, b, a0) && p_left(P, b, a, a1);
  return !(p_lefton(P, a, b, a1) && p_lefton(P, b, a, a0));
}
static int p_diagonal(const poly *P, int a, int b) { /* :192-194 */
  return p_incone(P, a, b) && p_incone(P, b, a) && p_diagonalie(P, a, b);
}
static float p_area_poly2(const poly *P) { /* :74-86 */
  float sum = 0.f;
  int a = P->v[P->head].next;
  do {
    sum += area2(P, P->head, a, P->v[a].next);
    a = P->v[a].next;
  } while (P->v[a].next != P->head);
  return sum;
}
static int p_simple_loop(const poly *P) { /* :198-222 */
  if (P->n < 4)
    return 1;
  int outerLeft = P->head, outerRight;
  do {
    outerRight = P->v[outerLeft].next;
    int innerLeft = P->v[outerRight].next, innerRight;
    do {
      innerRight = P->v[innerLeft].next;
      if (p_intersect(P, outerLeft, outerRight, innerLeft, innerRight))
        return 0;
      innerLeft = innerRight;
    } while (innerLeft != P->head && innerLeft != P->v[outerLeft].prev);
    outerLeft = outerRight;
  } while (outerLeft != P->v[P->v[P->head].prev].prev);
  return 1;
}

typedef struct {
  float *xy;
  int64_t cap, m;
} ptsink;

static int p_line(float x1, float y1, float x2, float y2, float *dxdy, float *x0) { /* :405-416 */
  float den = y2 - y1;
  if (den != 0) {
    *dxdy = (x2 - x1) / den;
    *x0 = x1 - *dxdy * y1;
    return 0;
  }
  return 1;
}

/* :349-403; v1,v2 share y */
static void p_flat(ptsink *s, float x1, float y1, float x2, float y2, float x3, float y3) {
  (void)y2;
  int dy = (int)(floor(y3) - floor(y1));
  int dx = (int)(floor(x2) - floor(x1));
  if (dx == 0 || dy == 0)
    return;
  float sx, sy, bx, by;
  if (dx > 0) {
    sx = x1; sy = y1; bx = x2; by = y2;
  } else {
    sx = x2; sy = y2; bx = x1; by = y1;
  }
  float dS = 0, dB = 0, oS = 0, oB = 0;
  p_line(sx, sy, x3, y3, &dS, &oS);
  p_line(bx, by, x3, y3, &dB, &oB);
  int jInit, jEnd;
  if (dy > 0) {
    jInit = (int)ceil(y1);
    jEnd = (int)ceil(y3);
  } else {
    jInit = (int)ceil(y3);
    jEnd = (int)ceil(y1);
  }
  for (int j = jInit; j < jEnd; ++j) {
    int iInit = (int)ceilf(dS * (float)j + oS);
    int iEnd = (int)ceilf(dB * (float)j + oB);
    for (int i = iInit; i < iEnd; ++i) {
      if (s->xy && s->m < s->cap) {
        s->xy[2 * s->m] = (float)i;
        s->xy[2 * s->m + 1] = (float)j;
      }
      ++s->m;
    }
  }
}

/* :283-347 (the three early flat-triangle calls at :285-298 discard their result) */
static void p_triangle(ptsink *s, const float *t) {
  float x[3] = {t[0], t[2], t[4]}, y[3] = {t[1], t[3], t[5]};
  int ymax, ymid, ymin;
  if (y[0] > y[1]) {
    if (y[1] > y[2]) { ymax = 0; ymid = 1; ymin = 2; }
    else if (y[2] > y[0]) { ymax = 2; ymid = 0; ymin = 1; }
    else { ymax = 0; ymid = 2; ymin = 1; }
  } else {
    if (y[0] > y[2]) { ymax = 1; ymid = 0; ymin = 2; }
    else if (y[2] > y[1]) { ymax = 2; ymid = 1; ymin = 0; }
    else { ymax = 1; ymid = 2; ymin = 0; }
  }
  float dxdy, x0;
  if (p_line(x[ymin], y[ymin], x[ymax], y[ymax], &dxdy, &x0))
    return;
  float newY = y[ymid], newX = dxdy * newY + x0;
  p_flat(s, x[ymid], y[ymid], newX, newY, x[ymax], y[ymax]);
  p_flat(s, x[ymid], y[ymid], newX, newY, x[ymin], y[ymin]);
}

int64_t lko_blob_points(const float *contour_xy, int nv, float *xy, int64_t cap) {
  if (nv < 3)
    return -1;
  poly P;
  P.v = (pvert *)malloc(sizeof(pvert) * (size_t)nv);
  P.n = nv;
  P.head = 0;
  for (int i = 0; i < nv; ++i) { /* vectorToVertex + add, :65-72, :3-14 */
    P.v[i].x = contour_xy[2 * i];
    P.v[i].y = contour_xy[2 * i + 1];
    P.v[i].ear = 0;
    P.v[i].next = (i + 1) % nv;
    P.v[i].prev = (i + nv - 1) % nv;
  }
  if (!p_simple_loop(&P)) { /* triangulate :224-229 */
    free(P.v);
    return -1;
  }
  if (p_area_poly2(&P) < 0) { /* reOrientPoly :88-102 */
    for (int i = 0; i < nv; ++i) {
      int t = P.v[i].prev;
      P.v[i].prev = P.v[i].next;
      P.v[i].next = t;
    }
  }
  { /* earInit :38-53 */
    int v1 = P.head;
    do {
      int v2 = P.v[v1].next, v0 = P.v[v1].prev;
      P.v[v1].ear = p_diagonal(&P, v0, v2);
      v1 = P.v[v1].next;
    } while (v1 != P.head);
  }
  float *tri = (float *)malloc(sizeof(float) * 6 * (size_t)(nv > 2 ? nv - 2 : 1));
  int ntri = 0;
  while (P.n > 3) { /* :241-275 */
    int v2 = P.head, found = 0;
    do {
      if (P.v[v2].ear) {
        int v3 = P.v[v2].next, v4 = P.v[v3].next, v1 = P.v[v2].prev, v0 = P.v[v1].prev;
        float *t = &tri[6 * ntri++];
        t[0] = P.v[v1].x; t[1] = P.v[v1].y;
        t[2] = P.v[v2].x; t[3] = P.v[v2].y;
        t[4] = P.v[v3].x; t[5] = P.v[v3].y;
        P.v[v1].ear = p_diagonal(&P, v0, v3);
        P.v[v3].ear = p_diagonal(&P, v1, v4);
        P.v[v1].next = v3;
        P.v[v3].prev = v1;
        P.head = v3;
        P.n--;
        found = 1;
        break;
      }
      v2 = P.v[v2].next;
    } while (v2 != P.head);
    if (!found) { /* the reference would spin forever here */
      free(tri);
      free(P.v);
      return -1;
    }
  }
  {
    int v2 = P.head, v1 = P.v[v2].prev, v3 = P.v[v2].next;
    float *t = &tri[6 * ntri++];
    t[0] = P.v[v1].x; t[1] = P.v[v1].y;
    t[2] = P.v[v2].x; t[3] = P.v[v2].y;
    t[4] = P.v[v3].x; t[5] = P.v[v3].y;
  }
  ptsink s = {xy, cap, 0};
  for (int k = 0; k < ntri; ++k) /* getInsidePoints :418-429 */
    p_triangle(&s, &tri[6 * k]);
  free(tri);
  free(P.v);
  return s.m;
}

/* manager_class.cpp:2602-2707 for one sector; all arrays have 6 slots */
void lko_adjust_initial_guess(int model, int frame, int constant_velocity,
                              const float *global_guess, float sector_cx, float sector_cy,
                              float global_cx, float global_cy, const float *resulting,
                              float *previous_resulting, float *guess_out) {
  int P = lko_n_params(model);
  if (frame == 0) {
    for (int i = 0; i < P; ++i)
      guess_out[i] = global_guess[i];
    float dx = sector_cx - global_cx, dy = sector_cy - global_cy;
    if (model == LKO_FM_UVUXUYVXVY) {
      float Ux = global_guess[2], Uy = global_guess[3], Vx = global_guess[4], Vy = global_guess[5];
      guess_out[0] += dx * Ux + dy * Uy;
      guess_out[1] += dx * Vx + dy * Vy;
    } else {
      float Vx = global_guess[2];
      guess_out[0] += -dy * Vx;
      guess_out[1] += dx * Vx;
    }
    for (int i = 0; i < P; ++i)
      previous_resulting[i] = guess_out[i];
  } else {
    for (int i = 0; i < P; ++i)
      guess_out[i] = constant_velocity ? resulting[i] + (resulting[i] - previous_resulting[i])
                                       : resulting[i];
    for (int i = 0; i < P; ++i)
      previous_resulting[i] = resulting[i];
  }
}
