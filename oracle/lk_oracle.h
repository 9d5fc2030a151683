/* lk_oracle.h - CPU ORACLE for the Lucas-Kanade correlation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's CPU
 * algorithm (namascar/correlation: correlation_class.cpp, interpolation_class.cpp,
 * model_class.cpp, pyramid_class.cpp, polygon_class.cpp and the pure functions of
 * manager_class.cpp that define the hot path's inputs).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (correlation_amd/, include/lk_engine.h) never does.
 *
 * PARITY STATUS: "parity unpinned" for bicubic sampling, pyramid, LM driver and the
 * 6x6 solve - the reference ships no tests / golden vectors, and its CPU path cannot
 * be compiled in this image without writing stand-ins for OpenCV and Eigen (absent).
 * Pinned pieces: the warp model (model_class.cpp) and the blob polygon rasteriser
 * (polygon_class.cpp, parameters.cpp) are checked against the reference's own objects
 * compiled from /root/reference by oracle/Makefile into oracle/_ref/ (no third-party
 * code needed), and the 16x16 bicubic matrix is checked against the reference's
 * literal table (read as text).  See DESIGN.md section "Oracle".
 *
 * All arithmetic is float32 in the reference's operation order; build with
 * -ffp-contract=off (oracle/Makefile) so no FMA contraction changes the roundings.
 */
#ifndef LK_ORACLE_H
#define LK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LKO_MAX_LEVELS 8
#define LKO_MAX_PARAMS 6

/* enums.hpp:10-35 */
enum { LKO_IM_NEAREST = 0, LKO_IM_BILINEAR = 1, LKO_IM_BICUBIC = 2 };
enum { LKO_FM_U = 0, LKO_FM_UV = 1, LKO_FM_UVQ = 2, LKO_FM_UVUXUYVXVY = 3 };
enum {
  LKO_ERR_NONE = 0,
  LKO_ERR_MODEL_OUT_OF_IMAGE = 1,
  LKO_ERR_INTERP_OUT_OF_IMAGE = 2,
  LKO_ERR_MAX_ITERS = 3,
  LKO_ERR_BAD_DOMAIN = 4
};
enum { LKO_IMG_UND = 0, LKO_IMG_DEF = 1, LKO_IMG_NXT = 2 };

typedef struct {
  int interp;       /* LKO_IM_*   (CorrelationClass ctor, correlation_class.hpp:132-137) */
  int model;        /* LKO_FM_* */
  float precision;  /* required_precision */
  int max_iters;    /* maximum_iterations */
  int py_start, py_step, py_stop;
  int n_threads;    /* number_of_threads of the reference (correlation_class.cpp:169-186,
                       :253-275): samples are split into this many contiguous chunks, each
                       summed from zero, chunk totals added in thread order.  0/1 = one
                       chunk.  (The reference really spawns threads; the arithmetic is the
                       same, only this split matters for the bits.) */
  int solver;       /* 0: the reference's solver, Eigen ColPivHouseholderQR restated (default).
                       Yardsticks only (never the parity target): 1 = float32 root-free
                       Cholesky, 2 = float64 Gaussian elimination ("exact" solve of the same
                       float32 system).  They measure how far ANY other backward-stable
                       solver lands from the QR's float32 rounding. */
  int cache_mode;   /* 0: coefficients computed per use (values identical to the
                       reference's lazy cache as long as no out-of-image error has
                       occurred on this def image); 1: emulate the lazy per-pixel
                       cache including its poisoning after an error
                       (interpolation_class.cpp:228-250) */
} lko_config;

/* layout-identical to CorrelationResult (domains.hpp:110-118), 48 bytes */
typedef struct {
  float p[6];
  float chi;
  int n_points;
  int iterations;
  int error_code;
  float und_cx, und_cy;
} lko_result;

/* one record per evaluation+solve, for golden traces */
typedef struct {
  int level;
  int kind;          /* 0 = evaluation #0, 1 = reject-path re-evaluation, 2 = tentative */
  int iteration;     /* LM trip (0 for evaluation #0) */
  float p_in[6];     /* parameters the evaluation ran at */
  float chi;         /* scaled chi */
  float lambda;      /* damping used by the solve that followed */
  float A[36];       /* raw sums (upper triangle valid), before scaling */
  float b[6];
  float dp[6];
  int error;
} lko_trace_rec;

typedef struct lko_engine lko_engine;

int lko_n_params(int model);                                  /* model_class.cpp:216-231 */

lko_engine *lko_create(const lko_config *cfg);
void lko_destroy(lko_engine *e);
/* copies the level-0 pixels (step == cols) and builds levels 1..py_stop
 * (pyramid_class.cpp:137-209) */
int lko_set_image(lko_engine *e, int which, const uint8_t *px, int rows, int cols);
void lko_und_from_def(lko_engine *e);                         /* pyramid_class.cpp:211-226 */
void lko_def_from_nxt(lko_engine *e);                         /* pyramid_class.cpp:228-258 */
const uint8_t *lko_get_level(lko_engine *e, int which, int level, int *rows, int *cols);

/* CorrelationClass::Newton_Raphson (correlation_class.cpp:306-640).
 * p is in/out (level-0 parameters).  use_center=0 -> centre = float mean of samples
 * (pyramid_class.cpp:325-340).  trace may be NULL; at most trace_cap records written,
 * *n_trace receives the number produced. */
int lko_newton_raphson(lko_engine *e, float *p, int n0, const float *xy,
                       int use_center, float cx, float cy, lko_result *out,
                       lko_trace_rec *trace, int trace_cap, int *n_trace);

/* batch of sectors, processed sequentially like manager_class.cpp:304-307 does.
 * xy: concatenated AoS sample lists; off[s]..off[s]+cnt[s]; centers[2s],[2s+1]
 * (ignored if use_center==0); guesses [S][6] in, results [S] out.
 * nthreads>1: OpenMP across sectors with per-thread engine clones (cache_mode 0 only,
 * stale-state quirks not reproduced). */
int lko_correlate_sectors(lko_engine *e, int S, const int64_t *off, const int *cnt,
                          const float *xy, int use_center, const float *centers,
                          const float *guesses, lko_result *results, int nthreads);

/* ---- stand-alone pieces for known-answer tests ---- */
/* one pyramid level: dst (rows/2 x cols/2, zero border) from src (pyramid_class.cpp:83-122) */
void lko_pyramid_level(const uint8_t *src, int rows, int cols, uint8_t *dst);
/* 16 bicubic coefficients of def pixel (ix,iy) (interpolation_class.cpp:243-336) */
void lko_bicubic_coeffs(const uint8_t *img, int step, int ix, int iy, float a[16]);
/* sample value + gradient; returns 0 ok / 1 out of image (interpolation_class.cpp:79-226) */
int lko_interpolate(int interp, const uint8_t *img, int rows, int cols, float x, float y,
                    float *w, float *wx, float *wy);
/* warp of one sample (model_class.cpp:48-202): def position and dT/dp rows */
void lko_model_point(int model, float x, float y, float cx, float cy, const float *p,
                     float *xd, float *yd, float dTx[6], float dTy[6]);
/* one evaluation on explicit data (interpolation_class.cpp:671-764); A upper-valid */
int lko_evaluate(int interp, int model, const uint8_t *und, int urows, int ucols,
                 const uint8_t *def, int drows, int dcols, const float *xy, int n,
                 float cx, float cy, const float *p, float A[36], float b[6], float *chi);
/* scale, mirror, damp and solve: compute_model_parameters + solve
 * (correlation_class.cpp:642-768); Eigen 3.4.0 ColPivHouseholderQR restated.
 * A,b are modified in place like the reference does; dp receives the step. */
void lko_damped_solve(int n, float *A, float *b, float lambda, float scaling, float *dp);
/* plain  x = A.colPivHouseholderQr().solve(b)  on a column-major n x n float matrix */
void lko_colpiv_qr_solve(int n, const float *A, const float *b, float *x);
/* per-level decimation (pyramid_class.cpp:289-323); returns count written to out */
int lko_decimate(const float *xy_prev, int n_prev, int level_delta, float *out);
/* parameter rescale between levels (pyramid_class.cpp:260-287) */
void lko_translate_parameters(int n_params, float *p, int level_src, int level_dst);

/* ---- ROI -> sample lists (manager_class.cpp / polygon_class.cpp) ---- */
/* sector geometry of a rectangular domain (manager_class.cpp:276-310): returns half
 * sizes and writes hs*vs centres (cx,cy ints) in iSector = i*vs+j order */
void lko_rect_sector_geometry(float x_begin, float y_begin, float x_end, float y_end,
                              int hs, int vs, int *xdim, int *ydim, int *centers);
/* manager_class.cpp:1596-1614: x outer, y inner, inclusive; returns count */
int lko_rect_points(int x0, int y0, int x1, int y1, float *xy, int cap);
/* manager_class.cpp:816-940 with one OpenMP thread; returns count (may exceed cap) */
int64_t lko_annular_points(float r, float dr, float a, float da, float cx, float cy,
                           int as, float *xy, int64_t cap);
/* polygonBlob_class (polygon_class.cpp:224-429): returns count, or -1 on a
 * self-intersecting contour (error_bad_domain) */
int64_t lko_blob_points(const float *contour_xy, int n_vertices, float *xy, int64_t cap);
/* managerClass::adjust_initial_guess (manager_class.cpp:2602-2707), one sector.
 * frame 0: guess = global + strain*(centre-global centre); later frames:
 * constant_velocity ? 2*p_prev - p_prevprev : p_prev.  Updates p_prevprev. */
void lko_adjust_initial_guess(int model, int frame, int constant_velocity,
                              const float *global_guess, float sector_cx, float sector_cy,
                              float global_cx, float global_cy, const float *resulting,
                              float *previous_resulting, float *guess_out);

#ifdef __cplusplus
}
#endif
#endif
