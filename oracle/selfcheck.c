/* selfcheck.c - TEST INFRASTRUCTURE: runs every entry point of the CPU oracle (lk_oracle.c) on
 * small synthetic inputs, for `make -C oracle asan` (AddressSanitizer + UBSan build, SURVEY.md
 * section 5: the reference itself has no sanitizer coverage).  Checks only sanity properties;
 * the values are pinned elsewhere (tests/test_oracle_pins.py, tests/test_oracle_selfcheck.py). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lk_oracle.h"

#define CHECK(x)                                                                  \
  do {                                                                            \
    if (!(x)) {                                                                   \
      fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #x);       \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

static void smooth_image(uint8_t *px, int rows, int cols, float shift) {
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float xs = (float)x - shift, v = 128.f + 60.f * sinf(0.21f * xs) * cosf(0.17f * (float)y) + 40.f * sinf(0.05f * xs + 0.09f * (float)y);
      px[(size_t)y * cols + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

int main(void) {
  enum { R = 96, C = 112 };
  /* two guard rows: the reference's nearest / bilinear samplers read the pixel right of and below
   * the rounded position (interpolation_class.cpp:376-406), i.e. one row past the image for
   * positions in its last half pixel - the restatement keeps that read and every caller pads */
  static uint8_t und[(R + 2) * C], def[(R + 2) * C], lvl[(R / 2) * (C / 2)];
  smooth_image(und, R, C, 0.f);
  smooth_image(def, R, C, 1.25f);
  lko_pyramid_level(und, R, C, lvl);
  float a[16], w, wx, wy;
  lko_bicubic_coeffs(und, C, 10, 12, a);
  for (int interp = 0; interp < 3; ++interp) {
    CHECK(lko_interpolate(interp, und, R, C, 20.3f, 30.7f, &w, &wx, &wy) == 0);
    CHECK(lko_interpolate(interp, und, R, C, -3.f, 30.7f, &w, &wx, &wy) == 1);
    CHECK(lko_interpolate(interp, und, R, C, (float)C - 1.5f, (float)R - 1.5f, &w, &wx, &wy) == (interp == 2 ? 1 : 0));
  }
  static float xy[2 * 41 * 41], dec[2 * 41 * 41];
  const int n = lko_rect_points(30, 28, 70, 68, xy, 41 * 41);
  CHECK(n == 41 * 41 && lko_rect_points(30, 28, 70, 68, xy, 10) == 41 * 41);
  CHECK(lko_decimate(xy, n, 1, dec) > n / 5);
  for (int model = 0; model < 4; ++model)
    for (int interp = 0; interp < 3; ++interp)
      for (int threads = 1; threads <= 3; threads += 2) {
        lko_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.interp = interp, cfg.model = model, cfg.precision = 1e-3f, cfg.max_iters = 50;
        cfg.py_start = 0, cfg.py_step = 1, cfg.py_stop = 2, cfg.n_threads = threads, cfg.cache_mode = threads == 1;
        lko_engine *e = lko_create(&cfg);
        CHECK(e);
        CHECK(lko_set_image(e, 0, und, R, C) == 0 && lko_set_image(e, 1, def, R, C) == 0 && lko_set_image(e, 2, def, R, C) == 0);
        int rr, cc;
        CHECK(lko_get_level(e, 1, 2, &rr, &cc) && rr == R / 4 && cc == C / 4);
        float p[6] = {0, 0, 0, 0, 0, 0};
        lko_result out;
        lko_trace_rec trace[8];
        int nt = 0;
        lko_newton_raphson(e, p, n, xy, 1, 50.f, 48.f, &out, trace, 8, &nt);
        CHECK(out.n_points == n && nt > 0);
        if (interp == 2 && (model == 1 || model == 3))
          CHECK(out.error_code == 0 && fabsf(out.p[0] - 1.25f) < 0.2f);
        /* a batch with an out-of-image sector and a one-sample sector, OpenMP across sectors */
        int64_t off[3] = {0, 0, 0};
        int cnt[3] = {n, 1, n};
        float centers[6] = {50.f, 48.f, 30.f, 28.f, 50.f, 48.f}, guesses[18];
        memset(guesses, 0, sizeof guesses);
        guesses[12] = 200.f; /* sector 2 starts far outside the image */
        lko_result res[3];
        CHECK(lko_correlate_sectors(e, 3, off, cnt, xy, 1, centers, guesses, res, 2) == 0);
        CHECK(res[2].error_code == LKO_ERR_INTERP_OUT_OF_IMAGE);
        lko_und_from_def(e);
        lko_def_from_nxt(e);
        lko_destroy(e);
      }
  /* ROI code */
  int xd, yd, cen[2 * 6 * 4];
  lko_rect_sector_geometry(3.5f, 7.25f, 90.f, 80.5f, 6, 4, &xd, &yd, cen);
  CHECK(xd > 0 && yd > 0);
  const int64_t na = lko_annular_points(10.f, 25.f, 0.4f, 1.1f, 48.f, 50.f, 4, NULL, 0);
  CHECK(na > 50);
  float *axy = (float *)malloc(2 * (size_t)na * sizeof(float));
  CHECK(lko_annular_points(10.f, 25.f, 0.4f, 1.1f, 48.f, 50.f, 4, axy, na) == na);
  CHECK(lko_annular_points(10.f, 25.f, 0.4f, 1.1f, 48.f, 50.f, 4, axy, na / 2) == na);
  free(axy);
  float star[32];
  for (int i = 0; i < 16; ++i) {
    const float r = i % 2 ? 15.f : 35.f, t = 6.2831853f * (float)i / 16.f;
    star[2 * i] = 48.f + r * cosf(t), star[2 * i + 1] = 48.f + r * sinf(t);
  }
  const int64_t nb = lko_blob_points(star, 16, NULL, 0);
  CHECK(nb > 300);
  float *bxy = (float *)malloc(2 * (size_t)nb * sizeof(float));
  CHECK(lko_blob_points(star, 16, bxy, nb) == nb);
  free(bxy);
  const float bow[8] = {0, 0, 10, 10, 10, 0, 0, 10};
  CHECK(lko_blob_points(bow, 4, NULL, 0) == -1);
  /* solver, guesses */
  float A[36], b[6] = {1, 2, 3, 4, 5, 6}, dp[6], x[6];
  for (int i = 0; i < 36; ++i)
    A[i] = (i / 6 == i % 6) ? 4.f + (float)(i / 6) : 0.1f * (float)((i * 7) % 5);
  lko_colpiv_qr_solve(6, A, b, x);
  lko_damped_solve(6, A, b, 1e-4f, 0.01f, dp);
  memset(A, 0, sizeof A); /* singular */
  lko_colpiv_qr_solve(6, A, b, x);
  float prev[6] = {0, 0, 0, 0, 0, 0}, g[6], gg[6] = {0.5f, 0.25f, 0.001f, 0, 0, 0.002f}, r6[6] = {1, 1, 0, 0, 0, 0};
  for (int model = 0; model < 4; ++model)
    for (int frame = 0; frame < 3; ++frame)
      lko_adjust_initial_guess(model, frame, frame == 2, gg, 10.f, 12.f, 50.f, 48.f, r6, prev, g);
  float q[6] = {1, 2, 3, 4, 5, 6};
  lko_translate_parameters(6, q, 0, 2);
  CHECK(q[0] == 0.25f && q[2] == 3.f);
  puts("oracle selfcheck ok");
  return 0;
}
