// tracker_driver.cpp - TEST INFRASTRUCTURE: exercises the HOST-ONLY product code - csrc/lk_tracker.cpp
// (tracking bookkeeping, report writer, PGM reader, frame loop with its helper threads) and
// csrc/lk_roi.hpp (ROI -> sample lists) - so that it can run under AddressSanitizer/UBSan and under
// ThreadSanitizer in the build container (the reference has no sanitizer
// coverage; its known races are listed there).  The engine behind lk_sequence_run is the CPU mock.
//   tracker_driver <scratch dir>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lk_tracker.h"

#define CHECK(x)                                                                   \
  do {                                                                             \
    if (!(x)) {                                                                    \
      std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #x);   \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

struct Frames {
  int rows = 64, cols = 64;
  std::vector<std::vector<uint8_t>> px;
  std::vector<std::string> names;
};
static const uint8_t *provide(void *user, int index, int *rows, int *cols, int *step, const char **name) {
  Frames *f = static_cast<Frames *>(user);
  *rows = f->rows, *cols = f->cols, *step = f->cols;
  *name = f->names[(size_t)index].c_str();
  return f->px[(size_t)index].data();
}

static void roi_functions() {
  int xd = 0, yd = 0;
  std::vector<int> cen(2 * 7 * 5);
  CHECK(lk_roi_rect_grid(3.5f, 7.25f, 490.f, 300.5f, 7, 5, &xd, &yd, cen.data()) == 0 && xd > 0 && yd > 0);
  const int64_t n = lk_roi_annular_points(40.f, 30.f, 0.3f, 0.9f, 100.f, 90.f, 4, nullptr, 0);
  CHECK(n > 100);
  std::vector<float> xy(2 * (size_t)n);
  CHECK(lk_roi_annular_points(40.f, 30.f, 0.3f, 0.9f, 100.f, 90.f, 4, xy.data(), n) == n);
  std::vector<float> half(2 * (size_t)n / 2);   // a too-small buffer must not be overrun
  CHECK(lk_roi_annular_points(40.f, 30.f, 0.3f, 0.9f, 100.f, 90.f, 4, half.data(), n / 2) == n);
  std::vector<float> dec(2 * (size_t)n);
  CHECK(lk_roi_decimate(xy.data(), (int)n, 1, dec.data()) <= (int)n);
  std::vector<float> star;
  for (int i = 0; i < 16; ++i) {
    const float r = i % 2 ? 20.f : 45.f, t = 6.2831853f * (float)i / 16.f;
    star.push_back(60.f + r * std::cos(t)), star.push_back(60.f + r * std::sin(t));
  }
  const int64_t nb = lk_roi_blob_points(star.data(), 16, nullptr, 0);
  CHECK(nb > 500);
  std::vector<float> bxy(2 * (size_t)nb);
  CHECK(lk_roi_blob_points(star.data(), 16, bxy.data(), nb) == nb);
  const float bow[8] = {0, 0, 10, 10, 10, 0, 0, 10};   // self-intersecting
  CHECK(lk_roi_blob_points(bow, 4, nullptr, 0) == -1);
}

static void pgm(const std::string &dir) {
  const std::string p = dir + "/t.pgm";
  FILE *f = std::fopen(p.c_str(), "wb");
  CHECK(f);
  std::fprintf(f, "P5\n# comment\n5 3\n255\n");
  for (int i = 0; i < 15; ++i)
    std::fputc(i * 10, f);
  std::fclose(f);
  uint8_t *px = nullptr;
  int rows = 0, cols = 0;
  CHECK(lk_load_pgm(p.c_str(), &px, &rows, &cols) == 0 && rows == 3 && cols == 5 && px[14] == 140);
  lk_free_image(px);
  f = std::fopen(p.c_str(), "wb");
  std::fprintf(f, "P5\n5 3\n255\nshort");   // truncated
  std::fclose(f);
  px = nullptr;
  CHECK(lk_load_pgm(p.c_str(), &px, &rows, &cols) != 0);
  CHECK(lk_load_pgm((dir + "/missing.pgm").c_str(), &px, &rows, &cols) != 0);
}

static void bookkeeping() {
  std::vector<float> star;
  for (int i = 0; i < 12; ++i) {
    const float r = i % 2 ? 60.f : 110.f, t = 6.2831853f * (float)i / 12.f;
    star.push_back(256.f + r * std::cos(t)), star.push_back(256.f + r * std::sin(t));
  }
  for (int domain = 0; domain < 3; ++domain)
    for (int deformation = 0; deformation < 3; ++deformation)
      for (int reference = 0; reference < 2; ++reference)
        for (int model : {LK_FM_UV, LK_FM_UVUXUYVXVY}) {
          lk_tracker_config cfg{};
          cfg.fitting_model = model, cfg.domain_type = domain, cfg.deformation = deformation;
          cfg.reference_image = reference, cfg.error_mode = deformation == 0 ? LK_ERRMODE_STOP_FRAME : LK_ERRMODE_CONTINUE;
          cfg.global_guess[0] = 0.5f, cfg.global_guess[1] = -0.25f;
          lk_tracker *t = nullptr;
          CHECK(lk_tracker_create(&cfg, &t) == 0);
          if (domain == LK_DOMAIN_RECT)
            CHECK(lk_tracker_set_rect_domain(t, 24.f, 24.f, 487.f, 487.f, 255.5f, 255.5f, 9, 7) == 0);
          else if (domain == LK_DOMAIN_ANNULAR)
            CHECK(lk_tracker_set_annular_domain(t, 60.f, 200.f, 256.f, 250.f, 3, 8) == 0);
          else
            CHECK(lk_tracker_set_blob_domain(t, star.data(), 12, 256.f, 256.f) == 0);
          const int S = lk_tracker_sector_count(t);
          CHECK(S >= 1);
          std::vector<lk_sector_command> cmd((size_t)S);
          std::vector<float> g(6 * (size_t)S);
          std::vector<lk_result> res((size_t)S);
          std::vector<lk_frame_result> fr((size_t)S);
          for (int frame = 0; frame < 4; ++frame) {
            CHECK(lk_tracker_begin_frame(t, frame, cmd.data(), g.data()) == 0);
            for (int s = 0; s < S; ++s) {
              lk_result &r = res[(size_t)s];
              std::memset(&r, 0, sizeof r);
              for (int i = 0; i < 6; ++i)
                r.resultingParameters[i] = g[6 * (size_t)s + i] + (i < 2 ? 0.8f - 0.1f * (float)i : 1e-4f);
              r.chi = 2.f + (float)s;
              r.numberOfPoints = 100 + s;
              r.iterations = 2;
              r.errorCode = (frame == 2 && s == S / 2 && deformation == 0) ? LK_ERROR_INTERPOLATION_OUT_OF_IMAGE : 0;
              r.undCenterX = cmd[(size_t)s].use_center ? cmd[(size_t)s].center_x : 200.f + (float)s;
              r.undCenterY = cmd[(size_t)s].use_center ? cmd[(size_t)s].center_y : 210.f;
            }
            int first_unsolved = -1, stop = -1;
            CHECK(lk_tracker_end_frame(t, frame, "und.pgm", "def.pgm", res.data(), &first_unsolved, &stop) == 0);
            CHECK(first_unsolved >= 0 && first_unsolved <= S && (stop == 0 || stop == 1));
          }
          CHECK(lk_tracker_get_results(t, fr.data()) == 0);
          size_t need = 0;
          CHECK(lk_tracker_report(t, nullptr, 0, &need) == 0 && need > 100);
          std::vector<char> text(need);
          CHECK(lk_tracker_report(t, text.data(), need, &need) == 0 && std::strlen(text.data()) + 1 == need);
          lk_tracker_destroy(t);
        }
}

static void frame_loop() {
  Frames f;
  unsigned seed = 12345u;
  for (int i = 0; i < 6; ++i) {
    std::vector<uint8_t> p((size_t)f.rows * f.cols);
    for (auto &v : p)
      v = (uint8_t)((seed = seed * 1664525u + 1013904223u) >> 24);
    f.px.push_back(p);
    f.names.push_back("frame" + std::to_string(i) + ".pgm");
  }
  for (int deformation : {LK_DEF_EULERIAN, LK_DEF_LAGRANGIAN, LK_DEF_STRICT_LAGRANGIAN})
    for (int reference : {LK_REF_FIRST, LK_REF_PREVIOUS}) {
      lk_config ec{LK_IM_BICUBIC, LK_FM_UVUXUYVXVY, 0.001f, 50, 0, 1, 2, 0};
      lk_engine *e = nullptr;
      CHECK(lk_create(&ec, &e) == 0);
      lk_tracker_config cfg{};
      cfg.fitting_model = LK_FM_UVUXUYVXVY, cfg.domain_type = LK_DOMAIN_RECT, cfg.deformation = deformation;
      cfg.reference_image = reference, cfg.error_mode = LK_ERRMODE_CONTINUE;
      lk_tracker *t = nullptr;
      CHECK(lk_tracker_create(&cfg, &t) == 0);
      CHECK(lk_tracker_set_rect_domain(t, 8.f, 8.f, 55.f, 55.f, 31.5f, 31.5f, 12, 11) == 0);
      int pairs = 0;
      CHECK(lk_sequence_run(e, t, 6, provide, &f, &pairs) == 0 && pairs == 5);
      size_t need = 0;
      CHECK(lk_tracker_report(t, nullptr, 0, &need) == 0 && need > 1000);
      lk_tracker_destroy(t);
      lk_destroy(e);
    }
}

int main(int argc, char **argv) {
  CHECK(argc == 2);
  roi_functions();
  pgm(argv[1]);
  bookkeeping();
  frame_loop();
  std::puts("tracker_driver ok");
  return 0;
}
