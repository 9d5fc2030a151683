// track_sequence.cpp - a headless C++ consumer of the C ABI (include/lk_engine.h, lk_tracker.h):
// what MainApp::correlate() + managerClass::perform_multiframe_correlation do in the reference
// application (mainapp.cpp:699-960, manager_class.cpp:1296-1496), without Qt.
//
//   g++ -std=c++17 -O2 -Iinclude examples/track_sequence.cpp -Lcorrelation_amd -llk_engine
//       -Wl,-rpath,$PWD/correlation_amd -o track_sequence          (one command line)
//   ./track_sequence report.csv eulerian|lagrangian|strict  hs vs  frame0.tif frame1.tif [frame2.tif ...]
//
// Frames are whatever lk_load_image decodes (PNG, TIFF, BMP, PNM) - the 8-bit grey cv::imread(path, IMREAD_GRAYSCALE) would give.  The rectangular domain is the image minus a 24-pixel margin, split
// into hs x vs sectors; affine model, bicubic interpolation, pyramid levels 0/1/2, zero global guess.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lk_engine.h"
#include "lk_tracker.h"

namespace {
struct Frames {
  std::vector<std::string> paths;
  uint8_t *held[2] = {nullptr, nullptr}; // the provider contract: frame i stays valid until i + 2 is returned
  int rows = 0, cols = 0;
};

const uint8_t *provide(void *user, int index, int *rows, int *cols, int *step, const char **name) {
  Frames *f = static_cast<Frames *>(user);
  uint8_t *px = nullptr;
  int r = 0, c = 0;
  if (lk_load_image(f->paths[(size_t)index].c_str(), &px, &r, &c) != LK_ERROR_NONE)
    return nullptr;
  lk_free_image(f->held[index & 1]);
  f->held[index & 1] = px;
  *rows = r;
  *cols = c;
  *step = c;
  *name = f->paths[(size_t)index].c_str();
  return px;
}
} // namespace

int main(int argc, char **argv) {
  if (argc < 7) {
    std::fprintf(stderr, "usage: %s report.csv eulerian|lagrangian|strict hs vs frame0.pgm frame1.pgm [...]\n", argv[0]);
    return 2;
  }
  const std::string mode = argv[2];
  const int hs = std::atoi(argv[3]), vs = std::atoi(argv[4]);
  Frames frames;
  for (int i = 5; i < argc; ++i)
    frames.paths.push_back(argv[i]);
  uint8_t *probe = nullptr;
  if (lk_load_image(frames.paths[0].c_str(), &probe, &frames.rows, &frames.cols) != LK_ERROR_NONE) {
    std::fprintf(stderr, "cannot read %s\n", frames.paths[0].c_str());
    return 1;
  }
  lk_free_image(probe);

  lk_config ec{LK_IM_BICUBIC, LK_FM_UVUXUYVXVY, 1e-3f, 50, 0, 1, 2, 0};
  lk_engine *engine = nullptr;
  if (lk_create(&ec, &engine) != LK_ERROR_NONE) {
    std::fprintf(stderr, "no usable HIP device (there is no CPU fallback)\n");
    return 1;
  }
  lk_tracker_config tc{};
  tc.fitting_model = LK_FM_UVUXUYVXVY;
  tc.domain_type = LK_DOMAIN_RECT;
  tc.deformation = mode == "strict" ? LK_DEF_STRICT_LAGRANGIAN : mode == "lagrangian" ? LK_DEF_LAGRANGIAN : LK_DEF_EULERIAN;
  tc.reference_image = tc.deformation == LK_DEF_EULERIAN ? LK_REF_FIRST : LK_REF_PREVIOUS;
  tc.error_mode = LK_ERRMODE_CONTINUE;
  lk_tracker *tracker = nullptr;
  if (lk_tracker_create(&tc, &tracker) != LK_ERROR_NONE)
    return 1;
  const float x0 = 24.f, y0 = 24.f, x1 = (float)(frames.cols - 25), y1 = (float)(frames.rows - 25);
  int rc = lk_tracker_set_rect_domain(tracker, x0, y0, x1, y1, 0.5f * (x0 + x1), 0.5f * (y0 + y1), hs, vs);
  int pairs = 0;
  if (!rc)
    rc = lk_sequence_run(engine, tracker, (int)frames.paths.size(), provide, &frames, &pairs);
  if (rc) {
    std::fprintf(stderr, "error %d: %s / %s\n", rc, lk_last_error_string(engine), lk_tracker_last_error(tracker));
    return 1;
  }
  size_t need = 0;
  lk_tracker_report(tracker, nullptr, 0, &need);
  std::string csv(need, '\0');
  lk_tracker_report(tracker, &csv[0], need, &need);
  FILE *out = std::fopen(argv[1], "wb");
  if (!out)
    return 1;
  std::fwrite(csv.data(), 1, need - 1, out);
  std::fclose(out);
  lk_stats st{};
  lk_get_stats(engine, &st);
  std::printf("%d pairs, %d sectors per pair, last solve %.3f ms, report %zu bytes -> %s\n", pairs,
              lk_tracker_sector_count(tracker), st.solve_ms, need - 1, argv[1]);
  lk_free_image(frames.held[0]);
  lk_free_image(frames.held[1]);
  lk_tracker_destroy(tracker);
  lk_destroy(engine);
  return 0;
}
