// adapter_driver.cpp - TEST INFRASTRUCTURE: drives include/lk_cuda_class_adapter.hpp (HipCudaClass)
// in the reference manager's exact call order for a rectangular domain and two frames:
//   frame 0  for every sector: resetPolygon(iSector, x0, y0, x1, y1), correlate(iSector, guess, results)
//            (perform_single_frame_correlation_rectangular, manager_class.cpp:304-340, :449)
//   frame 1  makeDefPyramidFromNxt (:234); for every sector: updatePolygon(iSector, def_Lagrangian)
//            (cuda_class.cu:569), correlate with the previous result as the guess (:2688-2694)
// and writes the 2 * S CorrelationResult records.  Linked against the CPU mock
// (tests/host/lk_engine_mock.cpp, -DADAPTER_DRIVER_MOCK) in the build container, against
// liblk_engine.so on the GPU box.
//   adapter_driver und.raw def.raw nxt.raw rows cols x_begin x_end hs vs out.bin
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lk_cuda_class_adapter.hpp"

#ifdef ADAPTER_DRIVER_MOCK
extern "C" const char *lk_mock_journal(void);
extern "C" int lk_mock_commits(void);
#endif

static std::vector<uint8_t> slurp(const char *path, size_t n) {
  std::vector<uint8_t> v(n);
  FILE *f = std::fopen(path, "rb");
  if (!f || std::fread(v.data(), 1, n, f) != n) {
    std::fprintf(stderr, "cannot read %zu bytes from %s\n", n, path);
    std::exit(3);
  }
  std::fclose(f);
  return v;
}

int main(int argc, char **argv) {
  if (argc != 11 && argc != 12) {
    std::fprintf(stderr, "usage: adapter_driver und.raw def.raw nxt.raw rows cols x_begin x_end hs vs out.bin [devices]\n");
    return 2;
  }
  const int devices = argc == 12 ? std::atoi(argv[11]) : 1;
  const int rows = std::atoi(argv[4]), cols = std::atoi(argv[5]);
  const float fx0 = (float)std::atof(argv[6]), fx1 = (float)std::atof(argv[7]);
  const int hs = std::atoi(argv[8]), vs = std::atoi(argv[9]);
  const auto und = slurp(argv[1], (size_t)rows * cols), def = slurp(argv[2], (size_t)rows * cols),
             nxt = slurp(argv[3], (size_t)rows * cols);
  HipCudaClass c;
  if (c.initialize() < 1)
    return 4;
  c.set_deviceCount(devices); // > 1: one engine per device behind the same calls (include/lk_group.h)
  c.set_max_iters(50);
  c.set_precision(0.001f);
  c.set_fitting_model(fm_UVUxUyVxVy);
  c.set_interpolation_model(im_bicubic);
  if (c.resetImagePyramids(und.data(), def.data(), nxt.data(), rows, cols, cols, color_monochrome, 0, 1, 2) != error_none)
    return 5;
#ifndef ADAPTER_DRIVER_MOCK
  if (c.handle())
    lk_set_batch_invariant(c.handle(), 1); // a sector's record then does not depend on what else is in the launch
  if (c.group_handle())
    for (int r = 0; r < lk_group_size(c.group_handle()); ++r) {
      lk_engine *e = nullptr;
      if (lk_group_engine(c.group_handle(), r, &e) == LK_ERROR_NONE)
        lk_set_batch_invariant(e, 1);
    }
#endif
  // sector geometry of the rectangular domain, manager_class.cpp:283-310
  const int x0 = (int)fx0, x1 = (int)fx1;
  const int xdim = (std::abs(x1 - x0) / hs - 1) / 2, ydim = (std::abs(x1 - x0) / vs - 1) / 2;
  const float fxdim = (std::abs(fx1 - fx0) / (float)hs - 1.f) / 2.f, fydim = (std::abs(fx1 - fx0) / (float)vs - 1.f) / 2.f;
  const int S = hs * vs;
  std::vector<CorrelationResult> out((size_t)2 * S);
  std::vector<float> guess((size_t)6 * S, 0.f);
  frame_results fr{};
  const auto t_f0 = std::chrono::steady_clock::now();
  for (int i = 0; i < hs; ++i)
    for (int j = 0; j < vs; ++j) {
      const int iSector = i * vs + j;
      const int cx = (int)(0.5f + fx0 + fxdim + (2.f * fxdim + 1.f) * (float)i);
      const int cy = (int)(0.5f + fx0 + fydim + (2.f * fydim + 1.f) * (float)j);
      if (c.resetPolygon(iSector, cx - xdim, cy - ydim, cx + xdim, cy + ydim) != error_none)
        return 6;
      if (devices == 1)
        out[(size_t)iSector] = *c.correlate(iSector, &guess[(size_t)6 * iSector], fr);
    }
  if (devices > 1) { // several devices: every sector registered first, then the frame in one sharded solve
    int n = 0;
    const CorrelationResult *all = c.correlateAll(guess.data(), &n);
    if (!all || n != S)
      return 8;
    for (int s = 0; s < S; ++s) {
      out[(size_t)s] = all[s];
      for (int i = 0; i < 6; ++i)
        guess[(size_t)6 * s + i] = all[s].resultingParameters[i];
    }
  }
  const double ms_f0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_f0).count();
  // the preload queue (CudaClass::tempQ): the next frame comes from the queue, pushed ahead of time
  if (c.resetNextPyramid() == error_none) // nothing preloaded yet: must be refused
    return 10;
  c.preloadNextImage(nxt.data(), rows, cols, cols);
  c.preloadNextImage(def.data(), rows, cols, cols); // (a second one stays queued)
  if (c.resetNextPyramid() != error_none || c.preloaded.size() != 1)
    return 11;
  c.makeDefPyramidFromNxt();
  const auto t_f1 = std::chrono::steady_clock::now();
  for (int iSector = 0; iSector < S; ++iSector) {
    c.updatePolygon(iSector, def_Lagrangian);
    if (devices == 1)
      out[(size_t)S + iSector] = *c.correlate(iSector, &guess[(size_t)6 * iSector], fr); // guess = the previous result
  }
  if (devices > 1) {
    int n = 0;
    const CorrelationResult *all = c.correlateAll(guess.data(), &n);
    if (!all || n != S)
      return 9;
    for (int s = 0; s < S; ++s)
      out[(size_t)S + s] = all[s];
  }
  const double ms_f1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_f1).count();
  std::printf("frame 0 (register + solve, sector by sector): %.1f ms; frame 1 (move + solve): %.1f ms\n", ms_f0, ms_f1);
  FILE *f = std::fopen(argv[10], "wb");
  if (!f || std::fwrite(out.data(), sizeof(CorrelationResult), out.size(), f) != out.size())
    return 7;
  std::fclose(f);
  // the sample lists follow the sector (getUndXY0ToCPU after the move)
  const v_points p0 = c.getUndXY0ToCPU(0);
  std::printf("sectors %d first sample of sector 0 after the move: %g %g (n = %zu)\n", S, p0.empty() ? -1.f : p0[0].first,
              p0.empty() ? -1.f : p0[0].second, p0.size());
#ifdef ADAPTER_DRIVER_MOCK
  std::printf("commits %d\n%s", lk_mock_commits(), lk_mock_journal());
#endif
  return 0;
}
