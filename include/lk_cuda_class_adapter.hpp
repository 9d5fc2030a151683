// lk_cuda_class_adapter.hpp - header-only C++ adapter that gives the MI355X engine
// (include/lk_engine.h, liblk_engine.so) the public surface of the reference's GPU engine
// `CudaClass` (cuda_class.cuh:46-79), so that managerClass / MainApp keep their call sites:
//
//   reference call site (manager_class.cpp / mainapp.cpp)        forwards to
//   ---------------------------------------------------------    ---------------------------
//   cuda_manager->initialize()                mainapp.cpp:824     lk_device_count
//   set_deviceCount(n)                           :1677            n > 1: lk_group (one engine per device)
//   set_max_iters / set_precision /
//   set_fitting_model / set_interpolation_model  :1677-1685       stored, applied at (re)create
//   resetImagePyramids(und,def,nxt,color,start,step,stop) :915    lk_create + lk_set_image x3
//   resetNextPyramid(path)              manager_class.cpp:257     lk_set_image(LK_IMG_NXT)
//   tempQ (preloaded frames)            mainapp.cpp:926-954       preloadNextImage / resetNextPyramid()
//   makeUndPyramidFromDef / makeDefPyramidFromNxt    :194,:234    lk_rotate_*
//   resetPolygon(iSector,x0,y0,x1,y1)                :340         lk_set_sector_rect
//   resetPolygon(iSector,r,dr,a,da,cx,cy,as)         :610         lk_set_sector_annular
//   resetPolygon(v_points)                           :1037        lk_set_sector_blob
//   correlate(iSector, guess, results)               :449         lk_correlate
//   getUndXY0ToCPU / getDefXY0ToCPU                  :344,:475    lk_get_und_xy / lk_get_def_xy
//
// It is compiled INSIDE the reference tree (it includes the reference's own enums.hpp and
// domains.hpp for CorrelationResult, frame_results, v_points and the enums); nothing of the
// reference is copied here.  Image decoding stays with the caller: the pyramid entry points
// take decoded 8-bit monochrome pixels (the reference decodes with cv::imread inside
// CudaClass, cuda_class.cu:498-510); define LK_ADAPTER_WITH_OPENCV to get the path-based
// overloads back.
//
// Differences a maintainer should know (all follow the CPU engine, the parity target):
//   * correlate() returns CPU-engine semantics (look-ahead parameters, last level's
//     iteration count, 1/n-scaled chi), not the CUDA path's;
//   * the returned CorrelationResult* points into this adapter (valid until the next
//     correlate() on it), like the reference's pinned staging struct (cuda_polygon.cuh:339);
//   * HIP failures come back as error_cuda in errorCode instead of exit(EXIT_FAILURE).
//   * For whole frames use correlateAll(): one device-resident batch instead of one call
//     per sector - that is where the speed is.
#pragma once

#include <cstdint>
#include <cstring>
#include <queue>
#include <string>
#include <utility>
#include <vector>

#include "domains.hpp" // reference: CorrelationResult, frame_results, v_points
#include "enums.hpp"   // reference: errorEnum, fittingModelEnum, ...
#include "lk_engine.h"
#include "lk_group.h"

#ifdef LK_ADAPTER_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/imgcodecs.hpp>
#endif

class HipCudaClass {
  lk_engine *engine_ = nullptr;
  lk_group *group_ = nullptr; // set_deviceCount(n > 1): one engine per device, sectors sharded (lk_group.h)
  lk_config cfg_{LK_IM_BICUBIC, LK_FM_UVUXUYVXVY, 0.001f, 50, 0, 1, 2, 0};
  int deviceCount_ = 1;
  bool sectors_dirty_ = false;
  CorrelationResult last_{};
  int last_rc_ = LK_ERROR_NONE;
  std::vector<lk_result> batch_;

  static_assert(sizeof(CorrelationResult) == sizeof(lk_result), "lk_result must mirror CorrelationResult");

  bool ensure_engine() {
    if (engine_ || group_)
      return true;
    if (deviceCount_ > 1)
      return lk_group_create(&cfg_, deviceCount_, nullptr, &group_) == LK_ERROR_NONE;
    return lk_create(&cfg_, &engine_) == LK_ERROR_NONE;
  }
  void recreate() {
    if (engine_)
      lk_destroy(engine_);
    if (group_)
      lk_group_destroy(group_);
    engine_ = nullptr;
    group_ = nullptr;
  }
  // Annular / blob sectors are rasterised at commit (on the device), so an empty one is reported HERE
  // (LK_ERROR_BAD_DOMAIN) rather than by resetPolygon; a failed commit stays pending - the next call tries
  // again after the caller re-registered the offending sector - and its code is what correlate() reports.
  int commit_rc_ = LK_ERROR_NONE;
  bool commit() {
    if (!sectors_dirty_)
      return true;
    commit_rc_ = group_ ? lk_group_commit_sectors(group_) : lk_commit_sectors(engine_);
    sectors_dirty_ = commit_rc_ != LK_ERROR_NONE;
    return commit_rc_ == LK_ERROR_NONE;
  }
  // group mode: the engine that owns a global sector index and the sector's index there
  lk_engine *owner(int iSector, int *local) {
    if (!group_) {
      *local = iSector;
      return engine_;
    }
    for (int r = 0; r < lk_group_size(group_); ++r) {
      int first = 0, count = 0;
      lk_engine *e = nullptr;
      if (lk_group_shard(group_, r, &first, &count) == LK_ERROR_NONE && iSector >= first && iSector < first + count &&
          lk_group_engine(group_, r, &e) == LK_ERROR_NONE) {
        *local = iSector - first;
        return e;
      }
    }
    return nullptr;
  }

public:
  HipCudaClass() = default;
  HipCudaClass(const HipCudaClass &) = delete;
  HipCudaClass &operator=(const HipCudaClass &) = delete;
  ~HipCudaClass() { recreate(); }

  int initialize() { return lk_device_count(); } // cuda_class.cu:39-75

  // cuda_class.cu:77-83.  The reference keeps this at 1 (README.md:33, `iGPU = 0` cuda_class.cu:333); here
  // n > 1 shards the sectors over n devices (include/lk_group.h): register every sector, then solve the
  // frame with correlateAll().  correlate(iSector) works too (on the owning device) once all sectors are
  // registered; interleaving registration and per-sector solves re-deals the shards at every commit.
  void set_deviceCount(int n) {
    n = n < 1 ? 1 : n;
    if (n != deviceCount_) {
      deviceCount_ = n;
      recreate();
    }
  }
  void set_max_iters(int n) {
    if (cfg_.max_iters != n) {
      cfg_.max_iters = n;
      recreate();
    }
  }
  void set_precision(float p) {
    if (cfg_.precision != p) {
      cfg_.precision = p;
      recreate();
    }
  }
  void set_fitting_model(fittingModelEnum m) {
    if (cfg_.fitting_model != (int)m) {
      cfg_.fitting_model = (int)m;
      recreate();
    }
  }
  void set_interpolation_model(interpolationModelEnum m) {
    if (cfg_.interpolation != (int)m) {
      cfg_.interpolation = (int)m;
      recreate();
    }
  }

  // decoded monochrome pixels, `step` bytes per row; nxt may be null
  errorEnum resetImagePyramids(const uint8_t *und, const uint8_t *def, const uint8_t *nxt, int rows,
                               int cols, int step, colorEnum /*monochrome only*/, int start, int stepLvl,
                               int stop) {
    if (cfg_.py_start != start || cfg_.py_step != stepLvl || cfg_.py_stop != stop) {
      cfg_.py_start = start;
      cfg_.py_step = stepLvl;
      cfg_.py_stop = stop;
      recreate();
    }
    if (!ensure_engine())
      return error_cuda;
    int rc = set_image(LK_IMG_UND, und, rows, cols, step);
    if (!rc)
      rc = set_image(LK_IMG_DEF, def, rows, cols, step);
    if (!rc && nxt)
      rc = set_image(LK_IMG_NXT, nxt, rows, cols, step);
    return (errorEnum)rc;
  }
  errorEnum resetNextPyramid(const uint8_t *nxt, int rows, int cols, int step) {
    return ensure_engine() ? (errorEnum)set_image(LK_IMG_NXT, nxt, rows, cols, step) : error_cuda;
  }
  // The preload queue (CudaClass::tempQ, cuda_class.cuh:79): MainApp::loadNxtGpuImages (mainapp.cpp:926-954) decodes
  // frames 2.. ahead of time and pushes them; resetNextPyramid takes the front instead of decoding
  // (cuda_class.cu:532-552).  Here as decoded 8-bit frames: preloadNextImage() pushes a copy, resetNextPyramid()
  // without arguments uploads the front and pops it (error_bad_domain when the queue is empty).  With
  // LK_ADAPTER_WITH_OPENCV the reference's own member - std::queue<cv::Mat> tempQ - exists as well and the
  // path-based resetNextPyramid prefers it, exactly as the reference does.
  struct PreloadedFrame {
    std::vector<uint8_t> pixels; // dense rows
    int rows = 0, cols = 0;
  };
  std::queue<PreloadedFrame> preloaded;
  void preloadNextImage(const uint8_t *pixels, int rows, int cols, int step) {
    PreloadedFrame f;
    f.rows = rows, f.cols = cols;
    f.pixels.resize((size_t)rows * (size_t)cols);
    for (int r = 0; r < rows; ++r)
      std::memcpy(f.pixels.data() + (size_t)r * (size_t)cols, pixels + (size_t)r * (size_t)step, (size_t)cols);
    preloaded.push(std::move(f));
  }
  errorEnum resetNextPyramid() {
    if (preloaded.empty())
      return (errorEnum)LK_ERROR_BAD_DOMAIN;
    const PreloadedFrame &f = preloaded.front();
    const errorEnum rc = resetNextPyramid(f.pixels.data(), f.rows, f.cols, f.cols);
    preloaded.pop();
    return rc;
  }
#ifdef LK_ADAPTER_WITH_OPENCV
  std::queue<cv::Mat> tempQ;
  void resetImagePyramids(const std::string undPath, const std::string defPath, const std::string nxtPath,
                          colorEnum color, const int start, const int step, const int stop) {
    cv::Mat u = cv::imread(undPath, cv::IMREAD_GRAYSCALE), d = cv::imread(defPath, cv::IMREAD_GRAYSCALE);
    cv::Mat n = nxtPath.empty() ? cv::Mat() : cv::imread(nxtPath, cv::IMREAD_GRAYSCALE);
    resetImagePyramids(u.data, d.data, n.empty() ? nullptr : n.data, u.rows, u.cols, (int)u.step1(), color,
                       start, step, stop);
  }
  void resetNextPyramid(const std::string nxtPath) {
    cv::Mat n;
    if (tempQ.empty()) {
      n = cv::imread(nxtPath, cv::IMREAD_GRAYSCALE);
    } else {
      n = tempQ.front();
      tempQ.pop();
    }
    resetNextPyramid(n.data, n.rows, n.cols, (int)n.step1());
  }
#endif
  void makeUndPyramidFromDef() {
    if (group_)
      lk_group_rotate_und_from_def(group_);
    else if (engine_)
      lk_rotate_und_from_def(engine_);
  }
  void makeDefPyramidFromNxt() {
    if (group_)
      lk_group_rotate_def_from_nxt(group_);
    else if (engine_)
      lk_rotate_def_from_nxt(engine_);
  }

  errorEnum resetPolygon(int iSector, int x0, int y0, int x1, int y1) {
    if (!ensure_engine())
      return error_cuda;
    sectors_dirty_ = true;
    return (errorEnum)(group_ ? lk_group_set_sector_rect(group_, iSector, x0, y0, x1, y1)
                              : lk_set_sector_rect(engine_, iSector, x0, y0, x1, y1));
  }
  errorEnum resetPolygon(int iSector, float r, float dr, float a, float da, float cx, float cy, int as) {
    if (!ensure_engine())
      return error_cuda;
    sectors_dirty_ = true;
    return (errorEnum)(group_ ? lk_group_set_sector_annular(group_, iSector, r, dr, a, da, cx, cy, as)
                              : lk_set_sector_annular(engine_, iSector, r, dr, a, da, cx, cy, as));
  }
  errorEnum resetPolygon(v_points blobContour) { // the blob domain is always sector 0
    if (!ensure_engine())
      return error_cuda;
    if (group_)
      return error_bad_domain; // one sector: nothing to shard - use set_deviceCount(1)
    std::vector<float> c;
    c.reserve(2 * blobContour.size());
    for (const auto &pt : blobContour) {
      c.push_back(pt.first);
      c.push_back(pt.second);
    }
    sectors_dirty_ = true;
    return (errorEnum)lk_set_sector_blob(engine_, 0, c.data(), (int)blobContour.size());
  }
  // Lagrangian domain updates (cuda_polygon.cu:268-415), CPU-manager semantics
  // (manager_class.cpp:354-419): the sector follows its own last record.  One sector per
  // call rebuilds the engine's lists before the next solve; whole frames go through
  // lk_sequence_frame / lk_translate_sectors / lk_rewarp_sectors (include/lk_tracker.h).
  void updatePolygon(int iSector, deformationDescriptionEnum deformationDescription) {
    int local = 0;
    if (ensure_engine() && commit())
      if (lk_engine *e = owner(iSector, &local))
        lk_update_sector(e, local, (int)deformationDescription);
  }

  CorrelationResult *correlate(int iSector, float *initial_guess_, frame_results & /*results*/) {
    lk_result r{};
    int local = 0;
    commit_rc_ = LK_ERROR_NONE;
    lk_engine *e = ensure_engine() && commit() ? owner(iSector, &local) : nullptr;
    int rc = e ? lk_correlate(e, local, initial_guess_, &r) : (commit_rc_ ? commit_rc_ : LK_ERROR_DEVICE);
    if (rc)
      r.errorCode = rc;
    for (int i = 0; i < 6; ++i)
      last_.resultingParameters[i] = r.resultingParameters[i];
    last_.chi = r.chi;
    last_.numberOfPoints = r.numberOfPoints;
    last_.iterations = r.iterations;
    last_.errorCode = (errorEnum)r.errorCode;
    last_.undCenterX = r.undCenterX;
    last_.undCenterY = r.undCenterY;
    return &last_;
  }

  // the batched fast path: every registered sector of the pair in one device-resident solve
  // (nullptr on failure; lastError() tells why - e.g. error_bad_domain for an empty annular sector)
  errorEnum lastError() const { return (errorEnum)last_rc_; }
  const CorrelationResult *correlateAll(const float *guesses /*[S][6]*/, int *count) {
    last_rc_ = LK_ERROR_DEVICE;
    commit_rc_ = LK_ERROR_NONE;
    if (!ensure_engine() || !commit()) {
      if (commit_rc_)
        last_rc_ = commit_rc_;
      return nullptr;
    }
    int S = group_ ? lk_group_sector_count(group_) : lk_sector_count(engine_);
    batch_.resize((size_t)S);
    last_rc_ = group_ ? lk_group_correlate_all(group_, guesses, batch_.data()) : lk_correlate_all(engine_, guesses, batch_.data());
    if (last_rc_ != LK_ERROR_NONE)
      return nullptr;
    if (count)
      *count = S;
    return reinterpret_cast<const CorrelationResult *>(batch_.data());
  }

  v_points getUndXY0ToCPU(int iSector) {
    v_points out;
    int n = 0, local = 0;
    lk_engine *e = (engine_ || (group_ && commit())) ? owner(iSector, &local) : nullptr;
    if (!e || lk_get_und_xy(e, local, nullptr, 0, &n) != LK_ERROR_NONE)
      return out;
    std::vector<float> xy(2 * (size_t)n);
    lk_get_und_xy(e, local, xy.data(), n, &n);
    out.resize((size_t)n);
    for (int i = 0; i < n; ++i)
      out[(size_t)i] = std::make_pair(xy[2 * (size_t)i], xy[2 * (size_t)i + 1]);
    return out;
  }
  // the reference warps with the sector's last result; pass it explicitly
  v_points getDefXY0ToCPU(int iSector, const float *parameters) {
    v_points out;
    int n = 0, local = 0;
    lk_engine *e = (engine_ || group_) && commit() ? owner(iSector, &local) : nullptr;
    if (!e || lk_get_def_xy(e, local, parameters, nullptr, 0, &n) != LK_ERROR_NONE)
      return out;
    std::vector<float> xy(2 * (size_t)n);
    lk_get_def_xy(e, local, parameters, xy.data(), n, &n);
    out.resize((size_t)n);
    for (int i = 0; i < n; ++i)
      out[(size_t)i] = std::make_pair(xy[2 * (size_t)i], xy[2 * (size_t)i + 1]);
    return out;
  }
  v_points getDefXY0ToCPU(int iSector) { return getDefXY0ToCPU(iSector, last_.resultingParameters); }

  lk_engine *handle() { return engine_; }
  lk_group *group_handle() { return group_; }

private:
  int set_image(int slot, const uint8_t *px, int rows, int cols, int step) {
    return group_ ? lk_group_set_image(group_, slot, px, rows, cols, step) : lk_set_image(engine_, slot, px, rows, cols, step);
  }
};
