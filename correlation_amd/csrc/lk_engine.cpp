// lk_engine.cpp - host side of the gfx950 Lucas-Kanade engine behind include/lk_engine.h.
//
// What lives where (HBM layout, one engine per GPU):
//   images      3 slots (und, def, nxt) x (py_stop+1) levels, row-major u8, pitch = cols,
//               two zeroed guard rows per level.  2048^2: 4 MiB + 1 MiB + 256 KiB per slot.
//   sample lists per level L: one concatenated float2 array + uint32 offsets [S+1]
//               (level 0 = the ROI's samples in the CPU engine's order, level L = the
//               decimated list of pyramid_class.cpp:301-322).
//   sector state center[S] (float2), guess[S][6], last_p[S][6], prev_p[S][6],
//               result[S] (48 B, layout of CorrelationResult), stats[S][4].
//   rectangular sectors have no list at all: int4 {x_first, y_first, width, n} per level.
//   sequences   a ring of K deformed-frame pyramids + per-window records / counters / granule chain (frame-pipelined
//               windows: lk_correlate_sequence_async - every sector advances through the frames of a window on its own).
// Everything for an image pair is resident; lk_correlate_all* is one launch per size class
// (16 / 32 / 64 lanes, 4 / 8 wavefronts, teams) on one stream, preceded by the one-lane kernel
// and its finisher when the class has a starved pyramid level.
#include "lk_device.hpp"
#include "lk_internal.hpp"
#include "lk_roi.hpp"

#include <rocprofiler-sdk-roctx/roctx.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

hipError_t lk_launch_solve(const LkSolveArgs &a, int model, int interp, int group, hipStream_t st);
hipError_t lk_launch_solve_seq(const LkSolveArgs &a, int model, int interp, int group, int flavour, hipStream_t st);
hipError_t lk_launch_eval(const LkEvalArgs &a, int model, int interp, int group, hipStream_t st);
hipError_t lk_launch_solve_only(int n, const float *d_in, float *d_out, hipStream_t st);
hipError_t lk_launch_sample(int interp, const uint8_t *def, int rows, int cols, const float2 *pts, int n,
                            float4 *out, hipStream_t st);
hipError_t lk_launch_pyramid(const uint8_t *src, int srows, int scols, uint8_t *dst, hipStream_t st);
hipError_t lk_launch_set_views(const LkLevelView *h_views, LkLevelView *d_views, hipStream_t st);
hipError_t lk_launch_pyramid2(int n_images, const uint8_t *const *src, const int *step, int rows, int cols,
                              uint8_t *const *l0, uint8_t *const *l1, uint8_t *const *l2, hipStream_t st);
hipError_t lk_launch_guess(const float2 *center, const float *last_p, float *prev_p, float *guess,
                           const float *global_guess, float gcx, float gcy, int n_sectors, int model,
                           int frame, int constant_velocity, hipStream_t st);
hipError_t lk_launch_warp_points(const float2 *xy, int n, float cx, float cy, int model, const float *d_p,
                                 float2 *out, hipStream_t st);
hipError_t lk_launch_rewarp(const LkRewarpArgs &a, int model, hipStream_t st);
int lk_decimate_tiles(uint32_t n_max);
hipError_t lk_launch_decimate(const float2 *xy_prev, const uint32_t *off_prev, const uint32_t *n_prev, uint32_t n_max,
                              int level_delta, int n_sectors, uint32_t *pos, uint32_t *tiles, float2 *xy_out,
                              uint32_t *off_out, uint32_t *n_out, hipStream_t st);
hipError_t lk_launch_roi_count(const LkRoiSector *sectors, const LkRoiFlat *flats, const uint32_t *tile_begin, int n_sectors,
                               uint32_t n_tiles, uint32_t *tiles, uint32_t *n_out, hipStream_t st, int rows = 0);
hipError_t lk_launch_roi_fill(const LkRoiSector *sectors, const LkRoiFlat *flats, const uint32_t *tile_begin, int n_sectors,
                              uint32_t n_tiles, const uint32_t *tiles, float2 *xy, uint32_t *off, hipStream_t st, int rows = 0);
hipError_t lk_launch_stale_iterations(lk_result *r, int n, const int *carry_in, int *carry_out, hipStream_t st);
hipError_t lk_launch_append_sector(const LkAppendArgs &a, hipStream_t st);
hipError_t lk_launch_mean_center(const float2 *xy, const uint32_t *off, int n_sectors, float2 *center, hipStream_t st);
size_t lk_mean_center_int_scratch_bytes(uint32_t n_samples, int n_sectors);
hipError_t lk_launch_mean_center_int(const float2 *xy, const uint32_t *off, uint32_t n_samples, int n_sectors, void *scratch,
                                     float2 *center, hipStream_t st);

namespace {

// roctx ranges where the reference has NVTX ranges (cuda_class.cu:133,277,310-326,497-511,528,554,
// cuda_polygon.cu:237,247, cuda_pyramid.cu:219,226): upload + pyramid, sector commit / rebuild,
// solve, record gather.  `rocprofv3 --marker-trace --kernel-trace` shows them over the kernels;
// without a profiler attached a push / pop pair is two calls into an empty table.
struct Range {
  explicit Range(const char *name) { (void)roctxRangePushA(name); }
  ~Range() { (void)roctxRangePop(); }
  Range(const Range &) = delete;
  Range &operator=(const Range &) = delete;
};

int n_params_of(int model) {
  switch (model) {
  case LK_FM_U: return 1;
  case LK_FM_UV: return 2;
  case LK_FM_UVQ: return 3;
  case LK_FM_UVUXUYVXVY: return 6;
  default: return -1;
  }
}

struct DevImage {
  uint8_t *lvl[LK_MAX_LEVELS] = {};
  size_t cap[LK_MAX_LEVELS] = {};
  int guard_rows[LK_MAX_LEVELS] = {-1, -1, -1, -1, -1, -1, -1, -1}; // geometry the guard rows were zeroed for
  int guard_cols[LK_MAX_LEVELS] = {-1, -1, -1, -1, -1, -1, -1, -1};
  int rows = 0, cols = 0;
  bool valid = false;
};

struct HostSector {
  std::vector<float> xy; // level-0 AoS (empty for rectangular sectors: generated on demand)
  bool is_rect = false;
  int x0 = 0, y0 = 0, x1 = -1, y1 = -1; // inclusive rectangle (is_rect)
  float cx = 0.f, cy = 0.f;
  bool has_center = false; // the solve centre was given (rectangular path) rather than the samples' mean
  bool set = false;
  bool fresh = true; // registered (lk_set_sector_*) since the last commit: its sequence state starts from zero
  // Annular and blob sectors are registered by their DESCRIPTION; the sample list is made by the
  // device mask at commit (lk_roi_tile_kernel) or, on demand, by the host scan (realize()).
  int lazy = 0;      // 0: xy / rectangle are the sector; 1: annular geometry `ag`; 2: blob half triangles `flats`
  lkroi::AnnularGeometry ag;
  std::vector<lkroi::BlobPolygon::Flat> flats;
  int n0_device = 0; // lazy sectors after a device commit: their sample count
  int n0() const { return is_rect ? (x1 - x0 + 1) * (y1 - y0 + 1) : (lazy ? n0_device : (int)(xy.size() / 2)); }
  void realize() {   // the host scan of a lazily registered sector: same samples, same order
    if (!lazy)
      return;
    xy.clear();
    if (lazy == 1)
      lkroi::annular_points(ag, xy);
    else
      lkroi::BlobPolygon::fill(flats, xy);
    if (!xy.empty())
      lkroi::mean_center(xy.data(), (int)(xy.size() / 2), cx, cy); // correlation_class.cpp:337-339
    lazy = 0;
  }
  std::vector<float> points() const { // level-0 list in the CPU engine's order
    if (!is_rect)
      return xy;
    std::vector<float> v;
    v.reserve(2 * (size_t)n0());
    lkroi::rect_points(x0, y0, x1, y1, v);
    return v;
  }
};

// floor(a / 2^l) and ceil(a / 2^l) for any sign
static int floor_shift(int a, int l) { return a >> l; }
static int ceil_shift(int a, int l) { return -((-a) >> l); }

template <class T> struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t ensure(size_t want) {
    if (want <= n && p)
      return hipSuccess;
    if (p)
      (void)hipFree(p);
    p = nullptr;
    n = 0;
    hipError_t e = hipMalloc((void **)&p, std::max<size_t>(want, 1) * sizeof(T));
    if (e == hipSuccess)
      n = want;
    return e;
  }
  // grow, keeping the first `keep` elements (a commit that adds sectors keeps the state of the others)
  hipError_t ensure_keep(size_t want, size_t keep) {
    if (want <= n && p)
      return hipSuccess;
    T *q = nullptr;
    hipError_t e = hipMalloc((void **)&q, std::max<size_t>(want, 1) * sizeof(T));
    if (e != hipSuccess)
      return e;
    if (p && keep)
      e = hipMemcpy(q, p, std::min(keep, n) * sizeof(T), hipMemcpyDeviceToDevice);
    if (p)
      (void)hipFree(p);
    p = q;
    n = want;
    return e;
  }
  // the same with room to spare (sector-by-sector registration: lk_commit_sectors' append path)
  hipError_t reserve_keep(size_t want, size_t keep) {
    if (want <= n && p)
      return hipSuccess;
    return ensure_keep(std::max(want, 2 * n + 64), keep);
  }
  void release() {
    if (p)
      (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

} // namespace

// lanes that own one sector: a 16-lane DPP row (4 sectors per wavefront), one wavefront,
// a workgroup of 4 / 8 wavefronts, or - for giant sectors (a blob over most of the image) -
// a team of up to kMaxTeam 8-wavefront workgroups
static const int kNumClasses = 6;
static const int kTeamClass = 5;
static const int kGroupOfClass[kNumClasses] = {16, 32, 64, 256, 512, 512};
static const int kTeamSamples = 32768; // level-0 samples per workgroup of a team (64 per lane)
static const int kMaxTeam = kLkMaxTeam; // (256: config 3's blob alone 0.405 -> 0.372 ms against 128; the launch clamps a team to what is resident)
static const int kTeamMinSamples = 4096; // a team workgroup is worth its all-to-all from 8 samples per lane on
static const int kFewBigSectors = 128;   // at most this many 8-wavefront sectors: give them teams
static int size_class(int n0) {
  return n0 <= 512 ? 0 : (n0 <= 2048 ? 1 : (n0 <= 8192 ? 2 : (n0 <= 65536 ? 3 : (n0 <= 8 * kTeamSamples ? 4 : 5))));
}

struct lk_engine {
  lk_config cfg{};
  int P = 0;
  hipStream_t own_stream = nullptr, stream = nullptr, nxt_stream = nullptr;
  hipEvent_t nxt_done = nullptr, ev_s0 = nullptr, ev_s1 = nullptr, ev_p0 = nullptr, ev_p1 = nullptr;
  // size classes are independent sector sets: every class after the first solves on its own
  // stream (forked from / joined into `stream`), so that one class's tail overlaps the others
  hipStream_t class_stream[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool nxt_pending = false, solve_timed = false, pyr_timed = false;
  bool batch_invariant = false; // lk_set_batch_invariant
  DevBuf<int> d_stale;          // reference-order mode: reached_iterations as the last solved sector left it ([2], alternating)
  int stale_par = 0;
  bool defer_stale = false;     // lk_group member: the markers are resolved over the gathered records, in global sector order
  int reference_order = 0;      // lk_set_reference_order: 0 = off, T = the reference's number_of_threads to reproduce
  int pairs_in_flight = 1;      // lk_set_pairs_in_flight: launches that share the GPU
  bool timing = true; // HIP events around pyramid builds and solves (lk_stats.solve_ms / pyramid_ms)
  std::mutex nxt_mu;
  std::string err;

  DevImage img[3];
  LkLevelView h_lv[LK_MAX_LEVELS]{}, h_lv_sent[LK_MAX_LEVELS]{};
  DevBuf<LkLevelView> d_lv;
  bool lv_dirty = true;

  std::vector<HostSector> hs; // staging until commit
  std::vector<HostSector> hs_backup; // the lists before the last lk_translate / lk_rewarp_sectors
  bool committed = false;
  // Sector-by-sector registration (the reference's first-frame loop: resetPolygon(i), correlate(i),
  // manager_class.cpp:340, :449): a commit that only ADDS rectangles behind the committed ones appends them
  // (O(1) host work and one small launch per sector) instead of rebuilding everything; the size-class
  // analysis of the whole domain is then redone lazily, by the next batch solve.
  bool append_ok = false;     // the committed state can be extended in place (plain sectors, no pending rebuilds)
  bool classes_dirty = false; // h_order / class_begin / promotions are stale (sectors were appended)
  bool recommit_pending = false; // lk_update_sector moved sample lists: rebuild before the next solve
  // lk_rewarp_sectors rebuilds the lists on the device: hs[].xy / cx / cy are stale until a host
  // consumer asks for them (materialize_host); the previous level-0 lists stay in d_xy0_alt
  bool lists_on_device = false, backup_on_device = false;
  DevBuf<float2> d_xy0_alt;
  DevBuf<uint32_t> d_off0_alt, d_pos, d_tiles, d_level_total;
  DevBuf<LkRoiSector> d_roi_sectors; // device ROI masks (commit of annular / blob sectors)
  DevBuf<LkRoiFlat> d_roi_flats;
  DevBuf<uint32_t> d_roi_tile_begin;
  std::vector<float> h_center_prev, h_offsets;
  DevBuf<float2> d_offsets;
  int S = 0;
  std::vector<uint32_t> h_off[LK_MAX_LEVELS];
  DevBuf<float2> d_xy[LK_MAX_LEVELS];
  // the row-major evaluation copy of the lists (LkLevelView::xy_eval; same offsets): built with the device ROI masks
  // when the domain has annular sectors, dropped when the lists move (Lagrangian descriptions)
  DevBuf<float2> d_xy_eval[LK_MAX_LEVELS];
  DevBuf<uint32_t> d_off_eval;   // scratch: the offsets the second pass computes (equal to d_off)
  DevBuf<float2> d_xy_eval0_alt; // level 0 of the copy while the lists move (lk_rewarp_sectors)
  bool eval_lists = false;
  DevBuf<uint32_t> d_off[LK_MAX_LEVELS];
  DevBuf<int4> d_rect[LK_MAX_LEVELS];
  std::vector<int4> h_rect[LK_MAX_LEVELS];
  std::vector<float> h_center; // [S][2]
  DevBuf<float2> d_center;
  DevBuf<float> d_guess, d_last_p, d_prev_p, d_last_eval_p;
  DevBuf<lk_result> d_result;
  DevBuf<uint32_t> d_stats;
  DevBuf<uint32_t> d_order;
  std::vector<uint32_t> h_order; // sectors grouped by size class
  int class_begin[kNumClasses + 1] = {0, 0, 0, 0, 0, 0, 0};
  bool class_starved[kNumClasses] = {false, false, false, false, false, false}; // a sector of the class has a starved level
  int team_share_permille = 0; // > 0: the team class and the one-workgroup class of this domain split the resident slots (classify_sectors)
  bool force_safe = false; // LK_FORCE_SAFE: QR fallback for ill-conditioned systems in the lane-group kernels
  std::vector<int> h_class; // size class of every sector
  DevBuf<uint32_t> d_single, d_queue;
  DevBuf<LkHandoff> d_handoff;
  DevBuf<uint32_t> d_mid, d_finish_list, d_finish_count; // stragglers of the starved-level kernel
  DevBuf<uint32_t> d_ill_list, d_ill_count;              // sectors whose damped system met a bad pivot
  DevBuf<uint32_t> d_mean_scratch;                       // chunk table / sums / maps of lk_mean_center_int_kernel
  int starved_max = -1; // LK_STARVED_MAX as read by the last commit (-1: the default, 2 P)
  int eval_cap = 20; // evaluations a lane of the starved-level kernel spends on one sector (0: no cap; config 4: 12 / 16 / 20 / 24 / 32 -> 2.00 / 1.91 / 1.88 / 1.92 / 2.08 ms)
  int team_w = 0; // workgroups per sector of the team class
  int team_min_samples = 0; // per-sector team sizing (0: every team has team_w workgroups)
  DevBuf<float> d_team_partials;
  DevBuf<uint32_t> d_team_arrivals;
  DevBuf<float> d_scratch; // 64 floats for the stand-alone entry points
  DevBuf<float2> d_warp;
  // Frame-pipelined windows (lk_correlate_sequence_async): a ring of resident deformed-frame pyramids, filled on the
  // next-frame stream, and the buffers of one window - records and counters per frame, the per-sector granule chain that
  // hands a sector's parameters from frame to frame, the per-frame image table, two flag words.
  std::vector<DevImage> ring;
  std::vector<hipEvent_t> ring_ready;   // per slot: its pyramid is built (recorded on nxt_stream)
  std::vector<char> ring_fresh;         // per slot: ring_ready has to be waited for by the next window that reads it
  std::vector<unsigned> ring_window;    // per slot: the last window that read it (+ 1; 0: none)
  hipEvent_t seq_done[4] = {nullptr, nullptr, nullptr, nullptr}; // recorded behind window w on the engine's stream: [w % 4]
  unsigned seq_windows = 0;             // windows launched so far
  DevBuf<lk_result> d_seq_result;       // [frames][S]
  DevBuf<uint32_t> d_seq_stats;         // [frames][S][4]
  DevBuf<float> d_seq_guess;            // [frames][S][6] (LK_SEQ_CHECK / tests)
  DevBuf<unsigned long long> d_seq_chain;
  DevBuf<LkSeqFrame> d_seq_img;
  DevBuf<uint32_t> d_seq_flags;         // [4]: wait bound hit, bad pivot in the fast flavour
  DevBuf<float> d_prev_p_alt;           // previous_resulting_parameters as the window's last frame leaves them
  struct SeqWindow {
    int und_slot = -1, first_slot = 0, n_frames = 0, reference_previous = 0, velocity = 0, want_host = 0, want_guesses = 0;
    bool pipelined = false, outstanding = false;
  } seq;
  uint32_t *h_seq_flags = nullptr;      // pinned [4]
  lk_result *h_seq_results[2] = {nullptr, nullptr}; // pinned [frames][S], alternating by window: the caller may digest window w's
  size_t h_seq_results_n[2] = {0, 0};               //   records (lk_sequence_host_records) while window w + 1 is being solved
  int h_seq_cur = 0;                                // buffer of the window launched last
  hipEvent_t ev_seq = nullptr;
  int stats_frames = 1;                 // frames the counters of the last solve cover (a window: its frames)
  lk_result *h_results = nullptr; // pinned: where lk_correlate_all_async leaves the records
  size_t h_results_n = 0;
  hipEvent_t ev_results = nullptr;
  bool results_pending = false;
  bool stats_valid = false;
  lk_stats stats{};

  int fail(int code, const std::string &what) {
    err = what;
    return code;
  }
  int hipfail(hipError_t e, const char *where) {
    err = std::string(where) + ": " + hipGetErrorString(e);
    return LK_ERROR_DEVICE;
  }
};

#define HIPCHK(call)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (call);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return e->hipfail(_e, #call);                                                                   \
  } while (0)

extern "C" {

int lk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

int lk_create(const lk_config *cfg, lk_engine **out) {
  if (!cfg || !out)
    return LK_ERROR_BAD_DOMAIN;
  *out = nullptr;
  if (n_params_of(cfg->fitting_model) < 0 || cfg->interpolation < 0 || cfg->interpolation > LK_IM_BICUBIC_SEPARABLE)
    return LK_ERROR_BAD_DOMAIN;
  if (cfg->py_step < 1 || cfg->py_start < 0 || cfg->py_stop < cfg->py_start ||
      cfg->py_stop >= LK_MAX_LEVELS || (cfg->py_stop - cfg->py_start) % cfg->py_step != 0)
    return LK_ERROR_BAD_DOMAIN;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return LK_ERROR_DEVICE; // no silent CPU fallback: the HIP device is the product
  if (cfg->device < 0 || cfg->device >= ndev)
    return LK_ERROR_DEVICE;
  lk_engine *e = new lk_engine();
  e->cfg = *cfg;
  e->P = n_params_of(cfg->fitting_model);
  bool ok = hipSetDevice(cfg->device) == hipSuccess &&
            hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&e->nxt_stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&e->nxt_done, hipEventDisableTiming) == hipSuccess &&
            hipEventCreate(&e->ev_s0) == hipSuccess && hipEventCreate(&e->ev_s1) == hipSuccess &&
            hipEventCreate(&e->ev_p0) == hipSuccess && hipEventCreate(&e->ev_p1) == hipSuccess;
  if (!ok) {
    delete e;
    return LK_ERROR_DEVICE;
  }
  e->stream = e->own_stream;
  *out = e;
  return LK_ERROR_NONE;
}

void lk_destroy(lk_engine *e) {
  if (!e)
    return;
  (void)hipSetDevice(e->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto &im : e->img)
    for (auto &p : im.lvl)
      if (p)
        (void)hipFree(p);
  e->d_lv.release();
  for (int l = 0; l < LK_MAX_LEVELS; ++l) {
    e->d_xy[l].release();
    e->d_xy_eval[l].release();
    e->d_off[l].release();
    e->d_rect[l].release();
  }
  e->d_center.release();
  e->d_xy0_alt.release();
  e->d_xy_eval0_alt.release();
  e->d_off_eval.release();
  e->d_off0_alt.release();
  e->d_pos.release();
  e->d_tiles.release();
  e->d_level_total.release();
  e->d_roi_sectors.release();
  e->d_roi_flats.release();
  e->d_roi_tile_begin.release();
  e->d_offsets.release();
  e->d_guess.release();
  e->d_last_p.release();
  e->d_last_eval_p.release();
  e->d_prev_p.release();
  e->d_result.release();
  e->d_stats.release();
  e->d_order.release();
  e->d_single.release();
  e->d_queue.release();
  e->d_handoff.release();
  e->d_mid.release();
  e->d_finish_list.release();
  e->d_finish_count.release();
  e->d_ill_list.release();
  e->d_ill_count.release();
  e->d_mean_scratch.release();
  e->d_scratch.release();
  e->d_stale.release();
  e->d_warp.release();
  e->d_team_partials.release();
  e->d_team_arrivals.release();
  for (auto &im : e->ring)
    for (auto &p : im.lvl)
      if (p)
        (void)hipFree(p);
  for (hipEvent_t ev : e->ring_ready)
    if (ev)
      (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->seq_done)
    if (ev)
      (void)hipEventDestroy(ev);
  if (e->ev_seq)
    (void)hipEventDestroy(e->ev_seq);
  e->d_seq_result.release();
  e->d_seq_stats.release();
  e->d_seq_guess.release();
  e->d_seq_chain.release();
  e->d_seq_img.release();
  e->d_seq_flags.release();
  e->d_prev_p_alt.release();
  if (e->h_seq_flags)
    (void)hipHostFree(e->h_seq_flags);
  for (lk_result *hp : e->h_seq_results)
    if (hp)
      (void)hipHostFree(hp);
  for (hipStream_t cs : e->class_stream)
    if (cs)
      (void)hipStreamDestroy(cs);
  for (hipEvent_t ev : e->ev_join)
    if (ev)
      (void)hipEventDestroy(ev);
  if (e->ev_fork)
    (void)hipEventDestroy(e->ev_fork);
  if (e->ev_results)
    (void)hipEventDestroy(e->ev_results);
  if (e->h_results)
    (void)hipHostFree(e->h_results);
  if (e->own_stream)
    (void)hipStreamDestroy(e->own_stream);
  if (e->nxt_stream)
    (void)hipStreamDestroy(e->nxt_stream);
  for (hipEvent_t ev : {e->nxt_done, e->ev_s0, e->ev_s1, e->ev_p0, e->ev_p1})
    if (ev)
      (void)hipEventDestroy(ev);
  delete e;
}

const char *lk_last_error_string(const lk_engine *e) { return e ? e->err.c_str() : "null engine"; }

int lk_set_stream(lk_engine *e, void *hip_stream) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
  return LK_ERROR_NONE;
}

int lk_set_timing(lk_engine *e, int enabled) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  e->timing = enabled != 0;
  return LK_ERROR_NONE;
}

int lk_set_batch_invariant(lk_engine *e, int enabled) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  e->batch_invariant = enabled != 0;
  e->lv_dirty = true; // (which copy of the lists the lane groups walk)
  return LK_ERROR_NONE;
}

int lk_set_reference_order(lk_engine *e, int threads) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (threads < 0 || threads > 4096)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_reference_order: threads must be in 0..4096");
  if ((threads > 0) != (e->reference_order > 0) && e->committed)
    e->recommit_pending = true; // the lane groups depend on the mode (commit_impl)
  e->reference_order = threads;
  return LK_ERROR_NONE;
}

int lk_set_pairs_in_flight(lk_engine *e, int n) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (n < 1 || n > 64)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_pairs_in_flight: n must be in 1..64");
  e->pairs_in_flight = n;
  return LK_ERROR_NONE;
}

int lk_synchronize(lk_engine *e) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->nxt_stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return LK_ERROR_NONE;
}

// ------------------------------------------------------------------------------------
// images
// ------------------------------------------------------------------------------------
// level buffers of one slot for a rows x cols frame (+ two zeroed guard rows per level; the
// pyramid kernels write every target pixel themselves, so the guard rows are only redone when
// the geometry of the level changes)
static int prepare_slot(lk_engine *e, DevImage &im, int rows, int cols, hipStream_t st) {
  int r = rows, c = cols;
  for (int l = 0; l <= e->cfg.py_stop; ++l) {
    size_t need = (size_t)(r + 2) * (size_t)c + 16;
    if (need > im.cap[l]) {
      if (im.lvl[l]) {
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipFree(im.lvl[l]));
        im.lvl[l] = nullptr;
      }
      HIPCHK(hipMalloc((void **)&im.lvl[l], need));
      im.cap[l] = need;
      im.guard_rows[l] = -1;
    }
    if (im.guard_rows[l] != r || im.guard_cols[l] != c) {
      HIPCHK(hipMemsetAsync(im.lvl[l] + (size_t)r * (size_t)c, 0, 2 * (size_t)c + 16, st));
      im.guard_rows[l] = r;
      im.guard_cols[l] = c;
    }
    r /= 2;
    c /= 2;
  }
  return LK_ERROR_NONE;
}

// Is a host pointer pinned (hipHostMalloc / hipHostRegister: lk_pin_host_memory)?  A copy from pinned memory is a DMA the
// stream orders; from pageable memory the runtime stages it, and the caller's buffer is only safe once the stream has
// passed the copy.
static bool host_pinned(const void *p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError(); // (pageable memory: not an error of ours)
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// upload (or device copy) of one frame into `im` + its pyramid levels, on stream `st`
static int fill_image(lk_engine *e, DevImage &im, const void *src, bool src_on_device, int rows, int cols, int step,
                      hipStream_t st, bool timed) {
  {
    int rc = prepare_slot(e, im, rows, cols, st);
    if (rc)
      return rc;
  }
  int r = rows, c = cols;
  // two or more pyramid levels on a device-resident frame: upload copy + levels 1, 2 in ONE
  // launch (lk_pyramid2_kernel); host frames are copied first and the kernel runs in place
  const bool fused = e->cfg.py_stop >= 2 && rows >= 4 && cols >= 4;
  if (!fused || !src_on_device)
    HIPCHK(hipMemcpy2DAsync(im.lvl[0], (size_t)cols, src, (size_t)step, (size_t)cols, (size_t)rows,
                            src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  if (timed)
    HIPCHK(hipEventRecord(e->ev_p0, st));
  r = rows;
  c = cols;
  int l = 1;
  if (fused) {
    const bool in_place = !src_on_device;
    const uint8_t *srcs[1] = {in_place ? im.lvl[0] : (const uint8_t *)src};
    const int steps[1] = {in_place ? cols : step};
    HIPCHK(lk_launch_pyramid2(1, srcs, steps, rows, cols, &im.lvl[0], &im.lvl[1], &im.lvl[2], st));
    r = rows / 4;
    c = cols / 4;
    l = 3;
  }
  for (; l <= e->cfg.py_stop; ++l) { // (remaining) levels up to stop (pyramid_class.cpp:92)
    HIPCHK(lk_launch_pyramid(im.lvl[l - 1], r, c, im.lvl[l], st));
    r /= 2;
    c /= 2;
  }
  if (timed) {
    HIPCHK(hipEventRecord(e->ev_p1, st));
    e->pyr_timed = true;
  }
  im.rows = rows;
  im.cols = cols;
  im.valid = true;
  return LK_ERROR_NONE;
}

// after / consumed (lk_group: frames that arrive by a collective on another stream): the fill waits for `after` on its
// stream instead of the host waiting for the collective, and `consumed` is recorded behind the last kernel that reads src
static int set_image_common(lk_engine *e, int slot, const void *src, bool src_on_device, int rows,
                            int cols, int step, hipEvent_t after = nullptr, hipEvent_t consumed = nullptr) {
  Range range_("lk:upload+pyramid");
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (slot < 0 || slot > 2 || !src || rows < 1 || cols < 1 || step < cols)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_image: bad arguments");
  HIPCHK(hipSetDevice(e->cfg.device));
  // the next-frame slot is filled on its own stream so it can overlap a running solve
  // (manager_class.cpp:1438-1447 / nxtStream in cuda_pyramid.cu:504,557)
  std::unique_lock<std::mutex> lock(e->nxt_mu, std::defer_lock);
  hipStream_t st = e->stream;
  if (slot == LK_IMG_NXT) {
    lock.lock();
    st = e->nxt_stream;
  }
  DevImage &im = e->img[slot];
  if (after)
    HIPCHK(hipStreamWaitEvent(st, after, 0));
  {
    int rc = fill_image(e, im, src, src_on_device, rows, cols, step, st, slot != LK_IMG_NXT && e->timing);
    if (rc)
      return rc;
  }
  if (consumed)
    HIPCHK(hipEventRecord(consumed, st));
  if (slot == LK_IMG_NXT) {
    HIPCHK(hipEventRecord(e->nxt_done, st));
    e->nxt_pending = true;
  } else {
    e->lv_dirty = true;
  }
  if (!src_on_device && !host_pinned(src)) // pageable host memory: the copy must have left it
    HIPCHK(hipStreamSynchronize(st));
  return LK_ERROR_NONE;
}

int lk_pin_host_memory(void *ptr, size_t bytes) {
  if (!ptr || !bytes)
    return LK_ERROR_BAD_DOMAIN;
  return hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess ? LK_ERROR_NONE : LK_ERROR_DEVICE;
}
int lk_unpin_host_memory(void *ptr) {
  if (!ptr)
    return LK_ERROR_BAD_DOMAIN;
  return hipHostUnregister(ptr) == hipSuccess ? LK_ERROR_NONE : LK_ERROR_DEVICE;
}

int lk_set_image(lk_engine *e, int slot, const uint8_t *host_pixels, int rows, int cols, int step) {
  return set_image_common(e, slot, host_pixels, false, rows, cols, step);
}

int lk_set_image_device(lk_engine *e, int slot, const void *device_pixels, int rows, int cols, int step) {
  return set_image_common(e, slot, device_pixels, true, rows, cols, step);
}

static void swap_images(DevImage &a, DevImage &b) { std::swap(a, b); }

// CudaClass::resetImagePyramids takes the frames of a pair together (cuda_class.cu:475-519);
// for device-resident frames both uploads and both pairs of pyramid levels share ONE launch.
int lk_set_image_pair_device(lk_engine *e, const void *und_pixels, int und_step, const void *def_pixels,
                             int def_step, int rows, int cols) {
  Range range_("lk:upload+pyramid pair");
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!und_pixels || !def_pixels || rows < 1 || cols < 1 || und_step < cols || def_step < cols)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_image_pair_device: bad arguments");
  if (e->cfg.py_stop < 2 || rows < 4 || cols < 4) { // nothing to fuse
    int rc = set_image_common(e, LK_IMG_UND, und_pixels, true, rows, cols, und_step);
    return rc ? rc : set_image_common(e, LK_IMG_DEF, def_pixels, true, rows, cols, def_step);
  }
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->stream;
  DevImage &u = e->img[LK_IMG_UND], &d = e->img[LK_IMG_DEF];
  int rc = prepare_slot(e, u, rows, cols, st);
  if (!rc)
    rc = prepare_slot(e, d, rows, cols, st);
  if (rc)
    return rc;
  if (e->timing)
    HIPCHK(hipEventRecord(e->ev_p0, st));
  const uint8_t *srcs[2] = {(const uint8_t *)und_pixels, (const uint8_t *)def_pixels};
  const int steps[2] = {und_step, def_step};
  uint8_t *l0[2] = {u.lvl[0], d.lvl[0]}, *l1[2] = {u.lvl[1], d.lvl[1]}, *l2[2] = {u.lvl[2], d.lvl[2]};
  HIPCHK(lk_launch_pyramid2(2, srcs, steps, rows, cols, l0, l1, l2, st));
  for (DevImage *im : {&u, &d}) {
    int r = rows / 4, c = cols / 4;
    for (int l = 3; l <= e->cfg.py_stop; ++l) {
      HIPCHK(lk_launch_pyramid(im->lvl[l - 1], r, c, im->lvl[l], st));
      r /= 2;
      c /= 2;
    }
    im->rows = rows;
    im->cols = cols;
    im->valid = true;
  }
  if (e->timing) {
    HIPCHK(hipEventRecord(e->ev_p1, st));
    e->pyr_timed = true;
  }
  e->lv_dirty = true;
  return LK_ERROR_NONE;
}

int lk_rotate_und_from_def(lk_engine *e) { // pyramid_class.cpp:211-226: def is emptied
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->img[LK_IMG_DEF].valid)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_rotate_und_from_def: no deformed image");
  swap_images(e->img[LK_IMG_UND], e->img[LK_IMG_DEF]);
  e->img[LK_IMG_DEF].valid = false; // keeps its allocation for reuse
  e->lv_dirty = true;
  return LK_ERROR_NONE;
}

int lk_rotate_def_from_nxt(lk_engine *e) { // pyramid_class.cpp:228-258
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  std::lock_guard<std::mutex> lock(e->nxt_mu);
  if (!e->img[LK_IMG_NXT].valid)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_rotate_def_from_nxt: no next image");
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->nxt_pending) { // fence: the solve stream must see the finished next pyramid
    HIPCHK(hipStreamWaitEvent(e->stream, e->nxt_done, 0));
    e->nxt_pending = false;
  }
  swap_images(e->img[LK_IMG_DEF], e->img[LK_IMG_NXT]);
  e->img[LK_IMG_NXT].valid = false;
  e->lv_dirty = true;
  return LK_ERROR_NONE;
}

int lk_get_pyramid_level(lk_engine *e, int slot, int level, uint8_t *host_out, int *rows, int *cols) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (slot < 0 || slot > 2 || level < 0 || level > e->cfg.py_stop || !e->img[slot].valid)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_pyramid_level: bad slot/level");
  const DevImage &im = e->img[slot];
  int r = im.rows >> level, c = im.cols >> level;
  if (rows)
    *rows = r;
  if (cols)
    *cols = c;
  if (host_out) {
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = slot == LK_IMG_NXT ? e->nxt_stream : e->stream;
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(host_out, im.lvl[level], (size_t)r * (size_t)c, hipMemcpyDeviceToHost));
  }
  return LK_ERROR_NONE;
}

// ------------------------------------------------------------------------------------
// sectors
// ------------------------------------------------------------------------------------
// Level-0 lists that live on the device only (after lk_rewarp_sectors) -> HostSector records.
static int lists_from_device(lk_engine *e, const float2 *d_xy0, const std::vector<float> &centers,
                             std::vector<HostSector> &out) {
  const size_t S = (size_t)e->S, total = e->h_off[0][S];
  std::vector<float> xy(2 * total + 2);
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipMemcpyAsync(xy.data(), d_xy0, total * sizeof(float2), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (size_t s = 0; s < S; ++s) {
    HostSector &h = out[s];
    h.xy.assign(xy.begin() + 2 * (size_t)e->h_off[0][s], xy.begin() + 2 * (size_t)e->h_off[0][s + 1]);
    h.is_rect = false;
    h.lazy = 0;
    h.cx = centers[2 * s];
    h.cy = centers[2 * s + 1];
  }
  return LK_ERROR_NONE;
}

// every entry point that reads or edits e->hs calls this first
static int materialize_host(lk_engine *e) {
  if (!e->lists_on_device)
    return LK_ERROR_NONE;
  int rc = lists_from_device(e, e->d_xy[0].p, e->h_center, e->hs);
  if (rc)
    return rc;
  e->lists_on_device = false;
  return LK_ERROR_NONE;
}

int lk_clear_sectors(lk_engine *e) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  e->lists_on_device = e->backup_on_device = false;
  e->hs.clear();
  e->committed = false;
  e->append_ok = false;
  e->S = 0;
  return LK_ERROR_NONE;
}

// LK_HOST_ROI=1 (test / comparison hook, read per call): annular and blob sectors are rasterised by
// the host scan at registration, as in round 1, instead of by the device mask at commit
static bool host_roi() {
  const char *f = std::getenv("LK_HOST_ROI");
  return f && std::atoi(f) != 0;
}

static HostSector *sector_slot(lk_engine *e, int sector) {
  if (sector < 0 || materialize_host(e))
    return nullptr;
  if ((size_t)sector >= e->hs.size())
    e->hs.resize((size_t)sector + 1);
  if (sector < e->S)
    e->append_ok = false; // a committed sector is registered anew: the next commit rebuilds
  e->committed = false;
  e->hs[(size_t)sector].fresh = true;
  e->hs[(size_t)sector].lazy = 0; // (annular / blob registration sets it again)
  return &e->hs[(size_t)sector];
}

int lk_set_sector_rect(lk_engine *e, int sector, int x0, int y0, int x1, int y1) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  HostSector *s = sector_slot(e, sector);
  if (!s || x1 < x0 || y1 < y0)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sector_rect: bad rectangle");
  s->xy.clear();
  s->is_rect = true;
  s->x0 = x0;
  s->y0 = y0;
  s->x1 = x1;
  s->y1 = y1;
  s->cx = (float)(x0 + x1) * 0.5f;
  s->cy = (float)(y0 + y1) * 0.5f;
  s->has_center = true;
  s->set = true;
  return LK_ERROR_NONE;
}

int lk_set_rect_grid(lk_engine *e, float x_begin, float y_begin, float x_end, float y_end, int hs, int vs,
                     int first, int count) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (hs < 1 || vs < 1)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_rect_grid: subdivisions must be >= 1");
  const int total = hs * vs;
  if (count < 0)
    count = total - first;
  if (first < 0 || count < 0 || first + count > total)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_rect_grid: sector range outside the grid");
  lkroi::RectGrid g = lkroi::rect_grid(x_begin, y_begin, x_end, y_end, hs, vs);
  if (g.xdim < 0 || g.ydim < 0)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_rect_grid: domain smaller than the grid");
  e->lists_on_device = e->backup_on_device = false;
  e->hs.clear();
  e->hs.resize((size_t)count);
  e->committed = false;
  e->append_ok = false;
  for (int k = 0; k < count; ++k) {
    int iSector = first + k, i = iSector / vs, j = iSector % vs; // iSector = i*vs + j
    HostSector &s = e->hs[(size_t)k];
    int cx = g.cx[i], cy = g.cy[j];
    s.is_rect = true;
    s.x0 = cx - g.xdim;
    s.y0 = cy - g.ydim;
    s.x1 = cx + g.xdim;
    s.y1 = cy + g.ydim;
    s.cx = (float)cx; // manager_class.cpp:438-441 passes the integer centre
    s.cy = (float)cy;
    s.has_center = true;
    s.set = true;
  }
  return LK_ERROR_NONE;
}

int lk_set_sector_annular(lk_engine *e, int sector, float r, float dr, float a, float da, float cx, float cy,
                          int as) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  HostSector *s = sector_slot(e, sector);
  if (!s)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sector_annular: bad sector index");
  s->xy.clear();
  s->is_rect = false;
  s->flats.clear();
  if (!lkroi::annular_geometry(r, dr, a, da, cx, cy, as, s->ag))
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sector_annular: bad sector description");
  s->lazy = 1; // the samples come from the device mask at commit (or from realize())
  if (host_roi()) {
    s->realize();
    if (s->xy.empty())
      return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sector_annular: empty sector");
  }
  s->has_center = false;
  s->set = true;
  return LK_ERROR_NONE;
}

int lk_set_sectors_annular(lk_engine *e, int first_sector, int count, const float *params, int as) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (first_sector < 0 || count < 1 || !params)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sectors_annular: bad arguments");
  if (!sector_slot(e, first_sector + count - 1)) // sizes the table once, before the threads
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sectors_annular: bad sector range");
  const bool on_host = host_roi();
  const unsigned hw = std::thread::hardware_concurrency();
  const int workers = on_host ? std::max(1, std::min({count, 16, (int)(hw ? hw : 1)})) : 1;
  std::vector<int> empty((size_t)workers, 0);
  auto work = [&](int w) {
    for (int k = w; k < count; k += workers) { // interleaved: neighbouring rings have similar sizes
      const float *q = params + 6 * (size_t)k;
      HostSector &s = e->hs[(size_t)(first_sector + k)];
      s.xy.clear();
      s.is_rect = false;
      s.flats.clear();
      s.fresh = true;
      if (!lkroi::annular_geometry(q[0], q[1], q[2], q[3], q[4], q[5], as, s.ag)) {
        empty[(size_t)w] = 1;
        continue;
      }
      s.lazy = 1; // the device mask makes the list at commit; LK_HOST_ROI=1: the host scan, here
      if (on_host) {
        s.realize();
        if (s.xy.empty()) {
          empty[(size_t)w] = 1;
          continue;
        }
      }
      s.has_center = false;
      s.set = true;
    }
  };
  std::vector<std::thread> th;
  for (int w = 1; w < workers; ++w)
    th.emplace_back(work, w);
  work(0);
  for (std::thread &t : th)
    t.join();
  for (int v : empty)
    if (v)
      return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sectors_annular: empty sector");
  return LK_ERROR_NONE;
}

int lk_set_sector_blob(lk_engine *e, int sector, const float *contour_xy, int n_vertices) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  HostSector *s = sector_slot(e, sector);
  if (!s || !contour_xy)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sector_blob: bad arguments");
  s->xy.clear();
  s->is_rect = false;
  s->flats.clear();
  s->lazy = 0;
  if (host_roi()) { // round 1's path: ear clipping + scan fill on a few host threads
    if (!lkroi::BlobPolygon::inside_points(contour_xy, n_vertices, s->xy) || s->xy.empty()) {
      s->set = false;
      return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sector_blob: contour is not a simple polygon");
    }
    lkroi::mean_center(s->xy.data(), (int)(s->xy.size() / 2), s->cx, s->cy);
  } else { // the ear clipper stays on the host (O(vertices^2)); the fill is the device mask's, at commit
    if (!lkroi::BlobPolygon::flat_triangles(contour_xy, n_vertices, s->flats) || s->flats.empty()) {
      s->set = false;
      return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sector_blob: contour is not a simple polygon");
    }
    s->lazy = 2;
  }
  s->has_center = false;
  s->set = true;
  return LK_ERROR_NONE;
}

int lk_set_sector_points(lk_engine *e, int sector, const float *xy, int n, int use_center, float cx,
                         float cy) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  HostSector *s = sector_slot(e, sector);
  if (!s || !xy || n < 1)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_set_sector_points: bad arguments");
  s->xy.assign(xy, xy + 2 * (size_t)n);
  s->is_rect = false;
  if (use_center) {
    s->cx = cx;
    s->cy = cy;
  } else {
    lkroi::mean_center(xy, n, s->cx, s->cy);
  }
  s->has_center = use_center != 0;
  s->set = true;
  return LK_ERROR_NONE;
}

// Starved levels (at most 2P samples for P parameters; always the coarsest ones) are solved
// first by the one-lane-per-sector kernel, bit-identically to the reference.  Needs h_class,
// h_rect and h_off of the coarsest level.
// (the tuning hook LK_STARVED_MAX is read at every commit and kept with the engine: the host's class table and
// the kernel's hand-over rule must agree)
static int starved_max(const lk_engine *e) { return e->starved_max >= 0 ? e->starved_max : 2 * e->P; }
static int refresh_starved(lk_engine *e) {
  const int S = (int)e->h_class.size();
  {
    const char *f = std::getenv("LK_STARVED_MAX"); // tuning hook (tests/tools/c4_parity.py)
    e->starved_max = f ? std::atoi(f) : -1;
  }
  for (int c = 0; c < kNumClasses; ++c)
    e->class_starved[c] = false;
  if (const char *f = std::getenv("LK_FORCE_SAFE")) // tuning / test hook
    e->force_safe = std::atoi(f) != 0;
  const int l = e->cfg.py_stop; // counts only shrink with the level: test the coarsest
  bool any_starved = false;
  for (int s = 0; s < S; ++s) {
    const int4 r = e->h_rect[l][(size_t)s];
    const int n = r.z > 0 ? r.w : (int)(e->h_off[l][(size_t)s + 1] - e->h_off[l][(size_t)s]);
    if (n <= starved_max(e))
      e->class_starved[e->h_class[(size_t)s]] = any_starved = true;
  }
  if (any_starved) {
    HIPCHK(e->d_finish_list.ensure((size_t)S));
    HIPCHK(e->d_finish_count.ensure(kNumClasses));
  }
  return LK_ERROR_NONE;
}

static int level0_count(const lk_engine *e, int s) { // samples of a sector at level 0, from the committed tables
  const int4 r = e->h_rect[0][(size_t)s];
  return r.z > 0 ? r.w : (int)(e->h_off[0][(size_t)s + 1] - e->h_off[0][(size_t)s]);
}

// Device ROI masks: every sector of the domain is annular or a blob and registered by description.
// Replaces cudaPolygon{Annular,Blob}'s thrust rasterise + remove_if (cuda_polygon.cuh:180-292,
// cuda_polygon.cu:589-627) with the CPU engine's predicates and sample order.  Leaves d_xy / d_off
// of every level, d_center, h_off, h_center and zeroed rectangles behind; the lists stay on the device.
static int build_lists_roi_device(lk_engine *e, const std::vector<int> &levels) {
  Range range_("lk:roi masks");
  const int S = (int)e->hs.size();
  hipStream_t st = e->stream;
  std::vector<LkRoiSector> sec((size_t)S);
  std::vector<LkRoiFlat> flats;
  std::vector<uint32_t> tile_begin((size_t)S + 1, 0u);
  uint64_t tiles = 0;
  bool non_negative = true; // every coordinate the masks can produce is >= 0 and below 2^15 (lk_mean_center_int_kernel)
  for (int s = 0; s < S; ++s) {
    const HostSector &h = e->hs[(size_t)s];
    LkRoiSector &q = sec[(size_t)s];
    std::memset(&q, 0, sizeof(q));
    tile_begin[(size_t)s] = (uint32_t)tiles;
    if (h.lazy == 1) {
      const lkroi::AnnularGeometry &g = h.ag;
      q.kind = 0;
      q.x0 = g.x0, q.y0 = g.y0, q.x1 = g.x1, q.y1 = g.y1;
      q.cx = g.cx, q.cy = g.cy, q.ri2 = g.ri2, q.ro2 = g.ro2;
      q.q00x = g.q00x, q.q01x = g.q01x, q.q10x = g.q10x, q.q11x = g.q11x;
      q.q00y = g.q00y, q.q01y = g.q01y, q.q10y = g.q10y, q.q11y = g.q11y;
      q.as = g.as;
      const int64_t w = std::max(0, g.x1 - g.x0), hh = std::max(0, g.y1 - g.y0);
      if (w * hh >= (int64_t)1 << 31)
        return e->fail(LK_ERROR_BAD_DOMAIN, "lk_commit_sectors: annular sector bounding box too large");
      q.x1 = q.x0 + (int)w, q.y1 = q.y0 + (int)hh;
      // kept samples lie strictly inside the outer circle (and inside the bounding box)
      non_negative = non_negative && std::max((float)q.x0, g.cx - g.ro) >= 0.f && std::max((float)q.y0, g.cy - g.ro) >= 0.f &&
                     q.x1 < (1 << 15) && q.y1 < (1 << 15);
      tiles += (uint64_t)((w * hh + kLkRoiTile - 1) / kLkRoiTile);
    } else {
      q.kind = 1;
      q.flat_begin = (int)flats.size();
      q.flat_count = (int)h.flats.size();
      uint64_t rows = 0;
      for (const lkroi::BlobPolygon::Flat &f : h.flats) {
        LkRoiFlat d;
        d.ls = f.ls, d.li = f.li, d.rs = f.rs, d.ri = f.ri, d.j0 = f.j0, d.j1 = f.j1;
        d.row_begin = (int)rows;
        rows += (uint64_t)(f.j1 - f.j0);
        flats.push_back(d);
        // pixel range of its two end rows (the edges are straight: the extremes are there)
        for (int j : {f.j0, f.j1 - 1}) {
          const float a = std::ceil(f.ls * (float)j + f.li), b = std::ceil(f.rs * (float)j + f.ri);
          non_negative = non_negative && j >= 0 && j < (1 << 15) && a >= 0.f && b < 32768.f;
        }
      }
      tiles += rows;
    }
    if (tiles >= (uint64_t)1 << 31)
      return e->fail(LK_ERROR_BAD_DOMAIN, "lk_commit_sectors: region of interest too large for the device mask");
  }
  tile_begin[(size_t)S] = (uint32_t)tiles;
  const uint32_t n_tiles = (uint32_t)tiles;
  if (n_tiles == 0)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_commit_sectors: empty region of interest");
  HIPCHK(e->d_roi_sectors.ensure((size_t)S));
  HIPCHK(e->d_roi_flats.ensure(flats.size() + 1));
  HIPCHK(e->d_roi_tile_begin.ensure((size_t)S + 1));
  HIPCHK(hipMemcpyAsync(e->d_roi_sectors.p, sec.data(), (size_t)S * sizeof(LkRoiSector), hipMemcpyHostToDevice, st));
  if (!flats.empty())
    HIPCHK(hipMemcpyAsync(e->d_roi_flats.p, flats.data(), flats.size() * sizeof(LkRoiFlat), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(e->d_roi_tile_begin.p, tile_begin.data(), ((size_t)S + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  HIPCHK(e->d_level_total.ensure(2 * LK_MAX_LEVELS)); // (second half: scratch of the evaluation copy's passes)
  // (the tile table may be larger than the decimation's: both share d_tiles)
  HIPCHK(e->d_tiles.ensure(4 * (size_t)n_tiles + 2)); // (also enough for the decimation passes below in all but pathological blobs)
  HIPCHK(lk_launch_roi_count(e->d_roi_sectors.p, e->d_roi_flats.p, e->d_roi_tile_begin.p, S, n_tiles, e->d_tiles.p,
                             e->d_level_total.p, st));
  uint32_t total = 0;
  HIPCHK(hipMemcpyAsync(&total, e->d_level_total.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st)); // the list buffers are sized by the count
  if (total == 0)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_commit_sectors: empty region of interest");
  for (int l : levels) {
    HIPCHK(e->d_xy[l].ensure((size_t)total + 1));
    HIPCHK(e->d_off[l].ensure((size_t)S + 1));
    HIPCHK(e->d_rect[l].ensure((size_t)S));
    HIPCHK(hipMemsetAsync(e->d_rect[l].p, 0, (size_t)S * sizeof(int4), st));
    e->h_rect[l].assign((size_t)S, make_int4(0, 0, 0, 0));
    e->h_off[l].assign((size_t)S + 1, 0u);
  }
  HIPCHK(lk_launch_roi_fill(e->d_roi_sectors.p, e->d_roi_flats.p, e->d_roi_tile_begin.p, S, n_tiles, e->d_tiles.p,
                            e->d_xy[0].p, e->d_off[0].p, st));
  HIPCHK(hipMemcpyAsync(e->h_off[0].data(), e->d_off[0].p, ((size_t)S + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  // coarser levels: pyramid_class.cpp:289-323 for all sectors at once (the kernels of lk_rewarp_sectors)
  HIPCHK(e->d_pos.ensure((size_t)total + 1));
  HIPCHK(e->d_tiles.ensure((size_t)std::max<uint32_t>(n_tiles, (uint32_t)lk_decimate_tiles(total)) + 2));
  for (size_t li = 1; li < levels.size(); ++li) {
    const int l = levels[li], pl = levels[li - 1];
    HIPCHK(lk_launch_decimate(e->d_xy[pl].p, e->d_off[pl].p, e->d_level_total.p + pl, total, l - pl, S, e->d_pos.p,
                              e->d_tiles.p, e->d_xy[l].p, e->d_off[l].p, e->d_level_total.p + l, st));
    HIPCHK(hipMemcpyAsync(e->h_off[l].data(), e->d_off[l].p, ((size_t)S + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  }
  // centres: the float mean of the samples in list order (correlation_class.cpp:337-339)
  HIPCHK(e->d_center.ensure((size_t)S));
  e->h_center.resize(2 * (size_t)S);
  // (the mask's coordinates are integers; non-negative ones - the ROI lies in the image - let the
  // chain be evaluated in parallel, see lk_mean_center_int_kernel)
  if (non_negative) {
    HIPCHK(e->d_mean_scratch.ensure((lk_mean_center_int_scratch_bytes(total, S) + 3) / 4));
    HIPCHK(lk_launch_mean_center_int(e->d_xy[0].p, e->d_off[0].p, total, S, e->d_mean_scratch.p, e->d_center.p, st));
  }
  else
    HIPCHK(lk_launch_mean_center(e->d_xy[0].p, e->d_off[0].p, S, e->d_center.p, st));
  HIPCHK(hipMemcpyAsync(e->h_center.data(), e->d_center.p, 2 * (size_t)S * sizeof(float), hipMemcpyDeviceToHost, st));
  // The evaluation copy: the same masks walked row by row (only annular sectors come out in another order; a blob's
  // scan lines are rows already), then the same decimation.  Same sample sets, hence the same offsets at every level.
  e->eval_lists = false;
  bool any_annular = false;
  for (int s = 0; s < S; ++s)
    any_annular = any_annular || e->hs[(size_t)s].lazy == 1;
  const char *eval_flag = std::getenv("LK_EVAL_LISTS"); // test / tuning hook, read per commit
  const bool eval_env = eval_flag ? std::atoi(eval_flag) != 0 : true;
  if (any_annular && eval_env) {
    HIPCHK(e->d_off_eval.ensure((size_t)S + 1));
    uint32_t *totals = e->d_level_total.p + LK_MAX_LEVELS; // (scratch: the counts are the canonical lists')
    for (int l : levels)
      HIPCHK(e->d_xy_eval[l].ensure((size_t)total + 1));
    HIPCHK(e->d_tiles.ensure(4 * (size_t)n_tiles + 2));
    HIPCHK(lk_launch_roi_count(e->d_roi_sectors.p, e->d_roi_flats.p, e->d_roi_tile_begin.p, S, n_tiles, e->d_tiles.p, totals, st, 1));
    HIPCHK(lk_launch_roi_fill(e->d_roi_sectors.p, e->d_roi_flats.p, e->d_roi_tile_begin.p, S, n_tiles, e->d_tiles.p,
                              e->d_xy_eval[0].p, e->d_off_eval.p, st, 1));
    HIPCHK(e->d_tiles.ensure((size_t)std::max<uint32_t>(n_tiles, (uint32_t)lk_decimate_tiles(total)) + 2));
    for (size_t li = 1; li < levels.size(); ++li) {
      const int l = levels[li], pl = levels[li - 1];
      HIPCHK(lk_launch_decimate(e->d_xy_eval[pl].p, e->d_off[pl].p, e->d_level_total.p + pl, total, l - pl, S, e->d_pos.p,
                                e->d_tiles.p, e->d_xy_eval[l].p, e->d_off_eval.p, totals + l, st));
    }
    e->eval_lists = true;
  }
  HIPCHK(hipStreamSynchronize(st));
  for (int s = 0; s < S; ++s) {
    HostSector &h = e->hs[(size_t)s];
    h.n0_device = (int)(e->h_off[0][(size_t)s + 1] - e->h_off[0][(size_t)s]);
    if (h.n0_device == 0)
      return e->fail(LK_ERROR_BAD_DOMAIN, "lk_commit_sectors: sector " + std::to_string(s) + " is empty");
    h.cx = e->h_center[2 * (size_t)s];
    h.cy = e->h_center[2 * (size_t)s + 1];
  }
  e->lists_on_device = true;
  e->backup_on_device = false;
  return LK_ERROR_NONE;
}

// keep_state: a re-commit after the sample lists moved (Lagrangian descriptions) keeps the
// sequence state of the sectors (guess history, last results)
// Size classes -> lanes per sector, the launch order of the classes, starved-level flags and team widths of the
// committed domain (e->S sectors; needs h_rect / h_off).  Part of every full commit; after sectors were APPENDED
// (lk_commit_sectors' fast path) it is redone lazily by the next batch solve.
static int classify_sectors(lk_engine *e) {
  const int S = e->S;
  // size classes -> lanes per sector.  A class whose sectors are too few to fill the chip
  // (< 2048 wavefronts) and still large per lane is promoted to the next wider group.
  e->h_class.assign((size_t)S, 0);
  size_t cnt[kNumClasses] = {0, 0, 0, 0, 0, 0}, tot[kNumClasses] = {0, 0, 0, 0, 0, 0};
  for (int s = 0; s < S; ++s) {
    int n0 = level0_count(e, s), c = size_class(n0);
    e->h_class[(size_t)s] = c;
    cnt[c]++;
    tot[c] += (size_t)n0;
  }
  for (int c = 0; c + 1 < kTeamClass; ++c) { // (nothing is promoted into the team class)
    if (!cnt[c] || e->batch_invariant || e->reference_order > 0) // (the group depends on the sector alone)
      continue;
    // Wider groups shorten a sector's own critical path and cost lane packing: they pay only
    // while the narrow grouping leaves SIMDs without a wavefront (fewer wavefronts than the
    // 1024 SIMDs).  Measured on 19x19-sample sectors (scripts/quick_solve.py, LK_GRID /
    // LK_FORCE_GROUP): 1024 sectors 0.110 ms in 32-lane groups against 0.092 ms in 64-lane ones,
    // 2025 sectors 0.122 / 0.123, 4096 sectors 0.166 / 0.188, 8100 sectors 0.240 / 0.324.
    size_t waves = cnt[c] * (size_t)kGroupOfClass[c] / 64 * (size_t)e->pairs_in_flight; // (the other launches' too)
    size_t per_lane = tot[c] / cnt[c] / (size_t)kGroupOfClass[c];
    const bool small_group = kGroupOfClass[c] < 64;
    const size_t enough = kGroupOfClass[c] == 16 ? 4096 : 1024; // (16-lane groups: 10 000 sectors 0.31 against 0.27 ms)
    if ((small_group && waves < enough && per_lane >= 4) || (!small_group && waves < 2048 && per_lane >= 32)) {
      for (int s = 0; s < S; ++s)
        if (e->h_class[(size_t)s] == c)
          e->h_class[(size_t)s] = c + 1;
      cnt[c + 1] += cnt[c];
      tot[c + 1] += tot[c];
      cnt[c] = tot[c] = 0;
    }
  }
  if (const char *f = std::getenv("LK_FORCE_GROUP")) { // tuning experiments only
    int g = std::atoi(f);
    for (int c = 0; c < kNumClasses; ++c)
      if (kGroupOfClass[c] == g)
        std::fill(e->h_class.begin(), e->h_class.end(), c);
  }
  int force_team = 0;
  if (const char *f = std::getenv("LK_FORCE_TEAM")) { // test hook: every sector gets a team of this width
    force_team = std::atoi(f);
    if (force_team >= 1) // (1: the team class's kernel with a single workgroup per sector)
      std::fill(e->h_class.begin(), e->h_class.end(), kTeamClass);
  }
  e->team_w = 0;
  {
    // A handful of big sectors (one ROI in the GUI, BASELINE config 1) cannot fill the chip with
    // one workgroup each: when at most half of the CUs would be busy, every 8-wavefront sector
    // gets a team as well.  The kernel sizes each sector's team by its own sample count
    // (kTeamMinSamples per workgroup), the launch clamps the width to what is resident.
    int n_big = 0;
    for (int s = 0; s < S; ++s)
      n_big += e->h_class[(size_t)s] >= kTeamClass - 1;
    if (n_big > 0 && n_big <= kFewBigSectors && !e->batch_invariant && e->reference_order == 0)
      for (int s = 0; s < S; ++s)
        if (e->h_class[(size_t)s] == kTeamClass - 1)
          e->h_class[(size_t)s] = kTeamClass;
    int n_team = 0, n0_max = 0;
    for (int s = 0; s < S; ++s)
      if (e->h_class[(size_t)s] == kTeamClass) {
        ++n_team;
        n0_max = std::max(n0_max, level0_count(e, s));
      }
    if (n_team) {
      int w = force_team >= 1 ? force_team : (n0_max + kTeamMinSamples - 1) / kTeamMinSamples;
      int max_team = kMaxTeam;
      if (const char *f = std::getenv("LK_MAX_TEAM")) // tuning hook
        max_team = std::min(std::max(2, std::atoi(f)), kLkMaxTeam);
      e->team_w = force_team == 1 ? 1 : std::min(std::max(w, 2), max_team); // the launch clamps it to what is resident
      e->team_min_samples = force_team >= 1 ? 0 : kTeamMinSamples;
      HIPCHK(e->d_team_partials.ensure((size_t)n_team * 2 * (size_t)e->team_w * 32));
      HIPCHK(e->d_team_arrivals.ensure(2 * (size_t)n_team)); // arrival counters + broken flags
    }
  }
  e->h_order.clear();
  e->h_order.reserve((size_t)S);
  for (int c = 0; c < kNumClasses; ++c) {
    e->class_begin[c] = (int)e->h_order.size();
    for (int s = 0; s < S; ++s)
      if (e->h_class[(size_t)s] == c)
        e->h_order.push_back((uint32_t)s);
  }
  e->class_begin[kNumClasses] = (int)e->h_order.size();
  // A team launch needs all its workgroups resident at once, and the one-workgroup class (one 512-thread workgroup per
  // sector, as many as there are CUs) would hold every slot until its first sectors are done: the two launches then run
  // one after the other whatever the streams say (config 3: 256 annular sectors 0.42 ms + the blob's team 0.40 ms).
  // Instead they split the slots by their share of the samples: the team gets its share (at least a quarter), the
  // one-workgroup class a persistent grid on the rest with its largest sectors first in the queue.
  e->team_share_permille = 0;
  {
    const int cb = kTeamClass - 1;
    const int n_one = e->class_begin[cb + 1] - e->class_begin[cb], n_team = e->class_begin[kTeamClass + 1] - e->class_begin[kTeamClass];
    const char *f = std::getenv("LK_TEAM_SHARE"); // tuning hook: permille of the slots for the team class (0: no split)
    if (n_one > 0 && n_team > 0 && e->team_w > 1 && !e->batch_invariant && e->reference_order == 0 && !(f && std::atoi(f) == 0)) {
      double s_one = 0, s_team = 0;
      for (int s = 0; s < S; ++s) {
        if (e->h_class[(size_t)s] == cb)
          s_one += level0_count(e, s);
        else if (e->h_class[(size_t)s] == kTeamClass)
          s_team += level0_count(e, s);
      }
      // (its share of the samples.  Round 3 gave the team a quarter more for its all-to-all; with round 4's kernels config 3
      // measures 72 / 80 / 88 / 96 / 104 team workgroups -> 0.595 / 0.568 / 0.580 / 0.597 / 0.620 ms: 80 = the plain share)
      int share = f ? std::atoi(f) : (int)(1000.0 * s_team / (s_team + s_one) + 0.5);
      e->team_share_permille = std::min(std::max(share, 250), 750);
      std::stable_sort(e->h_order.begin() + e->class_begin[cb], e->h_order.begin() + e->class_begin[cb + 1],
                       [&](uint32_t x, uint32_t y) { return level0_count(e, (int)x) > level0_count(e, (int)y); });
    }
  }
  {
    int rc = refresh_starved(e);
    if (rc)
      return rc;
  }
  HIPCHK(e->d_order.ensure((size_t)S));
  HIPCHK(hipMemcpy(e->d_order.p, e->h_order.data(), (size_t)S * sizeof(uint32_t), hipMemcpyHostToDevice));
  e->classes_dirty = false;
  return LK_ERROR_NONE;
}

static int commit_impl(lk_engine *e, bool keep_state) {
  Range range_("lk:commit sectors");
  if (int rc = materialize_host(e)) // (a no-op after any edit: editors fetch the lists first)
    return rc;
  const int S = (int)e->hs.size();
  if (S == 0)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_commit_sectors: no sectors");
  for (int s = 0; s < S; ++s)
    if (!e->hs[(size_t)s].set)
      return e->fail(LK_ERROR_BAD_DOMAIN, "lk_commit_sectors: sector " + std::to_string(s) + " was never set");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream));
  const lk_config &cfg = e->cfg;
  const int first = cfg.py_start == 0 ? cfg.py_step : cfg.py_start; // pyramid_class.cpp:299
  // level lists: level 0 always; first, first+step, ... <= stop, each from the previous
  std::vector<int> levels{0};
  for (int l = first; l <= cfg.py_stop; l += cfg.py_step)
    levels.push_back(l);
  // Annular / blob sectors registered by description: the device mask makes every list (level 0 by
  // lk_roi_tile_kernel, the coarser levels by the order-preserving compaction, centres by
  // lk_mean_center_kernel); only offsets and centres come back.  A domain that mixes them with
  // rectangles or explicit lists, or LK_HOST_ROI=1, takes the host scans instead (same lists).
  bool any_lazy = false, all_lazy = true;
  for (int s = 0; s < S; ++s) {
    any_lazy = any_lazy || e->hs[(size_t)s].lazy != 0;
    all_lazy = all_lazy && e->hs[(size_t)s].lazy != 0;
  }
  const bool device_roi = any_lazy && all_lazy && !host_roi();
  if (any_lazy && !device_roi)
    for (int s = 0; s < S; ++s)
      e->hs[(size_t)s].realize();
  if (any_lazy && !device_roi)
    for (int s = 0; s < S; ++s)
      if (!e->hs[(size_t)s].is_rect && e->hs[(size_t)s].xy.empty())
        return e->fail(LK_ERROR_BAD_DOMAIN, "lk_commit_sectors: sector " + std::to_string(s) + " is empty");
  e->eval_lists = false; // (only the device ROI path below builds the evaluation copy)
  if (device_roi) {
    int rc = build_lists_roi_device(e, levels);
    if (rc)
      return rc;
  }
  std::vector<std::vector<float>> cat(LK_MAX_LEVELS);
  if (!device_roi) {
  for (int l = 0; l < LK_MAX_LEVELS; ++l)
    e->h_off[l].assign(1, 0u);
  size_t total0 = 0;
  for (int s = 0; s < S; ++s)
    total0 += e->hs[(size_t)s].xy.size();
  if (total0 / 2 >= 0xffffffffull)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_commit_sectors: more than 2^32 samples");
  cat[0].reserve(total0);
  e->h_center.resize(2 * (size_t)S);
  for (int l : levels)
    e->h_rect[l].assign((size_t)S, make_int4(0, 0, 0, 0));
  // Long explicit lists (annular / blob domains) are decimated on the device: one order-
  // preserving compaction per coarser level over all sectors at once (the kernels of
  // lk_rewarp_sectors); LK_HOST_REWARP=1 keeps the restated reference loop (tests compare).
  const char *host_env = std::getenv("LK_HOST_REWARP");
  const bool device_levels = !(host_env && std::atoi(host_env) != 0) && total0 / 2 >= 32768 && levels.size() > 1;
  std::vector<float> prev, cur;
  for (int s = 0; s < S; ++s) {
    const HostSector &hs = e->hs[(size_t)s];
    e->h_center[2 * (size_t)s] = hs.cx;
    e->h_center[2 * (size_t)s + 1] = hs.cy;
    if (hs.is_rect) {
      // the decimation rule of pyramid_class.cpp:301-322 applied to a full rectangle keeps
      // exactly the coordinates divisible by 2^l, i.e. another full rectangle, same order
      for (int l : levels) {
        int xs = ceil_shift(hs.x0, l), xe = floor_shift(hs.x1, l);
        int ys = ceil_shift(hs.y0, l), ye = floor_shift(hs.y1, l);
        int w = std::max(0, xe - xs + 1), h = std::max(0, ye - ys + 1);
        e->h_rect[l][(size_t)s] = make_int4(xs, ys, std::max(w, 1), w * h);
        e->h_off[l].push_back((uint32_t)(cat[l].size() / 2));
      }
      continue;
    }
    cat[0].insert(cat[0].end(), hs.xy.begin(), hs.xy.end());
    e->h_off[0].push_back((uint32_t)(cat[0].size() / 2));
    if (device_levels)
      continue;
    const float *pxy = hs.xy.data();
    int pn = (int)(hs.xy.size() / 2), plevel = 0;
    for (size_t li = 1; li < levels.size(); ++li) {
      int l = levels[li];
      cur.clear();
      int kept = lkroi::decimate(pxy, pn, l - plevel, cur);
      cat[l].insert(cat[l].end(), cur.begin(), cur.end());
      e->h_off[l].push_back((uint32_t)(cat[l].size() / 2));
      prev.swap(cur);
      pxy = prev.data();
      pn = kept;
      plevel = l;
    }
  }
  for (int l : levels) {
    const bool on_device = device_levels && l > 0;
    HIPCHK(e->d_xy[l].ensure((on_device ? cat[0].size() : cat[l].size()) / 2 + 1));
    HIPCHK(e->d_off[l].ensure((size_t)S + 1));
    if (!on_device) {
      if (!cat[l].empty())
        HIPCHK(hipMemcpy(e->d_xy[l].p, cat[l].data(), cat[l].size() * sizeof(float), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(e->d_off[l].p, e->h_off[l].data(), ((size_t)S + 1) * sizeof(uint32_t),
                       hipMemcpyHostToDevice));
    }
    HIPCHK(e->d_rect[l].ensure((size_t)S));
    HIPCHK(hipMemcpy(e->d_rect[l].p, e->h_rect[l].data(), (size_t)S * sizeof(int4), hipMemcpyHostToDevice));
  }
  if (device_levels) {
    const uint32_t total = (uint32_t)(cat[0].size() / 2);
    HIPCHK(e->d_level_total.ensure(2 * LK_MAX_LEVELS)); // (second half: scratch of the evaluation copy's passes)
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)e->d_level_total.p, (int)total, 1, e->stream));
    HIPCHK(e->d_pos.ensure((size_t)total + 1));
    HIPCHK(e->d_tiles.ensure((size_t)lk_decimate_tiles(total) + 2));
    for (size_t li = 1; li < levels.size(); ++li) {
      const int l = levels[li], pl = levels[li - 1];
      HIPCHK(lk_launch_decimate(e->d_xy[pl].p, e->d_off[pl].p, e->d_level_total.p + pl, total, l - pl, S, e->d_pos.p,
                                e->d_tiles.p, e->d_xy[l].p, e->d_off[l].p, e->d_level_total.p + l, e->stream));
      e->h_off[l].resize((size_t)S + 1);
      HIPCHK(hipMemcpyAsync(e->h_off[l].data(), e->d_off[l].p, ((size_t)S + 1) * sizeof(uint32_t),
                            hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  // The row-major evaluation copy for lists that came from the host (LkLevelView::xy_eval; any permutation of a sector's
  // samples is a valid walk for the unordered sums): a stable counting sort by image row of every sector whose list
  // jumps back in y often - lists in the reference's x outer / y inner order do so once per column.  Left alone: lists
  // that are row-major already, short sectors, LK_EVAL_LISTS=0.
  {
    const char *eval_flag = std::getenv("LK_EVAL_LISTS"); // test / tuning hook, read per commit
    const bool eval_on = !(eval_flag && std::atoi(eval_flag) == 0);
    std::vector<float> ecat[LK_MAX_LEVELS];
    bool any_sorted = false;
    std::vector<uint32_t> count;
    std::vector<float> tmp;
    auto sort_rows = [&](std::vector<float> &v, size_t b, size_t n) { // samples [b, b + n) of v: stable by (int)y
      if (n < 64)
        return false;
      int ymin = INT_MAX, ymax = INT_MIN;
      size_t back = 0;
      for (size_t k = 0; k < n; ++k) {
        const float y = v[2 * (b + k) + 1];
        if (!(y > -1e6f && y < 1e6f))
          return false; // (NaN / absurd coordinates: the solve reports them; nothing to gain here)
        const int r = (int)y;
        ymin = std::min(ymin, r), ymax = std::max(ymax, r);
        back += k > 0 && y < v[2 * (b + k) - 1];
      }
      if (back < 8 || back * 2048 < n) // (row-major already, or nearly: a column-major list jumps back once per column)
        return false;
      count.assign((size_t)(ymax - ymin) + 2, 0u);
      for (size_t k = 0; k < n; ++k)
        ++count[(size_t)((int)v[2 * (b + k) + 1] - ymin) + 1];
      for (size_t r = 1; r < count.size(); ++r)
        count[r] += count[r - 1];
      tmp.resize(2 * n);
      for (size_t k = 0; k < n; ++k) {
        const uint32_t at = count[(size_t)((int)v[2 * (b + k) + 1] - ymin)]++;
        tmp[2 * at] = v[2 * (b + k)], tmp[2 * at + 1] = v[2 * (b + k) + 1];
      }
      std::memcpy(&v[2 * b], tmp.data(), 2 * n * sizeof(float));
      return true;
    };
    if (eval_on && !cat[0].empty()) {
      for (int l : levels) {
        if (device_levels && l > 0)
          break;
        ecat[l] = cat[l];
        for (int s = 0; s < S; ++s)
          if (e->h_rect[l][(size_t)s].z == 0) {
            const size_t b = e->h_off[l][(size_t)s], n = e->h_off[l][(size_t)s + 1] - b;
            any_sorted = sort_rows(ecat[l], b, n) || any_sorted;
          }
      }
    }
    if (any_sorted) {
      for (int l : levels) {
        const bool on_device = device_levels && l > 0;
        HIPCHK(e->d_xy_eval[l].ensure((on_device ? cat[0].size() : cat[l].size()) / 2 + 1));
        if (!on_device && !ecat[l].empty())
          HIPCHK(hipMemcpy(e->d_xy_eval[l].p, ecat[l].data(), ecat[l].size() * sizeof(float), hipMemcpyHostToDevice));
      }
      if (device_levels) { // the same decimation of the copy: the same sets, the offsets already computed
        const uint32_t total = (uint32_t)(cat[0].size() / 2);
        HIPCHK(e->d_off_eval.ensure((size_t)S + 1));
        for (size_t li = 1; li < levels.size(); ++li) {
          const int l = levels[li], pl = levels[li - 1];
          HIPCHK(lk_launch_decimate(e->d_xy_eval[pl].p, e->d_off[pl].p, e->d_level_total.p + pl, total, l - pl, S, e->d_pos.p,
                                    e->d_tiles.p, e->d_xy_eval[l].p, e->d_off_eval.p, e->d_level_total.p + LK_MAX_LEVELS + l, e->stream));
        }
        HIPCHK(hipStreamSynchronize(e->stream));
      }
      e->eval_lists = true;
    }
  }
  e->lists_on_device = false;
  } // !device_roi
  HIPCHK(e->d_center.ensure((size_t)S));
  HIPCHK(hipMemcpy(e->d_center.p, e->h_center.data(), 2 * (size_t)S * sizeof(float), hipMemcpyHostToDevice));
  // Per-sector sequence state (guess history, last record, last evaluated parameters) survives a
  // commit for every sector that was not registered anew since the previous one: the reference's
  // manager registers and solves sector after sector on the first frame (resetPolygon(i), then
  // correlate(i): manager_class.cpp:340, :449) and moves them from their own records on the next
  // (updatePolygon: cuda_polygon.cu:268-415), so a commit that adds sector i must not forget 0..i-1.
  {
    const size_t old = (size_t)std::min(e->S, S);
    HIPCHK(e->d_guess.ensure_keep(6 * (size_t)S, 6 * old));
    HIPCHK(e->d_last_p.ensure_keep(6 * (size_t)S, 6 * old));
    HIPCHK(e->d_prev_p.ensure_keep(6 * (size_t)S, 6 * old));
    HIPCHK(e->d_result.ensure_keep((size_t)S, old));
    HIPCHK(e->d_stats.ensure_keep(4 * (size_t)S, 4 * old));
    HIPCHK(e->d_last_eval_p.ensure_keep(6 * (size_t)S, 6 * old));
    for (int s = 0; s < S;) { // zero the state of every run of fresh sectors
      const bool fresh = (size_t)s >= old || (!keep_state && e->hs[(size_t)s].fresh);
      int t = s + 1;
      if (fresh) {
        while (t < S && ((size_t)t >= old || (!keep_state && e->hs[(size_t)t].fresh)))
          ++t;
        const size_t n = (size_t)(t - s);
        HIPCHK(hipMemset(e->d_guess.p + 6 * (size_t)s, 0, 6 * n * sizeof(float)));
        HIPCHK(hipMemset(e->d_last_p.p + 6 * (size_t)s, 0, 6 * n * sizeof(float)));
        HIPCHK(hipMemset(e->d_prev_p.p + 6 * (size_t)s, 0, 6 * n * sizeof(float)));
        HIPCHK(hipMemset(e->d_last_eval_p.p + 6 * (size_t)s, 0, 6 * n * sizeof(float)));
        HIPCHK(hipMemset(e->d_stats.p + 4 * (size_t)s, 0, 4 * n * sizeof(uint32_t)));
        HIPCHK(hipMemset(e->d_result.p + (size_t)s, 0, n * sizeof(lk_result)));
      }
      s = t;
    }
    for (int s = 0; s < S; ++s)
      e->hs[(size_t)s].fresh = false;
  }
  e->S = S;
  {
    int rc = classify_sectors(e);
    if (rc)
      return rc;
  }
  HIPCHK(e->d_single.ensure(1));
  HIPCHK(e->d_queue.ensure(8 * kNumClasses));
  HIPCHK(hipMemset(e->d_queue.p, 0, 8 * kNumClasses * sizeof(uint32_t))); // (the SAFE pass rewinds its own queue)
  HIPCHK(e->d_handoff.ensure((size_t)S));
  HIPCHK(e->d_mid.ensure((size_t)S * kLkMidWords));
  HIPCHK(e->d_ill_list.ensure((size_t)S));
  HIPCHK(e->d_ill_count.ensure(kNumClasses)); // one list region and one counter per class: classes solve concurrently
  HIPCHK(hipMemset(e->d_ill_count.p, 0, kNumClasses * sizeof(uint32_t)));
  if (const char *f = std::getenv("LK_EVAL_CAP")) // tuning / test hook
    e->eval_cap = std::atoi(f);
  HIPCHK(e->d_scratch.ensure(64));
  e->S = S;
  e->committed = true;
  e->lv_dirty = true;
  e->stats_valid = false;
  // may the next commit append? (plain sectors only; every per-sector buffer of this commit is S long)
  e->append_ok = !any_lazy && !e->lists_on_device;
  return LK_ERROR_NONE;
}

// lk_commit_sectors, fast path: every sector committed so far is untouched and the new ones are plain rectangles
// registered behind them - the reference's first-frame loop, resetPolygon(i) then correlate(i), sector after
// sector (manager_class.cpp:304-460).  Appending costs O(levels) host work, amortised buffer growth and ONE small
// launch that writes the sector's rectangles, offsets, centre and zeroed sequence state (the values travel in the
// kernel arguments: no staging, no synchronisation); the full commit would rebuild and re-upload all S sectors,
// O(S^2) over the loop.
static const int kMaxAppend = 64; // more new sectors than this: the full commit is cheaper
static bool can_append(const lk_engine *e) {
  const int S0 = e->S, S1 = (int)e->hs.size();
  if (!e->append_ok || S0 <= 0 || S1 <= S0 || S1 - S0 > kMaxAppend || e->recommit_pending || e->lists_on_device ||
      e->backup_on_device || !e->hs_backup.empty())
    return false;
  for (int s = S0; s < S1; ++s) {
    const HostSector &h = e->hs[(size_t)s];
    if (!h.set || !h.is_rect || h.lazy != 0 || size_class(h.n0()) >= kTeamClass - 1)
      return false;
  }
  return true;
}
static int append_rect_sectors(lk_engine *e) {
  Range range_("lk:append sectors");
  const int S0 = e->S, S1 = (int)e->hs.size();
  HIPCHK(hipSetDevice(e->cfg.device));
  const lk_config &cfg = e->cfg;
  const int first = cfg.py_start == 0 ? cfg.py_step : cfg.py_start; // pyramid_class.cpp:299
  std::vector<int> levels{0};
  for (int l = first; l <= cfg.py_stop; l += cfg.py_step)
    levels.push_back(l);
  // room for the new sectors (geometric growth; the committed part is kept)
  const size_t keep = (size_t)S0, want = (size_t)S1;
  {
    // A growth step reallocates: hipMalloc, a copy on the NULL stream, hipFree - none of which waits for the engine's
    // (non-blocking) stream.  Append kernels, in-place patches or an unsynchronised solve (lk_correlate_all_device /
    // _async) still queued there would write the old buffer behind the copy, and the hipFree would drop what they wrote.
    // So the stream is drained first - only on the geometric growth steps, the O(1) path of the other commits is untouched.
    bool grows = e->d_center.n < want || e->d_guess.n < 6 * want || e->d_last_p.n < 6 * want || e->d_prev_p.n < 6 * want ||
                 e->d_last_eval_p.n < 6 * want || e->d_result.n < want || e->d_stats.n < 4 * want || e->d_handoff.n < want ||
                 e->d_mid.n < want * kLkMidWords || e->d_ill_list.n < want || e->d_order.n < want || e->d_finish_list.n < want;
    for (int l : levels)
      grows = grows || e->d_rect[l].n < want || e->d_off[l].n < want + 1;
    if (grows)
      HIPCHK(hipStreamSynchronize(e->stream));
  }
  // (a failure from here on leaves the host tables as they were and sends the next commit through the full rebuild)
  const size_t rect0 = e->h_rect[0].size();
  auto undo = [&, rect0]() {
    for (int l : levels) {
      if (e->h_rect[l].size() > rect0)
        e->h_rect[l].resize(rect0);
      if (e->h_off[l].size() > rect0 + 1)
        e->h_off[l].resize(rect0 + 1);
    }
    e->h_center.resize(2 * keep);
    e->h_class.resize(keep);
    e->append_ok = false;
  };
  struct Guard {
    std::function<void()> f;
    bool armed = true;
    ~Guard() {
      if (armed)
        f();
    }
  } guard{undo};
  HIPCHK(e->d_center.reserve_keep(want, keep));
  HIPCHK(e->d_guess.reserve_keep(6 * want, 6 * keep));
  HIPCHK(e->d_last_p.reserve_keep(6 * want, 6 * keep));
  HIPCHK(e->d_prev_p.reserve_keep(6 * want, 6 * keep));
  HIPCHK(e->d_last_eval_p.reserve_keep(6 * want, 6 * keep));
  HIPCHK(e->d_result.reserve_keep(want, keep));
  HIPCHK(e->d_stats.reserve_keep(4 * want, 4 * keep));
  for (int l : levels) {
    HIPCHK(e->d_rect[l].reserve_keep(want, keep));
    HIPCHK(e->d_off[l].reserve_keep(want + 1, keep + 1));
  }
  // (scratch of the solve: contents do not survive a solve)
  if (e->d_handoff.n < want)
    HIPCHK(e->d_handoff.ensure(2 * want + 64));
  if (e->d_mid.n < want * kLkMidWords)
    HIPCHK(e->d_mid.ensure((2 * want + 64) * kLkMidWords));
  if (e->d_ill_list.n < want)
    HIPCHK(e->d_ill_list.ensure(2 * want + 64));
  if (e->d_order.n < want)
    HIPCHK(e->d_order.ensure(2 * want + 64));
  e->h_center.resize(2 * want);
  e->h_class.resize(want, 0);
  bool any_starved = false;
  for (int s = S0; s < S1; ++s) {
    HostSector &hs = e->hs[(size_t)s];
    e->h_center[2 * (size_t)s] = hs.cx;
    e->h_center[2 * (size_t)s + 1] = hs.cy;
    LkAppendArgs a{};
    a.sector = s;
    a.center = make_float2(hs.cx, hs.cy);
    a.d_center = e->d_center.p;
    a.d_guess = e->d_guess.p, a.d_last_p = e->d_last_p.p, a.d_prev_p = e->d_prev_p.p, a.d_last_eval_p = e->d_last_eval_p.p;
    a.d_result = e->d_result.p;
    a.d_stats = e->d_stats.p;
    a.n_levels = (int)levels.size();
    for (size_t li = 0; li < levels.size(); ++li) {
      const int l = levels[li];
      // the decimation rule of pyramid_class.cpp:301-322 applied to a full rectangle: see commit_impl
      const int xs = ceil_shift(hs.x0, l), xe = floor_shift(hs.x1, l);
      const int ys = ceil_shift(hs.y0, l), ye = floor_shift(hs.y1, l);
      const int w = std::max(0, xe - xs + 1), h = std::max(0, ye - ys + 1);
      const int4 rc = make_int4(xs, ys, std::max(w, 1), w * h);
      e->h_rect[l].push_back(rc);
      e->h_off[l].push_back(e->h_off[l].back()); // (a rectangle has no list entries)
      a.rect[li] = rc;
      a.off_end[li] = e->h_off[l].back();
      a.d_rect[li] = e->d_rect[l].p;
      a.d_off[li] = e->d_off[l].p;
    }
    const int c = size_class(hs.n0()); // (promotions are a property of the whole batch: classify_sectors)
    e->h_class[(size_t)s] = c;
    const int4 top = e->h_rect[cfg.py_stop][(size_t)s];
    if (top.w <= starved_max(e))
      e->class_starved[c] = any_starved = true;
    HIPCHK(lk_launch_append_sector(a, e->stream));
    hs.fresh = false;
  }
  if (any_starved) {
    if (e->d_finish_list.n < want)
      HIPCHK(e->d_finish_list.ensure(2 * want + 64));
    HIPCHK(e->d_finish_count.ensure(kNumClasses));
  }
  guard.armed = false;
  e->S = S1;
  e->committed = true;
  e->classes_dirty = true; // the batch's size-class analysis and launch order: redone by the next batch solve
  e->lv_dirty = true;
  e->stats_valid = false;
  return LK_ERROR_NONE;
}

int lk_commit_sectors(lk_engine *e) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  int rc = materialize_host(e);
  if (rc)
    return rc;
  if (can_append(e))
    return append_rect_sectors(e);
  return commit_impl(e, false);
}

// Lagrangian description, CPU-engine semantics (manager_class.cpp:381-419): every sample of
// the sector moves by add_pair(offset), i.e. (int)(offset + v + 0.5f) per coordinate
// (manager_class.cpp:38-47).  A rectangle whose columns and rows all move by one integer
// stays an implicit rectangle; otherwise the sector becomes an explicit list.
static void translate_one(HostSector &h, float ox, float oy, const float *center) {
  if (h.is_rect) {
    const int sx = (int)lkroi::add_pair_round(ox, (float)h.x0) - h.x0;
    const int sy = (int)lkroi::add_pair_round(oy, (float)h.y0) - h.y0;
    bool uniform = true;
    for (int x = h.x0; x <= h.x1 && uniform; ++x)
      uniform = (int)lkroi::add_pair_round(ox, (float)x) - x == sx;
    for (int y = h.y0; y <= h.y1 && uniform; ++y)
      uniform = (int)lkroi::add_pair_round(oy, (float)y) - y == sy;
    if (uniform) {
      h.x0 += sx, h.x1 += sx, h.y0 += sy, h.y1 += sy;
    } else {
      h.xy = h.points();
      h.is_rect = false;
    }
  }
  if (!h.is_rect) {
    const size_t n = h.xy.size() / 2;
    for (size_t i = 0; i < n; ++i) {
      h.xy[2 * i] = lkroi::add_pair_round(ox, h.xy[2 * i]);
      h.xy[2 * i + 1] = lkroi::add_pair_round(oy, h.xy[2 * i + 1]);
    }
  }
  if (center) {
    h.cx = center[0];
    h.cy = center[1];
  } else { // Newton_Raphson(p, n, xy): float mean of the samples (pyramid_class.cpp:325-340)
    const std::vector<float> pts = h.points();
    lkroi::mean_center(pts.data(), (int)(pts.size() / 2), h.cx, h.cy);
  }
}

static void rewarp_one(HostSector &h, int model, const float *p, const float *center) {
  if (h.is_rect) {
    h.xy = h.points();
    h.is_rect = false;
  }
  const size_t n = h.xy.size() / 2;
  const float cx = h.cx, cy = h.cy;
  for (size_t i = 0; i < n; ++i) {
    float xd, yd;
    lkroi::warp_point(model, h.xy[2 * i], h.xy[2 * i + 1], cx, cy, p, xd, yd);
    h.xy[2 * i] = xd;
    h.xy[2 * i + 1] = yd;
  }
  if (center) {
    h.cx = center[0];
    h.cy = center[1];
  } else {
    lkroi::mean_center(h.xy.data(), (int)n, h.cx, h.cy);
  }
}

static int rewarp_on_device(lk_engine *e, const float *centers_xy, const float *offsets_xy = nullptr);

// A grid of implicit rectangles that only moved (Lagrangian description): sizes, classes, launch
// order and the (empty) explicit lists are what they were; only the per-level rectangles, the
// centres and - through the parity of the new positions - the starved-level flags change.
static int recommit_rects(lk_engine *e) {
  const int S = e->S;
  const lk_config &cfg = e->cfg;
  HIPCHK(hipSetDevice(cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream)); // the previous uploads from these host tables are done
  const int first = cfg.py_start == 0 ? cfg.py_step : cfg.py_start; // pyramid_class.cpp:299
  std::vector<int> levels{0};
  for (int l = first; l <= cfg.py_stop; l += cfg.py_step)
    levels.push_back(l);
  for (int s = 0; s < S; ++s) {
    const HostSector &hs = e->hs[(size_t)s];
    e->h_center[2 * (size_t)s] = hs.cx;
    e->h_center[2 * (size_t)s + 1] = hs.cy;
    for (int l : levels) { // (the rule of commit_impl)
      int xs = ceil_shift(hs.x0, l), xe = floor_shift(hs.x1, l);
      int ys = ceil_shift(hs.y0, l), ye = floor_shift(hs.y1, l);
      int w = std::max(0, xe - xs + 1), h = std::max(0, ye - ys + 1);
      e->h_rect[l][(size_t)s] = make_int4(xs, ys, std::max(w, 1), w * h);
    }
  }
  for (int l : levels)
    HIPCHK(hipMemcpyAsync(e->d_rect[l].p, e->h_rect[l].data(), (size_t)S * sizeof(int4), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_center.p, e->h_center.data(), 2 * (size_t)S * sizeof(float), hipMemcpyHostToDevice, e->stream));
  e->stats_valid = false;
  return refresh_starved(e);
}

int lk_translate_sectors(lk_engine *e, const float *offsets_xy, const float *centers_xy) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed || !offsets_xy)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_translate_sectors: sectors are not committed");
  {
    // Explicit lists (annular / blob / point sectors) move on the device like the strict
    // Lagrangian ones; implicit rectangles stay with the host records - a rectangle that moves
    // by whole pixels stays implicit, which is O(sectors) work and keeps its faster sampling.
    const char *f = std::getenv("LK_HOST_REWARP"); // test hook, read per call
    bool device_path = !(f && std::atoi(f) != 0) && !e->recommit_pending;
    for (int s = 0; s < e->S && device_path; ++s)
      device_path = e->h_rect[0][(size_t)s].z == 0;
    if (device_path) {
      HIPCHK(hipSetDevice(e->cfg.device));
      return rewarp_on_device(e, centers_xy, offsets_xy);
    }
  }
  {
    int rc = materialize_host(e);
    if (rc)
      return rc;
  }
  e->hs_backup.resize(e->hs.size());
  e->backup_on_device = false;
  // sectors are independent: blocks of them on a few threads when there are tens of thousands
  const int S = e->S;
  const unsigned hw = std::thread::hardware_concurrency();
  const int workers = std::max(1, std::min({S / 8192, 8, (int)(hw ? hw : 1)}));
  std::vector<char> all_rect((size_t)workers, 1);
  auto work = [&](int w) {
    bool rects = true;
    for (int s = (int)((int64_t)S * w / workers), last = (int)((int64_t)S * (w + 1) / workers); s < last; ++s) {
      HostSector &h = e->hs[(size_t)s];
      e->hs_backup[(size_t)s] = h;
      rects = rects && h.is_rect;
      translate_one(h, offsets_xy[2 * (size_t)s], offsets_xy[2 * (size_t)s + 1],
                    centers_xy ? centers_xy + 2 * (size_t)s : nullptr);
      rects = rects && h.is_rect;
    }
    all_rect[(size_t)w] = rects;
  };
  std::vector<std::thread> th;
  for (int w = 1; w < workers; ++w)
    th.emplace_back(work, w);
  work(0);
  for (std::thread &t : th)
    t.join();
  bool rects_only = true;
  for (char r : all_rect)
    rects_only = rects_only && r;
  return rects_only ? recommit_rects(e) : commit_impl(e, true);
}

// Strict Lagrangian description (manager_class.cpp:369-380): the deformed positions of the
// last solve - CorrelationClass::getDefXY0, i.e. the samples warped with the parameters of
// the LAST level-0 evaluation about the solve's centre (correlation_class.cpp:884-896) -
// become the undeformed samples of the next frame.
//
// Device path (default): the lists never leave the GPU.  One kernel warps every level-0 sample
// (implicit rectangles become explicit lists), one order-preserving compaction per coarser level
// applies the decimation rule to all sectors at once, a one-lane-per-sector kernel recomputes the
// mean centres where the caller gives none; the host reads back only the per-level offsets
// (starved-level bookkeeping).  LK_HOST_REWARP=1 keeps the host path (tests compare the two).
static int rewarp_on_device(lk_engine *e, const float *centers_xy, const float *offsets_xy) {
  Range range_("lk:rebuild moved lists");
  const int S = e->S;
  const lk_config &cfg = e->cfg;
  hipStream_t st = e->stream;
  const int first = cfg.py_start == 0 ? cfg.py_step : cfg.py_start; // pyramid_class.cpp:299
  std::vector<int> levels{0};
  for (int l = first; l <= cfg.py_stop; l += cfg.py_step)
    levels.push_back(l);
  std::vector<uint32_t> off0((size_t)S + 1, 0u);
  bool explicit_already = true;
  uint64_t sum = 0;
  for (int s = 0; s < S; ++s) {
    const int4 r = e->h_rect[0][(size_t)s];
    sum += r.z > 0 ? (uint32_t)r.w : e->h_off[0][(size_t)s + 1] - e->h_off[0][(size_t)s];
    explicit_already = explicit_already && r.z == 0;
    off0[(size_t)s + 1] = (uint32_t)sum;
  }
  if (sum >= 0xffffffffull)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_rewarp_sectors: more than 2^32 samples");
  const uint32_t total = (uint32_t)sum;
  HIPCHK(e->d_xy0_alt.ensure((size_t)total + 1));
  const uint32_t *dst_off = e->d_off[0].p;
  if (!explicit_already) { // (a blocking copy: the alternate table has no reader in flight)
    HIPCHK(e->d_off0_alt.ensure((size_t)S + 1));
    HIPCHK(hipMemcpy(e->d_off0_alt.p, off0.data(), ((size_t)S + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    dst_off = e->d_off0_alt.p;
  }
  LkRewarpArgs a{};
  a.src_xy = e->d_xy[0].p;
  a.src_off = e->d_off[0].p;
  a.src_rect = e->d_rect[0].p;
  a.dst_off = dst_off;
  a.dst_xy = e->d_xy0_alt.p;
  a.center = e->d_center.p;
  a.p = e->d_last_eval_p.p;
  a.n_sectors = S;
  a.total = total;
  if (offsets_xy) { // Lagrangian description: the sectors' offsets instead of the warp
    e->h_offsets.assign(offsets_xy, offsets_xy + 2 * (size_t)S);
    HIPCHK(e->d_offsets.ensure((size_t)S));
    HIPCHK(hipMemcpyAsync(e->d_offsets.p, e->h_offsets.data(), 2 * (size_t)S * sizeof(float), hipMemcpyHostToDevice, st));
    a.offset = e->d_offsets.p;
  }
  HIPCHK(lk_launch_rewarp(a, cfg.fitting_model, st));
  // The evaluation copy moves with the lists (sample by sample: it stays a permutation of them with the same offsets);
  // an implicit rectangle that becomes a list here gets one - its samples row by row, where the list proper has the
  // reference's x outer / y inner (a 19 x 19 sector: 32 consecutive samples on 19 image rows instead of 2).
  bool any_rect = false;
  for (int s = 0; s < S; ++s)
    any_rect = any_rect || e->h_rect[0][(size_t)s].z > 0;
  const char *eval_flag = std::getenv("LK_EVAL_LISTS"); // test / tuning hook, read per call
  const bool keep_eval = (e->eval_lists || any_rect) && !(eval_flag && std::atoi(eval_flag) == 0);
  if (keep_eval) {
    HIPCHK(e->d_xy_eval0_alt.ensure((size_t)total + 1));
    LkRewarpArgs b = a;
    b.src_xy = e->eval_lists ? e->d_xy_eval[0].p : e->d_xy[0].p; // (explicit sectors of a mixed domain without a copy: list order)
    b.dst_xy = e->d_xy_eval0_alt.p;
    b.rows = 1;
    HIPCHK(lk_launch_rewarp(b, cfg.fitting_model, st));
  }
  // what lk_restore_sectors needs: the previous lists stay in the alternate buffer when they
  // were device-built too, otherwise the host records are still good
  if (e->lists_on_device) {
    e->backup_on_device = true;
  } else {
    e->hs_backup = e->hs;
    e->backup_on_device = false;
  }
  e->h_center_prev = e->h_center;
  e->eval_lists = false; // (until its coarser levels are rebuilt below)
  if (keep_eval)
    std::swap(e->d_xy_eval[0], e->d_xy_eval0_alt);
  for (HostSector &h : e->hs) // (descriptions of annular / blob sectors no longer describe the moved lists)
    h.lazy = 0;
  std::swap(e->d_xy[0], e->d_xy0_alt);
  if (!explicit_already)
    std::swap(e->d_off[0], e->d_off0_alt);
  e->h_off[0] = off0;
  for (int l : levels) {
    HIPCHK(hipMemsetAsync(e->d_rect[l].p, 0, (size_t)S * sizeof(int4), st));
    e->h_rect[l].assign((size_t)S, make_int4(0, 0, 0, 0));
  }
  HIPCHK(e->d_level_total.ensure(2 * LK_MAX_LEVELS));
  HIPCHK(hipMemsetD32Async((hipDeviceptr_t)e->d_level_total.p, (int)total, 1, st));
  HIPCHK(e->d_pos.ensure((size_t)total + 1));
  HIPCHK(e->d_tiles.ensure((size_t)lk_decimate_tiles(total) + 2));
  for (size_t li = 1; li < levels.size(); ++li) {
    const int l = levels[li], pl = levels[li - 1];
    HIPCHK(e->d_xy[l].ensure((size_t)total + 1)); // any count up to the finer level's can be kept
    HIPCHK(lk_launch_decimate(e->d_xy[pl].p, e->d_off[pl].p, e->d_level_total.p + pl, total, l - pl, S, e->d_pos.p,
                              e->d_tiles.p, e->d_xy[l].p, e->d_off[l].p, e->d_level_total.p + l, st));
    e->h_off[l].resize((size_t)S + 1);
    HIPCHK(hipMemcpyAsync(e->h_off[l].data(), e->d_off[l].p, ((size_t)S + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost,
                          st));
  }
  if (keep_eval) { // the same decimation of the moved copy: the same sets, the offsets just computed
    HIPCHK(e->d_off_eval.ensure((size_t)S + 1));
    for (size_t li = 1; li < levels.size(); ++li) {
      const int l = levels[li], pl = levels[li - 1];
      HIPCHK(e->d_xy_eval[l].ensure((size_t)total + 1));
      HIPCHK(lk_launch_decimate(e->d_xy_eval[pl].p, e->d_off[pl].p, e->d_level_total.p + pl, total, l - pl, S, e->d_pos.p,
                                e->d_tiles.p, e->d_xy_eval[l].p, e->d_off_eval.p, e->d_level_total.p + LK_MAX_LEVELS + l, st));
    }
    e->eval_lists = true;
  }
  if (centers_xy) {
    e->h_center.assign(centers_xy, centers_xy + 2 * (size_t)S);
    HIPCHK(hipMemcpyAsync(e->d_center.p, e->h_center.data(), 2 * (size_t)S * sizeof(float), hipMemcpyHostToDevice, st));
  } else {
    HIPCHK(lk_launch_mean_center(e->d_xy[0].p, e->d_off[0].p, S, e->d_center.p, st));
    HIPCHK(hipMemcpyAsync(e->h_center.data(), e->d_center.p, 2 * (size_t)S * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  e->lists_on_device = true;
  e->lv_dirty = true;
  e->stats_valid = false;
  return refresh_starved(e);
}

int lk_rewarp_sectors(lk_engine *e, const float *centers_xy) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_rewarp_sectors: sectors are not committed");
  HIPCHK(hipSetDevice(e->cfg.device));
  const char *f = std::getenv("LK_HOST_REWARP"); // test hook, read per call
  const bool host_path = f && std::atoi(f) != 0;
  if (!host_path && !e->recommit_pending)
    return rewarp_on_device(e, centers_xy);
  {
    int rc = materialize_host(e);
    if (rc)
      return rc;
  }
  std::vector<float> ev(6 * (size_t)e->S);
  HIPCHK(hipMemcpyAsync(ev.data(), e->d_last_eval_p.p, ev.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->hs_backup = e->hs;
  e->backup_on_device = false;
  for (int s = 0; s < e->S; ++s)
    rewarp_one(e->hs[(size_t)s], e->cfg.fitting_model, &ev[6 * (size_t)s],
               centers_xy ? centers_xy + 2 * (size_t)s : nullptr);
  return commit_impl(e, true);
}

// cudaPolygon::updatePolygon's call shape (cuda_class.cu:569, one sector, the engine's own
// last record of it) with the CPU engine's meaning: the sector moves to where update_results
// (manager_class.cpp:2406-2411) puts its centre, c + (u, v).  The re-commit is deferred to
// the next solve.  mode: deformationDescriptionEnum (0 strict Lagrangian, 1 Lagrangian, 2 Eulerian).
int lk_update_sector(lk_engine *e, int sector, int mode) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed || sector < 0 || sector >= e->S || mode < 0 || mode > 2)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_update_sector: unknown sector / mode");
  if (mode == 2)
    return LK_ERROR_NONE;
  HIPCHK(hipSetDevice(e->cfg.device));
  {
    int rc = materialize_host(e);
    if (rc)
      return rc;
  }
  lk_result r;
  float ev[6];
  HIPCHK(hipMemcpyAsync(&r, e->d_result.p + sector, sizeof(r), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(ev, e->d_last_eval_p.p + 6 * (size_t)sector, sizeof(ev), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  HostSector &h = e->hs[(size_t)sector];
  const float *m = r.resultingParameters;
  const float def_cx = h.cx + m[0], def_cy = e->P > 1 ? h.cy + m[1] : h.cy; // interpolation_class.cpp:3-43
  const float center[2] = {(float)(int)(def_cx + 0.5f), (float)(int)(def_cy + 0.5f)}; // manager_class.cpp:2087-2088
  const bool was_rect = h.is_rect;
  if (mode == 1)
    translate_one(h, def_cx - h.cx, def_cy - h.cy, h.has_center ? center : nullptr);
  else
    rewarp_one(h, e->cfg.fitting_model, ev, h.has_center ? center : nullptr);
  // A rectangle that moved by whole pixels is still an implicit rectangle of the same size: its records are
  // patched in place (O(levels) host work, one small launch; sequence state untouched), so that the reference's
  // per-sector loop - updatePolygon(i), correlate(i) for every sector (cuda_class.cu:569, manager_class.cpp:449) -
  // stays O(S).  Anything else (a sector that became a list) rebuilds everything before the next solve.
  if (was_rect && h.is_rect && e->append_ok && !e->recommit_pending && !e->lists_on_device) {
    const lk_config &cfg = e->cfg;
    const int first = cfg.py_start == 0 ? cfg.py_step : cfg.py_start;
    LkAppendArgs a{};
    a.sector = sector;
    a.keep_state = 1;
    a.center = make_float2(h.cx, h.cy);
    a.d_center = e->d_center.p;
    e->h_center[2 * (size_t)sector] = h.cx;
    e->h_center[2 * (size_t)sector + 1] = h.cy;
    int li = 0;
    for (int l = 0; l <= cfg.py_stop; l = (l == 0 ? first : l + cfg.py_step), ++li) {
      const int xs = ceil_shift(h.x0, l), xe = floor_shift(h.x1, l);
      const int ys = ceil_shift(h.y0, l), ye = floor_shift(h.y1, l);
      const int w = std::max(0, xe - xs + 1), hh = std::max(0, ye - ys + 1);
      const int4 rc = make_int4(xs, ys, std::max(w, 1), w * hh);
      e->h_rect[l][(size_t)sector] = rc;
      a.rect[li] = rc;
      a.d_rect[li] = e->d_rect[l].p;
      if (l == cfg.py_stop && rc.w <= starved_max(e) && !e->class_starved[e->h_class[(size_t)sector]]) {
        e->class_starved[e->h_class[(size_t)sector]] = true;
        HIPCHK(e->d_finish_list.ensure((size_t)std::max<size_t>(e->d_order.n, (size_t)e->S)));
        HIPCHK(e->d_finish_count.ensure(kNumClasses));
      }
    }
    a.n_levels = li;
    HIPCHK(lk_launch_append_sector(a, e->stream));
    return LK_ERROR_NONE;
  }
  e->recommit_pending = true;
  return LK_ERROR_NONE;
}

// Sectors the manager's loop never reached on a frame that stopped at an error
// (manager_class.cpp:520-546) keep the samples of the previous frame.
int lk_restore_sectors(lk_engine *e, int first_sector) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed || first_sector < 0)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_restore_sectors: nothing to restore");
  {
    int rc = materialize_host(e);
    if (!rc && e->backup_on_device) {
      e->hs_backup = e->hs;
      rc = lists_from_device(e, e->d_xy0_alt.p, e->h_center_prev, e->hs_backup);
      e->backup_on_device = false;
    }
    if (rc)
      return rc;
  }
  if (e->hs_backup.size() != e->hs.size())
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_restore_sectors: nothing to restore");
  for (size_t s = (size_t)first_sector; s < e->hs.size(); ++s)
    e->hs[s] = e->hs_backup[s];
  return commit_impl(e, true);
}

int lk_get_last_evaluated_parameters(lk_engine *e, float *out) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed || !out)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_last_evaluated_parameters: sectors are not committed");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipMemcpyAsync(out, e->d_last_eval_p.p, 6 * (size_t)e->S * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return LK_ERROR_NONE;
}

int lk_sector_count(const lk_engine *e) { return e ? (e->committed ? e->S : (int)e->hs.size()) : 0; }

int lk_get_sector_info(lk_engine *e, int sector, int *n_points, float *cx, float *cy) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (sector < 0 || (size_t)sector >= e->hs.size() || !e->hs[(size_t)sector].set)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_sector_info: unknown sector");
  if (e->lists_on_device && e->committed) { // device-built lists: counts and centres are on the host already
    const size_t q = (size_t)sector;
    const int4 r0 = e->h_rect[0][q];
    if (n_points)
      *n_points = r0.z > 0 ? r0.w : (int)(e->h_off[0][q + 1] - e->h_off[0][q]);
    if (cx)
      *cx = e->h_center[2 * q];
    if (cy)
      *cy = e->h_center[2 * q + 1];
    return LK_ERROR_NONE;
  }
  if (int rc = materialize_host(e))
    return rc;
  e->hs[(size_t)sector].realize();
  const HostSector &s = e->hs[(size_t)sector];
  if (n_points)
    *n_points = s.n0();
  if (cx)
    *cx = s.cx;
  if (cy)
    *cy = s.cy;
  return LK_ERROR_NONE;
}

int lk_get_sector_level_count(lk_engine *e, int sector, int level, int *n) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed || sector < 0 || sector >= e->S || level < 0 || level >= LK_MAX_LEVELS ||
      e->h_off[level].size() != (size_t)e->S + 1)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_sector_level_count: unknown sector/level");
  if (n) {
    const int4 r = e->h_rect[level][(size_t)sector];
    *n = r.z > 0 ? r.w : (int)(e->h_off[level][(size_t)sector + 1] - e->h_off[level][(size_t)sector]);
  }
  return LK_ERROR_NONE;
}

int lk_get_level_xy(lk_engine *e, int level, int which, int sector, float *xy, int cap, int *count) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed || sector < 0 || sector >= e->S || level < 0 || level >= LK_MAX_LEVELS ||
      e->h_off[level].size() != (size_t)e->S + 1)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_level_xy: unknown sector/level");
  if (which != 0 && !e->eval_lists)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_level_xy: the domain has no evaluation copy of its lists");
  const uint32_t b = e->h_off[level][(size_t)sector];
  const uint32_t n = e->h_rect[level][(size_t)sector].z > 0 ? 0u : e->h_off[level][(size_t)sector + 1] - b;
  if (count)
    *count = (int)n;
  if (xy && cap > 0 && n > 0) {
    const float2 *src = which ? e->d_xy_eval[level].p : e->d_xy[level].p;
    if (!src)
      return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_level_xy: the lists of this level are not on the device");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(xy, src + b, sizeof(float2) * (size_t)std::min((int)n, cap), hipMemcpyDeviceToHost));
  }
  return LK_ERROR_NONE;
}

int lk_get_und_xy(lk_engine *e, int sector, float *xy, int cap, int *count) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (sector < 0 || (size_t)sector >= e->hs.size() || !e->hs[(size_t)sector].set)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_und_xy: unknown sector");
  if (e->lists_on_device && e->committed && e->h_rect[0][(size_t)sector].z == 0) { // that sector's list only
    const uint32_t b = e->h_off[0][(size_t)sector], n = e->h_off[0][(size_t)sector + 1] - b;
    if (count)
      *count = (int)n;
    if (xy && cap > 0) {
      HIPCHK(hipSetDevice(e->cfg.device));
      HIPCHK(hipStreamSynchronize(e->stream));
      HIPCHK(hipMemcpy(xy, e->d_xy[0].p + b, sizeof(float2) * (size_t)std::min((int)n, cap), hipMemcpyDeviceToHost));
    }
    return LK_ERROR_NONE;
  }
  if (int rc = materialize_host(e))
    return rc;
  e->hs[(size_t)sector].realize();
  const HostSector &s = e->hs[(size_t)sector];
  int n = s.n0();
  if (count)
    *count = n;
  if (xy && cap > 0) {
    std::vector<float> pts = s.points();
    std::memcpy(xy, pts.data(), 2 * sizeof(float) * (size_t)std::min(n, cap));
  }
  return LK_ERROR_NONE;
}

int lk_get_def_xy(lk_engine *e, int sector, const float *p, float *xy, int cap, int *count) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed || sector < 0 || sector >= e->S || !p)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_def_xy: unknown sector");
  if (int rc = materialize_host(e))
    return rc;
  const std::vector<float> pts = e->hs[(size_t)sector].points();
  const int n = (int)(pts.size() / 2);
  if (count)
    *count = n;
  if (!xy || cap <= 0)
    return LK_ERROR_NONE;
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(e->d_warp.ensure(2 * (size_t)n));
  HIPCHK(hipMemcpyAsync(e->d_warp.p + n, pts.data(), 2 * sizeof(float) * (size_t)n, hipMemcpyHostToDevice,
                        e->stream));
  float pp[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < e->P; ++i)
    pp[i] = p[i];
  HIPCHK(hipMemcpyAsync(e->d_scratch.p, pp, sizeof(pp), hipMemcpyHostToDevice, e->stream));
  HIPCHK(lk_launch_warp_points(e->d_warp.p + n, n, e->h_center[2 * (size_t)sector],
                               e->h_center[2 * (size_t)sector + 1], e->cfg.fitting_model, e->d_scratch.p,
                               e->d_warp.p, e->stream));
  HIPCHK(hipMemcpyAsync(xy, e->d_warp.p, 2 * sizeof(float) * (size_t)std::min(n, cap),
                        hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return LK_ERROR_NONE;
}

// ------------------------------------------------------------------------------------
// solve
// ------------------------------------------------------------------------------------
static int refresh_level_views(lk_engine *e) {
  if (!e->committed)
    return e->fail(LK_ERROR_BAD_DOMAIN, "sectors are not committed (call lk_commit_sectors)");
  if (e->recommit_pending) {
    e->recommit_pending = false;
    int rc = commit_impl(e, true);
    if (rc)
      return rc;
  }
  const DevImage &u = e->img[LK_IMG_UND], &d = e->img[LK_IMG_DEF];
  if (!u.valid || !d.valid)
    return e->fail(LK_ERROR_BAD_DOMAIN, "undeformed and deformed images must be set before correlating");
  if (!e->lv_dirty)
    return LK_ERROR_NONE;
  // every level-0 sample must lie inside the undeformed image
  for (int l = 0; l < LK_MAX_LEVELS; ++l) {
    LkLevelView &v = e->h_lv[l];
    std::memset(&v, 0, sizeof(v));
    if (l > e->cfg.py_stop)
      continue;
    v.und = u.lvl[l];
    v.def = d.lvl[l];
    v.xy = e->d_xy[l].p;
    // (batch-invariant mode: a sector's arithmetic must not depend on HOW its list was built - device masks and moved
    // lists have the row-major copy, lists that came from the host do not - so every path walks the reference's order)
    v.xy_eval = e->eval_lists && !e->batch_invariant ? e->d_xy_eval[l].p : nullptr;
    v.off = e->d_off[l].p;
    v.rect = e->d_rect[l].p;
    v.urows = u.rows >> l;
    v.ucols = u.cols >> l;
    v.drows = d.rows >> l;
    v.dcols = d.cols >> l;
  }
  // Same buffers and geometry as the last launch (a frame loop that reuses its slots): nothing
  // to do.  Otherwise the table travels in the kernel arguments of a one-wavefront kernel:
  // stream-ordered, no staging buffer, no host synchronisation between the frames.
  if (!e->d_lv.p || std::memcmp(e->h_lv, e->h_lv_sent, sizeof(e->h_lv)) != 0) {
    HIPCHK(e->d_lv.ensure(LK_MAX_LEVELS));
    HIPCHK(lk_launch_set_views(e->h_lv, e->d_lv.p, e->stream));
    std::memcpy(e->h_lv_sent, e->h_lv, sizeof(e->h_lv));
  }
  e->lv_dirty = false;
  return LK_ERROR_NONE;
}

// Which flavour of the lane-group kernels a solve uses.  The fast flavour hands a sector whose damped system met a bad pivot
// to a second, SAFE pass; the SAFE flavour runs the reference's QR for such a system inside the kernel.  Batch-invariant mode
// takes the SAFE flavour throughout: a sector's arithmetic then is ONE kernel's from its guess to its record, whatever else
// is in the batch and whoever solves it - in particular the frame-pipelined instances (which have no second pass to hand a
// sector to) produce the very same bytes as the one-pair launches.
static bool safe_flavour(const lk_engine *e) { return e->force_safe || e->batch_invariant; }

static LkSolveArgs base_args(lk_engine *e, const float *d_guess, lk_result *d_result) {
  LkSolveArgs a{};
  a.lv = e->d_lv.p;
  a.center = e->d_center.p;
  a.guess = d_guess;
  a.result = d_result;
  a.last_p = e->d_last_p.p;
  a.last_eval_p = e->d_last_eval_p.p;
  a.stats = e->d_stats.p;
  a.py_start = e->cfg.py_start;
  a.py_step = e->cfg.py_step;
  a.py_stop = e->cfg.py_stop;
  a.precision = e->cfg.precision;
  a.max_iters = e->cfg.max_iters;
  a.solo = e->batch_invariant ? 0 : 1;
  a.gpu_share = e->pairs_in_flight;
  static const int align = [] { // tuning hook; alignment never changes a record's bits
    const char *f = std::getenv("LK_ALIGN");
    return f ? std::atoi(f) : 1;
  }();
  a.align = align;
  const char *ks = std::getenv("LK_KEEP_SUMS"); // test hook, read per call
  a.keep_sums = ks ? (std::atoi(ks) != 0 ? 1 : 0) : 1;
  a.starved_max = starved_max(e);
  return a;
}

// Starved levels: the one-lane-per-sector kernel (bit-identical sums and QR), whose lanes park
// a sector after eval_cap evaluations, then the 16-lane finisher for the parked ones.  Both
// leave an LkHandoff record per sector for the lane-group kernel that follows.
// `c` picks the class's own region of the parked-sector lists and its own counters (classes
// solve concurrently), `first` is where the class starts in d_order, `st` its stream.
static int launch_starved(lk_engine *e, LkSolveArgs &a, int c, int first, hipStream_t st) {
  a.handoff = e->d_handoff.p;
  a.eval_cap = e->eval_cap > 0 ? e->eval_cap : 0;
  a.mid_state = e->d_mid.p;
  a.finish_list = e->d_finish_list.p + first;
  a.finish_count = e->d_finish_count.p + c;
  if (a.eval_cap > 0)
    HIPCHK(hipMemsetAsync(a.finish_count, 0, sizeof(uint32_t), st));
  HIPCHK(lk_launch_solve(a, e->cfg.fitting_model, e->cfg.interpolation, 1, st));
  if (a.eval_cap > 0) {
    LkSolveArgs f = a;
    f.finisher = 1;
    f.safe = 1;
    f.align = 0; // one trip per evaluation whatever the level: rows take the next parked sector as they finish
    HIPCHK(lk_launch_solve(f, e->cfg.fitting_model, e->cfg.interpolation, 16, st));
  }
  a.eval_cap = 0;
  return LK_ERROR_NONE;
}

// The lane-group kernel of one class, then the SAFE 16-lane kernel for the sectors it parked
// because a damped system met a bad pivot (the reference's rank-revealing QR decides those
// steps; on textured images the list is empty and the second launch retires at once).
static int launch_groups(lk_engine *e, LkSolveArgs &a, int group, int c, int first, hipStream_t st) {
  static const bool ill_env = [] { const char *f = std::getenv("LK_ILL_PASS"); return f ? std::atoi(f) != 0 : true; }();
  const bool with_ill_pass = !a.safe && ill_env;
  if (with_ill_pass) {
    a.mid_state = e->d_mid.p;
    a.ill_list = e->d_ill_list.p + first;
    a.ill_count = e->d_ill_count.p + c; // (rewound by the SAFE pass itself)
  }
  HIPCHK(lk_launch_solve(a, e->cfg.fitting_model, e->cfg.interpolation, group, st));
  if (with_ill_pass) {
    LkSolveArgs r = a;
    r.resume = 1;
    r.safe = 1;
    r.ill_list = nullptr;
    r.ill_count = nullptr;
    r.finish_list = a.ill_list;
    r.finish_count = a.ill_count;
    r.team_w = 0;
    r.handoff = nullptr;
    r.queue = a.queue + 3;
    HIPCHK(lk_launch_solve(r, e->cfg.fitting_model, e->cfg.interpolation, 16, st));
  }
  return LK_ERROR_NONE;
}

// Team workgroups wait for each other, so two team launches that are both only partly resident
// would wait forever.  One engine never has two in flight; engines of one process take turns:
// every team launch waits for the previous one on the same device (whoever issued it).
static int team_fault_hook() { // LK_TEAM_FAULT=step (tests): a team workgroup goes missing -> the lone-workgroup fallback
  const char *f = std::getenv("LK_TEAM_FAULT");
  return f ? std::atoi(f) : 0;
}
static std::mutex g_team_mu;
static hipEvent_t g_team_done[64] = {};

// reference-order mode: records whose first evaluation failed report the iteration count the
// previously solved sector left behind (lk_stale_iterations_kernel)
static int stale_iterations(lk_engine *e, lk_result *d_result, int n) {
  if (!e->d_stale.p) {
    HIPCHK(e->d_stale.ensure(2));
    HIPCHK(hipMemsetAsync(e->d_stale.p, 0, 2 * sizeof(int), e->stream));
  }
  HIPCHK(lk_launch_stale_iterations(d_result, n, e->d_stale.p + e->stale_par, e->d_stale.p + (e->stale_par ^ 1), e->stream));
  e->stale_par ^= 1;
  return LK_ERROR_NONE;
}

static int launch_all(lk_engine *e, const float *d_guess, lk_result *d_result) {
  if (e->classes_dirty) { // sectors were appended since the last analysis of the whole domain
    HIPCHK(hipStreamSynchronize(e->stream)); // (earlier launches may still read the order table)
    int rc = classify_sectors(e);
    if (rc)
      return rc;
  }
  Range range_("lk:solve");
  if (e->timing)
    HIPCHK(hipEventRecord(e->ev_s0, e->stream));
  // Size classes are independent sector sets (own lists, counters and queue words): the first
  // one solves on the engine's stream, every further one on a stream of its own forked from it
  // and joined back, so that the tail of one class overlaps the others (config 3: 256 annular
  // sectors + the blob's team).
  static const bool overlap = [] { const char *f = std::getenv("LK_CLASS_STREAMS"); return f ? std::atoi(f) != 0 : true; }();
  int n_classes = 0, n_launched = 0;
  for (int c = 0; c < kNumClasses; ++c)
    n_classes += e->class_begin[c + 1] > e->class_begin[c];
  if (n_classes > 1 && overlap) { // guesses, views and images are ready at this point of the engine's stream
    if (!e->ev_fork)
      HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e->ev_fork, e->stream));
  }
  for (int c = 0; c < kNumClasses; ++c) {
    int n = e->class_begin[c + 1] - e->class_begin[c];
    if (n <= 0)
      continue;
    hipStream_t st = e->stream;
    if (n_launched > 0 && overlap) {
      if (!e->class_stream[c]) {
        HIPCHK(hipStreamCreateWithFlags(&e->class_stream[c], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&e->ev_join[c], hipEventDisableTiming));
      }
      st = e->class_stream[c];
      HIPCHK(hipStreamWaitEvent(st, e->ev_fork, 0));
    }
    LkSolveArgs a = base_args(e, d_guess, d_result);
    a.order = e->d_order.p + e->class_begin[c];
    a.n_sectors = n;
    a.queue = e->d_queue.p + 8 * c;
    a.safe = safe_flavour(e) ? 1 : 0;
    if (e->reference_order > 0) {
      // Reference-order mode: ONE launch per class, every level with the ordered sums and the
      // restated QR - a 16-lane row per small sector (four sectors share a wavefront's QR),
      // a wavefront per larger one.  No starved-level kernel, finisher, SAFE pass or team: the
      // ordered kernel is all of them.
      a.safe = 1;
      a.solo = 0;
      // Starved levels (at most 2P samples) first, by the one-lane kernel and its finisher: their sums and their QR
      // are the sequential reference's by construction, 64 sectors share one instruction stream of the QR instead of
      // four, and the records stay the same bytes (config 5: the ordered kernel then starts at level 2).  Only for
      // thread counts that leave such a level's summation sequential: T = 1, or T >= the level's sample count (every
      // chunk is one sample then, correlation_class.cpp:169-186).
      const char *chain_env = std::getenv("LK_REF_STARVED_CHAIN"); // test hook, read per call: 0 never, 2 whatever the sector count
      const int ref_chain = chain_env ? std::atoi(chain_env) : 1;
      // ... and only where the one-lane kernel has the wavefronts to fill the chip (64 sectors each): config 5's
      // 199 809 sectors 10.0 -> 8.6 ms; config 4's 50 176 would be 784 wavefronts waiting on their own QR chains
      // (2.12 -> 2.18 ms), so they stay with the ordered kernel.
      if (ref_chain != 0 && e->class_starved[c] && e->eval_cap > 0 && (n >= 131072 || ref_chain == 2) &&
          (e->reference_order == 1 || e->reference_order >= starved_max(e))) {
        a.reference_order = 0; // (the starved-level instances; `mark_stale` keeps the stale-iteration markers coming)
        a.mark_stale = 1;
        int rc = launch_starved(e, a, c, e->class_begin[c], st);
        if (rc)
          return rc;
      }
      a.mark_stale = 1;
      a.reference_order = e->reference_order;
      // (a 16-lane row per small sector, a wavefront per larger one, a 512-thread workgroup per sector of the two big
      // classes - and of the class below them while its sectors are no more than the CUs: a workgroup per sector then
      // shortens every sector's chain, beyond that the wavefronts' throughput wins)
      const int ref_group = c == 0 ? 16 : (c >= kTeamClass - 1 || (c == kTeamClass - 2 && n <= 256)) ? 512 : 64;
      // number_of_threads = T > 1 on the big classes: a TEAM of T workgroups per sector, one thread chunk of the reference
      // each (the launch falls back to one workgroup per sector when T x sectors are not all resident at once)
      std::unique_lock<std::mutex> ref_team_turn;
      hipEvent_t *ref_team_done = nullptr;
      if (c >= kTeamClass - 1 && e->reference_order > 1 && e->reference_order <= kLkMaxTeam && (long long)n * e->reference_order <= 256 &&
          !e->batch_invariant) {
        const int T = e->reference_order;
        HIPCHK(e->d_team_partials.ensure((size_t)n * 2 * (size_t)T * 32));
        HIPCHK(e->d_team_arrivals.ensure(2 * (size_t)n));
        a.team_w = T;
        a.team_min_samples = 0;
        a.team_partials = e->d_team_partials.p;
        a.team_arrivals = e->d_team_arrivals.p;
        a.team_fault = team_fault_hook();
        if (e->cfg.device >= 0 && e->cfg.device < 64) { // (team launches of one device take turns, see below)
          ref_team_turn = std::unique_lock<std::mutex>(g_team_mu);
          ref_team_done = &g_team_done[e->cfg.device];
          if (!*ref_team_done)
            HIPCHK(hipEventCreateWithFlags(ref_team_done, hipEventDisableTiming));
          else
            HIPCHK(hipStreamWaitEvent(st, *ref_team_done, 0));
        }
      }
      HIPCHK(lk_launch_solve(a, e->cfg.fitting_model, e->cfg.interpolation, ref_group, st));
      if (ref_team_done)
        HIPCHK(hipEventRecord(*ref_team_done, st));
      ref_team_turn = std::unique_lock<std::mutex>();
      if (st != e->stream) {
        HIPCHK(hipEventRecord(e->ev_join[c], st));
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join[c], 0));
      }
      ++n_launched;
      continue;
    }
    std::unique_lock<std::mutex> team_turn;
    hipEvent_t *team_done = nullptr;
    if (e->team_share_permille > 0 && (c == kTeamClass || c == kTeamClass - 1))
      a.slots_permille = c == kTeamClass ? e->team_share_permille : 1000 - e->team_share_permille;
    if (c == kTeamClass) {
      a.team_w = e->team_w;
      a.team_min_samples = e->team_min_samples;
      a.team_partials = e->d_team_partials.p;
      a.team_arrivals = e->d_team_arrivals.p;
      a.team_fault = team_fault_hook();
      if (e->team_w > 1 && e->cfg.device >= 0 && e->cfg.device < 64) {
        team_turn = std::unique_lock<std::mutex>(g_team_mu);
        team_done = &g_team_done[e->cfg.device];
        if (!*team_done)
          HIPCHK(hipEventCreateWithFlags(team_done, hipEventDisableTiming));
        else
          HIPCHK(hipStreamWaitEvent(st, *team_done, 0));
      }
    }
    if (e->class_starved[c]) { // coarsest level(s) first, one lane per sector (+ finisher)
      int rc = launch_starved(e, a, c, e->class_begin[c], st);
      if (rc)
        return rc;
    }
    {
      int rc = launch_groups(e, a, kGroupOfClass[c], c, e->class_begin[c], st);
      if (rc)
        return rc;
    }
    if (team_done)
      HIPCHK(hipEventRecord(*team_done, st));
    team_turn = std::unique_lock<std::mutex>();
    if (st != e->stream) {
      HIPCHK(hipEventRecord(e->ev_join[c], st));
      HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join[c], 0));
    }
    ++n_launched;
  }
  if (e->reference_order > 0 && !e->defer_stale) {
    int rc = stale_iterations(e, d_result, e->S);
    if (rc)
      return rc;
  }
  if (e->timing) {
    HIPCHK(hipEventRecord(e->ev_s1, e->stream));
    e->solve_timed = true;
  }
  e->stats_valid = false;
  e->stats_frames = 1;
  return LK_ERROR_NONE;
}

int lk_correlate_all_async(lk_engine *e) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (e->results_pending)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_correlate_all_async: the previous solve has not been waited for");
  HIPCHK(hipSetDevice(e->cfg.device));
  int rc = refresh_level_views(e);
  if (rc)
    return rc;
  if (e->h_results_n < (size_t)e->S) {
    if (e->h_results)
      HIPCHK(hipHostFree(e->h_results));
    e->h_results = nullptr;
    e->h_results_n = 0;
    HIPCHK(hipHostMalloc((void **)&e->h_results, (size_t)e->S * sizeof(lk_result), hipHostMallocDefault));
    e->h_results_n = (size_t)e->S;
  }
  if (!e->ev_results)
    HIPCHK(hipEventCreateWithFlags(&e->ev_results, hipEventDisableTiming));
  rc = launch_all(e, e->d_guess.p, e->d_result.p);
  if (rc)
    return rc;
  HIPCHK(hipMemcpyAsync(e->h_results, e->d_result.p, (size_t)e->S * sizeof(lk_result), hipMemcpyDeviceToHost,
                        e->stream));
  HIPCHK(hipEventRecord(e->ev_results, e->stream));
  e->results_pending = true;
  return LK_ERROR_NONE;
}

// Wait for an event with the latency of a polling loop: hipEventSynchronize may put the thread to sleep and wake it
// hundreds of microseconds late - more than a whole C2 solve takes.  Poll for a while first (a solve is 0.2-10 ms), then
// block.
static hipError_t wait_event(hipEvent_t ev) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipEventQuery(ev);
    if (q != hipErrorNotReady)
      return q;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
      return hipEventSynchronize(ev);
    std::this_thread::yield();
  }
}

int lk_wait_results(lk_engine *e, lk_result *out) {
  Range range_("lk:gather records");
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->results_pending || !out)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_wait_results: no solve outstanding / null buffer");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(wait_event(e->ev_results));
  e->results_pending = false;
  std::memcpy(out, e->h_results, (size_t)e->S * sizeof(lk_result));
  return LK_ERROR_NONE;
}

int lk_correlate_all_device(lk_engine *e, const void *d_guesses, void *d_results) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  HIPCHK(hipSetDevice(e->cfg.device));
  int rc = refresh_level_views(e);
  if (rc)
    return rc;
  return launch_all(e, d_guesses ? (const float *)d_guesses : e->d_guess.p,
                    d_results ? (lk_result *)d_results : e->d_result.p);
}

} // extern "C"

int lk_internal_set_defer_stale(lk_engine *e, int on) { // (lk_internal.hpp)
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  e->defer_stale = on != 0;
  return LK_ERROR_NONE;
}
int lk_internal_reference_order(const lk_engine *e) { return e ? e->reference_order : 0; }
int lk_internal_set_image_device_after(lk_engine *e, int slot, const void *device_pixels, int rows, int cols, int step,
                                       hipEvent_t after, hipEvent_t consumed) {
  return set_image_common(e, slot, device_pixels, true, rows, cols, step, after, consumed);
}
// does a solve of this engine launch teams (workgroups that wait for each other and must all be resident)?  A collective
// kernel on another stream could hold the slots one of them needs: the group then keeps its collectives in stream order.
int lk_internal_team_launches(const lk_engine *e) {
  if (!e || !e->committed)
    return 0;
  const int n_team = e->class_begin[kTeamClass + 1] - e->class_begin[kTeamClass];
  const int n_big = e->class_begin[kTeamClass + 1] - e->class_begin[kTeamClass - 1];
  return (n_team > 0 && e->team_w > 1) || (e->reference_order > 1 && n_big > 0) ? 1 : 0;
}
int lk_internal_sector_count(const lk_engine *e) { return e ? e->S : 0; }

extern "C" {

int lk_get_results_device(lk_engine *e, const void **d_records) {
  if (!e || !d_records)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_results_device: sectors are not committed");
  *d_records = e->d_result.p;
  return LK_ERROR_NONE;
}

int lk_correlate_all(lk_engine *e, const float *guesses, lk_result *out) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!out)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_correlate_all: null result buffer");
  HIPCHK(hipSetDevice(e->cfg.device));
  int rc = refresh_level_views(e);
  if (rc)
    return rc;
  if (guesses)
    HIPCHK(hipMemcpyAsync(e->d_guess.p, guesses, 6 * (size_t)e->S * sizeof(float), hipMemcpyHostToDevice,
                          e->stream));
  rc = launch_all(e, e->d_guess.p, e->d_result.p);
  if (rc)
    return rc;
  Range range_("lk:gather records");
  HIPCHK(hipMemcpyAsync(out, e->d_result.p, (size_t)e->S * sizeof(lk_result), hipMemcpyDeviceToHost,
                        e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return LK_ERROR_NONE;
}

int lk_correlate(lk_engine *e, int sector, float *guess_inout, lk_result *out) {
  Range range_("lk:correlate one sector");
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!guess_inout || !out)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_correlate: null argument");
  HIPCHK(hipSetDevice(e->cfg.device));
  int rc = refresh_level_views(e);
  if (rc)
    return rc;
  if (sector < 0 || sector >= e->S)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_correlate: unknown sector");
  float g[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < e->P; ++i)
    g[i] = guess_inout[i];
  uint32_t sidx = (uint32_t)sector;
  HIPCHK(hipMemcpyAsync(e->d_guess.p + 6 * (size_t)sector, g, sizeof(g), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_single.p, &sidx, sizeof(sidx), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  LkSolveArgs a = base_args(e, e->d_guess.p, e->d_result.p);
  a.order = e->d_single.p;
  const int group = kGroupOfClass[e->h_class[(size_t)sector]];
  a.n_sectors = 1;
  a.queue = e->d_queue.p;
  a.safe = safe_flavour(e) ? 1 : 0;
  if (e->reference_order > 0) { // (see launch_all)
    a.safe = 1;
    a.solo = 0;
    a.reference_order = e->reference_order;
    if (e->timing)
      HIPCHK(hipEventRecord(e->ev_s0, e->stream));
    HIPCHK(lk_launch_solve(a, e->cfg.fitting_model, e->cfg.interpolation, e->h_class[(size_t)sector] == 0 ? 16 : 64,
                           e->stream));
    if (int rc2 = stale_iterations(e, e->d_result.p + sector, 1))
      return rc2;
    if (e->timing) {
      HIPCHK(hipEventRecord(e->ev_s1, e->stream));
      e->solve_timed = true;
    }
    e->stats_valid = false;
    e->stats_frames = 1;
    HIPCHK(hipMemcpyAsync(out, e->d_result.p + sector, sizeof(lk_result), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int i = 0; i < e->P; ++i)
      guess_inout[i] = out->resultingParameters[i];
    return LK_ERROR_NONE;
  }
  std::unique_lock<std::mutex> team_turn;
  hipEvent_t *team_done = nullptr;
  if (e->h_class[(size_t)sector] == kTeamClass) {
    a.team_w = e->team_w;
    a.team_min_samples = e->team_min_samples;
    a.team_partials = e->d_team_partials.p;
    a.team_arrivals = e->d_team_arrivals.p;
    a.team_fault = team_fault_hook();
    if (e->team_w > 1 && e->cfg.device >= 0 && e->cfg.device < 64) { // (see launch_all)
      team_turn = std::unique_lock<std::mutex>(g_team_mu);
      team_done = &g_team_done[e->cfg.device];
      if (!*team_done)
        HIPCHK(hipEventCreateWithFlags(team_done, hipEventDisableTiming));
      else
        HIPCHK(hipStreamWaitEvent(e->stream, *team_done, 0));
    }
  }
  if (e->timing)
    HIPCHK(hipEventRecord(e->ev_s0, e->stream));
  if (e->class_starved[e->h_class[(size_t)sector]]) {
    int rc = launch_starved(e, a, 0, 0, e->stream);
    if (rc)
      return rc;
  }
  {
    int rc = launch_groups(e, a, group, 0, 0, e->stream);
    if (rc)
      return rc;
  }
  if (team_done)
    HIPCHK(hipEventRecord(*team_done, e->stream));
  team_turn = std::unique_lock<std::mutex>();
  if (e->timing) {
    HIPCHK(hipEventRecord(e->ev_s1, e->stream));
    e->solve_timed = true;
  }
  e->stats_valid = false;
  e->stats_frames = 1;
  HIPCHK(hipMemcpyAsync(out, e->d_result.p + sector, sizeof(lk_result), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int i = 0; i < e->P; ++i)
    guess_inout[i] = out->resultingParameters[i]; // cuda_class.cu:289-290
  return LK_ERROR_NONE;
}

int lk_adjust_initial_guess(lk_engine *e, int frame, int constant_velocity, const float *global_guess,
                            float global_cx, float global_cy) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_adjust_initial_guess: sectors are not committed");
  float gg[6] = {0, 0, 0, 0, 0, 0};
  if (global_guess)
    for (int i = 0; i < 6; ++i)
      gg[i] = global_guess[i];
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(lk_launch_guess(e->d_center.p, e->d_last_p.p, e->d_prev_p.p, e->d_guess.p, gg, global_cx, global_cy,
                         e->S, e->cfg.fitting_model, frame, constant_velocity, e->stream));
  return LK_ERROR_NONE;
}

int lk_get_guesses(lk_engine *e, float *guesses) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed || !guesses)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_guesses: sectors are not committed");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipMemcpyAsync(guesses, e->d_guess.p, 6 * (size_t)e->S * sizeof(float), hipMemcpyDeviceToHost,
                        e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return LK_ERROR_NONE;
}

// ------------------------------------------------------------------------------------
// Frame-pipelined windows of a sequence (perform_multiframe_correlation's frame loop,
// manager_class.cpp:1380-1496, with the Eulerian description: the sectors stay, the deformed
// image changes, and the guess of frame f + 1 of a sector is a function of that sector's own
// earlier results, :2677-2699).  K deformed frames are resident at once (the ring); ONE launch
// per size class solves all of them, every sector advancing from frame to frame on its own
// (the SEQ instances of lk_solve_kernel), instead of one launch - and one straggler tail - per pair.
// ------------------------------------------------------------------------------------
int lk_sequence_reserve(lk_engine *e, int n_slots) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (n_slots < 1 || n_slots > 4096)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_sequence_reserve: n_slots must be in 1..4096");
  if (e->seq.outstanding)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_sequence_reserve: a window is outstanding (lk_wait_sequence)");
  HIPCHK(hipSetDevice(e->cfg.device));
  std::lock_guard<std::mutex> lock(e->nxt_mu);
  if ((int)e->ring.size() < n_slots) {
    const size_t old = e->ring.size();
    e->ring.resize((size_t)n_slots);
    e->ring_ready.resize((size_t)n_slots, nullptr);
    e->ring_fresh.resize((size_t)n_slots, 0);
    e->ring_window.resize((size_t)n_slots, 0u);
    for (size_t i = old; i < (size_t)n_slots; ++i)
      HIPCHK(hipEventCreateWithFlags(&e->ring_ready[i], hipEventDisableTiming));
  }
  return LK_ERROR_NONE;
}

static int sequence_set_frame(lk_engine *e, int slot, const void *src, bool on_device, int rows, int cols, int step,
                              hipEvent_t after = nullptr, hipEvent_t consumed = nullptr) {
  Range range_("lk:upload+pyramid (ring)");
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!src || rows < 1 || cols < 1 || step < cols)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_sequence_set_frame: bad arguments");
  HIPCHK(hipSetDevice(e->cfg.device));
  // like the next-frame slot: filled on the next-frame stream, so that the frames of window w + 1 arrive and get their
  // pyramids while window w is being solved (manager_class.cpp:1438-1447)
  std::unique_lock<std::mutex> lock(e->nxt_mu);
  if (slot < 0 || slot >= (int)e->ring.size())
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_sequence_set_frame: unknown ring slot (lk_sequence_reserve)");
  hipStream_t st = e->nxt_stream;
  if (e->ring_window[(size_t)slot]) // a window read this slot: it must be through with it (stream order on the device)
    HIPCHK(hipStreamWaitEvent(st, e->seq_done[(e->ring_window[(size_t)slot] - 1) % 4], 0));
  if (after)
    HIPCHK(hipStreamWaitEvent(st, after, 0));
  int rc = fill_image(e, e->ring[(size_t)slot], src, on_device, rows, cols, step, st, false);
  if (rc)
    return rc;
  if (consumed)
    HIPCHK(hipEventRecord(consumed, st));
  HIPCHK(hipEventRecord(e->ring_ready[(size_t)slot], st));
  e->ring_fresh[(size_t)slot] = 1;
  lock.unlock();
  if (!on_device && !host_pinned(src)) // pageable host memory: the copy must have left it
    HIPCHK(hipStreamSynchronize(st));
  return LK_ERROR_NONE;
}

int lk_sequence_set_frame(lk_engine *e, int slot, const uint8_t *host_pixels, int rows, int cols, int step) {
  return sequence_set_frame(e, slot, host_pixels, false, rows, cols, step);
}
int lk_sequence_set_frame_device(lk_engine *e, int slot, const void *device_pixels, int rows, int cols, int step) {
  return sequence_set_frame(e, slot, device_pixels, true, rows, cols, step);
}
} // extern "C"
int lk_internal_sequence_set_frame_device_after(lk_engine *e, int slot, const void *device_pixels, int rows, int cols, int step,
                                                hipEvent_t after, hipEvent_t consumed) { // (lk_internal.hpp)
  return sequence_set_frame(e, slot, device_pixels, true, rows, cols, step, after, consumed);
}
extern "C" {

// lane group and flavour of a class inside a window (-1: the class has no frame-pipelined instance)
static int seq_group_of_class(const lk_engine *e, int c, int n) {
  if (e->reference_order > 0) { // (launch_all's rule)
    const int g = c == 0 ? 16 : (c >= kTeamClass - 1 || (c == kTeamClass - 2 && n <= 256)) ? 512 : 64;
    return g == 512 ? -1 : g;
  }
  const int g = kGroupOfClass[c];
  if (g > 64)
    return -1;
  if (e->class_starved[c]) {
    // (starved levels inside the window: the 16-lane rows' finisher arithmetic - whatever width the one-pair classifier
    // promoted a class of few small sectors to, e.g. one rank's block of a sharded sequence: the rule below keeps big ones out)
    // ... which pays while the sectors are small.  Config 5's geometry (17 x 17 samples, one starved level of four) measured
    // 9.1 ms per pair in a window against 5.9 (default) / 7.4 (batch-invariant) for the one-pair chain, whose one-lane kernel
    // shares one instruction stream of the QR among 64 sectors and whose lane groups widen: such classes keep the chain, frame
    // after frame.  9 x 9 samples: 1.12 against 1.66; 7 x 7: 0.98-1.21 against 1.6-1.9.
    static const int big_n0 = [] { const char *f = std::getenv("LK_SEQ_STARVED_MAX_N0"); return f ? std::atoi(f) : 128; }(); // tuning hook
    long long n0 = 0, n_starved = 0; // (over the sectors that HAVE a starved level: a class may mix them with others)
    const int top = e->cfg.py_stop;
    for (int i = e->class_begin[c]; i < e->class_begin[c + 1]; ++i) {
      const int s = (int)e->h_order[(size_t)i];
      const int4 r = e->h_rect[top][(size_t)s];
      const int n_top = r.z > 0 ? r.w : (int)(e->h_off[top][(size_t)s + 1] - e->h_off[top][(size_t)s]);
      if (n_top <= starved_max(e)) {
        n0 += level0_count(e, s);
        ++n_starved;
      }
    }
    if (n0 > (long long)big_n0 * n_starved)
      return -1;
    return 16;
  }
  return g;
}

static bool seq_pipelinable(const lk_engine *e) {
  static const bool off = [] { const char *f = std::getenv("LK_SEQ_PIPELINE"); return f && std::atoi(f) == 0; }(); // comparison hook
  if (off)
    return false;
  for (int c = 0; c < kNumClasses; ++c) {
    const int n = e->class_begin[c + 1] - e->class_begin[c];
    if (n > 0 && seq_group_of_class(e, c, n) < 0)
      return false;
  }
  return true;
}

// one window on the device: the pipelined launches, or - domains with classes that have no such instance (teams,
// workgroup-wide groups) - the frames one after the other through the ordinary launches, same buffers either way
static int launch_window(lk_engine *e, bool force_safe_flavour) {
  lk_engine::SeqWindow &w = e->seq;
  const int S = e->S, n = w.n_frames, R = (int)e->ring.size();
  const DevImage &u0 = w.und_slot >= 0 ? e->ring[(size_t)w.und_slot] : e->img[LK_IMG_UND];
  auto def_of = [&](int i) -> DevImage & { return e->ring[(size_t)((w.first_slot + i) % R)]; };
  auto und_of = [&](int i) -> const DevImage & { return (w.reference_previous && i > 0) ? def_of(i - 1) : u0; };
  if (e->timing)
    HIPCHK(hipEventRecord(e->ev_s0, e->stream));
  if (!w.pipelined) {
    // the one-pair loop on the device: images of frame i in the pair slots, guess, solve, records to their row
    DevImage keep_und = e->img[LK_IMG_UND], keep_def = e->img[LK_IMG_DEF];
    int rc = LK_ERROR_NONE;
    for (int i = 0; i < n && !rc; ++i) {
      e->img[LK_IMG_UND] = und_of(i);
      e->img[LK_IMG_DEF] = def_of(i);
      e->lv_dirty = true;
      rc = refresh_level_views(e);
      if (!rc && i > 0) {
        const float gg[6] = {0, 0, 0, 0, 0, 0};
        if (lk_launch_guess(e->d_center.p, e->d_last_p.p, e->d_prev_p.p, e->d_guess.p, gg, 0.f, 0.f, S, e->cfg.fitting_model, 1,
                            w.velocity, e->stream) != hipSuccess)
          rc = e->fail(LK_ERROR_DEVICE, "lk_correlate_sequence: guess launch failed");
      }
      if (!rc && w.want_guesses && hipMemcpyAsync(e->d_seq_guess.p + (size_t)i * 6 * (size_t)S, e->d_guess.p, 6 * (size_t)S * sizeof(float),
                                                 hipMemcpyDeviceToDevice, e->stream) != hipSuccess)
        rc = e->fail(LK_ERROR_DEVICE, "lk_correlate_sequence: guess copy failed");
      const bool timing = e->timing;
      e->timing = false; // (the window's own events bracket all frames)
      if (!rc)
        rc = launch_all(e, e->d_guess.p, e->d_seq_result.p + (size_t)i * (size_t)S);
      e->timing = timing;
      if (!rc && hipMemcpyAsync(e->d_seq_stats.p + (size_t)i * 4 * (size_t)S, e->d_stats.p, 4 * (size_t)S * sizeof(uint32_t),
                                hipMemcpyDeviceToDevice, e->stream) != hipSuccess)
        rc = e->fail(LK_ERROR_DEVICE, "lk_correlate_sequence: stats copy failed");
    }
    e->img[LK_IMG_UND] = keep_und;
    e->img[LK_IMG_DEF] = keep_def;
    e->lv_dirty = true;
    if (rc)
      return rc;
  } else {
    // per-frame image table, chain and flags of this window
    std::vector<LkSeqFrame> table((size_t)n);
    for (int i = 0; i < n; ++i) {
      const DevImage &u = und_of(i), &d = def_of(i);
      for (int l = 0; l < LK_MAX_LEVELS; ++l) {
        table[(size_t)i].und[l] = l <= e->cfg.py_stop ? u.lvl[l] : nullptr;
        table[(size_t)i].def[l] = l <= e->cfg.py_stop ? d.lvl[l] : nullptr;
      }
    }
    HIPCHK(hipMemcpyAsync(e->d_seq_img.p, table.data(), table.size() * sizeof(LkSeqFrame), hipMemcpyHostToDevice, e->stream)); // (pageable: staged before it returns)
    HIPCHK(hipMemsetAsync(e->d_seq_chain.p, 0, (size_t)S * kLkSeqChainWords * sizeof(unsigned long long), e->stream));
    HIPCHK(hipMemsetAsync(e->d_seq_flags.p, 0, 4 * sizeof(uint32_t), e->stream));
    int n_classes = 0, n_launched = 0;
    for (int c = 0; c < kNumClasses; ++c)
      n_classes += e->class_begin[c + 1] > e->class_begin[c];
    if (n_classes > 1) { // classes are independent sector sets: side by side, as in launch_all
      if (!e->ev_fork)
        HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
      HIPCHK(hipEventRecord(e->ev_fork, e->stream));
    }
    for (int c = 0; c < kNumClasses; ++c) {
      const int nc = e->class_begin[c + 1] - e->class_begin[c];
      if (nc <= 0)
        continue;
      hipStream_t st = e->stream;
      if (n_launched > 0) {
        if (!e->class_stream[c]) {
          HIPCHK(hipStreamCreateWithFlags(&e->class_stream[c], hipStreamNonBlocking));
          HIPCHK(hipEventCreateWithFlags(&e->ev_join[c], hipEventDisableTiming));
        }
        st = e->class_stream[c];
        HIPCHK(hipStreamWaitEvent(st, e->ev_fork, 0));
      }
      LkSolveArgs a = base_args(e, e->d_guess.p, e->d_seq_result.p);
      a.stats = e->d_seq_stats.p;
      a.order = e->d_order.p + e->class_begin[c];
      a.n_sectors = nc;
      a.queue = e->d_queue.p + 8 * c;
      a.solo = 0;
      a.seq_frames = n;
      a.seq_velocity = w.velocity;
      a.seq_stride = S;
      a.seq_img = e->d_seq_img.p;
      a.seq_chain = e->d_seq_chain.p;
      a.seq_prev_p = e->d_prev_p.p;
      a.seq_prev_p_out = e->d_prev_p_alt.p;
      a.seq_guess_out = w.want_guesses ? e->d_seq_guess.p : nullptr;
      a.seq_flags = e->d_seq_flags.p;
      if (const char *f = std::getenv("LK_SEQ_FAULT")) // test hook, read per launch: a frame that never publishes
        a.seq_fault = std::atoi(f);
      int flavour = (force_safe_flavour || safe_flavour(e) || e->class_starved[c]) ? 1 : 0;
      if (e->reference_order > 0) {
        flavour = 2;
        a.reference_order = e->reference_order;
        a.mark_stale = 1;
      } else if (e->class_starved[c]) {
        // Sectors that spend most of their evaluations on starved levels (config 4: 7 x 7 samples, 9-16 and 1-4 at levels 1
        // and 2).  The rows of a wavefront then sit at different levels most of the time: no level alignment (it never
        // changes a bit; 1.33 -> 1.21 ms per pair).  And in the default mode - whose arithmetic is free within the
        // reference's noise - such a class takes the reference-order instance with its lanes dealt by need, which IS the
        // CPU engine's arithmetic and the fastest instance at these sizes (0.98 ms per pair).
        long long n0 = 0;
        for (int i = e->class_begin[c]; i < e->class_begin[c + 1]; ++i)
          n0 += level0_count(e, (int)e->h_order[(size_t)i]);
        static const int small_n0 = [] { const char *f = std::getenv("LK_SEQ_SMALL"); return f ? std::atoi(f) : 64; }(); // tuning hook
        // ... while the class is in the throughput regime.  With fewer sectors than lane-group rows are resident (12 288 at three
        // wavefronts per SIMD: one rank's block of a sharded sequence) the window lasts as long as its slowest sector's chain, and
        // the unified rows' steps are the shorter ones: blocks of config 4's grid, windows of 16 pairs, reference-order instance /
        // unified rows: 25 088 sectors 0.77 / 0.82 ms per pair, 12 544: 0.67 / 0.61, 6272: 0.58 / 0.50, 3136: 0.54 / 0.48.
        static const int latency_nc = [] { const char *f = std::getenv("LK_SEQ_LATENCY_SECTORS"); return f ? std::atoi(f) : 16384; }(); // tuning hook
        // (and there the rows of a wavefront are better kept in step again - alignment never changes a bit: 6272 sectors, unified
        // rows, aligned / not: 0.50 / 0.59 ms per pair)
        if (n0 <= (long long)small_n0 * nc && nc > latency_nc) {
          a.align = 0;
          if (!safe_flavour(e) && !force_safe_flavour) {
            flavour = 2;
            a.reference_order = 1;
            a.mark_stale = -1; // (a first evaluation that fails reports 0 iterations, like every default-mode instance)
          }
        }
      }
      a.safe = flavour != 0;
      HIPCHK(lk_launch_solve_seq(a, e->cfg.fitting_model, e->cfg.interpolation, seq_group_of_class(e, c, nc), flavour, st));
      if (st != e->stream) {
        HIPCHK(hipEventRecord(e->ev_join[c], st));
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join[c], 0));
      }
      ++n_launched;
    }
    // reference-order mode: the stale iteration counts, in the order the reference solves - frame by frame, sector by
    // sector, which IS the layout of the window's records
    if (e->reference_order > 0 && !e->defer_stale) {
      int rc = stale_iterations(e, e->d_seq_result.p, n * S);
      if (rc)
        return rc;
    }
    HIPCHK(hipMemcpyAsync(e->h_seq_flags, e->d_seq_flags.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
  }
  if (e->timing) {
    HIPCHK(hipEventRecord(e->ev_s1, e->stream));
    e->solve_timed = true;
  }
  if (w.want_host)
    HIPCHK(hipMemcpyAsync(e->h_seq_results[e->h_seq_cur], e->d_seq_result.p, (size_t)n * (size_t)S * sizeof(lk_result), hipMemcpyDeviceToHost,
                          e->stream));
  HIPCHK(hipEventRecord(e->ev_seq, e->stream));
  e->stats_valid = false;
  e->stats_frames = n;
  return LK_ERROR_NONE;
}

// one of the two alternating pinned record buffers of the windows, at least `need` records
static int ensure_host_records(lk_engine *e, int b, size_t need) {
  if (e->h_seq_results_n[b] >= need)
    return LK_ERROR_NONE;
  if (e->h_seq_results[b])
    HIPCHK(hipHostFree(e->h_seq_results[b]));
  e->h_seq_results[b] = nullptr;
  e->h_seq_results_n[b] = 0;
  HIPCHK(hipHostMalloc((void **)&e->h_seq_results[b], need * sizeof(lk_result), hipHostMallocDefault));
  e->h_seq_results_n[b] = need;
  return LK_ERROR_NONE;
}

int lk_sequence_prepare_host_records(lk_engine *e, int n_frames) {
  if (!e || n_frames < 1)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed)
    return e->fail(LK_ERROR_BAD_DOMAIN, "sectors are not committed (call lk_commit_sectors)");
  HIPCHK(hipSetDevice(e->cfg.device));
  // (the buffer the running window's records go to - h_seq_cur - is not touched: this is the other one, which the next launch takes)
  return ensure_host_records(e, e->h_seq_cur ^ 1, (size_t)n_frames * (size_t)e->S);
}

int lk_correlate_sequence_async(lk_engine *e, int und_slot, int first_slot, int n_frames, int reference_previous,
                                int constant_velocity, int flags) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (e->seq.outstanding || e->results_pending)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_correlate_sequence_async: the previous solve has not been waited for");
  HIPCHK(hipSetDevice(e->cfg.device));
  const int R = (int)e->ring.size();
  if (n_frames < 1 || n_frames > R || first_slot < 0 || first_slot >= R || und_slot >= R)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_correlate_sequence_async: the window does not fit the ring (lk_sequence_reserve)");
  if (und_slot < 0 && !e->img[LK_IMG_UND].valid)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_correlate_sequence_async: no undeformed image");
  if (!e->committed)
    return e->fail(LK_ERROR_BAD_DOMAIN, "sectors are not committed (call lk_commit_sectors)");
  if (e->recommit_pending) {
    e->recommit_pending = false;
    int rc = commit_impl(e, true);
    if (rc)
      return rc;
  }
  if (e->classes_dirty) {
    HIPCHK(hipStreamSynchronize(e->stream));
    int rc = classify_sectors(e);
    if (rc)
      return rc;
  }
  lk_engine::SeqWindow &w = e->seq;
  w.und_slot = und_slot;
  w.first_slot = first_slot;
  w.n_frames = n_frames;
  w.reference_previous = reference_previous != 0;
  w.velocity = constant_velocity != 0;
  w.want_host = (flags & 1) != 0;
  w.want_guesses = (flags & 2) != 0;
  w.pipelined = seq_pipelinable(e);
  const int S = e->S;
  // (held until this window's done-event is on the stream: a concurrent lk_sequence_set_frame into one of its slots waits for it)
  std::unique_lock<std::mutex> ring_lock(e->nxt_mu);
  {
    // the frames of the window: built, of one geometry, and visible to the solve stream
    const DevImage &u0 = und_slot >= 0 ? e->ring[(size_t)und_slot] : e->img[LK_IMG_UND];
    if (!u0.valid)
      return e->fail(LK_ERROR_BAD_DOMAIN, "lk_correlate_sequence_async: the undeformed frame's ring slot is empty");
    for (int i = 0; i < n_frames; ++i) {
      const size_t slot = (size_t)((first_slot + i) % R);
      const DevImage &d = e->ring[slot];
      if (!d.valid || d.rows != u0.rows || d.cols != u0.cols)
        return e->fail(LK_ERROR_BAD_DOMAIN, "lk_correlate_sequence_async: empty ring slot, or frames of different sizes");
      if (e->ring_fresh[slot]) {
        HIPCHK(hipStreamWaitEvent(e->stream, e->ring_ready[slot], 0));
        e->ring_fresh[slot] = 0;
      }
      e->ring_window[slot] = e->seq_windows + 1;
    }
    if (und_slot >= 0) {
      if (e->ring_fresh[(size_t)und_slot]) {
        HIPCHK(hipStreamWaitEvent(e->stream, e->ring_ready[(size_t)und_slot], 0));
        e->ring_fresh[(size_t)und_slot] = 0;
      }
      e->ring_window[(size_t)und_slot] = e->seq_windows + 1;
    }
  }
  HIPCHK(e->d_seq_result.ensure((size_t)n_frames * (size_t)S));
  HIPCHK(e->d_seq_stats.ensure((size_t)n_frames * 4 * (size_t)S));
  if (w.want_guesses)
    HIPCHK(e->d_seq_guess.ensure((size_t)n_frames * 6 * (size_t)S));
  HIPCHK(e->d_seq_chain.ensure((size_t)S * kLkSeqChainWords));
  HIPCHK(e->d_seq_img.ensure((size_t)n_frames));
  HIPCHK(e->d_seq_flags.ensure(4));
  HIPCHK(e->d_prev_p_alt.ensure(6 * (size_t)S));
  if (!e->h_seq_flags)
    HIPCHK(hipHostMalloc((void **)&e->h_seq_flags, 4 * sizeof(uint32_t), hipHostMallocDefault));
  std::memset(e->h_seq_flags, 0, 4 * sizeof(uint32_t));
  if (w.want_host) {
    e->h_seq_cur ^= 1;
    if (int hrc = ensure_host_records(e, e->h_seq_cur, (size_t)n_frames * (size_t)S))
      return hrc;
  }
  if (!e->ev_seq)
    HIPCHK(hipEventCreateWithFlags(&e->ev_seq, hipEventDisableTiming));
  for (hipEvent_t &ev : e->seq_done)
    if (!ev)
      HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  if (w.pipelined) {
    // the level table: lists, rectangles and geometry (the images come from the per-frame table)
    DevImage keep_und = e->img[LK_IMG_UND], keep_def = e->img[LK_IMG_DEF];
    e->img[LK_IMG_UND] = und_slot >= 0 ? e->ring[(size_t)und_slot] : e->img[LK_IMG_UND];
    e->img[LK_IMG_DEF] = e->ring[(size_t)first_slot];
    e->lv_dirty = true;
    int rc = refresh_level_views(e);
    e->img[LK_IMG_UND] = keep_und;
    e->img[LK_IMG_DEF] = keep_def;
    e->lv_dirty = true;
    if (rc)
      return rc;
  }
  Range range_("lk:solve window");
  int rc = launch_window(e, false);
  if (rc)
    return rc;
  HIPCHK(hipEventRecord(e->seq_done[e->seq_windows % 4], e->stream));
  ++e->seq_windows;
  ring_lock.unlock();
  w.outstanding = true;
  return LK_ERROR_NONE;
}

int lk_wait_sequence(lk_engine *e, lk_result *out) {
  Range range_("lk:gather window records");
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  lk_engine::SeqWindow &w = e->seq;
  if (!w.outstanding)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_wait_sequence: no window outstanding");
  if (out && !w.want_host)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_wait_sequence: the window was launched without host records (flags & 1)");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(wait_event(e->ev_seq));
  w.outstanding = false;
  const int S = e->S, n = w.n_frames;
  if (w.pipelined) {
    if (e->h_seq_flags[0] != 0u)
      return e->fail(LK_ERROR_DEVICE, "lk_wait_sequence: a sector's wait for its previous frame ran into its bound - the window is void");
    if (e->h_seq_flags[1] != 0u) {
      // a bad pivot in the fast flavour: the whole window again with the SAFE instances (same guesses, same history -
      // the sequence state is only committed below)
      int rc = launch_window(e, true);
      if (rc)
        return rc;
      HIPCHK(hipEventRecord(e->seq_done[(e->seq_windows + 3) % 4], e->stream)); // (this window's event, again)
      HIPCHK(hipEventSynchronize(e->ev_seq));
      if (e->h_seq_flags[0] != 0u)
        return e->fail(LK_ERROR_DEVICE, "lk_wait_sequence: a sector's wait for its previous frame ran into its bound - the window is void");
    }
    // commit the sequence state: previous_resulting_parameters (the last frame but one), the engine's own record buffer
    if (n >= 2)
      HIPCHK(hipMemcpyAsync(e->d_prev_p.p, e->d_prev_p_alt.p, 6 * (size_t)S * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_result.p, e->d_seq_result.p + (size_t)(n - 1) * (size_t)S, (size_t)S * sizeof(lk_result),
                          hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_stats.p, e->d_seq_stats.p + (size_t)(n - 1) * 4 * (size_t)S, 4 * (size_t)S * sizeof(uint32_t),
                          hipMemcpyDeviceToDevice, e->stream));
  } else {
    HIPCHK(hipMemcpyAsync(e->d_result.p, e->d_seq_result.p + (size_t)(n - 1) * (size_t)S, (size_t)S * sizeof(lk_result),
                          hipMemcpyDeviceToDevice, e->stream));
  }
  if (out)
    std::memcpy(out, e->h_seq_results[e->h_seq_cur], (size_t)n * (size_t)S * sizeof(lk_result));
  return LK_ERROR_NONE;
}

int lk_sequence_host_records(lk_engine *e, const lk_result **records) {
  if (!e || !records)
    return LK_ERROR_BAD_DOMAIN;
  if (e->seq.outstanding || !e->seq.want_host || !e->h_seq_results[e->h_seq_cur])
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_sequence_host_records: no waited-for window with host records (flags & 1)");
  *records = e->h_seq_results[e->h_seq_cur];
  return LK_ERROR_NONE;
}

int lk_get_sequence_results_device(lk_engine *e, const void **d_records, const void **d_guesses) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->d_seq_result.p)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_sequence_results_device: no window was solved");
  if (d_records)
    *d_records = e->d_seq_result.p;
  if (d_guesses)
    *d_guesses = e->d_seq_guess.p;
  return LK_ERROR_NONE;
}

int lk_sequence_is_pipelined(lk_engine *e) { return e && e->seq.pipelined ? 1 : 0; }

int lk_copy_sequence_records_device(lk_engine *e, void *d_dst, size_t dst_pitch_records) {
  if (!e || !d_dst)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->d_seq_result.p || e->seq.n_frames < 1 || dst_pitch_records < (size_t)e->S)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_copy_sequence_records_device: no window was solved, or the pitch is below the sector count");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipMemcpy2DAsync(d_dst, dst_pitch_records * sizeof(lk_result), e->d_seq_result.p, (size_t)e->S * sizeof(lk_result),
                          (size_t)e->S * sizeof(lk_result), (size_t)e->seq.n_frames, hipMemcpyDeviceToDevice, e->stream));
  return LK_ERROR_NONE;
}

int lk_get_sequence_guesses(lk_engine *e, float *out) {
  if (!e || !out)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->seq.want_guesses || e->seq.outstanding || !e->d_seq_guess.p)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_sequence_guesses: the last window kept no guesses (flags & 2), or is still outstanding");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipMemcpyAsync(out, e->d_seq_guess.p, (size_t)e->seq.n_frames * 6 * (size_t)e->S * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return LK_ERROR_NONE;
}

int lk_evaluate(lk_engine *e, int sector, int level, const float *p, float *A36, float *b6, float *chi,
                int *error) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  HIPCHK(hipSetDevice(e->cfg.device));
  int rc = refresh_level_views(e);
  if (rc)
    return rc;
  if (sector < 0 || sector >= e->S || level < 0 || level > e->cfg.py_stop || !p ||
      e->h_off[level].size() != (size_t)e->S + 1)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_evaluate: unknown sector/level");
  LkEvalArgs a{};
  a.lv = e->d_lv.p;
  a.center = e->d_center.p;
  a.sector = sector;
  a.level = level;
  for (int i = 0; i < 6; ++i)
    a.p[i] = i < e->P ? p[i] : 0.f;
  a.out = e->d_scratch.p;
  a.ref_threads = e->reference_order;
  const int group = e->reference_order > 0 ? (e->h_class[(size_t)sector] == 0 ? 16 : 64)
                                           : kGroupOfClass[e->h_class[(size_t)sector]];
  HIPCHK(lk_launch_eval(a, e->cfg.fitting_model, e->cfg.interpolation, group, e->stream));
  float h[44];
  HIPCHK(hipMemcpyAsync(h, e->d_scratch.p, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (A36)
    std::memcpy(A36, h, 36 * sizeof(float));
  if (b6)
    std::memcpy(b6, h + 36, 6 * sizeof(float));
  if (chi)
    *chi = h[42];
  if (error)
    *error = h[43] != 0.f ? LK_ERROR_INTERPOLATION_OUT_OF_IMAGE : LK_ERROR_NONE;
  return LK_ERROR_NONE;
}

int lk_sample(lk_engine *e, int slot, int level, const float *xy, int n, float *out4) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (slot < 0 || slot > 2 || level < 0 || level > e->cfg.py_stop || !e->img[slot].valid || !xy || !out4 ||
      n < 1)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_sample: bad arguments");
  HIPCHK(hipSetDevice(e->cfg.device));
  const DevImage &im = e->img[slot];
  DevBuf<float2> d_pts;
  DevBuf<float4> d_out;
  HIPCHK(d_pts.ensure((size_t)n));
  HIPCHK(d_out.ensure((size_t)n));
  int rc = LK_ERROR_NONE;
  hipError_t he = hipMemcpyAsync(d_pts.p, xy, 2 * sizeof(float) * (size_t)n, hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess)
    he = lk_launch_sample(e->cfg.interpolation, im.lvl[level], im.rows >> level, im.cols >> level, d_pts.p, n,
                          d_out.p, e->stream);
  if (he == hipSuccess)
    he = hipMemcpyAsync(out4, d_out.p, 4 * sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, e->stream);
  if (he == hipSuccess)
    he = hipStreamSynchronize(e->stream);
  if (he != hipSuccess)
    rc = e->hipfail(he, "lk_sample");
  d_pts.release();
  d_out.release();
  return rc;
}

int lk_damped_solve(lk_engine *e, int n, const float *A, const float *b, float lambda, float scaling,
                    int reference_solver, float *dp) {
  if (!e)
    return LK_ERROR_BAD_DOMAIN;
  if (!(n == 1 || n == 2 || n == 3 || n == 6) || !A || !b || !dp)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_damped_solve: n must be 1, 2, 3 or 6");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(e->d_scratch.ensure(64));
  float h[48];
  std::memset(h, 0, sizeof(h));
  for (int i = 0; i < n; ++i) {
    h[36 + i] = b[i];
    for (int j = 0; j < n; ++j)
      h[i * 6 + j] = A[i * n + j];
  }
  h[42] = lambda;
  h[43] = scaling;
  h[44] = (float)reference_solver; // 0: fast path, 1: the restated QR, 2: the same QR spread over a 16-lane row
  HIPCHK(hipMemcpyAsync(e->d_scratch.p, h, sizeof(h), hipMemcpyHostToDevice, e->stream));
  HIPCHK(lk_launch_solve_only(n, e->d_scratch.p, e->d_scratch.p + 48, e->stream));
  float o[6];
  HIPCHK(hipMemcpyAsync(o, e->d_scratch.p + 48, sizeof(o), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int i = 0; i < n; ++i)
    dp[i] = o[i];
  return LK_ERROR_NONE;
}

int lk_get_stats(lk_engine *e, lk_stats *out) {
  if (!e || !out)
    return LK_ERROR_BAD_DOMAIN;
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (!e->stats_valid && e->committed) {
    // (after a frame-pipelined window: the counters of all its frames)
    const size_t rows = (size_t)e->S * (size_t)(e->stats_frames > 1 ? e->stats_frames : 1);
    std::vector<uint32_t> h(4 * rows);
    HIPCHK(hipMemcpy(h.data(), e->stats_frames > 1 ? e->d_seq_stats.p : e->d_stats.p, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    lk_stats s{};
    s.sectors = (uint64_t)rows;
    for (size_t i = 0; i < rows; ++i) {
      s.evaluations += h[4 * i];
      s.sample_evaluations += h[4 * i + 1];
      s.point_iterations += h[4 * i + 2];
      s.ill_conditioned_solves += h[4 * i + 3];
    }
    // 25 B per sample-evaluation + 196 B per evaluation
    s.algorithmic_bytes = 25ull * s.sample_evaluations + 196ull * s.evaluations;
    s.solve_ms = e->stats.solve_ms;
    s.pyramid_ms = e->stats.pyramid_ms;
    e->stats = s;
    e->stats_valid = true;
  }
  if (e->solve_timed) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev_s0, e->ev_s1) == hipSuccess)
      e->stats.solve_ms = ms;
  }
  if (e->pyr_timed) {
    float ms = 0.f;
    if (hipEventSynchronize(e->ev_p1) == hipSuccess && hipEventElapsedTime(&ms, e->ev_p0, e->ev_p1) == hipSuccess)
      e->stats.pyramid_ms = ms;
  }
  *out = e->stats;
  return LK_ERROR_NONE;
}

int lk_get_sector_stats(lk_engine *e, uint32_t *out) {
  if (!e || !out)
    return LK_ERROR_BAD_DOMAIN;
  if (!e->committed)
    return e->fail(LK_ERROR_BAD_DOMAIN, "lk_get_sector_stats: no committed sectors");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(out, e->d_stats.p, 4 * (size_t)e->S * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return LK_ERROR_NONE;
}

} // extern "C"
