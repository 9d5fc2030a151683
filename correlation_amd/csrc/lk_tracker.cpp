// lk_tracker.cpp - sequence / tracking bookkeeping and frame loop (include/lk_tracker.h).
//
// Host code only.  The arithmetic that decides sector positions and guesses is float32 in the
// reference's operation order (this file is compiled with -ffp-contract=off like the rest
// of the library); the report goes through std::ostream with default flags, as the
// reference's does (manager_class.cpp:2430-2471).
#include "../../include/lk_tracker.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <charconv>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <future>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "lk_roi.hpp"

namespace {
const float kPI = 3.14159265359f; // parameters.hpp:23

int n_params(int model) { return model == LK_FM_U ? 1 : model == LK_FM_UV ? 2 : model == LK_FM_UVQ ? 3 : 6; }
} // namespace

// The per-sector loops of a frame (positions and guesses, update_results, report rows) touch
// one frame_results record each: on grids of tens of thousands of sectors they take longer
// than the solve itself (50 176 sectors: 1.6 + 1.6 ms against 2.3 ms on the GPU), so they are
// split into blocks of sectors over a few helper threads that live as long as the tracker.
// Anything order-dependent (the float sums of update_global_results, the first-error scan of
// the stop policy, the order of the report rows) stays sequential or is joined in sector order.
class Workers {
public:
  ~Workers() {
    {
      std::lock_guard<std::mutex> lock(mu);
      quit = true;
    }
    go.notify_all();
    for (std::thread &t : threads)
      t.join();
  }
  // fn(first, last) over [0, n) in at most `max_blocks` blocks of at least `min_block` items
  void run(size_t n, size_t min_block, const std::function<void(size_t, size_t)> &fn) {
    size_t blocks = std::min<size_t>(n / std::max<size_t>(min_block, 1), kMaxThreads);
    if (blocks < 2) {
      fn(0, n);
      return;
    }
    if (threads.empty()) {
      size_t hw = std::thread::hardware_concurrency();
      size_t helpers = std::min<size_t>(kMaxThreads, hw > 1 ? hw : 1) - 1;
      for (size_t i = 0; i < helpers; ++i)
        threads.emplace_back([this, i] { loop(i + 1); });
    }
    blocks = std::min(blocks, threads.size() + 1);
    if (blocks < 2) {
      fn(0, n);
      return;
    }
    {
      std::lock_guard<std::mutex> lock(mu);
      job = &fn;
      job_n = n;
      job_blocks = blocks;
      pending = blocks - 1;
      ++generation;
    }
    go.notify_all();
    fn(0, n / blocks);
    std::unique_lock<std::mutex> lock(mu);
    done.wait(lock, [this] { return pending == 0; });
    job = nullptr;
  }

private:
  static constexpr size_t kMaxThreads = 32;
  void loop(size_t index) {
    size_t seen = 0;
    for (;;) {
      const std::function<void(size_t, size_t)> *fn = nullptr;
      size_t n = 0, blocks = 0;
      {
        std::unique_lock<std::mutex> lock(mu);
        go.wait(lock, [&] { return quit || generation != seen; });
        if (quit)
          return;
        seen = generation;
        if (index >= job_blocks)
          continue; // this round needs fewer blocks than there are helpers
        fn = job;
        n = job_n;
        blocks = job_blocks;
      }
      (*fn)(n * index / blocks, n * (index + 1) / blocks);
      std::lock_guard<std::mutex> lock(mu);
      if (--pending == 0)
        done.notify_one();
    }
  }
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable go, done;
  const std::function<void(size_t, size_t)> *job = nullptr;
  size_t job_n = 0, job_blocks = 0, pending = 0, generation = 0;
  bool quit = false;
};

struct lk_tracker {
  lk_tracker_config cfg{};
  int P = 0;
  bool domain_set = false;
  // rectangularDomainStruct / annularDomainStruct / blobDomainStruct
  float x_begin = 0, y_begin = 0, x_end = 0, y_end = 0, x_center = 0, y_center = 0;
  int hs = 0, vs = 0;
  float r_inside = 0, r_outside = 0;
  int rs = 0, as = 0;
  std::vector<float> contour;
  int results_i = 0, results_j = 0;
  std::vector<lk_frame_result> res, before; // frame_results[]; state before begin_frame
  std::string report;                 // the CSV header
  std::vector<std::string> report_blocks; // + the rows, one block per formatting thread and frame
  std::vector<lk_frame_result> report_res; // the records the rows in the making are read from
  std::future<void> report_job;            // formats report_res into the newest report_blocks
  bool report_enabled = true;
  std::string err;
  bool begun = false;
  Workers workers, report_workers;
  size_t min_block = 4096; // sectors per block of the threaded loops (LK_TRACKER_MIN_BLOCK: test hook)
  // scratch of lk_sequence_frame (kept: three multi-megabyte buffers per frame otherwise)
  std::vector<lk_sector_command> seq_cmds;
  std::vector<float> seq_guesses;
  std::vector<lk_result> seq_results;
  std::vector<float> global_terms_scratch;

  int fail(int code, const char *what) {
    err = what;
    return code;
  }
};

// std::ostream << float / int / bool with default flags, which is what the reference's report
// stream does: "%g" with 6 significant digits == std::to_chars(general, 6); an ostringstream
// round trip per number would cost more than the solve on 50 000-sector grids.
//
// Fast path: the float is widened to double (exact), scaled to a 6-digit integer by ONE
// correctly rounded multiplication or division with an exact power of ten (|error| < 2e-10),
// and rounded; whenever that product is within 1e-6 of a rounding boundary - which includes
// every exact tie - or the decade is outside the exact powers, std::to_chars decides.
static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

static void put_slow(std::string &r, float v) {
  char buf[48];
  auto res = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::general, 6);
  r.append(buf, res.ptr);
}

static void put(std::string &r, float v) {
  if (std::isnan(v)) { // num_put prints what printf does
    r += std::signbit(v) ? "-nan" : "nan";
    return;
  }
  const double a = std::fabs((double)v);
  if (!(a >= 1e-16 && a < 1e16)) { // zero, infinities, far decades
    put_slow(r, v);
    return;
  }
  int k = (int)std::floor(std::log10(a)); // decade estimate, corrected against the exact table
  if (k >= 0 ? a < kPow10[k] : a < 1.0 / kPow10[-k])
    --k; // (1/10^n is inexact: the boundary test below still sends doubtful cases to to_chars)
  const int shift = 5 - k; // digits = a * 10^shift in [1e5, 1e6)
  if (shift > 22 || shift < -22) {
    put_slow(r, v);
    return;
  }
  const double scaled = shift >= 0 ? a * kPow10[shift] : a / kPow10[-shift];
  const double fl = std::floor(scaled), frac = scaled - fl;
  if (!(scaled >= 100000.0 && scaled < 999999.0) || std::fabs(frac - 0.5) < 1e-6) {
    put_slow(r, v);
    return;
  }
  uint32_t digits = (uint32_t)fl + (frac > 0.5 ? 1u : 0u); // 100000 .. 999999
  char d[6];
  for (int i = 5; i >= 0; --i) {
    d[i] = (char)('0' + digits % 10);
    digits /= 10;
  }
  int nd = 6;
  while (nd > 1 && d[nd - 1] == '0')
    --nd; // %g drops trailing zeros
  char buf[24];
  char *o = buf;
  if (std::signbit(v))
    *o++ = '-';
  if (k >= -4 && k < 6) { // fixed notation
    if (k >= 0) {
      for (int i = 0; i <= k; ++i)
        *o++ = i < nd ? d[i] : '0';
      if (nd > k + 1) {
        *o++ = '.';
        for (int i = k + 1; i < nd; ++i)
          *o++ = d[i];
      }
    } else {
      *o++ = '0';
      *o++ = '.';
      for (int i = -1; i > k; --i)
        *o++ = '0';
      for (int i = 0; i < nd; ++i)
        *o++ = d[i];
    }
  } else { // d.ddddde+XX
    *o++ = d[0];
    if (nd > 1) {
      *o++ = '.';
      for (int i = 1; i < nd; ++i)
        *o++ = d[i];
    }
    *o++ = 'e';
    *o++ = k < 0 ? '-' : '+';
    const int ak = k < 0 ? -k : k;
    *o++ = (char)('0' + ak / 10);
    *o++ = (char)('0' + ak % 10);
  }
  r.append(buf, o);
}
static void put(std::string &r, int v) {
  char buf[16];
  auto res = std::to_chars(buf, buf + sizeof(buf), v);
  r.append(buf, res.ptr);
}

static void initialize_report(lk_tracker *t) { // manager_class.cpp:2473-2525
  std::string &r = t->report;
  r.clear();
  t->report_blocks.clear();
  for (const char *name : {"Frame#", "und_file_string", "def_file_string", "und_global_center_x",
                           "und_global_center_y", "und_center_x", "und_center_y", "def_global_center_x",
                           "def_global_center_y", "def_center_x", "def_center_y"}) {
    r += name;
    r += ',';
  }
  for (const char *stem : {"parameter_", "Initial_guess_"})
    for (int p = 0; p < t->P; ++p) {
      r += stem;
      put(r, p);
      r += ',';
    }
  for (const char *name : {"und_global_angle(rad)", "def_global_angle(rad)", "und_angle(rad)", "def_angle(rad)",
                           "def_angle(deg)", "chi", "number_of_points", "iterations", "error_status"}) {
    r += name;
    r += ',';
  }
  r += "error_code\n";
}

// one report row per sector of [first, last), appended to r
static void report_rows(const lk_tracker *t, const lk_frame_result *res, size_t first, size_t last, int frame,
                        const char *und, const char *def, std::string &r) {
  for (size_t i = first; i < last; ++i) {
    const lk_frame_result &s = res[i];
    put(r, frame);
    r += ',';
    r += und;
    r += ',';
    r += def;
    r += ',';
    for (float v : {s.und_global_center_x, s.und_global_center_y, s.und_center_x, s.und_center_y,
                    s.def_global_center_x, s.def_global_center_y, s.def_center_x, s.def_center_y}) {
      put(r, v);
      r += ',';
    }
    for (int p = 0; p < t->P; ++p) {
      put(r, s.resulting_parameters[p]);
      r += ',';
    }
    for (int p = 0; p < t->P; ++p) {
      put(r, s.initial_guess[p]);
      r += ',';
    }
    for (float v : {s.und_global_angle, s.def_global_angle, s.und_angle, s.def_angle, s.def_angle * 180 / kPI,
                    s.chi}) {
      put(r, v);
      r += ',';
    }
    put(r, s.number_of_points);
    r += ',';
    put(r, s.iterations);
    r += ',';
    put(r, s.error_status != 0 ? 1 : 0);
    r += ',';
    put(r, s.error_code);
    r += '\n';
  }
}

static void wait_for_report(lk_tracker *t) {
  if (t->report_job.valid())
    t->report_job.get();
}

// manager_class.cpp:2430-2471.  ~35 number conversions per sector: on grids of tens of
// thousands of sectors the rows are formatted by a few threads, each on its own block of
// sectors, and joined in sector order (the text is the same as the sequential loop's) - and
// they are formatted BEHIND the caller: the frame's records are copied, a background job
// formats the copy with helper threads of its own while the caller goes on to the next pair;
// whoever needs the text (lk_tracker_report, the next frame's rows, the destructor) waits.
static void add_frame_to_report(lk_tracker *t, int frame, const char *und, const char *def) {
  if (!t->report_enabled)
    return;
  wait_for_report(t);
  const size_t S = t->res.size(), kRowsPerBlock = std::min<size_t>(1024, t->min_block), kMaxBlocks = 32;
  const size_t blocks = std::max<size_t>(1, std::min(S / kRowsPerBlock, kMaxBlocks));
  const size_t base = t->report_blocks.size();
  t->report_blocks.resize(base + blocks); // blocks are joined only when the text is asked for
  // fixed row blocks (whatever the number of threads that format them): block b = rows
  // [S*b/blocks, S*(b+1)/blocks)
  auto format = [t, S, blocks, base, frame](const lk_frame_result *res, const std::string &u, const std::string &d,
                                            Workers &pool) {
    std::string *part = &t->report_blocks[base];
    pool.run(blocks, 1, [&](size_t b0, size_t b1) {
      for (size_t b = b0; b < b1; ++b) {
        const size_t first = S * b / blocks, last = S * (b + 1) / blocks;
        std::string rows; // (a local: the block headers sit side by side in one cache line)
        rows.reserve((last - first) * (size_t)(96 + 26 * t->P) + 64);
        report_rows(t, res, first, last, frame, u.c_str(), d.c_str(), rows);
        part[b] = std::move(rows);
      }
    });
  };
  if (blocks < 2) { // small grids: not worth a copy and a thread
    format(t->res.data(), und, def, t->workers);
    return;
  }
  t->report_res = t->res;
  t->report_job = std::async(std::launch::async, [t, format, u = std::string(und), d = std::string(def)] {
    format(t->report_res.data(), u, d, t->report_workers);
  });
}

// the Lagrangian branch shared by adjust_rectangular/annular/blob_domain
static void lagrangian_roll(lk_frame_result &s, bool annulus) {
  s.und_global_center_x = s.def_global_center_x;
  s.und_global_center_y = s.def_global_center_y;
  if (annulus) {
    s.und_global_ro = s.def_global_ro;
    s.und_global_ri = s.def_global_ri;
    s.und_global_e = s.def_global_e;
  }
  s.und_global_angle = s.def_global_angle;
  s.past_und_center_x = s.und_center_x;
  s.past_und_center_y = s.und_center_y;
  s.und_center_x = s.def_center_x;
  s.und_center_y = s.def_center_y;
  if (annulus)
    s.und_e = s.def_e;
  s.und_angle = s.def_angle;
}

static void adjust_initial_guess(lk_tracker *t, lk_frame_result &s, int frame, float *guess_out) { // :2602-2707
  const int P = t->P;
  const float *g = t->cfg.global_guess;
  if (frame == 0) {
    for (int p = 0; p < P; ++p)
      s.initial_guess[p] = g[p];
    float dx = s.und_center_x - s.und_global_center_x;
    float dy = s.und_center_y - s.und_global_center_y;
    if (t->cfg.fitting_model != LK_FM_UVUXUYVXVY) {
      // the reference reads global_initial_guess[2] for fm_U / fm_UV as well (outside their
      // arrays); taken as 0 here, and nothing is written past the model's own parameters
      float Vx = P > 2 ? g[2] : 0.f;
      s.initial_guess[0] += -dy * Vx;
      if (P > 1)
        s.initial_guess[1] += dx * Vx;
    } else {
      float Ux = g[2], Uy = g[3], Vx = g[4], Vy = g[5];
      s.initial_guess[0] += dx * Ux + dy * Uy;
      s.initial_guess[1] += dx * Vx + dy * Vy;
    }
    for (int p = 0; p < P; ++p)
      s.previous_resulting_parameters[p] = s.initial_guess[p];
  } else {
    if (t->cfg.deformation == LK_DEF_EULERIAN && t->cfg.reference_image == LK_REF_FIRST) {
      for (int i = 0; i < P; ++i)
        s.initial_guess[i] =
            s.resulting_parameters[i] + (s.resulting_parameters[i] - s.previous_resulting_parameters[i]);
    } else {
      for (int i = 0; i < P; ++i)
        s.initial_guess[i] = s.resulting_parameters[i];
    }
    for (int p = 0; p < P; ++p)
      s.previous_resulting_parameters[p] = s.resulting_parameters[p];
  }
  for (int p = 0; p < 6; ++p)
    guess_out[p] = p < P ? s.initial_guess[p] : 0.f;
}

static void update_results(lk_tracker *t, lk_frame_result &s, const lk_result &r) { // :2312-2428
  const int P = t->P;
  s.chi = r.chi;
  s.number_of_points = r.numberOfPoints;
  s.iterations = r.iterations;
  s.error_code = r.errorCode;
  s.error_status = r.errorCode != LK_ERROR_NONE;
  s.und_center_x = r.undCenterX;
  s.und_center_y = r.undCenterY;
  for (int p = 0; p < P; ++p)
    s.resulting_parameters[p] = r.resultingParameters[p];
  const float *m = s.resulting_parameters;
  switch (t->cfg.fitting_model) {
  case LK_FM_UVQ:
    s.def_angle = m[2] + s.und_angle;
    break;
  case LK_FM_UVUXUYVXVY: // best_rotation_UVUxUyVxVy, parameters.cpp:55-58
    s.def_angle = std::atan2((m[4] - m[3]), (m[2] + m[5] + 2.f)) + s.und_angle;
    break;
  default:
    s.def_angle = 0.f;
    break;
  }
  s.def_e = 0.f;
  // distortX(x, y, x, y, ...): the model applied to the centre about itself
  // (interpolation_class.cpp:3-43)
  const float x = s.und_center_x, y = s.und_center_y, cx = x, cy = y;
  switch (t->cfg.fitting_model) {
  case LK_FM_U:
    s.def_center_x = x + m[0];
    s.def_center_y = y;
    break;
  case LK_FM_UV:
    s.def_center_x = x + m[0];
    s.def_center_y = y + m[1];
    break;
  case LK_FM_UVQ:
    s.def_center_x = x + m[0] - (y - cy) * m[2];
    s.def_center_y = y + m[1] + (x - cx) * m[2];
    break;
  default:
    s.def_center_x = x + m[0] + (x - cx) * m[2] + (y - cy) * m[3];
    s.def_center_y = y + m[1] + (x - cx) * m[4] + (y - cy) * m[5];
    break;
  }
}

// :2709-2753.  The weighted sums run over the sectors in order, in float, like the reference's
// loop; `terms` ([S][5]: the products and n, filled by the update_results pass) keeps that
// sequential part on a compact array instead of a second walk over the 200-byte records.
static void update_global_results(lk_tracker *t, const std::vector<float> &terms) {
  float average_angle = 0.f, average_center_x = 0.f, average_center_y = 0.f, average_e = 0.f, total_n = 0.f;
  const size_t S = t->res.size();
  for (size_t k = 0; k < S; ++k) {
    const float *q = &terms[5 * k];
    average_angle += q[0];
    average_center_x += q[1];
    average_center_y += q[2];
    average_e += q[3];
    total_n += q[4];
  }
  average_angle = average_angle / total_n;
  average_center_x = average_center_x / total_n;
  average_center_y = average_center_y / total_n;
  average_e = average_e / total_n;
  float und_ro = t->res[0].und_global_ro, und_ri = t->res[0].und_global_ri;
  float def_ri = 1.f + average_e * (und_ro / und_ri - 1.f);
  t->workers.run(S, t->min_block, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; ++k) {
      lk_frame_result &s = t->res[k];
      s.def_global_angle = average_angle;
      s.def_global_center_x = average_center_x;
      s.def_global_center_y = average_center_y;
      s.def_global_e = average_e;
      s.def_global_ro = s.und_global_ro;
      s.def_global_ri = def_ri;
    }
  });
}

static void global_terms(const lk_frame_result &s, float *q) {
  const float n = (float)s.number_of_points;
  q[0] = s.def_angle * n;
  q[1] = s.def_center_x * n;
  q[2] = s.def_center_y * n;
  q[3] = s.def_e * n;
  q[4] = n;
}

extern "C" {

int lk_tracker_create(const lk_tracker_config *cfg, lk_tracker **out) {
  if (!cfg || !out)
    return LK_ERROR_BAD_DOMAIN;
  *out = nullptr;
  if (cfg->fitting_model < LK_FM_U || cfg->fitting_model > LK_FM_UVUXUYVXVY || cfg->domain_type < 0 ||
      cfg->domain_type > 2 || cfg->deformation < 0 || cfg->deformation > 2 || cfg->reference_image < 0 ||
      cfg->reference_image > 1 || cfg->error_mode < 0 || cfg->error_mode > 2)
    return LK_ERROR_BAD_DOMAIN;
  lk_tracker *t = new lk_tracker();
  t->cfg = *cfg;
  t->P = n_params(cfg->fitting_model);
  if (const char *f = std::getenv("LK_TRACKER_MIN_BLOCK"))
    t->min_block = (size_t)std::max(1, std::atoi(f));
  initialize_report(t);
  *out = t;
  return LK_ERROR_NONE;
}

void lk_tracker_destroy(lk_tracker *t) {
  if (!t)
    return;
  wait_for_report(t);
  delete t;
}

const char *lk_tracker_last_error(const lk_tracker *t) { return t ? t->err.c_str() : "null tracker"; }

static void reset_results(lk_tracker *t, int i, int j) {
  t->results_i = i;
  t->results_j = j;
  lk_frame_result z;
  std::memset(&z, 0, sizeof(z));
  t->res.assign((size_t)i * (size_t)j, z);
  t->domain_set = true;
  t->begun = false;
}

int lk_tracker_set_rect_domain(lk_tracker *t, float x_begin, float y_begin, float x_end, float y_end,
                               float x_center, float y_center, int hs, int vs) {
  if (!t)
    return LK_ERROR_BAD_DOMAIN;
  if (t->cfg.domain_type != LK_DOMAIN_RECT || hs < 1 || vs < 1)
    return t->fail(LK_ERROR_BAD_DOMAIN, "lk_tracker_set_rect_domain: not a rectangular tracker / bad subdivisions");
  t->x_begin = x_begin, t->y_begin = y_begin, t->x_end = x_end, t->y_end = y_end;
  t->x_center = x_center, t->y_center = y_center;
  t->hs = hs, t->vs = vs;
  reset_results(t, hs, vs);
  return LK_ERROR_NONE;
}

int lk_tracker_set_annular_domain(lk_tracker *t, float r_inside, float r_outside, float x_center,
                                  float y_center, int rs, int as) {
  if (!t)
    return LK_ERROR_BAD_DOMAIN;
  if (t->cfg.domain_type != LK_DOMAIN_ANNULAR || rs < 1 || as < 1)
    return t->fail(LK_ERROR_BAD_DOMAIN, "lk_tracker_set_annular_domain: not an annular tracker / bad subdivisions");
  t->r_inside = r_inside, t->r_outside = r_outside, t->x_center = x_center, t->y_center = y_center;
  t->rs = rs, t->as = as;
  reset_results(t, rs, as);
  return LK_ERROR_NONE;
}

int lk_tracker_set_blob_domain(lk_tracker *t, const float *contour_xy, int n_vertices, float x_center,
                               float y_center) {
  if (!t)
    return LK_ERROR_BAD_DOMAIN;
  if (t->cfg.domain_type != LK_DOMAIN_BLOB || !contour_xy || n_vertices < 3) // manager_class.cpp:1006,1116-1120
    return t->fail(LK_ERROR_BAD_DOMAIN, "lk_tracker_set_blob_domain: a blob needs at least 3 contour points");
  t->contour.assign(contour_xy, contour_xy + 2 * (size_t)n_vertices);
  t->x_center = x_center, t->y_center = y_center;
  reset_results(t, 1, 1);
  return LK_ERROR_NONE;
}

int lk_tracker_sector_count(const lk_tracker *t) { return t ? (int)t->res.size() : 0; }

int lk_tracker_enable_report(lk_tracker *t, int enabled) {
  if (!t)
    return LK_ERROR_BAD_DOMAIN;
  t->report_enabled = enabled != 0;
  return LK_ERROR_NONE;
}

int lk_tracker_blob_contour(const lk_tracker *t, const float **contour_xy, int *n_vertices) {
  if (!t || t->contour.empty())
    return LK_ERROR_BAD_DOMAIN;
  if (contour_xy)
    *contour_xy = t->contour.data();
  if (n_vertices)
    *n_vertices = (int)(t->contour.size() / 2);
  return LK_ERROR_NONE;
}

int lk_tracker_begin_frame(lk_tracker *t, int frame, lk_sector_command *commands, float *guesses) {
  if (!t)
    return LK_ERROR_BAD_DOMAIN;
  if (!t->domain_set || !commands || !guesses || frame < 0)
    return t->fail(LK_ERROR_BAD_DOMAIN, "lk_tracker_begin_frame: no domain / null buffers");
  if (t->cfg.error_mode != LK_ERRMODE_CONTINUE)
    t->before = t->res; // what lk_tracker_end_frame gives back to sectors a stopped frame never reached
  const int S = (int)t->res.size();
  const bool eulerian = t->cfg.deformation == LK_DEF_EULERIAN;
  auto later_frames = [&](lk_frame_result &s, lk_sector_command &c) { // manager_class.cpp:354-419
    if (eulerian) {
      c.kind = LK_SECTOR_KEEP;
    } else if (t->cfg.deformation == LK_DEF_STRICT_LAGRANGIAN) {
      c.kind = LK_SECTOR_REWARP;
    } else {
      c.kind = LK_SECTOR_TRANSLATE;
      c.offset_x = s.und_center_x - s.past_und_center_x;
      c.offset_y = s.und_center_y - s.past_und_center_y;
    }
  };
  std::memset(commands, 0, sizeof(lk_sector_command) * (size_t)S);
  switch (t->cfg.domain_type) {
  case LK_DOMAIN_RECT: { // manager_class.cpp:274-310 + adjust_rectangular_domain :2018-2090
    lkroi::RectGrid g = lkroi::rect_grid(t->x_begin, t->y_begin, t->x_end, t->y_end, t->hs, t->vs);
    if (g.xdim < 0 || g.ydim < 0)
      return t->fail(LK_ERROR_BAD_DOMAIN, "lk_tracker_begin_frame: domain smaller than the grid");
    t->workers.run((size_t)S, t->min_block, [&](size_t k0, size_t k1) {
      for (size_t kk = k0; kk < k1; ++kk) {
        const int k = (int)kk, i = k / t->vs, j = k % t->vs; // iSector = i*vs + j
        lk_frame_result &s = t->res[(size_t)k];
        lk_sector_command &c = commands[k];
        if (frame == 0) {
          s.und_global_center_x = t->x_center;
          s.und_global_center_y = t->y_center;
          s.und_global_angle = 0.f;
          s.und_global_e = 0.f;
          s.und_center_x = (float)g.cx[i];
          s.und_center_y = (float)g.cy[j];
          s.und_angle = 0.f;
          s.past_und_center_x = s.und_center_x;
          s.past_und_center_y = s.und_center_y;
        } else if (!eulerian) {
          lagrangian_roll(s, false);
        }
        const int center_x = (int)(s.und_center_x + 0.5f), center_y = (int)(s.und_center_y + 0.5f);
        adjust_initial_guess(t, s, frame, guesses + 6 * (size_t)k);
        c.use_center = 1;
        c.center_x = (float)center_x;
        c.center_y = (float)center_y;
        if (frame == 0) {
          c.kind = LK_SECTOR_RECT;
          c.x0 = center_x - g.xdim, c.y0 = center_y - g.ydim, c.x1 = center_x + g.xdim, c.y1 = center_y + g.ydim;
        } else {
          later_frames(s, c);
        }
      }
    });
    break;
  }
  case LK_DOMAIN_ANNULAR: { // manager_class.cpp:553-600 + adjust_annular_domain :2092-2237
    const float ri = t->r_inside, ro = t->r_outside;
    const float dr = (ro - ri) / (float)t->rs;
    const float da = 2.f * kPI / (float)t->as;
    for (int i = 0; i < t->rs; ++i)
      for (int j = 0; j < t->as; ++j) {
        const int k = i * t->as + j;
        lk_frame_result &s = t->res[(size_t)k];
        lk_sector_command &c = commands[k];
        if (frame == 0) {
          s.und_global_center_x = t->x_center;
          s.und_global_center_y = t->y_center;
          s.und_global_ro = ro;
          s.und_global_ri = ri;
          s.und_global_e = 0.f;
          s.und_global_angle = 0.f;
          if (t->as > 1) {
            float center_angle = 0 + j * da + da / 2.f;
            float center_r = ri + i * dr + dr / 2.f;
            s.und_center_x = s.und_global_center_x + center_r * (float)cos((double)center_angle);
            s.und_center_y = s.und_global_center_y + center_r * (float)sin((double)center_angle);
          } else {
            s.und_center_x = s.und_global_center_x;
            s.und_center_y = s.und_global_center_y;
          }
          s.past_und_center_x = s.und_center_x;
          s.past_und_center_y = s.und_center_y;
          s.und_e = 0.f;
          s.und_angle = 0.f;
        } else if (!eulerian) {
          lagrangian_roll(s, true);
        }
        const float r = ri + i * dr;
        const float a = s.und_global_angle + j * da;
        adjust_initial_guess(t, s, frame, guesses + 6 * (size_t)k);
        c.use_center = 0; // Newton_Raphson(p, n, xy): float mean of the samples (:703-705)
        if (frame == 0) {
          c.kind = LK_SECTOR_ANNULAR;
          c.r = r, c.dr = dr, c.a = a, c.da = da, c.cx = s.und_global_center_x, c.cy = s.und_global_center_y;
          c.as = t->as;
        } else {
          later_frames(s, c);
        }
      }
    break;
  }
  default: { // blob: manager_class.cpp:1000-1031 + adjust_blob_domain :2239-2310
    lk_frame_result &s = t->res[0];
    lk_sector_command &c = commands[0];
    if (frame == 0) {
      s.und_global_center_x = t->x_center;
      s.und_global_center_y = t->y_center;
      s.und_global_angle = 0.f;
      s.und_center_x = t->x_center;
      s.und_center_y = t->y_center;
      s.past_und_center_x = s.und_center_x;
      s.past_und_center_y = s.und_center_y;
      s.und_angle = 0.f;
    } else if (!eulerian) {
      lagrangian_roll(s, false);
    }
    adjust_initial_guess(t, s, frame, guesses);
    c.use_center = 0;
    if (frame == 0)
      c.kind = LK_SECTOR_BLOB;
    else
      later_frames(s, c);
    break;
  }
  }
  t->begun = true;
  return LK_ERROR_NONE;
}

int lk_tracker_end_frame(lk_tracker *t, int frame, const char *und_name, const char *def_name,
                         const lk_result *results, int *first_unsolved, int *stop_sequence) {
  if (!t)
    return LK_ERROR_BAD_DOMAIN;
  if (!t->begun || !results)
    return t->fail(LK_ERROR_BAD_DOMAIN, "lk_tracker_end_frame: lk_tracker_begin_frame has not been called");
  const int S = (int)t->res.size();
  const bool stop_mode = t->cfg.error_mode != LK_ERRMODE_CONTINUE;
  // the sector loop after the solve (manager_class.cpp:452-547): `error` is overwritten by
  // every sector, and under stopAll / stopFrame the loop ends after the first failing one
  int k = S;
  if (stop_mode)
    for (int u = 0; u < S; ++u)
      if (results[u].errorCode != LK_ERROR_NONE) {
        k = u + 1;
        break;
      }
  std::vector<float> &terms = t->global_terms_scratch;
  terms.resize(5 * (size_t)S);
  t->workers.run((size_t)k, t->min_block, [&](size_t u0, size_t u1) {
    for (size_t u = u0; u < u1; ++u) {
      update_results(t, t->res[u], results[u]);
      global_terms(t->res[u], &terms[5 * u]);
    }
  });
  const bool error = k > 0 && t->res[(size_t)k - 1].error_status != 0;
  for (int u = k; u < S; ++u) { // not reached by the reference's loop: as before this frame
    t->res[(size_t)u] = t->before[(size_t)u];
    global_terms(t->res[(size_t)u], &terms[5 * (size_t)u]);
  }
  if (first_unsolved)
    *first_unsolved = k;
  update_global_results(t, terms);
  add_frame_to_report(t, frame, und_name ? und_name : "", def_name ? def_name : "");
  if (stop_sequence) // manager_class.cpp:1485-1486
    *stop_sequence = error && t->cfg.error_mode == LK_ERRMODE_STOP_ALL;
  t->begun = false;
  return LK_ERROR_NONE;
}

int lk_tracker_get_results(const lk_tracker *t, lk_frame_result *out) {
  if (!t || !out)
    return LK_ERROR_BAD_DOMAIN;
  std::memcpy(out, t->res.data(), t->res.size() * sizeof(lk_frame_result));
  return LK_ERROR_NONE;
}

int lk_tracker_report(const lk_tracker *t, char *buf, size_t cap, size_t *needed) {
  if (!t)
    return LK_ERROR_BAD_DOMAIN;
  wait_for_report(const_cast<lk_tracker *>(t)); // the last frame's rows may still be in the making
  size_t size = t->report.size();
  for (const std::string &x : t->report_blocks)
    size += x.size();
  if (needed)
    *needed = size + 1;
  if (buf && cap > 0) {
    size_t at = 0;
    auto put_block = [&](const std::string &x) {
      const size_t n = std::min(x.size(), cap - 1 - at);
      std::memcpy(buf + at, x.data(), n);
      at += n;
    };
    put_block(t->report);
    for (const std::string &x : t->report_blocks)
      put_block(x);
    buf[at] = 0;
  }
  return LK_ERROR_NONE;
}

// ------------------------------------------------------------------------------------
// frame loop on an engine
// ------------------------------------------------------------------------------------
// frame 0: the tracker's commands become the engine's sectors
static int register_sectors(lk_engine *e, lk_tracker *t, const std::vector<lk_sector_command> &cmds) {
  const int S = (int)cmds.size();
  int rc = lk_clear_sectors(e);
  bool annular = S > 0;
  for (int s = 0; s < S && annular; ++s)
    annular = cmds[(size_t)s].kind == LK_SECTOR_ANNULAR && cmds[(size_t)s].as == cmds[0].as;
  if (annular && !rc) { // every sector of the annulus in one call: rasterised on several host threads
    std::vector<float> q(6 * (size_t)S);
    for (int s = 0; s < S; ++s) {
      const lk_sector_command &c = cmds[(size_t)s];
      const float v[6] = {c.r, c.dr, c.a, c.da, c.cx, c.cy};
      std::memcpy(&q[6 * (size_t)s], v, sizeof(v));
    }
    rc = lk_set_sectors_annular(e, 0, S, q.data(), cmds[0].as);
  }
  for (int s = 0; s < S && !rc && !annular; ++s) {
    const lk_sector_command &c = cmds[(size_t)s];
    switch (c.kind) {
    case LK_SECTOR_RECT: rc = lk_set_sector_rect(e, s, c.x0, c.y0, c.x1, c.y1); break;
    case LK_SECTOR_ANNULAR: rc = lk_set_sector_annular(e, s, c.r, c.dr, c.a, c.da, c.cx, c.cy, c.as); break;
    case LK_SECTOR_BLOB: {
      const float *xy = nullptr;
      int n = 0;
      rc = lk_tracker_blob_contour(t, &xy, &n);
      if (!rc)
        rc = lk_set_sector_blob(e, s, xy, n);
      break;
    }
    default: rc = LK_ERROR_BAD_DOMAIN; break;
    }
  }
  if (!rc)
    rc = lk_commit_sectors(e);
  return rc;
}

int lk_sequence_frame(lk_engine *e, lk_tracker *t, int frame, const char *und_name, const char *def_name,
                      int *stop_sequence) {
  if (!e || !t)
    return LK_ERROR_BAD_DOMAIN;
  const int S = lk_tracker_sector_count(t);
  std::vector<lk_sector_command> &cmds = t->seq_cmds;
  std::vector<float> &guesses = t->seq_guesses;
  cmds.resize((size_t)S);
  guesses.resize(6 * (size_t)S);
  int rc = lk_tracker_begin_frame(t, frame, cmds.data(), guesses.data());
  if (rc)
    return rc;
  bool moved = false;
  if (frame == 0) {
    rc = register_sectors(e, t, cmds);
  } else if (S > 0 && cmds[0].kind != LK_SECTOR_KEEP) {
    moved = true;
    std::vector<float> off(2 * (size_t)S), cen(2 * (size_t)S);
    for (int s = 0; s < S; ++s) {
      off[2 * (size_t)s] = cmds[(size_t)s].offset_x;
      off[2 * (size_t)s + 1] = cmds[(size_t)s].offset_y;
      cen[2 * (size_t)s] = cmds[(size_t)s].center_x;
      cen[2 * (size_t)s + 1] = cmds[(size_t)s].center_y;
    }
    const float *centers = cmds[0].use_center ? cen.data() : nullptr;
    rc = cmds[0].kind == LK_SECTOR_TRANSLATE ? lk_translate_sectors(e, off.data(), centers)
                                             : lk_rewarp_sectors(e, centers);
  }
  if (rc)
    return rc;
  std::vector<lk_result> &results = t->seq_results;
  results.resize((size_t)S);
  rc = lk_correlate_all(e, guesses.data(), results.data());
  if (rc)
    return rc;
  int first_unsolved = S;
  rc = lk_tracker_end_frame(t, frame, und_name, def_name, results.data(), &first_unsolved, stop_sequence);
  if (!rc && first_unsolved < S && moved)
    rc = lk_restore_sectors(e, first_unsolved); // their samples stay where the previous frame left them (Eulerian: nothing moved)
  return rc;
}

int lk_sequence_run(lk_engine *e, lk_tracker *t, int n_frames, lk_frame_provider provider, void *user,
                    int *pairs_done) {
  if (pairs_done)
    *pairs_done = 0;
  if (!e || !t || !provider || n_frames < 2)
    return LK_ERROR_BAD_DOMAIN;
  struct Frame {
    const uint8_t *px = nullptr;
    int rows = 0, cols = 0, step = 0;
    std::string name;
  };
  auto fetch = [&](int index) {
    Frame f;
    const char *name = nullptr;
    f.px = provider(user, index, &f.rows, &f.cols, &f.step, &name);
    f.name = name ? name : ("frame" + std::to_string(index));
    return f;
  };
  Frame f0 = fetch(0), f1 = fetch(1);
  if (!f0.px || !f1.px)
    return LK_ERROR_BAD_DOMAIN;
  int rc = lk_set_image(e, LK_IMG_UND, f0.px, f0.rows, f0.cols, f0.step);
  if (!rc)
    rc = lk_set_image(e, LK_IMG_DEF, f1.px, f1.rows, f1.cols, f1.step);
  std::string und_name = f0.name, def_name = f1.name;
  const int pairs = n_frames - 1;
  // Eulerian description on a rectangular grid, no stop policy: the guess of pair k+1 is a function
  // of the engine-held results alone (lk_adjust_initial_guess on the device = adjust_initial_guess,
  // manager_class.cpp:2602-2707), so pair k+1 is launched as soon as pair k's records are back and
  // the tracker's bookkeeping of pair k (update_results, global results, report) runs BEHIND that
  // solve instead of between two solves.
  const char *sync_env = std::getenv("LK_SEQ_SYNC"); // test / comparison hook
  // (stopAll - the sequence ends with the first frame that has an error, manager_class.cpp:1485-1486 - only in windows: they
  // compute ahead and the frames behind the stopping one are discarded; no frame before it had an error, so nothing a later
  // frame was started from changes.  stopFrame skips the rest of a frame and goes on: the frames computed ahead would have
  // started from results the reference never had - it keeps the synchronous loop)
  const char *win_env0 = std::getenv("LK_SEQ_WINDOW");
  const bool windows_wanted = !(win_env0 && std::atoi(win_env0) <= 1) && n_frames > 2;
  const bool overlapped = !rc && t->cfg.deformation == LK_DEF_EULERIAN &&
                          (t->cfg.error_mode == LK_ERRMODE_CONTINUE || (t->cfg.error_mode == LK_ERRMODE_STOP_ALL && windows_wanted)) &&
                          t->cfg.domain_type == LK_DOMAIN_RECT && !(sync_env && std::atoi(sync_env) != 0);
  if (overlapped) {
    const char *check_env = std::getenv("LK_SEQ_CHECK"); // tests: device guesses == the tracker's, bit for bit
    const bool check = check_env && std::atoi(check_env) != 0;
    const int S = lk_tracker_sector_count(t);
    const bool velocity = t->cfg.reference_image == LK_REF_FIRST;
    float gg[6] = {0, 0, 0, 0, 0, 0};
    for (int p = 0; p < t->P; ++p)
      gg[p] = t->cfg.global_guess[p];
    std::vector<lk_sector_command> &cmds = t->seq_cmds;
    std::vector<float> &guesses = t->seq_guesses;
    std::vector<lk_result> &results = t->seq_results;
    cmds.resize((size_t)S);
    guesses.resize(6 * (size_t)S);
    results.resize((size_t)S);
    std::vector<float> device_guesses(check ? 6 * (size_t)S : 0);
    auto launch = [&](int k) {
      int r = lk_adjust_initial_guess(e, k, velocity ? 1 : 0, gg, t->x_center, t->y_center);
      if (!r && check)
        r = lk_get_guesses(e, device_guesses.data());
      return r ? r : lk_correlate_all_async(e);
    };
    auto verify = [&]() { // (check mode) what the device solved from is what the tracker would have sent
      return !check || std::memcmp(device_guesses.data(), guesses.data(), device_guesses.size() * sizeof(float)) == 0;
    };
    static const bool timing_env = [] { const char *f = std::getenv("LK_SEQ_TIMING"); return f && std::atoi(f) != 0; }(); // tuning: where the set-up goes
    const auto t_begin = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
      if (timing_env)
        std::fprintf(stderr, "lk_sequence_run: %8.3f ms  %s\n",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), what);
    };
    rc = lk_tracker_begin_frame(t, 0, cmds.data(), guesses.data());
    if (!rc)
      rc = register_sectors(e, t, cmds);
    mark("sectors registered and committed");
    // Windows of K pairs (LK_SEQ_WINDOW, default 16; 1 = pair by pair, below): the K deformed frames of a window are
    // resident in the engine's ring and ONE launch per size class solves them all, every sector moving on to its next
    // frame as soon as its own previous frame is done (lk_correlate_sequence_async) - the guess of pair k + 1 needs
    // nothing but the sector's own earlier results (manager_class.cpp:2677-2699).  While window w is being solved a helper
    // thread fetches and uploads the frames of window w + 1 into the other half of the ring (:1438-1447, K frames ahead
    // instead of one), and the tracker's bookkeeping of window w - 1 (update_results, global results, report rows) runs
    // on this thread.  Same records, same report text as the pair-by-pair loop in batch-invariant and reference-order
    // mode (tests/test_sequence_window_gpu.py).
    const char *win_env = std::getenv("LK_SEQ_WINDOW");
    const int K = std::min(pairs, win_env ? std::max(1, std::atoi(win_env)) : 16);
    if (!rc && (K >= 2 || t->cfg.error_mode == LK_ERRMODE_STOP_ALL)) {
      const int R = 2 * K;
      rc = lk_sequence_reserve(e, R);
      std::vector<std::string> frame_name((size_t)n_frames);
      frame_name[0] = f0.name;
      frame_name[1] = f1.name;
      if (!rc)
        rc = lk_sequence_set_frame(e, 0, f1.px, f1.rows, f1.cols, f1.step);
      auto upload = [&](int f) -> int { // frame f (f >= 1) -> ring slot (f - 1) % R
        Frame fr = fetch(f);
        if (!fr.px)
          return (int)LK_ERROR_BAD_DOMAIN;
        frame_name[(size_t)f] = fr.name;
        return lk_sequence_set_frame(e, (f - 1) % R, fr.px, fr.rows, fr.cols, fr.step);
      };
      mark("ring reserved, frame 1 in");
      // (page-locking a window's record buffer takes milliseconds: the first one beside the first window's uploads, the
      // second one beside the first window's solve - see the uploads of the next window below)
      std::future<int> records_ready;
      if (!rc)
        records_ready = std::async(std::launch::async, [&] { return lk_sequence_prepare_host_records(e, K); });
      for (int f = 2; f <= K && !rc; ++f)
        rc = upload(f);
      if (records_ready.valid()) {
        const int prc = records_ready.get();
        if (!rc)
          rc = prc;
      }
      mark("first window's frames uploaded");
      // (the records of a window stay where the engine's copy left them: its pinned host buffers alternate, so window w's are
      // still there while window w + 1 is solved - and gone when window w + 2 is launched: the bookkeeping of w runs before that)
      const lk_result *win[2] = {nullptr, nullptr};
      std::vector<float> win_guess[2], win_first[2]; // check mode: the guesses the device solved from (first frame: the guess kernel's)
      if (check)
        for (int b = 0; b < 2; ++b) {
          win_guess[b].resize((size_t)K * 6 * (size_t)S);
          win_first[b].resize(6 * (size_t)S);
        }
      std::future<int> next;
      int have_first = -1, have_n = 0, have_buf = 0; // a solved window whose bookkeeping is still to be done
      bool stopped = false;                          // stopAll: a frame had an error - the frames behind it are discarded
      auto bookkeeping = [&]() -> int {
        int r = LK_ERROR_NONE;
        for (int i = 0; i < have_n && !r && !stopped; ++i) {
          const int k = have_first + i;
          const std::string &und_k = t->cfg.reference_image == LK_REF_PREVIOUS ? frame_name[(size_t)k] : frame_name[0];
          if (k > 0) // (frame 0's begin_frame ran before the sectors were registered; its guesses are still in `guesses`)
            r = lk_tracker_begin_frame(t, k, cmds.data(), guesses.data());
          if (!r && check) {
            const float *dev = i > 0 ? win_guess[have_buf].data() + (size_t)i * 6 * (size_t)S : win_first[have_buf].data();
            if (std::memcmp(dev, guesses.data(), 6 * (size_t)S * sizeof(float)) != 0)
              r = t->fail(LK_ERROR_BAD_DOMAIN, "lk_sequence_run: device guesses differ from the tracker's");
          }
          int first_unsolved = S, stop = 0; // (stopAll: the frame with the first error ends the sequence)
          if (!r)
            r = lk_tracker_end_frame(t, k, und_k.c_str(), frame_name[(size_t)k + 1].c_str(), win[have_buf] + (size_t)i * (size_t)S,
                                     &first_unsolved, &stop);
          if (!r && pairs_done)
            *pairs_done = k + 1;
          stopped = !r && stop != 0;
        }
        have_n = 0;
        return r;
      };
      for (int first = 0, w = 0; first < pairs && !rc; first += K, ++w) {
        const int n = std::min(K, pairs - first), buf = w & 1;
        if (next.valid()) { // this window's frames (uploaded behind the previous window's solve)
          const int nrc = next.get();
          if (nrc) // error_multiThread in the reference (manager_class.cpp:1470-1475)
            rc = nrc;
        }
        if (!rc)
          rc = lk_adjust_initial_guess(e, first, velocity ? 1 : 0, gg, t->x_center, t->y_center);
        if (!rc && check)
          rc = lk_get_guesses(e, win_first[buf].data());
        bool launched = false;
        if (!rc) {
          rc = lk_correlate_sequence_async(e, (t->cfg.reference_image == LK_REF_PREVIOUS && first > 0) ? (first - 1) % R : -1, first % R, n,
                                           t->cfg.reference_image == LK_REF_PREVIOUS ? 1 : 0, velocity ? 1 : 0, check ? 3 : 1);
          launched = !rc;
        }
        if (!rc && first + K < pairs) {
          const int f_begin = first + K + 1, f_end = std::min(first + 2 * K, pairs);
          next = std::async(std::launch::async, [&, f_begin, f_end] {
            int r = lk_sequence_prepare_host_records(e, std::min(K, pairs - (f_begin - 1))); // (the next window's; a no-op from the third window on)
            for (int f = f_begin; f <= f_end && !r; ++f)
              r = upload(f);
            return r;
          });
        }
        if (!rc && have_n > 0) // the window before: its bookkeeping runs behind the solve just launched
          rc = bookkeeping();
        if (launched) {
          const int wrc = lk_wait_sequence(e, nullptr);
          if (!rc)
            rc = wrc;
          if (!rc)
            rc = lk_sequence_host_records(e, &win[buf]);
          mark("window solved, records on the host");
        }
        if (!rc && check)
          rc = lk_get_sequence_guesses(e, win_guess[buf].data());
        if (!rc && !stopped) {
          have_first = first;
          have_n = n;
          have_buf = buf;
        }
        if (stopped)
          break;
      }
      if (next.valid())
        (void)next.get();
      if (!rc && have_n > 0)
        rc = bookkeeping();
      mark("last window's bookkeeping done");
      return rc;
    }
    if (!rc)
      rc = launch(0);
    if (!rc && !verify())
      rc = t->fail(LK_ERROR_BAD_DOMAIN, "lk_sequence_run: device guesses differ from the tracker's (frame 0)");
    Frame fn;
    std::future<int> next;
    auto prefetch = [&](int index) { // frame `index` -> the next-image slot, behind the running solve
      next = std::async(std::launch::async, [&, index] {
        fn = fetch(index);
        return fn.px ? lk_set_image(e, LK_IMG_NXT, fn.px, fn.rows, fn.cols, fn.step) : (int)LK_ERROR_BAD_DOMAIN;
      });
    };
    if (!rc && n_frames > 2)
      prefetch(2);
    for (int k = 0; k < pairs && !rc; ++k) {
      rc = lk_wait_results(e, results.data());
      const std::string und_k = und_name, def_k = def_name;
      if (next.valid()) {
        const int nrc = next.get();
        if (!rc && nrc) // error_multiThread in the reference (manager_class.cpp:1470-1475)
          rc = nrc;
      }
      if (!rc && k + 1 < pairs) { // image roles of the next pair (manager_class.cpp:1386-1407, :166-243)
        if (t->cfg.reference_image == LK_REF_PREVIOUS) {
          rc = lk_rotate_und_from_def(e);
          und_name = def_name;
        }
        if (!rc)
          rc = lk_rotate_def_from_nxt(e);
        def_name = fn.name;
        if (!rc)
          rc = launch(k + 1);
        if (!rc && k + 3 < n_frames)
          prefetch(k + 3);
      }
      if (rc)
        break;
      int first_unsolved = S, stop = 0; // (never a stop: the continue policy is a condition of this path)
      rc = lk_tracker_end_frame(t, k, und_k.c_str(), def_k.c_str(), results.data(), &first_unsolved, &stop);
      if (!rc && pairs_done)
        *pairs_done = k + 1;
      if (!rc && k + 1 < pairs) {
        rc = lk_tracker_begin_frame(t, k + 1, cmds.data(), guesses.data()); // bookkeeping; the solve is already running
        if (!rc && !verify())
          rc = t->fail(LK_ERROR_BAD_DOMAIN, "lk_sequence_run: device guesses differ from the tracker's");
      }
    }
    if (next.valid())
      (void)next.get();
    if (rc) { // leave no solve outstanding behind an error
      std::vector<lk_result> drop((size_t)S);
      (void)lk_wait_results(e, drop.data());
    }
    return rc;
  }
  for (int k = 0; k < pairs && !rc; ++k) {
    // load and upload frame k + 2 while pair k is being solved (manager_class.cpp:1438-1447:
    // std::async + set_next_image; here the upload and pyramid build run on the engine's
    // next-frame stream)
    std::future<int> next;
    Frame fn;
    const bool have_next = k + 2 < n_frames;
    if (have_next)
      next = std::async(std::launch::async, [&, k] {
        fn = fetch(k + 2);
        return fn.px ? lk_set_image(e, LK_IMG_NXT, fn.px, fn.rows, fn.cols, fn.step) : (int)LK_ERROR_BAD_DOMAIN;
      });
    int stop = 0;
    rc = lk_sequence_frame(e, t, k, und_name.c_str(), def_name.c_str(), &stop);
    int nrc = have_next ? next.get() : 0;
    if (!rc && pairs_done)
      *pairs_done = k + 1;
    if (rc || stop)
      break;
    if (have_next) {
      if (nrc) // error_multiThread in the reference (manager_class.cpp:1470-1475)
        return nrc;
      // image roles of the next pair (manager_class.cpp:1386-1407, :166-243)
      if (t->cfg.reference_image == LK_REF_PREVIOUS) {
        rc = lk_rotate_und_from_def(e);
        und_name = def_name;
      }
      if (!rc)
        rc = lk_rotate_def_from_nxt(e);
      def_name = fn.name;
    }
  }
  return rc;
}

// ------------------------------------------------------------------------------------
// ROI -> sample lists without an engine (host code of lk_roi.hpp; what lk_set_sector_* use)
// ------------------------------------------------------------------------------------
static int64_t copy_out(const std::vector<float> &v, float *xy, int64_t cap) {
  const int64_t n = (int64_t)(v.size() / 2);
  if (xy && cap > 0)
    std::memcpy(xy, v.data(), 2 * sizeof(float) * (size_t)(n < cap ? n : cap));
  return n;
}

int lk_roi_rect_grid(float x_begin, float y_begin, float x_end, float y_end, int hs, int vs, int *xdim, int *ydim,
                     int *centers_xy) {
  if (hs < 1 || vs < 1)
    return LK_ERROR_BAD_DOMAIN;
  lkroi::RectGrid g = lkroi::rect_grid(x_begin, y_begin, x_end, y_end, hs, vs);
  if (xdim)
    *xdim = g.xdim;
  if (ydim)
    *ydim = g.ydim;
  if (centers_xy)
    for (int i = 0; i < hs; ++i)
      for (int j = 0; j < vs; ++j) {
        centers_xy[2 * (i * vs + j)] = g.cx[i];
        centers_xy[2 * (i * vs + j) + 1] = g.cy[j];
      }
  return LK_ERROR_NONE;
}

int64_t lk_roi_annular_points(float r, float dr, float a, float da, float cx, float cy, int as, float *xy,
                              int64_t cap) {
  std::vector<float> v;
  if (!lkroi::annular_points(r, dr, a, da, cx, cy, as, v))
    return -1;
  return copy_out(v, xy, cap);
}

int64_t lk_roi_blob_points(const float *contour_xy, int n_vertices, float *xy, int64_t cap) {
  std::vector<float> v;
  if (!contour_xy || !lkroi::BlobPolygon::inside_points(contour_xy, n_vertices, v))
    return -1;
  return copy_out(v, xy, cap);
}

int lk_roi_decimate(const float *xy, int n, int level_delta, float *out) {
  std::vector<float> v;
  const int kept = lkroi::decimate(xy, n, level_delta, v);
  if (out)
    std::memcpy(out, v.data(), 2 * sizeof(float) * (size_t)kept);
  return kept;
}

int lk_load_pgm(const char *path, uint8_t **pixels, int *rows, int *cols) {
  if (!path || !pixels || !rows || !cols)
    return LK_ERROR_BAD_DOMAIN;
  *pixels = nullptr;
  FILE *f = std::fopen(path, "rb");
  if (!f)
    return LK_ERROR_BAD_DOMAIN;
  auto token = [&](int &v) { // whitespace / comment separated decimal
    int c = std::fgetc(f);
    for (;;) {
      while (c == ' ' || c == '\t' || c == '\n' || c == '\r')
        c = std::fgetc(f);
      if (c != '#')
        break;
      while (c != '\n' && c != EOF)
        c = std::fgetc(f);
    }
    if (c < '0' || c > '9')
      return false;
    long acc = 0;
    while (c >= '0' && c <= '9') {
      acc = acc * 10 + (c - '0');
      if (acc > 1 << 30)
        return false;
      c = std::fgetc(f);
    }
    v = (int)acc; // the single whitespace after the token has been consumed
    return true;
  };
  int w = 0, h = 0, maxval = 0;
  bool ok = std::fgetc(f) == 'P' && std::fgetc(f) == '5' && token(w) && token(h) && token(maxval) && w > 0 &&
            h > 0 && maxval > 0 && maxval <= 255;
  uint8_t *buf = nullptr;
  if (ok) {
    buf = (uint8_t *)std::malloc((size_t)w * (size_t)h);
    ok = buf && std::fread(buf, 1, (size_t)w * (size_t)h, f) == (size_t)w * (size_t)h;
  }
  std::fclose(f);
  if (!ok) {
    std::free(buf);
    return LK_ERROR_BAD_DOMAIN;
  }
  *pixels = buf;
  *rows = h;
  *cols = w;
  return LK_ERROR_NONE;
}

void lk_free_image(uint8_t *pixels) { std::free(pixels); }

} // extern "C"
