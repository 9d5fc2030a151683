// lk_engine_mock.cpp - TEST INFRASTRUCTURE: a CPU stand-in for the subset of include/lk_engine.h
// that include/lk_cuda_class_adapter.hpp calls, so that HipCudaClass can be EXECUTED in the build
// container (no GPU) in the reference manager's exact call order.  It keeps the engine's
// documented contracts (sectors must be committed before a solve; a commit keeps the state of
// sectors that were not registered anew; lk_update_sector moves a sector by its own last record;
// lk_correlate copies the result into the caller's guess) and journals every call.  The "solve"
// is a closed-form fake: p = guess + 0.25 * (sector centre - image centre) / 100 on (u, v).
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "lk_engine.h"
#include "lk_group.h"

namespace {
struct Sector {
  int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
  std::vector<float> xy;
  bool set = false, fresh = true;
  lk_result last{};
  bool solved = false;
};
}
struct lk_engine {
  lk_config cfg{};
  std::vector<Sector> hs;
  bool committed = false;
  int rows[3] = {0, 0, 0}, cols[3] = {0, 0, 0};
  unsigned sum[3] = {0, 0, 0};
  bool valid[3] = {false, false, false};
  std::vector<float> guess, last_p, prev_p; // [S][6], engine-held (lk_adjust_initial_guess)
  std::vector<lk_result> pending, own_records;
  bool outstanding = false;
  // frame-pipelined windows: the ring holds the image checksums; a window is the loop of one-pair solves
  std::vector<unsigned> ring_sum;
  std::vector<char> ring_valid;
  std::vector<lk_result> win_records, win_records_kept[2];
  int win_kept = 0;
  std::vector<float> win_guesses;
  int win_frames = 0;
  bool win_outstanding = false;
  std::string err;
};
static std::string g_journal;
static int g_commits = 0;
static void J(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#include <cstdarg>
static void J(const char *fmt, ...) {
  char b[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(b, sizeof b, fmt, ap);
  va_end(ap);
  static std::mutex mu; // lk_set_image(LK_IMG_NXT) may come from the caller's prefetch thread (lk_engine.h)
  std::lock_guard<std::mutex> lock(mu);
  g_journal += b;
  g_journal += '\n';
}

extern "C" {
const char *lk_mock_journal(void) { return g_journal.c_str(); }
int lk_mock_commits(void) { return g_commits; }

int lk_device_count(void) { return 1; }
int lk_create(const lk_config *cfg, lk_engine **out) {
  if (!cfg || !out)
    return LK_ERROR_BAD_DOMAIN;
  lk_engine *e = new lk_engine();
  e->cfg = *cfg;
  *out = e;
  J("create interp=%d model=%d prec=%g iters=%d py=%d/%d/%d", cfg->interpolation, cfg->fitting_model, cfg->precision,
    cfg->max_iters, cfg->py_start, cfg->py_step, cfg->py_stop);
  return 0;
}
void lk_destroy(lk_engine *e) {
  J("destroy");
  delete e;
}
const char *lk_last_error_string(const lk_engine *e) { return e ? e->err.c_str() : "null"; }
int lk_set_image(lk_engine *e, int slot, const uint8_t *px, int rows, int cols, int step) {
  if (!e || slot < 0 || slot > 2 || !px)
    return LK_ERROR_BAD_DOMAIN;
  unsigned s = 0;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c)
      s = s * 31u + px[(size_t)r * step + c];
  e->rows[slot] = rows, e->cols[slot] = cols, e->sum[slot] = s, e->valid[slot] = true;
  J("set_image slot=%d %dx%d sum=%u", slot, rows, cols, s);
  return 0;
}
int lk_rotate_und_from_def(lk_engine *e) {
  e->sum[0] = e->sum[1], e->valid[0] = e->valid[1], e->valid[1] = false;
  J("und_from_def");
  return 0;
}
int lk_rotate_def_from_nxt(lk_engine *e) {
  if (!e->valid[2])
    return LK_ERROR_BAD_DOMAIN;
  e->sum[1] = e->sum[2], e->valid[1] = true, e->valid[2] = false;
  J("def_from_nxt");
  return 0;
}
static Sector *slot(lk_engine *e, int s) {
  if (s < 0)
    return nullptr;
  if ((size_t)s >= e->hs.size())
    e->hs.resize((size_t)s + 1);
  e->committed = false;
  e->hs[(size_t)s].fresh = true;
  return &e->hs[(size_t)s];
}
int lk_set_sector_rect(lk_engine *e, int s, int x0, int y0, int x1, int y1) {
  Sector *q = slot(e, s);
  if (!q || x1 < x0 || y1 < y0)
    return LK_ERROR_BAD_DOMAIN;
  q->x0 = x0, q->y0 = y0, q->x1 = x1, q->y1 = y1, q->set = true;
  q->xy.clear();
  J("set_rect %d [%d,%d]-[%d,%d]", s, x0, y0, x1, y1);
  return 0;
}
int lk_set_sector_annular(lk_engine *e, int s, float r, float dr, float a, float da, float cx, float cy, int as) {
  Sector *q = slot(e, s);
  if (!q)
    return LK_ERROR_BAD_DOMAIN;
  q->xy = {cx + r, cy, cx + r + dr, cy};
  q->set = true;
  J("set_annular %d r=%g dr=%g a=%g da=%g c=(%g,%g) as=%d", s, r, dr, a, da, cx, cy, as);
  return 0;
}
int lk_set_sector_blob(lk_engine *e, int s, const float *c, int n) {
  Sector *q = slot(e, s);
  if (!q || n < 3)
    return LK_ERROR_BAD_DOMAIN;
  q->xy.assign(c, c + 2 * n);
  q->set = true;
  J("set_blob %d n=%d", s, n);
  return 0;
}
int lk_commit_sectors(lk_engine *e) {
  for (auto &q : e->hs)
    if (!q.set)
      return LK_ERROR_BAD_DOMAIN;
  for (auto &q : e->hs) {
    if (q.fresh) {
      q.last = lk_result{};
      q.solved = false;
    }
    q.fresh = false;
  }
  e->committed = true;
  ++g_commits;
  J("commit S=%zu", e->hs.size());
  return 0;
}
int lk_sector_count(const lk_engine *e) { return e ? (int)e->hs.size() : 0; }
int lk_update_sector(lk_engine *e, int s, int mode) {
  if (!e->committed || s < 0 || (size_t)s >= e->hs.size())
    return LK_ERROR_BAD_DOMAIN;
  Sector &q = e->hs[(size_t)s];
  J("update %d mode=%d solved=%d u=%g v=%g", s, mode, (int)q.solved, q.last.resultingParameters[0], q.last.resultingParameters[1]);
  if (mode == 1) { // Lagrangian: move by the rounded (u, v) of the sector's own last record
    const int du = (int)(q.last.resultingParameters[0] + (q.last.resultingParameters[0] < 0 ? -0.5f : 0.5f));
    const int dv = (int)(q.last.resultingParameters[1] + (q.last.resultingParameters[1] < 0 ? -0.5f : 0.5f));
    q.x0 += du, q.x1 += du, q.y0 += dv, q.y1 += dv;
  }
  return 0;
}
static lk_result fake_solve(lk_engine *e, const Sector &q, const float *g) {
  lk_result r{};
  const float cx = 0.5f * (float)(q.x0 + q.x1), cy = 0.5f * (float)(q.y0 + q.y1);
  for (int i = 0; i < 6; ++i)
    r.resultingParameters[i] = g[i];
  r.resultingParameters[0] += 0.0025f * (cx - 0.5f * (float)e->cols[0]);
  r.resultingParameters[1] += 0.0025f * (cy - 0.5f * (float)e->rows[0]);
  r.chi = 1.f + (float)(e->sum[1] % 97u);
  r.numberOfPoints = (q.x1 - q.x0 + 1) * (q.y1 - q.y0 + 1);
  r.iterations = 2;
  r.errorCode = 0;
  r.undCenterX = cx, r.undCenterY = cy;
  return r;
}
int lk_correlate(lk_engine *e, int s, float *guess, lk_result *out) {
  if (!e->committed || !e->valid[0] || !e->valid[1] || s < 0 || (size_t)s >= e->hs.size())
    return LK_ERROR_BAD_DOMAIN;
  float g[6] = {0, 0, 0, 0, 0, 0};
  const int P = e->cfg.fitting_model == LK_FM_U ? 1 : e->cfg.fitting_model == LK_FM_UV ? 2 : e->cfg.fitting_model == LK_FM_UVQ ? 3 : 6;
  for (int i = 0; i < P; ++i)
    g[i] = guess[i];
  Sector &q = e->hs[(size_t)s];
  q.last = fake_solve(e, q, g);
  q.solved = true;
  *out = q.last;
  for (int i = 0; i < P; ++i)
    guess[i] = out->resultingParameters[i];
  J("correlate %d guess=(%g,%g)", s, g[0], g[1]);
  return 0;
}
int lk_correlate_all(lk_engine *e, const float *guesses, lk_result *out) {
  if (!e->committed || !e->valid[0] || !e->valid[1])
    return LK_ERROR_BAD_DOMAIN;
  for (size_t s = 0; s < e->hs.size(); ++s) {
    float g[6] = {0, 0, 0, 0, 0, 0};
    if (guesses)
      std::memcpy(g, guesses + 6 * s, sizeof g);
    e->hs[s].last = out[s] = fake_solve(e, e->hs[s], g);
    e->hs[s].solved = true;
  }
  e->last_p.resize(6 * e->hs.size(), 0.f);
  for (size_t s = 0; s < e->hs.size(); ++s)
    std::memcpy(&e->last_p[6 * s], out[s].resultingParameters, 6 * sizeof(float));
  J("correlate_all S=%zu", e->hs.size());
  return 0;
}
// Device-buffer form, as the real engine has it ("device" memory is host memory here): records go to the
// caller's buffer when one is given - and then NOT into the engine's own record buffer, which is what
// lk_update_sector reads - or stay in the engine's own buffer (d_results = NULL, lk_get_results_device).
int lk_correlate_all_device(lk_engine *e, const void *d_guesses, void *d_results) {
  if (!e->committed || !e->valid[0] || !e->valid[1])
    return LK_ERROR_BAD_DOMAIN;
  const float *guesses = (const float *)d_guesses;
  lk_result *ext = (lk_result *)d_results;
  e->last_p.resize(6 * e->hs.size(), 0.f);
  for (size_t s = 0; s < e->hs.size(); ++s) {
    float g[6] = {0, 0, 0, 0, 0, 0};
    if (guesses)
      std::memcpy(g, guesses + 6 * s, sizeof g);
    else if (e->guess.size() == 6 * e->hs.size())
      std::memcpy(g, &e->guess[6 * s], sizeof g);
    const lk_result r = fake_solve(e, e->hs[s], g);
    if (ext) {
      ext[s] = r;
    } else {
      e->hs[s].last = r;
      e->hs[s].solved = true;
    }
    std::memcpy(&e->last_p[6 * s], r.resultingParameters, 6 * sizeof(float));
  }
  J("correlate_all_device S=%zu own=%d", e->hs.size(), ext ? 0 : 1);
  return 0;
}
int lk_get_results_device(lk_engine *e, const void **d_records) {
  if (!e->committed || !d_records)
    return LK_ERROR_BAD_DOMAIN;
  e->own_records.resize(e->hs.size());
  for (size_t s = 0; s < e->hs.size(); ++s)
    e->own_records[s] = e->hs[s].last;
  *d_records = e->own_records.data();
  return 0;
}
// ---- what lk_sequence_frame / lk_sequence_run (lk_tracker.cpp) call besides the above ----------
int lk_clear_sectors(lk_engine *e) {
  e->hs.clear();
  e->committed = false;
  J("clear");
  return 0;
}
int lk_set_sectors_annular(lk_engine *e, int first, int count, const float *q, int as) {
  for (int k = 0; k < count; ++k)
    if (int rc = lk_set_sector_annular(e, first + k, q[6 * k], q[6 * k + 1], q[6 * k + 2], q[6 * k + 3], q[6 * k + 4], q[6 * k + 5], as))
      return rc;
  return 0;
}
int lk_translate_sectors(lk_engine *e, const float *off, const float *centers) {
  if (!e->committed || !off)
    return LK_ERROR_BAD_DOMAIN;
  for (size_t s = 0; s < e->hs.size(); ++s) {
    Sector &q = e->hs[s];
    const int dx = (int)(off[2 * s] + 0.5f), dy = (int)(off[2 * s + 1] + 0.5f);
    q.x0 += dx, q.x1 += dx, q.y0 += dy, q.y1 += dy;
    for (size_t i = 0; i + 1 < q.xy.size(); i += 2)
      q.xy[i] += (float)dx, q.xy[i + 1] += (float)dy;
  }
  J("translate S=%zu centers=%d", e->hs.size(), centers ? 1 : 0);
  return 0;
}
int lk_rewarp_sectors(lk_engine *e, const float *centers) {
  if (!e->committed)
    return LK_ERROR_BAD_DOMAIN;
  J("rewarp S=%zu centers=%d", e->hs.size(), centers ? 1 : 0);
  return 0;
}
int lk_restore_sectors(lk_engine *e, int first) {
  J("restore from %d", first);
  return e->committed ? 0 : LK_ERROR_BAD_DOMAIN;
}
int lk_adjust_initial_guess(lk_engine *e, int frame, int cv, const float *gg, float gcx, float gcy) {
  if (!e->committed)
    return LK_ERROR_BAD_DOMAIN;
  const size_t S = e->hs.size();
  e->guess.assign(6 * S, 0.f);
  e->last_p.resize(6 * S, 0.f);
  e->prev_p.resize(6 * S, 0.f);
  for (size_t s = 0; s < S; ++s)
    for (int i = 0; i < 6; ++i) {
      float g;
      if (frame == 0) {
        g = gg ? gg[i] : 0.f;
        e->prev_p[6 * s + i] = g;
      } else {
        const float r = e->last_p[6 * s + i], q = e->prev_p[6 * s + i];
        g = cv ? r + (r - q) : r;
        e->prev_p[6 * s + i] = r;
      }
      e->guess[6 * s + i] = g;
    }
  (void)gcx, (void)gcy;
  J("adjust_guess frame=%d cv=%d", frame, cv);
  return 0;
}
int lk_get_guesses(lk_engine *e, float *g) {
  std::memcpy(g, e->guess.data(), e->guess.size() * sizeof(float));
  return 0;
}
int lk_correlate_all_async(lk_engine *e) {
  if (!e->committed || e->outstanding || !e->valid[0] || !e->valid[1])
    return LK_ERROR_BAD_DOMAIN;
  const size_t S = e->hs.size();
  if (e->guess.size() != 6 * S)
    e->guess.assign(6 * S, 0.f);
  e->pending.resize(S);
  e->last_p.resize(6 * S, 0.f);
  for (size_t s = 0; s < S; ++s) {
    e->pending[s] = e->hs[s].last = fake_solve(e, e->hs[s], &e->guess[6 * s]);
    e->hs[s].solved = true;
    std::memcpy(&e->last_p[6 * s], e->pending[s].resultingParameters, 6 * sizeof(float));
  }
  e->outstanding = true;
  J("correlate_all_async S=%zu", S);
  return 0;
}
int lk_sequence_reserve(lk_engine *e, int n_slots) {
  if (!e || n_slots < 1)
    return LK_ERROR_BAD_DOMAIN;
  if ((int)e->ring_sum.size() < n_slots) {
    e->ring_sum.resize((size_t)n_slots, 0u);
    e->ring_valid.resize((size_t)n_slots, 0);
  }
  J("sequence_reserve %d", n_slots);
  return 0;
}
int lk_sequence_set_frame(lk_engine *e, int slot, const uint8_t *px, int rows, int cols, int step) {
  if (!e || slot < 0 || slot >= (int)e->ring_sum.size() || !px)
    return LK_ERROR_BAD_DOMAIN;
  unsigned s = 0;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c)
      s = s * 31u + px[(size_t)r * step + c];
  e->ring_sum[(size_t)slot] = s;
  e->ring_valid[(size_t)slot] = 1;
  return 0; // (no journal line: a helper thread calls this while the caller's thread writes the journal)
}
int lk_correlate_sequence_async(lk_engine *e, int und_slot, int first_slot, int n_frames, int reference_previous, int cv, int flags) {
  const int R = (int)e->ring_sum.size();
  if (!e->committed || e->outstanding || e->win_outstanding || n_frames < 1 || n_frames > R || first_slot < 0 || first_slot >= R)
    return LK_ERROR_BAD_DOMAIN;
  const size_t S = e->hs.size();
  if (e->guess.size() != 6 * S)
    e->guess.assign(6 * S, 0.f);
  e->last_p.resize(6 * S, 0.f);
  e->prev_p.resize(6 * S, 0.f);
  e->win_records.assign((size_t)n_frames * S, lk_result{});
  e->win_guesses.assign((size_t)n_frames * 6 * S, 0.f);
  const unsigned keep_und = e->sum[0], keep_def = e->sum[1];
  for (int f = 0; f < n_frames; ++f) {
    const int slot = (first_slot + f) % R;
    if (!e->ring_valid[(size_t)slot])
      return LK_ERROR_BAD_DOMAIN;
    if (f == 0 && und_slot >= 0)
      e->sum[0] = e->ring_sum[(size_t)und_slot];
    else if (f > 0 && reference_previous)
      e->sum[0] = e->ring_sum[(size_t)((first_slot + f - 1) % R)];
    e->sum[1] = e->ring_sum[(size_t)slot];
    for (size_t s = 0; s < S; ++s) {
      if (f > 0)
        for (int i = 0; i < 6; ++i) { // adjust_initial_guess's rule, per sector
          const float r = e->last_p[6 * s + i], q = e->prev_p[6 * s + i];
          e->guess[6 * s + i] = cv ? r + (r - q) : r;
          e->prev_p[6 * s + i] = r;
        }
      std::memcpy(&e->win_guesses[((size_t)f * S + s) * 6], &e->guess[6 * s], 6 * sizeof(float));
      const lk_result r = e->hs[s].last = fake_solve(e, e->hs[s], &e->guess[6 * s]);
      e->hs[s].solved = true;
      e->win_records[(size_t)f * S + s] = r;
      std::memcpy(&e->last_p[6 * s], r.resultingParameters, 6 * sizeof(float));
    }
  }
  e->sum[0] = keep_und, e->sum[1] = keep_def;
  e->win_frames = n_frames;
  e->win_outstanding = true;
  J("correlate_sequence_async und_slot=%d first_slot=%d frames=%d ref_prev=%d cv=%d flags=%d", und_slot, first_slot, n_frames,
    reference_previous, cv, flags);
  return 0;
}
int lk_wait_sequence(lk_engine *e, lk_result *out) {
  if (!e->win_outstanding)
    return LK_ERROR_BAD_DOMAIN;
  if (out)
    std::memcpy(out, e->win_records.data(), e->win_records.size() * sizeof(lk_result));
  e->win_outstanding = false;
  return 0;
}
int lk_sequence_prepare_host_records(lk_engine *, int n_frames) { return n_frames < 1 ? LK_ERROR_BAD_DOMAIN : 0; }
int lk_sequence_host_records(lk_engine *e, const lk_result **records) {
  if (e->win_outstanding || e->win_records.empty())
    return LK_ERROR_BAD_DOMAIN;
  e->win_records_kept[e->win_kept ^= 1] = e->win_records; // (two alternating buffers, like the engine's pinned ones)
  *records = e->win_records_kept[e->win_kept].data();
  return 0;
}
int lk_get_sequence_guesses(lk_engine *e, float *g) {
  if (e->win_outstanding || e->win_guesses.empty())
    return LK_ERROR_BAD_DOMAIN;
  std::memcpy(g, e->win_guesses.data(), e->win_guesses.size() * sizeof(float));
  return 0;
}
int lk_wait_results(lk_engine *e, lk_result *out) {
  if (!e->outstanding || !out)
    return LK_ERROR_BAD_DOMAIN;
  std::memcpy(out, e->pending.data(), e->pending.size() * sizeof(lk_result));
  e->outstanding = false;
  return 0;
}

int lk_get_und_xy(lk_engine *e, int s, float *xy, int cap, int *count) {
  if (s < 0 || (size_t)s >= e->hs.size() || !e->hs[(size_t)s].set)
    return LK_ERROR_BAD_DOMAIN;
  const Sector &q = e->hs[(size_t)s];
  std::vector<float> v = q.xy;
  if (v.empty())
    for (int x = q.x0; x <= q.x1; ++x)
      for (int y = q.y0; y <= q.y1; ++y)
        v.push_back((float)x), v.push_back((float)y);
  *count = (int)(v.size() / 2);
  for (int i = 0; i < *count && i < cap; ++i)
    xy[2 * i] = v[2 * (size_t)i], xy[2 * i + 1] = v[2 * (size_t)i + 1];
  return 0;
}
int lk_get_level_xy(lk_engine *, int, int, int, float *, int, int *) { return LK_ERROR_BAD_DOMAIN; } // (no device lists here)
int lk_get_def_xy(lk_engine *e, int s, const float *p, float *xy, int cap, int *count) {
  int rc = lk_get_und_xy(e, s, xy, cap, count);
  if (!rc && p)
    for (int i = 0; i < *count && i < cap; ++i)
      xy[2 * i] += p[0], xy[2 * i + 1] += p[1];
  return rc;
}

// ---- include/lk_group.h over n mock engines: same partition rule, sectors re-dealt at commit --------
struct lk_group_mock_spec {
  int kind = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0, as = 1;
  float r = 0, dr = 0, a = 0, da = 0, cx = 0, cy = 0;
};
}
struct lk_group {
  std::vector<lk_engine *> e;
  std::vector<lk_group_mock_spec> secs;
  bool committed = false;
};
extern "C" {
int lk_group_shard_range(int S, int rank, int n, int *first, int *count) {
  *first = (int)((long long)S * rank / n);
  *count = (int)((long long)S * (rank + 1) / n) - *first;
  return 0;
}
int lk_group_create(const lk_config *cfg, int n, const int *, lk_group **out) {
  lk_group *g = new lk_group();
  for (int r = 0; r < n; ++r) {
    lk_engine *e = nullptr;
    lk_create(cfg, &e);
    g->e.push_back(e);
  }
  *out = g;
  J("group_create n=%d", n);
  return 0;
}
void lk_group_destroy(lk_group *g) {
  for (lk_engine *e : g->e)
    lk_destroy(e);
  delete g;
}
int lk_group_size(const lk_group *g) { return (int)g->e.size(); }
int lk_group_engine(lk_group *g, int rank, lk_engine **e) {
  *e = g->e[(size_t)rank];
  return 0;
}
int lk_group_shard(const lk_group *g, int rank, int *first, int *count) {
  return lk_group_shard_range((int)g->secs.size(), rank, (int)g->e.size(), first, count);
}
int lk_group_set_image(lk_group *g, int slot, const uint8_t *px, int rows, int cols, int step) {
  for (lk_engine *e : g->e)
    if (int rc = lk_set_image(e, slot, px, rows, cols, step))
      return rc;
  return 0;
}
int lk_group_rotate_und_from_def(lk_group *g) {
  for (lk_engine *e : g->e)
    lk_rotate_und_from_def(e);
  return 0;
}
int lk_group_rotate_def_from_nxt(lk_group *g) {
  for (lk_engine *e : g->e)
    if (int rc = lk_rotate_def_from_nxt(e))
      return rc;
  return 0;
}
int lk_group_set_sector_rect(lk_group *g, int s, int x0, int y0, int x1, int y1) {
  if ((size_t)s >= g->secs.size())
    g->secs.resize((size_t)s + 1);
  g->secs[(size_t)s] = lk_group_mock_spec{1, x0, y0, x1, y1, 1, 0, 0, 0, 0, 0, 0};
  g->committed = false;
  return 0;
}
int lk_group_set_sector_annular(lk_group *g, int s, float r, float dr, float a, float da, float cx, float cy, int as) {
  if ((size_t)s >= g->secs.size())
    g->secs.resize((size_t)s + 1);
  g->secs[(size_t)s] = lk_group_mock_spec{2, 0, 0, 0, 0, as, r, dr, a, da, cx, cy};
  g->committed = false;
  return 0;
}
int lk_group_commit_sectors(lk_group *g) {
  const int n = (int)g->e.size(), S = (int)g->secs.size();
  for (int r = 0; r < n; ++r) {
    int first, count;
    lk_group_shard_range(S, r, n, &first, &count);
    lk_clear_sectors(g->e[(size_t)r]);
    for (int k = 0; k < count; ++k) {
      const lk_group_mock_spec &q = g->secs[(size_t)(first + k)];
      if (q.kind == 1)
        lk_set_sector_rect(g->e[(size_t)r], k, q.x0, q.y0, q.x1, q.y1);
      else
        lk_set_sector_annular(g->e[(size_t)r], k, q.r, q.dr, q.a, q.da, q.cx, q.cy, q.as);
    }
    if (int rc = lk_commit_sectors(g->e[(size_t)r]))
      return rc;
  }
  g->committed = true;
  J("group_commit S=%d over %d", S, n);
  return 0;
}
int lk_group_sector_count(const lk_group *g) { return (int)g->secs.size(); }
int lk_group_correlate_all(lk_group *g, const float *guesses, lk_result *out) {
  if (!g->committed)
    return LK_ERROR_BAD_DOMAIN;
  const int n = (int)g->e.size(), S = (int)g->secs.size();
  for (int r = 0; r < n; ++r) {
    int first, count;
    lk_group_shard_range(S, r, n, &first, &count);
    // the real group's data flow (csrc/lk_group.cpp): the member solves into ITS OWN record buffer - the one
    // lk_update_sector reads - and the block for the gather is copied from there
    if (int rc = lk_correlate_all_device(g->e[(size_t)r], guesses ? guesses + 6 * (size_t)first : nullptr, nullptr))
      return rc;
    const void *own = nullptr;
    if (int rc = lk_get_results_device(g->e[(size_t)r], &own))
      return rc;
    if (out)
      std::memcpy(out + first, own, (size_t)count * sizeof(lk_result));
  }
  J("group_correlate_all S=%d", S);
  return 0;
}
}
