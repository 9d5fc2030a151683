// lk_group.cpp - single-process multi-GPU engine behind include/lk_group.h: one lk_engine, one host
// thread and one HIP stream per device; sectors in contiguous blocks of the global index; frames
// by ncclBroadcast, records by ncclAllGather (RCCL over xGMI).  SURVEY.md section 8(e).
//
// Threads.  Every entry point hands ONE job to all member threads and waits for them (the jobs
// are a few asynchronous enqueues each; what takes time runs on the devices).  Inside a job every
// member issues the same sequence of RCCL calls on its own communicator and stream - the
// one-thread-per-device model of ncclCommInitAll - so the collectives rendezvous without
// ncclGroupStart/End.  No host thread ever waits for another one inside a job, except in the
// one-GPU rehearsal mode (the same device listed several times), where device-to-device copies
// and a host barrier stand in for the two collectives.
//
// Streams.  Every member has its solve stream (the engine's) and a COMMUNICATION stream: frame uploads, broadcasts,
// all-gathers, the stale-count pass over gathered records and the copies of records to the host run there, ordered
// against the solves by events - never by the host.  The broadcast of frame k + 1 therefore travels while pair k is
// being solved (the reference's prefetch, manager_class.cpp:1438-1447, across GPUs), and the records of pair k leave
// while pair k + 1 is being solved.  Buffers are protected the same way: a frame buffer is overwritten only after the
// engine's pyramid kernel that read it (`consumed`), the padded record block only after the all-gather that read it.
// An engine whose solves launch TEAMS (workgroups that wait for each other) keeps its collectives on the solve stream:
// an RCCL kernel beside a team launch could hold a slot one of its workgroups needs.
//
// Failures.  A member that gave up before a collective would leave the others' collective - and rank 0's
// hipStreamSynchronize behind it - waiting for ever.  So every job is cut into phases: what can fail on one
// member alone (allocations, a deferred re-commit inside the engine, uploads, the solve launch) comes first, then
// the members AGREE on the outcome (lk_group::agree: a host rendezvous that hands everybody the first error of
// the phase), and only if nobody failed does anybody enqueue the collective.  A collective that fails at
// enqueue on one member is agreed on the same way afterwards: every member then aborts its communicator
// (ncclCommAbort ends the kernels of the ranks that did enqueue) and the group is broken for good - every
// later call returns LK_ERROR_DEVICE.  The rehearsal mode's barriers are reached by every member whatever
// happened before them; the error code travels past them.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lk_group.h"
#include "lk_internal.hpp"

namespace {

struct SectorSpec { // what lk_group_set_sector_* recorded, replayed on the owner at commit
  int kind = 0;     // 0 unset, 1 rect, 2 annular, 3 points
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  float r = 0, dr = 0, a = 0, da = 0, cx = 0, cy = 0;
  int as = 1, use_center = 0;
  std::vector<float> xy;
};

struct FrameBuf { // level-0 pixels of one image slot / ring slot on one member, dense rows
  uint8_t *p = nullptr;
  size_t cap = 0;
  hipEvent_t arrived = nullptr;  // comm stream: the pixels are here
  hipEvent_t consumed = nullptr; // engine's fill stream: the last kernel that reads them is done
  hipEvent_t up = nullptr;       // rehearsal transport, rank 0: the upload is done
  hipEvent_t copied = nullptr;   // rehearsal transport, other ranks: rank 0's pixels have been copied
  bool consumed_valid = false, copied_valid = false;
};

struct Member {
  int rank = 0, device = 0;
  lk_engine *e = nullptr;
  hipStream_t st = nullptr;
  hipStream_t cst = nullptr; // communication stream
  ncclComm_t comm = nullptr;
  FrameBuf frame[3];
  std::vector<FrameBuf> ring;   // frame-pipelined windows: the resident deformed frames - slices of ONE allocation, so that the
  uint8_t *d_ring = nullptr;    //   frames of a whole window travel in one broadcast
  size_t ring_slot_bytes = 0;
  hipEvent_t ev_solved = nullptr, ev_gathered = nullptr, ev_block = nullptr; // solve done (st) / records gathered (cst) / rehearsal: my block is in everybody's d_all
  bool gathered_valid = false;
  hipEvent_t probe[4] = {nullptr, nullptr, nullptr, nullptr}; // timing: last frame transfer begin / end (cst), last solve begin / end (st)
  bool probe_valid[2] = {false, false};
  lk_result *d_seq_rec = nullptr, *d_seq_all = nullptr; // windows: [frames][cap] of this member, [n][frames][cap] of everybody
  size_t seq_frames_cap = 0;
  int seq_frames = 0; // frames of the outstanding window (0: none)
  int exchanged_frames = 0; // frames of the window whose records the last exchange gathered into d_seq_all (0: none)
  float *d_guess = nullptr;   // [cap][6]
  lk_result *d_rec = nullptr; // [cap] this member's block (padded)
  lk_result *d_all = nullptr; // [n][cap] everybody's blocks
  int *d_stale = nullptr;     // [2] reference-order mode: the group-level stale iteration count, alternating
  int stale_par = 0;
  size_t cap = 0;
  int first = 0, count = 0;
  bool device_ok = true; // hipSetDevice of the current job worked
  std::thread th;
};

} // namespace

struct lk_group {
  lk_config cfg{};
  std::vector<Member> m;
  bool loopback = false; // duplicate devices: copies + host barrier instead of RCCL (rehearsal on one GPU)
  bool copy_frames = false; // frames by device-to-device copies from rank 0's buffer (copy engines over xGMI, no CU needed)
                            // instead of ncclBroadcast: always in the rehearsal transport, LK_GROUP_FRAMES=copy otherwise
  // sector registry (global index)
  std::vector<SectorSpec> secs;
  bool grid = false;
  float gx0 = 0, gy0 = 0, gx1 = 0, gy1 = 0;
  int ghs = 0, gvs = 0;
  int S = 0, cap = 0;
  bool committed = false;
  // job hand-off
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  unsigned long long gen = 0;
  int pending = 0;
  bool quit = false;
  std::function<int(Member &)> job;
  std::vector<int> rc;
  std::vector<std::string> msg;
  // rehearsal-mode barrier between the member threads
  std::mutex bmu;
  std::condition_variable bcv;
  int barrier_count = 0;
  unsigned long long barrier_gen = 0;
  int agree_rc = 0, agree_result = 0; // first error of the phase being agreed on / what the last agreement returned
  std::atomic<bool> broken{false};    // a collective failed on some member: the communicators are aborted
  bool job_collective = false;        // the current job has rendezvous points: every member must run it to the end
  std::string err;

  int fail(int code, const std::string &what) {
    err = what;
    return code;
  }
  void barrier() { (void)agree(0); }
  // Rendezvous of the member threads that also settles whether the phase before it succeeded everywhere:
  // every member passes its own code and all of them receive the same answer - the first non-zero code by
  // rank order of arrival, or 0.  Nobody proceeds into a collective unless the answer is 0.
  int agree(int rc) {
    std::unique_lock<std::mutex> lock(bmu);
    const unsigned long long g0 = barrier_gen;
    if (rc && !agree_rc)
      agree_rc = rc;
    if (++barrier_count == (int)m.size()) {
      barrier_count = 0;
      agree_result = agree_rc;
      agree_rc = 0;
      ++barrier_gen;
      bcv.notify_all();
      return agree_result;
    }
    bcv.wait(lock, [&] { return barrier_gen != g0; });
    return agree_result;
  }
  // run `f` on every member thread; returns the first error
  // collective = true: the job contains rendezvous points (lk_group::agree) that every member must reach
  int run(std::function<int(Member &)> f, bool collective = false) {
    if (broken)
      return fail(LK_ERROR_DEVICE, "the group is broken: a collective failed earlier (communicators aborted)");
    std::unique_lock<std::mutex> lock(mu);
    job = std::move(f);
    job_collective = collective;
    std::fill(rc.begin(), rc.end(), 0);
    for (std::string &t : msg)
      t.clear();
    pending = (int)m.size();
    ++gen;
    cv_job.notify_all();
    cv_done.wait(lock, [&] { return pending == 0; });
    // (a member that only relays the outcome of lk_group::agree has no message of its own: name the one that failed)
    for (int pass = 0; pass < 2; ++pass)
      for (size_t i = 0; i < m.size(); ++i)
        if (rc[i] && (pass == 1 || !msg[i].empty())) {
          err = "rank " + std::to_string(i) + ": " + msg[i];
          return rc[i];
        }
    return LK_ERROR_NONE;
  }
};

namespace {

void shard(int S, int rank, int n, int &first, int &count) { // [r*S/G, (r+1)*S/G)
  first = (int)((long long)S * rank / n);
  count = (int)((long long)S * (rank + 1) / n) - first;
}

void worker(lk_group *g, int rank) {
  Member &me = g->m[(size_t)rank];
  unsigned long long seen = 0;
  for (;;) {
    std::function<int(Member &)> f;
    bool collective = false;
    {
      std::unique_lock<std::mutex> lock(g->mu);
      g->cv_job.wait(lock, [&] { return g->quit || g->gen != seen; });
      if (g->quit)
        return;
      seen = g->gen;
      f = g->job;
      collective = g->job_collective;
    }
    // (a member that cannot even select its device still runs a COLLECTIVE job, with the failure latched, so
    // that it reaches every rendezvous of the job: see lk_group::agree)
    me.device_ok = hipSetDevice(me.device) == hipSuccess;
    int r = (me.device_ok || collective) ? f(me) : LK_ERROR_DEVICE;
    if (!me.device_ok && !r)
      r = LK_ERROR_DEVICE;
    {
      std::unique_lock<std::mutex> lock(g->mu);
      g->rc[(size_t)rank] = r;
      if (--g->pending == 0)
        g->cv_done.notify_all();
    }
  }
}

// error plumbing inside jobs
#define GHIP(call)                                                                                   \
  do {                                                                                               \
    hipError_t _e = (call);                                                                          \
    if (_e != hipSuccess) {                                                                          \
      g->msg[(size_t)me.rank] = std::string(#call) + ": " + hipGetErrorString(_e);                   \
      return LK_ERROR_DEVICE;                                                                        \
    }                                                                                                \
  } while (0)
#define GNCCL(call)                                                                                  \
  do {                                                                                               \
    ncclResult_t _e = (call);                                                                        \
    if (_e != ncclSuccess) {                                                                         \
      g->msg[(size_t)me.rank] = std::string(#call) + ": " + ncclGetErrorString(_e);                  \
      return LK_ERROR_DEVICE;                                                                        \
    }                                                                                                \
  } while (0)
#define GLK(call)                                                                                    \
  do {                                                                                               \
    int _e = (call);                                                                                 \
    if (_e != LK_ERROR_NONE) {                                                                       \
      g->msg[(size_t)me.rank] = std::string(#call) + ": " + lk_last_error_string(me.e);              \
      return _e;                                                                                     \
    }                                                                                                \
  } while (0)

hipStream_t comm_stream(const Member &me) { return lk_internal_team_launches(me.e) ? me.st : me.cst; }

int ensure_events(lk_group *g, Member &me, FrameBuf &fb) {
  for (hipEvent_t *ev : {&fb.arrived, &fb.consumed, &fb.up, &fb.copied})
    if (!*ev)
      GHIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  return LK_ERROR_NONE;
}

int ensure_frame(lk_group *g, Member &me, FrameBuf &fb, size_t bytes) {
  if (int rc = ensure_events(g, me, fb))
    return rc;
  if (fb.cap >= bytes)
    return LK_ERROR_NONE;
  // readers / writers of the old buffer: the engine's fill (its own streams), the collectives (cst)
  GLK(lk_synchronize(me.e));
  GHIP(hipStreamSynchronize(me.cst));
  if (fb.p)
    GHIP(hipFree(fb.p));
  fb.p = nullptr;
  fb.cap = 0;
  fb.consumed_valid = fb.copied_valid = false;
  GHIP(hipMalloc((void **)&fb.p, bytes));
  fb.cap = bytes;
  return LK_ERROR_NONE;
}

// after a collective failed on some member: end what the others enqueued, refuse everything from now on
int collective_failed(lk_group *g, Member &me, int rc) {
  if (me.comm) {
    (void)ncclCommAbort(me.comm);
    me.comm = nullptr;
  }
  g->broken = true; // (every member writes the same value; read by lk_group::run after the job)
  return rc;
}

// n consecutive frames (one image slot: n = 1; ring slots first .. first + n - 1, contiguous in memory) to every member:
// rank 0 uploads them into its buffer, the pixels travel in ONE transfer (ncclBroadcast; or device-to-device copies from
// rank 0's buffer), every member's engine builds the pyramids.  All of it on the communication stream and the engine's
// fill stream, ordered by events: the host returns when everything is ENQUEUED.
// which(member, i): the member's FrameBuf of frame i; src(i): its pixels (host, or rank 0's device);
// fill(member, i, buf): hands the pixels to the member's engine (image slot or ring slot).
// prefetch: the frames are not needed by the next solve (LK_IMG_NXT, ring slots of the next window) - they travel on the
// communication stream, beside the running solve; frames the next solve reads (LK_IMG_UND / LK_IMG_DEF) stay in stream
// order on the solve stream: nothing to overlap with, and every cross-stream hop costs a few microseconds.
int distribute_frames(lk_group *g, Member &me, int n_frames, const std::function<FrameBuf &(Member &, int)> &which,
                      const std::function<const void *(int)> &src, bool on_device0, int rows, int cols, int step,
                      const std::function<int(Member &, int, FrameBuf &)> &fill, bool prefetch) {
  const size_t bytes = (size_t)rows * (size_t)cols;
  FrameBuf &f0 = which(me, 0);
  hipStream_t cs = prefetch ? comm_stream(me) : me.st;
  const bool by_copy = g->loopback || g->copy_frames;
  int up = me.device_ok ? LK_ERROR_NONE : LK_ERROR_DEVICE;
  auto hip = [&](hipError_t he, const char *what, int &rc) {
    if (he != hipSuccess && !rc) {
      g->msg[(size_t)me.rank] = std::string(what) + ": " + hipGetErrorString(he);
      rc = LK_ERROR_DEVICE;
    }
  };
  if (!up) {
    if (me.probe[0]) {
      hip(hipEventRecord(me.probe[0], cs), "hipEventRecord", up);
      me.probe_valid[0] = true;
    }
    for (int i = 0; i < n_frames; ++i) {
      FrameBuf &fb = which(me, i);
      if (fb.consumed_valid) // the pyramid kernel of the frame this buffer held before
        hip(hipStreamWaitEvent(cs, fb.consumed, 0), "hipStreamWaitEvent", up);
    }
    if (me.rank == 0) {
      if (by_copy)
        for (Member &q : g->m) { // the other ranks copied the previous frames out of this buffer
          FrameBuf &fq = which(q, 0);
          if (q.rank != 0 && fq.copied_valid)
            hip(hipStreamWaitEvent(cs, fq.copied, 0), "hipStreamWaitEvent", up);
        }
      for (int i = 0; i < n_frames; ++i)
        hip(hipMemcpy2DAsync(which(me, i).p, (size_t)cols, src(i), (size_t)step, (size_t)cols, (size_t)rows,
                             on_device0 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, cs), "hipMemcpy2DAsync", up);
      if (by_copy)
        hip(hipEventRecord(f0.up, cs), "hipEventRecord", up);
    }
  }
  if (int rc = g->agree(up)) // nobody enters the collective unless every member got this far (and rank 0's upload is enqueued)
    return up ? up : rc;
  if (!by_copy) {
    int rc = LK_ERROR_NONE;
    const ncclResult_t ne = ncclBroadcast(f0.p, f0.p, (size_t)n_frames * bytes, ncclUint8, 0, me.comm, cs);
    if (ne != ncclSuccess) {
      g->msg[(size_t)me.rank] = std::string("ncclBroadcast: ") + ncclGetErrorString(ne);
      rc = LK_ERROR_DEVICE;
    }
    if (int all = g->agree(rc))
      return collective_failed(g, me, rc ? rc : all);
  } else if (g->m.size() > 1) {
    int rc = LK_ERROR_NONE;
    if (me.rank != 0) {
      FrameBuf &r0 = which(g->m[0], 0);
      hip(hipStreamWaitEvent(cs, r0.up, 0), "hipStreamWaitEvent", rc);
      hip(hipMemcpyAsync(f0.p, r0.p, (size_t)n_frames * bytes, hipMemcpyDeviceToDevice, cs), "hipMemcpyAsync", rc);
      hip(hipEventRecord(f0.copied, cs), "hipEventRecord", rc);
      f0.copied_valid = !rc;
    }
    if (int all = g->agree(rc))
      return rc ? rc : all;
  }
  for (int i = 0; i < n_frames; ++i)
    GHIP(hipEventRecord(which(me, i).arrived, cs));
  if (me.probe[1])
    GHIP(hipEventRecord(me.probe[1], cs));
  for (int i = 0; i < n_frames; ++i) {
    FrameBuf &fb = which(me, i);
    if (int rc = fill(me, i, fb))
      return rc;
    fb.consumed_valid = true;
  }
  return LK_ERROR_NONE;
}

} // namespace

extern "C" {

int lk_group_create(const lk_config *cfg, int n_devices, const int *devices, lk_group **out) {
  if (!cfg || !out || n_devices < 1 || n_devices > 64)
    return LK_ERROR_BAD_DOMAIN;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return LK_ERROR_DEVICE; // no silent CPU fallback
  lk_group *g = new lk_group();
  g->cfg = *cfg;
  g->m.resize((size_t)n_devices);
  g->rc.assign((size_t)n_devices, 0);
  g->msg.assign((size_t)n_devices, std::string());
  std::vector<int> devs((size_t)n_devices);
  // LK_GROUP_DEVICES=0,0,0 (rehearsal hook, only when the caller names no devices - HipCudaClass::set_deviceCount):
  // the device list, e.g. one GPU three times = the loopback transport on a one-GPU box
  std::vector<int> env_devs;
  if (!devices)
    if (const char *f = std::getenv("LK_GROUP_DEVICES"))
      for (const char *q = f; *q;) {
        env_devs.push_back(std::atoi(q));
        while (*q && *q != ',')
          ++q;
        if (*q == ',')
          ++q;
      }
  for (int r = 0; r < n_devices; ++r) {
    devs[(size_t)r] = devices ? devices[r] : ((size_t)r < env_devs.size() ? env_devs[(size_t)r] : r);
    if (devs[(size_t)r] < 0 || devs[(size_t)r] >= ndev) {
      delete g;
      return LK_ERROR_DEVICE;
    }
    for (int q = 0; q < r; ++q)
      g->loopback = g->loopback || devs[(size_t)q] == devs[(size_t)r];
  }
  if (const char *f = std::getenv("LK_GROUP_FRAMES"))
    g->copy_frames = std::string(f) == "copy";
  bool ok = true;
  for (int r = 0; r < n_devices && ok; ++r) {
    Member &me = g->m[(size_t)r];
    me.rank = r;
    me.device = devs[(size_t)r];
    lk_config c = *cfg;
    c.device = me.device;
    ok = lk_create(&c, &me.e) == LK_ERROR_NONE && hipSetDevice(me.device) == hipSuccess &&
         hipStreamCreateWithFlags(&me.st, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&me.cst, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&me.ev_solved, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&me.ev_gathered, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&me.ev_block, hipEventDisableTiming) == hipSuccess &&
         lk_set_stream(me.e, me.st) == LK_ERROR_NONE && lk_internal_set_defer_stale(me.e, 1) == LK_ERROR_NONE;
    for (hipEvent_t &ev : me.probe)
      ok = ok && hipEventCreate(&ev) == hipSuccess;
  }
  if (ok && !g->loopback) {
    std::vector<ncclComm_t> comms((size_t)n_devices);
    ok = ncclCommInitAll(comms.data(), n_devices, devs.data()) == ncclSuccess;
    if (ok)
      for (int r = 0; r < n_devices; ++r)
        g->m[(size_t)r].comm = comms[(size_t)r];
  }
  if (!ok) {
    lk_group_destroy(g);
    return LK_ERROR_DEVICE;
  }
  for (int r = 0; r < n_devices; ++r)
    g->m[(size_t)r].th = std::thread(worker, g, r);
  *out = g;
  return LK_ERROR_NONE;
}

void lk_group_destroy(lk_group *g) {
  if (!g)
    return;
  {
    std::unique_lock<std::mutex> lock(g->mu);
    g->quit = true;
    g->cv_job.notify_all();
  }
  for (Member &me : g->m) {
    if (me.th.joinable())
      me.th.join();
    (void)hipSetDevice(me.device);
    if (me.e)
      (void)lk_synchronize(me.e);
    if (me.comm)
      (void)ncclCommDestroy(me.comm);
    if (me.cst)
      (void)hipStreamSynchronize(me.cst);
    auto drop = [](FrameBuf &fb) {
      if (fb.p)
        (void)hipFree(fb.p);
      for (hipEvent_t ev : {fb.arrived, fb.consumed, fb.up, fb.copied})
        if (ev)
          (void)hipEventDestroy(ev);
    };
    for (FrameBuf &fb : me.frame)
      drop(fb);
    for (FrameBuf &fb : me.ring) {
      fb.p = nullptr; // (slices of d_ring)
      drop(fb);
    }
    if (me.d_ring)
      (void)hipFree(me.d_ring);
    for (void *p : {(void *)me.d_guess, (void *)me.d_rec, (void *)me.d_all, (void *)me.d_stale, (void *)me.d_seq_rec, (void *)me.d_seq_all})
      if (p)
        (void)hipFree(p);
    for (hipEvent_t ev : {me.ev_solved, me.ev_gathered, me.ev_block, me.probe[0], me.probe[1], me.probe[2], me.probe[3]})
      if (ev)
        (void)hipEventDestroy(ev);
    if (me.e)
      lk_destroy(me.e);
    if (me.st)
      (void)hipStreamDestroy(me.st);
    if (me.cst)
      (void)hipStreamDestroy(me.cst);
  }
  delete g;
}

const char *lk_group_last_error_string(const lk_group *g) { return g ? g->err.c_str() : "null group"; }
int lk_group_size(const lk_group *g) { return g ? (int)g->m.size() : 0; }
int lk_group_comm_ranks(const lk_group *g) {
  if (!g || g->m.empty() || !g->m[0].comm)
    return 0; // rehearsal transport (one device listed several times): no RCCL communicator
  int n = 0;
  return ncclCommCount(g->m[0].comm, &n) == ncclSuccess ? n : -1;
}

int lk_group_engine(lk_group *g, int rank, lk_engine **e) {
  if (!g || !e || rank < 0 || rank >= (int)g->m.size())
    return LK_ERROR_BAD_DOMAIN;
  *e = g->m[(size_t)rank].e;
  return LK_ERROR_NONE;
}

int lk_group_shard_range(int n_sectors, int rank, int n_ranks, int *first, int *count) {
  if (n_sectors < 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks || !first || !count)
    return LK_ERROR_BAD_DOMAIN;
  shard(n_sectors, rank, n_ranks, *first, *count);
  return LK_ERROR_NONE;
}

int lk_group_shard(const lk_group *g, int rank, int *first, int *count) {
  if (!g || rank < 0 || rank >= (int)g->m.size() || !first || !count)
    return LK_ERROR_BAD_DOMAIN;
  const int S = g->committed ? g->S : (g->grid ? g->ghs * g->gvs : (int)g->secs.size());
  shard(S, rank, (int)g->m.size(), *first, *count);
  return LK_ERROR_NONE;
}

// ---- images -------------------------------------------------------------------------------------
static int set_image_any(lk_group *g, int slot, const void *src, bool on_device0, int rows, int cols, int step) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  if (slot < 0 || slot > 2 || !src || rows < 1 || cols < 1 || step < cols)
    return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_set_image: bad arguments");
  // Growing the frame buffers can fail on one device alone (out of memory); a member that gave up before the
  // broadcast would leave the others' collective waiting for ever.  So allocations get a job of their own, and
  // the job with the collective starts only when every member has its buffer.
  const size_t bytes = (size_t)rows * (size_t)cols;
  bool grow = false;
  for (const Member &me : g->m)
    grow = grow || me.frame[slot].cap < bytes || !me.frame[slot].arrived;
  if (grow)
    if (int rc = g->run([=](Member &me) -> int { return ensure_frame(g, me, me.frame[slot], bytes); }))
      return rc;
  return g->run([=](Member &me) -> int {
    return distribute_frames(g, me, 1, [slot](Member &q, int) -> FrameBuf & { return q.frame[slot]; },
                             [src](int) { return src; }, on_device0, rows, cols, step,
                             [=](Member &q, int, FrameBuf &fb) -> int {
                               Member &me = q; // (GLK names `me`)
                               GLK(lk_internal_set_image_device_after(q.e, slot, fb.p, rows, cols, cols, fb.arrived, fb.consumed));
                               return LK_ERROR_NONE;
                             }, slot == LK_IMG_NXT);
  }, true);
}
int lk_group_set_image(lk_group *g, int slot, const uint8_t *host_pixels, int rows, int cols, int step) {
  return set_image_any(g, slot, host_pixels, false, rows, cols, step);
}
int lk_group_set_image_device(lk_group *g, int slot, const void *device0_pixels, int rows, int cols, int step) {
  return set_image_any(g, slot, device0_pixels, true, rows, cols, step);
}
int lk_group_rotate_und_from_def(lk_group *g) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  return g->run([=](Member &me) -> int {
    GLK(lk_rotate_und_from_def(me.e));
    return LK_ERROR_NONE;
  });
}
int lk_group_rotate_def_from_nxt(lk_group *g) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  return g->run([=](Member &me) -> int {
    GLK(lk_rotate_def_from_nxt(me.e));
    return LK_ERROR_NONE;
  });
}

// ---- sectors ------------------------------------------------------------------------------------
int lk_group_clear_sectors(lk_group *g) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  g->secs.clear();
  g->grid = false;
  g->committed = false;
  g->S = 0;
  return LK_ERROR_NONE;
}
static SectorSpec *spec_slot(lk_group *g, int sector) {
  if (!g || sector < 0)
    return nullptr;
  if (g->grid) { // a grid is replaced by individually registered sectors
    g->grid = false;
    g->secs.clear();
  }
  if ((size_t)sector >= g->secs.size())
    g->secs.resize((size_t)sector + 1);
  g->committed = false;
  return &g->secs[(size_t)sector];
}
int lk_group_set_sector_rect(lk_group *g, int sector, int x0, int y0, int x1, int y1) {
  SectorSpec *s = spec_slot(g, sector);
  if (!s || x1 < x0 || y1 < y0)
    return g ? g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_set_sector_rect: bad rectangle") : LK_ERROR_BAD_DOMAIN;
  *s = SectorSpec();
  s->kind = 1, s->x0 = x0, s->y0 = y0, s->x1 = x1, s->y1 = y1;
  return LK_ERROR_NONE;
}
int lk_group_set_rect_grid(lk_group *g, float x_begin, float y_begin, float x_end, float y_end, int hs, int vs) {
  if (!g || hs < 1 || vs < 1)
    return g ? g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_set_rect_grid: bad subdivision") : LK_ERROR_BAD_DOMAIN;
  g->secs.clear();
  g->grid = true;
  g->gx0 = x_begin, g->gy0 = y_begin, g->gx1 = x_end, g->gy1 = y_end, g->ghs = hs, g->gvs = vs;
  g->committed = false;
  return LK_ERROR_NONE;
}
int lk_group_set_sector_annular(lk_group *g, int sector, float r, float dr, float a, float da, float cx, float cy,
                                int as) {
  SectorSpec *s = spec_slot(g, sector);
  if (!s)
    return LK_ERROR_BAD_DOMAIN;
  *s = SectorSpec();
  s->kind = 2, s->r = r, s->dr = dr, s->a = a, s->da = da, s->cx = cx, s->cy = cy, s->as = as;
  return LK_ERROR_NONE;
}
int lk_group_set_sector_points(lk_group *g, int sector, const float *xy, int n, int use_center, float cx, float cy) {
  SectorSpec *s = spec_slot(g, sector);
  if (!s || !xy || n < 1)
    return g ? g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_set_sector_points: empty list") : LK_ERROR_BAD_DOMAIN;
  *s = SectorSpec();
  s->kind = 3, s->xy.assign(xy, xy + 2 * (size_t)n), s->use_center = use_center, s->cx = cx, s->cy = cy;
  return LK_ERROR_NONE;
}
int lk_group_sector_count(const lk_group *g) {
  return g ? (g->committed ? g->S : (g->grid ? g->ghs * g->gvs : (int)g->secs.size())) : 0;
}

int lk_group_commit_sectors(lk_group *g) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  const int n = (int)g->m.size();
  const int S = g->grid ? g->ghs * g->gvs : (int)g->secs.size();
  if (S < n)
    return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_commit_sectors: fewer sectors than devices");
  if (!g->grid)
    for (int s = 0; s < S; ++s)
      if (g->secs[(size_t)s].kind == 0)
        return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_commit_sectors: sector " + std::to_string(s) + " was never set");
  const int cap = (S + n - 1) / n; // equal all-gather blocks: the shards differ by at most one sector
  int rc = g->run([=](Member &me) -> int {
    shard(S, me.rank, n, me.first, me.count);
    if (g->grid) {
      GLK(lk_set_rect_grid(me.e, g->gx0, g->gy0, g->gx1, g->gy1, g->ghs, g->gvs, me.first, me.count));
    } else {
      GLK(lk_clear_sectors(me.e));
      for (int k = 0; k < me.count; ++k) {
        const SectorSpec &s = g->secs[(size_t)(me.first + k)];
        if (s.kind == 1)
          GLK(lk_set_sector_rect(me.e, k, s.x0, s.y0, s.x1, s.y1));
        else if (s.kind == 2)
          GLK(lk_set_sector_annular(me.e, k, s.r, s.dr, s.a, s.da, s.cx, s.cy, s.as));
        else
          GLK(lk_set_sector_points(me.e, k, s.xy.data(), (int)(s.xy.size() / 2), s.use_center, s.cx, s.cy));
      }
    }
    GLK(lk_commit_sectors(me.e));
    if (me.cap < (size_t)cap) {
      GLK(lk_synchronize(me.e));
      GHIP(hipStreamSynchronize(me.cst));
      me.gathered_valid = false;
      for (void *p : {(void *)me.d_guess, (void *)me.d_rec, (void *)me.d_all})
        if (p)
          GHIP(hipFree(p));
      me.d_guess = nullptr, me.d_rec = nullptr, me.d_all = nullptr, me.cap = 0;
      GHIP(hipMalloc((void **)&me.d_guess, 6 * (size_t)cap * sizeof(float)));
      GHIP(hipMalloc((void **)&me.d_rec, (size_t)cap * sizeof(lk_result)));
      GHIP(hipMalloc((void **)&me.d_all, (size_t)n * (size_t)cap * sizeof(lk_result)));
      me.cap = (size_t)cap;
    }
    GHIP(hipMemsetAsync(me.d_rec, 0, me.cap * sizeof(lk_result), me.st)); // (the padding record stays zero)
    GHIP(hipMemsetAsync(me.d_guess, 0, 6 * me.cap * sizeof(float), me.st));
    return LK_ERROR_NONE;
  });
  if (rc)
    return rc;
  g->S = S;
  g->cap = cap;
  g->committed = true;
  return LK_ERROR_NONE;
}

// ---- the solve ----------------------------------------------------------------------------------
// The exchange of one solve's (or one window's) records, on the communication stream behind `me.ev_solved`:
// send = this member's padded block, recv = n blocks; frames = 1 (one pair) or the window's frame count.
static int gather_records(lk_group *g, Member &me, lk_result *send, lk_result *recv, size_t block_bytes, int frames, int rc_before,
                          lk_result *out) {
  const int n = (int)g->m.size();
  hipStream_t cs = comm_stream(me);
  int rc = rc_before;
  auto hip = [&](hipError_t he, const char *what) {
    if (he != hipSuccess && !rc) {
      g->msg[(size_t)me.rank] = std::string(what) + ": " + hipGetErrorString(he);
      rc = LK_ERROR_DEVICE;
    }
  };
  if (int all = g->agree(rc)) // nobody enqueues the all-gather unless every member's solve is on its way
    return rc ? rc : all;
  hip(hipStreamWaitEvent(cs, me.ev_solved, 0), "hipStreamWaitEvent");
  if (!g->loopback) {
    if (!rc) {
      const ncclResult_t ne = ncclAllGather(send, recv, block_bytes, ncclUint8, me.comm, cs);
      if (ne != ncclSuccess) {
        g->msg[(size_t)me.rank] = std::string("ncclAllGather: ") + ncclGetErrorString(ne);
        rc = LK_ERROR_DEVICE;
      }
    }
    if (int all = g->agree(rc))
      return collective_failed(g, me, rc ? rc : all);
  } else {
    // rehearsal transport: every member copies its block into everybody's receive buffer and tells them by an event
    for (int q = 0; q < n && !rc; ++q) {
      lk_result *dst = frames > 1 ? g->m[(size_t)q].d_seq_all : g->m[(size_t)q].d_all;
      hip(hipMemcpyAsync((char *)dst + (size_t)me.rank * block_bytes, send, block_bytes, hipMemcpyDeviceToDevice, cs), "hipMemcpyAsync");
    }
    hip(hipEventRecord(me.ev_block, cs), "hipEventRecord");
    rc = g->agree(rc); // every member's event is on its stream (or everybody knows a copy failed)
    if (!rc)
      for (int q = 0; q < n; ++q)
        if (q != me.rank)
          hip(hipStreamWaitEvent(cs, g->m[(size_t)q].ev_block, 0), "hipStreamWaitEvent");
    if (int all = g->agree(rc)) // (nobody re-records its event - the next exchange - before everybody has waited for it)
      return rc ? rc : all;
  }
  // reference-order mode: the stale iteration counts, over the gathered records in the order the reference solves
  // (global sector order, frame by frame) with the group's own carry; every member resolves its copy - all copies end up
  // identical - and takes its own resolved block back into its engine's record buffer
  if (lk_internal_reference_order(me.e) > 0) {
    if (!me.d_stale) {
      GHIP(hipMalloc((void **)&me.d_stale, 2 * sizeof(int)));
      GHIP(hipMemsetAsync(me.d_stale, 0, 2 * sizeof(int), cs));
    }
    if (frames > 1)
      GHIP(lk_launch_stale_iterations_window(recv, g->S, n, g->cap, frames, me.d_stale + me.stale_par, me.d_stale + (me.stale_par ^ 1), cs));
    else
      GHIP(lk_launch_stale_iterations_blocks(recv, g->S, n, g->cap, me.d_stale + me.stale_par, me.d_stale + (me.stale_par ^ 1), cs));
    me.stale_par ^= 1;
    const void *d_own = nullptr;
    GLK(lk_get_results_device(me.e, &d_own));
    const lk_result *mine = (const lk_result *)((const char *)recv + (size_t)me.rank * block_bytes) + (size_t)(frames - 1) * (size_t)g->cap;
    GHIP(hipMemcpyAsync(const_cast<void *>(d_own), mine, (size_t)me.count * sizeof(lk_result), hipMemcpyDeviceToDevice, cs));
  }
  if (out && me.rank == 0) { // global sector order: block q starts at rank q's first sector
    for (int q = 0; q < n; ++q) {
      int first, count;
      shard(g->S, q, n, first, count);
      GHIP(hipMemcpy2DAsync(out + first, (size_t)g->S * sizeof(lk_result), (const char *)recv + (size_t)q * block_bytes,
                            (size_t)g->cap * sizeof(lk_result), (size_t)count * sizeof(lk_result), (size_t)frames, hipMemcpyDeviceToHost, cs));
    }
  }
  GHIP(hipEventRecord(me.ev_gathered, cs));
  me.gathered_valid = true;
  if (out && me.rank == 0)
    GHIP(hipStreamSynchronize(cs));
  return LK_ERROR_NONE;
}

int lk_group_correlate_all(lk_group *g, const float *guesses, lk_result *out) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  if (!g->committed)
    return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_correlate_all: sectors are not committed");
  const size_t block = (size_t)g->cap * sizeof(lk_result);
  return g->run([=](Member &me) -> int {
    // phase 1 - what can fail on this member alone: the guess upload, the engine's deferred re-commit /
    // allocations, the solve launches.  The records go to the ENGINE's own buffer: lk_update_sector
    // (HipCudaClass::updatePolygon) moves a sector by the engine's record of it.
    const void *d_own = nullptr;
    int rc = [&]() -> int {
      if (!me.device_ok)
        return LK_ERROR_DEVICE;
      if (const char *f = std::getenv("LK_GROUP_FAULT")) // test hook: this rank fails alone, before the collective
        if (std::atoi(f) == me.rank) {
          g->msg[(size_t)me.rank] = "LK_GROUP_FAULT: injected failure before the all-gather";
          return LK_ERROR_DEVICE;
        }
      // the previous exchange still reads the padded block (and, in reference-order mode, writes the engine's records)
      const bool ordered = lk_internal_reference_order(me.e) > 0;
      if (me.gathered_valid && ordered)
        GHIP(hipStreamWaitEvent(me.st, me.ev_gathered, 0));
      if (guesses)
        GHIP(hipMemcpyAsync(me.d_guess, guesses + 6 * (size_t)me.first, 6 * (size_t)me.count * sizeof(float),
                            hipMemcpyHostToDevice, me.st));
      GHIP(hipEventRecord(me.probe[2], me.st));
      GLK(lk_correlate_all_device(me.e, guesses ? me.d_guess : nullptr, nullptr));
      GHIP(hipEventRecord(me.probe[3], me.st));
      me.probe_valid[1] = true;
      GLK(lk_get_results_device(me.e, &d_own));
      if (me.gathered_valid && !ordered)
        GHIP(hipStreamWaitEvent(me.st, me.ev_gathered, 0));
      GHIP(hipMemcpyAsync(me.d_rec, d_own, (size_t)me.count * sizeof(lk_result), hipMemcpyDeviceToDevice, me.st)); // (the padded all-gather block)
      GHIP(hipEventRecord(me.ev_solved, me.st));
      return LK_ERROR_NONE;
    }();
    // phases 2, 3 - the exchange, behind the solve on the communication stream
    return gather_records(g, me, me.d_rec, me.d_all, block, 1, rc, out);
  }, true);
}

// ---- frame-pipelined windows (lk_correlate_sequence_async on every member) --------------------------------------
int lk_group_sequence_reserve(lk_group *g, int n_slots) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  if (n_slots < 1 || n_slots > 4096)
    return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_sequence_reserve: n_slots must be in 1..4096");
  return g->run([=](Member &me) -> int {
    GLK(lk_sequence_reserve(me.e, n_slots));
    if ((int)me.ring.size() < n_slots) { // (the pixels' allocation follows with the first frame: its size is not known yet)
      GLK(lk_synchronize(me.e));
      GHIP(hipStreamSynchronize(me.cst));
      if (me.d_ring)
        GHIP(hipFree(me.d_ring));
      me.d_ring = nullptr;
      me.ring_slot_bytes = 0;
      me.ring.resize((size_t)n_slots);
      for (FrameBuf &fb : me.ring) {
        fb.p = nullptr;
        fb.cap = 0;
        fb.consumed_valid = fb.copied_valid = false;
      }
    }
    return LK_ERROR_NONE;
  });
}

static int ensure_ring(lk_group *g, Member &me, size_t bytes) {
  for (FrameBuf &fb : me.ring)
    if (int rc = ensure_events(g, me, fb))
      return rc;
  if (me.d_ring && me.ring_slot_bytes >= bytes)
    return LK_ERROR_NONE;
  GLK(lk_synchronize(me.e));
  GHIP(hipStreamSynchronize(me.cst));
  if (me.d_ring)
    GHIP(hipFree(me.d_ring));
  me.d_ring = nullptr;
  me.ring_slot_bytes = 0;
  GHIP(hipMalloc((void **)&me.d_ring, me.ring.size() * bytes));
  me.ring_slot_bytes = bytes;
  for (size_t i = 0; i < me.ring.size(); ++i) {
    me.ring[i].p = me.d_ring + i * bytes;
    me.ring[i].cap = bytes;
    me.ring[i].consumed_valid = me.ring[i].copied_valid = false;
  }
  return LK_ERROR_NONE;
}

// n frames into the ring slots first_slot .. first_slot + n - 1 (no wrap-around: split the call), in ONE transfer
static int sequence_set_frames_any(lk_group *g, int first_slot, int n_frames, const std::function<const void *(int)> &src, bool on_device0,
                                   int rows, int cols, int step) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  if (n_frames < 1 || first_slot < 0 || first_slot + n_frames > (int)g->m[0].ring.size() || rows < 1 || cols < 1 || step < cols)
    return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_sequence_set_frames: bad arguments / slots outside the ring (lk_group_sequence_reserve)");
  for (int i = 0; i < n_frames; ++i)
    if (!src(i))
      return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_sequence_set_frames: null frame");
  const size_t bytes = (size_t)rows * (size_t)cols;
  bool grow = false;
  for (const Member &me : g->m)
    grow = grow || me.ring_slot_bytes != bytes || !me.ring[0].arrived;
  if (grow) // (one frame size per ring; allocations in a job of their own: see set_image_any)
    if (int rc = g->run([=](Member &me) -> int { return ensure_ring(g, me, bytes); }))
      return rc;
  return g->run([=](Member &me) -> int {
    return distribute_frames(g, me, n_frames, [first_slot](Member &q, int i) -> FrameBuf & { return q.ring[(size_t)(first_slot + i)]; },
                             src, on_device0, rows, cols, step,
                             [=](Member &q, int i, FrameBuf &fb) -> int {
                               Member &me = q;
                               GLK(lk_internal_sequence_set_frame_device_after(q.e, first_slot + i, fb.p, rows, cols, cols, fb.arrived, fb.consumed));
                               return LK_ERROR_NONE;
                             }, true);
  }, true);
}
int lk_group_sequence_set_frames(lk_group *g, int first_slot, int n_frames, const uint8_t *const *host_pixels, int rows, int cols, int step) {
  if (!host_pixels)
    return LK_ERROR_BAD_DOMAIN;
  return sequence_set_frames_any(g, first_slot, n_frames, [host_pixels](int i) -> const void * { return host_pixels[i]; }, false, rows, cols, step);
}
int lk_group_sequence_set_frames_device(lk_group *g, int first_slot, int n_frames, const void *device0_pixels, int rows, int cols, int step) {
  if (!device0_pixels)
    return LK_ERROR_BAD_DOMAIN;
  const char *base = (const char *)device0_pixels;
  const size_t pitch = (size_t)rows * (size_t)step; // frames back to back
  return sequence_set_frames_any(g, first_slot, n_frames, [base, pitch](int i) -> const void * { return base + (size_t)i * pitch; }, true, rows, cols, step);
}

int lk_group_correlate_sequence_async(lk_group *g, int und_slot, int first_slot, int n_frames, int reference_previous,
                                      int constant_velocity) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  if (!g->committed)
    return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_correlate_sequence_async: sectors are not committed");
  if (n_frames < 1)
    return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_correlate_sequence_async: no frames");
  return g->run([=](Member &me) -> int {
    if (me.seq_frames)
      return LK_ERROR_BAD_DOMAIN;
    GHIP(hipEventRecord(me.probe[2], me.st));
    GLK(lk_correlate_sequence_async(me.e, und_slot, first_slot, n_frames, reference_previous, constant_velocity, 0));
    GHIP(hipEventRecord(me.probe[3], me.st));
    me.probe_valid[1] = true;
    me.seq_frames = n_frames;
    return LK_ERROR_NONE;
  });
}

int lk_group_wait_sequence(lk_group *g, lk_result *out) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  const int n = (int)g->m.size();
  int frames = 0;
  for (const Member &me : g->m)
    frames = std::max(frames, me.seq_frames);
  if (frames < 1)
    return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_wait_sequence: no window outstanding");
  const size_t block = (size_t)frames * (size_t)g->cap * sizeof(lk_result);
  bool grow = false;
  for (const Member &me : g->m)
    grow = grow || me.seq_frames_cap < (size_t)frames;
  if (grow) // (allocations in a job of their own: see set_image_any)
    if (int rc = g->run([=](Member &me) -> int {
          GHIP(hipStreamSynchronize(me.cst));
          for (void *p : {(void *)me.d_seq_rec, (void *)me.d_seq_all})
            if (p)
              GHIP(hipFree(p));
          me.d_seq_rec = me.d_seq_all = nullptr;
          me.seq_frames_cap = 0;
          GHIP(hipMalloc((void **)&me.d_seq_rec, block));
          GHIP(hipMalloc((void **)&me.d_seq_all, (size_t)n * block));
          GHIP(hipMemsetAsync(me.d_seq_rec, 0, block, me.st)); // (the padding records stay zero)
          me.seq_frames_cap = (size_t)frames;
          return LK_ERROR_NONE;
        }))
      return rc;
  return g->run([=](Member &me) -> int {
    int rc = [&]() -> int {
      if (!me.device_ok || me.seq_frames != frames)
        return LK_ERROR_DEVICE;
      GLK(lk_wait_sequence(me.e, nullptr)); // (this member's window is solved; its sequence state is committed)
      const void *d_win = nullptr;
      GLK(lk_get_sequence_results_device(me.e, &d_win, nullptr));
      if (me.gathered_valid)
        GHIP(hipStreamWaitEvent(me.st, me.ev_gathered, 0)); // (the previous exchange read the padded block)
      GHIP(hipMemcpy2DAsync(me.d_seq_rec, (size_t)g->cap * sizeof(lk_result), d_win, (size_t)me.count * sizeof(lk_result),
                            (size_t)me.count * sizeof(lk_result), (size_t)frames, hipMemcpyDeviceToDevice, me.st));
      GHIP(hipEventRecord(me.ev_solved, me.st));
      return LK_ERROR_NONE;
    }();
    me.seq_frames = 0;
    me.exchanged_frames = 0;
    const int grc = gather_records(g, me, me.d_seq_rec, me.d_seq_all, block, frames, rc, out);
    if (!grc)
      me.exchanged_frames = frames;
    return grc;
  }, true);
}

// The records of the window that lk_group_wait_sequence(g, NULL) last exchanged, fetched separately: waits for THAT exchange
// only, so that the next window - launched in between - is solved while the records travel (between the devices and down to
// the host).  The reference's loop has the frame k + 1 upload behind the solve of pair k (manager_class.cpp:1438-1447); this is
// the same for the way back.
int lk_group_sequence_records(lk_group *g, lk_result *out) {
  if (!g || !out)
    return LK_ERROR_BAD_DOMAIN;
  for (const Member &me : g->m)
    if (me.exchanged_frames < 1 || !me.gathered_valid)
      return g->fail(LK_ERROR_BAD_DOMAIN, "lk_group_sequence_records: no exchanged window (lk_group_wait_sequence(g, NULL) first)");
  const int n = (int)g->m.size();
  return g->run([=](Member &me) -> int {
    const int frames = me.exchanged_frames;
    const size_t block = (size_t)frames * (size_t)g->cap * sizeof(lk_result);
    GHIP(hipEventSynchronize(me.ev_gathered)); // (every member: the exchange is complete on all devices when this returns)
    if (me.rank != 0)
      return LK_ERROR_NONE;
    hipStream_t cs = comm_stream(me);
    for (int q = 0; q < n; ++q) { // global sector order: block q starts at rank q's first sector
      int first, count;
      shard(g->S, q, n, first, count);
      GHIP(hipMemcpy2DAsync(out + first, (size_t)g->S * sizeof(lk_result), (const char *)me.d_seq_all + (size_t)q * block,
                            (size_t)g->cap * sizeof(lk_result), (size_t)count * sizeof(lk_result), (size_t)frames, hipMemcpyDeviceToHost, cs));
    }
    GHIP(hipStreamSynchronize(cs));
    return LK_ERROR_NONE;
  });
}

int lk_group_sequence_records_device(lk_group *g, int rank, const void **d_records) {
  if (!g || !d_records || rank < 0 || rank >= (int)g->m.size() || !g->m[(size_t)rank].d_seq_all)
    return LK_ERROR_BAD_DOMAIN;
  *d_records = g->m[(size_t)rank].d_seq_all;
  return LK_ERROR_NONE;
}

// what overlapped on `rank`'s device: ms_out[0] = duration of the last frame transfer (communication stream), [1] = of the
// last solve / window launch chain (solve stream), [2] = transfer begin - solve begin, [3] = solve end - transfer end
// (both positive: the transfer ran inside the solve).  Waits for both.
int lk_group_probe_overlap(lk_group *g, int rank, float *ms_out) {
  if (!g || !ms_out || rank < 0 || rank >= (int)g->m.size())
    return LK_ERROR_BAD_DOMAIN;
  return g->run([=](Member &me) -> int {
    if (me.rank != rank)
      return LK_ERROR_NONE;
    if (!me.probe_valid[0] || !me.probe_valid[1]) {
      g->msg[(size_t)me.rank] = "lk_group_probe_overlap: no frame transfer or no solve yet";
      return LK_ERROR_BAD_DOMAIN;
    }
    GHIP(hipEventSynchronize(me.probe[1]));
    GHIP(hipEventSynchronize(me.probe[3]));
    GHIP(hipEventElapsedTime(&ms_out[0], me.probe[0], me.probe[1]));
    GHIP(hipEventElapsedTime(&ms_out[1], me.probe[2], me.probe[3]));
    GHIP(hipEventElapsedTime(&ms_out[2], me.probe[2], me.probe[0]));
    GHIP(hipEventElapsedTime(&ms_out[3], me.probe[1], me.probe[3]));
    return LK_ERROR_NONE;
  });
}

int lk_group_adjust_initial_guess(lk_group *g, int frame, int constant_velocity, const float *global_guess,
                                  float global_cx, float global_cy) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  float gg[6] = {0, 0, 0, 0, 0, 0};
  if (global_guess)
    std::memcpy(gg, global_guess, sizeof gg);
  const std::vector<float> ggv(gg, gg + 6);
  return g->run([=](Member &me) -> int {
    GLK(lk_adjust_initial_guess(me.e, frame, constant_velocity, ggv.data(), global_cx, global_cy));
    return LK_ERROR_NONE;
  });
}

int lk_group_records_device(lk_group *g, int rank, const void **d_records) {
  if (!g || !d_records || rank < 0 || rank >= (int)g->m.size() || !g->committed)
    return LK_ERROR_BAD_DOMAIN;
  *d_records = g->m[(size_t)rank].d_all;
  return LK_ERROR_NONE;
}
int lk_group_block_records(const lk_group *g) { return g ? g->cap : 0; }

int lk_group_synchronize(lk_group *g) {
  if (!g)
    return LK_ERROR_BAD_DOMAIN;
  return g->run([=](Member &me) -> int {
    GLK(lk_synchronize(me.e));
    GHIP(hipStreamSynchronize(me.st));
    GHIP(hipStreamSynchronize(me.cst));
    return LK_ERROR_NONE;
  });
}

int lk_group_get_stats(lk_group *g, lk_stats *out) {
  if (!g || !out)
    return LK_ERROR_BAD_DOMAIN;
  std::vector<lk_stats> st(g->m.size());
  int rc = g->run([&](Member &me) -> int {
    GLK(lk_get_stats(me.e, &st[(size_t)me.rank]));
    return LK_ERROR_NONE;
  });
  if (rc)
    return rc;
  lk_stats s{};
  for (const lk_stats &q : st) {
    s.sectors += q.sectors;
    s.evaluations += q.evaluations;
    s.sample_evaluations += q.sample_evaluations;
    s.point_iterations += q.point_iterations;
    s.algorithmic_bytes += q.algorithmic_bytes;
    s.ill_conditioned_solves += q.ill_conditioned_solves;
    s.solve_ms = std::max(s.solve_ms, q.solve_ms);
    s.pyramid_ms = std::max(s.pyramid_ms, q.pyramid_ms);
  }
  *out = s;
  return LK_ERROR_NONE;
}

} // extern "C"
