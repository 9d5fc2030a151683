// lk_device.hpp - shared host/device structs of the gfx950 Lucas-Kanade engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lk_engine.h"

// One pyramid level as the solve kernel sees it.  Images are row-major u8 with
// pitch == cols (pyramid_class.cpp:153,177: step = cols) and two zeroed guard rows.
struct LkLevelView {
  const uint8_t *und;  // undeformed image, level L
  const uint8_t *def;  // deformed image, level L
  const float2 *xy;    // concatenated per-sector sample lists of level L (AoS x,y)
  const float2 *xy_eval; // the same lists with every sector's samples in row-major order (y outer, x inner), or null:
                         //     what the lane groups of the default mode walk - neighbouring lanes then read neighbouring
                         //     pixels of one image row.  The reference's order (x outer, y inner for annular sectors,
                         //     manager_class.cpp:907-918) puts 64 consecutive samples on 64 image rows: 64 cache lines
                         //     per load instruction.  Sums in the reference's order (starved levels, reference-order
                         //     mode) and the centres use `xy`.
  const uint32_t *off; // [S+1] start of each sector's list in xy
  const int4 *rect;    // [S] implicit rectangular sectors: {x_first, y_first, width, n} at this
                       //     level (the same sample SET as manager_class.cpp:1607-1611 + the
                       //     decimation of pyramid_class.cpp:301-322); width == 0 means
                       //     "use the explicit list"
  int urows, ucols;    // und dims at this level (rows0 >> L, pyramid_class.cpp:447-477)
  int drows, dcols;    // def dims
};

// What the starved-level kernel (one lane per sector) leaves for the lane-group kernel.
struct LkHandoff {
  float p[6];        // parameters in the scale of `level_old`
  int level;         // next pyramid level to solve; < py_start: the sector is finished
  int level_old;     // level the parameters are scaled for
  int reached;       // reached_iterations so far (correlation_class.cpp:452)
  uint32_t n_evals, n_sample_evals, n_point_iters;
};

// Frame-pipelined solve (lk_correlate_sequence*): the image pair of one frame of the window, level by level.
// The sample lists, rectangles and dimensions are those of LkLevelView (an Eulerian sequence: the sectors stay,
// only the images change from frame to frame).
struct LkSeqFrame {
  const uint8_t *und[LK_MAX_LEVELS];
  const uint8_t *def[LK_MAX_LEVELS];
};
// What one frame of a sector hands to the next: its returned parameters as six 8-byte granules {tag = frame + 1, float
// bits}, two slots per sector (frame parity) - one 128-byte line per sector.  A granule is written by ONE write-through
// store and read by write-through-aware loads; the tag is the flag, so no fence and no ordering between the six is needed
// (the consumer takes a value only when its tag is the frame it waits for).
constexpr int kLkSeqChainWords = 16; // uint64 per sector: [2 slots][8] (6 used)

constexpr int kLkMaxTeam = 256; // workgroups per team at most (the kernel stages a team's partial sums in LDS: 32 floats each)
constexpr int kLkMidWords = 32; // Cold (23) + p (6) + phase

struct LkSolveArgs {
  const LkLevelView *lv; // [LK_MAX_LEVELS] in device memory
  const float2 *center;  // [S] level-0 centre of each sector
  const float *guess;    // [S][6]
  lk_result *result;     // [S]
  float *last_p;         // [S][6] copy of the returned parameters (sequence state), may be null
  float *last_eval_p;    // [S][6] parameters of the LAST evaluation at level 0 - what CorrelationClass::getDefXY0
                         //   warps with (correlation_class.cpp:884-896); may be null
  uint32_t *stats;       // [S][4]: evaluations, sample evaluations, point iterations, -
  const uint32_t *order; // optional [n_sectors] indirection (size classes), may be null
  // teams: a giant sector is shared by team_w workgroups of the 512-thread kernel (0/1: off)
  int team_w;
  int team_min_samples;  // a sector's team has ceil(n0 / team_min_samples) workgroups (<= team_w)
  float *team_partials;  // [n_sectors][2][team_w][32]: per-workgroup sums, double-buffered by step parity
  uint32_t *team_arrivals; // [2][n_sectors]: monotonic arrival counters, then "team is broken" flags (zeroed per launch)
  int team_fault;          // test hook (LK_TEAM_FAULT = step): rank 1 of every team goes missing at that step
  // Stragglers of the starved-level kernel: after `eval_cap` evaluations a lane parks its sector
  // mid-level (kLkMidWords words of state) and appends it to `finish_list`; the 16-lane
  // finisher (`finisher` = 1) resumes it with reference-order sums spread over 16 lanes.
  int eval_cap;
  int finisher;
  int resume;              // 1: this launch takes its sectors, in the middle of a level, from finish_list
  // Fast-flavour kernels hand a sector whose damped system met a bad pivot to the SAFE 16-lane
  // kernel (resume = 1, the reference's QR) instead of zeroing that parameter's step:
  uint32_t *ill_list;      // [S] or null
  uint32_t *ill_count;     // [1]
  uint32_t *mid_state;    // [S][kLkMidWords]
  uint32_t *finish_list;  // [S]
  uint32_t *finish_count; // [1], zeroed before the starved-level kernel
  LkHandoff *handoff;    // [S] written by the starved-level kernel, read by the others (may be null)
  uint32_t *queue;       // next unclaimed slot of this launch (persistent mode), zeroed per launch
  int n_sectors;         // sectors in this launch
  int chunk;             // non-persistent: ceil(#workgroups / 8), XCD-contiguous chunk length
  int align;             // 1: the sectors of a wavefront (16/32-lane groups) descend the pyramid together
  int solo;              // 1: an idle half-wavefront may join its partner's sector (32-lane groups)
  int safe;              // 1: reference-exact handling of starved / ill-conditioned levels
  int reference_order;   // T > 0 (SAFE 16- / 64-lane kernels): every level with the reference's summation order for
                         //   number_of_threads = T and its QR - bit-identical records (lk_set_reference_order)
  int mark_stale;        // reference-order mode (any instance of its launch chain): a sector whose very first evaluation fails
                         //   reports the stale-iteration marker instead of 0 (lk_stale_iterations_kernel resolves it)
  int persistent;        // 1: groups pull sectors from `queue`; 0: one sector per group by position
  int rows_used;         // positional launches of 16-lane rows: rows of a wavefront that get a sector (0 / 4: all; 3, 2: the
                         //   reference-order rows deal the idle rows' lanes to the wavefront's sectors)
  int starved_max;       // a level with at most this many samples is "starved" (default 2 P)
  int keep_sums;         // 1 (default): a rejected trip continues from the kept sums of the last good parameters;
                         //   0 (LK_KEEP_SUMS=0, tests): it evaluates there again, as the reference does - same records
  int gpu_share;         // launches that may hold the GPU at the same time (>= 1): bounds a team launch's width
  int slots_permille;    // 0: the launch may use every resident workgroup slot; else its share of them in 1/1000 - the
                         //   team class and the one-workgroup class of one batch split the chip so that both are resident
                         //   at once (a team launch: bounds its width; the one-workgroup class: a persistent grid of that size)
  int py_start, py_step, py_stop;
  float precision;
  int max_iters;
  // Frame-pipelined solve of a sequence window (the SEQ instances): seq_frames > 0.  Tickets of the persistent queue
  // are (frame, sector) pairs drawn FRAME-MAJOR (ticket t: frame t / n_sectors, sector order[t % n_sectors]); the group
  // that draws (f, s) waits - without blocking the other sectors of its wavefront - until frame f - 1 of sector s has
  // published its parameters, forms the guess of managerClass::adjust_initial_guess (manager_class.cpp:2677-2699) from
  // them, solves, writes result[f * seq_stride + s] and publishes in turn.  Every wait targets a ticket that was drawn
  // earlier, i.e. one a running or finished wavefront holds: the grid drains whatever is resident.
  int seq_frames;
  int seq_velocity;             // 1: guess = 2 p(f-1) - p(f-2) (Eulerian description, first image as the reference), 0: p(f-1)
  int seq_stride;               // records per frame in `result` / `stats` / `seq_guess_out` (the engine's sector count)
  const LkSeqFrame *seq_img;    // [seq_frames]
  unsigned long long *seq_chain; // [S][kLkSeqChainWords] granules, zeroed before the launch
  const float *seq_prev_p;      // [S][6] previous_resulting_parameters before the window (p(f-2) of the window's frame 1)
  float *seq_prev_p_out;        // [S][6] ... and after it (written by the window's last frame; a buffer of its own)
  float *seq_guess_out;         // optional [seq_frames][seq_stride][6]: the guesses the frames were solved from
  uint32_t *seq_flags;          // [0]: a wait ran into its bound (the launch is void); [1]: a fast-flavour solve met a bad pivot
  int seq_fault;                // test hook (LK_SEQ_FAULT = f + 1): frame f of the launch's first sector never publishes - the bounded wait's exit
};

// ROI -> level-0 sample lists on the device (cudaPolygon's mask + compaction, cuda_polygon.cuh:180-292,
// cuda_polygon.cu:589-627, with the CPU engine's predicates and sample order: SURVEY.md 8a-a10).
// The work is cut into TILES whose outputs are consecutive in the final list:
//   annular sector  candidates = the bounding box walked x outer / y inner (manager_class.cpp:902-925),
//                   1024 candidates per tile, the predicate (:907-918) keeps some of them;
//   blob sector     one tile per scan line of each flat-sided half triangle, in the ear clipper's
//                   order (polygon_class.cpp:283-403); every pixel of the line is kept.
struct LkRoiSector {
  int kind;               // 0 annular, 1 blob
  int x0, y0, x1, y1;     // annular: candidates fx in [x0, x1), j in [y0, y1)
  float cx, cy, ri2, ro2; // annular: centre, squared radii
  float q00x, q01x, q10x, q11x, q00y, q01y, q10y, q11y; // annular: wedge corners (outer ones with the 1.2 "sag")
  int as;                 // annular: 1 = full ring (no wedge test)
  int flat_begin, flat_count; // blob: its half triangles in the flat table
};
struct LkRoiFlat { // one flat-sided half triangle: rows j0 .. j1-1, pixels ceil(ls*j+li) .. ceil(rs*j+ri)-1
  float ls, li, rs, ri;
  int j0, j1;
  int row_begin; // index of its first scan line among the sector's tiles
};
constexpr int kLkRoiTile = 1024;

struct LkRewarpArgs { // level-0 sample lists moved by the last level-0 evaluation's parameters (or by an offset)
  const float2 *src_xy;    // lists before (explicit sectors)
  const uint32_t *src_off; // [S+1]
  const int4 *src_rect;    // [S] implicit rectangles before (width 0: explicit)
  const uint32_t *dst_off; // [S+1] prefix of the level-0 counts: every sector explicit afterwards
  float2 *dst_xy;
  const float2 *center;    // [S] centres of the last solve
  const float *p;          // [S][6]
  const float2 *offset;    // [S] Lagrangian description: samples move by add_pair(offset) instead of the warp
  int n_sectors;
  uint32_t total;          // dst_off[S]
  int rows;                // 1: an implicit rectangle is walked row by row (y outer, x inner) - the evaluation copy of the lists
};

// One rectangular sector appended behind the committed ones (lk_commit_sectors' fast path): everything the
// device needs to know about it travels in the kernel arguments.
struct LkAppendArgs {
  int sector, n_levels;
  int keep_state; // 1: a committed rectangle that moved (lk_update_sector): rectangles and centre only
  float2 center;
  float2 *d_center;
  float *d_guess, *d_last_p, *d_prev_p, *d_last_eval_p; // [S][6] sequence state: zeroed
  lk_result *d_result;                                   // [S]: zeroed
  uint32_t *d_stats;                                     // [S][4]: zeroed
  int4 rect[LK_MAX_LEVELS];        // per committed level (in list order): the sector's implicit rectangle
  uint32_t off_end[LK_MAX_LEVELS]; //   and off[sector + 1] (= off[sector]: a rectangle has no list entries)
  int4 *d_rect[LK_MAX_LEVELS];
  uint32_t *d_off[LK_MAX_LEVELS];
};

struct LkEvalArgs { // stand-alone evaluation (known-answer tests)
  const LkLevelView *lv;
  const float2 *center;
  int sector, level;
  float p[6];
  float *out; // [36 + 6 + 1 + 1]: A row-major 6x6 (upper), b, chi, error
  int ref_threads; // > 0: the reference's summation order for number_of_threads = ref_threads (evaluate_ordered)
};
