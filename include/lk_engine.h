/* lk_engine.h - C-ABI of the MI355X (gfx950) Lucas-Kanade correlation engine.
 *
 * Drop-in boundary for the hot path of namascar/correlation: everything the reference's
 * managerClass asks of its GPU engine `CudaClass` (cuda_class.cuh:46-79) and of its CPU
 * engine `CorrelationClass` (correlation_class.hpp:132-179) for ONE image pair:
 * image pyramids, ROI -> sample lists, and the per-sector coarse-to-fine
 * Levenberg-Marquardt solve.  Plain C types only; no OpenCV / Qt / torch types.
 *
 * Conventions
 *  - every function returns an `lk_error` (the reference's errorEnum values,
 *    enums.hpp:25-35); the engine never calls exit() (the reference does on CUDA errors,
 *    cuda_class.cu:52-56).  lk_last_error_string() describes the last failure.
 *  - the caller owns every buffer it passes in or receives results in; the engine copies
 *    during the call and never returns pointers into its own memory (the reference
 *    returns a pointer to engine-owned pinned memory, cuda_polygon.cuh:339-340).
 *  - results follow the CPU engine's semantics (SURVEY.md section 8a-a13 lists where the
 *    reference's CUDA path differs): look-ahead parameters are returned, `iterations` is
 *    the last pyramid level's trip count, chi is the 1/n-scaled last_good_chi.
 *  - one caller thread at a time, except lk_set_image(LK_IMG_NXT) which may overlap a
 *    running lk_correlate_* (manager_class.cpp:1438-1447 prefetches the next frame).
 */
#ifndef LK_ENGINE_H
#define LK_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LK_MAX_LEVELS 8
#define LK_MAX_PARAMS 6

/* errorEnum, enums.hpp:25-35 (same numeric values) */
typedef enum {
  LK_ERROR_NONE = 0,
  LK_ERROR_MODEL_OUT_OF_IMAGE = 1,
  LK_ERROR_INTERPOLATION_OUT_OF_IMAGE = 2,
  LK_ERROR_CORRELATION_MAX_ITERS_REACHED = 3,
  LK_ERROR_BAD_DOMAIN = 4,
  LK_ERROR_SOLVER = 5, /* error_cuSolver */
  LK_ERROR_DEVICE = 6, /* error_cuda: any HIP failure */
  LK_ERROR_MULTITHREAD = 7
} lk_error;

/* interpolationModelEnum, enums.hpp:10-15 */
typedef enum {
  LK_IM_NEAREST = 0,
  LK_IM_BILINEAR = 1,
  LK_IM_BICUBIC = 2,
  /* Extension (not a value of the reference's interpolationModelEnum): the same bicubic
   * surface - the reference's 16-coefficient patch is the Catmull-Rom spline in exact
   * arithmetic - evaluated in separable form, 4 + 4 weights and their derivatives.  It is
   * closer to the exact spline than the reference's monomial evaluation (whose cancellation
   * costs ~1e-3 grey levels), so it agrees with LK_IM_BICUBIC to that rounding, not bit for bit,
   * and whole solves agree to ~1e-5 relative chi rather than inside the reference's own noise.
   * About 40 % fewer instructions per sample.  Never selected implicitly. */
  LK_IM_BICUBIC_SEPARABLE = 3
} lk_interpolation;
/* fittingModelEnum, enums.hpp:17-23: p = (u), (u,v), (u,v,q), (u,v,ux,uy,vx,vy) */
typedef enum { LK_FM_U = 0, LK_FM_UV = 1, LK_FM_UVQ = 2, LK_FM_UVUXUYVXVY = 3 } lk_fitting_model;
/* ImageType, enums.hpp:94-99 */
typedef enum { LK_IMG_UND = 0, LK_IMG_DEF = 1, LK_IMG_NXT = 2 } lk_image_slot;

/* CorrelationClass ctor arguments (correlation_class.hpp:132-137) +
 * CudaClass::set_* (cuda_class.cu:77-102) + resetImagePyramids' start/step/stop (:475) */
typedef struct {
  int interpolation;   /* lk_interpolation, default im_bicubic (mainapp.cpp:64) */
  int fitting_model;   /* lk_fitting_model */
  float precision;     /* required_precision, default 1e-3 */
  int max_iters;       /* maximum_iterations, default 50 */
  int py_start, py_step, py_stop; /* pyramid levels, default 0/1/2 */
  int device;          /* HIP device ordinal this engine lives on */
} lk_config;

/* layout-identical to CorrelationResult (domains.hpp:110-118), 48 bytes */
typedef struct {
  float resultingParameters[6];
  float chi;
  int numberOfPoints;
  int iterations;
  int errorCode; /* lk_error */
  float undCenterX;
  float undCenterY;
} lk_result;

/* counters of the last lk_correlate_* call, for the bench */
typedef struct {
  uint64_t sectors;
  uint64_t evaluations;       /* warp+sample+accumulate passes over one sector */
  uint64_t sample_evaluations;/* sum over evaluations of the sector's n_L */
  uint64_t point_iterations;  /* per sector and level: LM trips + 1 (evaluation #0) */
  uint64_t algorithmic_bytes; /* 25*sample_evaluations + 196*evaluations */
  uint64_t ill_conditioned_solves; /* damped solves (outside starved levels) that met a bad pivot */
  float solve_ms;             /* HIP-event time of the solve kernel(s) of the last call */
  float pyramid_ms;           /* HIP-event time of the last pyramid build */
} lk_stats;

typedef struct lk_engine lk_engine;

/* ---- life cycle ------------------------------------------------------------------- */
/* CudaClass::initialize (cuda_class.cu:39-75): number of usable devices */
int lk_device_count(void);
/* CorrelationClass ctor / CudaClass ctor + set_* */
int lk_create(const lk_config *cfg, lk_engine **out);
void lk_destroy(lk_engine *e);
const char *lk_last_error_string(const lk_engine *e);
/* run all engine work on a caller-provided hipStream_t (NULL = engine's own stream) */
int lk_set_stream(lk_engine *e, void *hip_stream);
/* HIP events around pyramid builds and solves (the DEBUG_TIME_* prints of defines.hpp:29-72):
 * on by default; off removes four event records per frame from the stream and leaves
 * lk_stats.solve_ms / pyramid_ms at their last values */
int lk_set_timing(lk_engine *e, int enabled);
/* A sector's record is always deterministic for a given batch.  By default it may differ in
 * the last bits between batches of different composition (a half-wavefront that runs out of
 * work joins its neighbour's sector, which changes the summation grouping - the same kind of
 * difference the reference shows between thread counts, correlation_class.cpp:169-186,253-275).
 * enabled = 1 switches that off (and fixes the lane group by the sector's own size instead of
 * the batch's): a sector then gets the same bits in any batch, shard or single-sector call
 * (about 20 % slower on small grids).  Call it before lk_commit_sectors. */
int lk_set_batch_invariant(lk_engine *e, int enabled);
/* Reference-order mode.  threads = T > 0: every evaluation of every sector at every pyramid level
 * adds its A, b and chi in the CPU engine's own order for number_of_threads = T - rounded
 * product, then rounded add, sample by sample (x outer / y inner for rectangles,
 * interpolation_class.cpp:722-749), in T contiguous chunks joined in thread order
 * (correlation_class.cpp:169-186, :253-275) - and every damped system goes through the restated
 * ColPivHouseholderQR (correlation_class.cpp:742-747).  The 48-byte records are then bit-identical
 * to CorrelationClass::Newton_Raphson's (as restated by the repository's CPU checker) for that
 * thread count (T = 1: one running sum; the reference's compile-time default is 20, defines.hpp).  The records
 * are also independent of batch composition.  The sample work stays parallel (a 16-lane row or a
 * wavefront per sector forms the per-sample products and transposes them through LDS; only the
 * additions of each sum run as a chain; the lanes of a wavefront are dealt to its four sectors by need), so the
 * mode costs 1.1-1.9x the default's time (measured, round 3: config 2 0.46 against 0.24 ms, config 4 2.1 against
 * 1.9 ms, config 5 8.6 against 5.4 ms), not the serial loop's.  threads = 0 (default): lane-parallel sums and the root-free Cholesky solve,
 * which differ from the reference by summation order and solver rounding only.
 * May be called at any time; the lane groups are re-chosen before the next solve. */
int lk_set_reference_order(lk_engine *e, int threads);
/* Independent image pairs can be solved side by side: one engine per pair in flight, each on
 * its own stream (lk_set_stream).  A solve ends in a tail of slow sectors that leaves most of
 * the GPU idle; the next pair's solve fills it (C2: 0.26 ms per pair one at a time, 0.15 ms
 * with three in flight).  n tells the engine how many launches share the GPU so that it keeps
 * the narrow, better-packed lane groups it would otherwise widen to shorten a lone solve
 * (default 1), and it bounds the width of the multi-workgroup teams that solve giant sectors:
 * team workgroups wait for each other, so engines whose team launches can be on the GPU at the
 * same time MUST declare it.  Call it before lk_commit_sectors.  Tracked sequences cannot use this: the
 * guess of pair k+1 needs the result of pair k (manager_class.cpp:2602-2707). */
int lk_set_pairs_in_flight(lk_engine *e, int n);
/* block until everything queued by this engine has finished */
int lk_synchronize(lk_engine *e);

/* ---- images and pyramids ---------------------------------------------------------- */
/* CorrelationClass::set_undeformed/deformed/next_image (correlation_class.cpp:38-51),
 * CudaClass::resetImagePyramids / resetNextPyramid (cuda_class.cu:475-559): upload the
 * level-0 pixels (monochrome u8, `step` bytes per row) and build levels 1..py_stop.
 * File decoding stays with the caller. */
int lk_set_image(lk_engine *e, int slot, const uint8_t *host_pixels, int rows, int cols, int step);
/* Frames in PINNED host memory ("pinned staging + hipMemcpyAsync on a dedicated stream"; the reference uploads
 * from pageable cv::Mat pixels, cuda_pyramid.cu:83-103).  lk_pin_host_memory page-locks a caller's frame buffer
 * (hipHostRegister; hipHostMalloc'ed memory needs nothing).  lk_set_image / lk_sequence_set_frame from pinned memory
 * return as soon as the copy is ENQUEUED - with LK_IMG_NXT or a ring slot on the next-frame stream, i.e. beside a running
 * solve, without a helper thread - instead of waiting for it as they do for pageable memory.  The buffer must stay
 * unchanged until the frame has been used: until the next lk_correlate_* that reads it has returned, or lk_synchronize. */
int lk_pin_host_memory(void *ptr, size_t bytes);
int lk_unpin_host_memory(void *ptr);
/* same, pixels already in this device's memory (used after an RCCL broadcast) */
int lk_set_image_device(lk_engine *e, int slot, const void *device_pixels, int rows, int cols, int step);
/* CudaClass::resetImagePyramids(und, def, ...) (cuda_class.cu:475-519): both frames of a pair,
 * already in this device's memory, uploaded and reduced in one launch */
int lk_set_image_pair_device(lk_engine *e, const void *und_pixels, int und_step, const void *def_pixels,
                             int def_step, int rows, int cols);
/* CorrelationClass::set_und_image_from_def / set_def_image_from_nxt
 * (correlation_class.cpp:53-61), CudaClass::makeUndPyramidFromDef / makeDefPyramidFromNxt
 * (cuda_class.cu:561-567): pointer rotation, no copies */
int lk_rotate_und_from_def(lk_engine *e);
int lk_rotate_def_from_nxt(lk_engine *e);
/* download one pyramid level (cudaImage::testPyramid, cuda_pyramid.cu:304-346);
 * host_out may be NULL to query the size */
int lk_get_pyramid_level(lk_engine *e, int slot, int level, uint8_t *host_out, int *rows, int *cols);

/* ---- sectors (ROI -> sample lists) ------------------------------------------------ */
/* forget all sectors */
int lk_clear_sectors(lk_engine *e);
/* CudaClass::resetPolygon(iSector,x0,y0,x1,y1) (cuda_class.cu:574-582) with the CPU
 * path's sample order (x outer, y inner, inclusive; manager_class.cpp:1596-1614) and
 * centre ((x0+x1)/2,(y0+y1)/2) = the integer centre the manager passes (:438-441) */
int lk_set_sector_rect(lk_engine *e, int sector, int x0, int y0, int x1, int y1);
/* the whole rectangular prologue (manager_class.cpp:276-310): hs*vs sectors,
 * iSector = i*vs + j; registers sectors [first, first+count) of that grid as engine
 * sectors 0..count-1 (count<0: all) - the multi-GPU shard entry point */
int lk_set_rect_grid(lk_engine *e, float x_begin, float y_begin, float x_end, float y_end,
                     int hs, int vs, int first, int count);
/* CudaClass::resetPolygon(iSector,r,dr,a,da,cx,cy,as) (cuda_class.cu:584-594) with the
 * CPU path's predicate and order (manager_class.cpp:816-940); centre = float mean */
int lk_set_sector_annular(lk_engine *e, int sector, float r, float dr, float a, float da,
                          float cx, float cy, int as);
/* the sectors first_sector .. first_sector+count-1 of an annular domain in one call - the sector
 * loop of perform_single_frame_correlation_annular (manager_class.cpp:600-720) - rasterised by a
 * few host threads, one sector each at a time (every list is the sequential scan's).
 * params [count][6] = {r, dr, a, da, cx, cy} per sector. */
int lk_set_sectors_annular(lk_engine *e, int first_sector, int count, const float *params, int as);
/* CudaClass::resetPolygon(v_points) (cuda_class.cu:596-605) with polygonBlob_class
 * semantics (polygon_class.cpp:224-429); LK_ERROR_BAD_DOMAIN on a self-intersecting
 * contour (manager_class.cpp:1028-1031) */
int lk_set_sector_blob(lk_engine *e, int sector, const float *contour_xy, int n_vertices);
/* CorrelationClass::Newton_Raphson(p, n, xy) / (p, n, cx, cy, xy)
 * (correlation_class.cpp:306-343): explicit AoS sample list; use_center=0 -> float mean */
int lk_set_sector_points(lk_engine *e, int sector, const float *xy, int n, int use_center,
                         float cx, float cy);
/* build per-level sample lists (pyramid_class.cpp:289-362) and upload; must be called
 * after the lk_set_sector_* calls and before lk_correlate_*.  Annular and blob sectors are
 * registered by their description and rasterised HERE (on the device): a sector that turns
 * out to be empty is reported by this call (LK_ERROR_BAD_DOMAIN, the message names the
 * sector), not by lk_set_sector_annular / _blob - unless LK_HOST_ROI=1, where the host scan
 * runs at registration like the reference's (manager_class.cpp:1028-1031).  Sectors registered since the
 * previous commit start with zeroed sequence state (guess history, last record); every other
 * sector keeps its state, so the reference's first-frame loop - resetPolygon(i), correlate(i),
 * sector after sector (manager_class.cpp:340, :449) - can commit once per sector and still
 * move every sector from its own record on the next frame (lk_update_sector). */
int lk_commit_sectors(lk_engine *e);
/* Domain tracking between the frames of a sequence, CPU-engine semantics (the CUDA engine's
 * cudaPolygon::updatePolygon, cuda_polygon.cu:268-415, moves by its own lastGood
 * parameters instead).  Both keep the per-sector sequence state (guess history).
 *  - Lagrangian (manager_class.cpp:381-419): sample (x,y) -> ((int)(ox+x+0.5f), (int)(oy+y+0.5f))
 *    with the sector's offset (ox,oy) = new und centre - past und centre (add_pair, :38-47);
 *  - strict Lagrangian (manager_class.cpp:369-380): the deformed sample positions of the last
 *    solve (CorrelationClass::getDefXY0, correlation_class.cpp:884-896: warped with the
 *    parameters of the last level-0 evaluation) become the undeformed samples.
 * offsets_xy [S][2]; centers_xy [S][2] = centres for the next solve (the rectangular path
 * passes integers, manager_class.cpp:438-441) or NULL = float mean of the new samples.
 * Both rebuild the lists of every pyramid level on the device (the samples never visit the
 * host), except that lk_translate_sectors keeps grids of implicit rectangles on the host
 * records: a rectangle that moves by whole pixels stays implicit. */
int lk_translate_sectors(lk_engine *e, const float *offsets_xy, const float *centers_xy);
int lk_rewarp_sectors(lk_engine *e, const float *centers_xy);
/* one sector, from the engine's own last record of it - the call shape of
 * CudaClass::updatePolygon(iSector, deformationDescription) (cuda_class.cu:569) with the CPU
 * manager's meaning; mode = deformationDescriptionEnum (0 strict Lagrangian, 1 Lagrangian,
 * 2 Eulerian = nothing).  The lists are rebuilt before the next solve. */
int lk_update_sector(lk_engine *e, int sector, int mode);
/* undo the last lk_translate_sectors / lk_rewarp_sectors for sectors >= first_sector: the
 * sectors a frame never reached because it stopped at an error (manager_class.cpp:520-546) */
int lk_restore_sectors(lk_engine *e, int first_sector);
/* [S][6]: the parameters the last evaluation of each sector ran at (level-0 scale) */
int lk_get_last_evaluated_parameters(lk_engine *e, float *out);
int lk_sector_count(const lk_engine *e);
/* CorrelationClass::get_number_of_points / get_und_x/y_center (correlation_class.cpp:850-868) */
int lk_get_sector_info(lk_engine *e, int sector, int *n_points, float *cx, float *cy);
/* number of samples of a sector at a pyramid level (pyramid_class.cpp:437-439) */
int lk_get_sector_level_count(lk_engine *e, int sector, int level, int *n);
/* CudaClass::getUndXY0ToCPU / CorrelationClass::getUndXY0 (cuda_class.cu:607-609):
 * returns the count; copies min(count, cap) AoS pairs */
int lk_get_und_xy(lk_engine *e, int sector, float *xy, int cap, int *count);
/* Diagnostics: the explicit sample list of a sector at a pyramid level as the device holds it (pyramid_class.cpp:289-323:
 * the decimated lists).  which = 0: the reference's order - what the centres, the starved levels and the reference-order
 * mode walk.  which = 1: the row-major evaluation copy the lane groups of the default mode walk (annular sectors rasterised
 * by the device masks, lists from the host in the reference's x outer / y inner order, moved lists; the same samples row by
 * row, so that neighbouring lanes read neighbouring pixels) -
 * LK_ERROR_BAD_DOMAIN when the domain has none.  Implicit rectangles have no list (count 0). */
int lk_get_level_xy(lk_engine *e, int level, int which, int sector, float *xy, int cap, int *count);
/* CudaClass::getDefXY0ToCPU / CorrelationClass::getDefXY0 (cuda_class.cu:611-613,
 * kModel_inPlace correlationKernel.cu:56-110): level-0 samples warped by p */
int lk_get_def_xy(lk_engine *e, int sector, const float *p, float *xy, int cap, int *count);

/* ---- the solve -------------------------------------------------------------------- */
/* CudaClass::correlate(iSector, guess, results) (cuda_class.cu:104-293) /
 * CorrelationClass::Newton_Raphson: one sector; guess is in/out like the reference
 * (cuda_class.cu:289-290, correlation_class.cpp:360) */
int lk_correlate(lk_engine *e, int sector, float *guess_inout, lk_result *out);
/* all sectors in one device-resident batch (one launch per size class).
 * guesses: [S][6] host floats (unused slots ignored); out: [S] host records */
int lk_correlate_all(lk_engine *e, const float *guesses, lk_result *out);
/* same with device buffers, asynchronous on the engine's stream; d_guesses may be NULL
 * to use the engine-held guesses written by lk_adjust_initial_guess; d_results may be NULL to
 * leave the records in the engine's own record buffer (lk_get_results_device).  Note that
 * lk_update_sector moves a sector by the engine's OWN record of it: a caller that solves into a
 * buffer of its own and then calls lk_update_sector must copy the records back (lk_group does). */
int lk_correlate_all_device(lk_engine *e, const void *d_guesses, void *d_results);
/* the engine's own record buffer, [S] lk_result in device memory (valid until the next commit) */
int lk_get_results_device(lk_engine *e, const void **d_records);
/* lk_correlate_all with the engine-held guesses (lk_adjust_initial_guess) that does not wait: the
 * records follow the solve into an engine-owned pinned buffer; lk_wait_results blocks until they
 * are there and copies them to out [S].  One solve may be outstanding.  A frame loop uses the
 * pair to keep the host's per-frame bookkeeping off the GPU's critical path (lk_sequence_run). */
int lk_correlate_all_async(lk_engine *e);
int lk_wait_results(lk_engine *e, lk_result *out);

/* ---- frame-pipelined windows of a sequence ----------------------------------------- */
/* perform_multiframe_correlation's frame loop (manager_class.cpp:1380-1496) for the Eulerian description: the
 * sectors stay where they are, every frame brings a new deformed image, and the guess of frame f + 1 of a sector is
 * a function of that SECTOR's own earlier results (adjust_initial_guess, :2677-2699: p(f), or 2 p(f) - p(f-1) with the
 * first image as the reference) - it waits for no other sector.  One launch per pair pays the slow tail of every
 * pair: a launch lasts as long as its slowest sector.  Here K deformed frames are resident at once (a ring of
 * pyramids, filled on the next-frame stream like LK_IMG_NXT, :1438-1447) and ONE launch per size class solves the
 * whole window: its work items are (frame, sector) pairs drawn frame-major from a device-wide queue, and a sector's
 * parameters travel from frame to frame through a per-sector chain in device memory; the group that draws (f, s)
 * before (f - 1, s) is done waits for it without blocking the other sectors of its wavefront.
 * Records: the arithmetic of a (frame, sector) is the one-pair kernels' - in batch-invariant and in reference-order
 * mode the window's records are byte-identical to solving the pairs one after the other with
 * lk_adjust_initial_guess + lk_correlate_all*.  The default mode uses the fixed lane groups and the fast flavour
 * inside a window (a window in which a damped system meets a bad pivot is solved again with the SAFE flavour).
 * Domains with sectors of more than 8192 samples (workgroup-wide groups, teams), and classes of more than 128 samples per
 * sector that have a starved pyramid level (config 5's geometry: their one-pair launch chain is the faster form), have
 * no pipelined instance: their windows run the frames one after the other on the device - same interface, same records. */
/* a ring of n_slots resident deformed-frame pyramids (grows; never shrinks) */
int lk_sequence_reserve(lk_engine *e, int n_slots);
/* upload + pyramid of one frame into ring slot `slot`, on the next-frame stream: may overlap a running
 * lk_correlate_* / window (the one concurrent call of section 8b's threading contract); a slot that a window
 * still reads is overwritten only after that window (stream order on the device) */
int lk_sequence_set_frame(lk_engine *e, int slot, const uint8_t *host_pixels, int rows, int cols, int step);
int lk_sequence_set_frame_device(lk_engine *e, int slot, const void *device_pixels, int rows, int cols, int step);
/* solve frames i = 0 .. n_frames-1 of a window: deformed image = ring slot (first_slot + i) % n_slots; undeformed
 * image = LK_IMG_UND (und_slot < 0) or ring slot und_slot for frame 0 and, with reference_previous, the previous
 * frame's deformed image for the others (image roles of manager_class.cpp:1386-1407).  Frame 0 starts from the
 * engine-held guesses (lk_adjust_initial_guess for that frame of the sequence, or whatever the caller put there),
 * every later frame from the guess rule above; the sequence state the next lk_adjust_initial_guess reads is left as
 * n_frames one-pair solves would leave it.  flags: 1 = copy the records to the host (lk_wait_sequence's out),
 * 2 = keep the guesses every frame started from (lk_get_sequence_results_device).  Does not wait. */
int lk_correlate_sequence_async(lk_engine *e, int und_slot, int first_slot, int n_frames, int reference_previous,
                                int constant_velocity, int flags);
/* block until the window is solved; out: [n_frames][S] records (frame-major) or NULL.  LK_ERROR_DEVICE if the
 * window is void (a bounded wait inside the kernel gave up: cannot happen unless the device loses wavefronts). */
int lk_wait_sequence(lk_engine *e, lk_result *out);
/* (flags & 1) the waited-for window's records [n_frames][S] in the engine's own pinned host memory - no copy; two buffers
 * alternate, so the pointer stays valid while the NEXT window is launched and solved (until the launch after that): a
 * frame loop digests window w while window w + 1 runs (lk_sequence_run) */
int lk_sequence_host_records(lk_engine *e, const lk_result **records);
/* page-locks, ahead of time, the record buffer the NEXT lk_correlate_sequence_async(..., flags & 1) of up to n_frames will
 * use (locking the 38 MB of a 16-pair window of 50 176 sectors takes 7 ms; the launch does it itself otherwise).  May be
 * called from another thread than the one that launches and waits, between a launch and the next one - a frame loop does
 * it beside its uploads and beside the running window (lk_sequence_run). */
int lk_sequence_prepare_host_records(lk_engine *e, int n_frames);
/* the window's records [n_frames][S] and (flags & 2) guesses [n_frames][S][6] in device memory */
int lk_get_sequence_results_device(lk_engine *e, const void **d_records, const void **d_guesses);
/* the last window's records into a caller's device buffer, frame f at d_dst + f * dst_pitch_records records
 * (dst_pitch_records >= S; the padded blocks of an all-gather), asynchronously on the engine's stream */
int lk_copy_sequence_records_device(lk_engine *e, void *d_dst, size_t dst_pitch_records);
/* 1: the last window ran on the frame-pipelined instances, 0: frame after frame */
int lk_sequence_is_pipelined(lk_engine *e);
/* (flags & 2) the guesses every frame of the last window started from, [n_frames][S][6], to the host */
int lk_get_sequence_guesses(lk_engine *e, float *guesses);

/* managerClass::adjust_initial_guess (manager_class.cpp:2602-2707), batched on the
 * device for every sector: frame 0 -> global guess + strain*(sector centre - global
 * centre); later frames -> constant_velocity ? 2*p_prev - p_prevprev : p_prev, where
 * p_prev are the results of the previous lk_correlate_all*. */
int lk_adjust_initial_guess(lk_engine *e, int frame, int constant_velocity,
                            const float *global_guess, float global_cx, float global_cy);
/* copy the engine-held guesses ([S][6]) to the host */
int lk_get_guesses(lk_engine *e, float *guesses);

/* ---- stand-alone pieces (known-answer tests, same kernels as the batch path) ------- */
/* one evaluation of one sector at one level: raw sums A (6x6 row-major, upper valid),
 * b, chi (unscaled), error flag (apply_model_and_interpolate, correlation_class.cpp:131) */
int lk_evaluate(lk_engine *e, int sector, int level, const float *p, float *A36, float *b6,
                float *chi, int *error);
/* value and gradient of image `slot` at pyramid `level` at n arbitrary points
 * (InterpolationClass::get_interpolation, interpolation_class.cpp:79-226), with the
 * engine's interpolation model: out4[k] = {W, dW/dx, dW/dy, out_of_image ? 1 : 0} */
int lk_sample(lk_engine *e, int slot, int level, const float *xy, int n, float *out4);
/* compute_model_parameters + solve (correlation_class.cpp:642-768) on the device.
 * reference_solver = 0: the engine's normal choice (root-free Cholesky, falling back to the
 * reference's pivoted QR when a pivot is small); 1: the pivoted QR always, as the solve
 * kernel does on starved pyramid levels (at most 2P samples); 2: the same QR spread over a
 * 16-lane row, as the finisher of parked sectors runs it (bit-identical to 1) */
int lk_damped_solve(lk_engine *e, int n, const float *A_rowmajor_upper, const float *b,
                    float lambda, float scaling, int reference_solver, float *dp);

int lk_get_stats(lk_engine *e, lk_stats *out);
/* the same counters per sector of the last solve, registration order:
 * out[s] = {evaluations, sample evaluations, point iterations, ill-conditioned solves} */
int lk_get_sector_stats(lk_engine *e, uint32_t *out_4_per_sector);

#ifdef __cplusplus
}
#endif
#endif /* LK_ENGINE_H */
