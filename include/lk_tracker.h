/* lk_tracker.h - C ABI of the sequence / tracking bookkeeping around the Lucas-Kanade engine.
 *
 * what managerClass does around the per-sector solve for a
 * sequence of frames - where each sector sits on each frame (Eulerian / Lagrangian / strict
 * Lagrangian descriptions), which initial guess it starts from, how the engine's 48-byte
 * records become frame_results and the CSV report - plus a frame loop that drives
 * include/lk_engine.h with the next frame uploaded behind the running solve.
 *
 *   lk_tracker  = the manager's bookkeeping, host only, no GPU inside:
 *                 perform_single_frame_correlation_{rectangular,annular,blob}
 *                 (manager_class.cpp:274-551, :553-813, :1000-1237) without the solve,
 *                 adjust_{rectangular,annular,blob}_domain (:2018-2310), adjust_initial_guess
 *                 (:2602-2707), update_results (:2312-2428), update_global_results
 *                 (:2709-2753), initializeReport / addFrameToReport (:2430-2525).
 *   lk_sequence_run = perform_multiframe_correlation (:1296-1496) on an lk_engine:
 *                 image roles (first / previous reference, und<-def<-nxt rotation), the
 *                 asynchronous load of frame k+2 during pair k (:1438-1447), stop policy.
 *
 * Enumerations carry the reference's values (enums.hpp): deformationDescriptionEnum
 * {strict_Lagrangian 0, Lagrangian 1, Eulerian 2}, errorHandlingModeEnum {stopAll 0,
 * stopFrame 1, continue 2}, referenceImageEnum {First 0, Previous 1}, domainEnum
 * {rectangular 0, annular 1, blob 2}.  GUI-only state (contours, overlay lists) is not kept.
 */
#ifndef LK_TRACKER_H
#define LK_TRACKER_H

#include <stddef.h>
#include <stdint.h>

#include "lk_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { LK_DEF_STRICT_LAGRANGIAN = 0, LK_DEF_LAGRANGIAN = 1, LK_DEF_EULERIAN = 2 };
enum { LK_ERRMODE_STOP_ALL = 0, LK_ERRMODE_STOP_FRAME = 1, LK_ERRMODE_CONTINUE = 2 };
enum { LK_REF_FIRST = 0, LK_REF_PREVIOUS = 1 };
enum { LK_DOMAIN_RECT = 0, LK_DOMAIN_ANNULAR = 1, LK_DOMAIN_BLOB = 2 };

typedef struct {
  int fitting_model;   /* fittingModelEnum */
  int domain_type;     /* domainEnum */
  int deformation;     /* deformationDescriptionEnum */
  int reference_image; /* referenceImageEnum */
  int error_mode;      /* errorHandlingModeEnum */
  float global_guess[6];
} lk_tracker_config;

/* the scalar part of frame_results (domains.hpp:59-108) */
typedef struct {
  float und_center_x, und_center_y, und_angle, und_e;
  float und_global_ro, und_global_ri, und_global_angle, und_global_center_x, und_global_center_y, und_global_e;
  float def_center_x, def_center_y, def_angle, def_e;
  float def_global_ro, def_global_ri, def_global_angle, def_global_center_x, def_global_center_y, def_global_e;
  float resulting_parameters[6], previous_resulting_parameters[6], initial_guess[6];
  int number_of_points;
  float chi;
  int iterations;
  int error_status;
  int error_code;
  float past_und_center_x, past_und_center_y;
} lk_frame_result;

/* what has to happen to a sector's sample list before the next solve */
enum {
  LK_SECTOR_KEEP = 0,      /* Eulerian, later frames */
  LK_SECTOR_RECT = 1,      /* frame 0: resetPolygon(iSector, x0, y0, x1, y1) */
  LK_SECTOR_ANNULAR = 2,   /* frame 0: resetPolygon(iSector, r, dr, a, da, cx, cy, as) */
  LK_SECTOR_BLOB = 3,      /* frame 0: resetPolygon(contour), see lk_tracker_blob_contour */
  LK_SECTOR_TRANSLATE = 4, /* Lagrangian: add_pair(offset) (manager_class.cpp:38-47, :381-419) */
  LK_SECTOR_REWARP = 5     /* strict Lagrangian: und <- def samples (:369-380) */
};
typedef struct {
  int kind;
  int use_center; /* 1: the solve is about (center_x, center_y) (rectangular path, :438-441) */
  float center_x, center_y;
  int x0, y0, x1, y1;        /* LK_SECTOR_RECT */
  float r, dr, a, da, cx, cy; /* LK_SECTOR_ANNULAR */
  int as;
  float offset_x, offset_y;  /* LK_SECTOR_TRANSLATE */
} lk_sector_command;

typedef struct lk_tracker lk_tracker;

int lk_tracker_create(const lk_tracker_config *cfg, lk_tracker **out);
void lk_tracker_destroy(lk_tracker *t);
const char *lk_tracker_last_error(const lk_tracker *t);
/* rectangularDomainStruct / annularDomainStruct / blobDomainStruct (domains.hpp:19-57) */
int lk_tracker_set_rect_domain(lk_tracker *t, float x_begin, float y_begin, float x_end, float y_end,
                               float x_center, float y_center, int hs, int vs);
int lk_tracker_set_annular_domain(lk_tracker *t, float r_inside, float r_outside, float x_center,
                                  float y_center, int rs, int as);
int lk_tracker_set_blob_domain(lk_tracker *t, const float *contour_xy, int n_vertices, float x_center,
                               float y_center);
int lk_tracker_sector_count(const lk_tracker *t);
/* the CSV rows cost ~35 number conversions per sector and frame: callers that only read
 * lk_tracker_get_results can switch them off (on by default) */
int lk_tracker_enable_report(lk_tracker *t, int enabled);
int lk_tracker_blob_contour(const lk_tracker *t, const float **contour_xy, int *n_vertices);

/* Top half of the sector loop of one frame: adjust_*_domain + adjust_initial_guess for every
 * sector.  commands [S], guesses [S][6]. */
int lk_tracker_begin_frame(lk_tracker *t, int frame, lk_sector_command *commands, float *guesses);
/* Bottom half: update_results per sector in loop order (stopping at the first error under
 * stopAll / stopFrame: later sectors keep the state they had before lk_tracker_begin_frame,
 * *first_unsolved receives the first such sector or S), update_global_results, and one report
 * row per sector.  *stop_sequence = 1 when the frame loop has to end (stopAll + error). */
int lk_tracker_end_frame(lk_tracker *t, int frame, const char *und_name, const char *def_name,
                         const lk_result *results, int *first_unsolved, int *stop_sequence);
int lk_tracker_get_results(const lk_tracker *t, lk_frame_result *out /* [S] */);
/* the report so far (initializeReport + addFrameToReport text); *needed = bytes incl. NUL */
int lk_tracker_report(const lk_tracker *t, char *buf, size_t cap, size_t *needed);

/* ---- frame loop on an engine ------------------------------------------------------------ */
/* Applies one frame's commands to the engine (frame 0: registers and commits the sectors;
 * later: lk_translate_sectors / lk_rewarp_sectors), solves every sector with the tracker's
 * guesses and feeds the records back.  Images are the caller's business here. */
int lk_sequence_frame(lk_engine *e, lk_tracker *t, int frame, const char *und_name, const char *def_name,
                      int *stop_sequence);

/* image source of lk_sequence_run: returns the level-0 pixels of frame `index` (monochrome u8,
 * *step bytes per row); the buffer of frame i must stay valid until the call for frame i + 2
 * returns (two frames are in flight).
 * Called from a helper thread for the prefetch of frame k + 2 (manager_class.cpp:1438-1447). */
typedef const uint8_t *(*lk_frame_provider)(void *user, int index, int *rows, int *cols, int *step,
                                            const char **name);
/* perform_multiframe_correlation (manager_class.cpp:1296-1496): n_frames images, n_frames - 1
 * pairs.  *pairs_done receives the number of pairs correlated.
 * Eulerian description + rectangular domain + continue policy: pair k+1 is launched from the
 * device-computed guesses (lk_adjust_initial_guess) as soon as the records of pair k are back,
 * and the tracker's bookkeeping of pair k runs behind that solve - same frame_results and report,
 * bit for bit, as the one-pair-at-a-time loop every other configuration uses. */
int lk_sequence_run(lk_engine *e, lk_tracker *t, int n_frames, lk_frame_provider provider, void *user,
                    int *pairs_done);

/* ---- ROI -> sample lists, host only (the code behind lk_set_rect_grid / lk_set_sector_annular /
 * lk_set_sector_blob / lk_commit_sectors; usable without a device, e.g. to size buffers or to
 * draw overlays) ------------------------------------------------------------------------------ */
/* sector geometry of a rectangular domain (manager_class.cpp:276-310): half sizes and hs*vs
 * integer centres in iSector = i*vs + j order */
int lk_roi_rect_grid(float x_begin, float y_begin, float x_end, float y_end, int hs, int vs, int *xdim, int *ydim,
                     int *centers_xy);
/* get_inside_points_annularDomain (manager_class.cpp:816-940); returns the count (may exceed cap) */
int64_t lk_roi_annular_points(float r, float dr, float a, float da, float cx, float cy, int as, float *xy,
                              int64_t cap);
/* polygonBlob_class (polygon_class.cpp:224-429); -1 for a self-intersecting contour */
int64_t lk_roi_blob_points(const float *contour_xy, int n_vertices, float *xy, int64_t cap);
/* per-level decimation (pyramid_class.cpp:289-323); returns the count written */
int lk_roi_decimate(const float *xy, int n, int level_delta, float *out);

/* ---- image ingest ------------------------------------------------------------------------ */
/* binary PGM (P5, maxval <= 255) reader for headless runs; *pixels is malloc'ed (free with
 * lk_free_image) */
int lk_load_pgm(const char *path, uint8_t **pixels, int *rows, int *cols);
void lk_free_image(uint8_t *pixels);
/* What cv::imread(path, cv::IMREAD_GRAYSCALE) hands the reference (manager_class.cpp:102-107,174,211,250;
 * cuda_class.cu:484-510; the reference's sample frames are PNG, mainapp.cpp:384-408): the file decoded to
 * 8-bit grey, rows x cols, row-major, malloc'ed (lk_free_image).  PNG (all colour types and depths, Adam7),
 * TIFF (strips or tiles, either byte order, 8 / 16 bits, uncompressed / PackBits / LZW / Deflate, horizontal
 * differencing; first page), uncompressed BMP, PNM P1-P6; 16-bit samples give their high byte, colour is converted with the fixed-point
 * coefficients of the decoder OpenCV uses for that container (csrc/lk_image_io.cpp has the formulas and what
 * is pinned).  LK_ERROR_BAD_DOMAIN for a missing, malformed, truncated or unsupported (JPEG, BigTIFF) file;
 * *pixels is NULL then.  Thread-safe; lk_decode_image is the same for a file already in memory. */
int lk_load_image(const char *path, uint8_t **pixels, int *rows, int *cols);
int lk_decode_image(const uint8_t *bytes, size_t n_bytes, uint8_t **pixels, int *rows, int *cols);

#ifdef __cplusplus
}
#endif
#endif
