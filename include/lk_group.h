/* lk_group.h - C-ABI of the single-process multi-GPU engine: one lk_engine per MI355X of a node,
 * sectors sharded over them, frames broadcast and records gathered over RCCL / xGMI.
 *
 * What it replaces in the reference: nothing that works - CudaClass carries only remnants of a
 * multi-GPU design (`deviceCount` plumbing cuda_class.cu:77-83, `MAX_GPU_COUNT 32` defines.hpp:11,
 * the commented-out `for iGPU` loops cuda_class.cu:336-337,422-423, the dead
 * k_aggregate_LS_problem_in_GPU0 kernels.cu:42-53 that summed one sector's A|b|chi across GPUs);
 * README.md:33 says "I set the number of GPUs = 1 always".  This is the design SURVEY.md
 * section 8(e) specifies instead, behind the same kind of C boundary as lk_engine.h so that the
 * C++ / Qt caller (HipCudaClass::set_deviceCount) reaches every GPU of the node:
 *
 *  - sectors are independent (own parameters, chi, iteration count; shared read-only images):
 *    rank r owns the contiguous block [r*S/G, (r+1)*S/G) of the sector index - for the rectangular
 *    grid (iSector = i*vs + j, manager_class.cpp:304-308) that is a band of image columns;
 *  - one host thread and one HIP stream per device (a 10 000-sector solve is ~0.25 ms on the GPU:
 *    launching for 8 devices from one thread would cost more than the solve);
 *  - per frame: the level-0 pixels go to device 0 and travel to the others in ONE ncclBroadcast
 *    (4 MiB at 2048^2, 64 MiB at 8192^2); every device builds its own pyramid (cheaper than
 *    broadcasting the 1.33x larger pyramid);
 *  - per solve: no collective inside; the 48-byte records are ncclAllGather-ed in equal padded
 *    blocks and handed to the caller in global sector order;
 *  - transfers run on a communication stream per member, ordered against the solves by events: the
 *    broadcast of the next frame (LK_IMG_NXT, or the next window's ring slots) travels while the current
 *    pair / window is being solved - the reference's prefetch (manager_class.cpp:1438-1447) across GPUs -
 *    and the records of one solve leave while the next one runs (LK_GROUP_FRAMES=copy: the frames by
 *    device-to-device copies from device 0's buffer - the copy engines over the point-to-point xGMI links,
 *    no compute unit needed - instead of ncclBroadcast);
 *  - sequence state (guess history for the constant-velocity guess, manager_class.cpp:2677-2686,
 *    moved sample lists) stays with the engine that owns the sector: a tracked sequence moves
 *    nothing between GPUs but the new frame and the records.
 *
 * Conventions as in lk_engine.h: every function returns an lk_error; the caller owns all buffers;
 * one caller thread at a time.  Records are bit-identical to what ONE engine gives for the same
 * sectors whenever the engine's records do not depend on batch composition
 * (lk_set_batch_invariant / lk_set_reference_order, applied to every member via lk_group_engine) - in
 * reference-order mode including the iteration count a sector reports when its very first evaluation fails
 * (the count the sector BEFORE it left behind, correlation_class.cpp:413-419, :870: resolved over the gathered
 * records in global sector order, with one carry for the whole group).
 * The members' own record buffers hold their shards' records after a solve, so lk_update_sector on a member
 * (HipCudaClass::updatePolygon) moves a sector by the right record.
 * Failures: a member that fails before a collective makes the CALL fail on every member (nobody enters the
 * collective); a collective that fails at enqueue aborts the communicators and breaks the group for good
 * (every later call returns LK_ERROR_DEVICE) - it never hangs the caller.
 */
#ifndef LK_GROUP_H
#define LK_GROUP_H

#include "lk_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lk_group lk_group;

/* n_devices engines on HIP devices devices[0..n-1] (NULL: 0..n-1), cfg->device is ignored.
 * Distinct devices talk over RCCL (ncclCommInitAll).  Listing the same device more than once is
 * allowed for rehearsals on a one-GPU box: those ranks exchange frames and records with
 * device-to-device copies instead (RCCL refuses duplicate devices); everything else - threads,
 * sharding, padding, gather order - is the same code.  (devices == NULL and the environment variable
 * LK_GROUP_DEVICES=0,0,0 set: that list - the rehearsal hook for callers that only pass a count, like
 * HipCudaClass::set_deviceCount.) */
int lk_group_create(const lk_config *cfg, int n_devices, const int *devices, lk_group **out);
void lk_group_destroy(lk_group *g);
const char *lk_group_last_error_string(const lk_group *g);
int lk_group_size(const lk_group *g);
/* ranks of the group's RCCL communicator as RCCL itself reports them (ncclCommCount); 0 in the one-GPU rehearsal
 * transport (the same device listed several times: copies and a host rendezvous, no communicator) */
int lk_group_comm_ranks(const lk_group *g);
/* member engine of a rank, for per-engine settings (lk_set_batch_invariant, lk_set_reference_order,
 * lk_get_stats ...); do not register sectors or images on it directly */
int lk_group_engine(lk_group *g, int rank, lk_engine **e);
/* the block of the global sector index rank owns: [first, first + count) */
int lk_group_shard(const lk_group *g, int rank, int *first, int *count);
/* the partition rule itself (no group, no device needed): rank r of n owns
 * [r*S/n, (r+1)*S/n) of S sectors - contiguous, covering, sizes differing by at most one */
int lk_group_shard_range(int n_sectors, int rank, int n_ranks, int *first, int *count);

/* ---- images: lk_set_image / lk_set_image_device / lk_rotate_* on every member ------------- */
/* level-0 pixels on the host: uploaded to device 0, broadcast, one pyramid build per device */
int lk_group_set_image(lk_group *g, int slot, const uint8_t *host_pixels, int rows, int cols, int step);
/* the same with the pixels already in device 0's memory (pitch = step bytes) */
int lk_group_set_image_device(lk_group *g, int slot, const void *device0_pixels, int rows, int cols, int step);
int lk_group_rotate_und_from_def(lk_group *g);
int lk_group_rotate_def_from_nxt(lk_group *g);

/* ---- sectors: registered by GLOBAL index, dealt to the members at commit ------------------ */
int lk_group_clear_sectors(lk_group *g);
int lk_group_set_sector_rect(lk_group *g, int sector, int x0, int y0, int x1, int y1);
int lk_group_set_rect_grid(lk_group *g, float x_begin, float y_begin, float x_end, float y_end, int hs, int vs);
int lk_group_set_sector_annular(lk_group *g, int sector, float r, float dr, float a, float da, float cx, float cy,
                                int as);
int lk_group_set_sector_points(lk_group *g, int sector, const float *xy, int n, int use_center, float cx, float cy);
int lk_group_commit_sectors(lk_group *g);
int lk_group_sector_count(const lk_group *g);

/* ---- the solve ---------------------------------------------------------------------------- */
/* guesses [S][6] host floats in global sector order, or NULL = the members' engine-held guesses
 * (lk_group_adjust_initial_guess); out [S] host records in global sector order, or NULL = leave
 * them on the devices (every device then holds all S records, see lk_group_records_device) and
 * return without waiting - lk_group_synchronize waits */
int lk_group_correlate_all(lk_group *g, const float *guesses, lk_result *out);
/* managerClass::adjust_initial_guess on every member's own sectors (lk_adjust_initial_guess) */
int lk_group_adjust_initial_guess(lk_group *g, int frame, int constant_velocity, const float *global_guess,
                                  float global_cx, float global_cy);
/* ---- frame-pipelined windows of a sequence (lk_correlate_sequence_async on every member) -- */
/* Every member keeps a ring of n_slots resident deformed frames (lk_sequence_reserve); the frames of a whole window
 * travel from device 0 to the others in ONE transfer on the members' communication streams - while the window before
 * is being solved - and every member solves the window for ITS block of sectors, each sector advancing through the
 * frames on its own (include/lk_engine.h, "frame-pipelined windows").  Per window the group moves n_frames frames out
 * and one all-gather of n_frames x S records back; guess history never leaves the owning engine
 * (manager_class.cpp:1380-1496, :1438-1447, :2677-2699). */
int lk_group_sequence_reserve(lk_group *g, int n_slots);
/* n_frames frames into the ring slots first_slot .. first_slot + n_frames - 1 (no wrap-around: split the call):
 * host frames (one pointer each), or frames back to back in device 0's memory (frame pitch = rows * step bytes).
 * Returns when the transfer is enqueued; ordered against the solves by events, not by the host. */
int lk_group_sequence_set_frames(lk_group *g, int first_slot, int n_frames, const uint8_t *const *host_pixels, int rows, int cols,
                                 int step);
int lk_group_sequence_set_frames_device(lk_group *g, int first_slot, int n_frames, const void *device0_pixels, int rows, int cols,
                                        int step);
/* lk_correlate_sequence_async on every member (frame 0 of the window from the members' engine-held guesses:
 * lk_group_adjust_initial_guess); does not wait */
int lk_group_correlate_sequence_async(lk_group *g, int und_slot, int first_slot, int n_frames, int reference_previous,
                                      int constant_velocity);
/* waits for every member's window, all-gathers the records (blocks of [n_frames][lk_group_block_records()] per
 * member) and hands them over in global sector order: out [n_frames][S] host records, or NULL = leave them on the
 * devices (lk_group_sequence_records_device; lk_group_synchronize waits for the exchange) */
int lk_group_wait_sequence(lk_group *g, lk_result *out);
/* The records of the window lk_group_wait_sequence(g, NULL) last exchanged, out [n_frames][S] in global sector order.
 * Waits for that exchange only: call it AFTER launching the next window (lk_group_correlate_sequence_async) and the
 * records of window w travel - between the devices and down to the host - while window w + 1 is solved.  Must be
 * called before the next lk_group_wait_sequence (which gathers into the same buffers). */
int lk_group_sequence_records(lk_group *g, lk_result *out);
int lk_group_sequence_records_device(lk_group *g, int rank, const void **d_records);
/* what overlapped on `rank`'s device (HIP event times, ms): [0] duration of the last frame transfer on the
 * communication stream, [1] of the last solve / window on the solve stream, [2] transfer begin - solve begin,
 * [3] solve end - transfer end (both positive: the transfer ran inside the solve).  Waits for both. */
int lk_group_probe_overlap(lk_group *g, int rank, float *ms4);

/* device pointer (on `rank`'s device) of the gathered records: lk_group_size() blocks of
 * lk_group_block_records() records each, block r holding rank r's sectors first */
int lk_group_records_device(lk_group *g, int rank, const void **d_records);
int lk_group_block_records(const lk_group *g);
int lk_group_synchronize(lk_group *g);
/* counters of the last solve summed over the members; solve_ms / pyramid_ms = the slowest member */
int lk_group_get_stats(lk_group *g, lk_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* LK_GROUP_H */
