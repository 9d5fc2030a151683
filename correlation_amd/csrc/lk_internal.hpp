// lk_internal.hpp - what lk_group.cpp needs from the engine and the kernels beyond the public C-ABI
// (same shared library; nothing here is part of include/*.h).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lk_engine.h"

// Reference-order mode inside a group: a member engine leaves the "stale iteration count" markers of its
// records unresolved (see lk_stale_iterations_kernel); the group resolves them over the gathered records in
// GLOBAL sector order with one group-level carry, so that a shard whose first sectors fail their very
// first evaluation reports what the last sector of the shard before it left behind - as the serial
// reference does (correlation_class.cpp:413-419, :870).
int lk_internal_set_defer_stale(lk_engine *e, int on);
int lk_internal_reference_order(const lk_engine *e);

// the same resolution over n_ranks padded blocks of `cap` records holding the shards [r*S/G, (r+1)*S/G)
hipError_t lk_launch_stale_iterations_blocks(lk_result *all, int n_sectors, int n_ranks, int cap, const int *carry_in,
                                             int *carry_out, hipStream_t st);

// Frames that reach a member through a collective on the group's communication stream: the engine's fill (upload copy +
// pyramid, on the engine's stream or - LK_IMG_NXT, ring slots - its next-frame stream) waits for `after` on the device
// instead of the host waiting for the collective, and records `consumed` behind the last kernel that reads the pixels
// (the next collective into that buffer waits for it).  Either event may be null.
int lk_internal_set_image_device_after(lk_engine *e, int slot, const void *device_pixels, int rows, int cols, int step,
                                       hipEvent_t after, hipEvent_t consumed);
int lk_internal_sequence_set_frame_device_after(lk_engine *e, int slot, const void *device_pixels, int rows, int cols, int step,
                                                hipEvent_t after, hipEvent_t consumed);
// 1: a solve of this engine launches teams of workgroups that wait for each other (collectives stay in stream order then)
int lk_internal_team_launches(const lk_engine *e);
int lk_internal_sector_count(const lk_engine *e);
// reference-order mode, a window of `frames` frames gathered as n_ranks blocks of [frames][cap] records: the stale
// iteration counts resolved in the order the reference solves - frame by frame, sector by sector
hipError_t lk_launch_stale_iterations_window(lk_result *all, int n_sectors, int n_ranks, int cap, int frames, const int *carry_in,
                                             int *carry_out, hipStream_t st);
