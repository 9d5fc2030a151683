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
