// mals_internal.h -- entry points shared by the translation units of libmyrrix_als.so, NOT part of the C-ABI
// (include/myrrix_als.h).  mals_group.cpp drives several handles from one thread: it must be able to enqueue the
// direct kernels of EVERY member before any host work (the eigendecomposition of the dual path) starts, and to
// decompose the group's Gramian once instead of once per member.
#pragma once
#include "../../include/myrrix_als.h"

extern "C" {
// mals_solve_chunk in two calls: BEGIN enqueues everything of the chunk that does not need the eigendecomposition
// of the opposite Gramian; when malsi_dual_pending(h) is then 1, the caller runs malsi_dual_host (on one member:
// from = NULL computes it; on the others: from = that member copies it) and END enqueues the rest.
__attribute__((visibility("hidden"))) int malsi_solve_chunk_begin(mals_handle h, int side, int32_t chunk);
__attribute__((visibility("hidden"))) int malsi_solve_chunk_end(mals_handle h, int side, int32_t chunk);
__attribute__((visibility("hidden"))) int malsi_dual_pending(mals_handle h);
__attribute__((visibility("hidden"))) int malsi_dual_host(mals_handle h, int side, mals_handle from);
// mals_gramian_partial / mals_set_gramian with the exact bound on |y| of the split-precision gather: the partial Gramian
// kernels also record max |element| of their rows (device_max: malsi_ymax_slots() caller-zeroed device words, bit patterns
// of floats >= 0, spread over several addresses because same-address atomics serialise; the maximum is what counts);
// the group all-reduces the members' maxima and installs the result with the summed Gramian.
__attribute__((visibility("hidden"))) int malsi_gramian_partial(mals_handle h, int side, int64_t row_begin, int64_t n_rows, double* device_out,
                                                              unsigned* device_max);
__attribute__((visibility("hidden"))) int malsi_ymax_slots(void);
__attribute__((visibility("hidden"))) int malsi_set_gramian(mals_handle h, int side, const double* G, int mem_kind, const unsigned* device_max);
// the event behind which everything a half-iteration sets up once (in its first chunk) is enqueued: a later chunk solved on
// ANOTHER stream waits for it (mals_group.cpp alternates two compute streams between consecutive chunks)
__attribute__((visibility("hidden"))) void* malsi_ready_event(mals_handle h);
// the thread's "why did create fail" text (mals_create_error): set by whichever create call fails, cleared by one that succeeds
__attribute__((visibility("hidden"))) void malsi_set_create_error(const char* text);
}
