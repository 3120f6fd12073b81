// mals_api.hip -- host side of libmyrrix_als.so: the C-ABI declared in include/myrrix_als.h.
// One handle = one GPU.  Holds the CSR shard of each side, the factor replicas, the Gramians and
// the work lists, and launches the gfx950 kernels of als_kernels.h on the handle's stream.
// No CPU fallback exists: every compute entry point needs a HIP device and fails with
// MALS_HIP_ERROR otherwise.
#include "../../include/myrrix_als.h"
#include "als_kernels.h"
#include "dual_kernels.h"
#include "lds_kernels.h"
#include "host_eigen.h"
#include "host_solver.h"
#include "topn_kernels.h"
#include "mals_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#ifndef MALS_D4
#define MALS_D4 4
#endif
using namespace mals;

namespace {

struct SideState {
  // factor replica
  int64_t n_total = 0;
  float* F = nullptr;
  bool F_owned = false;
  // local matrix shard
  int64_t row_offset = 0, n_local = 0, nnz = 0;
  int64_t* row_ptr = nullptr;
  int32_t* col = nullptr;
  float* val = nullptr;
  bool m_owned = false;
  bool has_matrix = false;
  std::vector<int64_t> h_row_ptr;  // host copy (work-list construction, chunked upload)
  int64_t append_rows = 0, append_nnz = 0;
  bool appending = false;
  // work lists
  WorkItem* itemsA = nullptr;  // rows no longer than segment_nnz, longest first
  int64_t nA = 0;
  int64_t nnzA = 0, nnzB = 0;  // entries handled by the rows kernel / the segments kernel
  WorkItem* itemsB = nullptr;  // segments of the long rows, longest first
  int64_t nB = 0;
  RowC* rowsC = nullptr;
  int64_t nC = 0;
  float* scratch = nullptr;
  uint8_t* refine = nullptr;   // per local row: marked for als_refine_kernel by the kernel that solved it
  size_t refine_cap = 0;
  int32_t col_min = 0, col_max = -1;  // range of the column indices (checked against the opposite replica)
  float max_abs_val = 0.f;  // bound on |value| used for the operand scale (>= local_max_abs_val)
  float local_max_abs_val = 0.f;  // largest |value| of the shard
  double mean_abs_val = 0.0;      // mean |value| used by the operand-range check (all shards if the caller synced it)
  double local_mean_abs_val = 0.0;  // mean |value| of the shard
  // the lists are stored chunk-major (contiguous ranges of cfg.chunk_rows rows of the shard), each
  // chunk sorted by length; a chunk can be solved on its own so that the caller can overlap the
  // exchange of finished chunks with the solve of the next one
  // itemsA of a chunk: [nA rows for the direct kernel, longest first, then ONE representative empty
  // row] [the dual classes: nD[3] rows with 48 < len <= 64, nD[2] ..., nD[0] rows with len <= 16 --
  // descending length throughout] [nZ further empty rows: x = 0, same verdict as the representative]
  struct ChunkRange {
    int64_t offA = 0, nA = 0, nnzA = 0, offB = 0, nB = 0, nnzB = 0, offC = 0, nC = 0;
    int64_t offP = 0, nP = 0;  // groups of partial slots reduced ahead of the finish kernel (entries of rowsC as well)
    int64_t nD[4] = {0, 0, 0, 0}, nnzD[4] = {0, 0, 0, 0}, nZ = 0;
    int64_t n_dual() const { return nD[0] + nD[1] + nD[2] + nD[3]; }
    int64_t nnz_dual() const { return nnzD[0] + nnzD[1] + nnzD[2] + nnzD[3]; }
  };
  int64_t n_dual_rows = 0;
  int64_t chunk_rows_override = -1;  // mals_set_chunk_rows: per-side value of cfg.chunk_rows (-1 = use cfg)
  std::vector<ChunkRange> chunks;
  // Gramian of THIS side's factors (consumed when solving the other side)
  double* G = nullptr;
  float* Gf = nullptr;
  double* partials = nullptr;   // wave / slab partials of K1 (doubles, or floats for the split kernel)
  size_t partial_bytes = 0;
  bool G_valid = false;
  uint64_t G_version = 0;  // bumped whenever G changes (cached operand scale of the split-precision gather)
  unsigned* d_ymax = nullptr;  // bit pattern of max |element| over the rows G was formed from, recorded by the Gramian kernels
  uint64_t ymax_version = 0;   // the G_version d_ymax belongs to (0 = none: G came from outside, gather_scale_kernel then
                               // bounds |y| by sqrt(max_f G_ff))
  uint64_t F_epoch = 0;    // bumped whenever the library itself writes the replica (uploads, solves, rebinding)
};

struct PendingEvent {
  hipEvent_t a, b;
  int kind;  // 0 = rows, 1 = segments, 2 = finish, 3 = gramian, 4 = dual, 5 = rotate
  double bytes;
};

}  // namespace

struct mals_handle_s {
  mals_config cfg;
  int T = 0;
  bool split_f16 = false;  // cfg.gramian_mode resolved
  bool split3 = false;     // MALS_GRAMIAN_SPLIT3_F16: three f16 terms per operand (features 49..64)
  int dual_blocks = 0;     // cfg.solve_mode resolved: rows up to 16*dual_blocks entries go to the dual lists (0 = none)
  // dual path state (dual_kernels.h): rotated copy of the gathered factor matrix, Q / Q^T / eigenvalues
  float* d_Mr = nullptr;
  size_t Mr_cap = 0;          // floats
  // gather table of the direct kernels when features % 16 != 0: zero-padded copy of the gathered factor matrix
  // (pad_rows_kernel), rebuilt once per half-iteration
  float* d_Mp = nullptr;
  size_t Mp_cap = 0;          // floats
  // mixed-precision refinement of ill-conditioned rows (als_refine_kernel): estimate above which a row is re-solved
  // (MALS_REFINE_LIMIT / mals_set_refine_limit; 0 = off), rows refined so far (device counter)
  float refine_limit = 64.f;
  bool exact_ready = false;   // als_exact_kernel's dynamic LDS limit raised
  int* d_gref_state = nullptr;   // {a row was marked in this half-iteration, Gref is valid, arrival ticket}
  double* d_Gref = nullptr;      // the reference-rounded Gramian of the gathered side (gramian_ref_kernel), on demand
  double* d_gref_part = nullptr;
  unsigned long long* d_refined = nullptr;
  int pad_side = -1;          // solved side whose opposite matrix the copy holds, version of that side's G and
  uint64_t pad_version = 0;   // factor-upload count at the time of the copy
  uint64_t pad_epoch = 0;
  std::vector<uint8_t> pad_done;  // chunks solved from the current copy: solving one again starts a new half-iteration
  char* d_eig = nullptr;      // device image of eig_stage; the four pointers below are views into it
  double* d_Q = nullptr;      // [2][16T][16T]: Q (k x 16T, zero padded) and Q^T
  float* d_Qf = nullptr;      // the same in fp32
  bool rotate_f64 = false;    // this half-iteration's forward rotation runs on the fp64 matrix cores
  bool rotate_split = false;  // k % 8 == 0: rotations on the f16 matrix pipe with split operands
  int32_t* d_Bs = nullptr;    // [2][KC][T][2][64][4]: Q and Q^T as split-f16 B operands
  size_t Bs_stride = 0;
  float* d_lam = nullptr;     // [2][16T]: eigenvalues, 1/sqrt(L + lambda alpha)
  unsigned* d_zbound = nullptr;
  double* h_G = nullptr;      // pinned k x k
  hipEvent_t ev_G = nullptr;
  // host half of the dual preparation: the eigendecomposition of G turned into what the device half uploads, in ONE
  // pinned block (asynchronous copies straight out of it, no stream synchronisation): Q | Q^T (doubles), the same in
  // fp32, eigenvalues | 1/sqrt(L + lambda alpha), and Q / Q^T as split-f16 B operands
  char* eig_stage = nullptr;
  size_t eig_stage_bytes = 0;
  bool eig_ok = false;        // the staged decomposition qualifies for the dual path
  double eig_gmax = 0.0;      // max_f G_ff of the decomposed Gramian
  bool dual_pending = false;  // a chunk's direct kernels are enqueued, its dual part waits for the eigendecomposition
  int dual_pending_side = -1, dual_pending_chunk = -1;
  double tl[4] = {0.0, 0.0, 0.0, 0.0};  // mals_get_timeline
  int dual_side = -1;         // what the rotated copy currently holds: solved side, version of the opposite G
  uint64_t dual_version = 0;
  bool dual_ok = false;       // the half-iteration's systems qualify for the dual path
  // k = 128 with the split-precision gather: the rows / segments kernels that stage the gather through LDS (lds_kernels.h)
  // and the Gramian image in their feature order (all zeros under lossIgnoresUnspecified), rebuilt when the opposite
  // side's Gramian changes
  // Chunks of one half-iteration on alternating streams (mals_group.cpp): ev_ready = everything the half-iteration sets up
  // once, in its first chunk (operand scale, padded / permuted images, the rotated copy of the dual path), is enqueued
  // behind it -- a later chunk on the OTHER stream waits for it; ev_refine serialises the chunks' refinement blocks
  // (gramian_ref_kernel's arrival ticket assumes one launch at a time)
  hipEvent_t ev_ready = nullptr, ev_refine = nullptr;
  bool refine_recorded = false;
  mals_iteration_fn iter_fn = nullptr;   // mals_set_iteration_callback
  void* iter_user = nullptr;
  bool lds_gather = false;
  float* d_Gperm = nullptr;
  int gperm_side = -1;
  uint64_t gperm_version = 0;
  float* d_zscale = nullptr;  // {S, 1/S^2} of the split-precision gather (gather_scale_kernel)
  int zs_side = -1;           // what d_zscale currently holds: solved side, version of the opposite G, value bound
  uint64_t zs_version = 0;
  float zs_bound = -1.f;
  double zs_mean = -1.0;
  unsigned* d_maxabs = nullptr;
  int* d_colrange = nullptr;
  int n_cu = 256;
  SideState side[2];
  hipStream_t stream = nullptr;
  // The three independent parts of a chunk -- rows list, long rows (segments + finish), dual lists (rotation + dual
  // kernels) -- are enqueued on three streams between a fork and a join event: the small launches (a few hundred
  // workgroups: the long rows of C2, the dual classes, the rotations) fill the slots the rows kernel's tail leaves
  // instead of each paying its own ramp-up and tail.  Measured (round 3, same box, one stream vs three): C2 2.22 ->
  // 2.18 ms, C4 81.8 -> 80.6, C5 rank 168.8 -> 165.3, C3 14.8 -> 15.3 (worse: its long rows then fight the rows
  // kernel for the cache-resident table) -- 1-2 %, and every per-kernel HIP-event time (the roofline figures of
  // mals_stats) then includes the other streams' contention.  So: OFF unless MALS_OVERLAP=1.
  hipStream_t aux[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  bool overlap = false;
  bool forked[2] = {false, false};
  std::string err;
  unsigned long long* d_bad = nullptr;   // [4]: first non-PD row per side, then the smallest-pivot suspect per side
  unsigned long long* h_bad = nullptr;   // pinned
  int32_t sing_side = -1;
  int64_t sing_row = -1;
  int32_t sing_rank = 0;
  std::atomic<int> cancelled{0};
  bool timing = false;
  mals_stats stats;
  std::vector<PendingEvent> pending;
  unsigned long long* d_trace = nullptr;  // MALS_DEBUG_TRACE
  void* tn_ws = nullptr;  // top-N workspace (topn_host.h), grow-only
  void* tn_front = nullptr;  // the serving front of mals_recommend*: queue, leader, passes in flight (topn_host.h)
  // userTagIDs as one bit per item (mals_set_tag_items): never recommended (RecommendIterator.java:72)
  uint32_t* tag_bits = nullptr;
  int64_t tag_bits_items = 0, n_tag_items = 0;
  // knownItemIDs (mals_set_known_items): what mals_recommend skips instead of the rows of R when present
  const int64_t* known_ptr = nullptr;
  const int32_t* known_idx = nullptr;
  int64_t known_rows = 0;
  int64_t* known_ptr_own = nullptr;   // copies of host arrays
  int32_t* known_idx_own = nullptr;
  double* d_sd = nullptr;       // mals_sample_dots: estimates, indices
  int64_t* d_sd_idx = nullptr;
  size_t sd_cap = 0, sd_idx_cap = 0;
  int64_t* d_idx = nullptr;  // gather scratch
  float* d_rows = nullptr;
  int32_t idx_cap = 0;
};

namespace {

int fail(mals_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

#define HIPCHK(h, call)                                                                     \
  do {                                                                                      \
    hipError_t _e = (call);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      const int _code = (_e == hipErrorOutOfMemory) ? MALS_OOM : MALS_HIP_ERROR;            \
      return fail(h, _code, std::string(#call) + ": " + hipGetErrorString(_e));             \
    }                                                                                       \
  } while (0)

#define CHECK_SIDE(h, side)                                                  \
  do {                                                                       \
    if (!(h)) return MALS_INVALID_ARG;                                       \
    if ((side) != MALS_SIDE_X && (side) != MALS_SIDE_Y)                      \
      return fail(h, MALS_INVALID_ARG, "side must be MALS_SIDE_X or _Y");    \
  } while (0)

int use_device(mals_handle h) {
  HIPCHK(h, hipSetDevice(h->cfg.device));
  return MALS_OK;
}

template <typename P>
void free_dev(P*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

void clear_known_items(mals_handle h) {
  free_dev(h->known_ptr_own);
  free_dev(h->known_idx_own);
  h->known_ptr = nullptr;
  h->known_idx = nullptr;
  h->known_rows = 0;
}

void free_matrix(SideState& s) {
  if (s.m_owned) {
    free_dev(s.row_ptr);
    free_dev(s.col);
    free_dev(s.val);
  }
  s.row_ptr = nullptr;
  s.col = nullptr;
  s.val = nullptr;
  s.m_owned = false;
  s.has_matrix = false;
  free_dev(s.itemsA);
  free_dev(s.itemsB);
  free_dev(s.rowsC);
  free_dev(s.scratch);
  free_dev(s.refine);
  s.refine_cap = 0;
  s.nA = s.nB = s.nC = 0;
  s.nnzA = s.nnzB = 0;
  s.chunks.clear();
  s.h_row_ptr.clear();
  s.h_row_ptr.shrink_to_fit();
}

int64_t slot_floats(int T) { return (int64_t)(tri(T) * 4 + T) * 64; }

// Split the rows of a shard into the three work lists (DESIGN.md "work decomposition"), chunk by chunk.
int build_work_lists(mals_handle h, SideState& s) {
  s.max_abs_val = 0.f;
  s.mean_abs_val = 0.0;
  s.local_mean_abs_val = 0.0;
  s.n_dual_rows = 0;
  if (s.nnz > 0) {  // one pass over the values: bounds the Gramian weights (gather_scale_kernel)
    HIPCHK(h, hipMemsetAsync(h->d_maxabs, 0, 4 * sizeof(unsigned), h->stream));
    const unsigned blocks = (unsigned)std::min<int64_t>(4096, (s.nnz + 255) / 256);
    hipLaunchKernelGGL(max_abs_kernel, dim3(blocks), dim3(256), 0, h->stream, s.val, s.nnz, h->d_maxabs,
                       reinterpret_cast<double*>(h->d_maxabs + 2));
    HIPCHK(h, hipGetLastError());
    unsigned raw[4];
    HIPCHK(h, hipMemcpyAsync(raw, h->d_maxabs, sizeof(raw), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::memcpy(&s.max_abs_val, &raw[0], sizeof(float));
    double sum = 0.0;
    std::memcpy(&sum, &raw[2], sizeof(double));
    s.mean_abs_val = sum / (double)s.nnz;
    s.local_mean_abs_val = s.mean_abs_val;
  }
  s.local_max_abs_val = s.max_abs_val;
  s.col_min = 0;
  s.col_max = -1;
  if (s.nnz > 0) {  // the same pass over the column indices: an index outside the opposite replica would be an
                    // out-of-bounds device access in the gather (and a write in the top-N mask)
    const int init[2] = {std::numeric_limits<int>::max(), std::numeric_limits<int>::min()};
    HIPCHK(h, hipMemcpyAsync(h->d_colrange, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
    const unsigned blocks = (unsigned)std::min<int64_t>(4096, (s.nnz + 255) / 256);
    hipLaunchKernelGGL(col_range_kernel, dim3(blocks), dim3(256), 0, h->stream, s.col, s.nnz, h->d_colrange);
    HIPCHK(h, hipGetLastError());
    int range[2];
    HIPCHK(h, hipMemcpyAsync(range, h->d_colrange, sizeof(range), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    s.col_min = range[0];
    s.col_max = range[1];
    if (s.col_min < 0) return fail(h, MALS_INVALID_ARG, "negative column index");
  }
  const int64_t n = s.n_local;
  const int seg = h->cfg.segment_nnz;
  const int64_t want_chunk = s.chunk_rows_override >= 0 ? s.chunk_rows_override : h->cfg.chunk_rows;
  const int64_t chunk_rows = want_chunk > 0 ? want_chunk : std::max<int64_t>(n, 1);
  const int n_chunks = (int)std::max<int64_t>(1, (n + chunk_rows - 1) / chunk_rows);
  const std::vector<int64_t>& rp = s.h_row_ptr;
  std::vector<WorkItem> order;
  std::vector<WorkItem> segs;
  std::vector<RowC> rowsC, groups;
  order.reserve((size_t)n);
  int64_t slot = 0;
  s.chunks.assign((size_t)n_chunks, SideState::ChunkRange());
  std::vector<int64_t> count((size_t)seg + 2);
  for (int c = 0; c < n_chunks; ++c) {
    const int64_t r0 = std::min(n, c * chunk_rows), r1 = std::min(n, (c + 1) * chunk_rows);
    SideState::ChunkRange& cr = s.chunks[(size_t)c];
    cr.offA = (int64_t)order.size();
    cr.offB = (int64_t)segs.size();
    cr.offC = (int64_t)rowsC.size();
    // counting sort of the chunk's short rows by length, longest first (stable: ascending row inside a length)
    std::fill(count.begin(), count.end(), 0);
    const int64_t dual_len = 16 * (int64_t)h->dual_blocks;  // rows with 1..dual_len entries: dual lists
    int64_t n_short = 0, n_direct = 0, n_empty = 0;
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t len = rp[r + 1] - rp[r];
      if (len < 0) return fail(h, MALS_INVALID_ARG, "row_ptr must be non-decreasing");
      if (len <= seg) {
        ++count[(size_t)(seg - len)];
        ++n_short;
        if (len == 0) {
          ++n_empty;
        } else if (len <= dual_len) {
          const int cls = (int)((len - 1) >> 4);
          ++cr.nD[cls];
          cr.nnzD[cls] += len;
        } else {
          ++n_direct;
          cr.nnzA += len;
        }
      } else {
        cr.nnzB += len;
      }
    }
    // How the long rows (more than segment_nnz entries) are cut: segments of up to segment_nnz entries when that gives
    // at least one per SIMD, shorter ones otherwise -- a wave takes ~20 ns per entry, and the user half of C2 (a handful
    // of rows above 4 096 entries) ran FOUR waves for 80 us on the critical path (round 3: 80 -> 22 us).  Each segment
    // costs a partial slot, written and read back one after the other by the finish kernel's one wave per row: with
    // six times the segments on C2's item half the finish kernel went from 23 to 70 us and ate the gain -- hence only
    // for lists that cannot even give every SIMD a segment, and never below 512 entries.
    int64_t seg_b = seg;
    {
      const int64_t want_segments = (int64_t)h->n_cu * 4;
      const int64_t lo = std::min<int64_t>(512, seg);
      seg_b = std::max<int64_t>(lo, std::min<int64_t>(seg, (cr.nnzB + want_segments - 1) / want_segments));
      seg_b = (seg_b + 31) & ~(int64_t)31;                   // whole 32-entry super-steps
      if (seg_b > seg) seg_b = seg;
    }
    int64_t acc = cr.offA;
    for (size_t b = 0; b < count.size(); ++b) {
      const int64_t cnt = count[b];
      count[b] = acc;
      acc += cnt;
    }
    order.resize((size_t)(cr.offA + n_short));
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t len = rp[r + 1] - rp[r];
      if (len <= seg) {
        WorkItem& w = order[(size_t)count[(size_t)(seg - len)]++];
        w.begin = rp[r];
        w.len = (int32_t)len;
        w.id = (int32_t)r;
      } else {
        const int64_t nseg = (len + seg_b - 1) / seg_b;
        int64_t per = (len + nseg - 1) / nseg;
        per = (per + 3) & ~(int64_t)3;  // whole 4-entry steps
        RowC rc;
        rc.first_slot = slot;
        rc.row = (int32_t)r;
        rc.nseg = 0;
        rc.stride = 1;
        rc.pad_ = 0;
        for (int64_t b = 0; b < len; b += per) {
          if (slot >= std::numeric_limits<int32_t>::max()) return fail(h, MALS_INVALID_ARG, "too many row segments");
          WorkItem sg;
          sg.begin = rp[r] + b;
          sg.len = (int32_t)std::min(per, len - b);
          sg.id = (int32_t)slot++;
          segs.push_back(sg);
          ++rc.nseg;
        }
        if (rc.nseg > FINISH_GROUP) {  // group sums first (als_prereduce_kernel), the finish kernel walks the leaders
          // (groups of max(8, sqrt(segments)) slots, more rows grouped: finish + group sums 0.29 -> 0.35 ms on C4 -- every
          // group sum is a slot read and written once more)
          const int32_t grp_len = FINISH_GROUP;
          for (int32_t g0 = 0; g0 < rc.nseg; g0 += grp_len) {
            RowC grp;
            grp.first_slot = rc.first_slot + g0;
            grp.row = (int32_t)r;
            grp.nseg = std::min<int32_t>(grp_len, rc.nseg - g0);
            grp.stride = 1;
            grp.pad_ = 0;
            groups.push_back(grp);
          }
          rc.nseg = (rc.nseg + grp_len - 1) / grp_len;
          rc.stride = grp_len;
        }
        rowsC.push_back(rc);
      }
    }
    // Every empty row has the same system (W = G, b = 0: ALS:447-494 with no entries): x = 0, and the
    // verdict "singular" is shared.  One representative (the smallest row) goes through the kernel, right
    // behind the direct rows; the others are only zero-filled.
    const int64_t n_dual = cr.n_dual();
    if (n_empty > 0 && n_dual > 0) {
      auto first = order.begin() + (size_t)(cr.offA + n_direct);
      std::rotate(first, first + (size_t)n_dual, first + (size_t)n_dual + 1);
    }
    cr.nA = n_direct + (n_empty > 0 ? 1 : 0);
    cr.nZ = n_empty > 0 ? n_empty - 1 : 0;
    s.n_dual_rows += n_dual;
    cr.nB = (int64_t)segs.size() - cr.offB;
    cr.nC = (int64_t)rowsC.size() - cr.offC;
    cr.offP = (int64_t)rowsC.size();  // the chunk's slot groups sit behind its rows in the same array
    cr.nP = (int64_t)groups.size();
    rowsC.insert(rowsC.end(), groups.begin(), groups.end());
    groups.clear();
    // longest segments first
    std::stable_sort(segs.begin() + cr.offB, segs.end(), [](const WorkItem& a, const WorkItem& b) { return a.len > b.len; });
    s.nnzA += cr.nnzA;
    s.nnzB += cr.nnzB;
  }
  s.nA = (int64_t)order.size();
  s.nB = (int64_t)segs.size();
  s.nC = (int64_t)rowsC.size();
  if (std::getenv("MALS_DEBUG_LISTS")) {  // work decomposition of the upload (tuning aid)
    int64_t nd[4] = {0, 0, 0, 0}, ed[4] = {0, 0, 0, 0}, na = 0, nz = 0, n_long = 0;   // (s.nC also counts the slot groups)
    for (const SideState::ChunkRange& cr : s.chunks) {
      n_long += cr.nC;
      for (int c = 0; c < 4; ++c) {
        nd[c] += cr.nD[c];
        ed[c] += cr.nnzD[c];
      }
      na += cr.nA;
      nz += cr.nZ;
    }
    std::fprintf(stderr, "[lists] rows %lld: direct %lld (%lld entries), dual 1-16: %lld (%lld) 17-32: %lld (%lld) 33-48: %lld (%lld) 49-64: %lld (%lld), "
                         "zero-filled %lld, segments %lld of %lld long rows (%lld entries)\n",
                 (long long)n, (long long)na, (long long)s.nnzA, (long long)nd[0], (long long)ed[0], (long long)nd[1], (long long)ed[1], (long long)nd[2],
                 (long long)ed[2], (long long)nd[3], (long long)ed[3], (long long)nz, (long long)s.nB, (long long)n_long, (long long)s.nnzB);
  }
  if (s.nA) {
    HIPCHK(h, hipMalloc(&s.itemsA, sizeof(WorkItem) * order.size()));
    HIPCHK(h, hipMemcpy(s.itemsA, order.data(), sizeof(WorkItem) * order.size(), hipMemcpyHostToDevice));
  }
  if (s.nB) {
    HIPCHK(h, hipMalloc(&s.itemsB, sizeof(WorkItem) * segs.size()));
    HIPCHK(h, hipMemcpy(s.itemsB, segs.data(), sizeof(WorkItem) * segs.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMalloc(&s.rowsC, sizeof(RowC) * rowsC.size()));
    HIPCHK(h, hipMemcpy(s.rowsC, rowsC.data(), sizeof(RowC) * rowsC.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMalloc(&s.scratch, sizeof(float) * (size_t)(slot * slot_floats(h->T))));
  }
  return MALS_OK;
}

int validate_matrix(mals_handle h, int side) {
  SideState& s = h->side[side];
  if (s.h_row_ptr.size() != (size_t)s.n_local + 1 || s.h_row_ptr[0] != 0 || s.h_row_ptr[(size_t)s.n_local] != s.nnz)
    return fail(h, MALS_INVALID_ARG, "row_ptr must have n_rows_local+1 entries, start at 0 and end at nnz");
  return MALS_OK;
}

// col_idx must index rows of the OPPOSITE side's factor replica (checked again at solve time: the
// replica may be declared after the matrix)
int validate_columns(mals_handle h, int side) {
  const SideState& s = h->side[side];
  const SideState& o = h->side[1 - side];
  if (s.nnz > 0 && o.n_total > 0 && (int64_t)s.col_max >= o.n_total)
    return fail(h, MALS_INVALID_ARG, "column index outside the opposite side's factor replica");
  return MALS_OK;
}

// ---- timing ------------------------------------------------------------------------------------
int begin_timed(mals_handle h, int kind, double bytes, PendingEvent& pe) {
  pe.kind = kind;
  pe.bytes = bytes;
  pe.a = pe.b = nullptr;
  if (!h->timing) return MALS_OK;
  HIPCHK(h, hipEventCreate(&pe.a));
  HIPCHK(h, hipEventCreate(&pe.b));
  HIPCHK(h, hipEventRecord(pe.a, h->stream));
  return MALS_OK;
}
int end_timed(mals_handle h, PendingEvent& pe) {
  if (!h->timing) return MALS_OK;
  HIPCHK(h, hipEventRecord(pe.b, h->stream));
  h->pending.push_back(pe);
  return MALS_OK;
}
int drain_events(mals_handle h) {
  for (PendingEvent& pe : h->pending) {
    HIPCHK(h, hipEventSynchronize(pe.b));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, pe.a, pe.b));
    mals_stats& st = h->stats;
    switch (pe.kind) {
      case 0: st.rows_ms += ms; st.rows_launches += 1; st.rows_bytes += pe.bytes; break;
      case 1: st.segments_ms += ms; st.segments_launches += 1; st.segments_bytes += pe.bytes; break;
      case 2: st.finish_ms += ms; st.finish_launches += 1; st.finish_bytes += pe.bytes; break;
      case 4: st.dual_ms += ms; st.dual_launches += 1; st.dual_bytes += pe.bytes; break;
      case 5: st.rotate_ms += ms; st.rotate_launches += 1; st.rotate_bytes += pe.bytes; break;
      default: st.gramian_ms += ms; st.gramian_launches += 1; st.gramian_bytes += pe.bytes; break;
    }
    (void)hipEventDestroy(pe.a);
    (void)hipEventDestroy(pe.b);
  }
  h->pending.clear();
  return MALS_OK;
}

// ---- kernel dispatch ---------------------------------------------------------------------------
// fp64 kernel below this many rows, split-f16 kernel (slabs of 512 rows = 32 accumulation steps, fp64 sum over the
// slabs) from there on: even when a few rows dominate G (no averaging over slabs) the fp32 slab sums stay within
// 6e-8 x sqrt(32) of it (als_kernels.h, gramian_split_kernel)
constexpr int64_t GRAMIAN_SPLIT_MIN_ROWS = 262144;
// Rows a wave of gramian_split_kernel sums in fp32 before its partial goes to the fp64 stages.  512 keeps a 262144-row matrix
// at 128 workgroups; from 4M rows on the 10 KB partial per slab (200 MB written and read again at 10M rows, a sixth of the
// input) is worth more than the extra waves: 2048 there (10M x 64: 0.67 -> 0.53 ms).  fp32 roundings per accumulator and slab:
// 3 per 32-row step, 192 at 2048 rows -- 8e-7 relative at random, averaged over thousands of slabs in fp64.
inline int64_t gramian_slab_rows(int64_t n_rows) {
  static const int64_t forced = std::getenv("MALS_GRAMIAN_SLAB_ROWS") ? std::atoll(std::getenv("MALS_GRAMIAN_SLAB_ROWS")) : 0;  // tuning override (a multiple of 64)
  if (forced > 0) return forced;
  int64_t slab = 512;
  while (slab < 2048 && n_rows / (2 * slab) / 4 >= 1024) slab *= 2;
  return slab;
}

template <int T>
int launch_gramian_T(mals_handle h, SideState& s, const float* M, int64_t n_rows, double* G_out, float* Gf_out, unsigned* ymax) {
  const int k = h->cfg.features;
  const int elems = tri(T) * 256;
  static const bool force_f64 = std::getenv("MALS_GRAMIAN_F64") != nullptr;  // A/B
  if (n_rows >= GRAMIAN_SPLIT_MIN_ROWS && !force_f64) {
    const int64_t slab_rows = gramian_slab_rows(n_rows);
    int64_t n_slabs = (n_rows + slab_rows - 1) / slab_rows;
    n_slabs = (n_slabs + 3) & ~(int64_t)3;  // whole workgroups
    constexpr int GROUPS = 64;              // first-stage sums (doubles) behind the slab partials (floats)
    const size_t slab_bytes = sizeof(float) * (size_t)n_slabs * tri(T) * 256;
    const size_t bytes = slab_bytes + sizeof(double) * (size_t)GROUPS * tri(T) * 256;
    if (s.partial_bytes < bytes) {
      free_dev(s.partials);
      s.partial_bytes = 0;
      void* pbuf = nullptr;
      HIPCHK(h, hipMalloc(&pbuf, bytes));
      s.partials = static_cast<double*>(pbuf);
      s.partial_bytes = bytes;
    }
    float* pf = reinterpret_cast<float*>(s.partials);
    double* pd = reinterpret_cast<double*>(reinterpret_cast<char*>(s.partials) + slab_bytes);  // slab_bytes is a multiple of 1024
    hipLaunchKernelGGL((gramian_split_kernel<T>), dim3((unsigned)(n_slabs / 4)), dim3(256), 0, h->stream, M, n_rows, k,
                       slab_rows, pf, ymax);
    hipLaunchKernelGGL(gramian_reduce_slabs_kernel, dim3((unsigned)((elems + 255) / 256), GROUPS), dim3(256), 0, h->stream, pf, n_slabs,
                       elems, GROUPS, pd);
    hipLaunchKernelGGL((gramian_finalize_kernel<T, false>), dim3(elems / 64), dim3(256), 0, h->stream, pd, (int64_t)GROUPS, k, G_out, Gf_out);
    HIPCHK(h, hipGetLastError());
    return MALS_OK;
  }
  // waves: at least 64 rows each, at most 1024 waves (the finalize pass costs per partial).  (256 rows each until round 3: a 17 770-row Y then ran on 72 of the
  // chip's 1024 SIMDs for 174 us at k = 100, C2's two matrices for 44 us each -- 7 % of its iteration.)
  int64_t n_waves = std::min<int64_t>(1024, std::max<int64_t>(1, (n_rows + 63) / 64));
  n_waves = (n_waves + 3) & ~(int64_t)3;
  int64_t rows_per_wave = (n_rows + n_waves - 1) / n_waves;
  rows_per_wave = std::max<int64_t>(4, (rows_per_wave + 3) & ~(int64_t)3);
  const size_t bytes = sizeof(double) * (size_t)n_waves * tri(T) * 256;
  if (s.partial_bytes < bytes) {
    free_dev(s.partials);
    s.partial_bytes = 0;
    HIPCHK(h, hipMalloc(&s.partials, bytes));
    s.partial_bytes = bytes;
  }
  // the fp64 kernel (small matrices: the diagonal's bound is at most 9 binades loose there) records no maximum; tracking
  // it cost the latency-bound kernel 20-40 % (measured on C2).  "Unknown" = a word above 3e38 that survives every max.
  if (ymax) HIPCHK(h, hipMemsetAsync(ymax, 0x7f, sizeof(unsigned), h->stream));
  hipLaunchKernelGGL((gramian_partial_kernel<T>), dim3((unsigned)(n_waves / 4)), dim3(256), 0, h->stream, M, n_rows, k,
                     rows_per_wave, s.partials);
  hipLaunchKernelGGL((gramian_finalize_kernel<T, true, 16>), dim3(elems / 16), dim3(256), 0, h->stream, s.partials, n_waves,
                     k, G_out, Gf_out);
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}

int launch_gramian(mals_handle h, SideState& s, const float* M, int64_t n_rows, double* G_out, float* Gf_out, unsigned* ymax = nullptr) {
  switch (h->T) {
    case 1: return launch_gramian_T<1>(h, s, M, n_rows, G_out, Gf_out, ymax);
    case 2: return launch_gramian_T<2>(h, s, M, n_rows, G_out, Gf_out, ymax);
    case 3: return launch_gramian_T<3>(h, s, M, n_rows, G_out, Gf_out, ymax);
    case 4: return launch_gramian_T<4>(h, s, M, n_rows, G_out, Gf_out, ymax);
    case 5: return launch_gramian_T<5>(h, s, M, n_rows, G_out, Gf_out, ymax);
    case 6: return launch_gramian_T<6>(h, s, M, n_rows, G_out, Gf_out, ymax);
    case 7: return launch_gramian_T<7>(h, s, M, n_rows, G_out, Gf_out, ymax);
    case 8: return launch_gramian_T<8>(h, s, M, n_rows, G_out, Gf_out, ymax);
  }
  return fail(h, MALS_INVALID_ARG, "unsupported feature count");
}

// Grid of the persistent kernels.  Work item i goes to wave i mod W of a length-sorted list, so every
// wave gets the same mix of row lengths; W is 16x the resident capacity (CUs x blocks/CU from the
// occupancy query), never more than the work needs.  Measured on C4 (profiles/): exactly-resident
// grids lose ~10% to waves that run slower than their peers (nothing rebalances a static
// assignment), 12-16x oversubscription lets the dispatcher level that out and each wave still
// walks >100 rows, which keeps the cross-row prefetch effective; beyond 16x nothing changes.
template <typename K>
int persistent_grid(mals_handle h, K kernel, int64_t n_work, unsigned* grid) {
  int per_cu = 0;
  HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0));
  if (per_cu < 1) per_cu = 1;
  per_cu *= 16;
  if (const char* e = std::getenv("MALS_BLOCKS_PER_CU")) per_cu = std::max(1, std::atoi(e));  // tuning override
  const int64_t cap = (int64_t)h->n_cu * per_cu;
  *grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_work + 3) / 4, cap));
  return MALS_OK;
}

// what one launch_solve call covers: the rows list (+ zero-fill) / the dual lists through the direct kernel / the long
// rows (segments + finish).  The three are independent of each other (disjoint output rows, read-only inputs).
enum { LISTS_ROWS = 1, LISTS_DUAL_ROWS = 2, LISTS_LONG = 4, LISTS_OWN = LISTS_ROWS | LISTS_LONG };

// One persistent launch, plus -- for a split-precision kernel -- its fp32-gather twin right behind it with
// flag bit 3: exactly one of the two does the work (gather_scale_kernel's range flag), the other returns at once.
template <typename K, typename KF>
int launch_persistent(mals_handle h, K kernel, KF fallback, const SolveParams& p, int kind, double bytes) {
  PendingEvent pe;
  unsigned grid = 1;
  if (int rc = persistent_grid(h, kernel, p.n_work, &grid)) return rc;
  if (int rc = begin_timed(h, kind, bytes, pe)) return rc;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, h->stream, p);
  if constexpr (!std::is_same<KF, std::nullptr_t>::value) {
    SolveParams pf = p;
    pf.flags |= 8;
    if (int rc = persistent_grid(h, fallback, p.n_work, &grid)) return rc;
    // the twin almost always returns at once: a resident-sized grid keeps that at a few microseconds (a full 16x
    // oversubscribed grid of empty workgroups costs 15-30); when it does run it only loses the oversubscription
    grid = std::min<unsigned>(grid, (unsigned)h->n_cu * 8u);
    hipLaunchKernelGGL(fallback, dim3(grid), dim3(256), 0, h->stream, pf);
  }
  return end_timed(h, pe);
}

template <typename KA, typename KB, typename KFA, typename KFB, typename KC, typename KP>
int launch_lists(mals_handle h, SideState& s, SolveParams p, int chunk, int which, KA rows_kernel, KB segments_kernel, KFA rows_fallback,
                 KFB segments_fallback, KC finish_kernel, KP prereduce_kernel) {
  const double per = 4.0 * p.k + 8.0;  // gathered row + col idx + value; written row + row_ptr
  const SideState::ChunkRange& cr = s.chunks[(size_t)chunk];
  PendingEvent pe;
  const bool own = which & LISTS_ROWS, longs = which & LISTS_LONG, dual_rows_too = which & LISTS_DUAL_ROWS;
  if (longs && cr.nB) {
    p.n_work = cr.nB;
    p.items = s.itemsB + cr.offB;
    if (int rc = launch_persistent(h, segments_kernel, segments_fallback, p, 1, (double)cr.nnzB * per)) return rc;
  }
  if (own && cr.nA) {
    p.n_work = cr.nA;
    p.items = s.itemsA + cr.offA;
    if (int rc = launch_persistent(h, rows_kernel, rows_fallback, p, 0, (double)cr.nnzA * per + (double)cr.nA * per)) return rc;
  }
  if (dual_rows_too && cr.n_dual()) {  // the dual lists through the direct kernel (the half-iteration does not qualify)
    p.n_work = cr.n_dual();
    p.items = s.itemsA + cr.offA + cr.nA;
    if (int rc = launch_persistent(h, rows_kernel, rows_fallback, p, 0, (double)cr.nnz_dual() * per + (double)cr.n_dual() * per)) return rc;
  }
  if (longs && cr.nC) {
    if (int rc = begin_timed(h, 2, (double)cr.nC * per, pe)) return rc;
    if (cr.nP) {
      p.n_work = cr.nP;
      p.rowsC = s.rowsC + cr.offP;
      hipLaunchKernelGGL(prereduce_kernel, dim3((unsigned)((cr.nP + 3) / 4)), dim3(256), 0, h->stream, p);
    }
    p.n_work = cr.nC;
    p.rowsC = s.rowsC + cr.offC;
    hipLaunchKernelGGL(finish_kernel, dim3((unsigned)((cr.nC + 3) / 4)), dim3(256), 0, h->stream, p);
    if (int rc = end_timed(h, pe)) return rc;
  }
  if (own && cr.nZ) {
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((cr.nZ + 31) / 32)), dim3(256), 0, h->stream,  // 8 rows per wave
                       s.itemsA + cr.offA + cr.nA + cr.n_dual(), cr.nZ, p.k, p.out);
  }
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}

// the persistent launch of an LDS-staged kernel (lds_kernels.h): one wave per workgroup, LDS-limited to 8 per CU, the same
// 16x oversubscription as persistent_grid
template <typename K, typename KF>
int launch_persistent_lds(mals_handle h, K kernel, KF fallback, const SolveParams& p, int kind, double bytes) {
  PendingEvent pe;
  int per_cu = 8 * 16;
  if (const char* e = std::getenv("MALS_BLOCKS_PER_CU")) per_cu = std::max(1, std::atoi(e)) * 4;  // tuning override (in 256-thread blocks)
  unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(p.n_work, (int64_t)h->n_cu * per_cu));
  if (int rc = begin_timed(h, kind, bytes, pe)) return rc;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, h->stream, p);
  SolveParams pf = p;
  pf.flags |= 8;
  if (int rc = persistent_grid(h, fallback, p.n_work, &grid)) return rc;
  grid = std::min<unsigned>(grid, (unsigned)h->n_cu * 8u);   // (as launch_persistent: the twin almost always returns at once)
  hipLaunchKernelGGL(fallback, dim3(grid), dim3(256), 0, h->stream, pf);
  return end_timed(h, pe);
}

// launch_lists for k = 128 through the LDS-staged kernels.  Their partial slots are in the kernels' own feature order, the
// fp32 twins' (range flag 0) in the plain one: both finish kernels are enqueued and the range flag decides on the device
// which of the two runs (bit 3 again: "only if the split-precision launch did not run").
int launch_lists_lds(mals_handle h, SideState& s, SolveParams p, int chunk, int which) {
  constexpr int T = 8, D = 2;
  const double per = 4.0 * p.k + 8.0;
  const SideState::ChunkRange& cr = s.chunks[(size_t)chunk];
  PendingEvent pe;
  const bool own = which & LISTS_ROWS, longs = which & LISTS_LONG, dual_rows_too = which & LISTS_DUAL_ROWS;
  if (longs && cr.nB) {
    p.n_work = cr.nB;
    p.items = s.itemsB + cr.offB;
    if (int rc = launch_persistent_lds(h, als_lds_kernel_h<1>, als_persistent_kernel<T, D, 1, true>, p, 1, (double)cr.nnzB * per)) return rc;
  }
  if (own && cr.nA) {
    p.n_work = cr.nA;
    p.items = s.itemsA + cr.offA;
    if (int rc = launch_persistent_lds(h, als_lds_kernel_h<0>, als_persistent_kernel<T, D, 0, true>, p, 0, (double)cr.nnzA * per + (double)cr.nA * per)) return rc;
  }
  if (dual_rows_too && cr.n_dual()) {
    p.n_work = cr.n_dual();
    p.items = s.itemsA + cr.offA + cr.nA;
    if (int rc = launch_persistent_lds(h, als_lds_kernel_h<0>, als_persistent_kernel<T, D, 0, true>, p, 0,
                                       (double)cr.nnz_dual() * per + (double)cr.n_dual() * per)) return rc;
  }
  if (longs && cr.nC) {
    if (int rc = begin_timed(h, 2, (double)cr.nC * per, pe)) return rc;
    if (cr.nP) {
      p.n_work = cr.nP;
      p.rowsC = s.rowsC + cr.offP;
      hipLaunchKernelGGL(als_prereduce_kernel<T>, dim3((unsigned)((cr.nP + 3) / 4)), dim3(256), 0, h->stream, p);
    }
    p.n_work = cr.nC;
    p.rowsC = s.rowsC + cr.offC;
    hipLaunchKernelGGL((als_finish_kernel<T, true>), dim3((unsigned)((cr.nC + 3) / 4)), dim3(256), 0, h->stream, p);
    SolveParams pf = p;
    pf.flags |= 8;
    hipLaunchKernelGGL((als_finish_kernel<T, false>), dim3((unsigned)((cr.nC + 3) / 4)), dim3(256), 0, h->stream, pf);
    if (int rc = end_timed(h, pe)) return rc;
  }
  if (own && cr.nZ) {
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((cr.nZ + 31) / 32)), dim3(256), 0, h->stream,
                       s.itemsA + cr.offA + cr.nA + cr.n_dual(), cr.nZ, p.k, p.out);
  }
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}

template <int T, int D, bool FULL>
int launch_solve_TF(mals_handle h, SideState& s, const SolveParams& p, int chunk, int which) {
  if constexpr (T == 8 && FULL) {
    if (h->split_f16 && h->lds_gather && p.Gperm) return launch_lists_lds(h, s, p, chunk, which);
  }
  if constexpr (T == 4) {
    if (h->split3)
      return launch_lists(h, s, p, chunk, which, als_persistent_kernel_h<T, 0, FULL, 3>, als_persistent_kernel_h<T, 1, FULL, 3>,
                          als_persistent_kernel<T, D, 0, FULL>, als_persistent_kernel<T, D, 1, FULL>, als_finish_kernel<T>,
                          als_prereduce_kernel<T>);
  }
  if (h->split_f16)
    return launch_lists(h, s, p, chunk, which, als_persistent_kernel_h<T, 0, FULL>, als_persistent_kernel_h<T, 1, FULL>,
                        als_persistent_kernel<T, D, 0, FULL>, als_persistent_kernel<T, D, 1, FULL>, als_finish_kernel<T>,
                        als_prereduce_kernel<T>);
  return launch_lists(h, s, p, chunk, which, als_persistent_kernel<T, D, 0, FULL>, als_persistent_kernel<T, D, 1, FULL>,
                      nullptr, nullptr, als_finish_kernel<T>, als_prereduce_kernel<T>);
}

template <int T, int D>
int launch_solve_T(mals_handle h, SideState& s, const SolveParams& p, int chunk, int which) {
  return p.k == 16 * T ? launch_solve_TF<T, D, true>(h, s, p, chunk, which) : launch_solve_TF<T, D, false>(h, s, p, chunk, which);
}

// the direct kernels over a chunk's lists (LISTS_*)
int launch_solve(mals_handle h, SideState& s, const SolveParams& p, int chunk, int which) {
  switch (h->T) {
    case 1: return launch_solve_T<1, 4>(h, s, p, chunk, which);
    case 2: return launch_solve_T<2, 4>(h, s, p, chunk, which);
    case 3: return launch_solve_T<3, 4>(h, s, p, chunk, which);
    case 4: return launch_solve_T<4, MALS_D4>(h, s, p, chunk, which);
    case 5: return launch_solve_T<5, 2>(h, s, p, chunk, which);
    case 6: return launch_solve_T<6, 2>(h, s, p, chunk, which);
    case 7: return launch_solve_T<7, 2>(h, s, p, chunk, which);
    case 8: return launch_solve_T<8, 2>(h, s, p, chunk, which);
  }
  return fail(h, MALS_INVALID_ARG, "unsupported feature count");
}

// ---- refinement of the rows the solving kernels marked (als_refine_kernel) -------------------------
template <int T>
int launch_refine_T(mals_handle h, const RefineParams& q) {
  // reads the marks on the device: no host round trip.  A resident-sized grid; a chunk without marked rows costs
  // one pass over its flag bytes.
  const int64_t rows = q.row_end - q.row_begin;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((rows + 255) / 256, (int64_t)h->n_cu * 8));
  hipLaunchKernelGGL((als_refine_kernel<T>), dim3(grid), dim3(256), 0, h->stream, q);
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}
int launch_refine(mals_handle h, const RefineParams& q) {
  switch (h->T) {
    case 1: return launch_refine_T<1>(h, q);
    case 2: return launch_refine_T<2>(h, q);
    case 3: return launch_refine_T<3>(h, q);
    case 4: return launch_refine_T<4>(h, q);
    case 5: return launch_refine_T<5>(h, q);
    case 6: return launch_refine_T<6>(h, q);
    case 7: return launch_refine_T<7>(h, q);
    case 8: return launch_refine_T<8>(h, q);
  }
  return fail(h, MALS_INVALID_ARG, "unsupported feature count");
}

// The reference's M^T M (MU:219-239 rounds every product to fp32) for the refinement of marked rows; the kernel
// returns at once unless a row has been marked in this half-iteration and the matrix is not there yet.
template <int T>
int launch_gramian_ref_T(mals_handle h, const SideState& o) {
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((o.n_total + 63) / 64, (int64_t)h->n_cu * 2));
  hipLaunchKernelGGL(gramian_ref_latch_kernel, dim3(1), dim3(1), 0, h->stream, h->d_gref_state);
  hipLaunchKernelGGL((gramian_ref_kernel<T>), dim3(grid), dim3(256), 0, h->stream, o.F, o.n_total, h->cfg.features, h->d_gref_state,
                     h->d_gref_part, h->d_Gref);
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}
int launch_gramian_ref(mals_handle h, const SideState& o) {
  switch (h->T) {
    case 1: return launch_gramian_ref_T<1>(h, o);
    case 2: return launch_gramian_ref_T<2>(h, o);
    case 3: return launch_gramian_ref_T<3>(h, o);
    case 4: return launch_gramian_ref_T<4>(h, o);
    case 5: return launch_gramian_ref_T<5>(h, o);
    case 6: return launch_gramian_ref_T<6>(h, o);
    case 7: return launch_gramian_ref_T<7>(h, o);
    case 8: return launch_gramian_ref_T<8>(h, o);
  }
  return fail(h, MALS_INVALID_ARG, "unsupported feature count");
}

int launch_exact(mals_handle h, const RefineParams& q, int level) {
  const int k = h->cfg.features;
  const size_t lds = sizeof(double) * ((size_t)k * (k + 1) + 2 * (size_t)k) + sizeof(float) * (size_t)k + sizeof(int) * 256;
  if (!h->exact_ready) {
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&als_exact_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->exact_ready = true;
  }
  const int64_t rows = q.row_end - q.row_begin;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((rows + 255) / 256, (int64_t)h->n_cu * 4));
  hipLaunchKernelGGL((als_exact_kernel<0>), dim3(grid), dim3(256), lds, h->stream, q, level);
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}

// ---- dual path (dual_kernels.h) ------------------------------------------------------------------
template <int T, bool LISTED, bool F64>
int launch_rotate_T(mals_handle h, const RotateParams& rp) {
  const int64_t tiles = (rp.n_rows + (F64 ? 31 : 63)) / (F64 ? 32 : 64);
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((tiles + 3) / 4, (int64_t)h->n_cu * 16));
  hipLaunchKernelGGL((rotate_rows_kernel<T, LISTED, F64>), dim3(grid), dim3(256), 0, h->stream, rp);
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}
template <bool LISTED, bool F64>
int launch_rotate(mals_handle h, const RotateParams& rp) {
  switch (h->T) {
    case 2: return launch_rotate_T<2, LISTED, F64>(h, rp);
    case 3: return launch_rotate_T<3, LISTED, F64>(h, rp);
    case 4: return launch_rotate_T<4, LISTED, F64>(h, rp);
    case 5: return launch_rotate_T<5, LISTED, F64>(h, rp);
    case 6: return launch_rotate_T<6, LISTED, F64>(h, rp);
    case 7: return launch_rotate_T<7, LISTED, F64>(h, rp);
    case 8: return launch_rotate_T<8, LISTED, F64>(h, rp);
  }
  return fail(h, MALS_INVALID_ARG, "unsupported feature count");
}

template <int T, bool LISTED>
int launch_rotate_split_T(mals_handle h, const RotateSplitParams& rp) {
  const int rows_per_wave = 16 * rotate_split_tiles(T);
  const int64_t tiles = (rp.n_rows + rows_per_wave - 1) / rows_per_wave;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((tiles + 3) / 4, (int64_t)h->n_cu * 16));
  hipLaunchKernelGGL((rotate_rows_split_kernel<T, LISTED>), dim3(grid), dim3(256), 0, h->stream, rp);
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}
template <bool LISTED>
int launch_rotate_split(mals_handle h, const RotateSplitParams& rp) {
  switch (h->T) {
    case 2: return launch_rotate_split_T<2, LISTED>(h, rp);
    case 3: return launch_rotate_split_T<3, LISTED>(h, rp);
    case 4: return launch_rotate_split_T<4, LISTED>(h, rp);
    case 5: return launch_rotate_split_T<5, LISTED>(h, rp);
    case 6: return launch_rotate_split_T<6, LISTED>(h, rp);
    case 7: return launch_rotate_split_T<7, LISTED>(h, rp);
    case 8: return launch_rotate_split_T<8, LISTED>(h, rp);
  }
  return fail(h, MALS_INVALID_ARG, "unsupported feature count");
}

// fp32 -> f16 bit pattern, round to nearest even (the compiler's conversion) / toward zero
uint16_t half_rne(float v) {
  const _Float16 x = (_Float16)v;
  uint16_t b;
  std::memcpy(&b, &x, 2);
  return b;
}
uint16_t half_rtz(float v) {
  uint16_t b = half_rne(v);
  _Float16 x;
  std::memcpy(&x, &b, 2);
  if (std::fabs((float)x) > std::fabs(v)) --b;  // one step toward zero: f16 magnitudes order like their bit patterns
  return b;
}
float half_to_float(uint16_t b) {
  _Float16 x;
  std::memcpy(&x, &b, 2);
  return (float)x;
}

// Q (or Q^T), times 2^13, as the split-f16 B operands of rotate_rows_split_kernel: Bs[q][t][hi|lo][lane][4 dwords],
// lane (g,c) slot s = element [32 q + 8 g + s][16 t + c]
void split_rotation_operand(const double* B, int k, int KP, int T, std::vector<int32_t>& out) {
  const int KC = (T + 1) / 2;
  out.assign((size_t)KC * T * 2 * 64 * 4, 0);
  for (int q = 0; q < KC; ++q)
    for (int t = 0; t < T; ++t)
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, c = lane & 15;
        uint16_t hi[8], lo[8];
        for (int s8 = 0; s8 < 8; ++s8) {
          const int f = 32 * q + 8 * g + s8;
          const float v = f < k ? (float)(B[(size_t)f * KP + 16 * t + c] * 8192.0) : 0.f;
          hi[s8] = half_rtz(v);
          lo[s8] = half_rne(v - half_to_float(hi[s8]));
        }
        int32_t* oh = &out[(((size_t)(q * T + t) * 2 + 0) * 64 + lane) * 4];
        int32_t* ol = &out[(((size_t)(q * T + t) * 2 + 1) * 64 + lane) * 4];
        for (int e = 0; e < 4; ++e) {
          oh[e] = (int32_t)((uint32_t)hi[2 * e] | ((uint32_t)hi[2 * e + 1] << 16));
          ol[e] = (int32_t)((uint32_t)lo[2 * e] | ((uint32_t)lo[2 * e + 1] << 16));
        }
      }
}

template <int T, int TN>
int launch_dual_TN(mals_handle h, DualParams dp) {
  unsigned grid = 1;
  if (int rc = persistent_grid(h, als_dual_kernel<T, TN>, dp.n_work, &grid)) return rc;
  hipLaunchKernelGGL((als_dual_kernel<T, TN>), dim3(grid), dim3(256), 0, h->stream, dp);
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}
template <int T>
int launch_dual_T(mals_handle h, const DualParams& dp, int tn) {
  if constexpr (dual_max_blocks(T) >= 1) { if (tn == 1) return launch_dual_TN<T, 1>(h, dp); }
  if constexpr (dual_max_blocks(T) >= 2) { if (tn == 2) return launch_dual_TN<T, 2>(h, dp); }
  if constexpr (dual_max_blocks(T) >= 3) { if (tn == 3) return launch_dual_TN<T, 3>(h, dp); }
  if constexpr (dual_max_blocks(T) >= 4) { if (tn == 4) return launch_dual_TN<T, 4>(h, dp); }
  return fail(h, MALS_INVALID_ARG, "no dual kernel for this row class");
}
int launch_dual(mals_handle h, const DualParams& dp, int tn) {
  switch (h->T) {
    case 2: return launch_dual_T<2>(h, dp, tn);
    case 3: return launch_dual_T<3>(h, dp, tn);
    case 4: return launch_dual_T<4>(h, dp, tn);
    case 5: return launch_dual_T<5>(h, dp, tn);
    case 6: return launch_dual_T<6>(h, dp, tn);
    case 7: return launch_dual_T<7>(h, dp, tn);
    case 8: return launch_dual_T<8>(h, dp, tn);
  }
  return fail(h, MALS_INVALID_ARG, "unsupported feature count");
}

// Once per half-iteration, in two halves.  HOST: G (opposite side, already on its way to h_G) -> eigendecomposition
// -> everything the device needs, staged in one pinned block.  A group computes it on ONE member and hands the block
// to the others (`from`): after the all-reduce every member holds the same G, and a k x k eigenproblem solved once
// per member by the one thread that drives them all would only delay the later members' kernels.  DEVICE:
// asynchronous uploads out of the block + the rotated copy of the opposite factors.  Both run AFTER the direct
// kernels of the first chunk have been enqueued, so the host work hides under them.  Sets h->dual_ok.
double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct EigStage {  // views into mals_handle_s::eig_stage
  double* Q;
  float* Qf;
  float* lam;
  int32_t* Bs;
  size_t nQ, nLam, nBs;  // elements: 2 KP^2, 2 KP, KC T 2 64 4 per operand
};
EigStage eig_views(mals_handle h) {
  const size_t KP = (size_t)16 * h->T, KC = (size_t)(h->T + 1) / 2;
  EigStage e;
  e.nQ = 2 * KP * KP;
  e.nLam = 2 * KP;
  e.nBs = KC * h->T * 2 * 64 * 4;
  char* b = h->eig_stage;
  e.Q = reinterpret_cast<double*>(b);
  e.Qf = reinterpret_cast<float*>(b + sizeof(double) * e.nQ);
  e.lam = e.Qf + e.nQ;
  e.Bs = reinterpret_cast<int32_t*>(e.lam + e.nLam);
  return e;
}
int ensure_eig_stage(mals_handle h) {
  if (h->eig_stage) return MALS_OK;
  const size_t KP = (size_t)16 * h->T, KC = (size_t)(h->T + 1) / 2;
  const size_t bytes = sizeof(double) * 2 * KP * KP + sizeof(float) * (2 * KP * KP + 2 * KP) + sizeof(int32_t) * 2 * KC * h->T * 2 * 64 * 4;
  HIPCHK(h, hipHostMalloc(&h->eig_stage, bytes));
  h->eig_stage_bytes = bytes;
  return MALS_OK;
}

int prepare_dual_host(mals_handle h, int side, mals_handle from) {
  SideState& o = h->side[1 - side];
  const int k = h->cfg.features, KP = 16 * h->T;
  h->dual_side = side;
  h->dual_version = o.G_version;
  h->dual_ok = false;
  h->eig_ok = false;
  if (int rc = ensure_eig_stage(h)) return rc;
  // h_G <- o.G was enqueued before the direct kernels, behind everything of the previous half-iteration: once it is
  // in, the uploads that read the block last time are done as well and the block may be rewritten
  HIPCHK(h, hipEventSynchronize(h->ev_G));
  h->tl[1] = now_us();
  if (from && from != h) {  // the same G, decomposed by another member of the group
    if (from->eig_stage_bytes != h->eig_stage_bytes) return fail(h, MALS_INVALID_ARG, "members of a group must share features");
    std::memcpy(h->eig_stage, from->eig_stage, h->eig_stage_bytes);
    h->eig_ok = from->eig_ok;
    h->eig_gmax = from->eig_gmax;
    h->rotate_f64 = from->rotate_f64;
    h->rotate_split = from->rotate_split;
    h->tl[2] = now_us();
    return MALS_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<double> evals((size_t)k), V((size_t)k * k);
  const bool ok = mals::symmetric_eigen(h->h_G, k, evals.data(), V.data());
  h->stats.eigen_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  h->tl[2] = now_us();
  if (!ok) return MALS_OK;
  double lmin = evals[0], lmax = evals[0];
  for (int f = 0; f < k; ++f) {
    lmin = std::min(lmin, evals[(size_t)f]);
    lmax = std::max(lmax, evals[(size_t)f]);
  }
  const double la = h->cfg.lambda * h->cfg.alpha;
  // A_u = G + lambda alpha n_u I must be safely positive definite for every n_u >= 1 (then W_u is too,
  // and the direct path could not have flagged the row either); tiny negative eigenvalues of a
  // rank-deficient G are rounding
  if (!(lmin + la >= 1.0e-4)) return MALS_OK;
  EigStage e = eig_views(h);
  std::fill(e.Q, e.Q + e.nQ, 0.0);
  for (int f = 0; f < k; ++f)
    for (int j = 0; j < k; ++j) {
      e.Q[(size_t)f * KP + j] = V[(size_t)f * k + j];                        // forward: y' = y Q
      e.Q[(size_t)KP * KP + (size_t)f * KP + j] = V[(size_t)j * k + f];      // back: x = x' Q^T
    }
  for (int f = 0; f < KP; ++f) {
    const double l = f < k ? std::max(evals[(size_t)f], 0.0) : 0.0;
    e.lam[(size_t)f] = (float)l;
    e.lam[(size_t)KP + f] = (float)(1.0 / std::sqrt(l + la));
  }
  for (size_t i = 0; i < e.nQ; ++i) e.Qf[i] = (float)e.Q[i];
  // fp32 rotation unless the spectrum is wide enough for a rotated coordinate to be a small difference of
  // large products (dual_kernels.h, rotate_rows_kernel)
  h->rotate_f64 = (lmax + la) > 1.0e5 * (std::max(lmin, 0.0) + la);
  if (const char* ev = std::getenv("MALS_ROTATE_F64")) h->rotate_f64 = std::atoi(ev) != 0;  // tests / A-B
  h->rotate_split = k % 8 == 0 && !std::getenv("MALS_ROTATE_NO_SPLIT");
  if (h->rotate_split) {
    std::vector<int32_t> Bs;
    for (int op = 0; op < 2; ++op) {
      split_rotation_operand(e.Q + (size_t)op * KP * KP, k, KP, h->T, Bs);
      std::memcpy(e.Bs + (size_t)op * e.nBs, Bs.data(), sizeof(int32_t) * e.nBs);
    }
  }
  double gmax = 0.0;   // max |y_f| <= sqrt(max_f G_ff)
  for (int f = 0; f < k; ++f) gmax = std::max(gmax, h->h_G[(size_t)f * k + f]);
  h->eig_gmax = gmax;
  h->eig_ok = true;
  return MALS_OK;
}

int prepare_dual_device(mals_handle h, int side) {
  SideState& o = h->side[1 - side];
  const int k = h->cfg.features, KP = 16 * h->T;
  h->dual_ok = false;
  if (!h->eig_ok) return MALS_OK;
  const EigStage e = eig_views(h);
  if (!h->d_eig) {  // one device block with the layout of the pinned one: ONE upload per half-iteration instead of four
    HIPCHK(h, hipMalloc(&h->d_eig, h->eig_stage_bytes));
    h->d_Q = reinterpret_cast<double*>(h->d_eig);
    h->d_Qf = reinterpret_cast<float*>(h->d_eig + (reinterpret_cast<char*>(e.Qf) - h->eig_stage));
    h->d_lam = reinterpret_cast<float*>(h->d_eig + (reinterpret_cast<char*>(e.lam) - h->eig_stage));
    h->d_Bs = reinterpret_cast<int32_t*>(h->d_eig + (reinterpret_cast<char*>(e.Bs) - h->eig_stage));
  }
  // {z bound of the dual kernels, x' bound of the un-rotation of the even chunks, of the odd chunks}: a chunk's un-rotation
  // picks its operand scale from the bound its OWN dual kernels left -- with the chunks on two alternating streams a
  // shared slot made the scale (hence the f16 split of small components) depend on how far the other stream had got
  if (!h->d_zbound) HIPCHK(h, hipMalloc(&h->d_zbound, 3 * sizeof(unsigned)));
  const size_t need = (size_t)o.n_total * KP;
  if (h->Mr_cap < need) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    free_dev(h->d_Mr);
    h->Mr_cap = 0;
    HIPCHK(h, hipMalloc(&h->d_Mr, sizeof(float) * need));
    h->Mr_cap = need;
  }
  // pinned source, rewritten no earlier than the next half-iteration's prepare_dual_host (which waits for this stream)
  HIPCHK(h, hipMemcpyAsync(h->d_eig, h->eig_stage, h->eig_stage_bytes, hipMemcpyHostToDevice, h->stream));
  h->Bs_stride = e.nBs;
  HIPCHK(h, hipMemsetAsync(h->d_zbound, 0, 3 * sizeof(unsigned), h->stream));
  RotateParams rp;
  rp.src = o.F;
  rp.dst = h->d_Mr;
  rp.B = h->d_Q;
  rp.Bf = h->d_Qf;
  rp.items = nullptr;
  rp.dmax = h->d_lam + KP;
  rp.zbound = h->d_zbound;
  rp.n_rows = o.n_total;
  rp.k = k;
  rp.src_stride = k;
  rp.dst_stride = KP;
  rp.dst_cols = KP;
  PendingEvent pe;
  if (int rc = begin_timed(h, 5, (double)o.n_total * (4.0 * k + 4.0 * KP), pe)) return rc;
  if (h->rotate_split && !h->rotate_f64) {
    RotateSplitParams sp;
    sp.src = rp.src; sp.dst = rp.dst; sp.items = nullptr; sp.dmax = rp.dmax; sp.zbound = rp.zbound; sp.n_rows = rp.n_rows; sp.k = k;
    sp.src_stride = rp.src_stride; sp.dst_stride = rp.dst_stride; sp.dst_cols = rp.dst_cols;
    sp.Bs = reinterpret_cast<const i32x4*>(h->d_Bs);
    sp.bound_bits = nullptr;
    sp.bound_host = (float)std::sqrt(h->eig_gmax);
    if (int rc = launch_rotate_split<false>(h, sp)) return rc;
  } else if (int rc = h->rotate_f64 ? launch_rotate<false, true>(h, rp) : launch_rotate<false, false>(h, rp)) {
    return rc;
  }
  if (int rc = end_timed(h, pe)) return rc;
  h->dual_ok = true;
  return MALS_OK;
}

// the dual lists of one chunk: als_dual_kernel per row class, then x = Q x' in place
int launch_dual_chunk(mals_handle h, int side, int chunk) {
  SideState& s = h->side[side];
  const SideState::ChunkRange& cr = s.chunks[(size_t)chunk];
  const int k = h->cfg.features, KP = 16 * h->T;
  const double per = 4.0 * k + 8.0;
  DualParams dp;
  dp.col = s.col;
  dp.val = s.val;
  dp.Mr = h->d_Mr;
  dp.lam = h->d_lam;
  dp.zbound = h->d_zbound;
  dp.out = s.F + s.row_offset * k;
  dp.bad_row = h->d_bad + side;
  dp.k = k;
  dp.alpha = (float)h->cfg.alpha;
  dp.lambda_alpha = (float)(h->cfg.lambda * h->cfg.alpha);
  dp.sqrt_w_max = (float)std::sqrt(std::fabs(h->cfg.alpha) * (double)s.max_abs_val);
  dp.xbound = h->d_zbound + 1 + (chunk & 1);
  HIPCHK(h, hipMemsetAsync(dp.xbound, 0, sizeof(unsigned), h->stream));  // same stream as the chunk two back, whose un-rotation is done
  dp.any_marked = h->d_gref_state;
  dp.refine_flag = h->refine_limit > 0.f ? s.refine : nullptr;
  dp.refine_limit = h->refine_limit;
  const WorkItem* base = s.itemsA + cr.offA + cr.nA;
  int64_t off = 0;
  PendingEvent pe;
  // (Two classes in flight at a time on two streams -- one class's tail under the next one's head -- was built in round 4
  // and measured flat on c4rank, 11.37 vs 11.36 ms: the dual kernels are issue bound, not tail bound.)
  for (int cls = 3; cls >= 0; --cls) {  // descending length, as stored
    if (!cr.nD[cls]) continue;
    dp.items = base + off;
    dp.n_work = cr.nD[cls];
    if (int rc = begin_timed(h, 4, (double)cr.nnzD[cls] * per + (double)cr.nD[cls] * per, pe)) return rc;
    if (int rc = launch_dual(h, dp, cls + 1)) return rc;
    if (int rc = end_timed(h, pe)) return rc;
    off += cr.nD[cls];
  }
  RotateParams rp;
  rp.src = dp.out;
  rp.dst = dp.out;
  rp.B = h->d_Q + (size_t)KP * KP;
  rp.Bf = h->d_Qf + (size_t)KP * KP;
  rp.items = base;
  rp.dmax = nullptr;
  rp.zbound = nullptr;
  rp.n_rows = cr.n_dual();
  rp.k = k;
  rp.src_stride = k;
  rp.dst_stride = k;
  rp.dst_cols = k;
  if (int rc = begin_timed(h, 5, (double)cr.n_dual() * 8.0 * k, pe)) return rc;
  if (h->rotate_split) {
    RotateSplitParams sp;
    sp.src = rp.src; sp.dst = rp.dst; sp.items = rp.items; sp.dmax = nullptr; sp.zbound = nullptr; sp.n_rows = rp.n_rows; sp.k = k;
    sp.src_stride = k; sp.dst_stride = k; sp.dst_cols = k;
    sp.Bs = reinterpret_cast<const i32x4*>(h->d_Bs + h->Bs_stride);
    sp.bound_bits = dp.xbound;
    sp.bound_host = 0.f;
    if (int rc = launch_rotate_split<true>(h, sp)) return rc;
  } else if (int rc = launch_rotate<true, false>(h, rp)) {
    return rc;
  }
  return end_timed(h, pe);
}

int ensure_gramian_buffers(mals_handle h, SideState& s) {
  const int k = h->cfg.features;
  if (!s.G) HIPCHK(h, hipMalloc(&s.G, sizeof(double) * (size_t)k * k));
  if (!s.Gf) HIPCHK(h, hipMalloc(&s.Gf, sizeof(float) * (size_t)tri(h->T) * 256));
  if (!s.d_ymax) HIPCHK(h, hipMalloc(&s.d_ymax, sizeof(unsigned) * YMAX_SLOTS));
  return MALS_OK;
}

// DoubleWeightedMean.increment (common/src/net/myrrix/common/stats/DoubleWeightedMean.java:73-81)
void dwm_increment(double& total_weight, double& mean, double datum, double weight) {
  const double old = total_weight;
  total_weight += weight;
  if (old <= 0) {
    mean = datum;
  } else {
    mean = mean * old / total_weight + datum * weight / total_weight;
  }
}

}  // namespace

namespace {
template <int T>
int launch_reconstruction(mals_handle h, SideState& s, SideState& o, unsigned grid, double* d_sum, unsigned long long* d_cnt) {
  hipLaunchKernelGGL((reconstruction_kernel<T>), dim3(grid), dim3(256), 0, h->stream, s.row_ptr, s.col,
                     s.F + s.row_offset * h->cfg.features, o.F, h->cfg.features, s.n_local, d_sum, d_cnt);
  HIPCHK(h, hipGetLastError());
  return MALS_OK;
}
}  // namespace


namespace {
#include "topn_host.h"
}  // namespace

// ================================================================================================
extern "C" {

int mals_abi_version(void) { return MALS_ABI_VERSION; }

int mals_default_config(mals_config* cfg) {
  if (!cfg) return MALS_INVALID_ARG;
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->struct_size = (int32_t)sizeof(mals_config);
  cfg->features = 30;   // MatrixFactorizer.java:34 DEFAULT_FEATURES
  cfg->alpha = 1.0;     // ALS:71
  cfg->lambda = 0.1;    // ALS:73
  cfg->singularity_threshold = 1.0e-5;
  cfg->flags = 0;
  cfg->device = 0;
  cfg->segment_nnz = 0;
  cfg->chunk_rows = 0;
  cfg->gramian_mode = MALS_GRAMIAN_AUTO;
  cfg->solve_mode = MALS_SOLVE_AUTO;
  return MALS_OK;
}

static thread_local std::string t_create_error;
void malsi_set_create_error(const char* text) { t_create_error = text ? text : ""; }
static int create_fail(int code, const std::string& msg) {
  t_create_error = msg;
  return code;
}
int mals_create_error(char* buf, size_t cap) {
  if (buf && cap > 0) {
    const size_t n = std::min(cap - 1, t_create_error.size());
    std::memcpy(buf, t_create_error.data(), n);
    buf[n] = 0;
  }
  return (int)t_create_error.size();
}
int mals_group_create_error(char* buf, size_t cap) { return mals_create_error(buf, cap); }

int mals_create(const mals_config* cfg, mals_handle* out) {
  if (!cfg || !out) return create_fail(MALS_INVALID_ARG, "mals_create: null config or output pointer");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(mals_config))
    return create_fail(MALS_INVALID_ARG, "mals_create: mals_config.struct_size " + std::to_string(cfg->struct_size) + " is not this library's " +
                                             std::to_string(sizeof(mals_config)) + " (mals_default_config fills it)");
  if (cfg->features <= 0 || cfg->features > 128)   // ALS:139
    return create_fail(MALS_INVALID_ARG, "mals_create: features must be in 1..128, got " + std::to_string(cfg->features));
  if (!(cfg->lambda >= 0.0) || !std::isfinite(cfg->alpha)) return create_fail(MALS_INVALID_ARG, "mals_create: lambda must be >= 0 and alpha finite");
  int n_dev = 0;
  {
    const hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
      (void)hipGetLastError();
      return create_fail(MALS_HIP_ERROR, std::string("mals_create: no HIP device (hipGetDeviceCount: ") + hipGetErrorString(e) +
                                             "); a GPU is required, there is no CPU fallback");
    }
  }
  if (cfg->device < 0 || cfg->device >= n_dev)
    return create_fail(MALS_HIP_ERROR, "mals_create: device ordinal " + std::to_string(cfg->device) + " outside 0.." + std::to_string(n_dev - 1));
  mals_handle h = new (std::nothrow) mals_handle_s();
  if (!h) return create_fail(MALS_OOM, "mals_create: out of host memory");
  h->cfg = *cfg;
  h->cfg.flags &= 3;
#ifdef MALS_PROFILING
  if (const char* dbg = std::getenv("MALS_DEBUG_FLAGS")) h->cfg.flags |= (std::atoi(dbg) & 0xff) << 8;  // ablations
#endif
  if (h->cfg.segment_nnz <= 0) h->cfg.segment_nnz = 4096;
  if (h->cfg.chunk_rows < 0) h->cfg.chunk_rows = 0;
  h->cfg.segment_nnz = (h->cfg.segment_nnz + 3) & ~3;
  h->T = (cfg->features + 15) / 16;
  switch (cfg->gramian_mode) {
    case MALS_GRAMIAN_AUTO: h->split_f16 = h->T >= 2; break;  // k <= 16: one tile, the fp32 products are not the bottleneck (k = 30: split is 8 % faster, measured)
    case MALS_GRAMIAN_FP32: h->split_f16 = false; break;
    case MALS_GRAMIAN_SPLIT_F16: h->split_f16 = true; break;
    case MALS_GRAMIAN_SPLIT3_F16:
      if (h->T != 4) {
        delete h;
        return create_fail(MALS_INVALID_ARG, "mals_create: MALS_GRAMIAN_SPLIT3_F16 is built for 49..64 features");
      }
      h->split_f16 = h->split3 = true;
      break;
    default: delete h; return create_fail(MALS_INVALID_ARG, "mals_create: unknown gramian_mode");
  }
  // a negative alpha (accepted by the reference, ALS:506-509) has no real sqrt(alpha |r|): fp32 gather
  if (cfg->alpha < 0.0) h->split_f16 = false;
  switch (cfg->solve_mode) {
    case MALS_SOLVE_AUTO: h->dual_blocks = h->T >= 3 ? dual_max_blocks(h->T) : 0; break;
    case MALS_SOLVE_DIRECT: h->dual_blocks = 0; break;
    case MALS_SOLVE_DUAL: h->dual_blocks = dual_max_blocks(h->T); break;
    default: delete h; return create_fail(MALS_INVALID_ARG, "mals_create: unknown solve_mode");
  }
  if (h->cfg.flags != 0 || !(cfg->alpha > 0.0)) h->dual_blocks = 0;  // the dual path covers the default mode only
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
      h->n_cu = prop.multiProcessorCount;
  }
  std::memset(&h->stats, 0, sizeof(h->stats));
  h->stats.struct_size = (int32_t)sizeof(mals_stats);
  if (hipSetDevice(cfg->device) != hipSuccess || hipMalloc(&h->d_bad, 4 * sizeof(unsigned long long)) != hipSuccess ||
      hipHostMalloc(&h->h_bad, 4 * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc(&h->d_zscale, 4 * sizeof(float)) != hipSuccess || hipMalloc(&h->d_maxabs, 4 * sizeof(unsigned)) != hipSuccess ||
      hipMalloc(&h->d_colrange, 2 * sizeof(int)) != hipSuccess ||
      hipMalloc(&h->d_refined, sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc(&h->d_gref_state, 4 * sizeof(int)) != hipSuccess || hipMemset(h->d_gref_state, 0, 4 * sizeof(int)) != hipSuccess ||
      hipMemset(h->d_refined, 0, sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(h->d_zscale, 0, 4 * sizeof(float)) != hipSuccess ||   // (mals_get_gather_scale before any split-precision gather: zeros)
      hipMemset(h->d_bad, 0xff, 4 * sizeof(unsigned long long)) != hipSuccess) {
    const hipError_t e = hipGetLastError();
    delete h;
    return create_fail(MALS_HIP_ERROR, "mals_create: device " + std::to_string(cfg->device) + ": " + hipGetErrorString(e));
  }
  if (const char* e = std::getenv("MALS_REFINE_LIMIT")) h->refine_limit = std::max(0.f, (float)std::atof(e));
  if (hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_refine, hipEventDisableTiming) != hipSuccess) {
    delete h;
    return create_fail(MALS_HIP_ERROR, "mals_create: hipEventCreate failed");
  }
  h->lds_gather = h->cfg.features == 128;   // the LDS-staged rows / segments kernels (lds_kernels.h); MALS_LDS_GATHER=0: A/B against the register-staged ones
  if (const char* e = std::getenv("MALS_LDS_GATHER")) h->lds_gather = h->lds_gather && std::atoi(e) != 0;
  if (const char* e = std::getenv("MALS_OVERLAP")) h->overlap = std::atoi(e) != 0;
  if (h->overlap) {
    bool ok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i)
      ok = hipStreamCreateWithFlags(&h->aux[i], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) h->overlap = false;
  }
#ifdef MALS_PROFILING
  if (std::getenv("MALS_DEBUG_TRACE")) {
    (void)hipMalloc(&h->d_trace, 64 * 64 * 6 * sizeof(unsigned long long));
    (void)hipMemset(h->d_trace, 0, 64 * 64 * 6 * sizeof(unsigned long long));
  }
#endif
  t_create_error.clear();
  h->tn_front = new TopnFront();
  if (const char* e = std::getenv("MALS_TOPN_FRONT_DEPTH")) topn_front(h)->depth = std::max(1, std::min(TOPN_SLOTS, std::atoi(e)));
  if (const char* e = std::getenv("MALS_TOPN_FRONT_SPIN_US")) topn_front(h)->spin_us = std::max(0, std::atoi(e));
  *out = h;
  return MALS_OK;
}

int mals_destroy(mals_handle h) {
  if (!h) return MALS_INVALID_ARG;
  (void)hipSetDevice(h->cfg.device);
  delete topn_front(h);
  h->tn_front = nullptr;
  free_dev(h->tag_bits);
  (void)hipStreamSynchronize(h->stream);
  if (h->d_trace) {  // dump the last launch's per-phase cycle stamps (profiling aid)
    std::vector<unsigned long long> t(64 * 64 * 6);
    (void)hipMemcpy(t.data(), h->d_trace, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    for (int r = 0; r < 64; r += 21)
      for (int w = 0; w < 4; w += 1) {
        const unsigned long long* o = &t[(size_t)(r * 64 + w) * 6];
        std::fprintf(stderr, "[trace] iter %2d wave %2d len %5llu gather %7llu chol %7llu solve+store %7llu gap_to_prev_end %lld\n", r, w, o[4],
                     o[1] - o[0], o[2] - o[1], o[3] - o[2], r ? (long long)(o[0] - t[(size_t)((r - 1) * 64 + w) * 6 + 3]) : 0ll);
      }
    for (int w = 0; w < 4; ++w) {  // shader clock: s_memtime ticks per 100 MHz wall tick over 63 traced rows
      const unsigned long long* a = &t[(size_t)w * 6];
      const unsigned long long* b = &t[(size_t)(63 * 64 + w) * 6];
      if (b[5] > a[5]) std::fprintf(stderr, "[trace] wave %d: %llu cycles in %llu wall ticks (100 MHz) = %.0f MHz; %.0f cycles per row\n", w, b[3] - a[3], b[5] - a[5], 100.0 * (double)(b[3] - a[3]) / (double)(b[5] - a[5]), (double)(b[3] - a[3]) / 63.0);
    }
    free_dev(h->d_trace);
  }
  for (PendingEvent& pe : h->pending) {
    (void)hipEventDestroy(pe.a);
    (void)hipEventDestroy(pe.b);
  }
  for (int sd = 0; sd < 2; ++sd) {
    SideState& s = h->side[sd];
    free_matrix(s);
    if (s.F_owned) free_dev(s.F);
    free_dev(s.G);
    free_dev(s.Gf);
    free_dev(s.d_ymax);
    free_dev(s.partials);
  }
  free_dev(h->d_bad);
  free_dev(h->d_refined);
  free_dev(h->d_gref_state);
  free_dev(h->d_Gref);
  free_dev(h->d_gref_part);
  free_dev(h->d_zscale);
  free_dev(h->d_Gperm);
  if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
  if (h->ev_refine) (void)hipEventDestroy(h->ev_refine);
  free_dev(h->d_maxabs);
  free_dev(h->d_colrange);
  free_dev(h->d_Mr);
  free_dev(h->d_Mp);
  free_dev(h->d_eig);  // d_Q, d_Qf, d_lam, d_Bs are views into it
  free_dev(h->d_zbound);
  if (h->h_G) (void)hipHostFree(h->h_G);
  if (h->eig_stage) (void)hipHostFree(h->eig_stage);
  for (int i = 0; i < 2; ++i) {
    if (h->aux[i]) (void)hipStreamSynchronize(h->aux[i]);
    if (h->aux[i]) (void)hipStreamDestroy(h->aux[i]);
    if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_G) (void)hipEventDestroy(h->ev_G);
  if (h->h_bad) (void)hipHostFree(h->h_bad);
  free_dev(h->d_idx);
  free_dev(h->d_rows);
  free_dev(h->d_sd);
  free_dev(h->d_sd_idx);
  topn_free(h);
  free_dev(h->known_ptr_own);
  free_dev(h->known_idx_own);
  delete h;
  return MALS_OK;
}

int mals_features(mals_handle h) { return h ? h->cfg.features : 0; }

const char* mals_last_error(mals_handle h) { return h ? h->err.c_str() : "null handle"; }

int mals_set_stream(mals_handle h, void* hip_stream) {
  if (!h) return MALS_INVALID_ARG;
  h->stream = (hipStream_t)hip_stream;
  return MALS_OK;
}

int mals_set_factor_rows(mals_handle h, int side, int64_t n_rows_total) {
  CHECK_SIDE(h, side);
  if (n_rows_total <= 0) return fail(h, MALS_INVALID_ARG, "n_rows_total must be positive");
  if (int rc = use_device(h)) return rc;
  SideState& s = h->side[side];
  if (s.F_owned) free_dev(s.F);
  s.F = nullptr;
  const size_t bytes = sizeof(float) * (size_t)n_rows_total * (size_t)h->cfg.features;
  HIPCHK(h, hipMalloc(&s.F, bytes));
  HIPCHK(h, hipMemsetAsync(s.F, 0, bytes, h->stream));
  s.F_owned = true;
  s.n_total = n_rows_total;
  s.G_valid = false;
  ++s.F_epoch;
  return MALS_OK;
}

int mals_bind_factors(mals_handle h, int side, float* device_ptr, int64_t n_rows_total) {
  CHECK_SIDE(h, side);
  if (!device_ptr || n_rows_total <= 0) return fail(h, MALS_INVALID_ARG, "null buffer or non-positive row count");
  if (int rc = use_device(h)) return rc;
  SideState& s = h->side[side];
  if (s.F_owned) free_dev(s.F);
  s.F = device_ptr;
  s.F_owned = false;
  s.n_total = n_rows_total;
  s.G_valid = false;
  ++s.F_epoch;
  return MALS_OK;
}

int mals_factor_device_ptr(mals_handle h, int side, void** out_device_ptr, int64_t* out_n_rows_total) {
  CHECK_SIDE(h, side);
  if (out_device_ptr) *out_device_ptr = h->side[side].F;
  if (out_n_rows_total) *out_n_rows_total = h->side[side].n_total;
  return h->side[side].F ? MALS_OK : fail(h, MALS_INVALID_ARG, "factor replica not allocated");
}

int mals_set_matrix(mals_handle h, int side, int64_t row_offset, int64_t n_rows_local, int64_t nnz, const int64_t* row_ptr,
                    const int32_t* col_idx, const float* val, int mem_kind) {
  CHECK_SIDE(h, side);
  if (row_offset < 0 || n_rows_local < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!col_idx || !val)))
    return fail(h, MALS_INVALID_ARG, "bad matrix arguments");
  if (n_rows_local > std::numeric_limits<int32_t>::max())
    return fail(h, MALS_INVALID_ARG, "at most 2^31-1 local rows per handle");
  if (int rc = use_device(h)) return rc;
  SideState& s = h->side[side];
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_matrix(s);
  if (side == MALS_SIDE_X) clear_known_items(h);  // they were a CSR over the OLD local rows (and maybe borrowed arrays)
  s.row_offset = row_offset;
  s.n_local = n_rows_local;
  s.nnz = nnz;
  s.h_row_ptr.resize((size_t)n_rows_local + 1);
  if (mem_kind == MALS_MEM_DEVICE) {
    HIPCHK(h, hipMemcpy(s.h_row_ptr.data(), row_ptr, sizeof(int64_t) * (size_t)(n_rows_local + 1), hipMemcpyDeviceToHost));
    s.row_ptr = const_cast<int64_t*>(row_ptr);
    s.col = const_cast<int32_t*>(col_idx);
    s.val = const_cast<float*>(val);
    s.m_owned = false;
  } else if (mem_kind == MALS_MEM_HOST) {
    std::memcpy(s.h_row_ptr.data(), row_ptr, sizeof(int64_t) * (size_t)(n_rows_local + 1));
    s.m_owned = true;
    HIPCHK(h, hipMalloc(&s.row_ptr, sizeof(int64_t) * (size_t)(n_rows_local + 1)));
    HIPCHK(h, hipMemcpy(s.row_ptr, row_ptr, sizeof(int64_t) * (size_t)(n_rows_local + 1), hipMemcpyHostToDevice));
    HIPCHK(h, hipMalloc(&s.col, sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)));
    HIPCHK(h, hipMalloc(&s.val, sizeof(float) * (size_t)std::max<int64_t>(nnz, 1)));
    if (nnz) {
      HIPCHK(h, hipMemcpy(s.col, col_idx, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice));
      HIPCHK(h, hipMemcpy(s.val, val, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice));
    }
  } else {
    return fail(h, MALS_INVALID_ARG, "mem_kind must be MALS_MEM_HOST or MALS_MEM_DEVICE");
  }
  if (int rc = validate_matrix(h, side)) {
    free_matrix(s);
    return rc;
  }
  if (int rc = build_work_lists(h, s)) {
    free_matrix(s);
    return rc;
  }
  if (int rc = validate_columns(h, side)) {
    free_matrix(s);
    return rc;
  }
  s.has_matrix = true;
  return MALS_OK;
}

int mals_begin_matrix(mals_handle h, int side, int64_t row_offset, int64_t n_rows_local, int64_t nnz) {
  CHECK_SIDE(h, side);
  if (row_offset < 0 || n_rows_local < 0 || nnz < 0) return fail(h, MALS_INVALID_ARG, "bad matrix arguments");
  if (n_rows_local > std::numeric_limits<int32_t>::max())
    return fail(h, MALS_INVALID_ARG, "at most 2^31-1 local rows per handle");
  if (int rc = use_device(h)) return rc;
  SideState& s = h->side[side];
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_matrix(s);
  if (side == MALS_SIDE_X) clear_known_items(h);
  s.row_offset = row_offset;
  s.n_local = n_rows_local;
  s.nnz = nnz;
  s.h_row_ptr.assign((size_t)n_rows_local + 1, 0);
  s.m_owned = true;
  HIPCHK(h, hipMalloc(&s.row_ptr, sizeof(int64_t) * (size_t)(n_rows_local + 1)));
  HIPCHK(h, hipMalloc(&s.col, sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)));
  HIPCHK(h, hipMalloc(&s.val, sizeof(float) * (size_t)std::max<int64_t>(nnz, 1)));
  s.append_rows = 0;
  s.append_nnz = 0;
  s.appending = true;
  return MALS_OK;
}

int mals_append_rows(mals_handle h, int side, int64_t n_rows, const int64_t* row_ptr_chunk, const int32_t* col_idx,
                     const float* val) {
  CHECK_SIDE(h, side);
  SideState& s = h->side[side];
  if (!s.appending) return fail(h, MALS_INVALID_ARG, "mals_append_rows without mals_begin_matrix");
  if (n_rows < 0 || !row_ptr_chunk || row_ptr_chunk[0] != 0) return fail(h, MALS_INVALID_ARG, "bad chunk");
  const int64_t cn = row_ptr_chunk[n_rows];
  if (s.append_rows + n_rows > s.n_local || s.append_nnz + cn > s.nnz || (cn > 0 && (!col_idx || !val)))
    return fail(h, MALS_INVALID_ARG, "chunk exceeds the declared matrix size");
  if (int rc = use_device(h)) return rc;
  for (int64_t r = 0; r < n_rows; ++r) {
    if (row_ptr_chunk[r + 1] < row_ptr_chunk[r]) return fail(h, MALS_INVALID_ARG, "row_ptr must be non-decreasing");
    s.h_row_ptr[(size_t)(s.append_rows + r + 1)] = s.append_nnz + row_ptr_chunk[r + 1];
  }
  if (cn) {
    HIPCHK(h, hipMemcpy(s.col + s.append_nnz, col_idx, sizeof(int32_t) * (size_t)cn, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(s.val + s.append_nnz, val, sizeof(float) * (size_t)cn, hipMemcpyHostToDevice));
  }
  s.append_rows += n_rows;
  s.append_nnz += cn;
  return MALS_OK;
}

int mals_end_matrix(mals_handle h, int side) {
  CHECK_SIDE(h, side);
  SideState& s = h->side[side];
  if (!s.appending) return fail(h, MALS_INVALID_ARG, "mals_end_matrix without mals_begin_matrix");
  s.appending = false;
  if (s.append_rows != s.n_local || s.append_nnz != s.nnz) {
    free_matrix(s);
    return fail(h, MALS_INVALID_ARG, "appended rows/entries do not match the declared matrix size");
  }
  if (int rc = use_device(h)) return rc;
  HIPCHK(h, hipMemcpy(s.row_ptr, s.h_row_ptr.data(), sizeof(int64_t) * (size_t)(s.n_local + 1), hipMemcpyHostToDevice));
  if (int rc = build_work_lists(h, s)) {
    free_matrix(s);
    return rc;
  }
  if (int rc = validate_columns(h, side)) {
    free_matrix(s);
    return rc;
  }
  s.has_matrix = true;
  return MALS_OK;
}

int mals_get_value_bound(mals_handle h, int side, float* max_abs_value) {
  CHECK_SIDE(h, side);
  if (!h->side[side].has_matrix || !max_abs_value) return fail(h, MALS_INVALID_ARG, "matrix of this side not set");
  *max_abs_value = h->side[side].max_abs_val;
  return MALS_OK;
}

int mals_set_value_bound(mals_handle h, int side, float max_abs_value) {
  CHECK_SIDE(h, side);
  SideState& s = h->side[side];
  if (!s.has_matrix) return fail(h, MALS_INVALID_ARG, "matrix of this side not set");
  if (!(max_abs_value >= s.local_max_abs_val)) return fail(h, MALS_INVALID_ARG, "bound below the largest local |value|");
  s.max_abs_val = max_abs_value;
  return MALS_OK;
}

int mals_get_value_stats(mals_handle h, int side, float* max_abs_value, double* sum_abs_value, int64_t* n_values) {
  CHECK_SIDE(h, side);
  const SideState& s = h->side[side];
  if (!s.has_matrix) return fail(h, MALS_INVALID_ARG, "matrix of this side not set");
  if (max_abs_value) *max_abs_value = s.local_max_abs_val;
  if (sum_abs_value) *sum_abs_value = s.local_mean_abs_val * (double)s.nnz;
  if (n_values) *n_values = s.nnz;
  return MALS_OK;
}

int mals_set_value_stats(mals_handle h, int side, float max_abs_value, double mean_abs_value) {
  CHECK_SIDE(h, side);
  if (!(mean_abs_value >= 0.0) || !std::isfinite(mean_abs_value)) return fail(h, MALS_INVALID_ARG, "mean |value| must be finite and >= 0");
  if (int rc = mals_set_value_bound(h, side, max_abs_value)) return rc;
  h->side[side].mean_abs_val = mean_abs_value;
  return MALS_OK;
}

int mals_set_factors(mals_handle h, int side, int64_t row_begin, int64_t n_rows, const float* host_rows) {
  CHECK_SIDE(h, side);
  SideState& s = h->side[side];
  if (!s.F) return fail(h, MALS_INVALID_ARG, "factor replica not allocated");
  if (row_begin < 0 || n_rows < 0 || row_begin + n_rows > s.n_total || !host_rows)
    return fail(h, MALS_INVALID_ARG, "row range outside the factor replica");
  if (int rc = use_device(h)) return rc;
  const int k = h->cfg.features;
  HIPCHK(h, hipMemcpyAsync(s.F + row_begin * k, host_rows, sizeof(float) * (size_t)n_rows * k, hipMemcpyHostToDevice,
                           h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  s.G_valid = false;
  ++s.F_epoch;
  return MALS_OK;
}

int mals_get_factors(mals_handle h, int side, int64_t row_begin, int64_t n_rows, float* host_out) {
  CHECK_SIDE(h, side);
  SideState& s = h->side[side];
  if (!s.F) return fail(h, MALS_INVALID_ARG, "factor replica not allocated");
  if (row_begin < 0 || n_rows < 0 || row_begin + n_rows > s.n_total || !host_out)
    return fail(h, MALS_INVALID_ARG, "row range outside the factor replica");
  if (int rc = use_device(h)) return rc;
  const int k = h->cfg.features;
  HIPCHK(h, hipMemcpyAsync(host_out, s.F + row_begin * k, sizeof(float) * (size_t)n_rows * k, hipMemcpyDeviceToHost,
                           h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MALS_OK;
}

int mals_get_rows(mals_handle h, int side, const int64_t* row_idx, int32_t n, float* host_out) {
  CHECK_SIDE(h, side);
  SideState& s = h->side[side];
  if (!s.F) return fail(h, MALS_INVALID_ARG, "factor replica not allocated");
  if (n < 0 || (n > 0 && (!row_idx || !host_out))) return fail(h, MALS_INVALID_ARG, "bad row list");
  if (n == 0) return MALS_OK;
  for (int i = 0; i < n; ++i)
    if (row_idx[i] < 0 || row_idx[i] >= s.n_total) return fail(h, MALS_INVALID_ARG, "row index outside the factor replica");
  if (int rc = use_device(h)) return rc;
  const int k = h->cfg.features;
  if (h->idx_cap < n) {
    free_dev(h->d_idx);
    free_dev(h->d_rows);
    HIPCHK(h, hipMalloc(&h->d_idx, sizeof(int64_t) * (size_t)n));
    HIPCHK(h, hipMalloc(&h->d_rows, sizeof(float) * (size_t)n * k));
    h->idx_cap = n;
  }
  HIPCHK(h, hipMemcpyAsync(h->d_idx, row_idx, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, h->stream, s.F, h->d_idx, n, k,
                     h->d_rows);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(host_out, h->d_rows, sizeof(float) * (size_t)n * k, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MALS_OK;
}

int mals_gramian(mals_handle h, int side, double* host_G) {
  CHECK_SIDE(h, side);
  SideState& s = h->side[side];
  if (!s.F) return fail(h, MALS_INVALID_ARG, "factor replica not allocated");  // MU:220-222 null/empty M
  if (int rc = use_device(h)) return rc;
  if (int rc = ensure_gramian_buffers(h, s)) return rc;
  PendingEvent pe;
  // only the split-f16 kernel (large matrices) records the maximum; below its threshold not even the two memsets are
  // spent (C2 is 50 launches of a few microseconds each)
  const bool with_max = s.n_total >= GRAMIAN_SPLIT_MIN_ROWS;
  if (with_max) HIPCHK(h, hipMemsetAsync(s.d_ymax, 0, sizeof(unsigned) * YMAX_SLOTS, h->stream));
  if (int rc = begin_timed(h, 3, (double)s.n_total * 4.0 * h->cfg.features, pe)) return rc;
  if (int rc = launch_gramian(h, s, s.F, s.n_total, s.G, s.Gf, with_max ? s.d_ymax : nullptr)) return rc;
  if (int rc = end_timed(h, pe)) return rc;
  s.G_valid = true;
  ++s.G_version;
  s.ymax_version = with_max ? s.G_version : 0;   // every element of the replica went through the kernel: its maximum is exact
  if (host_G) {
    const int k = h->cfg.features;
    HIPCHK(h, hipMemcpyAsync(host_G, s.G, sizeof(double) * (size_t)k * k, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return MALS_OK;
}

int malsi_gramian_partial(mals_handle h, int side, int64_t row_begin, int64_t n_rows, double* device_out, unsigned* device_max) {
  CHECK_SIDE(h, side);
  SideState& s = h->side[side];
  if (!s.F) return fail(h, MALS_INVALID_ARG, "factor replica not allocated");
  if (row_begin < 0 || n_rows <= 0 || row_begin + n_rows > s.n_total || !device_out)
    return fail(h, MALS_INVALID_ARG, "row range outside the factor replica");
  if (int rc = use_device(h)) return rc;
  PendingEvent pe;
  if (int rc = begin_timed(h, 3, (double)n_rows * 4.0 * h->cfg.features, pe)) return rc;
  if (int rc = launch_gramian(h, s, s.F + row_begin * h->cfg.features, n_rows, device_out, nullptr, device_max)) return rc;
  return end_timed(h, pe);
}

int mals_gramian_partial(mals_handle h, int side, int64_t row_begin, int64_t n_rows, double* device_out) {
  return malsi_gramian_partial(h, side, row_begin, n_rows, device_out, nullptr);
}

// device_max: YMAX_SLOTS device words (malsi_ymax_slots()) whose maximum is the bit pattern of max |element| over ALL rows the installed G was formed from (the
// group's all-reduced maximum of the members' malsi_gramian_partial maxima), or NULL when the caller has none
int malsi_set_gramian(mals_handle h, int side, const double* G, int mem_kind, const unsigned* device_max) {
  CHECK_SIDE(h, side);
  if (!G) return fail(h, MALS_INVALID_ARG, "null Gramian");
  SideState& s = h->side[side];
  if (int rc = use_device(h)) return rc;
  if (int rc = ensure_gramian_buffers(h, s)) return rc;
  const int k = h->cfg.features;
  HIPCHK(h, hipMemcpyAsync(s.G, G, sizeof(double) * (size_t)k * k,
                           mem_kind == MALS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(gramian_pack_kernel, dim3((unsigned)tri(h->T)), dim3(256), 0, h->stream, s.G, k, h->T, s.Gf);
  HIPCHK(h, hipGetLastError());
  if (device_max) HIPCHK(h, hipMemcpyAsync(s.d_ymax, device_max, sizeof(unsigned) * YMAX_SLOTS, hipMemcpyDeviceToDevice, h->stream));
  if (mem_kind != MALS_MEM_DEVICE) HIPCHK(h, hipStreamSynchronize(h->stream));
  s.G_valid = true;
  ++s.G_version;
  s.ymax_version = device_max ? s.G_version : 0;
  return MALS_OK;
}

int mals_set_gramian(mals_handle h, int side, const double* G, int mem_kind) { return malsi_set_gramian(h, side, G, mem_kind, nullptr); }

// fork: the side streams start behind everything enqueued on the main stream so far; join: the main stream continues
// behind whatever was put on them.  A scope swaps h->stream so that every launch helper (and its timing events) lands
// on the side stream.
struct SideStream {
  mals_handle h;
  hipStream_t saved;
  SideStream(mals_handle hh, int i) : h(hh), saved(hh->stream) {
    if (h->overlap && h->forked[i]) h->stream = h->aux[i];
  }
  ~SideStream() { h->stream = saved; }
};
static int fork_streams(mals_handle h) {
  if (!h->overlap) return MALS_OK;
  HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
  for (int i = 0; i < 2; ++i) {
    HIPCHK(h, hipStreamWaitEvent(h->aux[i], h->ev_fork, 0));
    h->forked[i] = true;
  }
  return MALS_OK;
}
static int join_streams(mals_handle h) {
  if (!h->overlap) return MALS_OK;
  for (int i = 0; i < 2; ++i) {
    if (!h->forked[i]) continue;
    HIPCHK(h, hipEventRecord(h->ev_join[i], h->aux[i]));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[i], 0));
    h->forked[i] = false;
  }
  return MALS_OK;
}

// phase: the whole of every chunk (ALL); or, for callers that drive several handles from one thread (mals_group.cpp),
// one chunk in two calls -- BEGIN enqueues everything that does not need the eigendecomposition of the dual path and
// returns (h->dual_pending says whether anything is left), the caller provides the decomposition (prepare_dual_host,
// once per group), END enqueues the rest.  BEGIN + host + END = ALL.
enum { SOLVE_ALL = 0, SOLVE_BEGIN = 1, SOLVE_END = 2 };

static int solve_chunks(mals_handle h, int side, int chunk_begin, int chunk_end, int phase = SOLVE_ALL) {
  SideState& s = h->side[side];
  SideState& o = h->side[1 - side];
  if (!s.has_matrix) return fail(h, MALS_INVALID_ARG, "matrix of this side not set");
  if (!s.F || !o.F) return fail(h, MALS_INVALID_ARG, "factor replicas not allocated");
  if (s.row_offset + s.n_local > s.n_total) return fail(h, MALS_INVALID_ARG, "matrix rows exceed the factor replica");
  if (int rc = validate_columns(h, side)) return rc;
  // the split-precision gather takes its scale from the Gramian's diagonal even when W does not start from G
  const bool use_g = !(h->cfg.flags & MALS_FLAG_LOSS_IGNORES_UNSPECIFIED) || h->split_f16;
  if (use_g && !o.G_valid) return fail(h, MALS_INVALID_ARG, "Gramian of the opposite side not computed");
  if (chunk_begin < 0 || chunk_end > (int)s.chunks.size() || chunk_begin >= chunk_end)
    return fail(h, MALS_INVALID_ARG, "chunk index out of range");
  const bool resume = phase == SOLVE_END;
  if (resume) {
    if (!h->dual_pending) return MALS_OK;  // BEGIN did the whole chunk
    if (h->dual_pending_side != side || h->dual_pending_chunk != chunk_begin || chunk_end != chunk_begin + 1)
      return fail(h, MALS_INVALID_ARG, "solve END does not match the pending BEGIN");
  } else {
    h->dual_pending = false;  // a BEGIN whose END never came belongs to a half-iteration that was abandoned on an error
  }
  if (int rc = use_device(h)) return rc;
  if (s.n_local == 0) return MALS_OK;
  const int k = h->cfg.features;
  SolveParams p;
  p.row_ptr = s.row_ptr;
  p.col = s.col;
  p.val = s.val;
  p.M = o.F;
  p.ldm = k;
  if (!resume) {
    // A new half-iteration?  The opposite factors do not change while a half-iteration's chunks are solved (in any
    // order): "new" = the side, the opposite Gramian or the opposite uploads changed, or a chunk comes round again.
    bool new_half = h->pad_side != side || h->pad_version != o.G_version || h->pad_epoch != o.F_epoch ||
                    h->pad_done.size() != s.chunks.size();
    for (int c = chunk_begin; c < chunk_end && !new_half; ++c) new_half = h->pad_done[(size_t)c] != 0;
    if (new_half) {
      h->pad_side = side;
      h->pad_version = o.G_version;
      h->pad_epoch = o.F_epoch;
      h->pad_done.assign(s.chunks.size(), 0);
      h->tl[0] = now_us();
      h->tl[1] = h->tl[2] = 0.0;
      // nothing marked yet, no reference-rounded Gramian yet, no arrival counted (gramian_ref_kernel)
      HIPCHK(h, hipMemsetAsync(h->d_gref_state, 0, 3 * sizeof(int), h->stream));
    }
    for (int c = chunk_begin; c < chunk_end; ++c) h->pad_done[(size_t)c] = 1;
    if (k % 16 != 0) {
      const int ld = 16 * h->T;
      const size_t need = (size_t)o.n_total * (size_t)ld;
      if (need > h->Mp_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        free_dev(h->d_Mp);
        h->Mp_cap = 0;
        new_half = true;   // the copy is gone
        HIPCHK(h, hipMalloc(&h->d_Mp, sizeof(float) * need));
        h->Mp_cap = need;
      }
      if (new_half) {
        PendingEvent pe;
        if (int rc = begin_timed(h, 5, (double)o.n_total * 4.0 * (k + ld), pe)) return rc;
        const int64_t n4 = o.n_total * (ld / 4);
        const unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, (int64_t)h->n_cu * 32);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, h->stream, o.F, o.n_total, k, ld, h->d_Mp);
        HIPCHK(h, hipGetLastError());
        if (int rc = end_timed(h, pe)) return rc;
      }
    }
  }
  if (k % 16 != 0) {
    p.M = h->d_Mp;
    p.ldm = 16 * h->T;
  }
  p.Gf = o.Gf;
  p.Gperm = nullptr;
  if (h->lds_gather && h->split_f16 && k == 128) {
    if (!h->d_Gperm) HIPCHK(h, hipMalloc(&h->d_Gperm, sizeof(float) * (size_t)tri(8) * 256));
    if (!resume && (h->gperm_side != side || h->gperm_version != o.G_version)) {
      if (h->cfg.flags & MALS_FLAG_LOSS_IGNORES_UNSPECIFIED) {  // W does not start from G (ALS:524-539): an image of zeros
        HIPCHK(h, hipMemsetAsync(h->d_Gperm, 0, sizeof(float) * (size_t)tri(8) * 256, h->stream));
      } else {
        hipLaunchKernelGGL(gramian_perm_kernel, dim3((unsigned)tri(8)), dim3(256), 0, h->stream, o.G, k, h->d_Gperm);
        HIPCHK(h, hipGetLastError());
      }
      h->gperm_side = side;
      h->gperm_version = o.G_version;
    }
    p.Gperm = h->d_Gperm;
  }
  p.out = s.F + s.row_offset * k;
  p.items = nullptr;
  p.rowsC = s.rowsC;
  p.scratch = s.scratch;
  p.bad_row = h->d_bad + side;
  p.suspect = h->d_bad + 2 + side;
  p.any_marked = h->d_gref_state;
  p.refine_flag = nullptr;
  // under lossIgnoresUnspecified W has no Gramian under it and every marked row goes to the fp64 restatement, which
  // also reproduces that mode's fp32-rounded products: the estimate is at its weakest there (measured 100x and more
  // below cond(W) for rows with about as many entries as features), so the bar is lower
  p.refine_limit = (h->cfg.flags & MALS_FLAG_LOSS_IGNORES_UNSPECIFIED) ? h->refine_limit / 16.f : h->refine_limit;
  p.gramian_weight = 0.25f;
  if (h->cfg.flags & MALS_FLAG_RECONSTRUCT_R) {   // no confidence weights: W = G + rho I (or the plain sum of y y^T)
    p.gramian_weight = 1.f;
    p.refine_limit *= 0.25f;
  }
  {
    if (s.refine_cap < (size_t)s.n_local) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      free_dev(s.refine);
      s.refine_cap = 0;
      HIPCHK(h, hipMalloc(&s.refine, (size_t)s.n_local));
      s.refine_cap = (size_t)s.n_local;
    }
    p.refine_flag = s.refine;
  }
  p.n_work = 0;
  p.trace = h->d_trace;
  p.trace_start = 0;
  if (const char* ts = std::getenv("MALS_DEBUG_TRACE")) p.trace_start = std::atoi(ts);
  if (const char* ts = std::getenv("MALS_DEBUG_TRACE_SIDE")) if (std::atoi(ts) != side) p.trace = nullptr;
  p.k = k;
  p.flags = h->cfg.flags;
  p.alpha = (float)h->cfg.alpha;
  p.lambda_alpha = (float)(h->cfg.lambda * h->cfg.alpha);  // ALS:435
  p.sing_threshold = (float)h->cfg.singularity_threshold;
  p.zscale = h->d_zscale;
  if (!resume && h->split_f16 && (h->zs_side != side || h->zs_version != o.G_version || h->zs_bound != s.max_abs_val || h->zs_mean != s.mean_abs_val)) {
    // once per half-iteration, not per chunk: the scale only depends on G and on the value bound
    const double base_w = (h->cfg.flags & MALS_FLAG_LOSS_IGNORES_UNSPECIFIED) ? 1.0 : 0.0;
    const double w_max = base_w + ((h->cfg.flags & MALS_FLAG_RECONSTRUCT_R) ? 0.0 : std::fabs(h->cfg.alpha) * (double)s.max_abs_val);
    const double w_mean = base_w + ((h->cfg.flags & MALS_FLAG_RECONSTRUCT_R) ? 0.0 : std::fabs(h->cfg.alpha) * s.mean_abs_val);
    int force = -1;  // MALS_FORCE_RANGE_FLAG=0/1 (tests): override the range decision
    if (const char* e = std::getenv("MALS_FORCE_RANGE_FLAG")) force = std::atoi(e) != 0;
    hipLaunchKernelGGL(gather_scale_kernel, dim3(1), dim3(64), 0, h->stream, o.G, k, (float)std::sqrt(w_max), (float)std::sqrt(w_mean),
                       (double)o.n_total, force, (o.ymax_version != 0 && o.ymax_version == o.G_version) ? o.d_ymax : nullptr, h->d_zscale);
    HIPCHK(h, hipGetLastError());
    h->zs_side = side;
    h->zs_version = o.G_version;
    h->zs_bound = s.max_abs_val;
    h->zs_mean = s.mean_abs_val;
  }
  // dual path (dual_kernels.h): the reference's default mode only; the Gramian goes to the host first so
  // that its eigendecomposition runs while the direct kernels of the first chunk execute
  // ... and only when it pays: the rotation of the gathered matrix costs ~1/30 of what a dual row saves
  // (measured, k = 64 and 128), so a side with few short rows against a huge opposite matrix (the item half of
  // C5: 1.25M long rows per rank gathering from 100M users) stays on the direct kernels
  const bool dual_pays = h->cfg.solve_mode == MALS_SOLVE_DUAL || s.n_dual_rows * 32 >= o.n_total;
  const bool want_dual = s.n_dual_rows > 0 && dual_pays && h->cfg.flags == 0 && h->cfg.alpha > 0.0 && o.G_valid;
  const bool dual_stale = resume || (want_dual && (h->dual_side != side || h->dual_version != o.G_version));
  if (dual_stale && !resume) {
    if (!h->h_G) HIPCHK(h, hipHostMalloc(&h->h_G, sizeof(double) * (size_t)k * k));
    if (!h->ev_G) HIPCHK(h, hipEventCreateWithFlags(&h->ev_G, hipEventDisableTiming));
    HIPCHK(h, hipMemcpyAsync(h->h_G, o.G, sizeof(double) * (size_t)k * k, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipEventRecord(h->ev_G, h->stream));
  }
  if (phase == SOLVE_BEGIN && chunk_end != chunk_begin + 1) return fail(h, MALS_INVALID_ARG, "solve BEGIN takes one chunk");
  // everything a half-iteration sets up once is enqueued by now unless the dual preparation is still to come (below)
  if (!resume && !(want_dual && dual_stale)) HIPCHK(h, hipEventRecord(h->ev_ready, h->stream));
  const int64_t want_chunk_rows = s.chunk_rows_override >= 0 ? s.chunk_rows_override : h->cfg.chunk_rows;
  const int64_t rows_per_chunk = want_chunk_rows > 0 ? want_chunk_rows : std::max<int64_t>(s.n_local, 1);  // as build_work_lists
  for (int c = chunk_begin; c < chunk_end; ++c) {
    const SideState::ChunkRange& cr = s.chunks[(size_t)c];
    const int64_t row0 = std::min<int64_t>(s.n_local, (int64_t)c * rows_per_chunk);
    const int64_t row1 = std::min<int64_t>(s.n_local, (int64_t)(c + 1) * rows_per_chunk);
    if (!resume && p.refine_flag && row1 > row0) HIPCHK(h, hipMemsetAsync(s.refine + row0, 0, (size_t)(row1 - row0), h->stream));
    // (per chunk, not per call: once the first chunk of a multi-chunk call has prepared the dual path the later ones use it)
    bool dual_now = want_dual && h->dual_ok && h->dual_side == side && h->dual_version == o.G_version;
    if (!resume)
      if (int rc = fork_streams(h)) return rc;
    if (want_dual && dual_stale && c == chunk_begin) {
      if (!resume) {
        {
          SideStream long_rows(h, 0);
          if (int rc = launch_solve(h, s, p, c, LISTS_LONG)) return rc;
        }
        if (int rc = launch_solve(h, s, p, c, LISTS_ROWS)) return rc;   // direct lists first ...
        if (phase == SOLVE_BEGIN) {                                     // ... the caller decomposes G under them
          h->dual_pending = true;
          h->dual_pending_side = side;
          h->dual_pending_chunk = c;
          return MALS_OK;
        }
        if (int rc = prepare_dual_host(h, side, nullptr)) return rc;    // ... host eigendecomposition under them
      } else if (h->dual_side != side || h->dual_version != o.G_version) {
        return fail(h, MALS_INVALID_ARG, "solve END without the eigendecomposition of this half-iteration");
      }
      h->dual_pending = false;
      {
        SideStream dual_rows(h, 1);
        if (int rc = prepare_dual_device(h, side)) return rc;
        HIPCHK(h, hipEventRecord(h->ev_ready, h->stream));   // (with MALS_OVERLAP: the side stream the rotation is on)
        dual_now = h->dual_ok;
        if (dual_now) {
          if (int rc = launch_dual_chunk(h, side, c)) return rc;
        } else if (cr.n_dual()) {  // does not qualify: the dual lists through the direct kernel after all
          if (int rc = launch_solve(h, s, p, c, LISTS_DUAL_ROWS)) return rc;
        }
      }
    } else {
      {
        SideStream long_rows(h, 0);
        if (int rc = launch_solve(h, s, p, c, LISTS_LONG)) return rc;
      }
      if (dual_now && cr.n_dual()) {
        SideStream dual_rows(h, 1);
        if (int rc = launch_dual_chunk(h, side, c)) return rc;
      }
      if (int rc = launch_solve(h, s, p, c, dual_now ? LISTS_ROWS : (LISTS_ROWS | LISTS_DUAL_ROWS))) return rc;
    }
    if (int rc = join_streams(h)) return rc;
    if (p.refine_flag && row1 > row0) {  // behind every kernel of the chunk (the un-rotation of the dual rows included)
      if (h->refine_recorded) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_refine, 0));   // one chunk's block at a time
      if (!h->d_Gref) {   // before the parameter block below takes the pointers
        const size_t kp2 = (size_t)(16 * h->T) * (size_t)(16 * h->T);
        HIPCHK(h, hipMalloc(&h->d_Gref, sizeof(double) * kp2));
        HIPCHK(h, hipMalloc(&h->d_gref_part, sizeof(double) * kp2 * (size_t)h->n_cu * 2));
      }
      RefineParams q;
      q.p = p;
      q.G = o.G;
      q.Gref = h->d_Gref;
      q.gref_state = h->d_gref_state;
      q.n_refined = h->d_refined;
      q.row_begin = row0;
      q.row_end = row1;
      q.alpha = h->cfg.alpha;
      q.lambda_alpha = h->cfg.lambda * h->cfg.alpha;
      const bool no_gramian = h->cfg.flags & MALS_FLAG_LOSS_IGNORES_UNSPECIFIED;
      if (!no_gramian)   // something marked in this half-iteration: M^T M once more, rounded like the reference's
        if (int rc = launch_gramian_ref(h, o)) return rc;
      if (!no_gramian && h->refine_limit > 0.f)
        if (int rc = launch_refine(h, q)) return rc;      // marks 1; may raise a mark to 2
      if (int rc = launch_exact(h, q, no_gramian ? 1 : 2)) return rc;
      HIPCHK(h, hipEventRecord(h->ev_refine, h->stream));
      h->refine_recorded = true;
    }
    h->stats.rows_solved += cr.nA + cr.nC + cr.n_dual() + cr.nZ;
    h->stats.nnz_gathered += cr.nnzA + cr.nnzB + cr.nnz_dual();
    if (dual_now) h->stats.rows_dual += cr.n_dual();
  }
  // this side's factors changed: its Gramian is stale.  (The OPPOSITE side's Gramian stays valid
  // for the remaining chunks of this half-iteration.)
  s.G_valid = false;
  ++s.F_epoch;
  return MALS_OK;
}

int mals_solve_side(mals_handle h, int side) {
  CHECK_SIDE(h, side);
  return solve_chunks(h, side, 0, (int)h->side[side].chunks.size());
}

int mals_solve_chunk(mals_handle h, int side, int32_t chunk) {
  CHECK_SIDE(h, side);
  return solve_chunks(h, side, chunk, chunk + 1);
}

// ---- csrc/mals_internal.h: the two-call form of mals_solve_chunk for mals_group.cpp (not part of the C-ABI)
int malsi_solve_chunk_begin(mals_handle h, int side, int32_t chunk) {
  CHECK_SIDE(h, side);
  return solve_chunks(h, side, chunk, chunk + 1, SOLVE_BEGIN);
}
int malsi_solve_chunk_end(mals_handle h, int side, int32_t chunk) {
  CHECK_SIDE(h, side);
  return solve_chunks(h, side, chunk, chunk + 1, SOLVE_END);
}
int malsi_dual_pending(mals_handle h) { return h && h->dual_pending ? 1 : 0; }
void* malsi_ready_event(mals_handle h) { return h ? (void*)h->ev_ready : nullptr; }
int malsi_ymax_slots(void) { return YMAX_SLOTS; }
int malsi_dual_host(mals_handle h, int side, mals_handle from) {
  CHECK_SIDE(h, side);
  if (!h->dual_pending || h->dual_pending_side != side) return fail(h, MALS_INVALID_ARG, "no chunk is waiting for the eigendecomposition");
  if (int rc = use_device(h)) return rc;
  return prepare_dual_host(h, side, from);
}

int mals_get_gather_scale(mals_handle h, float* out4) {
  if (!h || !out4) return MALS_INVALID_ARG;
  if (int rc = use_device(h)) return rc;
  HIPCHK(h, hipMemcpyAsync(out4, h->d_zscale, 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MALS_OK;
}

int mals_get_timeline(mals_handle h, double* out4) {
  if (!h || !out4) return MALS_INVALID_ARG;
  for (int i = 0; i < 4; ++i) out4[i] = h->tl[i];
  return MALS_OK;
}

int mals_set_chunk_rows(mals_handle h, int side, int64_t chunk_rows) {
  CHECK_SIDE(h, side);
  if (chunk_rows < 0) return fail(h, MALS_INVALID_ARG, "chunk_rows must be >= 0");
  h->side[side].chunk_rows_override = chunk_rows;
  return MALS_OK;
}

int mals_num_chunks(mals_handle h, int side, int32_t* n_chunks) {
  CHECK_SIDE(h, side);
  if (!h->side[side].has_matrix) return fail(h, MALS_INVALID_ARG, "matrix of this side not set");
  if (n_chunks) *n_chunks = (int32_t)h->side[side].chunks.size();
  return MALS_OK;
}

namespace {

// The reference's SingularMatrixSolverException carries RRQR's getRank(0.01) of the failing row's
// system (CMLSS:47).  Error path only: rebuild that one k x k system on the host in fp64 from the
// row's entries, the current opposite factors and their Gramian (ALS:450-492).  0 = unknown.
// the row's k x k system in fp64 (ALS:450-492); false = could not be rebuilt
bool build_row_system(mals_handle h, int side, int64_t local_row, std::vector<double>& W) {
  SideState& s = h->side[side];
  SideState& o = h->side[1 - side];
  const int k = h->cfg.features;
  if (!s.has_matrix || local_row < 0 || local_row >= s.n_local || !o.F) return false;
  const int64_t e0 = s.h_row_ptr[local_row], n_u = s.h_row_ptr[local_row + 1] - e0;
  std::vector<int32_t> col((size_t)n_u);
  std::vector<float> val((size_t)n_u);
  std::vector<int64_t> idx((size_t)n_u);
  std::vector<float> rows((size_t)n_u * k);
  W.assign((size_t)k * k, 0.0);
  if (n_u > 0) {
    if (n_u > (int64_t)INT32_MAX) return false;
    if (hipMemcpy(col.data(), s.col + e0, sizeof(int32_t) * (size_t)n_u, hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (hipMemcpy(val.data(), s.val + e0, sizeof(float) * (size_t)n_u, hipMemcpyDeviceToHost) != hipSuccess) return false;
    for (int64_t e = 0; e < n_u; ++e) idx[e] = col[e];
    if (mals_get_rows(h, 1 - side, idx.data(), (int32_t)n_u, rows.data()) != MALS_OK) return false;
  }
  const bool reconstruct = h->cfg.flags & MALS_FLAG_RECONSTRUCT_R;
  if (!(h->cfg.flags & MALS_FLAG_LOSS_IGNORES_UNSPECIFIED)) {
    if (!o.G || hipMemcpy(W.data(), o.G, sizeof(double) * W.size(), hipMemcpyDeviceToHost) != hipSuccess) return false;
  }
  for (int64_t e = 0; e < n_u; ++e) {
    const float* y = rows.data() + (size_t)e * k;
    double w = 0.0;
    if (h->cfg.flags & MALS_FLAG_LOSS_IGNORES_UNSPECIFIED) w += 1.0;             // ALS:524-539
    if (!reconstruct) w += h->cfg.alpha * std::fabs((double)val[e]);             // ALS:471-477
    for (int r = 0; r < k; ++r)
      for (int c = 0; c < k; ++c) W[(size_t)r * k + c] += w * (double)y[r] * (double)y[c];
  }
  for (int d = 0; d < k; ++d) W[(size_t)d * k + d] += h->cfg.lambda * h->cfg.alpha * (double)n_u;  // ALS:488
  return true;
}

int apparent_rank_of_row(mals_handle h, int side, int64_t local_row) {
  std::vector<double> W;
  if (!build_row_system(h, side, local_row, W)) return 0;
  return mals::PivotedQR(W.data(), h->cfg.features, h->cfg.singularity_threshold).rank(0.01);
}

}  // namespace

int mals_check(mals_handle h) {
  if (!h) return MALS_INVALID_ARG;
  if (int rc = use_device(h)) return rc;
  HIPCHK(h, hipMemcpyAsync(h->h_bad, h->d_bad, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // The kernels' verdict is "a Cholesky pivot <= threshold"; the reference's is "a diagonal element of R of the
  // column-pivoted QR <= threshold" (CMLSS:43-54), which can already hold when the smallest (unpivoted) Cholesky
  // pivot is still 100x the threshold (measured: 63 x 63 Gramian of 63 rows, pivot 5e-5, |R_dd| 2e-6).  So the row with
  // the smallest pivot within 1024x of the threshold is put to the reference's own test here, in fp64 on the host.
  for (int sd = 0; sd < 2; ++sd) {
    const unsigned long long key = h->h_bad[2 + sd];
    if (h->h_bad[sd] != ~0ull || key == ~0ull) continue;
    const int64_t row = (int64_t)(key & 0xffffffffull);
    SideState& s = h->side[sd];
    std::vector<double> W;
    if (row < s.n_local && s.h_row_ptr[(size_t)row + 1] - s.h_row_ptr[(size_t)row] <= 200000 && build_row_system(h, sd, row, W) &&
        !mals::PivotedQR(W.data(), h->cfg.features, h->cfg.singularity_threshold).non_singular())
      h->h_bad[sd] = (unsigned long long)row;
  }
  if (h->h_bad[2] != ~0ull || h->h_bad[3] != ~0ull)
    HIPCHK(h, hipMemsetAsync(h->d_bad + 2, 0xff, 2 * sizeof(unsigned long long), h->stream));
  for (int sd = 0; sd < 2; ++sd) {
    if (h->h_bad[sd] != ~0ull) {
      h->sing_side = sd;
      h->sing_row = h->side[sd].row_offset + (int64_t)h->h_bad[sd];
      h->sing_rank = apparent_rank_of_row(h, sd, (int64_t)h->h_bad[sd]);
      HIPCHK(h, hipMemsetAsync(h->d_bad, 0xff, 2 * sizeof(unsigned long long), h->stream));
      char buf[160];
      std::snprintf(buf, sizeof(buf), "near-singular system (pivot <= %g) for row %lld of side %c", h->cfg.singularity_threshold,
                    (long long)h->sing_row, sd == MALS_SIDE_X ? 'X' : 'Y');
      return fail(h, MALS_SINGULAR, buf);
    }
  }
  return MALS_OK;
}

int mals_singular_info(mals_handle h, int32_t* side, int64_t* row, int32_t* apparent_rank) {
  if (!h) return MALS_INVALID_ARG;
  if (side) *side = h->sing_side;
  if (row) *row = h->sing_row;
  if (apparent_rank) *apparent_rank = h->sing_rank;
  return MALS_OK;
}

// ---- Generation.recomputeSolver ---------------------------------------
struct mals_solver_s {
  mals::PivotedQR qr;
};

int mals_solver_create(const double* A, int32_t n, double singularity_threshold, mals_solver* out, int32_t* apparent_rank_out) {
  if (out) *out = nullptr;
  if (!A || !out || n <= 0) return MALS_INVALID_ARG;
  mals_solver s = new (std::nothrow) mals_solver_s{mals::PivotedQR(A, n, singularity_threshold)};
  if (!s) return MALS_OOM;
  if (!s->qr.non_singular()) {  // CMLSS:43-54
    if (apparent_rank_out) *apparent_rank_out = s->qr.rank(0.01);
    delete s;
    return MALS_SINGULAR;
  }
  *out = s;
  return MALS_OK;
}

int mals_solver_dim(mals_solver s) { return s ? s->qr.dim() : 0; }

int mals_solver_solve_dtof(mals_solver s, const double* b, float* x) {
  if (!s || !b || !x) return MALS_INVALID_ARG;
  std::vector<double> t((size_t)s->qr.dim());
  s->qr.solve(b, t.data());
  for (int i = 0; i < s->qr.dim(); ++i) x[i] = (float)t[i];  // CommonsMathSolver.java:40-42
  return MALS_OK;
}

int mals_solver_solve_ftod(mals_solver s, const float* b, double* x) {
  if (!s || !b || !x) return MALS_INVALID_ARG;
  std::vector<double> t(b, b + s->qr.dim());  // CommonsMathSolver.java:48-51
  s->qr.solve(t.data(), x);
  return MALS_OK;
}

int mals_solver_destroy(mals_solver s) {
  delete s;
  return MALS_OK;
}

int mals_recompute_solver(mals_handle h, int side, mals_solver* out, double* inf_norm_out) {
  CHECK_SIDE(h, side);
  if (!out) return fail(h, MALS_INVALID_ARG, "out must not be NULL");
  *out = nullptr;
  if (inf_norm_out) *inf_norm_out = 0.0;
  SideState& s = h->side[side];
  if (!s.F || s.n_total == 0) return MALS_OK;  // Generation.java:145-147: no solver for an empty matrix
  const int k = h->cfg.features;
  std::vector<double> G((size_t)k * k);
  if (int rc = mals_gramian(h, side, G.data())) return rc;  // Generation.java:148
  const double norm = mals::max_abs_column_sum(G.data(), k);  // :149
  if (inf_norm_out) *inf_norm_out = norm;
  if (!(norm >= 1.0)) {  // :150-153 (a NaN norm is ill-conditioned too)
    char buf[96];
    std::snprintf(buf, sizeof(buf), "infNorm: %g; try decreasing model.als.lambda", norm);
    return fail(h, MALS_ILL_CONDITIONED, buf);
  }
  int32_t rank = 0;
  const int rc = mals_solver_create(G.data(), k, h->cfg.singularity_threshold, out, &rank);
  if (rc == MALS_SINGULAR) {
    h->sing_side = side;
    h->sing_row = -1;
    h->sing_rank = rank;
    char buf[96];
    std::snprintf(buf, sizeof(buf), "Apparent rank: %d", rank);
    return fail(h, rc, buf);
  }
  return rc == MALS_OK ? MALS_OK : fail(h, rc, "mals_solver_create failed");
}

int mals_half_iteration(mals_handle h, int side) {
  CHECK_SIDE(h, side);
  if (!(h->cfg.flags & MALS_FLAG_LOSS_IGNORES_UNSPECIFIED) || h->split_f16 || !h->side[1 - side].G_valid) {
    if (int rc = mals_gramian(h, 1 - side, nullptr)) return rc;  // ALS:342 / ALS:369
  }
  if (int rc = mals_solve_side(h, side)) return rc;                // ALS:344 / ALS:371
  return mals_check(h);                                            // ALS:346-361 f.get()
}

int mals_sample_dots(mals_handle h, const int64_t* test_users, int32_t n_test_users, const int64_t* test_items, int32_t n_test_items,
                     double* host_out) {
  if (!h) return MALS_INVALID_ARG;
  SideState& x = h->side[MALS_SIDE_X];
  SideState& y = h->side[MALS_SIDE_Y];
  if (!x.F || !y.F) return fail(h, MALS_INVALID_ARG, "factor replicas not allocated");
  if (n_test_users < 0 || n_test_items < 0 || (n_test_users > 0 && !test_users) || (n_test_items > 0 && !test_items) ||
      (int64_t)n_test_users * n_test_items > (1 << 24))
    return fail(h, MALS_INVALID_ARG, "bad convergence sample");
  const int64_t n = (int64_t)n_test_users * n_test_items;
  if (n == 0) return MALS_OK;
  if (!host_out) return fail(h, MALS_INVALID_ARG, "null output");
  for (int i = 0; i < n_test_users; ++i)
    if (test_users[i] < 0 || test_users[i] >= x.n_total) return fail(h, MALS_INVALID_ARG, "test user outside the factor replica");
  for (int j = 0; j < n_test_items; ++j)
    if (test_items[j] < 0 || test_items[j] >= y.n_total) return fail(h, MALS_INVALID_ARG, "test item outside the factor replica");
  if (int rc = use_device(h)) return rc;
  const size_t idx_bytes = sizeof(int64_t) * (size_t)(n_test_users + n_test_items);
  if (h->sd_cap < (size_t)n || h->sd_idx_cap < idx_bytes) {
    free_dev(h->d_sd);
    free_dev(h->d_sd_idx);
    HIPCHK(h, hipMalloc(&h->d_sd, sizeof(double) * (size_t)n));
    HIPCHK(h, hipMalloc(&h->d_sd_idx, idx_bytes));
    h->sd_cap = (size_t)n;
    h->sd_idx_cap = idx_bytes;
  }
  HIPCHK(h, hipMemcpyAsync(h->d_sd_idx, test_users, sizeof(int64_t) * (size_t)n_test_users, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_sd_idx + n_test_users, test_items, sizeof(int64_t) * (size_t)n_test_items, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(sample_dots_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, x.F, y.F, h->d_sd_idx,
                     h->d_sd_idx + n_test_users, n_test_users, n_test_items, h->cfg.features, h->d_sd);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(host_out, h->d_sd, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MALS_OK;
}

int mals_factorize(mals_handle h, double convergence_threshold, int32_t max_iterations, int32_t random_y, int32_t iterate,
                   const int64_t* test_users, int32_t n_test_users, const int64_t* test_items, int32_t n_test_items,
                   int32_t* iterations_out, double* convergence_out) {
  if (!h) return MALS_INVALID_ARG;
  if (iterations_out) *iterations_out = 0;
  if (convergence_out) *convergence_out = std::numeric_limits<double>::quiet_NaN();
  // ALS:140-141 threshold must be in (0,1)
  if (!(convergence_threshold > 0.0 && convergence_threshold < 1.0))
    return fail(h, MALS_INVALID_ARG, "threshold must be in (0,1)");
  if (n_test_users < 0 || n_test_items < 0 || (n_test_users > 0 && !test_users) || (n_test_items > 0 && !test_items))
    return fail(h, MALS_INVALID_ARG, "bad convergence sample");
  h->cancelled.store(0);
  if (!iterate) return mals_half_iteration(h, MALS_SIDE_X);  // ALS:196-204
  std::vector<double> est((size_t)n_test_users * (size_t)n_test_items, 0.0);  // ALS:215: X empty => 0
  std::vector<double> fresh(est.size());
  int it = 0;
  for (;;) {
    const double t_it = now_us();
    const int64_t rows0 = h->stats.rows_solved, nnz0 = h->stats.nnz_gathered;
    if (h->cancelled.load()) return fail(h, MALS_CANCELLED, "cancelled");
    if (int rc = mals_half_iteration(h, MALS_SIDE_X)) return rc;  // ALS:228
    if (h->cancelled.load()) return fail(h, MALS_CANCELLED, "cancelled");
    if (int rc = mals_half_iteration(h, MALS_SIDE_Y)) return rc;  // ALS:229
    // the sample dots on the device (ALS:234), the order-dependent running mean on the host (ALS:237)
    if (int rc = mals_sample_dots(h, test_users, n_test_users, test_items, n_test_items, fresh.data())) return rc;
    double tw = 0.0, mean = std::numeric_limits<double>::quiet_NaN();
    for (int i = 0; i < n_test_users; ++i) {
      for (int j = 0; j < n_test_items; ++j) {  // ALS:231-238
        const double nv = fresh[(size_t)i * n_test_items + j];
        const double ov = est[(size_t)i * n_test_items + j];
        est[(size_t)i * n_test_items + j] = nv;
        dwm_increment(tw, mean, std::fabs(nv - ov), nv > 0.0 ? nv : 0.0);
      }
    }
    ++it;
    if (iterations_out) *iterations_out = it;
    if (convergence_out) *convergence_out = mean;
    if (h->iter_fn) {   // what the reference logs per iteration (ALS:241-246, 351-358)
      mals_iteration_info info;
      std::memset(&info, 0, sizeof(info));
      info.struct_size = (int32_t)sizeof(info);
      info.iteration = it;
      info.avg_abs_difference = mean;
      info.seconds = (now_us() - t_it) * 1e-6;
      info.x_rows = h->side[MALS_SIDE_X].n_local;
      info.y_rows = h->side[MALS_SIDE_Y].n_local;
      info.entries_gathered = h->stats.nnz_gathered - nnz0;
      info.algorithmic_bytes = (double)info.entries_gathered * (4.0 * h->cfg.features + 8.0) +
                               (double)(h->stats.rows_solved - rows0) * (4.0 * h->cfg.features + 8.0);
      info.devices = 1;
      h->iter_fn(h->iter_user, &info);
    }
    if (max_iterations > 0 && it >= max_iterations) break;                 // ALS:242-245
    if (!std::isfinite(mean)) break;                                       // ALS:248-251
    if (!(random_y && it == 1) && mean < convergence_threshold) break;     // ALS:253-256
  }
  return MALS_OK;
}

int mals_reconstruction_error(mals_handle h, double* sum_out, int64_t* count_out) {
  if (!h || !sum_out || !count_out) return MALS_INVALID_ARG;
  SideState& s = h->side[MALS_SIDE_X];
  SideState& o = h->side[MALS_SIDE_Y];
  if (!s.has_matrix || !s.F || !o.F) return fail(h, MALS_INVALID_ARG, "needs the user-side matrix and both factor replicas");
  if (int rc = use_device(h)) return rc;
  *sum_out = 0.0;
  *count_out = 0;
  if (s.n_local == 0) return MALS_OK;
  const unsigned grid = (unsigned)std::min<int64_t>((s.n_local + 3) / 4, (int64_t)h->n_cu * 32);
  const size_t n_waves = (size_t)grid * 4;
  double* d_sum = nullptr;
  unsigned long long* d_cnt = nullptr;
  HIPCHK(h, hipMalloc(&d_sum, sizeof(double) * n_waves));
  HIPCHK(h, hipMalloc(&d_cnt, sizeof(unsigned long long) * n_waves));
  int rc = MALS_INVALID_ARG;
  switch (h->T) {
    case 1: rc = launch_reconstruction<1>(h, s, o, grid, d_sum, d_cnt); break;
    case 2: rc = launch_reconstruction<2>(h, s, o, grid, d_sum, d_cnt); break;
    case 3: rc = launch_reconstruction<3>(h, s, o, grid, d_sum, d_cnt); break;
    case 4: rc = launch_reconstruction<4>(h, s, o, grid, d_sum, d_cnt); break;
    case 5: rc = launch_reconstruction<5>(h, s, o, grid, d_sum, d_cnt); break;
    case 6: rc = launch_reconstruction<6>(h, s, o, grid, d_sum, d_cnt); break;
    case 7: rc = launch_reconstruction<7>(h, s, o, grid, d_sum, d_cnt); break;
    case 8: rc = launch_reconstruction<8>(h, s, o, grid, d_sum, d_cnt); break;
  }
  std::vector<double> hs(n_waves);
  std::vector<unsigned long long> hc(n_waves);
  if (rc == MALS_OK) {
    hipError_t e = hipMemcpyAsync(hs.data(), d_sum, sizeof(double) * n_waves, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hc.data(), d_cnt, sizeof(unsigned long long) * n_waves, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) rc = fail(h, MALS_HIP_ERROR, hipGetErrorString(e));
  }
  (void)hipFree(d_sum);
  (void)hipFree(d_cnt);
  if (rc != MALS_OK) return rc;
  double sum = 0.0;
  unsigned long long cnt = 0;
  for (size_t w = 0; w < n_waves; ++w) {  // fixed order: deterministic
    sum += hs[w];
    cnt += hc[w];
  }
  *sum_out = sum;
  *count_out = (int64_t)cnt;
  return MALS_OK;
}

// (a failed argument check of a recommend call: the message under the front's mutex -- other request threads may be
// inside the same handle)
static int topn_fail(mals_handle h, int code, const char* msg) {
  std::lock_guard<std::mutex> lk(topn_front(h)->mu);
  h->err = msg;
  return code;
}

int mals_recommend(mals_handle h, const int64_t* user_idx, int32_t n_queries, int32_t how_many, int32_t consider_known_items,
                   int64_t* item_idx_out, float* score_out, int32_t* n_out) {
  if (!h) return MALS_INVALID_ARG;
  SideState& x = h->side[MALS_SIDE_X];
  SideState& y = h->side[MALS_SIDE_Y];
  if (!x.F || !y.F || y.n_total == 0) return topn_fail(h, MALS_INVALID_ARG, "factor replicas not available");
  if (n_queries < 0 || how_many <= 0 || how_many > 4096 || (n_queries > 0 && (!user_idx || !item_idx_out || !score_out)))
    return topn_fail(h, MALS_INVALID_ARG, "bad recommend arguments (how_many in 1..4096)");
  if (!consider_known_items && !x.has_matrix && !h->known_ptr)
    return topn_fail(h, MALS_INVALID_ARG, "the user-side matrix (or mals_set_known_items) is needed to skip known items");
  if (!consider_known_items && h->known_ptr && h->known_rows != x.n_local)
    return topn_fail(h, MALS_INVALID_ARG, "the installed known items do not match the local user rows");
  for (int q = 0; q < n_queries; ++q) {
    if (user_idx[q] < 0 || user_idx[q] >= x.n_total) return topn_fail(h, MALS_INVALID_ARG, "user index outside the factor replica");
    if (!consider_known_items && (user_idx[q] < x.row_offset || user_idx[q] >= x.row_offset + x.n_local))
      return topn_fail(h, MALS_INVALID_ARG, "known items of this user are not on this handle (row outside the local shard)");
  }
  if (n_queries == 0) return MALS_OK;
  if (h->tag_bits && h->tag_bits_items != y.n_total)
    return topn_fail(h, MALS_INVALID_ARG, "the tag items were set for another item count: call mals_set_tag_items again");
  if (hipSetDevice(h->cfg.device) != hipSuccess) return topn_fail(h, MALS_HIP_ERROR, "hipSetDevice failed");
  TopnTicket t;
  TopnRequest rq;
  if (n_queries < TOPN_FRONT_BULK && how_many <= TOPN_FILTER_MAX_N) {
    t.user_idx = user_idx;
    t.n = n_queries;
    t.how_many = how_many;
    t.skip_known = !consider_known_items;
    t.item_out = item_idx_out;
    t.score_out = score_out;
    t.n_out = n_out;
  } else {
    rq.n_queries = n_queries;
    rq.how_many = how_many;
    rq.user_idx = user_idx;
    rq.skip_known = !consider_known_items;
    rq.item_idx_out = item_idx_out;
    rq.score_out = score_out;
    rq.n_out = n_out;
    t.bulk = &rq;
    t.how_many = how_many;
  }
  return topn_front_submit(h, t);
}

int mals_set_known_items(mals_handle h, int64_t n_rows, const int64_t* row_ptr, const int32_t* item_idx, int mem_kind) {
  if (!h) return MALS_INVALID_ARG;
  if (mem_kind != MALS_MEM_HOST && mem_kind != MALS_MEM_DEVICE) return fail(h, MALS_INVALID_ARG, "mem_kind must be MALS_MEM_HOST or MALS_MEM_DEVICE");
  if (int rc = use_device(h)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  clear_known_items(h);
  if (!row_ptr) return MALS_OK;  // back to the rows of R
  SideState& x = h->side[MALS_SIDE_X];
  if (n_rows != x.n_local) return fail(h, MALS_INVALID_ARG, "known items: one row per local user row of side X");
  if (mem_kind == MALS_MEM_DEVICE) {
    h->known_ptr = row_ptr;
    h->known_idx = item_idx;
  } else {
    const int64_t n = row_ptr[n_rows];
    if (n < 0 || (n > 0 && !item_idx)) return fail(h, MALS_INVALID_ARG, "bad known-item arrays");
    HIPCHK(h, hipMalloc(&h->known_ptr_own, sizeof(int64_t) * (size_t)(n_rows + 1)));
    HIPCHK(h, hipMalloc(&h->known_idx_own, sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1)));
    HIPCHK(h, hipMemcpy(h->known_ptr_own, row_ptr, sizeof(int64_t) * (size_t)(n_rows + 1), hipMemcpyHostToDevice));
    if (n) HIPCHK(h, hipMemcpy(h->known_idx_own, item_idx, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    h->known_ptr = h->known_ptr_own;
    h->known_idx = h->known_idx_own;
  }
  h->known_rows = n_rows;
  return MALS_OK;
}

int mals_set_tag_items(mals_handle h, int64_t n, const int64_t* item_idx, int mem_kind) {
  if (!h) return MALS_INVALID_ARG;
  if (mem_kind != MALS_MEM_HOST && mem_kind != MALS_MEM_DEVICE) return fail(h, MALS_INVALID_ARG, "mem_kind must be MALS_MEM_HOST or MALS_MEM_DEVICE");
  if (n < 0 || (n > 0 && !item_idx)) return fail(h, MALS_INVALID_ARG, "bad tag item list");
  if (int rc = use_device(h)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_dev(h->tag_bits);
  h->tag_bits_items = h->n_tag_items = 0;
  if (n == 0) return MALS_OK;
  const int64_t n_items = h->side[MALS_SIDE_Y].n_total;
  if (n_items <= 0) return fail(h, MALS_INVALID_ARG, "tag items: the item factor replica comes first (mals_set_factor_rows)");
  const size_t words = ((size_t)((n_items + 31) / 32) + 1) & ~(size_t)1;   // even: the counter behind them is 8-byte aligned
  uint32_t* bits = nullptr;
  HIPCHK(h, hipMalloc(&bits, sizeof(uint32_t) * words + sizeof(unsigned long long)));
  unsigned long long* d_n = reinterpret_cast<unsigned long long*>(bits + words);
  HIPCHK(h, hipMemsetAsync(bits, 0, sizeof(uint32_t) * words + sizeof(unsigned long long), h->stream));
  int64_t* d_idx = nullptr;
  const int64_t* src = item_idx;
  if (mem_kind == MALS_MEM_HOST) {
    HIPCHK(h, hipMalloc(&d_idx, sizeof(int64_t) * (size_t)n));
    HIPCHK(h, hipMemcpyAsync(d_idx, item_idx, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    src = d_idx;
  }
  hipLaunchKernelGGL(topn_tag_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, src, n, n_items, bits, d_n);
  unsigned long long n_set = 0;
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&n_set, d_n, sizeof(n_set), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  free_dev(d_idx);
  if (e != hipSuccess) {
    free_dev(bits);
    return fail(h, MALS_HIP_ERROR, std::string("mals_set_tag_items: ") + hipGetErrorString(e));
  }
  if (n_set == 0) {  // none of them owns a row of Y: nothing to strike
    free_dev(bits);
    return MALS_OK;
  }
  h->tag_bits = bits;
  h->tag_bits_items = n_items;
  h->n_tag_items = (int64_t)n_set;
  return MALS_OK;
}

int mals_get_tag_item_count(mals_handle h, int64_t* n_out) {
  if (!h || !n_out) return MALS_INVALID_ARG;
  *n_out = h->n_tag_items;
  return MALS_OK;
}

int mals_recommend_to_many(mals_handle h, const float* vectors, const int64_t* vector_ptr, int32_t n_queries, int32_t how_many,
                           const int64_t* exclude_ptr, const int64_t* exclude_idx, int64_t* item_idx_out, float* score_out,
                           int32_t* n_out) {
  if (!h) return MALS_INVALID_ARG;
  SideState& y = h->side[MALS_SIDE_Y];
  if (!y.F || y.n_total == 0) return topn_fail(h, MALS_INVALID_ARG, "item factor replica not available");
  if (n_queries < 0 || how_many <= 0 || how_many > 4096 || (n_queries > 0 && (!vectors || !item_idx_out || !score_out)))
    return topn_fail(h, MALS_INVALID_ARG, "bad recommend arguments (how_many in 1..4096)");
  if ((exclude_ptr == nullptr) != (exclude_idx == nullptr) && exclude_ptr && exclude_ptr[n_queries] > 0)
    return topn_fail(h, MALS_INVALID_ARG, "exclude_ptr and exclude_idx go together");
  if (vector_ptr) {
    if (vector_ptr[0] != 0) return topn_fail(h, MALS_INVALID_ARG, "vector_ptr[0] must be 0");
    for (int q = 0; q < n_queries; ++q) {
      const int64_t n = vector_ptr[q + 1] - vector_ptr[q];
      if (n < 1) return topn_fail(h, MALS_INVALID_ARG, "features must not be empty");  // RecommendIterator.java:52
      if (n > (1 << 20)) return topn_fail(h, MALS_INVALID_ARG, "too many vectors in one query");
    }
  }
  if (n_queries == 0) return MALS_OK;
  if (h->tag_bits && h->tag_bits_items != y.n_total)
    return topn_fail(h, MALS_INVALID_ARG, "the tag items were set for another item count: call mals_set_tag_items again");
  if (hipSetDevice(h->cfg.device) != hipSuccess) return topn_fail(h, MALS_HIP_ERROR, "hipSetDevice failed");
  TopnRequest rq;
  TopnTicket t;
  t.how_many = how_many;
  const int64_t n_vec = vector_ptr ? vector_ptr[n_queries] : (int64_t)n_queries;
  const int64_t n_ex = (exclude_ptr && exclude_idx) ? exclude_ptr[n_queries] - exclude_ptr[0] : 0;
  if (n_queries < TOPN_FRONT_BULK && how_many <= TOPN_FILTER_MAX_N && n_vec <= 4 * TOPN_FRONT_BULK && n_ex <= (1 << 16) &&
      (!exclude_ptr || exclude_ptr[0] == 0)) {
    // a small call (an anonymous user, recommendToMany for a handful of users): folded into a pass with the other callers'
    t.vectors = vectors;
    t.vec_ptr = vector_ptr;
    t.excl_ptr = exclude_ptr;
    t.excl_idx = exclude_idx;
    t.n = n_queries;
    t.item_out = item_idx_out;
    t.score_out = score_out;
    t.n_out = n_out;
  } else {
    rq.n_queries = n_queries;
    rq.how_many = how_many;
    rq.vectors = vectors;
    rq.vec_ptr = vector_ptr;
    rq.excl_ptr = exclude_ptr;
    rq.excl_idx = exclude_idx;
    rq.item_idx_out = item_idx_out;
    rq.score_out = score_out;
    rq.n_out = n_out;
    t.bulk = &rq;
  }
  return topn_front_submit(h, t);
}

int mals_recommend_vectors(mals_handle h, const float* query_vectors, int32_t n_queries, int32_t how_many,
                           const int64_t* exclude_ptr, const int64_t* exclude_idx, int64_t* item_idx_out, float* score_out,
                           int32_t* n_out) {
  return mals_recommend_to_many(h, query_vectors, nullptr, n_queries, how_many, exclude_ptr, exclude_idx, item_idx_out, score_out, n_out);
}

int mals_recommend_front_stats(mals_handle h, int64_t* out4) {
  if (!h || !out4) return MALS_INVALID_ARG;
  TopnFront* f = topn_front(h);
  std::lock_guard<std::mutex> lk(f->mu);
  out4[0] = (int64_t)f->calls;
  out4[1] = (int64_t)f->queries;
  out4[2] = (int64_t)f->passes;
  out4[3] = (int64_t)f->bulk_calls;
  return MALS_OK;
}

int mals_recommend_set_spin_us(mals_handle h, int32_t spin_us) {
  if (!h || spin_us < 0 || spin_us > 1000000) return MALS_INVALID_ARG;
  TopnFront* f = topn_front(h);
  std::lock_guard<std::mutex> lk(f->mu);
  f->spin_us = spin_us;
  return MALS_OK;
}

int mals_recommend_set_depth(mals_handle h, int32_t passes_in_flight) {
  if (!h || passes_in_flight < 1 || passes_in_flight > TOPN_SLOTS) return MALS_INVALID_ARG;
  TopnFront* f = topn_front(h);
  std::lock_guard<std::mutex> lk(f->mu);
  if (f->n_busy > 0 || f->leader) { h->err = "mals_recommend_set_depth: calls are in flight"; return MALS_INVALID_ARG; }
  f->depth = passes_in_flight;
  f->next_slot = f->oldest = 0;
  return MALS_OK;
}

int mals_symmetric_eigen(const double* A, int32_t n, double* evals_out, double* V_out) {
  if (!A || !evals_out || !V_out || n <= 0) return MALS_INVALID_ARG;
  return mals::symmetric_eigen(A, n, evals_out, V_out) ? MALS_OK : MALS_INVALID_ARG;
}

int mals_cancel(mals_handle h) {
  if (!h) return MALS_INVALID_ARG;
  h->cancelled.store(1);
  return MALS_OK;
}

int mals_set_iteration_callback(mals_handle h, mals_iteration_fn fn, void* user) {
  if (!h) return MALS_INVALID_ARG;
  h->iter_fn = fn;
  h->iter_user = user;
  return MALS_OK;
}

int mals_enable_timing(mals_handle h, int32_t on) {
  if (!h) return MALS_INVALID_ARG;
  h->timing = on != 0;
  return MALS_OK;
}

int mals_reset_stats(mals_handle h) {
  if (!h) return MALS_INVALID_ARG;
  if (int rc = use_device(h)) return rc;
  if (int rc = drain_events(h)) return rc;
  std::memset(&h->stats, 0, sizeof(h->stats));
  h->stats.struct_size = (int32_t)sizeof(mals_stats);
  HIPCHK(h, hipMemsetAsync(h->d_refined, 0, sizeof(unsigned long long), h->stream));
  return MALS_OK;
}

int mals_get_stats(mals_handle h, mals_stats* out) {
  if (!h || !out) return MALS_INVALID_ARG;
  if (int rc = use_device(h)) return rc;
  if (int rc = drain_events(h)) return rc;
  unsigned long long refined = 0;
  HIPCHK(h, hipMemcpyAsync(&refined, h->d_refined, sizeof(refined), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->stats.rows_refined = (int64_t)refined;
  *out = h->stats;
  return MALS_OK;
}

int mals_set_refine_limit(mals_handle h, double limit) {
  if (!h || !(limit >= 0.0)) return h ? fail(h, MALS_INVALID_ARG, "refine limit must be >= 0") : MALS_INVALID_ARG;
  h->refine_limit = (float)limit;
  return MALS_OK;
}

}  // extern "C"
