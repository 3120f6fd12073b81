// ingest_api.hip -- host side of the ingest -> CSR path (mals_ingest_*, include/myrrix_als.h).
// Everything between "records appended" and "two CSR matrices + id tables in HBM" runs on the device
// (ingest_kernels.h); the host only sequences kernels and reads back three counters.
#include "../../include/myrrix_als.h"
#include "ingest_kernels.h"
#include "ingest_text_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace mals;

constexpr int64_t MALS_INGEST_MAX_RECORDS = (int64_t)1 << 36;       // (the record arrays alone are 1.6 TB there)
constexpr int64_t MALS_INGEST_ONE_SHOT_MAX = (int64_t)0x7fffff00;   // what one sort pipeline holds (32-bit positions)
constexpr int64_t MALS_INGEST_MIN_PART = (int64_t)1 << 26;          // smallest user range the automatic choice makes (ingest_big_host.h)

struct mals_ingest_s {
  int device = 0;
  float zero_threshold = 1.0e-4f;
  hipStream_t stream = nullptr;
  std::string err;
  // appended records (device), capacity-doubling
  int64_t n = 0, cap = 0;
  int64_t* d_user = nullptr;
  int64_t* d_item = nullptr;
  float* d_value = nullptr;
  // results
  bool finished = false;
  int64_t n_users = 0, n_items = 0, nnz = 0;
  int64_t* ids[2] = {nullptr, nullptr};      // dense index -> id, ascending
  int64_t* ptr[2] = {nullptr, nullptr};      // CSR row pointers (side X: by user, side Y: by item)
  int32_t* col[2] = {nullptr, nullptr};
  float* val[2] = {nullptr, nullptr};
  // sort/scan workspace, kept between finishes, one allocation per buffer.  On some boxes a fresh
  // hipMalloc of tens of GB takes 1-1.5 s and returns memory in which this pipeline runs ~2x slower
  // (driver memory placement, not under the library's control); workspace_ms reports the former.
  static constexpr int N_WS = 12;
  void* ws[N_WS] = {};
  size_t ws_bytes[N_WS] = {};
  double last_workspace_ms = 0.0;  // host time spent (re)allocating the workspace in the last finish
  double last_finish_ms = 0.0;
  double bytes_moved = 0.0;  // algorithmic bytes of the last finish (reads + writes of every pass)
  int radix_passes = 0;
  // ---- text -> records (ingest_text_host.h) ----
  size_t text_block_bytes = (size_t)256 << 20;
  uint8_t* d_text = nullptr;   // [carried tail][new bytes][padding]
  size_t text_cap = 0;
  uint8_t* d_carry = nullptr;  // the unterminated tail of the previous block
  size_t carry_cap = 0, carry_len = 0;
  bool carry_ends_cr = false;
  void* h_pinned = nullptr;    // staging buffer of mals_ingest_read_file
  size_t pinned_cap = 0;
  unsigned* t_block_counts = nullptr;
  size_t t_block_counts_cap = 0;
  unsigned* t_tile_sums = nullptr;
  size_t t_tile_sums_cap = 0;
  size_t t_line_cap = 0;       // per-line arrays of one block
  unsigned *t_starts = nullptr, *t_flag = nullptr, *t_flag_scan = nullptr, *t_defer = nullptr;
  uint8_t* t_status = nullptr;
  int64_t *t_user = nullptr, *t_item = nullptr;
  uint32_t* t_value = nullptr;
  mals::TextCounters* t_counters = nullptr;  // + two tag cursors
  int64_t* d_tags[2] = {nullptr, nullptr};   // [0]: itemTagIDs (tags seen in the user column), [1]: userTagIDs; with repeats
  size_t tag_cap[2] = {0, 0}, n_tags_raw[2] = {0, 0};
  int64_t lines = 0, bad_lines = 0, header_lines = 0, skipped_lines = 0, slow_lines = 0, text_bytes = 0;
  bool abort_armed = false;    // badLines > 100: the next line throws (IFR:96-98)
  bool text_failed = false;
  int text_fail_code = 0;
  std::string text_fail_msg;
  double parse_ms = 0.0, stage_ms = 0.0;
  hipEvent_t t_ev[2] = {nullptr, nullptr};
  // results of the last finish that come from the text path's extras
  int64_t* tag_ids[2] = {nullptr, nullptr};  // ascending, unique
  int64_t n_tag_ids[2] = {0, 0};
  bool want_known = false;
  int64_t n_known = 0;
  int64_t* known_ptr = nullptr;  // knownItemIDs as a CSR over the dense user / item indices
  int32_t* known_idx = nullptr;
  int64_t* tag_item_idx = nullptr;  // dense item index of every userTagID (ascending ids), -1: the tag owns no row of R^T
  int64_t part_cap = 0;       // MALS_INGEST_OPT_PARTITION_RECORDS (0: default) -- ingest_big_host.h
  int32_t last_partitions = 0, last_item_ranges = 0;
};

namespace {

int fail(mals_ingest g, int code, const std::string& msg) {
  if (g) g->err = msg;
  return code;
}

#define ICHK(g, call)                                                                  \
  do {                                                                                 \
    hipError_t _e = (call);                                                            \
    if (_e != hipSuccess) {                                                            \
      const int _code = (_e == hipErrorOutOfMemory) ? MALS_OOM : MALS_HIP_ERROR;       \
      return fail(g, _code, std::string(#call) + ": " + hipGetErrorString(_e));        \
    }                                                                                  \
  } while (0)

template <typename P>
void dfree(P*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

unsigned blocks_for(int64_t n, int per_block = 256, int64_t cap = 1 << 20) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + per_block - 1) / per_block, cap));
}

// scratch shared by the sorts and scans of one finish
struct Scratch {
  uint64_t* keys[2] = {nullptr, nullptr};
  unsigned* pay[2] = {nullptr, nullptr};
  uint64_t* pay64[2] = {nullptr, nullptr};
  unsigned* counts = nullptr;     // 256 * n_blocks
  unsigned* tile_sums = nullptr;  // scan tiles
  unsigned long long* digit_tot = nullptr;  // [8][256]
  unsigned* total = nullptr;      // grand total of a scan
  // all of it is carved out of the ingest object's arena
};

// exclusive scan of `n` uint32 (in != out allowed to alias); optional grand total copied to *host_total
int scan_u32(mals_ingest g, Scratch& s, const unsigned* in, unsigned* out, int64_t n, unsigned* host_total) {
  if (n <= 0) {
    if (host_total) *host_total = 0;
    return MALS_OK;
  }
  const int64_t tiles = (n + SC_TILE - 1) / SC_TILE;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)tiles), dim3(256), 0, g->stream, in, n, s.tile_sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, g->stream, s.tile_sums, tiles, s.total);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)tiles), dim3(256), 0, g->stream, in, n, out, s.tile_sums);
  ICHK(g, hipGetLastError());
  g->bytes_moved += 12.0 * (double)n;
  if (host_total) {
    ICHK(g, hipMemcpyAsync(host_total, s.total, sizeof(unsigned), hipMemcpyDeviceToHost, g->stream));
    ICHK(g, hipStreamSynchronize(g->stream));
  }
  return MALS_OK;
}

// Stable LSD radix sort of (keys[0], pay[0]) by the key digits >= first_digit; *result = index of the
// buffer pair holding the sorted data.  Digits on which all keys agree are skipped.  K = uint32_t when the caller knows
// that every key fits (ids below 2^32): 12 bytes less per record and pass, and a third workgroup per CU (LDS).
// key_bound (optional): a value no key exceeds, when the caller knows one (dense ranks): the digits above it are skipped
// without the pass over the keys that counts them.
template <typename K, typename P>
int radix_sort(mals_ingest g, Scratch& s, K* const (&keys)[2], P* const (&pay)[2], int64_t n, int* result, int first_digit = 0,
               uint64_t key_bound = 0) {
  constexpr int ND = (int)sizeof(K);
  *result = 0;
  if (n <= 1) return MALS_OK;
  std::vector<unsigned long long> tot(8 * 256, 0ull);
  if (key_bound != 0) {
    for (int d = 0; d < ND; ++d)
      if ((key_bound >> (8 * d)) == 0) tot[(size_t)d * 256] = (unsigned long long)n;  // every key has a zero there
  } else {
    ICHK(g, hipMemsetAsync(s.digit_tot, 0, 8 * 256 * sizeof(unsigned long long), g->stream));
    hipLaunchKernelGGL(rs_digit_totals_kernel<K>, dim3(blocks_for(n, 256 * 16, 4096)), dim3(256), 0, g->stream, keys[0], n, s.digit_tot);
    ICHK(g, hipGetLastError());
    ICHK(g, hipMemcpyAsync(tot.data(), s.digit_tot, tot.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, g->stream));
    ICHK(g, hipStreamSynchronize(g->stream));
    g->bytes_moved += (double)sizeof(K) * (double)n;
  }
  const int64_t n_blocks = (n + RS_BLOCK_TILE - 1) / RS_BLOCK_TILE;
  const unsigned grid = (unsigned)n_blocks;
  int cur = 0;
  for (int d = first_digit; d < ND; ++d) {
    bool trivial = false;
    for (int b = 0; b < 256; ++b)
      if (tot[(size_t)d * 256 + b] == (unsigned long long)n) trivial = true;
    if (trivial) continue;
    hipLaunchKernelGGL(rs_histogram_kernel<K>, dim3(grid), dim3(256), 0, g->stream, keys[cur], n, 8 * d, n_blocks, s.counts);
    ICHK(g, hipGetLastError());
    if (int rc = scan_u32(g, s, s.counts, s.counts, 256 * n_blocks, nullptr)) return rc;
    hipLaunchKernelGGL((rs_scatter_kernel<K, P>), dim3(grid), dim3(256), 0, g->stream, keys[cur], pay[cur], n, 8 * d, n_blocks,
                       s.counts, keys[1 - cur], pay[1 - cur]);
    ICHK(g, hipGetLastError());
    g->bytes_moved += ((double)sizeof(K) + 2.0 * ((double)sizeof(K) + sizeof(P))) * (double)n;  // histogram read; scatter read + write
    ++g->radix_passes;
    cur = 1 - cur;
  }
  *result = cur;
  return MALS_OK;
}

void free_results(mals_ingest g) {
  for (int sd = 0; sd < 2; ++sd) {
    dfree(g->ids[sd]);
    dfree(g->ptr[sd]);
    dfree(g->col[sd]);
    dfree(g->val[sd]);
    dfree(g->tag_ids[sd]);
    g->n_tag_ids[sd] = 0;
  }
  dfree(g->known_ptr);
  dfree(g->known_idx);
  dfree(g->tag_item_idx);
  g->n_known = 0;
  g->finished = false;
  g->n_users = g->n_items = g->nnz = 0;
}

}  // namespace

extern "C" {

int mals_ingest_create(int32_t device, float zero_threshold, mals_ingest* out) {
  if (!out) return MALS_INVALID_ARG;
  *out = nullptr;
  if (!(zero_threshold >= 0.f)) return MALS_INVALID_ARG;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return MALS_HIP_ERROR;
  mals_ingest g = new (std::nothrow) mals_ingest_s();
  if (!g) return MALS_OOM;
  g->device = device;
  g->zero_threshold = zero_threshold;
  *out = g;
  return MALS_OK;
}

int mals_ingest_destroy(mals_ingest g) {
  if (!g) return MALS_INVALID_ARG;
  (void)hipSetDevice(g->device);
  (void)hipStreamSynchronize(g->stream);
  free_results(g);
  dfree(g->d_user);
  dfree(g->d_item);
  dfree(g->d_value);
  for (int b = 0; b < mals_ingest_s::N_WS; ++b) dfree(g->ws[b]);
  dfree(g->d_text); dfree(g->d_carry); dfree(g->t_block_counts); dfree(g->t_tile_sums); dfree(g->t_starts); dfree(g->t_flag);
  dfree(g->t_flag_scan); dfree(g->t_defer); dfree(g->t_status); dfree(g->t_user); dfree(g->t_item); dfree(g->t_value);
  dfree(g->t_counters); dfree(g->d_tags[0]); dfree(g->d_tags[1]);
  if (g->h_pinned) (void)hipHostFree(g->h_pinned);
  if (g->t_ev[0]) (void)hipEventDestroy(g->t_ev[0]);
  if (g->t_ev[1]) (void)hipEventDestroy(g->t_ev[1]);
  delete g;
  return MALS_OK;
}

const char* mals_ingest_last_error(mals_ingest g) { return g ? g->err.c_str() : "null ingest handle"; }

int mals_ingest_append(mals_ingest g, int64_t n, const int64_t* user_ids, const int64_t* item_ids, const float* values,
                       int mem_kind) {
  if (!g) return MALS_INVALID_ARG;
  if (n < 0 || (n > 0 && (!user_ids || !item_ids || !values))) return fail(g, MALS_INVALID_ARG, "bad record arrays");
  if (mem_kind != MALS_MEM_HOST && mem_kind != MALS_MEM_DEVICE) return fail(g, MALS_INVALID_ARG, "mem_kind must be MALS_MEM_HOST or MALS_MEM_DEVICE");
  if (g->n + n >= MALS_INGEST_MAX_RECORDS) return fail(g, MALS_INVALID_ARG, "at most 2^36 records per ingest");
  if (n == 0) return MALS_OK;
  ICHK(g, hipSetDevice(g->device));
  if (g->finished) free_results(g);
  if (g->n + n > g->cap) {
    const int64_t cap = std::max<int64_t>(g->n + n, g->cap * 2);
    int64_t *u = nullptr, *it = nullptr;
    float* v = nullptr;
    ICHK(g, hipMalloc(&u, sizeof(int64_t) * (size_t)cap));
    ICHK(g, hipMalloc(&it, sizeof(int64_t) * (size_t)cap));
    ICHK(g, hipMalloc(&v, sizeof(float) * (size_t)cap));
    if (g->n) {
      ICHK(g, hipMemcpyAsync(u, g->d_user, sizeof(int64_t) * (size_t)g->n, hipMemcpyDeviceToDevice, g->stream));
      ICHK(g, hipMemcpyAsync(it, g->d_item, sizeof(int64_t) * (size_t)g->n, hipMemcpyDeviceToDevice, g->stream));
      ICHK(g, hipMemcpyAsync(v, g->d_value, sizeof(float) * (size_t)g->n, hipMemcpyDeviceToDevice, g->stream));
      ICHK(g, hipStreamSynchronize(g->stream));
    }
    dfree(g->d_user);
    dfree(g->d_item);
    dfree(g->d_value);
    g->d_user = u;
    g->d_item = it;
    g->d_value = v;
    g->cap = cap;
  }
  const hipMemcpyKind kind = mem_kind == MALS_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  ICHK(g, hipMemcpyAsync(g->d_user + g->n, user_ids, sizeof(int64_t) * (size_t)n, kind, g->stream));
  ICHK(g, hipMemcpyAsync(g->d_item + g->n, item_ids, sizeof(int64_t) * (size_t)n, kind, g->stream));
  ICHK(g, hipMemcpyAsync(g->d_value + g->n, values, sizeof(float) * (size_t)n, kind, g->stream));
  ICHK(g, hipStreamSynchronize(g->stream));
  g->n += n;
  return MALS_OK;
}

}  // extern "C"

static int alloc_results(mals_ingest g, unsigned n_users, unsigned n_items, unsigned nnz) {
  g->n_users = n_users;
  g->n_items = n_items;
  g->nnz = nnz;
  const size_t nnz1 = std::max<size_t>(nnz, 1);
  ICHK(g, hipMalloc(&g->ids[0], sizeof(int64_t) * std::max<size_t>(n_users, 1)));
  ICHK(g, hipMalloc(&g->ids[1], sizeof(int64_t) * std::max<size_t>(n_items, 1)));
  ICHK(g, hipMalloc(&g->ptr[0], sizeof(int64_t) * ((size_t)n_users + 1)));
  ICHK(g, hipMalloc(&g->ptr[1], sizeof(int64_t) * ((size_t)n_items + 1)));
  for (int sd = 0; sd < 2; ++sd) {
    ICHK(g, hipMalloc(&g->col[sd], sizeof(int32_t) * nnz1));
    ICHK(g, hipMalloc(&g->val[sd], sizeof(float) * nnz1));
  }
  return MALS_OK;
}

// what finish_impl shares with its two sort stages
struct FinishTmp {  // small per-finish device temporaries (sized by the number of distinct ids)
  unsigned *head = nullptr, *scan = nullptr, *ri = nullptr, *keep = nullptr, *present = nullptr;  // arena
  float* pair_val = nullptr;                                                   // arena
  int32_t* coo_row = nullptr;                                                  // arena
  unsigned *alive_u = nullptr, *alive_i = nullptr, *new_u = nullptr, *new_i = nullptr;
  int64_t *uid_all = nullptr, *iid_all = nullptr;
  ~FinishTmp() {
    dfree(alive_u); dfree(alive_i); dfree(new_u); dfree(new_i); dfree(uid_all); dfree(iid_all);
  }
};

// the records a sort pipeline works on: all of the ingest's, or one user range of them (ingest_big_host.h)
struct Records {
  const int64_t* user;
  const int64_t* item;
  const float* value;
  int64_t n;
};

// 1. records sorted by item id (stable: stream order inside an item); dense item rank of every position
template <typename K>
static int stage_items(mals_ingest g, Scratch& s, FinishTmp& t, const Records& rec, bool user32, int* ra_out, unsigned* n_i_all) {
  const int64_t n = rec.n;
  K* const kb[2] = {reinterpret_cast<K*>(s.keys[0]), reinterpret_cast<K*>(s.keys[1])};
  if (user32)
    hipLaunchKernelGGL((item_stage_kernel<K, true>), dim3(blocks_for(n)), dim3(256), 0, g->stream, rec.item, rec.user, rec.value, n, kb[0], s.pay64[0]);
  else
    hipLaunchKernelGGL((item_stage_kernel<K, false>), dim3(blocks_for(n)), dim3(256), 0, g->stream, rec.item, rec.user, rec.value, n, kb[0], s.pay64[0]);
  ICHK(g, hipGetLastError());
  g->bytes_moved += (12.0 + (user32 ? 8.0 : 0.0) + sizeof(K) + 8.0) * (double)n;
  int ra = 0;
  if (int rc = radix_sort<K, uint64_t>(g, s, kb, s.pay64, n, &ra)) return rc;
  hipLaunchKernelGGL(heads_kernel<K>, dim3(blocks_for(n)), dim3(256), 0, g->stream, kb[ra], n, t.head);
  ICHK(g, hipGetLastError());
  if (int rc = scan_u32(g, s, t.head, t.scan, n, n_i_all)) return rc;
  ICHK(g, hipMalloc(&t.iid_all, sizeof(int64_t) * (size_t)*n_i_all));
  hipLaunchKernelGGL(position_ranks_kernel<K>, dim3(blocks_for(n)), dim3(256), 0, g->stream, kb[ra], t.head, t.scan, n, t.ri, t.iid_all);
  ICHK(g, hipGetLastError());
  g->bytes_moved += (sizeof(K) + 4.0 + sizeof(K) + 8.0 + 4.0) * (double)n;
  *ra_out = ra;
  return MALS_OK;
}

// 2. ... then by user id (stable again): the records are now ordered by (user, item, stream order) -- one sort of the
//    composite key in two stable stages, with the item rank and the value riding along; 3. pair keys
//    (user rank << 32 | item rank) and values in that order.  *r_out = the key buffer that holds the pair keys.
template <typename K>
static int stage_users(mals_ingest g, Scratch& s, FinishTmp& t, const Records& rec, int ra, float* sorted_val, int* r_out, unsigned* n_u_all) {
  const int64_t n = rec.n;
  constexpr bool USER32 = sizeof(K) == 4;
  K* const kb[2] = {reinterpret_cast<K*>(s.keys[0]), reinterpret_cast<K*>(s.keys[1])};
  // (reads pay64[ra][i] and writes pay64[0][i]: the same thread, the same index -- safe when ra == 0)
  hipLaunchKernelGGL((user_stage_kernel<K, USER32>), dim3(blocks_for(n)), dim3(256), 0, g->stream, rec.user, s.pay64[ra], t.ri, n, kb[0], s.pay64[0]);
  ICHK(g, hipGetLastError());
  g->bytes_moved += (8.0 + 4.0 + (USER32 ? 0.0 : 8.0) + sizeof(K) + 8.0) * (double)n;
  int rb = 0;
  if (int rc = radix_sort<K, uint64_t>(g, s, kb, s.pay64, n, &rb)) return rc;
  hipLaunchKernelGGL(heads_kernel<K>, dim3(blocks_for(n)), dim3(256), 0, g->stream, kb[rb], n, t.head);
  ICHK(g, hipGetLastError());
  if (int rc = scan_u32(g, s, t.head, t.scan, n, n_u_all)) return rc;
  ICHK(g, hipMalloc(&t.uid_all, sizeof(int64_t) * (size_t)*n_u_all));
  const int r = 1 - rb;  // the other key buffer is free now
  hipLaunchKernelGGL(pair_from_sorted_kernel<K>, dim3(blocks_for(n)), dim3(256), 0, g->stream, kb[rb], s.pay64[rb], t.head, t.scan, n, s.keys[r],
                     sorted_val, t.uid_all);
  ICHK(g, hipGetLastError());
  g->bytes_moved += (sizeof(K) + 4.0 + sizeof(K) + 8.0 + 4.0 + 4.0 + 8.0 + 4.0) * (double)n;
  *r_out = r;
  return MALS_OK;
}

// the arena of one sort pipeline over n records (52 bytes per record + digit counts), carved into the views the stages use;
// scan_extra: the longest scan the caller runs beside the pipeline's own
static int setup_workspace(mals_ingest g, Scratch& s, FinishTmp& t, int64_t n, int64_t scan_extra) {
  const int64_t n_blocks = (n + RS_BLOCK_TILE - 1) / RS_BLOCK_TILE;
  const int64_t scan_len = std::max<int64_t>(std::max<int64_t>(n, 256 * n_blocks), scan_extra);
  const int64_t tiles = (scan_len + SC_TILE - 1) / SC_TILE + 1;
  const size_t k8 = sizeof(uint64_t) * (size_t)n, k4 = sizeof(unsigned) * (size_t)n;
  const size_t want[mals_ingest_s::N_WS] = {k8, k8, k4, k4, k8, k8, k4, k4, k4, sizeof(unsigned) * (size_t)(256 * n_blocks),
                                            sizeof(unsigned) * (size_t)tiles, sizeof(unsigned long long) * 8 * 256 + 256};
  const auto t0 = std::chrono::steady_clock::now();
  for (int b = 0; b < mals_ingest_s::N_WS; ++b) {
    if (want[b] > g->ws_bytes[b]) {
      dfree(g->ws[b]);
      g->ws_bytes[b] = 0;
      ICHK(g, hipMalloc(&g->ws[b], want[b]));
      g->ws_bytes[b] = want[b];
    }
  }
  g->last_workspace_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  s.keys[0] = (uint64_t*)g->ws[0];
  s.keys[1] = (uint64_t*)g->ws[1];
  s.pay[0] = (unsigned*)g->ws[2];
  s.pay[1] = (unsigned*)g->ws[3];
  s.pay64[0] = (uint64_t*)g->ws[4];
  s.pay64[1] = (uint64_t*)g->ws[5];
  t.head = (unsigned*)g->ws[6];
  t.scan = (unsigned*)g->ws[7];
  t.ri = (unsigned*)g->ws[8];
  s.counts = (unsigned*)g->ws[9];
  s.tile_sums = (unsigned*)g->ws[10];
  s.digit_tot = (unsigned long long*)g->ws[11];
  s.total = (unsigned*)((char*)g->ws[11] + sizeof(unsigned long long) * 8 * 256);
  // dead after the composite sort: the 64-bit payload buffers carry the replay outputs
  t.keep = (unsigned*)s.pay64[0];
  t.pair_val = (float*)((char*)s.pay64[0] + k4);
  t.coo_row = (int32_t*)s.pay64[1];
  t.present = g->want_known ? (unsigned*)((char*)s.pay64[1] + k4) : nullptr;  // coo_row needs 4 bytes per entry at most
  return MALS_OK;
}

#include "ingest_big_host.h"

// 8. tag id sets (IFR:159-165): sorted, unique; 9. userTagIDs as rows of R^T.  sort_cap: what the arena's sort buffers hold
static int finish_tags(mals_ingest g, Scratch& s, FinishTmp& t, int64_t sort_cap) {
  const int64_t n_items = g->n_items;
  for (int which = 0; which < 2; ++which) {
    const int64_t nt = (int64_t)g->n_tags_raw[which];
    if (nt == 0) continue;  // nt <= n: every tag comes from a record
    if (nt > sort_cap) return fail(g, MALS_INVALID_ARG, "ingest: more tag lines than a partition holds records");
    hipLaunchKernelGGL(ids_to_keys_kernel, dim3(blocks_for(nt)), dim3(256), 0, g->stream, g->d_tags[which], nt, s.keys[0], s.pay[0]);
    ICHK(g, hipGetLastError());
    int rt = 0;
    if (int rc = radix_sort<uint64_t, unsigned>(g, s, s.keys, s.pay, nt, &rt)) return rc;
    hipLaunchKernelGGL(heads_kernel<uint64_t>, dim3(blocks_for(nt)), dim3(256), 0, g->stream, s.keys[rt], nt, t.head);
    ICHK(g, hipGetLastError());
    unsigned n_unique = 0;
    if (int rc = scan_u32(g, s, t.head, t.scan, nt, &n_unique)) return rc;
    ICHK(g, hipMalloc(&g->tag_ids[which], sizeof(int64_t) * (size_t)n_unique));
    g->n_tag_ids[which] = n_unique;
    hipLaunchKernelGGL(unique_ids_kernel, dim3(blocks_for(nt)), dim3(256), 0, g->stream, s.keys[rt], t.head, t.scan, nt, g->tag_ids[which]);
    ICHK(g, hipGetLastError());
  }
  // 9. userTagIDs as rows of R^T: what top-N must never return (RecommendIterator.java:72)
  if (g->n_tag_ids[1] > 0) {
    ICHK(g, hipMalloc(&g->tag_item_idx, sizeof(int64_t) * (size_t)g->n_tag_ids[1]));
    hipLaunchKernelGGL(index_of_ids_kernel, dim3(blocks_for(g->n_tag_ids[1])), dim3(256), 0, g->stream, g->tag_ids[1], g->n_tag_ids[1], g->ids[1],
                       (int64_t)n_items, g->tag_item_idx);
    ICHK(g, hipGetLastError());
  }
  return MALS_OK;
}

static int finish_impl(mals_ingest g, hipEvent_t e0) {
  const int64_t n = g->n;
  g->last_partitions = g->last_item_ranges = 0;
  if (n == 0) {
    ICHK(g, hipEventRecord(e0, g->stream));
    if (int rc = alloc_results(g, 0, 0, 0)) return rc;
    ICHK(g, hipMemsetAsync(g->ptr[0], 0, sizeof(int64_t), g->stream));
    ICHK(g, hipMemsetAsync(g->ptr[1], 0, sizeof(int64_t), g->stream));
    if (g->want_known) {
      ICHK(g, hipMalloc(&g->known_ptr, sizeof(int64_t)));
      ICHK(g, hipMalloc(&g->known_idx, sizeof(int32_t)));
      ICHK(g, hipMemsetAsync(g->known_ptr, 0, sizeof(int64_t), g->stream));
    }
    return MALS_OK;
  }
  // more records than one sort pipeline holds (or than the caller wants it to hold): user range by user range
  if (n > MALS_INGEST_ONE_SHOT_MAX || (g->part_cap > 0 && n > g->part_cap))
    return finish_big(g, e0, g->part_cap);   // 0: ranges as large as the free memory holds
  FinishTmp t;
  Scratch s;
  unsigned n_u_all = 0, n_i_all = 0, n_users = 0, n_items = 0, nnz = 0;
  if (int rc = setup_workspace(g, s, t, n, 0)) return rc;
  ICHK(g, hipEventRecord(e0, g->stream));  // the pipeline proper starts here
  // 0. do the ids fit 32 bits?  Then the sorts run on 32-bit keys, and the user ids ride through the first sort so that
  //    nothing has to be gathered by record index
  ICHK(g, hipMemsetAsync(s.digit_tot, 0, 2 * sizeof(unsigned long long), g->stream));
  hipLaunchKernelGGL(ids_high_bits_kernel, dim3(blocks_for(n, 256, 8192)), dim3(256), 0, g->stream, g->d_user, n, s.digit_tot);
  hipLaunchKernelGGL(ids_high_bits_kernel, dim3(blocks_for(n, 256, 8192)), dim3(256), 0, g->stream, g->d_item, n, s.digit_tot + 1);
  unsigned long long high[2] = {1, 1};
  ICHK(g, hipMemcpyAsync(high, s.digit_tot, sizeof(high), hipMemcpyDeviceToHost, g->stream));
  ICHK(g, hipStreamSynchronize(g->stream));
  const bool narrow = !std::getenv("MALS_INGEST_WIDE_KEYS");   // A/B and test switch
  const bool user32 = high[0] == 0 && narrow, item32 = high[1] == 0 && narrow;
  g->bytes_moved += 16.0 * (double)n;
  int ra = 0;
  const Records rec = {g->d_user, g->d_item, g->d_value, n};
  if (int rc = item32 ? stage_items<uint32_t>(g, s, t, rec, user32, &ra, &n_i_all) : stage_items<uint64_t>(g, s, t, rec, user32, &ra, &n_i_all)) return rc;
  float* sorted_val = reinterpret_cast<float*>(s.pay[0]);
  int r = 0;
  if (int rc = user32 ? stage_users<uint32_t>(g, s, t, rec, ra, sorted_val, &r, &n_u_all) : stage_users<uint64_t>(g, s, t, rec, ra, sorted_val, &r, &n_u_all))
    return rc;
  // 4. replay every pair's records in order
  ICHK(g, hipMalloc(&t.alive_u, sizeof(unsigned) * (size_t)n_u_all));
  ICHK(g, hipMalloc(&t.alive_i, sizeof(unsigned) * (size_t)n_i_all));
  ICHK(g, hipMalloc(&t.new_u, sizeof(unsigned) * (size_t)n_u_all));
  ICHK(g, hipMalloc(&t.new_i, sizeof(unsigned) * (size_t)n_i_all));
  ICHK(g, hipMemsetAsync(t.alive_u, 0, sizeof(unsigned) * (size_t)n_u_all, g->stream));
  ICHK(g, hipMemsetAsync(t.alive_i, 0, sizeof(unsigned) * (size_t)n_i_all, g->stream));
  hipLaunchKernelGGL(replay_pairs_kernel, dim3(blocks_for(n)), dim3(256), 0, g->stream, s.keys[r], sorted_val, n, g->zero_threshold,
                     t.keep, t.pair_val, t.alive_u, t.alive_i, t.present);
  ICHK(g, hipGetLastError());
  g->bytes_moved += (8.0 + 4.0 + 8.0 + (t.present ? 4.0 : 0.0)) * (double)n;
  // 5. ids that still own an entry, renumbered densely (ascending id)
  if (int rc = scan_u32(g, s, t.alive_u, t.new_u, n_u_all, &n_users)) return rc;
  if (int rc = scan_u32(g, s, t.alive_i, t.new_i, n_i_all, &n_items)) return rc;
  // 6. surviving entries (|value| >= threshold): already sorted by (user, item)
  if (int rc = scan_u32(g, s, t.keep, t.scan, n, &nnz)) return rc;
  if (int rc = alloc_results(g, n_users, n_items, nnz)) return rc;
  hipLaunchKernelGGL(compact_ids_kernel, dim3(blocks_for(n_u_all)), dim3(256), 0, g->stream, t.uid_all, t.alive_u, t.new_u,
                     (int64_t)n_u_all, g->ids[0]);
  hipLaunchKernelGGL(compact_ids_kernel, dim3(blocks_for(n_i_all)), dim3(256), 0, g->stream, t.iid_all, t.alive_i, t.new_i,
                     (int64_t)n_i_all, g->ids[1]);
  hipLaunchKernelGGL(compact_pairs_kernel, dim3(blocks_for(n)), dim3(256), 0, g->stream, s.keys[r], t.keep, t.scan, t.pair_val, n,
                     t.new_u, t.new_i, t.coo_row, g->col[0], g->val[0]);
  hipLaunchKernelGGL(row_ptr_from_sorted_kernel, dim3(blocks_for((int64_t)nnz + 1)), dim3(256), 0, g->stream, t.coo_row, (int64_t)nnz,
                     (int64_t)n_users, g->ptr[0]);
  ICHK(g, hipGetLastError());
  g->bytes_moved += 16.0 * (double)n + 16.0 * (double)nnz + 8.0 * (double)n_users;
  // 6b. knownItemIDs (IFR:173-191): the pairs present at the end of the stream, pruned from R or not, as a CSR over
  //    the same dense indices (its users are exactly the rows of RbyRow, its items rows of RbyColumn)
  if (g->want_known) {
    unsigned n_known = 0;
    if (int rc = scan_u32(g, s, t.present, t.scan, n, &n_known)) return rc;
    g->n_known = n_known;
    ICHK(g, hipMalloc(&g->known_ptr, sizeof(int64_t) * ((size_t)n_users + 1)));
    ICHK(g, hipMalloc(&g->known_idx, sizeof(int32_t) * std::max<size_t>(n_known, 1)));
    int32_t* known_row = (int32_t*)t.ri;
    hipLaunchKernelGGL(compact_known_kernel, dim3(blocks_for(n)), dim3(256), 0, g->stream, s.keys[r], t.present, t.scan, n, t.new_u, t.new_i,
                       known_row, g->known_idx);
    hipLaunchKernelGGL(row_ptr_from_sorted_kernel, dim3(blocks_for((int64_t)n_known + 1)), dim3(256), 0, g->stream, known_row,
                       (int64_t)n_known, (int64_t)n_users, g->known_ptr);
    ICHK(g, hipGetLastError());
    g->bytes_moved += (4.0 + 12.0 + 8.0) * (double)n + 12.0 * (double)n_known + 8.0 * (double)n_users;
  }
  // 7. the transposed matrix: the entries are sorted by (user, item); a STABLE sort on the item half of
  //    the key alone leaves the users ascending inside every item, so the four low digits are not sorted
  if (nnz > 0) {
    hipLaunchKernelGGL(transpose_keys_kernel, dim3(blocks_for(nnz)), dim3(256), 0, g->stream, t.coo_row, g->col[0], g->val[0],
                       (int64_t)nnz, s.keys[0], s.pay[0]);
    ICHK(g, hipGetLastError());
    g->bytes_moved += 24.0 * (double)nnz;
    int r2 = 0;
    // (the item half of the key is a dense index below n_items: no counting pass)
    if (int rc = radix_sort<uint64_t, unsigned>(g, s, s.keys, s.pay, (int64_t)nnz, &r2, 4, ((uint64_t)(n_items > 0 ? n_items - 1 : 0) << 32) | 0xffffffffull)) return rc;
    hipLaunchKernelGGL(transpose_gather_kernel, dim3(blocks_for(nnz)), dim3(256), 0, g->stream, s.keys[r2], s.pay[r2], (int64_t)nnz,
                       t.coo_row, g->col[1], g->val[1]);
    ICHK(g, hipGetLastError());
    g->bytes_moved += 24.0 * (double)nnz;
  }
  hipLaunchKernelGGL(row_ptr_from_sorted_kernel, dim3(blocks_for((int64_t)nnz + 1)), dim3(256), 0, g->stream, t.coo_row, (int64_t)nnz,
                     (int64_t)n_items, g->ptr[1]);
  ICHK(g, hipGetLastError());
  g->bytes_moved += 4.0 * (double)nnz + 8.0 * (double)n_items;
  if (int rc = finish_tags(g, s, t, n)) return rc;
  ICHK(g, hipStreamSynchronize(g->stream));
  return MALS_OK;
}

extern "C" {

int mals_ingest_finish(mals_ingest g) {
  if (!g) return MALS_INVALID_ARG;
  if (g->text_failed) return fail(g, g->text_fail_code, g->text_fail_msg);
  if (g->carry_len) return fail(g, MALS_INVALID_ARG, "text pending: the last mals_ingest_append_text of a file must say end_of_file");
  ICHK(g, hipSetDevice(g->device));
  free_results(g);
  g->bytes_moved = 0.0;
  g->radix_passes = 0;
  hipEvent_t e0, e1;
  ICHK(g, hipEventCreate(&e0));
  ICHK(g, hipEventCreate(&e1));
  g->last_workspace_ms = 0.0;
  const int rc = finish_impl(g, e0);
  if (rc != MALS_OK) {
    free_results(g);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
  }
  ICHK(g, hipEventRecord(e1, g->stream));
  ICHK(g, hipEventSynchronize(e1));
  float ms = 0.f;
  ICHK(g, hipEventElapsedTime(&ms, e0, e1));
  g->last_finish_ms = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  g->finished = true;
  return MALS_OK;
}

int mals_ingest_counts(mals_ingest g, int64_t* n_records, int64_t* n_users, int64_t* n_items, int64_t* nnz) {
  if (!g) return MALS_INVALID_ARG;
  if (n_records) *n_records = g->n;
  if (!g->finished && (n_users || n_items || nnz)) return fail(g, MALS_INVALID_ARG, "mals_ingest_finish has not run");
  if (n_users) *n_users = g->n_users;
  if (n_items) *n_items = g->n_items;
  if (nnz) *nnz = g->nnz;
  return MALS_OK;
}

int mals_ingest_get_ids(mals_ingest g, int side, int64_t* host_ids_out) {
  if (!g || !host_ids_out) return MALS_INVALID_ARG;
  if (side != MALS_SIDE_X && side != MALS_SIDE_Y) return fail(g, MALS_INVALID_ARG, "side must be MALS_SIDE_X or _Y");
  if (!g->finished) return fail(g, MALS_INVALID_ARG, "mals_ingest_finish has not run");
  ICHK(g, hipSetDevice(g->device));
  const int64_t cnt = side == MALS_SIDE_X ? g->n_users : g->n_items;
  if (cnt) ICHK(g, hipMemcpy(host_ids_out, g->ids[side], sizeof(int64_t) * (size_t)cnt, hipMemcpyDeviceToHost));
  return MALS_OK;
}

int mals_ingest_get_csr(mals_ingest g, int side, int64_t* host_row_ptr, int32_t* host_col_idx, float* host_val) {
  if (!g) return MALS_INVALID_ARG;
  if (side != MALS_SIDE_X && side != MALS_SIDE_Y) return fail(g, MALS_INVALID_ARG, "side must be MALS_SIDE_X or _Y");
  if (!g->finished) return fail(g, MALS_INVALID_ARG, "mals_ingest_finish has not run");
  ICHK(g, hipSetDevice(g->device));
  const int64_t rows = side == MALS_SIDE_X ? g->n_users : g->n_items;
  if (host_row_ptr) ICHK(g, hipMemcpy(host_row_ptr, g->ptr[side], sizeof(int64_t) * (size_t)(rows + 1), hipMemcpyDeviceToHost));
  if (host_col_idx && g->nnz) ICHK(g, hipMemcpy(host_col_idx, g->col[side], sizeof(int32_t) * (size_t)g->nnz, hipMemcpyDeviceToHost));
  if (host_val && g->nnz) ICHK(g, hipMemcpy(host_val, g->val[side], sizeof(float) * (size_t)g->nnz, hipMemcpyDeviceToHost));
  return MALS_OK;
}

int mals_ingest_device_csr(mals_ingest g, int side, const int64_t** row_ptr, const int32_t** col_idx, const float** val) {
  if (!g) return MALS_INVALID_ARG;
  if (side != MALS_SIDE_X && side != MALS_SIDE_Y) return fail(g, MALS_INVALID_ARG, "side must be MALS_SIDE_X or _Y");
  if (!g->finished) return fail(g, MALS_INVALID_ARG, "mals_ingest_finish has not run");
  if (row_ptr) *row_ptr = g->ptr[side];
  if (col_idx) *col_idx = g->col[side];
  if (val) *val = g->val[side];
  return MALS_OK;
}

int mals_ingest_install(mals_ingest g, mals_handle h) {
  if (!g || !h) return MALS_INVALID_ARG;
  if (!g->finished) return fail(g, MALS_INVALID_ARG, "mals_ingest_finish has not run");
  if (int rc = mals_set_matrix(h, MALS_SIDE_X, 0, g->n_users, g->nnz, g->ptr[0], g->col[0], g->val[0], MALS_MEM_DEVICE))
    return fail(g, rc, mals_last_error(h));
  if (int rc = mals_set_matrix(h, MALS_SIDE_Y, 0, g->n_items, g->nnz, g->ptr[1], g->col[1], g->val[1], MALS_MEM_DEVICE))
    return fail(g, rc, mals_last_error(h));
  // knownItemIDs, if they were asked for: what mals_recommend skips (ServerRecommender.java:394-425), entries that
  // removeSmall pruned from R included
  // (mals_set_matrix(side X) has dropped whatever known items an earlier install left on the handle)
  if (g->known_ptr)
    if (int rc = mals_set_known_items(h, g->n_users, g->known_ptr, g->known_idx, MALS_MEM_DEVICE)) return fail(g, rc, mals_last_error(h));
  // userTagIDs: rows of Y top-N never returns (RecommendIterator.java:72).  Needs the item factor rows: declare them here if
  // the caller has not (a caller-bound or larger replica is left alone).
  void* fy = nullptr;
  int64_t fy_rows = 0;
  (void)mals_factor_device_ptr(h, MALS_SIDE_Y, &fy, &fy_rows);
  if (g->n_tag_ids[1] > 0 && (!fy || fy_rows < g->n_items))
    if (int rc = mals_set_factor_rows(h, MALS_SIDE_Y, g->n_items)) return fail(g, rc, mals_last_error(h));
  if (int rc = mals_set_tag_items(h, g->n_tag_ids[1], g->tag_item_idx, MALS_MEM_DEVICE)) return fail(g, rc, mals_last_error(h));
  return MALS_OK;
}

int mals_ingest_device(mals_ingest g, int32_t* device_out) {
  if (!g || !device_out) return MALS_INVALID_ARG;
  *device_out = g->device;
  return MALS_OK;
}

int mals_ingest_get_tag_items(mals_ingest g, int64_t* host_idx_out) {
  if (!g || !host_idx_out) return MALS_INVALID_ARG;
  if (!g->finished) return fail(g, MALS_INVALID_ARG, "mals_ingest_finish has not run");
  ICHK(g, hipSetDevice(g->device));
  if (g->n_tag_ids[1]) ICHK(g, hipMemcpy(host_idx_out, g->tag_item_idx, sizeof(int64_t) * (size_t)g->n_tag_ids[1], hipMemcpyDeviceToHost));
  return MALS_OK;
}

int mals_ingest_device_tag_items(mals_ingest g, const int64_t** device_idx_out, int64_t* n_out) {
  if (!g) return MALS_INVALID_ARG;
  if (!g->finished) return fail(g, MALS_INVALID_ARG, "mals_ingest_finish has not run");
  if (device_idx_out) *device_idx_out = g->tag_item_idx;
  if (n_out) *n_out = g->n_tag_ids[1];
  return MALS_OK;
}

int mals_ingest_partitions(mals_ingest g, int32_t* user_ranges, int32_t* item_ranges) {
  if (!g) return MALS_INVALID_ARG;
  if (user_ranges) *user_ranges = g->last_partitions;
  if (item_ranges) *item_ranges = g->last_item_ranges;
  return MALS_OK;
}

int mals_ingest_stats(mals_ingest g, double* finish_ms, double* workspace_ms, double* bytes_moved, int32_t* radix_passes) {
  if (!g) return MALS_INVALID_ARG;
  if (finish_ms) *finish_ms = g->last_finish_ms;
  if (workspace_ms) *workspace_ms = g->last_workspace_ms;
  if (bytes_moved) *bytes_moved = g->bytes_moved;
  if (radix_passes) *radix_passes = g->radix_passes;
  return MALS_OK;
}

}  // extern "C"

#include "ingest_text_host.h"
