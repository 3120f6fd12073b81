// ingest_big_host.h -- mals_ingest_finish for more records than ONE sort pipeline holds (included by ingest_api.hip).
//
// The pipeline of ingest_api.hip sorts all records at once: 32-bit positions, 52 bytes of workspace per record -- 2^31
// records at most, and 5e9 records (C5: InputFilesReader.readInputFiles, IFR:64-211, is the one entry point of every
// configuration) would need 260 GB beside 120 GB of records.  Here the same result is assembled from pieces that each fit:
//
//   1. the records are cut into P <= 250 ranges of USER id (splitters from a sorted sample of 65536 user ids, exact sizes
//      counted, P raised until every range fits `part_cap` records); a byte per record says which range it belongs to;
//   2. range by range, in ascending id order: the range's records are compacted (stable: stream order kept) into a
//      partition buffer and go through the unchanged stages of the one-shot pipeline -- composite stable sort by (user id,
//      item id), per-pair replay (MU:64-125, FBIFM:129-138), removeSmall (IFR:200-211).  Users of different ranges are
//      different users, so the range's user table, row pointers and surviving entries are final up to their offsets, which
//      are the running totals.  Items are shared between ranges: a range keeps its own ascending item table, its liveness
//      flags and writes the entries' columns as LOCAL item ranks;
//   3. the ranges' item tables are merged into one ascending table (binary-search merges, no sort), liveness is OR-ed
//      through the rank maps, the dense item index is the scan of it, and every range's columns are renumbered in place;
//   4. R^T: entries per item counted with atomics (an item's count is bounded by the users: no overflow), offsets by a
//      64-bit scan, then item range by item range (<= part_cap entries each): the entries whose item lies in the range are
//      compacted in user order, their row found by a search between the tile's first and last row, sorted stably on the
//      item half of (item << 32 | user) and written behind the item range's offset.
// Positions inside a partition stay 32-bit; everything that counts across partitions is 64-bit.  The result -- ids, both
// CSRs, values, knownItemIDs, tag sets -- is bit-identical to the one-shot pipeline's (tests/test_gpu_ingest_big.py runs
// the oracle suites through this path with a partition capacity of a few hundred records).
#pragma once

namespace mals {

constexpr int BIG_TILE = 2048;  // records (or entries) per workgroup of the selection kernels: 256 threads x 8

// part[i] = number of splitters <= user_ids[i] (splitters ascending and distinct): equal ids share a range, ranges ascend
__global__ void big_assign_part_kernel(const int64_t* __restrict__ user_ids, int64_t n, const int64_t* __restrict__ splitters, int n_split,
                                       uint8_t* __restrict__ part) {
  MALS_GRID_STRIDE(i, n) {
    const int64_t u = user_ids[i];
    int lo = 0, hi = n_split;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (splitters[mid] <= u) lo = mid + 1; else hi = mid;
    }
    part[i] = (uint8_t)lo;
  }
}
// records per range, all ranges in one read: hist[256] (block-private LDS counts, then global atomics)
__global__ __launch_bounds__(256) void big_part_histogram_kernel(const uint8_t* __restrict__ part, int64_t n, unsigned long long* __restrict__ hist) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  MALS_GRID_STRIDE(i, n) atomicAdd(&h[part[i]], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
// tile_counts[t] = records of range p in tile t
// (a thread's 8 range bytes are ONE 8-byte load -- the array is padded past n --, a thread's 8 columns two 16-byte loads:
// eight strided 1- or 4-byte loads per thread ran these kernels at 0.2-0.9 TB/s)
__device__ __forceinline__ unsigned big_match8(const uint8_t* __restrict__ part, int64_t b, int64_t n, uint8_t p, bool (&m)[8]) {
  const uint64_t w = b < n ? *reinterpret_cast<const uint64_t*>(part + b) : 0ull;
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    m[j] = b + j < n && (uint8_t)(w >> (8 * j)) == p;
    c += m[j] ? 1u : 0u;
  }
  return c;
}
__device__ __forceinline__ void big_load8(const int32_t* __restrict__ col, int64_t e, int64_t nnz, int32_t (&c)[8]) {
  if (e + 8 <= nnz) {
    const int4 lo = *reinterpret_cast<const int4*>(col + e), hi = *reinterpret_cast<const int4*>(col + e + 4);
    c[0] = lo.x, c[1] = lo.y, c[2] = lo.z, c[3] = lo.w, c[4] = hi.x, c[5] = hi.y, c[6] = hi.z, c[7] = hi.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = e + j < nnz ? col[e + j] : -1;
  }
}
__global__ __launch_bounds__(256) void big_count_part_kernel(const uint8_t* __restrict__ part, int64_t n, uint8_t p, unsigned* __restrict__ tile_counts) {
  const int64_t b = (int64_t)blockIdx.x * BIG_TILE + threadIdx.x * 8;
  bool m[8];
  const unsigned c = big_match8(part, b, n, p, m);
  unsigned total;
  block_exclusive_scan(c, &total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}
// the records of range p, in stream order, into the partition buffers (tile_offsets = exclusive scan of the tile counts)
__global__ __launch_bounds__(256) void big_compact_part_kernel(const uint8_t* __restrict__ part, int64_t n, uint8_t p,
                                                               const unsigned* __restrict__ tile_offsets, const int64_t* __restrict__ user,
                                                               const int64_t* __restrict__ item, const float* __restrict__ value,
                                                               int64_t* __restrict__ pu, int64_t* __restrict__ pi, float* __restrict__ pv) {
  const int64_t b = (int64_t)blockIdx.x * BIG_TILE + threadIdx.x * 8;
  bool m[8];
  const unsigned c = big_match8(part, b, n, p, m);
  unsigned pos = tile_offsets[blockIdx.x] + block_exclusive_scan(c, nullptr);
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (m[j]) {
      pu[pos] = user[b + j];
      pi[pos] = item[b + j];
      pv[pos] = value[b + j];
      ++pos;
    }
}
// every S-th user id as a sort key
__global__ void big_sample_kernel(const int64_t* __restrict__ user_ids, int64_t n, int64_t n_sample, uint64_t* __restrict__ keys, unsigned* __restrict__ pay) {
  MALS_GRID_STRIDE(j, n_sample) {
    int64_t i = (int64_t)((double)j * ((double)n / (double)n_sample));   // (any spread of positions does: it is a sample)
    i = i < 0 ? 0 : (i >= n ? n - 1 : i);
    keys[j] = id_to_key(user_ids[i]);
    pay[j] = 0u;
  }
}
__global__ void big_keys_to_ids_kernel(const uint64_t* __restrict__ keys, int64_t n, int64_t* __restrict__ ids) {
  MALS_GRID_STRIDE(i, n) ids[i] = key_to_id(keys[i]);
}
// the kept pairs of a partition: local row (dense among the partition's live users), LOCAL item rank as the column
__global__ void big_compact_pairs_kernel(const uint64_t* __restrict__ keys, const unsigned* __restrict__ keep, const unsigned* __restrict__ keep_scan,
                                         const float* __restrict__ pair_val, int64_t n, const unsigned* __restrict__ new_u,
                                         int32_t* __restrict__ row, int32_t* __restrict__ col, float* __restrict__ val) {
  MALS_GRID_STRIDE(i, n) {
    if (!keep[i]) continue;
    const unsigned q = keep_scan[i];
    row[q] = (int32_t)new_u[(unsigned)(keys[i] >> 32)];
    col[q] = (int32_t)(unsigned)(keys[i] & 0xffffffffu);
    if (val) val[q] = pair_val[i];
  }
}
// out[r] = base + local[r]
__global__ void big_add_base_kernel(const int64_t* __restrict__ local, int64_t n, int64_t base, int64_t* __restrict__ out) {
  MALS_GRID_STRIDE(i, n) out[i] = base + local[i];
}
// ---- merging ascending id tables without a sort ------------------------------------------------------------------------
// lower_bound of x in the ascending table
__device__ __forceinline__ int64_t big_lower_bound(const int64_t* __restrict__ table, int64_t n, int64_t x) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (table[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// fresh[i] = 1 where b[i] is not in a
__global__ void big_fresh_kernel(const int64_t* __restrict__ b, int64_t nb, const int64_t* __restrict__ a, int64_t na, unsigned* __restrict__ fresh) {
  MALS_GRID_STRIDE(i, nb) {
    const int64_t q = big_lower_bound(a, na, b[i]);
    fresh[i] = (q < na && a[q] == b[i]) ? 0u : 1u;
  }
}
__global__ void big_compact_fresh_kernel(const int64_t* __restrict__ b, int64_t nb, const unsigned* __restrict__ fresh, const unsigned* __restrict__ fresh_scan,
                                         int64_t* __restrict__ out) {
  MALS_GRID_STRIDE(i, nb)
    if (fresh[i]) out[fresh_scan[i]] = b[i];
}
// union of two DISJOINT ascending tables: an element's position = its own index + how many of the other table precede it
__global__ void big_merge_place_kernel(const int64_t* __restrict__ mine, int64_t n_mine, const int64_t* __restrict__ other, int64_t n_other,
                                       int64_t* __restrict__ out) {
  MALS_GRID_STRIDE(i, n_mine) out[i + big_lower_bound(other, n_other, mine[i])] = mine[i];
}
// alive_glob[map[l]] |= alive_local[l]
__global__ void big_mark_alive_kernel(const unsigned* __restrict__ alive_local, const int64_t* __restrict__ map, int64_t n_local,
                                      unsigned* __restrict__ alive_glob) {
  MALS_GRID_STRIDE(l, n_local)
    if (alive_local[l] && __atomic_load_n(&alive_glob[map[l]], __ATOMIC_RELAXED) == 0u) alive_glob[map[l]] = 1u;
}
// local item rank -> dense item index of the result
__global__ void big_final_map_kernel(const int64_t* __restrict__ map, int64_t n_local, const unsigned* __restrict__ new_i, int32_t* __restrict__ out) {
  MALS_GRID_STRIDE(l, n_local) out[l] = (int32_t)new_i[map[l]];
}
__global__ void big_remap_kernel(int32_t* __restrict__ col, int64_t n, const int32_t* __restrict__ final_map) {
  MALS_GRID_STRIDE(i, n) col[i] = final_map[col[i]];
}
// ---- R^T ------------------------------------------------------------------------------------------------------------------
// entries per item, from every `stride`-th entry: the item ranges of R^T are CUT by this estimate (an exact count of all
// entries is 5e9 atomics, a fifth of them on a few thousand popular items: 324 ms at 2.5e9 entries, a quarter of the whole
// finish); what a range really holds is counted exactly when it is selected, and R^T's offsets come out of the sorted ranges
__global__ void big_item_sample_kernel(const int32_t* __restrict__ col, int64_t nnz, int64_t stride, unsigned* __restrict__ cnt) {
  const int64_t n_s = (nnz + stride - 1) / stride;
  MALS_GRID_STRIDE(i, n_s) atomicAdd(&cnt[col[i * stride]], 1u);
}
// exclusive scan uint32 -> int64 offsets (out has n + 1 entries): tile sums, one block over them, apply
__global__ __launch_bounds__(256) void big_scan64_reduce_kernel(const unsigned* __restrict__ in, int64_t n, unsigned long long* __restrict__ tile_sums) {
  const int64_t b = (int64_t)blockIdx.x * SC_TILE + threadIdx.x * 8;
  unsigned v[8], s = 0;
  scan_load8(in, b, n, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];   // a tile holds 2048 counts below 2^31 / 2048 each in every real input; summed in 64 bits below
  __shared__ unsigned long long acc;
  if (threadIdx.x == 0) acc = 0;
  __syncthreads();
  atomicAdd(&acc, (unsigned long long)s);
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = acc;
}
__global__ void big_scan64_sums_kernel(unsigned long long* __restrict__ sums, int64_t n_tiles, unsigned long long* __restrict__ grand_total) {
  // one thread: a few million additions at most (n_items / 2048 tiles)
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  unsigned long long run = 0;
  for (int64_t t = 0; t < n_tiles; ++t) {
    const unsigned long long v = sums[t];
    sums[t] = run;
    run += v;
  }
  *grand_total = run;
}
__global__ __launch_bounds__(256) void big_scan64_apply_kernel(const unsigned* __restrict__ in, int64_t n, const unsigned long long* __restrict__ tile_offsets,
                                                               const unsigned long long* __restrict__ grand_total, int64_t* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * SC_TILE + threadIdx.x * 8;
  unsigned v[8], s = 0;
  scan_load8(in, b, n, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  unsigned long long pre = tile_offsets[blockIdx.x] + (unsigned long long)block_exclusive_scan(s, nullptr);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (b + i < n) out[b + i] = (int64_t)pre;
    pre += v[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (int64_t)*grand_total;
}
// tile_counts[t] = entries of tile t whose item lies in [a, b)
__global__ __launch_bounds__(256) void big_count_items_kernel(const int32_t* __restrict__ col, int64_t nnz, int32_t a, int32_t b,
                                                              unsigned* __restrict__ tile_counts) {
  const int64_t e = (int64_t)blockIdx.x * BIG_TILE + threadIdx.x * 8;
  int32_t cc[8];
  big_load8(col, e, nnz, cc);
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) c += (cc[j] >= a && cc[j] < b) ? 1u : 0u;
  unsigned total;
  block_exclusive_scan(c, &total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}
// the row of entry e: the last r with row_ptr[r] <= e, searched in [lo, hi]
__device__ __forceinline__ int64_t big_row_of(const int64_t* __restrict__ row_ptr, int64_t lo, int64_t hi, int64_t e) {
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo + 1) / 2;
    if (row_ptr[mid] <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// tile_row[t] = the row of the first entry of entry tile t (tile_row[n_tiles] = the last row): once per finish, one thread per
// tile.  (Searched inside big_select_items_kernel by one thread per workgroup it was 50 dependent loads in front of a
// barrier, per tile and per item range: 20 ms per range at 2.5e9 entries, a third of the finish.)
__global__ void big_tile_rows_kernel(const int64_t* __restrict__ row_ptr, int64_t n_rows, int64_t nnz, int64_t n_tiles, int32_t* __restrict__ tile_row) {
  MALS_GRID_STRIDE(t, n_tiles + 1)
    tile_row[t] = (t == n_tiles || t * BIG_TILE >= nnz) ? (int32_t)(n_rows - 1) : (int32_t)big_row_of(row_ptr, 0, n_rows - 1, t * BIG_TILE);
}
// the selected entries, in (user, item) order, as sort keys ((item - a) << 32 | user) with the value bits as payload
__global__ __launch_bounds__(256) void big_select_items_kernel(const int32_t* __restrict__ col, const float* __restrict__ val, int64_t nnz,
                                                               const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ tile_row, int32_t a, int32_t b,
                                                               const unsigned* __restrict__ tile_offsets, uint64_t* __restrict__ keys,
                                                               unsigned* __restrict__ pay) {
  const int64_t e0 = (int64_t)blockIdx.x * BIG_TILE;
  const int64_t r_lo = tile_row[blockIdx.x], r_hi = tile_row[blockIdx.x + 1];
  const int64_t e = e0 + threadIdx.x * 8;
  int32_t cc[8];
  big_load8(col, e, nnz, cc);
  bool m[8];
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    m[j] = cc[j] >= a && cc[j] < b;   // (entries past nnz read as -1: outside every range)
    c += m[j] ? 1u : 0u;
  }
  unsigned pos = tile_offsets[blockIdx.x] + block_exclusive_scan(c, nullptr);
  int64_t row = -1;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (m[j]) {
      // (a thread's 8 entries are consecutive: the row only moves forward)
      if (row < 0 || row_ptr[row + 1] <= e + j) row = big_row_of(row_ptr, row < 0 ? r_lo : row, r_hi, e + j);
      keys[pos] = ((uint64_t)(uint32_t)(cc[j] - a) << 32) | (uint32_t)row;
      pay[pos] = __float_as_uint(val[e + j]);
      ++pos;
    }
}
__global__ void big_transpose_write_kernel(const uint64_t* __restrict__ keys, const unsigned* __restrict__ pay, int64_t n, int32_t* __restrict__ t_row,
                                           int32_t* __restrict__ t_col, float* __restrict__ t_val) {
  MALS_GRID_STRIDE(i, n) {
    t_row[i] = (int32_t)(keys[i] >> 32);          // the item, counted from the range's first
    t_col[i] = (int32_t)(keys[i] & 0xffffffffu);  // the user
    t_val[i] = __uint_as_float(pay[i]);
  }
}
__global__ void big_add_base_inplace_kernel(int64_t* __restrict__ p, int64_t n, int64_t base) {
  MALS_GRID_STRIDE(i, n) p[i] += base;
}

}  // namespace mals

// ---- host ----------------------------------------------------------------------------------------------------------------
namespace {

struct BigPart {  // what a user range leaves behind until the items are known
  int64_t n_records = 0, n_users = 0, nnz = 0, n_known = 0;
  int64_t u_base = 0, nnz_base = 0, known_base = 0;
  int64_t n_items_local = 0;
  int64_t* item_ids = nullptr;     // the range's ascending item table
  unsigned* item_alive = nullptr;  // ... and which of them own an entry here
  int64_t* user_ids = nullptr;     // live users, ascending
  int64_t* ptr_local = nullptr;    // [n_users + 1] offsets into the range's entries
  int64_t* known_ptr_local = nullptr;
};

struct BigState {
  std::vector<BigPart> parts;
  uint8_t* part = nullptr;
  int64_t *pu = nullptr, *pi = nullptr;
  float* pv = nullptr;
  int64_t* d_split = nullptr;
  int64_t* item_glob = nullptr;   // merged ascending item table
  int64_t n_item_glob = 0;
  unsigned *alive_glob = nullptr, *new_i = nullptr, *cnt = nullptr;
  unsigned long long* sums64 = nullptr;
  int32_t* tile_row = nullptr;
  ~BigState() {
    dfree(tile_row);
    for (BigPart& p : parts) {
      dfree(p.item_ids); dfree(p.item_alive); dfree(p.user_ids); dfree(p.ptr_local); dfree(p.known_ptr_local);
    }
    dfree(part); dfree(pu); dfree(pi); dfree(pv); dfree(d_split); dfree(item_glob); dfree(alive_glob); dfree(new_i); dfree(cnt); dfree(sums64);
  }
};

unsigned big_tiles(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + mals::BIG_TILE - 1) / mals::BIG_TILE); }

// a = a U b for ascending tables (b's duplicates of a dropped); a is reallocated
int big_merge_tables(mals_ingest g, Scratch& s, FinishTmp& t, int64_t*& a, int64_t& na, const int64_t* b, int64_t nb) {
  if (nb == 0) return MALS_OK;
  if (na == 0) {
    dfree(a);
    ICHK(g, hipMalloc(&a, sizeof(int64_t) * (size_t)nb));
    ICHK(g, hipMemcpyAsync(a, b, sizeof(int64_t) * (size_t)nb, hipMemcpyDeviceToDevice, g->stream));
    na = nb;
    return MALS_OK;
  }
  hipLaunchKernelGGL(big_fresh_kernel, dim3(blocks_for(nb)), dim3(256), 0, g->stream, b, nb, a, na, t.head);
  ICHK(g, hipGetLastError());
  unsigned n_fresh = 0;
  if (int rc = scan_u32(g, s, t.head, t.scan, nb, &n_fresh)) return rc;
  if (n_fresh == 0) return MALS_OK;
  int64_t *fresh = nullptr, *merged = nullptr;
  ICHK(g, hipMalloc(&fresh, sizeof(int64_t) * (size_t)n_fresh));
  if (hipMalloc(&merged, sizeof(int64_t) * (size_t)(na + n_fresh)) != hipSuccess) {
    dfree(fresh);
    return fail(g, MALS_OOM, "item table merge: out of device memory");
  }
  hipLaunchKernelGGL(big_compact_fresh_kernel, dim3(blocks_for(nb)), dim3(256), 0, g->stream, b, nb, t.head, t.scan, fresh);
  hipLaunchKernelGGL(big_merge_place_kernel, dim3(blocks_for(na)), dim3(256), 0, g->stream, a, na, fresh, (int64_t)n_fresh, merged);
  hipLaunchKernelGGL(big_merge_place_kernel, dim3(blocks_for(n_fresh)), dim3(256), 0, g->stream, fresh, (int64_t)n_fresh, a, na, merged);
  const hipError_t e = hipGetLastError();
  const hipError_t e2 = hipStreamSynchronize(g->stream);
  dfree(fresh);
  dfree(a);
  a = merged;
  na += n_fresh;
  ICHK(g, e);
  ICHK(g, e2);
  g->bytes_moved += 24.0 * (double)nb + 16.0 * (double)na;
  return MALS_OK;
}

// splitters for `n_parts` ranges from the sorted sample (host): strictly ascending, at most n_parts - 1
std::vector<int64_t> big_pick_splitters(const std::vector<int64_t>& sample, int n_parts) {
  std::vector<int64_t> sp;
  const size_t S = sample.size();
  for (int j = 1; j < n_parts; ++j) {
    const int64_t c = sample[std::min(S - 1, (size_t)((unsigned long long)j * S / (unsigned)n_parts))];
    if (sp.empty() || c > sp.back()) sp.push_back(c);
  }
  return sp;
}

}  // namespace

static int setup_workspace(mals_ingest g, Scratch& s, FinishTmp& t, int64_t n, int64_t scan_extra);
static int finish_tags(mals_ingest g, Scratch& s, FinishTmp& t, int64_t sort_cap);

// part_cap = 0: as many records per range as the device's free memory holds (fewer ranges = fewer sweeps over the records)
static int finish_big(mals_ingest g, hipEvent_t e0, int64_t part_cap) {
  using namespace mals;
  const int64_t n = g->n;
  BigState B;
  // ---- 0. ranges of user id ------------------------------------------------------------------------------------------
  const int64_t n_sample = std::min<int64_t>(n, 65536);
  Scratch s;
  FinishTmp t;
  const auto t_ws = std::chrono::steady_clock::now();
  // results that are filled range by range: R by user with room for every record (entries <= records), knownItemIDs likewise
  ICHK(g, hipMalloc(&B.part, (size_t)n + 8));
  ICHK(g, hipMalloc(&B.d_split, sizeof(int64_t) * 256));
  ICHK(g, hipMalloc(&g->col[0], sizeof(int32_t) * (size_t)n));
  ICHK(g, hipMalloc(&g->val[0], sizeof(float) * (size_t)n));
  if (g->want_known) ICHK(g, hipMalloc(&g->known_idx, sizeof(int32_t) * (size_t)n));
  // ... and R by item likewise, now: a 20 GB hipMalloc in the middle of the pipeline is a second of host time on some boxes
  ICHK(g, hipMalloc(&g->col[1], sizeof(int32_t) * (size_t)n));
  ICHK(g, hipMalloc(&g->val[1], sizeof(float) * (size_t)n));
  if (part_cap <= 0) {
    // A range costs 76 bytes per record (24 in the partition buffer, 52 of sort workspace).  Of what is free now -- plus the
    // arena an earlier finish left, which is reused -- 70 % go to it: the id tables, the ranges' item tables and the row
    // bounds of R^T (8-16 bytes per user / item) come out of the rest.
    size_t free_b = 0, total_b = 0;
    ICHK(g, hipMemGetInfo(&free_b, &total_b));
    size_t arena = 0;
    for (int b = 0; b < mals_ingest_s::N_WS; ++b) arena += g->ws_bytes[b];
    const double budget = 0.7 * ((double)free_b + (double)arena);
    part_cap = (int64_t)std::min<double>((double)MALS_INGEST_ONE_SHOT_MAX, std::max<double>((double)MALS_INGEST_MIN_PART, budget / 76.0));
    // (no point in a range larger than the whole input spread over two ranges)
    part_cap = std::min<int64_t>(part_cap, std::max<int64_t>(MALS_INGEST_MIN_PART, n / 2 + n / 8));
  }
  // the workspace of one partition; the scans over all records run on tiles of 2048 (one count per tile) and the dense item
  // index is a scan over at most 2^31 items: both inside the same tile-sum buffer
  if (int rc = setup_workspace(g, s, t, std::max<int64_t>(part_cap, n_sample), std::max<int64_t>((n + BIG_TILE - 1) / BIG_TILE, (int64_t)1 << 31))) return rc;
  ICHK(g, hipMalloc(&B.pu, sizeof(int64_t) * (size_t)part_cap));
  ICHK(g, hipMalloc(&B.pi, sizeof(int64_t) * (size_t)part_cap));
  ICHK(g, hipMalloc(&B.pv, sizeof(float) * (size_t)part_cap));
  g->last_workspace_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ws).count();
  ICHK(g, hipEventRecord(e0, g->stream));

  ICHK(g, hipMemsetAsync(s.digit_tot, 0, 2 * sizeof(unsigned long long), g->stream));
  hipLaunchKernelGGL(ids_high_bits_kernel, dim3(blocks_for(n, 256, 8192)), dim3(256), 0, g->stream, g->d_user, n, s.digit_tot);
  hipLaunchKernelGGL(ids_high_bits_kernel, dim3(blocks_for(n, 256, 8192)), dim3(256), 0, g->stream, g->d_item, n, s.digit_tot + 1);
  unsigned long long high[2] = {1, 1};
  ICHK(g, hipMemcpyAsync(high, s.digit_tot, sizeof(high), hipMemcpyDeviceToHost, g->stream));
  ICHK(g, hipStreamSynchronize(g->stream));
  const bool narrow = !std::getenv("MALS_INGEST_WIDE_KEYS");
  const bool user32 = high[0] == 0 && narrow, item32 = high[1] == 0 && narrow;
  g->bytes_moved += 16.0 * (double)n;

  hipLaunchKernelGGL(big_sample_kernel, dim3(blocks_for(n_sample)), dim3(256), 0, g->stream, g->d_user, n, n_sample, s.keys[0], s.pay[0]);
  ICHK(g, hipGetLastError());
  int rs = 0;
  if (int rc = radix_sort<uint64_t, unsigned>(g, s, s.keys, s.pay, n_sample, &rs)) return rc;
  std::vector<int64_t> sample((size_t)n_sample);
  {
    std::vector<uint64_t> sk((size_t)n_sample);
    ICHK(g, hipMemcpy(sk.data(), s.keys[rs], sizeof(uint64_t) * (size_t)n_sample, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < sk.size(); ++i) sample[i] = key_to_id(sk[i]);
  }
  int n_parts = (int)std::min<int64_t>(250, std::max<int64_t>(2, (n + (part_cap * 3) / 4 - 1) / ((part_cap * 3) / 4)));
  std::vector<unsigned long long> hist(256);
  std::vector<int64_t> splitters;
  for (int attempt = 0;; ++attempt) {
    splitters = big_pick_splitters(sample, n_parts);
    ICHK(g, hipMemcpyAsync(B.d_split, splitters.data(), sizeof(int64_t) * splitters.size(), hipMemcpyHostToDevice, g->stream));
    hipLaunchKernelGGL(big_assign_part_kernel, dim3(blocks_for(n, 256, 1 << 16)), dim3(256), 0, g->stream, g->d_user, n, B.d_split, (int)splitters.size(), B.part);
    ICHK(g, hipMemsetAsync(s.digit_tot, 0, 256 * sizeof(unsigned long long), g->stream));
    hipLaunchKernelGGL(big_part_histogram_kernel, dim3(blocks_for(n, 256, 1 << 14)), dim3(256), 0, g->stream, B.part, n, s.digit_tot);
    ICHK(g, hipGetLastError());
    ICHK(g, hipMemcpyAsync(hist.data(), s.digit_tot, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, g->stream));
    ICHK(g, hipStreamSynchronize(g->stream));
    g->bytes_moved += 10.0 * (double)n;
    unsigned long long worst = 0;
    for (unsigned long long c : hist) worst = std::max(worst, c);
    if ((int64_t)worst <= part_cap) break;
    if (n_parts >= 250 || attempt >= 8)
      return fail(g, MALS_INVALID_ARG, "ingest: the records of one user-id range do not fit a partition (" + std::to_string(worst) + " records, capacity " +
                                           std::to_string(part_cap) + "): one user id owns too many lines, or MALS_INGEST_OPT_PARTITION_RECORDS is too small");
    n_parts = std::min(250, n_parts + std::max(1, n_parts / 2));
  }
  n_parts = (int)splitters.size() + 1;
  g->last_partitions = n_parts;

  // ---- 1. range by range ----------------------------------------------------------------------------------------------
  B.parts.resize((size_t)n_parts);
  int64_t u_base = 0, nnz_base = 0, known_base = 0;
  const unsigned n_tiles = big_tiles(n);
  float* sorted_val = reinterpret_cast<float*>(s.pay[0]);
  for (int p = 0; p < n_parts; ++p) {
    BigPart& bp = B.parts[(size_t)p];
    bp.u_base = u_base;
    bp.nnz_base = nnz_base;
    bp.known_base = known_base;
    const int64_t np = (int64_t)hist[(size_t)p];
    bp.n_records = np;
    if (np == 0) continue;
    hipLaunchKernelGGL(big_count_part_kernel, dim3(n_tiles), dim3(256), 0, g->stream, B.part, n, (uint8_t)p, t.head);
    ICHK(g, hipGetLastError());
    unsigned counted = 0;
    if (int rc = scan_u32(g, s, t.head, t.head, (int64_t)n_tiles, &counted)) return rc;
    if ((int64_t)counted != np) return fail(g, MALS_HIP_ERROR, "ingest: partition size mismatch");
    hipLaunchKernelGGL(big_compact_part_kernel, dim3(n_tiles), dim3(256), 0, g->stream, B.part, n, (uint8_t)p, t.head, g->d_user, g->d_item, g->d_value,
                       B.pu, B.pi, B.pv);
    ICHK(g, hipGetLastError());
    g->bytes_moved += 2.0 * (double)n + 48.0 * (double)np;
    const Records rec = {B.pu, B.pi, B.pv, np};
    unsigned n_u_all = 0, n_i_all = 0, n_users = 0, nnz = 0;
    FinishTmp tp;   // the range's temporaries (freed at the end of the iteration), sharing the arena views
    tp.head = t.head; tp.scan = t.scan; tp.ri = t.ri; tp.keep = t.keep; tp.pair_val = t.pair_val; tp.coo_row = t.coo_row; tp.present = t.present;
    int ra = 0, r = 0;
    if (int rc = item32 ? stage_items<uint32_t>(g, s, tp, rec, user32, &ra, &n_i_all) : stage_items<uint64_t>(g, s, tp, rec, user32, &ra, &n_i_all)) return rc;
    if (int rc = user32 ? stage_users<uint32_t>(g, s, tp, rec, ra, sorted_val, &r, &n_u_all) : stage_users<uint64_t>(g, s, tp, rec, ra, sorted_val, &r, &n_u_all))
      return rc;
    ICHK(g, hipMalloc(&tp.alive_u, sizeof(unsigned) * (size_t)n_u_all));
    ICHK(g, hipMalloc(&tp.alive_i, sizeof(unsigned) * (size_t)n_i_all));
    ICHK(g, hipMalloc(&tp.new_u, sizeof(unsigned) * (size_t)n_u_all));
    ICHK(g, hipMemsetAsync(tp.alive_u, 0, sizeof(unsigned) * (size_t)n_u_all, g->stream));
    ICHK(g, hipMemsetAsync(tp.alive_i, 0, sizeof(unsigned) * (size_t)n_i_all, g->stream));
    hipLaunchKernelGGL(replay_pairs_kernel, dim3(blocks_for(np)), dim3(256), 0, g->stream, s.keys[r], sorted_val, np, g->zero_threshold, tp.keep, tp.pair_val,
                       tp.alive_u, tp.alive_i, tp.present);
    ICHK(g, hipGetLastError());
    g->bytes_moved += (8.0 + 4.0 + 8.0 + (tp.present ? 4.0 : 0.0)) * (double)np;
    if (int rc = scan_u32(g, s, tp.alive_u, tp.new_u, n_u_all, &n_users)) return rc;
    if (int rc = scan_u32(g, s, tp.keep, tp.scan, np, &nnz)) return rc;
    bp.n_users = n_users;
    bp.nnz = nnz;
    bp.n_items_local = n_i_all;
    bp.item_ids = tp.iid_all;     // kept: the range's item table and its liveness
    tp.iid_all = nullptr;
    bp.item_alive = tp.alive_i;
    tp.alive_i = nullptr;
    ICHK(g, hipMalloc(&bp.user_ids, sizeof(int64_t) * std::max<size_t>(n_users, 1)));
    ICHK(g, hipMalloc(&bp.ptr_local, sizeof(int64_t) * ((size_t)n_users + 1)));
    hipLaunchKernelGGL(compact_ids_kernel, dim3(blocks_for(n_u_all)), dim3(256), 0, g->stream, tp.uid_all, tp.alive_u, tp.new_u, (int64_t)n_u_all, bp.user_ids);
    hipLaunchKernelGGL(big_compact_pairs_kernel, dim3(blocks_for(np)), dim3(256), 0, g->stream, s.keys[r], tp.keep, tp.scan, tp.pair_val, np, tp.new_u,
                       tp.coo_row, g->col[0] + nnz_base, g->val[0] + nnz_base);
    hipLaunchKernelGGL(row_ptr_from_sorted_kernel, dim3(blocks_for((int64_t)nnz + 1)), dim3(256), 0, g->stream, tp.coo_row, (int64_t)nnz, (int64_t)n_users,
                       bp.ptr_local);
    ICHK(g, hipGetLastError());
    g->bytes_moved += 16.0 * (double)np + 16.0 * (double)nnz + 8.0 * (double)n_users;
    if (g->want_known) {
      unsigned n_known = 0;
      if (int rc = scan_u32(g, s, tp.present, tp.scan, np, &n_known)) return rc;
      bp.n_known = n_known;
      ICHK(g, hipMalloc(&bp.known_ptr_local, sizeof(int64_t) * ((size_t)n_users + 1)));
      int32_t* known_row = (int32_t*)tp.ri;
      hipLaunchKernelGGL(big_compact_pairs_kernel, dim3(blocks_for(np)), dim3(256), 0, g->stream, s.keys[r], tp.present, tp.scan, (const float*)nullptr, np,
                         tp.new_u, known_row, g->known_idx + known_base, (float*)nullptr);
      hipLaunchKernelGGL(row_ptr_from_sorted_kernel, dim3(blocks_for((int64_t)n_known + 1)), dim3(256), 0, g->stream, known_row, (int64_t)n_known,
                         (int64_t)n_users, bp.known_ptr_local);
      ICHK(g, hipGetLastError());
      g->bytes_moved += (4.0 + 12.0 + 8.0) * (double)np + 12.0 * (double)n_known + 8.0 * (double)n_users;
      known_base += n_known;
    }
    // the merged item table grows by what this range saw for the first time
    if (int rc = big_merge_tables(g, s, t, B.item_glob, B.n_item_glob, bp.item_ids, bp.n_items_local)) return rc;
    ICHK(g, hipStreamSynchronize(g->stream));   // tp's allocations go away
    u_base += n_users;
    nnz_base += nnz;
    if (u_base > (int64_t)0x7fffff00 || B.n_item_glob > (int64_t)0x7fffff00)
      return fail(g, MALS_INVALID_ARG, "ingest: more than 2^31 distinct users or items (dense indices are 32-bit)");
  }
  const int64_t n_users = u_base, nnz = nnz_base, n_known = known_base;
  dfree(B.pu); dfree(B.pi); dfree(B.pv); dfree(B.part);

  // ---- 2. items: liveness through the rank maps, dense index, renumbering -------------------------------------------------
  const int64_t n_glob = B.n_item_glob;
  ICHK(g, hipMalloc(&B.alive_glob, sizeof(unsigned) * std::max<size_t>((size_t)n_glob, 1)));
  ICHK(g, hipMalloc(&B.new_i, sizeof(unsigned) * std::max<size_t>((size_t)n_glob, 1)));
  ICHK(g, hipMemsetAsync(B.alive_glob, 0, sizeof(unsigned) * std::max<size_t>((size_t)n_glob, 1), g->stream));
  // (the maps live in the keys arena: a range's item table is no longer than the range)
  int64_t* map = reinterpret_cast<int64_t*>(s.keys[0]);
  int32_t* final_map = reinterpret_cast<int32_t*>(s.keys[1]);
  for (BigPart& bp : B.parts) {
    if (bp.n_items_local == 0) continue;
    hipLaunchKernelGGL(index_of_ids_kernel, dim3(blocks_for(bp.n_items_local)), dim3(256), 0, g->stream, bp.item_ids, bp.n_items_local, B.item_glob, n_glob, map);
    hipLaunchKernelGGL(big_mark_alive_kernel, dim3(blocks_for(bp.n_items_local)), dim3(256), 0, g->stream, bp.item_alive, map, bp.n_items_local, B.alive_glob);
    ICHK(g, hipGetLastError());
  }
  unsigned n_items_u = 0;
  if (int rc = scan_u32(g, s, B.alive_glob, B.new_i, n_glob, &n_items_u)) return rc;
  const int64_t n_items = n_items_u;
  g->n_users = n_users;
  g->n_items = n_items;
  g->nnz = nnz;
  g->n_known = n_known;
  ICHK(g, hipMalloc(&g->ids[0], sizeof(int64_t) * std::max<size_t>((size_t)n_users, 1)));
  ICHK(g, hipMalloc(&g->ids[1], sizeof(int64_t) * std::max<size_t>((size_t)n_items, 1)));
  ICHK(g, hipMalloc(&g->ptr[0], sizeof(int64_t) * ((size_t)n_users + 1)));
  ICHK(g, hipMalloc(&g->ptr[1], sizeof(int64_t) * ((size_t)n_items + 1)));
  if (g->want_known) ICHK(g, hipMalloc(&g->known_ptr, sizeof(int64_t) * ((size_t)n_users + 1)));
  hipLaunchKernelGGL(compact_ids_kernel, dim3(blocks_for(n_glob)), dim3(256), 0, g->stream, B.item_glob, B.alive_glob, B.new_i, n_glob, g->ids[1]);
  ICHK(g, hipGetLastError());
  ICHK(g, hipMemsetAsync(g->ptr[0], 0, sizeof(int64_t), g->stream));   // (no range, no user: the one offset there is)
  if (g->want_known) ICHK(g, hipMemsetAsync(g->known_ptr, 0, sizeof(int64_t), g->stream));
  for (BigPart& bp : B.parts) {
    if (bp.n_records == 0) continue;
    hipLaunchKernelGGL(index_of_ids_kernel, dim3(blocks_for(bp.n_items_local)), dim3(256), 0, g->stream, bp.item_ids, bp.n_items_local, B.item_glob, n_glob, map);
    hipLaunchKernelGGL(big_final_map_kernel, dim3(blocks_for(bp.n_items_local)), dim3(256), 0, g->stream, map, bp.n_items_local, B.new_i, final_map);
    if (bp.nnz) hipLaunchKernelGGL(big_remap_kernel, dim3(blocks_for(bp.nnz)), dim3(256), 0, g->stream, g->col[0] + bp.nnz_base, bp.nnz, final_map);
    if (g->want_known && bp.n_known)
      hipLaunchKernelGGL(big_remap_kernel, dim3(blocks_for(bp.n_known)), dim3(256), 0, g->stream, g->known_idx + bp.known_base, bp.n_known, final_map);
    if (bp.n_users) ICHK(g, hipMemcpyAsync(g->ids[0] + bp.u_base, bp.user_ids, sizeof(int64_t) * (size_t)bp.n_users, hipMemcpyDeviceToDevice, g->stream));
    hipLaunchKernelGGL(big_add_base_kernel, dim3(blocks_for(bp.n_users + 1)), dim3(256), 0, g->stream, bp.ptr_local, bp.n_users + 1, bp.nnz_base,
                       g->ptr[0] + bp.u_base);
    if (g->want_known)
      hipLaunchKernelGGL(big_add_base_kernel, dim3(blocks_for(bp.n_users + 1)), dim3(256), 0, g->stream, bp.known_ptr_local, bp.n_users + 1, bp.known_base,
                         g->known_ptr + bp.u_base);
    ICHK(g, hipGetLastError());
    g->bytes_moved += 8.0 * (double)bp.nnz + 24.0 * (double)bp.n_users;
  }
  ICHK(g, hipStreamSynchronize(g->stream));
  for (BigPart& bp : B.parts) {
    dfree(bp.item_ids); dfree(bp.item_alive); dfree(bp.user_ids); dfree(bp.ptr_local); dfree(bp.known_ptr_local);
  }
  dfree(B.alive_glob); dfree(B.new_i); dfree(B.item_glob);

  // ---- 3. R^T ---------------------------------------------------------------------------------------------------------------
  // item ranges from a sample of the entries (every 61st), cut at 0.7 of a partition; a range that turns out larger is halved
  ICHK(g, hipMalloc(&B.cnt, sizeof(unsigned) * ((size_t)n_items + 8)));
  const int64_t tiles64 = (n_items + SC_TILE - 1) / SC_TILE + 1;
  ICHK(g, hipMalloc(&B.sums64, sizeof(unsigned long long) * ((size_t)tiles64 + 1)));
  ICHK(g, hipMemsetAsync(B.cnt, 0, sizeof(unsigned) * ((size_t)n_items + 8), g->stream));
  const int64_t stride = nnz > ((int64_t)1 << 24) ? 61 : 1;
  if (nnz) hipLaunchKernelGGL(big_item_sample_kernel, dim3(blocks_for((nnz + stride - 1) / stride, 256, 1 << 16)), dim3(256), 0, g->stream, g->col[0], nnz, stride, B.cnt);
  const unsigned t64 = (unsigned)std::max<int64_t>(1, (n_items + SC_TILE - 1) / SC_TILE);
  hipLaunchKernelGGL(big_scan64_reduce_kernel, dim3(t64), dim3(256), 0, g->stream, B.cnt, n_items, B.sums64);
  hipLaunchKernelGGL(big_scan64_sums_kernel, dim3(1), dim3(64), 0, g->stream, B.sums64, (int64_t)t64, B.sums64 + t64);
  hipLaunchKernelGGL(big_scan64_apply_kernel, dim3(t64), dim3(256), 0, g->stream, B.cnt, n_items, B.sums64, B.sums64 + t64, g->ptr[1]);
  ICHK(g, hipGetLastError());
  g->bytes_moved += 4.0 * (double)nnz / (double)stride + 16.0 * (double)n_items;
  std::vector<int64_t> cp((size_t)n_items + 1);   // estimated offsets, in sampled entries
  ICHK(g, hipMemcpyAsync(cp.data(), g->ptr[1], sizeof(int64_t) * cp.size(), hipMemcpyDeviceToHost, g->stream));
  ICHK(g, hipStreamSynchronize(g->stream));
  std::vector<std::pair<int64_t, int64_t>> todo;   // item ranges still to do, the first on top
  {
    const int64_t budget = std::max<int64_t>(1, (int64_t)(0.7 * (double)part_cap / (double)stride));
    std::vector<std::pair<int64_t, int64_t>> ranges;
    for (int64_t a = 0; a < n_items;) {
      int64_t b = std::upper_bound(cp.begin() + a + 1, cp.end(), cp[(size_t)a] + budget) - cp.begin() - 1;
      b = std::min(std::max(b, a + 1), n_items);
      ranges.emplace_back(a, b);
      a = b;
    }
    todo.assign(ranges.rbegin(), ranges.rend());
  }
  const unsigned e_tiles = big_tiles(nnz);
  ICHK(g, hipMalloc(&B.tile_row, sizeof(int32_t) * ((size_t)e_tiles + 1)));
  hipLaunchKernelGGL(big_tile_rows_kernel, dim3(blocks_for((int64_t)e_tiles + 1)), dim3(256), 0, g->stream, g->ptr[0], std::max<int64_t>(n_users, 1), nnz,
                     (int64_t)e_tiles, B.tile_row);
  ICHK(g, hipGetLastError());
  int64_t done_entries = 0;   // R^T's offsets so far: every range starts where the one before it ended
  ICHK(g, hipMemsetAsync(g->ptr[1], 0, sizeof(int64_t) * ((size_t)n_items + 1), g->stream));
  while (!todo.empty()) {
    const int64_t a = todo.back().first, b = todo.back().second;
    todo.pop_back();
    int64_t nq = 0;
    if (nnz > 0) {
      hipLaunchKernelGGL(big_count_items_kernel, dim3(e_tiles), dim3(256), 0, g->stream, g->col[0], nnz, (int32_t)a, (int32_t)b, t.head);
      ICHK(g, hipGetLastError());
      unsigned counted = 0;
      if (int rc = scan_u32(g, s, t.head, t.head, (int64_t)e_tiles, &counted)) return rc;
      nq = counted;
      g->bytes_moved += 4.0 * (double)nnz;
    }
    if (nq > part_cap) {   // the sample underestimated it: two halves (a single item that does not fit is refused)
      if (b - a < 2) return fail(g, MALS_INVALID_ARG, "ingest: the entries of one item do not fit a partition (" + std::to_string(nq) + ")");
      const int64_t mid = a + (b - a) / 2;
      todo.emplace_back(mid, b);
      todo.emplace_back(a, mid);
      continue;
    }
    ++g->last_item_ranges;
    if (nq > 0) {
      hipLaunchKernelGGL(big_select_items_kernel, dim3(e_tiles), dim3(256), 0, g->stream, g->col[0], g->val[0], nnz, g->ptr[0], B.tile_row, (int32_t)a, (int32_t)b,
                         t.head, s.keys[0], s.pay[0]);
      ICHK(g, hipGetLastError());
      int r2 = 0;
      if (int rc = radix_sort<uint64_t, unsigned>(g, s, s.keys, s.pay, nq, &r2, 4, ((uint64_t)(b - a - 1) << 32) | 0xffffffffull)) return rc;
      hipLaunchKernelGGL(big_transpose_write_kernel, dim3(blocks_for(nq)), dim3(256), 0, g->stream, s.keys[r2], s.pay[r2], nq, t.coo_row, g->col[1] + done_entries,
                         g->val[1] + done_entries);
      ICHK(g, hipGetLastError());
      g->bytes_moved += 4.0 * (double)nnz + 40.0 * (double)nq;
    }
    // offsets of the range's items: local from the sorted item column, then shifted behind the ranges before it
    hipLaunchKernelGGL(row_ptr_from_sorted_kernel, dim3(blocks_for(nq + 1)), dim3(256), 0, g->stream, t.coo_row, nq, b - a, g->ptr[1] + a);
    hipLaunchKernelGGL(big_add_base_inplace_kernel, dim3(blocks_for(b - a + 1)), dim3(256), 0, g->stream, g->ptr[1] + a, b - a + 1, done_entries);
    ICHK(g, hipGetLastError());
    done_entries += nq;
  }
  if (done_entries != nnz) return fail(g, MALS_HIP_ERROR, "ingest: the item ranges do not add up to the entries");
  // ---- 4. tag id sets, userTagIDs as rows of R^T ----------------------------------------------------------------------------
  if (int rc = finish_tags(g, s, t, std::max<int64_t>(part_cap, n_sample))) return rc;
  ICHK(g, hipStreamSynchronize(g->stream));
  return MALS_OK;
}
