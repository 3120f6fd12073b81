// ingest_kernels.h -- hand-written gfx950 kernels of the ingest -> CSR path (SURVEY.md section 8(f)
// row 2): what InputFilesReader.readInputFiles (online-local/.../generation/InputFilesReader.java:
// 64-211) + MatrixUtils.addTo / remove (common/.../math/MatrixUtils.java:64-125) +
// FastByIDFloatMap.increment (common/.../collection/FastByIDFloatMap.java:129-138) do to a stream of
// (user, item, value | NaN) records, restated as sorts, scans and one sequential pass per pair.
//
// All of it is HBM-bound integer work: an LSD radix sort (8-bit digits, stable) of 64-bit keys with a
// 32- or 64-bit payload, prefix sums, and elementwise passes.  Nothing here is shaped into a GEMM.
//   radix pass = histogram kernel (one read of the keys) + prefix sum of the per-workgroup digit
//   counts + scatter kernel (one read of keys+payload, one write).  A workgroup owns 4096 consecutive
//   keys; every wave walks its quarter 64 keys at a time in order and ranks a key among the equal
//   digits of its round with 8 ballots ("multi-split"), running counts in a wave-private LDS row, so
//   the sort is stable without atomics; the tile is then ordered by digit in LDS and written out in
//   runs instead of isolated 8-byte stores.
//   Digits on which all keys agree are skipped (one preliminary histogram of all 8 digits), which
//   is most of them for ids below 2^32 and for (row,col) keys of realistic shapes.
// Pipeline (ingest_api.hip): composite sort by (user id, item id) in two stable stages with the dense
// ranks read off the sorted orders, per-pair replay, compaction, transpose by a sort on the item half.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mals {

constexpr int RS_WAVE_TILE = 1024;  // keys per wave and pass (16 rounds of 64)
constexpr int RS_BLOCK_TILE = 4 * RS_WAVE_TILE;

// exclusive scan of one value per thread over a 256-thread workgroup
__device__ __forceinline__ unsigned block_exclusive_scan_256(unsigned v) {
  __shared__ unsigned wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned y = __shfl_up(x, off);
    if (lane >= off) x += y;
  }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  unsigned pre = 0;
  for (int i = 0; i < w; ++i) pre += wsum[i];
  return pre + x - v;
}

// ---- digit statistics of all 8 digits in one read: tot[8][256] (global atomics on block sums) -------
template <typename K>  // key: uint64_t, or uint32_t when every key of the sort fits (ids below 2^32: 12 bytes less per record and pass)
__global__ __launch_bounds__(256) void rs_digit_totals_kernel(const K* __restrict__ keys, int64_t n,
                                                              unsigned long long* __restrict__ tot) {
  constexpr int ND = (int)sizeof(K);
  __shared__ unsigned h[8 * 256];
  for (int i = threadIdx.x; i < 8 * 256; i += 256) h[i] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int lane = threadIdx.x & 63;
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += stride) {  // whole waves stay in the loop together
    const int64_t i = i0 + threadIdx.x;
    const bool ok = i < n;
    const K k = ok ? keys[i] : (K)0;
    const uint64_t active = __ballot(ok);
    if (!active) continue;
    const int leader = __ffsll((unsigned long long)active) - 1;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const int digit = (int)((k >> (8 * d)) & 255);
      // the high digits of real ids agree across a wave almost always: one LDS atomic instead of 64
      const int first = __builtin_amdgcn_readlane(digit, leader);
      if (__ballot(ok && digit != first) == 0) {
        if (lane == leader) atomicAdd(&h[d * 256 + first], (unsigned)__popcll(active));
      } else if (ok) {
        atomicAdd(&h[d * 256 + digit], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 256; i += 256)
    if (h[i]) atomicAdd(&tot[i], (unsigned long long)h[i]);
}

// ---- per-block digit histogram: counts[digit * n_blocks + block] ---------------------------------
template <typename K>
__global__ __launch_bounds__(256) void rs_histogram_kernel(const K* __restrict__ keys, int64_t n, int shift,
                                                           int64_t n_blocks, unsigned* __restrict__ counts) {
  __shared__ unsigned h[4][256];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 256; i += 64) h[w][i] = 0;
  // wave-private LDS rows while counting: no barrier needed, a wave executes in lock step
  const int64_t b = (int64_t)blockIdx.x * RS_BLOCK_TILE + w * RS_WAVE_TILE;
  // (all sixteen loads of the wave's quarter are issued before the first count: one memory latency per tile, not sixteen)
  K kk[RS_WAVE_TILE / 64];
#pragma unroll
  for (int r = 0; r < RS_WAVE_TILE / 64; ++r) {
    const int64_t i = b + 64 * r + lane;
    kk[r] = i < n ? keys[i] : (K)0;
  }
#pragma unroll
  for (int r = 0; r < RS_WAVE_TILE / 64; ++r)
    if (b + 64 * r + lane < n) atomicAdd(&h[w][(int)((kk[r] >> shift) & 255)], 1u);
  __syncthreads();
  const int d = threadIdx.x;
  counts[(int64_t)d * n_blocks + blockIdx.x] = h[0][d] + h[1][d] + h[2][d] + h[3][d];
}

// ---- stable scatter ------------------------------------------------------------------------------
// A workgroup owns RS_BLOCK_TILE consecutive keys (wave w the w-th quarter, walked 64 at a time in
// order).  Pass 1 ranks every key among the equal digits of its wave (8 ballots per round + a
// wave-private running count); the four waves' counts give each digit a contiguous segment of the
// tile; pass 2 moves the tile into LDS in digit order; pass 3 streams it out: consecutive LDS slots of
// a digit go to consecutive global addresses, so the writes are runs of ~RS_BLOCK_TILE/256 keys
// instead of isolated 8-byte stores that HBM would turn into read-modify-writes of whole lines.
template <typename K, typename P>  // key, payload: uint32_t or uint64_t
__global__ __launch_bounds__(256) void rs_scatter_kernel(const K* __restrict__ keys, const P* __restrict__ pay,
                                                         int64_t n, int shift, int64_t n_blocks,
                                                         const unsigned* __restrict__ offsets,  // scanned counts
                                                         K* __restrict__ keys_out, P* __restrict__ pay_out) {
  constexpr int ROUNDS = RS_WAVE_TILE / 64;
  __shared__ K skey[RS_BLOCK_TILE];
  __shared__ P spay[RS_BLOCK_TILE];
  __shared__ unsigned cnt[4][256];   // per-wave digit counts, then per-wave offsets inside the digit's segment
  __shared__ unsigned seg[256];      // start of the digit's segment in the tile
  __shared__ unsigned gbase[256];    // global position of the segment's first key
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 256; i += 64) cnt[w][i] = 0;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int64_t b = (int64_t)blockIdx.x * RS_BLOCK_TILE + w * RS_WAVE_TILE;
  K k[ROUNDS];
  P p[ROUNDS];
  unsigned short lrank[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int64_t i = b + 64 * r + lane;
    const bool ok = i < n;
    k[r] = ok ? keys[i] : (K)~(K)0;
    p[r] = ok ? pay[i] : (P)0;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const bool ok = b + 64 * r + lane < n;
    const int digit = (int)((k[r] >> shift) & 255);
    uint64_t peers = __ballot(ok);  // lanes of this round with the same digit (inactive lanes match nobody)
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (digit >> bit) & 1;
      const uint64_t m = __ballot(one);
      peers &= one ? m : ~m;
    }
    if (ok) {
      const unsigned pos = cnt[w][digit] + (unsigned)__popcll(peers & lt);
      lrank[r] = (unsigned short)pos;
      // the highest peer advances the digit's running count; every peer has read it already (LDS
      // operations of one wave complete in program order)
      if ((peers >> lane) == 1ull) cnt[w][digit] = pos + 1;
    }
  }
  __syncthreads();
  {  // segment starts: exclusive scan over the 256 digit totals (one digit per thread)
    const int d = threadIdx.x;
    const unsigned c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
    const unsigned start = block_exclusive_scan_256(c0 + c1 + c2 + c3);
    seg[d] = start;
    gbase[d] = offsets[(int64_t)d * n_blocks + blockIdx.x];
    cnt[0][d] = start;  // becomes: position of wave w's first key of digit d in the tile
    cnt[1][d] = start + c0;
    cnt[2][d] = start + c0 + c1;
    cnt[3][d] = start + c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    if (b + 64 * r + lane < n) {
      const int digit = (int)((k[r] >> shift) & 255);
      const unsigned q = cnt[w][digit] + lrank[r];
      skey[q] = k[r];
      spay[q] = p[r];
    }
  }
  __syncthreads();
  const int64_t tile_n = n - (int64_t)blockIdx.x * RS_BLOCK_TILE < RS_BLOCK_TILE ? n - (int64_t)blockIdx.x * RS_BLOCK_TILE : RS_BLOCK_TILE;
  for (int q = threadIdx.x; q < tile_n; q += 256) {
    const K kk = skey[q];
    const int digit = (int)((kk >> shift) & 255);
    const unsigned pos = gbase[digit] + ((unsigned)q - seg[digit]);
    keys_out[pos] = kk;
    pay_out[pos] = spay[q];
  }
}

// ---- exclusive scan of uint32 (reduce, scan of the tile sums, apply; block = 256 threads x 8 items):
// the data is read twice and written once -----------------------------------------------------------
constexpr int SC_TILE = 2048;
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* total) {  // v: per-thread value
  __shared__ unsigned ws[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned y = __shfl_up(x, off);
    if (lane >= off) x += y;
  }
  if (lane == 63) ws[w] = x;
  __syncthreads();
  unsigned pre = 0;
  for (int i = 0; i < w; ++i) pre += ws[i];
  if (total) *total = ws[0] + ws[1] + ws[2] + ws[3];
  __syncthreads();
  return pre + x - v;
}
// a thread's 8 consecutive items: two 16-byte loads where the whole group is inside the array
__device__ __forceinline__ void scan_load8(const unsigned* __restrict__ in, int64_t b, int64_t n, unsigned (&v)[8]) {
  if (b + 8 <= n) {
    const uint4 lo = *reinterpret_cast<const uint4*>(in + b), hi = *reinterpret_cast<const uint4*>(in + b + 4);
    v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w, v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = b + i < n ? in[b + i] : 0;
  }
}
// pass 1: the sum of every tile (the input is only read)
__global__ __launch_bounds__(256) void scan_reduce_kernel(const unsigned* __restrict__ in, int64_t n, unsigned* __restrict__ tile_sums) {
  const int64_t b = (int64_t)blockIdx.x * SC_TILE + threadIdx.x * 8;
  unsigned v[8], s = 0;
  scan_load8(in, b, n, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  unsigned total;
  block_exclusive_scan(s, &total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
// pass 3: exclusive scan inside each tile on top of its (already scanned) tile offset; in may alias out
__global__ __launch_bounds__(256) void scan_apply_kernel(const unsigned* in, int64_t n, unsigned* out, const unsigned* __restrict__ tile_offsets) {
  const int64_t b = (int64_t)blockIdx.x * SC_TILE + threadIdx.x * 8;
  unsigned v[8], s = 0;
  scan_load8(in, b, n, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  unsigned pre = tile_offsets[blockIdx.x] + block_exclusive_scan(s, nullptr);
  unsigned o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    o[i] = pre;
    pre += v[i];
  }
  if (b + 8 <= n) {
    *reinterpret_cast<uint4*>(out + b) = make_uint4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint4*>(out + b + 4) = make_uint4(o[4], o[5], o[6], o[7]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (b + i < n) out[b + i] = o[i];
  }
}
// one block scans the tile sums in place (exclusive), carrying across chunks of 2048
__global__ __launch_bounds__(256) void scan_sums_kernel(unsigned* __restrict__ sums, int64_t n_tiles, unsigned* __restrict__ grand_total) {
  unsigned carry = 0;
  for (int64_t b0 = 0; b0 < n_tiles; b0 += SC_TILE) {
    const int64_t b = b0 + threadIdx.x * 8;
    unsigned v[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = b + i < n_tiles ? sums[b + i] : 0;
      s += v[i];
    }
    unsigned total;
    unsigned pre = carry + block_exclusive_scan(s, &total);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (b + i < n_tiles) sums[b + i] = pre;
      pre += v[i];
    }
    carry += total;
  }
  if (threadIdx.x == 0 && grand_total) *grand_total = carry;
}
// ---- elementwise passes ------------------------------------------------------------------------------
#define MALS_GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// signed 64-bit id -> radix key with the same order
__device__ __host__ __forceinline__ uint64_t id_to_key(int64_t id) { return (uint64_t)id ^ 0x8000000000000000ull; }
__device__ __host__ __forceinline__ int64_t key_to_id(uint64_t k) { return (int64_t)(k ^ 0x8000000000000000ull); }
// a sort whose ids all lie in [0, 2^32) runs on 32-bit keys: the id itself
template <typename K>
__device__ __forceinline__ K make_key(int64_t id) {
  if constexpr (sizeof(K) == 4) return (K)(uint32_t)id;
  else return (K)id_to_key(id);
}
__device__ __forceinline__ int64_t key_to_id(uint32_t k) { return (int64_t)k; }

// Do all ids fit 32 unsigned bits?  (One read of the ids; the answer decides what rides through the first sort.)
__global__ __launch_bounds__(256) void ids_high_bits_kernel(const int64_t* __restrict__ ids, int64_t n, unsigned long long* __restrict__ high_or) {
  unsigned long long acc = 0;
  MALS_GRID_STRIDE(i, n) acc |= (unsigned long long)ids[i] >> 32;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc |= __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0 && acc) atomicOr(high_or, acc);
}
// first stage of the composite sort: key = item id, payload = (value bits << 32 | x): the value rides along through
// both stages.  x = the record's USER ID when every user id fits 32 bits (USER32; the usual case: the second stage then
// needs no gather at all -- 27.7 ms of a 215 ms finish at 1e9 records was that random 8-byte gather), else the record
// index, through which the second stage fetches the user id.
template <typename K, bool USER32>
__global__ void item_stage_kernel(const int64_t* __restrict__ item_ids, const int64_t* __restrict__ user_ids, const float* __restrict__ values,
                                  int64_t n, K* __restrict__ keys, uint64_t* __restrict__ pay) {
  MALS_GRID_STRIDE(i, n) {
    keys[i] = make_key<K>(item_ids[i]);
    pay[i] = ((uint64_t)__float_as_uint(values[i]) << 32) | (USER32 ? (uint64_t)(uint32_t)user_ids[i] : (uint64_t)i);
  }
}
// head[i] = 1 where a new key starts in a sorted key array
template <typename K>
__global__ void heads_kernel(const K* __restrict__ keys, int64_t n, unsigned* __restrict__ head) {
  MALS_GRID_STRIDE(i, n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// item-sorted order: dense item rank of every position + the ascending item id table
template <typename K>
__global__ void position_ranks_kernel(const K* __restrict__ keys, const unsigned* __restrict__ head,
                                      const unsigned* __restrict__ head_scan, int64_t n, unsigned* __restrict__ rank_of_position,
                                      int64_t* __restrict__ id_table) {
  MALS_GRID_STRIDE(i, n) {
    const unsigned r = head_scan[i] + head[i] - 1;
    rank_of_position[i] = r;
    if (head[i]) id_table[r] = key_to_id(keys[i]);
  }
}
// second stage of the composite sort: key = the record's user id (USER32: carried in the payload; else gathered through
// the item-sorted permutation), payload = (item rank << 32 | value bits)
template <typename K, bool USER32>   // USER32 <=> K = uint32_t
__global__ void user_stage_kernel(const int64_t* __restrict__ user_ids, const uint64_t* __restrict__ pay_a,
                                  const unsigned* __restrict__ item_rank, int64_t n, K* __restrict__ keys,
                                  uint64_t* __restrict__ pay) {
  MALS_GRID_STRIDE(i, n) {
    const uint64_t pa = pay_a[i];
    keys[i] = make_key<K>(USER32 ? (int64_t)(pa & 0xffffffffu) : user_ids[(unsigned)(pa & 0xffffffffu)]);
    pay[i] = ((uint64_t)item_rank[i] << 32) | (pa >> 32);
  }
}
// (user id, item rank, stream order) sorted records -> the pair keys and values replay_pairs wants
template <typename K>
__global__ void pair_from_sorted_kernel(const K* __restrict__ ukeys, const uint64_t* __restrict__ pay,
                                        const unsigned* __restrict__ head, const unsigned* __restrict__ head_scan, int64_t n,
                                        uint64_t* __restrict__ pair_keys, float* __restrict__ sorted_val,
                                        int64_t* __restrict__ user_table) {
  MALS_GRID_STRIDE(i, n) {
    const unsigned ru = head_scan[i] + head[i] - 1;
    pair_keys[i] = ((uint64_t)ru << 32) | (pay[i] >> 32);
    sorted_val[i] = __uint_as_float((unsigned)(pay[i] & 0xffffffffu));
    if (head[i]) user_table[ru] = key_to_id(ukeys[i]);
  }
}
// One thread per (user,item) pair = per run of equal keys in the stably sorted record array: replays
// the pair's records in stream order exactly as the reference does (IFR:165-171): NaN removes the
// entry (MU:102-125), a value starts it or is added to it in fp32 (FBIFM:129-138).  Marks the ids
// that own a live entry (MU:81-92: a row exists while it has entries).
__global__ void replay_pairs_kernel(const uint64_t* __restrict__ keys, const float* __restrict__ values,  // both in sorted order
                                    int64_t n, float zero_threshold,
                                    unsigned* __restrict__ keep, float* __restrict__ pair_val,
                                    unsigned* __restrict__ user_alive, unsigned* __restrict__ item_alive,
                                    unsigned* __restrict__ present_out) {  // present_out (may be null): knownItemIDs, IFR:173-191
  MALS_GRID_STRIDE(i, n) {
    keep[i] = 0;
    if (present_out) present_out[i] = 0;
    const uint64_t k = keys[i];
    if (i != 0 && keys[i - 1] == k) continue;  // not the head of its run
    bool present = false;
    float v = 0.f;
    for (int64_t t = i; t < n && keys[t] == k; ++t) {
      const float x = values[t];
      if (x != x) {
        present = false;
      } else if (!present) {
        present = true;
        v = x;
      } else {
        v = v + x;
      }
    }
    if (present) {
      // (test first: a popular item's flag is set by its first pair and then only read -- a billion 4-byte stores into the few
      // megabytes of the item table are write traffic the L2 has to serialise per line; the flags are idempotent, a race only
      // repeats a store)
      unsigned* ua = user_alive + (unsigned)(k >> 32);
      unsigned* ia = item_alive + (unsigned)(k & 0xffffffffu);
      if (__atomic_load_n(ua, __ATOMIC_RELAXED) == 0u) *ua = 1u;
      if (__atomic_load_n(ia, __ATOMIC_RELAXED) == 0u) *ia = 1u;
      pair_val[i] = v;
      if (present_out) present_out[i] = 1u;
      keep[i] = (fabsf(v) < zero_threshold) ? 0u : 1u;  // removeSmall, IFR:200-211 (NaN sums are kept, like there)
    }
  }
}
// compact the existing ids: new dense index = exclusive scan of alive
__global__ void compact_ids_kernel(const int64_t* __restrict__ id_table, const unsigned* __restrict__ alive,
                                   const unsigned* __restrict__ alive_scan, int64_t n_ids, int64_t* __restrict__ ids_out) {
  MALS_GRID_STRIDE(i, n_ids)
    if (alive[i]) ids_out[alive_scan[i]] = id_table[i];
}
// out[i] = the position of ids[i] in the ascending table, -1 if it is not there
__global__ void index_of_ids_kernel(const int64_t* __restrict__ ids, int64_t n, const int64_t* __restrict__ table, int64_t n_table,
                                    int64_t* __restrict__ out) {
  MALS_GRID_STRIDE(i, n) {
    const int64_t want = ids[i];
    int64_t lo = 0, hi = n_table;
    while (lo < hi) {
      const int64_t mid = lo + (hi - lo) / 2;
      if (table[mid] < want) lo = mid + 1; else hi = mid;
    }
    out[i] = (lo < n_table && table[lo] == want) ? lo : -1;
  }
}
// compact the kept pairs into COO sorted by (row, col); count entries per row
__global__ void compact_pairs_kernel(const uint64_t* __restrict__ keys, const unsigned* __restrict__ keep,
                                     const unsigned* __restrict__ keep_scan, const float* __restrict__ pair_val, int64_t n,
                                     const unsigned* __restrict__ new_u, const unsigned* __restrict__ new_i,
                                     int32_t* __restrict__ row, int32_t* __restrict__ col, float* __restrict__ val) {
  MALS_GRID_STRIDE(i, n) {
    if (!keep[i]) continue;
    const unsigned p = keep_scan[i];
    row[p] = (int32_t)new_u[(unsigned)(keys[i] >> 32)];
    col[p] = (int32_t)new_i[(unsigned)(keys[i] & 0xffffffffu)];
    val[p] = pair_val[i];
  }
}
// Row pointers of a COO array sorted by row (rows without entries included): entry p opens every row
// in (row[p-1], row[p]]; the rows after the last entry point at nnz.  No atomics: a popular item
// would serialise millions of them on one counter.
__global__ void row_ptr_from_sorted_kernel(const int32_t* __restrict__ row, int64_t nnz, int64_t n_rows, int64_t* __restrict__ ptr) {
  MALS_GRID_STRIDE(p, nnz + 1) {
    const int64_t lo = p == 0 ? 0 : (int64_t)row[p - 1] + 1;
    const int64_t hi = p == nnz ? n_rows : (int64_t)row[p];
    for (int64_t r = lo; r <= hi; ++r) ptr[r] = p;
  }
}
__global__ void transpose_keys_kernel(const int32_t* __restrict__ row, const int32_t* __restrict__ col,
                                      const float* __restrict__ val, int64_t nnz, uint64_t* __restrict__ keys,
                                      unsigned* __restrict__ pay) {
  MALS_GRID_STRIDE(i, nnz) {
    keys[i] = ((uint64_t)(uint32_t)col[i] << 32) | (uint32_t)row[i];
    pay[i] = __float_as_uint(val[i]);  // the value rides along: no gather after the sort
  }
}
__global__ void transpose_gather_kernel(const uint64_t* __restrict__ keys, const unsigned* __restrict__ pay,
                                        int64_t nnz, int32_t* __restrict__ t_row, int32_t* __restrict__ t_col,
                                        float* __restrict__ t_val) {
  MALS_GRID_STRIDE(i, nnz) {
    t_row[i] = (int32_t)(keys[i] >> 32);          // the item index
    t_col[i] = (int32_t)(keys[i] & 0xffffffffu);  // the user index
    t_val[i] = __uint_as_float(pay[i]);
  }
}

}  // namespace mals
