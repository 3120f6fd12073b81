// topn_kernels.h -- hand-written gfx950 kernels of top-N scoring: what
// ServerRecommender.multithreadedTopN (online/src/net/myrrix/online/ServerRecommender.java:443-508)
// does with RecommendIterator (RecommendIterator.java:62-109) and TopN (common/.../TopN.java:49-128):
// score every item against the query vector, skip the user's known items, keep the N best.
//
//   scores   fp64 matrix cores, 16 items x 16 queries x 4 features per instruction; all queries of
//            the batch are scored per item read, so Y is streamed once per batch: HBM-bound
//            (n_items * 4k bytes).  fp64 accumulation like the reference's dot
//            (SimpleVectorMath.java:34-41), cast to fp32 at the end.
//   mask     known items of each query's user -> -inf (RecommendIterator.java:75-82).
//   select   4-pass radix select (8-bit digits of an order-preserving integer image of the score,
//            grid-wide histograms per query) finds the N-th largest score exactly, then everything
//            above it plus the ties are handed back; no sort of the whole row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mals {

constexpr int TOPN_MAX_QUERIES = 64;  // queries scored per item read (x vectors staged in LDS)

// fp32 -> uint32 with the same order (NaN sorts above +inf; the scores here are finite or -inf)
__device__ __host__ __forceinline__ uint32_t score_key(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Scores on the fp64 matrix cores: D[item][query] += A[item][feature] * B[feature][query] with
// v_mfma_f64_16x16x4_f64 -- 16 items x 16 queries x 4 features per instruction, operands widened
// from fp32 (their product is exact in fp64), fp64 accumulation, one cast to fp32 at the end.  The
// reference rounds each product to fp32 before it widens (SimpleVectorMath.java:37); the difference
// is below half an fp32 ulp of the score, so a score can differ from the reference's in its last bit
// (tests compare at 2 ulp).  A wave owns 16 consecutive items and all query tiles; the queries sit
// in LDS as doubles in operand order.  Y is read once per batch: HBM-bound (n_items * 4k bytes).
typedef double f64x4_t __attribute__((ext_vector_type(4)));
struct TopnState {
  uint32_t prefix;     // digits decided so far (in place, high bits); after 4 passes the N-th best key
  uint32_t remaining;  // rank of the N-th score inside the candidates that match the prefix
  uint32_t above;      // collected: scores strictly above the N-th (filter path: all candidates)
  uint32_t ties;       // collected: scores equal to the N-th (all of them counted, cap_ties stored)
};
__device__ __forceinline__ uint32_t topn_threshold(const TopnState* st, int q) { return st[q].prefix; }
__device__ __forceinline__ unsigned topn_append(TopnState* st, int q) { return atomicAdd(&st[q].above, 1u); }
__device__ __forceinline__ unsigned topn_count(const TopnState* st, int q) { return st[q].above; }

// MODE 0: dense score rows for the 16-item tiles whose index is a multiple of tile_stride
//         (1 = every item; 16 = the 1/16 sample that yields the filter thresholds), row length n_out;
// MODE 1: filter -- only (item, score) pairs whose score reaches the query's threshold key are
//         appended to the query's candidate list (st[q].above counts them, also past the capacity).
// NT = query tiles of 16 (compile time: the MFMA block below is straight-line code; a wave-uniform
// runtime test per MFMA made the fully unrolled kernel 2x slower).
template <int T, int MODE, int NT>
__global__ __launch_bounds__(256) void topn_scores_kernel(const float* __restrict__ Y, int64_t n_items, int k,
                                                          const float* __restrict__ Q, int n_queries, int tile_stride,
                                                          int64_t n_out, float* __restrict__ scores,  // MODE 0: [n_queries][n_out]
                                                          TopnState* __restrict__ st, int cap, uint32_t* __restrict__ cand) {
  constexpr int KP = 16 * T;                                  // padded feature count
  __shared__ double sq[NT * KP * 16];                          // [query tile][feature][query in tile]
  constexpr int n_tiles = NT;
  for (int i = threadIdx.x; i < n_tiles * KP * 16; i += 256) {
    const int j = i & 15, f = (i >> 4) % KP, qt = i / (16 * KP);
    const int q = 16 * qt + j;
    sq[i] = (q < n_queries && f < k) ? (double)Q[(int64_t)q * k + f] : 0.0;
  }
  __shared__ uint32_t sthr[TOPN_MAX_QUERIES];
  if (MODE == 1 && threadIdx.x < TOPN_MAX_QUERIES) sthr[threadIdx.x] = threadIdx.x < n_queries ? topn_threshold(st, threadIdx.x) : 0xffffffffu;
  __syncthreads();
  const int lane = threadIdx.x & 63, kk = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  const int64_t step = (int64_t)tile_stride * 16;             // items between processed tiles
  // The order of the features inside the contraction is free (it only moves fp64 rounding), so lane
  // (kk, c) takes the CONTIGUOUS quarter [kk*4T, (kk+1)*4T) of item c's row: at k = 64 exactly one
  // 64-byte line per lane, read with T 16-byte loads.  MFMA step s contracts feature kk*4T + s.
  constexpr int CH = 4 * T;
  auto load_rows16 = [&](int64_t i0, float (&yv)[CH]) {
    const int64_t item = i0 + c;
    const bool ok = item < n_items;
    const float* y = Y + (ok ? item : 0) * k + kk * CH;
    if (k == KP) {
      const float4* y4 = reinterpret_cast<const float4*>(y);
#pragma unroll
      for (int v = 0; v < T; ++v) {
        const float4 t4 = y4[v];
        yv[4 * v] = t4.x; yv[4 * v + 1] = t4.y; yv[4 * v + 2] = t4.z; yv[4 * v + 3] = t4.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < CH; ++s) yv[s] = kk * CH + s < k ? y[s] : 0.f;
    }
  };
  float ynext[CH];
  if (wave * step < n_items) load_rows16(wave * step, ynext);
  for (int64_t i0 = wave * step; i0 < n_items; i0 += n_waves * step) {
    const bool ok = i0 + c < n_items;
    float yv[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s) yv[s] = ynext[s];
    if (i0 + n_waves * step < n_items) load_rows16(i0 + n_waves * step, ynext);  // next tile's rows fly during the MFMAs
    f64x4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f64x4_t{0., 0., 0., 0.};
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      const double a = ok ? (double)yv[s] : 0.0;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double b = sq[(t * KP + kk * CH + s) * 16 + c];  // lane (kk, c) = feature kk*4T+s, query c
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
    // D layout: lane (g = lane>>4, c) reg r = D[row = g + 4r][col = c]: item i0+g+4r, query 16t+c
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int q = 16 * t + c;
      if (q < n_queries) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t it = i0 + kk + 4 * r;
          if (it >= n_items) continue;
          const float sc = (float)acc[t][r];                  // RecommendIterator.java:104
          if (MODE == 0) {
            scores[(int64_t)q * n_out + (i0 / step) * 16 + kk + 4 * r] = sc;
          } else {
            const uint32_t key = score_key(sc);
            if (key >= sthr[q]) {
              const unsigned p = topn_append(st, q);
              if ((int)p < cap) {
                cand[((int64_t)q * cap + p) * 2] = (uint32_t)it;
                cand[((int64_t)q * cap + p) * 2 + 1] = key;
              }
            }
          }
        }
      }
    }
  }
}

// known items of the query's user are never recommended (RecommendIterator.java:75-82)
// position of an item in a score row that holds every tile_stride-th 16-item tile (-1: not in it)
__device__ __forceinline__ int64_t topn_row_slot(int64_t item, int tile_stride) {
  const int64_t tile = item >> 4;
  return tile % tile_stride ? -1 : (tile / tile_stride) * 16 + (item & 15);
}
__global__ void topn_mask_kernel(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                 const int64_t* __restrict__ query_row, int n_queries, int tile_stride, int64_t n_out,
                                 float* __restrict__ scores) {
  const int q = blockIdx.y;
  if (q >= n_queries) return;
  const int64_t r = query_row[q];
  if (r < 0) return;
  const int64_t b = row_ptr[r], e = row_ptr[r + 1];
  for (int64_t i = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t slot = topn_row_slot(col[i], tile_stride);
    if (slot >= 0) scores[(int64_t)q * n_out + slot] = -__builtin_huge_valf();
  }
}
// caller-supplied exclusion lists (anonymous users: the items they were built from, SR:561-606)
__global__ void topn_exclude_kernel(const int64_t* __restrict__ excl_ptr, const int64_t* __restrict__ excl_idx, int n_queries,
                                    int64_t n_items, int tile_stride, int64_t n_out, float* __restrict__ scores) {
  const int q = blockIdx.y;
  if (q >= n_queries) return;
  const int64_t b = excl_ptr[q], e = excl_ptr[q + 1];
  for (int64_t i = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t it = excl_idx[i];
    const int64_t slot = (it >= 0 && it < n_items) ? topn_row_slot(it, tile_stride) : -1;
    if (slot >= 0) scores[(int64_t)q * n_out + slot] = -__builtin_huge_valf();
  }
}
// Filter path: candidates that are known / excluded items are struck out (key 0 never qualifies).
// One workgroup row per query; a query has a few hundred candidates, so every list entry is compared
// against all of them.
__global__ __launch_bounds__(256) void topn_strike_kernel(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                          const int64_t* __restrict__ query_row,
                                                          const int64_t* __restrict__ excl_ptr, const int64_t* __restrict__ excl_idx,
                                                          int n_queries, const TopnState* __restrict__ st, int cap,
                                                          uint32_t* __restrict__ cand) {
  const int q = blockIdx.y;
  if (q >= n_queries) return;
  int64_t b = 0, e = 0;
  const int64_t* list64 = nullptr;
  const int32_t* list32 = nullptr;
  if (query_row) {
    const int64_t r = query_row[q];
    if (r >= 0) { b = row_ptr[r]; e = row_ptr[r + 1]; list32 = col; }
  } else if (excl_ptr) {
    b = excl_ptr[q]; e = excl_ptr[q + 1]; list64 = excl_idx;
  }
  const unsigned n_c = topn_count(st, q) < (unsigned)cap ? topn_count(st, q) : (unsigned)cap;
  uint32_t* cq = cand + (int64_t)q * cap * 2;
  for (int64_t i = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t it = list32 ? (int64_t)list32[i] : list64[i];
    for (unsigned p = 0; p < n_c; ++p)
      if ((int64_t)cq[2 * p] == it) cq[2 * p + 1] = 0u;
  }
}

// Selection state per query: {prefix, remaining} + a 256-bin histogram.  The N-th largest score is
// found by a 4-pass radix select, most significant digit first; every pass is one grid-wide scan of
// the score rows (grid = slabs x queries) into the per-query histogram, followed by a one-thread-per-
// query pick of the digit in which the N-th score lies.  -inf scores (masked items) never qualify.

__global__ void topn_init_kernel(TopnState* __restrict__ st, unsigned* __restrict__ hist, int n_queries, int how_many) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_queries) st[i] = TopnState{0u, (uint32_t)how_many, 0u, 0u};
  if (i < n_queries * 256) hist[i] = 0;
}

__global__ __launch_bounds__(256) void topn_hist_kernel(const float* __restrict__ scores, int64_t n_items, int pass,
                                                        const TopnState* __restrict__ st, unsigned* __restrict__ hist) {
  __shared__ unsigned h[256];
  const int q = blockIdx.y;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int shift = 24 - 8 * pass;
  const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
  const uint32_t prefix = st[q].prefix;
  const uint32_t ninf_key = score_key(-__builtin_huge_valf());
  const float* row = scores + (int64_t)q * n_items;
  const int lane = threadIdx.x & 63;
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n_items; i0 += (int64_t)gridDim.x * 256) {  // whole waves iterate together
    const int64_t i = i0 + threadIdx.x;
    const uint32_t key = i < n_items ? score_key(row[i]) : 0u;
    const bool cand = i < n_items && key > ninf_key && (key & mask) == prefix;
    const int digit = (int)((key >> shift) & 255);
    // scores of one query crowd into a few digits: aggregate equal digits of a wave with ballots and
    // let one lane add the count (an LDS atomic per lane on the same bin serialises)
    uint64_t peers = __ballot(cand);
    if (!peers) continue;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (digit >> bit) & 1;
      const uint64_t m = __ballot(one);
      peers &= one ? m : ~m;
    }
    if (cand && (peers & ((1ull << lane) - 1)) == 0) atomicAdd(&h[digit], (unsigned)__popcll(peers));
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[q * 256 + threadIdx.x], h[threadIdx.x]);
}

// one workgroup per query: pick the digit, clear the histogram for the next pass
__global__ __launch_bounds__(256) void topn_pick_kernel(TopnState* __restrict__ st, unsigned* __restrict__ hist, int pass) {
  __shared__ unsigned h[256];
  const int q = blockIdx.x;
  h[threadIdx.x] = hist[q * 256 + threadIdx.x];
  hist[q * 256 + threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int shift = 24 - 8 * pass;
    unsigned rem = st[q].remaining, d = 255;
    for (;; --d) {
      if (h[d] >= rem || d == 0) break;  // fewer candidates than asked for: ends at digit 0
      rem -= h[d];
    }
    st[q].prefix |= (uint32_t)d << shift;
    st[q].remaining = rem;
  }
}

// out per query: [how_many] (index, key) pairs strictly above the N-th score, then [cap_ties] pairs of ties
__global__ __launch_bounds__(256) void topn_collect_kernel(const float* __restrict__ scores, int64_t n_items, TopnState* __restrict__ st,
                                                           int how_many, int cap_ties, uint32_t* __restrict__ out) {
  const int q = blockIdx.y;
  const uint32_t thr = st[q].prefix;
  const uint32_t ninf_key = score_key(-__builtin_huge_valf());
  const float* row = scores + (int64_t)q * n_items;
  uint32_t* o = out + (int64_t)q * 2 * ((int64_t)how_many + cap_ties);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 256) {
    const uint32_t key = score_key(row[i]);
    if (key <= ninf_key || key < thr) continue;
    if (key > thr) {
      const unsigned p = atomicAdd(&st[q].above, 1u);
      if ((int)p < how_many) {
        o[2 * p] = (uint32_t)i;
        o[2 * p + 1] = key;
      }
    } else {
      const unsigned p = atomicAdd(&st[q].ties, 1u);
      if ((int)p < cap_ties) {
        o[2 * (how_many + p)] = (uint32_t)i;
        o[2 * (how_many + p) + 1] = key;
      }
    }
  }
}

}  // namespace mals
