// topn_kernels.h -- hand-written gfx950 kernels of top-N scoring: what
// ServerRecommender.multithreadedTopN (online/src/net/myrrix/online/ServerRecommender.java:443-508)
// does with RecommendIterator (RecommendIterator.java:62-109) and TopN (common/.../TopN.java:49-128):
// score every item against the query vector, skip the user's known items, keep the N best.
//
//   scores   one wave per 4 items and step, lane (g,c) = item 4*step+g, feature lanes c of every
//            16-block (the gather layout of the factorizer); all queries of the batch are scored per
//            item read, so Y is streamed once per batch: HBM-bound (n_items * 4k bytes).  The dot is
//            the reference's (SimpleVectorMath.java:34-41): fp32 products, fp64 sum, cast to fp32.
//   mask     known items of each query's user -> -inf (RecommendIterator.java:75-82).
//   select   one workgroup per query: 4-pass radix select (8-bit digits of an order-preserving
//            integer image of the score) finds the N-th largest score exactly, then everything above
//            it plus the ties are handed back; no sort of the whole row.
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

template <int T>
__global__ __launch_bounds__(256) void topn_scores_kernel(const float* __restrict__ Y, int64_t n_items, int k,
                                                          const float* __restrict__ Q, int n_queries,
                                                          float* __restrict__ scores) {  // [n_queries][n_items]
  __shared__ float sq[TOPN_MAX_QUERIES * 16 * T];
  for (int i = threadIdx.x; i < n_queries * 16 * T; i += 256) {
    const int q = i / (16 * T), f = i % (16 * T);
    sq[i] = f < k ? Q[(int64_t)q * k + f] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  for (int64_t i0 = wave * 4; i0 < n_items; i0 += n_waves * 4) {
    const int64_t item = i0 + g;
    const bool ok = item < n_items;
    const float* y = Y + (ok ? item : 0) * k;
    float yv[T];
#pragma unroll
    for (int v = 0; v < T; ++v) yv[v] = (ok && 16 * v + c < k) ? y[16 * v + c] : 0.f;
    for (int q = 0; q < n_queries; ++q) {
      double d = 0.0;
#pragma unroll
      for (int v = 0; v < T; ++v) d += (double)__fmul_rn(yv[v], sq[q * 16 * T + 16 * v + c]);
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) d += __shfl_xor(d, off);
      if (ok && c == 0) scores[(int64_t)q * n_items + item] = (float)d;  // RecommendIterator.java:104
    }
  }
}

// known items of the query's user are never recommended (RecommendIterator.java:75-82)
__global__ void topn_mask_kernel(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                 const int64_t* __restrict__ query_row, int n_queries, int64_t n_items,
                                 float* __restrict__ scores) {
  const int q = blockIdx.y;
  if (q >= n_queries) return;
  const int64_t r = query_row[q];
  if (r < 0) return;
  const int64_t b = row_ptr[r], e = row_ptr[r + 1];
  for (int64_t i = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x)
    scores[(int64_t)q * n_items + col[i]] = -__builtin_huge_valf();
}
// caller-supplied exclusion lists (anonymous users: the items they were built from, SR:561-606)
__global__ void topn_exclude_kernel(const int64_t* __restrict__ excl_ptr, const int64_t* __restrict__ excl_idx, int n_queries,
                                    int64_t n_items, float* __restrict__ scores) {
  const int q = blockIdx.y;
  if (q >= n_queries) return;
  const int64_t b = excl_ptr[q], e = excl_ptr[q + 1];
  for (int64_t i = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t it = excl_idx[i];
    if (it >= 0 && it < n_items) scores[(int64_t)q * n_items + it] = -__builtin_huge_valf();
  }
}

// One workgroup per query.  out layout per query: header {n_above, n_ties_total, n_ties_stored, key},
// then up to cap (index, score-bits) pairs: first the n_above items strictly above the N-th score, then
// the stored ties.  -inf scores (masked items) never qualify.
__global__ __launch_bounds__(256) void topn_select_kernel(const float* __restrict__ scores, int64_t n_items, int how_many,
                                                          int cap, uint32_t* __restrict__ out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_remaining, s_above, s_ties, s_stored;
  const float* row = scores + (int64_t)blockIdx.x * n_items;
  uint32_t* o = out + (int64_t)blockIdx.x * (4 + 2 * (int64_t)cap);
  const uint32_t ninf_key = score_key(-__builtin_huge_valf());
  if (threadIdx.x == 0) {
    s_prefix = 0;
    s_remaining = (unsigned)how_many;
    s_above = s_ties = s_stored = 0;
  }
  __syncthreads();
  // radix select of the how_many-th largest key, most significant digit first
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    for (int64_t i = threadIdx.x; i < n_items; i += 256) {
      const uint32_t key = score_key(row[i]);
      if (key > ninf_key && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned rem = s_remaining, d = 255;
      for (;; --d) {
        if (hist[d] >= rem || d == 0) break;
        rem -= hist[d];
      }
      // fewer than how_many candidates in total: d reaches 0 with rem still larger than hist[0]
      s_prefix = prefix | ((uint32_t)d << shift);
      s_remaining = rem;
    }
    __syncthreads();
  }
  const uint32_t thr = s_prefix;  // key of the how_many-th best score (or the smallest candidate key)
  for (int64_t i = threadIdx.x; i < n_items; i += 256) {
    const float sc = row[i];
    const uint32_t key = score_key(sc);
    if (key <= ninf_key) continue;
    if (key > thr) {
      const unsigned p = atomicAdd(&s_above, 1u);
      if ((int)p < cap) {
        o[4 + 2 * p] = (uint32_t)i;
        o[4 + 2 * p + 1] = key;
      }
    } else if (key == thr) {
      atomicAdd(&s_ties, 1u);
    }
  }
  __syncthreads();
  // ties after the strictly better ones, as many as fit
  const unsigned above = s_above < (unsigned)cap ? s_above : (unsigned)cap;
  for (int64_t i = threadIdx.x; i < n_items; i += 256) {
    if (score_key(row[i]) == thr && thr > ninf_key) {
      const unsigned p = atomicAdd(&s_stored, 1u);
      if (above + p < (unsigned)cap) {
        o[4 + 2 * (above + p)] = (uint32_t)i;
        o[4 + 2 * (above + p) + 1] = thr;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    o[0] = s_above;
    o[1] = s_ties;
    o[2] = (above + s_stored <= (unsigned)cap) ? s_stored : (unsigned)cap - above;
    o[3] = thr;
  }
}

}  // namespace mals
