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

// Selection state per query: {prefix, remaining} + a 256-bin histogram.  The N-th largest score is
// found by a 4-pass radix select, most significant digit first; every pass is one grid-wide scan of
// the score rows (grid = slabs x queries) into the per-query histogram, followed by a one-thread-per-
// query pick of the digit in which the N-th score lies.  -inf scores (masked items) never qualify.
struct TopnState {
  uint32_t prefix;     // digits decided so far (in place, high bits)
  uint32_t remaining;  // rank of the N-th score inside the candidates that match the prefix
  uint32_t above;      // collected: scores strictly above the N-th
  uint32_t ties;       // collected: scores equal to the N-th (all of them counted, cap_ties stored)
};

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
