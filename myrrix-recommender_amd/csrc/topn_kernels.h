// topn_kernels.h -- hand-written gfx950 kernels of top-N scoring: what
// ServerRecommender.multithreadedTopN (online/src/net/myrrix/online/ServerRecommender.java:443-508) does with
// RecommendIterator (online/.../RecommendIterator.java:62-109) and TopN (common/src/net/myrrix/common/TopN.java:
// 49-128): score every item against the query's vector(s), skip the known items, keep the N best.
//
// The score of item i for a query with vectors f_1..f_n IS the reference's (RecommendIterator.java:93-104):
//     (float)( (dot(Y_i, f_1) + ... + dot(Y_i, f_n)) / n ),   dot = SimpleVectorMath.dot (SVM:34-41): every product
//     rounded to fp32, summed in fp64 in feature order
// computed by topn_exact_* below with exactly those operations, so scores and -- with ties in ascending item index --
// indices are bit-identical to the oracle's.  What makes that affordable on a million items is a FILTER in front:
//   1. sample     every `stride`-th 16-item tile is scored approximately against all queries of the pass on
//                 v_mfma_f32_16x16x32_bf16 (item rows split into two bf16 halves, query vectors one bf16), fp32
//                 accumulate.  |approx - exact| <= 2^-8 |x| |y_i| (bound below), so lb_i = approx - margin_i is a
//                 LOWER bound of the exact score.  Known items are masked out.
//   2. threshold  per query tau = (a lower estimate of) the N-th largest lb of the sample: at least N unmasked items
//                 score >= tau exactly, hence the exact N-th best score is >= tau.
//   3. filter     ALL items streamed once per pass (Y is read once for up to 256 queries: HBM-bound, n_items * 4k
//                 bytes), same approximate score; item i is a candidate of query q iff approx + margin_i >= tau_q --
//                 a superset of {i: exact score >= tau_q}, which contains the exact top N with all its ties.
//   4. rescore    exact reference arithmetic on the candidates only (a few hundred per query), known items struck.
//   5. final      per query the N largest (score key, ~index) pairs: the N best, ties by ascending index.
// Nothing approximate reaches the output; if a query's candidates overflow their buffer, or its sample holds fewer than
// N unmasked items, the pass is answered by the dense path: exact scores of every item (topn_exact_dense_kernel), known
// items masked, 4-pass radix select of the N-th best, everything above it plus the ties sorted on the host.
//
// Error bound of the approximate score.  The item rows enter split, y = hi + lo + r with hi = bf16(y), lo = bf16(y - hi),
// |r| <= 2^-18 |y|; the query vector enters as one bf16, x = xh + e, |e| <= 2^-9 |x|.  approx = sum (hi + lo) xh on
// v_mfma_f32_16x16x32_bf16 (exact products, fp32 accumulate): |approx - sum x y| <= (2^-9 + 2^-18) sum |x_f y_f| plus the
// accumulation of <= 128 terms (2^-17 of it); the reference's own roundings (fp32 products, final cast) are <= 2^-23 of
// it.  In total < 2^-8.9 sum |x_f y_f| <= 2^-8.9 |x|_2 |y_i|_2 (Cauchy-Schwarz); the kernels use 2^-8, round both
// factors of the margin UP to bf16, and add an absolute floor for the subnormal range.  For a query of n vectors the
// filter vector is their mean and |x| is the mean of their norms (an upper bound of the norm of the mean, and of the
// per-vector error sum).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mals {

constexpr int TOPN_MAX_QUERIES = 64;      // dense path: queries per pass
constexpr int TOPN_FILTER_QUERIES = 256;  // filter path: queries scored per read of Y
constexpr int TOPN_FILTER_MAX_N = 64;     // largest how_many the filter path takes (the candidates of a query sit in LDS)
constexpr float TOPN_MARGIN = 0.00390625f;  // 2^-8
constexpr float TOPN_MARGIN_FLOOR = 1e-30f;
constexpr int TOPN_COUNT_STRIDE = 32;  // candidate counters one per 128-byte line: atomics on one LINE serialise in L2 (97 us per pass when packed)

// fp32 -> uint32 with the same order (NaN sorts above +inf; the scores here are finite or -inf)
__device__ __host__ __forceinline__ uint32_t score_key(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __host__ __forceinline__ float key_score(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}

// ---- the reference's arithmetic ----------------------------------------------------------------------------------
// SimpleVectorMath.dot (SVM:34-41)
__device__ __forceinline__ double topn_ref_dot(const float* __restrict__ y, const float* __restrict__ x, int k) {
  double d = 0.0;
  for (int f = 0; f < k; ++f) d += (double)__fmul_rn(y[f], x[f]);
  return d;
}
// RecommendIterator.java:93-104 for the vectors [v0, v1) of the pass: vector v = row vrow[v] of vecs (vrow NULL: row v)
__device__ __forceinline__ float topn_ref_score(const float* __restrict__ y, const float* __restrict__ vecs, const int64_t* __restrict__ vrow,
                                                int v0, int v1, int k) {
  double sum = 0.0;
  int count = 0;
  for (int v = v0; v < v1; ++v) {
    sum += topn_ref_dot(y, vecs + (vrow ? vrow[v] : (int64_t)v) * k, k);
    ++count;
  }
  return (float)(sum / (double)count);
}

// ---- approximate scores on the bf16 matrix pipe -----------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4t __attribute__((ext_vector_type(4)));

// position of an item in a score row that holds every tile_stride-th 16-item tile (-1: not in it)
__device__ __forceinline__ int64_t topn_row_slot(int64_t item, int tile_stride) {
  const int64_t tile = item >> 4;
  return tile % tile_stride ? -1 : (tile / tile_stride) * 16 + (item & 15);
}

// bf16 >= v (v finite): the margin operands are rounded toward +infinity, so that what the matrix pipe adds is never
// less than the bound asks for
__device__ __forceinline__ __bf16 bf16_up(float v) {
  uint32_t u = __float_as_uint(v);
  if (!(u & 0x80000000u)) u += 0xffffu;  // positive: up = away from zero; negative: truncation is already up
  const uint16_t h = (uint16_t)(u >> 16);
  return __builtin_bit_cast(__bf16, h);
}

// The queries of a pass as MFMA B operands in the order the filter kernel's LDS wants them: per query tile t, S + 1
// entries of 64 lanes x 8 bf16:
//   entry s < S : lane (g, c) = bf16(xbar[query 16 t + c][features 8 S g + 8 s + 0..7]), xbar = the mean of the query's
//                 vectors (fp32)
//   entry S     : the "margin step": lane (0, c) = {2^-8 |x_q| rounded up, floor, 0 (-tau hi), 0 (-tau lo), 0...},
//                 |x_q| = the mean of the norms of the query's vectors; the filter kernel fills the tau slots once the
//                 thresholds are known.  Its A operand is {|y_i|, 1, 1, 1, 0..}, so one more MFMA adds margin_i - tau_q
//                 to every accumulator.
// One workgroup per query tile (16 queries), once per pass: gathers the vectors (row vrow[v] of vecs; vrow NULL: row
// v), builds the tile's image, and clears the pass's counters.
__global__ __launch_bounds__(256) void topn_prepare_kernel(const float* __restrict__ vecs, const int64_t* __restrict__ vrow,
                                                           const int32_t* __restrict__ vptr, int n_queries, int k, int S,
                                                           bf16x8* __restrict__ img, unsigned* __restrict__ count,
                                                           unsigned* __restrict__ overflow) {
  __shared__ float xb[16][129];
  __shared__ float nrm[16];
  const int t = blockIdx.x, qi = threadIdx.x >> 4, sub = threadIdx.x & 15;  // 16 threads per query
  const int q = 16 * t + qi;
  if (blockIdx.x == 0 && threadIdx.x == 0) *overflow = 0u;
  if (sub == 0 && q < n_queries) count[(size_t)q * TOPN_COUNT_STRIDE] = 0u;
  int v0 = 0, v1 = 0;
  if (q < n_queries) {
    v0 = vptr[q];
    v1 = vptr[q + 1];
  }
  const int n = v1 - v0;
  for (int f = sub; f < 128; f += 16) {
    double mean = 0.0;
    if (f < k)
      for (int v = v0; v < v1; ++v) mean += (double)vecs[(vrow ? vrow[v] : (int64_t)v) * k + f];
    xb[qi][f] = n ? (float)(mean / n) : 0.f;
  }
  double norms = 0.0;
  for (int v = v0; v < v1; ++v) {
    const float* x = vecs + (vrow ? vrow[v] : (int64_t)v) * k;
    double ss = 0.0;
    for (int f = sub; f < k; f += 16) ss += (double)x[f] * (double)x[f];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) ss += __shfl_xor(ss, off);  // the 16 lanes of this query
    norms += sqrt(ss);
  }
  if (sub == 0) nrm[qi] = n ? (float)(norms / n * 1.000001) : 0.f;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    for (int s = 0; s < S; ++s) {
      bf16x8 hi;
#pragma unroll
      for (int j = 0; j < 8; ++j) hi[j] = (__bf16)xb[c][8 * S * g + 8 * s + j];
      img[(t * (S + 1) + s) * 64 + lane] = hi;
    }
    bf16x8 m;
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = (__bf16)0.f;
    if (g == 0) {
      if (16 * t + c < n_queries) {
        m[0] = bf16_up(TOPN_MARGIN * nrm[c]);
        m[1] = bf16_up(TOPN_MARGIN_FLOOR);
      } else {
        m[2] = (__bf16)(-1e30f);  // a padding query never has a candidate
      }
    }
    img[(t * (S + 1) + S) * 64 + lane] = m;
  }
}

// S = contraction steps of 32 features (features padded to 32 S); NT = query tiles of 16 per workgroup; the workgroup's
// query tiles are [NT blockIdx.y, NT blockIdx.y + NT).
// MODE 0: sample -- lower bounds (approx - margin) of every tile_stride-th tile into lb[q][slot], row length n_out;
// MODE 1: filter -- (item i, query q) is a hit iff approx + margin_i - tau_q >= 0.  Hits go to a list PRIVATE to the
//         wave (wave_hits[wave][..], wave_count[wave] counts them, also past wave_cap): positions come from a ballot, not
//         from an atomic -- a returning atomic in this loop waits on the same counter as the prefetched rows, and with
//         240 queries nearly every tile has a hit (measured: 284 us per pass with atomics).  topn_scatter_kernel
//         sorts the hits into per-query candidate lists afterwards.
// A wave owns U 16-item tiles at a time (U = 2 in MODE 1: every B operand fetched from LDS serves both): lane (g, c)
// reads the contiguous quarter [8 S g, 8 S (g + 1)) of item c's row (the order of the features inside the contraction
// is free); MFMA step s contracts features 8 S g + 8 s + j.  The item rows are split (hi + lo, 16 bits), the queries are
// not (8 bits): |approx - exact| <= 2^-9 sum |x y|, the margin uses 2^-8.  The grid is persistent (as many workgroups as
// fit the chip at once): the LDS image is loaded once per workgroup.
template <int S, int NT, int MODE, bool ALIGNED>
__global__ __launch_bounds__(256) void topn_filter_kernel(const float* __restrict__ Y, int64_t n_items, int k,
                                                          const bf16x8* __restrict__ img, int n_queries,
                                                          int tile_stride, int64_t n_out, float* __restrict__ lb,
                                                          const float* __restrict__ tau, int wave_cap, unsigned* __restrict__ wave_count,
                                                          uint2* __restrict__ wave_hits) {
  constexpr int CH = 8 * S;  // features per lane
  constexpr int U = MODE == 1 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bf16x8* bq = reinterpret_cast<bf16x8*>(smem);  // [NT][S + 1][64 lanes]
  const int t_base = NT * blockIdx.y;
  for (int i = threadIdx.x; i < NT * (S + 1) * 64; i += 256) bq[i] = img[(int64_t)t_base * (S + 1) * 64 + i];
  __syncthreads();
  if (MODE == 1) {  // -tau_q = hi + lo (lo rounded up) into slots 2, 3 of the margin entries
    for (int i = threadIdx.x; i < NT * 16; i += 256) {
      const int q = 16 * t_base + i;
      if (q < n_queries) {
        const float tq = tau[q];
        float v = -tq;
        if (!(tq > -__builtin_huge_valf())) v = 1e30f;  // no threshold: everything is a candidate (the pass falls back)
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = bf16_up(v - (float)hi);
        bf16x8 m = bq[((i >> 4) * (S + 1) + S) * 64 + (i & 15)];
        m[2] = hi;
        m[3] = lo;
        bq[((i >> 4) * (S + 1) + S) * 64 + (i & 15)] = m;
      }
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  const int64_t step = (int64_t)tile_stride * 16;  // items between consecutive processed tiles
  // ALIGNED (k == 32 S): four-float loads.  Otherwise one load per feature, the index clamped into the row: the padding
  // features meet zeros in the query operand, so what they hold does not matter -- and nothing here may branch on or
  // touch a loaded value (a wait for the loads would turn the prefetch into a plain load).
  auto load_rows16 = [&](int64_t i0, float (&yv)[CH]) {
    const int64_t item = i0 + c;
    const float* row = Y + (item < n_items ? item : 0) * k;  // rows past the end read row 0; dropped in the epilogue
    if (ALIGNED) {
      const float4* y4 = reinterpret_cast<const float4*>(row + g * CH);
#pragma unroll
      for (int v = 0; v < CH / 4; ++v) {
        const float4 t4 = y4[v];
        yv[4 * v] = t4.x; yv[4 * v + 1] = t4.y; yv[4 * v + 2] = t4.z; yv[4 * v + 3] = t4.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        const int f = g * CH + s;
        yv[s] = row[f < k ? f : k - 1];
      }
    }
  };
  // wave w takes tile groups w, w + n_waves, ...; group j = tiles U j .. U j + U - 1 (of the processed tiles)
  const int64_t n_proc = (n_items + step - 1) / step;       // processed tiles
  const int64_t n_groups = (n_proc + U - 1) / U;
  unsigned n_hits = 0;  // wave-uniform
  uint2* my_hits = MODE == 1 ? wave_hits + wave * (int64_t)wave_cap : nullptr;
  float ynext[U][CH];
  if (wave < n_groups)
#pragma unroll
    for (int u = 0; u < U; ++u) load_rows16((wave * U + u) * step, ynext[u]);
  for (int64_t grp = wave; grp < n_groups; grp += n_waves) {
    float yv[U][CH];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int s = 0; s < CH; ++s) yv[u][s] = ynext[u][s];
    if (grp + n_waves < n_groups)
#pragma unroll
      for (int u = 0; u < U; ++u) load_rows16(((grp + n_waves) * U + u) * step, ynext[u]);  // the next rows fly during the MFMAs
    bf16x8 ah[U][S], al[U][S], am[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // |y_c| of the tile's 16 items: this lane's quarter, then the other three
      float nsq = 0.f;
#pragma unroll
      for (int s = 0; s < CH; ++s)
        if (ALIGNED || g * CH + s < k) nsq = __builtin_fmaf(yv[u][s], yv[u][s], nsq);
      nsq += __shfl_xor(nsq, 16);
      nsq += __shfl_xor(nsq, 32);
      const float ny = __builtin_sqrtf(nsq) * 1.0000005f;
#pragma unroll
      for (int j = 0; j < 8; ++j) am[u][j] = (__bf16)0.f;
      if (g == 0) {
        const float sgn = MODE == 0 ? -1.f : 1.f;  // the sample wants approx - margin
        am[u][0] = MODE == 0 ? (__bf16)(-(float)bf16_up(ny)) : bf16_up(ny);
        am[u][1] = (__bf16)sgn;
        am[u][2] = (__bf16)1.f;
        am[u][3] = (__bf16)1.f;
      }
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = yv[u][8 * s + j];
          const __bf16 h = (__bf16)v;
          ah[u][s][j] = h;
          al[u][s][j] = (__bf16)(v - (float)h);
        }
    }
    f32x4t acc[U][NT];
    // The B operands come from LDS one step ahead, into two alternating register sets, pinned with scheduling barriers:
    // left to itself the compiler reloads ONE register set right before its use and waits out the LDS latency in front of
    // every group of matrix instructions (lgkmcnt(0) 45 times per tile pair: the matrix pipe idled half the time).
    bf16x8 bb[2];
    bb[0] = bq[lane];
#pragma unroll
    for (int j = 0; j < NT * (S + 1); ++j) {
      const int t = j / (S + 1), s = j % (S + 1);
      if (j + 1 < NT * (S + 1)) bb[(j + 1) & 1] = bq[(j + 1) * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 b = bb[j & 1];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (s == 0) acc[u][t] = f32x4t{0.f, 0.f, 0.f, 0.f};
        if (s < S) {
          acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[u][s], b, acc[u][t], 0, 0, 0);
          acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[u][s], b, acc[u][t], 0, 0, 0);
        } else {
          acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[u], b, acc[u][t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // D layout: lane (g, c) register r = D[row 4 g + r][col c]: item 4 g + r of the tile, query 16 (t_base + t) + c
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i0 = (grp * U + u) * step;
      if (i0 >= n_items) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int q = 16 * (t_base + t) + c;
        if (MODE == 0) {
          if (q < n_queries) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (i0 + 4 * g + r < n_items) lb[(int64_t)q * n_out + (grp * U + u) * 16 + 4 * g + r] = acc[u][t][r];
          }
        } else {
          // a candidate is an accumulator that is not negative: one test per wave and query tile, the rare hits inside
          const uint32_t all_neg = __float_as_uint(acc[u][t][0]) & __float_as_uint(acc[u][t][1]) & __float_as_uint(acc[u][t][2]) &
                                   __float_as_uint(acc[u][t][3]);
          if (__ballot(!(all_neg >> 31))) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int64_t it = i0 + 4 * g + r;
              const bool hit = !(acc[u][t][r] < 0.f) && it < n_items && q < n_queries;  // a NaN (overflow of the approximation) is a hit too
              const uint64_t hm = __ballot(hit);
              if (hm) {
                const unsigned at = n_hits + (unsigned)__popcll(hm & ((1ull << lane) - 1ull));
                if (hit && at < (unsigned)wave_cap) my_hits[at] = make_uint2((uint32_t)it, (uint32_t)q);
                n_hits += (unsigned)__popcll(hm);
              }
            }
          }
        }
      }
    }
  }
  if (MODE == 1 && lane == 0) wave_count[wave] = n_hits;
}

// the waves' hit lists -> per-query candidate lists (count[q] counts them, also past cap).  One thread per hit slot;
// here a returning atomic costs nothing else its latency.
__global__ __launch_bounds__(256) void topn_scatter_kernel(const unsigned* __restrict__ wave_count, const uint2* __restrict__ wave_hits,
                                                           int wave_cap, int n_waves, int cap, unsigned* __restrict__ count,
                                                           uint32_t* __restrict__ cand, unsigned* __restrict__ overflow) {
  const int w = blockIdx.x;
  if (w >= n_waves) return;
  const unsigned n = wave_count[w];
  if (n > (unsigned)wave_cap) {
    if (threadIdx.x == 0) atomicAdd(overflow, 1u);
    return;
  }
  for (unsigned i = threadIdx.x; i < n; i += 256) {
    const uint2 hq = wave_hits[(int64_t)w * wave_cap + i];
    const unsigned p = atomicAdd(&count[(size_t)hq.y * TOPN_COUNT_STRIDE], 1u);
    if ((int)p < cap) cand[(int64_t)hq.y * cap + p] = hq.x;
  }
}

// known items of the query's user are never recommended (RecommendIterator.java:75-82)
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

// tau[q] = a value that at least how_many entries of row q reach: the how_many-th largest of the 1024 per-thread
// maxima of the row (thread t owns entries t, t + 1024, ...).  Those maxima are distinct entries, so the claim holds;
// it is the exact how_many-th largest unless two of the best how_many share a thread, and a lower tau only lets a
// few more candidates through.  -inf if fewer than how_many threads hold a finite entry (the dense path answers).
// The known / excluded items of the query (RecommendIterator.java:75-82) are taken out of its row first, by this
// workgroup (the row is its own).
__global__ __launch_bounds__(1024) void topn_threshold_kernel(float* __restrict__ rows, int64_t n_row, int how_many,
                                                              const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                              const int64_t* __restrict__ query_row, const int64_t* __restrict__ excl_ptr,
                                                              const int64_t* __restrict__ excl_idx, int64_t n_items, int tile_stride,
                                                              float* __restrict__ tau) {
  __shared__ unsigned h[256], sfx[256];
  __shared__ uint32_t s_prefix, s_rem;
  const int q = blockIdx.x;
  float* row = rows + (int64_t)q * n_row;
  const uint32_t ninf_key = score_key(-__builtin_huge_valf());
  if (query_row) {
    const int64_t r = query_row[q];
    if (r >= 0)
      for (int64_t i = row_ptr[r] + threadIdx.x; i < row_ptr[r + 1]; i += 1024) {
        const int64_t slot = topn_row_slot(col[i], tile_stride);
        if (slot >= 0) row[slot] = -__builtin_huge_valf();
      }
  }
  if (excl_ptr)
    for (int64_t i = excl_ptr[q] + threadIdx.x; i < excl_ptr[q + 1]; i += 1024) {
      const int64_t it = excl_idx[i];
      const int64_t slot = (it >= 0 && it < n_items) ? topn_row_slot(it, tile_stride) : -1;
      if (slot >= 0) row[slot] = -__builtin_huge_valf();
    }
  __threadfence_block();
  __syncthreads();
  float best = -__builtin_huge_valf();
  for (int64_t i = threadIdx.x; i < n_row; i += 1024) best = fmaxf(best, row[i]);   // NaN lower bounds are dropped by fmaxf
  const uint32_t key = score_key(best);
  if (threadIdx.x == 0) {
    s_prefix = 0;
    s_rem = (uint32_t)how_many;
  }
  const int lane = threadIdx.x & 63;
  for (int pass = 0; pass < 4; ++pass) {
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    const uint32_t prefix = s_prefix;
    const bool cnd = key > ninf_key && (key & mask) == prefix;
    const int digit = (int)((key >> shift) & 255);
    uint64_t peers = __ballot(cnd);
    if (peers) {
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const bool one = (digit >> bit) & 1;
        const uint64_t m = __ballot(one);
        peers &= one ? m : ~m;
      }
      if (cnd && (peers & ((1ull << lane) - 1)) == 0) atomicAdd(&h[digit], (unsigned)__popcll(peers));
    }
    __syncthreads();
    // the digit d with  sum_{j > d} h[j] < rem <= sum_{j >= d} h[j]: suffix sums by 256 threads (a serial walk over the
    // bins by one thread cost 10 us per pass)
    if (threadIdx.x < 256) sfx[threadIdx.x] = h[threadIdx.x];
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      unsigned v = 0;
      if (threadIdx.x < 256 && threadIdx.x + off < 256) v = sfx[threadIdx.x + off];
      __syncthreads();
      if (threadIdx.x < 256) sfx[threadIdx.x] += v;
      __syncthreads();
    }
    const unsigned rem = s_rem;
    __syncthreads();
    if (threadIdx.x < 256) {
      const unsigned ge = sfx[threadIdx.x], gt = threadIdx.x < 255 ? sfx[threadIdx.x + 1] : 0u;
      if (rem < 0x80000000u && gt < rem && rem <= ge) {
        s_prefix = prefix | ((uint32_t)threadIdx.x << shift);
        s_rem = rem - gt;
      }
      if (threadIdx.x == 0 && (rem >= 0x80000000u || ge < rem)) s_rem = 0x80000000u;  // fewer candidates than asked for
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) tau[q] = (s_rem >= 0x80000000u || s_prefix <= ninf_key) ? -__builtin_huge_valf() : key_score(s_prefix);
}

// ---- exact rescoring of the candidates --------------------------------------------------------------------------------
// pairs[q][p] = (score key << 32) | ~item for p < min(count[q], cap), 0 for a candidate that is a known / excluded item
// of the query (RecommendIterator.java:75-82).  One thread per candidate.
__global__ __launch_bounds__(256) void topn_rescore_kernel(const float* __restrict__ Y, int k, const float* __restrict__ vecs,
                                                           const int64_t* __restrict__ vrow, const int32_t* __restrict__ vptr, const unsigned* __restrict__ count, int cap,
                                                           const uint32_t* __restrict__ cand, const int64_t* __restrict__ row_ptr,
                                                           const int32_t* __restrict__ col, const int64_t* __restrict__ query_row,
                                                           const int64_t* __restrict__ excl_ptr, const int64_t* __restrict__ excl_idx,
                                                           uint64_t* __restrict__ pairs) {
  const int q = blockIdx.y;
  const unsigned cq = count[(size_t)q * TOPN_COUNT_STRIDE];
  const unsigned n = cq < (unsigned)cap ? cq : (unsigned)cap;
  int64_t kb = 0, ke = 0, eb = 0, ee = 0;
  if (query_row) {
    const int64_t r = query_row[q];
    if (r >= 0) {
      kb = row_ptr[r];
      ke = row_ptr[r + 1];
    }
  }
  if (excl_ptr) {
    eb = excl_ptr[q];
    ee = excl_ptr[q + 1];
  }
  for (unsigned p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
    const uint32_t it = cand[(int64_t)q * cap + p];
    bool struck = false;
    for (int64_t i = kb; i < ke; ++i) struck |= (uint32_t)col[i] == it;           // the same list for the whole workgroup
    for (int64_t i = eb; i < ee; ++i) struck |= excl_idx[i] == (int64_t)it;
    uint64_t out = 0;
    if (!struck) {
      const float sc = topn_ref_score(Y + (int64_t)it * k, vecs, vrow, vptr[q], vptr[q + 1], k);
      out = ((uint64_t)score_key(sc) << 32) | (uint64_t)(0xffffffffu - it);
    }
    pairs[(int64_t)q * cap + p] = out;
  }
}
// The N best of every query: how_many rounds of "largest remaining pair" over the query's candidates in LDS (a few
// hundred pairs, N <= 64: cheaper than sorting them).  Pairs are unique (the item is part of them), larger = better
// score, then lower index.  out_pairs[q][j], j < how_many: the j-th best (0 = none).
// The pass's results in ONE block for one copy to the host: out_pairs [n_queries][how_many], then per query its
// candidate count and tau, then the overflow word.
__global__ __launch_bounds__(256) void topn_final_kernel(const uint64_t* __restrict__ pairs, const unsigned* __restrict__ count, int cap,
                                                         int how_many, uint64_t* __restrict__ out_pairs, unsigned* __restrict__ count_out,
                                                         const float* __restrict__ tau, float* __restrict__ tau_out,
                                                         const unsigned* __restrict__ overflow, unsigned* __restrict__ overflow_out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint64_t* a = reinterpret_cast<uint64_t*>(smem);
  __shared__ uint64_t wbest[4];
  __shared__ int wwhere[4];
  const int q = blockIdx.x;
  const unsigned cq = count[(size_t)q * TOPN_COUNT_STRIDE];
  if (threadIdx.x == 0) {
    count_out[q] = cq;
    tau_out[q] = tau[q];
    if (q == 0) *overflow_out = *overflow;
  }
  const int n = (int)(cq < (unsigned)cap ? cq : (unsigned)cap);
  for (int i = threadIdx.x; i < n; i += 256) a[i] = pairs[(int64_t)q * cap + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int j = 0; j < how_many; ++j) {
    uint64_t best = 0;
    int where = -1;
    for (int i = threadIdx.x; i < n; i += 256) {
      const uint64_t v = a[i];
      if (v > best) {
        best = v;
        where = i;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint64_t ob = __shfl_down(best, off);
      const int ow = __shfl_down(where, off);
      if (ob > best) {
        best = ob;
        where = ow;
      }
    }
    if (lane == 0) {
      wbest[w] = best;
      wwhere[w] = where;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint64_t b = wbest[0];
      int wh = wwhere[0];
      for (int x = 1; x < 4; ++x)
        if (wbest[x] > b) {
          b = wbest[x];
          wh = wwhere[x];
        }
      out_pairs[(int64_t)q * how_many + j] = b;
      if (wh >= 0) a[wh] = 0;
    }
    __syncthreads();
  }
}

// ---- the dense path: exact scores of every item -----------------------------------------------------------------------
// scores[q][i] for a 64-item tile per workgroup: the tile's rows are staged in LDS (coalesced), wave w takes the queries
// w, w + 4, ...: lane = item, the query's vectors are read with wave-uniform addresses.
__global__ __launch_bounds__(256) void topn_exact_dense_kernel(const float* __restrict__ Y, int64_t n_items, int k,
                                                               const float* __restrict__ vecs, const int64_t* __restrict__ vrow,
                                                               const int32_t* __restrict__ vptr, int n_queries, float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  float* ys = reinterpret_cast<float*>(smem);  // [64][k + 1]
  const int pitch = k + 1;
  for (int64_t i0 = (int64_t)blockIdx.x * 64; i0 < n_items; i0 += (int64_t)gridDim.x * 64) {
    const int rows = (int)(n_items - i0 < 64 ? n_items - i0 : 64);
    __syncthreads();
    for (int e = threadIdx.x; e < rows * k; e += 256) ys[(e / k) * pitch + (e % k)] = Y[i0 * k + e];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < rows) {
      const float* y = ys + lane * pitch;
      for (int q = w; q < n_queries; q += 4) {
        const int v0 = __builtin_amdgcn_readfirstlane(vptr[q]), v1 = __builtin_amdgcn_readfirstlane(vptr[q + 1]);
        scores[(int64_t)q * n_items + i0 + lane] = topn_ref_score(y, vecs, vrow, v0, v1, k);
      }
    }
  }
}

// Selection state per query (dense path): {prefix, remaining} + a 256-bin histogram.  The N-th largest score is
// found by a 4-pass radix select, most significant digit first; every pass is one grid-wide scan of
// the score rows (grid = slabs x queries) into the per-query histogram, followed by a one-thread-per-
// query pick of the digit in which the N-th score lies.  -inf scores (masked items) never qualify.
struct TopnState {
  uint32_t prefix;     // digits decided so far (in place, high bits); after 4 passes the N-th best key
  uint32_t remaining;  // rank of the N-th score inside the candidates that match the prefix
  uint32_t above;      // collected: scores strictly above the N-th
  uint32_t ties;       // collected: scores equal to the N-th (all of them counted, cap_ties stored)
};

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
