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
//                 v_mfma_f32_16x16x32_bf16 (item rows and query vectors one bf16 each), fp32 accumulate.
//                 |approx - exact| <= margin_i = 1.25 * 2^-8 |x| |y_i| (bound below), so lb_i = approx - margin_i is a
//                 LOWER bound of the exact score.  The sample is reduced on the fly to a few thousand BUCKET maxima per
//                 query (bucket = a fixed subset of the sample's items) with the item that attains each.
//   2. threshold  buckets whose best item is a known / excluded item of the query are dropped; tau = (a lower estimate
//                 of) the N-th largest of the remaining bucket maxima: at least N distinct unmasked items score >= tau
//                 exactly, hence the exact N-th best score is >= tau.
//   3. filter     ALL items streamed once per pass (Y is read once for up to 256 queries: HBM-bound, n_items * 4k
//                 bytes), same approximate score; item i is a candidate of query q iff approx + margin_i >= tau_q --
//                 a superset of {i: exact score >= tau_q}, which contains the exact top N with all its ties.
//   4. rescore    exact reference arithmetic on the candidates only (a few hundred per query), known items struck.
//   5. final      per query the N largest (score key, ~index) pairs: the N best, ties by ascending index.
// Nothing approximate reaches the output; if a query's candidates overflow their buffer, or its sample holds fewer than
// N unmasked buckets, that query is answered by the dense path (all queries of the pass if a wave's hit list overflowed): exact scores of every item (topn_exact_dense_kernel), known
// items masked, 4-pass radix select of the N-th best, everything above it plus the ties sorted on the host.
//
// Error bound of the approximate score.  Both operands enter as ONE bf16: y = yh + ey, |ey| <= 2^-9 |y|; x = xh + ex,
// |ex| <= 2^-9 |x|.  approx = sum yh xh on v_mfma_f32_16x16x32_bf16 (exact products, fp32 accumulate):
// |approx - sum x y| <= sum |x| |ey| + |ex| |yh| <= 2^-8 (1 + 2^-10) sum |x_f y_f|, plus the accumulation of <= 128 terms
// (< 2^-16 of it) and the reference's own roundings (fp32 products, final cast: <= 2^-23 of it): in total
// < 1.01 * 2^-8 sum |x_f y_f| <= 1.01 * 2^-8 |x|_2 |y_i|_2 (Cauchy-Schwarz).  The kernels use 1.25 * 2^-8, round both
// factors of the margin UP to bf16, and add an absolute floor for the subnormal range.  (Until the middle of round 5 the item
// rows entered split into two bf16 halves with a 2^-8 margin: one more matrix instruction and three more vector
// instructions per element for 20 % fewer candidates -- not worth it: the filter is bound by instruction issue.)  For a
// query of n vectors the filter vector is their mean and |x| is the mean of their norms (an upper bound of the norm of
// the mean, and of the per-vector error sum).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mals {

constexpr int TOPN_MAX_QUERIES = 64;      // dense path: queries per pass
constexpr int TOPN_FILTER_QUERIES = 256;  // filter path: queries scored per read of Y
constexpr int TOPN_FILTER_MAX_N = 64;     // largest how_many the filter path takes (the candidates of a query sit in LDS)
constexpr int TOPN_SAMPLE_GROUPS = 512;   // most workgroups of the sample kernel: 16 buckets per workgroup and query
constexpr float TOPN_MARGIN = 0.0048828125f;  // 1.25 * 2^-8
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

// ---- the streaming kernel: sample (MODE 0) and filter (MODE 1) ---------------------------------------------------------
// S = contraction steps of 32 features (features padded to 32 S); QT = query tiles of 16 per wave, 4 QT per workgroup.
// Query-stationary: wave w of the workgroup keeps its QT query tiles (tiles 4 j + w) as B operands in REGISTERS for the
// whole kernel (QT (S + 1) x 4 registers).  A stage is 64 items = 4 item tiles: wave w loads tile w (prefetched one stage
// ahead; lane (g, c) reads the contiguous quarter [8 S g, 8 S (g + 1)) of item c's row -- the order of the features inside
// the contraction is free), converts it to bf16, computes the margin operand and writes the tile's S + 1 A operands to
// LDS (3 KB at S = 2); after the barrier every wave runs all four tiles against its own query tiles: 4 QT accumulator
// registers per tile, 4-5 waves per SIMD.  (__launch_bounds__(256, 2): with at most 256 registers per lane the compiler puts
// the accumulators into ordinary registers; with the default bound they land in AGPRs and every one of them costs a
// v_accvgpr_read before the hit test -- a third of the kernel's vector instructions.)  MFMA step s contracts features 8 S g + 8 s + j; the last step is the "margin
// step": A = {|y_i|, 1, 1, 1, 0..}, B = the query's margin entry, so every accumulator ends as approx -+ margin_i (- tau_q).
// (The first kernel of round 5 kept the ITEM operands in registers and fetched every query operand from LDS for every
// pair of item tiles -- 75 KB of LDS reads per 32 items, 120 accumulator registers, one wave per SIMD: 138 us per pass
// of 240 queries where the matrix pipe needs 31.)
// MODE 0: sample -- every tile_stride-th tile; acc = approx - margin = a lower bound of the exact score.  Lane (g, c)
//         register r of query tile j keeps the best lower bound it has seen and the item that attained it: the workgroup's
//         BUCKET 4 g + r of query 16 (4 j + w) + c.  bmax / bidx [q][16 blockIdx.x + 4 g + r], row length 16 gridDim.x.
//         An item's bucket follows from its index: stage = item / (64 tile_stride), workgroup = stage % gridDim.x,
//         bucket = 16 workgroup + (item & 15).
// MODE 1: filter -- all tiles; acc = approx + margin_i - tau_q; (item i, query q) is a hit iff acc is not below zero.
//         Hits go to a list PRIVATE to the wave (wave_hits[wave][..], wave_count[wave] counts them, also past wave_cap):
//         positions come from a ballot, not from an atomic -- a returning atomic in this loop waits on the same counter as
//         the prefetched rows.  When a wave has run out of item tiles it sorts its own hits into the per-query candidate lists.
// The filter's grid is persistent (as many workgroups as fit the chip at once).
template <int S, int QT, int MODE, int LM>
__global__ __launch_bounds__(256, 2) void topn_stream_kernel(const float* __restrict__ Y, int64_t n_items, int k,
                                                          const bf16x8* __restrict__ img, int n_queries, int tile_stride,
                                                          float* __restrict__ bmax, uint32_t* __restrict__ bidx,
                                                          const float* __restrict__ tau, int wave_cap, unsigned* __restrict__ wave_count,
                                                          uint2* __restrict__ wave_hits, int cap, unsigned* __restrict__ count,
                                                          uint32_t* __restrict__ cand, unsigned* __restrict__ overflow) {
  constexpr int CH = 8 * S;  // features per lane
  constexpr int E = S + 1;   // A operands per item tile: S contraction steps + the margin step
  __shared__ __attribute__((aligned(16))) bf16x8 sa2[2][4 * E * 64];  // two stages x [item tile of the stage][operand][lane]
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  // this wave's query tiles, once
  bf16x8 bq[QT][S], bm[QT];
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    const int t = 4 * j + w;
#pragma unroll
    for (int s = 0; s < S; ++s) bq[j][s] = img[(t * (S + 1) + s) * 64 + lane];
    bf16x8 m = img[(t * (S + 1) + S) * 64 + lane];
    if (MODE == 1 && g == 0) {  // -tau_q = hi + lo (lo rounded up) into slots 2, 3 of the margin entry
      const int q = 16 * t + c;
      if (q < n_queries) {
        const float tq = tau[q];
        float v = -tq;
        if (!(tq > -__builtin_huge_valf())) v = 1e30f;  // no threshold: everything is a candidate (the pass falls back)
        const __bf16 hi = (__bf16)v;
        m[2] = hi;
        m[3] = bf16_up(v - (float)hi);
      }
    }
    bm[j] = m;
  }
  const int64_t step = (int64_t)tile_stride * 16;       // items between consecutive processed tiles
  const int64_t n_proc = (n_items + step - 1) / step;   // processed tiles
  const int64_t n_stages = (n_proc + 3) / 4;
  // LM = how a lane loads its CH features: 1: k == 32 S, four-float loads; 2: k % 4 == 0, four-float loads with the index
  // clamped into the row; 3: k % 2 == 0, two-float loads, clamped; 0: one load per feature, clamped.  The padding features
  // meet zeros in the query operand, so what the clamped loads bring does not matter (they are left out of |y|) -- and
  // nothing here may branch on or touch a loaded value (a wait for the loads would turn the prefetch into a plain load).
  auto load_rows16 = [&](int64_t i0, float (&yv)[CH]) {
    const int64_t item = i0 + c;
    const float* row = Y + (item < n_items ? item : 0) * k;  // rows past the end read row 0; dropped in the epilogue
    if (LM == 1) {
      const float4* y4 = reinterpret_cast<const float4*>(row + g * CH);
#pragma unroll
      for (int v = 0; v < CH / 4; ++v) {
        const float4 t4 = y4[v];
        yv[4 * v] = t4.x; yv[4 * v + 1] = t4.y; yv[4 * v + 2] = t4.z; yv[4 * v + 3] = t4.w;
      }
    } else if (LM == 2) {
#pragma unroll
      for (int v = 0; v < CH / 4; ++v) {
        const int f = g * CH + 4 * v;
        const float4 t4 = *reinterpret_cast<const float4*>(row + (f < k ? f : k - 4));
        yv[4 * v] = t4.x; yv[4 * v + 1] = t4.y; yv[4 * v + 2] = t4.z; yv[4 * v + 3] = t4.w;
      }
    } else if (LM == 3) {
#pragma unroll
      for (int v = 0; v < CH / 2; ++v) {
        const int f = g * CH + 2 * v;
        const float2 t2 = *reinterpret_cast<const float2*>(row + (f < k ? f : k - 2));
        yv[2 * v] = t2.x; yv[2 * v + 1] = t2.y;
      }
    } else {
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        const int f = g * CH + s;
        yv[s] = row[f < k ? f : k - 1];
      }
    }
  };
  const int64_t wave = (int64_t)blockIdx.x * 4 + w;
  unsigned n_hits = 0;  // wave-uniform
  uint2* my_hits = MODE == 1 ? wave_hits + wave * (int64_t)wave_cap : nullptr;
  f32x4t best[MODE == 0 ? QT : 1];
  uint32_t besti[MODE == 0 ? QT : 1][4];
  if (MODE == 0) {
#pragma unroll
    for (int j = 0; j < QT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        best[j][r] = -__builtin_huge_valf();
        besti[j][r] = 0xffffffffu;
      }
  }
  float ynext[CH];
  if ((int64_t)blockIdx.x < n_stages) load_rows16((4 * (int64_t)blockIdx.x + w) * step, ynext);
  int buf = 0;
  for (int64_t st = blockIdx.x; st < n_stages; st += gridDim.x) {
    float yv[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s) yv[s] = ynext[s];
    if (st + gridDim.x < n_stages) load_rows16((4 * (st + gridDim.x) + w) * step, ynext);  // the next stage's rows fly during this one
    // this wave's tile of the stage as A operands
    bf16x8 ah[S], am;
    {
      float nsq = 0.f;
#pragma unroll
      for (int s = 0; s < CH; ++s)
        if (LM == 1 || g * CH + s < k) nsq = __builtin_fmaf(yv[s], yv[s], nsq);
      nsq += __shfl_xor(nsq, 16);
      nsq += __shfl_xor(nsq, 32);
      const float ny = __builtin_sqrtf(nsq) * 1.0000005f;  // |y_c| of the tile's 16 items
#pragma unroll
      for (int j = 0; j < 8; ++j) am[j] = (__bf16)0.f;
      if (g == 0) {
        am[0] = MODE == 0 ? (__bf16)(-(float)bf16_up(ny)) : bf16_up(ny);  // the sample wants approx - margin
        am[1] = (__bf16)(MODE == 0 ? -1.f : 1.f);
        am[2] = (__bf16)1.f;
        am[3] = (__bf16)1.f;
      }
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) ah[s][j] = (__bf16)yv[8 * s + j];
    }
    // two LDS buffers, one barrier per stage: a wave writes buffer b again only after the barrier of the stage in between,
    // which every wave reaches after it has finished reading b (the second barrier per stage cost a quarter of the kernel)
    bf16x8* sa = sa2[buf];
    buf ^= 1;
#pragma unroll
    for (int s = 0; s < S; ++s) sa[(w * E + s) * 64 + lane] = ah[s];
    sa[(w * E + S) * 64 + lane] = am;
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int64_t i0 = (4 * st + jt) * step;
      bf16x8 a[E];
#pragma unroll
      for (int e = 0; e < E; ++e) a[e] = sa[(jt * E + e) * 64 + lane];
      f32x4t acc[QT];
#pragma unroll
      for (int j = 0; j < QT; ++j) acc[j] = f32x4t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int j = 0; j < QT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s], bq[j][s], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < QT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[S], bm[j], acc[j], 0, 0, 0);
      if (i0 >= n_items) continue;  // uniform
      // D layout: lane (g, c) register r = D[row 4 g + r][col c]: item 4 g + r of the tile, query 16 (4 j + w) + c
      if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t it = i0 + 4 * g + r;
          const bool in = it < n_items;
#pragma unroll
          for (int j = 0; j < QT; ++j)
            if (in && acc[j][r] > best[j][r]) {  // a NaN never wins
              best[j][r] = acc[j][r];
              besti[j][r] = (uint32_t)it;
            }
        }
      } else {
        // A candidate is an accumulator that is not below zero.  As signed integers the non-negative floats (and the
        // quiet NaN the matrix pipe makes of an overflowing approximation, 0x7fc00000) are >= 0 and everything below zero
        // is < 0: one signed maximum over the wave's 4 QT accumulators per item tile, the rare hits inside.
        int32_t mj[QT];
#pragma unroll
        for (int j = 0; j < QT; ++j)
          mj[j] = max(max((int32_t)__float_as_uint(acc[j][0]), (int32_t)__float_as_uint(acc[j][1])),
                      max((int32_t)__float_as_uint(acc[j][2]), (int32_t)__float_as_uint(acc[j][3])));
        int32_t mx = mj[0];
#pragma unroll
        for (int j = 1; j < QT; ++j) mx = max(mx, mj[j]);
        // (TOPN_HIT: also the NaNs with the sign bit set -- as signed integers they lie above -inf's pattern 0xff800000, every
        // finite negative float at or below it.  A non-finite candidate is not dropped silently: its exact score sends the
        // pass to the dense path, which reports it like the reference's checkState(isFinite), RecommendIterator.java:105)
        constexpr int32_t TOPN_HIT = (int32_t)0xff800000;
        if (__ballot(mx > TOPN_HIT)) {
#pragma unroll
          for (int j = 0; j < QT; ++j) {
            if (!__ballot(mj[j] > TOPN_HIT)) continue;
            const int q = 16 * (4 * j + w) + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int64_t it = i0 + 4 * g + r;
              const bool hit = (int32_t)__float_as_uint(acc[j][r]) > TOPN_HIT && it < n_items && q < n_queries;
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
  if (MODE == 0) {
    const int64_t row_len = 16 * (int64_t)gridDim.x;
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      const int q = 16 * (4 * j + w) + c;
      if (q < n_queries) {
        const int64_t at = (int64_t)q * row_len + 16 * (int64_t)blockIdx.x + 4 * g;
        *reinterpret_cast<float4*>(bmax + at) = make_float4(best[j][0], best[j][1], best[j][2], best[j][3]);
        *reinterpret_cast<uint4*>(bidx + at) = make_uint4(besti[j][0], besti[j][1], besti[j][2], besti[j][3]);
      }
    }
  }
  if (MODE == 1) {
    if (lane == 0) wave_count[wave] = n_hits;
    // The wave's own hit list -> the per-query candidate lists (count[q] counts them, also past cap).  Until round 6 a kernel
    // of its own (one launch and its gap per pass: 7 us); here every wave does it for its ~100 hits when it has run out of
    // item tiles: the returning atomics wait on nothing the wave still needs, and the other waves of the CU stream on.
    if (n_hits > (unsigned)wave_cap) {
      if (lane == 0) atomicAdd(overflow, 1u);   // whose hits are missing is not known: the whole pass falls back
    } else if (n_hits) {
      // the list was written by this wave's own lanes: their stores before the other lanes' loads.  WORKGROUP scope: the wave
      // only waits for its own stores (the lines were never read on this CU: nothing stale to hit).  An agent-scope fence
      // here writes back and invalidates the XCD's whole L2 once per wave -- 2048 times per pass: measured, a pass of 240
      // queries went from 86 to 232 us with every streaming kernel beside it losing its cache.
      __threadfence_block();
      for (unsigned i = lane; i < n_hits; i += 64) {
        const uint2 hq = my_hits[i];
        const unsigned pos = atomicAdd(&count[(size_t)hq.y * TOPN_COUNT_STRIDE], 1u);
        if ((int)pos < cap) cand[(int64_t)hq.y * cap + pos] = hq.x;
      }
    }
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
// userTagIDs (RecommendIterator.java:72: "if (userTagIDs.contains(itemID)) return null"): items that are tag pseudo-items
// are never recommended to anybody.  The handle keeps them as one bit per item (mals_set_tag_items).
__device__ __forceinline__ bool topn_tagged(const uint32_t* __restrict__ tag_bits, int64_t item) {
  return tag_bits != nullptr && ((tag_bits[item >> 5] >> (item & 31)) & 1u) != 0u;
}
__global__ void topn_tag_bits_kernel(const int64_t* __restrict__ item_idx, int64_t n, int64_t n_items, uint32_t* __restrict__ bits,
                                     unsigned long long* __restrict__ n_set) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t it = item_idx[i];
  if (it < 0 || it >= n_items) return;   // a tag that owns no row of Y (its entries were all removed)
  const uint32_t bit = 1u << (it & 31);
  if ((atomicOr(&bits[it >> 5], bit) & bit) == 0u) atomicAdd(n_set, 1ull);
}
// dense path: the tagged items of every query's score row
__global__ void topn_mask_tags_kernel(const uint32_t* __restrict__ tag_bits, int n_queries, int64_t n_items, float* __restrict__ scores) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 32-item word per thread
  if (w >= (n_items + 31) / 32) return;
  uint32_t bits = tag_bits[w];
  while (bits) {
    const int b = __ffs((int)bits) - 1;
    bits &= bits - 1;
    const int64_t it = w * 32 + b;
    if (it < n_items)
      for (int q = 0; q < n_queries; ++q) scores[(int64_t)q * n_items + it] = -__builtin_huge_valf();
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

// tau[q] = a value that at least how_many unmasked items of the sample reach: the how_many-th largest of the 1024
// per-thread maxima of the query's bucket row (thread t owns buckets t, t + 1024, ...).  Bucket maxima belong to distinct
// items, so the claim holds; it is a lower estimate of the true how_many-th best of the sample, and a lower tau only lets
// a few more candidates through.  -inf if fewer than how_many threads hold a finite entry (the dense path answers).
// The known / excluded items of the query (RecommendIterator.java:75-82) go first: a bucket whose best item is one of
// them is dropped whole (what else it holds is not known), by this workgroup (the row is its own).
__global__ __launch_bounds__(1024) void topn_threshold_kernel(float* __restrict__ bmax, const uint32_t* __restrict__ bidx, int n_groups,
                                                              int how_many, const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                              const int64_t* __restrict__ query_row, const int64_t* __restrict__ excl_ptr,
                                                              const int64_t* __restrict__ excl_idx, int64_t n_items, int tile_stride,
                                                              const uint32_t* __restrict__ tag_bits, float* __restrict__ tau) {
  __shared__ unsigned h[256], sfx[256];
  __shared__ uint32_t s_prefix, s_rem;
  const int q = blockIdx.x;
  const int64_t n_row = 16 * (int64_t)n_groups;
  float* row = bmax + (int64_t)q * n_row;
  const uint32_t* idx = bidx + (int64_t)q * n_row;
  const uint32_t ninf_key = score_key(-__builtin_huge_valf());
  auto drop = [&](int64_t it) {
    if (it < 0 || it >= n_items) return;
    const int64_t tile = it >> 4;
    if (tile % tile_stride) return;  // not in the sample
    const int64_t b = 16 * (((tile / tile_stride) >> 2) % n_groups) + (it & 15);
    if (idx[b] == (uint32_t)it) row[b] = -__builtin_huge_valf();
  };
  if (query_row) {
    const int64_t r = query_row[q];
    if (r >= 0)
      for (int64_t i = row_ptr[r] + threadIdx.x; i < row_ptr[r + 1]; i += 1024) drop(col[i]);
  }
  if (excl_ptr)
    for (int64_t i = excl_ptr[q] + threadIdx.x; i < excl_ptr[q + 1]; i += 1024) drop(excl_idx[i]);
  __threadfence_block();
  __syncthreads();
  float best = -__builtin_huge_valf();
  // (a bucket won by a tag item is dropped like one won by a known item; an empty bucket's idx is never read as an item:
  // its maximum is -inf already)
  for (int64_t i = threadIdx.x; i < n_row; i += 1024) {
    const float v = row[i];
    if (v > -__builtin_huge_valf() && !topn_tagged(tag_bits, (int64_t)idx[i])) best = fmaxf(best, v);
  }
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
// of the query (RecommendIterator.java:75-82).  One wave per workgroup, 64 candidates at a time: their rows are staged in
// LDS with coalesced loads (16 lanes per row), then lane l computes candidate l's score from LDS in the reference's
// order.  (A lane walking its own row in global memory is 64 loads that each touch 64 different lines: 60 us per pass of
// 240 queries.)  The query's known and excluded items pass through LDS a thousand at a time.
__global__ __launch_bounds__(64) void topn_rescore_kernel(const float* __restrict__ Y, int k, const float* __restrict__ vecs,
                                                          const int64_t* __restrict__ vrow, const int32_t* __restrict__ vptr, const unsigned* __restrict__ count, int cap,
                                                          const uint32_t* __restrict__ cand, const int64_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ col, const int64_t* __restrict__ query_row,
                                                          const int64_t* __restrict__ excl_ptr, const int64_t* __restrict__ excl_idx,
                                                          const uint32_t* __restrict__ tag_bits, uint64_t* __restrict__ pairs,
                                                          unsigned* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  float* ys = reinterpret_cast<float*>(smem);  // [64][k + 1]
  __shared__ uint32_t sk[1024];
  const int q = blockIdx.y, lane = threadIdx.x, pitch = k + 1;
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
  const int64_t n_list = (ke - kb) + (ee - eb);
  const int v0 = vptr[q], v1 = vptr[q + 1];
  for (unsigned base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
    const unsigned p = base + lane;
    const bool valid = p < n;
    const uint32_t it = valid ? cand[(int64_t)q * cap + p] : 0u;
    bool struck = valid && topn_tagged(tag_bits, (int64_t)it);
    for (int64_t c0 = 0; c0 < n_list; c0 += 1024) {
      __syncthreads();
      for (int i = lane; i < 1024 && c0 + i < n_list; i += 64) {
        const int64_t at = c0 + i;
        uint32_t v;
        if (at < ke - kb) {
          v = (uint32_t)col[kb + at];
        } else {
          const int64_t e = excl_idx[eb + (at - (ke - kb))];
          v = (e >= 0 && e < 0xffffffffll) ? (uint32_t)e : 0xffffffffu;  // no candidate has this index
        }
        sk[i] = v;
      }
      __syncthreads();
      const int len = (int)(n_list - c0 < 1024 ? n_list - c0 : 1024);
      for (int i = 0; i < len; ++i) struck |= sk[i] == it;
    }
    // the 64 candidates' rows -> LDS: lane group (lane >> 4) takes row r + (lane >> 4), 16 lanes across the row
    __syncthreads();
    // (all sixteen loads of a sweep are issued before the first store: one memory latency per sweep, not sixteen)
    if ((k & 3) == 0) {
      for (int e0 = 0; e0 < k; e0 += 64) {
        const int e = e0 + 4 * (lane & 15);
        float4 t4[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t item = (uint32_t)__shfl((int)it, 4 * r + (lane >> 4));
          t4[r] = e < k ? *reinterpret_cast<const float4*>(Y + (int64_t)item * k + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (e < k) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* d = ys + (4 * r + (lane >> 4)) * pitch + e;
            d[0] = t4[r].x; d[1] = t4[r].y; d[2] = t4[r].z; d[3] = t4[r].w;
          }
        }
      }
    } else {
      for (int e0 = 0; e0 < k; e0 += 16) {
        const int e = e0 + (lane & 15);
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t item = (uint32_t)__shfl((int)it, 4 * r + (lane >> 4));
          t[r] = e < k ? Y[(int64_t)item * k + e] : 0.f;
        }
        if (e < k) {
#pragma unroll
          for (int r = 0; r < 16; ++r) ys[(4 * r + (lane >> 4)) * pitch + e] = t[r];
        }
      }
    }
    __syncthreads();
    if (valid) {
      uint64_t out = 0;
      if (!struck) {
        const float sc = topn_ref_score(ys + lane * pitch, vecs, vrow, v0, v1, k);
        if (!(fabsf(sc) < __builtin_huge_valf())) atomicAdd(overflow, 1u);   // NaN / infinite: the dense path decides and reports
        out = ((uint64_t)score_key(sc) << 32) | (uint64_t)(0xffffffffu - it);
      }
      pairs[(int64_t)q * cap + p] = out;
    }
  }
}
// The N best of every query: how_many rounds of "largest remaining pair" over the query's candidates in LDS (a few
// hundred pairs, N <= 64: cheaper than sorting them).  Pairs are unique (the item is part of them), larger = better
// score, then lower index.  out_pairs[q][j], j < how_many: the j-th best (0 = none).
// The pass's results in ONE block (the host's pinned block, written from here): out_pairs [n_queries][how_many], then per
// query its candidate count and tau, then the overflow word.
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
  if (threadIdx.x == 0) __threadfence_system();  // the block may be host memory (the pass's pinned result block)
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
        const float sc = topn_ref_score(y, vecs, vrow, v0, v1, k);
        // (a NaN of either sign as THE NaN that score_key sorts above +inf: a non-finite score is then the first result of its
        // query, where topn_emit looks for it -- RecommendIterator.java:105)
        scores[(int64_t)q * n_items + i0 + lane] = sc != sc ? __builtin_nanf("") : sc;
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
