// dual_kernels.h -- the "short row" solve path of the ALS half-iteration (gfx950, wave64).
//
// AlternatingLeastSquares.Worker.call (ALS:447-494) solves, per row u with n_u entries,
//     W_u x_u = b_u,   W_u = G + sum_i w_i y_i y_i^T + rho_u I,   b_u = sum_i cb_i y_i,
//     w_i = alpha |r_ui|,  cb_i = [r_ui > 0] (1 + w_i),  rho_u = lambda alpha n_u,  G = Y^T Y.
// W_u is k x k however short the row is.  With the eigendecomposition G = Q diag(L) Q^T (host fp64,
// host_eigen.h, once per half-iteration) and y' = Q^T y,  A_u = G + rho_u I = Q diag(L + rho_u) Q^T is
// diagonal in the rotated coordinates, W_u = A_u + V^T C V (V = the n_u gathered rows, C = diag(w)) is
// a rank-n_u update of it, and the push-through identity gives
//     x_u = W_u^-1 V^T cb = A_u^-1 V^T C^1/2 (I + C^1/2 V A_u^-1 V^T C^1/2)^-1 C^-1/2 cb
//         = Q D^1/2 Z^T S^-1 q,       D = diag(1 / (L + rho_u)),  Z = C^1/2 V' D^1/2  (n_u x k),
//                                     S = I + Z Z^T  (n_u x n_u, SPD, eigenvalues >= 1),  q_i = cb_i / sqrt(w_i)
// -- the SAME x_u (not an approximation), from an n_u x n_u Cholesky instead of a k x k one.  For
// n_u <= k/2 that is 8x fewer factorization flops and a 4x smaller per-row Gramian (Z Z^T contracts
// over k, n_u^2 k / 2 products instead of n_u k^2 / 2).  Measured error against the fp64 oracle is on a
// par with the direct path (tests/test_gpu_dual.py; the algorithm alone, numpy on the CPU: tests/dual_emulation.py; 2-5e-7 relative, also
// with cond(G) = 1e6 and lambda = 0).
//
// Pieces (all launched by mals_api.hip on the handle's stream):
//   rotate_rows_kernel<T,false>  Mr = M Q, row stride padded to 16T floats (fp32 or fp64 matrix cores), plus the
//                                bound max |y'_f| / sqrt(L_f + rho_1) for the f16 operand scale
//   als_dual_kernel<T,TN>        one wave per row with 16(TN-1) < n_u <= 16 TN: gather the rotated rows
//                                with lane <-> entry (the MFMA operand layout of Z, no cross-lane moves),
//                                S on v_mfma_f32_16x16x32_f16 with split-f16 operands (as gather_row_h),
//                                in-register Cholesky + solves on the TN x TN tiles (als_kernels.h),
//                                x' = D^1/2 Z^T v through an MFMA transposition of Z
//   rotate_rows_kernel<T,true>   x = Q x' in place over the rows of the dual work lists
// The path is only taken in the reference's default mode (no reconstructR / lossIgnoresUnspecified),
// alpha > 0, and min_f L_f + lambda alpha >= 1e-4 (A_u safely positive definite: the direct path could
// not flag such a row as singular either); otherwise the same lists run through the direct kernels.
#pragma once
#include "als_kernels.h"

namespace mals {

struct DualParams {
  const int32_t* col;
  const float* val;
  const float* Mr;          // rotated opposing factors, row stride 16T floats, zero padded
  const float* lam;         // eigenvalues L_f of G, 16T floats (0 for the padding features)
  const unsigned* zbound;   // bit pattern of max_{rows,f} |y'_f| / sqrt(L_f + rho_1)  (rotate_rows_kernel)
  float* out;               // this side's factor replica + row_offset*k: x' is written here
  const WorkItem* items;
  unsigned long long* bad_row;
  int64_t n_work;
  int32_t k;
  float alpha;
  float lambda_alpha;
  float sqrt_w_max;         // sqrt(alpha * max |r|): bound of C^1/2
  unsigned* xbound;         // bit pattern of the largest |x'| stored so far (operand scale of the un-rotation)
  int* any_marked;          // as in SolveParams
  uint8_t* refine_flag;     // as in SolveParams: rows whose S = I + Z Z^T is ill-conditioned for fp32 go to als_refine_kernel
  float refine_limit;
};

struct RotateParams {
  const float* src;
  float* dst;
  const double* B;          // k x 16T row-major: forward Q (zero padded columns), listed Q^T
  const float* Bf;          // the same in fp32
  const WorkItem* items;    // LISTED: the rows are items[i].id
  const float* dmax;        // forward: 1 / sqrt(L_f + rho_1) per output column (16T floats)
  unsigned* zbound;
  int64_t n_rows;           // rows (forward) or list length (LISTED)
  int32_t k;
  int32_t src_stride, dst_stride, dst_cols;
};

// largest row class of the dual path for T feature blocks: the n_u x n_u system must be clearly smaller than the
// k x k one and its operands must fit the registers next to it (T = 8: TN = 5 spills 268 B and loses, measured)
__host__ __device__ constexpr int dual_max_blocks(int T) { return T >= 8 ? 4 : (T >= 6 ? 4 : (T >= 4 ? 3 : T / 2)); }
__host__ __device__ constexpr int dual_waves(int T, int TN) {
  const int regs = 4 * T * TN + 4 * tri(TN) + 72;
  return regs > 168 ? 2 : (regs > 128 ? 3 : (regs > 102 ? 4 : 5));
}

// dst rows = src rows x B on the matrix cores, rounded to fp32 at the end.  One wave per 64 (fp32) or 32
// (fp64) rows; B (k x 16T, <= 128 KB as doubles) is read through L1/L2.
// Two arithmetics, chosen per half-iteration on the host from the spectrum of G:
//   F64 = false  v_mfma_f32_16x16x4_f32 (33 cycles): fp32 products and sums -- fine as long as no rotated
//                coordinate is a tiny difference of large products, i.e. for cond(G + rho I) <= 1e5;
//   F64 = true   v_mfma_f64_16x16x4_f64 (64 cycles): with one factor row 1e4 times the others, G's leading
//                eigenvector is that row's direction and its OTHER rotated coordinates are 1e-7 of its norm --
//                an fp32 accumulation loses them entirely (measured 1e-2 error in x).
// The un-rotation x = Q x' (LISTED) has no such cancellation and always runs in fp32.
// C/D layouts: f32: lane l reg r = D[row 4 (l>>4) + r][col l&15];  f64: D[row (l>>4) + 4 r][col l&15].
template <int T, bool LISTED, bool F64>
__global__ __launch_bounds__(256, 2) void rotate_rows_kernel(RotateParams p) {
  constexpr int KP = 16 * T;
  // 16-row tiles per wave: every B operand read from L1/L2 feeds NT matrix instructions, enough of them
  // (NT T x 33 or 64 cycles per T loads) to cover the latency of the next step's reads
  constexpr int NT = F64 ? 2 : 4;
  typedef typename std::conditional<F64, double, float>::type real;
  typedef typename std::conditional<F64, f64x4, f32x4>::type real4;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t n_tiles = (p.n_rows + 16 * NT - 1) / (16 * NT);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const real* B = reinterpret_cast<const real*>(F64 ? (const void*)p.B : (const void*)p.Bf);
  float dmaxcol[T];
#pragma unroll
  for (int t = 0; t < T; ++t) dmaxcol[t] = (!LISTED && p.dmax) ? p.dmax[16 * t + c] : 0.f;
  float zmax = 0.f;
  const int n_steps = (p.k + 3) >> 2;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += n_waves) {
    const float* sp[NT];
    bool okc[NT];
#pragma unroll
    for (int h = 0; h < NT; ++h) {
      const int64_t rc = tile * (16 * NT) + 16 * h + c;
      okc[h] = rc < p.n_rows;
      const int64_t row_c = LISTED ? (int64_t)p.items[okc[h] ? rc : p.n_rows - 1].id : (okc[h] ? rc : 0);
      sp[h] = p.src + row_c * (int64_t)p.src_stride + g;
    }
    real4 acc[NT][T];
#pragma unroll
    for (int h = 0; h < NT; ++h)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[h][t] = real4{0, 0, 0, 0};
    for (int s = 0; s < n_steps; ++s) {
      const int f = 4 * s + g;
      const bool okf = f < p.k;
      real a[NT];
#pragma unroll
      for (int h = 0; h < NT; ++h) a[h] = (okc[h] && okf) ? (real)sp[h][4 * s] : (real)0;
      const real* bp = B + (int64_t)(okf ? f : 0) * KP + c;
      real b[T];
#pragma unroll
      for (int t = 0; t < T; ++t) b[t] = bp[16 * t];
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int h = 0; h < NT; ++h) {
          if constexpr (F64) {
            acc[h][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h], b[t], acc[h][t], 0, 0, 0);
          } else {
            acc[h][t] = mfma4(a[h], b[t], acc[h][t]);
          }
        }
    }
#pragma unroll
    for (int h = 0; h < NT; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t rr = tile * (16 * NT) + 16 * h + (F64 ? g + 4 * r : 4 * g + r);
        if (rr < p.n_rows) {
          const int64_t row = LISTED ? (int64_t)p.items[rr].id : rr;
          float* o = p.dst + row * (int64_t)p.dst_stride;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const float v = (float)acc[h][t][r];
            if (16 * t + c < p.dst_cols) o[16 * t + c] = v;
            if (!LISTED) zmax = fmaxf(zmax, fabsf(v) * dmaxcol[t]);
          }
        }
      }
  }
  if (!LISTED && p.zbound) {
    for (int off = 32; off > 0; off >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, off));
    if (lane == 0 && zmax > __uint_as_float(__builtin_nontemporal_load(p.zbound))) atomicMax(p.zbound, __float_as_uint(zmax));
  }
}

// The same product on the f16 matrix pipe (v_mfma_f32_16x16x32_f16), operands split into two f16 halves
// like the per-row Gramians: 3 x 16 cycles per 16 x 16 x 32 block instead of 8 x 33 -- the forward rotation
// of a well-conditioned half-iteration and every un-rotation run here (k a multiple of 8).
//   A (rows): lane (g,c) reads features 32 q + 8 g .. + 7 of row c of a 16-row tile (two 16-byte loads), times a
//             power of two sA with max |value| sA <= 2^14 -- forward: max |y| <= sqrt(max_f G_ff) from the host;
//             un-rotation: the largest |x'| the dual kernels stored (a device scalar they maintain) -- then
//             hi = the value rounded to f16, lo = the exact residual rounded to f16 (pk_rn16);
//   B (Q):    split once on the host into the operand layout  Bs[q][t][hi|lo][lane] (16 bytes each), scale 2^13.
// One wave per NT 16-row tiles that share every B operand.
struct RotateSplitParams {
  const float* src;
  float* dst;
  const i32x4* Bs;          // [KC][T][2][64] operand-layout halves of Q (forward) or Q^T (LISTED), times 2^13
  const WorkItem* items;
  const float* dmax;
  unsigned* zbound;
  const unsigned* bound_bits;  // device scalar: bit pattern of a bound on |src| (LISTED); NULL: bound_host
  float bound_host;
  int64_t n_rows;
  int32_t k;
  int32_t src_stride, dst_stride, dst_cols;
};

// 16-row tiles per wave
__host__ __device__ constexpr int rotate_split_tiles(int T) { return T >= 7 ? 2 : 4; }

template <int T, bool LISTED>
__global__ __launch_bounds__(256, 2) void rotate_rows_split_kernel(RotateSplitParams p) {
  constexpr int KC = (T + 1) / 2, NT = rotate_split_tiles(T);
  // the B operands (<= 64 KB) once per workgroup in LDS: read from L2 right before their use they cost one memory
  // latency per pair of tiles (measured: 0.3 ns per row instead of the 0.07 ns the matrix pipe needs)
  __shared__ i32x4 sB[KC * T * 2 * 64];
  for (int e = threadIdx.x; e < KC * T * 2 * 64; e += 256) sB[e] = p.Bs[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t n_tiles = (p.n_rows + 16 * NT - 1) / (16 * NT);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  float sa, inv_scale;
  {
    const float bound = p.bound_bits ? __uint_as_float(uniform((int)*p.bound_bits)) : p.bound_host;
    const int eb = ((__float_as_int(bound) >> 23) & 255) - 126;   // bound < 2^eb
    int pw = 14 - eb;
    pw = pw < -100 ? -100 : (pw > 100 ? 100 : pw);
    if (!(bound > 0.f)) pw = 0;
    sa = __int_as_float((pw + 127) << 23);
    inv_scale = __int_as_float((127 - pw) << 23) * (1.0f / 8192.0f);   // 1 / (sA * 2^13)
  }
  float dmaxcol[T];
#pragma unroll
  for (int t = 0; t < T; ++t) dmaxcol[t] = (!LISTED && p.dmax) ? p.dmax[16 * t + c] : 0.f;
  float zmax = 0.f;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += n_waves) {
    const float* sp[NT];
    bool okc[NT];
#pragma unroll
    for (int h = 0; h < NT; ++h) {
      const int64_t rc = tile * (16 * NT) + 16 * h + c;
      okc[h] = rc < p.n_rows;
      const int64_t row_c = LISTED ? (int64_t)p.items[okc[h] ? rc : p.n_rows - 1].id : (okc[h] ? rc : 0);
      sp[h] = p.src + row_c * (int64_t)p.src_stride + 8 * g;
    }
    f32x4 acc[NT][T];
#pragma unroll
    for (int h = 0; h < NT; ++h)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // chunk q+1 is requested before the products of chunk q
    f32x4 raw[NT][2], nxt[NT][2];
    auto load_chunk = [&](int q, f32x4 (&dstv)[NT][2]) {
#pragma unroll
      for (int h = 0; h < NT; ++h) {
        const bool okf = okc[h] && 32 * q + 8 * g < p.k;   // k is a multiple of 8: a lane's 8 features are all in or all out
        dstv[h][0] = okf ? *reinterpret_cast<const f32x4*>(sp[h] + 32 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        dstv[h][1] = okf ? *reinterpret_cast<const f32x4*>(sp[h] + 32 * q + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    load_chunk(0, raw);
#pragma unroll
    for (int q = 0; q < KC; ++q) {
      if (q + 1 < KC) load_chunk(q + 1, nxt);
      ZOp<8> ah[NT], al[NT];
#pragma unroll
      for (int h = 0; h < NT; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 z = raw[h][u] * sa;
          const int h01 = pk_rn16(z[0], z[1]), h23 = pk_rn16(z[2], z[3]);
          ah[h].r[2 * u] = h01;
          ah[h].r[2 * u + 1] = h23;
          al[h].r[2 * u] = pk_rn16(residual_lo(h01, z[0]), residual_hi(h01, z[1]));
          al[h].r[2 * u + 1] = pk_rn16(residual_lo(h23, z[2]), residual_hi(h23, z[3]));
        }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const i32x4* bp = sB + ((q * T + t) * 2) * 64 + lane;
        const i32x4 b_hi = bp[0], b_lo = bp[64];
        ZOp<8> bh, bl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bh.r[e] = b_hi[e];
          bl.r[e] = b_lo[e];
        }
#pragma unroll
        for (int h = 0; h < NT; ++h) {
          acc[h][t] = mfma_h<8>(ah[h], bh, acc[h][t]);
          acc[h][t] = mfma_h<8>(ah[h], bl, acc[h][t]);
          acc[h][t] = mfma_h<8>(al[h], bh, acc[h][t]);
        }
        // a few LDS reads ahead are enough; left alone the scheduler hoists all KC x T of them (640 B of spills)
        if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int h = 0; h < NT; ++h) {
        raw[h][0] = nxt[h][0];
        raw[h][1] = nxt[h][1];
      }
    }
#pragma unroll
    for (int h = 0; h < NT; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t rr = tile * (16 * NT) + 16 * h + 4 * g + r;
        if (rr < p.n_rows) {
          const int64_t row = LISTED ? (int64_t)p.items[rr].id : rr;
          float* o = p.dst + row * (int64_t)p.dst_stride;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const float v = acc[h][t][r] * inv_scale;
            if (16 * t + c < p.dst_cols) o[16 * t + c] = v;
            if (!LISTED) zmax = fmaxf(zmax, fabsf(v) * dmaxcol[t]);
          }
        }
      }
  }
  if (!LISTED && p.zbound) {
    for (int off = 32; off > 0; off >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, off));
    if (lane == 0 && zmax > __uint_as_float(__builtin_nontemporal_load(p.zbound))) atomicMax(p.zbound, __float_as_uint(zmax));
  }
}

// rows outside every work list that need no arithmetic: x = 0 (empty rows: b = 0)
// (one wave per row, a lane per feature: a thread per element with its 64-bit division ran at 1.9 TB/s on C4's 1.4M
// empty user rows)
__global__ void zero_rows_kernel(const WorkItem* __restrict__ items, int64_t n, int k, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  constexpr int ROWS = 8;  // rows per wave
  if ((k & 3) == 0 && k <= 256) {
    // 16-byte stores, 64 / (k / 4) rows per instruction (k = 128: two): rows are k floats apart, 16-byte aligned
    const int lanes_per_row = k >> 2, rows_per_inst = 64 / lanes_per_row;
    const int sub = lane / lanes_per_row, c4 = lane - sub * lanes_per_row;
    if (sub >= rows_per_inst) return;
    for (int r = sub; r < ROWS; r += rows_per_inst) {
      const int64_t i = wave * ROWS + r;
      if (i < n) reinterpret_cast<f32x4*>(out + (int64_t)items[i].id * k)[c4] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  } else {
    for (int r = 0; r < ROWS; ++r) {
      const int64_t i = wave * ROWS + r;
      if (i >= n) return;
      float* o = out + (int64_t)items[i].id * k;
      for (int c = lane; c < k; c += 64) o[c] = 0.f;
    }
  }
}

template <int TN>
struct DualEntries {
  int col[TN];
  float r[TN];
};

template <int TN>
__device__ __forceinline__ DualEntries<TN> dual_load_entries(const DualParams& p, const WorkItem& w, int c) {
  DualEntries<TN> e;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = 16 * b + c;
    const int nn = n < w.len ? n : w.len - 1;
    e.col[b] = __builtin_nontemporal_load(p.col + w.begin + nn);
#ifdef MALS_DUAL_CACHED_COLS   // ablation (exp builds, wrong results): every dual gather hits the cache
    e.col[b] &= 0xfff;
#endif
    e.r[b] = __builtin_nontemporal_load(p.val + w.begin + nn);
  }
  return e;
}

__device__ __forceinline__ WorkItem dual_load_item(const DualParams& p, int64_t it) {
  WorkItem w;
  if (it < p.n_work) {
    w = p.items[it];
  } else {
    w.begin = 0;
    w.len = -1;
    w.id = 0;
  }
  return w;
}

template <int T, int TN>
__global__ __launch_bounds__(256, dual_waves(T, TN)) void als_dual_kernel(DualParams p) {
  constexpr int KP = 16 * T, KC = (T + 1) / 2, NMAX = 16 * TN;
  __shared__ float sD[NMAX * KP];  // sD[n-1][f] = 1 / sqrt(L_f + lambda alpha n)
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  for (int e = threadIdx.x; e < NMAX * KP; e += 256) {
    const int n = e / KP + 1, f = e - (n - 1) * KP;
    // v_rsq_f32 (1 ulp): a workgroup of a short class list has four rows to solve, the IEEE sqrt + division of its
    // NMAX x KP table entries was a quarter of its instructions
    sD[e] = __builtin_amdgcn_rsqf(p.lam[f] + p.lambda_alpha * (float)n);
  }
  __syncthreads();
  const int wave = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  int64_t it = wave;
  if (it >= p.n_work) return;
  // operand scale: |z| <= sqrt_w_max * zbound;  Sc = 2^pw with |z| Sc <= 2^14 (f16 cannot overflow)
  float sc, inv_sc, inv_sc2;
  {
    const float bound = __uint_as_float(uniform((int)*p.zbound)) * p.sqrt_w_max;
    int e = ((__float_as_int(bound) >> 23) & 255) - 126;  // bound < 2^e
    int pw = 14 - e;
    pw = pw < -60 ? -60 : (pw > 60 ? 60 : pw);
    if (!(bound > 0.f)) pw = 0;
    sc = __int_as_float((pw + 127) << 23);
    inv_sc = __int_as_float((127 - pw) << 23);
    inv_sc2 = __int_as_float((127 - 2 * pw) << 23);
  }
  // identity operand of the transposition MFMA (v_mfma_f32_16x16x16_f16): B[k = 4g+s][j = c] = [k == j]
  i32x2h ident;
  {
    const int s = c - 4 * g;
    ident[0] = s == 0 ? 0x3c00 : (s == 1 ? 0x3c000000 : 0);
    ident[1] = s == 2 ? 0x3c00 : (s == 3 ? 0x3c000000 : 0);
  }
  float idn[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) idn[r] = (4 * g + r == c) ? 1.f : 0.f;

  WorkItem cur = dual_load_item(p, it);
  WorkItem nxt = dual_load_item(p, it + n_waves);
  DualEntries<TN> en = dual_load_entries<TN>(p, cur, c);
  float xmax = 0.f;
  for (;;) {
    const int n = cur.len;
    // (1) all gathers of the row: lane (g,c) reads, for entry c of every 16-entry block, 16 bytes at
    // float offset 16 j + 4 g of the rotated row -- 64 contiguous bytes per entry and instruction
    // (entries past the end of the row carry a clamped column and a zero weight, dual_load_entries: their
    // gathers are valid and contribute nothing -- no predicate, no zero fill)
    f32x4 raw[TN][T];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const float* ptr = p.Mr + ((uint64_t)(uint32_t)en.col[b] * (uint32_t)KP + (uint32_t)(4 * g));
#pragma unroll
      for (int j = 0; j < T; ++j) raw[b][j] = *reinterpret_cast<const f32x4*>(ptr + 16 * j);
    }
    // weights of this lane's entries (ALS:471-482)
    float ws[TN], qcol[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const bool ok = 16 * b + c < n;
      const float r = en.r[b];
      const float ar = p.alpha * fabsf(r);
      // v_sqrt_f32 / v_rcp_f32 (1 ulp each, as chunk_weights_h): the IEEE expansions were 60 instructions per entry block
      const float w = ok ? __builtin_amdgcn_sqrtf(ar) : 0.f;
      const float cb = (ok && r > 0.f) ? 1.f + ar : 0.f;
      qcol[b] = w > 0.f ? cb * __builtin_amdgcn_rcpf(w) : 0.f;
      ws[b] = w * sc;
    }
    // next row's entries and the item after it land during this row's arithmetic
    if (nxt.len > 0) en = dual_load_entries<TN>(p, nxt, c);
    const WorkItem nx2 = dual_load_item(p, it + 2 * n_waves);
    __builtin_amdgcn_sched_barrier(0);
    // (2) z = sqrt(w) Sc d y', split into two f16 halves (22 significand bits, as gather_row_h)
    ZOp<8> zh[TN][KC], zl[TN][KC];
    const float* dn = sD + (n - 1) * KP + 4 * g;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
#pragma unroll
      for (int j = 0; j < 2 * KC; ++j) {
        if (j < T) {
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(dn + 16 * j);
          // whole-vector products (v_pk_mul_f32: two values per issue slot); the residuals z - (float)zh as
          // v_fma_mix_f32 on the packed halves, written out: hipcc only finds that form on scalar products
          const f32x4 z4 = raw[b][j] * d4 * ws[b];
          const int h01 = pk_rn16(z4[0], z4[1]), h23 = pk_rn16(z4[2], z4[3]);
          zh[b][j >> 1].r[2 * (j & 1)] = h01;
          zh[b][j >> 1].r[2 * (j & 1) + 1] = h23;
          zl[b][j >> 1].r[2 * (j & 1)] = pk_rn16(residual_lo(h01, z4[0]), residual_hi(h01, z4[1]));
          zl[b][j >> 1].r[2 * (j & 1) + 1] = pk_rn16(residual_lo(h23, z4[2]), residual_hi(h23, z4[3]));
        } else {  // odd T: the upper half of the last 32-feature chunk is padding
          zh[b][j >> 1].r[2 * (j & 1)] = 0;
          zh[b][j >> 1].r[2 * (j & 1) + 1] = 0;
          zl[b][j >> 1].r[2 * (j & 1)] = 0;
          zl[b][j >> 1].r[2 * (j & 1) + 1] = 0;
        }
      }
    }
    // (3) S = I + Z Z^T on the f16 matrix pipe: zh zh^T + zh zl^T + zl zh^T, contraction over the features
    f32x4 acc[tri(TN)];
#pragma unroll
    for (int t = 0; t < tri(TN); ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KC; ++q) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = i; j < TN; ++j) acc[tidx(TN, i, j)] = mfma_h<8>(zh[i][q], zh[j][q], acc[tidx(TN, i, j)]);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = i; j < TN; ++j) acc[tidx(TN, i, j)] = mfma_h<8>(zh[i][q], zl[j][q], acc[tidx(TN, i, j)]);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = i; j < TN; ++j) acc[tidx(TN, i, j)] = mfma_h<8>(zl[i][q], zh[j][q], acc[tidx(TN, i, j)]);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = i; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[tidx(TN, i, j)][r] = i == j ? fmaf(acc[tidx(TN, i, j)][r], inv_sc2, idn[r]) : acc[tidx(TN, i, j)][r] * inv_sc2;
    // (4) S v = q: Cholesky + triangular solves on the TN x TN tiles, in registers
    float minpiv = 3.0e38f, smax;
    float vcol[TN];
    if constexpr (TN >= 2) {
      const float inv_s2row = row_scale<TN>(acc, qcol, lane, smax);
      cholesky_tiles<TN, true>(acc, lane, minpiv);
      minpiv *= inv_s2row;
    } else {
      smax = __int_as_float(max_entry_bits<TN>(acc, lane));
      cholesky_tiles<TN>(acc, lane, minpiv);
    }
    solve_tiles<TN>(acc, qcol, vcol, lane);
    // (5) x' = D^1/2 Z^T v / Sc.  Z sits with lane <-> entry; an MFMA against the identity turns one
    // 16-entry x 16-feature piece into accumulator layout (lane (g,c) reg r = z[entry 4g+r][feature c],
    // zh + zl added exactly in fp32), where the sum over the entries is 4 FMAs and one group reduction.
    float xacc[T];
#pragma unroll
    for (int j = 0; j < T; ++j) xacc[j] = 0.f;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const f32x4 vrow = col_to_row(vcol[b], lane);
#pragma unroll
      for (int j = 0; j < T; ++j) {
        i32x2h ah, al;
        ah[0] = zh[b][j >> 1].r[2 * (j & 1)];
        ah[1] = zh[b][j >> 1].r[2 * (j & 1) + 1];
        al[0] = zl[b][j >> 1].r[2 * (j & 1)];
        al[1] = zl[b][j >> 1].r[2 * (j & 1) + 1];
        f32x4 t = mfma16h(ah, ident, f32x4{0.f, 0.f, 0.f, 0.f});
        t = mfma16h(al, ident, t);
        xacc[j] = fmaf(t[0], vrow[0], xacc[j]);
        xacc[j] = fmaf(t[1], vrow[1], xacc[j]);
        xacc[j] = fmaf(t[2], vrow[2], xacc[j]);
        xacc[j] = fmaf(t[3], vrow[3], xacc[j]);
      }
    }
    const bool bad = !(minpiv > 0.5f);  // S >= I: only a non-finite input gets here
    if (bad && lane == 0) atomicMin(p.bad_row, (unsigned long long)cur.id);
    if (!bad && p.refine_flag && p.refine_limit > 0.f && smax > p.refine_limit * minpiv && lane == 0) {
      p.refine_flag[cur.id] = 1;
      *p.any_marked = 1;
    }
    float* o = p.out + (int64_t)cur.id * p.k;
    if constexpr (T >= 3 && T * TN < 32) {  // (T = 8, TN = 4 is at the register limit: this form spills there, 8.2 -> 8.9 ns per row)
      // The sum over the four lane groups by halving: after the exchange with lane ^ 32 a lane keeps the blocks of its
      // half of the wave, after lane ^ 16 those of its group -- group g ends up with blocks NB g .. NB g + NB - 1
      // complete and the row leaves as NB full-wave stores instead of T quarter-wave ones (T = 8: 6 ds_bpermute
      // instead of 16, 2 predicated stores instead of 8).
      constexpr int P = T > 4 ? 8 : 4, NB = P / 4;
      // (lane-derived addresses and masks of this block are recomputed per row from an opaque copy of the lane id:
      // hoisted out of the row loop they stay live across the gather and the TN = 4 kernel at T = 8 spills)
      int l2 = lane;
      asm volatile("" : "+v"(l2));
      const bool hi = l2 & 32, odd = l2 & 16;
      const int g2 = l2 >> 4, c2 = l2 & 15;
      float u[P / 2], v[NB];
#pragma unroll
      for (int i = 0; i < P / 2; ++i) {
        const float lo_blk = xacc[i], hi_blk = i + P / 2 < T ? xacc[i + P / 2] : 0.f;
        u[i] = (hi ? hi_blk : lo_blk) + bperm((l2 ^ 32) << 2, hi ? lo_blk : hi_blk);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) v[i] = (odd ? u[i + NB] : u[i]) + bperm((l2 ^ 16) << 2, odd ? u[i] : u[i + NB]);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int feat = 16 * (NB * g2 + i) + c2;
        if (feat < p.k) {
          const float x = v[i] * sD[(n - 1) * KP + feat] * inv_sc;
          o[feat] = bad ? 0.f : x;
          xmax = fmaxf(xmax, bad ? 0.f : fabsf(x));
        }
      }
    } else {
      const float* dc = sD + (n - 1) * KP + c;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const float x = reduce_groups(xacc[j], lane) * dc[16 * j] * inv_sc;
        if (lane < 16 && 16 * j + lane < p.k) {
          o[16 * j + lane] = bad ? 0.f : x;
          xmax = fmaxf(xmax, bad ? 0.f : fabsf(x));
        }
      }
    }
    if (nxt.len <= 0) break;
    cur = nxt;
    nxt = nx2;
    it += n_waves;
  }
  if (p.xbound) {
    for (int off = 32; off > 0; off >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, off));
    // same-address atomics serialise in L2 (~10 ns each, 1e5 waves): only waves that raise the bound issue one
    if (lane == 0 && xmax > __uint_as_float(__builtin_nontemporal_load(p.xbound))) atomicMax(p.xbound, __float_as_uint(xmax));
  }
}

}  // namespace mals
