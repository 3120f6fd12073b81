// als_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the ALS hot path.
//
// K1  gramian_partial_kernel / gramian_finalize_kernel : G = M^T M in fp64 on
//     v_mfma_f64_16x16x4_f64 (replaces MatrixUtils.transposeTimesSelf, MU:219-239).
// K2  gather + per-row weighted Gramian + RHS, one 64-lane wave per row (replaces the inner loop of
//     AlternatingLeastSquares.Worker.call, ALS:447-492): split-precision on v_mfma_f32_16x16x32_f16
//     (gather_row_h, the default above k = 16) or fp32 on v_mfma_f32_16x16x4_f32 (gather_row).
// K3  blocked Cholesky + triangular solves on the accumulator tiles, in registers, fused behind K2
//     (replaces MatrixUtils.getSolver(Wu).solveDToF, ALS:494 -> CMLSS:37-55, CMS:37-44).
// K4  the rows fp32 cannot solve as the reference's fp64 does (marked by K3's conditioning estimate):
//     als_refine_kernel (CG on the exact system, K3's factor as preconditioner), als_exact_kernel (the reference's
//     arithmetic in fp64, roundings included), gramian_ref_kernel (MU:232's fp32-rounded products, on demand).
//     K1 also has a split-f16 variant for large matrices (gramian_split_kernel); pad_rows_kernel makes the
//     64-byte-aligned gather table when k % 16 != 0.
//
// Fragment layouts (cdna_hip_programming.md section 3):
//   v_mfma_f32_16x16x4_f32  : lane l supplies A[i=l&15][kk=l>>4], B[kk=l>>4][j=l&15];
//                             C/D "acc layout": lane l reg r = D[row=4*(l>>4)+r][col=l&15].
//   v_mfma_f64_16x16x4_f64  : same A/B; C/D lane l reg r = D[row=(l>>4)+4*r][col=l&15].
// With g = lane>>4 (lane group) and c = lane&15: a feature vector of k floats is cut into
// T = ceil(k/16) blocks of 16; W (k x k, symmetric) is held as its T(T+1)/2 upper 16x16 tiles in
// acc layout, 4 VGPRs per tile per lane.  The lane-level choreography below is mirrored
// one-to-one by tests/wave_emulation.py, which is checked against the oracle on CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace mals {

#ifndef MALS_WAVES
#define MALS_WAVES(T, MODE) ((T) <= 4 ? 4 : ((T) == 5 ? 3 : 2))
#endif
#ifndef MALS_WAVES_H
#define MALS_WAVES_H(T, MODE) ((T) <= 2 ? 5 : ((T) <= 4 ? 3 : 2))
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int tri(int T) { return T * (T + 1) / 2; }
// index of upper tile (i <= j), row-major over the upper triangle
__host__ __device__ constexpr int tidx(int T, int i, int j) { return i * T - (i * (i - 1)) / 2 + (j - i); }

constexpr int YMAX_SLOTS = 64;  // addresses the Gramian kernels spread their max |element| atomics over (power of two)

struct WorkItem {   // one row (list A) or one segment of a long row (list B); 16 bytes, s_load_dwordx4
  int64_t begin;    // absolute offset into col/val
  int32_t len;      // entries
  int32_t id;       // list A: local row; list B: scratch slot
};
struct RowC {       // a long row to be finished from its segment partials: slots first_slot + i * stride, i < nseg
  int64_t first_slot;
  int32_t row;
  int32_t nseg;
  int32_t stride;   // 1, or the group length once als_prereduce_kernel has summed every group of slots into its first
  int32_t pad_;
};
// A wave sums a row's partial slots one after the other (a dependent chain of ~0.7 us each): the most popular item of
// C4 (7M entries, 1 700 segments) kept the finish kernel running for 1.15 ms with everything else long done.  Rows with
// more than FINISH_GROUP segments are first reduced group-wise, one wave per group of FINISH_GROUP consecutive slots,
// in place (als_prereduce_kernel); the finish kernel then walks the group leaders.
constexpr int FINISH_GROUP = 32;

struct SolveParams {
  const int64_t* row_ptr;   // local CSR
  const int32_t* col;
  const float* val;
  const float* M;           // gather table: the opposing factor replica (row-major, stride ldm = k) when k % 16 == 0,
                            // else its zero-padded copy (stride ldm = 16 T, pad_rows_kernel): every row starts on a
                            // 64-byte boundary and the last feature block loads like the others
  const float* Gf;          // fp32 image of G in acc layout: [upper tile][lane][reg]
  const float* Gperm;       // the same image in the feature order of the LDS-staged k = 128 kernels (lds_kernels.h), or null
  float* out;               // this side's factor replica + row_offset*k
  const WorkItem* items;    // list A or B (whichever this launch handles), sorted by length (desc)
  const RowC* rowsC;        // list C
  float* scratch;           // segment partials
  unsigned long long* bad_row; // first (smallest) local row with a non-PD system
  unsigned long long* suspect; // (pivot bits << 32 | local row) of the smallest pivot within 1024x of the threshold:
                               // mals_check puts that row to the reference's own singularity test (pivoted QR)
  int* any_marked;          // set to 1 with the first mark of a half-iteration (gramian_ref_kernel waits for it)
  uint8_t* refine_flag;     // per local row: set to 1 by the solving kernel when (largest entry of W) / (smallest pivot)
  float refine_limit;       // exceeds refine_limit -- the row is then re-solved with fp64 residuals (als_refine_kernel)
  float gramian_weight;     // what the largest entry of all of W counts for in that estimate: 1/4 normally (store_row),
                            // 1 under reconstructR, where W IS the Gramian plus the ridge
  int64_t n_work;           // waves of work in the list this launch handles
  int32_t k;
  int32_t ldm;              // row stride of M in floats
  int32_t flags;            // bit0 reconstructR, bit1 lossIgnoresUnspecified, bit3 run only if zscale[2] == 0
  float alpha;
  float lambda_alpha;       // lambda*alpha
  float sing_threshold;
  const float* zscale;      // split-precision gather only: {S, 1/S^2, range flag}, written by gather_scale_kernel
  unsigned long long* trace;  // profiling only (MALS_DEBUG_TRACE): per-phase s_memtime stamps
  int trace_start;            // first traced row of a wave's list (MALS_DEBUG_TRACE=<n>)
};

// ------------------------------------------------------------------------------------------------
// cross-lane helpers
__device__ __forceinline__ float bperm(int byte_idx, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_idx, __float_as_int(v)));
}
__device__ __forceinline__ float readlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// 4-way select on the two low bits of q, written as independent bit tests so that it lowers to three
// v_cndmask (a ?: chain on q==0/1/2 becomes a switch, which hipcc lowers to exec-mask branches)
__device__ __forceinline__ float select4(int q, float t0, float t1, float t2, float t3) {
  const bool b0 = q & 1, b1 = q & 2;
  const float lo = b0 ? t1 : t0, hi = b0 ? t3 : t2;
  return b1 ? hi : lo;
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | lo;
}
template <int N>
__device__ __forceinline__ float row_ror(float v) {  // DPP rotate within each 16-lane row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
template <int LANE>
__device__ __forceinline__ float row_bcast(float v) {  // DPP row_newbcast: lane LANE of each 16-lane row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + LANE, 0xf, 0xf, false));
}
__device__ __forceinline__ float reduce_row16(float x) {  // all 16 lanes of a row get the row sum
  x += row_ror<8>(x);
  x += row_ror<4>(x);
  x += row_ror<2>(x);
  x += row_ror<1>(x);
  return x;
}
#ifndef MALS_REDUCE_GROUPS_BPERM
#define MALS_REDUCE_GROUPS_BPERM 0
#endif
// sum over the 4 lane groups (same c), every lane gets the total.  gfx950's v_permlane32_swap / v_permlane16_swap move
// whole 32- / 16-lane halves between two registers on the VALU: with both operands = x the pair comes back as
// ([x0 x1 x0 x1], [x2 x3 x2 x3]) resp. ([s0 s0 s2 s2], [s1 s1 s3 s3]) (rows of 16 lanes), so two swap + add stages
// replace two dependent ds_bpermute round trips (the block solves chain 16 of these per row).
__device__ __forceinline__ float reduce_groups(float x, int lane) {
#if MALS_REDUCE_GROUPS_BPERM
  x += bperm((lane ^ 16) << 2, x);
  x += bperm((lane ^ 32) << 2, x);
  return x;
#else
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned xi = __float_as_uint(x);
  const u32x2 a = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned si = __float_as_uint(s);
  const u32x2 b = __builtin_amdgcn_permlane16_swap(si, si, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
#endif
}
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// D = P^T Q accumulated onto C, for acc-layout tiles P, Q (contraction over the tile-row index)
__device__ __forceinline__ f32x4 tile_ptq(const f32x4& P, const f32x4& Q, f32x4 C) {
  C = mfma4(P[0], Q[0], C);
  C = mfma4(P[1], Q[1], C);
  C = mfma4(P[2], Q[2], C);
  C = mfma4(P[3], Q[3], C);
  return C;
}
__device__ __forceinline__ f32x4 tile_neg_ptq(const f32x4& P, const f32x4& Q, f32x4 C) {
  C = mfma4(-P[0], Q[0], C);
  C = mfma4(-P[1], Q[1], C);
  C = mfma4(-P[2], Q[2], C);
  C = mfma4(-P[3], Q[3], C);
  return C;
}

// ------------------------------------------------------------------------------------------------
// K3a: in-register factorization of one full symmetric 16x16 tile D (acc layout).  Returns
// Uinv = U^{-1} (acc layout) where U^T U = D.  Read through the symmetry of D, lane (g,c) owns
// ROW c of the tile, columns 4g..4g+3 (d[r] = D[c][4g+r]); it keeps the same part of row c of the
// inverse factor next to it (e[r]).  The elimination runs LDL^T-style on [D | I] with unscaled rows:
// step m subtracts l_cm = D[c][m] / D[m][m] times row m from every row c > m.  Row m at this lane's
// columns sits in lane m of the same 16-lane group (a DPP row_newbcast operand of the FMA); only the
// multiplier's numerator D[c][m] lives in another group (lane (m>>2, c)): ONE ds_bpermute per step.
// At the end row c of [.. | L~^-1] is scaled by 1/sqrt(pivot c), which every lane holds for its own
// row, and e[r] = L^-1[c][4g+r] = Uinv[4g+r][c] is already the acc layout of Uinv: no transposition.
// (ds_bpermute is the scarce resource: tools/ubench/bperm_rate.hip measures one per 2.5 ns per CU,
// shared by all waves, and it heads the dependency chain of every step.)
// minpiv watches the smallest pivot (a pivot <= singularity threshold flags a non-PD system), per lane: see the end of factor_diag.
// Pivot order: step t eliminates index p = 4*(t&3) + (t>>2), i.e. register t>>2 of lane group t&3 --
// register by register instead of 0..15.  Any symmetric pivot order factors an SPD tile (U is then a
// row permutation of a triangle, which neither the TRSM, the SYRK nor the solves care about: they only
// use U_kk^-1), and with this one whole registers drop out of the work: register r of the D half is
// finished after step 4r+3, register r of the inverse half is still zero before step 4r.  76 DPP FMAs
// per tile instead of 128 (a DPP instruction costs two issue slots).
__device__ __forceinline__ float spread_to_col(float v, int lane);
template <int T_>
__device__ __forceinline__ void diag_step(f32x4& D, f32x4& E, int lane, int pos, float& minpiv, float& piv, float& num) {
  constexpr int gm = T_ & 3, rm = T_ >> 2, P = 4 * gm + rm;
  constexpr int gn = (T_ + 1) & 3, rn = ((T_ + 1) >> 2) & 3, PN = 4 * gn + rn;  // the next step's pivot
  const int c = lane & 15;
  const float rinv = __builtin_amdgcn_rcpf(piv);
  float nl = -(num * rinv);        // num = D[c][p] (from lane (gm, c)), piv = D[p][p]: fetched a step ahead
  nl = pos > T_ ? nl : 0.f;        // finished rows (and row p itself) stay put: D[p][p] keeps the pivot
  // d += row_newbcast(d) * nl as ONE instruction each (hipcc emits v_mov_b32_dpp + v_fma instead).  A
  // DPP read needs 2 wait states after a VALU write of the same register, which the compiler does not
  // track into asm: step 0 (whose operands may just have been copied) opens with an s_nop; later the
  // registers were last written by the previous step's block, 7+ instructions earlier
  // (tools/check_dpp_hazards.py checks that on the generated ISA).  Blocks, not one statement per FMA
  // (the assembler's .if drops the finished / still-zero registers): a statement per FMA would let the
  // compiler put a register copy right in front of a DPP read.  The register of the next pivot goes first, so that the next step's pivot and
  // multipliers (the ds_bpermute heads the step's dependency chain) are fetched under the other FMAs.
#define MALS_FMAC_BCAST(n, cond) ".if " cond "\n\tv_fmac_f32_dpp %" #n ", %" #n ", %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t.endif\n\t"
  if constexpr (T_ < 15) {
    asm volatile(".if %10 == 0\n\ts_nop 1\n\t.endif\n\t"
                 MALS_FMAC_BCAST(0, "%11 == 0") MALS_FMAC_BCAST(1, "%11 == 1") MALS_FMAC_BCAST(2, "%11 == 2") MALS_FMAC_BCAST(3, "%11 == 3")
                 : "+v"(D[0]), "+v"(D[1]), "+v"(D[2]), "+v"(D[3]), "+v"(E[0]), "+v"(E[1]), "+v"(E[2]), "+v"(E[3])
                 : "v"(nl), "n"(P), "n"(T_), "n"(rn));
    __builtin_amdgcn_sched_barrier(0);
    num = bperm(((16 * gn) | c) << 2, D[rn]);
    piv = readlane(D[rn], 16 * gn + PN);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile(MALS_FMAC_BCAST(0, "(3 > %10) && (%11 != 0)") MALS_FMAC_BCAST(1, "(7 > %10) && (%11 != 1)")
                 MALS_FMAC_BCAST(2, "(11 > %10) && (%11 != 2)") MALS_FMAC_BCAST(3, "(15 > %10) && (%11 != 3)")
                 MALS_FMAC_BCAST(4, "0 <= %10") MALS_FMAC_BCAST(5, "4 <= %10") MALS_FMAC_BCAST(6, "8 <= %10") MALS_FMAC_BCAST(7, "12 <= %10")
                 : "+v"(D[0]), "+v"(D[1]), "+v"(D[2]), "+v"(D[3]), "+v"(E[0]), "+v"(E[1]), "+v"(E[2]), "+v"(E[3])
                 : "v"(nl), "n"(P), "n"(T_), "n"(rn));
  } else {
    asm volatile(MALS_FMAC_BCAST(4, "0 <= %10") MALS_FMAC_BCAST(5, "4 <= %10") MALS_FMAC_BCAST(6, "8 <= %10") MALS_FMAC_BCAST(7, "12 <= %10")
                 : "+v"(D[0]), "+v"(D[1]), "+v"(D[2]), "+v"(D[3]), "+v"(E[0]), "+v"(E[1]), "+v"(E[2]), "+v"(E[3])
                 : "v"(nl), "n"(P), "n"(T_), "n"(rn));
  }
#undef MALS_FMAC_BCAST
}

__device__ __forceinline__ f32x4 factor_diag(f32x4 D, int lane, float& minpiv) {
  const int g = lane >> 4, c = lane & 15;
  const int pos = 4 * (c & 3) + (c >> 2);  // the step at which row c is the pivot row
  f32x4 E;
#pragma unroll
  for (int r = 0; r < 4; ++r) E[r] = (4 * g + r == c) ? 1.f : 0.f;
  float num = bperm(c << 2, D[0]);   // step 0: pivot index 0 = register 0 of group 0
  float piv = readlane(D[0], 0);
  diag_step<0>(D, E, lane, pos, minpiv, piv, num);
  diag_step<1>(D, E, lane, pos, minpiv, piv, num);
  diag_step<2>(D, E, lane, pos, minpiv, piv, num);
  diag_step<3>(D, E, lane, pos, minpiv, piv, num);
  diag_step<4>(D, E, lane, pos, minpiv, piv, num);
  diag_step<5>(D, E, lane, pos, minpiv, piv, num);
  diag_step<6>(D, E, lane, pos, minpiv, piv, num);
  diag_step<7>(D, E, lane, pos, minpiv, piv, num);
  diag_step<8>(D, E, lane, pos, minpiv, piv, num);
  diag_step<9>(D, E, lane, pos, minpiv, piv, num);
  diag_step<10>(D, E, lane, pos, minpiv, piv, num);
  diag_step<11>(D, E, lane, pos, minpiv, piv, num);
  diag_step<12>(D, E, lane, pos, minpiv, piv, num);
  diag_step<13>(D, E, lane, pos, minpiv, piv, num);
  diag_step<14>(D, E, lane, pos, minpiv, piv, num);
  diag_step<15>(D, E, lane, pos, minpiv, piv, num);
  // row c is scaled by 1/sqrt(its pivot), which lane (c>>2, c) still holds as D[c][c]
  // (its pivot also enters the smallest-pivot watch here, per LANE -- lane (g,c) watches pivot c of every tile; cholesky_tiles
  // takes the minimum over the 16 lanes of a row once per system.  Until round 4 every step folded its pivot into a uniform
  // minimum: a v_min3 and a register copy of the scalar operands per two steps, 16 vector instructions per tile.)
  const float pvc = spread_to_col(select4(c & 3, D[0], D[1], D[2], D[3]), lane);
  minpiv = fminf(minpiv, pvc == pvc ? pvc : 0.f);   // (a pivot <= 0 poisons the rows after it: a NaN on the diagonal is a failed pivot, and v_min would drop it)
  const float s = __builtin_amdgcn_rsqf(pvc);
#pragma unroll
  for (int r = 0; r < 4; ++r) E[r] *= s;
  return E;
}

// ---- split-precision tile products for the rank-16 updates of the factorization (SYRK) ------------
// Same idea as the split-precision gather (gather_row_h): U_ki^T U_kj with both operands cut into two
// f16 halves (22 significand bits, exact products, fp32 accumulate) costs 3 x 17 cycles on the f16
// matrix pipe instead of 4 x 33 on the fp32 one.  The operands are entries of U, bounded by
// sqrt(max_i W_ii): the row's system is scaled by a power of two s^2 first (row_scale) so that they
// stay below 2^13 -- f16 cannot overflow, and entries down to 2^-17 of the largest keep 22 bits.
typedef _Float16 f16x4h __attribute__((ext_vector_type(4)));
typedef int i32x2h __attribute__((ext_vector_type(2)));
struct TileH {
  i32x2h h, l;  // contraction index 4g + r: reg 0 = (r0 | r1 << 16), reg 1 = (r2 | r3 << 16)
};
// two floats -> packed f16, round to nearest even (gfx950's v_cvt_pk_f16_f32; same issue cost as v_cvt_pkrtz_f16_f32,
// tools/ubench/op_rate.hip).  BOTH halves of every split are rounded.  Until round 4 they were truncated (v_cvt_pkrtz): a truncated
// residual makes hi + lo fall short of the value by up to 2^-22 of it, always towards zero -- a bias every product of a
// half-iteration shares, which ten chained iterations showed as a linear drift away from the oracle's chain (+1.6e-7 per
// iteration, tests/test_gpu_multi_iteration.py).  Rounded low half: no sign, no drift; rounded high half as well: the residual is at
// most half an f16 ulp, one more bit for the low half -- one half-iteration 4.2e-7 -> 2.7e-7 from the oracle, ten 2.3e-6 -> 1.1e-6.
__device__ __forceinline__ int pk_rn16(float a, float b) {
  typedef float f32x2r __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2r __attribute__((ext_vector_type(2)));
  const f32x2r v = {a, b};
  return __builtin_bit_cast(int, __builtin_convertvector(v, f16x2r));
}
// z - (float)h.lo / z - (float)h.hi for a packed f16 pair h: one v_fma_mix_f32 each (f16 source 0, fp32 constant and addend)
__device__ __forceinline__ float residual_lo(int h, float z) {
  float o;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(o) : "v"(h), "v"(z));
  return o;
}
__device__ __forceinline__ float residual_hi(int h, float z) {
  float o;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(o) : "v"(h), "v"(z));
  return o;
}

__device__ __forceinline__ TileH split_tile(const f32x4& t) {
  // the value rounded to f16 (pk_rn16), then what is left: one v_fma_mix_f32 per value on the packed halves (hipcc's own
  // rendering of t - (float)half was v_cvt_f32_f16 + v_sub_f32: two instructions per value, 224 per k = 128 row)
  const int h01 = pk_rn16(t[0], t[1]), h23 = pk_rn16(t[2], t[3]);
  const int l01 = pk_rn16(residual_lo(h01, t[0]), residual_hi(h01, t[1]));
  const int l23 = pk_rn16(residual_lo(h23, t[2]), residual_hi(h23, t[3]));
  TileH o;
  o.h[0] = h01;
  o.h[1] = h23;
  o.l[0] = l01;
  o.l[1] = l23;
  return o;
}
__device__ __forceinline__ TileH negate_tile(const TileH& t) {
  TileH o;
  o.h = t.h ^ (int)0x80008000;
  o.l = t.l ^ (int)0x80008000;
  return o;
}
__device__ __forceinline__ f32x4 mfma16h(const i32x2h& a, const i32x2h& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4h, a), __builtin_bit_cast(f16x4h, b), c, 0, 0, 0);
}
// C += P^T Q for acc-layout tiles given as split operands (the lo x lo term, < 2^-22 relative, is dropped)
__device__ __forceinline__ f32x4 tile_ptq_h(const TileH& P, const TileH& Q, f32x4 C) {
  C = mfma16h(P.h, Q.h, C);
  C = mfma16h(P.h, Q.l, C);
  C = mfma16h(P.l, Q.h, C);
  return C;
}

// Two block rows at once: C += P0^T Q0 + P1^T Q1 as three v_mfma_f32_16x16x32_f16 -- the contraction runs over the 32 rows of
// both tiles (each lane supplies its four rows of tile 0 and its four rows of tile 1 to A and to B alike; the order inside the
// contraction does not matter as long as both operands use the same one).  Same 16 cycles as the 16-row instruction.
typedef int i32x4h __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));
struct TileH2 {
  i32x4h h, l;
};
__device__ __forceinline__ TileH2 pair_tiles(const TileH& a, const TileH& b) {
  TileH2 o;
  o.h = i32x4h{a.h[0], a.h[1], b.h[0], b.h[1]};
  o.l = i32x4h{a.l[0], a.l[1], b.l[0], b.l[1]};
  return o;
}
__device__ __forceinline__ f32x4 mfma32h(const i32x4h& a, const i32x4h& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8h, a), __builtin_bit_cast(f16x8h, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 tile_ptq_h2(const TileH2& P, const TileH2& Q, f32x4 C) {
  C = mfma32h(P.h, Q.h, C);
  C = mfma32h(P.h, Q.l, C);
  C = mfma32h(P.l, Q.h, C);
  return C;
}

// K3b: blocked right-looking Cholesky W = U^T U on the upper tiles.  On return the off-diagonal
// tiles hold U_ij and the diagonal tiles hold U_ii^{-1}.  TRSM and SYRK run on the matrix cores.
// SPLIT: the SYRK products on the f16 matrix pipe (the caller has scaled the system, row_scale).
#ifndef MALS_SYRK_RESPLIT_MINT
#define MALS_SYRK_RESPLIT_MINT 8
#endif
// MALS_SYRK_PAIRS_MINT: from this T on (even T only) the rank-16 updates of TWO consecutive block rows are applied to the
// trailing tiles together, as one K = 32 product per tile (tile_ptq_h2): block row kb is factored and applied to block row
// kb + 1 alone (K = 16), block row kb + 1 is factored, then every tile below gets both rows' updates in three
// v_mfma_f32_16x16x32_f16 instead of six v_mfma_f32_16x16x16_f16 -- T = 8: 150 matrix instructions instead of 252 for the
// updates of a row, the same number of split_tile.  PAIRS is set by the LDS-staged k = 128 kernel only: measured there -1.3 % on
// the user-half rows kernel of c5rank (the matrix pipe is a third busy: an f16 matrix instruction less is worth ~3.5 issue cycles,
// not its 16 pipe cycles); in the register-staged T = 8 kernels the two operand pairs cost 28 more spilled dwords.
#ifndef MALS_SYRK_PAIRS_MINT
#define MALS_SYRK_PAIRS_MINT 8
#endif
template <int T>
__device__ __forceinline__ void cholesky_factor_row(f32x4 (&acc)[tri(T)], int kb, int lane, float& minpiv) {
  const f32x4 Uinv = factor_diag(acc[tidx(T, kb, kb)], lane, minpiv);
  acc[tidx(T, kb, kb)] = Uinv;
#pragma unroll
  for (int j = 0; j < T; ++j) {  // U_kj = Uinv^T A_kj
    if (j > kb) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      acc[tidx(T, kb, j)] = tile_ptq(Uinv, acc[tidx(T, kb, j)], zero);
    }
  }
}
// smallest pivot of the system out of factor_diag's per-lane watch: minimum over the 16 lanes of a row (every lane group holds
// the same 16 values), uniform afterwards
__device__ __forceinline__ float min_row16(float x) {
  x = fminf(x, row_ror<8>(x));
  x = fminf(x, row_ror<4>(x));
  x = fminf(x, row_ror<2>(x));
  x = fminf(x, row_ror<1>(x));
  return x;
}
template <int T>
__device__ __forceinline__ void cholesky_tiles_pairs(f32x4 (&acc)[tri(T)], int lane, float& minpiv) {
  static_assert(T % 2 == 0, "block rows are taken two at a time");
#pragma unroll
  for (int kb = 0; kb < T; kb += 2) {
    cholesky_factor_row<T>(acc, kb, lane, minpiv);
    {  // block row kb -> block row kb + 1 only
      const int i = kb + 1;
      const TileH qi = split_tile(acc[tidx(T, kb, i)]);
      const TileH np = negate_tile(qi);
      acc[tidx(T, i, i)] = tile_ptq_h(np, qi, acc[tidx(T, i, i)]);
#pragma unroll
      for (int j = 0; j < T; ++j) {
        if (j > i) {
          const TileH qj = split_tile(acc[tidx(T, kb, j)]);
          acc[tidx(T, i, j)] = tile_ptq_h(np, qj, acc[tidx(T, i, j)]);
        }
      }
    }
    cholesky_factor_row<T>(acc, kb + 1, lane, minpiv);
#pragma unroll
    for (int i = 0; i < T; ++i) {   // both block rows -> everything below them
      if (i >= kb + 2) {
        const TileH2 qi = pair_tiles(split_tile(acc[tidx(T, kb, i)]), split_tile(acc[tidx(T, kb + 1, i)]));
        TileH2 np;
        np.h = qi.h ^ (int)0x80008000;
        np.l = qi.l ^ (int)0x80008000;
        acc[tidx(T, i, i)] = tile_ptq_h2(np, qi, acc[tidx(T, i, i)]);
#pragma unroll
        for (int j = 0; j < T; ++j) {
          if (j > i) {
            const TileH2 qj = pair_tiles(split_tile(acc[tidx(T, kb, j)]), split_tile(acc[tidx(T, kb + 1, j)]));
            acc[tidx(T, i, j)] = tile_ptq_h2(np, qj, acc[tidx(T, i, j)]);
          }
        }
      }
    }
  }
}
template <int T, bool SPLIT = false, bool RESPLIT = (T >= MALS_SYRK_RESPLIT_MINT), bool PAIRS = false>
__device__ __forceinline__ void cholesky_tiles(f32x4 (&acc)[tri(T)], int lane, float& minpiv) {
  if constexpr (PAIRS && SPLIT && RESPLIT && T % 2 == 0 && T >= MALS_SYRK_PAIRS_MINT) {
    cholesky_tiles_pairs<T>(acc, lane, minpiv);
    minpiv = min_row16(minpiv);
    return;
  }
#pragma unroll
  for (int kb = 0; kb < T; ++kb) {
#ifdef MALS_DOUBLE_DIAG   // ablation by duplication (exp builds): what one more factor_diag per tile costs in place
    {
      float mp2 = minpiv;
      const f32x4 twice = factor_diag(acc[tidx(T, kb, kb)], lane, mp2);
      asm volatile("" ::"v"(twice[0]), "v"(twice[1]), "v"(twice[2]), "v"(twice[3]), "v"(mp2));
    }
#endif
    const f32x4 Uinv = factor_diag(acc[tidx(T, kb, kb)], lane, minpiv);
    acc[tidx(T, kb, kb)] = Uinv;
#pragma unroll
    for (int j = kb + 1; j < T; ++j) {  // U_kj = Uinv^T A_kj
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      acc[tidx(T, kb, j)] = tile_ptq(Uinv, acc[tidx(T, kb, j)], zero);
    }
    if constexpr (SPLIT && RESPLIT) {
      // T = 8: the split tiles of block row kb are made where they are used instead of being kept in q[T] -- 28 registers
      // next to 144 accumulators and the next row's 32 raw registers in flight.  +56 split_tile per row (~670 VALU),
      // 42 -> 15 spilled dwords, and the spill traffic was the larger cost: c5rank 168.2 -> 164.6 ms, its user-half
      // rows kernel 61.7 -> 58.8 ms, c5shard8 183.1 -> 180.0 (round 3, same box)
#pragma unroll
      for (int i = kb + 1; i < T; ++i) {
        const TileH qi = split_tile(acc[tidx(T, kb, i)]);
        const TileH np = negate_tile(qi);
        acc[tidx(T, i, i)] = tile_ptq_h(np, qi, acc[tidx(T, i, i)]);
#pragma unroll
        for (int j = i + 1; j < T; ++j) {    // A_ij -= U_ki^T U_kj
          const TileH qj = split_tile(acc[tidx(T, kb, j)]);
          acc[tidx(T, i, j)] = tile_ptq_h(np, qj, acc[tidx(T, i, j)]);
        }
      }
    } else if constexpr (SPLIT) {
      TileH q[T];
#pragma unroll
      for (int j = kb + 1; j < T; ++j) q[j] = split_tile(acc[tidx(T, kb, j)]);
#pragma unroll
      for (int i = kb + 1; i < T; ++i) {
        const TileH np = negate_tile(q[i]);
#pragma unroll
        for (int j = i; j < T; ++j)     // A_ij -= U_ki^T U_kj
          acc[tidx(T, i, j)] = tile_ptq_h(np, q[j], acc[tidx(T, i, j)]);
      }
    } else {
#pragma unroll
      for (int i = kb + 1; i < T; ++i) {
#pragma unroll
        for (int j = i; j < T; ++j)     // A_ij -= U_ki^T U_kj
          acc[tidx(T, i, j)] = tile_neg_ptq(acc[tidx(T, kb, i)], acc[tidx(T, kb, j)], acc[tidx(T, i, j)]);
      }
    }
  }
  minpiv = min_row16(minpiv);
}

// vector layout conversions: "col layout" = lane (g,c) holds v[c]; "row layout" = lanes of group g
// hold v[4g+r] in reg r.
__device__ __forceinline__ f32x4 col_to_row(float vcol, int lane) {
  const int g = lane >> 4;
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = bperm(((lane & 48) | (4 * g + r)) << 2, vcol);
  return o;
}
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {  // DPP quad_perm, CTRL = a | b<<2 | c<<4 | d<<6
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// Sum each of the four registers over the 16 lanes of a row; lane c ends up with the total of register
// c & 3.  A butterfly that halves the live data per stage: 5 DPP adds instead of the 16 of four
// reduce_row16, and the result is spread over the lanes the way the next cross-group move wants it.
__device__ __forceinline__ float reduce4_row16(const f32x4& v, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  const float a0 = (b0 ? v[1] : v[0]) + quad_perm<0xB1>(b0 ? v[0] : v[1]);  // lanes ^1: registers with (r & 1) == b0
  const float a1 = (b0 ? v[3] : v[2]) + quad_perm<0xB1>(b0 ? v[2] : v[3]);
  float t = (b1 ? a1 : a0) + quad_perm<0x4E>(b1 ? a0 : a1);                  // lanes ^2: register 2*b1 + b0
  t += row_ror<4>(t);
  t += row_ror<8>(t);
  return t;
}
// row-layout vector spread as reduce4_row16 leaves it (lane (g,c): element 4g + (c&3)) -> col layout
__device__ __forceinline__ float spread_to_col(float v, int lane) {
  const int c = lane & 15;
  return bperm(((16 * (c >> 2)) | c) << 2, v);
}

// K3c: x = W^{-1} b using the factor tiles (forward z = U^{-T} b, backward x = U^{-1} z).
template <int T>
__device__ __forceinline__ void solve_tiles(const f32x4 (&acc)[tri(T)], const float (&bcol)[T], float (&xcol)[T], int lane) {
  f32x4 zrow[T];
  float zcol[T];
#pragma unroll
  for (int kb = 0; kb < T; ++kb) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < kb; ++i) {
      const f32x4& U = acc[tidx(T, i, kb)];
      t = fmaf(U[0], zrow[i][0], t);
      t = fmaf(U[1], zrow[i][1], t);
      t = fmaf(U[2], zrow[i][2], t);
      t = fmaf(U[3], zrow[i][3], t);
    }
    float rhs = bcol[kb];
    if (kb > 0) rhs -= reduce_groups(t, lane);
    const f32x4 rr = col_to_row(rhs, lane);
    const f32x4& Ui = acc[tidx(T, kb, kb)];
    float zt = Ui[0] * rr[0];
    zt = fmaf(Ui[1], rr[1], zt);
    zt = fmaf(Ui[2], rr[2], zt);
    zt = fmaf(Ui[3], rr[3], zt);
    zcol[kb] = reduce_groups(zt, lane);
    if (kb < T - 1) zrow[kb] = col_to_row(zcol[kb], lane);
  }
  // backward: contractions over the columns = over the 16 lanes of a row (DPP); one cross-group move
  // per product brings the row-indexed result back to the col layout the next product wants
#pragma unroll
  for (int kb = T - 1; kb >= 0; --kb) {
    float rhs_col = zcol[kb];
    if (kb < T - 1) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = kb + 1; j < T; ++j) {
        const f32x4& U = acc[tidx(T, kb, j)];
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaf(U[r], xcol[j], t[r]);
      }
      rhs_col -= spread_to_col(reduce4_row16(t, lane), lane);
    }
    const f32x4& Ui = acc[tidx(T, kb, kb)];
    f32x4 xr;
#pragma unroll
    for (int r = 0; r < 4; ++r) xr[r] = Ui[r] * rhs_col;
    xcol[kb] = spread_to_col(reduce4_row16(xr, lane), lane);
  }
}

// ------------------------------------------------------------------------------------------------
// K2: gather phase.  One wave walks the entries [0,len) of a row (or row segment) four at a time:
// lane (g,c) owns entry 4*step+g and feature lanes c of every 16-block; per step it issues T
// 64-byte-coalesced dword gathers of the opposing factor row straight into MFMA operand registers
// (no LDS hop: a gathered row is consumed once, by this wave only), then T(T+1)/2 MFMAs
// acc_ij += (w*y_i) y_j^T and the RHS update.
//   * (col,val) are read 64 entries at a time, one entry per lane, fully coalesced; the weights
//     w/cb are computed once per entry and handed to the lane group that needs them with
//     ds_bpermute one step ahead of their use.
//   * factor rows are gathered D-1 steps ahead into a ring of D register slots.
//   * the waves are persistent: wave w owns work items w, w+W, w+2W, ... of a list sorted by length
//     (longest first), and while it factors/solves row i it already has the first chunk and the
//     first D-1 row gathers of row i+1 in flight, so the next gather starts with data on chip.
struct Chunk {
  int col;   // column index of this lane's entry (clamped inside the row)
  float w;   // Gramian weight: (c-1) = alpha*|r|   [+1 if lossIgnoresUnspecified; 0 if reconstructR]
  float cb;  // RHS weight:     c if r>0 else 0      [r if reconstructR]
};

__device__ __forceinline__ int bperm_i(int byte_idx, int v) { return __builtin_amdgcn_ds_bpermute(byte_idx, v); }

// ds_bpermute with the constant part of the lane address in the instruction's offset field.  hipcc
// never folds an add into that field, so every distinct (base + constant) costs a live VGPR -- 16 of
// them in the split-precision gather loop.  Written as asm the constants are free; the block ends
// with its own lgkmcnt(0) because the compiler's waitcnt pass does not see inside it.
template <int O0, int O1>
__device__ __forceinline__ void bperm2x2(int addr, float a, float b, float& a0, float& a1, float& b0, float& b1) {
  asm volatile(
      "ds_bpermute_b32 %0, %4, %5 offset:%7\n\t"
      "ds_bpermute_b32 %1, %4, %5 offset:%8\n\t"
      "ds_bpermute_b32 %2, %4, %6 offset:%7\n\t"
      "ds_bpermute_b32 %3, %4, %6 offset:%8\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1)
      : "v"(addr), "v"(a), "v"(b), "n"(O0), "n"(O1));
}
// The same in two halves: the two ds_bpermute now, the wait where the values are needed.  The wait statement names the
// values as operands, so nothing that reads them can be scheduled ahead of it; that the register allocator puts no copy or
// spill of them in between either is checked on the generated ISA (tools/check_lds_windows.py, tests/test_dpp_hazards.py).
template <int O0, int O1>
__device__ __forceinline__ void bperm2_i_start(int addr, int v, int& r0, int& r1) {
  asm volatile(
      "ds_bpermute_b32 %0, %2, %3 offset:%4\n\t"
      "ds_bpermute_b32 %1, %2, %3 offset:%5"
      : "=&v"(r0), "=&v"(r1)
      : "v"(addr), "v"(v), "n"(O0), "n"(O1));
}
__device__ __forceinline__ void bperm2_i_land(int& r0, int& r1) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1));
}
template <int O0, int O1>
__device__ __forceinline__ void bperm2_i(int addr, int v, int& r0, int& r1) {
  asm volatile(
      "ds_bpermute_b32 %0, %2, %3 offset:%4\n\t"
      "ds_bpermute_b32 %1, %2, %3 offset:%5\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0), "=&v"(r1)
      : "v"(addr), "v"(v), "n"(O0), "n"(O1));
}

// Entry base+lane of a row with len > 0.  Issued in two halves so that no arithmetic waits on the
// loads right after they are issued: chunk_issue only loads (col, raw value), chunk_weights turns
// the raw value into the two weights when the chunk is about to be used.
// ROWMAJOR (the split-precision kernels): lane 16 g + m holds entry base + 4 m + g instead of base + lane -- the entries
// lane group g consumes (4 e + g, e = 0..15) then sit in ITS OWN 16 lanes, and handing entry e's weights to the group
// is a DPP row broadcast of lane e (v_mov_b32_dpp row_newbcast, 4.4 issue cycles, no LDS) instead of a ds_bpermute
// (8 cycles on the SIMD + 5 on the CU's shared crossbar + a wait on its latency).  The 64 lanes still read 64
// consecutive entries.
template <bool ROWMAJOR = false>
__device__ __forceinline__ Chunk chunk_issue(const SolveParams& p, int64_t begin, int len, int base, int lane) {
  const int n = base + (ROWMAJOR ? (((lane & 15) << 2) | (lane >> 4)) : lane);
  const int nn = n < len ? n : len - 1;
  Chunk e;
  // the entry stream is read exactly once: non-temporal, so that it does not push factor rows
  // out of L2 / Infinity Cache
  e.col = __builtin_nontemporal_load(p.col + begin + nn);
  e.w = __builtin_nontemporal_load(p.val + begin + nn);  // raw r_ui until chunk_weights
  e.cb = n < len ? 1.f : 0.f;        // validity
  return e;
}
__device__ __forceinline__ void chunk_weights(const SolveParams& p, Chunk& e) {
  const float r = e.w, ok = e.cb;
  const float base_w = (p.flags & 2) ? 1.f : 0.f;
  float w, cb;
  if (p.flags & 1) {  // ALS:466-469
    w = base_w;
    cb = r;
  } else {            // ALS:471-482
    const float ar = p.alpha * fabsf(r);
    w = base_w + ar;
    cb = r > 0.f ? 1.f + ar : 0.f;
  }
  e.w = ok * w;
  e.cb = ok * cb;
}

// Gather one factor row: T dword loads, 16 lanes x 4 B = one 64-byte segment per lane group each.
// No arithmetic may depend on the loaded values here (the loads must stay in flight).  The table is
// zero-padded to 16 T floats per row when k % 16 != 0 (SolveParams::M), so the last block needs no
// predication and no row straddles more 128-byte lines than its length asks for (measured at k = 30 on the
// unpadded replica, 120-byte rows: 1.6x the algorithmic bytes from HBM, all of it as 128-byte requests).
template <int T>
__device__ __forceinline__ void load_rows(const float* __restrict__ M, int ldm, int col, int c, float (&y)[T]) {
  const float* p = M + ((uint64_t)(uint32_t)col * (uint32_t)ldm + (uint32_t)c);  // one v_mad_u64_u32
#pragma unroll
  for (int v = 0; v < T; ++v) y[v] = p[16 * v];
}

template <int T>
__device__ __forceinline__ void gram_step(const float (&y)[T], float w, float cb, f32x4 (&acc)[tri(T)], float (&bpart)[T]) {
  float a[T];
#pragma unroll
  for (int v = 0; v < T; ++v) a[v] = w * y[v];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = i; j < T; ++j) acc[tidx(T, i, j)] = mfma4(a[i], y[j], acc[tidx(T, i, j)]);
#pragma unroll
  for (int v = 0; v < T; ++v) bpart[v] = fmaf(cb, y[v], bpart[v]);
}

// per-wave gather pipeline state (all in registers)
template <int T, int D>
struct Pipe {
  Chunk ch;       // current 64-entry chunk
  float y[D][T];  // ring of gathered rows
  float wcur, cbcur;  // weights of the next step to execute
  int colpf;          // column of the next step to gather
};

// Start a row (len > 0): ch = chunk 0 already loaded; issue the gathers of steps 0..D-2 and fetch the
// weights of step 0 and the column of step D-1.  Steps past the end of a short row gather a valid
// (clamped) row that is never used.
template <int T, int D, bool FULL>
__device__ __forceinline__ void prime_row(const SolveParams& p, int lane, Pipe<T, D>& pp) {
  const int c = lane & 15, gb = (lane >> 4) << 2;
#pragma unroll
  for (int i = 0; i < D - 1; ++i) load_rows<T>(p.M, p.ldm, bperm_i(gb + 16 * i, pp.ch.col), c, pp.y[i]);
  pp.wcur = bperm(gb, pp.ch.w);
  pp.cbcur = bperm(gb, pp.ch.cb);
  pp.colpf = bperm_i(gb + 16 * (D - 1), pp.ch.col);
}

template <int T, int D, bool FULL>
__device__ __forceinline__ void gather_row(const SolveParams& p, int64_t begin, int len, int lane, Pipe<T, D>& pp,
                                           f32x4 (&acc)[tri(T)], float (&bpart)[T]) {
  static_assert(16 % D == 0, "ring depth must divide the chunk length in steps");
  const int c = lane & 15, gb = (lane >> 4) << 2;
  const int nsteps = (len + 3) >> 2;
  for (int q = 0; 16 * q < nsteps; ++q) {
    // next chunk (clamped inside the row, so always safe to issue); lands during this chunk
    Chunk chn = chunk_issue(p, begin, len, 64 * (q + 1), lane);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int s = 16 * q + j;
      if (s < nsteps) {
        // (1) gather the rows of step s+D-1 into the slot step s-1 has just released
        load_rows<T>(p.M, p.ldm, pp.colpf, c, pp.y[(j + D - 1) % D]);
        __builtin_amdgcn_sched_barrier(0);
        if (j == 14) chunk_weights(p, chn);  // first use of the next chunk's weights is step 15
        // (2) cross-lane fetches for the next step's weights and the next gather's column
        const int jn = j + 1, jt = j + D;
        const float wn = bperm(gb + 16 * (jn & 15), jn < 16 ? pp.ch.w : chn.w);
        const float cbn = bperm(gb + 16 * (jn & 15), jn < 16 ? pp.ch.cb : chn.cb);
        const int coln = bperm_i(gb + 16 * (jt & 15), jt < 16 ? pp.ch.col : chn.col);
        // (3) the matrix-core work of step s
        gram_step<T>(pp.y[j % D], pp.wcur, pp.cbcur, acc, bpart);
        pp.wcur = wn;
        pp.cbcur = cbn;
        pp.colpf = coln;
        // keep the hand-built pipeline: without this fence the scheduler hoists the gathers of all 16
        // unrolled steps to the top of the chunk and the kernel needs > 200 VGPRs
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    pp.ch = chn;
  }
}

// ------------------------------------------------------------------------------------------------
// K2, split-precision variant (T = 3, 4).  Measured on gfx950 (tools/ubench/coexec2.hip): an
// fp32-input MFMA occupies its SIMD for 33 cycles per 16x16x4 and nothing else issues meanwhile, so
// at k = 64 the fp32 Gramian above costs 10 x 33 cycles per 4 entries and the gather is matrix-issue
// bound, not HBM bound.  The f16 matrix pipe does a 16x16x32 product in 17 cycles, 16x the
// contraction depth per cycle.  Here the per-row Gramian  sum_n w_n y_n y_n^T  is therefore computed
// as  sum_n z_n z_n^T  with z_n = sqrt(w_n) S y_n  split into two f16 numbers z = zh + zl
// (zh = z rounded to f16, zl = the exact residual z - zh rounded to f16: pk_rn16) and three f16 MFMAs
// per tile:  zh zh^T + zh zl^T + zl zh^T  (fp32 accumulate; the dropped zl zl^T term is < 2^-22
// relative).  Every f16 x f16 product is exact in fp32, so the only error is the 22-bit
// representation of z: ~4x the rounding error of the fp32 path, still two orders of magnitude
// inside the 1e-4 bar (tests/test_gpu_parity.py).  S is a power of two chosen per launch so that
// max |z| <= 2^14 (gather_scale_kernel): f16 overflow cannot happen, and values down to 2^-18 of the
// largest keep all 22 bits.  The right-hand side is still accumulated in fp32 from the raw rows.
//
// One "super-step" = 32 entries = the contraction depth of one MFMA.  Lane (g,c) holds, for entry
// n = 4e+g (e = 0..7) of the super-step and every 16-block v, feature 16v+c: T*8 dword gathers per
// lane, each instruction again reading 4 rows x 64 B.  Slot e of the lane is contraction index
// k = 8g+e of both MFMA operands (the same bijection entry <-> k on both sides, so the sum is over
// the 32 entries whatever the hardware's k order is).  Pipeline: convert super-step s (raw ->
// zh, zl; RHS update), issue the gathers of s+1 into the same raw registers, then the 3*tri(T)
// MFMAs of s while they fly; across rows the first super-step of the next row is in flight during
// the factorization, like above.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// E = entry slots per lane and super-step: 8 -> 32 entries, v_mfma_f32_16x16x32_f16 (T <= 7);
// 4 -> 16 entries, v_mfma_f32_16x16x16_f16 (T = 8: 144 accumulators + 64 raw + 64 operand registers do not fit).
// Both MFMAs take 16 cycles, so E = 8 does twice the work per matrix-pipe cycle.  Round 3: T = 6 (238 VGPRs, no
// spill) and T = 7 (256 VGPRs + 84 B of scratch in the rows kernel, 44 B in the segments kernel) moved to E = 8:
// C3 (k = 100) 16.2 -> 15.35 ms, k = 112 38.2 -> 36.3 ms per iteration on the same box.
template <int E>
struct ZOp {  // one f16 MFMA operand: E halves = E/2 dwords
  int r[E / 2];
};
template <int E>
__device__ __forceinline__ f32x4 mfma_h(const ZOp<E>& a, const ZOp<E>& b, f32x4 c) {
  if constexpr (E == 8) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
  }
}
#ifndef MALS_SPLIT8_MAXT
#define MALS_SPLIT8_MAXT 7
#endif
__host__ __device__ constexpr int split_slots(int T) { return T <= MALS_SPLIT8_MAXT ? 8 : 4; }

// the Gramian weight of the chunk becomes sqrt(w) * S
__device__ __forceinline__ void chunk_weights_h(const SolveParams& p, Chunk& e, float zscale) {
  chunk_weights(p, e);
  e.w = __builtin_amdgcn_sqrtf(e.w) * zscale;
}

// Gather the two entries 4e+g, e = 2*E2 and 2*E2+1, of one super-step: their columns sit in lanes
// (off/4 + e) of col_src (row-major chunk: off = 4 (16 g + E*part)).  Entries past the end of the row
// have a clamped column (chunk_issue) and zero weights, so they gather a valid, cached row that
// contributes nothing: no per-entry predication.  As in load_rows, the features past k of a partial last
// block are the zero padding of the table.
template <int T, int E, bool FULL, int E2>
__device__ __forceinline__ void issue_pair_h(const SolveParams& p, int col_src, int off, int lane, float (&raw)[T][E]) {
  const int c = lane & 15;
  int col[2];
  bperm2_i<8 * E2, 8 * E2 + 4>(off, col_src, col[0], col[1]);  // entries 4 (2 E2 + i) + g of the part: lanes 16 g + 2 E2 + i (+ E part)
#ifdef MALS_PROFILING
  if (p.flags & 0x400) {  // ablation (MALS_DEBUG_FLAGS=4): every gather hits the cache
    col[0] &= 0xfff;
    col[1] &= 0xfff;
  }
#endif
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = 2 * E2 + i;
    const float* ptr = p.M + ((uint64_t)(uint32_t)col[i] * (uint32_t)p.ldm + (uint32_t)c);
#pragma unroll
    for (int v = 0; v < T; ++v) raw[v][e] = ptr[16 * v];
  }
}

// issue_pair_h in two halves (convert_refill_h): the columns are requested before the conversion of the pair whose raw
// registers they will refill, the loads issued after it -- the LDS round trip runs under ~70 VALU instructions instead
// of stalling the wave four times per super-step
template <int E2>
__device__ __forceinline__ void fetch_pair_cols_h(int col_src, int off, int (&col)[2]) {
  bperm2_i_start<8 * E2, 8 * E2 + 4>(off, col_src, col[0], col[1]);
}
template <int T, int E, bool FULL, int E2>
__device__ __forceinline__ void issue_pair_loads_h(const SolveParams& p, int (&col)[2], int lane, float (&raw)[T][E]) {
  const int c = lane & 15;
  bperm2_i_land(col[0], col[1]);
#ifdef MALS_PROFILING
  if (p.flags & 0x400) {  // ablation (MALS_DEBUG_FLAGS=4): every gather hits the cache
    col[0] &= 0xfff;
    col[1] &= 0xfff;
  }
#endif
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = 2 * E2 + i;
    const float* ptr = p.M + ((uint64_t)(uint32_t)col[i] * (uint32_t)p.ldm + (uint32_t)c);
#pragma unroll
    for (int v = 0; v < T; ++v) raw[v][e] = ptr[16 * v];
  }
}

// raw rows of one entry pair -> scaled, split f16 operands; RHS partial sums (fp32, raw rows)
// NTERM = 3 (MALS_GRAMIAN_SPLIT3_F16): z = zh + zm + zl, three f16 numbers -- 33 significand bits, every fp32 z exactly -- and
// the six products at or above 2^-24 of the leading one (gram_super_step).
template <int T, int E, bool FULL, int PART, int E2, int NTERM = 2>
__device__ __forceinline__ void convert_pair_h(const SolveParams& p, const Chunk& ch, int lane, const float (&raw)[T][E], ZOp<E> (&zh)[T],
                                               ZOp<E> (&zl)[T], float (&bpart)[T], ZOp<E> (&zm)[NTERM == 3 ? T : 1]) {
  constexpr int m0 = E * PART + 2 * E2;  // entry 4 m0 + g of the chunk: lane m0 of this group's row (chunk_issue<true>)
  const float s0 = row_bcast<m0>(ch.w), s1 = row_bcast<m0 + 1>(ch.w);
  const float c0 = row_bcast<m0>(ch.cb), c1 = row_bcast<m0 + 1>(ch.cb);
#pragma unroll
  for (int v = 0; v < T; ++v) {
    const float y0 = raw[v][2 * E2], y1 = raw[v][2 * E2 + 1];
    const float z0 = y0 * s0, z1 = y1 * s1;
    // zh = z rounded to f16 (pk_rn16), zl = what is left: written
    // as an FMA on the widened half so that hipcc emits one v_fma_mix_f32 per value (and folds the
    // scaling in: zl is the residual of the exact product)
    const int hp = pk_rn16(z0, z1);
    const f16x2 hh = __builtin_bit_cast(f16x2, hp);
    zh[v].r[E2] = hp;
    if constexpr (NTERM == 3) {
      const float r0 = fmaf((float)hh[0], -1.f, z0), r1 = fmaf((float)hh[1], -1.f, z1);   // exact: the bits rounding dropped
      const int mp = pk_rn16(r0, r1);
      const f16x2 mh = __builtin_bit_cast(f16x2, mp);
      zm[v].r[E2] = mp;
      zl[v].r[E2] = pk_rn16(fmaf((float)mh[0], -1.f, r0), fmaf((float)mh[1], -1.f, r1));
    } else {
      zl[v].r[E2] = pk_rn16(fmaf((float)hh[0], -1.f, z0), fmaf((float)hh[1], -1.f, z1));
    }
    bpart[v] = fmaf(c0, y0, bpart[v]);
    bpart[v] = fmaf(c1, y1, bpart[v]);
  }
}

// Convert super-step (ch, PART) out of the raw registers and, pair by pair, refill them with the
// gathers of whatever comes next (columns in next_col at next_off): the registers are in flight
// again as soon as they have been read.  The fences keep that order: left alone, the scheduler
// issues the refills first and copies the old rows aside, at twice the registers.
template <int T, int E, bool FULL, int PART, int NTERM = 2>
__device__ __forceinline__ void convert_refill_h(const SolveParams& p, const Chunk& ch, int next_col, int next_off, int lane,
                                                 float (&raw)[T][E], ZOp<E> (&zh)[T], ZOp<E> (&zl)[T], float (&bpart)[T], ZOp<E> (&zm)[NTERM == 3 ? T : 1]) {
#define MALS_SB __builtin_amdgcn_sched_barrier(0)
  int col[2];
  fetch_pair_cols_h<0>(next_col, next_off, col);
  MALS_SB;
  convert_pair_h<T, E, FULL, PART, 0, NTERM>(p, ch, lane, raw, zh, zl, bpart, zm);
  MALS_SB;
  issue_pair_loads_h<T, E, FULL, 0>(p, col, lane, raw);
  fetch_pair_cols_h<1>(next_col, next_off, col);
  MALS_SB;
  convert_pair_h<T, E, FULL, PART, 1, NTERM>(p, ch, lane, raw, zh, zl, bpart, zm);
  MALS_SB;
  issue_pair_loads_h<T, E, FULL, 1>(p, col, lane, raw);
  if constexpr (E == 8) {
    fetch_pair_cols_h<2>(next_col, next_off, col);
    MALS_SB;
    convert_pair_h<T, E, FULL, PART, 2, NTERM>(p, ch, lane, raw, zh, zl, bpart, zm);
    MALS_SB;
    issue_pair_loads_h<T, E, FULL, 2>(p, col, lane, raw);
    fetch_pair_cols_h<3>(next_col, next_off, col);
    MALS_SB;
    convert_pair_h<T, E, FULL, PART, 3, NTERM>(p, ch, lane, raw, zh, zl, bpart, zm);
    MALS_SB;
    issue_pair_loads_h<T, E, FULL, 3>(p, col, lane, raw);
  }
#undef MALS_SB
}

template <int T, int E>
__device__ __forceinline__ void gram_super_step(const ZOp<E> (&zh)[T], const ZOp<E> (&zl)[T], f32x4 (&acc)[tri(T)]) {
  // three passes over the tiles, so that consecutive MFMAs never wait on each other's accumulator
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = i; j < T; ++j) acc[tidx(T, i, j)] = mfma_h<E>(zh[i], zh[j], acc[tidx(T, i, j)]);
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = i; j < T; ++j) acc[tidx(T, i, j)] = mfma_h<E>(zh[i], zl[j], acc[tidx(T, i, j)]);
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = i; j < T; ++j) acc[tidx(T, i, j)] = mfma_h<E>(zl[i], zh[j], acc[tidx(T, i, j)]);
}

// three terms: hh + (hm + mh) + (mm + hl + lh); what is dropped (ml, lm, ll) is below 2^-35 of the leading product
template <int T, int E>
__device__ __forceinline__ void gram_super_step3(const ZOp<E> (&zh)[T], const ZOp<E> (&zm)[T], const ZOp<E> (&zl)[T], f32x4 (&acc)[tri(T)]) {
#define MALS_GS3(A, B)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < T; ++i) _Pragma("unroll") for (int j = i; j < T; ++j)              \
      acc[tidx(T, i, j)] = mfma_h<E>(A[i], B[j], acc[tidx(T, i, j)])
  // smallest terms first: they are added to the accumulator before the large ones swamp their low bits
  MALS_GS3(zl, zh);
  MALS_GS3(zh, zl);
  MALS_GS3(zm, zm);
  MALS_GS3(zm, zh);
  MALS_GS3(zh, zm);
  MALS_GS3(zh, zh);
#undef MALS_GS3
}

// The gathers of a whole super-step (start of a wave's first row only).
template <int T, int E, bool FULL>
__device__ __forceinline__ void prime_row_h(const SolveParams& p, int col_src, int lane, float (&raw)[T][E]) {
  const int rb = (lane & 48) << 2;  // byte address of lane 0 of this group's row
  issue_pair_h<T, E, FULL, 0>(p, col_src, rb, lane, raw);
  issue_pair_h<T, E, FULL, 1>(p, col_src, rb, lane, raw);
  if constexpr (E == 8) {
    issue_pair_h<T, E, FULL, 2>(p, col_src, rb, lane, raw);
    issue_pair_h<T, E, FULL, 3>(p, col_src, rb, lane, raw);
  }
}

// One super-step of the row pipeline: PART-th group of 4E entries of chunk ch.
template <int T, int E, bool FULL, int PART, int NTERM = 2>
__device__ __forceinline__ void super_step_h(const SolveParams& p, const Chunk& ch, const Chunk& chn, int next_col, bool last, int lane,
                                             float (&raw)[T][E], f32x4 (&acc)[tri(T)], float (&bpart)[T]) {
  constexpr int NP = 16 / E;  // super-steps per 64-entry chunk
  const int rb = (lane & 48) << 2;  // byte address of lane 0 of this group's row
  ZOp<E> zh[T], zl[T], zm[NTERM == 3 ? T : 1];
  // what the raw registers are refilled with: the next part of this chunk, part 0 of the next chunk,
  // or (after the row's last super-step) the first super-step of the next row
  const int same_row_col = PART == NP - 1 ? chn.col : ch.col;
  const int same_row_off = PART == NP - 1 ? rb : rb + 4 * E * (PART + 1);
  convert_refill_h<T, E, FULL, PART, NTERM>(p, ch, last ? next_col : same_row_col, last ? rb : same_row_off, lane, raw, zh, zl, bpart, zm);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NTERM == 3) gram_super_step3<T, E>(zh, zm, zl, acc);
  else gram_super_step<T, E>(zh, zl, acc);
  __builtin_amdgcn_sched_barrier(0);
}

// acc (zero on entry) += S^2 * sum_n w_n y_n y_n^T,  bpart += sum_n cb_n y_n  for a row with len > 0.
// On entry ch = chunk 0 (weights done) and raw = super-step 0 in flight; on exit raw = super-step 0
// of the NEXT row (columns in the first lanes of next_col) in flight.
template <int T, int E, bool FULL, int NTERM = 2>
__device__ __forceinline__ void gather_row_h(const SolveParams& p, int64_t begin, int len, int lane, float zscale, Chunk& ch,
                                             int next_col, float (&raw)[T][E], f32x4 (&acc)[tri(T)], float (&bpart)[T]) {
  constexpr int NS = 4 * E, NP = 16 / E;
  const int n_ss = (len + NS - 1) / NS;
  for (int q = 0; NP * q < n_ss; ++q) {
    // next chunk (clamped inside the row, so always safe to issue); lands during this chunk
    Chunk chn = chunk_issue<true>(p, begin, len, 64 * (q + 1), lane);
    const int ss = NP * q;
    super_step_h<T, E, FULL, 0, NTERM>(p, ch, chn, next_col, ss + 1 >= n_ss, lane, raw, acc, bpart);
    if (ss + 1 < n_ss) {
      if (NP == 2) chunk_weights_h(p, chn, zscale);
      super_step_h<T, E, FULL, 1, NTERM>(p, ch, chn, next_col, ss + 2 >= n_ss, lane, raw, acc, bpart);
    }
    if constexpr (NP == 4) {
      if (ss + 2 < n_ss) super_step_h<T, E, FULL, 2, NTERM>(p, ch, chn, next_col, ss + 3 >= n_ss, lane, raw, acc, bpart);
      if (ss + 3 < n_ss) {
        chunk_weights_h(p, chn, zscale);
        super_step_h<T, E, FULL, 3, NTERM>(p, ch, chn, next_col, ss + 4 >= n_ss, lane, raw, acc, bpart);
      }
    }
    ch = chn;
  }
}

// acc <- shared Gramian image (ALS:447-450: start from YTY unless lossIgnoresUnspecified).  Gf is
// stored in acc layout, [tile][lane] float4, so this is tri(T) 16-byte loads and no VALU.
template <int T>
__device__ __forceinline__ void init_acc(const SolveParams& p, f32x4 (&acc)[tri(T)], int lane) {
  if (!(p.flags & 2)) {
    const f32x4* G4 = reinterpret_cast<const f32x4*>(p.Gf) + lane;
#pragma unroll
    for (int t = 0; t < tri(T); ++t) acc[t] = G4[t * 64];
  } else {
#pragma unroll
    for (int t = 0; t < tri(T); ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// W += lambda*alpha*n_u I (ALS:488-492); identity on the padding features
template <int T, bool FULL = false>
__device__ __forceinline__ void add_ridge(const SolveParams& p, f32x4 (&acc)[tri(T)], int n_u, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const float ridge = p.lambda_alpha * (float)n_u;
  // whole blocks: the diagonal sits at register r of lane (g, 4g+r) in every diagonal tile, so the four
  // selects are made once and the tiles get plain adds
  float rd[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rd[r] = (4 * g + r == c) ? ridge : 0.f;
  float w00 = 1.f;
  if constexpr (!FULL) w00 = readlane(acc[tidx(T, 0, 0)][0], 0) + ridge;  // lane 0, register 0 of tile (0,0) = W[0][0]
#pragma unroll
  for (int v = 0; v < T; ++v) {
    if (FULL || v < T - 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[tidx(T, v, v)][r] += rd[r];
    } else {  // partial last block: the padding features are decoupled from the others (zero rows and columns);
              // their diagonal gets W[0][0], a value on the scale of the real pivots, so that the smallest pivot and
              // the conditioning estimate of store_row speak about the real system only
      const int feat = 16 * v + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * g + r == c) acc[tidx(T, v, v)][r] = feat < p.k ? acc[tidx(T, v, v)][r] + ridge : w00;
      }
    }
  }
}

// bit pattern of the largest |entry| of the diagonal tiles of an SPD matrix = its largest diagonal element = its
// largest |entry| (no need to pick the diagonal out of the tiles); uniform over the wave.  Non-negative floats order
// like their bit patterns: integer max (a float max of a DPP / bpermute result costs an extra canonicalising v_max).
__device__ __forceinline__ int wave_max_bits(float mf, int lane) {
  int m = __float_as_int(mf);
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x120 + 8, 0xf, 0xf, false));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x120 + 4, 0xf, 0xf, false));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x120 + 2, 0xf, 0xf, false));
  m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x120 + 1, 0xf, 0xf, false));
  m = max(m, bperm_i((lane ^ 16) << 2, m));
  m = max(m, bperm_i((lane ^ 32) << 2, m));
  return m;
}
template <int T>
__device__ __forceinline__ int max_entry_bits(const f32x4 (&acc)[tri(T)], int lane) {
  float mf = 0.f;
#pragma unroll
  for (int v = 0; v < T; ++v) {
    const f32x4& d = acc[tidx(T, v, v)];
    mf = fmaxf(mf, fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), fabsf(d[3]))));
  }
  return wave_max_bits(mf, lane);
}

// the same for W minus the Gramian image it started from: the largest entry of the row's own part sum w y y^T
// (gimg: [tile][lane] float4 image, global or LDS; zeros under lossIgnoresUnspecified)
template <int T>
__device__ __forceinline__ float row_part_max(const f32x4 (&acc)[tri(T)], const f32x4* gimg, int lane) {
  float mf = 0.f;
#pragma unroll
  for (int v = 0; v < T; ++v) {
    const f32x4 d = acc[tidx(T, v, v)] - gimg[tidx(T, v, v) * 64 + lane];
    mf = fmaxf(mf, fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), fabsf(d[3]))));
  }
  return __int_as_float(wave_max_bits(mf, lane));
}

// Packed fp32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32 on whole accumulator tiles) up to this many feature blocks
#ifndef MALS_PK_MAXT
#define MALS_PK_MAXT 7
#endif
// Scale the row's system W x = b by s^2 = 2^(2p) with s * sqrt(max_i W_ii) <= 2^13 (entries of the
// Cholesky factor of s^2 W are then <= 2^13): exact, and undone by comparing the pivots against
// threshold * s^2 (the caller multiplies minpiv by the returned 1/s^2) -- x itself is unchanged.
// wmax <- the largest entry of W before the scaling
template <int T>
__device__ __forceinline__ float row_scale(f32x4 (&acc)[tri(T)], float (&bcol)[T], int lane, float& wmax) {
  const int m = max_entry_bits<T>(acc, lane);
  wmax = __int_as_float(m);
  const int e = ((m >> 23) & 255) - 126;  // largest entry < 2^e
  int p2 = 2 * (13 - ((e + 1) >> 1));
  p2 = p2 < -100 ? -100 : (p2 > 100 ? 100 : p2);
  const float s2 = __int_as_float((p2 + 127) << 23), inv_s2 = __int_as_float((127 - p2) << 23);
#pragma unroll
  for (int t = 0; t < tri(T); ++t) {
    if constexpr (T <= MALS_PK_MAXT) {
      acc[t] = acc[t] * s2;  // whole-tile products: v_pk_mul_f32, two values per issue slot
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] *= s2;
    }
  }
#pragma unroll
  for (int v = 0; v < T; ++v) bcol[v] *= s2;
  return inv_s2;
}

// store x (the cast to fp32 of CMS:40-42 is implicit: all arithmetic here is fp32); flag non-PD rows
// wmax = max(largest entry of the row's own part sum w y y^T, a quarter of the largest entry of W): wmax / minpiv
// estimates from below how much the fp32 roundings are amplified in x.  The row's own part is an fp32 accumulation
// (measured 2-5e-7 of x per unit of the ratio); the shared Gramian under it is an fp64 sum rounded once and only
// suffers the factorization's rounding (measured 1e-7 per unit, on rows of the C5 shape whose W is all Gramian) --
// but it must count: with reconstructR there is no row part at all, and G + rho I with fewer factor rows than
// features is as ill-conditioned as anything (sweep cases 2145, 2550) and loses ~1e-6 per unit on the split-f16
// factorization (case 3047): there the whole of W counts in full and the limit is a quarter (host).  Rows above
// refine_limit are marked.
// PERM: xcol is in the feature order of lds_kernels.h (block v, lane c <-> feature 8c + (v ^ 4 (c >> 3)), T = 8, k = 128):
// lane (g, c), g < 2, holds features 8c + 4g .. 8c + 4g + 3 in blocks 0..3 or 4..7 and writes them as 16 bytes
template <int T, bool PERM = false>
__device__ __forceinline__ void store_row(const SolveParams& p, float (&xcol)[T], float minpiv, float wmax, int row, int lane) {
  if (!(minpiv > p.sing_threshold)) {
    // an fp32 pivot at the threshold is rounding noise once cond(W) passes ~1e7: the verdict belongs to the fp64
    // restatement (als_exact_kernel, mark 2); without a mark array this kernel's word is final
    if (lane == 0) {
      if (p.refine_flag) {
        p.refine_flag[row] = 2;
        *p.any_marked = 1;
      } else {
        atomicMin(p.bad_row, (unsigned long long)row);
      }
    }
#pragma unroll
    for (int v = 0; v < T; ++v) xcol[v] = 0.f;
  } else {
    if (minpiv <= 1024.f * p.sing_threshold && lane == 0) {
      const unsigned long long key = ((unsigned long long)__float_as_uint(minpiv) << 32) | (unsigned)row;
      // same-address atomics serialise: only rows that lower the minimum issue one
      if (key < __builtin_nontemporal_load(p.suspect)) atomicMin(p.suspect, key);
    }
    if (p.refine_flag && p.refine_limit > 0.f && wmax > p.refine_limit * minpiv && lane == 0) {
      p.refine_flag[row] = 1;
      *p.any_marked = 1;
    }
  }
  // xcol[v] is the same in all four lane groups: group g writes blocks g, 4 + g -- (T + 3) / 4 full-wave stores, and only a
  // store that can reach the last block needs the feature bound (k > 16 (T - 1)); T quarter-wave stores under T
  // hoisted predicates cost 190-250 instructions per row, most of them reloads of spilled exec masks
  if constexpr (PERM) {
    static_assert(T == 8, "the permuted feature order belongs to the k = 128 kernels");
    float* o = p.out + (int64_t)row * p.k;
    int l2 = lane;
    asm volatile("" : "+v"(l2));
    const int g = l2 >> 4, c = l2 & 15;
    const bool hi = ((g ^ (c >> 3)) & 1) != 0;
    f32x4 x;
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = hi ? xcol[4 + q] : xcol[q];
    if (g < 2) *reinterpret_cast<f32x4*>(o + 8 * c + 4 * g) = x;
  } else {
    float* o = p.out + (int64_t)row * p.k;
    int l2 = lane;
    asm volatile("" : "+v"(l2));  // (the masks and offsets below recomputed here, not kept live from the top of the row loop)
    const int g = l2 >> 4, c = l2 & 15;
#pragma unroll
    for (int m = 0; 4 * m < T; ++m) {
      const float x = select4(g, xcol[4 * m], 4 * m + 1 < T ? xcol[4 * m + 1] : 0.f, 4 * m + 2 < T ? xcol[4 * m + 2] : 0.f,
                              4 * m + 3 < T ? xcol[4 * m + 3] : 0.f);
      const int feat = 16 * (4 * m + g) + c;
      if (4 * m + 3 < T - 1 || feat < p.k) o[feat] = x;
    }
  }
}

// ridge + factor + solve + store, no prefetch hook (used by the long-row finish kernel)
template <int T, bool PERM = false>
__device__ __forceinline__ void finish_row(const SolveParams& p, f32x4 (&acc)[tri(T)], const float (&bcol)[T],
                                           int n_u, int row, int lane, float rmax = 0.f) {
  add_ridge<T, PERM>(p, acc, n_u, lane);   // (PERM implies k = 128 = 16 T: the ridge is a plain diagonal)
  float minpiv = 3.0e38f;
  float xcol[T];
  const float wmax = fmaxf(rmax, p.gramian_weight * __int_as_float(max_entry_bits<T>(acc, lane)));
  cholesky_tiles<T>(acc, lane, minpiv);
  solve_tiles<T>(acc, bcol, xcol, lane);
  store_row<T, PERM>(p, xcol, minpiv, wmax, row, lane);
}

__device__ __forceinline__ WorkItem load_item(const SolveParams& p, int64_t it) {
  WorkItem w;
  if (it < p.n_work) {
    w = p.items[it];
  } else {
    w.begin = 0;
    w.len = -1;  // sentinel: no more work
    w.id = 0;
  }
  return w;
}

// Lists A (MODE 0: rows no longer than segment_nnz, fused K2+K3) and B (MODE 1: segments of long
// rows, K2 only, partial tiles + RHS to scratch).  Persistent waves, see the K2 header comment.
template <int T, int D, int MODE, bool FULL>
__global__ __launch_bounds__(256, MALS_WAVES(T, MODE)) void als_persistent_kernel(SolveParams p) {
  __shared__ f32x4 sG[MODE == 0 ? tri(T) * 64 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  if (MODE == 0) {  // stage the acc-layout Gramian image once per workgroup
    const f32x4* G4 = reinterpret_cast<const f32x4*>(p.Gf);
    for (int e = threadIdx.x; e < tri(T) * 64; e += 256) sG[e] = (p.flags & 2) ? f32x4{0.f, 0.f, 0.f, 0.f} : G4[e];
    __syncthreads();
  }
  int64_t it = wave;
  if (it >= p.n_work) return;
  if ((p.flags & 8) && p.zscale[2] != 0.f) return;  // enqueued as the fallback of a split-precision launch that ran
  WorkItem cur = load_item(p, it);
  WorkItem nxt = load_item(p, it + n_waves);
  Pipe<T, D> pp;
  pp.wcur = pp.cbcur = 0.f;
  pp.colpf = 0;
  pp.ch.col = 0;
  pp.ch.w = pp.ch.cb = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int v = 0; v < T; ++v) pp.y[i][v] = 0.f;
  if (cur.len > 0) {
    pp.ch = chunk_issue(p, cur.begin, cur.len, 0, lane);
    chunk_weights(p, pp.ch);
    prime_row<T, D, FULL>(p, lane, pp);
  }
  for (;;) {
    f32x4 acc[tri(T)];
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < tri(T); ++t) acc[t] = sG[t * 64 + lane];
    } else {
#pragma unroll
      for (int t = 0; t < tri(T); ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float bpart[T];
#pragma unroll
    for (int v = 0; v < T; ++v) bpart[v] = 0.f;
#ifdef MALS_PROFILING  // per-phase cycle stamps + ablation switches; not compiled into the product library
    const int64_t trow = it / n_waves - p.trace_start;
    const bool tr = MODE == 0 && p.trace && wave < 64 && trow >= 0 && trow < 64;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (tr) t0 = __builtin_readcyclecounter();
    if (!(p.flags & 0x200)) gather_row<T, D, FULL>(p, cur.begin, cur.len, lane, pp, acc, bpart);
    if (tr) t1 = __builtin_readcyclecounter();
#else
    gather_row<T, D, FULL>(p, cur.begin, cur.len, lane, pp, acc, bpart);
#endif
    float bcol[T];
#pragma unroll
    for (int v = 0; v < T; ++v) bcol[v] = reduce_groups(bpart[v], lane);
    // next row: first chunk now, item after next (scalar load) now, row gathers after the factorization
    const bool prime_next = nxt.len > 0;
    if (prime_next) pp.ch = chunk_issue(p, nxt.begin, nxt.len, 0, lane);
    const WorkItem nxt2 = load_item(p, it + 2 * n_waves);
    if (MODE == 0) {
      const float wmax = p.refine_flag ? fmaxf(row_part_max<T>(acc, sG, lane), p.gramian_weight * __int_as_float(max_entry_bits<T>(acc, lane))) : 0.f;
      add_ridge<T, FULL>(p, acc, cur.len, lane);
      float minpiv = 3.0e38f;
      float xcol[T];
#ifdef MALS_PROFILING
      if (p.flags & 0x100) {  // ablation (MALS_DEBUG_FLAGS): skip K3
#pragma unroll
        for (int v = 0; v < T; ++v) xcol[v] = 1e-3f + 1e-9f * (bcol[v] + acc[tidx(T, v, v)][0] + acc[tidx(T, 0, v)][1]);
        if (prime_next) {
          chunk_weights(p, pp.ch);
          prime_row<T, D, FULL>(p, lane, pp);
        }
      } else
#endif
      {
        cholesky_tiles<T>(acc, lane, minpiv);
#ifdef MALS_PROFILING
        if (tr) t2 = __builtin_readcyclecounter();
#endif
        if (prime_next) {  // gathers land during the solves
          chunk_weights(p, pp.ch);
          prime_row<T, D, FULL>(p, lane, pp);
        }
        solve_tiles<T>(acc, bcol, xcol, lane);
      }
      store_row<T>(p, xcol, minpiv, wmax, cur.id, lane);
#ifdef MALS_PROFILING
      if (tr) {
        t3 = __builtin_readcyclecounter();
        if (lane == 0) {
          unsigned long long* o = p.trace + (trow * 64 + wave) * 6;
          o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = (unsigned long long)cur.len; o[5] = wall_clock64();
        }
      }
#endif
    } else {
      // partial slot: tri(T) tiles as one float4 per lane (16-byte stores), then the T RHS blocks
      float* s = p.scratch + (int64_t)cur.id * ((tri(T) * 4 + T) * 64);
#pragma unroll
      for (int t = 0; t < tri(T); ++t) reinterpret_cast<f32x4*>(s)[t * 64 + lane] = acc[t];
#pragma unroll
      for (int v = 0; v < T; ++v) s[(tri(T) * 4 + v) * 64 + lane] = bcol[v];
      if (prime_next) {
        chunk_weights(p, pp.ch);
        prime_row<T, D, FULL>(p, lane, pp);
      }
    }
    if (nxt.len < 0) break;
    cur = nxt;
    nxt = nxt2;
    it += n_waves;
  }
}

// The same two lists with the split-precision gather (gather_row_h).  Pipeline per wave: the raw
// registers always hold the NEXT super-step in flight -- of this row or, from the last convert of
// a row on, super-step 0 of the next row, which therefore flies during the whole factorization.
// That needs the next row's first chunk (col, value) on chip a row ahead (nch) and the work items three ahead.  Rows are sorted by length, so the empty rows
// (nothing to gather: W = G, b = 0) form the tail of a wave's list and are handled after the loop.
template <int T, int MODE, bool FULL, int NTERM = 2>
__global__ __launch_bounds__(256, MALS_WAVES_H(T, MODE)) void als_persistent_kernel_h(SolveParams p) {
  __shared__ f32x4 sG[MODE == 0 ? tri(T) * 64 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  if (MODE == 0) {  // stage the acc-layout Gramian image once per workgroup
    const f32x4* G4 = reinterpret_cast<const f32x4*>(p.Gf);
    for (int e = threadIdx.x; e < tri(T) * 64; e += 256) sG[e] = (p.flags & 2) ? f32x4{0.f, 0.f, 0.f, 0.f} : G4[e];
    __syncthreads();
  }
  int64_t it = wave;
  if (it >= p.n_work) return;
  if (p.zscale[2] == 0.f) return;  // operand range too wide for the f16 split: the fp32 kernels behind this launch run
  const float zscale = __int_as_float(uniform(__float_as_int(p.zscale[0])));
  const float inv_s2 = __int_as_float(uniform(__float_as_int(p.zscale[1])));
  WorkItem cur = load_item(p, it);
  WorkItem nxt = load_item(p, it + n_waves);
  WorkItem nx2 = load_item(p, it + 2 * n_waves);
  constexpr int E = split_slots(T);
  float raw[T][E];
#pragma unroll
  for (int v = 0; v < T; ++v)
#pragma unroll
    for (int e = 0; e < E; ++e) raw[v][e] = 0.f;
  if (cur.len > 0) {
    Chunk ch = chunk_issue<true>(p, cur.begin, cur.len, 0, lane);
    // first chunks of the next two rows; where there is no such row, any valid columns do
    Chunk nch = nxt.len > 0 ? chunk_issue<true>(p, nxt.begin, nxt.len, 0, lane) : ch;
    chunk_weights_h(p, ch, zscale);
    prime_row_h<T, E, FULL>(p, ch.col, lane, raw);
    for (;;) {
      const WorkItem nx3 = load_item(p, it + 3 * n_waves);
      f32x4 acc[tri(T)];
#pragma unroll
      for (int t = 0; t < tri(T); ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      float bpart[T];
#pragma unroll
      for (int v = 0; v < T; ++v) bpart[v] = 0.f;
#ifdef MALS_PROFILING
      const int64_t trow = it / n_waves - p.trace_start;
      const bool tr = MODE == 0 && p.trace && wave < 64 && trow >= 0 && trow < 64;
      unsigned long long t0 = 0, t1 = 0, t2 = 0;
      if (tr) t0 = __builtin_readcyclecounter();
#endif
      gather_row_h<T, E, FULL, NTERM>(p, cur.begin, cur.len, lane, zscale, ch, nch.col, raw, acc, bpart);
#ifdef MALS_PROFILING
      if (tr) t1 = __builtin_readcyclecounter();
#endif
      float bcol[T];
#pragma unroll
      for (int v = 0; v < T; ++v) bcol[v] = reduce_groups(bpart[v], lane);
      ch = nch;
      if (nxt.len > 0) chunk_weights_h(p, ch, zscale);
      if (MODE == 0) {
        const float rmax = p.refine_flag ? __int_as_float(max_entry_bits<T>(acc, lane)) * inv_s2 : 0.f;
        // back to the unscaled system (S^2 is a power of two: exact), on top of the shared Gramian
#pragma unroll
        for (int t = 0; t < tri(T); ++t) {
          const f32x4 g4 = sG[t * 64 + lane];
          if constexpr (T <= MALS_PK_MAXT) {
            acc[t] = __builtin_elementwise_fma(acc[t], f32x4{inv_s2, inv_s2, inv_s2, inv_s2}, g4);  // v_pk_fma_f32
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(acc[t][r], inv_s2, g4[r]);
          }
        }
        add_ridge<T, FULL>(p, acc, cur.len, lane);
        float minpiv = 3.0e38f, wmax;
        float xcol[T];
        if constexpr (T >= 2) {  // the rank-16 updates of the factorization on the f16 pipe as well
          const float inv_s2row = row_scale<T>(acc, bcol, lane, wmax);
          cholesky_tiles<T, true>(acc, lane, minpiv);
          minpiv *= inv_s2row;
        } else {
          wmax = __int_as_float(max_entry_bits<T>(acc, lane));
          cholesky_tiles<T>(acc, lane, minpiv);
        }
#ifdef MALS_PROFILING
        if (tr) t2 = __builtin_readcyclecounter();
#endif
        solve_tiles<T>(acc, bcol, xcol, lane);
        store_row<T>(p, xcol, minpiv, fmaxf(rmax, p.gramian_weight * wmax), cur.id, lane);
#ifdef MALS_PROFILING
        if (tr) {
          const unsigned long long t3 = __builtin_readcyclecounter();
          if (lane == 0) {
            unsigned long long* o = p.trace + (trow * 64 + wave) * 6;
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = (unsigned long long)cur.len; o[5] = wall_clock64();
          }
        }
#endif
      } else {
        float* s = p.scratch + (int64_t)cur.id * ((tri(T) * 4 + T) * 64);
#pragma unroll
        for (int t = 0; t < tri(T); ++t) reinterpret_cast<f32x4*>(s)[t * 64 + lane] = acc[t] * inv_s2;
#pragma unroll
        for (int v = 0; v < T; ++v) s[(tri(T) * 4 + v) * 64 + lane] = bcol[v];
      }
      cur = nxt;
      nxt = nx2;
      nx2 = nx3;
      // first chunk of the row after the next (nxt, after the shift): requested a whole row (~10-20 us) before the end
      // of the next row needs its columns.  (Until round 3 it was requested two rows ahead, in three more registers:
      // measured neutral on C4 / C5, 1 % slower on C3, where the registers spill.)
      nch = nxt.len > 0 ? chunk_issue<true>(p, nxt.begin, nxt.len, 0, lane) : nch;
      it += n_waves;
      if (cur.len <= 0) break;
    }
  }
  if (MODE == 0) {
    while (cur.len == 0) {  // empty rows: W = G (+ 0 ridge), b = 0
      f32x4 acc[tri(T)];
#pragma unroll
      for (int t = 0; t < tri(T); ++t) acc[t] = sG[t * 64 + lane];
      float bcol[T];
#pragma unroll
      for (int v = 0; v < T; ++v) bcol[v] = 0.f;
      finish_row<T>(p, acc, bcol, 0, cur.id, lane);
      it += n_waves;
      cur = load_item(p, it);
    }
  }
}

// acc, bcol += one partial slot.  All loads of a pass first (16 bytes per lane and tile), then the adds; from T = 6 on in
// two passes of half the tiles -- a whole slot in flight next to the accumulators is 2 x tri(T) x 4 registers, which put
// the T = 7 finish kernel at one wave per SIMD (247 VGPRs + 92 AGPRs)
template <int T>
__device__ __forceinline__ void add_slot(const float* __restrict__ s, int lane, f32x4 (&acc)[tri(T)], float (&bcol)[T]) {
  constexpr int PASSES = T >= 6 ? 2 : 1, PER = (tri(T) + PASSES - 1) / PASSES;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    f32x4 part[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t)
      if (ps * PER + t < tri(T)) part[t] = reinterpret_cast<const f32x4*>(s)[(ps * PER + t) * 64 + lane];
    if (ps == 0) {
      float bp[T];
#pragma unroll
      for (int v = 0; v < T; ++v) bp[v] = s[(tri(T) * 4 + v) * 64 + lane];
#pragma unroll
      for (int v = 0; v < T; ++v) bcol[v] += bp[v];
    }
#pragma unroll
    for (int t = 0; t < PER; ++t)
      if (ps * PER + t < tri(T)) acc[ps * PER + t] += part[t];
    if (PASSES > 1) __builtin_amdgcn_sched_barrier(0);  // keep the passes apart (the scheduler would merge their loads again)
  }
}

// groups of list C (RowC with stride 1, row unused): one wave per group sums its slots in order into the first
template <int T>
__global__ __launch_bounds__(256) void als_prereduce_kernel(SolveParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= p.n_work) return;
  RowC rc = p.rowsC[wave];
  rc.first_slot = uniform64(rc.first_slot);
  rc.nseg = uniform(rc.nseg);
  constexpr int SLOT = (tri(T) * 4 + T) * 64;
  float* s0 = p.scratch + rc.first_slot * (int64_t)SLOT;
  f32x4 acc[tri(T)];
  float bcol[T];
#pragma unroll
  for (int t = 0; t < tri(T); ++t) acc[t] = reinterpret_cast<const f32x4*>(s0)[t * 64 + lane];
#pragma unroll
  for (int v = 0; v < T; ++v) bcol[v] = s0[(tri(T) * 4 + v) * 64 + lane];
  for (int sgi = 1; sgi < rc.nseg; ++sgi) {
    add_slot<T>(s0 + sgi * (int64_t)SLOT, lane, acc, bcol);
  }
#pragma unroll
  for (int t = 0; t < tri(T); ++t) reinterpret_cast<f32x4*>(s0)[t * 64 + lane] = acc[t];
#pragma unroll
  for (int v = 0; v < T; ++v) s0[(tri(T) * 4 + v) * 64 + lane] = bcol[v];
}

// list C: one wave per long row: sum the segment partials (or their group sums) in order, then K3
// PERM: the partial slots come from als_lds_kernel_h<1> (lds_kernels.h) -- its feature order, its Gramian image
template <int T, bool PERM = false>
__global__ __launch_bounds__(256, T >= 6 ? 2 : 1) void als_finish_kernel(SolveParams p) {  // (T >= 6: without the bound hipcc takes 290-370 registers)
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= p.n_work) return;
  // the slots are in the order of whichever segments kernel ran: the LDS-staged one (range flag set) or its fp32 twin
  if (PERM && p.zscale[2] == 0.f) return;
  if (!PERM && (p.flags & 8) && p.zscale[2] != 0.f) return;
  RowC rc = p.rowsC[wave];
  rc.first_slot = uniform64(rc.first_slot);
  rc.row = uniform(rc.row);
  rc.nseg = uniform(rc.nseg);
  rc.stride = uniform(rc.stride);
  f32x4 acc[tri(T)];
#pragma unroll
  for (int t = 0; t < tri(T); ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bcol[T];
#pragma unroll
  for (int v = 0; v < T; ++v) bcol[v] = 0.f;
  for (int sgi = 0; sgi < rc.nseg; ++sgi) {
    const float* s = p.scratch + (rc.first_slot + (int64_t)sgi * rc.stride) * (int64_t)((tri(T) * 4 + T) * 64);
    add_slot<T>(s, lane, acc, bcol);
  }
  const int n_u = uniform((int)(p.row_ptr[rc.row + 1] - p.row_ptr[rc.row]));
  // the row's own part first (its largest entry feeds the conditioning estimate), then the shared Gramian under it
  const float rmax = p.refine_flag ? __int_as_float(max_entry_bits<T>(acc, lane)) : 0.f;
  if (!(p.flags & 2)) {
    const f32x4* G4 = reinterpret_cast<const f32x4*>(PERM ? p.Gperm : p.Gf) + lane;
#pragma unroll
    for (int t = 0; t < tri(T); ++t) acc[t] += G4[t * 64];
  }
  finish_row<T, PERM>(p, acc, bcol, n_u, rc.row, lane, rmax);
}

// ------------------------------------------------------------------------------------------------
// Mixed-precision refinement of the rows the solving kernels marked (store_row: largest entry of W over smallest
// pivot above refine_limit).  The reference solves every row in fp64 (ALS:494, CMLSS:37-55); an fp32 accumulation +
// factorization loses about cond(W) 6e-8 of x, which passes the 1e-4 bar up to cond(W) ~ 1e3 and not beyond
// (confidence weights alpha |r| in the thousands against a small lambda).  Per marked row, one wave:
//   W~, b~ and the Cholesky factor exactly as the fp32 rows kernel builds them, x0 = W~^-1 b~;
//   then conjugate gradients on the EXACT system W x = b, preconditioned with that factor (in registers):
//   every product with W is  sum_e w_e (y_e . v) y_e + G v + rho v  in fp64, straight from the entries, the fp32
//   factor rows and the fp64 Gramian -- one pass over the row's entries per iteration.  With a good factor this is
//   classical iterative refinement (one or two passes); with a poor one (cond(W) 1e6 and more, where a plain
//   correction step can make a good x0 WORSE -- measured) it still converges, monotonically in the energy norm.
//   Stops when a step is below 1e-6 |x| (at most 12 iterations).
// Rows of any length (a long row is simply gathered by its one wave: this is the slow path), either solve path
// (a dual row is re-solved in the original basis, after the un-rotation).  Not marked: nothing happens.
struct RefineParams {
  SolveParams p;                  // the half-iteration's parameters (gather table, CSR, Gramian image, out, flags)
  const double* G;                // k x k row-major fp64 Gramian of the gathered side (exact products)
  const double* Gref;             // the same with every product rounded to fp32 first, as the reference forms it
  const int* gref_state;          // [1] != 0: Gref is there (gramian_ref_kernel)
  unsigned long long* n_refined;  // statistics
  int64_t row_begin, row_end;     // local rows of the chunk
  double alpha, lambda_alpha;
};

// out = (with_rhs ? b : 0) - W v  for the row's exact system, fp64.  v and out in "col" layout: every lane (g, c)
// holds features 16 v' + c (all four lane groups the same values).  xs: 16 T doubles of LDS owned by the wave.
template <int T>
__device__ __forceinline__ void refine_matvec(const RefineParams& q, int64_t begin, int len, int lane, volatile double* xs,
                                              const double (&v)[T], bool with_rhs, double (&out)[T]) {
  const SolveParams& p = q.p;
  const int g = lane >> 4, c = lane & 15;
  const double base_w = (p.flags & 2) ? 1.0 : 0.0;
#pragma unroll
  for (int u = 0; u < T; ++u) out[u] = 0.0;
  for (int s4 = 0; s4 < len; s4 += 4) {  // four entries per step, one per lane group
    const int e = s4 + g;
    const bool ok = e < len;
    const int64_t at = begin + (ok ? e : len - 1);
    const int col = p.col[at];
    const double r = (double)p.val[at];
    const float* yp = p.M + ((uint64_t)(uint32_t)col * (uint32_t)p.ldm + (uint32_t)c);
    double y[T];
#pragma unroll
    for (int u = 0; u < T; ++u) y[u] = (double)yp[16 * u];
    double dot = 0.0;
#pragma unroll
    for (int u = 0; u < T; ++u) dot = fma(y[u], v[u], dot);
    for (int off = 8; off > 0; off >>= 1) dot += __shfl_xor(dot, off);  // over the 16 lanes of the group
    double wgt, cb;
    if (p.flags & 1) {  // ALS:466-469
      wgt = base_w;
      cb = r;
    } else {            // ALS:471-482
      const double ar = q.alpha * fabs(r);
      wgt = base_w + ar;
      cb = r > 0.0 ? 1.0 + ar : 0.0;
    }
    const double t = ok ? (with_rhs ? cb : 0.0) - wgt * dot : 0.0;
#pragma unroll
    for (int u = 0; u < T; ++u) out[u] = fma(t, y[u], out[u]);
  }
#pragma unroll
  for (int u = 0; u < T; ++u) {
    out[u] += __shfl_xor(out[u], 16);
    out[u] += __shfl_xor(out[u], 32);
  }
  if (!(p.flags & 2)) {  // - G v: v through LDS, row j of the (symmetric) Gramian read along the features
    if (g == 0) {
#pragma unroll
      for (int u = 0; u < T; ++u) xs[16 * u + c] = v[u];
    }
    __builtin_amdgcn_wave_barrier();
    double gv[T];
#pragma unroll
    for (int u = 0; u < T; ++u) gv[u] = 0.0;
    const double* G = q.gref_state[1] ? q.Gref : q.G;
    for (int j = g; j < p.k; j += 4) {
      const double vj = xs[j];
      const double* gr = G + (int64_t)j * p.k;
#pragma unroll
      for (int u = 0; u < T; ++u) {
        const int f = 16 * u + c;
        gv[u] = fma(f < p.k ? gr[f] : 0.0, vj, gv[u]);
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < T; ++u) {
      gv[u] += __shfl_xor(gv[u], 16);
      gv[u] += __shfl_xor(gv[u], 32);
      out[u] -= gv[u];
    }
  }
  const double rho = q.lambda_alpha * (double)len;
#pragma unroll
  for (int u = 0; u < T; ++u) out[u] -= rho * v[u];
}

// sum over the features of a[.] b[.] (col layout: 16 lanes x T values), the same in every lane
template <int T>
__device__ __forceinline__ double refine_dot(const double (&a)[T], const double (&b)[T]) {
  double d = 0.0;
#pragma unroll
  for (int u = 0; u < T; ++u) d = fma(a[u], b[u], d);
  for (int off = 8; off > 0; off >>= 1) d += __shfl_xor(d, off);
  return d;
}

template <int T>
__global__ __launch_bounds__(256, T >= 7 ? 1 : 2) void als_refine_kernel(RefineParams q) {
  const SolveParams& p = q.p;
  __shared__ double sx[4][16 * T];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv, n_waves = (int64_t)gridDim.x * 4;
  unsigned long long done = 0;
  // 64 rows per wave and step: one flag byte per lane, then the marked ones of the block one after the other
  for (int64_t blk = q.row_begin + 64 * wave; blk < q.row_end; blk += 64 * n_waves) {
    unsigned long long marks = __ballot(blk + lane < q.row_end && p.refine_flag[blk + lane] == 1);
    while (marks) {
      const int64_t row = blk + __builtin_ctzll(marks);
      marks &= marks - 1;
      const int64_t begin = uniform64(p.row_ptr[row]);
      const int len = (int)(uniform64(p.row_ptr[row + 1]) - begin);
      if (len <= 0) continue;
      // ---- preconditioner: the fp32 system and its factor, as als_persistent_kernel<T, 2, 0> builds them
      f32x4 acc[tri(T)];
      init_acc<T>(p, acc, lane);
      float bpart[T];
#pragma unroll
      for (int u = 0; u < T; ++u) bpart[u] = 0.f;
      {
        Pipe<T, 2> pp;
        pp.wcur = pp.cbcur = 0.f;
        pp.colpf = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int u = 0; u < T; ++u) pp.y[i][u] = 0.f;
        pp.ch = chunk_issue(p, begin, len, 0, lane);
        chunk_weights(p, pp.ch);
        prime_row<T, 2, true>(p, lane, pp);
        gather_row<T, 2, true>(p, begin, len, lane, pp, acc, bpart);
      }
      float bcol[T], xcol[T];
#pragma unroll
      for (int u = 0; u < T; ++u) bcol[u] = reduce_groups(bpart[u], lane);
      add_ridge<T, false>(p, acc, len, lane);
      float minpiv = 3.0e38f;
      cholesky_tiles<T>(acc, lane, minpiv);
      if (!(minpiv > p.sing_threshold)) {  // no usable fp32 factor: the fp64 restatement takes the row
        if (lane == 0) p.refine_flag[row] = 2;
        continue;
      }
      solve_tiles<T>(acc, bcol, xcol, lane);
      // ---- preconditioned conjugate gradients on the exact system, from x0
      double x[T], r[T], pd[T], wp[T];
#pragma unroll
      for (int u = 0; u < T; ++u) x[u] = (double)xcol[u];
      refine_matvec<T>(q, begin, len, lane, sx[wv], x, true, r);  // r = b - W x0
      float rf[T], zf[T];
#pragma unroll
      for (int u = 0; u < T; ++u) rf[u] = (float)r[u];
      solve_tiles<T>(acc, rf, zf, lane);
#pragma unroll
      for (int u = 0; u < T; ++u) pd[u] = (double)zf[u];
      double rz = refine_dot<T>(r, pd);
      bool converged = !(rz > 0.0);
      for (int it = 0; it < 12 && rz > 0.0; ++it) {
        refine_matvec<T>(q, begin, len, lane, sx[wv], pd, false, wp);  // wp = -W p
        const double pwp = -refine_dot<T>(pd, wp);
        if (!(pwp > 0.0)) break;
        const double a = rz / pwp;
        double smax = 0.0, xmax = 0.0;
#pragma unroll
        for (int u = 0; u < T; ++u) {
          x[u] = fma(a, pd[u], x[u]);
          r[u] = fma(a, wp[u], r[u]);
          smax = fmax(smax, fabs(a * pd[u]));
          xmax = fmax(xmax, fabs(x[u]));
        }
        for (int off = 8; off > 0; off >>= 1) {
          smax = fmax(smax, __shfl_xor(smax, off));
          xmax = fmax(xmax, __shfl_xor(xmax, off));
        }
        if (!(smax > 1e-6 * xmax)) {  // uniform: every lane group holds the same vectors
          converged = true;
          break;
        }
#pragma unroll
        for (int u = 0; u < T; ++u) rf[u] = (float)r[u];
        solve_tiles<T>(acc, rf, zf, lane);
        double z[T];
#pragma unroll
        for (int u = 0; u < T; ++u) z[u] = (double)zf[u];
        const double rz2 = refine_dot<T>(r, z);
        const double beta = rz2 / rz;
        rz = rz2;
#pragma unroll
        for (int u = 0; u < T; ++u) pd[u] = fma(beta, pd[u], z[u]);
      }
      if (!converged && lane == 0) p.refine_flag[row] = 2;  // the factor is no preconditioner: fp64 restatement
      if (lane < 16) {
        float* o = p.out + row * p.k;
#pragma unroll
        for (int u = 0; u < T; ++u) {
          const int feat = 16 * u + lane;
          if (feat < p.k) o[feat] = (float)x[u];
        }
      }
      ++done;
    }
  }
  if (lane == 0 && done) atomicAdd(q.n_refined, done);
}

// ------------------------------------------------------------------------------------------------
// Last resort: the reference's arithmetic, restated on the device in fp64, one workgroup per row with the k x k
// system in LDS (k = 128: 132 KB of the CU's 160).  For the rows on which the fp32 factorization gives no usable
// answer at all -- a pivot at or below the singularity threshold, which at cond(W) ~ 1e7 and beyond is rounding
// noise and not a verdict (measured: lossIgnoresUnspecified with lambda = 0.01 and factor rows of norm 30, the
// reference solves such rows, fp32 calls them singular) -- and, under lossIgnoresUnspecified, for every marked
// row: there W starts from the row's own sum of (float)(y_r y_c) (ALS:524-539 rounds each product to fp32, like
// MU:232), which conjugate gradients on the exact system cannot reproduce and this kernel does, product for
// product.  Then W += (y_r (c_u - 1)) y_c in fp64 in the reference's order (ALS:474-477), the ridge (ALS:488),
// an LDL^T in fp64 -- the verdict "singular" is given HERE (pivot <= threshold), or by mals_check's pivoted QR
// for the smallest-pivot suspect -- and x cast to fp32 (CMS:40-42).  Slow (tens of microseconds per row and CU);
// rows of level `level` and above in refine_flag: 1 = marked ill-conditioned, 2 = no usable fp32 factor.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void als_exact_kernel(RefineParams q, int level) {
  const SolveParams& p = q.p;
  extern __shared__ double lds[];
  const int k = p.k, ld = k + 1;
  double* W = lds;                       // k x (k + 1)
  double* bv = W + (size_t)k * ld;       // k
  double* yd = bv + k;                   // k: the entry's factor row, widened
  float* yf = reinterpret_cast<float*>(yd + k);   // k: the same in fp32
  int* list = reinterpret_cast<int*>(yf + k);     // up to 256 marked rows of the block + count
  __shared__ int n_marked;
  __shared__ double s_piv, s_minpiv;
  const int tid = threadIdx.x;
  unsigned long long done = 0;
  for (int64_t blk = q.row_begin + 256 * (int64_t)blockIdx.x; blk < q.row_end; blk += 256 * (int64_t)gridDim.x) {
    if (tid == 0) n_marked = 0;
    __syncthreads();
    if (blk + tid < q.row_end && p.refine_flag[blk + tid] >= level) list[atomicAdd(&n_marked, 1)] = tid;
    __syncthreads();
    const int nm = n_marked;
    for (int mi = 0; mi < nm; ++mi) {
      const int64_t row = blk + list[mi];   // arrival order: the rows are independent
      const int64_t begin = p.row_ptr[row];
      const int64_t len = p.row_ptr[row + 1] - begin;
      const double* G = q.gref_state[1] ? q.Gref : q.G;
      for (int e = tid; e < k * k; e += 256) {
        const int r = e / k, c = e - r * k;
        W[r * ld + c] = (p.flags & 2) ? 0.0 : G[(int64_t)r * k + c];   // ALS:447-450
      }
      if (tid < k) bv[tid] = 0.0;
      __syncthreads();
      for (int64_t en = 0; en < len; ++en) {
        const int col = p.col[begin + en];
        const double xu = (double)p.val[begin + en];
        if (tid < k) {
          const float y = p.M[(uint64_t)(uint32_t)col * (uint32_t)p.ldm + (uint32_t)tid];
          yf[tid] = y;
          yd[tid] = (double)y;
        }
        __syncthreads();
        const double cu1 = (p.flags & 1) ? 0.0 : q.alpha * fabs(xu);   // c_u - 1 (ALS:471)
        for (int e = tid; e < k * k; e += 256) {
          const int r = e / k, c = e - r * k;
          double w = W[r * ld + c];
          if (p.flags & 2) w += (double)(yf[r] * yf[c]);            // ALS:524-539: the product is rounded to fp32
          if (!(p.flags & 1)) w += (yd[r] * cu1) * yd[c];            // ALS:474-477
          W[r * ld + c] = w;
        }
        if (tid < k) {
          if (p.flags & 1) {
            bv[tid] += xu * yd[tid];                                  // ALS:466-469
          } else if (xu > 0.0) {
            bv[tid] += yd[tid] * (1.0 + cu1);                         // ALS:480-482
          }
        }
        __syncthreads();
      }
      if (tid < k) W[tid * ld + tid] += q.lambda_alpha * (double)len;  // ALS:488
      if (tid == 0) s_minpiv = 1.0e300;
      __syncthreads();
      // LDL^T, right-looking; column j keeps d_j l_ij = the entries as they stood when j was eliminated
      for (int j = 0; j < k; ++j) {
        if (tid == 0) {
          s_piv = W[j * ld + j];
          s_minpiv = s_piv < s_minpiv || !(s_piv == s_piv) ? s_piv : s_minpiv;
        }
        __syncthreads();
        const double inv_d = 1.0 / s_piv;
        const int m = k - j - 1;
        for (int e = tid; e < m * m; e += 256) {
          const int a = j + 1 + e / m, b2 = j + 1 + e % m;
          W[a * ld + b2] -= W[a * ld + j] * W[b2 * ld + j] * inv_d;
        }
        __syncthreads();
      }
      const double minpiv = s_minpiv;
      const bool bad = !(minpiv > (double)p.sing_threshold);
      // forward: z = L^-1 b (in place), then z_j / d_j, then backward with L^T
      for (int j = 0; j < k; ++j) {
        const double zj = bv[j];
        const double inv_d = 1.0 / W[j * ld + j];
        __syncthreads();
        if (tid > j && tid < k) bv[tid] -= W[tid * ld + j] * inv_d * zj;
        __syncthreads();
      }
      if (tid < k) bv[tid] /= W[tid * ld + tid];
      __syncthreads();
      for (int j = k - 1; j > 0; --j) {
        const double xj = bv[j];
        __syncthreads();
        if (tid < j) bv[tid] -= W[j * ld + tid] / W[tid * ld + tid] * xj;
        __syncthreads();
      }
      if (tid < k) p.out[row * k + tid] = bad ? 0.f : (float)bv[tid];
      if (tid == 0) {
        if (bad) {
          atomicMin(p.bad_row, (unsigned long long)row);
        } else if (minpiv <= 1024.0 * (double)p.sing_threshold) {
          const unsigned long long key = ((unsigned long long)__float_as_uint((float)minpiv) << 32) | (unsigned)row;
          if (key < __builtin_nontemporal_load(p.suspect)) atomicMin(p.suspect, key);
        }
        ++done;
      }
      __syncthreads();
    }
    __syncthreads();
  }
  if (tid == 0 && done) atomicAdd(q.n_refined, done);
}

// ------------------------------------------------------------------------------------------------
// K1: G = M^T M, fp64 accumulate on the fp64 matrix cores.  Each wave owns a contiguous slab of
// rows, walks it 4 rows per step (lane (g,c): row r0+g, features 16v+c), and keeps the T(T+1)/2
// upper tiles in fp64 accumulators; partials are summed in a fixed order by the finalize kernel,
// so the result is deterministic.  (The reference rounds each product to fp32 before widening,
// MU:232; the exact fp64 product used here differs from that by < 2^-24 relative per term.)
template <int T>
__global__ __launch_bounds__(256) void gramian_partial_kernel(const float* __restrict__ M, int64_t n_rows, int k,
                                                              int64_t rows_per_wave, double* __restrict__ partial) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t r0 = wave * rows_per_wave;
  int64_t r1 = r0 + rows_per_wave;
  if (r1 > n_rows) r1 = n_rows;
  f64x4 acc[tri(T)];
#pragma unroll
  for (int t = 0; t < tri(T); ++t) acc[t] = f64x4{0., 0., 0., 0.};
  for (int64_t r = r0; r < r1; r += 4) {
    const int64_t row = r + g;
    const bool ok = row < r1;
    const float* p = M + (ok ? row : r0) * k;
    double y[T];
#pragma unroll
    for (int v = 0; v < T; ++v) {
      const int f = 16 * v + c;
      const float x = p[f < k ? f : k - 1];
      y[v] = (ok && f < k) ? (double)x : 0.0;
    }
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int j = i; j < T; ++j)
        acc[tidx(T, i, j)] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[i], y[j], acc[tidx(T, i, j)], 0, 0, 0);
  }
  double* o = partial + wave * (int64_t)(tri(T) * 4 * 64) + lane;
#pragma unroll
  for (int t = 0; t < tri(T); ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[(t * 4 + r) * 64] = acc[t][r];
}

// 64 (tile, reg, lane) elements per workgroup: each of the 4 waves sums a contiguous quarter of the
// wave partials in order (coalesced 512-byte reads), the quarters are then added in order -- a fixed
// summation tree, so the result is deterministic.
// first stage for the many slab partials of gramian_split_kernel: group q of `groups` sums its share of the slabs
// (fixed order) into one set of doubles, element by element, coalesced -- thousands of workgroups instead of the 144 of
// the finalize kernel
__global__ __launch_bounds__(256) void gramian_reduce_slabs_kernel(const float* __restrict__ partial, int64_t n_slabs, int elems,
                                                                   int groups, double* __restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int q = blockIdx.y;
  if (e >= elems) return;
  const int64_t per = (n_slabs + groups - 1) / groups;
  const int64_t w0 = q * per, w1 = (q + 1) * per < n_slabs ? (q + 1) * per : n_slabs;
  double acc = 0.0;
  for (int64_t w = w0; w < w1; ++w) acc += (double)partial[w * (int64_t)elems + e];
  out[(int64_t)q * elems + e] = acc;
}

// PARTS threads per element, each summing a contiguous range of the partials in order; the PARTS sums are then added in
// order -- a fixed association whatever the launch.  PARTS = 16 for the fp64 kernel's per-wave partials (up to 1024 of
// them: with 4 ranges and 40 workgroups the finalize pass cost as much as the Gramian itself on C2).
template <int T, bool F64_LAYOUT, int PARTS = 4>
__global__ __launch_bounds__(256) void gramian_finalize_kernel(const double* __restrict__ partial, int64_t n_waves, int k,
                                                               double* __restrict__ G, float* __restrict__ Gf) {
  constexpr int EPB = 256 / PARTS;     // elements per workgroup
  __shared__ double part[PARTS][EPB];
  const int q = threadIdx.x / EPB, ln = threadIdx.x % EPB;
  const int e = blockIdx.x * EPB + ln;  // tri(T)*256 elements, a multiple of 64
  const int64_t per = (n_waves + PARTS - 1) / PARTS;
  const int64_t w_end = (q + 1) * per < n_waves ? (q + 1) * per : n_waves;
  double acc = 0.0;
  for (int64_t w = q * per; w < w_end; ++w) acc += partial[w * (int64_t)(tri(T) * 256) + e];
  part[q][ln] = acc;
  __syncthreads();
  if (q != 0) return;
  double s = part[0][ln];
#pragma unroll
  for (int i = 1; i < PARTS; ++i) s += part[i][ln];
  const int t = e >> 8, reg = (e >> 6) & 3, lane = e & 63;
  // decode tile (i,j) from t
  int i = 0, rem = t;
  while (rem >= T - i) {
    rem -= T - i;
    ++i;
  }
  const int j = i + rem;
  // C/D layout of the partials: f64 instruction: row (lane>>4) + 4 reg; f32-accumulating instructions: row 4 (lane>>4) + reg
  const int row = F64_LAYOUT ? 16 * i + (lane >> 4) + 4 * reg : 16 * i + 4 * (lane >> 4) + reg;
  const int col = 16 * j + (lane & 15);
  // A diagonal tile holds both (a,b) and (b,a).  The fp64 instruction computes them bit-identically; with split
  // operands the three partial products reach the two in a different order (an ulp apart): only the upper element
  // writes, to both places -- no two threads ever store different values to one address.
  if (i == j && row > col) return;
  if (G && row < k && col < k) {
    G[(int64_t)row * k + col] = s;
    G[(int64_t)col * k + row] = s;
  }
  if (Gf) {
    // fp32 image in the f32 MFMA acc layout, [tile][lane][reg]: element (row, col) of tile (i,j)
    // lives at lane 16*((row%16)/4) + col%16, reg (row%16)%4.  Diagonal tiles are stored in full.
    const float f = (row < k && col < k) ? (float)s : 0.f;
    const int lr = row - 16 * i, lc = col - 16 * j;
    Gf[(t * 64 + 16 * (lr >> 2) + lc) * 4 + (lr & 3)] = f;
    if (i == j) Gf[(t * 64 + 16 * (lc >> 2) + lr) * 4 + (lc & 3)] = f;
  }
}

// K1 as the reference rounds it: G[r][c] = sum_i (double)(float)(M[i][r] M[i][c]) (MU:219-239: the product of two
// floats is a float in Java, widened afterwards).  VALU work (a multiply, a conversion and an fp64 add per product:
// ~4 ms for 10M x 64), so it runs only where it can matter: when a row of the half-iteration has been marked
// ill-conditioned (state[0], set by the solving kernels) and the matrix is not there yet (state[1]).  Those rows'
// answers move by more than 1e-4 with the rounding of these products (cond(W) ~ 1e7; sweep case 2550), and parity is
// with the reference's arithmetic.  Per-workgroup partial sums, combined in workgroup order by the last to arrive.
// state[3] = "run": the go / no-go of the NEXT gramian_ref_kernel launch, decided once.  With the chunks of a half-
// iteration on two alternating streams the other stream's solving kernels may set state[0] while this launch's
// workgroups are still being dispatched: were every workgroup to read state[0] itself, some would return and some
// compute, the arrival ticket would never reach gridDim - 1 and stay non-zero, and the next launch would sum stale
// partials.  One thread latches the decision in front of the launch (same stream), every workgroup reads the latch.
__global__ void gramian_ref_latch_kernel(int* state) { state[3] = (state[0] != 0 && state[1] == 0) ? 1 : 0; }

template <int T>
__global__ __launch_bounds__(256) void gramian_ref_kernel(const float* __restrict__ M, int64_t n_rows, int k, int* state,
                                                          double* __restrict__ part, double* __restrict__ Gref) {
  if (state[3] == 0) return;
  constexpr int KP = 16 * T, NE = T * T;   // KP^2 / 256 entries per thread
  __shared__ float srow[8][KP];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  double acc[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) acc[j] = 0.0;
  const int64_t per = (n_rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = per * blockIdx.x, r1 = r0 + per < n_rows ? r0 + per : n_rows;
  for (int64_t base = r0; base < r1; base += 8) {
    __syncthreads();
    for (int e = tid; e < 8 * KP; e += 256) {
      const int i = e / KP, f = e - i * KP;
      srow[i][f] = (base + i < r1 && f < k) ? M[(base + i) * k + f] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int e = tid + 256 * j, r = e / KP, c = e - r * KP;
        acc[j] += (double)__fmul_rn(srow[i][r], srow[i][c]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NE; ++j) part[(size_t)blockIdx.x * (KP * KP) + tid + 256 * j] = acc[j];
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(&state[2], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int e = tid + 256 * j, r = e / KP, c = e - r * KP;
    double sum = 0.0;
    for (unsigned w = 0; w < gridDim.x; ++w) sum += part[(size_t)w * (KP * KP) + e];
    if (r < k && c < k) Gref[(int64_t)r * k + c] = sum;
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    state[2] = 0;
    state[1] = 1;
  }
}

// K1 for LARGE matrices on the f16 matrix pipe.  The fp64 instruction above costs 64 cycles per 16x16x4 block and
// is 8 % of a k = 128 iteration; v_mfma_f32_16x16x16_f16 on operands split into two f16 halves (hi = the value rounded
// to f16, lo = the exact residual rounded to f16: 22 significand bits, unbiased, every product exact) does a
// 16x16x16 block in 3 x 16 cycles.  What fp64 bought -- no rounding in the sum -- is kept where it matters: a wave
// accumulates in fp32 only over its slab of rows_per_slab rows (512 .. 2048: gramian_slab_rows in mals_api.hip), the slab partials
// are summed in fp64 in a fixed order by gramian_finalize_kernel.  Per slab the fp32 sum carries <= 6e-8 x sqrt(roundings) relative (random; 48 .. 192 roundings);
// over the hundreds of slabs this kernel is used for that averages far below the reference's own product rounding
// (MU:232 rounds every product to fp32), and stays at 3e-7 even when a handful of rows dominate G.  Small matrices keep the fp64 kernel (launch_gramian).
// Operand scale: a power of two per wave, lowered (with an exact rescale of the accumulators) whenever a 16-row
// step brings a larger |value| than any before it -- no bound is needed from outside and an outlier row costs
// the rows after it a few low bits relative to sums it already dominates.
// Layout: lane (g,c) holds, for rows r0 + 4g + s (s = 0..3) and every 16-block v, feature 16v + c -- the A and the B
// operand of the instruction at once (contraction over the 16 rows of the step).
// E = rows per lane and step (a step = 4 E rows = one MFMA contraction): 8 -> v_mfma_f32_16x16x32_f16 for T <= 4 (both
// instructions take 16 cycles, so the x32 one halves the matrix-pipe time, and the per-step maximum / rescale logic runs half
// as often; round 5), 4 -> v_mfma_f32_16x16x16_f16 where two buffers of 8 raw rows per tile do not fit next to the accumulators
// (T = 6 spilled 188 bytes with E = 8).
template <int T>
__global__ __launch_bounds__(256, 2) void gramian_split_kernel(const float* __restrict__ M, int64_t n_rows, int k,
                                                               int64_t rows_per_slab, float* __restrict__ partial,
                                                               unsigned* __restrict__ ymax) {
  constexpr int E = T <= 4 ? 8 : 4, STEP = 4 * E;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t slab = uniform64((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
  const int64_t r0 = slab * rows_per_slab;
  int64_t r1 = r0 + rows_per_slab;
  if (r1 > n_rows) r1 = n_rows;
  f32x4 acc[tri(T)];
#pragma unroll
  for (int t = 0; t < tri(T); ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int pw = 100;  // current scale 2^pw: max |z| <= 2^14
  int slab_max = 0;  // bit pattern of the largest |element| of the slab (-> ymax: operand bound of the split-precision gather)
  // Loads ahead of their use (until round 4 a step waited for its own loads: 8 waves x 4 KB per CU in flight, 4.6 TB/s): a step's
  // raw registers are reloaded as soon as it has converted them, under its matrix instructions.  Steps past the end of the
  // slab read clamped rows and contribute zeros.
  // Addresses are 32-bit element offsets from the slab's (wave-uniform) base, one add per row of the step; rows and
  // features are clamped / zeroed only in the steps that need it (the last step of a ragged slab; the last 16-block
  // when k < 16 T).  Until the second half of round 5 every load carried a 64-bit row x k multiply and a clamp and every
  // element a select: 550 vector instructions per step against 30 matrix instructions, 4.8 TB/s.
  const int n_slab = (int)(r1 - r0);            // rows of this slab (uniform; <= 0 for the padding slabs of the last workgroup)
  const float* __restrict__ base = M + (n_slab > 0 ? r0 : 0) * k;
  const bool kfull = k == 16 * T;               // uniform
  int fo[T];                                    // this lane's feature of 16-block v, clamped into the row
#pragma unroll
  for (int v = 0; v < T; ++v) fo[v] = 16 * v + c < k ? 16 * v + c : k - 1;
  auto load_step = [&](int rl, float (&raw)[T][E]) {
    if (rl + STEP <= n_slab) {  // uniform: every row of the step exists
      // byte offsets: uniform base + 32-bit lane offset + the 16-block as the instruction's immediate
      const char* bb = reinterpret_cast<const char*>(base);
      const unsigned o0 = 4u * (unsigned)((rl + E * g) * k + c), last = 4u * (unsigned)(fo[T - 1] - c);
      if (kfull) {  // uniform
#pragma unroll
        for (int s4 = 0; s4 < E; ++s4) {
          const unsigned o = o0 + 4u * (unsigned)(s4 * k);
#pragma unroll
          for (int v = 0; v < T; ++v) raw[v][s4] = *reinterpret_cast<const float*>(bb + o + 64 * v);
        }
      } else {
#pragma unroll
        for (int s4 = 0; s4 < E; ++s4) {
          const unsigned o = o0 + 4u * (unsigned)(s4 * k);
#pragma unroll
          for (int v = 0; v < T - 1; ++v) raw[v][s4] = *reinterpret_cast<const float*>(bb + o + 64 * v);
          raw[T - 1][s4] = *reinterpret_cast<const float*>(bb + (o + last));
        }
      }
    } else {
#pragma unroll
      for (int s4 = 0; s4 < E; ++s4) {
        const int row = rl + E * g + s4;
        const unsigned o = (unsigned)((row < n_slab ? row : 0) * k);
#pragma unroll
        for (int v = 0; v < T; ++v) raw[v][s4] = base[o + (unsigned)fo[v]];
      }
    }
  };
  auto do_step = [&](int rl, float (&raw)[T][E], int reload) {
    if (rl + STEP > n_slab) {  // uniform: rows past the end of the slab contribute zeros
#pragma unroll
      for (int s4 = 0; s4 < E; ++s4)
        if (!(rl + E * g + s4 < n_slab))
#pragma unroll
          for (int v = 0; v < T; ++v) raw[v][s4] = 0.f;
    }
    if (!kfull) {  // uniform: so do the padding features of the last 16-block
      if (!(16 * (T - 1) + c < k))
#pragma unroll
        for (int s4 = 0; s4 < E; ++s4) raw[T - 1][s4] = 0.f;
    }
    float amax = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < E; ++s4)
#pragma unroll
      for (int v = 0; v < T; ++v) amax = fmaxf(amax, fabsf(raw[v][s4]));
    int m = __float_as_int(amax);  // non-negative floats order like their bit patterns
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    m = uniform(m);
    slab_max = max(slab_max, m);
    if (m != 0) {
      const int eb = ((m >> 23) & 255) - 126;  // step max < 2^eb
      const int want = 14 - eb;
      if (want < pw) {  // a larger value than any before: lower the scale, bring the sums along (exact)
        if (pw != 100) {
          int d2 = 2 * (want - pw);
          d2 = d2 < -120 ? -120 : d2;
          const float f = __int_as_float((d2 + 127) << 23);
#pragma unroll
          for (int t = 0; t < tri(T); ++t) acc[t] *= f;
        }
        pw = want;
      }
    }
    const int pwc = pw == 100 ? 0 : (pw < -100 ? -100 : (pw > 100 ? 100 : pw));
    const float sc = __int_as_float((pwc + 127) << 23);
    ZOp<E> zh[T], zl[T];
#pragma unroll
    for (int v = 0; v < T; ++v) {
#pragma unroll
      for (int pr = 0; pr < E / 2; ++pr) {
        const float z0 = raw[v][2 * pr] * sc, z1 = raw[v][2 * pr + 1] * sc;
        const int h01 = pk_rn16(z0, z1);
        zh[v].r[pr] = h01;
        zl[v].r[pr] = pk_rn16(residual_lo(h01, z0), residual_hi(h01, z1));
      }
    }
    // the raw registers are free: the step after the next (T <= 6: two buffers) or the next one (T = 7, 8: one buffer, the
    // accumulators leave no room for a second) is requested now and lands during the 3 tri(T) matrix instructions below
    __builtin_amdgcn_sched_barrier(0);
    load_step(reload, raw);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int j = i; j < T; ++j) acc[tidx(T, i, j)] = mfma_h<E>(zh[i], zh[j], acc[tidx(T, i, j)]);
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int j = i; j < T; ++j) acc[tidx(T, i, j)] = mfma_h<E>(zh[i], zl[j], acc[tidx(T, i, j)]);
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int j = i; j < T; ++j) acc[tidx(T, i, j)] = mfma_h<E>(zl[i], zh[j], acc[tidx(T, i, j)]);
  };
  if constexpr (T <= 6) {
    float ra[T][E], rb[T][E];
    load_step(0, ra);
    load_step(STEP, rb);
    for (int rl = 0; rl < n_slab; rl += 2 * STEP) {
      do_step(rl, ra, rl + 2 * STEP);
      do_step(rl + STEP, rb, rl + 3 * STEP);
    }
  } else {
    float ra[T][E];
    load_step(0, ra);
    for (int rl = 0; rl < n_slab; rl += STEP) do_step(rl, ra, rl + STEP);
  }
  int d2 = pw == 100 ? 0 : -2 * pw;
  d2 = d2 < -126 ? -126 : (d2 > 126 ? 126 : d2);
  const float back = __int_as_float((d2 + 127) << 23);
  float* o = partial + slab * (int64_t)(tri(T) * 4 * 64) + lane;
#pragma unroll
  for (int t = 0; t < tri(T); ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[(t * 4 + r) * 64] = acc[t][r] * back;
  if (ymax) {  // wave-uniform
    // one atomic per workgroup, spread over YMAX_SLOTS addresses: same-address atomics serialise (~0.1 us each) and the
    // waves of a launch finish together; the consumer takes the maximum of the slots
    __shared__ int s_max[4];
    if (lane == 0) s_max[threadIdx.x >> 6] = slab_max;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned m4 = (unsigned)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
      unsigned* slot = ymax + (blockIdx.x & (YMAX_SLOTS - 1));
      if (m4 > __builtin_nontemporal_load(slot)) atomicMax(slot, m4);
    }
  }
}

// G (k x k fp64 row-major) -> fp32 acc-layout image used by K2 (see gramian_finalize_kernel)
__global__ void gramian_pack_kernel(const double* __restrict__ G, int k, int T, float* __restrict__ Gf) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= tri(T) * 256) return;
  const int t = e >> 8, lane = (e >> 2) & 63, reg = e & 3;
  int i = 0, rem = t;
  while (rem >= T - i) {
    rem -= T - i;
    ++i;
  }
  const int j = i + rem;
  const int row = 16 * i + 4 * (lane >> 4) + reg, col = 16 * j + (lane & 15);
  Gf[e] = (row < k && col < k) ? (float)G[(int64_t)row * k + col] : 0.f;
}

// Split-precision gather: S = 2^p with  max|z| = sqrt(w_max) * S * max|y| <= 2^14.  max|y| is
// bounded by sqrt(max_f G_ff) (G = M^T M of the gathered factor matrix, always at hand) -- loose by up to
// sqrt(n_rows): 13 binades of the 16 the range flag allows at the 1e8 rows of C5's X -- or, when the split-f16
// Gramian kernel of this library made G (>= 262144 rows, where the looseness matters; it computes every step's
// max |element| for its own scaling anyway), by the exact max |y| it recorded (ymax; round 3; a word above 3e38 =
// "some part of G was formed without one": diagonal bound).  w_max = the largest |value| of the matrix side (max_abs_kernel at upload).
// out = {S, 1/S^2, range flag, bound on |y| used}.
// Range flag: a typical operand, sqrt(w_mean) * rms|y_f| (rms over the n_rows rows that make up G), sits
// log2(bound / typical) binades below the bound; both f16 halves of z keep all their bits while
// z S >= 2^-2, i.e. up to 16 binades.  Beyond that (an outlier value or factor row stretching the
// bound) the launch is NOT run in split precision: flag = 0 makes the split kernels return at once and
// the fp32-gather kernels enqueued behind them do the work (no host round trip).
__global__ void gather_scale_kernel(const double* __restrict__ G, int k, float sqrt_w_max, float sqrt_w_mean, double n_rows,
                                    int force_flag, const unsigned* __restrict__ ymax, float* __restrict__ out) {
  // one wave: the lanes fetch the diagonal (and the maximum's slots) side by side -- a single thread walking k dependent
  // loads took 13 us on the critical path of every half-iteration
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  double d = 0.0, tr = 0.0;
  for (int f = lane; f < k; f += 64) {
    const double g = G[(int64_t)f * k + f];
    d = fmax(d, g);
    tr += g;
  }
  unsigned bits = (ymax && lane < YMAX_SLOTS) ? ymax[lane] : 0u;
  static_assert(YMAX_SLOTS <= 64, "one slot per lane");
  for (int off = 32; off > 0; off >>= 1) {
    d = fmax(d, __shfl_xor(d, off));
    tr += __shfl_xor(tr, off);
    bits = max(bits, (unsigned)__shfl_xor((int)bits, off));
  }
  if (lane != 0) return;
  double ybound = sqrt(d);
  if (ymax) {
    const float ym = __uint_as_float(bits);
    if (ym > 0.f && ym < 3.0e38f && (double)ym < ybound) ybound = (double)ym;   // non-finite: keep the diagonal's verdict
  }
  out[3] = (float)ybound;
  const double bound = ybound * (double)sqrt_w_max;
  int e = 0;
  if (bound > 0.0 && bound < 1.0e300) {
    int eb;
    (void)frexp(bound, &eb);  // bound < 2^eb
    e = 14 - eb;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
  }
  out[0] = (float)ldexp(1.0, e);
  out[1] = (float)ldexp(1.0, -2 * e);
  const double typical = (double)sqrt_w_mean * sqrt(tr / ((double)k * fmax(n_rows, 1.0)));
  float ok = 1.f;
  if (typical > 0.0 && bound > 0.0 && bound / typical > 65536.0) ok = 0.f;
  if (!(bound < 1.0e300)) ok = 0.f;  // non-finite Gramian: leave it to the fp32 kernels
  if (force_flag >= 0) ok = (float)force_flag;
  out[2] = ok;
}

// max |v| over a value array, as the bit pattern of a non-negative float (atomicMax on uint), and
// sum |v| (for the mean: the dynamic-range check of the split-precision gather)
__global__ void max_abs_kernel(const float* __restrict__ v, int64_t n, unsigned* __restrict__ out, double* __restrict__ sum_out) {
  float m = 0.f;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = fabsf(v[i]);
    m = fmaxf(m, a);
    s += (double)a;
  }
  for (int off = 32; off > 0; off >>= 1) {
    m = fmaxf(m, __shfl_xor(m, off));
    s += __shfl_xor(s, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out, __float_as_uint(m));
    atomicAdd(sum_out, s);
  }
}

// {min, max} over a column-index array (matrix upload: indices must stay inside the opposite replica)
__global__ void col_range_kernel(const int32_t* __restrict__ col, int64_t n, int* __restrict__ out) {
  int lo = 0x7fffffff, hi = (int)0x80000000;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = col[i];
    lo = min(lo, c);
    hi = max(hi, c);
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off));
    hi = max(hi, __shfl_xor(hi, off));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(out, lo);
    atomicMax(out + 1, hi);
  }
}

// Reconstruction error over the stored entries of the local rows of R:
// sum over (u,i) of max(0, 1 - x_u . y_i), the quantity ReconstructionEvaluator averages
// (online/src/net/myrrix/online/eval/ReconstructionEvaluator.java:91-102), with the reference's dot
// (SimpleVectorMath.java:34-41: fp32 products, fp64 sum).  One wave per row, four entries per step
// like the fp32 gather; per-wave partial sums, summed in a fixed order by the caller.
template <int T>
__global__ __launch_bounds__(256) void reconstruction_kernel(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                             const float* __restrict__ X, const float* __restrict__ Y, int k,
                                                             int64_t n_rows, double* __restrict__ partial_sum,
                                                             unsigned long long* __restrict__ partial_cnt) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  double acc = 0.0;
  unsigned long long cnt = 0;
  for (int64_t row = wave; row < n_rows; row += n_waves) {
    const int64_t b = row_ptr[row], e = row_ptr[row + 1];
    float x[T];
#pragma unroll
    for (int v = 0; v < T; ++v) x[v] = 16 * v + c < k ? X[row * k + 16 * v + c] : 0.f;
    for (int64_t s = b; s < e; s += 4) {
      const bool ok = s + g < e;
      const float* y = Y + (int64_t)col[ok ? s + g : b] * k;
      double d = 0.0;
#pragma unroll
      for (int v = 0; v < T; ++v) {
        const float yv = 16 * v + c < k ? y[16 * v + c] : 0.f;
        d += (double)__fmul_rn(x[v], yv);  // the reference's float * float, widened before the sum
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) d += __shfl_xor(d, off);  // over the 16 lanes of the entry
      if (ok && c == 0) {
        acc += fmax(0.0, 1.0 - d);
        ++cnt;
      }
    }
  }
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  cnt += __shfl_xor(cnt, 16);
  cnt += __shfl_xor(cnt, 32);
  if (lane == 0) {
    partial_sum[wave] = acc;
    partial_cnt[wave] = cnt;
  }
}

// The convergence sample of call() (ALS:230-238): est[i][j] = SimpleVectorMath.dot(X[test_users[i]],
// Y[test_items[j]]) (SimpleVectorMath.java:34-41: float product, double sum, features in order), one thread
// per pair, from the resident factors -- bit-identical to the host loop it replaces.
__global__ void sample_dots_kernel(const float* __restrict__ X, const float* __restrict__ Y, const int64_t* __restrict__ users,
                                   const int64_t* __restrict__ items, int n_users, int n_items, int k, double* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_users * n_items) return;
  const float* x = X + users[e / n_items] * (int64_t)k;
  const float* y = Y + items[e % n_items] * (int64_t)k;
  double d = 0.0;
  for (int f = 0; f < k; ++f) d += (double)__fmul_rn(x[f], y[f]);
  out[e] = d;
}

// Gather table of the rows kernels when k % 16 != 0 (SolveParams::M): dst row r = src row r followed by zeros up to
// ld = 16 T floats.  One 16-byte store per thread; a streaming copy (n (k + ld) 4 bytes) once per half-iteration.
__global__ void pad_rows_kernel(const float* __restrict__ src, int64_t n, int k, int ld, float* __restrict__ dst) {
  const int q = ld >> 2;
  const int64_t n4 = n * q;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / q;
    const int f = 4 * (int)(e - r * q);
    const float* s = src + r * k + f;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f + i < k ? s[i] : 0.f;
    reinterpret_cast<f32x4*>(dst)[e] = v;
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ F, const int64_t* __restrict__ idx, int n, int k,
                                   float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * k) return;
  out[e] = F[idx[e / k] * (int64_t)k + (e % k)];
}

}  // namespace mals
