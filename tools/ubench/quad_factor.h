// K3a for four rows at a time: the diagonal tiles of the four waves of a workgroup factored together by ONE of them.
//
// factor_diag (als_kernels.h) eliminates one 16x16 tile spread over all 64 lanes: 16 steps, each headed by a
// ds_bpermute + v_readlane round trip and carrying ~10 instructions of per-step bookkeeping next to ~5 DPP FMAs --
// 240 instructions and ~1.1K issue cycles per tile, 40-45 % of a short row's fixed cost at every rank.  Four tiles
// side by side turn the same wave into four 16-lane groups that each own a whole tile: lane (t, c) holds ROW c of
// tile t, all 16 columns (d[0..15]) and the same row of the inverse factor (e[0..15]).  Step m then needs nothing from
// another lane group: the multiplier's numerator d[c][m] is this lane's own register m, the pivot row is lane m of
// this group (a DPP row_newbcast operand of the FMA), 1/pivot is v_rcp of register m broadcast the same way.  No LDS
// round trip inside the chain, the bookkeeping (rcp, multiplier, two masks) is paid once per step for four tiles:
// 20 instructions per step and batch = 80 per tile instead of 240.
//
// The four tiles come from the four waves of a workgroup, which walk their rows in lockstep: at block step kb every
// wave writes its diagonal tile into the exchange buffer (read through the symmetry of D: lane (g,c) of the
// accumulator layout holds D[c][4g..4g+3], one ds_write_b128), the workgroup meets at a barrier, wave kb & 3 reads
// the four tiles (lane (t,c): row c of tile t, 4 x ds_read_b128), factors them and writes the four inverse factors back
// in place, second barrier, every wave reads its own U_kk^-1 back in accumulator layout (row c of L^-1, columns
// 4g..4g+3 = Uinv[4g+r][c]: one ds_read_b128) together with its tile's pivots.  Rows of 20 floats: both the 64 B-per-lane
// and the 16 B-per-lane accesses fall on distinct 16-byte bank slots.
//
// Pivot order is the natural one (0..15): with a lane owning a whole row, register m is finished after step m whatever
// the order.  e[m] is born at step m as the multiplier itself (the identity's column m has a single 1, in row m), so the
// inverse half costs m FMAs at step m, the D half 15 - m.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mals {

typedef __attribute__((address_space(3))) float qf_lds_float;
typedef float qf_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) qf_f32x4 qf_lds_f32x4;

constexpr int QF_ROW_FLOATS = 20;                       // 16 + 4 of padding: conflict-free b128 rows
constexpr int QF_TILE_FLOATS = 16 * QF_ROW_FLOATS;      // 320
constexpr int QF_PIV_OFFSET = 4 * QF_TILE_FLOATS;       // 64 pivots behind the four tiles
constexpr int QF_FLOATS = QF_PIV_OFFSET + 64;
constexpr int QF_BYTES = QF_FLOATS * 4;                 // 5 376 B per workgroup

__host__ __device__ constexpr unsigned long long qf_mask_eq(int m) { return 0x0001000100010001ull << m; }
__host__ __device__ constexpr unsigned long long qf_mask_gt(int m) {
  return ((0xffffull << (m + 1)) & 0xffffull) * 0x0001000100010001ull;
}

// one elimination step of the batch.  r = 1/d[m] (every lane's own; only lane m's is used) comes in, 1/d[m+1] goes out.
#define QF_D_FIRST(n) ".if %[m] + 1 == " #n "\n\tv_fmac_f32_dpp %[d" #n "], %[d" #n "], %[nl] row_newbcast:%[m] row_mask:0xf bank_mask:0xf\n\t" \
                      "v_rcp_f32_e32 %[r], %[d" #n "]\n\t.endif\n\t"
#define QF_D_REST(n) ".if %[m] + 1 < " #n "\n\tv_fmac_f32_dpp %[d" #n "], %[d" #n "], %[nl] row_newbcast:%[m] row_mask:0xf bank_mask:0xf\n\t.endif\n\t"
#define QF_E_NEW(n) ".if %[m] == " #n "\n\tv_cndmask_b32_e64 %[e" #n "], %[nl], 1.0, %[eq]\n\t.endif\n\t"
#define QF_E_OLD(n) ".if %[m] > " #n "\n\tv_fmac_f32_dpp %[e" #n "], %[e" #n "], %[nl] row_newbcast:%[m] row_mask:0xf bank_mask:0xf\n\t.endif\n\t"
template <int M>
__device__ __forceinline__ void qf_step(float (&d)[16], float (&e)[16], float& r, float& pk) {
  float nl;
  const unsigned long long eq = qf_mask_eq(M), gt = qf_mask_gt(M);
  // (DPP reads need two wait states after a VALU write of the register: r was written >= 14 instructions ago -- the
  // caller of step 0 provides the distance --, d[m+1..15] by the previous step's block, nl is never read through DPP)
  asm volatile("v_mul_f32_dpp %[nl], %[r], %[dm] row_newbcast:%[m] row_mask:0xf bank_mask:0xf\n\t"
               "v_cndmask_b32_e64 %[pk], %[pk], %[dm], %[eq]\n\t"
               "v_cndmask_b32_e64 %[nl], 0, -%[nl], %[gt]\n\t"
               QF_D_FIRST(1) QF_D_FIRST(2) QF_D_FIRST(3) QF_D_FIRST(4) QF_D_FIRST(5) QF_D_FIRST(6) QF_D_FIRST(7) QF_D_FIRST(8)
               QF_D_FIRST(9) QF_D_FIRST(10) QF_D_FIRST(11) QF_D_FIRST(12) QF_D_FIRST(13) QF_D_FIRST(14) QF_D_FIRST(15)
               QF_D_REST(2) QF_D_REST(3) QF_D_REST(4) QF_D_REST(5) QF_D_REST(6) QF_D_REST(7) QF_D_REST(8)
               QF_D_REST(9) QF_D_REST(10) QF_D_REST(11) QF_D_REST(12) QF_D_REST(13) QF_D_REST(14) QF_D_REST(15)
               : [nl] "=&v"(nl), [r] "+v"(r), [pk] "+v"(pk),
                 [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]), [d5] "+v"(d[5]), [d6] "+v"(d[6]), [d7] "+v"(d[7]),
                 [d8] "+v"(d[8]), [d9] "+v"(d[9]), [d10] "+v"(d[10]), [d11] "+v"(d[11]), [d12] "+v"(d[12]), [d13] "+v"(d[13]),
                 [d14] "+v"(d[14]), [d15] "+v"(d[15])
               : [dm] "v"(d[M]), [eq] "s"(eq), [gt] "s"(gt), [m] "n"(M));
  asm volatile(QF_E_NEW(0) QF_E_NEW(1) QF_E_NEW(2) QF_E_NEW(3) QF_E_NEW(4) QF_E_NEW(5) QF_E_NEW(6) QF_E_NEW(7)
               QF_E_NEW(8) QF_E_NEW(9) QF_E_NEW(10) QF_E_NEW(11) QF_E_NEW(12) QF_E_NEW(13) QF_E_NEW(14) QF_E_NEW(15)
               QF_E_OLD(0) QF_E_OLD(1) QF_E_OLD(2) QF_E_OLD(3) QF_E_OLD(4) QF_E_OLD(5) QF_E_OLD(6) QF_E_OLD(7)
               QF_E_OLD(8) QF_E_OLD(9) QF_E_OLD(10) QF_E_OLD(11) QF_E_OLD(12) QF_E_OLD(13) QF_E_OLD(14)
               : [e0] "+v"(e[0]), [e1] "+v"(e[1]), [e2] "+v"(e[2]), [e3] "+v"(e[3]), [e4] "+v"(e[4]), [e5] "+v"(e[5]), [e6] "+v"(e[6]),
                 [e7] "+v"(e[7]), [e8] "+v"(e[8]), [e9] "+v"(e[9]), [e10] "+v"(e[10]), [e11] "+v"(e[11]), [e12] "+v"(e[12]),
                 [e13] "+v"(e[13]), [e14] "+v"(e[14]), [e15] "+v"(e[15])
               : [nl] "v"(nl), [eq] "s"(eq), [m] "n"(M));
}
#undef QF_D_FIRST
#undef QF_D_REST
#undef QF_E_NEW
#undef QF_E_OLD

// the factoring wave's part: four tiles in, four inverse factors (rows scaled by 1/sqrt(pivot)) and the pivots out
__device__ __forceinline__ void qf_factor_batch(qf_lds_float* xb, int lane) {
  qf_lds_float* row = xb + lane * QF_ROW_FLOATS;   // lane (t,c) -> tile t, row c: (16 t + c) rows of 20 floats
  float d[16], e[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const qf_f32x4 v = *reinterpret_cast<qf_lds_f32x4*>(row + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) d[4 * q + r] = v[r];
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) e[j] = 0.f;
  float pk = 1.f;
  float r = __builtin_amdgcn_rcpf(d[0]);
  asm volatile("s_nop 1" : "+v"(r));   // the first step reads r through DPP
  qf_step<0>(d, e, r, pk);
  qf_step<1>(d, e, r, pk);
  qf_step<2>(d, e, r, pk);
  qf_step<3>(d, e, r, pk);
  qf_step<4>(d, e, r, pk);
  qf_step<5>(d, e, r, pk);
  qf_step<6>(d, e, r, pk);
  qf_step<7>(d, e, r, pk);
  qf_step<8>(d, e, r, pk);
  qf_step<9>(d, e, r, pk);
  qf_step<10>(d, e, r, pk);
  qf_step<11>(d, e, r, pk);
  qf_step<12>(d, e, r, pk);
  qf_step<13>(d, e, r, pk);
  qf_step<14>(d, e, r, pk);
  qf_step<15>(d, e, r, pk);
  const float s = __builtin_amdgcn_rsqf(pk);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    qf_f32x4 v;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) v[rr] = e[4 * q + rr] * s;
    *reinterpret_cast<qf_lds_f32x4*>(row + 4 * q) = v;
  }
  xb[QF_PIV_OFFSET + lane] = pk;
}

// Every wave of the workgroup calls this with its own diagonal tile (accumulator layout, full symmetric tile) at the same
// block step; `turn` (uniform over the workgroup) names the wave that factors.  Returns U^-1 in accumulator layout and
// folds the tile's pivots into the lane's minpiv (lane (g,c): pivot c -- reduce over the 16 lanes of a row at the end).
__device__ __forceinline__ qf_f32x4 factor_diag_quad(const qf_f32x4& D, int lane, int wave, int turn, float& minpiv_lane,
                                                     qf_lds_float* xb) {
  const int g = lane >> 4, c = lane & 15;
  qf_lds_float* mine = xb + (wave * 16 + c) * QF_ROW_FLOATS + 4 * g;
  *reinterpret_cast<qf_lds_f32x4*>(mine) = D;
  // lgkmcnt(0): the tile is in LDS before the barrier.  (Not __syncthreads: its fences would also drain vmcnt, i.e. wait
  // for the next row's gathers in flight.)  The "memory" clobber keeps the compiler from moving LDS accesses across.
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (wave == turn) qf_factor_batch(xb, lane);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  const qf_f32x4 U = *reinterpret_cast<qf_lds_f32x4*>(mine);
  minpiv_lane = fminf(minpiv_lane, xb[QF_PIV_OFFSET + wave * 16 + c]);
  return U;
}

}  // namespace mals
