// lds_kernels.h -- the k = 128 rows / segments kernels with the gather staged through LDS (round 4).
//
// Why: at T = 8 the register-staged gather of als_kernels.h (gather_row_h) cannot hold a 32-entry super-step -- 144
// accumulator registers + 64 operand registers + 64 raw registers in flight -- and runs 16 entries per step on
// v_mfma_f32_16x16x16_f16: twice the matrix cycles per entry (36 tiles x 3 split products x 16 cycles per 16 entries) and
// half the prefetch distance.  Here the gathered factor rows never touch a VGPR on their way in: global_load_lds_dwordx4
// deposits them in a per-wave LDS ring (32 entries x 512 B), the conversion reads them back 16 bytes at a time, and the
// raw registers are gone -- 144 + 64 + ~40 fit, the super-step is 32 entries on v_mfma_f32_16x16x32_f16, and a whole
// super-step (16 KB per wave) is in flight while the previous one is multiplied.  Same arithmetic as gather_row_h
// (ALS:447-492: z = sqrt(w) S y split into two f16 halves, zh zh^T + zh zl^T + zl zh^T in fp32), same epilogue (K3).
//
// One wave per workgroup (nothing is shared between waves: the Gramian image comes from L2 once per row), two waves per
// SIMD, 17.5 KB of LDS per wave (8 waves = 140 of the CU's 160 KB).
//
// Feature order inside the kernel.  A gathered row lands in LDS as it lies in memory, 32 lanes x 16 bytes; the lane that
// converts features for the MFMA reads 32 contiguous bytes of it (two ds_read_b128) instead of eight dwords 64 bytes
// apart.  Lane (g, c) therefore owns features 8c .. 8c+7 of its entries, and "block v, lane c" of the tile algebra is
//     feature  f(v, c) = 8 c + (v ^ 4 (c >> 3))          (v = 0..7, c = 0..15)
// -- a fixed permutation of the 128 features (the xor swaps the two 16-byte halves for lanes 8..15, which makes every
// ds_read_b128 lane group hit 16 distinct 16-byte slots: conflict free).  A symmetric permutation of W x = b changes
// nothing but the order of the pivots: the Gramian image is laid out in the same order (gramian_perm_kernel), the ridge is
// a diagonal, the factorization and the solves never look at feature numbers, and the store undoes it (each lane writes
// 16 contiguous bytes of x).  Partial slots of long rows are in the permuted order too; their finish kernel is
// als_finish_kernel<8, true>.
//
// LDS ring and the entry <-> slot map.  A super-step is 32 entries n = 4 e + g (lane group g converts entries 4e+g,
// e = 0..7, as in gather_row_h: its weights sit in its own 16 lanes of the row-major chunk).  They are converted pair by
// pair (E2 = e >> 1: the two entries one v_cvt_pk_f16_f32 packs), and as soon as pair E2 -- 8 entries -- has been read, its 4 KB are
// refilled with the same pair of the NEXT super-step: four global_load_lds_dwordx4, instruction i covering entry
// (g = i, e' = 0) with lanes 0-31 and (g = i, e' = 1) with lanes 32-63 (the destination is lane-linear: M0 + 16 lane).  So
//     slot(n) = 8 E2 + 2 g + e'      (512 bytes each),
// and the four column indices a lane needs for a pair are 16 contiguous bytes of the chunk's column array.
//
// Completion.  LDS-DMA is counted by vmcnt and nothing else orders a ds_read behind it.  Steady state: when pair E2 of
// super-step s is read, the loads issued after its own are pairs E2+1..3 of s and pairs 0..E2-1 of s+1 -- always 12 --
// so "s_waitcnt vmcnt(12)" retires it (anything else issued in between only makes the wait stricter).  A row starts with
// vmcnt(0): its first super-step and the next row's first (col, value) chunk were issued before the previous row's
// factorization.  The (col, value) stream goes through LDS as well (global_load_lds_dword, 64 entries per instruction,
// three 512-byte buffers per wave: this chunk, the next of this row, the first of the next row), so no register ever
// waits for a load the compiler does not know about.
#pragma once
#include "als_kernels.h"

namespace mals {

constexpr int LDSK_T = 8, LDSK_E = 8;
#ifndef MALS_LDSK_RESPLIT
#define MALS_LDSK_RESPLIT 1
#endif
constexpr bool LDSK_RESPLIT = MALS_LDSK_RESPLIT != 0;   // cholesky_tiles: split the panel tiles where they are used (fewer registers) or once per block row
constexpr int LDSK_ROW_BYTES = 512;                          // one gathered factor row, k = 128
constexpr int LDSK_RING_BYTES = 32 * LDSK_ROW_BYTES;         // one super-step
constexpr int LDSK_CHUNK_BYTES = 512;                        // 64 column indices + 64 values
constexpr int LDSK_NCHUNK = 3;
constexpr int LDSK_WAVE_BYTES = LDSK_RING_BYTES + LDSK_NCHUNK * LDSK_CHUNK_BYTES;  // 17 920

// feature of (block v, lane c) -- see the header comment
__host__ __device__ constexpr int ldsk_feature(int v, int c) { return 8 * c + (v ^ (4 * (c >> 3))); }

typedef __attribute__((address_space(3))) char lds_char;

// ---- LDS-DMA -------------------------------------------------------------------------------------------------------
// one global_load_lds_dwordx4: 64 lanes x 16 bytes, destination dst + 16 lane (M0 is the compiler's: saved and restored
// inside the statement)
__device__ __forceinline__ void glds16(const void* a, unsigned dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(a), "s"(dst)
      : "memory");
}
// the (col, value) chunk: two global_load_lds_dword, 64 lanes x 4 bytes each, non-temporal (the entry stream is read once)
__device__ __forceinline__ void glds4x2(const void* a0, const void* a1, unsigned dst) {
  unsigned keep;
  const unsigned d1 = dst + 256;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, off nt\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, off nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(a0), "v"(a1), "s"(dst), "s"(d1)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef const __attribute__((address_space(3))) f32x4* lds_f4;
typedef const __attribute__((address_space(3))) i32x4* lds_i4;
typedef const __attribute__((address_space(3))) float* lds_f1;

// what a lane keeps about itself
struct LdskLane {
  unsigned rd0;        // LDS byte address of the lane's first 16-byte read of slot (g, e' = 0) of pair 0; the second is rd0 ^ 16
  const char* gsrc;    // gather table + the byte offset of the lane's 16 bytes inside a gathered row
};

// request chunk [base, base + 64) of a row into chunk buffer `buf` (LDS byte address); entries past the end repeat the
// row's last one, so every column in the buffer is a valid row of the table (their weights are zero: ldsk_weights)
__device__ __forceinline__ void ldsk_chunk_dma(const SolveParams& p, int64_t begin, int len, int base, unsigned buf, int lane) {
  const int n = base + lane;
  const int nn = n < len ? n : len - 1;
  glds4x2(p.col + begin + nn, p.val + begin + nn, buf);
}

// weights of chunk buffer `buf` (its DMA has landed), row-major lane order: lane 16 g + m <- entry 4 m + g
__device__ __forceinline__ void ldsk_weights(const SolveParams& p, const lds_char* lds, unsigned buf, int base, int len, int lane, float zscale,
                                             float& w, float& cb) {
  const int widx = 4 * (lane & 15) + (lane >> 4);
  Chunk e;
  e.col = 0;
  e.w = *reinterpret_cast<lds_f1>(lds + buf + 256 + 4 * widx);
  e.cb = base + widx < len ? 1.f : 0.f;
  chunk_weights_h(p, e, zscale);
  w = e.w;
  cb = e.cb;
}

// the four loads of pair E2 of a super-step: columns at LDS byte address cols_at (a chunk buffer's column array, + 128
// for its second half; + 16 for lanes 32-63: ldsk_cols_addr)
template <int E2>
__device__ __forceinline__ void ldsk_issue_pair(const lds_char* lds, unsigned ring, unsigned cols_at, const LdskLane& L) {
  const i32x4 cols = *reinterpret_cast<lds_i4>(lds + cols_at + 32 * E2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // gsrc + 512 col as ONE v_mad_u64_u32 (hipcc strength-reduces the constant product into v_mov + v_lshlrev_b64 +
    // v_lshl_add_u64: three half-rate instructions per load)
    const char* a;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(a) : "v"(cols[i]), "s"((unsigned)LDSK_ROW_BYTES), "v"(L.gsrc) : "vcc");
    glds16(a, ring + 4096 * E2 + 1024 * i);
  }
}
__device__ __forceinline__ unsigned ldsk_cols_addr(unsigned buf_cols, int lane) { return buf_cols + 16 * (unsigned)(lane >> 5); }

// Half a pair: the two entries (g, e' = 0 / 1) of pair E2, four features each (blocks V0 .. V0 + 3) -> scaled, split f16
// operands in contraction slots 2 E2, 2 E2 + 1; RHS partial sums in fp32 from the raw values.  The arithmetic of
// convert_pair_h (als_kernels.h), value for value.
template <int E2, int V0>
__device__ __forceinline__ void ldsk_convert_half(const f32x4& ya, const f32x4& yb, float s0, float s1, float c0, float c1,
                                                  ZOp<LDSK_E> (&zh)[LDSK_T], ZOp<LDSK_E> (&zl)[LDSK_T], float (&bpart)[LDSK_T]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int v = V0 + q;
    const float y0 = ya[q], y1 = yb[q];
    const float z0 = y0 * s0, z1 = y1 * s1;
    const int hp = pk_rn16(z0, z1);
    const f16x2 hh = __builtin_bit_cast(f16x2, hp);
    zh[v].r[E2] = hp;
    zl[v].r[E2] = pk_rn16(fmaf((float)hh[0], -1.f, z0), fmaf((float)hh[1], -1.f, z1));
    bpart[v] = fmaf(c0, y0, bpart[v]);
    bpart[v] = fmaf(c1, y1, bpart[v]);
  }
}

// One super-step: convert the 32 entries in the ring -- their weights in lanes 0..7 of every 16-lane row of (w, cb) --,
// refilling every pair's slots with the same pair of the next super-step (columns at next_cols, ldsk_cols_addr), then the
// 108 matrix instructions.
template <int E2>
__device__ __forceinline__ void ldsk_pair(const lds_char* lds, unsigned ring, float w, float cb, unsigned next_cols, const LdskLane& L,
                                          ZOp<LDSK_E> (&zh)[LDSK_T], ZOp<LDSK_E> (&zl)[LDSK_T], float (&bpart)[LDSK_T]) {
  wait_vm<12>();
  const float s0 = row_bcast<2 * E2>(w), s1 = row_bcast<2 * E2 + 1>(w);
  const float c0 = row_bcast<2 * E2>(cb), c1 = row_bcast<2 * E2 + 1>(cb);
  {
    const f32x4 ya = *reinterpret_cast<lds_f4>(lds + L.rd0 + 4096 * E2);
    const f32x4 yb = *reinterpret_cast<lds_f4>(lds + L.rd0 + 4096 * E2 + 512);
    ldsk_convert_half<E2, 0>(ya, yb, s0, s1, c0, c1, zh, zl, bpart);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    const f32x4 ya = *reinterpret_cast<lds_f4>(lds + (L.rd0 ^ 16u) + 4096 * E2);
    const f32x4 yb = *reinterpret_cast<lds_f4>(lds + (L.rd0 ^ 16u) + 4096 * E2 + 512);
    ldsk_convert_half<E2, 4>(ya, yb, s0, s1, c0, c1, zh, zl, bpart);
  }
  __builtin_amdgcn_sched_barrier(0);
  ldsk_issue_pair<E2>(lds, ring, next_cols, L);   // the slots just read are free: the same pair of the next super-step
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void ldsk_super_step(const lds_char* lds, unsigned ring, float w, float cb, unsigned next_cols, const LdskLane& L,
                                                f32x4 (&acc)[tri(LDSK_T)], float (&bpart)[LDSK_T]) {
  ZOp<LDSK_E> zh[LDSK_T], zl[LDSK_T];
  ldsk_pair<0>(lds, ring, w, cb, next_cols, L, zh, zl, bpart);
  ldsk_pair<1>(lds, ring, w, cb, next_cols, L, zh, zl, bpart);
  ldsk_pair<2>(lds, ring, w, cb, next_cols, L, zh, zl, bpart);
  ldsk_pair<3>(lds, ring, w, cb, next_cols, L, zh, zl, bpart);
  gram_super_step<LDSK_T, LDSK_E>(zh, zl, acc);
  __builtin_amdgcn_sched_barrier(0);
}

// image of G for the permuted feature order, acc layout [upper tile][lane][reg]
__global__ void gramian_perm_kernel(const double* __restrict__ G, int k, float* __restrict__ Gp) {
  constexpr int T = LDSK_T;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // tri(T) * 256 elements
  if (e >= tri(T) * 256) return;
  const int t = e >> 8, lane = (e >> 2) & 63, reg = e & 3;
  int i = 0, rem = t;
  while (rem >= T - i) {
    rem -= T - i;
    ++i;
  }
  const int j = i + rem;
  const int row = ldsk_feature(i, 4 * (lane >> 4) + reg), col = ldsk_feature(j, lane & 15);
  Gp[e] = (row < k && col < k) ? (float)G[(int64_t)row * k + col] : 0.f;
}

// Lists A (MODE 0) and B (MODE 1) at k = 128.  Launch: one wave per workgroup, grid = persistent waves.
template <int MODE>
__global__ __launch_bounds__(64, 2) void als_lds_kernel_h(SolveParams p) {
  constexpr int T = LDSK_T;
  __shared__ __attribute__((aligned(1024))) char lds_raw[LDSK_WAVE_BYTES];
  const lds_char* lds = (const lds_char*)lds_raw;
  const unsigned lbase = (unsigned)(uintptr_t)lds;   // LDS byte address of the wave's block (0: the only array)
  const int lane = threadIdx.x;
  const int64_t wave = blockIdx.x, n_waves = gridDim.x;
  int64_t it = wave;
  if (it >= p.n_work) return;
  if (p.zscale[2] == 0.f) return;  // operand range too wide for the f16 split: the fp32 kernels behind this launch run
  const float zscale = __int_as_float(uniform(__float_as_int(p.zscale[0])));
  const float inv_s2 = __int_as_float(uniform(__float_as_int(p.zscale[1])));
  LdskLane L;
  {
    const int g = lane >> 4, c = lane & 15;
    L.rd0 = (unsigned)(g * 1024 + 32 * c + 16 * (c >> 3));
    L.gsrc = reinterpret_cast<const char*>(p.M) + 16 * (lane & 31);
  }
  const unsigned ring = lbase;
  // chunk buffers (byte offsets inside the wave's block): this chunk, the next of this row, the first of the next row
  unsigned b_cur = LDSK_RING_BYTES, b_next = LDSK_RING_BYTES + LDSK_CHUNK_BYTES, b_nrow = LDSK_RING_BYTES + 2 * LDSK_CHUNK_BYTES;
  const f32x4* Gp4 = reinterpret_cast<const f32x4*>(p.Gperm);   // uniform: the lane goes into the index (scalar base + 32-bit offset loads)
  WorkItem cur = load_item(p, it);
  WorkItem nxt = load_item(p, it + n_waves);
  WorkItem nx2 = load_item(p, it + 2 * n_waves);
  if (cur.len > 0) {
    ldsk_chunk_dma(p, cur.begin, cur.len, 0, lbase + b_cur, lane);
    if (nxt.len > 0) ldsk_chunk_dma(p, nxt.begin, nxt.len, 0, lbase + b_nrow, lane);
    wait_vm<0>();
    {
      const unsigned ca = ldsk_cols_addr(b_cur, lane);
      ldsk_issue_pair<0>(lds, ring, ca, L);
      ldsk_issue_pair<1>(lds, ring, ca, L);
      ldsk_issue_pair<2>(lds, ring, ca, L);
      ldsk_issue_pair<3>(lds, ring, ca, L);
    }
    for (;;) {
      const WorkItem nx3 = load_item(p, it + 3 * n_waves);
      f32x4 acc[tri(T)];
#pragma unroll
      for (int t = 0; t < tri(T); ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      float bpart[T];
#pragma unroll
      for (int v = 0; v < T; ++v) bpart[v] = 0.f;
#ifdef MALS_PROFILING   // per-phase cycle stamps of 64 rows of the first 64 waves (MALS_DEBUG_TRACE=<first traced row>)
      const int64_t trow = it / n_waves - p.trace_start;
      const bool tr = MODE == 0 && p.trace && wave < 64 && trow >= 0 && trow < 64;
      unsigned long long t0 = 0, t1 = 0, t2 = 0;
      if (tr) t0 = __builtin_readcyclecounter();
#endif
      // the row's first super-step and the next row's first chunk were requested before the previous row's epilogue
      wait_vm<0>();
      const int n_ss = (cur.len + 31) >> 5;
      float w = 0.f, cb = 0.f;
      for (int ss = 0; ss < n_ss; ++ss) {
        const bool odd = ss & 1, last = ss + 1 == n_ss;
        if (!odd) {  // a new 64-entry chunk (its DMA landed: at least 16 row loads were issued and retired behind it)
          if (ss > 0) {
            const unsigned t = b_cur;
            b_cur = b_next;
            b_next = t;
          }
          ldsk_weights(p, lds, b_cur, 32 * ss, cur.len, lane, zscale, w, cb);
          if (32 * ss + 64 < cur.len) ldsk_chunk_dma(p, cur.begin, cur.len, 32 * ss + 64, lbase + b_next, lane);
        } else {     // second half of the chunk: its weights into lanes 0..7 of every row
          w = row_ror<8>(w);
          cb = row_ror<8>(cb);
        }
        // what the ring is refilled with: the other half of this chunk, the next chunk, or the next row's first super-step
        // (no next row: this chunk's first rows again -- valid columns, never used)
        const unsigned src = last ? (nxt.len > 0 ? b_nrow : b_cur) : (odd ? b_next : b_cur + 128);
        ldsk_super_step(lds, ring, w, cb, ldsk_cols_addr(src, lane), L, acc, bpart);
      }
#ifdef MALS_PROFILING
      if (tr) t1 = __builtin_readcyclecounter();
#endif
      float bcol[T];
#pragma unroll
      for (int v = 0; v < T; ++v) bcol[v] = reduce_groups(bpart[v], lane);
      // this row's chunk buffers are free: the first chunk of the row after the next
      {
        const unsigned t = b_cur;
        b_cur = b_nrow;
        b_nrow = t;
      }
      if (nx2.len > 0) ldsk_chunk_dma(p, nx2.begin, nx2.len, 0, lbase + b_nrow, lane);
      if (MODE == 0) {
        const float rmax = p.refine_flag ? __int_as_float(max_entry_bits<T>(acc, lane)) * inv_s2 : 0.f;
        // back to the unscaled system (S^2 is a power of two: exact), on top of the shared Gramian (image from L2)
        // (the image -- all zeros under lossIgnoresUnspecified: the host provides that -- is read nine tiles at a time: all 36
        // at once would be 144 registers in flight next to the accumulators)
        {
          int l2 = lane;
          asm volatile("" : "+v"(l2));   // (the 36 addresses computed here, not hoisted out of the row loop and spilled)
#pragma unroll
          for (int t0 = 0; t0 < tri(T); t0 += 9) {
            f32x4 g4[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) g4[t] = Gp4[(t0 + t) * 64 + l2];
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[t0 + t][r] = fmaf(acc[t0 + t][r], inv_s2, g4[t][r]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        add_ridge<T, true>(p, acc, cur.len, lane);
        float minpiv = 3.0e38f, wmax;
        float xcol[T];
        const float inv_s2row = row_scale<T>(acc, bcol, lane, wmax);
        cholesky_tiles<T, true, LDSK_RESPLIT, true>(acc, lane, minpiv);
        minpiv *= inv_s2row;
#ifdef MALS_PROFILING
        if (tr) t2 = __builtin_readcyclecounter();
#endif
        solve_tiles<T>(acc, bcol, xcol, lane);
        store_row<T, true>(p, xcol, minpiv, fmaxf(rmax, p.gramian_weight * wmax), cur.id, lane);
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
      it += n_waves;
      if (cur.len <= 0) break;
    }
    wait_vm<0>();   // the loads requested for a row that never came must not land in another workgroup's LDS
  }
  if (MODE == 0) {
    while (cur.len == 0) {  // empty rows: W = G (+ 0 ridge), b = 0
      f32x4 acc[tri(T)];
#pragma unroll
      for (int t = 0; t < tri(T); ++t) acc[t] = Gp4[t * 64 + lane];
      float bcol[T];
#pragma unroll
      for (int v = 0; v < T; ++v) bcol[v] = 0.f;
      finish_row<T, true>(p, acc, bcol, 0, cur.id, lane);
      it += n_waves;
      cur = load_item(p, it);
    }
  }
}

}  // namespace mals
