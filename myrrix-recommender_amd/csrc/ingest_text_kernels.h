// ingest_text_kernels.h -- gfx950 kernels of the TEXT half of the ingest path:
// the bytes of the input files -> lines -> (user id, item id, value | NaN) records in file order, i.e. the loop
// InputFilesReader.readInputFiles runs over FileLineIterable (online-local/src/net/myrrix/online/generation/
// InputFilesReader.java:92-158, common/src/net/myrrix/common/iterator/FileLineIterator.java:104-114).
//
// HBM-bound byte work, nothing shaped into a GEMM:
//   line_count_kernel / line_starts_kernel   16 bytes per thread (one dwordx4 load), a line starts after '\n', after
//       a '\r' that is not followed by '\n', and at offset 0 (java.io.BufferedReader.readLine); two passes around one
//       prefix sum of the per-workgroup counts
//   parse_lines_kernel      one thread per line, the FAST instantiation of csrc/text_parse.h (ASCII numeric lines:
//       the bulk of any real file); the line's bytes come through an 8-byte register window, so a 20-byte line costs
//       three or four aligned 8-byte loads instead of twenty byte loads; adjacent threads read adjacent lines
//   parse_deferred_kernel   the FULL instantiation (tags -> MD5, non-ASCII whitespace and digits, malformed UTF-8,
//       hexadecimal floats, > 19-digit significands at a rounding boundary) for the lines the fast parser handed on
//   line_summary_kernel     record flags for the stable compaction + the counters the host needs
//   compact_records_kernel  records in file order appended to the ingest object's record arrays
//   collect_tags_kernel     hashed tag ids (IFR:159-165), only launched when a block has any
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ingest_kernels.h"
#include "text_parse.h"

namespace mals {

constexpr int LT_BYTES_PER_THREAD = 16;
constexpr int LT_BLOCK_BYTES = 256 * LT_BYTES_PER_THREAD;

// counters of one block of text (device, copied back once per block)
struct TextCounters {
  unsigned records, bad, fatal, header, skipped, deferred, user_tags, item_tags;
};

// 0x80 in every byte of x that equals c (exact: no borrow crosses a byte)
__device__ __forceinline__ uint64_t byte_eq(uint64_t x, uint8_t c) {
  x ^= 0x0101010101010101ull * c;
  return ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);
}
// bit 8j+7 of lo (j < 8) / hi (j >= 8): byte j of this thread's 16 starts a line.  prev = the byte in front of them.
// A line starts after '\n', and after a '\r' that is not followed by '\n' (java.io.BufferedReader.readLine).
__device__ __forceinline__ void line_start_words(uint64_t w0, uint64_t w1, uint8_t prev, uint64_t& lo, uint64_t& hi) {
  const uint64_t n0 = byte_eq(w0, '\n'), n1 = byte_eq(w1, '\n'), r0 = byte_eq(w0, '\r'), r1 = byte_eq(w1, '\r');
  const uint64_t pn0 = (n0 << 8) | (prev == '\n' ? 0x80ull : 0ull), pn1 = (n1 << 8) | (n0 >> 56);
  const uint64_t pr0 = (r0 << 8) | (prev == '\r' ? 0x80ull : 0ull), pr1 = (r1 << 8) | (r0 >> 56);
  lo = pn0 | (pr0 & ~n0);
  hi = pn1 | (pr1 & ~n1);
}
// the thread's 16 bytes at p0 and the byte in front (the last byte of the previous lane's 16, or of the previous wave's)
__device__ __forceinline__ void line_start_bits(const uint8_t* __restrict__ text, int64_t n, int64_t p0, uint64_t& lo, uint64_t& hi) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (p0 < n) v = *reinterpret_cast<const uint4*>(text + p0);  // the buffer is padded to a multiple of 16
  const uint64_t w0 = ((uint64_t)v.y << 32) | v.x, w1 = ((uint64_t)v.w << 32) | v.z;
  unsigned last = __shfl_up(v.w >> 24, 1);
  if ((threadIdx.x & 63) == 0) last = p0 > 0 && p0 <= n ? text[p0 - 1] : (unsigned)'\n';  // offset 0 starts a line
  line_start_words(w0, w1, (uint8_t)last, lo, hi);
  // only bytes below n count
  const int64_t left = n - p0;
  if (left < 16) {
    const uint64_t keep0 = left <= 0 ? 0ull : (left >= 8 ? ~0ull : ((1ull << (8 * left)) - 1ull));
    const uint64_t keep1 = left <= 8 ? 0ull : ((1ull << (8 * (left - 8))) - 1ull);
    lo &= keep0;
    hi &= keep1;
  }
}

__global__ __launch_bounds__(256) void line_count_kernel(const uint8_t* __restrict__ text, int64_t n, unsigned* __restrict__ counts) {
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * LT_BYTES_PER_THREAD;
  uint64_t lo, hi;
  line_start_bits(text, n, p0, lo, hi);
  unsigned c = (unsigned)(__popcll(lo) + __popcll(hi));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  __shared__ unsigned ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void line_starts_kernel(const uint8_t* __restrict__ text, int64_t n,
                                                          const unsigned* __restrict__ block_offsets, unsigned* __restrict__ starts) {
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * LT_BYTES_PER_THREAD;
  uint64_t lo, hi;
  line_start_bits(text, n, p0, lo, hi);
  unsigned at = block_offsets[blockIdx.x] + block_exclusive_scan_256((unsigned)(__popcll(lo) + __popcll(hi)));
  while (lo) {
    starts[at++] = (unsigned)(p0 + ((__ffsll((unsigned long long)lo) - 1) >> 3));
    lo &= lo - 1;
  }
  while (hi) {
    starts[at++] = (unsigned)(p0 + 8 + ((__ffsll((unsigned long long)hi) - 1) >> 3));
    hi &= hi - 1;
  }
}

// the line's bytes through an aligned 8-byte window held in registers
struct WindowSrc {
  const uint8_t* p;
  mutable uint64_t w;
  mutable uint32_t base;  // multiple of 8; 0xffffffff = nothing loaded
  __device__ __forceinline__ uint8_t operator()(uint32_t i) const {
    const uint32_t b = i & ~7u;
    if (b != base) {
      w = *reinterpret_cast<const uint64_t*>(p + b);
      base = b;
    }
    return (uint8_t)(w >> (8 * (i & 7u)));
  }
};

// the same over the workgroup's LDS copy of the text span that starts at text offset `origin` (a multiple of 16)
struct LdsWindowSrc {
  const uint8_t* lds;
  uint32_t origin;
  mutable uint64_t w;
  mutable uint32_t base;
  __device__ __forceinline__ uint8_t operator()(uint32_t i) const {
    const uint32_t b = (i - origin) & ~7u;
    if (b != base) {
      w = *reinterpret_cast<const uint64_t*>(lds + b);
      base = b;
    }
    return (uint8_t)(w >> (8 * (i & 7u)));
  }
};

// [s, e) of line i without its terminator
__device__ __forceinline__ void line_span(const uint8_t* __restrict__ text, const unsigned* __restrict__ starts, int64_t n_lines,
                                          unsigned region_end, int64_t i, unsigned& s, unsigned& e) {
  s = starts[i];
  e = (i + 1 < n_lines) ? starts[i + 1] : region_end;
  if (e > s && text[e - 1] == '\n') --e;
  if (e > s && text[e - 1] == '\r') --e;
}

__device__ __forceinline__ void store_parsed(const text::Parsed& r, int64_t i, uint8_t* __restrict__ status, int64_t* __restrict__ user,
                                             int64_t* __restrict__ item, uint32_t* __restrict__ value) {
  status[i] = (uint8_t)(r.status | (r.flags << 4));
  user[i] = r.user;
  item[i] = r.item;
  value[i] = r.value_bits;
}

// ---- the shape nearly every line of a real input file has: digits ',' digits [',' plain decimal] ------------------
// Parsed a token at a time instead of a byte at a time: the line's words come out of LDS together, commas are found
// with exact SWAR zero-byte tests, a token is fetched right-aligned with two aligned 8-byte reads and a funnel shift,
// checked to be eight-or-fewer digits with the nibble test and converted four digits per 32-bit lane with 24-bit
// multiply-adds.  The value is w / 10^f with w < 10^7 < 2^24 and f <= 7: both exact in binary32, so ONE IEEE division
// is the correctly rounded result (Clinger's fast path).  Anything else about the line -- whitespace, signs on ids,
// exponents, tags, more than three columns, longer tokens, errors -- returns false and the byte-wise parser of
// text_parse.h takes the line from the start; both give the same answer where both apply (tests/test_gpu_ingest_text).
__device__ __forceinline__ uint64_t lds_u64(const uint8_t* lds, uint32_t off) { return *reinterpret_cast<const uint64_t*>(lds + off); }
// the 8 bytes lds[o, o + 8) for any o
__device__ __forceinline__ uint64_t lds_unaligned_u64(const uint8_t* lds, uint32_t o) {
  const uint32_t a = o & ~7u, sh = (o & 7u) * 8u;
  const uint64_t r0 = lds_u64(lds, a), r1 = lds_u64(lds, a + 8u);
  return sh ? (r0 >> sh) | (r1 << (64u - sh)) : r0;
}
// v = n (1..8) ASCII digits right-aligned in the high bytes, zeros below: their value, or -1 if one is not a digit
__device__ __forceinline__ int32_t digits8(uint64_t v, uint32_t n) {
  const uint64_t fill = n >= 8u ? 0ull : (0x3030303030303030ull >> (8u * n));   // '0' in the bytes the token does not reach
  v |= fill;
  if (((v & 0xF0F0F0F0F0F0F0F0ull) | (((v + 0x0606060606060606ull) & 0xF0F0F0F0F0F0F0F0ull) >> 4)) != 0x3333333333333333ull) return -1;
  uint32_t lo = (uint32_t)v & 0x0F0F0F0Fu, hi = (uint32_t)(v >> 32) & 0x0F0F0F0Fu;   // lo: the four more significant digits
  lo = ((lo << 3) + (lo << 1) + (lo >> 8)) & 0x00FF00FFu;    // d0*10 + d1 | d2*10 + d3
  hi = ((hi << 3) + (hi << 1) + (hi >> 8)) & 0x00FF00FFu;
  lo = (lo & 0xFFu) * 100u + (lo >> 16);
  hi = (hi & 0xFFu) * 100u + (hi >> 16);
  return (int32_t)(lo * 10000u + hi);
}
// token lds[a, b), 1..16 chars, all digits -> value; false otherwise
__device__ __forceinline__ bool digits16(const uint8_t* lds, uint32_t a, uint32_t b, int64_t* out) {
  const uint32_t n = b - a;
  if (n - 1u > 15u) return false;
  const uint32_t n_lo = n > 8u ? 8u : n;
  uint64_t v = lds_unaligned_u64(lds, b - 8u);
  if (n_lo < 8u) v &= ~0ull << (8u * (8u - n_lo));
  const int32_t lo = digits8(v, n_lo);
  if (lo < 0) return false;
  if (n <= 8u) {
    *out = lo;
    return true;
  }
  uint64_t u = lds_unaligned_u64(lds, b - 16u);
  const uint32_t n_hi = n - 8u;
  if (n_hi < 8u) u &= ~0ull << (8u * (8u - n_hi));
  const int32_t hi = digits8(u, n_hi);
  if (hi < 0) return false;
  *out = (int64_t)((uint64_t)(uint32_t)hi * 100000000ull + (uint32_t)lo);
  return true;
}

constexpr int PL_LDS_BYTES = 16384;  // text bytes a workgroup stages
constexpr int PL_LDS_FRONT = 16;     // zeros in front (a right-aligned fetch of a token at the very start reaches back 15 bytes)
constexpr int PL_LDS_BACK = 48;      // readable behind (the five words of a line, the windows)

// lds[A, A + len) = the line.  true: r is the record.
__device__ __forceinline__ bool simple_line(const uint8_t* lds, uint32_t A, uint32_t len, text::Parsed& r) {
  if (len - 3u > 29u) return false;  // 3..32 bytes
  const uint32_t w0 = A & ~7u, end = A + len;
  uint32_t n_commas = 0, p1 = 0, p2 = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const uint32_t wb = w0 + 8u * k;
    uint64_t x = lds_u64(lds, wb) ^ 0x2C2C2C2C2C2C2C2Cull;
    uint64_t z = ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);   // 0x80 where the byte is ','
    // bytes of this word inside the line
    const uint32_t lo = wb < A ? A - wb : 0u;                       // first valid byte (only word 0 can be clipped below)
    const uint32_t hi = end > wb ? (end - wb > 8u ? 8u : end - wb) : 0u;   // one past the last valid byte
    uint64_t valid = hi > lo ? ((hi >= 8u ? ~0ull : ((1ull << (8u * hi)) - 1ull)) & (~0ull << (8u * lo))) : 0ull;
    z &= valid;
    if (z) {
      const uint32_t c = (uint32_t)__popcll(z);
      if (n_commas == 0) {
        p1 = wb + ((uint32_t)__ffsll((unsigned long long)z) - 1u) / 8u;
        if (c > 1) {
          const uint64_t z2 = z & (z - 1);
          p2 = wb + ((uint32_t)__ffsll((unsigned long long)z2) - 1u) / 8u;
        }
      } else if (n_commas == 1) {
        p2 = wb + ((uint32_t)__ffsll((unsigned long long)z) - 1u) / 8u;
      }
      n_commas += c;
    }
  }
  if (n_commas - 1u > 1u) return false;  // one or two commas
  int64_t user, item;
  const uint32_t t1_end = n_commas == 2 ? p2 : end;
  if (!digits16(lds, A, p1, &user) || !digits16(lds, p1 + 1u, t1_end, &item)) return false;
  uint32_t bits = 0x3f800000u;  // absent: 1.0f (IFR:136)
  if (n_commas == 2) {
    const uint32_t vb = p2 + 1u, vn = end - vb;
    if (vn == 0) {
      bits = 0x7fc00000u;  // empty: NaN = remove (IFR:134)
    } else {
      if (vn > 8u) return false;
      uint64_t v = lds_unaligned_u64(lds, vb);
      uint32_t left = vn;
      bool neg = false;
      if ((v & 0xFF) == '-') {
        neg = true;
        v >>= 8;
        --left;
      }
      uint32_t w = 0, n_dig = 0;
      float scale = 1.f;
      bool point = false;
      for (; left; --left, v >>= 8) {
        const uint32_t c = (uint32_t)(v & 0xFF);
        if (c - '0' <= 9u) {
          w = w * 10u + (c - '0');
          ++n_dig;
          if (point) scale *= 10.f;
        } else if (c == '.' && !point) {
          point = true;
        } else {
          return false;
        }
      }
      if (n_dig == 0 || n_dig > 7u) return false;
      const float f = (float)w / scale;  // correctly rounded: both operands exact, one IEEE division
      bits = __float_as_uint(f) | (neg ? 0x80000000u : 0u);
    }
  }
  r.user = user;
  r.item = item;
  r.value_bits = bits;
  r.flags = 0;
  r.status = text::ST_RECORD;
  return true;
}

// One workgroup = 256 consecutive lines.  Their bytes are one contiguous span of the text: it is copied into LDS
// with coalesced 16-byte loads (all in flight together) and parsed out of LDS.  A span that does not fit (lines
// averaging > 64 bytes) is parsed straight from global memory by the byte-wise parser.
__global__ __launch_bounds__(256) void parse_lines_kernel(const uint8_t* __restrict__ text, const unsigned* __restrict__ starts,
                                                          int64_t n_lines, unsigned region_end, int first_line_is_first,
                                                          uint8_t* __restrict__ status, int64_t* __restrict__ user,
                                                          int64_t* __restrict__ item, uint32_t* __restrict__ value,
                                                          unsigned* __restrict__ defer_list, TextCounters* __restrict__ counters) {
  __shared__ __attribute__((aligned(16))) uint8_t stage_all[PL_LDS_FRONT + PL_LDS_BYTES + PL_LDS_BACK];
  uint8_t* stage = stage_all + PL_LDS_FRONT;
  const int64_t l0 = (int64_t)blockIdx.x * 256;
  const int64_t l1 = l0 + 256 < n_lines ? l0 + 256 : n_lines;
  const unsigned span_b = starts[l0] & ~15u;                         // aligned down: the copy is 16 bytes per lane
  const unsigned span_e = l1 < n_lines ? starts[l1] : region_end;
  const bool staged = span_e - span_b <= (unsigned)PL_LDS_BYTES;
  if (staged) {
    if (threadIdx.x < PL_LDS_FRONT / 4) reinterpret_cast<uint32_t*>(stage_all)[threadIdx.x] = 0u;
    for (unsigned o = threadIdx.x * 16u; span_b + o < span_e + (unsigned)PL_LDS_BACK; o += 256u * 16u)   // the text buffer is padded
      *reinterpret_cast<uint4*>(stage + o) = *reinterpret_cast<const uint4*>(text + span_b + o);
    __syncthreads();
  }
  const int64_t i = l0 + threadIdx.x;
  if (i >= n_lines) return;
  const unsigned s = starts[i];
  unsigned e = (i + 1 < n_lines) ? starts[i + 1] : region_end;
  text::Parsed r;
  if (staged) {
    if (e > s && stage[e - 1 - span_b] == '\n') --e;
    if (e > s && stage[e - 1 - span_b] == '\r') --e;
    if (!simple_line(stage_all, (uint32_t)PL_LDS_FRONT + (s - span_b), e - s, r)) {
      LdsWindowSrc src{stage, span_b, 0, 0xffffffffu};
      r = text::parse_line<false>(src, s, e, first_line_is_first && i == 0);
    }
  } else {
    if (e > s && text[e - 1] == '\n') --e;
    if (e > s && text[e - 1] == '\r') --e;
    WindowSrc src{text, 0, 0xffffffffu};
    r = text::parse_line<false>(src, s, e, first_line_is_first && i == 0);
  }
  if (r.status == text::ST_DEFER) {
    defer_list[atomicAdd(&counters->deferred, 1u)] = (unsigned)i;
    return;
  }
  store_parsed(r, i, status, user, item, value);
}

__global__ __launch_bounds__(64) void parse_deferred_kernel(const uint8_t* __restrict__ text, const unsigned* __restrict__ starts,
                                                            int64_t n_lines, unsigned region_end, int first_line_is_first,
                                                            uint8_t* __restrict__ status, int64_t* __restrict__ user,
                                                            int64_t* __restrict__ item, uint32_t* __restrict__ value,
                                                            const unsigned* __restrict__ defer_list,
                                                            const TextCounters* __restrict__ counters) {
  const unsigned n = counters->deferred;
  for (unsigned j = blockIdx.x * 64 + threadIdx.x; j < n; j += gridDim.x * 64) {
    const int64_t i = defer_list[j];
    unsigned s, e;
    line_span(text, starts, n_lines, region_end, i, s, e);
    const text::PtrSrc src{text};
    const text::Parsed r = text::parse_line<true>(src, s, e, first_line_is_first && i == 0);
    store_parsed(r, i, status, user, item, value);
  }
}

// Stable compaction of the record lines in two passes over tiles of 2048 lines: (1) records per tile + the block's
// counters, (2) after a prefix sum of the tile counts, every tile ranks its own records and writes them out.
constexpr int CT_TILE = 2048;  // 256 threads x 8 lines
__global__ __launch_bounds__(256) void line_summary_kernel(const uint8_t* __restrict__ status, int64_t n_lines,
                                                           unsigned* __restrict__ tile_records, TextCounters* __restrict__ counters) {
  const int64_t b = (int64_t)blockIdx.x * CT_TILE + threadIdx.x * 8;
  unsigned c[7] = {0, 0, 0, 0, 0, 0, 0};
  if (b < n_lines) {
    const uint64_t w = *reinterpret_cast<const uint64_t*>(status + b);  // the array is padded to a multiple of 8
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (b + j >= n_lines) break;
      const int x = (int)((w >> (8 * j)) & 0xFF), st = x & 15, fl = x >> 4;
      c[0] += st == text::ST_RECORD;
      c[1] += st == text::ST_BAD;
      c[2] += st == text::ST_FATAL;
      c[3] += st == text::ST_HEADER;
      c[4] += st == text::ST_SKIP;
      c[5] += st == text::ST_RECORD && (fl & text::FL_USER_TAG);
      c[6] += st == text::ST_RECORD && (fl & text::FL_ITEM_TAG);
    }
  }
  __shared__ unsigned ws[4][7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    unsigned v = c[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const unsigned v = ws[0][threadIdx.x] + ws[1][threadIdx.x] + ws[2][threadIdx.x] + ws[3][threadIdx.x];
    if (threadIdx.x == 0) tile_records[blockIdx.x] = v;
    // seven addresses for the whole grid: only the classes that are rare may use an atomic per tile
    if (threadIdx.x >= 1 && v) {
      unsigned* dst = threadIdx.x == 1 ? &counters->bad : threadIdx.x == 2 ? &counters->fatal : threadIdx.x == 3 ? &counters->header
                    : threadIdx.x == 4 ? &counters->skipped : threadIdx.x == 5 ? &counters->user_tags : &counters->item_tags;
      atomicAdd(dst, v);
    }
  }
}

__global__ __launch_bounds__(256) void compact_records_kernel(const uint8_t* __restrict__ status, const unsigned* __restrict__ tile_offsets,
                                                              int64_t n_lines, const int64_t* __restrict__ user,
                                                              const int64_t* __restrict__ item, const uint32_t* __restrict__ value,
                                                              int64_t* __restrict__ user_out, int64_t* __restrict__ item_out,
                                                              float* __restrict__ value_out) {
  // eight rounds of 256 consecutive lines, one line per thread: reads and writes stay coalesced
  unsigned base = tile_offsets[blockIdx.x];
#pragma unroll 1
  for (int j = 0; j < CT_TILE / 256; ++j) {
    const int64_t i = (int64_t)blockIdx.x * CT_TILE + j * 256 + threadIdx.x;
    const bool rec = i < n_lines && (status[i] & 15) == text::ST_RECORD;
    unsigned total;
    const unsigned at = base + block_exclusive_scan(rec ? 1u : 0u, &total);
    if (rec) {
      user_out[at] = user[i];
      item_out[at] = item[i];
      value_out[at] = __uint_as_float(value[i]);
    }
    base += total;
  }
}

// IFR:159-165: a tag in the user column is an "item tag" id, a tag in the item column a "user tag" id.  Order does
// not matter (the ids are sorted and made unique when the ingest is finished).
__global__ void collect_tags_kernel(const uint8_t* __restrict__ status, int64_t n_lines, const int64_t* __restrict__ user,
                                    const int64_t* __restrict__ item, int64_t* __restrict__ item_tag_out,
                                    int64_t* __restrict__ user_tag_out, unsigned* __restrict__ cursors) {
  MALS_GRID_STRIDE(i, n_lines) {
    const uint8_t st = status[i];
    if ((st & 15) != text::ST_RECORD) continue;
    if ((st >> 4) & text::FL_USER_TAG) item_tag_out[atomicAdd(&cursors[0], 1u)] = user[i];
    if ((st >> 4) & text::FL_ITEM_TAG) user_tag_out[atomicAdd(&cursors[1], 1u)] = item[i];
  }
}

// sorted unique ids out of a sorted key array
__global__ void unique_ids_kernel(const uint64_t* __restrict__ keys, const unsigned* __restrict__ head, const unsigned* __restrict__ head_scan,
                                  int64_t n, int64_t* __restrict__ out) {
  MALS_GRID_STRIDE(i, n)
    if (head[i]) out[head_scan[i]] = key_to_id(keys[i]);
}
__global__ void ids_to_keys_kernel(const int64_t* __restrict__ ids, int64_t n, uint64_t* __restrict__ keys, unsigned* __restrict__ pay) {
  MALS_GRID_STRIDE(i, n) {
    keys[i] = id_to_key(ids[i]);
    pay[i] = 0;
  }
}

// knownItemIDs (IFR:173-191): the pairs that are present at the end of the stream, pruned or not
__global__ void compact_known_kernel(const uint64_t* __restrict__ keys, const unsigned* __restrict__ present,
                                     const unsigned* __restrict__ present_scan, int64_t n, const unsigned* __restrict__ new_u,
                                     const unsigned* __restrict__ new_i, int32_t* __restrict__ row, int32_t* __restrict__ col) {
  MALS_GRID_STRIDE(i, n) {
    if (!present[i]) continue;
    const unsigned p = present_scan[i];
    row[p] = (int32_t)new_u[(unsigned)(keys[i] >> 32)];
    col[p] = (int32_t)new_i[(unsigned)(keys[i] & 0xffffffffu)];
  }
}

}  // namespace mals
