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

// bit p of the result: byte p of this thread's 16 starts a line
__device__ __forceinline__ unsigned line_start_mask(const uint8_t* __restrict__ text, int64_t n, int64_t p0) {
  if (p0 >= n) return 0u;
  const uint4 v = *reinterpret_cast<const uint4*>(text + p0);  // the buffer is padded to a multiple of 16
  uint8_t prev = p0 > 0 ? text[p0 - 1] : (uint8_t)'\n';
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  unsigned mask = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint8_t c = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
    const bool start = prev == '\n' || (prev == '\r' && c != '\n');
    if (start && p0 + j < n) mask |= 1u << j;
    prev = c;
  }
  return mask;
}

__global__ __launch_bounds__(256) void line_count_kernel(const uint8_t* __restrict__ text, int64_t n, unsigned* __restrict__ counts) {
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * LT_BYTES_PER_THREAD;
  unsigned c = __popc(line_start_mask(text, n, p0));
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
  unsigned mask = line_start_mask(text, n, p0);
  unsigned at = block_offsets[blockIdx.x] + block_exclusive_scan_256(__popc(mask));
  while (mask) {
    const int j = __ffs(mask) - 1;
    mask &= mask - 1;
    starts[at++] = (unsigned)(p0 + j);
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

__global__ __launch_bounds__(256) void parse_lines_kernel(const uint8_t* __restrict__ text, const unsigned* __restrict__ starts,
                                                          int64_t n_lines, unsigned region_end, int first_line_is_first,
                                                          uint8_t* __restrict__ status, int64_t* __restrict__ user,
                                                          int64_t* __restrict__ item, uint32_t* __restrict__ value,
                                                          unsigned* __restrict__ defer_list, TextCounters* __restrict__ counters) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_lines) return;
  unsigned s, e;
  line_span(text, starts, n_lines, region_end, i, s, e);
  WindowSrc src{text, 0, 0xffffffffu};
  const text::Parsed r = text::parse_line<false>(src, s, e, first_line_is_first && i == 0);
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

// flag[i] = 1 for a record line; the block's counters
__global__ __launch_bounds__(256) void line_summary_kernel(const uint8_t* __restrict__ status, int64_t n_lines, unsigned* __restrict__ flag,
                                                           TextCounters* __restrict__ counters) {
  __shared__ unsigned c[7];
  if (threadIdx.x < 7) c[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int st = -1, fl = 0;
  if (i < n_lines) {
    st = status[i] & 15;
    fl = status[i] >> 4;
    flag[i] = st == text::ST_RECORD ? 1u : 0u;
  }
  // one atomic per wave and class
  const int cls[5] = {text::ST_RECORD, text::ST_BAD, text::ST_FATAL, text::ST_HEADER, text::ST_SKIP};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const unsigned long long b = __ballot(st == cls[k]);
    if (b && (threadIdx.x & 63) == 0) atomicAdd(&c[k], (unsigned)__popcll(b));
  }
  const unsigned long long bu = __ballot(st == text::ST_RECORD && (fl & text::FL_USER_TAG));
  const unsigned long long bi = __ballot(st == text::ST_RECORD && (fl & text::FL_ITEM_TAG));
  if ((threadIdx.x & 63) == 0) {
    if (bu) atomicAdd(&c[5], (unsigned)__popcll(bu));
    if (bi) atomicAdd(&c[6], (unsigned)__popcll(bi));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (c[0]) atomicAdd(&counters->records, c[0]);
    if (c[1]) atomicAdd(&counters->bad, c[1]);
    if (c[2]) atomicAdd(&counters->fatal, c[2]);
    if (c[3]) atomicAdd(&counters->header, c[3]);
    if (c[4]) atomicAdd(&counters->skipped, c[4]);
    if (c[5]) atomicAdd(&counters->user_tags, c[5]);
    if (c[6]) atomicAdd(&counters->item_tags, c[6]);
  }
}

__global__ void compact_records_kernel(const unsigned* __restrict__ flag, const unsigned* __restrict__ flag_scan, int64_t n_lines,
                                       const int64_t* __restrict__ user, const int64_t* __restrict__ item,
                                       const uint32_t* __restrict__ value, int64_t* __restrict__ user_out,
                                       int64_t* __restrict__ item_out, float* __restrict__ value_out) {
  MALS_GRID_STRIDE(i, n_lines) {
    if (!flag[i]) continue;
    const unsigned p = flag_scan[i];
    user_out[p] = user[i];
    item_out[p] = item[i];
    value_out[p] = __uint_as_float(value[i]);
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
