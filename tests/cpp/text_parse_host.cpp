// TEST INFRASTRUCTURE: csrc/text_parse.h compiled for the host (plain g++), so that the line parser the device
// kernels run -- the same inline functions -- can be fuzzed against oracle/ingest_text_oracle.py without a GPU.
#include "../../myrrix-recommender_amd/csrc/text_parse.h"

using namespace mals::text;

extern "C" {

uint32_t tp_el_float(uint64_t w, int32_t q) { return el_float_bits(w, q); }

// status; full = 0: the fast parser (may answer ST_DEFER)
int tp_parse_line(const uint8_t* bytes, uint32_t n, int first, int full, int64_t* user, int64_t* item, uint32_t* value_bits,
                  int* flags) {
  const PtrSrc s{bytes};
  const Parsed r = full ? parse_line<true>(s, 0, n, first != 0) : parse_line<false>(s, 0, n, first != 0);
  *user = r.user;
  *item = r.item;
  *value_bits = r.value_bits;
  *flags = r.flags;
  return r.status;
}

// many lines of one buffer: line i = bytes[start[i], end[i])
void tp_parse_lines(const uint8_t* bytes, const uint32_t* start, const uint32_t* end, int64_t n_lines, int first_line_is_first,
                    int full, int64_t* user, int64_t* item, uint32_t* value_bits, uint8_t* status, uint8_t* flags) {
  const PtrSrc s{bytes};
  for (int64_t i = 0; i < n_lines; ++i) {
    const bool first = first_line_is_first && i == 0;
    const Parsed r = full ? parse_line<true>(s, start[i], end[i], first) : parse_line<false>(s, start[i], end[i], first);
    user[i] = r.user;
    item[i] = r.item;
    value_bits[i] = r.value_bits;
    status[i] = r.status;
    flags[i] = r.flags;
  }
}

int tp_parse_float(const uint8_t* bytes, uint32_t n, int full, uint32_t* bits) {
  const PtrSrc s{bytes};
  return full ? parse_float<true>(s, 0, n, bits) : parse_float<false>(s, 0, n, bits);
}

int64_t tp_tag_to_long(const uint8_t* bytes, uint32_t n) {
  const PtrSrc s{bytes};
  return tag_to_long(s, 0, n);
}

int tp_parse_long(const uint8_t* bytes, uint32_t n, int64_t* out) {
  const PtrSrc s{bytes};
  return parse_long<true>(s, 0, n, out) ? 1 : 0;
}
}
