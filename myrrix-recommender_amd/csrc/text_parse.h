// text_parse.h -- one input line -> one record, with the reference's semantics (SURVEY.md section 8(f) row 2,
// the text half).  Plain inline functions over a byte accessor, compiled for the device by
// ingest_text_kernels.h (one thread per line) and for the host by tests/cpp/text_parse_host.cpp, which is how
// the arithmetic below is fuzzed against exact rational arithmetic without a GPU.
//
// What is restated (reference paths relative to /root/reference):
//   InputFilesReader.readInputFiles, the body of the line loop
//       online-local/src/net/myrrix/online/generation/InputFilesReader.java:92-158        ("IFR")
//     IFR:101      empty line or first char '#'                      -> skipped
//     IFR:51,105   Splitter.on(',').trimResults(): lazy split, tokens trimmed of CharMatcher.WHITESPACE
//     IFR:114-130  token starting with '"' = tag: substring(1, length-1) -> OneWayMigrator.toLongID;
//                  otherwise Long.parseLong
//     IFR:132-137  third token: empty -> NaN ("remove"), else LangUtils.parseFloat; absent -> 1.0f
//     IFR:139-143  NoSuchElementException (fewer than two tokens)    -> bad line
//     IFR:144-151  IllegalArgumentException: line 1 -> header, ignored; else bad line
//     IFR:153-157  two tags -> bad line
//   LangUtils.parseFloat   common/src/net/myrrix/common/LangUtils.java:42-46  (non-finite rejected)
//   OneWayMigrator         common/src/net/myrrix/common/OneWayMigrator.java (pinned by OneWayMigratorTest.java:28-29)
// Third-party / JDK behaviour NOT under /root/reference, restated from the published behaviour:
//   org.apache.mahout:mahout-core:0.8 AbstractIDMigrator.hash: MD5 of the UTF-8 bytes, first 8 digest
//       bytes as a big-endian long (pinned by the reference's own OneWayMigratorTest vectors)
//   com.google.guava:guava:14.0.1 CharMatcher.WHITESPACE: U+0009-000D, 0020, 0085, 00A0, 1680, 180E,
//       2000-200A, 2028, 2029, 202F, 205F, 3000                                            (unpinned)
//   java.lang.Long.parseLong (JDK 7+: leading '+' accepted; digits = Character.digit, i.e. any BMP Nd)
//   java.lang.Float.parseFloat = sun.misc.FloatingDecimal.readJavaFormatString: trim() of chars <= U+0020,
//       sign, "NaN" / "Infinity", decimal or hexadecimal literal, optional f/F/d/D suffix, the exponent
//       clamp `expLimit = 324 + nDigits + nTrailZero` INCLUDING its quirk (a clamped exponent replaces
//       the decimal-point position instead of adding to it); value correctly rounded (JDK 8+)  (unpinned)
//   sun.nio.cs.UTF_8 decoder with REPLACE (InputStreamReader): one U+FFFD per malformed sequence with
//       JDK 8's malformed lengths; String.getBytes(UTF-8) of a lone surrogate = '?'           (unpinned)
// Decimal -> binary32: Eisel-Lemire with the full 128-bit table (exact for <= 19 significant digits:
// Mushtak & Lemire 2023); longer significands: the truncated value and its successor, and when they
// round differently an exact big-integer comparison with the midpoint.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MALS_HD __host__ __device__ __forceinline__
#else
#define MALS_HD inline
#endif
#define MALS_TABLE static constexpr
#include "text_tables.h"

namespace mals {
namespace text {

// ---- outcome of a line ------------------------------------------------------------------------------
enum : uint8_t {
  ST_RECORD = 0,  // (user, item, value | NaN) goes to MatrixUtils.addTo / remove
  ST_SKIP = 1,    // empty or comment
  ST_BAD = 2,     // counted in badLines
  ST_HEADER = 3,  // unparseable FIRST line of the stream: ignored, not counted
  ST_FATAL = 4,   // the token '"' alone: String.substring(1, 0) throws out of readInputFiles
  ST_DEFER = 5    // fast mode only: the line needs the full parser
};
enum : uint8_t { FL_USER_TAG = 1, FL_ITEM_TAG = 2 };

struct Parsed {
  int64_t user, item;
  uint32_t value_bits;
  uint8_t status, flags;
};

struct PtrSrc {
  const uint8_t* p;
  MALS_HD uint8_t operator()(uint32_t i) const { return p[i]; }
};

// ---- small helpers -----------------------------------------------------------------------------------
MALS_HD void mul64(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  lo = a * b;
  hi = __umul64hi(a, b);
#else
  const unsigned __int128 p = (unsigned __int128)a * b;
  lo = (uint64_t)p;
  hi = (uint64_t)(p >> 64);
#endif
}
MALS_HD int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)x);
#else
  return __builtin_clzll(x);
#endif
}
MALS_HD int clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clz((int)x);
#else
  return __builtin_clz(x);
#endif
}

// ---- decimal significand x 10^q -> binary32 bits (positive), Eisel-Lemire -------------------------------
MALS_HD uint32_t el_float_bits(uint64_t w, int32_t q) {
  if (w == 0 || q < P5_QMIN) return 0u;
  if (q > P5_QMAX) return 0x7f800000u;
  const int lz = clz64(w);
  w <<= lz;
  uint64_t hi, lo;
  mul64(w, P5_128[2 * (q - P5_QMIN)], hi, lo);
  const uint64_t mask = 0xFFFFFFFFFFFFFFFFull >> 26;  // 23 explicit bits + 3
  if ((hi & mask) == mask) {
    uint64_t h2, l2;
    mul64(w, P5_128[2 * (q - P5_QMIN) + 1], h2, l2);
    lo += h2;
    if (h2 > lo) ++hi;
  }
  const int upper = (int)(hi >> 63);
  const int shift = upper + 64 - 23 - 3;
  uint64_t m = hi >> shift;
  int32_t p2 = (int32_t)((((152170 + 65536) * q) >> 16) + 63) + upper - lz + 127;
  if (p2 <= 0) {  // subnormal
    if (-p2 + 1 >= 64) return 0u;
    m >>= (-p2 + 1);
    m += (m & 1);
    m >>= 1;
    return (uint32_t)m;  // m == 2^23 is the smallest normal: the same bit pattern
  }
  if (lo <= 1 && q >= -17 && q <= 10 && (m & 3) == 1) {  // exactly halfway: to even
    if ((m << shift) == hi) m &= ~1ull;
  }
  m += (m & 1);
  m >>= 1;
  if (m >= (2ull << 23)) {
    m = 1ull << 23;
    ++p2;
  }
  m &= ~(1ull << 23);
  if (p2 >= 0xFF) return 0x7f800000u;
  return ((uint32_t)p2 << 23) | (uint32_t)m;
}

// ---- exact comparison of T x 10^q with M x 2^E (T given digit by digit) --------------------------------
struct Big {
  static constexpr int N = 40;  // 1280 bits; the largest operand of cmp_with_midpoint is < 1100
  uint32_t l[N];
  int n;
  bool overflow;
};
MALS_HD void big_set(Big& b, uint32_t v) {
  b.n = v ? 1 : 0;
  b.l[0] = v;
  b.overflow = false;
}
MALS_HD void big_mul_add(Big& b, uint32_t m, uint32_t a) {
  uint64_t c = a;
  for (int i = 0; i < b.n; ++i) {
    const uint64_t t = (uint64_t)b.l[i] * m + c;
    b.l[i] = (uint32_t)t;
    c = t >> 32;
  }
  if (c) {
    if (b.n < Big::N) b.l[b.n++] = (uint32_t)c;
    else b.overflow = true;
  }
}
MALS_HD void big_mul_pow5(Big& b, int e) {
  while (e >= 13) {
    big_mul_add(b, 1220703125u, 0);  // 5^13
    e -= 13;
  }
  uint32_t m = 1;
  for (int i = 0; i < e; ++i) m *= 5;
  if (m > 1) big_mul_add(b, m, 0);
}
MALS_HD void big_shl(Big& b, int s) {
  if (b.n == 0 || s == 0) return;
  const int w = s >> 5, r = s & 31;
  if (b.n + w + 1 > Big::N) {
    b.overflow = true;
    return;
  }
  if (r) {
    uint32_t carry = 0;
    for (int i = 0; i < b.n; ++i) {
      const uint32_t v = b.l[i];
      b.l[i] = (v << r) | carry;
      carry = v >> (32 - r);
    }
    if (carry) b.l[b.n++] = carry;
  }
  if (w) {
    for (int i = b.n - 1; i >= 0; --i) b.l[i + w] = b.l[i];
    for (int i = 0; i < w; ++i) b.l[i] = 0;
    b.n += w;
  }
}
MALS_HD int big_cmp(const Big& a, const Big& b) {
  if (a.n != b.n) return a.n < b.n ? -1 : 1;
  for (int i = a.n - 1; i >= 0; --i)
    if (a.l[i] != b.l[i]) return a.l[i] < b.l[i] ? -1 : 1;
  return 0;
}

// ---- character classes -------------------------------------------------------------------------------
// CharMatcher.WHITESPACE of guava 14.0.1
MALS_HD bool guava_ws(uint32_t c) {
  return (c >= 0x09 && c <= 0x0D) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || c == 0x180E ||
         (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
MALS_HD bool is_cont(uint8_t b) { return (b & 0xC0) == 0x80; }

// Character.digit(c, 10) for a BMP char, -1 if none
MALS_HD int java_digit(uint32_t c) {
  if (c < 0x80) return (c >= '0' && c <= '9') ? (int)(c - '0') : -1;
  for (int i = 1; i < ND_ZERO_COUNT; ++i)
    if (c >= ND_ZERO[i] && c < (uint32_t)ND_ZERO[i] + 10u) return (int)(c - ND_ZERO[i]);
  return -1;
}

// One step of sun.nio.cs.UTF_8's decoder over s[p, e) (REPLACE action).  Returns the number of bytes
// consumed; *cp = the code point, or 0xFFFD for a malformed sequence.  *valid says the bytes consumed
// are the encoding of *cp (so re-encoding gives them back); bytes beyond e count as non-continuation.
template <class Src>
MALS_HD int utf8_step(const Src& s, uint32_t p, uint32_t e, uint32_t* cp, bool* valid) {
  const uint8_t b1 = s(p);
  *valid = true;
  if (b1 < 0x80) {
    *cp = b1;
    return 1;
  }
  const uint8_t b2 = (p + 1 < e) ? s(p + 1) : 0;
  const uint8_t b3 = (p + 2 < e) ? s(p + 2) : 0;
  const uint8_t b4 = (p + 3 < e) ? s(p + 3) : 0;
  *valid = false;
  *cp = 0xFFFD;
  if (b1 >= 0xC2 && b1 <= 0xDF) {
    if (!is_cont(b2)) return 1;
    *cp = ((uint32_t)(b1 & 0x1F) << 6) | (b2 & 0x3F);
    *valid = true;
    return 2;
  }
  if ((b1 & 0xF0) == 0xE0) {
    if ((b1 == 0xE0 && (b2 & 0xE0) == 0x80) || !is_cont(b2)) return 1;
    if (!is_cont(b3)) return 2;
    const uint32_t c = ((uint32_t)(b1 & 0x0F) << 12) | ((uint32_t)(b2 & 0x3F) << 6) | (b3 & 0x3F);
    if (c >= 0xD800 && c <= 0xDFFF) return 3;  // an encoded surrogate: malformed, length 3
    *cp = c;
    *valid = true;
    return 3;
  }
  if ((b1 & 0xF8) == 0xF0) {
    if (b1 > 0xF4 || (b1 == 0xF0 && (b2 < 0x90 || b2 > 0xBF)) || (b1 == 0xF4 && (b2 & 0xF0) != 0x80) || !is_cont(b2)) return 1;
    if (!is_cont(b3)) return 2;
    if (!is_cont(b4)) return 3;
    *cp = ((uint32_t)(b1 & 0x07) << 18) | ((uint32_t)(b2 & 0x3F) << 12) | ((uint32_t)(b3 & 0x3F) << 6) | (b4 & 0x3F);
    *valid = true;
    return 4;
  }
  return 1;  // 80..BF, C0, C1, F8..FF
}

// Splitter.trimResults(): [b, e) shrunk by guava whitespace on both sides.  Every multi-byte whitespace
// starts with a lead byte that is never a continuation byte, so a match at the end of the token is a
// character boundary of the decoder whatever precedes it.
template <class Src>
MALS_HD void trim_token(const Src& s, uint32_t& b, uint32_t& e) {
  while (b < e) {
    const uint8_t c = s(b);
    if (c < 0x80) {
      if (!guava_ws(c)) break;
      ++b;
      continue;
    }
    uint32_t cp;
    bool valid;
    const int len = utf8_step(s, b, e, &cp, &valid);
    if (!valid || !guava_ws(cp)) break;
    b += (uint32_t)len;
  }
  while (e > b) {
    const uint8_t c = s(e - 1);
    if (c < 0x80) {
      if (!guava_ws(c)) break;
      --e;
      continue;
    }
    if (!is_cont(c)) break;
    // candidates: 2 bytes C2 85 / C2 A0; 3 bytes E1 9A 80, E1 A0 8E, E2 80 80..8A/A8/A9/AF, E2 81 9F, E3 80 80
    if (e - b >= 2 && s(e - 2) == 0xC2) {
      if (c == 0x85 || c == 0xA0) {
        e -= 2;
        continue;
      }
      break;
    }
    if (e - b >= 3 && is_cont(s(e - 2))) {
      const uint8_t l = s(e - 3);
      if (l >= 0xE1 && l <= 0xE3) {
        const uint32_t cp = ((uint32_t)(l & 0x0F) << 12) | ((uint32_t)(s(e - 2) & 0x3F) << 6) | (c & 0x3F);
        if (cp >= 0x800 && guava_ws(cp)) {
          e -= 3;
          continue;
        }
      }
    }
    break;
  }
}

// ---- Long.parseLong ---------------------------------------------------------------------------------------
// false = NumberFormatException
template <bool FULL, class Src>
MALS_HD bool parse_long(const Src& s, uint32_t b, uint32_t e, int64_t* out) {
  if (b >= e) return false;
  bool neg = false;
  const uint8_t c0 = s(b);
  if (c0 == '-' || c0 == '+') {
    neg = c0 == '-';
    ++b;
    if (b >= e) return false;
  }
  uint64_t mag = 0;
  while (b < e) {
    int d;
    const uint8_t c = s(b);
    if (c < 0x80) {
      d = (c >= '0' && c <= '9') ? (int)(c - '0') : -1;
      ++b;
    } else if (FULL) {
      uint32_t cp;
      bool valid;
      const int len = utf8_step(s, b, e, &cp, &valid);
      d = (valid && cp < 0x10000) ? java_digit(cp) : -1;
      b += (uint32_t)len;
    } else {
      d = -1;
    }
    if (d < 0) return false;
    if (mag > 922337203685477580ull) return false;  // Long.MAX_VALUE / 10: one more digit leaves Long's range
    mag = mag * 10u + (uint64_t)d;
    if (mag > 9223372036854775808ull) return false;
  }
  if (!neg && mag > 9223372036854775807ull) return false;
  *out = neg ? (int64_t)(0ull - mag) : (int64_t)mag;
  return true;
}

// ---- MD5 (RFC 1321), streamed byte by byte --------------------------------------------------------------
struct Md5 {
  uint32_t a, b, c, d;
  uint32_t w[16];
  uint64_t len;
};
MALS_HD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
MALS_HD void md5_init(Md5& m) {
  m.a = 0x67452301u;
  m.b = 0xefcdab89u;
  m.c = 0x98badcfeu;
  m.d = 0x10325476u;
  m.len = 0;
  for (int i = 0; i < 16; ++i) m.w[i] = 0;
}
MALS_HD void md5_block(Md5& m) {
  uint32_t a = m.a, b = m.b, c = m.c, d = m.d;
  for (int i = 0; i < 64; ++i) {
    uint32_t f;
    int g, r;
    if (i < 16) {
      f = (b & c) | (~b & d);
      g = i;
      r = (i & 3) == 0 ? 7 : (i & 3) == 1 ? 12 : (i & 3) == 2 ? 17 : 22;
    } else if (i < 32) {
      f = (d & b) | (~d & c);
      g = (5 * i + 1) & 15;
      r = (i & 3) == 0 ? 5 : (i & 3) == 1 ? 9 : (i & 3) == 2 ? 14 : 20;
    } else if (i < 48) {
      f = b ^ c ^ d;
      g = (3 * i + 5) & 15;
      r = (i & 3) == 0 ? 4 : (i & 3) == 1 ? 11 : (i & 3) == 2 ? 16 : 23;
    } else {
      f = c ^ (b | ~d);
      g = (7 * i) & 15;
      r = (i & 3) == 0 ? 6 : (i & 3) == 1 ? 10 : (i & 3) == 2 ? 15 : 21;
    }
    const uint32_t t = d;
    d = c;
    c = b;
    b = b + rotl32(a + f + MD5_K[i] + m.w[g], r);
    a = t;
  }
  m.a += a;
  m.b += b;
  m.c += c;
  m.d += d;
  for (int i = 0; i < 16; ++i) m.w[i] = 0;
}
MALS_HD void md5_byte(Md5& m, uint8_t x) {
  const int pos = (int)(m.len & 63);
  m.w[pos >> 2] |= (uint32_t)x << (8 * (pos & 3));
  ++m.len;
  if (pos == 63) md5_block(m);
}
// first 8 digest bytes as a big-endian long (AbstractIDMigrator.hash)
MALS_HD int64_t md5_finish_long(Md5& m) {
  const uint64_t bits = m.len * 8;
  md5_byte(m, 0x80);
  while ((m.len & 63) != 56) md5_byte(m, 0);
  m.w[14] = (uint32_t)bits;
  m.w[15] = (uint32_t)(bits >> 32);
  md5_block(m);
  // digest bytes = a, b, c, d little-endian; the long reads the first eight big-endian
  const uint32_t a = m.a, b = m.b;
  const uint64_t hi = ((uint64_t)(a & 0xFF) << 24) | ((uint64_t)((a >> 8) & 0xFF) << 16) | ((uint64_t)((a >> 16) & 0xFF) << 8) | (a >> 24);
  const uint64_t lo = ((uint64_t)(b & 0xFF) << 24) | ((uint64_t)((b >> 8) & 0xFF) << 16) | ((uint64_t)((b >> 16) & 0xFF) << 8) | (b >> 24);
  return (int64_t)((hi << 32) | lo);
}

// toLongID(token.substring(1, token.length() - 1)) for a token [b, e) that starts with '"' and has more
// than one char: the chars after the quote, minus the last one, re-encoded as UTF-8.
template <class Src>
MALS_HD int64_t tag_to_long(const Src& s, uint32_t b, uint32_t e) {
  Md5 m;
  md5_init(m);
  uint32_t p = b + 1;
  // one unit of delay: the last char of the token is not hashed
  uint32_t pend_at = 0;
  int pend_len = 0;  // 0 = nothing pending; -1 = U+FFFD pending
  while (p < e) {
    uint32_t cp;
    bool valid;
    const int len = utf8_step(s, p, e, &cp, &valid);
    if (pend_len > 0) {
      for (int i = 0; i < pend_len; ++i) md5_byte(m, s(pend_at + (uint32_t)i));
    } else if (pend_len < 0) {
      md5_byte(m, 0xEF);
      md5_byte(m, 0xBF);
      md5_byte(m, 0xBD);
    }
    pend_at = p;
    pend_len = valid ? len : -1;
    p += (uint32_t)len;
  }
  // a supplementary code point is two chars: dropping the last one leaves a lone high surrogate, which
  // String.getBytes(UTF-8) writes as '?'
  if (pend_len == 4) md5_byte(m, '?');
  return md5_finish_long(m);
}

// ---- Float.parseFloat + LangUtils.parseFloat -----------------------------------------------------------------
enum { PF_OK = 0, PF_ERROR = 1, PF_DEFER = 2 };  // ERROR: NumberFormatException or a non-finite value

// exact decision between bits1 (the rounding of the truncated significand) and bits1 + 1: compares the full
// significand (digits of s[b, e), '.' skipped, after `skip` leading zeros... see the caller) with the midpoint
template <class Src>
MALS_HD uint32_t decide_with_midpoint(const Src& s, uint32_t dig_b, uint32_t dig_e, int32_t n_digits, int32_t dec_exp,
                                      uint32_t bits1) {
  // significand: the first min(n_digits, 128) digits (leading zeros already skipped by the caller's dig_b)
  Big T;
  big_set(T, 0);
  int taken = 0;
  uint32_t chunk = 0;
  int in_chunk = 0;
  bool sticky = false;
  for (uint32_t p = dig_b; p < dig_e && taken < n_digits; ++p) {
    const uint8_t c = s(p);
    if (c == '.') continue;
    if (taken < 128) {
      chunk = chunk * 10u + (uint32_t)(c - '0');
      if (++in_chunk == 9) {
        big_mul_add(T, 1000000000u, chunk);
        chunk = 0;
        in_chunk = 0;
      }
    } else if (c != '0') {
      sticky = true;
    }
    ++taken;
  }
  if (in_chunk) {
    uint32_t p10 = 1;
    for (int i = 0; i < in_chunk; ++i) p10 *= 10u;
    big_mul_add(T, p10, chunk);
  }
  const int used = n_digits < 128 ? n_digits : 128;
  const int32_t q = dec_exp - used;  // value = T x 10^q (+ a little if sticky)
  // midpoint between bits1 and bits1 + 1: (2 mant + 1) x 2^(e - 1)
  const uint32_t ex = bits1 >> 23, fr = bits1 & 0x7fffffu;
  const uint32_t mant = ex ? (fr | 0x800000u) : fr;
  const int32_t e2 = (ex ? (int32_t)ex : 1) - 127 - 23;
  Big M;
  big_set(M, 2u * mant + 1u);
  const int32_t E = e2 - 1;
  // T 5^q 2^q  vs  M 2^E
  if (q >= 0) {
    big_mul_pow5(T, q);
    if (q >= E) big_shl(T, q - E);
    else big_shl(M, E - q);
  } else {
    big_mul_pow5(M, -q);
    if (E - q >= 0) big_shl(M, E - q);
    else big_shl(T, q - E);
  }
  int c = big_cmp(T, M);
  if (c == 0 && sticky) c = 1;
  if (T.overflow || M.overflow) return 0x7fc00000u;  // cannot happen (sizes in the header); poisons the value if it does
  if (c < 0) return bits1;
  if (c > 0) return bits1 + 1u;
  return (bits1 & 1u) ? bits1 + 1u : bits1;
}

// hexadecimal literal after "0x": s[b, e); correctly rounded
template <class Src>
MALS_HD int parse_hex_float(const Src& s, uint32_t b, uint32_t e, uint32_t* bits) {
  uint64_t sig = 0;
  int sig_bits_dropped = 0;  // hex digits dropped after 15 significant ones (x4 bits each)
  bool sticky = false, any_digit = false, seen_point = false, any_after_point = false, started = false;
  int frac_digits = 0, taken = 0, int_digits_seen = 0;
  uint32_t p = b;
  for (; p < e; ++p) {
    const uint8_t c = s(p);
    int d;
    if (c >= '0' && c <= '9') d = c - '0';
    else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
    else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
    else if (c == '.') {
      if (seen_point) return PF_ERROR;
      seen_point = true;
      continue;
    } else break;
    any_digit = true;
    if (seen_point) {
      any_after_point = true;
      ++frac_digits;
    } else {
      ++int_digits_seen;
    }
    if (d) started = true;
    if (started) {
      if (taken < 15) {
        sig = (sig << 4) | (uint64_t)d;
        ++taken;
      } else {
        ++sig_bits_dropped;
        if (d) sticky = true;
      }
    }
  }
  // HexDigits+ '.'?   |   HexDigits* '.' HexDigits+
  if (!any_digit) return PF_ERROR;
  if (seen_point && int_digits_seen == 0 && !any_after_point) return PF_ERROR;
  if (p >= e || (s(p) != 'p' && s(p) != 'P')) return PF_ERROR;
  ++p;
  bool eneg = false;
  if (p < e && (s(p) == '-' || s(p) == '+')) {
    eneg = s(p) == '-';
    ++p;
  }
  int64_t ev = 0;
  int edigits = 0;
  for (; p < e; ++p) {
    const uint8_t c = s(p);
    if (c < '0' || c > '9') break;
    if (ev < 100000000) ev = ev * 10 + (c - '0');
    ++edigits;
  }
  if (!edigits) return PF_ERROR;
  if (p < e) {
    const uint8_t c = s(p);
    if (p != e - 1 || !(c == 'f' || c == 'F' || c == 'd' || c == 'D')) return PF_ERROR;
  }
  if (sig == 0) {
    *bits = 0;
    return PF_OK;
  }
  // value = sig x 16^sig_bits_dropped x 2^(-4 frac_digits) x 2^(+-ev)
  int64_t e2 = 4 * (int64_t)sig_bits_dropped - 4 * (int64_t)frac_digits + (eneg ? -ev : ev);
  // normalise sig to 64 bits
  const int lz = clz64(sig);
  sig <<= lz;
  e2 -= lz;  // value = sig x 2^e2, sig in [2^63, 2^64)
  int64_t ue = e2 + 63;  // unbiased exponent of the leading bit
  if (ue > 127) {
    *bits = 0x7f800000u;
    return PF_OK;
  }
  int keep = 24;  // significant bits kept
  if (ue < -126) keep = 24 - (int)((-126) - ue > 64 ? 64 : (-126) - ue);
  if (keep < 0) {  // below half of the smallest subnormal
    *bits = 0;
    return PF_OK;
  }
  uint64_t mant = keep == 0 ? 0 : (sig >> (64 - keep));
  const uint64_t rest = keep == 0 ? sig : (sig << keep);
  const uint64_t half = 0x8000000000000000ull;
  bool up;
  if (rest > half || (rest == half && sticky)) up = true;
  else if (rest == half) up = (mant & 1) != 0;
  else up = false;
  if (up) ++mant;
  uint32_t out;
  if (ue < -126) {
    out = (uint32_t)mant;  // subnormal (mant == 2^23 = the smallest normal)
  } else {
    if (mant == (1ull << 24)) {
      mant >>= 1;
      ++ue;
    }
    if (ue > 127) {
      *bits = 0x7f800000u;
      return PF_OK;
    }
    out = ((uint32_t)(ue + 127) << 23) | ((uint32_t)mant & 0x7fffffu);
  }
  *bits = out;
  return PF_OK;
}

// s[b, e) = the (guava-trimmed, non-empty) value token.  FULL = false: anything but a plain decimal literal of
// at most 19 significant digits is deferred when the answer is not immediate.
template <bool FULL, class Src>
MALS_HD int parse_float(const Src& s, uint32_t b, uint32_t e, uint32_t* bits_out) {
  // String.trim(): chars <= U+0020
  while (b < e && s(b) <= 0x20) ++b;
  while (e > b && s(e - 1) <= 0x20) --e;
  if (b >= e) return PF_ERROR;
  uint32_t i = b;
  bool neg = false;
  if (s(i) == '-' || s(i) == '+') {
    neg = s(i) == '-';
    ++i;
    if (i >= e) return PF_ERROR;
  }
  uint8_t c = s(i);
  if (c == 'N' || c == 'I') return PF_ERROR;  // "NaN" / "Infinity" parse, then LangUtils rejects them; anything else is malformed
  if (c == '0' && i + 1 < e && (s(i + 1) == 'x' || s(i + 1) == 'X')) {
    if (!FULL) return PF_DEFER;
    uint32_t hb;
    const int rc = parse_hex_float(s, i + 2, e, &hb);
    if (rc != PF_OK || hb == 0x7f800000u) return PF_ERROR;
    *bits_out = hb | (neg ? 0x80000000u : 0u);
    return PF_OK;
  }
  // leading zeros and the decimal point
  int32_t n_lead_zero = 0, dec_pt = 0;
  bool dec_seen = false;
  const uint32_t body = i;  // decPt is counted from here (Java subtracts the sign position)
  for (; i < e; ++i) {
    c = s(i);
    if (c == '0') ++n_lead_zero;
    else if (c == '.') {
      if (dec_seen) return PF_ERROR;
      dec_pt = (int32_t)(i - body);
      dec_seen = true;
    } else break;
  }
  // digits
  const uint32_t dig_b = i;
  int32_t n_digits = 0, n_trail_zero = 0;
  uint64_t w = 0;  // the first 19 digits
  for (; i < e; ++i) {
    c = s(i);
    if (c >= '1' && c <= '9') {
      if (n_digits < 19) w = w * 10u + (uint64_t)(c - '0');
      ++n_digits;
      n_trail_zero = 0;
    } else if (c == '0') {
      if (n_digits < 19) w = w * 10u;
      ++n_digits;
      ++n_trail_zero;
    } else if (c == '.') {
      if (dec_seen) return PF_ERROR;
      dec_pt = (int32_t)(i - body);
      dec_seen = true;
    } else break;
  }
  const uint32_t dig_e = i;
  const int32_t all_digits = n_digits;  // with the trailing zeros
  n_digits -= n_trail_zero;
  const bool is_zero = n_digits == 0;
  if (is_zero && n_lead_zero == 0) return PF_ERROR;  // no digit at all
  int64_t dec_exp = dec_seen ? (int64_t)dec_pt - n_lead_zero : (int64_t)all_digits;
  if (i < e && (s(i) == 'e' || s(i) == 'E')) {
    ++i;
    if (i >= e) return PF_ERROR;
    int exp_sign = 1;
    if (s(i) == '-' || s(i) == '+') {
      exp_sign = s(i) == '-' ? -1 : 1;
      ++i;
    }
    const uint32_t exp_at = i;
    int32_t exp_val = 0;
    bool exp_overflow = false;
    for (; i < e; ++i) {
      if (exp_val >= 214748364) exp_overflow = true;  // Integer.MAX_VALUE / 10
      c = s(i);
      if (c < '0' || c > '9') break;
      if (!exp_overflow) exp_val = exp_val * 10 + (c - '0');
    }
    const int64_t exp_limit = 324 + (int64_t)all_digits;  // bigDecimalExponent + nDigits + nTrailZero
    if (exp_overflow || exp_val > exp_limit) dec_exp = exp_sign * exp_limit;  // sic: replaces the point position
    else dec_exp += (int64_t)exp_sign * exp_val;
    if (i == exp_at) return PF_ERROR;
  }
  if (i < e) {
    c = s(i);
    if (i != e - 1 || !(c == 'f' || c == 'F' || c == 'd' || c == 'D')) return PF_ERROR;
  }
  if (is_zero) {
    *bits_out = neg ? 0x80000000u : 0u;
    return PF_OK;
  }
  // value = 0.d1 d2 ... d_n x 10^dec_exp
  uint32_t bits;
  if (n_digits <= 19) {
    // w holds min(all_digits, 19) digits, of which the first n_digits are significant
    const int held = all_digits < 19 ? all_digits : 19;
    int64_t q = dec_exp - held;
    if (q < -400) q = -400;
    if (q > 400) q = 400;
    bits = el_float_bits(w, (int32_t)q);
  } else {
    int64_t q = dec_exp - 19;
    if (q < -400) q = -400;
    if (q > 400) q = 400;
    const uint32_t b1 = el_float_bits(w, (int32_t)q);
    const uint32_t b2 = el_float_bits(w + 1, (int32_t)q);
    if (b1 == b2) {
      bits = b1;
    } else {
      if (!FULL) return PF_DEFER;
      int64_t de = dec_exp;
      if (de < -100000) de = -100000;
      if (de > 100000) de = 100000;
      bits = decide_with_midpoint(s, dig_b, dig_e, n_digits, (int32_t)de, b1);
    }
  }
  if (bits >= 0x7f800000u) return PF_ERROR;  // Infinity (or the poisoned value): Preconditions.checkArgument fails
  *bits_out = bits | (neg ? 0x80000000u : 0u);
  return PF_OK;
}

// ---- one line ---------------------------------------------------------------------------------------------------
// s[b, e) = the line without its terminator; first = it is line 1 of the whole input (IFR:145).
template <bool FULL, class Src>
MALS_HD Parsed parse_line(const Src& s, uint32_t b, uint32_t e, bool first) {
  Parsed r;
  r.user = r.item = 0;
  r.value_bits = 0x3f800000u;
  r.flags = 0;
  r.status = ST_SKIP;
  if (b >= e || s(b) == '#') return r;
  const uint8_t illegal = first ? ST_HEADER : ST_BAD;
  // the first three tokens
  uint32_t tb[3], te[3];
  int n_tok = 0;
  {
    uint32_t p = b, start = b;
    bool odd = false;
    for (; p < e && n_tok < 3; ++p) {
      const uint8_t c = s(p);
      if (!FULL && (c >= 0x80 || (c < 0x20 && c != '\t') || c == '"')) odd = true;
      if (c == ',') {
        tb[n_tok] = start;
        te[n_tok] = p;
        ++n_tok;
        start = p + 1;
      }
    }
    if (n_tok < 3) {
      tb[n_tok] = start;
      te[n_tok] = e;
      ++n_tok;
    }
    if (!FULL && odd) {
      r.status = ST_DEFER;
      return r;
    }
  }
  // user, then item (IFR:114-130)
  for (int t = 0; t < 2; ++t) {
    if (t >= n_tok) {
      r.status = ST_BAD;  // NoSuchElementException: never a header
      return r;
    }
    uint32_t x = tb[t], y = te[t];
    trim_token(s, x, y);
    int64_t id;
    if (FULL && x < y && s(x) == '"') {
      // token.length() in chars: 1 iff the token is the quote alone
      if (y - x == 1) {
        r.status = ST_FATAL;
        return r;
      }
      id = tag_to_long(s, x, y);
      r.flags |= t == 0 ? FL_USER_TAG : FL_ITEM_TAG;
    } else if (!parse_long<FULL>(s, x, y, &id)) {
      r.status = illegal;
      return r;
    }
    if (t == 0) r.user = id;
    else r.item = id;
  }
  if (n_tok >= 3) {
    uint32_t x = tb[2], y = te[2];
    trim_token(s, x, y);
    if (x >= y) {
      r.value_bits = 0x7fc00000u;  // Float.NaN: remove
    } else {
      uint32_t bits;
      const int rc = parse_float<FULL>(s, x, y, &bits);
      if (rc == PF_DEFER) {
        r.status = ST_DEFER;
        return r;
      }
      if (rc != PF_OK) {
        r.status = illegal;
        return r;
      }
      r.value_bits = bits;
    }
  }
  if ((r.flags & FL_USER_TAG) && (r.flags & FL_ITEM_TAG)) {
    r.status = ST_BAD;  // IFR:153-157
    return r;
  }
  r.status = ST_RECORD;
  return r;
}

}  // namespace text
}  // namespace mals
