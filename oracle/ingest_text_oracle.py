"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the TEXT half of the reference's ingest (SURVEY.md
section 8(f) row 2): bytes of the input files -> lines -> records, tag id sets, knownItemIDs.  Not
product code: only tests/ (and tools/bench_ingest.py's cpu_baseline leg) may import it.

Follows, in the reference's own order of evaluation (paths relative to /root/reference):
  InputFilesReader.readInputFiles   online-local/src/net/myrrix/online/generation/InputFilesReader.java
      :71-86    *.csv / *.csv.gz / *.csv.zip, sorted by last-modified time (ByLastModifiedComparator.java:34-44)
      :92-98    one line counter and one bad-line counter over ALL files; "Too many bad lines" is thrown at the
                top of the line that FOLLOWS the 101st bad line
      :101      empty line / first char '#' skipped (after the counter was incremented)
      :105      COMMA.split: Splitter.on(',').trimResults(), consumed lazily (only 3 tokens are ever looked at)
      :114-137  user token, item token ('"' prefix = tag -> OneWayMigrator.toLongID(substring(1, length-1))),
                value token (empty -> NaN, absent -> 1.0f, else LangUtils.parseFloat)
      :139-151  NoSuchElementException -> bad; IllegalArgumentException -> header on line 1, else bad
      :153-157  two tags -> bad
      :159-165  userIsTag -> itemTagIDs.add(userID); itemIsTag -> userTagIDs.add(itemID)     (sic)
      :167-171  NaN -> MatrixUtils.remove, else MatrixUtils.addTo        (oracle/ingest_oracle.py)
      :173-191  knownItemIDs: add the item / remove it, dropping the user's set when it empties
  FileLineIterator.getFileInputStream   common/src/net/myrrix/common/iterator/FileLineIterator.java:92-102
      .gz -> GZIPInputStream; .zip -> a ZipInputStream on which getNextEntry() is never called, which reads
      as an EMPTY stream (java.util.zip.ZipInputStream.read returns -1 while entry == null)
  LangUtils.parseFloat    common/src/net/myrrix/common/LangUtils.java:42-46
  OneWayMigrator          common/src/net/myrrix/common/OneWayMigrator.java
Pinned by the reference's OneWayMigratorTest.java:28-29 and LangUtilsTest.java:23-52 in
tests/test_ingest_text_oracle.py.

Behaviour of code that is NOT under /root/reference (the JDK, guava 14.0.1, mahout-core 0.8), restated from
its published behaviour -- PARITY UNPINNED for these corners, no JVM exists in the image:
  java.io.BufferedReader.readLine (\\n, \\r, \\r\\n), sun.nio.cs.UTF_8 decoding with replacement (JDK 8 malformed
  lengths), String.getBytes(UTF-8), Long.parseLong (JDK 7+), Float.parseFloat (FloatingDecimal.
  readJavaFormatString, correctly rounded), CharMatcher.WHITESPACE, AbstractIDMigrator.hash.
Strings are kept as lists of UTF-16 code units, like Java's.
"""
import gzip
import hashlib
import os
import re
from fractions import Fraction

import numpy as np

from . import ingest_oracle

# ---- java.lang / java.io pieces ---------------------------------------------------------------------------------


def _cont(b):
    return b is not None and (b & 0xC0) == 0x80


def java_utf8_decode(data):
    """bytes -> UTF-16 code units, one U+FFFD per malformed sequence (sun.nio.cs.UTF_8, JDK 8)."""
    out = []
    n = len(data)
    p = 0

    def at(i):
        return data[i] if i < n else None

    while p < n:
        b1 = data[p]
        if b1 < 0x80:
            out.append(b1)
            p += 1
        elif 0xC2 <= b1 <= 0xDF:
            b2 = at(p + 1)
            if _cont(b2):
                out.append(((b1 & 0x1F) << 6) | (b2 & 0x3F))
                p += 2
            else:
                out.append(0xFFFD)
                p += 1
        elif 0xE0 <= b1 <= 0xEF:
            b2, b3 = at(p + 1), at(p + 2)
            if (b1 == 0xE0 and b2 is not None and (b2 & 0xE0) == 0x80) or not _cont(b2):
                out.append(0xFFFD)
                p += 1
            elif not _cont(b3):
                out.append(0xFFFD)
                p += 2
            else:
                c = ((b1 & 0x0F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F)
                out.append(0xFFFD if 0xD800 <= c <= 0xDFFF else c)
                p += 3
        elif 0xF0 <= b1 <= 0xF7:
            b2, b3, b4 = at(p + 1), at(p + 2), at(p + 3)
            if b1 > 0xF4 or not _cont(b2) or (b1 == 0xF0 and not 0x90 <= b2 <= 0xBF) or (b1 == 0xF4 and (b2 & 0xF0) != 0x80):
                out.append(0xFFFD)
                p += 1
            elif not _cont(b3):
                out.append(0xFFFD)
                p += 2
            elif not _cont(b4):
                out.append(0xFFFD)
                p += 3
            else:
                c = ((b1 & 0x07) << 18) | ((b2 & 0x3F) << 12) | ((b3 & 0x3F) << 6) | (b4 & 0x3F)
                c -= 0x10000
                out.append(0xD800 + (c >> 10))
                out.append(0xDC00 + (c & 0x3FF))
                p += 4
        else:
            out.append(0xFFFD)
            p += 1
    return out


def java_utf8_encode(units):
    """String.getBytes(UTF_8): an unpaired surrogate is written as '?'."""
    out = bytearray()
    i = 0
    while i < len(units):
        c = units[i]
        if 0xD800 <= c <= 0xDBFF and i + 1 < len(units) and 0xDC00 <= units[i + 1] <= 0xDFFF:
            cp = 0x10000 + ((c - 0xD800) << 10) + (units[i + 1] - 0xDC00)
            out += bytes([0xF0 | (cp >> 18), 0x80 | ((cp >> 12) & 0x3F), 0x80 | ((cp >> 6) & 0x3F), 0x80 | (cp & 0x3F)])
            i += 2
            continue
        if 0xD800 <= c <= 0xDFFF:
            out.append(0x3F)
        elif c < 0x80:
            out.append(c)
        elif c < 0x800:
            out += bytes([0xC0 | (c >> 6), 0x80 | (c & 0x3F)])
        else:
            out += bytes([0xE0 | (c >> 12), 0x80 | ((c >> 6) & 0x3F), 0x80 | (c & 0x3F)])
        i += 1
    return bytes(out)


def read_lines(units):
    """BufferedReader.readLine over a whole stream."""
    lines, cur, i, n = [], [], 0, len(units)
    pending = False       # chars seen since the last terminator
    while i < n:
        c = units[i]
        if c == 0x0A or c == 0x0D:
            lines.append(cur)
            cur = []
            pending = False
            if c == 0x0D and i + 1 < n and units[i + 1] == 0x0A:
                i += 1
        else:
            cur.append(c)
            pending = True
        i += 1
    if pending:
        lines.append(cur)
    return lines


_GUAVA_WS = set([0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, 0x180E, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000]) \
    | set(range(0x2000, 0x200B))


def guava_trim(tok):
    b, e = 0, len(tok)
    while b < e and tok[b] in _GUAVA_WS:
        b += 1
    while e > b and tok[e - 1] in _GUAVA_WS:
        e -= 1
    return tok[b:e]


def comma_split(line):
    """Iterator of trimmed tokens, like Splitter.on(',').trimResults().split(line).iterator()."""
    start = 0
    for i, c in enumerate(line):
        if c == 0x2C:
            yield guava_trim(line[start:i])
            start = i + 1
    yield guava_trim(line[start:])


class NumberFormatException(ValueError):        # an IllegalArgumentException in Java
    pass


class IllegalArgumentException(ValueError):
    pass


class NoSuchElementException(Exception):
    pass


class StringIndexOutOfBoundsException(Exception):
    pass


# BMP zeros of the Unicode 6.2 decimal-digit blocks (Character.digit in JDK 7/8)
_ND_ZERO = [0x0030, 0x0660, 0x06F0, 0x07C0, 0x0966, 0x09E6, 0x0A66, 0x0AE6, 0x0B66, 0x0BE6, 0x0C66, 0x0CE6, 0x0D66,
            0x0E50, 0x0ED0, 0x0F20, 0x1040, 0x1090, 0x17E0, 0x1810, 0x1946, 0x19D0, 0x1A80, 0x1A90, 0x1B50, 0x1BB0,
            0x1C40, 0x1C50, 0xA620, 0xA8D0, 0xA900, 0xA9D0, 0xAA50, 0xABF0, 0xFF10]


def character_digit(c):
    for z in _ND_ZERO:
        if z <= c < z + 10:
            return c - z
    return -1


def parse_long(s):
    """Long.parseLong(String) of JDK 7+."""
    if not s:
        raise NumberFormatException()
    i, neg = 0, False
    limit = -(2 ** 63 - 1)
    if s[0] < 0x30:                      # possible leading '+' or '-'
        if s[0] == 0x2D:
            neg = True
            limit = -(2 ** 63)
        elif s[0] != 0x2B:
            raise NumberFormatException()
        if len(s) == 1:
            raise NumberFormatException()
        i = 1
    multmin = -((-limit) // 10)          # Java's truncating limit / 10
    result = 0
    while i < len(s):
        d = character_digit(s[i])
        i += 1
        if d < 0 or result < multmin:
            raise NumberFormatException()
        result *= 10
        if result < limit + d:
            raise NumberFormatException()
        result -= d
    return result if neg else -result


def round_to_float32_bits(x):
    """Exact rational x >= 0 -> binary32 bit pattern, round to nearest even; overflow -> 0x7f800000."""
    if x == 0:
        return 0
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if Fraction(2) ** e > x:
        e -= 1
    assert Fraction(2) ** e <= x < Fraction(2) ** (e + 1)
    qe = max(e, -126) - 23
    n = x / Fraction(2) ** qe
    f = n.numerator // n.denominator
    rem = n - f
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (f & 1)):
        f += 1
    if e < -126:
        return f                                  # subnormal, or 2^23 = the smallest normal
    if f == 1 << 24:
        f >>= 1
        e += 1
    if e > 127:
        return 0x7F800000
    return ((e + 127) << 23) | (f & 0x7FFFFF)


_HEX = re.compile(r"([-+])?0[xX](((?P<i1>[0-9a-fA-F]+)\.?)|((?P<i2>[0-9a-fA-F]*)\.(?P<f2>[0-9a-fA-F]+)))[pP](?P<es>[-+])?(?P<e>[0-9]+)[fFdD]?")


def float_parse_float(s):
    """Float.parseFloat(String) -> binary32 bits (may be NaN / infinite); s = UTF-16 units."""
    # String.trim()
    b, e = 0, len(s)
    while b < e and s[b] <= 0x20:
        b += 1
    while e > b and s[e - 1] <= 0x20:
        e -= 1
    t = s[b:e]
    n = len(t)
    if n == 0:
        raise NumberFormatException("empty String")
    if any(c > 0x7F for c in t):
        raise NumberFormatException()             # no non-ASCII char is part of the grammar
    a = "".join(chr(c) for c in t)
    i, neg = 0, False
    if a[0] in "+-":
        neg = a[0] == "-"
        i = 1
    sign_bit = 0x80000000 if neg else 0
    if i >= n:
        raise NumberFormatException()
    if a[i] == "N":
        if a[i:] == "NaN":
            return 0x7FC00000
        raise NumberFormatException()
    if a[i] == "I":
        if a[i:] == "Infinity":
            return sign_bit | 0x7F800000
        raise NumberFormatException()
    if a[i] == "0" and i + 1 < n and a[i + 1] in "xX":
        m = _HEX.fullmatch(a)
        if not m:
            raise NumberFormatException()
        if m.group("i1") is not None:
            ip, fp = m.group("i1"), ""
        else:
            ip, fp = m.group("i2"), m.group("f2")
        sig = Fraction(int(ip + fp, 16)) / Fraction(16) ** len(fp)
        ex = int(m.group("e"))
        if m.group("es") == "-":
            ex = -ex
        ex = max(-100000, min(100000, ex))
        if sig == 0:
            return sign_bit
        return sign_bit | round_to_float32_bits(sig * Fraction(2) ** ex)
    # FloatingDecimal.readJavaFormatString
    n_lead_zero, dec_pt, dec_seen = 0, 0, False
    sign_seen = i == 1
    while i < n:
        c = a[i]
        if c == "0":
            n_lead_zero += 1
        elif c == ".":
            if dec_seen:
                raise NumberFormatException("multiple points")
            dec_pt = i - (1 if sign_seen else 0)
            dec_seen = True
        else:
            break
        i += 1
    digits, n_trail_zero = [], 0
    while i < n:
        c = a[i]
        if "1" <= c <= "9":
            digits.append(c)
            n_trail_zero = 0
        elif c == "0":
            digits.append(c)
            n_trail_zero += 1
        elif c == ".":
            if dec_seen:
                raise NumberFormatException("multiple points")
            dec_pt = i - (1 if sign_seen else 0)
            dec_seen = True
        else:
            break
        i += 1
    n_digits = len(digits) - n_trail_zero
    is_zero = n_digits == 0
    if is_zero and n_lead_zero == 0:
        raise NumberFormatException()
    dec_exp = dec_pt - n_lead_zero if dec_seen else n_digits + n_trail_zero
    if i < n and a[i] in "eE":
        exp_sign, exp_val, really_big, exp_overflow = 1, 0, (2 ** 31 - 1) // 10, False
        i += 1
        if i >= n:
            raise NumberFormatException()         # charAt(++i) throws StringIndexOutOfBounds -> NFE
        if a[i] == "-":
            exp_sign = -1
            i += 1
        elif a[i] == "+":
            i += 1
        exp_at = i
        while i < n:
            if exp_val >= really_big:
                exp_overflow = True
            c = a[i]
            i += 1
            if "0" <= c <= "9":
                exp_val = exp_val * 10 + (ord(c) - 48)
            else:
                i -= 1
                break
        exp_limit = 324 + n_digits + n_trail_zero
        if exp_overflow or exp_val > exp_limit:
            dec_exp = exp_sign * exp_limit
        else:
            dec_exp = dec_exp + exp_sign * exp_val
        if i == exp_at:
            raise NumberFormatException()
    if i < n and (i != n - 1 or a[i] not in "fFdD"):
        raise NumberFormatException()
    if is_zero:
        return sign_bit
    value = Fraction(int("".join(digits[:n_digits]))) * Fraction(10) ** (dec_exp - n_digits)
    return sign_bit | round_to_float32_bits(value)


def lang_utils_parse_float(s):
    """LangUtils.parseFloat (LangUtils.java:42-46) -> binary32 bits of a finite value."""
    bits = float_parse_float(s)
    if (bits & 0x7F800000) == 0x7F800000:
        raise IllegalArgumentException("Bad value")
    return bits


def to_long_id(units):
    """OneWayMigrator.toLongID = AbstractIDMigrator.hash (mahout-core 0.8)."""
    d = hashlib.md5(java_utf8_encode(units)).digest()
    v = int.from_bytes(d[:8], "big")
    return v - (1 << 64) if v >= (1 << 63) else v


# ---- the line loop ------------------------------------------------------------------------------------------------

RECORD, SKIP, BAD, HEADER, FATAL = 0, 1, 2, 3, 4
NAN_BITS = 0x7FC00000


def parse_line(line, lines):
    """One pass of the loop body IFR:99-157 for a line (UTF-16 units) that is line number `lines` (1-based).
    Returns (status, user, item, value bits, userIsTag, itemIsTag)."""
    if len(line) == 0 or line[0] == 0x23:
        return SKIP, 0, 0, 0, False, False
    it = comma_split(line)
    try:
        try:
            u = next(it)
        except StopIteration:
            raise NoSuchElementException()
        user_is_tag = len(u) > 0 and u[0] == 0x22
        if user_is_tag:
            if 1 > len(u) - 1:
                raise StringIndexOutOfBoundsException()
            user = to_long_id(u[1:len(u) - 1])
        else:
            user = parse_long(u)
        try:
            t = next(it)
        except StopIteration:
            raise NoSuchElementException()
        item_is_tag = len(t) > 0 and t[0] == 0x22
        if item_is_tag:
            if 1 > len(t) - 1:
                raise StringIndexOutOfBoundsException()
            item = to_long_id(t[1:len(t) - 1])
        else:
            item = parse_long(t)
        v = next(it, None)
        if v is not None:
            value = NAN_BITS if len(v) == 0 else lang_utils_parse_float(v)
        else:
            value = 0x3F800000
    except NoSuchElementException:
        return BAD, 0, 0, 0, False, False
    except (NumberFormatException, IllegalArgumentException):
        return (HEADER if lines == 1 else BAD), 0, 0, 0, False, False
    except StringIndexOutOfBoundsException:
        return FATAL, 0, 0, 0, False, False
    if user_is_tag and item_is_tag:
        return BAD, 0, 0, 0, False, False
    return RECORD, user, item, value, user_is_tag, item_is_tag


class TooManyBadLines(IOError):
    pass


class UncaughtStringIndexOutOfBounds(RuntimeError):
    pass


def file_bytes(path):
    """FileLineIterator.getFileInputStream (FLI:92-102)."""
    name = os.path.basename(path)
    if name.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    if name.endswith(".zip"):
        return b""            # ZipInputStream without getNextEntry(): an empty stream
    with open(path, "rb") as f:
        return f.read()


_CSV = re.compile(r".+\.csv(\.(zip|gz))?")


def list_input_files(input_dir):
    """IFR:71-86.  Equal timestamps keep the order listFiles() gave them (Arrays.sort is stable), which the JDK
    leaves unspecified: by name here, as the library does."""
    names = sorted(n for n in os.listdir(input_dir) if _CSV.fullmatch(n) and os.path.isfile(os.path.join(input_dir, n)))
    paths = [os.path.join(input_dir, n) for n in names]
    paths.sort(key=lambda p: int(os.stat(p).st_mtime * 1000))       # File.lastModified(): milliseconds
    return paths


def read_streams(streams):
    """The loops IFR:92-192 over the decoded bytes of the input files (in order).  Returns a dict:
    users / items / values (records in order, NaN bits = remove), statuses (per line), lines, bad_lines,
    item_tag_ids, user_tag_ids (sorted), known (dict user -> sorted item list)."""
    users, items, values, statuses = [], [], [], []
    item_tags, user_tags, known = set(), set(), {}
    lines = bad = 0
    for data in streams:
        for line in read_lines(java_utf8_decode(data)):
            if bad > 100:
                raise TooManyBadLines("Too many bad lines; aborting")
            lines += 1
            st, u, i, v, ut, it = parse_line(line, lines)
            statuses.append(st)
            if st == FATAL:
                raise UncaughtStringIndexOutOfBounds()
            if st == BAD:
                bad += 1
            if st != RECORD:
                continue
            if ut:
                item_tags.add(u)
            if it:
                user_tags.add(i)
            users.append(u)
            items.append(i)
            values.append(v)
            if v == NAN_BITS:
                s = known.get(u)
                if s is not None:
                    s.discard(i)
                    if not s:
                        del known[u]
            else:
                known.setdefault(u, set()).add(i)
    return {"users": np.array(users, dtype=np.int64), "items": np.array(items, dtype=np.int64),
            "values": np.array(values, dtype=np.uint32).view(np.float32), "statuses": np.array(statuses, dtype=np.uint8),
            "lines": lines, "bad_lines": bad, "item_tag_ids": np.array(sorted(item_tags), dtype=np.int64),
            "user_tag_ids": np.array(sorted(user_tags), dtype=np.int64),
            "known": {u: sorted(s) for u, s in known.items()}}


def read_input_files(input_dir):
    return read_streams(file_bytes(p) for p in list_input_files(input_dir))


def expected(streams, zero_threshold=1.0e-4):
    """Everything readInputFiles leaves behind: the parse products of read_streams plus the two CSR matrices
    (oracle/ingest_oracle.py) and knownItemIDs as a CSR over the SAME dense indices."""
    r = read_streams(streams)
    (uid, rp, col, val), (iid, cp, ccol, cval) = ingest_oracle.expected_matrices(r["users"], r["items"], r["values"], zero_threshold)
    r["csr_x"] = (uid, rp, col, val)
    r["csr_y"] = (iid, cp, ccol, cval)
    # knownItemIDs has exactly the users that still own an entry before pruning = the rows of RbyRow, and its
    # items are rows of RbyColumn: the same dense indices apply
    assert sorted(r["known"].keys()) == uid.tolist()
    i_index = {int(x): k for k, x in enumerate(iid.tolist())}
    kp, kc = [0], []
    for u in uid.tolist():
        kc.extend(i_index[i] for i in r["known"][u])
        kp.append(len(kc))
    r["known_ptr"] = np.array(kp, dtype=np.int64)
    r["known_idx"] = np.array(kc, dtype=np.int32)
    return r
