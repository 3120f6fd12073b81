"""Seeded generator of input-file text for the ingest tests (CPU: the host build of csrc/text_parse.h against
oracle/ingest_text_oracle.py; GPU: mals_ingest_append_text against the same oracle).  Everything a line of an
input file can be, by the reference's reader (InputFilesReader.java:92-158): plain records, signs, exponents,
suffixes, whitespace of every kind guava trims, CR / LF / CRLF, tags (ASCII, non-ASCII, supplementary, malformed
UTF-8), removals, missing values, extra columns, comments, empty lines, headers, bad lines of every exception
class, ids at the edges of Long, non-ASCII decimal digits, long significands at rounding boundaries."""
import struct
from fractions import Fraction

import numpy as np

WS_ASCII = [" ", "\t", "  ", " \t "]
WS_ODD = ["\x0b", "\x0c", "\u0085", "\u00a0", "\u1680", "\u180e", "\u2000", "\u2005", "\u200a", "\u2028", "\u2029",
          "\u202f", "\u205f", "\u3000"]
NOT_WS = ["\u200b", "\ufeff", "\x1c", "\x1f", "\x00", "\x7f"]
TAG_WORDS = ["foobar", "", "a", "rock", "jazz fusion", "caf\u00e9", "\u6771\u4eac", "na\u00efve", "x\U0001F600", "\U0001F600",
             "tag,with", "q\"uote", " spaced ", "\u00e9", "\u20ac\u20ac", "z" * 70, "y" * 55, "w" * 56, "v" * 64, "0123456789" * 13]
BAD_UTF8 = [b"\xff", b"\xc0\x80", b"\xe0\x80\x80", b"\xe2\x82", b"\xed\xa0\x80", b"\xf0\x80\x80\x80", b"\xf4\x90\x80\x80",
            b"\xf0\x9f\x98", b"\x80", b"\xc2", b"\xe1\x80", b"\xf5\x80\x80\x80", b"\xf0\x9f", b"\xbf\xbf"]


def f32(bits):
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def exact_decimal(fr, max_digits=200):
    """Exact decimal expansion of a Fraction whose denominator is a power of two (or 2^a 5^b)."""
    n, d = fr.numerator, fr.denominator
    ip = n // d
    r = n % d
    out = []
    while r and len(out) < max_digits:
        r *= 10
        out.append(str(r // d))
        r %= d
    return str(ip) + ("." + "".join(out) if out else "")


def sci(fr, digits):
    """fr (positive Fraction) as d.ddd...e[+-]xx with `digits` significant digits, truncated."""
    e = 0
    while fr >= 10:
        fr /= 10
        e += 1
    while fr < 1:
        fr *= 10
        e -= 1
    s = []
    for _ in range(digits):
        d = int(fr)
        s.append(str(d))
        fr = (fr - d) * 10
    return s[0] + "." + "".join(s[1:]) + "e" + str(e)


def boundary_float_text(rng):
    """A decimal literal at, just below or just above the midpoint of two adjacent binary32 values."""
    kind = rng.integers(0, 6)
    if kind == 0:
        bits = int(rng.integers(0, 0x7F800000))
    elif kind == 1:
        bits = int(rng.integers(0, 1 << 24))            # subnormals and small normals
    elif kind == 2:
        bits = int(0x7F7FFFFF - rng.integers(0, 4))     # the top of the range (the last midpoint rounds to infinity)
    elif kind == 3:
        bits = int(rng.integers(0x3F000000, 0x41000000))  # around 1
    elif kind == 4:
        bits = int(rng.integers(0, 8))
    else:
        bits = int(rng.integers(0x3A000000, 0x4B000000))
    lo = Fraction(f32(bits))
    hi = Fraction(f32(bits + 1)) if bits + 1 < 0x7F800000 else Fraction(2) ** 128
    mid = (lo + hi) / 2
    mode = rng.integers(0, 6)
    if mode == 0:
        return exact_decimal(mid, 400)
    if mode == 1:
        return exact_decimal(mid, 400) + "0000001"
    if mode == 2:
        s = exact_decimal(mid, 400)
        # one unit in the last place less: still 100+ digits
        digs = list(s)
        k = len(digs) - 1
        while k >= 0 and digs[k] in ".0":
            if digs[k] == "0":
                digs[k] = "9"
            k -= 1
        if k >= 0:
            digs[k] = str(int(digs[k]) - 1)
        return "".join(digs)
    if mode == 3:
        return sci(mid, int(rng.integers(17, 40)))
    if mode == 4:
        return sci(mid, int(rng.integers(1, 19)))
    return sci(lo if (lo > 0 and rng.random() < 0.5) else mid, 19) + str(int(rng.integers(0, 10**6)))


def float_text(rng):
    k = rng.integers(0, 30)
    if k < 8:
        return str(int(rng.integers(1, 6)))
    if k < 12:
        return "%.1f" % (rng.integers(1, 11) * 0.5)
    if k < 15:
        return repr(float(np.float32(rng.standard_normal() * 3)))
    if k < 17:
        return "%.*e" % (int(rng.integers(0, 12)), rng.standard_normal() * 10.0 ** int(rng.integers(-50, 45)))
    if k == 17:
        return str(rng.choice(["1e5", "1E5", "1e+5", "1e-5", "2.5f", "2.5F", "2.5d", "2.5D", "1.", ".5", "0.", ".0", "00012.500",
                               "+3", "-3", "-0", "-0.0", "1e0", "0e99999999999", "0.00001", "0.0001", "0.00009999", "1e-4",
                               "9.9999e-5", "-1e-4", "1e38", "3.4028235e38", "3.4028234663852886e38", "1e-45", "1.4e-45", "7e-46",
                               "7.1e-46", "1e-46", "0.0000000000000000000000000000000000000000000001", "100000000000000000000",
                               "123456789012345678901234567890", "16777217", "16777216.000000000000001", "33554433", "0.1", "0.3",
                               "3.1", "1e2147483647", "1e-2147483648", "1e214748364", "1e214748365", "1e99999999999999999999"]))
    if k == 18:
        return str(rng.choice(["NaN", "Infinity", "-Infinity", "+Infinity", "nan", "inf", "1e39", "3.4028236e38", "1e400", "4e38",
                               "1e", "e5", "1e+", "1e-", ".", "-", "+", "1..2", "1.2.3", "--1", "1-", "1e5.5", "1ff", "1fd", "f", "0x",
                               "1,5", "one", "1 2", "1_0", "\u0661", "1\u00a0", "0x1", "0x1p", "0xp1", "0x.p1", "0xg.p1", "1e5e5",
                               "1d5", "Infinityx", "NaNx", "N", "I"]))
    if k == 19:
        return str(rng.choice(["0x1p0", "0x1.8p1", "0X1.8P1", "0x.8p1", "0x1.p-1", "-0x1p0", "0x1p-149", "0x1p-150", "0x1.000001p-150",
                               "0x1p-151", "0x1.fffffep127", "0x1.ffffffp127", "0x1.fffffefp127", "0x1p128", "0x0p0", "0x0.0p0",
                               "0x1.000001p0", "0x1.0000010000000001p0", "0x1.000003p0", "0x1.0000030p0f", "0x123456789abcdef012345p-60",
                               "0x0.00000000000000000001p80", "0x1p99999999999", "0x1p-99999999999", "0x1.8p+1d", "0x1.fffffe0000001p-127",
                               "0x0.fffffep-126", "0x0.ffffffp-126", "0x0.000002p-126", "0x0.000001p-126", "0x0.0000011p-126"]))
    if k < 24:
        return boundary_float_text(rng)
    if k < 26:
        n = int(rng.integers(1, 60))
        s = "".join(str(int(d)) for d in rng.integers(0, 10, n))
        p = int(rng.integers(0, n + 1))
        s = s[:p] + "." + s[p:]
        if rng.random() < 0.5:
            s += "e%d" % int(rng.integers(-60, 60))
        return s
    if k == 26:
        return "0" * int(rng.integers(1, 30)) + "." + "0" * int(rng.integers(0, 60)) + str(int(rng.integers(1, 10**9)))
    if k == 27:
        # the exponent clamp: a huge exponent that the digits would cancel
        z = int(rng.integers(300, 420))
        return "0." + "0" * z + "1e%d" % (z + int(rng.integers(-2, 3)))
    if k == 28:
        return str(int(rng.integers(1, 10**9))) + "0" * int(rng.integers(0, 40))
    return "%de%d" % (int(rng.integers(1, 10**18)), int(rng.integers(-70, 30)))


def id_text(rng, n_ids, side):
    k = rng.integers(0, 60)
    if k < 48:
        return str(int(rng.integers(0, n_ids)) + (1000 if side else 0))
    if k < 50:
        return str(rng.choice(["9223372036854775807", "-9223372036854775808", "9223372036854775808", "-9223372036854775809",
                               "92233720368547758070", "18446744073709551616", "18446744073709551617", "-0", "+5", "+", "-", "",
                               "0000000000000000000000000000007", "-00000000000000000000000009223372036854775808", "12a", "a12", "1 2",
                               "1.0", "1e3", "0x10", "\u0663", "\u0661\u0662\u0663", "\uff11\uff12", "1\u0663", "\U0001D7CE", "922337203685477580", "922337203685477581",
                               "9223372036854775799", "1844674407370955161", "18446744073709551615", "18446744073709551619"]))
    if k < 53:
        return str(int(rng.integers(-2**63, 2**63 - 1)))
    if k < 59:
        return "\"" + str(rng.choice(TAG_WORDS)) + "\""
    return str(rng.choice(["\"", "\"x", "\"\"", "\"abc", "\"a\"b\"", "x\"y\"", "\"\U0001F600", "\"ab\U0001F600"]))


def pad(rng, s, odd=0.03):
    def ws():
        r = rng.random()
        if r < 0.85:
            return ""
        if r < 0.85 + 0.15 * (1 - odd):
            return str(rng.choice(WS_ASCII))
        if rng.random() < 0.8:
            return str(rng.choice(WS_ODD))
        return str(rng.choice(NOT_WS))
    return ws() + s + ws()


def line_bytes(rng, n_users, n_items, p_odd=0.25):
    """One line (without terminator) as bytes."""
    r = rng.random()
    if r > p_odd:                                   # the bulk: plain numeric records
        u = str(int(rng.integers(0, n_users)))
        i = str(int(rng.integers(0, n_items)) + 1000)
        k = rng.integers(0, 20)
        if k < 14:
            v = str(rng.choice(["1", "2", "3", "4", "5", "0.5", "1.5", "2.5", "3.5", "4.5", "1.0", "5.0", "-1", "0.00003", "-2.5"]))
            return ("%s,%s,%s" % (u, i, v)).encode()
        if k < 16:
            return ("%s,%s" % (u, i)).encode()
        if k < 18:
            return ("%s,%s," % (u, i)).encode()
        return ("%s,%s,%s" % (u, i, float_text(rng))).encode()
    k = rng.integers(0, 40)
    if k < 2:
        return b""
    if k < 4:
        return ("#" + str(rng.choice(["", " comment", "1,2,3", ",,,"]))).encode()
    if k < 6:
        return str(rng.choice([" ", "\t", " #x", ",", ",,", ",,,", "1", "1,", ",1", "1,2,3,4", "1,2,3,", "1,2,,4", "abc", "user,item,value",
                               "\"t\",\"u\",1", "\"t\",\"u\"", "1,\"", "\",1", "x,\"", "1,2,\"", "\ufeff1,2,3", "1;2;3", "1\t2\t3",
                               "1,2,3\x00", "\x001,2,3", "1,2,\x003", "1,2,3\x1f", "1,2,\x1c3\x1d", "1,2, 3 ,", " 1 , 2 , 3 "])).encode()
    if k < 8:
        # malformed UTF-8 somewhere
        bad = BAD_UTF8[int(rng.integers(0, len(BAD_UTF8)))]
        where = rng.integers(0, 5)
        u = str(int(rng.integers(0, n_users))).encode()
        i = str(int(rng.integers(0, n_items)) + 1000).encode()
        if where == 0:
            return b"\"ab" + bad + b"cd\"," + i + b",1"
        if where == 1:
            return b"\"ab" + bad + b"\"," + i + b",2"
        if where == 2:
            return u + b",\"" + bad + b"\",3"
        if where == 3:
            return u + bad + b"," + i + b",1"
        return u + b"," + i + b",1" + bad
    u = pad(rng, id_text(rng, n_users, 0), 0.3)
    i = pad(rng, id_text(rng, n_items, 1), 0.3)
    k2 = rng.integers(0, 10)
    if k2 == 0:
        s = "%s,%s" % (u, i)
    elif k2 == 1:
        s = "%s,%s,%s" % (u, i, pad(rng, "", 0.3))
    elif k2 == 2:
        s = "%s,%s,%s,%s" % (u, i, pad(rng, float_text(rng), 0.3), str(rng.choice(["", "x", "1,2", "\"", "\xff"])))
    else:
        s = "%s,%s,%s" % (u, i, pad(rng, float_text(rng), 0.3))
    return s.encode("utf-8", "surrogatepass")


def corpus(seed, n_lines, n_users=50, n_items=40, p_odd=0.25, first_line=None, final_newline=True, terminators=("\n",)):
    """bytes of one input file."""
    rng = np.random.default_rng(seed)
    out = bytearray()
    for k in range(n_lines):
        if k == 0 and first_line is not None:
            out += first_line
        else:
            out += line_bytes(rng, n_users, n_items, p_odd)
        if k < n_lines - 1 or final_newline:
            out += str(rng.choice(list(terminators))).encode()
    return bytes(out)
