"""csrc/text_parse.h -- the per-line parser the device kernels run -- compiled for the host
(tests/cpp/text_parse_host.cpp) and fuzzed against oracle/ingest_text_oracle.py: every outcome, id, value bit and
flag must be identical.  The GPU tests (tests/test_gpu_ingest_text.py) then only have to show that the kernels
around these functions feed them the right bytes."""
import ctypes
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

from oracle import ingest_text_oracle as to
from tests import text_corpus as tc

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def tp():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp"), "libtext_parse_host.so"])
    L = ctypes.CDLL(os.path.join(HERE, "cpp", "libtext_parse_host.so"))
    L.tp_el_float.restype = ctypes.c_uint32
    L.tp_el_float.argtypes = [ctypes.c_uint64, ctypes.c_int32]
    L.tp_tag_to_long.restype = ctypes.c_int64
    return L


def host_parse(tp, line, first, full=True):
    u, i, v, f = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_uint32(), ctypes.c_int()
    st = tp.tp_parse_line(line, len(line), int(first), int(full), ctypes.byref(u), ctypes.byref(i), ctypes.byref(v), ctypes.byref(f))
    return st, u.value, i.value, v.value, bool(f.value & 1), bool(f.value & 2)


def oracle_parse(line, first):
    units = to.java_utf8_decode(line)
    assert not any(c in (0x0A, 0x0D) for c in units)
    return to.parse_line(units, 1 if first else 2)


def same(a, b):
    if a[0] != b[0]:
        return False
    return a[0] != to.RECORD or a == b


def test_eisel_lemire_matches_exact_rounding(tp):
    """w x 10^q for random 64-bit w and every q the table covers (and beyond, both sides)."""
    rng = np.random.default_rng(7)
    for q in range(-70, 45):
        ws = [1, 2, 9, 10, 2**24 - 1, 2**24, 2**24 + 1, 2**53 + 1, 10**19 - 1, 2**64 - 1, 9999999999999999999, 1 << 63]
        ws += [int(x) for x in rng.integers(1, 2**63, 40, dtype=np.uint64)] + [int(x) for x in rng.integers(1, 10**6, 20)]
        for w in ws:
            want = to.round_to_float32_bits(Fraction(w) * Fraction(10) ** q)
            assert tp.tp_el_float(w, q) == want, (w, q)


def test_eisel_lemire_at_the_midpoints(tp):
    """Decimal significands that sit exactly on, one below and one above the midpoint of two adjacent floats
    (midpoints with at most 19 digits exist for moderate exponents: those are the ties the algorithm must break)."""
    rng = np.random.default_rng(11)
    n = 0
    for _ in range(20000):
        bits = int(rng.integers(0x30000000, 0x5E000000))
        lo, hi = Fraction(tc.f32(bits)), Fraction(tc.f32(bits + 1))
        mid = (lo + hi) / 2
        # mid = w x 10^q exactly?
        q = 0
        m = mid
        while m.denominator != 1 and q > -40:
            m *= 10
            q -= 1
        while m.denominator == 1 and m.numerator % 10 == 0 and m.numerator > 0:
            m /= 10
            q += 1
        if m.denominator != 1 or m.numerator >= 2**64:
            continue
        w = int(m)
        n += 1
        for dw in (-1, 0, 1):
            want = to.round_to_float32_bits(Fraction(w + dw) * Fraction(10) ** q)
            assert tp.tp_el_float(w + dw, q) == want, (w + dw, q)
    assert n > 3000


def test_float_tokens(tp):
    rng = np.random.default_rng(3)
    seen_deferred = 0
    for k in range(60000):
        s = tc.float_text(rng)
        b = s.encode("utf-8")
        try:
            want = (0, to.lang_utils_parse_float(to.java_utf8_decode(b)))
        except (to.NumberFormatException, to.IllegalArgumentException):
            want = (1, None)
        bits = ctypes.c_uint32()
        rc = tp.tp_parse_float(b, len(b), 1, ctypes.byref(bits))
        got = (rc, bits.value if rc == 0 else None)
        assert got == want, (s, got, want)
        rc2 = tp.tp_parse_float(b, len(b), 0, ctypes.byref(bits))       # the fast parser: the same answer or "defer"
        if rc2 == 2:
            seen_deferred += 1
        else:
            assert (rc2, bits.value if rc2 == 0 else None) == want, s
    assert seen_deferred > 50          # the boundary literals with > 19 digits do get deferred


def test_tags(tp):
    rng = np.random.default_rng(5)
    words = [w.encode("utf-8") for w in tc.TAG_WORDS] + tc.BAD_UTF8
    for _ in range(3000):
        n = int(rng.integers(0, 4))
        body = b"".join(words[int(rng.integers(0, len(words)))] for _ in range(n))
        if rng.random() < 0.3:
            body += bytes(rng.integers(0, 256, int(rng.integers(0, 6)), dtype=np.uint8)).replace(b"\n", b"").replace(b"\r", b"").replace(b",", b"")
        tok = b"\"" + body + (b"\"" if rng.random() < 0.8 else b"")
        units = to.java_utf8_decode(tok)
        if len(units) < 2:
            continue
        assert tp.tp_tag_to_long(tok, len(tok)) == to.to_long_id(units[1:len(units) - 1]), tok


def test_lines_fuzz(tp):
    rng = np.random.default_rng(2024)
    counts = {}
    for k in range(120000):
        line = tc.line_bytes(rng, 50, 40, p_odd=0.6)
        first = bool(rng.random() < 0.05)
        want = oracle_parse(line, first)
        got = host_parse(tp, line, first)
        assert same(got, want), (line, got, want)
        fast = host_parse(tp, line, first, full=False)
        assert fast[0] == 5 or same(fast, want), (line, fast, want)
        counts[want[0]] = counts.get(want[0], 0) + 1
        counts["defer"] = counts.get("defer", 0) + (fast[0] == 5)
    assert all(counts.get(s, 0) > 100 for s in (to.RECORD, to.SKIP, to.BAD, to.HEADER, to.FATAL)), counts
    assert counts["defer"] < 0.5 * 120000


def test_plain_lines_never_defer(tp):
    """The fast parser must take the whole bulk: plain numeric lines with ordinary floats."""
    rng = np.random.default_rng(8)
    for _ in range(20000):
        u, i = int(rng.integers(0, 10**7)), int(rng.integers(0, 10**6))
        v = rng.choice(["1", "2.5", "%.6f" % rng.random(), "%.3e" % (rng.random() * 100), " 4 ", "\t3", "-1", ""])
        line = ("%d,%d,%s" % (u, i, v)).encode()
        got = host_parse(tp, line, False, full=False)
        assert got[0] == to.RECORD and same(got, oracle_parse(line, False)), line
