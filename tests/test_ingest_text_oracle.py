"""The text oracle (oracle/ingest_text_oracle.py) against the vectors the reference's own tests hold for this
path, and against independent arithmetic for what they do not cover."""
import struct
from fractions import Fraction

import numpy as np
import pytest

from oracle import ingest_text_oracle as to


def units(s):
    return to.java_utf8_decode(s.encode("utf-8", "surrogatepass") if isinstance(s, str) else s)


def test_one_way_migrator_vectors():
    """OneWayMigratorTest.java:28-29"""
    assert to.to_long_id(units("foobar")) == 4060265690780417169
    assert to.to_long_id(units("")) == -3162216497309240828


def test_lang_utils_vectors():
    """LangUtilsTest.java:23-52"""
    for s in ("NaN", "Infinity", "-Infinity"):
        with pytest.raises(to.IllegalArgumentException):
            to.lang_utils_parse_float(units(s))
    assert to.lang_utils_parse_float(units("3.1")) == struct.unpack("<I", struct.pack("<f", 3.1))[0]


def test_float_rounding_against_numpy_where_double_rounding_cannot_bite():
    """<= 7 significant digits and a moderate exponent: float(s) is the correctly rounded double, and rounding
    that to binary32 can only differ from the single rounding when the double lands exactly on a binary32
    midpoint, which a 7-digit decimal cannot approach to within 2^-53 relative unless it IS the midpoint."""
    rng = np.random.default_rng(1)
    for _ in range(3000):
        s = "%.*e" % (int(rng.integers(0, 7)), rng.standard_normal() * 10.0 ** int(rng.integers(-30, 30)))
        want = int(np.float32(float(s)).view(np.uint32))
        assert to.float_parse_float(units(s)) == want, s


def test_round_to_float32_bits_edges():
    F = Fraction
    assert to.round_to_float32_bits(F(1)) == 0x3F800000
    assert to.round_to_float32_bits(F(2) ** -149) == 1
    assert to.round_to_float32_bits(F(2) ** -150) == 0            # tie -> even (zero)
    assert to.round_to_float32_bits(F(2) ** -150 + F(2) ** -200) == 1
    assert to.round_to_float32_bits(F(2) ** -126) == 0x00800000
    assert to.round_to_float32_bits(F(2) ** -126 - F(2) ** -150) == 0x00800000   # tie at the top of the subnormals -> even
    assert to.round_to_float32_bits(F(2) ** 128 - F(2) ** 104) == 0x7F7FFFFF
    assert to.round_to_float32_bits(F(2) ** 128 - F(2) ** 103) == 0x7F800000     # the last midpoint rounds to infinity
    assert to.round_to_float32_bits(F(2) ** 128 - F(2) ** 103 - 1) == 0x7F7FFFFF
    assert to.round_to_float32_bits(1 + F(2) ** -24) == 0x3F800000               # tie -> even
    assert to.round_to_float32_bits(1 + 3 * F(2) ** -24) == 0x3F800002


def test_read_lines_is_buffered_reader():
    def L(b):
        return ["".join(map(chr, x)) for x in to.read_lines(to.java_utf8_decode(b))]
    assert L(b"a\nb") == ["a", "b"]
    assert L(b"a\nb\n") == ["a", "b"]
    assert L(b"a\r\nb\r") == ["a", "b"]
    assert L(b"a\n\nb") == ["a", "", "b"]
    assert L(b"\n") == [""]
    assert L(b"") == []
    assert L(b"a\r\r\nb") == ["a", "", "b"]
    assert L(b"a\n\rb") == ["a", "", "b"]


def test_decoder_replacement_lengths():
    D = to.java_utf8_decode
    assert D(b"\xe2\x82,") == [0xFFFD, 0x2C]
    assert D(b"\xed\xa0\x80") == [0xFFFD]                  # an encoded surrogate is ONE malformed sequence of 3
    assert D(b"\xe0\x80\x80") == [0xFFFD] * 3
    assert D(b"\xf0\x9f\x98\x80") == [0xD83D, 0xDE00]
    assert D(b"\xf0\x9f\x98") == [0xFFFD]
    assert D(b"\xf4\x90\x80\x80") == [0xFFFD] * 4
    assert D(b"\xc0\x80") == [0xFFFD] * 2
    assert to.java_utf8_encode([0xD83D]) == b"?"
    assert to.java_utf8_encode([0xD83D, 0xDE00]) == b"\xf0\x9f\x98\x80"


def test_line_semantics():
    def P(s, n=2):
        return to.parse_line(units(s), n)
    assert P("1,2,3")[:4] == (to.RECORD, 1, 2, 0x40400000)
    assert P("1,2")[:4] == (to.RECORD, 1, 2, 0x3F800000)
    assert P("1,2,")[:4] == (to.RECORD, 1, 2, to.NAN_BITS)
    assert P(" 1 ,\t2 , 3.5 , junk")[:4] == (to.RECORD, 1, 2, 0x40600000)
    assert P("")[0] == to.SKIP and P("#1,2")[0] == to.SKIP
    assert P(" #1,2")[0] == to.BAD
    assert P("1")[0] == to.BAD and P("1", 1)[0] == to.BAD              # NoSuchElement is never a header
    assert P("user,item", 1)[0] == to.HEADER and P("user,item")[0] == to.BAD
    assert P("1,2,NaN")[0] == to.BAD and P("1,2,1e39")[0] == to.BAD
    assert P("\"a\",\"b\",1")[0] == to.BAD and P("\"a\",\"b\",1", 1)[0] == to.BAD
    assert P("\"foobar\",7")[:3] == (to.RECORD, 4060265690780417169, 7) and P("\"foobar\",7")[4:] == (True, False)
    assert P("7,\"\"")[:3] == (to.RECORD, 7, -3162216497309240828)
    assert P("\"foobarX,7")[1] == 4060265690780417169                   # substring(1, length-1) drops the last char, quote or not
    assert P("\",7")[0] == to.FATAL and P("x,\"")[0] == to.BAD and P("1,\"")[0] == to.FATAL
    assert P("+5,-0")[:3] == (to.RECORD, 5, 0)
    assert P("\u0663,\uff11\uff12")[:3] == (to.RECORD, 3, 12)        # Character.digit knows every Nd digit
    assert P("1,2,0x1.8p1")[3] == 0x40400000
    assert P("1,2,\x003\x1f")[3] == 0x40400000                          # String.trim() drops chars <= U+0020
    assert P("1\x00,2")[0] == to.BAD


def test_too_many_bad_lines_is_thrown_by_the_line_after_the_101st():
    ok = b"1,2,3\n"
    bad = b"x\n"
    r = to.read_streams([ok + bad * 101])
    assert r["bad_lines"] == 101 and r["lines"] == 102
    with pytest.raises(to.TooManyBadLines):
        to.read_streams([ok + bad * 101 + b"\n"])
    with pytest.raises(to.TooManyBadLines):
        to.read_streams([ok + bad * 101, b"#\n"])
    assert to.read_streams([b"x\n" + bad * 100])["bad_lines"] == 100    # line 1 is a header


def test_known_items_and_tags():
    r = to.expected([b"1,10,1\n1,11,1\n2,10,0.00001\n1,10,\n3,12,1\n3,12,\n\"t\",10,2\n4,\"s\",1\n"])
    assert r["known"] == {1: [11], 2: [10], to.to_long_id(units("t")): [10], 4: [to.to_long_id(units("s"))]}
    assert r["item_tag_ids"].tolist() == [to.to_long_id(units("t"))] and r["user_tag_ids"].tolist() == [to.to_long_id(units("s"))]
    uid = r["csr_x"][0].tolist()
    assert uid == sorted(r["known"].keys())
    # user 2's only entry is pruned from R (|v| < threshold) but stays a known item
    k = uid.index(2)
    assert r["csr_x"][1][k + 1] - r["csr_x"][1][k] == 0 and r["known_ptr"][k + 1] - r["known_ptr"][k] == 1
