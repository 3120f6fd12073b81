"""The reference's random stream (commons-math3 MersenneTwister as RandomManager seeds it), restated in
myrrix_recommender_amd/random_mt.py and include/myrrix/random.hpp, pinned to what can be pinned without a JVM:
the published output of mt19937ar.c, CPython's own Mersenne Twister, and the two mirrors against each other."""
import math
import os
import random
import subprocess

import numpy as np

from myrrix_recommender_amd.random_mt import MersenneTwister

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_word_stream_is_mt19937ar():
    # mt19937ar.out (Matsumoto & Nishimura): init_by_array({0x123, 0x234, 0x345, 0x456}), first outputs of genrand_int32
    mt = MersenneTwister(0)
    mt.setSeedArray([0x123, 0x234, 0x345, 0x456])
    got = [mt.next(32) for _ in range(10)]
    assert got == [1067595299, 955945823, 477289528, 4107218783, 4228976476, 3344332714, 3355579695, 227628506, 810200273, 2591290167]


def test_long_seed_matches_cpython_twister():
    # setSeed(long) -> init_by_array({hi, lo}); CPython seeds by init_by_array of the 32-bit digits of |seed|, low digit
    # first: the integer hi + lo * 2^32 has the digits [hi, lo] (lo != 0, so that there ARE two digits)
    for seed in (1234567890, 1, 2**40 + 17, 2**63 - 1, -5):
        u = seed & 0xFFFFFFFFFFFFFFFF
        hi, lo = u >> 32, u & 0xFFFFFFFF
        ref = random.Random()
        ref.seed(hi + (lo << 32))
        mt = MersenneTwister(seed)
        assert [mt.next(32) for _ in range(1000)] == [ref.getrandbits(32) for _ in range(1000)], seed


def test_bit_stream_generator_rules():
    mt, ref = MersenneTwister(1234567890), random.Random()
    ref.seed(1234567890 << 32)
    w = [ref.getrandbits(32) for _ in range(64)]
    assert mt.nextDouble() == (((w[0] >> 6) << 26) | (w[1] >> 6)) * 2.0 ** -52
    assert mt.nextInt(1 << 10) == (w[2] >> 1) * (1 << 10) >> 31          # power of two
    assert mt.nextInt(1000) == (w[3] >> 1) % 1000                          # no rejection at this size
    assert mt.nextBoolean() == bool(w[4] >> 31)
    assert mt.nextInt() == (w[5] - (1 << 32) if w[5] >> 31 else w[5])
    x, y = (((w[6] >> 6) << 26) | (w[7] >> 6)) * 2.0 ** -52, (((w[8] >> 6) << 26) | (w[9] >> 6)) * 2.0 ** -52
    g0 = mt.nextGaussian()
    g1 = mt.nextGaussian()                                                  # the cached sine half: no new words
    r = math.sqrt(-2.0 * math.log(y))
    assert g0 == r * math.cos(2.0 * math.pi * x) and g1 == r * math.sin(2.0 * math.pi * x)
    assert mt.nextLong() == ((w[10] << 32) | w[11]) - ((1 << 64) if w[10] >> 31 else 0)


def test_vectorised_draws_equal_scalar_draws():
    a, b = MersenneTwister(42), MersenneTwister(42)
    for n in (1, 2, 5, 30, 31, 64, 7):
        va = a.nextGaussians(n)
        vb = np.array([b.nextGaussian() for _ in range(n)])
        assert np.allclose(va, vb, rtol=1e-15, atol=0)
        assert np.array_equal(a.nextInts(100, 12345), np.array([b.nextInt(12345) for _ in range(100)]))
        assert a.nextDouble() == b.nextDouble()
    # a bound that rejects often: 2^30 + 1
    n = (1 << 30) + 1
    assert np.array_equal(a.nextInts(200, n), np.array([b.nextInt(n) for _ in range(200)]))
    assert a.next(32) == b.next(32)


def test_cpp_mirror_draws_the_same_stream():
    exe = os.path.join(ROOT, "tests", "cpp", "test_random_stream")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "test_random_stream"], stdout=subprocess.DEVNULL)
    out = subprocess.check_output([exe, "1234567890"], text=True).split()
    mt = MersenneTwister(1234567890)
    vals = []
    for _ in range(50):
        vals += [mt.nextInt(), mt.nextInt(1000), mt.nextInt(1 << 12), mt.nextDouble(), mt.nextLong(), int(mt.nextBoolean())]
    got = [float.fromhex(t) if i % 6 == 3 else int(t) for i, t in enumerate(out[:len(vals)])]
    assert got == vals
    gs = [float.fromhex(t) for t in out[len(vals):len(vals) + 101]]
    gp = [mt.nextGaussian() for _ in range(101)]
    assert np.allclose(gs, gp, rtol=4e-16, atol=0)   # libm on both sides; the compilers may contract differently


def test_convergence_sample_starts_a_fresh_generator():
    """ALS:206 calls RandomManager.getRandom() again -- under the test seed a NEW MersenneTwister(1234567890)
    (RandomManager.java:61-64) -- so the ~100 x ~100 sample does not depend on how much of the stream
    constructInitialY (ALS:266, its own generator) consumed."""
    from myrrix_recommender_amd import factorizer
    from myrrix_recommender_amd.random_mt import MersenneTwister
    seed = 1234567890
    users, items = list(range(5000)), list(range(100, 2100))
    tu, ti = factorizer.AlternatingLeastSquares._convergence_sample(seed, users, items)
    fresh = MersenneTwister(seed)
    assert tu == factorizer._choose_about_n(100, users, fresh)
    assert ti == factorizer._choose_about_n(100, items, fresh)       # the same generator goes on to the items (ALS:210-213)
    used = MersenneTwister(seed)
    for _ in range(1000):
        used.nextGaussian()                                          # what a cold start draws first
    assert tu != factorizer._choose_about_n(100, users, used)
    assert 40 < len(tu) < 180 and 40 < len(ti) < 180 and tu == sorted(set(tu))
