"""The random stream of the reference: org.apache.commons.math3.random.MersenneTwister as RandomManager creates it
(common/src/net/myrrix/common/random/RandomManager.java:63-73, test seed 1234567890 at RM:52).

commons-math3 (3.2, an un-vendored dependency of the reference) is restated from its published algorithm:
  * MersenneTwister = MT19937 of Matsumoto & Nishimura (mt19937ar.c): setSeed(long) -> init_by_array({hi, lo}),
    next(bits) = tempered word >>> (32 - bits);
  * BitsStreamGenerator: nextDouble = ((long) next(26) << 26 | next(26)) * 2^-52, nextFloat = next(23) * 2^-23,
    nextInt(n) = java.util.Random's algorithm on next(31), nextLong = next(32) << 32 | next(32) & 0xffffffff,
    nextGaussian = Box-Muller on two nextDouble()s (r cos, cached r sin).
The 32-bit word stream comes from numpy's MT19937 bit generator seeded through init_by_array (the same published
algorithm, an independent implementation); tests/test_random_stream.py pins it to mt19937ar's published output and to
CPython's own Mersenne Twister.  nextGaussian uses libm's log / cos / sin where commons-math uses FastMath's (both
within 1 ulp of the true value, not necessarily the same double): a Gaussian can differ from the JVM's in its last
bit, which the cast to float of RandomUtils.java:93 almost always removes.  PARITY UNPINNED against a JVM: there is
none here, and the reference's tests hold no vector of its random stream.
"""
import math

import numpy as np

_TWO_M52 = 2.0 ** -52
_TWO_M23 = 2.0 ** -23


class MersenneTwister:
    def __init__(self, seed=None):
        self._bg = np.random.MT19937()
        self._buf = np.zeros(0, dtype=np.uint64)
        self._pos = 0
        self._next_gaussian = math.nan
        self.setSeed(1234567890 if seed is None else seed)

    # -- seeding (MersenneTwister.setSeed(long) -> setSeed(int[]{(int)(seed >>> 32), (int)(seed & 0xffffffffL)})) ------
    def setSeed(self, seed):
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.setSeedArray([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF])

    def setSeedArray(self, key):
        self._bg._legacy_seeding(np.asarray(key, dtype=np.uint32))   # init_by_array of mt19937ar.c
        self._buf = np.zeros(0, dtype=np.uint64)
        self._pos = 0
        self._next_gaussian = math.nan                                 # BitsStreamGenerator.clear()

    # -- the 32-bit word stream ---------------------------------------------------------------------------------------
    def _words(self, n):
        """The next n tempered 32-bit words as a uint64 array (consumed)."""
        have = len(self._buf) - self._pos
        if have < n:
            more = self._bg.random_raw(max(4096, n - have))
            self._buf = np.concatenate([self._buf[self._pos:], more])
            self._pos = 0
        out = self._buf[self._pos:self._pos + n]
        self._pos += n
        return out

    def next(self, bits):
        return int(self._words(1)[0]) >> (32 - bits)

    # -- BitsStreamGenerator ------------------------------------------------------------------------------------------
    def nextInt(self, n=None):
        if n is None:
            v = self.next(32)
            return v - (1 << 32) if v & 0x80000000 else v
        if n <= 0:
            raise ValueError("n must be strictly positive")
        if n & -n == n:                                    # power of two
            return (n * self.next(31)) >> 31
        while True:
            bits = self.next(31)
            val = bits % n
            if bits - val + (n - 1) < (1 << 31):           # no int overflow
                return val

    def nextInts(self, count, n):
        """count calls of nextInt(n), vectorised (a rejection, 1 in 2^31 / n, falls back to the scalar loop)."""
        w = self._words(count) >> np.uint64(1)
        if n & -n == n:
            return ((np.uint64(n) * w) >> np.uint64(31)).astype(np.int64)
        val = w % np.uint64(n)
        if np.all(w - val + np.uint64(n - 1) < np.uint64(1 << 31)):
            return val.astype(np.int64)
        self._pos -= count                                  # un-read them: the scalar loop consumes what it needs
        return np.array([self.nextInt(n) for _ in range(count)], dtype=np.int64)

    def nextLong(self):
        hi, lo = (int(x) for x in self._words(2))
        v = (hi << 32) | lo
        return v - (1 << 64) if v & (1 << 63) else v

    def nextBoolean(self):
        return self.next(1) != 0

    def nextFloat(self):
        return float(np.float32(self.next(23) * _TWO_M23))

    def nextDouble(self):
        a, b = (int(x) for x in self._words(2))
        return (((a >> 6) << 26) | (b >> 6)) * _TWO_M52

    def nextDoubles(self, count):
        w = self._words(2 * count)
        return (((w[0::2] >> np.uint64(6)) << np.uint64(26)) | (w[1::2] >> np.uint64(6))).astype(np.float64) * _TWO_M52

    def nextGaussian(self):
        if not math.isnan(self._next_gaussian):
            g, self._next_gaussian = self._next_gaussian, math.nan
            return g
        x, y = self.nextDouble(), self.nextDouble()
        alpha = 2.0 * math.pi * x
        r = math.sqrt(-2.0 * math.log(y)) if y > 0.0 else math.inf
        self._next_gaussian = r * math.sin(alpha)
        return r * math.cos(alpha)

    def nextGaussians(self, count):
        """count calls of nextGaussian() in stream order (the cached second value of a pair included)."""
        out = np.empty(count, dtype=np.float64)
        i = 0
        if count and not math.isnan(self._next_gaussian):
            out[0], self._next_gaussian = self._next_gaussian, math.nan
            i = 1
        pairs = (count - i + 1) // 2
        if pairs:
            d = self.nextDoubles(2 * pairs)
            alpha = 2.0 * math.pi * d[0::2]
            with np.errstate(divide="ignore"):
                r = np.sqrt(-2.0 * np.log(d[1::2]))
            both = np.empty(2 * pairs, dtype=np.float64)
            both[0::2] = r * np.cos(alpha)
            both[1::2] = r * np.sin(alpha)
            take = count - i
            out[i:] = both[:take]
            if take < 2 * pairs:
                self._next_gaussian = float(both[take])
        return out
