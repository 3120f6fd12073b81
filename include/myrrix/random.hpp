// random.hpp -- the reference's random stream for the C++ mirror: org.apache.commons.math3.random.MersenneTwister as
// net.myrrix.common.random.RandomManager creates it (RandomManager.java:63-73; test seed 1234567890, RM:52).
// commons-math3 3.2 is an un-vendored dependency of the reference; restated from its published algorithm:
//   MersenneTwister      MT19937 (mt19937ar.c of Matsumoto & Nishimura): setSeed(long) = init_by_array({hi, lo}),
//                        next(bits) = tempered word >>> (32 - bits)
//   BitsStreamGenerator  nextDouble = ((long) next(26) << 26 | next(26)) * 2^-52; nextInt(n) = java.util.Random's
//                        algorithm on next(31); nextLong = next(32) << 32 | next(32); nextGaussian = Box-Muller on two
//                        nextDouble()s (r cos, the r sin half cached)
// Pinned to mt19937ar's published output and to the Python mirror (tests/test_random_stream.py); a Gaussian can
// differ from the JVM's in its last bit (libm here, FastMath there).  PARITY UNPINNED against a JVM.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>

namespace myrrix {

class MersenneTwister {
 public:
  explicit MersenneTwister(int64_t seed = 1234567890LL) { setSeed(seed); }

  void setSeed(int64_t seed) {
    const uint64_t s = (uint64_t)seed;
    const uint32_t key[2] = {(uint32_t)(s >> 32), (uint32_t)(s & 0xffffffffu)};
    setSeed(key, 2);
  }
  void setSeed(const uint32_t* key, int len) {   // init_by_array
    init(19650218u);
    int i = 1, j = 0;
    for (int kk = N > len ? N : len; kk; --kk) {
      mt_[i] = (mt_[i] ^ ((mt_[i - 1] ^ (mt_[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      if (++i >= N) { mt_[0] = mt_[N - 1]; i = 1; }
      if (++j >= len) j = 0;
    }
    for (int kk = N - 1; kk; --kk) {
      mt_[i] = (mt_[i] ^ ((mt_[i - 1] ^ (mt_[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      if (++i >= N) { mt_[0] = mt_[N - 1]; i = 1; }
    }
    mt_[0] = 0x80000000u;
    nextGaussian_ = std::numeric_limits<double>::quiet_NaN();
  }

  uint32_t next(int bits) {
    if (mti_ >= N) twist();
    uint32_t y = mt_[mti_++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y >> (32 - bits);
  }
  int32_t nextInt() { return (int32_t)next(32); }
  int32_t nextInt(int32_t n) {
    if (n <= 0) throw std::invalid_argument("n must be strictly positive");
    if ((n & -n) == n) return (int32_t)(((int64_t)n * (int64_t)next(31)) >> 31);
    for (;;) {
      const int64_t bits = next(31), val = bits % n;
      if (bits - val + (n - 1) < (int64_t(1) << 31)) return (int32_t)val;   // Java: no int overflow
    }
  }
  int64_t nextLong() {
    const uint64_t hi = next(32), lo = next(32);
    return (int64_t)((hi << 32) | lo);
  }
  bool nextBoolean() { return next(1) != 0; }
  float nextFloat() { return (float)next(23) * 0x1.0p-23f; }
  double nextDouble() {
    const uint64_t high = (uint64_t)next(26) << 26, low = next(26);
    return (double)(high | low) * 0x1.0p-52;
  }
  double nextGaussian() {
    if (!std::isnan(nextGaussian_)) {
      const double g = nextGaussian_;
      nextGaussian_ = std::numeric_limits<double>::quiet_NaN();
      return g;
    }
    const double x = nextDouble(), y = nextDouble();
    const double alpha = 2.0 * 3.141592653589793 * x;
    const double r = std::sqrt(-2.0 * std::log(y));
    nextGaussian_ = r * std::sin(alpha);
    return r * std::cos(alpha);
  }

 private:
  static constexpr int N = 624, M = 397;
  uint32_t mt_[N];
  int mti_ = N;
  double nextGaussian_ = std::numeric_limits<double>::quiet_NaN();

  void init(uint32_t s) {
    mt_[0] = s;
    for (int i = 1; i < N; ++i) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + (uint32_t)i;
    mti_ = N;
  }
  void twist() {
    for (int kk = 0; kk < N; ++kk) {
      const uint32_t y = (mt_[kk] & 0x80000000u) | (mt_[(kk + 1) % N] & 0x7fffffffu);
      mt_[kk] = mt_[(kk + M) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    mti_ = 0;
  }
};

}  // namespace myrrix
