// prints draws of myrrix::MersenneTwister (include/myrrix/random.hpp) for tests/test_random_stream.py
#include <cstdio>
#include <cstdlib>

#include "../../include/myrrix/random.hpp"

int main(int argc, char** argv) {
  myrrix::MersenneTwister mt(argc > 1 ? std::atoll(argv[1]) : 1234567890LL);
  for (int i = 0; i < 50; ++i) {
    std::printf("%d\n", mt.nextInt());
    std::printf("%d\n", mt.nextInt(1000));
    std::printf("%d\n", mt.nextInt(1 << 12));
    std::printf("%a\n", mt.nextDouble());
    std::printf("%lld\n", (long long)mt.nextLong());
    std::printf("%d\n", mt.nextBoolean() ? 1 : 0);
  }
  for (int i = 0; i < 101; ++i) std::printf("%a\n", mt.nextGaussian());
  return 0;
}
