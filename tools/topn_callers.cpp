// topn_callers.cpp -- bench harness (tools/bench_topn.py, tests/test_gpu_topn_serving.py): N native threads, each making
// single-user mals_recommend calls on ONE handle, the way the reference's request threads enter ServerRecommender.recommend
// (ServerRecommender.java:359-441).  Python threads would measure the interpreter lock, not the library.
// Build: g++ -O2 -shared -fPIC -pthread tools/topn_callers.cpp -Iinclude -Lmyrrix-recommender_amd/csrc -lmyrrix_als \
//        -Wl,-rpath,'$ORIGIN/../myrrix-recommender_amd/csrc' -o tools/libtopn_callers.so
#include "myrrix_als.h"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>

extern "C" int topn_callers_run(void* handle, int32_t n_threads, int32_t calls_per_thread, int64_t n_users, int32_t how_many, uint64_t seed,
                                double* latencies_us, double* wall_s, int64_t* checksum) {
  mals_handle h = static_cast<mals_handle>(handle);
  std::atomic<int> ready{0}, failed{0};
  std::atomic<bool> go{false};
  std::atomic<long long> sum{0};
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&, t] {
      uint64_t x = seed * 0x9E3779B97F4A7C15ull + (uint64_t)t * 0xD1B54A32D192ED03ull + 1;
      std::vector<int64_t> items((size_t)how_many);
      std::vector<float> scores((size_t)how_many);
      long long local = 0;
      ready.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      for (int c = 0; c < calls_per_thread; ++c) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int64_t user = (int64_t)(x % (uint64_t)n_users);
        int32_t cnt = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = mals_recommend(h, &user, 1, how_many, 0, items.data(), scores.data(), &cnt);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc != MALS_OK) {
          failed.store(rc);
          break;
        }
        latencies_us[(size_t)t * calls_per_thread + c] = std::chrono::duration<double, std::micro>(t1 - t0).count();
        for (int j = 0; j < cnt; ++j) local += items[(size_t)j] * (j + 1);
      }
      sum.fetch_add(local);
    });
  while (ready.load() < n_threads) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (std::thread& t : th) t.join();
  *wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (checksum) *checksum = sum.load();
  return failed.load();
}
