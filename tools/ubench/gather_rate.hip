// gather_rate.hip -- the ceiling for gathering 512-byte rows at random from a table of a given size, by a kernel
// that does nothing else: every lane reads 16 bytes (a wave instruction = two whole rows), 8 independent loads in
// flight per lane, 8 waves per SIMD, the values summed into a checksum.  Context: the k = 128 item half of a C5 rank
// gathers from a 51.2 GB replica at 4.85 TB/s of HBM traffic -- how far is that from
// what the memory system gives ANY kernel for this access pattern?
// build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip ; run: ./gather_rate
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ M, const int32_t* __restrict__ idx, int64_t n_idx,
                                                     float* __restrict__ out) {
  const int lane = threadIdx.x & 63, half = lane >> 5, q = lane & 31;   // lanes 0-31: one row, 32-63: the next
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t base = wave * 16; base + 16 <= n_idx; base += n_waves * 16) {
    int32_t r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = idx[base + 2 * u + half];
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(M + (int64_t)r[u] * 128 + 4 * q);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];   // keep the loads alive
}

__global__ void fill_kernel(float* M, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) M[i] = (float)(i & 1023) * 1e-3f;
}

__global__ void index_kernel(int32_t* idx, int64_t n, uint32_t n_rows, int sorted_blocks) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x1234567ull;   // splitmix-style hash
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    idx[i] = (int32_t)(x % n_rows);
  }
  (void)sorted_blocks;
}

int main() {
  const int64_t n_idx = 200000000;   // 102.4 GB gathered per pass
  int32_t* idx = nullptr;
  float* out = nullptr;
  hipMalloc(&idx, sizeof(int32_t) * n_idx);
  hipMalloc(&out, 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (uint32_t n_rows : {400000u, 1000000u, 10000000u, 100000000u}) {
    float* M = nullptr;
    if (hipMalloc(&M, sizeof(float) * 128 * (size_t)n_rows) != hipSuccess) return 1;
    fill_kernel<<<4096, 256>>>(M, (int64_t)n_rows * 128);
    index_kernel<<<4096, 256>>>(idx, n_idx, n_rows, 0);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      gather_kernel<<<256 * 8, 256>>>(M, idx, n_idx, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    std::printf("table %7.2f GB: %lld random 512-byte rows in %.2f ms = %.2f TB/s\n", (double)n_rows * 512 / 1e9, (long long)n_idx, ms,
                (double)n_idx * 512 / (ms * 1e-3) / 1e12);
    hipFree(M);
  }
  return 0;
}
