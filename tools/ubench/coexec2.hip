// Do matrix-pipe instructions of one wave and VALU instructions of ANOTHER wave on the same SIMD
// overlap?  Role is decided once, outside the loops (no per-iteration branches).
// 512-thread blocks = 2 waves/SIMD: waves 0-3 role A, waves 4-7 role B.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// KIND: 0 = idle, 1 = f32 MFMA 16x16x4, 2 = f16 MFMA 16x16x32, 3 = VALU fma
template <int KIND>
__device__ float work(int iters, float x, float y) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(x + i); hb[i] = (_Float16)(y - i); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (KIND == 1) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      } else if (KIND == 2) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, a3, 0, 0, 0);
      } else if (KIND == 4) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 qa = {ha[0], ha[1], ha[2], ha[3]}, qb = {hb[0], hb[1], hb[2], hb[3]};
        a0 = __builtin_amdgcn_mfma_f32_16x16x16f16(qa, qb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x16f16(qa, qb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x16f16(qa, qb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x16f16(qa, qb, a3, 0, 0, 0);
      } else if (KIND == 3) {
        v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x);
        v4 = fmaf(v4, y, x); v5 = fmaf(v5, y, x); v6 = fmaf(v6, y, x); v7 = fmaf(v7, y, x);
        v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x);
        v4 = fmaf(v4, y, x); v5 = fmaf(v5, y, x); v6 = fmaf(v6, y, x); v7 = fmaf(v7, y, x);
      }
    }
  }
  return a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int A, int B>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float r;
  if (wave < 4) r = work<A>(iters, x, y);
  else r = work<B>(iters, x, y);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int A, int B>
void run(const char* name, float* d, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-44s %8.3f ms -> %6.1f ns per u-iter (4 MFMA | 16 VALU per wave)\n", name, ms, ms * 1e6 / (iters * 4.0));
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int it = 200000;
  run<4, 0>("f16 MFMA 16x16x16 | idle", d, it);
  run<1, 0>("f32 MFMA | idle", d, it);
  run<2, 0>("f16 MFMA | idle", d, it);
  run<3, 0>("VALU | idle", d, it);
  run<1, 1>("f32 MFMA | f32 MFMA", d, it);
  run<2, 2>("f16 MFMA | f16 MFMA", d, it);
  run<3, 3>("VALU | VALU", d, it);
  run<1, 3>("f32 MFMA | VALU", d, it);
  run<2, 3>("f16 MFMA | VALU", d, it);
  run<1, 2>("f32 MFMA | f16 MFMA", d, it);
  return 0;
}
