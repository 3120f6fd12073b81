// Microbenchmark: do fp32-input MFMAs and plain VALU co-execute on one SIMD?
// Launch 256 CUs x (1 block of 256 threads = 1 wave/SIMD) or 512 threads (2 waves/SIMD).
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// mode 0: MFMA f32 only; 1: VALU only; 2: interleaved 1 MFMA : R VALU in one wave;
// 3: waves 0-3 MFMA-only, waves 4-7 VALU-only (2 waves/SIMD); 4: bf16 MFMA only; 5: bf16 MFMA + VALU interleaved
template <int MODE, int R>
__global__ void k(float* out, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
  bf16x8 ba = {1,2,3,4,5,6,7,8}, bb = {8,7,6,5,4,3,2,1};
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 2 || MODE == 5 || (MODE == 3 && wave >= 4);
  const bool do_bf = MODE == 4 || MODE == 5;
  if (MODE == 6 || MODE == 7) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#define ONE(acc, va, vb)                                                                        \
        if (MODE == 6) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);         \
        else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc, 0, 0, 0);              \
        _Pragma("unroll") for (int r = 0; r < R; ++r) { va = fmaf(va, y, x); vb = fmaf(vb, y, x); }
        ONE(a0, v0, v1) ONE(a1, v2, v3) ONE(a2, v4, v5) ONE(a3, v6, v7)
      }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    return;
  }
  const bool do_bf8 = MODE == 8 && wave < 4;
  const bool do_valu8 = MODE == 8 && wave >= 4;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (do_bf8) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, a3, 0, 0, 0);
      }
      if (do_valu8) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x);
          v4 = fmaf(v4, y, x); v5 = fmaf(v5, y, x); v6 = fmaf(v6, y, x); v7 = fmaf(v7, y, x);
        }
      }
      if (do_mfma) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      }
      if (do_bf) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, a3, 0, 0, 0);
      }
      if (do_valu) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x);
          v4 = fmaf(v4, y, x); v5 = fmaf(v5, y, x); v6 = fmaf(v6, y, x); v7 = fmaf(v7, y, x);
        }
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int MODE, int R>
void run(const char* name, int threads, float* d, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(threads), 0, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per inner u-iteration: 4 MFMA and/or 8R VALU per wave
  printf("%-46s threads %3d  %8.3f ms  -> %7.1f ns per u-iter (4 MFMA / %d VALU)\n", name, threads, ms, ms * 1e6 / (iters * 4.0), 8 * R);
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int it = 200000;
  run<0, 1>("f32 MFMA only (1 wave/SIMD)", 256, d, it);
  run<1, 1>("VALU only 8 fma (1 wave/SIMD)", 256, d, it);
  run<1, 2>("VALU only 16 fma (1 wave/SIMD)", 256, d, it);
  run<2, 1>("f32 MFMA + 8 fma interleaved, same wave", 256, d, it);
  run<2, 2>("f32 MFMA + 16 fma interleaved, same wave", 256, d, it);
  run<3, 2>("f32 MFMA waves + VALU(16) waves, 2 waves/SIMD", 512, d, it);
  run<0, 1>("f32 MFMA only (2 waves/SIMD)", 512, d, it);
  run<1, 2>("VALU only 16 fma (2 waves/SIMD)", 512, d, it);
  run<4, 1>("bf16 MFMA only (1 wave/SIMD)", 256, d, it);
  run<5, 1>("bf16 MFMA + 8 fma interleaved, same wave", 256, d, it);
  run<5, 2>("bf16 MFMA + 16 fma interleaved, same wave", 256, d, it);
  run<5, 4>("bf16 MFMA + 32 fma interleaved, same wave", 256, d, it);
  run<6, 1>("f32 MFMA, 2 fma after EACH mfma (8/iter)", 256, d, it);
  run<6, 2>("f32 MFMA, 4 fma after EACH mfma (16/iter)", 256, d, it);
  run<6, 4>("f32 MFMA, 8 fma after EACH mfma (32/iter)", 256, d, it);
  run<7, 1>("bf16 MFMA, 2 fma after EACH mfma (8/iter)", 256, d, it);
  run<7, 2>("bf16 MFMA, 4 fma after EACH mfma (16/iter)", 256, d, it);
  run<7, 4>("bf16 MFMA, 8 fma after EACH mfma (32/iter)", 256, d, it);
  run<8, 1>("bf16 MFMA waves + VALU(8) waves, 2 waves/SIMD", 512, d, it);
  run<8, 2>("bf16 MFMA waves + VALU(16) waves, 2 waves/SIMD", 512, d, it);
  run<8, 4>("bf16 MFMA waves + VALU(32) waves, 2 waves/SIMD", 512, d, it);
  run<3, 1>("f32 MFMA waves + VALU(8) waves, 2 waves/SIMD", 512, d, it);
  run<3, 4>("f32 MFMA waves + VALU(32) waves, 2 waves/SIMD", 512, d, it);
  return 0;
}
