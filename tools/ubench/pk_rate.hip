// Is v_pk_fma_f32 (2 fp32 FMAs per lane) issued at the rate of a plain v_fma_f32 on gfx950?
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  f32x2 a0 = {x, x + 1}, a1 = {x + 2, x + 3}, a2 = {x + 4, x + 5}, a3 = {x + 6, x + 7}, a4 = {x + 8, x + 9}, a5 = {x + 10, x + 11},
        a6 = {x + 12, x + 13}, a7 = {x + 14, x + 15};
  const f32x2 yy = {y, y}, xx = {x, x};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) {  // 16 plain fma
        a0[0] = fmaf(a0[0], y, x); a0[1] = fmaf(a0[1], y, x); a1[0] = fmaf(a1[0], y, x); a1[1] = fmaf(a1[1], y, x);
        a2[0] = fmaf(a2[0], y, x); a2[1] = fmaf(a2[1], y, x); a3[0] = fmaf(a3[0], y, x); a3[1] = fmaf(a3[1], y, x);
        a4[0] = fmaf(a4[0], y, x); a4[1] = fmaf(a4[1], y, x); a5[0] = fmaf(a5[0], y, x); a5[1] = fmaf(a5[1], y, x);
        a6[0] = fmaf(a6[0], y, x); a6[1] = fmaf(a6[1], y, x); a7[0] = fmaf(a7[0], y, x); a7[1] = fmaf(a7[1], y, x);
      } else {          // 8 packed fma = the same 16 FMAs
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(yy), "v"(xx));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(yy), "v"(xx));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(yy), "v"(xx));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(yy), "v"(xx));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(yy), "v"(xx));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(yy), "v"(xx));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(yy), "v"(xx));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(yy), "v"(xx));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a0[1] + a1[0] + a1[1] + a2[0] + a2[1] + a3[0] + a3[1] + a4[0] + a4[1] + a5[0] + a5[1] + a6[0] + a6[1] + a7[0] + a7[1];
}
template <int MODE>
void run(const char* name, int threads, float* d, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-40s threads %3d %8.3f ms -> %6.2f ns per 16 FMAs per wave\n", name, threads, ms, ms * 1e6 / (iters * 4.0));
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  const int it = 200000;
  run<0>("16 v_fma_f32 (1 wave/SIMD)", 256, d, it);
  run<1>("8 v_pk_fma_f32 (1 wave/SIMD)", 256, d, it);
  run<0>("16 v_fma_f32 (2 waves/SIMD)", 512, d, it);
  run<1>("8 v_pk_fma_f32 (2 waves/SIMD)", 512, d, it);
  run<0>("16 v_fma_f32 (4 waves/SIMD)", 1024, d, it);
  run<1>("8 v_pk_fma_f32 (4 waves/SIMD)", 1024, d, it);
  return 0;
}
