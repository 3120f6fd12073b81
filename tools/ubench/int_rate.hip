// Issue cost of the integer instructions a gather address can be built from (gfx950).
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
template <int MODE>
__global__ void k(unsigned long long* out, int iters, unsigned kk) {
  unsigned a[8];
  unsigned long long acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7 + i; acc[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(kk) : "vcc");
      if (MODE == 1) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
      if (MODE == 2) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(kk), "v"(a[(i + 1) & 7]));
      if (MODE == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(kk));
      if (MODE == 4) asm volatile("v_lshl_add_u32 %0, %0, 6, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
      if (MODE == 5) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : "vcc");
    }
  }
  unsigned long long s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i] + a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int threads, unsigned long long* d, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, 10, 64u);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, iters, 64u);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-24s %d waves/SIMD %8.3f ms -> %6.2f ns per instruction per SIMD\n", name, threads / 256, ms, ms * 1e6 / iters / 8.0 / (threads / 256));
}
int main() {
  unsigned long long* d; hipMalloc(&d, 256 * 1024 * 8);
  const int it = 100000;
  for (int threads : {256, 768}) {
    run<0>("v_mad_u64_u32", threads, d, it);
    run<1>("v_lshl_add_u64", threads, d, it);
    run<2>("v_mad_u32_u24", threads, d, it);
    run<3>("v_mul_lo_u32", threads, d, it);
    run<4>("v_lshl_add_u32", threads, d, it);
    run<5>("v_add_co_u32", threads, d, it);
  }
  return 0;
}
