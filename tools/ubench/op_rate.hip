// Issue cost per wave-instruction of the VALU operations the ALS kernels are made of (gfx950, 3 waves/SIMD).
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#define OP8(stmt) _Pragma("unroll") for (int i = 0; i < 8; ++i) { stmt; }
template <int MODE>
__global__ void k(float* out, int iters, float kk) {
  float a[8], b[8];
  unsigned long long mask = 0x00ff00ff00ff00ffull ^ (unsigned long long)iters;
  int sg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 1.0f + i * 1e-3f; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) OP8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(kk)))
    if (MODE == 1) OP8(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 2) OP8(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 3) OP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 16) OP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(mask)))
    if (MODE == 17) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[0]), "v"(b[0]) : "vcc"); OP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]))) }
    if (MODE == 18) OP8(asm volatile("v_readlane_b32 %0, %1, 5\n\ts_nop 3" : "=s"(sg[i]) : "v"(a[i])))
    if (MODE == 4) OP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 5) OP8(asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,1]" : "+v"(a[i]) : "v"(b[i]), "v"(kk)))
    if (MODE == 6) OP8(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 7) OP8(asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 8) OP8(asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(kk)))
    if (MODE == 9) OP8(asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i])))
    if (MODE == 10) OP8(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 11) OP8(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(kk)))
    if (MODE == 12) OP8(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&a[i & 6])) : "v"(*reinterpret_cast<double*>(&b[i & 6]))))
    if (MODE == 13) OP8(asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i])))
    if (MODE == 14) OP8(asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i])))
    if (MODE == 15) OP8(asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + b[i] + sg[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* d, int iters) {
  const int threads = 768;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, 10, 0.5f);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, iters, 0.5f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-24s %6.2f ns per instruction per SIMD\n", name, ms * 1e6 / iters / 8.0 / 3.0);
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  const int it = 100000;
  run<0>("v_fma_f32", d, it); run<11>("v_fmac_f32", d, it); run<1>("v_mul_f32", d, it); run<2>("v_add_f32", d, it);
  run<10>("v_max_f32", d, it); run<8>("v_min3_f32", d, it); run<3>("v_cndmask_b32", d, it); run<7>("v_mov_b32", d, it);
  run<6>("v_xor_b32", d, it); run<15>("v_and_b32", d, it); run<4>("v_cvt_pkrtz_f16_f32", d, it); run<13>("v_cvt_f32_f16", d, it);
  run<5>("v_fma_mix_f32", d, it); run<12>("v_pk_mul_f32", d, it); run<16>("v_cndmask_b32_e64 sgpr", d, it); run<17>("v_cmp + 8 v_cndmask vcc", d, it); run<18>("v_readlane + s_nop 3", d, it); run<9>("v_rcp_f32", d, it); run<14>("v_sqrt_f32", d, it);
  return 0;
}
