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
    if (MODE == 24) OP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 25) { unsigned long long m2; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(a[0]), "v"(b[0])); OP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(m2))) }
    if (MODE == 26) { asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc"); OP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]))) }
    if (MODE == 27) OP8(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7])))
    if (MODE == 28) OP8(asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" :: "v"(a[i]), "v"(b[i]) : "vcc"))
    if (MODE == 29) { unsigned long long m2; OP8(asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(a[i]), "v"(b[i]))) }
    if (MODE == 30) OP8(asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc"))
    if (MODE == 31) OP8(asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc"))
    if (MODE == 32) OP8(asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(*reinterpret_cast<double*>(&a[i & 6])) : "v"(*reinterpret_cast<double*>(&b[i & 6]))))
    if (MODE == 33) OP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*reinterpret_cast<double*>(&a[i & 6])) : "v"(b[i]), "v"(kk) : "vcc"))
    if (MODE == 34) OP8(asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 35) OP8(asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]), "v"(kk)))
    if (MODE == 36) OP8(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 37) OP8(asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[i])))
    if (MODE == 38) OP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&a[i & 6])) : "v"(*reinterpret_cast<double*>(&b[i & 6]))))
    if (MODE == 39) OP8(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
    if (MODE == 19) OP8(asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(a[i]) : "v"(b[i]), "v"(kk)))
    if (MODE == 20) OP8(asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(a[i]) : "v"(b[i]), "v"(kk)))
    if (MODE == 21) OP8(asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(a[i]) : "v"(b[i]), "v"(kk), "v"(b[(i + 1) & 7])))
    // one entry pair of the split-precision conversion, as the kernels do it (8 instructions incl. the two RHS FMAs) ...
    if (MODE == 22) OP8(asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4\n\tv_cvt_pkrtz_f16_f32 %2, %0, %1\n\t"
                                     "v_fma_mix_f32 %0, %2, -1.0, %0 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                                     "v_cvt_pkrtz_f16_f32 %3, %0, %1\n\tv_fmac_f32 %0, %4, %2\n\tv_fmac_f32 %1, %4, %3"
                                     : "+v"(a[i]), "+v"(b[i]), "+v"(a[(i + 1) & 7]), "+v"(b[(i + 1) & 7]) : "v"(kk)))
    // ... and on v_fma_mixlo/mixhi_f16 (hi = RNE(y s), lo = RNE(y s - hi) from the exact product; 6 instructions)
    if (MODE == 23) OP8(asm volatile("v_fma_mixlo_f16 %0, %2, %4, 0\n\tv_fma_mixhi_f16 %0, %3, %4, 0\n\t"
                                     "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                                     "v_fmac_f32 %2, %4, %3\n\tv_fmac_f32 %3, %4, %2"
                                     : "+v"(a[i]), "+v"(b[i]), "+v"(a[(i + 1) & 7]), "+v"(b[(i + 1) & 7]) : "v"(kk)))
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
  run<24>("v_cndmask_b32_e64 vcc", d, it); run<25>("v_cmp_e64 sgpr + 8 cndmask", d, it); run<26>("s_mov vcc + 8 cndmask vcc", d, it); run<27>("v_cndmask vcc, dst != src", d, it);
  run<28>("v_cmp_gt_f32_e32 (vcc)", d, it); run<29>("v_cmp_gt_f32_e64 (sgpr)", d, it); run<30>("v_add_co_u32_e32", d, it); run<31>("v_addc_co_u32_e32", d, it);
  run<32>("v_lshl_add_u64", d, it); run<33>("v_mad_u64_u32", d, it); run<34>("v_mov_b32_dpp", d, it); run<35>("v_fmac_f32_dpp", d, it);
  run<36>("v_add_u32", d, it); run<37>("v_lshlrev_b32", d, it); run<38>("v_pk_fma_f32", d, it);
  run<39>("v_cvt_pk_f16_f32 (RNE)", d, it);
  run<19>("v_fma_mixlo_f16", d, it); run<20>("v_fma_mixhi_f16", d, it); run<21>("v_fma_mixlo_f16 f16 src2", d, it);
  run<22>("pair: mul/cvt/mix/cvt (8)", d, it); run<23>("pair: mixlo/mixhi (6)", d, it);
  return 0;
}
