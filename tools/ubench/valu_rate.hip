// Microbenchmark: issue cost of plain, packed and DPP VALU ops, and of ds_bpermute, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { v[i] = x + i; p[i] = f32x2{x + i, x - i}; }
  const f32x2 yy = {y, y}, xx = {x, x};
  const int lane4 = ((threadIdx.x & 63) ^ 16) << 2;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MODE == 0) v[j] = fmaf(v[j], y, x);
        if (MODE == 1) p[j] = __builtin_elementwise_fma(p[j], yy, xx);
        if (MODE == 2) v[j] = fmaf(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[j]), 0x150 + 3, 0xf, 0xf, false)), y, v[j]);
        if (MODE == 3) v[j] = __int_as_float(__builtin_amdgcn_ds_bpermute(lane4, __float_as_int(v[j]))) + x;
        if (MODE == 4) v[j] = (threadIdx.x & (1 << j)) ? v[j] * y : x;   // mul + cndmask
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int threads, float* d, int iters, int ops_per_iter) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, 10);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-34s waves/SIMD %d : %6.2f ns per op (%5.2f cycles @2.3GHz)\n", name, threads / 256, ms * 1e6 / ((double)iters * ops_per_iter), ms * 1e6 / ((double)iters * ops_per_iter) * 2.3);
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
  const int it = 100000;
  for (int t = 256; t <= 1024; t *= 2) {
    run<0>("v_fma_f32", t, d, it, 32);
    run<1>("v_pk_fma_f32", t, d, it, 32);
    run<2>("v_fmac_f32_dpp (row_newbcast)", t, d, it, 32);
    run<3>("ds_bpermute_b32 + v_add", t, d, it, 32);
    run<4>("v_mul + v_cndmask", t, d, it, 32);
  }
  return 0;
}
