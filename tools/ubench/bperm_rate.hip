// What does a cross-lane move cost on gfx950?  ds_bpermute_b32 (through the LDS crossbar, one per CU),
// a DPP row_newbcast operand, v_permlane32_swap / v_permlane16_swap (VALU), and whether bpermutes
// overlap with VALU work of the same wave / other waves of the CU.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
  const int idx = ((lane ^ 16) << 2);
  const float y = 1.0001f, x = 0.5f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 3) {  // 16 independent bpermutes
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(a[i])));
    }
    if (MODE == 1) {  // 16 multiplies with a DPP row_newbcast source
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = a[i] * __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[(i + 1) & 15]), 0x150 + 3, 0xf, 0xf, false));
    }
    if (MODE == 2) {  // 16 permlane32 swaps (8 pairs twice)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; i += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1]));
    }
    if (MODE == 5) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; i += 2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1]));
    }
    if (MODE == 3 || MODE == 4) {  // 32 plain FMAs
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], y, x);
    }
    if (MODE == 7 || MODE == 8 || MODE == 9) {  // the same 16 values through explicit LDS: per-wave region, in-order LDS queue
      extern __shared__ float lds[];
      float* mine = lds + (threadIdx.x >> 6) * 256;
      const int gsrc = (it & 3) * 16 + (lane & 15);
      if (MODE == 7) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          *reinterpret_cast<volatile f2*>(mine + 2 * lane) = f2{a[i], a[i + 1]};
          const f2 v = *reinterpret_cast<volatile f2*>(mine + 2 * gsrc);
          a[i] = v.x; a[i + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
          if (MODE == 8 || (lane >> 4) == (it & 3)) *reinterpret_cast<volatile f4*>(mine + 4 * (MODE == 8 ? lane : (lane & 15))) = f4{a[i], a[i + 1], a[i + 2], a[i + 3]};
          const f4 v = *reinterpret_cast<volatile f4*>(mine + 4 * (MODE == 8 ? gsrc : (lane & 15)));
          a[i] = v.x; a[i + 1] = v.y; a[i + 2] = v.z; a[i + 3] = v.w;
        }
      }
    }
    if (MODE == 6) {  // 16 v_readlane + v_mov from sgpr (broadcast through SGPR)
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = a[i] + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[(i + 1) & 15]), 17));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int threads, float* d, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 16384, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 16384, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-44s %d waves/SIMD %8.3f ms -> %7.1f ns per iteration\n", name, threads / 256, ms, ms * 1e6 / iters);
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  const int it = 100000;
  for (int threads : {256, 768}) {
    run<0>("16 ds_bpermute", threads, d, it);
    run<1>("16 v_mul dpp row_newbcast", threads, d, it);
    run<2>("16 v_permlane32_swap", threads, d, it);
    run<5>("16 v_permlane16_swap", threads, d, it);
    run<6>("16 v_readlane + v_add sgpr", threads, d, it);
    run<7>("8 x (ds_write_b64 + ds_read_b64 bcast)", threads, d, it);
    run<8>("4 x (ds_write_b128 + ds_read_b128 bcast)", threads, d, it);
    run<9>("4 x (16-lane ds_write_b128 + ds_read_b128)", threads, d, it);
    run<4>("32 v_fma", threads, d, it);
    run<3>("16 ds_bpermute + 32 v_fma", threads, d, it);
  }
  return 0;
}
