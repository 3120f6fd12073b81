// factor_diag (one tile per wave) against factor_diag_quad (the four waves of a workgroup, one of them factoring the four
// tiles): correctness on random SPD tiles and throughput at a given occupancy, with and without independent matrix work
// between factorizations.   usage: quad_diag [workgroups_per_cu=2] [mfma_per_iter=64] [reps=200]
#include "../../myrrix-recommender_amd/csrc/als_kernels.h"
#include "quad_factor.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include <cstdlib>
using namespace mals;

template <bool QUAD>
__global__ __launch_bounds__(256) void k(const float* in, float* out, float* out_piv, int reps, int work, int skew) {
  extern __shared__ char dyn[];   // sized by the host to set the occupancy; the exchange buffer sits at its start
  qf_lds_float* xb = (qf_lds_float*)dyn;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = blockIdx.x * 4 + wave;
  const int g = lane >> 4, c = lane & 15;
  f32x4 D0, D, U = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < 4; ++r) D0[r] = in[w * 256 + (4 * g + r) * 16 + c];
  D = D0;
  float minpiv = 3e38f;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  f32x4 a0 = D0, a1 = D0, a2 = D0, a3 = D0;
  const h4 ha = {(_Float16)1.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
  if (skew && (blockIdx.x & 1))   // every other workgroup starts half a block step late
    for (int i = 0; i < work / 2 + 16; i += 4) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, ha, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, ha, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, ha, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, ha, a3, 0, 0, 0);
    }
  for (int rep = 0; rep < reps; ++rep) {
    if (QUAD) U = factor_diag_quad(D, lane, wave, (rep + skew * (int)blockIdx.x) & 3, minpiv, xb);
    else U = factor_diag(D, lane, minpiv);
    for (int i = 0; i < work; i += 4) {   // stand-in for the TRSM/SYRK of the block step: four independent chains
      a0 = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, ha, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, ha, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, ha, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, ha, a3, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[r] = D0[r] + 1e-30f * (U[r] + a0[r] + a1[r] + a2[r] + a3[r]);
  }
  minpiv = fminf(fminf(minpiv, row_ror<8>(minpiv)), fminf(row_ror<4>(minpiv), row_ror<12>(minpiv))),
  minpiv = fminf(fminf(minpiv, row_ror<1>(minpiv)), fminf(row_ror<2>(minpiv), row_ror<3>(minpiv)));
  for (int r = 0; r < 4; ++r) out[w * 256 + (4 * g + r) * 16 + c] = U[r];
  if (lane == 0) out_piv[w] = minpiv;
}

int main(int argc, char** argv) {
  const int wg_per_cu = argc > 1 ? atoi(argv[1]) : 2, work = argc > 2 ? atoi(argv[2]) : 64, reps = argc > 3 ? atoi(argv[3]) : 200, skew = argc > 4 ? atoi(argv[4]) : 1;
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  const int n_wg = n_cu * wg_per_cu, n = n_wg * 4;
  const size_t lds = (size_t)(160 * 1024 / wg_per_cu) / 1024 * 1024 - 1024;
  std::vector<float> h((size_t)n * 256);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd;
  for (int w = 0; w < n; ++w) {
    float A[24][16];
    for (auto& row : A) for (auto& x : row) x = nd(rng);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = (i == j) ? 0.5 : 0.0;
        for (int t = 0; t < 24; ++t) s += (double)A[t][i] * A[t][j];
        h[(size_t)w * 256 + i * 16 + j] = (float)s;
      }
  }
  float *d, *du, *dp;
  (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&du, h.size() * 4); (void)hipMalloc(&dp, n * 4);
  (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int rc = 0;
  for (int quad = 0; quad < 2; ++quad) {
    float ms_best = 1e9f;
    for (int trial = 0; trial < 4; ++trial) {
      (void)hipEventRecord(e0, 0);
      if (quad) hipLaunchKernelGGL(k<true>, dim3(n_wg), dim3(256), lds, 0, d, du, dp, reps, work, skew);
      else hipLaunchKernelGGL(k<false>, dim3(n_wg), dim3(256), lds, 0, d, du, dp, reps, work, skew);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (trial) ms_best = fminf(ms_best, ms);
    }
    std::vector<float> u(h.size()), piv(n);
    (void)hipMemcpy(u.data(), du, h.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(piv.data(), dp, n * 4, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    for (int w = 0; w < n; w += 7) {  // D^-1 = Uinv Uinv^T
      double M[16][16];
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int t = 0; t < 16; ++t) s += (double)u[(size_t)w*256+i*16+t] * u[(size_t)w*256+j*16+t]; M[i][j] = s; }
      double err = 0;
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int t = 0; t < 16; ++t) s += M[i][t] * h[(size_t)w*256+t*16+j]; err = fmax(err, fabs(s - (i == j))); }
      if (!(err < 1e-3) || !(piv[w] > 0.f && piv[w] < 1e6f)) { if (bad < 5) printf("tile %d: |Uinv Uinv^T D - I| = %g, min pivot %g\n", w, err, piv[w]); ++bad; }
      worst = fmax(worst, err);
    }
    // per SIMD: waves_per_simd tiles per rep -> time per tile-factorization slot
    const double per_rep_ns = ms_best * 1e6 / reps;
    printf("%s: %d workgroups per CU (%d waves per SIMD), %d f16 MFMAs between: %.3f ms, %.0f ns per block step per wave, "
           "%.1f ns per tile and SIMD; worst residual %.2g, bad %d\n", quad ? "quad     " : "one-tile ", wg_per_cu, wg_per_cu, work,
           ms_best, per_rep_ns, per_rep_ns / wg_per_cu, worst, bad);
    rc |= bad != 0;
  }
  return rc;
}
