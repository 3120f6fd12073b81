// Unit test of factor_diag on the GPU: random SPD 16x16 tiles; checks Uinv Uinv^T D = I and that
// Uinv is upper triangular.
#include "../../myrrix-recommender_amd/csrc/als_kernels.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
using namespace mals;
__global__ void k(const float* in, float* out_uinv, float* out_e, int n) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n) return;
  const int g = lane >> 4, c = lane & 15;
  f32x4 F;
  for (int r = 0; r < 4; ++r) F[r] = in[w * 256 + (4 * g + r) * 16 + c];
  float minpiv = 3e38f;
  const f32x4 U = factor_diag(F, lane, minpiv);
  for (int r = 0; r < 4; ++r) {
    out_uinv[w * 256 + (4 * g + r) * 16 + c] = U[r];
    out_e[w * 256 + (4 * g + r) * 16 + c] = U[r];
  }
}
int main() {
  const int n = 4096;
  std::vector<float> h(n * 256);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd;
  for (int w = 0; w < n; ++w) {
    float A[24][16];
    for (auto& row : A) for (auto& x : row) x = nd(rng);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = (i == j) ? 0.5 : 0.0;
        for (int t = 0; t < 24; ++t) s += (double)A[t][i] * A[t][j];
        h[w * 256 + i * 16 + j] = (float)s;
      }
  }
  float *d, *du, *de;
  (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&du, h.size() * 4); (void)hipMalloc(&de, h.size() * 4);
  (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 4), dim3(256), 0, 0, d, du, de, n);
  std::vector<float> u(h.size()), e(h.size());
  (void)hipMemcpy(u.data(), du, h.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(e.data(), de, h.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, worst_t = 0; int bad = 0;
  for (int w = 0; w < n; ++w) {
    // check Uinv * Uinv^T * D = I  (D^-1 = Uinv Uinv^T) and E == Uinv^T
    double M[16][16] = {};
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int t = 0; t < 16; ++t) s += (double)u[w*256+i*16+t] * u[w*256+j*16+t]; M[i][j] = s; }
    double err = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int t = 0; t < 16; ++t) s += M[i][t] * h[w*256+t*16+j]; err = fmax(err, fabs(s - (i == j))); }
    double et = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < i; ++j) et = fmax(et, fabs((double)u[w*256+i*16+j]));  // Uinv is upper triangular
    if (err > 1e-3 || et > 1e-6) { if (bad < 5) printf("tile %d: |Dinv D - I| = %g, |E - Uinv^T| = %g\n", w, err, et); ++bad; }
    worst = fmax(worst, err); worst_t = fmax(worst_t, et);
  }
  printf("worst |Uinv Uinv^T D - I| = %g, worst |E-Uinv^T| = %g, bad tiles %d / %d\n", worst, worst_t, bad, n);
  return bad != 0;
}
