// host_eigen.h -- fp64 eigendecomposition of the k x k shared Gramian G = M^T M on the host.
//
// Used by the dual ("short row") solve path of dual_kernels.h: with G = Q diag(L) Q^T the per-row
// system of AlternatingLeastSquares.Worker.call (ALS:447-494),
//     W_u = G + sum_i (c_ui - 1) y_i y_i^T + lambda*alpha*n_u I,
// is, in the rotated coordinates y' = Q^T y, a diagonal matrix plus a rank-n_u update, and for rows
// with fewer entries than features the n_u x n_u system of the push-through identity is cheaper to
// factor than the k x k one.  k <= 128, once per half-iteration, overlapped with the kernels of the
// long rows -- O(10 k^3) flops on one core.
//
// Method: Householder reduction to tridiagonal form (transformations accumulated), then the implicit
// symmetric QR iteration with Wilkinson shifts and deflation (Golub & Van Loan, Matrix Computations,
// sections 8.3.1-8.3.5).  Written from the textbook algorithm; no third-party code.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace mals {

// A: row-major symmetric n x n (only read).  On success evals[n] (ascending is NOT guaranteed) and
// V (row-major n x n, column j = unit eigenvector of evals[j]) satisfy A = V diag(evals) V^T.
// Returns false when the input holds a non-finite value or the iteration does not converge.
inline bool symmetric_eigen(const double* A, int n, double* evals, double* V) {
  if (n <= 0) return false;
  std::vector<double> a((size_t)n * n);
  for (size_t i = 0; i < a.size(); ++i) {
    if (!std::isfinite(A[i])) return false;
    a[i] = A[i];
  }
  auto at = [&](int r, int c) -> double& { return a[(size_t)r * n + c]; };
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) V[(size_t)r * n + c] = r == c ? 1.0 : 0.0;
  std::vector<double> v(n), p(n), w(n);
  // --- Householder tridiagonalisation: for every column k annihilate rows k+2.. ------------------
  for (int k = 0; k + 2 < n; ++k) {
    double norm2 = 0.0;
    for (int r = k + 1; r < n; ++r) norm2 += at(r, k) * at(r, k);
    const double x0 = at(k + 1, k);
    double tail2 = norm2 - x0 * x0;
    if (!(tail2 > 0.0)) continue;  // already tridiagonal in this column
    const double alpha = x0 > 0.0 ? -std::sqrt(norm2) : std::sqrt(norm2);
    for (int r = k + 1; r < n; ++r) v[r] = at(r, k);
    v[k + 1] -= alpha;
    const double vnorm2 = tail2 + v[k + 1] * v[k + 1];
    const double beta = 2.0 / vnorm2;  // H = I - beta v v^T
    // p = beta * A22 v,  K = beta/2 * v^T p,  w = p - K v,  A22 -= v w^T + w v^T
    double vtp = 0.0;
    for (int r = k + 1; r < n; ++r) {
      double s = 0.0;
      const double* row = &a[(size_t)r * n];
      for (int c = k + 1; c < n; ++c) s += row[c] * v[c];
      p[r] = beta * s;
      vtp += v[r] * p[r];
    }
    const double K = 0.5 * beta * vtp;
    for (int r = k + 1; r < n; ++r) w[r] = p[r] - K * v[r];
    for (int r = k + 1; r < n; ++r) {
      double* row = &a[(size_t)r * n];
      const double vr = v[r], wr = w[r];
      for (int c = k + 1; c < n; ++c) row[c] -= vr * w[c] + wr * v[c];
    }
    at(k + 1, k) = at(k, k + 1) = alpha;
    for (int r = k + 2; r < n; ++r) at(r, k) = at(k, r) = 0.0;
    // V <- V H on columns k+1..
    for (int r = 0; r < n; ++r) {
      double* row = &V[(size_t)r * n];
      double s = 0.0;
      for (int c = k + 1; c < n; ++c) s += row[c] * v[c];
      s *= beta;
      for (int c = k + 1; c < n; ++c) row[c] -= s * v[c];
    }
  }
  std::vector<double> d(n), e(n, 0.0);
  for (int i = 0; i < n; ++i) d[i] = at(i, i);
  for (int i = 0; i + 1 < n; ++i) e[i] = at(i, i + 1);
  // --- implicit symmetric QR with Wilkinson shift on the unreduced trailing block [lo, hi] ------
  // the rotations are accumulated in Vt (transposed: rows = eigenvectors) so that each one touches two
  // contiguous rows
  std::vector<double> Vt((size_t)n * n);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) Vt[(size_t)c * n + r] = V[(size_t)r * n + c];
  const double eps = 2.220446049250313e-16;
  int hi = n - 1;
  long budget = 60L * n;
  while (hi > 0) {
    if (std::fabs(e[hi - 1]) <= eps * (std::fabs(d[hi - 1]) + std::fabs(d[hi]))) {
      e[hi - 1] = 0.0;
      --hi;
      continue;
    }
    int lo = hi - 1;
    while (lo > 0 && std::fabs(e[lo - 1]) > eps * (std::fabs(d[lo - 1]) + std::fabs(d[lo]))) --lo;
    if (lo > 0) e[lo - 1] = 0.0;
    if (--budget < 0) return false;
    const double dd = 0.5 * (d[hi - 1] - d[hi]);
    const double eh = e[hi - 1];
    const double denom = dd + (dd >= 0.0 ? std::hypot(dd, eh) : -std::hypot(dd, eh));
    const double mu = denom != 0.0 ? d[hi] - eh * eh / denom : d[hi];
    double x = d[lo] - mu, z = e[lo];
    for (int k = lo; k < hi; ++k) {
      const double r = std::hypot(x, z);
      double c = 1.0, s = 0.0;
      if (r != 0.0) {
        c = x / r;
        s = -z / r;
      }
      if (k > lo) e[k - 1] = r;
      const double d1 = d[k], d2 = d[k + 1], ek = e[k];
      d[k] = c * c * d1 - 2.0 * c * s * ek + s * s * d2;
      d[k + 1] = s * s * d1 + 2.0 * c * s * ek + c * c * d2;
      e[k] = c * s * (d1 - d2) + (c * c - s * s) * ek;
      x = e[k];
      if (k + 1 < hi) {
        z = -s * e[k + 1];
        e[k + 1] = c * e[k + 1];
      }
      double* vk = &Vt[(size_t)k * n];
      double* vk1 = vk + n;
      for (int i = 0; i < n; ++i) {
        const double t = vk[i];
        vk[i] = c * t - s * vk1[i];
        vk1[i] = s * t + c * vk1[i];
      }
    }
  }
  for (int i = 0; i < n; ++i) evals[i] = d[i];
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) V[(size_t)r * n + c] = Vt[(size_t)c * n + r];
  return true;
}

}  // namespace mals
