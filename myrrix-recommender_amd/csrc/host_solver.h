// host_solver.h -- fp64 k x k solver for the MODEL Gramians X^T X / Y^T Y on the host.
//
// Generation.recomputeSolver (online/src/net/myrrix/online/generation/Generation.java:142-158)
// builds MatrixUtils.getSolver(M^T M) once per generation; the serving layer then calls
// Solver.solveFToD / solveDToF (common/src/net/myrrix/common/math/Solver.java:35-41) once per
// fold-in request.  M^T M is the device kernel K1; the k x k factorization is O(k^3) <= 2 MFLOP
// and done once, so it stays on the host in fp64 -- the reference's own precision here.
//
// Semantics follow the reference's CommonsMathLinearSystemSolver (CMLSS:37-55): a column-pivoted
// Householder QR  A P = Q R  (pivot = remaining column of largest 2-norm), "non-singular" iff every
// |R_ii| > threshold, and for a singular matrix an apparent rank from the Frobenius norms of R's
// trailing blocks with drop tolerance 0.01.  Reflectors are stored LAPACK-style: H_j = I - tau_j
// v_j v_j^T with v_j[j] = 1 implicit and v_j[j+1:] kept below R's diagonal (column-major).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

namespace mals {

class PivotedQR {
 public:
  // A: row-major n x n.
  PivotedQR(const double* A, int n, double threshold) : n_(n), thr_(threshold), a_((size_t)n * n), tau_(n), piv_(n) {
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) at(r, c) = A[(size_t)r * n + c];
    for (int j = 0; j < n; ++j) piv_[j] = j;
    for (int j = 0; j < n; ++j) {
      bring_largest_column_to(j);
      reflect(j);
    }
  }

  int dim() const { return n_; }

  bool non_singular() const {
    for (int j = 0; j < n_; ++j)
      if (std::fabs(at(j, j)) <= thr_) return false;
    return true;
  }

  // Smallest r such that the trailing block R[r:, r:] is negligible:
  // stop when ||R[r:,r:]||_F is 0 or (||R[r:,r:]||_F / ||R[r-1:,r-1:]||_F) * ||R||_F < drop.
  int rank(double drop) const {
    std::vector<double> tail(n_ + 1, 0.0);  // tail[r] = squared Frobenius norm of rows >= r of R
    for (int r = n_ - 1; r >= 0; --r) {
      double s = 0.0;
      for (int c = r; c < n_; ++c) s += at(r, c) * at(r, c);
      tail[r] = tail[r + 1] + s;
    }
    const double whole = std::sqrt(tail[0]);
    double prev = whole;
    int r = 1;
    for (; r < n_; ++r) {
      const double cur = std::sqrt(tail[r]);
      if (cur == 0.0 || (cur / prev) * whole < drop) break;
      prev = cur;
    }
    return r;
  }

  // x = A^-1 b (both length n, may alias).  Precondition: non_singular().
  void solve(const double* b, double* x) const {
    std::vector<double> y(b, b + n_);
    for (int j = 0; j < n_; ++j) {  // y <- H_j y
      if (tau_[j] == 0.0) continue;
      double s = y[j];
      for (int r = j + 1; r < n_; ++r) s += at(r, j) * y[r];
      s *= tau_[j];
      y[j] -= s;
      for (int r = j + 1; r < n_; ++r) y[r] -= s * at(r, j);
    }
    for (int r = n_ - 1; r >= 0; --r) {  // R z = y
      double s = y[r];
      for (int c = r + 1; c < n_; ++c) s -= at(r, c) * y[c];
      y[r] = s / at(r, r);
    }
    for (int j = 0; j < n_; ++j) x[piv_[j]] = y[j];
  }

 private:
  double& at(int r, int c) { return a_[(size_t)c * n_ + r]; }
  double at(int r, int c) const { return a_[(size_t)c * n_ + r]; }

  void bring_largest_column_to(int j) {
    int best = j;
    double best_norm = 0.0;
    auto consider = [&](int c, double s) {
      if (s > best_norm) {
        best_norm = s;
        best = c;
      }
    };
    int c = j;
    for (; c + 4 <= n_; c += 4) {
      const double* c0 = &a_[(size_t)c * n_];
      const double *c1 = c0 + n_, *c2 = c1 + n_, *c3 = c2 + n_;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int r = j; r < n_; ++r) {
        s0 += c0[r] * c0[r];
        s1 += c1[r] * c1[r];
        s2 += c2[r] * c2[r];
        s3 += c3[r] * c3[r];
      }
      consider(c, s0);
      consider(c + 1, s1);
      consider(c + 2, s2);
      consider(c + 3, s3);
    }
    for (; c < n_; ++c) {
      const double* cc = &a_[(size_t)c * n_];
      double s = 0.0;
      for (int r = j; r < n_; ++r) s += cc[r] * cc[r];
      consider(c, s);
    }
    if (best != j) {
      std::swap_ranges(a_.begin() + (size_t)j * n_, a_.begin() + (size_t)(j + 1) * n_, a_.begin() + (size_t)best * n_);
      std::swap(piv_[j], piv_[best]);
    }
  }

  // Annihilate column j below the diagonal; R_jj = -sign(x_0) ||x||.
  void reflect(int j) {
    double norm2 = 0.0;
    for (int r = j; r < n_; ++r) norm2 += at(r, j) * at(r, j);
    const double x0 = at(j, j);
    const double beta = x0 > 0.0 ? -std::sqrt(norm2) : std::sqrt(norm2);
    if (beta == 0.0) {
      tau_[j] = 0.0;
      return;
    }
    const double v0 = x0 - beta;
    tau_[j] = -v0 / beta;  // = 2 / (v^T v) with v scaled to v[0] = 1
    for (int r = j + 1; r < n_; ++r) at(r, j) /= v0;
    at(j, j) = beta;
    // apply H_j to the trailing columns, four at a time (independent dot-product chains; the
    // summation order inside each dot product does not change)
    const double* v = &a_[(size_t)j * n_];
    const double tau = tau_[j];
    int c = j + 1;
    for (; c + 4 <= n_; c += 4) {
      double* c0 = &a_[(size_t)c * n_];
      double* c1 = c0 + n_;
      double* c2 = c1 + n_;
      double* c3 = c2 + n_;
      double s0 = c0[j], s1 = c1[j], s2 = c2[j], s3 = c3[j];
      for (int r = j + 1; r < n_; ++r) {
        const double vr = v[r];
        s0 += vr * c0[r];
        s1 += vr * c1[r];
        s2 += vr * c2[r];
        s3 += vr * c3[r];
      }
      s0 *= tau; s1 *= tau; s2 *= tau; s3 *= tau;
      c0[j] -= s0; c1[j] -= s1; c2[j] -= s2; c3[j] -= s3;
      for (int r = j + 1; r < n_; ++r) {
        const double vr = v[r];
        c0[r] -= s0 * vr;
        c1[r] -= s1 * vr;
        c2[r] -= s2 * vr;
        c3[r] -= s3 * vr;
      }
    }
    for (; c < n_; ++c) {
      double* cc = &a_[(size_t)c * n_];
      double s = cc[j];
      for (int r = j + 1; r < n_; ++r) s += v[r] * cc[r];
      s *= tau;
      cc[j] -= s;
      for (int r = j + 1; r < n_; ++r) cc[r] -= s * v[r];
    }
  }

  int n_;
  double thr_;
  std::vector<double> a_;  // column-major: R on and above the diagonal, reflector tails below
  std::vector<double> tau_;
  std::vector<int> piv_;
};

// AbstractRealMatrix.getNorm() as used at Generation.java:150: maximum absolute column sum.
inline double max_abs_column_sum(const double* A, int n) {
  double best = 0.0;
  for (int c = 0; c < n; ++c) {
    double s = 0.0;
    for (int r = 0; r < n; ++r) s += std::fabs(A[(size_t)r * n + c]);
    best = std::max(best, s);
  }
  return best;
}

}  // namespace mals
