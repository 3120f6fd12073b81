// Replays, in plain C++ against libmyrrix_als.so, the C-ABI calls that the Java side of the Solver SPI issues:
//   net.myrrix.common.math.JBlasLinearSystemSolver.getSolver / isNonSingular   (java/net/myrrix/common/math/, the class
//       MatrixUtils loads under -Dcommon.matrix.nativeMath=true, MatrixUtils.java:44-49)
//   net.myrrix.common.math.NativeSolver.solveDToF / solveFToD / close / recompute (jni/myrrix_solver_jni.c)
// on the inputs of Generation.recomputeSolver (Generation.java:142-158):
//   MTM = MatrixUtils.transposeTimesSelf(M) (MatrixUtils.java:219-239: float product, double sum);
//   infNorm = MTM.getNorm() (max absolute column sum) < 1.0  -> IllConditionedSolverException, getSolver never called;
//   MatrixUtils.getSolver(MTM) -> mals_solver_create: MALS_SINGULAR + apparent rank -> SingularMatrixSolverException(rank);
//   then Solver.solveDToF / solveFToD from the serving threads.
// Host only by default (mals_solver_* need no GPU); with the argument "device" also NativeSolver.recompute's
// sequence: mals_group_create(1 device) -> set_factor_rows / set_factors -> mals_group_local -> mals_recompute_solver.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/myrrix_als.h"

static int failures = 0;
#define CHECK(cond)                                             \
  do {                                                          \
    if (!(cond)) {                                              \
      std::printf("FAIL line %d: %s\n", __LINE__, #cond);      \
      ++failures;                                               \
    }                                                           \
  } while (0)

static unsigned long long lcg_state = 1234567890ull;
static double uniform01() {
  lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(lcg_state >> 11) / 9007199254740992.0;
}

// MatrixUtils.transposeTimesSelf (MatrixUtils.java:219-239)
static std::vector<double> transpose_times_self(const std::vector<float>& M, int n, int k) {
  std::vector<double> G((size_t)k * k, 0.0);
  for (int r = 0; r < n; ++r)
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) {
        const volatile float p = M[(size_t)r * k + a] * M[(size_t)r * k + b];  // a Java float * float is a float
        G[(size_t)a * k + b] += (double)p;
      }
  return G;
}

// RealMatrix.getNorm(): maximum absolute column sum
static double inf_norm(const std::vector<double>& A, int k) {
  double best = 0.0;
  for (int c = 0; c < k; ++c) {
    double s = 0.0;
    for (int r = 0; r < k; ++r) s += std::fabs(A[(size_t)r * k + c]);
    best = std::fmax(best, s);
  }
  return best;
}

// independent check: Gaussian elimination with partial pivoting, fp64
static std::vector<double> dense_solve(std::vector<double> A, std::vector<double> b, int k) {
  for (int c = 0; c < k; ++c) {
    int p = c;
    for (int r = c + 1; r < k; ++r)
      if (std::fabs(A[(size_t)r * k + c]) > std::fabs(A[(size_t)p * k + c])) p = r;
    for (int j = 0; j < k; ++j) std::swap(A[(size_t)c * k + j], A[(size_t)p * k + j]);
    std::swap(b[(size_t)c], b[(size_t)p]);
    for (int r = c + 1; r < k; ++r) {
      const double f = A[(size_t)r * k + c] / A[(size_t)c * k + c];
      for (int j = c; j < k; ++j) A[(size_t)r * k + j] -= f * A[(size_t)c * k + j];
      b[(size_t)r] -= f * b[(size_t)c];
    }
  }
  std::vector<double> x((size_t)k);
  for (int r = k - 1; r >= 0; --r) {
    double s = b[(size_t)r];
    for (int j = r + 1; j < k; ++j) s -= A[(size_t)r * k + j] * x[(size_t)j];
    x[(size_t)r] = s / A[(size_t)r * k + r];
  }
  return x;
}

enum Outcome { SOLVER, ILL_CONDITIONED, SINGULAR };

// Generation.recomputeSolver + the solves, as the Java classes issue them.  Returns what the JVM would see.
static Outcome recompute_solver_sequence(const std::vector<float>& M, int n, int k, double threshold, int* rank_seen) {
  const std::vector<double> MTM = transpose_times_self(M, n, k);                     // Generation.java:148
  const double norm = inf_norm(MTM, k);                                              // :149
  if (norm < 1.0) return ILL_CONDITIONED;                                            // :150-153: getSolver is never reached
  mals_solver s = nullptr;                                                           // :154 -> JBlasLinearSystemSolver.getSolver
  int32_t rank = -7;
  const int rc = mals_solver_create(MTM.data(), k, threshold, &s, &rank);           // NativeSolver.nativeCreate
  if (rc == MALS_SINGULAR) {
    CHECK(s == nullptr);
    *rank_seen = rank;
    return SINGULAR;                                                                 // SingularMatrixSolverException(rank, "Apparent rank: " + rank)
  }
  CHECK(rc == MALS_OK && s != nullptr);
  if (rc != MALS_OK) return SINGULAR;
  CHECK(mals_solver_dim(s) == k);                                                    // the shim sizes every array with it
  for (int trial = 0; trial < 3; ++trial) {
    std::vector<double> b((size_t)k), xd((size_t)k);
    std::vector<float> bf((size_t)k), xf((size_t)k);
    for (int i = 0; i < k; ++i) {
      b[(size_t)i] = 2.0 * uniform01() - 1.0;
      bf[(size_t)i] = (float)b[(size_t)i];
    }
    CHECK(mals_solver_solve_dtof(s, b.data(), xf.data()) == MALS_OK);               // Solver.solveDToF (Solver.java:35)
    CHECK(mals_solver_solve_ftod(s, bf.data(), xd.data()) == MALS_OK);              // Solver.solveFToD (Solver.java:41)
    const std::vector<double> want = dense_solve(MTM, b, k);
    std::vector<double> bfd(bf.begin(), bf.end());
    const std::vector<double> want_f = dense_solve(MTM, bfd, k);
    double scale = 0.0;
    for (int i = 0; i < k; ++i) scale = std::fmax(scale, std::fabs(want[(size_t)i]));
    for (int i = 0; i < k; ++i) {
      CHECK(xf[(size_t)i] == (float)want[(size_t)i] || std::fabs((double)xf[(size_t)i] - want[(size_t)i]) <= 2e-7 * scale);  // (float) of the fp64 solution
      CHECK(std::fabs(xd[(size_t)i] - want_f[(size_t)i]) <= 1e-9 * scale);
    }
  }
  CHECK(mals_solver_destroy(s) == MALS_OK);                                          // NativeSolver.close()
  return SOLVER;
}

int main(int argc, char** argv) {
  const bool device = argc > 1 && std::strcmp(argv[1], "device") == 0;
  const double threshold = 1.0e-5;   // LinearSystemSolver.SINGULARITY_THRESHOLD (LinearSystemSolver.java:33-34)

  // 1. a healthy model side: 500 x 30 factors (MatrixFactorizer.DEFAULT_FEATURES)
  {
    const int n = 500, k = 30;
    std::vector<float> M((size_t)n * k);
    for (float& v : M) v = (float)(uniform01() - 0.3);
    int rank = -1;
    CHECK(recompute_solver_sequence(M, n, k, threshold, &rank) == SOLVER);
  }
  // 2. inf norm below 1 (what too large a model.als.lambda produces): IllConditionedSolverException, no solver built
  {
    const int n = 40, k = 10;
    std::vector<float> M((size_t)n * k);
    for (float& v : M) v = (float)(0.01 * (uniform01() - 0.5));
    int rank = -1;
    CHECK(recompute_solver_sequence(M, n, k, threshold, &rank) == ILL_CONDITIONED);
  }
  // 3. fewer rows than features (ReducedFeaturesTest's situation): singular, apparent rank = rows
  {
    const int n = 5, k = 8;
    std::vector<float> M((size_t)n * k);
    for (float& v : M) v = (float)(2.0 * uniform01() - 1.0);
    int rank = -1;
    CHECK(recompute_solver_sequence(M, n, k, threshold, &rank) == SINGULAR);
    CHECK(rank == 5);
  }
  // 4. isNonSingular(M): the same create, the solver released at once; a null matrix is not "singular"
  {
    const double I2[4] = {1.0, 0.0, 0.0, 1.0}, Z2[4] = {1.0, 1.0, 1.0, 1.0};
    mals_solver s = nullptr;
    int32_t rank = -1;
    CHECK(mals_solver_create(I2, 2, threshold, &s, &rank) == MALS_OK && s);
    CHECK(mals_solver_destroy(s) == MALS_OK);
    s = nullptr;
    CHECK(mals_solver_create(Z2, 2, threshold, &s, &rank) == MALS_SINGULAR && !s && rank == 1);
    CHECK(mals_solver_create(nullptr, 2, threshold, &s, &rank) == MALS_INVALID_ARG);
    CHECK(mals_solver_dim(nullptr) == 0);                                            // the shim rejects a closed handle by it
  }
  // 5. NativeSolver.recompute: the factors are resident in the factorizer's group, M^T M runs on the device
  if (device) {
    const int n = 3000, k = 64;
    std::vector<float> M((size_t)n * k);
    for (float& v : M) v = (float)(uniform01() - 0.4);
    mals_config cfg;
    CHECK(mals_default_config(&cfg) == MALS_OK);
    cfg.features = k;
    const int32_t dev0 = 0;
    mals_group g = nullptr;
    if (mals_group_create(&cfg, &dev0, 1, MALS_GROUP_PEER_COPY, &g) != MALS_OK) {
      std::printf("mals_group_create failed: a HIP device is required for the device part\n");
      return 3;
    }
    CHECK(mals_group_set_factor_rows(g, MALS_SIDE_X, n) == MALS_OK);
    CHECK(mals_group_set_factors(g, MALS_SIDE_X, 0, n, M.data()) == MALS_OK);
    mals_handle h = nullptr;
    CHECK(mals_group_local(g, 0, &h, nullptr) == MALS_OK && h);
    mals_solver s = nullptr;
    double norm = 0.0;
    CHECK(mals_recompute_solver(h, MALS_SIDE_X, &s, &norm) == MALS_OK && s);
    const std::vector<double> MTM = transpose_times_self(M, n, k);
    CHECK(std::fabs(norm - inf_norm(MTM, k)) <= 1e-6 * norm);
    std::vector<double> b((size_t)k, 1.0);
    std::vector<float> xf((size_t)k);
    CHECK(mals_solver_solve_dtof(s, b.data(), xf.data()) == MALS_OK);
    const std::vector<double> want = dense_solve(MTM, b, k);
    double scale = 0.0, worst = 0.0;
    for (int i = 0; i < k; ++i) scale = std::fmax(scale, std::fabs(want[(size_t)i]));
    for (int i = 0; i < k; ++i) worst = std::fmax(worst, std::fabs((double)xf[(size_t)i] - want[(size_t)i]));
    CHECK(worst <= 1e-5 * scale);   // the device Gramian keeps exact products where MU:232 rounds each to fp32
    CHECK(mals_solver_destroy(s) == MALS_OK);
    // a side scaled down until getNorm() < 1: MALS_ILL_CONDITIONED with the norm, no solver
    for (float& v : M) v *= 1e-3f;
    CHECK(mals_group_set_factors(g, MALS_SIDE_X, 0, n, M.data()) == MALS_OK);
    s = nullptr;
    CHECK(mals_recompute_solver(h, MALS_SIDE_X, &s, &norm) == MALS_ILL_CONDITIONED && !s && norm < 1.0 && norm > 0.0);
    // fewer rows than features: MALS_SINGULAR, rank through mals_singular_info (row = -1: a model Gramian)
    CHECK(mals_group_set_factor_rows(g, MALS_SIDE_Y, 7) == MALS_OK);
    std::vector<float> Y((size_t)7 * k);
    for (float& v : Y) v = (float)(4.0 * uniform01() - 2.0);
    CHECK(mals_group_set_factors(g, MALS_SIDE_Y, 0, 7, Y.data()) == MALS_OK);
    CHECK(mals_recompute_solver(h, MALS_SIDE_Y, &s, &norm) == MALS_SINGULAR && !s);
    int32_t sd = -2, rank = -2;
    int64_t row = -2;
    CHECK(mals_singular_info(h, &sd, &row, &rank) == MALS_OK && sd == MALS_SIDE_Y && row == -1 && rank == 7);
    CHECK(mals_group_destroy(g) == MALS_OK);
  }
  if (failures) {
    std::printf("%d FAILED\n", failures);
    return 1;
  }
  std::printf("ALL PASSED (%s)\n", device ? "host + device" : "host");
  return 0;
}
