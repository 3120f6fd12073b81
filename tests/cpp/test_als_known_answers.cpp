// C++ re-statement of the reference's factorizer unit tests on the C++ mirror interface
// (include/myrrix/factorizer.hpp -> include/myrrix_als.h -> libmyrrix_als.so):
//   AlternatingLeastSquaresTest.testALS / testALSPredictingR  (ALST:38-78, data ALST:92-115)
//   NegativeInputTest.testALS                                  (NIT:36-80)
//   MatrixUtilsTest.testAddTo / testRemove                     (MatrixUtilsTest.java:33-60)
// plus the solver state of Generation (include/myrrix/generation.hpp; Generation.java:132-158).
// The expected matrices are the known-answer data of those tests.  Tolerance 1e-6, the reference's own (the GPU path is
// fp32; the reference's fp64 path is pinned at 1e-6 by the oracle, tests/test_oracle_golden.py).
// Exit code 0 = all passed.  Needs a GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../include/myrrix/factorizer.hpp"
#include "../../include/myrrix/generation.hpp"

using namespace myrrix;

static int failures = 0;
#define CHECK(cond)                                                  \
  do {                                                               \
    if (!(cond)) {                                                   \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
      ++failures;                                                    \
    }                                                                \
  } while (0)

static void checkProduct(const char* name, const std::vector<std::vector<double>>& product,
                         const std::vector<std::vector<float>>& expected) {
  double worst = 0.0;
  CHECK(product.size() == expected.size());
  for (size_t r = 0; r < expected.size(); ++r) {
    CHECK(product[r].size() == expected[r].size());
    for (size_t c = 0; c < expected[r].size(); ++c) worst = std::fmax(worst, std::fabs((float)product[r][c] - expected[r][c]));
  }
  std::printf("%-22s max |X*Y^T - expected| = %.3g\n", name, worst);
  CHECK(worst <= 1e-6);  // the reference's own tolerance (MyrrixTest.java:34)
}

static std::vector<std::vector<double>> buildTestXYTProduct(bool reconstructR) {  // ALST:86-125
  System::setProperty("model.reconstructRMatrix", reconstructR ? "true" : "false");
  FastByIDMap<FastByIDFloatMap> byRow, byCol;
  // Octave: R = [ 0 2 3 1 0 ; 0 0 4 5 0 ; 1 0 0 0 2 ; 3 0 1 0 5 ; 0 2 2 2 0 ]
  const float R[5][5] = {{0, 2, 3, 1, 0}, {0, 0, 4, 5, 0}, {1, 0, 0, 0, 2}, {3, 0, 1, 0, 5}, {0, 2, 2, 2, 0}};
  for (int r = 0; r < 5; ++r)
    for (int c = 0; c < 5; ++c)
      if (R[r][c] != 0) MatrixUtils::addTo(r, c, R[r][c], byRow, byCol);
  FastByIDMap<FloatVector> previousY = {{0, {0.1f, 0.2f}}, {1, {0.2f, 0.5f}}, {2, {0.3f, 0.1f}}, {3, {0.2f, 0.2f}}, {4, {0.5f, 0.4f}}};
  AlternatingLeastSquares als(byRow, byCol, 2, 0.0001, 40);
  als.setPreviousY(&previousY);
  als.call();
  System::clearProperty("model.reconstructRMatrix");
  return MatrixUtils::multiplyXYT(als.getX(), als.getY());
}

int main() {
  {  // MatrixUtilsTest.testAddTo / testRemove
    FastByIDMap<FastByIDFloatMap> byRow, byCol;
    MatrixUtils::addTo(0, 0, -1.0f, byRow, byCol);
    MatrixUtils::addTo(4, 1, 2.0f, byRow, byCol);
    CHECK(byRow[0][0] == -1.0f && byCol[0][0] == -1.0f);
    CHECK(byRow.count(1) == 0);
    CHECK(byRow[4][1] == 2.0f && byCol[1][4] == 2.0f);
    MatrixUtils::remove(0, 0, byRow, byCol);
    CHECK(byRow.count(0) == 0);
    CHECK(byRow[4][1] == 2.0f && byCol[1][4] == 2.0f);
  }
  try {
    checkProduct("testALS", buildTestXYTProduct(false),
                 {{-0.030258f, 0.852781f, 1.004839f, 1.024087f, -0.036206f},
                  {0.077046f, 0.751232f, 0.949796f, 0.910322f, 0.073047f},
                  {0.916777f, -0.196005f, 0.335926f, -0.163591f, 0.929028f},
                  {0.987400f, 0.130943f, 0.772403f, 0.235522f, 0.998354f},
                  {-0.028683f, 0.850540f, 1.003130f, 1.021514f, -0.034598f}});
    checkProduct("testALSPredictingR", buildTestXYTProduct(true),
                 {{0.0678369f, 0.6574759f, 2.1020291f, 2.0976211f, 0.1115919f},
                  {-0.0176293f, 1.3062225f, 4.1365933f, 4.1739127f, -0.0380586f},
                  {1.0854513f, -0.0344434f, 0.1725342f, -0.1564803f, 1.8502977f},
                  {2.8377915f, 0.0528524f, 0.9041158f, 0.0474437f, 4.8365208f},
                  {-0.0057799f, 0.6608552f, 2.0936351f, 2.1115670f, -0.0139042f}});
    {  // NegativeInputTest
      FastByIDMap<FastByIDFloatMap> byRow, byCol;
      // Octave: R = [ 1 1 1 0 ; 0 -1 1 1 ; -1 0 0 1 ]
      const float R[3][4] = {{1, 1, 1, 0}, {0, -1, 1, 1}, {-1, 0, 0, 1}};
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
          if (R[r][c] != 0) MatrixUtils::addTo(r, c, R[r][c], byRow, byCol);
      FastByIDMap<FloatVector> previousY = {{0, {0.1f, 0.2f}}, {1, {0.2f, 0.5f}}, {2, {0.3f, 0.1f}}, {3, {0.2f, 0.2f}}};
      AlternatingLeastSquares als(byRow, byCol, 2, 0.0001, 40);
      als.setPreviousY(&previousY);
      als.call();
      checkProduct("NegativeInputTest", MatrixUtils::multiplyXYT(als.getX(), als.getY()),
                   {{0.899032f, 0.900162f, 0.990150f, -0.026642f},
                    {0.181214f, 0.089988f, 0.787198f, 1.012226f},
                    {-0.104165f, -0.178240f, 0.360391f, 0.825856f}});
    }
    {  // no previousY: random unit initial Y, finite factors of the right shape, never converges after 1 iteration
      FastByIDMap<FastByIDFloatMap> byRow, byCol;
      for (int u = 0; u < 40; ++u)
        for (int i = 0; i < 25; ++i)
          if ((u * 7 + i * 3) % 5 == 0) MatrixUtils::addTo(1000 + u, 5000 + i, 1.0f + (u + i) % 4, byRow, byCol);
      AlternatingLeastSquares als(byRow, byCol, 5, 0.001, 10);
      als.call();
      CHECK(als.getX().size() == byRow.size() && als.getY().size() == byCol.size());
      CHECK(als.iterations() >= 2);  // ALS:252-253
      for (auto& e : als.getX())
        for (float f : e.second) CHECK(std::isfinite(f));
    }
    {  // Generation.recomputeSolver + Solver (Generation.java:142-158): (Y^T Y) x = b round trip
      const int k = 6;
      FastByIDMap<FloatVector> X, Y;
      std::mt19937_64 rng(7);
      std::normal_distribution<float> nd;
      for (int i = 0; i < 300; ++i) {
        FloatVector v(k), w(k);
        for (int f = 0; f < k; ++f) { v[f] = nd(rng); w[f] = nd(rng); }
        X[10 + i] = v;
        Y[500 + 3 * i] = w;
      }
      Generation gen(X, Y);
      CHECK(gen.getNumUsers() == 300 && gen.getNumItems() == 300);
      CHECK(gen.getXTXSolver() != nullptr && gen.getYTYSolver() != nullptr);
      std::vector<double> G((size_t)k * k, 0.0);  // fp64 Y^T Y on the host for the residual
      for (auto& e : Y)
        for (int r = 0; r < k; ++r)
          for (int c = 0; c < k; ++c) G[(size_t)r * k + c] += (double)e.second[r] * (double)e.second[c];
      FloatVector b = {1.f, -2.f, 0.5f, 3.f, 0.f, -1.f};
      std::vector<double> x = gen.getYTYSolver()->solveFToD(b);
      double worst = 0.0;
      for (int r = 0; r < k; ++r) {
        double s = 0.0;
        for (int c = 0; c < k; ++c) s += G[(size_t)r * k + c] * x[c];
        worst = std::fmax(worst, std::fabs(s - b[r]));
      }
      std::printf("%-22s max |Y^T Y x - b| = %.3g\n", "Generation solver", worst);
      CHECK(worst < 1e-4);
      FloatVector xf = gen.getYTYSolver()->solveDToF(std::vector<double>(b.begin(), b.end()));
      for (int r = 0; r < k; ++r) CHECK(std::fabs(xf[r] - (float)x[r]) <= 1e-6f * std::fmax(1.f, std::fabs(xf[r])));
      // ill-conditioned (Generation.java:150-153) and rank-deficient (CMLSS:46-54) factors
      FastByIDMap<FloatVector> tiny = {{1, {1e-3f, 0.f}}, {2, {0.f, 1e-3f}}}, none;
      bool threw = false;
      try { Generation bad(none, tiny); } catch (const IllConditionedSolverException&) { threw = true; }
      CHECK(threw);
      FastByIDMap<FloatVector> flat = {{1, {1.f, 2.f, 3.f}}, {2, {2.f, 4.f, 6.f}}, {3, {-1.f, -2.f, -3.f}}};
      int rank = -1;
      try { Generation bad(none, flat); } catch (const SingularMatrixSolverException& e) { rank = e.getApparentRank(); }
      CHECK(rank == 1);
      CHECK(Generation(none, none).getYTYSolver() == nullptr);
      CHECK(isNonSingular({2.0, 0.0, 0.0, 3.0}, 2) && !isNonSingular({1.0, 1.0, 1.0, 1.0}, 2));
    }
    {  // constructor preconditions (ALS:139-141)
      FastByIDMap<FastByIDFloatMap> a, b;
      bool threw = false;
      try { AlternatingLeastSquares bad(a, b, 0, 0.001, 1); } catch (const std::invalid_argument&) { threw = true; }
      CHECK(threw);
      threw = false;
      try { AlternatingLeastSquares bad(a, b, 2, 1.0, 1); } catch (const std::invalid_argument&) { threw = true; }
      CHECK(threw);
    }
  } catch (const std::exception& e) {
    std::printf("FAIL: exception %s\n", e.what());
    ++failures;
  }
  std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
  return failures ? 1 : 0;
}
