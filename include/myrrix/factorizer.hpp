// factorizer.hpp -- C++ host-side mirror of the reference's factorizer interface over the C-ABI
// (include/myrrix_als.h).  The reference is compiled (Java) code and no JDK exists in the build
// image, so this header stands where the Java adapter of INTEGRATION.md would: same type names,
// method names, argument meaning and error behaviour as
//   net.myrrix.online.factorizer.MatrixFactorizer            online/src/.../MatrixFactorizer.java:31-77
//   net.myrrix.online.factorizer.als.AlternatingLeastSquares online/src/.../als/AlternatingLeastSquares.java:66-262
//   net.myrrix.common.math.MatrixUtils (addTo / remove / multiplyXYT) common/src/.../math/MatrixUtils.java:64-125,155-165
//   net.myrrix.common.math.SingularMatrixSolverException       common/src/.../math/SingularMatrixSolverException.java:23-53
// It only densifies ids, streams CSR through the C-ABI and copies factors back; every flop of the
// hot path runs in libmyrrix_als.so.  Header-only, C++17.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../myrrix_als.h"
#include "random.hpp"

namespace myrrix {

// FastByIDMap<V> / FastByIDFloatMap (common/src/.../collection/FastByIDMap.java:44,
// FastByIDFloatMap.java:43): long-keyed maps.  std::map gives a deterministic iteration order,
// which is all the factorizer needs of them (the reference iterates in hash-slot order).
template <class V>
using FastByIDMap = std::map<int64_t, V>;
using FastByIDFloatMap = std::map<int64_t, float>;
using FloatVector = std::vector<float>;

// java.lang.System properties: every model knob of the reference is a system property
class System {
 public:
  static void setProperty(const std::string& k, const std::string& v) { props()[k] = v; }
  static void clearProperty(const std::string& k) { props().erase(k); }
  static std::string getProperty(const std::string& k, const std::string& dflt) {
    auto it = props().find(k);
    return it == props().end() ? dflt : it->second;
  }

 private:
  static std::map<std::string, std::string>& props() {
    static std::map<std::string, std::string> p;
    return p;
  }
};

struct SolverException : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct SingularMatrixSolverException : SolverException {
  SingularMatrixSolverException(int apparentRank, const std::string& msg) : SolverException(msg), apparentRank_(apparentRank) {}
  int getApparentRank() const { return apparentRank_; }

 private:
  int apparentRank_;
};
// java.util.concurrent.ExecutionException: worker failures surface wrapped (ALS:349)
struct ExecutionException : std::runtime_error {
  explicit ExecutionException(std::shared_ptr<std::exception> cause)
      : std::runtime_error(cause ? cause->what() : "ExecutionException"), cause_(std::move(cause)) {}
  const std::exception* getCause() const { return cause_.get(); }

 private:
  std::shared_ptr<std::exception> cause_;
};
struct InterruptedException : std::runtime_error {
  InterruptedException() : std::runtime_error("interrupted") {}
};

struct MatrixUtils {
  // MU:64-92: increment an entry in two parallel sparse matrices (duplicates sum)
  static void addTo(int64_t row, int64_t column, float value, FastByIDMap<FastByIDFloatMap>& RbyRow,
                    FastByIDMap<FastByIDFloatMap>& RbyColumn) {
    RbyRow[row][column] += value;
    RbyColumn[column][row] += value;
  }
  // MU:94-125: remove an entry; an emptied row is deleted
  static void remove(int64_t row, int64_t column, FastByIDMap<FastByIDFloatMap>& RbyRow,
                     FastByIDMap<FastByIDFloatMap>& RbyColumn) {
    auto rm = [](FastByIDMap<FastByIDFloatMap>& M, int64_t a, int64_t b) {
      auto it = M.find(a);
      if (it == M.end()) return;
      it->second.erase(b);
      if (it->second.empty()) M.erase(it);
    };
    rm(RbyRow, row, column);
    rm(RbyColumn, column, row);
  }
  // SimpleVectorMath.dot (SVM:34-41): float product, double accumulation
  static double dot(const FloatVector& x, const FloatVector& y) {
    double d = 0.0;
    for (size_t i = 0; i < x.size(); ++i) {
      volatile float p = x[i] * y[i];
      d += (double)p;
    }
    return d;
  }
  // MU:155-165: dense product over ids 0..n-1
  static std::vector<std::vector<double>> multiplyXYT(const FastByIDMap<FloatVector>& X, const FastByIDMap<FloatVector>& Y) {
    std::vector<std::vector<double>> out(X.size(), std::vector<double>(Y.size(), 0.0));
    for (size_t r = 0; r < X.size(); ++r)
      for (size_t c = 0; c < Y.size(); ++c) out[r][c] = dot(X.at((int64_t)r), Y.at((int64_t)c));
    return out;
  }
};

// MatrixFactorizer.java:31-77
class MatrixFactorizer {
 public:
  static constexpr int DEFAULT_FEATURES = 30;  // MF:34
  virtual ~MatrixFactorizer() = default;
  virtual void call() = 0;  // throws ExecutionException, InterruptedException
  virtual void setPreviousX(const FastByIDMap<FloatVector>* previousX) = 0;
  virtual void setPreviousY(const FastByIDMap<FloatVector>* previousY) = 0;
  virtual const FastByIDMap<FloatVector>& getX() const = 0;
  virtual const FastByIDMap<FloatVector>& getY() const = 0;
};

// AlternatingLeastSquares.java:66 -- same constructor and methods; the work happens in libmyrrix_als.so
class AlternatingLeastSquares final : public MatrixFactorizer {
 public:
  static constexpr double DEFAULT_ALPHA = 1.0;                    // ALS:71
  static constexpr double DEFAULT_LAMBDA = 0.1;                   // ALS:73
  static constexpr double DEFAULT_CONVERGENCE_THRESHOLD = 0.001;  // ALS:74
  static constexpr int DEFAULT_MAX_ITERATIONS = 30;               // ALS:75
  static constexpr int NUM_USER_ITEMS_TO_TEST_CONVERGENCE = 100;  // ALS:80
  static constexpr size_t MAX_FAR_FROM_VECTORS = 100000;          // ALS:83

  AlternatingLeastSquares(const FastByIDMap<FastByIDFloatMap>& RbyRow, const FastByIDMap<FastByIDFloatMap>& RbyColumn,
                          int features, double estimateErrorConvergenceThreshold, int maxIterations, int device = 0)
      : RbyRow_(RbyRow), RbyColumn_(RbyColumn), features_(features), threshold_(estimateErrorConvergenceThreshold),
        maxIterations_(maxIterations), device_(device) {
    if (!(features > 0)) throw std::invalid_argument("features must be positive");  // ALS:139
    if (!(threshold_ > 0.0 && threshold_ < 1.0)) throw std::invalid_argument("threshold must be in (0,1)");  // ALS:140
  }

  void setPreviousX(const FastByIDMap<FloatVector>*) override {}  // ALS:162-165: does nothing
  void setPreviousY(const FastByIDMap<FloatVector>* previousY) override { previousY_ = previousY; }
  const FastByIDMap<FloatVector>& getX() const override { return X_; }
  const FastByIDMap<FloatVector>& getY() const override { return Y_; }
  int iterations() const { return iterations_; }
  double convergenceValue() const { return convergence_; }

  void call() override {  // ALS:176-262
    const int k = features_;
    const bool randomY = previousY_ == nullptr || previousY_->empty();  // ALS:181
    MersenneTwister rng(std::stoll(System::getProperty("model.test.seed", "1234567890")));  // RandomManager.java:52,63-73
    FastByIDMap<FloatVector> Y0 = constructInitialY(rng);                                      // ALS:182
    std::vector<int64_t> userIDs, itemIDs, yIDs;
    for (auto& e : RbyRow_) userIDs.push_back(e.first);
    for (auto& e : RbyColumn_) itemIDs.push_back(e.first);
    yIDs = itemIDs;
    for (auto& e : Y0)
      if (!RbyColumn_.count(e.first)) yIDs.push_back(e.first);  // stale rows still count in Y^T Y
    std::map<int64_t, int32_t> userIndex, itemIndex;
    for (size_t i = 0; i < userIDs.size(); ++i) userIndex[userIDs[i]] = (int32_t)i;
    for (size_t i = 0; i < yIDs.size(); ++i) itemIndex[yIDs[i]] = (int32_t)i;
    X_.clear();
    Y_.clear();
    if (userIDs.empty() || yIDs.empty()) {
      for (auto id : yIDs) Y_[id] = Y0[id];
      return;
    }
    Csr r = toCsr(RbyRow_, userIDs, itemIndex), c = toCsr(RbyColumn_, itemIDs, userIndex);
    std::vector<float> y0((size_t)yIDs.size() * k);
    for (size_t i = 0; i < yIDs.size(); ++i) std::copy(Y0[yIDs[i]].begin(), Y0[yIDs[i]].end(), y0.begin() + i * k);
    // ALS:206 asks RandomManager for ANOTHER generator (constructInitialY had its own, ALS:266): under the test seed every
    // getRandom() is a fresh MersenneTwister(TEST_SEED) (RandomManager.java:61-64) -- the sample's skips start at the
    // beginning of the stream whatever constructInitialY consumed
    MersenneTwister rngSample(std::stoll(System::getProperty("model.test.seed", "1234567890")));
    std::vector<int64_t> tu = chooseAboutN(NUM_USER_ITEMS_TO_TEST_CONVERGENCE, userIDs.size(), rngSample);  // ALS:206-209
    std::vector<int64_t> ti = chooseAboutN(NUM_USER_ITEMS_TO_TEST_CONVERGENCE, itemIDs.size(), rngSample);  // ALS:210-213

    mals_config cfg;
    mals_default_config(&cfg);
    cfg.features = k;
    cfg.alpha = std::stod(System::getProperty("model.als.alpha", "1.0"));    // ALS:506-509
    cfg.lambda = std::stod(System::getProperty("model.als.lambda", "0.1"));  // ALS:511-514
    cfg.singularity_threshold = std::stod(System::getProperty("common.matrix.singularityThreshold", "1.0e-5"));
    cfg.flags = (System::getProperty("model.reconstructRMatrix", "false") == "true" ? MALS_FLAG_RECONSTRUCT_R : 0) |  // ALS:85-87
                (System::getProperty("model.lossIgnoresUnspecified", "false") == "true" ? MALS_FLAG_LOSS_IGNORES_UNSPECIFIED : 0);
    cfg.device = device_;
    const bool iterate = System::getProperty("model.als.iterate", "true") == "true";  // ALS:196
    mals_handle h = nullptr;
    if (mals_create(&cfg, &h) != MALS_OK)
      throw ExecutionException(std::make_shared<std::runtime_error>("mals_create failed: a HIP device is required (no CPU fallback)"));
    struct Guard {
      mals_handle h;
      ~Guard() { mals_destroy(h); }
    } guard{h};
    check(h, mals_set_factor_rows(h, MALS_SIDE_X, (int64_t)userIDs.size()));
    check(h, mals_set_factor_rows(h, MALS_SIDE_Y, (int64_t)yIDs.size()));
    check(h, mals_set_matrix(h, MALS_SIDE_X, 0, (int64_t)userIDs.size(), (int64_t)r.col.size(), r.rowPtr.data(), r.col.data(), r.val.data(), MALS_MEM_HOST));
    check(h, mals_set_matrix(h, MALS_SIDE_Y, 0, (int64_t)itemIDs.size(), (int64_t)c.col.size(), c.rowPtr.data(), c.col.data(), c.val.data(), MALS_MEM_HOST));
    check(h, mals_set_factors(h, MALS_SIDE_Y, 0, (int64_t)yIDs.size(), y0.data()));
    int32_t iters = 0;
    double conv = std::numeric_limits<double>::quiet_NaN();
    check(h, mals_factorize(h, threshold_, maxIterations_, randomY ? 1 : 0, iterate ? 1 : 0, tu.data(), (int32_t)tu.size(),
                            ti.data(), (int32_t)ti.size(), &iters, &conv));
    iterations_ = iters;
    convergence_ = conv;
    std::vector<float> xs((size_t)userIDs.size() * k), ys((size_t)yIDs.size() * k);
    check(h, mals_get_factors(h, MALS_SIDE_X, 0, (int64_t)userIDs.size(), xs.data()));
    check(h, mals_get_factors(h, MALS_SIDE_Y, 0, (int64_t)yIDs.size(), ys.data()));
    for (size_t i = 0; i < userIDs.size(); ++i) X_[userIDs[i]] = FloatVector(xs.begin() + i * k, xs.begin() + (i + 1) * k);
    for (size_t i = 0; i < yIDs.size(); ++i) Y_[yIDs[i]] = FloatVector(ys.begin() + i * k, ys.begin() + (i + 1) * k);
  }

 private:
  struct Csr {
    std::vector<int64_t> rowPtr;
    std::vector<int32_t> col;
    std::vector<float> val;
  };

  static void check(mals_handle h, int rc) {
    if (rc == MALS_OK) return;
    const std::string msg = mals_last_error(h);
    if (rc == MALS_SINGULAR) {  // ALS:349 wraps the worker's SingularMatrixSolverException (CMLSS:46-54)
      int32_t side = 0, rank = 0;
      int64_t row = 0;
      mals_singular_info(h, &side, &row, &rank);
      throw ExecutionException(std::make_shared<SingularMatrixSolverException>(rank, msg));
    }
    if (rc == MALS_CANCELLED) throw InterruptedException();
    throw ExecutionException(std::make_shared<std::runtime_error>(msg));
  }

  static Csr toCsr(const FastByIDMap<FastByIDFloatMap>& R, const std::vector<int64_t>& rowIDs, const std::map<int64_t, int32_t>& colIndex) {
    Csr m;
    m.rowPtr.push_back(0);
    for (int64_t id : rowIDs) {
      for (auto& e : R.at(id)) {
        auto it = colIndex.find(e.first);
        // the reference logs "No vector for {}. This should not happen." (ALS:460-463)
        if (it == colIndex.end()) throw std::invalid_argument("matrix references an id with no row on the other side");
        m.col.push_back(it->second);
        m.val.push_back(e.second);
      }
      m.rowPtr.push_back((int64_t)m.col.size());
    }
    return m;
  }

  static void normalize(FloatVector& v) {  // SimpleVectorMath.normalize (SVM:80-85)
    double t = 0.0;
    for (float f : v) t += (double)(f * f);
    const float n = (float)std::sqrt(t);
    for (float& f : v) f /= n;
  }

  // RandomUtils.randomUnitVectorFarFrom (RandomUtils.java:110-140) on the reference's own random stream
  // (random.hpp: commons-math3's MersenneTwister as RandomManager seeds it), drawn in the reference's order: the
  // Gaussians of doRandomUnitVector (RU:88-100), one nextInt(size) per sampled earlier vector when there are more
  // than 100 (RU:124), one nextDouble() for the acceptance (RU:136).
  static FloatVector randomUnitVectorFarFrom(int k, const std::vector<const FloatVector*>& farFrom, MersenneTwister& rng) {
    const size_t size = farFrom.size(), numSamples = std::min<size_t>(100, size);
    for (;;) {
      FloatVector v(k);
      double total = 0.0;
      for (int i = 0; i < k; ++i) {
        const double d = rng.nextGaussian();
        v[i] = (float)d;
        total += d * d;
      }
      const float nrm = (float)std::sqrt(total);
      for (float& f : v) f /= nrm;
      double smallest = std::numeric_limits<double>::infinity();
      for (size_t s = 0; s < numSamples; ++s) {
        const FloatVector& other = *farFrom[size == numSamples ? s : (size_t)rng.nextInt((int32_t)size)];
        const double d2 = 2.0 - 2.0 * MatrixUtils::dot(v, other);
        if (std::isfinite(d2) && d2 < smallest) smallest = d2;
      }
      if (std::isfinite(smallest) && !(k == 1 && smallest == 0.0)) {
        if (rng.nextDouble() < smallest / 4.0) return v;
      } else {
        return v;
      }
    }
  }

  FastByIDMap<FloatVector> constructInitialY(MersenneTwister& rng) {  // ALS:264-335
    const int k = features_;
    FastByIDMap<FloatVector> Y;
    if (previousY_ && !previousY_->empty()) {
      const size_t oldK = previousY_->begin()->second.size();
      for (auto& e : *previousY_) {
        FloatVector v(k, 0.f);
        for (size_t i = 0; i < std::min<size_t>(oldK, (size_t)k); ++i) v[i] = e.second[i];
        if (oldK > (size_t)k) {  // ALS:277-287
          normalize(v);
        } else if (oldK < (size_t)k) {  // ALS:289-302
          for (size_t i = oldK; i < (size_t)k; ++i) v[i] = (float)rng.nextGaussian();   // ALS:297-299
          normalize(v);
        }
        Y[e.first] = v;
      }
    }
    std::vector<const FloatVector*> recent;
    for (auto& e : Y) {
      if (recent.size() >= MAX_FAR_FROM_VECTORS) break;
      recent.push_back(&e.second);
    }
    for (auto& e : RbyColumn_) {  // ALS:318-328
      if (!Y.count(e.first)) {
        auto ins = Y.emplace(e.first, randomUnitVectorFarFrom(k, recent, rng));
        if (recent.size() < MAX_FAR_FROM_VECTORS) recent.push_back(&ins.first->second);
      }
    }
    return Y;
  }

  // RandomUtils.chooseAboutNFromStream (RandomUtils.java:202-217): everything when n >= size, else what
  // SamplingLongPrimitiveIterator keeps: it skips PascalDistribution(random, 1, rate).sample() elements, which in
  // commons-math3 is inverseCumulativeProbability(random.nextDouble()) = floor(log(1-u) / log(1-rate)) -- one
  // nextDouble() per skip, the JVM's value except for a u within rounding of a step of the CDF.  Dense indices.
  static std::vector<int64_t> chooseAboutN(int n, size_t size, MersenneTwister& rng) {
    std::vector<int64_t> out;
    if ((size_t)n >= size) {
      for (size_t i = 0; i < size; ++i) out.push_back((int64_t)i);
      return out;
    }
    const double rate = (double)n / (double)size;
    int64_t pos = -1;
    for (;;) {
      const double u = rng.nextDouble();
      pos += 1 + (int64_t)std::floor(std::log1p(-u) / std::log1p(-rate));
      if ((size_t)pos >= size) break;
      out.push_back(pos);
    }
    return out;
  }

  const FastByIDMap<FastByIDFloatMap>& RbyRow_;
  const FastByIDMap<FastByIDFloatMap>& RbyColumn_;
  int features_;
  double threshold_;
  int maxIterations_;
  int device_;
  const FastByIDMap<FloatVector>* previousY_ = nullptr;
  FastByIDMap<FloatVector> X_, Y_;
  int iterations_ = 0;
  double convergence_ = std::numeric_limits<double>::quiet_NaN();
};

}  // namespace myrrix
