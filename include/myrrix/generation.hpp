// generation.hpp -- C++ host-side mirror of the model-load consumer of the Gramian kernel
//  over the C-ABI:
//   net.myrrix.common.math.Solver                       common/src/.../math/Solver.java:27-43
//   MatrixUtils.getSolver / isNonSingular               common/src/.../math/MatrixUtils.java:130-139
//   net.myrrix.common.math.IllConditionedSolverException
//   Generation.recomputeState / recomputeSolver / get{XTX,YTY}Solver
//                                                       online/src/.../generation/Generation.java:132-158,181-205
// Only the solver state of Generation is mirrored; candidate filters, known-item maps, clusters and
// locks are serving-plane state (out of scope).  Header-only, C++17.
#pragma once
#include <ios>
#include <memory>

#include "factorizer.hpp"

namespace myrrix {

struct IllConditionedSolverException : SolverException {
  using SolverException::SolverException;
};

class Solver {
 public:
  explicit Solver(mals_solver s) : s_(s) {}
  Solver(const Solver&) = delete;
  Solver& operator=(const Solver&) = delete;
  ~Solver() { mals_solver_destroy(s_); }

  FloatVector solveDToF(const std::vector<double>& b) const {
    requireDim(b.size());
    FloatVector x(b.size());
    mals_solver_solve_dtof(s_, b.data(), x.data());
    return x;
  }
  std::vector<double> solveFToD(const FloatVector& b) const {
    requireDim(b.size());
    std::vector<double> x(b.size());
    mals_solver_solve_ftod(s_, b.data(), x.data());
    return x;
  }

 private:
  void requireDim(size_t n) const {
    if ((int)n != mals_solver_dim(s_)) throw std::invalid_argument("vector length != solver dimension");
  }
  mals_solver s_;
};

inline double singularityThreshold() {  // LinearSystemSolver.java:33-34
  return std::stod(System::getProperty("common.matrix.singularityThreshold", "1.0E-5"));
}

// MatrixUtils.getSolver (MU:137-139, CMLSS:37-55); M row-major n x n.
inline std::unique_ptr<Solver> getSolver(const std::vector<double>& M, int n) {
  mals_solver s = nullptr;
  int32_t rank = 0;
  const int rc = mals_solver_create(M.data(), n, singularityThreshold(), &s, &rank);
  if (rc == MALS_SINGULAR) throw SingularMatrixSolverException(rank, "Apparent rank: " + std::to_string(rank));
  if (rc != MALS_OK) throw std::invalid_argument("mals_solver_create failed");
  return std::make_unique<Solver>(s);
}

// MatrixUtils.isNonSingular (MU:130-132, CMLSS:58-63)
inline bool isNonSingular(const std::vector<double>& M, int n) {
  try {
    getSolver(M, n);
    return true;
  } catch (const SingularMatrixSolverException&) {
    return false;
  }
}

class Generation {
 public:
  Generation(FastByIDMap<FloatVector> X, FastByIDMap<FloatVector> Y, int device = 0)
      : X_(std::move(X)), Y_(std::move(Y)), device_(device) {
    recomputeState();
  }

  void recomputeState() {  // Generation.java:132-140
    if (System::getProperty("model.solver.xtx.compute", "true") == "true") XTXsolver_ = recomputeSolver(X_);
    if (System::getProperty("model.solver.yty.compute", "true") == "true") YTYsolver_ = recomputeSolver(Y_);
  }

  const Solver* getXTXSolver() const { return XTXsolver_.get(); }
  const Solver* getYTYSolver() const { return YTYsolver_.get(); }
  size_t getNumUsers() const { return X_.size(); }
  size_t getNumItems() const { return Y_.size(); }
  const FastByIDMap<FloatVector>& getX() const { return X_; }
  const FastByIDMap<FloatVector>& getY() const { return Y_; }

 private:
  std::unique_ptr<Solver> recomputeSolver(const FastByIDMap<FloatVector>& M) const {  // Generation.java:142-158
    if (M.empty()) return nullptr;
    const int k = (int)M.begin()->second.size();
    std::vector<float> rows;
    rows.reserve(M.size() * (size_t)k);
    for (const auto& kv : M) rows.insert(rows.end(), kv.second.begin(), kv.second.end());
    mals_config cfg;
    mals_default_config(&cfg);
    cfg.features = k;
    cfg.device = device_;
    cfg.singularity_threshold = singularityThreshold();
    mals_handle h = nullptr;
    if (mals_create(&cfg, &h) != MALS_OK) throw std::runtime_error("mals_create failed: a HIP device is required");
    struct Guard {
      mals_handle h;
      ~Guard() { mals_destroy(h); }
    } guard{h};
    auto ok = [&](int rc) {
      if (rc != MALS_OK) throw std::runtime_error(mals_last_error(h));
    };
    ok(mals_set_factor_rows(h, MALS_SIDE_X, (int64_t)M.size()));
    ok(mals_set_factors(h, MALS_SIDE_X, 0, (int64_t)M.size(), rows.data()));
    mals_solver s = nullptr;
    double norm = 0.0;
    const int rc = mals_recompute_solver(h, MALS_SIDE_X, &s, &norm);
    if (rc == MALS_ILL_CONDITIONED) throw IllConditionedSolverException("infNorm: " + std::to_string(norm));
    if (rc == MALS_SINGULAR) {
      int32_t rank = 0;
      mals_singular_info(h, nullptr, nullptr, &rank);
      throw SingularMatrixSolverException(rank, mals_last_error(h));
    }
    ok(rc);
    return std::make_unique<Solver>(s);
  }

  FastByIDMap<FloatVector> X_, Y_;
  int device_;
  std::unique_ptr<Solver> XTXsolver_, YTYsolver_;
};

// InputFilesReader.readInputFiles (online-local/.../generation/InputFilesReader.java:64-211) on the device: everything
// the reference leaves behind after reading `inputDir` -- ids, R by user and by item as CSR over dense indices, the two
// tag id sets, knownItemIDs (absent under model.noKnownItems) -- through mals_ingest_read_dir / _finish.  Throws
// std::ios_base::failure where the reference throws IOException ("Too many bad lines; aborting", unreadable file).
struct InputMatrices {
  std::vector<int64_t> userIDs, itemIDs;                  // dense index -> id, ascending
  std::vector<int64_t> rowPtr[2];                         // [0]: R by user, [1]: R^T by item
  std::vector<int32_t> colIdx[2];
  std::vector<float> values[2];
  std::vector<int64_t> itemTagIDs, userTagIDs;            // IFR:159-165
  bool hasKnownItems = false;
  std::vector<int64_t> knownPtr;                          // knownItemIDs over the dense users (IFR:173-191)
  std::vector<int32_t> knownItems;
  int64_t lines = 0, badLines = 0;
};

inline InputMatrices readInputFiles(const std::string& inputDir, int device = 0, float zeroThreshold = 1.0e-4f, bool knownItems = true) {
  mals_ingest g = nullptr;
  if (mals_ingest_create(device, zeroThreshold, &g) != MALS_OK) throw std::runtime_error("mals_ingest_create failed: a HIP device is required");
  struct Guard {
    mals_ingest g;
    ~Guard() { mals_ingest_destroy(g); }
  } guard{g};
  auto ok = [&](int rc) {
    if (rc == MALS_IO_ERROR) throw std::ios_base::failure(mals_ingest_last_error(g));
    if (rc != MALS_OK) throw std::runtime_error(mals_ingest_last_error(g));
  };
  if (knownItems) ok(mals_ingest_set_option(g, MALS_INGEST_OPT_KNOWN_ITEMS, 1));
  int32_t files = 0;
  ok(mals_ingest_read_dir(g, inputDir.c_str(), &files));
  ok(mals_ingest_finish(g));
  InputMatrices m;
  int64_t records = 0, users = 0, items = 0, nnz = 0;
  ok(mals_ingest_counts(g, &records, &users, &items, &nnz));
  m.userIDs.resize((size_t)users);
  m.itemIDs.resize((size_t)items);
  ok(mals_ingest_get_ids(g, MALS_SIDE_X, m.userIDs.data()));
  ok(mals_ingest_get_ids(g, MALS_SIDE_Y, m.itemIDs.data()));
  for (int side = 0; side < 2; ++side) {
    m.rowPtr[side].resize((size_t)(side == 0 ? users : items) + 1);
    m.colIdx[side].resize((size_t)nnz);
    m.values[side].resize((size_t)nnz);
    ok(mals_ingest_get_csr(g, side, m.rowPtr[side].data(), m.colIdx[side].data(), m.values[side].data()));
  }
  mals_ingest_text_info_t info;
  info.struct_size = (int32_t)sizeof(info);
  ok(mals_ingest_text_info(g, &info));
  m.lines = info.lines;
  m.badLines = info.bad_lines;
  m.itemTagIDs.resize((size_t)info.n_item_tag_ids);
  m.userTagIDs.resize((size_t)info.n_user_tag_ids);
  ok(mals_ingest_get_tag_ids(g, MALS_ITEM_TAG_IDS, m.itemTagIDs.data()));
  ok(mals_ingest_get_tag_ids(g, MALS_USER_TAG_IDS, m.userTagIDs.data()));
  if (knownItems) {
    m.hasKnownItems = true;
    m.knownPtr.resize((size_t)users + 1);
    m.knownItems.resize((size_t)std::max<int64_t>(info.n_known_items, 0));
    ok(mals_ingest_get_known_items(g, m.knownPtr.data(), m.knownItems.data()));
  }
  return m;
}

}  // namespace myrrix
