// Replays, in plain C++ against libmyrrix_als.so, the exact sequence of C-ABI calls that
// jni/myrrix_als_jni.c issues for one HipAlternatingLeastSquares.call()
// (java/net/myrrix/online/factorizer/als/HipAlternatingLeastSquares.java):
//   mals_default_config, mals_group_create(devices, n)
//   mals_group_set_refine_limit (optional), mals_group_set_factor_rows(X), (Y)
//   mals_group_begin_matrix / [mals_group_pending_entries +] mals_group_append_rows (pieces of whole rows) /
//   mals_group_end_matrix, for R and R^T; mals_group_features (array size checks of the shim)
//   mals_group_set_factors(Y, pieces)                      setPreviousY / initial Y
//   mals_group_factorize                                    call()
//   mals_group_get_factors(X), (Y)                          getX() / getY()
//   mals_group_destroy
// and, since round 4: a create that must fail (device ordinal outside the box) whose reason nativeCreateError hands to the
// JVM (mals_group_create_error), and the per-iteration callback behind the adapter's log lines
// (mals_group_set_iteration_callback -> IterationListener.iteration, ALS:241-246, 351-358).
// on the reference's known-answer cases (AlternatingLeastSquaresTest.java:42-77, NegativeInputTest.java:71-79;
// tests/golden/reference_known_answers.json, written to a text file by tests/test_jni_sequence.py):
// X*Y^T must equal the expected matrix to the reference's own 1e-6.
// usage: test_jni_call_sequence <case file> <n_members> <backend 0|1> <rows per piece>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/myrrix_als.h"

#define REQUIRE_OK(call)                                                                              \
  do {                                                                                                \
    const int _rc = (call);                                                                           \
    if (_rc != MALS_OK) {                                                                             \
      std::printf("FAIL %s -> %d (%s)\n", #call, _rc, g ? mals_group_last_error(g) : "no group");     \
      return 1;                                                                                       \
    }                                                                                                 \
  } while (0)

static void stream(std::vector<int64_t>& row_ptr, std::vector<int32_t>& col, std::vector<float>& val, const std::vector<float>& R,
                   int n_rows, int n_cols, bool transposed) {
  row_ptr.assign(1, 0);
  col.clear();
  val.clear();
  for (int r = 0; r < n_rows; ++r) {
    for (int c = 0; c < n_cols; ++c) {
      const float v = transposed ? R[(size_t)c * n_rows + r] : R[(size_t)r * n_cols + c];
      if (v != 0.f) {
        col.push_back(c);
        val.push_back(v);
      }
    }
    row_ptr.push_back((int64_t)col.size());
  }
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  std::FILE* f = std::fopen(argv[1], "r");
  if (!f) return 2;
  const int n_members = std::atoi(argv[2]), backend = std::atoi(argv[3]), piece = std::atoi(argv[4]);
  int features, max_iterations, flags, n_users, n_items;
  double threshold, tol;
  if (std::fscanf(f, "%d %lf %d %d %d %d %lf", &features, &threshold, &max_iterations, &flags, &n_users, &n_items, &tol) != 7) return 2;
  std::vector<float> R((size_t)n_users * n_items), Y0((size_t)n_items * features), expected((size_t)n_users * n_items);
  for (float& v : R) if (std::fscanf(f, "%f", &v) != 1) return 2;
  for (float& v : Y0) if (std::fscanf(f, "%f", &v) != 1) return 2;
  for (float& v : expected) if (std::fscanf(f, "%f", &v) != 1) return 2;
  std::fclose(f);

  mals_group g = nullptr;
  mals_config cfg;
  REQUIRE_OK(mals_default_config(&cfg));
  cfg.features = features;
  cfg.flags = flags;
  std::vector<int32_t> devices((size_t)n_members, 0);
  {   // nativeCreate with -Dmodel.als.gpus=4095: returns 0, and the JVM learns WHY (nativeCreateError)
    std::vector<int32_t> nowhere(1, 4095);
    mals_group bad = nullptr;
    char why[512];
    if (mals_group_create(&cfg, nowhere.data(), 1, backend, &bad) == MALS_OK || bad) {
      std::printf("a group on device 4095 must not exist\n");
      return 4;
    }
    const int need = mals_group_create_error(why, sizeof(why));
    if (need <= 0 || !std::strstr(why, "4095")) {
      std::printf("mals_group_create_error does not name the reason: \"%s\"\n", why);
      return 4;
    }
    std::printf("create on device 4095: %s\n", why);
  }
  if (mals_group_create(&cfg, devices.data(), n_members, backend, &g) != MALS_OK) {
    char why[512];
    (void)mals_group_create_error(why, sizeof(why));
    std::printf("mals_group_create failed: %s\n", why);
    return 3;
  }
  {
    char why[8];
    if (mals_group_create_error(why, sizeof(why)) != 0 || why[0]) {
      std::printf("mals_group_create_error must be empty after a success\n");
      return 4;
    }
  }
  REQUIRE_OK(mals_group_set_refine_limit(g, 128.0));   // -Dmodel.als.gpu.refineLimit, when set
  REQUIRE_OK(mals_group_set_factor_rows(g, MALS_SIDE_X, n_users));
  REQUIRE_OK(mals_group_set_factor_rows(g, MALS_SIDE_Y, n_items));
  std::vector<int64_t> row_ptr;
  std::vector<int32_t> col;
  std::vector<float> val;
  for (int side = 0; side < 2; ++side) {
    const int n_rows = side == MALS_SIDE_X ? n_users : n_items;
    stream(row_ptr, col, val, R, n_rows, side == MALS_SIDE_X ? n_items : n_users, side == MALS_SIDE_Y);
    REQUIRE_OK(mals_group_begin_matrix(g, side, n_rows, row_ptr.data()));
    for (int r0 = 0; r0 < n_rows; r0 += piece) {
      const int r1 = r0 + piece < n_rows ? r0 + piece : n_rows;
      int64_t expected_entries = -1;   // nativeAppendRows checks its arrays against this before the library reads them
      REQUIRE_OK(mals_group_pending_entries(g, side, r1 - r0, &expected_entries));
      if (expected_entries != row_ptr[(size_t)r1] - row_ptr[(size_t)r0]) {
        std::printf("FAIL mals_group_pending_entries: %lld\n", (long long)expected_entries);
        return 1;
      }
      REQUIRE_OK(mals_group_append_rows(g, side, r1 - r0, col.data() + row_ptr[(size_t)r0], val.data() + row_ptr[(size_t)r0]));
    }
    REQUIRE_OK(mals_group_end_matrix(g, side));
  }
  if (mals_group_features(g) != features) {   // nativeSetFactors / nativeGetFactors size-check their float[] with it
    std::printf("FAIL mals_group_features\n");
    return 1;
  }
  for (int r0 = 0; r0 < n_items; r0 += piece) {
    const int n = r0 + piece < n_items ? piece : n_items - r0;
    REQUIRE_OK(mals_group_set_factors(g, MALS_SIDE_Y, r0, n, Y0.data() + (size_t)r0 * features));
  }
  std::vector<int64_t> test_users((size_t)n_users), test_items((size_t)n_items);  // tiny inputs: the whole population
  for (int i = 0; i < n_users; ++i) test_users[(size_t)i] = i;
  for (int i = 0; i < n_items; ++i) test_items[(size_t)i] = i;
  int32_t iterations = 0;
  double convergence = 0.0;
  struct Seen {
    int calls = 0, last_iteration = 0;
    double last_value = 0.0;
    long long rows = 0;
    bool ok = true;
  } seen;
  // nativeFactorize: the listener behind the adapter's "Finished iteration {}" / "Avg absolute difference ..." lines
  REQUIRE_OK(mals_group_set_iteration_callback(g, [](void* user, const mals_iteration_info* info) {
    Seen* sn = static_cast<Seen*>(user);
    sn->ok = sn->ok && info->struct_size == (int32_t)sizeof(mals_iteration_info) && info->iteration == sn->last_iteration + 1 &&
             info->seconds > 0.0 && info->devices >= 1 && info->algorithmic_bytes > 0.0;
    ++sn->calls;
    sn->last_iteration = info->iteration;
    sn->last_value = info->avg_abs_difference;
    sn->rows = (long long)(info->x_rows + info->y_rows);
  }, &seen));
  REQUIRE_OK(mals_group_factorize(g, threshold, max_iterations, /*random_y=*/0, /*iterate=*/1, test_users.data(), n_users,
                                  test_items.data(), n_items, &iterations, &convergence));
  REQUIRE_OK(mals_group_set_iteration_callback(g, nullptr, nullptr));
  if (!seen.ok || seen.calls != iterations || seen.last_iteration != iterations || seen.last_value != convergence ||
      seen.rows != (long long)n_users + n_items) {
    std::printf("iteration callback: %d calls for %d iterations, last value %.17g vs %.17g, rows %lld\n", seen.calls, iterations,
                seen.last_value, convergence, seen.rows);
    return 5;
  }
  std::printf("iteration callback: %d calls, last avg abs difference %.6g\n", seen.calls, seen.last_value);
  std::vector<float> X((size_t)n_users * features), Y((size_t)n_items * features);
  for (int r0 = 0; r0 < n_users; r0 += piece)
    REQUIRE_OK(mals_group_get_factors(g, MALS_SIDE_X, r0, r0 + piece < n_users ? piece : n_users - r0, X.data() + (size_t)r0 * features));
  for (int r0 = 0; r0 < n_items; r0 += piece)
    REQUIRE_OK(mals_group_get_factors(g, MALS_SIDE_Y, r0, r0 + piece < n_items ? piece : n_items - r0, Y.data() + (size_t)r0 * features));
  double worst = 0.0;
  for (int u = 0; u < n_users; ++u)
    for (int i = 0; i < n_items; ++i) {
      double d = 0.0;  // SimpleVectorMath.dot: float product, double sum
      for (int k = 0; k < features; ++k) {
        const volatile float p = X[(size_t)u * features + k] * Y[(size_t)i * features + k];
        d += (double)p;
      }
      worst = std::fmax(worst, std::fabs((double)(float)d - (double)expected[(size_t)u * n_items + i]));
    }
  REQUIRE_OK(mals_group_destroy(g));
  std::printf("members %d backend %d piece %d: %d iterations, convergence %.3g, max |X*Y^T - expected| = %.3g (tol %.1g)\n", n_members,
              backend, piece, iterations, convergence, worst, tol);
  if (!(worst <= tol)) {
    std::printf("FAIL\n");
    return 1;
  }
  std::printf("ALL PASSED\n");
  return 0;
}
