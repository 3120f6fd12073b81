/*
 * als_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64 arithmetic on fp32 storage) of the ALS hot path of
 * myrrix-recommender, used ONLY as the parity checker by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The shipped path (libmyrrix_als.so, HIP) never links, loads or
 * calls anything in this directory.
 *
 * Parity status: PINNED for the ALS arithmetic -- checked in tests/test_oracle_golden.py against
 * every known-answer vector the reference's own tests hold for this path:
 *   online/test/.../als/AlternatingLeastSquaresTest.java:42-56 (default), :63-77 (reconstructR),
 *   online/test/.../als/NegativeInputTest.java:71-79,
 *   common/test/.../math/MatrixUtilsTest.java:63-71 (Gramian), SimpleVectorMathTest.java:28-37.
 * UNPINNED: the apparent-rank value reported for a singular system (no reference test pins it;
 * restated from the published commons-math3 3.2 RRQRDecomposition.getRank algorithm).
 *
 * Third-party arithmetic not under /root/reference: org.apache.commons:commons-math3:3.2
 * (pom.xml:81) -- RRQRDecomposition / QRDecomposition.Solver.  Restated below from the published
 * algorithm (Householder QR on the transposed storage with column pivoting by largest remaining
 * column norm; solve = apply Q^T, back-substitute R, undo the permutation).
 *
 * Each function cites the reference file:line it follows.  Abbreviations:
 *   ALS = online/src/net/myrrix/online/factorizer/als/AlternatingLeastSquares.java
 *   MU  = common/src/net/myrrix/common/math/MatrixUtils.java
 *   SVM = common/src/net/myrrix/common/math/SimpleVectorMath.java
 *   CMLSS = common/src/net/myrrix/common/math/CommonsMathLinearSystemSolver.java
 *   CMS = common/src/net/myrrix/common/math/CommonsMathSolver.java
 *   LSS = common/src/net/myrrix/common/math/LinearSystemSolver.java
 *   DWM = common/src/net/myrrix/common/stats/DoubleWeightedMean.java
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_FLAG_RECONSTRUCT_R 1          /* ALS:85-87  model.reconstructRMatrix          */
#define ORACLE_FLAG_LOSS_IGNORES_UNSPECIFIED 2 /* ALS:89-91  model.lossIgnoresUnspecified    */

#define ORACLE_OK 0
#define ORACLE_SINGULAR 1

#define WORK_UNIT_SIZE 100 /* ALS:77 */

/* ------------------------------------------------------------------------------------------ */
/* MU:219-239 transposeTimesSelf: G[r][c] += (float)(v[r]*v[c]) widened to double.            */
/* Rows are visited in index order (the reference visits hash-slot order).         */
void oracle_gramian(const float* M, int64_t n, int k, double* G) {
  memset(G, 0, sizeof(double) * (size_t)k * (size_t)k);
  for (int64_t i = 0; i < n; i++) {
    const float* v = M + i * (int64_t)k;
    for (int r = 0; r < k; r++) {
      float rv = v[r];
      double* Gr = G + (size_t)r * k;
      for (int c = 0; c < k; c++) {
        float prod = rv * v[c]; /* Java float*float: rounded to fp32 before widening */
        Gr[c] += (double)prod;
      }
    }
  }
}

/* SVM:34-41 dot: sum of (float)(x[i]*y[i]) accumulated in double. */
double oracle_dot(const float* x, const float* y, int k) {
  double dot = 0.0;
  for (int i = 0; i < k; i++) {
    float p = x[i] * y[i];
    dot += (double)p;
  }
  return dot;
}

/* SVM:46-52 norm(float[]) */
double oracle_norm(const float* x, int k) {
  double total = 0.0;
  for (int i = 0; i < k; i++) {
    float p = x[i] * x[i];
    total += (double)p;
  }
  return sqrt(total);
}

/* ------------------------------------------------------------------------------------------ */
/* commons-math3 3.2 RRQRDecomposition(M, threshold) + getSolver().solve(b) as used at
 * CMLSS:37-55 and CMS:37-44.  Storage is the transpose qrt[col][row] like the library.
 * Returns ORACLE_OK and x (cast to fp32, CMS:40-42), or ORACLE_SINGULAR and *apparent_rank
 * (= getRank(0.01), CMLSS:47). work must hold k*k + 2*k doubles and perm k ints. */
static int rrqr_solve(const double* W, const double* b, int k, double threshold, float* x_out,
                      int* apparent_rank, double* work, int* perm) {
  double* qrt = work;              /* [col][row] */
  double* rdiag = work + (size_t)k * k;
  double* y = rdiag + k;
  for (int c = 0; c < k; c++)
    for (int r = 0; r < k; r++) qrt[(size_t)c * k + r] = W[(size_t)r * k + c];
  for (int i = 0; i < k; i++) perm[i] = i;

  for (int minor = 0; minor < k; minor++) {
    /* RRQR pivot: column (>= minor) with the largest squared norm over rows >= minor */
    double best = 0.0;
    int best_col = minor;
    for (int c = minor; c < k; c++) {
      double s = 0.0;
      const double* col = qrt + (size_t)c * k;
      for (int r = minor; r < k; r++) s += col[r] * col[r];
      if (s > best) {
        best = s;
        best_col = c;
      }
    }
    if (best_col != minor) {
      for (int r = 0; r < k; r++) {
        double t = qrt[(size_t)minor * k + r];
        qrt[(size_t)minor * k + r] = qrt[(size_t)best_col * k + r];
        qrt[(size_t)best_col * k + r] = t;
      }
      int t = perm[minor];
      perm[minor] = perm[best_col];
      perm[best_col] = t;
    }
    /* Householder reflection on column `minor` */
    double* qm = qrt + (size_t)minor * k;
    double xnorm2 = 0.0;
    for (int r = minor; r < k; r++) xnorm2 += qm[r] * qm[r];
    double a = (qm[minor] > 0.0) ? -sqrt(xnorm2) : sqrt(xnorm2);
    rdiag[minor] = a;
    if (a != 0.0) {
      qm[minor] -= a;
      for (int c = minor + 1; c < k; c++) {
        double* qc = qrt + (size_t)c * k;
        double alpha = 0.0;
        for (int r = minor; r < k; r++) alpha -= qc[r] * qm[r];
        alpha /= a * qm[minor];
        for (int r = minor; r < k; r++) qc[r] -= alpha * qm[r];
      }
    }
  }

  /* isNonSingular(): every |R_dd| > threshold (LSS:33-34 default 1e-5) */
  int singular = 0;
  for (int d = 0; d < k; d++)
    if (fabs(rdiag[d]) <= threshold) singular = 1;

  if (singular) {
    /* getRank(0.01): walk Frobenius norms of trailing sub-matrices of R.
     * R[i][j] = rdiag[i] if i==j, qrt[j][i] if j>i, 0 otherwise. */
    if (apparent_rank) {
      int rank = 1;
      double fro = 0.0;
      for (int i = 0; i < k; i++) {
        fro += rdiag[i] * rdiag[i];
        for (int j = i + 1; j < k; j++) fro += qrt[(size_t)j * k + i] * qrt[(size_t)j * k + i];
      }
      double last = sqrt(fro), rnorm = last;
      while (rank < k) {
        double s = 0.0;
        for (int i = rank; i < k; i++) {
          s += rdiag[i] * rdiag[i];
          for (int j = i + 1; j < k; j++) s += qrt[(size_t)j * k + i] * qrt[(size_t)j * k + i];
        }
        double th = sqrt(s);
        if (th == 0.0 || (th / last) * rnorm < 0.01) break;
        last = th;
        rank++;
      }
      *apparent_rank = rank;
    }
    return ORACLE_SINGULAR;
  }

  /* y = Q^T b */
  for (int i = 0; i < k; i++) y[i] = b[i];
  for (int minor = 0; minor < k; minor++) {
    const double* qm = qrt + (size_t)minor * k;
    double dp = 0.0;
    for (int r = minor; r < k; r++) dp += y[r] * qm[r];
    dp /= rdiag[minor] * qm[minor];
    for (int r = minor; r < k; r++) y[r] += dp * qm[r];
  }
  /* back-substitute R x = y */
  for (int r = k - 1; r >= 0; r--) {
    y[r] /= rdiag[r];
    double yr = y[r];
    const double* qr = qrt + (size_t)r * k;
    for (int i = 0; i < r; i++) y[i] -= yr * qr[i];
  }
  /* undo the column permutation, cast to float (CMS:40-42) */
  for (int i = 0; i < k; i++) x_out[perm[i]] = (float)y[i];
  return ORACLE_OK;
}

int oracle_rrqr_solve(const double* W, const double* b, int k, double threshold, float* x_out,
                      int* apparent_rank) {
  double* work = (double*)malloc(sizeof(double) * ((size_t)k * k + 2 * (size_t)k));
  int* perm = (int*)malloc(sizeof(int) * (size_t)k);
  int rc = rrqr_solve(W, b, k, threshold, x_out, apparent_rank, work, perm);
  free(work);
  free(perm);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* ALS:432-504 Worker.call for ONE row u.  M = opposing factor matrix (dense index, row-major
 * n x k fp32), G = M^T M.  Scratch: W k*k, b k, work k*k+2k doubles, perm k ints.            */
static int solve_one_row(const int32_t* cols, const float* vals, int64_t n_u, const float* M, int k,
                         const double* G, double alpha, double lambda_times_alpha, int flags,
                         double threshold, float* x_out, int* apparent_rank, double* W, double* b,
                         double* work, int* perm) {
  const int reconstruct = flags & ORACLE_FLAG_RECONSTRUCT_R;
  const int loss_ignores = flags & ORACLE_FLAG_LOSS_IGNORES_UNSPECIFIED;
  if (loss_ignores) {
    /* ALS:524-539 partialTransposeTimesSelf: sum over the row's entries of (float)(v[r]*v[c]) */
    memset(W, 0, sizeof(double) * (size_t)k * k);
    for (int64_t e = 0; e < n_u; e++) {
      const float* v = M + (int64_t)cols[e] * k;
      for (int r = 0; r < k; r++) {
        float rv = v[r];
        for (int c = 0; c < k; c++) {
          float p = rv * v[c];
          W[(size_t)r * k + c] += (double)p;
        }
      }
    }
  } else {
    memcpy(W, G, sizeof(double) * (size_t)k * k); /* ALS:450 YTY.copy() */
  }
  for (int i = 0; i < k; i++) b[i] = 0.0;

  for (int64_t e = 0; e < n_u; e++) {
    double xu = (double)vals[e];                   /* ALS:457 */
    const float* v = M + (int64_t)cols[e] * k;     /* ALS:459 Y.get(id) */
    if (reconstruct) {
      for (int r = 0; r < k; r++) b[r] += xu * (double)v[r]; /* ALS:466-469 */
    } else {
      double cu = 1.0 + alpha * fabs(xu);          /* ALS:471 */
      for (int r = 0; r < k; r++) {
        float vr = v[r];
        double row_value = (double)vr * (cu - 1.0); /* ALS:474 */
        double* Wr = W + (size_t)r * k;
        for (int c = 0; c < k; c++) Wr[c] += row_value * (double)v[c]; /* ALS:477 */
        if (xu > 0.0) b[r] += (double)vr * cu;      /* ALS:480-482 */
      }
    }
  }
  double ltc = lambda_times_alpha * (double)n_u;    /* ALS:488 ru.size() */
  for (int d = 0; d < k; d++) W[(size_t)d * k + d] += ltc; /* ALS:489-492 */
  return rrqr_solve(W, b, k, threshold, x_out, apparent_rank, work, perm); /* ALS:494 */
}

typedef struct {
  const int64_t* row_ptr;
  const int32_t* col_idx;
  const float* val;
  int64_t row_begin, row_end;
  const float* M;
  int k;
  const double* G;
  double alpha, lambda_times_alpha, threshold;
  int flags;
  float* out; /* n_rows x k, indexed by absolute row */
  /* shared */
  int64_t next_unit;
  pthread_mutex_t mu;
  int status;
  int64_t bad_row;
  int bad_rank;
} solve_job;

static void* solve_worker(void* arg) {
  solve_job* j = (solve_job*)arg;
  int k = j->k;
  double* W = (double*)malloc(sizeof(double) * (2 * (size_t)k * k + 3 * (size_t)k));
  double* b = W + (size_t)k * k;
  double* work = b + k;
  int* perm = (int*)malloc(sizeof(int) * (size_t)k);
  for (;;) {
    pthread_mutex_lock(&j->mu);
    int64_t begin = j->next_unit;
    j->next_unit += WORK_UNIT_SIZE; /* ALS:391-410: work units of 100 rows */
    int stop = j->status != ORACLE_OK;
    pthread_mutex_unlock(&j->mu);
    if (stop || begin >= j->row_end) break;
    int64_t end = begin + WORK_UNIT_SIZE;
    if (end > j->row_end) end = j->row_end;
    for (int64_t u = begin; u < end; u++) {
      int64_t s = j->row_ptr[u], e = j->row_ptr[u + 1];
      int rank = 0;
      int rc = solve_one_row(j->col_idx + s, j->val + s, e - s, j->M, k, j->G, j->alpha,
                             j->lambda_times_alpha, j->flags, j->threshold,
                             j->out + u * (int64_t)k, &rank, W, b, work, perm);
      if (rc != ORACLE_OK) {
        pthread_mutex_lock(&j->mu);
        if (j->status == ORACLE_OK) {
          j->status = rc;
          j->bad_row = u;
          j->bad_rank = rank;
        }
        pthread_mutex_unlock(&j->mu);
        break;
      }
    }
  }
  free(W);
  free(perm);
  return NULL;
}

/* ALS:391-410 addWorkers + ALS:432-504: solve rows [row_begin,row_end) of one side given the
 * opposing factors M and their Gramian G.  `out` is the full n_rows x k output matrix.       */
int oracle_solve_rows(const int64_t* row_ptr, const int32_t* col_idx, const float* val,
                      int64_t row_begin, int64_t row_end, const float* M, int k, const double* G,
                      double alpha, double lambda, int flags, double threshold, float* out,
                      int threads, int64_t* bad_row, int* bad_rank) {
  solve_job j;
  memset(&j, 0, sizeof(j));
  j.row_ptr = row_ptr;
  j.col_idx = col_idx;
  j.val = val;
  j.row_begin = row_begin;
  j.row_end = row_end;
  j.M = M;
  j.k = k;
  j.G = G;
  j.alpha = alpha;
  j.lambda_times_alpha = lambda * alpha; /* ALS:435 */
  j.threshold = threshold;
  j.flags = flags;
  j.out = out;
  j.next_unit = row_begin;
  j.status = ORACLE_OK;
  j.bad_row = -1;
  pthread_mutex_init(&j.mu, NULL);
  if (threads < 1) threads = 1;
  if (threads == 1) {
    solve_worker(&j);
  } else {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, solve_worker, &j);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
  }
  pthread_mutex_destroy(&j.mu);
  if (bad_row) *bad_row = j.bad_row;
  if (bad_rank) *bad_rank = j.bad_rank;
  return j.status;
}

/* ALS:340-362 iterateXFromY (or ALS:367-389 with roles swapped): G = M^T M over ALL n_m rows of
 * the opposing matrix (stale rows included), then solve every row of this side.    */
int oracle_half_iteration(const int64_t* row_ptr, const int32_t* col_idx, const float* val,
                          int64_t n_rows, const float* M, int64_t n_m, int k, double alpha,
                          double lambda, int flags, double threshold, float* out, int threads,
                          int64_t* bad_row, int* bad_rank) {
  double* G = (double*)malloc(sizeof(double) * (size_t)k * k);
  oracle_gramian(M, n_m, k, G); /* ALS:342 -- serial on the caller thread, like the reference */
  int rc = oracle_solve_rows(row_ptr, col_idx, val, 0, n_rows, M, k, G, alpha, lambda, flags,
                             threshold, out, threads, bad_row, bad_rank);
  free(G);
  return rc;
}

/* DWM:73-81 increment(datum, weight) */
static void dwm_increment(double* total_weight, double* mean, double datum, double weight) {
  double old = *total_weight;
  *total_weight += weight;
  if (old <= 0) {
    *mean = datum;
  } else {
    *mean = *mean * old / *total_weight + datum * weight / *total_weight;
  }
}

/* ALS:176-262 call().  X: n_users x k (out; its input content is ignored -- ALS:179 builds X from
 * scratch, estimates start at 0, ALS:215-223).  Y: n_y x k (in: initial Y, ALS:182; out: final).
 * Only the first n_items rows of Y are re-solved; rows beyond are stale rows that still count in
 * Y^T Y.  test_users/test_items: the convergence sample (ALS:206-215; the reference
 * draws it with RandomUtils.chooseAboutNFromStream -- when n >= stream size that is "all ids").
 * random_y: Y was random (ALS:181) => never converge after iteration 1 (ALS:252-253).
 * iterate: model.als.iterate (ALS:196-204): if 0, run iterateXFromY once and return.          */
int oracle_als_call(const int64_t* r_row_ptr, const int32_t* r_col_idx, const float* r_val,
                    const int64_t* c_row_ptr, const int32_t* c_col_idx, const float* c_val,
                    int64_t n_users, int64_t n_items, int64_t n_y, int k, double alpha,
                    double lambda, int flags, double sing_threshold, double conv_threshold,
                    int max_iterations, int random_y, int iterate, const int64_t* test_users,
                    int n_tu, const int64_t* test_items, int n_ti, float* X, float* Y, int threads,
                    int* iterations_out, double* conv_out, int64_t* bad_row, int* bad_rank) {
  int rc;
  if (iterations_out) *iterations_out = 0;
  if (conv_out) *conv_out = NAN;
  if (!iterate) {
    return oracle_half_iteration(r_row_ptr, r_col_idx, r_val, n_users, Y, n_y, k, alpha, lambda,
                                 flags, sing_threshold, X, threads, bad_row, bad_rank);
  }
  double* est = (double*)calloc((size_t)n_tu * (size_t)n_ti + 1, sizeof(double));
  int it = 0;
  for (;;) {
    rc = oracle_half_iteration(r_row_ptr, r_col_idx, r_val, n_users, Y, n_y, k, alpha, lambda,
                               flags, sing_threshold, X, threads, bad_row, bad_rank); /* ALS:228 */
    if (rc != ORACLE_OK) break;
    rc = oracle_half_iteration(c_row_ptr, c_col_idx, c_val, n_items, X, n_users, k, alpha, lambda,
                               flags, sing_threshold, Y, threads, bad_row, bad_rank); /* ALS:229 */
    if (rc != ORACLE_OK) break;
    double tw = 0.0, mean = NAN; /* DWM:40-42 */
    for (int i = 0; i < n_tu; i++) {
      for (int j = 0; j < n_ti; j++) {
        double nv = oracle_dot(X + test_users[i] * (int64_t)k, Y + test_items[j] * (int64_t)k, k);
        double ov = est[(size_t)i * n_ti + j];
        est[(size_t)i * n_ti + j] = nv;
        dwm_increment(&tw, &mean, fabs(nv - ov), nv > 0.0 ? nv : 0.0); /* ALS:236 */
      }
    }
    it++;
    if (iterations_out) *iterations_out = it;
    if (conv_out) *conv_out = mean;
    if (max_iterations > 0 && it >= max_iterations) break; /* ALS:242-245 */
    if (!isfinite(mean)) break;                               /* ALS:248-251 */
    if (!(random_y && it == 1) && mean < conv_threshold) break; /* ALS:253-256 */
  }
  free(est);
  return rc;
}

/* MU:155-165 multiplyXYT restricted to dense ids: P[i][j] = dot(X[i], Y[j]). */
void oracle_multiply_xyt(const float* X, int64_t nx, const float* Y, int64_t ny, int k, double* P) {
  for (int64_t i = 0; i < nx; i++)
    for (int64_t j = 0; j < ny; j++) P[i * ny + j] = oracle_dot(X + i * k, Y + j * k, k);
}
