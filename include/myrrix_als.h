/*
 * myrrix_als.h -- C-ABI of the MI355X-native ALS matrix-factorization core (libmyrrix_als.so).
 *
 * This is the drop-in boundary behind myrrix-recommender's
 *   net.myrrix.online.factorizer.MatrixFactorizer            (online/src/.../MatrixFactorizer.java:31-77)
 *   net.myrrix.online.factorizer.als.AlternatingLeastSquares (online/src/.../als/AlternatingLeastSquares.java:66)
 * Plain pointers and sizes only; no C++/torch types.  The reference has no FFI for this path (it is
 * pure Java); the JNI stub a maintainer would add to bind these entry points is shown in
 * INTEGRATION.md, next to the Java adapter that keeps the 5-argument constructor of
 * AlternatingLeastSquares.java:132-136.  Abbreviations below: ALS = that file, MU =
 * common/src/net/myrrix/common/math/MatrixUtils.java, CMLSS =
 * common/src/net/myrrix/common/math/CommonsMathLinearSystemSolver.java.
 *
 * Data model.  The Java side keeps FastByIDMap<FastByIDFloatMap> RbyRow / RbyColumn and
 * FastByIDMap<float[]> X / Y; the adapter maps 64-bit ids to dense row indices and hands the native
 * side CSR arrays: "side X" = rows of R (users; its factor matrix is X, solved from Y), "side Y" =
 * rows of R^T (items; factor matrix Y, solved from X).  Factor matrices are row-major fp32,
 * n_rows_total x features, resident in HBM.  n_rows_total may exceed the number of matrix rows: Y may
 * carry stale rows that are never re-solved but still count in Y^T Y (ALS:304-308,342), and a
 * multi-GPU shard layout may pad each rank's slice.  One handle drives ONE GPU and holds the
 * matrix rows [row_offset, row_offset+n_rows_local) of each side plus FULL replicas of X and Y;
 * the caller exchanges freshly solved slices between handles (all-gather) -- see
 * mals_factor_device_ptr / mals_bind_factors.
 *
 * Every function returns a status code and never throws across the ABI.
 */
#ifndef MYRRIX_ALS_H
#define MYRRIX_ALS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MALS_ABI_VERSION 5

typedef struct mals_handle_s* mals_handle;

/* Status codes.  Java adapter mapping (ALS:349, CMLSS:46-54, MatrixFactorizer.java:41-47):
 * SINGULAR -> ExecutionException(SingularMatrixSolverException(apparentRank)),
 * CANCELLED -> InterruptedException, everything else -> ExecutionException(IllegalStateException). */
enum {
  MALS_OK = 0,
  MALS_SINGULAR = 1,
  MALS_INVALID_ARG = 2,
  MALS_HIP_ERROR = 3,
  MALS_COMM_ERROR = 4,
  MALS_CANCELLED = 5,
  MALS_OOM = 6,
  MALS_ILL_CONDITIONED = 7, /* IllConditionedSolverException (Generation.java:150-153) */
  MALS_IO_ERROR = 8         /* java.io.IOException (unreadable / corrupt / truncated model file) */
};

enum { MALS_SIDE_X = 0, MALS_SIDE_Y = 1 };

/* ALS:85-91: model.reconstructRMatrix / model.lossIgnoresUnspecified (static finals there). */
enum { MALS_FLAG_RECONSTRUCT_R = 1, MALS_FLAG_LOSS_IGNORES_UNSPECIFIED = 2 };

enum { MALS_MEM_HOST = 0, MALS_MEM_DEVICE = 1 };

enum { MALS_GRAMIAN_AUTO = 0, MALS_GRAMIAN_FP32 = 1, MALS_GRAMIAN_SPLIT_F16 = 2, MALS_GRAMIAN_SPLIT3_F16 = 3 };

enum { MALS_SOLVE_AUTO = 0, MALS_SOLVE_DIRECT = 1, MALS_SOLVE_DUAL = 2 };

typedef struct mals_config {
  int32_t struct_size;          /* sizeof(mals_config), for ABI evolution                      */
  int32_t features;             /* k; ALS:134 "features must be positive"; 1..128 supported    */
  double alpha;                 /* model.als.alpha,  default 1.0 (ALS:71,506-509)              */
  double lambda;                /* model.als.lambda, default 0.1 (ALS:73,511-514); the ridge
                                   applied per row is lambda*alpha*n_u (ALS:435,488)           */
  double singularity_threshold; /* common.matrix.singularityThreshold, default 1e-5
                                   (LinearSystemSolver.java:33-34)                             */
  int32_t flags;                /* MALS_FLAG_*                                                 */
  int32_t device;               /* HIP device ordinal                                          */
  int32_t segment_nnz;          /* rows longer than this are split across waves; 0 = default   */
  int32_t chunk_rows;           /* >0: the shard's rows are cut into contiguous ranges of this many
                                   rows that can be solved one by one (mals_solve_chunk), so that
                                   the caller can exchange a finished range while the next one is
                                   being solved; 0 = one chunk                                  */
  int32_t gramian_mode;         /* arithmetic of the per-row Gramian sum (c-1) y y^T (ALS:471-477):
                                   MALS_GRAMIAN_FP32: fp32 products, fp32 accumulate;
                                   MALS_GRAMIAN_SPLIT_F16: operands split into two f16 halves (22
                                   significand bits), exact products, fp32 accumulate (the rank-16
                                   updates of the factorization use the same operands) -- 2.5x less
                                   matrix-pipe time, rounding error on a par with FP32 (measured
                                   2-4e-7 vs the fp64 reference for both);
                                   MALS_GRAMIAN_SPLIT3_F16 (features 49..64 only; a measured alternative,
                                   DESIGN.md section 6): THREE f16 terms per operand -- every fp32 operand
                                   exactly -- and the six products at or above 2^-24 of the leading one:
                                   fp32-operand arithmetic on the f16 pipe, twice SPLIT_F16's matrix time;
                                   MALS_GRAMIAN_AUTO (default): FP32 for features <= 16 (where the
                                   products are not the bottleneck), SPLIT_F16 above            */
  int32_t solve_mode;           /* how a row with FEWER ENTRIES THAN FEATURES is solved (ALS:447-494 is the
                                   same k x k system either way):
                                   MALS_SOLVE_DIRECT: the k x k system W x = b, whatever the row length;
                                   MALS_SOLVE_DUAL: rows with n_u <= 16*floor(ceil(k/16)/2) entries through
                                   the n_u x n_u system of the push-through identity in the eigenbasis of
                                   the shared Gramian (csrc/dual_kernels.h: same x, 8x fewer factorization
                                   flops at n_u <= k/2), the others directly; needs the reference's default
                                   mode (flags = 0), alpha > 0 and min eig(G) + lambda*alpha >= 1e-4 and
                                   otherwise falls back to DIRECT for that half-iteration;
                                   MALS_SOLVE_AUTO (default): DUAL for features > 32, DIRECT below    */
} mals_config;

typedef struct mals_stats {
  int32_t struct_size;
  int32_t reserved;
  /* Accumulated since mals_reset_stats.  Times are HIP-event milliseconds measured on the handle's
   * stream around each kernel launch, only while timing is enabled (mals_enable_timing); bytes are
   * the ALGORITHMIC bytes of SURVEY.md section 8(d) attributed to that kernel's launches:
   * entries*(4k+4+4) gathered + rows*(4k+8) written.                                            */
  double rows_ms;           /* als_rows_kernel: fused gather+Gramian+Cholesky, rows <= segment_nnz */
  double segments_ms;       /* als_segments_kernel: gather+Gramian partials of long rows           */
  double finish_ms;         /* als_finish_kernel: partial sums + Cholesky of long rows             */
  double gramian_ms;        /* gramian_partial_kernel + gramian_finalize_kernel (K1)               */
  int64_t rows_launches;
  int64_t segments_launches;
  int64_t finish_launches;
  int64_t gramian_launches;
  double rows_bytes;
  double segments_bytes;
  double finish_bytes;
  double gramian_bytes;     /* rows*4k read                                                        */
  int64_t rows_solved;
  int64_t nnz_gathered;
  /* appended in ABI version 2 */
  double dual_ms;           /* als_dual_kernel: short rows through the n_u x n_u system               */
  double rotate_ms;         /* rotate_rows_kernel: M Q before, x = Q x' after the dual kernels        */
  int64_t dual_launches;
  int64_t rotate_launches;
  double dual_bytes;        /* entries*(4k+8) + rows*(4k+8), like rows_bytes                          */
  double rotate_bytes;      /* rows*(4k + 64*ceil(k/16)) forward, rows*8k in place                    */
  int64_t rows_dual;        /* rows solved by the dual kernels (included in rows_solved)              */
  double eigen_host_ms;     /* host time of the k x k eigendecompositions (overlapped with kernels)   */
  int64_t rows_refined;     /* rows re-solved with fp64 residuals (mals_set_refine_limit)             */
} mals_stats;

int mals_abi_version(void);

/* Fill cfg with the reference's defaults (k=30, alpha=1, lambda=0.1, threshold 1e-5;
 * MatrixFactorizer.java:34, ALS:71-75). */
int mals_default_config(mals_config* cfg);

/* Create a handle on cfg->device.  Replaces `new AlternatingLeastSquares(...)` (ALS:132-147). */
int mals_create(const mals_config* cfg, mals_handle* out);
int mals_destroy(mals_handle h);

/* Why the last mals_create / mals_group_create / mals_group_create_rank / mals_group_unique_id ON THIS THREAD failed --
 * there is no handle to ask then ("no HIP device", "device ordinal 3 outside 0..0", "librccl.so.1: cannot open ...",
 * "ncclCommInitAll: ...").  Copies at most cap - 1 characters + NUL into buf, returns the full length ("" / 0 after a
 * success).  mals_group_create_error is the same call under the group's name.  What the JVM adapter puts into the
 * ExecutionException instead of a bare "create failed" (jni/myrrix_als_jni.c nativeCreateError). */
int mals_create_error(char* buf, size_t cap);
int mals_group_create_error(char* buf, size_t cap);

/* cfg.features of the handle (0 for a null handle): lets a binding size-check factor and query-vector arrays. */
int mals_features(mals_handle h);

/* Human-readable message for the last non-OK status on this handle ("" if none). */
const char* mals_last_error(mals_handle h);

/* Use an existing hipStream_t (e.g. the caller's current stream) for all work; NULL = default. */
int mals_set_stream(mals_handle h, void* hip_stream);

/* Declare the number of rows of a side's factor replica and (re)allocate it zero-filled, unless
 * the caller binds its own device buffer with mals_bind_factors. */
int mals_set_factor_rows(mals_handle h, int side, int64_t n_rows_total);

/* Use caller-owned device memory (row-major n_rows_total x features fp32) as the factor replica
 * of `side`.  The caller keeps it alive until mals_destroy or the next bind. */
int mals_bind_factors(mals_handle h, int side, float* device_ptr, int64_t n_rows_total);

/* Device pointer of the factor replica of `side` (for the caller's collective). */
int mals_factor_device_ptr(mals_handle h, int side, void** out_device_ptr, int64_t* out_n_rows_total);

/* Upload the local matrix rows of a side: rows [row_offset, row_offset+n_rows_local) of R (side X)
 * or R^T (side Y) in CSR with local row_ptr (n_rows_local+1 entries, row_ptr[0]=0), col_idx indexing
 * rows of the OPPOSITE side's factor replica.  This is what iterating RbyRow / RbyColumn delivers
 * (ALS:399,455).  mem_kind = MALS_MEM_DEVICE borrows the caller's device arrays (no copy; keep
 * them alive); MALS_MEM_HOST copies.  n_u of a row = its entry count (ALS:488 ru.size()). */
int mals_set_matrix(mals_handle h, int side, int64_t row_offset, int64_t n_rows_local, int64_t nnz,
                    const int64_t* row_ptr, const int32_t* col_idx, const float* val, int mem_kind);

/* Chunked variant for callers that cannot hold one array per matrix (Java arrays are < 2^31):
 * begin, then append consecutive row chunks (host memory; row_ptr_chunk has n_rows+1 entries
 * starting at 0), then end. */
int mals_begin_matrix(mals_handle h, int side, int64_t row_offset, int64_t n_rows_local, int64_t nnz);
int mals_append_rows(mals_handle h, int side, int64_t n_rows, const int64_t* row_ptr_chunk,
                     const int32_t* col_idx, const float* val);
int mals_end_matrix(mals_handle h, int side);

/* Largest |value| of the local matrix rows of `side` (computed at upload), and an override for it.
 * The split-precision Gramian takes its operand scale from this bound; a multi-GPU caller installs
 * the maximum over all shards so that every rank uses the same scale and the factors do not depend
 * on how the rows are sharded.  The override must be >= the local maximum. */
int mals_get_value_bound(mals_handle h, int side, float* max_abs_value);
int mals_set_value_bound(mals_handle h, int side, float max_abs_value);
/* The same with the second statistic the split-precision gather looks at: sum of |value| and number of
 * entries of the local rows (their ratio is the "typical" weight of the operand-range check that decides
 * per launch between the split-precision and the fp32 gather).  A multi-GPU caller adds the sums and
 * counts of all shards and installs max and mean everywhere. */
int mals_get_value_stats(mals_handle h, int side, float* max_abs_value, double* sum_abs_value, int64_t* n_values);
int mals_set_value_stats(mals_handle h, int side, float max_abs_value, double mean_abs_value);

/* Host <-> device factor rows.  setPreviousY (ALS:172-174) = mals_set_factors(MALS_SIDE_Y, ...);
 * getX()/getY() (ALS:149-157) = mals_get_factors. */
int mals_set_factors(mals_handle h, int side, int64_t row_begin, int64_t n_rows, const float* host_rows);
int mals_get_factors(mals_handle h, int side, int64_t row_begin, int64_t n_rows, float* host_out);
int mals_get_rows(mals_handle h, int side, const int64_t* row_idx, int32_t n, float* host_out);

/* K1.  G = M^T M over ALL rows of `side`'s factor replica (MU:219-239 as called at ALS:342,369),
 * fp64, left on the device for the next mals_solve_side of the OTHER side; optionally copied to
 * host_G (features*features doubles, row-major) when host_G != NULL. */
int mals_gramian(mals_handle h, int side, double* host_G);

/* Multi-GPU variant: partial Gramian of rows [row_begin,row_begin+n_rows) of `side`'s replica into
 * device_out (features*features doubles); the caller all-reduces and installs the sum. */
int mals_gramian_partial(mals_handle h, int side, int64_t row_begin, int64_t n_rows, double* device_out);
int mals_set_gramian(mals_handle h, int side, const double* G, int mem_kind);

/* K2+K3.  Solve the local rows of `side` from the opposite side's factors and Gramian
 * (ALS:391-410 addWorkers + ALS:432-504 Worker.call): per row W = G + sum (c-1) y y^T +
 * lambda*alpha*n_u I, b = sum_{r>0} c y, x = W^-1 b, written into rows
 * [row_offset, row_offset+n_rows_local) of `side`'s factor replica.  Asynchronous on the handle's
 * stream; errors (singular rows) are reported by the next mals_check. */
int mals_solve_side(mals_handle h, int side);

/* The same for one chunk: local rows [chunk*cfg.chunk_rows, (chunk+1)*cfg.chunk_rows) of the shard
 * (all chunks, in any order, = mals_solve_side).  mals_num_chunks = ceil(n_rows_local/chunk_rows). */
int mals_solve_chunk(mals_handle h, int side, int32_t chunk);
int mals_num_chunks(mals_handle h, int side, int32_t* n_chunks);
/* cfg.chunk_rows for one side only (the two sides of a shard usually differ in size); takes effect at the
 * next matrix upload of that side.  0 = one chunk. */
int mals_set_chunk_rows(mals_handle h, int side, int64_t chunk_rows);

/* Synchronise the stream and report MALS_SINGULAR if any row solved since the last check had a
 * non-positive-definite system (the reference throws SingularMatrixSolverException, CMLSS:46-54). */
int mals_check(mals_handle h);
/* Details of the last MALS_SINGULAR: the side, the (global) row -- -1 when it was a model Gramian
 * (mals_recompute_solver) -- and the apparent rank the reference would report (CMLSS:47,
 * RRQRDecomposition.getRank(0.01)), which DelegateGenerationManager.java:345-354 uses to lower
 * model.features and retry.  For a row, the rank is recomputed on the host in fp64 from the row's
 * entries and the CURRENT opposite factors / Gramian, i.e. it is exact when the check directly
 * follows the solve (mals_half_iteration, mals_factorize); 0 = could not be determined. */
int mals_singular_info(mals_handle h, int32_t* side, int64_t* row, int32_t* apparent_rank);

/* iterateXFromY (ALS:340-362) for side X, iterateYFromX (ALS:367-389) for side Y:
 * mals_gramian(opposite) + mals_solve_side(side) + mals_check. Single-GPU convenience. */
int mals_half_iteration(mals_handle h, int side);

/* call() (ALS:176-262) on one GPU: alternate half-iterations until the convergence rule fires.
 * test_users / test_items: dense indices of the convergence sample (the adapter draws them with the
 * reference's own RandomUtils.chooseAboutNFromStream, ALS:206-215).  random_y: Y was random
 * (ALS:181,253).  iterate = 0: model.als.iterate=false (ALS:196-204).  max_iterations <= 0: no cap. */
int mals_factorize(mals_handle h, double convergence_threshold, int32_t max_iterations,
                   int32_t random_y, int32_t iterate, const int64_t* test_users, int32_t n_test_users,
                   const int64_t* test_items, int32_t n_test_items, int32_t* iterations_out,
                   double* convergence_out);

/* The convergence sample on the device (ALS:230-238): host_out[i*n_test_items+j] =
 * SimpleVectorMath.dot(X[test_users[i]], Y[test_items[j]]) (float product, double sum, features in order --
 * bit-identical to the Java loop) from the resident factors; 8 bytes per pair cross PCIe instead of the
 * sampled rows.  mals_factorize / mals_group_factorize use it every iteration and keep only the
 * order-dependent DoubleWeightedMean on the host. */
int mals_sample_dots(mals_handle h, const int64_t* test_users, int32_t n_test_users, const int64_t* test_items,
                     int32_t n_test_items, double* host_out);

/* The host eigensolver of the dual path (csrc/host_eigen.h; no GPU needed): A = V diag(evals) V^T for a symmetric
 * row-major n x n matrix (Householder tridiagonalisation + implicit QR; evals in no particular order, column j of
 * the row-major V is the unit eigenvector of evals[j]).  MALS_INVALID_ARG for a non-finite input or no convergence. */
int mals_symmetric_eigen(const double* A, int32_t n, double* evals_out, double* V_out);

/* Cooperative cancellation (InterruptedException path, MatrixFactorizer.java:43-44): checked
 * between half-iterations of mals_factorize. */
int mals_cancel(mals_handle h);

/* ---- the model-load consumer of K1 --------------------------------
 * Generation.recomputeSolver (online/src/net/myrrix/online/generation/Generation.java:142-158):
 * M^T M of `side`'s factors (K1 on the device), its max-abs-column-sum norm (getNorm(), :149), and
 * MatrixUtils.getSolver(M^T M) (MU:137, CMLSS:37-55).  The k x k factorization is host fp64.
 *   MALS_OK, *out = NULL        the side has no factor rows ("M == null || M.isEmpty()", :145)
 *   MALS_ILL_CONDITIONED        norm < 1.0 (:150-153); *inf_norm_out holds the norm
 *   MALS_SINGULAR               getSolver threw; rank via mals_singular_info (row = -1)
 * The Gramian stays installed on the device like after mals_gramian. */
typedef struct mals_solver_s* mals_solver;
int mals_recompute_solver(mals_handle h, int side, mals_solver* out, double* inf_norm_out);

/* MatrixUtils.getSolver(A) for a caller-supplied row-major n x n matrix (host only, no device
 * needed).  MALS_SINGULAR: *apparent_rank_out = getRank(0.01) and *out = NULL. */
int mals_solver_create(const double* A, int32_t n, double singularity_threshold, mals_solver* out,
                       int32_t* apparent_rank_out);
int mals_solver_dim(mals_solver s);
/* Solver.solveDToF / solveFToD (Solver.java:35-41, CommonsMathSolver.java:37-59). */
int mals_solver_solve_dtof(mals_solver s, const double* b, float* x);
int mals_solver_solve_ftod(mals_solver s, const float* b, double* x);
int mals_solver_destroy(mals_solver s);

/* ---- reconstruction metric on the device -------------------------------
 * ReconstructionEvaluator.evaluate (online/src/net/myrrix/online/eval/ReconstructionEvaluator.java:
 * 91-102): over every stored entry (u,i) of the local rows of side X, err = max(0, 1 - dot(X_u, Y_i))
 * with SimpleVectorMath.dot (fp32 products, fp64 sum).  Returns the sum and the number of entries; the
 * evaluator's result is sum/count (a multi-GPU caller adds the ranks' sums and counts first).
 * (The convergence statistic of call() already crosses PCIe as 200 sampled rows only, mals_factorize.) */
int mals_reconstruction_error(mals_handle h, double* sum_out, int64_t* count_out);

/* ---- top-N scoring -------------------------------------------------------
 * ServerRecommender.recommend(userID, howMany, considerKnownItems, null) (online/src/net/myrrix/online/
 * ServerRecommender.java:382-441) -> multithreadedTopN (:443-508) -> RecommendIterator (RecommendIterator.
 * java:62-109) -> TopN (common/src/net/myrrix/common/TopN.java:49-128), for a batch of model users given by
 * their dense indices.  The score of item i IS the reference's: (float)(sum_j dot(Y_i, f_j) / n) over the
 * query's n vectors with SimpleVectorMath.dot (every product rounded to fp32, fp64 sum in feature order,
 * RecommendIterator.java:93-104) -- bit-identical; the user's known items (the entries of its row of R on this
 * handle) are skipped unless consider_known_items; the how_many best come back best first, equal scores in
 * ascending item index (the reference leaves ties in hash order).  item_idx_out / score_out: n_queries x
 * how_many, padded with -1 / -inf; n_out (may be NULL): results per query.  Rescorers, candidate filters and
 * tags stay with the caller (tag items: pass them as exclusions).  On large item sets a bf16 MFMA pass with a
 * proven error margin only narrows the items down to a few hundred candidates per query, whose scores are then
 * computed exactly (csrc/topn_kernels.h); Y is streamed once per up to 240 queries, the passes of a call overlap
 * on six internal streams (ordered after the work already on the handle's stream; the call returns when all have
 * finished). */
int mals_recommend(mals_handle h, const int64_t* user_idx, int32_t n_queries, int32_t how_many, int32_t consider_known_items,
                   int64_t* item_idx_out, float* score_out, int32_t* n_out);
/* THREADS.  The reference's top-N is entered by every request thread at once, one user per call (ServerRecommender.java:
 * 359-441 -> multithreadedTopN :443-508).  mals_recommend, mals_recommend_vectors and mals_recommend_to_many may be
 * called on ONE handle from any number of threads concurrently (with each other -- not with calls that change the
 * handle's matrices, factors, known or tag items).  A call is cheap only as part of a pass (one read of Y answers up to 240
 * queries), so concurrent calls are folded into passes behind the ABI: every call becomes a ticket in the handle's queue;
 * the first thread that finds no leader leads -- it packs the queued calls of fewer than 64 queries into the next pass
 * (by-user calls together, calls with their own vectors -- an anonymous user, recommendToMany -- together), enqueues it,
 * decodes finished passes, wakes their callers -- and hands leadership on when its own answer is there.
 * At most `passes_in_flight` passes (default 2, environment MALS_TOPN_FRONT_DEPTH at mals_create) are on the device at a
 * time; what arrives meanwhile forms the next one.  Larger calls run exclusively, in queue order.  Every result is what the
 * call alone would have returned (bit-identical).  mals_recommend_front_stats:
 * out4 = {calls, queries, coalesced passes, exclusive calls} since mals_create. */
int mals_recommend_front_stats(mals_handle h, int64_t* out4);
int mals_recommend_set_depth(mals_handle h, int32_t passes_in_flight);   /* 1..6; not while calls are in flight */
/* How long a caller whose call is in somebody else's pass polls for its answer before it blocks on a condition variable
 * (default 0: block at once; a process with cores to spare trades them for wake-up latency). */
int mals_recommend_set_spin_us(mals_handle h, int32_t spin_us);
/* userTagIDs (RecommendIterator.java:72 -- "if (userTagIDs.contains(itemID)) return null"; likewise MostSimilarItemIterator
 * .java:77): rows of Y that stand for tags of users (InputFilesReader.java:159-165) are never recommended, to anybody, by
 * any mals_recommend* call.  item_idx: n dense item indices (host or device; entries outside [0, rows of Y) are ignored:
 * a tag whose entries were all removed owns no row).  n = 0 clears.  Needs the item factor rows declared
 * (mals_set_factor_rows).  mals_ingest_install / mals_ingest_install_group hand the ingest's userTagIDs over. */
int mals_set_tag_items(mals_handle h, int64_t n, const int64_t* item_idx, int mem_kind);
int mals_get_tag_item_count(mals_handle h, int64_t* n_out);   /* distinct rows of Y currently struck */
/* The same for caller-supplied query vectors (n_queries x features, host) -- anonymous users / fold-in
 * (SR:561-606) -- with optional per-query lists of item indices to skip (CSR: exclude_ptr has
 * n_queries+1 entries; both NULL = none). */
int mals_recommend_vectors(mals_handle h, const float* query_vectors, int32_t n_queries, int32_t how_many,
                           const int64_t* exclude_ptr, const int64_t* exclude_idx, int64_t* item_idx_out, float* score_out,
                           int32_t* n_out);
/* knownItemIDs for mals_recommend (ServerRecommender.java:394-425 takes them from generation.getKnownItemIDs(), which
 * differ from the rows of R by the entries InputFilesReader.removeSmall pruned): a CSR over the handle's local user rows
 * (n_rows = the local rows of side X) of dense item indices.  MALS_MEM_DEVICE arrays are borrowed, host arrays copied.
 * row_ptr NULL: back to the rows of R.  mals_ingest_install hands them over when the ingest built them. */
int mals_set_known_items(mals_handle h, int64_t n_rows, const int64_t* row_ptr, const int32_t* item_idx, int mem_kind);
/* ... and for queries of SEVERAL vectors each -- recommendToMany (ServerRecommender.java:366-441: the feature
 * vectors of all the users asked for; the caller passes the intersection of their known items, :398-425, as the
 * exclusion list): query q owns vectors[vector_ptr[q] .. vector_ptr[q+1]) (rows of `vectors`, features floats
 * each; at least one: RecommendIterator.java:52), its score is the mean of the dots, divided last
 * (RecommendIterator.java:93-104).  vector_ptr NULL = one vector per query. */
int mals_recommend_to_many(mals_handle h, const float* vectors, const int64_t* vector_ptr, int32_t n_queries, int32_t how_many,
                           const int64_t* exclude_ptr, const int64_t* exclude_idx, int64_t* item_idx_out, float* score_out,
                           int32_t* n_out);

/* ---- ingest -> CSR ------------------------------------------------------
 * What InputFilesReader.readInputFiles (online-local/src/net/myrrix/online/generation/
 * InputFilesReader.java:64-211) does to the parsed records of the input files, on the device: the
 * caller appends (user id, item id, value) records IN FILE ORDER -- value NaN = the line had an empty
 * value token = "remove this entry" (IFR:137,165-167) -- and mals_ingest_finish leaves in HBM exactly
 * the matrices the reference ends up with in RbyRow / RbyColumn:
 *   per (user,item) pair the records are replayed in order: NaN removes the entry (MU:102-125), a
 *   value starts it or is added to it with one fp32 add per record (MU:64-92, FastByIDFloatMap.java:
 *   129-138); a user / item exists iff it still owns an entry at the end of the stream (MU:117-125);
 *   then entries with |value| < zero_threshold (model.decay.zeroThreshold, IFR:58-59) are dropped,
 *   which may leave existing rows empty (IFR:200-211 removes entries, not rows).
 * Dense indices are assigned in ascending id order; CSR columns ascend within a row.  Side X = R by
 * user, side Y = R^T by item.  Records come from the caller (mals_ingest_append) or from the text of the
 * input files (mals_ingest_append_text / _read_file / _read_dir, below).  Up to 2^31 - 256 records go through one sort
 * pipeline; more (C5: 5e9 lines) are finished user-id range by user-id range (MALS_INGEST_OPT_PARTITION_RECORDS), bounded
 * by device memory: 24 bytes per record held + 8 per entry and matrix + the workspace of one range.  Dense indices are
 * 32-bit: at most 2^31 - 256 distinct users and items. */
typedef struct mals_ingest_s* mals_ingest;
int mals_ingest_create(int32_t device, float zero_threshold, mals_ingest* out);
int mals_ingest_destroy(mals_ingest g);
const char* mals_ingest_last_error(mals_ingest g);
int mals_ingest_append(mals_ingest g, int64_t n, const int64_t* user_ids, const int64_t* item_ids,
                       const float* values, int mem_kind);
int mals_ingest_finish(mals_ingest g);
int mals_ingest_counts(mals_ingest g, int64_t* n_records, int64_t* n_users, int64_t* n_items, int64_t* nnz);
/* dense index -> 64-bit id (n_users / n_items entries) */
int mals_ingest_get_ids(mals_ingest g, int side, int64_t* host_ids_out);
/* host copies of one CSR (any pointer may be NULL) / the device arrays themselves */
int mals_ingest_get_csr(mals_ingest g, int side, int64_t* host_row_ptr, int32_t* host_col_idx, float* host_val);
int mals_ingest_device_csr(mals_ingest g, int side, const int64_t** row_ptr, const int32_t** col_idx, const float** val);
/* mals_set_matrix(MALS_MEM_DEVICE) of both sides into a factorizer handle on the same device; the
 * handle borrows the arrays, so the ingest object must outlive their use. */
int mals_ingest_install(mals_ingest g, mals_handle h);
/* The device the ingest lives on, and its userTagIDs as rows of R^T: device_idx_out = a device array owned by the ingest
 * (valid until the next finish / destroy) holding, per userTagID in ascending id order (mals_ingest_get_tag_ids), its dense
 * item index, or -1 for a tag that owns no row at the end of the stream; n_out = the number of userTagIDs.  What
 * mals_ingest_install passes to mals_set_tag_items. */
int mals_ingest_device(mals_ingest g, int32_t* device_out);
int mals_ingest_device_tag_items(mals_ingest g, const int64_t** device_idx_out, int64_t* n_out);
int mals_ingest_get_tag_items(mals_ingest g, int64_t* host_idx_out);   /* the same array, copied to the host */
/* last finish: HIP-event milliseconds of the pipeline, host milliseconds spent (re)allocating its
 * workspace (52 bytes per record, kept for later finishes; hipMalloc of tens of GB is slow), algorithmic
 * bytes read+written by all passes, radix passes run */
int mals_ingest_stats(mals_ingest g, double* finish_ms, double* workspace_ms, double* bytes_moved, int32_t* radix_passes);
/* last finish: user-id ranges and item ranges it was cut into (0, 0: one pipeline) */
int mals_ingest_partitions(mals_ingest g, int32_t* user_ranges, int32_t* item_ranges);

/* ---- the TEXT half of the same row: bytes of the input files -> records, on the device -----------------
 * Replaces the line loop of InputFilesReader.readInputFiles (IFR = online-local/src/net/myrrix/online/generation/
 * InputFilesReader.java:88-192) over FileLineIterable (common/src/net/myrrix/common/iterator/FileLineIterator.java):
 *   lines      java.io.BufferedReader.readLine: '\n', '\r' or "\r\n" ends a line; bytes after the last terminator
 *              of a file are its last line
 *   IFR:92-98  one line counter and one bad-line counter over all files; the line FOLLOWING the 101st bad line
 *              fails the whole read with "Too many bad lines; aborting" (MALS_IO_ERROR, like the IOException)
 *   IFR:101    empty lines and lines starting with '#' are skipped
 *   IFR:51,105 Splitter.on(',').trimResults(): only the first three tokens are looked at; guava whitespace
 *              (ASCII and non-ASCII) is trimmed from each
 *   IFR:114-130 user and item tokens: Long.parseLong, or -- token starting with '"' -- a tag:
 *              OneWayMigrator.toLongID(token.substring(1, token.length()-1)) = first 8 bytes of the MD5 of the
 *              UTF-8 bytes, big-endian; a token that is a lone '"' makes the reference throw an unchecked
 *              StringIndexOutOfBoundsException out of readInputFiles: MALS_INVALID_ARG here
 *   IFR:132-137 value token: absent -> 1.0f; empty -> NaN = "remove this entry"; else LangUtils.parseFloat
 *              (Float.parseFloat's grammar, correctly rounded; NaN / infinite / overflowing values rejected)
 *   IFR:139-157 fewer than two tokens, or two tags -> bad line; an unparseable token -> bad line, except on line 1
 *              of the whole input, which is taken for a header and ignored
 *   IFR:159-165 tag in the user column -> its id joins itemTagIDs; in the item column -> userTagIDs
 *   IFR:173-191 knownItemIDs (MALS_INGEST_OPT_KNOWN_ITEMS): per user the items whose last line was not a remove
 * The parsed records join those of mals_ingest_append in stream order and go through the same mals_ingest_finish.
 * Everything after the bytes reach HBM runs on the device (csrc/ingest_text_kernels.h, csrc/text_parse.h); the
 * host lists and inflates files (zlib) and keeps the two sequential counters.  Behaviour of the JDK / guava /
 * mahout pieces this restates is listed, with what is pinned and what is not, at the top of csrc/text_parse.h.
 *
 * mals_ingest_append_text: `bytes` continue the current file (any split, also inside a line or between '\r' and
 * '\n'); end_of_file != 0 closes the file: its unterminated last line, if any, is a line.  MALS_MEM_DEVICE: the
 * bytes are already in HBM. */
/* KNOWN_ITEMS: also build knownItemIDs (off by default); TEXT_BLOCK_BYTES: bytes handed to the device at a time
 * (default 256 MiB); RESERVE_RECORDS: allocate the record arrays for this many records now (they grow by copying
 * otherwise) */
/* PARTITION_RECORDS: most records ONE sort pipeline is given (52 bytes of workspace each); an ingest with more is finished
 * user-id range by user-id range (csrc/ingest_big_host.h: same result, bit for bit).  0 = the default: one pipeline up to
 * 2^31 - 256 records; beyond, ranges as large as 70 % of the device's free memory hold at 76 bytes per record (at least
 * 2^26) -- which is how C5's 5e9 lines fit one 288 GB device next to their 120 GB of records.  (Tests set a few hundred
 * to run the oracle suites through the partitioned path.) */
enum { MALS_INGEST_OPT_KNOWN_ITEMS = 1, MALS_INGEST_OPT_TEXT_BLOCK_BYTES = 2, MALS_INGEST_OPT_RESERVE_RECORDS = 3, MALS_INGEST_OPT_PARTITION_RECORDS = 4 };
enum { MALS_ITEM_TAG_IDS = 0, MALS_USER_TAG_IDS = 1 };
int mals_ingest_set_option(mals_ingest g, int32_t option, int64_t value);
int mals_ingest_append_text(mals_ingest g, const void* bytes, int64_t n_bytes, int mem_kind, int32_t end_of_file);
/* One input file, by name like FileLineIterator.getFileInputStream (FLI:92-102): "*.gz" is inflated (all members, like
 * GZIPInputStream); "*.zip" yields NO lines -- the reference wraps it in a ZipInputStream on which getNextEntry()
 * is never called, and such a stream reads as empty; anything else is read as it is. */
int mals_ingest_read_file(mals_ingest g, const char* path);
/* IFR:71-86: the files of input_dir matching .+\.csv(\.(zip|gz))? in ascending last-modified order (equal
 * timestamps: by name; the reference leaves that order to File.listFiles()).  A missing directory reads nothing. */
int mals_ingest_read_dir(mals_ingest g, const char* input_dir, int32_t* n_files_read);
typedef struct mals_ingest_text_info_t {
  int32_t struct_size; /* sizeof(mals_ingest_text_info_t) */
  int32_t reserved;
  int64_t lines;             /* IFR:99 `lines` */
  int64_t bad_lines;         /* IFR:93 `badLines` */
  int64_t header_lines;      /* 0 or 1 */
  int64_t skipped_lines;     /* empty or comment */
  int64_t full_parser_lines; /* lines the fast (ASCII, numeric) parser handed to the full one */
  int64_t text_bytes;
  int64_t records;           /* records held (text and mals_ingest_append) */
  double parse_ms;           /* HIP-event milliseconds of the text kernels so far */
  double stage_ms;           /* ... of bringing the blocks in front of them: host-to-device copies (PCIe) for
                                MALS_MEM_HOST bytes, a device-to-device copy for MALS_MEM_DEVICE */
  int64_t n_item_tag_ids;    /* after mals_ingest_finish; -1 before */
  int64_t n_user_tag_ids;
  int64_t n_known_items;     /* entries of knownItemIDs; -1 if not requested / not finished */
} mals_ingest_text_info_t;
int mals_ingest_text_info(mals_ingest g, mals_ingest_text_info_t* out);
/* after mals_ingest_finish: ascending ids of itemTagIDs / userTagIDs */
int mals_ingest_get_tag_ids(mals_ingest g, int32_t which, int64_t* host_ids_out);
/* after mals_ingest_finish with MALS_INGEST_OPT_KNOWN_ITEMS: knownItemIDs as a CSR over the dense user indices
 * (n_users + 1 offsets) holding dense item indices, ascending within a user.  Its users are exactly the rows of
 * side X and its items rows of side Y; it differs from R's pattern by the entries removeSmall pruned (IFR:194-211). */
int mals_ingest_get_known_items(mals_ingest g, int64_t* host_ptr, int32_t* host_item_idx);
int mals_ingest_device_known_items(mals_ingest g, const int64_t** ptr, const int32_t** item_idx, int64_t* n_known);

/* ---- model.bin.gz, the file through which the factors leave and re-enter
 * the unmodified Java server.  Replaces GenerationSerializer.writeGeneration / readGeneration
 * (online-local/src/net/myrrix/online/generation/GenerationSerializer.java:84-95; body :96-262) and
 * IOUtils.writeObjectToFile / readObjectFromFile (common/src/net/myrrix/common/io/IOUtils.java:259-283):
 * a gzip member holding a Java Object Serialization stream (protocol version 2) with ONE object of
 * class net.myrrix.online.generation.GenerationSerializer (serialVersionUID 1, GS:51) whose custom
 * writeObject data (GS:96-105) is, big-endian, in 1024-byte block-data records:
 *   knownItemIDs  int count | -1 for null, then per user: long id, int n, n x long      GS:129-165
 *   X, Y          int count, then per row: long id, int length, length x float          GS:167-201
 *   itemTagIDs, userTagIDs   int count, count x long                                    GS:203-222
 *   userClusters, itemClusters  int count, per cluster: int n, n x long members,
 *                               int length, length x float centroid                     GS:224-262
 * Host only (no GPU, no handle).  Rows are written in the order given (the reference's order is the
 * hash order of its maps, which its reader does not depend on).  A non-finite factor is rejected on
 * both sides like Preconditions.checkState(LangUtils.isFinite(f)) (GS:167-201): MALS_INVALID_ARG.
 * Rows of X and Y must share one length ("features"); a file with ragged rows is MALS_INVALID_ARG.
 * Parity note: no JVM exists in the build image, so interoperability is pinned to the published
 * stream grammar (Java Object Serialization Specification, section 6.4) and to an independent
 * restatement in oracle/model_oracle.py -- not to a file written by the reference itself. */
typedef struct mals_model_view {
  int32_t struct_size; /* sizeof(mals_model_view) */
  int32_t features;
  int64_t n_users;
  const int64_t* user_ids;
  const float* X; /* n_users x features, row-major */
  int64_t n_items;
  const int64_t* item_ids;
  const float* Y;
  int64_t n_known; /* users with a known-item set; -1 = knownItemIDs is null (GS:131-133,148-149) */
  const int64_t* known_user_ids;
  const int64_t* known_ptr; /* n_known + 1 offsets into known_item_ids */
  const int64_t* known_item_ids;
  int64_t n_item_tags;
  const int64_t* item_tag_ids;
  int64_t n_user_tags;
  const int64_t* user_tag_ids;
  /* clusters: members of cluster c = members[member_ptr[c] .. member_ptr[c+1]), its centroid =
   * centroids[centroid_ptr[c] .. centroid_ptr[c+1]) */
  int64_t n_user_clusters;
  const int64_t* user_cluster_member_ptr;
  const int64_t* user_cluster_members;
  const int64_t* user_cluster_centroid_ptr;
  const float* user_cluster_centroids;
  int64_t n_item_clusters;
  const int64_t* item_cluster_member_ptr;
  const int64_t* item_cluster_members;
  const int64_t* item_cluster_centroid_ptr;
  const float* item_cluster_centroids;
} mals_model_view;
typedef struct mals_model_s* mals_model;
/* path must end in ".gz" (IOUtils.java:276) */
int mals_model_write(const char* path, const mals_model_view* model);
int mals_model_read(const char* path, mals_model* out);
/* the arrays of a model that was read; the pointers stay valid until mals_model_destroy */
int mals_model_get(mals_model m, mals_model_view* out);
int mals_model_destroy(mals_model m);
/* message of the last failed mals_model_* call of this thread */
const char* mals_model_last_error(void);

/* ---- the multi-GPU half-iteration below the C-ABI ------------------------------
 * Rows within a half-iteration are independent given the full opposite factor matrix and its Gramian --
 * how the reference threads it (ALS:391-410, one writer per output row ALS:497-499).  A GROUP is `world`
 * ranks, one GPU each, every rank holding the CSR rows of its slice of users and of items plus FULL
 * replicas of X and Y.  Per half-iteration: partial Gramian of the rank's own slice + k x k fp64 all-reduce;
 * solve the slice in `exchange_chunks` row chunks; as soon as a chunk is solved its rows go to every other
 * replica (in place, on a second stream, while the next chunk is being solved).  Slices are contiguous row
 * ranges balanced by COST (entries + row_cost per row, mals_plan_shards), not by row count, so the
 * exchange is an all-gather with per-rank counts: one grouped ncclSend/ncclRecv per peer pair (RCCL; on a
 * fully connected xGMI node every pair has its own link).
 * Two ways to form a group, same calls afterwards:
 *   mals_group_create         ONE process drives all GPUs (the JVM deployment): ncclCommInitAll;
 *   mals_group_create_rank    one process per GPU (torchrun / MPI style): rank 0 calls mals_group_unique_id,
 *                             ships the 128 bytes to the others by its own means, every rank calls
 *                             create_rank; all later mals_group_* calls are collective (every rank, same
 *                             order, same arguments apart from host buffers).
 * backend: MALS_GROUP_RCCL (RCCL, dlopen'ed: librccl.so.1 must be loadable) or MALS_GROUP_PEER_COPY
 * (single-process groups only: hipMemcpyPeerAsync between the replicas -- the SDMA engines move the
 * slices, no CU is taken from the solve -- and the k x k sum through the host).
 * Errors: MALS_COMM_ERROR for a failed RCCL call; a per-row error (MALS_SINGULAR ...) found by any rank is
 * returned by every rank. */
typedef struct mals_group_s* mals_group;
enum { MALS_GROUP_RCCL = 0, MALS_GROUP_PEER_COPY = 1 };

/* Contiguous row slices of equal cost: row r costs (row_ptr[r+1]-row_ptr[r]) + row_cost (row_cost < 0: a
 * default for `features`: the factorization of a row in entry-gathers, ~ features^2 / 200).
 * bounds_out[world+1]: slice j = rows [bounds_out[j], bounds_out[j+1]).  Host only. */
int mals_plan_shards(const int64_t* row_ptr, int64_t n_rows, int32_t world, double row_cost, int32_t features,
                     int64_t* bounds_out);

int mals_group_create(const mals_config* cfg, const int32_t* devices, int32_t n_devices, int32_t backend, mals_group* out);
int mals_group_unique_id(void* id_out_128_bytes);
int mals_group_create_rank(const mals_config* cfg, int32_t world, int32_t rank, const void* id_128_bytes, mals_group* out);
int mals_group_destroy(mals_group g);
const char* mals_group_last_error(mals_group g);
int mals_group_world(mals_group g);
int mals_group_features(mals_group g);   /* cfg.features (0 for a null group): lets a binding size-check factor arrays */
/* the handle of local member i (0 .. n_local-1; n_local = world for mals_group_create, 1 for create_rank)
 * and its rank -- for per-GPU calls such as mals_get_stats, mals_recommend, mals_reconstruction_error */
int mals_group_local(mals_group g, int32_t i, mals_handle* handle_out, int32_t* rank_out);
/* Row chunks per slice (default 4).  Takes effect with the NEXT matrix upload of each side: a side that already
 * holds a matrix keeps the count its work lists were built for. */
int mals_group_set_exchange_chunks(mals_group g, int32_t n_chunks);
/* Consecutive chunks of a half-iteration are solved on two alternating compute streams (default on): a chunk's tail -- the
 * last waves of its persistent kernels -- then overlaps the next chunk's head instead of draining the GPU at every chunk
 * boundary; each chunk's completion event is handed to the exchange stream as before and the streams are joined at the end
 * of the half-iteration.  0 = one stream (A/B; the factors are bitwise the same either way). */
int mals_group_set_alternate_streams(mals_group g, int32_t on);
/* Which shared library is loaded as RCCL: by default librccl.so.1 (a copy the process already mapped first).  A
 * process that wants a particular build -- or the tests' stand-in transport, tests/cpp/libmock_rccl.so -- says so
 * HERE, once, before its first group or unique id (MALS_INVALID_ARG afterwards); NULL = the default.  Deliberately
 * not an environment variable: what runs as "RCCL" inside a server process is the process's own decision. */
int mals_group_use_transport(const char* library_path);
/* What the communicator itself reports for local member i (ncclCommCount / ncclCommUserRank / ncclCommCuDevice;
 * 0 / -1 when the group has no communicator: world 1 or the peer-copy backend), the HIP device ordinal of the member
 * and its PCI bus id ("0000:c1:00.0"; pci_bus_id may be NULL) -- so that a benchmark line can prove how many ranks
 * RCCL saw and on which devices. */
int mals_group_comm_info(mals_group g, int32_t local_member, int32_t* comm_size, int32_t* comm_rank, int32_t* hip_device,
                         char* pci_bus_id, int32_t pci_len);
int mals_group_set_refine_limit(mals_group g, double limit);        /* mals_set_refine_limit on every local member */

/* Replicas: n_rows_total rows per side on every rank (>= the matrix rows: stale Y rows, ALS:304-308). */
int mals_group_set_factor_rows(mals_group g, int side, int64_t n_rows_total);
/* host rows -> every local replica / rows of local member 0's replica -> host */
int mals_group_set_factors(mals_group g, int side, int64_t row_begin, int64_t n_rows, const float* host_rows);
int mals_group_get_factors(mals_group g, int side, int64_t row_begin, int64_t n_rows, float* host_out);
int mals_group_get_rows(mals_group g, int side, const int64_t* row_idx, int32_t n, float* host_out);

/* The FULL matrix of a side (CSR, n_rows rows, global row_ptr; host or device arrays of the calling
 * process): slices are planned from row_ptr (mals_plan_shards with row_cost < 0) and every local member
 * uploads its own.  Device arrays are borrowed (col_idx / val) like mals_set_matrix(MALS_MEM_DEVICE). */
int mals_group_set_matrix(mals_group g, int side, int64_t n_rows, int64_t nnz, const int64_t* row_ptr,
                          const int32_t* col_idx, const float* val, int mem_kind);
/* The same for callers that cannot hold one array per matrix (Java arrays are < 2^31 entries): the full
 * row_ptr first (n_rows+1 longs always fit), then the entries of consecutive rows in any number of
 * pieces (each piece = whole rows), then end. */
int mals_group_begin_matrix(mals_group g, int side, int64_t n_rows, const int64_t* row_ptr);
int mals_group_append_rows(mals_group g, int side, int64_t n_rows, const int32_t* col_idx, const float* val);
int mals_group_end_matrix(mals_group g, int side);
/* How many entries the next n_rows rows of the chunked upload in progress hold (from the row_ptr given to begin): what a
 * binding checks its arrays against BEFORE mals_group_append_rows reads them. */
int mals_group_pending_entries(mals_group g, int side, int64_t n_rows, int64_t* n_entries_out);
/* InputFilesReader.readInputFiles -> the GROUP, without the detour through the host: after mals_ingest_finish both CSRs
 * are cut at the group's cost-balanced bounds (mals_plan_shards on the ingest's row pointers) and every local member takes
 * its slice of R and of R^T device to device -- borrowed in place when the member sits on the ingest's device, copied with
 * hipMemcpyPeerAsync into arrays the group owns otherwise -- together with its slice of knownItemIDs (when the ingest built
 * them) and the userTagIDs item mask.  Factor rows of both sides are declared from the ingest's counts (n_users, n_items)
 * unless already declared at least that large.  In a one-process-per-GPU group every rank calls this with its OWN ingest of
 * the same input (collective like mals_group_set_matrix).  flags = 0: the ingest must outlive the group's use of the
 * borrowed slices; MALS_INSTALL_COPY: members on the ingest's device copy their slices too (8 bytes per entry and side more
 * on that device) and the ingest -- its records and its 52 bytes per record of workspace -- can be destroyed right away. */
enum { MALS_INSTALL_COPY = 1 };
int mals_ingest_install_group(mals_ingest g, mals_group grp, int32_t flags);
/* ServerRecommender.recommend for model users on a group (arguments as mals_recommend): every member holds complete replicas
 * of X and Y but only ITS users' rows of R / knownItemIDs, so a query is answered by the local member whose slice holds the
 * user's row (consider_known_items: by any local member).  Callable from any number of request threads like mals_recommend:
 * it only reads the group and enters the members' serving fronts.  MALS_INVALID_ARG for a user whose owner is a rank of
 * another process (a one-process-per-GPU deployment routes the request to that process). */
int mals_group_recommend(mals_group g, const int64_t* user_idx, int32_t n_queries, int32_t how_many, int32_t consider_known_items,
                         int64_t* item_idx_out, float* score_out, int32_t* n_out);
/* slice bounds of a side after its matrix was set: bounds_out[world+1] */
int mals_group_bounds(mals_group g, int side, int64_t* bounds_out);

/* iterateXFromY / iterateYFromX (ALS:340-389) and call() (ALS:176-262) on the group; arguments as for
 * mals_half_iteration / mals_factorize. */
int mals_group_half_iteration(mals_group g, int side);
int mals_group_factorize(mals_group g, double convergence_threshold, int32_t max_iterations, int32_t random_y,
                         int32_t iterate, const int64_t* test_users, int32_t n_test_users, const int64_t* test_items,
                         int32_t n_test_items, int32_t* iterations_out, double* convergence_out);
/* details of the last MALS_SINGULAR as mals_singular_info reports them, from whichever local member hit it
 * (side = -1: none of the local members did) */
int mals_group_singular_info(mals_group g, int32_t* side, int64_t* row, int32_t* apparent_rank);
/* diagnostic: the exchange of a side on its own (the current slices into every replica again), to price the
 * wire separately from the solve it normally hides behind */
int mals_group_exchange_only(mals_group g, int side);
int mals_group_cancel(mals_group g);
/* wait for everything enqueued on the local members' streams (timing brackets) */
int mals_group_synchronize(mals_group g);

/* Accuracy on ill-conditioned rows.  The reference solves every row's k x k system in fp64 (ALS:494 ->
 * CMLSS:37-55); the kernels here accumulate and factor it in fp32, which costs about cond(W) 6e-8 of x -- inside the
 * 1e-4 bar up to cond(W) ~ 1e3, not beyond (confidence weights alpha|r| in the thousands against a small lambda).
 * Every solving kernel therefore estimates cond(W) from below -- max(largest entry of the row's own sum w y y^T,
 * a quarter of the largest entry of W) / (smallest pivot) -- and a row above `limit` is solved again
 * (als_refine_kernel): the same fp32 factor as preconditioner, conjugate gradients on the exact system with every
 * product in fp64 straight from the entries, the factor rows and the fp64 Gramian, until a step is below 1e-6 |x|.
 * Default 64 (environment MALS_REFINE_LIMIT overrides it at mals_create; a sixteenth of it applies under
 * MALS_FLAG_LOSS_IGNORES_UNSPECIFIED); 0 = never.  Independently of the limit, a row whose fp32 factorization
 * breaks down (pivot <= singularity_threshold) is re-done in fp64 with the reference's own roundings
 * (als_exact_kernel) before it is called singular.  mals_stats.rows_refined counts both. */
int mals_set_refine_limit(mals_handle h, double limit);

/* The operand scale of the split-precision gather as the last half-iteration computed it on the device (diagnostic):
 * out4 = {S, 1/S^2, range flag (1 = split-precision kernels ran, 0 = their fp32 twins), the bound on |y| that was
 * used}.  The bound is the exact max |element| of the gathered factor matrix whenever this library formed its Gramian
 * (mals_gramian, mals_half_iteration, mals_factorize, the group calls: the Gramian kernels record it, a group all-reduces
 * it), and sqrt(max_f G_ff) -- loose by up to sqrt(rows) -- after mals_set_gramian. */
int mals_get_gather_scale(mals_handle h, float* out4);

/* Host-side timeline of the current / last half-iteration on this handle (diagnostic), microseconds of one
 * process-wide steady clock: out4[0] = its first kernel was enqueued, out4[1] / out4[2] = begin / end of the host half
 * of the dual preparation (the k x k eigendecomposition, computed here or received from the group member that
 * computed it; 0 = none in this half-iteration), out4[3] reserved.  tests/test_gpu_group.py uses it to check that a
 * single-process group enqueues every member's kernels before any member's host work. */
int mals_get_timeline(mals_handle h, double* out4);

/* What call() logs per iteration in the reference -- "Finished iteration {}", "Avg absolute difference in estimate vs prior
 * iteration: {}" (ALS:241-246), "{} X/tag rows computed" (ALS:351-358) -- needs values out of the library WHILE
 * mals_factorize / mals_group_factorize run: the callback is called on the calling thread after every iteration's
 * convergence statistic, before the stop rules (ALS:242-256) are applied.  seconds = host wall time of the iteration;
 * rows / entries / algorithmic bytes of the iteration are what THIS process's members solved (a
 * one-process group: everything).  fn = NULL removes it. */
typedef struct mals_iteration_info {
  int32_t struct_size;
  int32_t iteration;             /* 1-based */
  double avg_abs_difference;     /* DoubleWeightedMean of |new - old| over the convergence sample (ALS:230-238) */
  double seconds;
  int64_t x_rows, y_rows;        /* rows solved by the two half-iterations */
  int64_t entries_gathered;
  double algorithmic_bytes;
  int32_t devices;               /* GPUs of this process taking part */
  int32_t reserved;
} mals_iteration_info;
typedef void (*mals_iteration_fn)(void* user, const mals_iteration_info* info);
int mals_set_iteration_callback(mals_handle h, mals_iteration_fn fn, void* user);
int mals_group_set_iteration_callback(mals_group g, mals_iteration_fn fn, void* user);

int mals_enable_timing(mals_handle h, int32_t on);
int mals_reset_stats(mals_handle h);
int mals_get_stats(mals_handle h, mals_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* MYRRIX_ALS_H */
