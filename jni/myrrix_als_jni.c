/*
 * myrrix_als_jni.c -- JNI shim between net.myrrix.online.factorizer.als.HipAlternatingLeastSquares
 * (java/net/myrrix/online/factorizer/als/HipAlternatingLeastSquares.java) and the C-ABI of
 * include/myrrix_als.h.  One native method per mals_group_* entry point the adapter uses; a group of one
 * device is the single-GPU case.  The reference has no FFI on this path (it is pure Java): this is the
 * binding a maintainer adds, together with the one-line change at DelegateGenerationManager.java:406.
 *
 * NOT BUILT IN THIS REPOSITORY'S IMAGE (no JDK: jni.h is absent).  Complete source, reviewed by eye; the
 * exact call sequence it issues is replayed against the library by tests/cpp/test_jni_call_sequence.cpp.
 * Build where a JDK exists:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       -o libmyrrix_als_jni.so jni/myrrix_als_jni.c -Lmyrrix-recommender_amd/csrc -lmyrrix_als
 *
 * Array handling: every array is pinned or copied only for the duration of one native call
 * (Get<Type>ArrayElements, released with JNI_ABORT when only read) -- pieces are at most 2^26 entries, so no
 * call holds the collector for long, and no GetPrimitiveArrayCritical is used (the serving threads of the
 * same JVM must not stall on a factorization).  No C++ exception can cross: the
 * library reports status codes only.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>

#include "myrrix_als.h"

#define JNI_FN(name) Java_net_myrrix_online_factorizer_als_HipAlternatingLeastSquares_##name

static mals_group as_group(jlong handle) { return (mals_group)(intptr_t)handle; }

/* Why the last nativeCreate of THIS thread returned 0 (mals_group_create_error: "no HIP device ...", "device ordinal 3
 * outside 0..0", "librccl.so.1: cannot open shared object file", "ncclCommInitAll: ..."); "" after a success */
static __thread char t_shim_error[256];   /* failures of the shim itself, before the library was asked */
JNIEXPORT jstring JNICALL JNI_FN(nativeCreateError)(JNIEnv* env, jclass cls) {
  (void)cls;
  char why[1024];
  why[0] = 0;
  if (t_shim_error[0]) return (*env)->NewStringUTF(env, t_shim_error);
  (void)mals_group_create_error(why, sizeof(why));
  return (*env)->NewStringUTF(env, why);
}
static jlong shim_fail(const char* text) {
  size_t i = 0;
  for (; text[i] && i + 1 < sizeof(t_shim_error); ++i) t_shim_error[i] = text[i];
  t_shim_error[i] = 0;
  return 0;
}

/* mals_group_create: devices of one node, RCCL or peer copies; returns 0 on failure */
JNIEXPORT jlong JNICALL JNI_FN(nativeCreate)(JNIEnv* env, jclass cls, jint features, jdouble alpha, jdouble lambda,
                                             jdouble singularity_threshold, jint flags, jintArray devices, jboolean peer_copy) {
  (void)cls;
  mals_config cfg;
  t_shim_error[0] = 0;
  if (mals_default_config(&cfg) != MALS_OK) return shim_fail("mals_default_config failed");
  cfg.features = features;
  cfg.alpha = alpha;
  cfg.lambda = lambda;
  cfg.singularity_threshold = singularity_threshold;
  cfg.flags = flags;
  const jsize n = (*env)->GetArrayLength(env, devices);
  if (n <= 0) return shim_fail("model.als.gpus: empty device list");
  jint* dev = (*env)->GetIntArrayElements(env, devices, NULL);
  if (!dev) return shim_fail("out of memory (device list)");
  int32_t* list = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  if (!list) {
    (*env)->ReleaseIntArrayElements(env, devices, dev, JNI_ABORT);
    return shim_fail("out of memory (device list)");
  }
  for (jsize i = 0; i < n; ++i) list[i] = dev[i];
  (*env)->ReleaseIntArrayElements(env, devices, dev, JNI_ABORT);
  cfg.device = list[0];
  mals_group g = NULL;
  const int rc = mals_group_create(&cfg, list, (int32_t)n, peer_copy ? MALS_GROUP_PEER_COPY : MALS_GROUP_RCCL, &g);
  free(list);
  return rc == MALS_OK ? (jlong)(intptr_t)g : 0;
}

JNIEXPORT void JNICALL JNI_FN(nativeDestroy)(JNIEnv* env, jclass cls, jlong group) {
  (void)env;
  (void)cls;
  if (group) (void)mals_group_destroy(as_group(group));
}

JNIEXPORT jstring JNICALL JNI_FN(nativeLastError)(JNIEnv* env, jclass cls, jlong group) {
  (void)cls;
  return (*env)->NewStringUTF(env, group ? mals_group_last_error(as_group(group)) : "null group");
}

/* mals_group_set_refine_limit: the conditioning estimate above which a row is solved again with fp64 residuals */
JNIEXPORT jint JNICALL JNI_FN(nativeSetRefineLimit)(JNIEnv* env, jclass cls, jlong group, jdouble limit) {
  (void)env;
  (void)cls;
  return mals_group_set_refine_limit(as_group(group), limit);
}

JNIEXPORT jint JNICALL JNI_FN(nativeSetFactorRows)(JNIEnv* env, jclass cls, jlong group, jint side, jlong n_rows_total) {
  (void)env;
  (void)cls;
  return mals_group_set_factor_rows(as_group(group), side, n_rows_total);
}

/* mals_group_begin_matrix: the full row_ptr (n_rows + 1 longs) */
JNIEXPORT jint JNICALL JNI_FN(nativeBeginMatrix)(JNIEnv* env, jclass cls, jlong group, jint side, jlong n_rows, jlongArray row_ptr) {
  (void)cls;
  if ((*env)->GetArrayLength(env, row_ptr) < n_rows + 1) return MALS_INVALID_ARG;
  jlong* rp = (*env)->GetLongArrayElements(env, row_ptr, NULL);
  if (!rp) return MALS_OOM;
  /* jlong and int64_t are the same 64-bit type on every platform the library supports */
  const int rc = mals_group_begin_matrix(as_group(group), side, n_rows, (const int64_t*)rp);
  (*env)->ReleaseLongArrayElements(env, row_ptr, rp, JNI_ABORT);
  return rc;
}

/* mals_group_append_rows: the entries of the next n_rows rows (n_entries of them are valid) */
JNIEXPORT jint JNICALL JNI_FN(nativeAppendRows)(JNIEnv* env, jclass cls, jlong group, jint side, jlong n_rows, jintArray col_idx,
                                                jfloatArray val, jint n_entries) {
  (void)cls;
  if ((*env)->GetArrayLength(env, col_idx) < n_entries || (*env)->GetArrayLength(env, val) < n_entries) return MALS_INVALID_ARG;
  /* the library reads row_ptr[b] - row_ptr[a] entries whatever the caller believes: the two must agree BEFORE it does */
  int64_t expected = -1;
  const int prc = mals_group_pending_entries(as_group(group), side, n_rows, &expected);
  if (prc != MALS_OK) return prc;
  if (expected != (int64_t)n_entries) return MALS_INVALID_ARG;
  jint* c = (*env)->GetIntArrayElements(env, col_idx, NULL);
  if (!c) return MALS_OOM;
  jfloat* v = (*env)->GetFloatArrayElements(env, val, NULL);
  if (!v) {
    (*env)->ReleaseIntArrayElements(env, col_idx, c, JNI_ABORT);
    return MALS_OOM;
  }
  const int rc = mals_group_append_rows(as_group(group), side, n_rows, (const int32_t*)c, (const float*)v);
  (*env)->ReleaseFloatArrayElements(env, val, v, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, col_idx, c, JNI_ABORT);
  return rc;
}

JNIEXPORT jint JNICALL JNI_FN(nativeEndMatrix)(JNIEnv* env, jclass cls, jlong group, jint side) {
  (void)env;
  (void)cls;
  return mals_group_end_matrix(as_group(group), side);
}

/* setPreviousY / initial Y: rows [row_begin, row_begin + n_rows) of a side into every replica */
JNIEXPORT jint JNICALL JNI_FN(nativeSetFactors)(JNIEnv* env, jclass cls, jlong group, jint side, jlong row_begin, jint n_rows,
                                                jfloatArray rows) {
  (void)cls;
  if (n_rows < 0 || (int64_t)(*env)->GetArrayLength(env, rows) < (int64_t)n_rows * mals_group_features(as_group(group))) return MALS_INVALID_ARG;
  jfloat* r = (*env)->GetFloatArrayElements(env, rows, NULL);
  if (!r) return MALS_OOM;
  const int rc = mals_group_set_factors(as_group(group), side, row_begin, n_rows, (const float*)r);
  (*env)->ReleaseFloatArrayElements(env, rows, r, JNI_ABORT);
  return rc;
}

/* getX() / getY() */
JNIEXPORT jint JNICALL JNI_FN(nativeGetFactors)(JNIEnv* env, jclass cls, jlong group, jint side, jlong row_begin, jint n_rows,
                                                jfloatArray out) {
  (void)cls;
  if (n_rows < 0 || (int64_t)(*env)->GetArrayLength(env, out) < (int64_t)n_rows * mals_group_features(as_group(group))) return MALS_INVALID_ARG;
  jfloat* r = (*env)->GetFloatArrayElements(env, out, NULL);
  if (!r) return MALS_OOM;
  const int rc = mals_group_get_factors(as_group(group), side, row_begin, n_rows, (float*)r);
  (*env)->ReleaseFloatArrayElements(env, out, r, rc == MALS_OK ? 0 : JNI_ABORT);  /* 0: copy back and free */
  return rc;
}

/* per-iteration values for the adapter's log lines (ALS:241-246, 351-358): mals_group_set_iteration_callback -> the
 * listener's iteration(...) method, called on the thread that runs nativeFactorize (its JNIEnv is valid for the call) */
typedef struct {
  JNIEnv* env;
  jobject listener;
  jmethodID method;
} iteration_ctx;
static void on_iteration(void* user, const mals_iteration_info* info) {
  iteration_ctx* c = (iteration_ctx*)user;
  if (!c->listener || !c->method) return;
  (*c->env)->CallVoidMethod(c->env, c->listener, c->method, (jint)info->iteration, (jdouble)info->avg_abs_difference, (jdouble)info->seconds,
                            (jlong)info->x_rows, (jlong)info->y_rows, (jlong)info->entries_gathered, (jdouble)info->algorithmic_bytes,
                            (jint)info->devices);
  if ((*c->env)->ExceptionCheck(c->env)) (*c->env)->ExceptionClear(c->env);   /* a logging failure must not fail the build */
}

/* call() (ALS:176-262) */
JNIEXPORT jint JNICALL JNI_FN(nativeFactorize)(JNIEnv* env, jclass cls, jlong group, jdouble threshold, jint max_iterations,
                                               jboolean random_y, jboolean iterate, jlongArray test_users, jlongArray test_items,
                                               jintArray iterations_out, jdoubleArray convergence_out, jobject listener) {
  (void)cls;
  const jsize nu = (*env)->GetArrayLength(env, test_users), ni = (*env)->GetArrayLength(env, test_items);
  jlong* tu = (*env)->GetLongArrayElements(env, test_users, NULL);
  jlong* ti = (*env)->GetLongArrayElements(env, test_items, NULL);
  int rc = MALS_OOM;
  if (tu && ti) {
    int32_t iterations = 0;
    double convergence = 0.0;
    iteration_ctx ctx = {env, listener, NULL};
    if (listener) {
      jclass lc = (*env)->GetObjectClass(env, listener);
      ctx.method = (*env)->GetMethodID(env, lc, "iteration", "(IDDJJJDI)V");
      if (!ctx.method) (*env)->ExceptionClear(env);
    }
    (void)mals_group_set_iteration_callback(as_group(group), ctx.method ? on_iteration : NULL, &ctx);
    rc = mals_group_factorize(as_group(group), threshold, max_iterations, random_y ? 1 : 0, iterate ? 1 : 0, (const int64_t*)tu,
                              (int32_t)nu, (const int64_t*)ti, (int32_t)ni, &iterations, &convergence);
    (void)mals_group_set_iteration_callback(as_group(group), NULL, NULL);
    const jint it = iterations;
    const jdouble cv = convergence;
    (*env)->SetIntArrayRegion(env, iterations_out, 0, 1, &it);
    (*env)->SetDoubleArrayRegion(env, convergence_out, 0, 1, &cv);
  }
  if (ti) (*env)->ReleaseLongArrayElements(env, test_items, ti, JNI_ABORT);
  if (tu) (*env)->ReleaseLongArrayElements(env, test_users, tu, JNI_ABORT);
  return rc;
}

/* Thread.interrupt() of the calling thread -> cooperative cancellation between half-iterations */
JNIEXPORT jint JNICALL JNI_FN(nativeCancel)(JNIEnv* env, jclass cls, jlong group) {
  (void)env;
  (void)cls;
  return mals_group_cancel(as_group(group));
}

/* {side, row, apparent rank} of the last MALS_SINGULAR (SingularMatrixSolverException.getApparentRank) */
JNIEXPORT jint JNICALL JNI_FN(nativeSingularInfo)(JNIEnv* env, jclass cls, jlong group, jlongArray out) {
  (void)cls;
  int32_t side = -1, rank = 0;
  int64_t row = -1;
  const int rc = mals_group_singular_info(as_group(group), &side, &row, &rank);
  const jlong v[3] = {side, row, rank};
  (*env)->SetLongArrayRegion(env, out, 0, 3, v);
  return rc;
}
