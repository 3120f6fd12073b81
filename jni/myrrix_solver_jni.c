/*
 * myrrix_solver_jni.c -- JNI functions of net.myrrix.common.math.NativeSolver
 * (java/net/myrrix/common/math/NativeSolver.java) over the mals_solver_* / mals_recompute_solver entry points of
 * include/myrrix_als.h: the Solver SPI of the reference (Solver.java:26-42, LinearSystemSolver.java:27-48) behind
 * its own reflective hook (MatrixUtils.java:44-49, -Dcommon.matrix.nativeMath=true; see
 * java/net/myrrix/common/math/JBlasLinearSystemSolver.java).  Compiled into the same libmyrrix_als_jni.so as
 * jni/myrrix_als_jni.c.
 *
 * NOT BUILT IN THIS REPOSITORY'S IMAGE (no JDK: jni.h is absent).  The exact call sequence is replayed against the
 * library by tests/cpp/test_solver_spi_sequence.cpp.  Every array is validated against the solver's dimension
 * before the library reads or writes it.
 */
#include <jni.h>
#include <stdint.h>

#include "myrrix_als.h"

#define JNI_FN(name) Java_net_myrrix_common_math_NativeSolver_##name

static mals_solver as_solver(jlong handle) { return (mals_solver)(intptr_t)handle; }

/* MatrixUtils.getSolver(A): mals_solver_create.  Returns 0 with rank_out[0] = apparent rank (singular) or -1 (bad input). */
JNIEXPORT jlong JNICALL JNI_FN(nativeCreate)(JNIEnv* env, jclass cls, jdoubleArray row_major, jint n, jdouble threshold,
                                             jintArray rank_out) {
  (void)cls;
  jint rank = -1;
  jlong result = 0;
  if (n > 0 && (int64_t)(*env)->GetArrayLength(env, row_major) >= (int64_t)n * n && (*env)->GetArrayLength(env, rank_out) >= 1) {
    jdouble* a = (*env)->GetDoubleArrayElements(env, row_major, NULL);
    if (a) {
      mals_solver s = NULL;
      int32_t apparent = 0;
      const int rc = mals_solver_create((const double*)a, n, threshold, &s, &apparent);
      (*env)->ReleaseDoubleArrayElements(env, row_major, a, JNI_ABORT);
      if (rc == MALS_OK) {
        result = (jlong)(intptr_t)s;
        rank = n;
      } else if (rc == MALS_SINGULAR) {
        rank = apparent;
      }
    }
  }
  if ((*env)->GetArrayLength(env, rank_out) >= 1) (*env)->SetIntArrayRegion(env, rank_out, 0, 1, &rank);
  return result;
}

/* Generation.recomputeSolver on the factors resident in the factorizer's group (member 0): mals_recompute_solver.
 * status_out = {status, apparent rank}; norm_out[0] = getNorm() of M^T M. */
JNIEXPORT jlong JNICALL JNI_FN(nativeRecompute)(JNIEnv* env, jclass cls, jlong group, jint side, jdoubleArray norm_out,
                                                jintArray status_out) {
  (void)cls;
  jint st[2] = {MALS_INVALID_ARG, 0};
  jdouble norm = 0.0;
  mals_solver s = NULL;
  mals_handle h = NULL;
  if (group && (*env)->GetArrayLength(env, norm_out) >= 1 && (*env)->GetArrayLength(env, status_out) >= 2 &&
      mals_group_local((mals_group)(intptr_t)group, 0, &h, NULL) == MALS_OK) {
    double n = 0.0;
    st[0] = mals_recompute_solver(h, side, &s, &n);
    norm = n;
    if (st[0] == MALS_SINGULAR) {
      int32_t sd = -1, rank = 0;
      int64_t row = -1;
      (void)mals_singular_info(h, &sd, &row, &rank);
      st[1] = rank;
    }
    if (st[0] != MALS_OK) s = NULL;
  }
  if ((*env)->GetArrayLength(env, norm_out) >= 1) (*env)->SetDoubleArrayRegion(env, norm_out, 0, 1, &norm);
  if ((*env)->GetArrayLength(env, status_out) >= 2) (*env)->SetIntArrayRegion(env, status_out, 0, 2, st);
  return (jlong)(intptr_t)s;
}

/* Solver.solveDToF (Solver.java:35) */
JNIEXPORT jint JNICALL JNI_FN(nativeSolveDToF)(JNIEnv* env, jclass cls, jlong handle, jdoubleArray b, jfloatArray x_out) {
  (void)cls;
  const int n = mals_solver_dim(as_solver(handle));
  if (n <= 0 || (*env)->GetArrayLength(env, b) != n || (*env)->GetArrayLength(env, x_out) != n) return MALS_INVALID_ARG;
  jdouble* bb = (*env)->GetDoubleArrayElements(env, b, NULL);
  if (!bb) return MALS_OOM;
  jfloat* xx = (*env)->GetFloatArrayElements(env, x_out, NULL);
  if (!xx) {
    (*env)->ReleaseDoubleArrayElements(env, b, bb, JNI_ABORT);
    return MALS_OOM;
  }
  const int rc = mals_solver_solve_dtof(as_solver(handle), (const double*)bb, (float*)xx);
  (*env)->ReleaseFloatArrayElements(env, x_out, xx, rc == MALS_OK ? 0 : JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, b, bb, JNI_ABORT);
  return rc;
}

/* Solver.solveFToD (Solver.java:41) */
JNIEXPORT jint JNICALL JNI_FN(nativeSolveFToD)(JNIEnv* env, jclass cls, jlong handle, jfloatArray b, jdoubleArray x_out) {
  (void)cls;
  const int n = mals_solver_dim(as_solver(handle));
  if (n <= 0 || (*env)->GetArrayLength(env, b) != n || (*env)->GetArrayLength(env, x_out) != n) return MALS_INVALID_ARG;
  jfloat* bb = (*env)->GetFloatArrayElements(env, b, NULL);
  if (!bb) return MALS_OOM;
  jdouble* xx = (*env)->GetDoubleArrayElements(env, x_out, NULL);
  if (!xx) {
    (*env)->ReleaseFloatArrayElements(env, b, bb, JNI_ABORT);
    return MALS_OOM;
  }
  const int rc = mals_solver_solve_ftod(as_solver(handle), (const float*)bb, (double*)xx);
  (*env)->ReleaseDoubleArrayElements(env, x_out, xx, rc == MALS_OK ? 0 : JNI_ABORT);
  (*env)->ReleaseFloatArrayElements(env, b, bb, JNI_ABORT);
  return rc;
}

JNIEXPORT void JNICALL JNI_FN(nativeDestroy)(JNIEnv* env, jclass cls, jlong handle) {
  (void)env;
  (void)cls;
  if (handle) (void)mals_solver_destroy(as_solver(handle));
}
