/*
 * net.myrrix.common.math.Solver (common/src/net/myrrix/common/math/Solver.java:26-42) over one mals_solver of
 * libmyrrix_als.so: the fp64 pivoted-QR factorization of a k x k matrix A, solve(b) per call.  What
 * CommonsMathSolver (CommonsMathSolver.java:28-59) is for commons-math's DecompositionSolver.
 *
 * The factorization is immutable after creation: solveDToF / solveFToD may be called from any number of serving
 * threads at once (ServerRecommender's fold-in does, online/src/net/myrrix/online/ServerRecommender.java:561-606).
 *
 * NOT BUILT IN THIS REPOSITORY'S IMAGE (no JDK).  JNI functions: jni/myrrix_solver_jni.c.
 */
package net.myrrix.common.math;

import java.util.concurrent.locks.ReentrantReadWriteLock;

public final class NativeSolver implements Solver, AutoCloseable {

  static {
    System.loadLibrary("myrrix_als_jni");   // links libmyrrix_als.so
  }

  private long handle;
  private final int n;
  /**
   * Solves share the lock, close() takes it exclusively: the native factorization cannot be freed while a serving
   * thread is inside mals_solver_solve_*.  That covers an explicit close() from another thread AND the finalizer: once
   * handle() has returned, `this` is no longer referenced by the solving thread, so the JIT may treat it as unreachable
   * during the native call and the collector may run finalize() -- which then blocks in close() until the solve has
   * released its read lock (round 3 read the handle under a monitor and called the native solve outside it).
   */
  private final ReentrantReadWriteLock lifecycle = new ReentrantReadWriteLock();

  NativeSolver(long handle, int n) {
    this.handle = handle;
    this.n = n;
  }

  /**
   * Generation.recomputeSolver (Generation.java:142-158) in one call for a factor matrix that is already resident in
   * a factorizer's HBM: M^T M on the device (the iteration's own Gramian kernel), getNorm(), the < 1.0 test, getSolver.
   *
   * @param group handle of HipAlternatingLeastSquares' native group (member 0's replica is used)
   * @param side 0 = X, 1 = Y
   * @return null when the side has no rows (Generation.java:145)
   * @throws IllConditionedSolverException infNorm < 1.0 (Generation.java:150-153)
   * @throws SingularMatrixSolverException with the apparent rank (CommonsMathLinearSystemSolver.java:47-54)
   */
  public static NativeSolver recompute(long group, int side, int features) {
    double[] infNorm = new double[1];
    int[] status = new int[2];            // {status, apparent rank}
    long h = nativeRecompute(group, side, infNorm, status);
    switch (status[0]) {
      case 0:
        return h == 0L ? null : new NativeSolver(h, features);
      case 7:                             // MALS_ILL_CONDITIONED
        throw new IllConditionedSolverException("infNorm: " + infNorm[0]);
      case 1:                             // MALS_SINGULAR
        throw new SingularMatrixSolverException(status[1], "Apparent rank: " + status[1]);
      default:
        throw new IllegalStateException("mals_recompute_solver failed with status " + status[0]);
    }
  }

  /** Solver.java:35: x = A^-1 b in fp64, each entry cast to float (CommonsMathSolver.java:37-44). */
  @Override
  public float[] solveDToF(double[] b) {
    checkLength(b.length);
    float[] x = new float[n];
    lifecycle.readLock().lock();
    try {
      check(nativeSolveDToF(handle(), b, x));
    } finally {
      lifecycle.readLock().unlock();
    }
    return x;
  }

  /** Solver.java:41: float input widened, fp64 result (CommonsMathSolver.java:46-58). */
  @Override
  public double[] solveFToD(float[] b) {
    checkLength(b.length);
    double[] x = new double[n];
    lifecycle.readLock().lock();
    try {
      check(nativeSolveFToD(handle(), b, x));
    } finally {
      lifecycle.readLock().unlock();
    }
    return x;
  }

  private void checkLength(int length) {
    if (length != n) {
      // commons-math: DimensionMismatchException extends IllegalArgumentException
      throw new IllegalArgumentException("vector of length " + length + " for a " + n + " x " + n + " system");
    }
  }

  /** Call with the read lock held. */
  private long handle() {
    if (handle == 0L) {
      throw new IllegalStateException("solver already closed");
    }
    return handle;
  }

  private static void check(int status) {
    if (status != 0) {
      throw new IllegalStateException("mals_solver_solve failed with status " + status);
    }
  }

  @Override
  public void close() {
    lifecycle.writeLock().lock();     // waits for every solve in flight
    try {
      if (handle != 0L) {
        nativeDestroy(handle);
        handle = 0L;
      }
    } finally {
      lifecycle.writeLock().unlock();
    }
  }

  @Override
  protected void finalize() {   // the reference targets Java 6/7 (pom.xml): no java.lang.ref.Cleaner there
    close();
  }

  /** mals_solver_create; 0 with apparentRankOut[0] >= 0: singular (the rank); 0 with -1: invalid input. */
  static native long nativeCreate(double[] rowMajor, int n, double singularityThreshold, int[] apparentRankOut);

  private static native long nativeRecompute(long group, int side, double[] infNormOut, int[] statusAndRankOut);

  private static native int nativeSolveDToF(long handle, double[] b, float[] xOut);

  private static native int nativeSolveFToD(long handle, float[] b, double[] xOut);

  private static native void nativeDestroy(long handle);

}
