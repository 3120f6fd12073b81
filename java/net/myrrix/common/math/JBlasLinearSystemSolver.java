/*
 * Drop-in for the reference's own native-math hook.
 *
 * MatrixUtils (common/src/net/myrrix/common/math/MatrixUtils.java:44-49) loads, by name and reflectively,
 *   net.myrrix.common.math.JBlasLinearSystemSolver     when -Dcommon.matrix.nativeMath=true
 *   net.myrrix.common.math.CommonsMathLinearSystemSolver otherwise
 * and the class of that name in the reference tree (JBlasLinearSystemSolver.java:35-80) is a stub whose getSolver()
 * throws UnsupportedOperationException.  This class has the SAME fully-qualified name and the same package-private
 * interface (LinearSystemSolver.java:27-48): put libmyrrix-hip.jar before myrrix-common on the classpath, start the
 * JVM with -Dcommon.matrix.nativeMath=true, and every MatrixUtils.getSolver(...) -- Generation.recomputeSolver's
 * XTX / YTY solvers (online/src/net/myrrix/online/generation/Generation.java:142-158) first of all -- goes through
 * libmyrrix_als.so (mals_solver_create: column-pivoted Householder QR in fp64, the reference's singularity rule and
 * apparent rank, csrc/host_solver.h) without one changed line in the reference.
 *
 * NOT BUILT IN THIS REPOSITORY'S IMAGE (no JDK).  Complete source; the call sequence it issues is replayed against
 * the library by tests/cpp/test_solver_spi_sequence.cpp.
 */
package net.myrrix.common.math;

import org.apache.commons.math3.linear.RealMatrix;

public final class JBlasLinearSystemSolver implements LinearSystemSolver {

  /** MatrixUtils.getSolver(M) (MatrixUtils.java:137-139) -> CommonsMathLinearSystemSolver.getSolver's contract
   *  (CommonsMathLinearSystemSolver.java:37-55): null for null, a Solver when every |R_ii| of the pivoted QR exceeds
   *  SINGULARITY_THRESHOLD, else SingularMatrixSolverException carrying getRank(0.01). */
  @Override
  public Solver getSolver(RealMatrix M) {
    if (M == null) {
      return null;
    }
    int n = M.getRowDimension();
    if (n != M.getColumnDimension()) {
      throw new IllegalArgumentException("square matrix expected: " + n + " x " + M.getColumnDimension());
    }
    double[] rowMajor = new double[n * n];
    for (int r = 0; r < n; r++) {
      // getRow copies one row; MatrixUtils' reflective grab of Array2DRowRealMatrix.data (MatrixUtils.java:171-177)
      // is private to it, and this runs once per generation / fold-in setup, not per row of R
      System.arraycopy(M.getRow(r), 0, rowMajor, r * n, n);
    }
    int[] apparentRank = new int[1];
    long handle = NativeSolver.nativeCreate(rowMajor, n, SINGULARITY_THRESHOLD, apparentRank);
    if (handle == 0L) {
      if (apparentRank[0] < 0) {
        throw new IllegalStateException("mals_solver_create failed for a " + n + " x " + n + " matrix (non-finite input?)");
      }
      // same message as CommonsMathLinearSystemSolver.java:54
      throw new SingularMatrixSolverException(apparentRank[0], "Apparent rank: " + apparentRank[0]);
    }
    return new NativeSolver(handle, n);
  }

  /** CommonsMathLinearSystemSolver.java:57-62. */
  @Override
  public boolean isNonSingular(RealMatrix M) {
    try {
      Solver s = getSolver(M);
      if (s instanceof NativeSolver) {
        ((NativeSolver) s).close();
      }
    } catch (SolverException ignored) {
      return false;
    }
    return true;
  }

}
