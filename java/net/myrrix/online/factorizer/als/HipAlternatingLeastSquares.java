/*
 * HipAlternatingLeastSquares -- the Java side of the MI355X-native ALS core: a MatrixFactorizer with the
 * constructor of net.myrrix.online.factorizer.als.AlternatingLeastSquares
 * (online/src/net/myrrix/online/factorizer/als/AlternatingLeastSquares.java:132-147) that keeps the
 * reference's data structures at the boundary (FastByIDMap<FastByIDFloatMap> in, FastByIDMap<float[]> out)
 * and runs call() (ALS:176-262) on one or more GPUs through libmyrrix_als.so (include/myrrix_als.h, via
 * jni/myrrix_als_jni.c).
 *
 * NOT BUILT IN THIS REPOSITORY'S IMAGE: there is no JDK here (java, javac, jni.h absent).  The file is
 * complete source, written against the reference's public classes, and reviewed by eye only.  Build with
 * the reference's own classpath (myrrix-common, myrrix-online, commons-math3 3.2, guava 14); the single
 * production call site to switch is DelegateGenerationManager.java:406 (`new AlternatingLeastSquares(...)`),
 * see INTEGRATION.md.
 *
 * What stays in Java (SURVEY.md section 8(a) rows a4, a14, a15): the initial Y (previous generation's
 * vectors, truncated / padded, new items from RandomUtils.randomUnitVectorFarFrom with the JVM's own
 * generator -- so a cold start draws exactly the reference's random stream), the choice of the ~100 x 100
 * convergence sample (RandomUtils.chooseAboutNFromStream), the id <-> dense index maps.  Everything else
 * -- Gramians, the per-row systems, the convergence statistic -- is native.
 *
 * System properties (identical names and defaults to the reference): model.als.alpha (1.0),
 * model.als.lambda (0.1), model.reconstructRMatrix, model.lossIgnoresUnspecified, model.als.iterate,
 * common.matrix.singularityThreshold (1e-5); new: model.als.gpus (comma-separated HIP device ordinals,
 * default "0"), model.als.gpu.peerCopy (true: slices move by peer copies instead of RCCL).
 */
package net.myrrix.online.factorizer.als;

import java.util.ArrayList;
import java.util.List;
import java.util.concurrent.Callable;
import java.util.concurrent.ExecutionException;
import java.util.concurrent.FutureTask;

import org.apache.commons.math3.random.RandomGenerator;
import org.slf4j.Logger;
import org.slf4j.LoggerFactory;

import net.myrrix.common.collection.FastByIDFloatMap;
import net.myrrix.common.collection.FastByIDMap;
import net.myrrix.common.math.SimpleVectorMath;
import net.myrrix.common.math.SingularMatrixSolverException;
import net.myrrix.common.random.RandomManager;
import net.myrrix.common.random.RandomUtils;
import net.myrrix.online.factorizer.MatrixFactorizer;

import org.apache.mahout.cf.taste.impl.common.LongPrimitiveIterator;

public final class HipAlternatingLeastSquares implements MatrixFactorizer {

  /** The reference logs under its own class name (ALS:68); an operator's logback / log4j filters keep working when the
   *  logger NAME is the reference's.  -Dmodel.als.gpu.logAsReference=false logs under this class instead. */
  private static final Logger log = LoggerFactory.getLogger(
      Boolean.parseBoolean(System.getProperty("model.als.gpu.logAsReference", "true"))
          ? "net.myrrix.online.factorizer.als.AlternatingLeastSquares"
          : HipAlternatingLeastSquares.class.getName());

  /** Called from native code (jni/myrrix_als_jni.c, on_iteration) after every iteration of nativeFactorize. */
  interface IterationListener {
    void iteration(int iteration, double avgAbsDifference, double seconds, long xRows, long yRows, long entriesGathered,
                   double algorithmicBytes, int devices);
  }

  // status codes of include/myrrix_als.h
  private static final int MALS_OK = 0;
  private static final int MALS_SINGULAR = 1;
  private static final int MALS_CANCELLED = 5;
  private static final int SIDE_X = 0;
  private static final int SIDE_Y = 1;
  private static final int FLAG_RECONSTRUCT_R = 1;
  private static final int FLAG_LOSS_IGNORES_UNSPECIFIED = 2;

  /** Same constants as the reference (ALS:71-76, 78-81). */
  public static final double DEFAULT_ALPHA = 1.0;
  public static final double DEFAULT_LAMBDA = 0.1;
  public static final double DEFAULT_CONVERGENCE_THRESHOLD = 0.001;
  public static final int DEFAULT_MAX_ITERATIONS = 30;
  private static final int NUM_TO_TEST_CONVERGENCE = 100;
  private static final int MAX_FAR_FROM_VECTORS = 100000;

  /** Entries handed to the native side per call: Java arrays stay far below 2^31 elements. */
  private static final int ENTRIES_PER_PIECE = 1 << 26;
  private static final int ROWS_PER_FACTOR_PIECE = 1 << 18;

  static {
    System.loadLibrary("myrrix_als_jni");   // which loads libmyrrix_als.so
  }

  // ---- native methods: one per mals_group_* entry point used (jni/myrrix_als_jni.c) ----------------
  private static native long nativeCreate(int features, double alpha, double lambda, double singularityThreshold,
                                          int flags, int[] devices, boolean peerCopy);
  /** Why the last nativeCreate on this thread returned 0 (mals_group_create_error). */
  private static native String nativeCreateError();
  private static native void nativeDestroy(long group);
  private static native String nativeLastError(long group);
  private static native int nativeSetRefineLimit(long group, double limit);
  private static native int nativeSetFactorRows(long group, int side, long nRowsTotal);
  private static native int nativeBeginMatrix(long group, int side, long nRows, long[] rowPtr);
  private static native int nativeAppendRows(long group, int side, long nRows, int[] colIdx, float[] val, int nEntries);
  private static native int nativeEndMatrix(long group, int side);
  private static native int nativeSetFactors(long group, int side, long rowBegin, int nRows, float[] rows);
  private static native int nativeGetFactors(long group, int side, long rowBegin, int nRows, float[] out);
  private static native int nativeFactorize(long group, double threshold, int maxIterations, boolean randomY,
                                            boolean iterate, long[] testUsers, long[] testItems,
                                            int[] iterationsOut, double[] convergenceOut, IterationListener listener);
  private static native int nativeCancel(long group);
  /** out = {side, row, apparentRank} of the last MALS_SINGULAR. */
  private static native int nativeSingularInfo(long group, long[] out);

  private final FastByIDMap<FastByIDFloatMap> RbyRow;
  private final FastByIDMap<FastByIDFloatMap> RbyColumn;
  private final int features;
  private final double estimateErrorConvergenceThreshold;
  private final int maxIterations;
  private FastByIDMap<float[]> X;
  private FastByIDMap<float[]> Y;
  private FastByIDMap<float[]> previousY;
  private int iterationsRun;
  private double lastConvergenceValue = Double.NaN;

  public HipAlternatingLeastSquares(FastByIDMap<FastByIDFloatMap> RbyRow,
                                    FastByIDMap<FastByIDFloatMap> RbyColumn,
                                    int features) {
    this(RbyRow, RbyColumn, features, DEFAULT_CONVERGENCE_THRESHOLD, DEFAULT_MAX_ITERATIONS);
  }

  /** Arguments and checks as ALS:132-147. */
  public HipAlternatingLeastSquares(FastByIDMap<FastByIDFloatMap> RbyRow,
                                    FastByIDMap<FastByIDFloatMap> RbyColumn,
                                    int features,
                                    double estimateErrorConvergenceThreshold,
                                    int maxIterations) {
    if (RbyRow == null || RbyColumn == null) {
      throw new NullPointerException();
    }
    if (features <= 0) {
      throw new IllegalArgumentException("features must be positive: " + features);
    }
    if (!(estimateErrorConvergenceThreshold > 0.0 && estimateErrorConvergenceThreshold < 1.0)) {
      throw new IllegalArgumentException("threshold must be in (0,1): " + estimateErrorConvergenceThreshold);
    }
    this.RbyRow = RbyRow;
    this.RbyColumn = RbyColumn;
    this.features = features;
    this.estimateErrorConvergenceThreshold = estimateErrorConvergenceThreshold;
    this.maxIterations = maxIterations;
  }

  @Override
  public FastByIDMap<float[]> getX() {
    return X;
  }

  @Override
  public FastByIDMap<float[]> getY() {
    return Y;
  }

  /** Ignored, like the reference (ALS:162-165). */
  @Override
  public void setPreviousX(FastByIDMap<float[]> previousX) {
    // nothing to do
  }

  @Override
  public void setPreviousY(FastByIDMap<float[]> previousY) {
    this.previousY = previousY;
  }

  public int getIterationsRun() {
    return iterationsRun;
  }

  public double getLastConvergenceValue() {
    return lastConvergenceValue;
  }

  // ---------------------------------------------------------------------------------------------------
  @Override
  public Void call() throws ExecutionException, InterruptedException {
    X = new FastByIDMap<float[]>(RbyRow.size());
    boolean randomY = previousY == null || previousY.isEmpty();
    FastByIDMap<float[]> initialY = buildInitialY(previousY);

    // dense indices: users in the iteration order of RbyRow; items of RbyColumn first (they own matrix
    // rows), then the rows of Y no current item owns (kept, never re-solved, counted in Y^T Y: ALS:304-308)
    long[] userIDs = keysOf(RbyRow.keySetIterator(), RbyRow.size());
    LongIndex userIndex = new LongIndex(userIDs);
    int numItems = RbyColumn.size();
    long[] itemIDs = new long[initialY.size()];
    int filled = 0;
    LongPrimitiveIterator itemIt = RbyColumn.keySetIterator();
    while (itemIt.hasNext()) {
      itemIDs[filled++] = itemIt.nextLong();
    }
    for (FastByIDMap.MapEntry<float[]> entry : initialY.entrySet()) {
      if (!RbyColumn.containsKey(entry.getKey())) {
        itemIDs[filled++] = entry.getKey();
      }
    }
    LongIndex itemIndex = new LongIndex(itemIDs);

    boolean iterate = Boolean.parseBoolean(System.getProperty("model.als.iterate", "true"));
    long[] testUsers = new long[0];
    long[] testItems = new long[0];
    if (iterate) {
      // the same two draws, from the same generator, as ALS:206-215
      RandomGenerator random = RandomManager.getRandom();
      long[] testUserIDs = RandomUtils.chooseAboutNFromStream(NUM_TO_TEST_CONVERGENCE, RbyRow.keySetIterator(),
                                                              RbyRow.size(), random);
      long[] testItemIDs = RandomUtils.chooseAboutNFromStream(NUM_TO_TEST_CONVERGENCE, RbyColumn.keySetIterator(),
                                                              RbyColumn.size(), random);
      testUsers = toIndices(testUserIDs, userIndex);
      testItems = toIndices(testItemIDs, itemIndex);
    }

    int flags = 0;
    if (Boolean.parseBoolean(System.getProperty("model.reconstructRMatrix", "false"))) {
      flags |= FLAG_RECONSTRUCT_R;
    }
    if (Boolean.parseBoolean(System.getProperty("model.lossIgnoresUnspecified", "false"))) {
      flags |= FLAG_LOSS_IGNORES_UNSPECIFIED;
    }
    final int[] devices = parseDevices(System.getProperty("model.als.gpus", "0"));
    // ALS:193 logs "Iterating using {} threads"; the workers here are GPUs
    log.info("Iterating using {} GPU(s) {} ({} features, {} users, {} items)", devices.length, java.util.Arrays.toString(devices),
             features, userIDs.length, numItems);
    final long group = nativeCreate(features,
                                    doubleProperty("model.als.alpha", DEFAULT_ALPHA),
                                    doubleProperty("model.als.lambda", DEFAULT_LAMBDA),
                                    doubleProperty("common.matrix.singularityThreshold", 1.0e-5),
                                    flags,
                                    devices,
                                    Boolean.parseBoolean(System.getProperty("model.als.gpu.peerCopy", "false")));
    if (group == 0L) {
      // "no HIP device", "device ordinal 3 outside 0..0", "librccl.so.1: cannot open ...", "ncclCommInitAll: ..." --
      // the library's own words (there is no CPU fallback to fall back to)
      String why = nativeCreateError();
      throw new ExecutionException(new IllegalStateException(
          "mals_group_create failed: " + (why == null || why.isEmpty() ? "unknown reason" : why)));
    }
    try {
      // rows whose system is too ill-conditioned for fp32 are solved again with fp64 residuals: the conditioning
      // estimate above which that happens (include/myrrix_als.h, mals_set_refine_limit); unset = the library's default
      String refineLimit = System.getProperty("model.als.gpu.refineLimit");
      if (refineLimit != null) {
        check(group, nativeSetRefineLimit(group, Double.parseDouble(refineLimit)));
      }
      check(group, nativeSetFactorRows(group, SIDE_X, Math.max(1, userIDs.length)));
      check(group, nativeSetFactorRows(group, SIDE_Y, Math.max(1, itemIDs.length)));
      streamRows(group, SIDE_X, RbyRow, userIDs, userIDs.length, itemIndex);
      streamRows(group, SIDE_Y, RbyColumn, itemIDs, numItems, userIndex);
      uploadFactors(group, SIDE_Y, itemIDs, initialY);

      final long[] tu = testUsers;
      final long[] ti = testItems;
      final boolean iter = iterate;
      final boolean rnd = randomY;
      final int[] iterationsOut = new int[1];
      final double[] convergenceOut = new double[1];
      final int maxIt = maxIterations;
      final double threshold = estimateErrorConvergenceThreshold;
      // the reference's per-iteration lines (ALS:241-256) and its "rows computed" progress (ALS:351-358, 378-385: there
      // every 10000 work units with the JVM heap; here once per half-iteration pair, with what a GPU operator watches:
      // rows/s and the algorithmic GB/s of SURVEY.md 8(d))
      final IterationListener listener = new IterationListener() {
        @Override
        public void iteration(int iteration, double avgAbsDifference, double seconds, long xRows, long yRows,
                              long entriesGathered, double algorithmicBytes, int nDevices) {
          log.info("{} X/tag rows computed, {} Y/tag rows computed ({} entries gathered in {} s on {} GPU(s): {} rows/s, {} GB/s)",
                   xRows, yRows, entriesGathered, String.format("%.3f", seconds), nDevices,
                   String.format("%.3g", (xRows + yRows) / Math.max(seconds, 1.0e-9)),
                   String.format("%.1f", algorithmicBytes / Math.max(seconds, 1.0e-9) / 1.0e9));
          log.info("Finished iteration {}", iteration);
          if (maxIt > 0 && iteration >= maxIt) {
            log.info("Reached iteration limit");
            return;
          }
          log.info("Avg absolute difference in estimate vs prior iteration: {}", avgAbsDifference);
          if (Double.isNaN(avgAbsDifference) || Double.isInfinite(avgAbsDifference)) {
            log.warn("Invalid convergence value, aborting iteration! {}", avgAbsDifference);
          } else if (!(rnd && iteration == 1) && avgAbsDifference < threshold) {
            log.info("Converged");
          }
        }
      };
      // the factorization runs on its own thread so that THIS thread stays interruptible: an interrupt
      // becomes mals_group_cancel, honoured between half-iterations (MatrixFactorizer.java:43-44)
      FutureTask<Integer> task = new FutureTask<Integer>(new Callable<Integer>() {
        @Override
        public Integer call() {
          return nativeFactorize(group, estimateErrorConvergenceThreshold, maxIterations, rnd, iter, tu, ti,
                                 iterationsOut, convergenceOut, listener);
        }
      });
      Thread worker = new Thread(task, "HipALS");
      worker.setDaemon(true);
      worker.start();
      int status;
      try {
        status = task.get();
      } catch (InterruptedException ie) {
        nativeCancel(group);
        worker.join();
        throw ie;
      }
      iterationsRun = iterationsOut[0];
      lastConvergenceValue = convergenceOut[0];
      check(group, status);

      X = downloadFactors(group, SIDE_X, userIDs, userIDs.length);
      Y = downloadFactors(group, SIDE_Y, itemIDs, itemIDs.length);
    } finally {
      nativeDestroy(group);
    }
    return null;
  }

  // ---- initial Y: the semantics of ALS:264-335, calling the reference's own random utilities ----------
  private FastByIDMap<float[]> buildInitialY(FastByIDMap<float[]> previous) {
    RandomGenerator random = RandomManager.getRandom();
    FastByIDMap<float[]> start;
    if (previous == null || previous.isEmpty()) {
      log.info("Starting from new, random Y matrix");                                   // ALS:271
      start = new FastByIDMap<float[]>(RbyColumn.size());
    } else {
      int oldFeatures = previous.entrySet().iterator().next().getValue().length;
      if (oldFeatures == features) {
        log.info("Starting from previous generation's Y matrix");                       // ALS:306
        start = previous;                      // reused in place; the caller passed a clone (DGM:419-422)
      } else {
        if (oldFeatures > features) {                                                   // ALS:279, 290
          log.info("Feature count has decreased to {}, projecting down previous generation's Y matrix", features);
        } else {
          log.info("Feature count has increased to {}, using previous generation's Y matrix as subspace", features);
        }
        start = new FastByIDMap<float[]>(previous.size());
        for (FastByIDMap.MapEntry<float[]> entry : previous.entrySet()) {
          float[] old = entry.getValue();
          float[] resized = new float[features];
          System.arraycopy(old, 0, resized, 0, Math.min(old.length, features));
          for (int i = old.length; i < features; i++) {   // more features now: random in the new dimensions
            resized[i] = (float) random.nextGaussian();
          }
          SimpleVectorMath.normalize(resized);
          start.put(entry.getKey(), resized);
        }
      }
    }
    List<float[]> farFrom = new ArrayList<float[]>();
    for (FastByIDMap.MapEntry<float[]> entry : start.entrySet()) {
      if (farFrom.size() >= MAX_FAR_FROM_VECTORS) {
        break;
      }
      farFrom.add(entry.getValue());
    }
    LongPrimitiveIterator it = RbyColumn.keySetIterator();
    int fresher = 0;
    while (it.hasNext()) {
      long id = it.nextLong();
      if (!start.containsKey(id)) {
        float[] fresh = RandomUtils.randomUnitVectorFarFrom(features, farFrom, random);
        start.put(id, fresh);
        if (farFrom.size() < MAX_FAR_FROM_VECTORS) {
          farFrom.add(fresh);
        }
        if (++fresher % 10000 == 0) {
          log.info("Computed {} initial Y rows", fresher);                              // ALS:330
        }
      }
    }
    log.info("Constructed initial Y");                                                  // ALS:333
    return start;
  }

  // ---- R (or R^T) as CSR over dense indices, in pieces of whole rows ----------------------------------
  private static void streamRows(long group, int side, FastByIDMap<FastByIDFloatMap> matrix, long[] rowIDs,
                                 int numRows, LongIndex columnIndex) throws ExecutionException {
    long[] rowPtr = new long[numRows + 1];
    for (int r = 0; r < numRows; r++) {
      FastByIDFloatMap row = matrix.get(rowIDs[r]);
      rowPtr[r + 1] = rowPtr[r] + (row == null ? 0 : row.size());
    }
    check(group, nativeBeginMatrix(group, side, numRows, rowPtr));
    int begin = 0;
    while (begin < numRows) {
      int end = begin;
      long entries = 0;
      while (end < numRows && (end == begin || entries + (rowPtr[end + 1] - rowPtr[end]) <= ENTRIES_PER_PIECE)) {
        entries += rowPtr[end + 1] - rowPtr[end];
        end++;
      }
      if (entries > Integer.MAX_VALUE - 8) {
        throw new ExecutionException(new IllegalStateException("a single row holds more than 2^31 entries"));
      }
      int[] colIdx = new int[(int) entries];
      float[] val = new float[(int) entries];
      int at = 0;
      for (int r = begin; r < end; r++) {
        FastByIDFloatMap row = matrix.get(rowIDs[r]);
        if (row == null) {
          continue;
        }
        for (FastByIDFloatMap.MapEntry e : row.entrySet()) {
          colIdx[at] = columnIndex.indexOf(e.getKey());
          val[at] = e.getValue();
          at++;
        }
      }
      check(group, nativeAppendRows(group, side, end - begin, colIdx, val, at));
      begin = end;
    }
    check(group, nativeEndMatrix(group, side));
  }

  private void uploadFactors(long group, int side, long[] ids, FastByIDMap<float[]> vectors) throws ExecutionException {
    float[] buffer = new float[Math.min(ids.length, ROWS_PER_FACTOR_PIECE) * features];
    for (int begin = 0; begin < ids.length; begin += ROWS_PER_FACTOR_PIECE) {
      int n = Math.min(ROWS_PER_FACTOR_PIECE, ids.length - begin);
      for (int r = 0; r < n; r++) {
        System.arraycopy(vectors.get(ids[begin + r]), 0, buffer, r * features, features);
      }
      check(group, nativeSetFactors(group, side, begin, n, buffer));
    }
  }

  private FastByIDMap<float[]> downloadFactors(long group, int side, long[] ids, int count) throws ExecutionException {
    FastByIDMap<float[]> result = new FastByIDMap<float[]>(count);
    float[] buffer = new float[Math.min(Math.max(count, 1), ROWS_PER_FACTOR_PIECE) * features];
    for (int begin = 0; begin < count; begin += ROWS_PER_FACTOR_PIECE) {
      int n = Math.min(ROWS_PER_FACTOR_PIECE, count - begin);
      check(group, nativeGetFactors(group, side, begin, n, buffer));
      for (int r = 0; r < n; r++) {
        float[] vector = new float[features];
        System.arraycopy(buffer, r * features, vector, 0, features);
        result.put(ids[begin + r], vector);
      }
    }
    return result;
  }

  // ---- status -> the reference's exceptions (MF:41-47, ALS:349, CMLSS:46-54) --------------------------
  private static void check(long group, int status) throws ExecutionException {
    if (status == MALS_OK) {
      return;
    }
    String message = nativeLastError(group);
    if (status == MALS_SINGULAR) {
      long[] info = new long[3];
      nativeSingularInfo(group, info);
      throw new ExecutionException(new SingularMatrixSolverException((int) info[2], message));
    }
    if (status == MALS_CANCELLED) {
      // only reachable if the cancel raced with the end of nativeFactorize; treated like any failure
      throw new ExecutionException(new IllegalStateException("cancelled: " + message));
    }
    throw new ExecutionException(new IllegalStateException("native ALS failed with status " + status + ": " + message));
  }

  private static double doubleProperty(String name, double defaultValue) {
    String value = System.getProperty(name);
    return value == null ? defaultValue : Double.parseDouble(value);
  }

  private static int[] parseDevices(String spec) {
    String[] tokens = spec.split(",");
    int[] devices = new int[tokens.length];
    for (int i = 0; i < tokens.length; i++) {
      devices[i] = Integer.parseInt(tokens[i].trim());
    }
    return devices;
  }

  private static long[] keysOf(LongPrimitiveIterator it, int size) {
    long[] keys = new long[size];
    int i = 0;
    while (it.hasNext()) {
      keys[i++] = it.nextLong();
    }
    return keys;
  }

  private static long[] toIndices(long[] ids, LongIndex index) {
    long[] out = new long[ids.length];
    for (int i = 0; i < ids.length; i++) {
      out[i] = index.indexOf(ids[i]);
    }
    return out;
  }

  /** 64-bit id -> dense index: open addressing with linear probing over a power-of-two table. */
  private static final class LongIndex {
    private final long[] keys;
    private final int[] values;
    private final int mask;

    LongIndex(long[] ids) {
      int capacity = 16;
      while (capacity < 2L * ids.length) {
        capacity <<= 1;
      }
      keys = new long[capacity];
      values = new int[capacity];
      java.util.Arrays.fill(values, -1);
      mask = capacity - 1;
      for (int i = 0; i < ids.length; i++) {
        int slot = slotOf(ids[i]);
        while (values[slot] >= 0) {
          slot = (slot + 1) & mask;
        }
        keys[slot] = ids[i];
        values[slot] = i;
      }
    }

    private int slotOf(long id) {
      long h = id * 0x9E3779B97F4A7C15L;
      return (int) (h >>> 40) & mask;
    }

    int indexOf(long id) {
      int slot = slotOf(id);
      while (values[slot] >= 0) {
        if (keys[slot] == id) {
          return values[slot];
        }
        slot = (slot + 1) & mask;
      }
      throw new IllegalStateException("id not present in the matrix: " + id);
    }
  }

}
