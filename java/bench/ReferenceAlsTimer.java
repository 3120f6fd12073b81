/*
 * ReferenceAlsTimer -- times the REAL net.myrrix.online.factorizer.als.AlternatingLeastSquares (the reference's
 * multi-threaded CPU path) on a matrix written by bench.py, for the opportunistic "reference JVM path" leg of
 * cpu_baseline (BASELINE.md section 4).  Ours, not the reference's; compiled and run by bench.py only when
 * `java`, `javac` and the reference's jars are present (env MYRRIX_CP = class path with myrrix-common,
 * myrrix-online, commons-math3-3.2, guava-14, mahout-core-0.8, slf4j-api).  NOT built in this repository's image
 * (no JDK): reviewed source.
 *
 * usage: java -cp $MYRRIX_CP:. bench.ReferenceAlsTimer <entries.bin> <features> <iterations>
 *   entries.bin: little-endian int32 n, then n x (int64 user, int64 item, float32 value)
 * prints one line: rows_per_s=<(users + items) * iterations / seconds> threads=<availableProcessors>
 * AlternatingLeastSquares.call() with maxIterations = N runs the initial-Y construction once and N full
 * iterations (ALS:176-262); two runs (N and 1) are differenced so that the one-off parts cancel.
 */
package bench;

import java.io.DataInputStream;
import java.io.FileInputStream;
import java.io.BufferedInputStream;

import net.myrrix.common.collection.FastByIDFloatMap;
import net.myrrix.common.collection.FastByIDMap;
import net.myrrix.common.math.MatrixUtils;
import net.myrrix.online.factorizer.MatrixFactorizer;
import net.myrrix.online.factorizer.als.AlternatingLeastSquares;

public final class ReferenceAlsTimer {

  private ReferenceAlsTimer() {
  }

  private static long readLongLE(DataInputStream in) throws java.io.IOException {
    return Long.reverseBytes(in.readLong());
  }

  private static int readIntLE(DataInputStream in) throws java.io.IOException {
    return Integer.reverseBytes(in.readInt());
  }

  private static double timeCall(FastByIDMap<FastByIDFloatMap> byRow, FastByIDMap<FastByIDFloatMap> byColumn,
                                 int features, int iterations) throws Exception {
    // threshold tiny: only the iteration cap ends the loop
    MatrixFactorizer als = new AlternatingLeastSquares(byRow, byColumn, features, 1.0e-12, iterations);
    long start = System.nanoTime();
    als.call();
    return (System.nanoTime() - start) * 1.0e-9;
  }

  public static void main(String[] args) throws Exception {
    int features = Integer.parseInt(args[1]);
    int iterations = Integer.parseInt(args[2]);
    FastByIDMap<FastByIDFloatMap> byRow = new FastByIDMap<FastByIDFloatMap>();
    FastByIDMap<FastByIDFloatMap> byColumn = new FastByIDMap<FastByIDFloatMap>();
    DataInputStream in = new DataInputStream(new BufferedInputStream(new FileInputStream(args[0]), 1 << 20));
    try {
      int n = readIntLE(in);
      for (int i = 0; i < n; i++) {
        long user = readLongLE(in);
        long item = readLongLE(in);
        float value = Float.intBitsToFloat(readIntLE(in));
        MatrixUtils.addTo(user, item, value, byRow, byColumn);
      }
    } finally {
      in.close();
    }
    timeCall(byRow, byColumn, features, 1);                              // JIT warm-up
    double one = timeCall(byRow, byColumn, features, 1);
    double many = timeCall(byRow, byColumn, features, 1 + iterations);
    double perIteration = (many - one) / iterations;
    System.out.println("rows_per_s=" + ((byRow.size() + byColumn.size()) / perIteration)
                       + " threads=" + Runtime.getRuntime().availableProcessors());
  }

}
