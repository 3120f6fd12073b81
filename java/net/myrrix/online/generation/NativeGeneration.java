/*
 * NativeGeneration -- the Java side of the two rows either side of the factorization that the MI355X-native library also
 * covers (SURVEY.md section 8(f) rows 2 and 4), via jni/myrrix_serving_jni.c:
 *
 *   readInputFiles(...)   same signature and effect as InputFilesReader.readInputFiles
 *                         (online-local/src/net/myrrix/online/generation/InputFilesReader.java:64-211): the files of the input
 *                         directory are read, split, parsed and merged ON THE DEVICE (mals_ingest_read_dir + _finish) and the
 *                         reference's own structures are filled from the resulting CSR -- RbyRow, RbyColumn, itemTagIDs,
 *                         userTagIDs, knownItemIDs -- so DelegateGenerationManager keeps working unchanged; a caller that goes
 *                         on to the native factorizer can skip the maps and call install(handle) instead.
 *   recommend(...) /      ServerRecommender.recommend / recommendToMany / recommendToAnonymous without rescorer or candidate
 *   recommendToMany(...)  filter (online/src/net/myrrix/online/ServerRecommender.java:366-508,561-606) for dense user / item
 *                         indices: the scores are the reference's (RecommendIterator.java:93-104) bit for bit.
 *
 * NOT BUILT IN THIS REPOSITORY'S IMAGE (no JDK); complete source against the reference's public classes, like
 * HipAlternatingLeastSquares.  Same package as InputFilesReader so that the call site at DelegateGenerationManager.java:336
 * switches by class name only.
 */
package net.myrrix.online.generation;

import java.io.File;
import java.io.IOException;
import java.nio.charset.Charset;

import net.myrrix.common.collection.FastByIDFloatMap;
import net.myrrix.common.collection.FastByIDMap;
import net.myrrix.common.collection.FastIDSet;

public final class NativeGeneration implements AutoCloseable {

  private static final float ZERO_THRESHOLD = Float.parseFloat(System.getProperty("model.decay.zeroThreshold", "0.0001"));   // IFR:58-59

  static {
    System.loadLibrary("myrrix_serving_jni");   // which loads libmyrrix_als.so
  }

  private long ingest;            // mals_ingest
  private long[] userIDs;         // dense index -> id, after readInputFiles
  private long[] itemIDs;

  private static native long nativeIngestCreate(int device, float zeroThreshold, boolean wantKnownItems);
  private static native void nativeIngestDestroy(long ingest);
  private static native String nativeIngestLastError(long ingest);
  private static native int nativeReadInputDir(long ingest, long[] dirBytes, long[] info);
  private static native int nativeIngestIds(long ingest, int side, long[] ids);
  private static native int nativeIngestTagIds(long ingest, int which, long[] ids);
  private static native int nativeIngestSetSizes(long ingest, long[] out);
  private static native int nativeIngestCsr(long ingest, int side, long[] rowPtr, int[] colIdx, float[] val);
  private static native int nativeIngestKnownItems(long ingest, long[] ptr, int[] itemIdx);
  private static native int nativeIngestInstall(long ingest, long handle);
  private static native int nativeIngestInstallGroup(long ingest, long group, int flags);
  private static native long nativeGroupHandle(long group, int member);
  private static native int nativeSetKnownItems(long handle, long[] rowPtr, int[] itemIdx);
  private static native int nativeRecommend(long handle, long[] userIdx, int howMany, boolean considerKnownItems,
                                            long[] items, float[] scores, int[] counts);
  private static native int nativeRecommendToMany(long handle, float[] vectors, long[] vectorPtr, int nQueries, int howMany,
                                                  long[] excludePtr, long[] excludeIdx, long[] items, float[] scores, int[] counts);
  private static native String nativeLastError(long handle);

  /** The reference's own six arguments (InputFilesReader.java:64-69; call site DelegateGenerationManager.java:336): a switch by
   *  class name only.  The device is the first of -Dmodel.als.gpus (default 0). */
  public static NativeGeneration readInputFiles(FastByIDMap<FastIDSet> knownItemIDs,
                                                FastByIDMap<FastByIDFloatMap> rbyRow,
                                                FastByIDMap<FastByIDFloatMap> rbyColumn,
                                                FastIDSet itemTagIDs,
                                                FastIDSet userTagIDs,
                                                File inputDir) throws IOException {
    String gpus = System.getProperty("model.als.gpus", "0");
    int device = Integer.parseInt(gpus.split(",")[0].trim());
    return readInputFiles(knownItemIDs, rbyRow, rbyColumn, itemTagIDs, userTagIDs, inputDir, device);
  }

  /** InputFilesReader.readInputFiles (IFR:64-211) with the work on device `device`; the object keeps the matrices in HBM
   *  until close() so that install() can hand them to a factorizer without a round trip. */
  public static NativeGeneration readInputFiles(FastByIDMap<FastIDSet> knownItemIDs,
                                                FastByIDMap<FastByIDFloatMap> rbyRow,
                                                FastByIDMap<FastByIDFloatMap> rbyColumn,
                                                FastIDSet itemTagIDs,
                                                FastIDSet userTagIDs,
                                                File inputDir,
                                                int device) throws IOException {
    NativeGeneration g = new NativeGeneration();
    g.ingest = nativeIngestCreate(device, ZERO_THRESHOLD, knownItemIDs != null);
    if (g.ingest == 0) {
      throw new IOException("no native ingest on device " + device);
    }
    boolean ok = false;
    try {
      byte[] path = inputDir.getAbsolutePath().getBytes(Charset.forName("UTF-8"));
      long[] wide = new long[path.length];
      for (int i = 0; i < path.length; i++) {
        wide[i] = path[i];
      }
      long[] info = new long[8];   // lines, badLines, headerLines, skippedLines, records, users, items, nnz
      check(g.ingest, nativeReadInputDir(g.ingest, wide, info));   // "Too many bad lines" (IFR:96-98) arrives here
      // Java arrays end at 2^31 - 1 elements: an input with more users, items or entries than that cannot become the reference's
      // maps here (nor could the reference hold it) -- such inputs go to the factorizer with install() / installGroup() instead
      if (info[5] > Integer.MAX_VALUE || info[6] > Integer.MAX_VALUE || info[7] > Integer.MAX_VALUE) {
        throw new IOException("input too large for Java maps (" + info[5] + " users, " + info[6] + " items, " + info[7]
            + " entries): read it with NativeGeneration.readInputFiles(inputDir, device) and install() it");
      }
      int nUsers = (int) info[5];
      int nItems = (int) info[6];
      int nnz = (int) info[7];
      g.userIDs = new long[nUsers];
      g.itemIDs = new long[nItems];
      check(g.ingest, nativeIngestIds(g.ingest, 0, g.userIDs));
      check(g.ingest, nativeIngestIds(g.ingest, 1, g.itemIDs));
      fill(g, 0, nUsers, nnz, g.userIDs, g.itemIDs, rbyRow);
      fill(g, 1, nItems, nnz, g.itemIDs, g.userIDs, rbyColumn);
      long[] sizes = new long[3];
      check(g.ingest, nativeIngestSetSizes(g.ingest, sizes));
      if (sizes[0] > Integer.MAX_VALUE || sizes[1] > Integer.MAX_VALUE || sizes[2] > Integer.MAX_VALUE) {
        throw new IOException("input too large for Java sets (" + sizes[0] + " / " + sizes[1] + " tag ids, " + sizes[2] + " known items)");
      }
      addAll(g.ingest, 0, (int) sizes[0], itemTagIDs);
      addAll(g.ingest, 1, (int) sizes[1], userTagIDs);
      if (knownItemIDs != null && sizes[2] >= 0) {
        long[] ptr = new long[nUsers + 1];
        int[] idx = new int[(int) sizes[2]];
        check(g.ingest, nativeIngestKnownItems(g.ingest, ptr, idx));
        for (int u = 0; u < nUsers; u++) {
          FastIDSet known = new FastIDSet((int) (ptr[u + 1] - ptr[u]));
          for (long e = ptr[u]; e < ptr[u + 1]; e++) {
            known.add(g.itemIDs[idx[(int) e]]);
          }
          knownItemIDs.put(g.userIDs[u], known);
        }
      }
      ok = true;
      return g;
    } finally {
      if (!ok) {
        g.close();
      }
    }
  }

  private static void fill(NativeGeneration g, int side, int nRows, int nnz, long[] rowIDs, long[] colIDs,
                           FastByIDMap<FastByIDFloatMap> out) throws IOException {
    long[] rowPtr = new long[nRows + 1];
    int[] colIdx = new int[nnz];
    float[] val = new float[nnz];
    check(g.ingest, nativeIngestCsr(g.ingest, side, rowPtr, colIdx, val));
    for (int r = 0; r < nRows; r++) {
      FastByIDFloatMap row = new FastByIDFloatMap((int) (rowPtr[r + 1] - rowPtr[r]));
      for (long e = rowPtr[r]; e < rowPtr[r + 1]; e++) {
        row.put(colIDs[colIdx[(int) e]], val[(int) e]);
      }
      out.put(rowIDs[r], row);   // rows emptied by removeSmall stay, as in the reference (IFR:202-211 vs MU:117-125)
    }
  }

  private static void addAll(long ingest, int which, int n, FastIDSet into) throws IOException {
    if (n <= 0) {
      return;
    }
    long[] ids = new long[n];
    check(ingest, nativeIngestTagIds(ingest, which, ids));
    for (long id : ids) {
      into.add(id);
    }
  }

  private static void check(long ingest, int status) throws IOException {
    if (status != 0) {
      throw new IOException(nativeIngestLastError(ingest));   // e.g. "Too many bad lines; aborting" (IFR:96-98)
    }
  }

  /** Both matrices and knownItemIDs to a native factorizer handle on the same device, without leaving HBM
   *  (mals_ingest_install); the handle borrows them: keep this object open while it factorizes and serves. */
  public void install(long handle) throws IOException {
    check(ingest, nativeIngestInstall(ingest, handle));
  }

  /** The same for a multi-GPU factorizer (mals_ingest_install_group): both matrices cut at the group's cost-balanced bounds, every
   *  member its slices device to device, its users' knownItemIDs and the userTagIDs mask.  copy = true: the members own copies
   *  and this object may be closed right away; false: members on this object's device borrow, keep it open. */
  public void installGroup(long group, boolean copy) throws IOException {
    check(ingest, nativeIngestInstallGroup(ingest, group, copy ? 1 : 0));
  }

  public long[] getUserIDs() {
    return userIDs;
  }

  public long[] getItemIDs() {
    return itemIDs;
  }

  /** The single-GPU handle of a member of the factorizer's group (HipAlternatingLeastSquares): factors and R stay resident
   *  after call(), so the generation just built is served from the same memory. */
  public static long handleOf(long group, int member) {
    long handle = nativeGroupHandle(group, member);
    if (handle == 0) {
      throw new IllegalStateException("no member " + member + " in this group");
    }
    return handle;
  }

  /** generation.getKnownItemIDs() for a handle that was not fed by install(): CSR over its user rows, dense item indices. */
  public static void setKnownItems(long handle, long[] rowPtr, int[] itemIdx) {
    checkHandle(handle, nativeSetKnownItems(handle, rowPtr, itemIdx));
  }

  /** ServerRecommender.recommend(userID, howMany, considerKnownItems, null) (SR:382-441) for users by dense index:
   *  items[q * howMany + j] = dense index of the j-th best item of query q (-1 beyond counts[q]), scores likewise. */
  public static void recommend(long handle, long[] userIdx, int howMany, boolean considerKnownItems,
                               long[] items, float[] scores, int[] counts) {
    checkHandle(handle, nativeRecommend(handle, userIdx, howMany, considerKnownItems, items, scores, counts));
  }

  /** recommendToMany (SR:366-441; the caller passes the intersection of the users' known items as the exclusion list,
   *  SR:398-425) and recommendToAnonymous (SR:561-606; the query vector comes from the Solver of the generation). */
  public static void recommendToMany(long handle, float[] vectors, long[] vectorPtr, int nQueries, int howMany,
                                     long[] excludePtr, long[] excludeIdx, long[] items, float[] scores, int[] counts) {
    checkHandle(handle, nativeRecommendToMany(handle, vectors, vectorPtr, nQueries, howMany, excludePtr, excludeIdx, items, scores, counts));
  }

  private static void checkHandle(long handle, int status) {
    if (status != 0) {
      throw new IllegalStateException("native top-N failed with status " + status + ": " + nativeLastError(handle));
    }
  }

  @Override
  public void close() {
    if (ingest != 0) {
      nativeIngestDestroy(ingest);
      ingest = 0;
    }
  }
}
