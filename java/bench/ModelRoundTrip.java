/*
 * ModelRoundTrip -- pins model.bin.gz interoperability (SURVEY.md section 8(f) row 5) to the REFERENCE'S OWN serializer
 * wherever a JDK and the reference's jars exist.  Ours, not the reference's; compiled and run by
 * tests/test_model_java_interop.py only when `java`, `javac` and MYRRIX_CP (class path with myrrix-common,
 * myrrix-online, myrrix-online-local, commons-math3-3.2, guava-14, mahout-core-0.8, slf4j-api) are present -- never in
 * the build image (no JDK): reviewed source.
 *
 *   java -cp $MYRRIX_CP:. bench.ModelRoundTrip write <file.bin.gz>
 *       builds a small fixed Generation, writes it with GenerationSerializer.writeGeneration (GenerationSerializer.java:
 *       92-94 -> writeObject :96-105 -> IOUtils.writeObjectToFile, IOUtils.java:259-271) and prints its canonical dump
 *   java -cp $MYRRIX_CP:. bench.ModelRoundTrip read <file.bin.gz>
 *       reads the file with GenerationSerializer.readGeneration (:84-86) -- a file written by mals_model_write -- and
 *       prints the canonical dump of what the JVM got
 * Canonical dump (ids ascending, floats as raw bits, so that the comparison is exact):
 *   known null | known <n>, then  k <user> <item> <item> ...
 *   x <n>,  then  x <id> <bits> <bits> ...          (the same for y)
 *   itemtags <id> ...     usertags <id> ...
 *   userclusters <n>, then per cluster  c <member> ... | <centroid bits> ...   (clusters in list order; the same for items)
 */
package bench;

import java.io.File;
import java.util.ArrayList;
import java.util.Arrays;
import java.util.List;

import org.apache.mahout.cf.taste.impl.common.LongPrimitiveIterator;

import net.myrrix.common.collection.FastByIDMap;
import net.myrrix.common.collection.FastIDSet;
import net.myrrix.online.generation.IDCluster;
import net.myrrix.online.generation.Generation;
import net.myrrix.online.generation.GenerationSerializer;

public final class ModelRoundTrip {

  private ModelRoundTrip() {
  }

  private static long[] sortedKeys(FastByIDMap<?> map) {
    long[] ids = new long[map.size()];
    int i = 0;
    LongPrimitiveIterator it = map.keySetIterator();
    while (it.hasNext()) {
      ids[i++] = it.nextLong();
    }
    Arrays.sort(ids);
    return ids;
  }

  private static long[] sorted(FastIDSet set) {
    long[] ids = set.toArray();
    Arrays.sort(ids);
    return ids;
  }

  private static void dumpMatrix(String tag, FastByIDMap<float[]> m, StringBuilder out) {
    out.append(tag).append(' ').append(m.size()).append('\n');
    for (long id : sortedKeys(m)) {
      out.append(tag).append(' ').append(id);
      for (float f : m.get(id)) {
        out.append(' ').append(Integer.toHexString(Float.floatToRawIntBits(f)));
      }
      out.append('\n');
    }
  }

  private static void dumpClusters(String tag, List<IDCluster> clusters, StringBuilder out) {
    out.append(tag).append(' ').append(clusters.size()).append('\n');
    for (IDCluster c : clusters) {
      out.append('c');
      for (long id : sorted(c.getMembers())) {
        out.append(' ').append(id);
      }
      out.append(" |");
      for (float f : c.getCentroid()) {
        out.append(' ').append(Integer.toHexString(Float.floatToRawIntBits(f)));
      }
      out.append('\n');
    }
  }

  static String dump(Generation g) {
    StringBuilder out = new StringBuilder();
    FastByIDMap<FastIDSet> known = g.getKnownItemIDs();
    if (known == null) {
      out.append("known null\n");
    } else {
      out.append("known ").append(known.size()).append('\n');
      for (long user : sortedKeys(known)) {
        out.append("k ").append(user);
        for (long item : sorted(known.get(user))) {
          out.append(' ').append(item);
        }
        out.append('\n');
      }
    }
    dumpMatrix("x", g.getX(), out);
    dumpMatrix("y", g.getY(), out);
    out.append("itemtags");
    for (long id : sorted(g.getItemTagIDs())) {
      out.append(' ').append(id);
    }
    out.append("\nusertags");
    for (long id : sorted(g.getUserTagIDs())) {
      out.append(' ').append(id);
    }
    out.append('\n');
    dumpClusters("userclusters", g.getUserClusters(), out);
    dumpClusters("itemclusters", g.getItemClusters(), out);
    return out.toString();
  }

  /** A fixed, well-conditioned model: Generation's constructor factors X'X and Y'Y (Generation.java:132-158). */
  static Generation sample() {
    int features = 3;
    java.util.Random r = new java.util.Random(1234567890L);
    FastByIDMap<float[]> x = new FastByIDMap<float[]>();
    FastByIDMap<float[]> y = new FastByIDMap<float[]>();
    FastByIDMap<FastIDSet> known = new FastByIDMap<FastIDSet>();
    long[] users = new long[40];
    long[] items = new long[25];
    for (int i = 0; i < users.length; i++) {
      users[i] = i % 7 == 0 ? -(1L << 40) - i : 1000L + 17L * i;       // negative and large ids (hashed tags look like this)
      float[] v = new float[features];
      for (int f = 0; f < features; f++) {
        v[f] = (float) (r.nextGaussian() + (f == i % features ? 2.0 : 0.0));
      }
      x.put(users[i], v);
    }
    for (int i = 0; i < items.length; i++) {
      items[i] = i % 5 == 0 ? Long.MAX_VALUE - i : 5L + 3L * i;
      float[] v = new float[features];
      for (int f = 0; f < features; f++) {
        v[f] = (float) (r.nextGaussian() + (f == i % features ? 2.0 : 0.0));
      }
      y.put(items[i], v);
    }
    for (int i = 0; i < users.length; i += 2) {
      FastIDSet s = new FastIDSet();
      for (int j = 0; j < 1 + i % 4; j++) {
        s.add(items[(i * 3 + j * 7) % items.length]);
      }
      known.put(users[i], s);
    }
    FastIDSet itemTags = new FastIDSet();
    itemTags.add(users[0]);
    itemTags.add(users[7]);
    FastIDSet userTags = new FastIDSet();
    userTags.add(items[5]);
    List<IDCluster> userClusters = new ArrayList<IDCluster>();
    FastIDSet members = new FastIDSet();
    members.add(users[1]);
    members.add(users[2]);
    members.add(users[3]);
    userClusters.add(new IDCluster(members, new float[] {0.5f, -1.25f, 3.0f}));
    List<IDCluster> itemClusters = new ArrayList<IDCluster>();
    FastIDSet im = new FastIDSet();
    im.add(items[4]);
    itemClusters.add(new IDCluster(im, new float[] {1.0f, 2.0f, -0.0f}));
    FastIDSet im2 = new FastIDSet();
    im2.add(items[6]);
    im2.add(items[8]);
    itemClusters.add(new IDCluster(im2, new float[] {Float.MIN_VALUE, 1e30f, -7.5f}));
    return new Generation(known, x, y, itemTags, userTags, userClusters, itemClusters);
  }

  public static void main(String[] args) throws Exception {
    if (args.length != 2) {
      System.err.println("usage: ModelRoundTrip write|read <file.bin.gz>");
      System.exit(2);
    }
    File f = new File(args[1]);
    if ("write".equals(args[0])) {
      Generation g = sample();
      GenerationSerializer.writeGeneration(g, f);
      System.out.print(dump(g));
    } else if ("read".equals(args[0])) {
      System.out.print(dump(GenerationSerializer.readGeneration(f)));
    } else {
      System.exit(2);
    }
  }

}
