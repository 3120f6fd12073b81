// The C++ mirror of GenerationSerializer (include/myrrix/serializer.hpp) on the C-ABI: host code, no GPU.
// Writes a model, checks the stream against the bytes spelled out by hand in tests/test_model_oracle.py
// (passed in as a hex string), reads it back, and checks the reference's error behaviour.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/myrrix/serializer.hpp"

using namespace myrrix;

static int failures = 0;
#define CHECK(c)                                              \
  do {                                                        \
    if (!(c)) {                                               \
      std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); \
      ++failures;                                             \
    }                                                         \
  } while (0)

static std::string gunzip(const std::string& path) {
  gzFile f = gzopen(path.c_str(), "rb");
  std::string out;
  char buf[4096];
  int n;
  while (f && (n = gzread(f, buf, sizeof buf)) > 0) out.append(buf, (size_t)n);
  if (f) gzclose(f);
  return out;
}
static std::string hex(const std::string& s) {
  static const char* d = "0123456789abcdef";
  std::string o;
  for (unsigned char c : s) {
    o.push_back(d[c >> 4]);
    o.push_back(d[c & 15]);
  }
  return o;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::printf("usage: test_model_file <dir> <expected hex of the tiny stream>\n");
    return 2;
  }
  const std::string dir = argv[1], want = argv[2];
  SerializedGeneration tiny;
  tiny.knownItemIDs[5] = {7};
  tiny.X[5] = {1.0f, -2.0f};
  tiny.Y[7] = {0.5f, 0.25f};
  GenerationSerializer::writeGeneration(tiny, dir + "/model.bin.gz");
  CHECK(hex(gunzip(dir + "/model.bin.gz")) == want);
  const SerializedGeneration back = GenerationSerializer::readGeneration(dir + "/model.bin.gz");
  CHECK(back.hasKnownItemIDs && back.knownItemIDs == tiny.knownItemIDs && back.X == tiny.X && back.Y == tiny.Y);
  CHECK(back.itemTagIDs.empty() && back.userClusters.empty());

  SerializedGeneration g;
  g.hasKnownItemIDs = false;
  for (int64_t u = -3; u < 400; ++u) g.X[u * 1000003] = FloatVector(30, 0.5f + (float)u);
  for (int64_t i = 0; i < 90; ++i) g.Y[i - 45] = FloatVector(30, (float)i / 7.0f);
  g.itemTagIDs = {11, -12};
  g.userTagIDs = {13};
  g.userClusters.push_back(IDCluster{{1, 2, 3}, FloatVector(30, 0.25f)});
  g.itemClusters.push_back(IDCluster{{}, {}});
  GenerationSerializer::writeGeneration(g, dir + "/big.bin.gz");
  const SerializedGeneration b2 = GenerationSerializer::readGeneration(dir + "/big.bin.gz");
  CHECK(!b2.hasKnownItemIDs && b2.X == g.X && b2.Y == g.Y && b2.itemTagIDs == g.itemTagIDs && b2.userTagIDs == g.userTagIDs);
  CHECK(b2.userClusters.size() == 1 && b2.userClusters[0].members == g.userClusters[0].members &&
        b2.userClusters[0].centroid == g.userClusters[0].centroid && b2.itemClusters.size() == 1 && b2.itemClusters[0].members.empty());

  bool threw = false;
  try {
    GenerationSerializer::writeGeneration(tiny, dir + "/model.bin");  // IOUtils.java:276
  } catch (const IllegalStateException&) {
    threw = true;
  }
  CHECK(threw);
  threw = false;
  try {
    SerializedGeneration bad = tiny;
    bad.Y[7][1] = INFINITY;  // GS:196
    GenerationSerializer::writeGeneration(bad, dir + "/bad.bin.gz");
  } catch (const IllegalStateException&) {
    threw = true;
  }
  CHECK(threw);
  threw = false;
  try {
    GenerationSerializer::readGeneration(dir + "/absent.bin.gz");
  } catch (const IOException&) {
    threw = true;
  }
  CHECK(threw);
  std::printf(failures ? "%d FAILED\n" : "ALL PASSED\n", failures);
  return failures ? 1 : 0;
}
