// The C++ mirror of InputFilesReader.readInputFiles (include/myrrix/generation.hpp) on the C-ABI: a directory of
// input files in, ids / CSR / tag sets / knownItemIDs out.  Needs a GPU (the parsing runs on the device).
// usage: test_read_input_files <empty scratch dir>
#include <sys/stat.h>
#include <utime.h>

#include <cstdio>
#include <fstream>
#include <string>

#include "../../include/myrrix/generation.hpp"

static int failures = 0;
#define CHECK(c)                                                \
  do {                                                          \
    if (!(c)) {                                                 \
      std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); \
      ++failures;                                               \
    }                                                           \
  } while (0)

static void put(const std::string& path, const std::string& text, long mtime) {
  std::ofstream(path, std::ios::binary) << text;
  utimbuf t{mtime, mtime};
  utime(path.c_str(), &t);
}

int main(int argc, char** argv) {
  if (argc != 2) return 2;
  const std::string dir = argv[1];
  // b.csv is older than a.csv: read first (ByLastModifiedComparator); its first line is the only possible header
  put(dir + "/b.csv", "user,item,value\n1,10,1\n1,11,2.5\n2,10,0.00001\n1,10,\n\"foobar\",10,2\n", 1600000000);
  put(dir + "/a.csv", "user,item\n3,12,1\n3,12,\n4,\"\",1\n#comment\n\n5,13", 1600000100);
  put(dir + "/notes.txt", "9,9,9\n", 1600000200);
  const myrrix::InputMatrices m = myrrix::readInputFiles(dir);
  CHECK(m.lines == 13 && m.badLines == 1);                       // a.csv's "header" is line 7: a bad line
  // users alive at the end: 1 (item 11), 2 (item 10, pruned from R but known), "foobar", 4, 5; user 3's entry was removed
  CHECK(m.userIDs.size() == 5 && m.userIDs[0] == 1 && m.userIDs[1] == 2 && m.userIDs[2] == 4 && m.userIDs[3] == 5 &&
        m.userIDs[4] == 4060265690780417169LL);                  // OneWayMigratorTest.java:28
  CHECK(m.itemIDs.size() == 4 && m.itemIDs[0] == -3162216497309240828LL && m.itemIDs[1] == 10 && m.itemIDs[2] == 11 && m.itemIDs[3] == 13);
  CHECK(m.itemTagIDs.size() == 1 && m.itemTagIDs[0] == 4060265690780417169LL);
  CHECK(m.userTagIDs.size() == 1 && m.userTagIDs[0] == -3162216497309240828LL);   // OneWayMigratorTest.java:29 (the empty tag)
  CHECK(m.rowPtr[0].back() == 4);                                // R: (1,11) (4,"") (5,13) ("foobar",10); (2,10) pruned
  CHECK(m.hasKnownItems && m.knownPtr.back() == 5);              // knownItemIDs keeps (2,10)
  CHECK(m.values[0].size() == 4 && m.values[0][0] == 2.5f);
  bool threw = false;
  try {
    std::string bad = "1,2,3\n";
    for (int i = 0; i < 102; ++i) bad += "x\n";
    put(dir + "/c.csv", bad, 1600000300);
    (void)myrrix::readInputFiles(dir);
  } catch (const std::ios_base::failure& e) {
    threw = std::string(e.what()).find("Too many bad lines") != std::string::npos;
  }
  CHECK(threw);
  std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
  return failures ? 1 : 0;
}
