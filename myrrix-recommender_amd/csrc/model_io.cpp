// model.bin.gz reader / writer (include/myrrix_als.h, "section 8(f) row 5").  Host code only.
//
// What is in the file (GenerationSerializer.java:96-262 through IOUtils.java:259-283): gzip( Java
// Object Serialization stream ).  The stream, by the grammar of the Java Object Serialization
// Specification section 6.4.2:
//   magic 0xACED, version 5,
//   TC_OBJECT, TC_CLASSDESC, className (modified UTF-8, u16 length), serialVersionUID (8 bytes),
//   classDescFlags = SC_SERIALIZABLE | SC_WRITE_METHOD (the class has a private writeObject),
//   fields: one, the non-transient `private Generation generation` (GS:55): 'L', "generation",
//           TC_STRING "Lnet/myrrix/online/generation/Generation;",
//   classAnnotation: TC_ENDBLOCKDATA, superClassDesc: TC_NULL,
//   then, because writeObject never calls defaultWriteObject, only the objectAnnotation: the bytes the
//   DataOutput calls of GS:96-105 produced, cut into block-data records of at most 1024 bytes
//   (TC_BLOCKDATA + u8 length up to 255 bytes, TC_BLOCKDATALONG + i32 length above), TC_ENDBLOCKDATA.
// The reader accepts any record sizes and any field list in the class descriptor.
#include <zlib.h>

#include <cmath>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/myrrix_als.h"

namespace {

constexpr uint8_t TC_NULL = 0x70, TC_REFERENCE = 0x71, TC_CLASSDESC = 0x72, TC_OBJECT = 0x73, TC_STRING = 0x74,
                  TC_BLOCKDATA = 0x77, TC_ENDBLOCKDATA = 0x78, TC_BLOCKDATALONG = 0x7A, TC_LONGSTRING = 0x7C;
constexpr uint8_t SC_WRITE_METHOD = 0x01, SC_SERIALIZABLE = 0x02;
constexpr int MAX_BLOCK = 1024;
const char* const CLASS_NAME = "net.myrrix.online.generation.GenerationSerializer";
const char* const FIELD_NAME = "generation";
const char* const FIELD_TYPE = "Lnet/myrrix/online/generation/Generation;";

thread_local std::string g_err;

struct Failure {
  int status;
  std::string what;
};
[[noreturn]] void fail(int status, const std::string& what) { throw Failure{status, what}; }

// ---------------------------------------------------------------- writing
struct GzOut {
  gzFile f = nullptr;
  std::vector<uint8_t> buf;
  explicit GzOut(const char* path) {
    f = gzopen(path, "wb6");
    if (!f) fail(MALS_IO_ERROR, std::string("cannot open for writing: ") + path);
    gzbuffer(f, 1 << 20);
    buf.reserve(1 << 20);
  }
  ~GzOut() {
    if (f) gzclose(f);
  }
  void flush() {
    if (!buf.empty() && gzwrite(f, buf.data(), (unsigned)buf.size()) != (int)buf.size()) fail(MALS_IO_ERROR, "write failed");
    buf.clear();
  }
  void raw(const void* p, size_t n) {
    if (buf.size() + n > buf.capacity()) flush();
    const uint8_t* b = (const uint8_t*)p;
    buf.insert(buf.end(), b, b + n);
  }
  void u8(uint8_t v) { raw(&v, 1); }
  void be16(uint16_t v) {
    uint8_t b[2] = {(uint8_t)(v >> 8), (uint8_t)v};
    raw(b, 2);
  }
  void be32(uint32_t v) {
    uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
    raw(b, 4);
  }
  void be64(uint64_t v) {
    be32((uint32_t)(v >> 32));
    be32((uint32_t)v);
  }
  void utf(const char* s) {
    be16((uint16_t)strlen(s));
    raw(s, strlen(s));
  }
  void close() {
    flush();
    int rc = gzclose(f);
    f = nullptr;
    if (rc != Z_OK) fail(MALS_IO_ERROR, "close failed");
  }
};

// the block-data mode of ObjectOutputStream: a 1024-byte buffer drained as one record when full
struct BlockOut {
  GzOut& out;
  uint8_t blk[MAX_BLOCK];
  int pos = 0;
  explicit BlockOut(GzOut& o) : out(o) {}
  void drain() {
    if (!pos) return;
    if (pos <= 0xFF) {
      out.u8(TC_BLOCKDATA);
      out.u8((uint8_t)pos);
    } else {
      out.u8(TC_BLOCKDATALONG);
      out.be32((uint32_t)pos);
    }
    out.raw(blk, pos);
    pos = 0;
  }
  void byte(uint8_t v) {
    if (pos == MAX_BLOCK) drain();
    blk[pos++] = v;
  }
  void i32(int32_t v) {
    uint32_t u = (uint32_t)v;
    byte(u >> 24), byte(u >> 16), byte(u >> 8), byte(u);
  }
  void i64(int64_t v) {
    i32((int32_t)((uint64_t)v >> 32));
    i32((int32_t)(uint64_t)v);
  }
  void f32(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);  // writeFloat = floatToIntBits; NaN never gets here for factors, centroids keep their bits
    i32((int32_t)u);
  }
};

int32_t checked_count(int64_t n, const char* what) {
  if (n < 0 || n > INT32_MAX) fail(MALS_INVALID_ARG, std::string(what) + ": count does not fit a Java int");
  return (int32_t)n;
}

void write_matrix(BlockOut& b, int64_t n, const int64_t* ids, const float* rows, int32_t k, const char* what) {
  b.i32(checked_count(n, what));
  if (n && (!ids || (k > 0 && !rows))) fail(MALS_INVALID_ARG, std::string(what) + ": null array");
  for (int64_t r = 0; r < n; ++r) {
    b.i64(ids[r]);
    b.i32(k);
    for (int32_t j = 0; j < k; ++j) {
      const float v = rows[r * (int64_t)k + j];
      if (!std::isfinite(v)) fail(MALS_INVALID_ARG, std::string(what) + ": non-finite factor");
      b.f32(v);
    }
  }
}

void write_id_set(BlockOut& b, int64_t n, const int64_t* ids, const char* what) {
  b.i32(checked_count(n, what));
  if (n && !ids) fail(MALS_INVALID_ARG, std::string(what) + ": null array");
  for (int64_t i = 0; i < n; ++i) b.i64(ids[i]);
}

void write_clusters(BlockOut& b, int64_t n, const int64_t* mptr, const int64_t* members, const int64_t* cptr, const float* cent,
                    const char* what) {
  b.i32(checked_count(n, what));
  if (n && (!mptr || !cptr)) fail(MALS_INVALID_ARG, std::string(what) + ": null array");
  for (int64_t c = 0; c < n; ++c) {
    write_id_set(b, mptr[c + 1] - mptr[c], members + mptr[c], what);
    b.i32(checked_count(cptr[c + 1] - cptr[c], what));
    for (int64_t j = cptr[c]; j < cptr[c + 1]; ++j) b.f32(cent[j]);
  }
}

void write_model(const char* path, const mals_model_view& m) {
  const size_t len = strlen(path);
  if (len < 3 || strcmp(path + len - 3, ".gz") != 0) fail(MALS_INVALID_ARG, std::string("File should end in .gz: ") + path);
  if (m.features < 0) fail(MALS_INVALID_ARG, "negative feature count");
  GzOut out(path);
  out.be16(0xACED);
  out.be16(5);
  out.u8(TC_OBJECT);
  out.u8(TC_CLASSDESC);
  out.utf(CLASS_NAME);
  out.be64(1);  // serialVersionUID, GS:51
  out.u8(SC_SERIALIZABLE | SC_WRITE_METHOD);
  out.be16(1);
  out.u8('L');
  out.utf(FIELD_NAME);
  out.u8(TC_STRING);
  out.utf(FIELD_TYPE);
  out.u8(TC_ENDBLOCKDATA);
  out.u8(TC_NULL);
  BlockOut b(out);
  if (m.n_known < 0) {
    b.i32(-1);  // NULL_COUNT, GS:53
  } else {
    b.i32(checked_count(m.n_known, "knownItemIDs"));
    if (m.n_known && (!m.known_user_ids || !m.known_ptr)) fail(MALS_INVALID_ARG, "knownItemIDs: null array");
    for (int64_t u = 0; u < m.n_known; ++u) {
      b.i64(m.known_user_ids[u]);
      write_id_set(b, m.known_ptr[u + 1] - m.known_ptr[u], m.known_item_ids + m.known_ptr[u], "knownItemIDs");
    }
  }
  write_matrix(b, m.n_users, m.user_ids, m.X, m.features, "X");
  write_matrix(b, m.n_items, m.item_ids, m.Y, m.features, "Y");
  write_id_set(b, m.n_item_tags, m.item_tag_ids, "itemTagIDs");
  write_id_set(b, m.n_user_tags, m.user_tag_ids, "userTagIDs");
  write_clusters(b, m.n_user_clusters, m.user_cluster_member_ptr, m.user_cluster_members, m.user_cluster_centroid_ptr,
                 m.user_cluster_centroids, "userClusters");
  write_clusters(b, m.n_item_clusters, m.item_cluster_member_ptr, m.item_cluster_members, m.item_cluster_centroid_ptr,
                 m.item_cluster_centroids, "itemClusters");
  b.drain();
  out.u8(TC_ENDBLOCKDATA);
  out.close();
}

// ---------------------------------------------------------------- reading
struct GzIn {
  gzFile f = nullptr;
  std::vector<uint8_t> buf;
  size_t pos = 0, end = 0;
  explicit GzIn(const char* path) {
    f = gzopen(path, "rb");
    if (!f) fail(MALS_IO_ERROR, std::string("cannot open: ") + path);
    gzbuffer(f, 1 << 20);
    buf.resize(1 << 20);
  }
  ~GzIn() {
    if (f) gzclose(f);
  }
  bool fill() {
    const int n = gzread(f, buf.data(), (unsigned)buf.size());
    if (n < 0) {
      int e = 0;
      fail(MALS_IO_ERROR, std::string("corrupt gzip stream: ") + gzerror(f, &e));
    }
    pos = 0;
    end = (size_t)n;
    return n > 0;
  }
  void raw(void* p, size_t n) {
    uint8_t* b = (uint8_t*)p;
    while (n) {
      if (pos == end && !fill()) fail(MALS_IO_ERROR, "unexpected end of stream");
      const size_t take = std::min(n, end - pos);
      memcpy(b, buf.data() + pos, take);
      pos += take, b += take, n -= take;
    }
  }
  uint8_t u8() {
    uint8_t v;
    raw(&v, 1);
    return v;
  }
  uint16_t be16() {
    uint8_t b[2];
    raw(b, 2);
    return (uint16_t)(b[0] << 8 | b[1]);
  }
  uint32_t be32() {
    uint8_t b[4];
    raw(b, 4);
    return (uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | b[3];
  }
  uint64_t be64() {
    const uint64_t hi = be32();
    return hi << 32 | be32();
  }
  std::string utf() {
    std::string s(be16(), '\0');
    raw(s.data(), s.size());
    return s;
  }
  void skip(uint64_t n) {
    uint8_t tmp[256];
    while (n) {
      const size_t take = (size_t)std::min<uint64_t>(n, sizeof tmp);
      raw(tmp, take);
      n -= take;
    }
  }
};

struct BlockIn {
  GzIn& in;
  uint64_t left = 0;  // unread bytes of the current record
  explicit BlockIn(GzIn& i) : in(i) {}
  void next_record() {
    const uint8_t tc = in.u8();
    if (tc == TC_BLOCKDATA)
      left = in.u8();
    else if (tc == TC_BLOCKDATALONG) {
      const int32_t n = (int32_t)in.be32();
      if (n < 0) fail(MALS_IO_ERROR, "negative block length");
      left = (uint64_t)n;
    } else if (tc == TC_ENDBLOCKDATA)
      fail(MALS_IO_ERROR, "model data ends early (java.io.EOFException)");
    else
      fail(MALS_IO_ERROR, "unexpected tag inside the model data (java.io.StreamCorruptedException)");
  }
  void bytes(uint8_t* p, size_t n) {
    while (n) {
      while (!left) next_record();
      const size_t take = (size_t)std::min<uint64_t>(n, left);
      in.raw(p, take);
      p += take, n -= take, left -= take;
    }
  }
  int32_t i32() {
    uint8_t b[4];
    bytes(b, 4);
    return (int32_t)((uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | b[3]);
  }
  int64_t i64() {
    const uint64_t hi = (uint32_t)i32();
    return (int64_t)(hi << 32 | (uint32_t)i32());
  }
  float f32() {
    const uint32_t u = (uint32_t)i32();
    float v;
    memcpy(&v, &u, 4);
    return v;
  }
  int32_t count(const char* what) {
    const int32_t n = i32();
    if (n < 0) fail(MALS_IO_ERROR, std::string(what) + ": negative count");
    return n;
  }
  void i64s(std::vector<int64_t>& dst, int32_t n) {
    for (int32_t i = 0; i < n; ++i) dst.push_back(i64());
  }
};

struct Clusters {
  std::vector<int64_t> member_ptr{0}, members, centroid_ptr{0};
  std::vector<float> centroids;
};

}  // namespace

struct mals_model_s {
  int32_t features = 0;
  std::vector<int64_t> user_ids, item_ids;
  std::vector<float> X, Y;
  bool known_null = false;
  std::vector<int64_t> known_user_ids, known_ptr{0}, known_item_ids;
  std::vector<int64_t> item_tags, user_tags;
  Clusters user_clusters, item_clusters;
};

namespace {

void skip_class_name_object(GzIn& in) {  // className1 of an object-typed field: a string or a back reference
  const uint8_t tc = in.u8();
  if (tc == TC_STRING)
    in.skip(in.be16());
  else if (tc == TC_LONGSTRING)
    in.skip(in.be64());
  else if (tc == TC_REFERENCE)
    in.skip(4);
  else
    fail(MALS_IO_ERROR, "unexpected tag in a field descriptor");
}

void read_matrix(BlockIn& b, std::vector<int64_t>& ids, std::vector<float>& rows, int32_t& features, bool& have_features,
                 const char* what) {
  const int32_t n = b.count(what);
  // the counts come from the file: a corrupt header must not drive a multi-GB allocation before any data is read
  ids.reserve(std::min<size_t>((size_t)n, (size_t)1 << 20));
  for (int32_t r = 0; r < n; ++r) {
    ids.push_back(b.i64());
    const int32_t k = b.count(what);
    if (!have_features) {
      features = k;
      have_features = true;
      rows.reserve(std::min<size_t>((size_t)n * (size_t)k, (size_t)1 << 24));
    } else if (k != features) {
      fail(MALS_INVALID_ARG, std::string(what) + ": rows of different lengths");
    }
    for (int32_t j = 0; j < k; ++j) {
      const float v = b.f32();
      if (!std::isfinite(v)) fail(MALS_INVALID_ARG, std::string(what) + ": non-finite factor (GenerationSerializer.readMatrix)");
      rows.push_back(v);
    }
  }
}

void read_clusters(BlockIn& b, Clusters& c, const char* what) {
  const int32_t n = b.count(what);
  for (int32_t i = 0; i < n; ++i) {
    b.i64s(c.members, b.count(what));
    c.member_ptr.push_back((int64_t)c.members.size());
    const int32_t k = b.count(what);
    for (int32_t j = 0; j < k; ++j) c.centroids.push_back(b.f32());
    c.centroid_ptr.push_back((int64_t)c.centroids.size());
  }
}

void read_model(const char* path, mals_model_s& m) {
  GzIn in(path);
  if (in.be16() != 0xACED) fail(MALS_IO_ERROR, "not a Java serialization stream (java.io.StreamCorruptedException: invalid stream header)");
  if (in.be16() != 5) fail(MALS_IO_ERROR, "unsupported serialization stream version");
  if (in.u8() != TC_OBJECT || in.u8() != TC_CLASSDESC) fail(MALS_IO_ERROR, "the stream does not start with a new object of a new class");
  const std::string cls = in.utf();
  if (cls != CLASS_NAME) fail(MALS_IO_ERROR, "unexpected class " + cls);
  if (in.be64() != 1) fail(MALS_IO_ERROR, "serialVersionUID mismatch (java.io.InvalidClassException)");
  const uint8_t flags = in.u8();
  if (!(flags & SC_SERIALIZABLE) || !(flags & SC_WRITE_METHOD)) fail(MALS_IO_ERROR, "unexpected class descriptor flags");
  const uint16_t n_fields = in.be16();
  for (uint16_t i = 0; i < n_fields; ++i) {
    const uint8_t type = in.u8();
    in.skip(in.be16());
    if (type == 'L' || type == '[') skip_class_name_object(in);
  }
  if (in.u8() != TC_ENDBLOCKDATA) fail(MALS_IO_ERROR, "unexpected class annotation");
  if (in.u8() != TC_NULL) fail(MALS_IO_ERROR, "unexpected superclass descriptor");
  BlockIn b(in);
  const int32_t n_known = b.i32();
  if (n_known == -1) {
    m.known_null = true;
  } else {
    if (n_known < 0) fail(MALS_IO_ERROR, "knownItemIDs: negative count");
    for (int32_t u = 0; u < n_known; ++u) {
      m.known_user_ids.push_back(b.i64());
      b.i64s(m.known_item_ids, b.count("knownItemIDs"));
      m.known_ptr.push_back((int64_t)m.known_item_ids.size());
    }
  }
  bool have = false;
  read_matrix(b, m.user_ids, m.X, m.features, have, "X");
  read_matrix(b, m.item_ids, m.Y, m.features, have, "Y");
  b.i64s(m.item_tags, b.count("itemTagIDs"));
  b.i64s(m.user_tags, b.count("userTagIDs"));
  read_clusters(b, m.user_clusters, "userClusters");
  read_clusters(b, m.item_clusters, "itemClusters");
  // whatever a later version appended is skipped like ObjectInputStream.skipCustomData does
}

template <typename F>
int guarded(F&& f) {
  try {
    f();
    return MALS_OK;
  } catch (const Failure& e) {
    g_err = e.what;
    return e.status;
  } catch (const std::bad_alloc&) {
    g_err = "out of host memory";
    return MALS_OOM;
  } catch (const std::exception& e) {
    g_err = e.what();
    return MALS_IO_ERROR;
  }
}

}  // namespace

extern "C" {

int mals_model_write(const char* path, const mals_model_view* model) {
  return guarded([&] {
    if (!path || !model || model->struct_size != (int32_t)sizeof(mals_model_view)) fail(MALS_INVALID_ARG, "null argument or wrong struct_size");
    try {
      write_model(path, *model);
    } catch (...) {
      remove(path);  // never leave half a model behind
      throw;
    }
  });
}

int mals_model_read(const char* path, mals_model* out) {
  return guarded([&] {
    if (!path || !out) fail(MALS_INVALID_ARG, "null argument");
    *out = nullptr;
    auto* m = new mals_model_s;
    try {
      read_model(path, *m);
    } catch (...) {
      delete m;
      throw;
    }
    *out = m;
  });
}

int mals_model_get(mals_model m, mals_model_view* v) {
  return guarded([&] {
    if (!m || !v) fail(MALS_INVALID_ARG, "null argument");
    memset(v, 0, sizeof *v);
    v->struct_size = (int32_t)sizeof *v;
    v->features = m->features;
    v->n_users = (int64_t)m->user_ids.size(), v->user_ids = m->user_ids.data(), v->X = m->X.data();
    v->n_items = (int64_t)m->item_ids.size(), v->item_ids = m->item_ids.data(), v->Y = m->Y.data();
    v->n_known = m->known_null ? -1 : (int64_t)m->known_user_ids.size();
    v->known_user_ids = m->known_user_ids.data(), v->known_ptr = m->known_ptr.data(), v->known_item_ids = m->known_item_ids.data();
    v->n_item_tags = (int64_t)m->item_tags.size(), v->item_tag_ids = m->item_tags.data();
    v->n_user_tags = (int64_t)m->user_tags.size(), v->user_tag_ids = m->user_tags.data();
    const Clusters& uc = m->user_clusters;
    v->n_user_clusters = (int64_t)uc.member_ptr.size() - 1;
    v->user_cluster_member_ptr = uc.member_ptr.data(), v->user_cluster_members = uc.members.data();
    v->user_cluster_centroid_ptr = uc.centroid_ptr.data(), v->user_cluster_centroids = uc.centroids.data();
    const Clusters& ic = m->item_clusters;
    v->n_item_clusters = (int64_t)ic.member_ptr.size() - 1;
    v->item_cluster_member_ptr = ic.member_ptr.data(), v->item_cluster_members = ic.members.data();
    v->item_cluster_centroid_ptr = ic.centroid_ptr.data(), v->item_cluster_centroids = ic.centroids.data();
  });
}

int mals_model_destroy(mals_model m) {
  if (!m) return MALS_INVALID_ARG;
  delete m;
  return MALS_OK;
}

const char* mals_model_last_error(void) { return g_err.c_str(); }

}  // extern "C"
