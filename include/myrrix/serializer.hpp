// serializer.hpp -- C++ host-side mirror of the model file of the reference (SURVEY.md section 8(f)
// row 5) over the C-ABI (mals_model_*, csrc/model_io.cpp):
//   net.myrrix.online.generation.GenerationSerializer.readGeneration / writeGeneration
//                                   online-local/src/.../generation/GenerationSerializer.java:84-95
//   (stream layout :96-262; container IOUtils.writeObjectToFile / readObjectFromFile,
//    common/src/net/myrrix/common/io/IOUtils.java:259-283)
// Same names, argument meaning and error behaviour (java.io.IOException -> IOException,
// Preconditions.checkState -> IllegalStateException).  The seven constructor arguments of Generation
// that the file carries (GS:117-123) are held in the reference's own container shapes.
// Header-only, C++17.
#pragma once
#include <set>
#include <string>
#include <utility>

#include "factorizer.hpp"

namespace myrrix {

struct IOException : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct IllegalStateException : std::runtime_error {
  using std::runtime_error::runtime_error;
};

using FastIDSet = std::set<int64_t>;
struct IDCluster {  // online/src/net/myrrix/online/generation/IDCluster.java
  FastIDSet members;
  FloatVector centroid;
};

struct SerializedGeneration {
  bool hasKnownItemIDs = true;  // false: model.noKnownItems, the map is null (GS:131-133,148-149)
  FastByIDMap<FastIDSet> knownItemIDs;
  FastByIDMap<FloatVector> X, Y;
  FastIDSet itemTagIDs, userTagIDs;
  std::vector<IDCluster> userClusters, itemClusters;
};

class GenerationSerializer {
 public:
  static void writeGeneration(const SerializedGeneration& g, const std::string& f) {
    Flat fx = flatten(g.X, "X"), fy = flatten(g.Y, "Y");
    if (!g.X.empty() && !g.Y.empty() && fx.features != fy.features)
      throw IllegalStateException("X and Y differ in their number of features");
    std::vector<int64_t> ku, kp{0}, ki;
    for (const auto& e : g.knownItemIDs) {
      ku.push_back(e.first);
      ki.insert(ki.end(), e.second.begin(), e.second.end());
      kp.push_back((int64_t)ki.size());
    }
    const std::vector<int64_t> it(g.itemTagIDs.begin(), g.itemTagIDs.end()), ut(g.userTagIDs.begin(), g.userTagIDs.end());
    const Clusters uc = flatten(g.userClusters), ic = flatten(g.itemClusters);
    mals_model_view v{};
    v.struct_size = (int32_t)sizeof v;
    v.features = g.X.empty() ? fy.features : fx.features;
    v.n_users = (int64_t)fx.ids.size(), v.user_ids = fx.ids.data(), v.X = fx.rows.data();
    v.n_items = (int64_t)fy.ids.size(), v.item_ids = fy.ids.data(), v.Y = fy.rows.data();
    v.n_known = g.hasKnownItemIDs ? (int64_t)ku.size() : -1;
    v.known_user_ids = ku.data(), v.known_ptr = kp.data(), v.known_item_ids = ki.data();
    v.n_item_tags = (int64_t)it.size(), v.item_tag_ids = it.data();
    v.n_user_tags = (int64_t)ut.size(), v.user_tag_ids = ut.data();
    v.n_user_clusters = (int64_t)g.userClusters.size();
    v.user_cluster_member_ptr = uc.mptr.data(), v.user_cluster_members = uc.members.data();
    v.user_cluster_centroid_ptr = uc.cptr.data(), v.user_cluster_centroids = uc.centroids.data();
    v.n_item_clusters = (int64_t)g.itemClusters.size();
    v.item_cluster_member_ptr = ic.mptr.data(), v.item_cluster_members = ic.members.data();
    v.item_cluster_centroid_ptr = ic.cptr.data(), v.item_cluster_centroids = ic.centroids.data();
    check(mals_model_write(f.c_str(), &v));
  }

  static SerializedGeneration readGeneration(const std::string& f) {
    mals_model m = nullptr;
    check(mals_model_read(f.c_str(), &m));
    struct Guard {
      mals_model m;
      ~Guard() { mals_model_destroy(m); }
    } guard{m};
    mals_model_view v{};
    check(mals_model_get(m, &v));
    SerializedGeneration g;
    g.hasKnownItemIDs = v.n_known >= 0;
    for (int64_t u = 0; u < v.n_known; ++u)
      g.knownItemIDs[v.known_user_ids[u]] = FastIDSet(v.known_item_ids + v.known_ptr[u], v.known_item_ids + v.known_ptr[u + 1]);
    for (int64_t r = 0; r < v.n_users; ++r) g.X[v.user_ids[r]] = FloatVector(v.X + r * v.features, v.X + (r + 1) * v.features);
    for (int64_t r = 0; r < v.n_items; ++r) g.Y[v.item_ids[r]] = FloatVector(v.Y + r * v.features, v.Y + (r + 1) * v.features);
    g.itemTagIDs = FastIDSet(v.item_tag_ids, v.item_tag_ids + v.n_item_tags);
    g.userTagIDs = FastIDSet(v.user_tag_ids, v.user_tag_ids + v.n_user_tags);
    g.userClusters = clusters(v.n_user_clusters, v.user_cluster_member_ptr, v.user_cluster_members, v.user_cluster_centroid_ptr,
                              v.user_cluster_centroids);
    g.itemClusters = clusters(v.n_item_clusters, v.item_cluster_member_ptr, v.item_cluster_members, v.item_cluster_centroid_ptr,
                              v.item_cluster_centroids);
    return g;
  }

 private:
  struct Flat {
    std::vector<int64_t> ids;
    std::vector<float> rows;
    int32_t features = 0;
  };
  struct Clusters {
    std::vector<int64_t> mptr{0}, members, cptr{0};
    std::vector<float> centroids;
  };
  static Flat flatten(const FastByIDMap<FloatVector>& M, const char* what) {
    Flat f;
    for (const auto& e : M) {
      if (f.ids.empty()) f.features = (int32_t)e.second.size();
      if ((int32_t)e.second.size() != f.features) throw IllegalStateException(std::string(what) + ": rows of different lengths");
      f.ids.push_back(e.first);
      f.rows.insert(f.rows.end(), e.second.begin(), e.second.end());
    }
    return f;
  }
  static Clusters flatten(const std::vector<IDCluster>& cs) {
    Clusters c;
    for (const IDCluster& k : cs) {
      c.members.insert(c.members.end(), k.members.begin(), k.members.end());
      c.mptr.push_back((int64_t)c.members.size());
      c.centroids.insert(c.centroids.end(), k.centroid.begin(), k.centroid.end());
      c.cptr.push_back((int64_t)c.centroids.size());
    }
    return c;
  }
  static std::vector<IDCluster> clusters(int64_t n, const int64_t* mptr, const int64_t* members, const int64_t* cptr, const float* cent) {
    std::vector<IDCluster> out;
    for (int64_t c = 0; c < n; ++c)
      out.push_back(IDCluster{FastIDSet(members + mptr[c], members + mptr[c + 1]), FloatVector(cent + cptr[c], cent + cptr[c + 1])});
    return out;
  }
  static void check(int rc) {
    if (rc == MALS_OK) return;
    const std::string msg = mals_model_last_error();
    if (rc == MALS_IO_ERROR) throw IOException(msg);
    throw IllegalStateException(msg);
  }
};

}  // namespace myrrix
