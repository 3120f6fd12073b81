// mals_group.cpp -- the multi-GPU half-iteration of include/myrrix_als.h, built
// entirely on the per-GPU C-ABI of mals_api.hip plus streams, events and the exchange.
//
// The reference runs one process with N worker threads, every output row written by exactly one of
// them (ALS:391-410, ALS:497-499); here the workers are GPUs, a row belongs to the rank whose slice
// holds it, and the only communication per half-iteration is (1) the k x k fp64 sum of the partial
// Gramians and (2) the freshly solved rows of every slice into every replica -- an all-gather with
// per-rank counts, issued chunk by chunk behind the solve.
//
// RCCL is resolved with dlopen at group creation (no link-time dependency for single-GPU users; inside
// a process that already loaded PyTorch's copy the same library is reused).  Collectives of several
// local members are always enclosed in ncclGroupStart/End (one thread drives all devices), and every RCCL call
// of a rank is issued on that rank's comm stream -- a communicator is never used from two streams at once.
#include "../../include/myrrix_als.h"
#include "mals_internal.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  std::string forced;  // mals_group_use_transport: this library and no other

  bool load(std::string& err) {
    if (lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // No environment variable selects the library: what gets loaded as "RCCL" into a server process is decided by the
    // process itself (mals_group_use_transport: a particular RCCL build, or the tests' stand-in transport).
    if (!forced.empty()) {
      lib = dlopen(forced.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (!lib) {
        const char* e = dlerror();  // once: dlerror() clears its state
        err = "transport library " + forced + " could not be loaded: " + (e ? e : "");
        return false;
      }
    }
    for (const char* n : names) {
      if (lib) break;
      lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // a copy the process already mapped (e.g. PyTorch's) first
    }
    std::string first_error;
    for (const char* n : names) {
      if (lib) break;
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (!lib && first_error.empty()) {  // right behind the failing dlopen: later attempts overwrite the message
        const char* e = dlerror();
        first_error = e ? e : "";
      }
    }
    if (!lib) {
      err = "librccl.so.1 could not be loaded: " + first_error;
      return false;
    }
#define MALS_SYM(field, name)                                                  \
  field = reinterpret_cast<decltype(field)>(dlsym(lib, name));                 \
  if (!field) {                                                                \
    err = std::string("RCCL symbol missing: ") + name;                         \
    dlclose(lib);                                                              \
    lib = nullptr;                                                             \
    return false;                                                              \
  }
    MALS_SYM(GetUniqueId, "ncclGetUniqueId")
    MALS_SYM(CommInitRank, "ncclCommInitRank")
    MALS_SYM(CommInitAll, "ncclCommInitAll")
    MALS_SYM(CommDestroy, "ncclCommDestroy")
    MALS_SYM(AllReduce, "ncclAllReduce")
    MALS_SYM(Send, "ncclSend")
    MALS_SYM(Recv, "ncclRecv")
    MALS_SYM(GroupStart, "ncclGroupStart")
    MALS_SYM(GroupEnd, "ncclGroupEnd")
    MALS_SYM(GetErrorString, "ncclGetErrorString")
    MALS_SYM(CommCount, "ncclCommCount")
    MALS_SYM(CommUserRank, "ncclCommUserRank")
    MALS_SYM(CommCuDevice, "ncclCommCuDevice")
#undef MALS_SYM
    return true;
  }
};

Rccl g_rccl;

struct Member {
  mals_handle h = nullptr;
  int device = 0;
  int rank = 0;
  hipStream_t compute = nullptr, comm = nullptr;
  hipStream_t compute2 = nullptr;   // odd chunks of a half-iteration (alternate_streams)
  hipEvent_t ev_solved = nullptr, ev_exchanged = nullptr;
  hipEvent_t ev_half = nullptr, ev_join = nullptr;
  ncclComm_t nccl = nullptr;
  double* d_gp = nullptr;    // k*k: partial Gramian / all-reduce buffer
  double* d_stat = nullptr;  // 4 doubles: value statistics, status
  float* d_ymax = nullptr;   // max |element| of the rows of this member's partial Gramian (bit pattern = non-negative float)
  float* F[2] = {nullptr, nullptr};
  int64_t* d_row_ptr[2] = {nullptr, nullptr};  // rebased row pointers of borrowed device matrices
  // mals_ingest_install_group: the member's slices when the ingest lives on ANOTHER device (peer copies), and its slice of
  // knownItemIDs (rebased offsets always; the indices only when copied)
  int32_t* own_col[2] = {nullptr, nullptr};
  float* own_val[2] = {nullptr, nullptr};
  int64_t* own_known_ptr = nullptr;
  int32_t* own_known_idx = nullptr;
  int64_t* own_tag_idx = nullptr;
  // chunked upload
  int64_t up_rows = 0;
};

}  // namespace

struct mals_group_s {
  mals_config cfg;
  int world = 1;
  int backend = MALS_GROUP_RCCL;
  bool single_process = true;
  std::vector<Member> m;
  int64_t n_total[2] = {0, 0};
  int64_t n_rows[2] = {0, 0};
  std::vector<int64_t> bounds[2];
  std::vector<int64_t> up_row_ptr[2];  // chunked upload: the full row_ptr
  int64_t up_next_row[2] = {0, 0};
  int exchange_chunks = 4;        // what the NEXT matrix upload of a side is cut into
  int side_chunks[2] = {4, 4};    // what each side's CURRENT matrix was cut into (plan_side): the members' work lists are
                                  // built for this count, so the solve loop, the exchange ranges and the members agree
  // Consecutive chunks of a half-iteration on two alternating compute streams: every chunk boundary otherwise drains the
  // persistent kernels (rows, long rows, dual classes each have a tail) -- +3.7 % at 4 chunks on C4 (round 3).  A chunk's
  // completion event goes to the exchange stream as before; the streams are joined before anything that assumes one.
  bool alternate_streams = true;
  std::atomic<int> cancelled{0};
  mals_iteration_fn iter_fn = nullptr;   // mals_group_set_iteration_callback
  void* iter_user = nullptr;
  std::string err;
};

namespace {

int gfail(mals_group g, int code, const std::string& msg) {
  if (g) g->err = msg;
  return code;
}

#define GHIP(g, call)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (call);                                                                             \
    if (_e != hipSuccess) return gfail(g, _e == hipErrorOutOfMemory ? MALS_OOM : MALS_HIP_ERROR,        \
                                       std::string(#call) + ": " + hipGetErrorString(_e));              \
  } while (0)

#define GNCCL(g, call)                                                                                  \
  do {                                                                                                  \
    ncclResult_t _r = (call);                                                                           \
    if (_r != ncclSuccess) return gfail(g, MALS_COMM_ERROR, std::string(#call) + ": " + g_rccl.GetErrorString(_r)); \
  } while (0)

// a member call failed: keep its message
int mfail(mals_group g, const Member& mb, int rc) {
  return gfail(g, rc, std::string("rank ") + std::to_string(mb.rank) + ": " + mals_last_error(mb.h));
}

#define GSIDE(g, side)                                                                    \
  do {                                                                                    \
    if (!(g)) return MALS_INVALID_ARG;                                                    \
    if ((side) != MALS_SIDE_X && (side) != MALS_SIDE_Y) return gfail(g, MALS_INVALID_ARG, "side must be MALS_SIDE_X or _Y"); \
  } while (0)

int64_t chunk_rows_of(const mals_group g, int side, int rank) {
  const int64_t n = g->bounds[side][(size_t)rank + 1] - g->bounds[side][(size_t)rank];
  return std::max<int64_t>(1, (n + g->side_chunks[side] - 1) / g->side_chunks[side]);
}

// rows [lo, hi) (global) of chunk c of rank's slice
void chunk_range(const mals_group g, int side, int rank, int c, int64_t* lo, int64_t* hi) {
  const int64_t b0 = g->bounds[side][(size_t)rank], b1 = g->bounds[side][(size_t)rank + 1];
  const int64_t cr = chunk_rows_of(g, side, rank);
  *lo = std::min(b1, b0 + (int64_t)c * cr);
  *hi = std::min(b1, b0 + (int64_t)(c + 1) * cr);
}

int init_member(mals_group g, Member& mb, const mals_config& cfg, int device, int rank) {
  mb.device = -1;   // until the handle exists: destroy_member must not select a device that may not be there
  mb.rank = rank;
  mals_config c = cfg;
  c.device = device;
  if (int rc = mals_create(&c, &mb.h)) {
    char why[512];
    (void)mals_create_error(why, sizeof(why));
    return gfail(g, rc, std::string(why[0] ? why : "mals_create failed") + " (group member " + std::to_string(rank) + ", device " + std::to_string(device) + ")");
  }
  mb.device = device;
  GHIP(g, hipSetDevice(device));
  GHIP(g, hipStreamCreateWithFlags(&mb.compute, hipStreamNonBlocking));
  GHIP(g, hipStreamCreateWithFlags(&mb.comm, hipStreamNonBlocking));
  GHIP(g, hipStreamCreateWithFlags(&mb.compute2, hipStreamNonBlocking));
  GHIP(g, hipEventCreateWithFlags(&mb.ev_half, hipEventDisableTiming));
  GHIP(g, hipEventCreateWithFlags(&mb.ev_join, hipEventDisableTiming));
  GHIP(g, hipEventCreateWithFlags(&mb.ev_solved, hipEventDisableTiming));
  GHIP(g, hipEventCreateWithFlags(&mb.ev_exchanged, hipEventDisableTiming));
  GHIP(g, hipEventRecord(mb.ev_exchanged, mb.comm));
  const size_t kk = (size_t)cfg.features * cfg.features;
  GHIP(g, hipMalloc(&mb.d_gp, sizeof(double) * kk));
  GHIP(g, hipMalloc(&mb.d_stat, sizeof(double) * 4));
  GHIP(g, hipMalloc(&mb.d_ymax, sizeof(float) * (size_t)malsi_ymax_slots()));
  if (int rc = mals_set_stream(mb.h, mb.compute)) return mfail(g, mb, rc);
  return MALS_OK;
}

void destroy_member(Member& mb) {
  if (mb.device >= 0) (void)hipSetDevice(mb.device);
  if (mb.compute) (void)hipStreamSynchronize(mb.compute);
  if (mb.compute2) (void)hipStreamSynchronize(mb.compute2);
  if (mb.comm) (void)hipStreamSynchronize(mb.comm);
  if (mb.nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(mb.nccl);
  if (mb.h) (void)mals_destroy(mb.h);
  for (int sd = 0; sd < 2; ++sd) {
    if (mb.d_row_ptr[sd]) (void)hipFree(mb.d_row_ptr[sd]);
    if (mb.own_col[sd]) (void)hipFree(mb.own_col[sd]);
    if (mb.own_val[sd]) (void)hipFree(mb.own_val[sd]);
  }
  if (mb.own_known_ptr) (void)hipFree(mb.own_known_ptr);
  if (mb.own_known_idx) (void)hipFree(mb.own_known_idx);
  if (mb.own_tag_idx) (void)hipFree(mb.own_tag_idx);
  if (mb.d_gp) (void)hipFree(mb.d_gp);
  if (mb.d_stat) (void)hipFree(mb.d_stat);
  if (mb.d_ymax) (void)hipFree(mb.d_ymax);
  if (mb.ev_solved) (void)hipEventDestroy(mb.ev_solved);
  if (mb.ev_half) (void)hipEventDestroy(mb.ev_half);
  if (mb.ev_join) (void)hipEventDestroy(mb.ev_join);
  if (mb.compute2) (void)hipStreamDestroy(mb.compute2);
  if (mb.ev_exchanged) (void)hipEventDestroy(mb.ev_exchanged);
  if (mb.comm) (void)hipStreamDestroy(mb.comm);
  if (mb.compute) (void)hipStreamDestroy(mb.compute);
  mb = Member();
}

// out[i] = op over all ranks of the members' d_stat[0..n) (op: 0 = sum, 1 = max); result in every d_stat
int allreduce_stat(mals_group g, int n, int op) {
  if (g->world == 1 && !g->m[0].nccl) return MALS_OK;
  if (g->backend == MALS_GROUP_PEER_COPY) {
    std::vector<double> acc((size_t)n, op ? -std::numeric_limits<double>::infinity() : 0.0), tmp((size_t)n);
    for (Member& mb : g->m) {
      GHIP(g, hipSetDevice(mb.device));
      GHIP(g, hipMemcpyAsync(tmp.data(), mb.d_stat, sizeof(double) * n, hipMemcpyDeviceToHost, mb.compute));
      GHIP(g, hipStreamSynchronize(mb.compute));
      for (int i = 0; i < n; ++i) acc[(size_t)i] = op ? std::max(acc[(size_t)i], tmp[(size_t)i]) : acc[(size_t)i] + tmp[(size_t)i];
    }
    for (Member& mb : g->m) {
      GHIP(g, hipSetDevice(mb.device));
      GHIP(g, hipMemcpyAsync(mb.d_stat, acc.data(), sizeof(double) * n, hipMemcpyHostToDevice, mb.compute));
      GHIP(g, hipStreamSynchronize(mb.compute));
    }
    return MALS_OK;
  }
  // every RCCL call of a rank goes to ITS comm stream: one communicator is never used from two streams at once
  GNCCL(g, g_rccl.GroupStart());
  for (Member& mb : g->m) {
    const ncclResult_t r = g_rccl.AllReduce(mb.d_stat, mb.d_stat, (size_t)n, ncclDouble, op ? ncclMax : ncclSum, mb.nccl, mb.comm);
    if (r != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return gfail(g, MALS_COMM_ERROR, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
    }
  }
  GNCCL(g, g_rccl.GroupEnd());
  for (Member& mb : g->m) {
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipStreamSynchronize(mb.comm));
  }
  return MALS_OK;
}

// every member's d_stat[0..n) <- host values (one per member), then all-reduce, then read back (member 0)
int allreduce_host(mals_group g, const std::vector<std::vector<double>>& per_member, int n, int op, double* out) {
  for (size_t i = 0; i < g->m.size(); ++i) {
    Member& mb = g->m[i];
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipMemcpyAsync(mb.d_stat, per_member[i].data(), sizeof(double) * n, hipMemcpyHostToDevice, mb.compute));
    GHIP(g, hipStreamSynchronize(mb.compute));  // the source is a pageable temporary
  }
  if (int rc = allreduce_stat(g, n, op)) return rc;
  Member& m0 = g->m[0];
  GHIP(g, hipSetDevice(m0.device));
  GHIP(g, hipMemcpyAsync(out, m0.d_stat, sizeof(double) * n, hipMemcpyDeviceToHost, m0.compute));
  GHIP(g, hipStreamSynchronize(m0.compute));
  return MALS_OK;
}

// A status every rank agrees on: the largest code any rank saw (collective in multi-process groups).
int agree_status(mals_group g, int local_rc, const std::string& local_msg) {
  if (g->single_process) {
    if (local_rc != MALS_OK) g->err = local_msg;
    return local_rc;
  }
  std::vector<std::vector<double>> v(g->m.size(), std::vector<double>(1, (double)local_rc));
  double worst = 0.0;
  if (int rc = allreduce_host(g, v, 1, 1, &worst)) return rc;
  const int agreed = (int)worst;
  if (agreed != MALS_OK) g->err = local_rc != MALS_OK ? local_msg : "another rank reported status " + std::to_string(agreed);
  return agreed;
}

int sync_value_stats(mals_group g, int side) {
  std::vector<std::vector<double>> mx(g->m.size(), std::vector<double>(1, 0.0)), sm(g->m.size(), std::vector<double>(2, 0.0));
  for (size_t i = 0; i < g->m.size(); ++i) {
    float m = 0.f;
    double s = 0.0;
    int64_t n = 0;
    if (int rc = mals_get_value_stats(g->m[i].h, side, &m, &s, &n)) return mfail(g, g->m[i], rc);
    mx[i][0] = (double)m;
    sm[i][0] = s;
    sm[i][1] = (double)n;
  }
  double vmax = 0.0, vs[2] = {0.0, 0.0};
  if (int rc = allreduce_host(g, mx, 1, 1, &vmax)) return rc;
  if (int rc = allreduce_host(g, sm, 2, 0, vs)) return rc;
  for (Member& mb : g->m)
    if (int rc = mals_set_value_stats(mb.h, side, (float)vmax, vs[1] > 0.0 ? vs[0] / vs[1] : 0.0)) return mfail(g, mb, rc);
  return MALS_OK;
}

int refresh_replica_ptrs(mals_group g, int side) {
  for (Member& mb : g->m) {
    void* p = nullptr;
    int64_t n = 0;
    if (int rc = mals_factor_device_ptr(mb.h, side, &p, &n)) return mfail(g, mb, rc);
    mb.F[side] = static_cast<float*>(p);
  }
  return MALS_OK;
}

// slices planned, every member told its chunking; called with the full row_ptr on the host
int plan_side(mals_group g, int side, const int64_t* row_ptr, int64_t n_rows) {
  g->n_rows[side] = n_rows;
  g->side_chunks[side] = g->exchange_chunks;
  g->bounds[side].assign((size_t)g->world + 1, 0);
  if (int rc = mals_plan_shards(row_ptr, n_rows, g->world, -1.0, g->cfg.features, g->bounds[side].data()))
    return gfail(g, rc, "mals_plan_shards failed");
  for (Member& mb : g->m)
    if (int rc = mals_set_chunk_rows(mb.h, side, chunk_rows_of(g, side, mb.rank))) return mfail(g, mb, rc);
  return MALS_OK;
}

// The freshly solved rows of chunk c of every slice into every replica (comm streams).
int exchange_chunk(mals_group g, int side, int c) {
  const int k = g->cfg.features;
  if (g->world == 1) return MALS_OK;
  if (g->backend == MALS_GROUP_PEER_COPY) {
    for (Member& src : g->m) {
      int64_t lo, hi;
      chunk_range(g, side, src.rank, c, &lo, &hi);
      if (hi <= lo) continue;
      GHIP(g, hipSetDevice(src.device));
      for (Member& dst : g->m) {
        if (dst.rank == src.rank) continue;
        GHIP(g, hipMemcpyPeerAsync(dst.F[side] + lo * k, dst.device, src.F[side] + lo * k, src.device, sizeof(float) * (size_t)(hi - lo) * k,
                                   src.comm));
      }
    }
    return MALS_OK;
  }
  GNCCL(g, g_rccl.GroupStart());
  for (Member& mb : g->m) {
    int64_t lo, hi;
    chunk_range(g, side, mb.rank, c, &lo, &hi);
    for (int q = 0; q < g->world; ++q) {
      if (q == mb.rank) continue;
      ncclResult_t r = ncclSuccess;
      if (hi > lo) r = g_rccl.Send(mb.F[side] + lo * k, (size_t)(hi - lo) * k, ncclFloat, q, mb.nccl, mb.comm);
      int64_t qlo, qhi;
      chunk_range(g, side, q, c, &qlo, &qhi);
      if (r == ncclSuccess && qhi > qlo) r = g_rccl.Recv(mb.F[side] + qlo * k, (size_t)(qhi - qlo) * k, ncclFloat, q, mb.nccl, mb.comm);
      if (r != ncclSuccess) {
        (void)g_rccl.GroupEnd();
        return gfail(g, MALS_COMM_ERROR, std::string("ncclSend/ncclRecv: ") + g_rccl.GetErrorString(r));
      }
    }
  }
  GNCCL(g, g_rccl.GroupEnd());
  return MALS_OK;
}

// G = sum over ranks of the partial Gramians of `side`'s replica, installed on every member.  Every
// replica is complete when this runs (the caller has waited for the exchange), so the rows -- stale Y rows
// behind the matrix included (ALS:304-308) -- are simply cut into `world` equal ranges: the work of M^T M
// is proportional to rows, not to entries.
// A LOCAL failure of a member's partial Gramian or of installing the sum (MALS_OOM ...) does not leave the collective:
// the member contributes zeros, every rank's all-reduces stay matched, and the failure is handed back through
// local_rc / local_msg for the caller to skip the solves and agree on the status at the end of the half-iteration.
// The return value is for failures that end the half-iteration on the spot (communication, device).
int group_gramian(mals_group g, int side, int* local_rc, std::string* local_msg) {
  auto member_failed = [&](const Member& mb, int rc) {
    if (*local_rc == MALS_OK) {
      *local_rc = rc;
      *local_msg = std::string("rank ") + std::to_string(mb.rank) + ": " + mals_last_error(mb.h);
    }
  };
  const int k = g->cfg.features;
  const size_t kk = (size_t)k * k;
  const int64_t per = (g->n_total[side] + g->world - 1) / g->world;
  for (Member& mb : g->m) {
    GHIP(g, hipSetDevice(mb.device));
    const int64_t r0 = std::min(g->n_total[side], per * mb.rank);
    const int64_t r1 = std::min(g->n_total[side], per * (mb.rank + 1));
    // the kernels that form the partial Gramian also record the largest |element| of their rows: summed Gramian + the
    // maximum over all ranks give every member the exact operand bound of the split-precision gather (instead of
    // sqrt(max_f G_ff), which at 1e8 rows is 13 binades loose)
    GHIP(g, hipMemsetAsync(mb.d_ymax, 0, sizeof(float) * (size_t)malsi_ymax_slots(), mb.compute));
    int prc = MALS_OK;
    if (r1 > r0) {
      prc = malsi_gramian_partial(mb.h, side, r0, r1 - r0, mb.d_gp, reinterpret_cast<unsigned*>(mb.d_ymax));
      if (prc == MALS_HIP_ERROR) return mfail(g, mb, prc);
      if (prc != MALS_OK) member_failed(mb, prc);
    }
    if (r1 <= r0 || prc != MALS_OK) GHIP(g, hipMemsetAsync(mb.d_gp, 0, sizeof(double) * kk, mb.compute));
  }
  if (g->world > 1 || g->m[0].nccl) {
    if (g->backend == MALS_GROUP_PEER_COPY) {  // fixed summation order (rank 0, 1, ...): deterministic
      std::vector<double> acc(kk, 0.0), tmp(kk);
      const size_t ns = (size_t)malsi_ymax_slots();
      std::vector<float> ymax(ns, 0.f), ytmp(ns, 0.f);
      for (Member& mb : g->m) {
        GHIP(g, hipSetDevice(mb.device));
        GHIP(g, hipMemcpyAsync(tmp.data(), mb.d_gp, sizeof(double) * kk, hipMemcpyDeviceToHost, mb.compute));
        GHIP(g, hipMemcpyAsync(ytmp.data(), mb.d_ymax, sizeof(float) * ns, hipMemcpyDeviceToHost, mb.compute));
        GHIP(g, hipStreamSynchronize(mb.compute));
        for (size_t i = 0; i < kk; ++i) acc[i] += tmp[i];
        for (size_t i = 0; i < ns; ++i)
          if (!(ytmp[i] <= ymax[i])) ymax[i] = ytmp[i];   // an inf pattern wins and makes the consumer fall back
      }
      for (Member& mb : g->m) {
        GHIP(g, hipSetDevice(mb.device));
        GHIP(g, hipMemcpyAsync(mb.d_gp, acc.data(), sizeof(double) * kk, hipMemcpyHostToDevice, mb.compute));
        GHIP(g, hipMemcpyAsync(mb.d_ymax, ymax.data(), sizeof(float) * ns, hipMemcpyHostToDevice, mb.compute));
        GHIP(g, hipStreamSynchronize(mb.compute));
      }
    } else {
      // partial Gramian (compute stream) -> all-reduce (comm stream, behind whatever exchange is still on it) -> compute
      for (Member& mb : g->m) {
        GHIP(g, hipSetDevice(mb.device));
        GHIP(g, hipEventRecord(mb.ev_solved, mb.compute));
        GHIP(g, hipStreamWaitEvent(mb.comm, mb.ev_solved, 0));
      }
      GNCCL(g, g_rccl.GroupStart());
      for (Member& mb : g->m) {
        ncclResult_t r = g_rccl.AllReduce(mb.d_gp, mb.d_gp, kk, ncclDouble, ncclSum, mb.nccl, mb.comm);
        if (r == ncclSuccess) r = g_rccl.AllReduce(mb.d_ymax, mb.d_ymax, (size_t)malsi_ymax_slots(), ncclFloat, ncclMax, mb.nccl, mb.comm);
        if (r != ncclSuccess) {
          (void)g_rccl.GroupEnd();
          return gfail(g, MALS_COMM_ERROR, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
        }
      }
      GNCCL(g, g_rccl.GroupEnd());
      for (Member& mb : g->m) {
        GHIP(g, hipSetDevice(mb.device));
        GHIP(g, hipEventRecord(mb.ev_exchanged, mb.comm));
        GHIP(g, hipStreamWaitEvent(mb.compute, mb.ev_exchanged, 0));
      }
    }
  }
  for (Member& mb : g->m) {
    GHIP(g, hipSetDevice(mb.device));
    if (int rc = malsi_set_gramian(mb.h, side, mb.d_gp, MALS_MEM_DEVICE, reinterpret_cast<const unsigned*>(mb.d_ymax))) {
      if (rc == MALS_HIP_ERROR) return mfail(g, mb, rc);
      member_failed(mb, rc);
    }
  }
  return MALS_OK;
}

int finish_matrix(mals_group g, int side) {
  if (int rc = sync_value_stats(g, side)) return rc;
  return MALS_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int mals_plan_shards(const int64_t* row_ptr, int64_t n_rows, int32_t world, double row_cost, int32_t features, int64_t* bounds_out) {
  if (!row_ptr || !bounds_out || n_rows < 0 || world <= 0) return MALS_INVALID_ARG;
  if (row_cost < 0.0) row_cost = features > 0 ? (double)features * features / 200.0 : 0.0;
  const double total = (double)(row_ptr[n_rows] - row_ptr[0]) + row_cost * (double)n_rows;
  bounds_out[0] = 0;
  int64_t r = 0;
  for (int j = 1; j < world; ++j) {
    const double target = total * (double)j / (double)world;
    // first r with cost(rows [0, r)) >= target, then the closer of r-1 and r
    auto cost = [&](int64_t rr) { return (double)(row_ptr[rr] - row_ptr[0]) + row_cost * (double)rr; };
    int64_t lo = r, hi = n_rows;
    while (lo < hi) {
      const int64_t mid = lo + (hi - lo) / 2;
      if (cost(mid) >= target) hi = mid; else lo = mid + 1;
    }
    int64_t b = lo;
    if (b > r && target - cost(b - 1) < cost(b) - target) b = b - 1;
    bounds_out[j] = std::max(b, r);
    r = bounds_out[j];
  }
  bounds_out[world] = n_rows;
  return MALS_OK;
}

int mals_group_create(const mals_config* cfg, const int32_t* devices, int32_t n_devices, int32_t backend, mals_group* out) {
  if (!cfg || !devices || !out || n_devices <= 0) {
    malsi_set_create_error("mals_group_create: null argument or empty device list");
    return MALS_INVALID_ARG;
  }
  *out = nullptr;
  if (backend != MALS_GROUP_RCCL && backend != MALS_GROUP_PEER_COPY) {
    malsi_set_create_error("mals_group_create: backend must be MALS_GROUP_RCCL or MALS_GROUP_PEER_COPY");
    return MALS_INVALID_ARG;
  }
  mals_group g = new (std::nothrow) mals_group_s();
  if (!g) {
    malsi_set_create_error("mals_group_create: out of host memory");
    return MALS_OOM;
  }
  g->cfg = *cfg;
  g->world = n_devices;
  g->backend = backend;
  g->single_process = true;
  g->m.resize((size_t)n_devices);
  int rc = MALS_OK;
  for (int i = 0; i < n_devices && rc == MALS_OK; ++i) rc = init_member(g, g->m[(size_t)i], *cfg, devices[i], i);
  if (rc == MALS_OK && backend == MALS_GROUP_RCCL && n_devices > 1) {
    std::string e;
    if (!g_rccl.load(e)) {
      rc = gfail(g, MALS_COMM_ERROR, e);
    } else {
      std::vector<ncclComm_t> comms((size_t)n_devices);
      std::vector<int> devs(devices, devices + n_devices);
      const ncclResult_t r = g_rccl.CommInitAll(comms.data(), n_devices, devs.data());
      if (r != ncclSuccess) {
        rc = gfail(g, MALS_COMM_ERROR, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
      } else {
        for (int i = 0; i < n_devices; ++i) g->m[(size_t)i].nccl = comms[(size_t)i];
      }
    }
  }
  if (rc == MALS_OK && backend == MALS_GROUP_PEER_COPY) {
    for (Member& a : g->m)
      for (Member& b : g->m) {
        if (a.device == b.device) continue;
        (void)hipSetDevice(a.device);
        const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();  // copies then stage through the host
      }
  }
  if (rc != MALS_OK) {
    std::fprintf(stderr, "mals_group_create: %s\n", g->err.c_str());
    malsi_set_create_error(("mals_group_create: " + g->err).c_str());
    for (Member& mb : g->m) destroy_member(mb);
    delete g;
    (void)hipGetLastError();   // a failed create leaves no sticky error behind for the next group of this process
    return rc;
  }
  malsi_set_create_error("");
  *out = g;
  return MALS_OK;
}

int mals_group_unique_id(void* id_out_128_bytes) {
  if (!id_out_128_bytes) return MALS_INVALID_ARG;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  std::string e;
  if (!g_rccl.load(e)) {
    malsi_set_create_error(("mals_group_unique_id: " + e).c_str());
    return MALS_COMM_ERROR;
  }
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) {
    malsi_set_create_error("mals_group_unique_id: ncclGetUniqueId failed");
    return MALS_COMM_ERROR;
  }
  std::memcpy(id_out_128_bytes, &id, sizeof(id));
  return MALS_OK;
}

int mals_group_create_rank(const mals_config* cfg, int32_t world, int32_t rank, const void* id_128_bytes, mals_group* out) {
  if (!cfg || !out || world <= 0 || rank < 0 || rank >= world || (world > 1 && !id_128_bytes)) {
    malsi_set_create_error("mals_group_create_rank: null argument, rank outside 0..world-1, or no unique id for world > 1");
    return MALS_INVALID_ARG;
  }
  *out = nullptr;
  mals_group g = new (std::nothrow) mals_group_s();
  if (!g) {
    malsi_set_create_error("mals_group_create_rank: out of host memory");
    return MALS_OOM;
  }
  g->cfg = *cfg;
  g->world = world;
  g->backend = MALS_GROUP_RCCL;
  g->single_process = world == 1;
  g->m.resize(1);
  int rc = init_member(g, g->m[0], *cfg, cfg->device, rank);
  if (rc == MALS_OK && id_128_bytes) {  // also for world = 1 when an id is given: a real one-rank communicator
    std::string e;
    if (!g_rccl.load(e)) {
      rc = gfail(g, MALS_COMM_ERROR, e);
    } else {
      ncclUniqueId id;
      std::memcpy(&id, id_128_bytes, sizeof(id));
      (void)hipSetDevice(cfg->device);
      const ncclResult_t r = g_rccl.CommInitRank(&g->m[0].nccl, world, id, rank);
      if (r != ncclSuccess) rc = gfail(g, MALS_COMM_ERROR, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
    }
  }
  if (rc != MALS_OK) {
    std::fprintf(stderr, "mals_group_create_rank: %s\n", g->err.c_str());
    malsi_set_create_error(("mals_group_create_rank: " + g->err).c_str());
    destroy_member(g->m[0]);
    delete g;
    (void)hipGetLastError();
    return rc;
  }
  malsi_set_create_error("");
  *out = g;
  return MALS_OK;
}

int mals_group_destroy(mals_group g) {
  if (!g) return MALS_INVALID_ARG;
  for (Member& mb : g->m) destroy_member(mb);
  delete g;
  return MALS_OK;
}

int mals_group_set_alternate_streams(mals_group g, int32_t on) {
  if (!g) return MALS_INVALID_ARG;
  g->alternate_streams = on != 0;
  return MALS_OK;
}

int mals_group_set_iteration_callback(mals_group g, mals_iteration_fn fn, void* user) {
  if (!g) return MALS_INVALID_ARG;
  g->iter_fn = fn;
  g->iter_user = user;
  return MALS_OK;
}

const char* mals_group_last_error(mals_group g) { return g ? g->err.c_str() : "null group"; }

int mals_group_world(mals_group g) { return g ? g->world : 0; }

int mals_group_features(mals_group g) { return g ? g->cfg.features : 0; }

int mals_group_pending_entries(mals_group g, int side, int64_t n_rows, int64_t* n_entries_out) {
  GSIDE(g, side);
  const std::vector<int64_t>& rp = g->up_row_ptr[side];
  if (rp.empty() || !n_entries_out) return gfail(g, MALS_INVALID_ARG, "no chunked upload in progress");
  const int64_t a = g->up_next_row[side], b = a + n_rows;
  if (n_rows < 0 || b > g->n_rows[side]) return gfail(g, MALS_INVALID_ARG, "piece exceeds the declared matrix");
  *n_entries_out = rp[(size_t)b] - rp[(size_t)a];
  return MALS_OK;
}

int mals_group_local(mals_group g, int32_t i, mals_handle* handle_out, int32_t* rank_out) {
  if (!g) return MALS_INVALID_ARG;
  if (i < 0 || (size_t)i >= g->m.size()) return gfail(g, MALS_INVALID_ARG, "no such local member");
  if (handle_out) *handle_out = g->m[(size_t)i].h;
  if (rank_out) *rank_out = g->m[(size_t)i].rank;
  return MALS_OK;
}

int mals_group_set_exchange_chunks(mals_group g, int32_t n_chunks) {
  if (!g || n_chunks <= 0) return MALS_INVALID_ARG;
  // takes effect with the next matrix upload of each side: a side that already holds a matrix keeps the count its
  // members' work lists were built for (side_chunks), so a later call can never leave part of a slice unsolved
  g->exchange_chunks = n_chunks;
  return MALS_OK;
}

int mals_group_use_transport(const char* library_path) {
  if (g_rccl.lib) return MALS_INVALID_ARG;  // already loaded: the choice is made once per process, before the first group
  g_rccl.forced = library_path ? library_path : "";
  return MALS_OK;
}

int mals_group_comm_info(mals_group g, int32_t local_member, int32_t* comm_size, int32_t* comm_rank, int32_t* hip_device,
                         char* pci_bus_id, int32_t pci_len) {
  if (!g) return MALS_INVALID_ARG;
  if (local_member < 0 || (size_t)local_member >= g->m.size()) return gfail(g, MALS_INVALID_ARG, "no such local member");
  const Member& mb = g->m[(size_t)local_member];
  int n = 0, r = -1, dev = mb.device;
  if (mb.nccl) {  // read back from the communicator itself, not from what this library was told
    GNCCL(g, g_rccl.CommCount(mb.nccl, &n));
    GNCCL(g, g_rccl.CommUserRank(mb.nccl, &r));
    GNCCL(g, g_rccl.CommCuDevice(mb.nccl, &dev));
  }
  if (comm_size) *comm_size = n;
  if (comm_rank) *comm_rank = r;
  if (hip_device) *hip_device = dev;
  if (pci_bus_id && pci_len > 0) {
    pci_bus_id[0] = 0;
    if (hipDeviceGetPCIBusId(pci_bus_id, pci_len, mb.device) != hipSuccess) {
      (void)hipGetLastError();
      pci_bus_id[0] = 0;
    }
  }
  return MALS_OK;
}

int mals_group_set_refine_limit(mals_group g, double limit) {
  if (!g) return MALS_INVALID_ARG;
  for (Member& mb : g->m)
    if (int rc = mals_set_refine_limit(mb.h, limit)) return mfail(g, mb, rc);
  return MALS_OK;
}

int mals_group_set_factor_rows(mals_group g, int side, int64_t n_rows_total) {
  GSIDE(g, side);
  for (Member& mb : g->m)
    if (int rc = mals_set_factor_rows(mb.h, side, n_rows_total)) return mfail(g, mb, rc);
  g->n_total[side] = n_rows_total;
  return refresh_replica_ptrs(g, side);
}

int mals_group_set_factors(mals_group g, int side, int64_t row_begin, int64_t n_rows, const float* host_rows) {
  GSIDE(g, side);
  for (Member& mb : g->m)
    if (int rc = mals_set_factors(mb.h, side, row_begin, n_rows, host_rows)) return mfail(g, mb, rc);
  return MALS_OK;
}

int mals_group_get_factors(mals_group g, int side, int64_t row_begin, int64_t n_rows, float* host_out) {
  GSIDE(g, side);
  if (int rc = mals_group_synchronize(g)) return rc;  // rows other ranks solved arrive on the comm streams
  Member& mb = g->m[0];
  if (int rc = mals_get_factors(mb.h, side, row_begin, n_rows, host_out)) return mfail(g, mb, rc);
  return MALS_OK;
}

int mals_group_get_rows(mals_group g, int side, const int64_t* row_idx, int32_t n, float* host_out) {
  GSIDE(g, side);
  if (int rc = mals_group_synchronize(g)) return rc;
  Member& mb = g->m[0];
  if (int rc = mals_get_rows(mb.h, side, row_idx, n, host_out)) return mfail(g, mb, rc);
  return MALS_OK;
}

int mals_group_set_matrix(mals_group g, int side, int64_t n_rows, int64_t nnz, const int64_t* row_ptr, const int32_t* col_idx,
                          const float* val, int mem_kind) {
  GSIDE(g, side);
  if (n_rows < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!col_idx || !val))) return gfail(g, MALS_INVALID_ARG, "bad matrix arguments");
  if (mem_kind != MALS_MEM_HOST && mem_kind != MALS_MEM_DEVICE) return gfail(g, MALS_INVALID_ARG, "mem_kind must be MALS_MEM_HOST or MALS_MEM_DEVICE");
  std::vector<int64_t> rp_host;
  const int64_t* rp = row_ptr;
  if (mem_kind == MALS_MEM_DEVICE) {
    rp_host.resize((size_t)n_rows + 1);
    GHIP(g, hipMemcpy(rp_host.data(), row_ptr, sizeof(int64_t) * (size_t)(n_rows + 1), hipMemcpyDeviceToHost));
    rp = rp_host.data();
  }
  if (rp[0] != 0 || rp[n_rows] != nnz) return gfail(g, MALS_INVALID_ARG, "row_ptr must start at 0 and end at nnz");
  if (int rc = plan_side(g, side, rp, n_rows)) return rc;
  for (Member& mb : g->m) {
    const int64_t r0 = g->bounds[side][(size_t)mb.rank], r1 = g->bounds[side][(size_t)mb.rank + 1];
    const int64_t e0 = rp[r0], e1 = rp[r1];
    std::vector<int64_t> local((size_t)(r1 - r0) + 1);
    for (int64_t r = r0; r <= r1; ++r) local[(size_t)(r - r0)] = rp[r] - e0;
    GHIP(g, hipSetDevice(mb.device));
    int rc;
    if (mem_kind == MALS_MEM_HOST) {
      rc = mals_set_matrix(mb.h, side, r0, r1 - r0, e1 - e0, local.data(), col_idx + e0, val + e0, MALS_MEM_HOST);
    } else {
      if (mb.d_row_ptr[side]) (void)hipFree(mb.d_row_ptr[side]);
      mb.d_row_ptr[side] = nullptr;
      // (slices an earlier mals_ingest_install_group copied for this member are not this matrix: let them go once the handle
      // has taken the new arrays below -- mals_set_matrix synchronises the member's stream before it drops the old ones)
      GHIP(g, hipMalloc(&mb.d_row_ptr[side], sizeof(int64_t) * local.size()));
      GHIP(g, hipMemcpy(mb.d_row_ptr[side], local.data(), sizeof(int64_t) * local.size(), hipMemcpyHostToDevice));
      rc = mals_set_matrix(mb.h, side, r0, r1 - r0, e1 - e0, mb.d_row_ptr[side], col_idx + e0, val + e0, MALS_MEM_DEVICE);
    }
    if (rc) return mfail(g, mb, rc);
    if (mb.own_col[side]) (void)hipFree(mb.own_col[side]);
    if (mb.own_val[side]) (void)hipFree(mb.own_val[side]);
    mb.own_col[side] = nullptr;
    mb.own_val[side] = nullptr;
  }
  return finish_matrix(g, side);
}

// InputFilesReader.readInputFiles -> DelegateGenerationManager.java:406-410, for a group: see include/myrrix_als.h
int mals_ingest_install_group(mals_ingest in, mals_group g, int32_t flags) {
  if (!g) return MALS_INVALID_ARG;
  if (!in) return gfail(g, MALS_INVALID_ARG, "null ingest handle");
  int64_t n_rows[2] = {0, 0}, nnz = 0;
  if (int rc = mals_ingest_counts(in, nullptr, &n_rows[0], &n_rows[1], &nnz)) return gfail(g, rc, mals_ingest_last_error(in));
  int32_t in_dev = 0;
  if (int rc = mals_ingest_device(in, &in_dev)) return gfail(g, rc, "mals_ingest_device failed");
  if (flags & ~MALS_INSTALL_COPY) return gfail(g, MALS_INVALID_ARG, "unknown install flag");
  const bool force_copy = (flags & MALS_INSTALL_COPY) != 0;   // members on the ingest's own device copy too: the ingest may go
  auto free_p = [](auto*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
  };
  // replicas first: validate_columns of the matrix upload checks the column indices against the opposite replica
  for (int sd = 0; sd < 2; ++sd)
    if (g->n_total[sd] < n_rows[sd])
      if (int rc = mals_group_set_factor_rows(g, sd, n_rows[sd])) return rc;
  for (int sd = 0; sd < 2; ++sd) {
    const int64_t* d_ptr = nullptr;
    const int32_t* d_col = nullptr;
    const float* d_val = nullptr;
    if (int rc = mals_ingest_device_csr(in, sd, &d_ptr, &d_col, &d_val)) return gfail(g, rc, mals_ingest_last_error(in));
    // the row pointers cross to the host once (8 bytes per row): the plan is made there, like for every other upload
    std::vector<int64_t> rp((size_t)n_rows[sd] + 1);
    GHIP(g, hipSetDevice(in_dev));
    GHIP(g, hipMemcpy(rp.data(), d_ptr, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost));
    if (rp[0] != 0 || rp[(size_t)n_rows[sd]] != nnz) return gfail(g, MALS_INVALID_ARG, "the ingest's row pointers do not match its entry count");
    if (int rc = plan_side(g, sd, rp.data(), n_rows[sd])) return rc;
    for (Member& mb : g->m) {
      const int64_t r0 = g->bounds[sd][(size_t)mb.rank], r1 = g->bounds[sd][(size_t)mb.rank + 1];
      const int64_t e0 = rp[(size_t)r0], e1 = rp[(size_t)r1];
      std::vector<int64_t> local((size_t)(r1 - r0) + 1);
      for (int64_t r = r0; r <= r1; ++r) local[(size_t)(r - r0)] = rp[(size_t)r] - e0;
      GHIP(g, hipSetDevice(mb.device));
      free_p(mb.d_row_ptr[sd]);
      free_p(mb.own_col[sd]);
      free_p(mb.own_val[sd]);
      GHIP(g, hipMalloc(&mb.d_row_ptr[sd], sizeof(int64_t) * local.size()));
      GHIP(g, hipMemcpy(mb.d_row_ptr[sd], local.data(), sizeof(int64_t) * local.size(), hipMemcpyHostToDevice));
      const int32_t* col = d_col + e0;
      const float* val = d_val + e0;
      if ((mb.device != in_dev || force_copy) && e1 > e0) {
        GHIP(g, hipMalloc(&mb.own_col[sd], sizeof(int32_t) * (size_t)(e1 - e0)));
        GHIP(g, hipMalloc(&mb.own_val[sd], sizeof(float) * (size_t)(e1 - e0)));
        GHIP(g, hipMemcpyPeerAsync(mb.own_col[sd], mb.device, col, in_dev, sizeof(int32_t) * (size_t)(e1 - e0), mb.compute));
        GHIP(g, hipMemcpyPeerAsync(mb.own_val[sd], mb.device, val, in_dev, sizeof(float) * (size_t)(e1 - e0), mb.compute));
        GHIP(g, hipStreamSynchronize(mb.compute));
        col = mb.own_col[sd];
        val = mb.own_val[sd];
      }
      if (int rc = mals_set_matrix(mb.h, sd, r0, r1 - r0, e1 - e0, mb.d_row_ptr[sd], col, val, MALS_MEM_DEVICE)) return mfail(g, mb, rc);
    }
    if (int rc = finish_matrix(g, sd)) return rc;
  }
  // knownItemIDs: every member the rows of its own users (mals_recommend by user index is answered by the member that
  // holds the user's row); userTagIDs: every member the whole mask
  const int64_t *k_ptr = nullptr, *t_idx = nullptr;
  const int32_t* k_idx = nullptr;
  int64_t n_known = 0, n_tags = 0;
  (void)mals_ingest_device_known_items(in, &k_ptr, &k_idx, &n_known);   // fails when the ingest did not build them: k_ptr stays null
  if (int rc = mals_ingest_device_tag_items(in, &t_idx, &n_tags)) return gfail(g, rc, mals_ingest_last_error(in));
  std::vector<int64_t> kp;
  if (k_ptr) {
    kp.resize((size_t)n_rows[0] + 1);
    GHIP(g, hipSetDevice(in_dev));
    GHIP(g, hipMemcpy(kp.data(), k_ptr, sizeof(int64_t) * kp.size(), hipMemcpyDeviceToHost));
  }
  for (Member& mb : g->m) {
    GHIP(g, hipSetDevice(mb.device));
    free_p(mb.own_known_ptr);
    free_p(mb.own_known_idx);
    free_p(mb.own_tag_idx);
    if (k_ptr) {
      const int64_t r0 = g->bounds[0][(size_t)mb.rank], r1 = g->bounds[0][(size_t)mb.rank + 1];
      const int64_t e0 = kp[(size_t)r0], e1 = kp[(size_t)r1];
      std::vector<int64_t> local((size_t)(r1 - r0) + 1);
      for (int64_t r = r0; r <= r1; ++r) local[(size_t)(r - r0)] = kp[(size_t)r] - e0;
      GHIP(g, hipMalloc(&mb.own_known_ptr, sizeof(int64_t) * local.size()));
      GHIP(g, hipMemcpy(mb.own_known_ptr, local.data(), sizeof(int64_t) * local.size(), hipMemcpyHostToDevice));
      const int32_t* idx = k_idx + e0;
      if ((mb.device != in_dev || force_copy) && e1 > e0) {
        GHIP(g, hipMalloc(&mb.own_known_idx, sizeof(int32_t) * (size_t)(e1 - e0)));
        GHIP(g, hipMemcpyPeerAsync(mb.own_known_idx, mb.device, idx, in_dev, sizeof(int32_t) * (size_t)(e1 - e0), mb.compute));
        GHIP(g, hipStreamSynchronize(mb.compute));
        idx = mb.own_known_idx;
      }
      if (int rc = mals_set_known_items(mb.h, r1 - r0, mb.own_known_ptr, idx, MALS_MEM_DEVICE)) return mfail(g, mb, rc);
    }
    const int64_t* tags = t_idx;
    if (n_tags > 0 && (mb.device != in_dev || force_copy)) {
      GHIP(g, hipMalloc(&mb.own_tag_idx, sizeof(int64_t) * (size_t)n_tags));
      GHIP(g, hipMemcpyPeerAsync(mb.own_tag_idx, mb.device, t_idx, in_dev, sizeof(int64_t) * (size_t)n_tags, mb.compute));
      GHIP(g, hipStreamSynchronize(mb.compute));
      tags = mb.own_tag_idx;
    }
    if (int rc = mals_set_tag_items(mb.h, n_tags, tags, MALS_MEM_DEVICE)) return mfail(g, mb, rc);
  }
  return MALS_OK;
}

// ServerRecommender.recommend(userID, ...) on a group: every member holds full replicas of X and Y but only its own users' rows
// of R / knownItemIDs, so a user is answered by the member whose slice holds its row.
int mals_group_recommend(mals_group g, const int64_t* user_idx, int32_t n_queries, int32_t how_many, int32_t consider_known_items,
                         int64_t* item_idx_out, float* score_out, int32_t* n_out) {
  if (!g) return MALS_INVALID_ARG;
  if (n_queries < 0 || how_many <= 0 || (n_queries > 0 && (!user_idx || !item_idx_out || !score_out))) return MALS_INVALID_ARG;
  if (g->bounds[MALS_SIDE_X].empty()) return MALS_INVALID_ARG;   // (no message: request threads share the group's error string)
  const std::vector<int64_t>& b = g->bounds[MALS_SIDE_X];
  // runs of consecutive queries with the same owner go to that member in one call (a one-user call is one run)
  for (int32_t q0 = 0; q0 < n_queries;) {
    const int64_t u = user_idx[q0];
    if (u < 0 || u >= b.back()) return MALS_INVALID_ARG;
    const int owner = (int)(std::upper_bound(b.begin(), b.end(), u) - b.begin()) - 1;
    int32_t q1 = q0 + 1;
    while (q1 < n_queries && user_idx[q1] >= b[(size_t)owner] && user_idx[q1] < b[(size_t)owner + 1]) ++q1;
    const Member* mb = nullptr;
    for (const Member& m : g->m)
      if (m.rank == owner) mb = &m;
    // consider_known_items: any member can answer (the replicas are complete); else only the owner knows the user's items
    if (!mb && consider_known_items) mb = &g->m[0];
    if (!mb) return MALS_INVALID_ARG;   // the owner is another process's rank
    if (int rc = mals_recommend(mb->h, user_idx + q0, q1 - q0, how_many, consider_known_items, item_idx_out + (size_t)q0 * how_many,
                                score_out + (size_t)q0 * how_many, n_out ? n_out + q0 : nullptr))
      return rc;
    q0 = q1;
  }
  return MALS_OK;
}

int mals_group_begin_matrix(mals_group g, int side, int64_t n_rows, const int64_t* row_ptr) {
  GSIDE(g, side);
  if (n_rows < 0 || !row_ptr || row_ptr[0] != 0) return gfail(g, MALS_INVALID_ARG, "bad matrix arguments");
  g->up_row_ptr[side].assign(row_ptr, row_ptr + n_rows + 1);
  g->up_next_row[side] = 0;
  if (int rc = plan_side(g, side, row_ptr, n_rows)) return rc;
  for (Member& mb : g->m) {
    const int64_t r0 = g->bounds[side][(size_t)mb.rank], r1 = g->bounds[side][(size_t)mb.rank + 1];
    if (int rc = mals_begin_matrix(mb.h, side, r0, r1 - r0, row_ptr[r1] - row_ptr[r0])) return mfail(g, mb, rc);
    mb.up_rows = 0;
  }
  return MALS_OK;
}

int mals_group_append_rows(mals_group g, int side, int64_t n_rows, const int32_t* col_idx, const float* val) {
  GSIDE(g, side);
  const std::vector<int64_t>& rp = g->up_row_ptr[side];
  if (rp.empty()) return gfail(g, MALS_INVALID_ARG, "mals_group_append_rows without mals_group_begin_matrix");
  const int64_t a = g->up_next_row[side], b = a + n_rows;
  if (n_rows < 0 || b > g->n_rows[side]) return gfail(g, MALS_INVALID_ARG, "piece exceeds the declared matrix");
  const int64_t base = rp[(size_t)a];
  if (rp[(size_t)b] > base && (!col_idx || !val)) return gfail(g, MALS_INVALID_ARG, "null entry arrays");
  for (Member& mb : g->m) {  // the part of [a, b) inside this member's slice
    const int64_t r0 = std::max(a, g->bounds[side][(size_t)mb.rank]), r1 = std::min(b, g->bounds[side][(size_t)mb.rank + 1]);
    if (r1 <= r0) continue;
    std::vector<int64_t> local((size_t)(r1 - r0) + 1);
    for (int64_t r = r0; r <= r1; ++r) local[(size_t)(r - r0)] = rp[(size_t)r] - rp[(size_t)r0];
    const int64_t off = rp[(size_t)r0] - base;
    if (int rc = mals_append_rows(mb.h, side, r1 - r0, local.data(), col_idx ? col_idx + off : nullptr, val ? val + off : nullptr))
      return mfail(g, mb, rc);
  }
  g->up_next_row[side] = b;
  return MALS_OK;
}

int mals_group_end_matrix(mals_group g, int side) {
  GSIDE(g, side);
  if (g->up_row_ptr[side].empty()) return gfail(g, MALS_INVALID_ARG, "mals_group_end_matrix without mals_group_begin_matrix");
  const bool complete = g->up_next_row[side] == g->n_rows[side];
  g->up_row_ptr[side].clear();
  g->up_row_ptr[side].shrink_to_fit();
  if (!complete) return gfail(g, MALS_INVALID_ARG, "appended rows do not match the declared matrix size");
  for (Member& mb : g->m)
    if (int rc = mals_end_matrix(mb.h, side)) return mfail(g, mb, rc);
  return finish_matrix(g, side);
}

int mals_group_bounds(mals_group g, int side, int64_t* bounds_out) {
  GSIDE(g, side);
  if (!bounds_out || g->bounds[side].empty()) return gfail(g, MALS_INVALID_ARG, "matrix of this side not set");
  std::memcpy(bounds_out, g->bounds[side].data(), sizeof(int64_t) * g->bounds[side].size());
  return MALS_OK;
}

int mals_group_half_iteration(mals_group g, int side) {
  GSIDE(g, side);
  if (g->bounds[side].empty()) return gfail(g, MALS_INVALID_ARG, "matrix of this side not set");
  if (!g->m[0].F[side] || !g->m[0].F[1 - side]) return gfail(g, MALS_INVALID_ARG, "factor replicas not allocated");
  // A failure of the local solve (MALS_OOM, MALS_INVALID_ARG ...) must not leave the peers alone in a collective:
  // the remaining exchanges of the half-iteration are still issued (only the solves are skipped), so that every rank
  // arrives at the agreed status with matching calls behind it.  Only a communication / device failure ends it here.
  int solve_rc = MALS_OK;
  std::string solve_msg;
  auto member_failed = [&](const Member& mb, int rc) {
    if (solve_rc == MALS_OK) {
      solve_rc = rc;
      solve_msg = std::string("rank ") + std::to_string(mb.rank) + ": " + mals_last_error(mb.h);
    }
  };
  auto run = [&]() -> int {
    // the gather of this half reads every row of the opposite replica: all of the previous exchange must be in
    for (Member& mb : g->m) {
      GHIP(g, hipSetDevice(mb.device));
      GHIP(g, hipStreamWaitEvent(mb.compute, mb.ev_exchanged, 0));
    }
    if (g->single_process && g->backend == MALS_GROUP_PEER_COPY)  // peer copies INTO a replica run on the source's stream
      for (Member& mb : g->m)
        for (Member& other : g->m) {
          GHIP(g, hipSetDevice(mb.device));
          GHIP(g, hipStreamWaitEvent(mb.compute, other.ev_exchanged, 0));
        }
    if (int rc = group_gramian(g, 1 - side, &solve_rc, &solve_msg)) return rc;  // ALS:342 / ALS:369
    const bool alternate = g->alternate_streams && g->side_chunks[side] > 1;
    if (alternate)
      for (Member& mb : g->m) {   // the second stream starts behind everything the first has seen so far
        GHIP(g, hipSetDevice(mb.device));
        GHIP(g, hipEventRecord(mb.ev_half, mb.compute));
        GHIP(g, hipStreamWaitEvent(mb.compute2, mb.ev_half, 0));
      }
    for (int c = 0; c < g->side_chunks[side]; ++c) {
      const bool odd = alternate && (c & 1);
      if (alternate)
        for (Member& mb : g->m) {
          GHIP(g, hipSetDevice(mb.device));
          if (int rc = mals_set_stream(mb.h, odd ? mb.compute2 : mb.compute)) return mfail(g, mb, rc);
          // ... and behind what the half-iteration sets up once, in its first chunk (operand scale, images, rotated copy)
          if (c == 1) GHIP(g, hipStreamWaitEvent(mb.compute2, (hipEvent_t)malsi_ready_event(mb.h), 0));
        }
      // Like the reference's pool, where every worker is started before any result is awaited (ALS:186-191,391-410):
      // the direct kernels of EVERY member are enqueued before any host work; the k x k eigendecomposition of the
      // dual path (the same G on every member after the all-reduce) is then computed once, under those kernels, and
      // handed to the others; only then the members' dual kernels follow.
      std::vector<char> has(g->m.size(), 0);
      for (size_t i = 0; i < g->m.size(); ++i) {
        Member& mb = g->m[i];
        GHIP(g, hipSetDevice(mb.device));
        int32_t mine = 0;
        if (int rc = mals_num_chunks(mb.h, side, &mine)) return mfail(g, mb, rc);
        has[i] = c < mine && solve_rc == MALS_OK;
        if (has[i])
          if (int rc = malsi_solve_chunk_begin(mb.h, side, c)) {  // ALS:344 / ALS:371
            member_failed(mb, rc);
            has[i] = 0;
          }
      }
      Member* decomposed = nullptr;
      for (size_t i = 0; i < g->m.size(); ++i) {
        Member& mb = g->m[i];
        if (!has[i] || !malsi_dual_pending(mb.h)) continue;
        GHIP(g, hipSetDevice(mb.device));
        if (int rc = malsi_dual_host(mb.h, side, decomposed ? decomposed->h : nullptr)) {
          member_failed(mb, rc);
          has[i] = 0;
          continue;
        }
        if (!decomposed) decomposed = &mb;
      }
      for (size_t i = 0; i < g->m.size(); ++i) {
        Member& mb = g->m[i];
        GHIP(g, hipSetDevice(mb.device));
        if (has[i])
          if (int rc = malsi_solve_chunk_end(mb.h, side, c)) member_failed(mb, rc);
        GHIP(g, hipEventRecord(mb.ev_solved, odd ? mb.compute2 : mb.compute));
        GHIP(g, hipStreamWaitEvent(mb.comm, mb.ev_solved, 0));
      }
      if (int rc = exchange_chunk(g, side, c)) return rc;
    }
    if (alternate)
      for (Member& mb : g->m) {   // join: everything after this half-iteration assumes the one compute stream
        GHIP(g, hipSetDevice(mb.device));
        GHIP(g, hipEventRecord(mb.ev_join, mb.compute2));
        GHIP(g, hipStreamWaitEvent(mb.compute, mb.ev_join, 0));
        if (int rc = mals_set_stream(mb.h, mb.compute)) return mfail(g, mb, rc);
      }
    for (Member& mb : g->m) {
      GHIP(g, hipSetDevice(mb.device));
      GHIP(g, hipEventRecord(mb.ev_exchanged, mb.comm));
    }
    if (solve_rc != MALS_OK) return gfail(g, solve_rc, solve_msg);
    for (Member& mb : g->m) {  // ALS:346-361: f.get() of every worker
      GHIP(g, hipSetDevice(mb.device));
      if (int rc = mals_check(mb.h)) return mfail(g, mb, rc);
    }
    return MALS_OK;
  };
  const int local_rc = run();
  const std::string local_msg = g->err;
  for (Member& mb : g->m) (void)mals_set_stream(mb.h, mb.compute);   // (also after a failure in the middle of the chunk loop)
  // a communication failure cannot be agreed upon over the same communicator
  if (local_rc == MALS_COMM_ERROR || local_rc == MALS_HIP_ERROR) return local_rc;
  return agree_status(g, local_rc, local_msg);
}

int mals_group_singular_info(mals_group g, int32_t* side, int64_t* row, int32_t* apparent_rank) {
  if (!g) return MALS_INVALID_ARG;
  if (side) *side = -1;
  if (row) *row = -1;
  if (apparent_rank) *apparent_rank = 0;
  for (Member& mb : g->m) {  // the local member that reported it (multi-process groups: each process knows its own)
    int32_t sd = -1, rk = 0;
    int64_t rw = -1;
    if (mals_singular_info(mb.h, &sd, &rw, &rk) == MALS_OK && sd >= 0) {
      if (side) *side = sd;
      if (row) *row = rw;
      if (apparent_rank) *apparent_rank = rk;
      return MALS_OK;
    }
  }
  return MALS_OK;
}

int mals_group_exchange_only(mals_group g, int side) {
  GSIDE(g, side);
  if (g->bounds[side].empty()) return gfail(g, MALS_INVALID_ARG, "matrix of this side not set");
  for (Member& mb : g->m) {
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipEventRecord(mb.ev_solved, mb.compute));
    GHIP(g, hipStreamWaitEvent(mb.comm, mb.ev_solved, 0));
  }
  for (int c = 0; c < g->side_chunks[side]; ++c)
    if (int rc = exchange_chunk(g, side, c)) return rc;
  for (Member& mb : g->m) {
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipEventRecord(mb.ev_exchanged, mb.comm));
  }
  return MALS_OK;
}

int mals_group_cancel(mals_group g) {
  if (!g) return MALS_INVALID_ARG;
  g->cancelled.store(1);
  return MALS_OK;
}

int mals_group_synchronize(mals_group g) {
  if (!g) return MALS_INVALID_ARG;
  for (Member& mb : g->m) {
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipStreamSynchronize(mb.compute));
    GHIP(g, hipStreamSynchronize(mb.comm));
  }
  return MALS_OK;
}

int mals_group_factorize(mals_group g, double convergence_threshold, int32_t max_iterations, int32_t random_y, int32_t iterate,
                         const int64_t* test_users, int32_t n_test_users, const int64_t* test_items, int32_t n_test_items,
                         int32_t* iterations_out, double* convergence_out) {
  if (!g) return MALS_INVALID_ARG;
  if (iterations_out) *iterations_out = 0;
  if (convergence_out) *convergence_out = std::numeric_limits<double>::quiet_NaN();
  if (!(convergence_threshold > 0.0 && convergence_threshold < 1.0)) return gfail(g, MALS_INVALID_ARG, "threshold must be in (0,1)");  // ALS:140-141
  if (n_test_users < 0 || n_test_items < 0 || (n_test_users > 0 && !test_users) || (n_test_items > 0 && !test_items))
    return gfail(g, MALS_INVALID_ARG, "bad convergence sample");
  g->cancelled.store(0);
  if (!iterate) return mals_group_half_iteration(g, MALS_SIDE_X);  // ALS:196-204
  std::vector<double> est((size_t)n_test_users * (size_t)n_test_items, 0.0), fresh(est.size());
  int it = 0;
  for (;;) {
    const auto t_it = std::chrono::steady_clock::now();
    mals_stats st0;
    // the callback is sampled ONCE per iteration (one installed while an iteration runs starts with the next one)
    const mals_iteration_fn iter_fn = g->iter_fn;
    void* const iter_user = g->iter_user;
    std::vector<std::pair<int64_t, int64_t>> before;   // (rows_solved, nnz_gathered) of every local member
    auto rows_now = [&]() {
      int64_t r = 0;
      for (Member& mb : g->m) {
        (void)mals_get_stats(mb.h, &st0);
        r += st0.rows_solved;
      }
      return r;
    };
    int64_t rows_before = 0, rows_after_x = 0;
    if (iter_fn) {
      for (Member& mb : g->m) {
        (void)mals_get_stats(mb.h, &st0);
        before.emplace_back(st0.rows_solved, st0.nnz_gathered);
        rows_before += st0.rows_solved;
      }
    }
    // a cancellation is local knowledge: agree on it before entering a collective
    if (int rc = agree_status(g, g->cancelled.load() ? MALS_CANCELLED : MALS_OK, "cancelled")) return rc;
    if (int rc = mals_group_half_iteration(g, MALS_SIDE_X)) return rc;  // ALS:228
    if (iter_fn) rows_after_x = rows_now();  // what THIS process's members solved in the X half, exactly
    if (int rc = agree_status(g, g->cancelled.load() ? MALS_CANCELLED : MALS_OK, "cancelled")) return rc;
    if (int rc = mals_group_half_iteration(g, MALS_SIDE_Y)) return rc;  // ALS:229
    // ALS:231-238: the sample dots (SimpleVectorMath.dot) on the device from a complete replica (every replica
    // holds the same factors once the exchange is in), DoubleWeightedMean.increment on the host
    if (int rc = mals_group_synchronize(g)) return rc;
    if (int rc = mals_sample_dots(g->m[0].h, test_users, n_test_users, test_items, n_test_items, fresh.data())) return mfail(g, g->m[0], rc);
    double tw = 0.0, mean = std::numeric_limits<double>::quiet_NaN();
    for (int i = 0; i < n_test_users; ++i)
      for (int j = 0; j < n_test_items; ++j) {
        const double nv = fresh[(size_t)i * n_test_items + j];
        double& slot = est[(size_t)i * n_test_items + j];
        const double datum = std::fabs(nv - slot), weight = nv > 0.0 ? nv : 0.0;
        slot = nv;
        const double old = tw;
        tw += weight;
        mean = old <= 0 ? datum : mean * old / tw + datum * weight / tw;
      }
    ++it;
    if (iterations_out) *iterations_out = it;
    if (convergence_out) *convergence_out = mean;
    if (iter_fn && before.size() == g->m.size()) {   // what the reference logs per iteration (ALS:241-246, 351-358)
      mals_iteration_info info;
      std::memset(&info, 0, sizeof(info));
      info.struct_size = (int32_t)sizeof(info);
      info.iteration = it;
      info.avg_abs_difference = mean;
      info.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_it).count();
      int64_t rows = 0;
      for (size_t i = 0; i < g->m.size(); ++i) {
        (void)mals_get_stats(g->m[i].h, &st0);
        rows += st0.rows_solved - before[i].first;
        info.entries_gathered += st0.nnz_gathered - before[i].second;
      }
      info.x_rows = rows_after_x - rows_before;
      info.y_rows = rows - info.x_rows;
      info.algorithmic_bytes = (double)(info.entries_gathered + rows) * (4.0 * g->cfg.features + 8.0);
      info.devices = (int32_t)g->m.size();
      iter_fn(iter_user, &info);
    }
    if (max_iterations > 0 && it >= max_iterations) break;              // ALS:242-245
    if (!std::isfinite(mean)) break;                                    // ALS:248-251
    if (!(random_y && it == 1) && mean < convergence_threshold) break;  // ALS:253-256
  }
  return MALS_OK;
}

}  // extern "C"
