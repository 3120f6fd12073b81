// mock_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in transport with RCCL's entry points, so that the RCCL code
// path of csrc/mals_group.cpp (ncclCommInitAll / ncclCommInitRank, grouped ncclSend + ncclRecv exchange,
// ncclAllReduce, all on each rank's comm stream) can be EXECUTED with N > 1 ranks on a box that has one GPU.
// RCCL itself refuses two ranks on one device; this library does not care where a rank lives.  Selected with
// mals_group_use_transport(<path to libmock_rccl.so>) (csrc/mals_group.cpp Rccl::load); never used by the product path.
//
// What it checks that a real communicator would punish with a hang or silent corruption:
//   * every ncclSend meets an ncclRecv of the SAME element count on the peer, in the same order per pair
//     (a mismatch returns ncclInvalidArgument instead of deadlocking);
//   * every rank of the communicator joins an all-reduce with the same count, type and operation;
//   * data is read from / written to the device buffers behind the stream the call was given and ONLY that
//     stream is synchronised -- a missing event between the compute and the comm stream shows up as stale data.
// Transport: ranks of one process (ncclCommInitAll) exchange with device-to-device copies when the outermost
// ncclGroupEnd runs; ranks of different processes (ncclCommInitRank) through files under /dev/shm/<unique id>/
// (sends are buffered first, receives then poll with a timeout).  Correctness only: every call is synchronous.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Context {           // one communicator clique
  int world = 0;
  bool in_process = false;
  std::string dir;         // multi-process: mailbox directory
};

struct Op {
  int kind;                // 0 send, 1 recv, 2 all-reduce
  const void* src;
  void* dst;
  size_t count;
  ncclDataType_t type;
  ncclRedOp_t op;
  int peer;
  struct ncclComm* comm;
  hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_queue;

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclFloat: return 4;
    case ncclDouble: return 8;
    case ncclInt32: case ncclUint32: return 4;
    case ncclInt64: case ncclUint64: return 8;
    case ncclInt8: case ncclUint8: return 1;
    default: return 0;
  }
}

}  // namespace

struct ncclComm {
  std::shared_ptr<Context> ctx;
  int rank = 0;
  int device = 0;
  std::map<int, long> send_seq, recv_seq;  // multi-process: per-peer message counters
  long ar_seq = 0;
};

namespace {

bool read_device(const Op& o, const void* p, size_t bytes, std::vector<char>& host) {
  host.resize(bytes);
  if (hipSetDevice(o.comm->device) != hipSuccess) return false;
  if (hipMemcpyAsync(host.data(), p, bytes, hipMemcpyDeviceToHost, o.stream) != hipSuccess) return false;
  return hipStreamSynchronize(o.stream) == hipSuccess;
}
bool write_device(const Op& o, void* p, size_t bytes, const std::vector<char>& host) {
  if (hipSetDevice(o.comm->device) != hipSuccess) return false;
  if (hipMemcpyAsync(p, host.data(), bytes, hipMemcpyHostToDevice, o.stream) != hipSuccess) return false;
  return hipStreamSynchronize(o.stream) == hipSuccess;
}

template <typename T>
void reduce_into(std::vector<char>& acc, const std::vector<char>& in, ncclRedOp_t op) {
  T* a = reinterpret_cast<T*>(acc.data());
  const T* b = reinterpret_cast<const T*>(in.data());
  const size_t n = acc.size() / sizeof(T);
  for (size_t i = 0; i < n; ++i) a[i] = op == ncclMax ? std::max(a[i], b[i]) : (op == ncclMin ? std::min(a[i], b[i]) : a[i] + b[i]);
}
bool reduce_typed(std::vector<char>& acc, const std::vector<char>& in, ncclDataType_t t, ncclRedOp_t op) {
  if (op != ncclSum && op != ncclMax && op != ncclMin) return false;
  switch (t) {
    case ncclFloat: reduce_into<float>(acc, in, op); return true;
    case ncclDouble: reduce_into<double>(acc, in, op); return true;
    case ncclInt32: reduce_into<int32_t>(acc, in, op); return true;
    case ncclInt64: reduce_into<int64_t>(acc, in, op); return true;
    default: return false;
  }
}

bool write_file(const std::string& path, const std::vector<char>& data, size_t count) {
  const std::string tmp = path + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const unsigned long long n = count;
  bool ok = std::fwrite(&n, sizeof(n), 1, f) == 1 && (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size());
  ok = std::fclose(f) == 0 && ok;
  return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}
// 0 ok, 1 timeout, 2 count mismatch / io error
int read_file(const std::string& path, size_t count, size_t bytes, std::vector<char>& data, bool remove_after) {
  const auto t0 = std::chrono::steady_clock::now();
  struct stat st;
  while (stat(path.c_str(), &st) != 0) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return 1;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return 2;
  unsigned long long n = 0;
  data.resize(bytes);
  bool ok = std::fread(&n, sizeof(n), 1, f) == 1 && n == count && (bytes == 0 || std::fread(data.data(), 1, bytes, f) == bytes);
  std::fclose(f);
  if (remove_after) (void)std::remove(path.c_str());
  return ok ? 0 : 2;
}

ncclResult_t run_in_process(std::vector<Op>& q) {
  // point-to-point: FIFO per (source, destination) pair
  std::map<std::pair<int, int>, std::vector<const Op*>> sends, recvs;
  std::map<ncclComm*, std::vector<const Op*>> reduces;
  for (const Op& o : q) {
    if (o.kind == 0) sends[{o.comm->rank, o.peer}].push_back(&o);
    else if (o.kind == 1) recvs[{o.peer, o.comm->rank}].push_back(&o);
    else reduces[o.comm].push_back(&o);
  }
  for (auto& kv : sends) {
    auto it = recvs.find(kv.first);
    if (it == recvs.end() || it->second.size() != kv.second.size()) {
      std::fprintf(stderr, "mock_rccl: %zu sends %d -> %d without matching receives\n", kv.second.size(), kv.first.first, kv.first.second);
      return ncclInvalidArgument;
    }
  }
  for (auto& kv : recvs)
    if (!sends.count(kv.first)) {
      std::fprintf(stderr, "mock_rccl: receives %d <- %d without sends\n", kv.first.second, kv.first.first);
      return ncclInvalidArgument;
    }
  std::vector<char> host;
  for (auto& kv : sends) {
    const auto& rs = recvs[kv.first];
    for (size_t i = 0; i < kv.second.size(); ++i) {
      const Op& s = *kv.second[i];
      const Op& r = *rs[i];
      if (s.count != r.count || s.type != r.type) {
        std::fprintf(stderr, "mock_rccl: send %d -> %d of %zu elements meets a receive of %zu\n", kv.first.first, kv.first.second, s.count, r.count);
        return ncclInvalidArgument;
      }
      const size_t bytes = s.count * type_size(s.type);
      if (!read_device(s, s.src, bytes, host) || !write_device(r, r.dst, bytes, host)) return ncclUnhandledCudaError;
    }
  }
  // all-reduces: the i-th of every rank belong together
  if (!reduces.empty()) {
    const size_t n_calls = reduces.begin()->second.size();
    const int world = reduces.begin()->first->ctx->world;
    if ((int)reduces.size() != world) {
      std::fprintf(stderr, "mock_rccl: all-reduce joined by %zu of %d ranks\n", reduces.size(), world);
      return ncclInvalidArgument;
    }
    for (auto& kv : reduces)
      if (kv.second.size() != n_calls) return ncclInvalidArgument;
    for (size_t i = 0; i < n_calls; ++i) {
      std::vector<std::pair<int, const Op*>> by_rank;
      for (auto& kv : reduces) by_rank.push_back({kv.first->rank, kv.second[i]});
      std::sort(by_rank.begin(), by_rank.end(), [](const std::pair<int, const Op*>& a, const std::pair<int, const Op*>& b) { return a.first < b.first; });
      const Op& first = *by_rank[0].second;
      const size_t bytes = first.count * type_size(first.type);
      std::vector<char> acc, in;
      for (size_t j = 0; j < by_rank.size(); ++j) {
        const Op& o = *by_rank[j].second;
        if (o.count != first.count || o.type != first.type || o.op != first.op) return ncclInvalidArgument;
        if (!read_device(o, o.src, bytes, j == 0 ? acc : in)) return ncclUnhandledCudaError;
        if (j > 0 && !reduce_typed(acc, in, o.type, o.op)) return ncclInvalidArgument;
      }
      for (auto& pr : by_rank)
        if (!write_device(*pr.second, pr.second->dst, bytes, acc)) return ncclUnhandledCudaError;
    }
  }
  return ncclSuccess;
}

ncclResult_t run_multi_process(std::vector<Op>& q) {
  std::vector<char> host;
  // buffered sends first, then the receives: no rank ever waits for a peer before it has posted its own data
  for (const Op& o : q) {
    if (o.kind != 0) continue;
    const size_t bytes = o.count * type_size(o.type);
    if (!read_device(o, o.src, bytes, host)) return ncclUnhandledCudaError;
    const long seq = o.comm->send_seq[o.peer]++;
    const std::string path = o.comm->ctx->dir + "/p2p_" + std::to_string(o.comm->rank) + "_" + std::to_string(o.peer) + "_" + std::to_string(seq);
    if (!write_file(path, host, o.count)) return ncclSystemError;
  }
  for (const Op& o : q) {
    if (o.kind != 1) continue;
    const size_t bytes = o.count * type_size(o.type);
    const long seq = o.comm->recv_seq[o.peer]++;
    const std::string path = o.comm->ctx->dir + "/p2p_" + std::to_string(o.peer) + "_" + std::to_string(o.comm->rank) + "_" + std::to_string(seq);
    const int rc = read_file(path, o.count, bytes, host, true);
    if (rc == 1) {
      std::fprintf(stderr, "mock_rccl: rank %d timed out waiting for message %ld from rank %d\n", o.comm->rank, seq, o.peer);
      return ncclSystemError;
    }
    if (rc == 2) {
      std::fprintf(stderr, "mock_rccl: rank %d: message %ld from rank %d does not have %zu elements\n", o.comm->rank, seq, o.peer, o.count);
      return ncclInvalidArgument;
    }
    if (!write_device(o, o.dst, bytes, host)) return ncclUnhandledCudaError;
  }
  for (const Op& o : q) {
    if (o.kind != 2) continue;
    const size_t bytes = o.count * type_size(o.type);
    const long seq = o.comm->ar_seq++;
    if (!read_device(o, o.src, bytes, host)) return ncclUnhandledCudaError;
    const std::string base = o.comm->ctx->dir + "/ar_" + std::to_string(seq) + "_" + std::to_string((int)o.op) + "_";
    if (!write_file(base + std::to_string(o.comm->rank), host, o.count)) return ncclSystemError;
    std::vector<char> acc, in;
    for (int r = 0; r < o.comm->ctx->world; ++r) {   // rank order: every rank computes the same sum
      const int rc = read_file(base + std::to_string(r), o.count, bytes, r == 0 ? acc : in, false);
      if (rc != 0) {
        std::fprintf(stderr, "mock_rccl: rank %d: all-reduce %ld: contribution of rank %d %s\n", o.comm->rank, seq, r,
                     rc == 1 ? "never arrived" : "has another count or operation");
        return rc == 1 ? ncclSystemError : ncclInvalidArgument;
      }
      if (r > 0 && !reduce_typed(acc, in, o.type, o.op)) return ncclInvalidArgument;
    }
    if (!write_device(o, o.dst, bytes, acc)) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

ncclResult_t flush() {
  std::vector<Op> q;
  q.swap(g_queue);
  if (q.empty()) return ncclSuccess;
  const bool in_process = q[0].comm->ctx->in_process;
  for (const Op& o : q)
    if (o.comm->ctx->in_process != in_process) return ncclInvalidUsage;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const ncclResult_t r = in_process ? run_in_process(q) : run_multi_process(q);
  (void)hipSetDevice(dev);
  return r;
}

ncclResult_t enqueue(const Op& o) {
  if (!o.comm || type_size(o.type) == 0) return ncclInvalidArgument;
  if (o.kind != 2 && (o.peer < 0 || o.peer >= o.comm->ctx->world || o.peer == o.comm->rank)) return ncclInvalidArgument;
  g_queue.push_back(o);
  return g_depth > 0 ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id, 0, sizeof(*id));
  static int counter = 0;
  std::snprintf(id->internal, sizeof(id->internal), "mockrccl_%ld_%lld_%d", (long)getpid(),
                (long long)std::chrono::steady_clock::now().time_since_epoch().count(), counter++);
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  if (!comms || ndev <= 0) return ncclInvalidArgument;
  auto ctx = std::make_shared<Context>();
  ctx->world = ndev;
  ctx->in_process = true;
  for (int i = 0; i < ndev; ++i) {
    ncclComm* c = new ncclComm();
    c->ctx = ctx;
    c->rank = i;
    c->device = devlist ? devlist[i] : i;
    comms[i] = c;
  }
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks <= 0 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  auto ctx = std::make_shared<Context>();
  ctx->world = nranks;
  ctx->in_process = false;
  ctx->dir = std::string("/dev/shm/") + id.internal;
  if (mkdir(ctx->dir.c_str(), 0700) != 0 && access(ctx->dir.c_str(), W_OK) != 0) return ncclSystemError;
  ncclComm* c = new ncclComm();
  c->ctx = ctx;
  c->rank = rank;
  (void)hipGetDevice(&c->device);
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclInvalidArgument;
  if (!comm->ctx->in_process) {  // own leftovers only (all-reduce contributions); the directory goes when it is empty
    for (long s = 0; s < comm->ar_seq; ++s)
      for (int op = 0; op < 5; ++op)
        (void)std::remove((comm->ctx->dir + "/ar_" + std::to_string(s) + "_" + std::to_string(op) + "_" + std::to_string(comm->rank)).c_str());
    (void)rmdir(comm->ctx->dir.c_str());
  }
  delete comm;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  return enqueue(Op{2, sendbuff, recvbuff, count, datatype, op, -1, comm, stream});
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return enqueue(Op{0, sendbuff, nullptr, count, datatype, ncclSum, peer, comm, stream});
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return enqueue(Op{1, nullptr, recvbuff, count, datatype, ncclSum, peer, comm, stream});
}

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  return --g_depth == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  if (!comm || !count) return ncclInvalidArgument;
  *count = comm->ctx->world;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
  if (!comm || !rank) return ncclInvalidArgument;
  *rank = comm->rank;
  return ncclSuccess;
}

ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int* device) {
  if (!comm || !device) return ncclInvalidArgument;
  *device = comm->device;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t result) {
  switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "mock transport: HIP error";
    case ncclSystemError: return "mock transport: system error or timeout";
    case ncclInvalidArgument: return "mock transport: mismatched call (count, type, peer or participation)";
    case ncclInvalidUsage: return "mock transport: invalid usage";
    default: return "mock transport: error";
  }
}

}  // extern "C"
