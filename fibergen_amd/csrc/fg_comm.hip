#include "fg_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>

#include "fg_hip_util.h"

namespace fg {

// =====================================================================================================================
// RCCL, loaded at run time.  A process that already carries an RCCL (PyTorch ships one under the same soname) shares
// it; otherwise the ROCm one is loaded.  Only the handful of entry points below are used.
namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  static std::string error;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (!api.handle) {
      error = std::string("cannot load librccl: ") + dlerror();
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(api.handle, n);
      if (!p && error.empty()) error = std::string("librccl lacks ") + n;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  });
  if (!error.empty()) throw std::runtime_error("RCCL: " + error);
  return api;
}

#define FG_NCCL_CHECK(expr)                                                                                   \
  do {                                                                                                        \
    ncclResult_t fg_r__ = (expr);                                                                             \
    if (fg_r__ != ncclSuccess)                                                                                \
      throw std::runtime_error(std::string("RCCL error: ") + rccl().GetErrorString(fg_r__) + " in " #expr);   \
  } while (0)

class RcclComm : public Comm {
 public:
  RcclComm(const char* id128, int rank, int nranks, int device) : rank_(rank), size_(nranks) {
    static_assert(sizeof(ncclUniqueId) == kUniqueIdBytes, "ncclUniqueId is 128 bytes");
    FG_HIP_CHECK(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    FG_NCCL_CHECK(rccl().CommInitRank(&comm_, nranks, id, rank));
  }
  ~RcclComm() override {
    if (comm_) (void)rccl().CommDestroy(comm_);
  }
  int rank() const override { return rank_; }
  int size() const override { return size_; }
  const char* name() const override { return "rccl"; }
  void exchange(const XOp* ops, int n, hipStream_t stream) override {
    if (n == 0) return;
    RcclApi& a = rccl();
    FG_NCCL_CHECK(a.GroupStart());
    for (int i = 0; i < n; ++i) {
      const XOp& o = ops[i];
      if (o.send) FG_NCCL_CHECK(a.Send(o.ptr, o.bytes / 8, ncclDouble, o.peer, comm_, stream));
      else FG_NCCL_CHECK(a.Recv(o.ptr, o.bytes / 8, ncclDouble, o.peer, comm_, stream));
    }
    FG_NCCL_CHECK(a.GroupEnd());
  }
  void allreduce(double* buf, int n, bool min_op, hipStream_t stream) override {
    FG_NCCL_CHECK(rccl().AllReduce(buf, buf, (size_t)n, ncclDouble, min_op ? ncclMin : ncclSum, comm_, stream));
  }
  void group_begin() override { FG_NCCL_CHECK(rccl().GroupStart()); }
  void group_end() override { FG_NCCL_CHECK(rccl().GroupEnd()); }

 private:
  int rank_, size_;
  ncclComm_t comm_ = nullptr;
};

}  // namespace

void rccl_unique_id(char* out128) {
  ncclUniqueId id;
  FG_NCCL_CHECK(rccl().GetUniqueId(&id));
  std::memcpy(out128, &id, sizeof(id));
}

std::unique_ptr<Comm> make_rccl_comm(const char* id128, int rank, int nranks, int device) {
  return std::unique_ptr<Comm>(new RcclComm(id128, rank, nranks, device));
}

// =====================================================================================================================
// In-process group: every member posts its ops; when the last member has posted exchange number k, the copies of
// exchange k are enqueued (all members share one stream, so enqueue order is execution order).
constexpr int kMaxLocalRanks = 16;

struct ReducePtrs {
  double* p[kMaxLocalRanks];
};

// fixed rank order => identical, reproducible results on every member
__global__ void k_local_allreduce(ReducePtrs bufs, int nranks, int n, int min_op) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = bufs.p[0][i];
  for (int r = 1; r < nranks; ++r) {
    const double b = bufs.p[r][i];
    a = min_op ? (b < a ? b : a) : a + b;
  }
  for (int r = 0; r < nranks; ++r) bufs.p[r][i] = a;
}

// The device copies of one in-process exchange as ONE launch (the 8 members of a 256^3 problem post 56 blocks per all-to-all:
// as separate hipMemcpyAsync calls those tiny copies, not the kernels, bound the emulation of the multi-GPU run on one GPU).
constexpr int kCopyBatch = 96;
struct CopyBatch {
  const double2* src[kCopyBatch];
  double2* dst[kCopyBatch];
  unsigned n2[kCopyBatch];   // 16-byte elements
};
__global__ void k_local_copy(CopyBatch ops) {
  const double2* s = ops.src[blockIdx.y];
  double2* d = ops.dst[blockIdx.y];
  const unsigned n = ops.n2[blockIdx.y];
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = s[i];
}

class LocalHub {
 public:
  explicit LocalHub(int n) : n_(n), posted_(n), reduce_(n) {
    if (n < 1 || n > kMaxLocalRanks) throw std::runtime_error("in-process slab group: 1..16 members");
  }
  int size() const { return n_; }

  void post_exchange(int rank, const XOp* ops, int n, hipStream_t stream) {
    posted_[rank].emplace_back(ops, ops + n);
    // exchange k is complete when every member's queue holds one
    for (int r = 0; r < n_; ++r)
      if (posted_[r].empty()) return;
    std::vector<std::vector<XOp>> cur(n_);
    for (int r = 0; r < n_; ++r) {
      cur[r] = std::move(posted_[r].front());
      posted_[r].pop_front();
    }
    CopyBatch batch;
    int nb = 0;
    size_t longest = 0;
    auto flush = [&]() {
      if (nb == 0) return;
      unsigned gx = (unsigned)((longest / 16 + 255) / 256);
      if (gx > 64) gx = 64;
      if (gx < 1) gx = 1;
      hipLaunchKernelGGL(k_local_copy, dim3(gx, nb), dim3(256), 0, stream, batch);
      FG_HIP_CHECK(hipGetLastError());
      nb = 0;
      longest = 0;
    };
    for (int dst = 0; dst < n_; ++dst) {
      std::vector<size_t> next(n_, 0);   // per source: position of the next unmatched send to dst
      for (const XOp& rv : cur[dst]) {
        if (rv.send) continue;
        const int src = rv.peer;
        if (src < 0 || src >= n_) throw std::runtime_error("in-process exchange: bad peer");
        size_t& k = next[src];
        while (k < cur[src].size() && !(cur[src][k].send && cur[src][k].peer == dst)) ++k;
        if (k == cur[src].size()) throw std::runtime_error("in-process exchange: receive without a matching send");
        if (cur[src][k].bytes != rv.bytes) throw std::runtime_error("in-process exchange: message sizes differ");
        if (rv.bytes % 16 || ((uintptr_t)rv.ptr | (uintptr_t)cur[src][k].ptr) % 16 || rv.bytes / 16 > 0xffffffffu) {
          FG_HIP_CHECK(hipMemcpyAsync(rv.ptr, cur[src][k].ptr, rv.bytes, hipMemcpyDeviceToDevice, stream));
        } else {
          batch.src[nb] = static_cast<const double2*>(cur[src][k].ptr);
          batch.dst[nb] = static_cast<double2*>(rv.ptr);
          batch.n2[nb] = (unsigned)(rv.bytes / 16);
          if (rv.bytes > longest) longest = rv.bytes;
          if (++nb == kCopyBatch) flush();
        }
        ++k;
      }
    }
    flush();
    for (int src = 0; src < n_; ++src) {   // every send must have found its receive
      size_t sends = 0, recvs_of_it = 0;
      for (const XOp& o : cur[src]) sends += o.send ? 1 : 0;
      for (int dst = 0; dst < n_; ++dst)
        for (const XOp& o : cur[dst]) recvs_of_it += (!o.send && o.peer == src) ? 1 : 0;
      if (sends != recvs_of_it) throw std::runtime_error("in-process exchange: send without a matching receive");
    }
  }

  void post_allreduce(int rank, double* buf, int n, bool min_op, hipStream_t stream) {
    reduce_[rank].push_back({buf, n, min_op});
    for (int r = 0; r < n_; ++r)
      if (reduce_[r].empty()) return;
    ReducePtrs bufs;
    for (int r = 0; r < n_; ++r) {
      const Red& q = reduce_[r].front();
      if (q.n != n || q.min_op != min_op) throw std::runtime_error("in-process all-reduce: members disagree");
      bufs.p[r] = q.buf;
    }
    for (int r = 0; r < n_; ++r) reduce_[r].pop_front();
    hipLaunchKernelGGL(k_local_allreduce, dim3((n + 63) / 64), dim3(64), 0, stream, bufs, n_, n, min_op ? 1 : 0);
    FG_HIP_CHECK(hipGetLastError());
  }

 private:
  struct Red {
    double* buf;
    int n;
    bool min_op;
  };
  int n_;
  std::vector<std::deque<std::vector<XOp>>> posted_;
  std::vector<std::deque<Red>> reduce_;
};

namespace {
class LocalComm : public Comm {
 public:
  LocalComm(std::shared_ptr<LocalHub> hub, int rank) : hub_(std::move(hub)), rank_(rank) {}
  int rank() const override { return rank_; }
  int size() const override { return hub_->size(); }
  const char* name() const override { return "local"; }
  void exchange(const XOp* ops, int n, hipStream_t stream) override { hub_->post_exchange(rank_, ops, n, stream); }
  void allreduce(double* buf, int n, bool min_op, hipStream_t stream) override {
    hub_->post_allreduce(rank_, buf, n, min_op, stream);
  }

 private:
  std::shared_ptr<LocalHub> hub_;
  int rank_;
};

// =====================================================================================================================
// The caller moves the bytes: the stream is drained, then the callback sees device pointers (exchange) or host values
// (all-reduce).  For multi-process tests over gloo; not a performance path.
class CallbackComm : public Comm {
 public:
  CallbackComm(int rank, int nranks, fg_exchange_fn p2p, fg_allreduce_fn ar, void* user)
      : rank_(rank), size_(nranks), p2p_(p2p), ar_(ar), user_(user) {
    if (!p2p || !ar) throw std::runtime_error("callback transport needs both callbacks");
  }
  int rank() const override { return rank_; }
  int size() const override { return size_; }
  const char* name() const override { return "callback"; }
  void exchange(const XOp* ops, int n, hipStream_t stream) override {
    FG_HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<fg_xop> c(n);
    for (int i = 0; i < n; ++i) c[i] = fg_xop{ops[i].send, ops[i].peer, ops[i].ptr, (unsigned long)ops[i].bytes};
    if (p2p_(user_, c.data(), n) != 0) throw std::runtime_error("exchange callback failed");
  }
  void allreduce(double* buf, int n, bool min_op, hipStream_t stream) override {
    std::vector<double> h(n);
    FG_HIP_CHECK(hipMemcpyAsync(h.data(), buf, n * sizeof(double), hipMemcpyDeviceToHost, stream));
    FG_HIP_CHECK(hipStreamSynchronize(stream));
    if (ar_(user_, h.data(), n, min_op ? 1 : 0) != 0) throw std::runtime_error("all-reduce callback failed");
    FG_HIP_CHECK(hipMemcpyAsync(buf, h.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
    FG_HIP_CHECK(hipStreamSynchronize(stream));
  }

 private:
  int rank_, size_;
  fg_exchange_fn p2p_;
  fg_allreduce_fn ar_;
  void* user_;
};
}  // namespace

std::shared_ptr<LocalHub> make_local_hub(int nranks) { return std::make_shared<LocalHub>(nranks); }
std::unique_ptr<Comm> make_local_comm(std::shared_ptr<LocalHub> hub, int rank) {
  return std::unique_ptr<Comm>(new LocalComm(std::move(hub), rank));
}
std::unique_ptr<Comm> make_callback_comm(int rank, int nranks, fg_exchange_fn p2p, fg_allreduce_fn allreduce, void* user) {
  return std::unique_ptr<Comm>(new CallbackComm(rank, nranks, p2p, allreduce, user));
}

}  // namespace fg
