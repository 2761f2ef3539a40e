// Communication layer of the slab-decomposed solver: point-to-point exchanges (all-to-all blocks of
// the pencil transpose, +-1 halo planes) and tiny all-reduces (norms, means, min / max), all stream-ordered.
//
// Three transports behind one interface:
//   * RcclComm   -- one process per GPU, RCCL over xGMI (librccl is dlopen'ed on first use, so single-GPU users never
//                   load it): ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd per exchange, ncclAllReduce.
//   * LocalHub   -- all slabs in ONE process on ONE device and ONE stream (tests on a single GPU, bench.py
//                   --force-slab): device-to-device copies once every member has posted its part.
//   * CallbackComm -- the caller moves the bytes (multi-process tests over gloo, staged through the host).
// The op lists come from slab_plan() (fg_slab_plan.h) for every transport, so what is tested on one GPU is what
// RCCL executes.
//
// Rule for callers: within one step a comm call is the LAST thing enqueued that later work of the same step could depend
// on; its results are consumed in a later step (LocalHub executes an exchange when the last member posts it).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <memory>
#include <string>
#include <vector>

#include "../../include/fibergen_amd.h"

namespace fg {

struct XOp {
  int send;      // 1 = send, 0 = receive
  int peer;      // != rank()
  void* ptr;     // device pointer
  size_t bytes;  // multiple of 8
};

class Comm {
 public:
  virtual ~Comm() {}
  virtual int rank() const = 0;
  virtual int size() const = 0;
  // Sends to / receives from one peer are matched in issue order (NCCL semantics).
  // CONTRACT -- receiver-gated: bytes reach a receive buffer of THIS rank no earlier than this rank's own exchange() call that
  // names the buffer has been reached on `stream` (RCCL: the receive is a kernel on the receiver's stream; LocalHub: one
  // stream for all members; CallbackComm: synchronises).  The slab driver relies on it: the y-slab spectrum lands in fu_
  // (Solver::slab_buffer, FG_BUF_SPECTRUM_Y), which holds the divergence field until the forward y pass of the component has
  // consumed it, and is written again only behind comm_wait(kXHaloU).  A transport that PUSHES into a peer's buffer when the
  // sender is ready (put / IPC writes) breaks that aliasing and must report pushes() = true: the driver then keeps the
  // spectrum in the second half of the polarisation field instead (one more field in the working set of a pass).
  virtual void exchange(const XOp* ops, int n, hipStream_t stream) = 0;
  virtual bool pushes() const { return false; }
  // in place on a device buffer; min_op: element-wise minimum instead of the sum
  virtual void allreduce(double* buf, int n, bool min_op, hipStream_t stream) = 0;
  // collectives issued between the two calls may be fused into one launch (RCCL group); no-ops elsewhere
  virtual void group_begin() {}
  virtual void group_end() {}
  virtual const char* name() const = 0;
};

// ---- RCCL ----------------------------------------------------------------------------------------------------------
constexpr int kUniqueIdBytes = 128;
void rccl_unique_id(char* out128);   // ncclGetUniqueId (throws if librccl cannot be loaded)
std::unique_ptr<Comm> make_rccl_comm(const char* id128, int rank, int nranks, int device);

// ---- in-process group ----------------------------------------------------------------------------------------------
class LocalHub;
std::shared_ptr<LocalHub> make_local_hub(int nranks);
std::unique_ptr<Comm> make_local_comm(std::shared_ptr<LocalHub> hub, int rank);

// ---- caller-driven -------------------------------------------------------------------------------------------------
std::unique_ptr<Comm> make_callback_comm(int rank, int nranks, fg_exchange_fn p2p, fg_allreduce_fn allreduce, void* user);

}  // namespace fg
