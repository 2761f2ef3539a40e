// Host <-> device hand-over of padded fields at the boundary (fg_set_phase / fg_set_normals / fg_set_field / fg_get_field).
//
// The reference's GetField / SetField F:26931-27010 copy row by row between the NumPy array and the TensorField; the first
// form of this boundary mirrored that with hipMemcpy2D straight into pageable memory (2-KB rows: 16 GB/s).  HostStager moves
// the same bytes as a pipeline: a device kernel strips (adds) the row padding into (out of) a contiguous staging buffer,
// the copy engine moves whole chunks between it and pinned host buffers, and a team of host threads copies between the pinned
// buffers and the caller's pageable array while the next chunks are on the link.  Values are byte-identical by construction.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <mutex>
#include <vector>

namespace fg {

// one run of `nrows` rows of `len` doubles: device rows `pitch` doubles apart, host rows contiguous
struct RowBlock {
  double* dev;
  double* host;   // read-only for uploads
  long nrows;
};

class HostStager {
 public:
  static constexpr size_t kCapacity = 16u << 20;   // bytes per staging buffer (three pinned + three device buffers)

  // the stager of a device: created on first use, shared by the solvers of the process (transfers are serialised), never
  // freed (pinned allocations cost tens of milliseconds; the runtime may be gone when static destructors run)
  static HostStager& of_device(int device);

  // device (padded, pitch) -> host (contiguous); the caller has synchronised the producing stream.
  // chunk_bytes <= kCapacity: size of a pipeline stage (tests shrink it to run many stages on small fields)
  void download(const std::vector<RowBlock>& blocks, long len, long pitch, size_t chunk_bytes = kCapacity);
  // host (contiguous) -> device (padded, pitch); padding columns are left untouched; complete on return
  void upload(const std::vector<RowBlock>& blocks, long len, long pitch, size_t chunk_bytes = kCapacity);

  static int host_threads();   // size of the copy team (1 ... 8, from the process's CPU affinity)

 private:
  explicit HostStager(int device);
  HostStager(const HostStager&) = delete;
  HostStager& operator=(const HostStager&) = delete;
  struct Chunk {
    double* dev;
    double* host;
    long nrows;
  };
  static std::vector<Chunk> cut(const std::vector<RowBlock>& blocks, long len, long pitch, size_t chunk_bytes);

  static constexpr int kBuffers = 3;
  int device_;
  std::mutex mu_;
  hipStream_t stream_ = nullptr;
  hipEvent_t done_[kBuffers] = {nullptr, nullptr, nullptr};
  double* pinned_[kBuffers] = {nullptr, nullptr, nullptr};
  double* dstage_[kBuffers] = {nullptr, nullptr, nullptr};
};

}  // namespace fg
