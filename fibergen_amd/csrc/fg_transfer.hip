// HostStager: pipelined host <-> device transfers of padded fields (see fg_transfer.h).
#include "fg_transfer.h"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <thread>

#include "fg_hip_util.h"

namespace fg {

namespace {

// rows of `len` doubles, `pitch` apart -> contiguous (and back); one thread per pair of doubles where the rows allow
template <bool TO_STAGE>
__global__ __launch_bounds__(256) void k_rows(double* padded, long pitch, double* stage, long len, long total) {
  const long step = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const long r = i / len, c = i - r * len;
    if (TO_STAGE) stage[i] = padded[r * pitch + c];
    else padded[r * pitch + c] = stage[i];
  }
}

template <bool TO_STAGE>
void launch_rows(double* padded, long pitch, double* stage, long len, long nrows, hipStream_t s) {
  const long total = len * nrows;
  if (total == 0) return;
  const long blocks = std::min<long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL((k_rows<TO_STAGE>), dim3((unsigned)blocks), dim3(256), 0, s, padded, pitch, stage, len, total);
  FG_HIP_CHECK(hipGetLastError());
}

// The host side of the pipeline: T threads, thread t moves slice t of every chunk once the chunk is released.
class CopyTeam {
 public:
  struct Job {
    char* dst;
    const char* src;
    size_t bytes;
  };
  CopyTeam(std::vector<Job> jobs, int nthreads) : jobs_(std::move(jobs)), n_(nthreads), finished_(new std::atomic<int>[jobs_.size() + 1]) {
    for (size_t i = 0; i <= jobs_.size(); ++i) finished_[i].store(0, std::memory_order_relaxed);
    for (int t = 0; t < n_; ++t) threads_.emplace_back([this, t] { work(t); });
  }
  ~CopyTeam() {
    abort_.store(true, std::memory_order_release);
    for (auto& th : threads_) th.join();
  }
  void release(long upto) { released_.store(upto, std::memory_order_release); }   // chunks [0, upto) may be copied
  void wait_finished(long i) const {
    unsigned spins = 0;
    while (finished_[i].load(std::memory_order_acquire) < n_) pause(spins);
  }

 private:
  static void pause(unsigned& spins) {
    if (++spins < 64) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  void work(int t) {
    for (size_t i = 0; i < jobs_.size(); ++i) {
      unsigned spins = 0;
      while (released_.load(std::memory_order_acquire) <= (long)i) {
        if (abort_.load(std::memory_order_acquire)) return;
        pause(spins);
      }
      // slices on 4-KB boundaries of the chunk
      const size_t pages = (jobs_[i].bytes + 4095) / 4096;
      const size_t lo = std::min(jobs_[i].bytes, pages * t / n_ * 4096), hi = std::min(jobs_[i].bytes, pages * (t + 1) / n_ * 4096);
      if (hi > lo) std::memcpy(jobs_[i].dst + lo, jobs_[i].src + lo, hi - lo);
      finished_[i].fetch_add(1, std::memory_order_release);
    }
  }
  std::vector<Job> jobs_;
  int n_;
  std::unique_ptr<std::atomic<int>[]> finished_;
  std::atomic<long> released_{0};
  std::atomic<bool> abort_{false};
  std::vector<std::thread> threads_;
};

}  // namespace

int HostStager::host_threads() {
  cpu_set_t set;
  int n = 1;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  return std::max(1, std::min(8, n - 1));   // (12 threads measured equal: 16.6-19.2 ms per 805 MB either way)
}

HostStager& HostStager::of_device(int device) {
  static std::mutex mu;
  static HostStager* table[kMaxDevices] = {};
  if (device < 0 || device >= kMaxDevices) throw std::runtime_error("device index out of range");
  std::lock_guard<std::mutex> lk(mu);
  if (!table[device]) table[device] = new HostStager(device);
  return *table[device];
}

HostStager::HostStager(int device) : device_(device) {
  FG_HIP_CHECK(hipSetDevice(device_));
  FG_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  for (int b = 0; b < kBuffers; ++b) {
    FG_HIP_CHECK(hipEventCreateWithFlags(&done_[b], hipEventDisableTiming));
    FG_HIP_CHECK(hipHostMalloc(&pinned_[b], kCapacity));
    FG_HIP_CHECK(hipMalloc(&dstage_[b], kCapacity));
  }
}

std::vector<HostStager::Chunk> HostStager::cut(const std::vector<RowBlock>& blocks, long len, long pitch, size_t chunk_bytes) {
  if (len <= 0 || pitch < len) throw std::runtime_error("HostStager: bad row geometry");
  chunk_bytes = std::min(chunk_bytes, kCapacity);
  const long rows_per_chunk = std::max<long>(1, (long)(chunk_bytes / (len * sizeof(double))));
  if ((size_t)len * sizeof(double) > kCapacity) throw std::runtime_error("HostStager: a row exceeds the staging buffer");
  std::vector<Chunk> out;
  for (const RowBlock& b : blocks)
    for (long r = 0; r < b.nrows; r += rows_per_chunk)
      out.push_back(Chunk{b.dev + r * pitch, b.host + r * len, std::min(rows_per_chunk, b.nrows - r)});
  return out;
}

void HostStager::download(const std::vector<RowBlock>& blocks, long len, long pitch, size_t chunk_bytes) {
  std::lock_guard<std::mutex> lk(mu_);
  FG_HIP_CHECK(hipSetDevice(device_));
  const std::vector<Chunk> ch = cut(blocks, len, pitch, chunk_bytes);
  const long n = (long)ch.size();
  if (n == 0) return;
  std::vector<CopyTeam::Job> jobs;
  for (long i = 0; i < n; ++i)
    jobs.push_back({reinterpret_cast<char*>(ch[i].host), reinterpret_cast<const char*>(pinned_[i % kBuffers]),
                    (size_t)(ch[i].nrows * len) * sizeof(double)});
  CopyTeam team(std::move(jobs), host_threads());
  for (long i = 0; i <= n; ++i) {
    if (i < n) {
      const int b = (int)(i % kBuffers);
      if (i >= kBuffers) team.wait_finished(i - kBuffers);   // the pinned buffer has been emptied
      launch_rows<true>(ch[i].dev, pitch, dstage_[b], len, ch[i].nrows, stream_);
      FG_HIP_CHECK(hipMemcpyAsync(pinned_[b], dstage_[b], (size_t)(ch[i].nrows * len) * sizeof(double), hipMemcpyDeviceToHost, stream_));
      FG_HIP_CHECK(hipEventRecord(done_[b], stream_));
    }
    if (i >= 1) {
      FG_HIP_CHECK(hipEventSynchronize(done_[(i - 1) % kBuffers]));
      team.release(i);   // chunk i - 1 is in its pinned buffer
    }
  }
  team.wait_finished(n - 1);
}

void HostStager::upload(const std::vector<RowBlock>& blocks, long len, long pitch, size_t chunk_bytes) {
  std::lock_guard<std::mutex> lk(mu_);
  FG_HIP_CHECK(hipSetDevice(device_));
  const std::vector<Chunk> ch = cut(blocks, len, pitch, chunk_bytes);
  const long n = (long)ch.size();
  if (n == 0) return;
  std::vector<CopyTeam::Job> jobs;
  for (long i = 0; i < n; ++i)
    jobs.push_back({reinterpret_cast<char*>(pinned_[i % kBuffers]), reinterpret_cast<const char*>(ch[i].host),
                    (size_t)(ch[i].nrows * len) * sizeof(double)});
  CopyTeam team(std::move(jobs), host_threads());
  long released = std::min<long>(kBuffers, n);
  team.release(released);
  for (long i = 0; i < n; ++i) {
    const int b = (int)(i % kBuffers);
    // chunk i + kBuffers - 1 may be staged once chunk i - 1 has left its pinned buffer and been expanded on the device
    if (i >= 1 && released < n) {
      FG_HIP_CHECK(hipEventSynchronize(done_[(i - 1) % kBuffers]));
      team.release(++released);
    }
    team.wait_finished(i);
    FG_HIP_CHECK(hipMemcpyAsync(dstage_[b], pinned_[b], (size_t)(ch[i].nrows * len) * sizeof(double), hipMemcpyHostToDevice, stream_));
    launch_rows<false>(ch[i].dev, pitch, dstage_[b], len, ch[i].nrows, stream_);
    FG_HIP_CHECK(hipEventRecord(done_[b], stream_));
  }
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
}

}  // namespace fg
