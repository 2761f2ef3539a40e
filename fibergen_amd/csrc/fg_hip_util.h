// HIP error handling: internal code throws, the C ABI layer (fg_capi.cpp)
// converts to return codes -- no exception crosses the extern "C" boundary.
#pragma once

#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>

#define FG_HIP_CHECK(expr)                                                                          \
  do {                                                                                              \
    hipError_t fg_err__ = (expr);                                                                   \
    if (fg_err__ != hipSuccess)                                                                     \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(fg_err__) + " at " + \
                               __FILE__ + ":" + std::to_string(__LINE__) + " in " #expr);          \
  } while (0)

#include <atomic>
#include <exception>
#include <mutex>

namespace fg {

// Per-device "done once" flag for launch attributes (hipFuncSetAttribute is per device) and the cached CU count:
// one process may hold solvers on several devices (fg_create(..., device)), and distinct solvers may be driven from
// distinct threads.
constexpr int kMaxDevices = 64;

inline int current_device() {
  int d = 0;
  FG_HIP_CHECK(hipGetDevice(&d));
  return d < 0 || d >= kMaxDevices ? 0 : d;
}

// first_use() hands the first caller on a device a guard that keeps the other callers of that device waiting until the
// guarded block (hipFuncSetAttribute ...) has run; the flag is set only when the block left without an exception:
//     static PerDeviceOnce configured;
//     if (auto once = configured.first_use()) { FG_HIP_CHECK(hipFuncSetAttribute(...)); }
struct PerDeviceOnce {
  std::atomic<bool> done[kMaxDevices] = {};
  std::mutex mu;
  struct Guard {
    PerDeviceOnce* owner;
    int device;
    int exceptions;
    std::unique_lock<std::mutex> lock;
    explicit operator bool() const { return owner != nullptr; }
    ~Guard() {
      if (owner && std::uncaught_exceptions() == exceptions) owner->done[device].store(true, std::memory_order_release);
    }
  };
  Guard first_use() {
    const int d = current_device();
    if (done[d].load(std::memory_order_acquire)) return Guard{nullptr, d, 0, {}};
    std::unique_lock<std::mutex> lk(mu);
    if (done[d].load(std::memory_order_acquire)) return Guard{nullptr, d, 0, {}};
    return Guard{this, d, std::uncaught_exceptions(), std::move(lk)};
  }
};

inline int device_cu_count() {
  static std::atomic<int> cus[kMaxDevices] = {};
  const int d = current_device();
  int c = cus[d].load();
  if (!c) {
    FG_HIP_CHECK(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d));
    cus[d].store(c);
  }
  return c;
}

}  // namespace fg
