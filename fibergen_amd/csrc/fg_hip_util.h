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

struct PerDeviceOnce {
  std::atomic<bool> done[kMaxDevices] = {};
  bool first_use() { return !done[current_device()].exchange(true); }
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
