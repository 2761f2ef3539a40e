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
