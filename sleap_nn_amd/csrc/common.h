// Shared helpers for libposehip (host side error plumbing + small device utilities).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/posehip.h"

namespace ph {

void set_error(const char* fmt, ...);

#define PH_HIP_CHECK(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      ph::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return PH_E_HIP;                                                                       \
    }                                                                                        \
  } while (0)

#define PH_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      ph::set_error(__VA_ARGS__);    \
      return PH_E_INVALID;           \
    }                                \
  } while (0)

// Compute units of the CURRENT device (persistent kernels launch one workgroup per CU).  Cached per device index, so a process
// that drives two different devices gets each one's own count.
int device_cu_count(int* out);

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int pad16(int c) { return (c + 15) / 16 * 16; }

}  // namespace ph
