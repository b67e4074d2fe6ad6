// common.h -- shared host-side helpers of libccc_amd (error reporting, HIP call checking).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/ccc_amd.h"

namespace ccc_amd
{
// Last error text of the calling thread (returned by ccc_last_error_string()).
std::string & last_error();

inline int fail(int code, const char * fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

// Select `device` after checking that it exists and is a gfx950 part: the product path has no CPU
// (or other-arch) fallback and must fail loudly instead.
int select_device(int device);

// Zero `words` 32-bit words at `p` on `stream` with a kernel of the library's own.  The per-launch counters (work-list
// lengths, tickets) are reset with this rather than hipMemsetAsync: captured in a hipGraph, a 4-byte memset node
// followed a foreign kernel with a memory access fault on replay (ROCm 7.2; tests/test_graph_capture_gpu.py).
int zero_words(void * p, int words, void * stream);
} // namespace ccc_amd

#define CCC_HIP_CHECK(expr)                                                                              \
  do                                                                                                     \
  {                                                                                                      \
    hipError_t err__ = (expr);                                                                           \
    if(err__ != hipSuccess)                                                                              \
      return ccc_amd::fail(CCC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), __FILE__, \
                           __LINE__);                                                                    \
  } while(0)
