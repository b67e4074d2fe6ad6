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

// Check that `device` exists and is a gfx950 part: the product path has no CPU (or other-arch) fallback and must fail
// loudly instead.  Does NOT change the calling thread's current device (see DeviceGuard).
int select_device(int device);

// Every C-ABI entry point works on its handle's device and restores the caller's current HIP device on return (a host
// that drives several GPUs from one thread -- or torch, which reads the runtime's current device -- must not find it
// switched behind its back).
struct DeviceGuard
{
  int prev = -1, target = -1;
  bool ok = false;
  explicit DeviceGuard(int device) : target(device)
  {
    ok = hipGetDevice(&prev) == hipSuccess && (prev == device || hipSetDevice(device) == hipSuccess);
  }
  ~DeviceGuard()
  {
    if(prev >= 0 && prev != target) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard & operator=(const DeviceGuard &) = delete;
};

// Zero `words` 32-bit words at `p` on `stream` with a kernel of the library's own.  The per-launch counters (work-list
// lengths, tickets) are reset with this rather than hipMemsetAsync: captured in a hipGraph, a 4-byte memset node
// followed a foreign kernel with a memory access fault on replay (ROCm 7.2; tests/test_graph_capture_gpu.py).
int zero_words(void * p, int words, void * stream);
// order[0 .. n) = the items sorted by hist[], largest first (in-stream, two launches; see common.hip); `scratch`:
// kOrderScratchInts ints of the caller's; optionally zeroes nwords words at `zero` and stores n in *count_out on the way
constexpr int kOrderBlocks = 128;
constexpr int kOrderScratchInts = kOrderBlocks * 256 + 4 * kOrderBlocks; // (the table, then two 64-bit sums per chunk)
// diff / flag (optional): per-item |count - count of the call before| and where the verdict "the history predicts" goes
// (page-locked host memory; see common.hip); order = nullptr: the verdict alone
int order_by_count(const int * hist, int n, int * order, int * scratch, void * zero, int nwords, int * count_out, void * stream,
                   const int * diff = nullptr, int * flag = nullptr);

// The `_device` entry points are asynchronous and graph-capturable EXCEPT when a handle's workspace has to grow (a batch
// larger than any seen before): hipMalloc / hipFree synchronise the device and invalidate an active capture.  Growth
// during a capture is refused with CCC_ERR_INVALID_ARGUMENT instead (call once eagerly with the largest batch first).
int refuse_growth_in_capture(void * stream, const char * who);
} // namespace ccc_amd

#define CCC_DEVICE_GUARD(device)                                                                   \
  ccc_amd::DeviceGuard ccc_device_guard__(device);                                                 \
  if(!ccc_device_guard__.ok) return ccc_amd::fail(CCC_ERR_HIP, "cannot select HIP device %d", (int)(device))

#define CCC_NO_CAPTURE(stream, who)                                                      \
  do                                                                                     \
  {                                                                                      \
    if(int rc__ = ccc_amd::refuse_growth_in_capture((void *)(stream), who)) return rc__; \
  } while(0)

#define CCC_HIP_CHECK(expr)                                                                              \
  do                                                                                                     \
  {                                                                                                      \
    hipError_t err__ = (expr);                                                                           \
    if(err__ != hipSuccess)                                                                              \
      return ccc_amd::fail(CCC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), __FILE__, \
                           __LINE__);                                                                    \
  } while(0)
