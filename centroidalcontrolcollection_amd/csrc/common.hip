// common.hip -- error reporting and device selection of libccc_amd.
#include "common.h"

#include <algorithm>

#include <cstring>

namespace ccc_amd
{
std::string & last_error()
{
  static thread_local std::string err;
  return err;
}

__global__ void zero_words_kernel(unsigned * p, int words)
{
  for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) p[i] = 0u;
}

int zero_words(void * p, int words, void * stream)
{
  if(words <= 0) return CCC_OK;
  const int blocks = words < 256 * 64 ? (words + 255) / 256 : 64;
  hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     static_cast<unsigned *>(p), words);
  CCC_HIP_CHECK(hipGetLastError());
  return CCC_OK;
}

// Schedules from the last call's counts (round 5; csrc/zmp.hip: pivot trips per QP, csrc/xy.hip: sweeps per instance): a
// counting sort of the n items by count, largest first; within a count the order is whatever the atomics give -- the
// answers never depend on it.  Two launches over kOrderBlocks chunks of the items (one workgroup doing all of it took
// 85 us for the headline's 131072 QPs, a sixth of the solve): (1) every workgroup counts its chunk into a row of
// `scratch` [blocks][256]; (2) every workgroup finds where its items of each count start -- the items of larger counts of
// all chunks, then the same count of the chunks before it -- and places them.  Optionally zeroes `nwords` words and stores
// n in *count_out, so that a caller that needs those done in-stream as well does not pay a launch for each.
constexpr int kOrderBuckets = 256;
__device__ __forceinline__ int order_bucket(int t) { return kOrderBuckets - 1 - (t < 0 ? 0 : (t > kOrderBuckets - 1 ? kOrderBuckets - 1 : t)); }

// Does the history predict?  A kernel that keeps per-item counts can also keep, per item, |count now - count of the call
// before| (`diff`); the counting pass adds them up beside the counts themselves (two more words per chunk behind the table),
// and the verdict -- independent draws of a batch's counts differ by about 0.6 of their sum, the consecutive cycles of
// a closed loop by a few per cent: the history is trusted below kTrustNum / kTrustDen = a fifth -- goes to *flag: page-locked
// host memory that the launching code reads, without waiting, a call or two later.  (The per-chunk sums travel as 64-bit
// words: a chunk of the largest batch the callers allow, 2^30 items / 128 chunks x several hundred trips, overflows 32.)
constexpr long long kTrustNum = 1, kTrustDen = 5;
__device__ void order_verdict(const int * __restrict__ scratch, int nblk, int * flag, int tid)
{
  if(tid < 64)
  {
    long long a = 0, sum = 0;
    const long long * agree = reinterpret_cast<const long long *>(scratch + kOrderBlocks * kOrderBuckets);
    for(int b = tid; b < nblk; b += 64)
    {
      a += agree[2 * b];
      sum += agree[2 * b + 1];
    }
    for(int d = 32; d >= 1; d >>= 1)
    {
      a += __shfl_xor(a, d);
      sum += __shfl_xor(sum, d);
    }
    if(tid == 0 && sum > 0)
      __hip_atomic_store(flag, kTrustDen * a < kTrustNum * sum ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(64) void order_verdict_kernel(const int * __restrict__ scratch, int nblk, int * flag)
{
  order_verdict(scratch, nblk, flag, threadIdx.x);
}

__global__ __launch_bounds__(256) void order_count_kernel(const int * __restrict__ hist, int n, int chunk, int * __restrict__ scratch,
                                                          unsigned * __restrict__ zero, int nwords, int * count_out,
                                                          const int * __restrict__ diff)
{
  __shared__ int cnt[kOrderBuckets];
  __shared__ unsigned long long agree[2];
  const int tid = threadIdx.x, blk = blockIdx.x;
  cnt[tid] = 0;
  if(tid < 2) agree[tid] = 0ull;
  if(blk == 0)
  {
    for(int k = tid; k < nwords; k += 256) zero[k] = 0u;
    if(tid == 0 && count_out) *count_out = n;
  }
  __syncthreads();
  const int lo = blk * chunk, hi = lo + chunk < n ? lo + chunk : n;
  long long da = 0, ds = 0;
  for(int i = lo + tid; i < hi; i += 256)
  {
    const int t = hist[i];
    atomicAdd(&cnt[order_bucket(t)], 1);
    if(diff)
    {
      da += diff[i];
      ds += t;
    }
  }
  if(diff)
  {
    for(int d = 32; d >= 1; d >>= 1)
    {
      da += __shfl_xor(da, d);
      ds += __shfl_xor(ds, d);
    }
    if((tid & 63) == 0)
    {
      atomicAdd(&agree[0], (unsigned long long)da);
      atomicAdd(&agree[1], (unsigned long long)ds);
    }
  }
  __syncthreads();
  scratch[blk * kOrderBuckets + tid] = cnt[tid];
  if(diff && tid < 2) reinterpret_cast<long long *>(scratch + kOrderBlocks * kOrderBuckets)[2 * blk + tid] = (long long)agree[tid];
}

__global__ __launch_bounds__(256) void order_place_kernel(const int * __restrict__ hist, int n, int chunk, const int * __restrict__ scratch,
                                                          int * __restrict__ order, int * flag)
{
  __shared__ int pos[kOrderBuckets];
  __shared__ int wsum[4];
  const int tid = threadIdx.x, blk = blockIdx.x, nblk = gridDim.x;
  if(flag && blk == 0) order_verdict(scratch, nblk, flag, tid);
  // bucket tid: items of it in all chunks, and in the chunks before this one
  // (the loads of sixteen chunks in flight at a time: one after the other this loop was most of the sort's 20 us)
  int total = 0, before = 0;
  for(int b0 = 0; b0 < nblk; b0 += 16)
  {
    int c[16];
#pragma unroll
    for(int k = 0; k < 16; k++) c[k] = b0 + k < nblk ? scratch[(b0 + k) * kOrderBuckets + tid] : 0;
#pragma unroll
    for(int k = 0; k < 16; k++)
    {
      total += c[k];
      before += b0 + k < blk ? c[k] : 0;
    }
  }
  // exclusive prefix of `total` over the buckets (four wavefronts)
  int incl = total;
  for(int d = 1; d < 64; d <<= 1)
  {
    const int o = __shfl_up(incl, d);
    if((tid & 63) >= d) incl += o;
  }
  if((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int base = 0;
  for(int w = 0; w < (tid >> 6); ++w) base += wsum[w];
  pos[tid] = base + incl - total + before;
  __syncthreads();
  const int lo = blk * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for(int i = lo + tid; i < hi; i += 256) order[atomicAdd(&pos[order_bucket(hist[i])], 1)] = i;
}

int order_by_count(const int * hist, int n, int * order, int * scratch, void * zero, int nwords, int * count_out, void * stream,
                   const int * diff, int * flag)
{
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int blocks = std::max(1, std::min(kOrderBlocks, (n + 1023) / 1024));
  const int chunk = (n + blocks - 1) / blocks;
  hipLaunchKernelGGL(order_count_kernel, dim3(blocks), dim3(256), 0, s, hist, n, chunk, scratch, static_cast<unsigned *>(zero),
                     nwords, count_out, diff);
  if(!order) // (the verdict alone: a history that is not followed is still watched)
    hipLaunchKernelGGL(order_verdict_kernel, dim3(1), dim3(64), 0, s, scratch, blocks, flag);
  else
    hipLaunchKernelGGL(order_place_kernel, dim3(blocks), dim3(256), 0, s, hist, n, chunk, scratch, order, diff ? flag : nullptr);
  if(hipGetLastError() != hipSuccess) return fail(CCC_ERR_HIP, "order_by_count: launch failed");
  return CCC_OK;
}

int refuse_growth_in_capture(void * stream, const char * who)
{
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if(stream && hipStreamIsCapturing(reinterpret_cast<hipStream_t>(stream), &st) == hipSuccess &&
     st != hipStreamCaptureStatusNone)
    return fail(CCC_ERR_INVALID_ARGUMENT,
                "%s: the workspace has to grow for this batch size, which cannot happen inside a stream capture; "
                "call once eagerly with the largest batch before capturing",
                who);
  return CCC_OK;
}

int select_device(int device)
{
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if(e != hipSuccess || count <= 0)
    return fail(CCC_ERR_NO_DEVICE, "no HIP device visible (%s): libccc_amd has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if(device < 0 || device >= count)
    return fail(CCC_ERR_INVALID_ARGUMENT, "device %d out of range [0, %d)", device, count);
  hipDeviceProp_t prop;
  CCC_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if(std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(CCC_ERR_NO_DEVICE, "device %d is %s; libccc_amd is built for gfx950 (MI355X) only", device,
                prop.gcnArchName);
  return CCC_OK;
}
} // namespace ccc_amd

extern "C" const char * ccc_last_error_string(void)
{
  return ccc_amd::last_error().c_str();
}

extern "C" int ccc_abi_version(void)
{
  // 2: ccc_ddp_config_t::reg_type, sharded entry points, ccc_device_count; 3: max_ridges in ccc_ddp_params_t /
  // ccc_xy_params_t; 4: ccc_ddp_config_t::warm_start_guard (the struct grew), CCC_DDP_STATUS_WARM_REPLACED_BIT
  return CCC_ABI_VERSION;
}

extern "C" int ccc_device_count(void)
{
  int count = 0;
  if(hipGetDeviceCount(&count) != hipSuccess) return 0;
  int ok = 0;
  for(int d = 0; d < count; d++)
  {
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}
