// wave_group.h -- lane-group collectives inside one 64-wide CDNA wavefront.
//
// A "group" is LG consecutive lanes (LG = 32: two groups per wavefront, LG = 64: the whole wavefront)
// that cooperate on one small problem, one lane per row.  Everything here is register-to-register
// (ds_bpermute / DPP); no LDS storage and no barriers are involved.
#pragma once

#include <hip/hip_runtime.h>

namespace ccc_amd
{
template<int LG>
struct WaveGroup
{
  static_assert(LG == 32 || LG == 64, "a group is half a wavefront or a whole one");

  // value held by lane `src` (group-relative) of the caller's group
  static __device__ __forceinline__ double bcast(double v, int src)
  {
    return __shfl(v, src, LG);
  }
  static __device__ __forceinline__ int bcast(int v, int src)
  {
    return __shfl(v, src, LG);
  }

  // (max key, lowest index attaining it) over the group, replicated in every lane
  static __device__ __forceinline__ void argmax(double & key, int & idx)
  {
#pragma unroll
    for(int off = LG / 2; off > 0; off >>= 1)
    {
      double ok = __shfl_xor(key, off, LG);
      int oi = __shfl_xor(idx, off, LG);
      bool take = (ok > key) || (ok == key && oi < idx);
      key = take ? ok : key;
      idx = take ? oi : idx;
    }
  }

  static __device__ __forceinline__ void argmin(double & key, int & idx)
  {
#pragma unroll
    for(int off = LG / 2; off > 0; off >>= 1)
    {
      double ok = __shfl_xor(key, off, LG);
      int oi = __shfl_xor(idx, off, LG);
      bool take = (ok < key) || (ok == key && oi < idx);
      key = take ? ok : key;
      idx = take ? oi : idx;
    }
  }

  static __device__ __forceinline__ double sum(double v)
  {
#pragma unroll
    for(int off = LG / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, LG);
    return v;
  }

  static __device__ __forceinline__ bool any(bool pred)
  {
    unsigned long long m = __ballot(pred);
    if(LG == 64) return m != 0ull;
    unsigned long long mine = (threadIdx.x & 32) ? (m >> 32) : (m & 0xffffffffull);
    return mine != 0ull;
  }

  // maximum over the whole wavefront of a value that is uniform inside each group
  static __device__ __forceinline__ int wave_max_of_group_uniform(int v)
  {
    int a = __builtin_amdgcn_readlane(v, 0);
    if(LG == 64) return a;
    int b = __builtin_amdgcn_readlane(v, 32);
    return a > b ? a : b;
  }
};
} // namespace ccc_amd
