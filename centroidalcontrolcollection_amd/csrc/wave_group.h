// wave_group.h -- lane-group collectives inside one 64-wide CDNA4 wavefront.
//
// A "group" is LG consecutive lanes (LG = 32: two groups per wavefront, LG = 64: the whole wavefront)
// that cooperate on one small problem, one lane per row.  Reductions run on the VALU cross-lane paths
// of gfx950 (DPP quad_perm / row_mirror inside a 16-lane row, v_permlane16_swap / v_permlane32_swap
// across rows) -- no LDS round trips, no barriers.
#pragma once

#include <hip/hip_runtime.h>

namespace ccc_amd
{
// DPP controls (gfx9 encoding)
constexpr int kDppQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141; // lane i <-> 7-i inside each 8 lanes
constexpr int kDppRowMirror = 0x140;     // lane i <-> 15-i inside each 16-lane row

template<int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
  // every lane of these controls has a valid source: no "old" operand to preserve (saves the copies
  // __builtin_amdgcn_update_dpp would need)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

template<int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}

// max of two doubles that are known not to be signalling NaNs: the bare instruction, without the
// canonicalisation hipcc wraps around fmax() for IEEE sNaN quieting
__device__ __forceinline__ double max_raw(double a, double b)
{
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max_raw(float a, float b)
{
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// 1/x to within a couple of ulp: v_rcp_f64 + two Newton steps (the IEEE division expansion costs 13 fp64 ops)
__device__ __forceinline__ double fast_rcp(double x)
{
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// a = rows {0,0,2,2} of v, b = rows {1,1,3,3} of v (a row = 16 lanes)
__device__ __forceinline__ void rows_pair16(double v, double & a, double & b)
{
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  a = __hiloint2double((int)rh[0], (int)rl[0]);
  b = __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ __forceinline__ void rows_pair16(float v, float & a, float & b)
{
  const unsigned x = (unsigned)__float_as_int(v);
  auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  a = __int_as_float((int)r[0]);
  b = __int_as_float((int)r[1]);
}

// a = lanes 0-31 of v in both halves, b = lanes 32-63 of v in both halves
__device__ __forceinline__ void halves_pair32(double v, double & a, double & b)
{
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  a = __hiloint2double((int)rh[0], (int)rl[0]);
  b = __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ __forceinline__ void halves_pair32(float v, float & a, float & b)
{
  const unsigned x = (unsigned)__float_as_int(v);
  auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  a = __int_as_float((int)r[0]);
  b = __int_as_float((int)r[1]);
}

template<int LG>
struct WaveGroup
{
  static_assert(LG == 32 || LG == 64, "a group is half a wavefront or a whole one");

  // max over the group, replicated in every lane of the group
  // (inputs must not be signalling NaNs; a quiet NaN input yields an unspecified non-signalling result)
  static __device__ __forceinline__ double max(double v)
  {
    v = max_raw(v, dpp_f64<kDppQuadXor1>(v));
    v = max_raw(v, dpp_f64<kDppQuadXor2>(v));
    v = max_raw(v, dpp_f64<kDppRowHalfMirror>(v));
    v = max_raw(v, dpp_f64<kDppRowMirror>(v));
    double a, b;
    rows_pair16(v, a, b);
    v = max_raw(a, b);
    if(LG == 64)
    {
      halves_pair32(v, a, b);
      v = max_raw(a, b);
    }
    return v;
  }

  static __device__ __forceinline__ float max(float v)
  {
    v = max_raw(v, dpp_f32<kDppQuadXor1>(v));
    v = max_raw(v, dpp_f32<kDppQuadXor2>(v));
    v = max_raw(v, dpp_f32<kDppRowHalfMirror>(v));
    v = max_raw(v, dpp_f32<kDppRowMirror>(v));
    float a, b;
    rows_pair16(v, a, b);
    v = max_raw(a, b);
    if(LG == 64)
    {
      halves_pair32(v, a, b);
      v = max_raw(a, b);
    }
    return v;
  }

  static __device__ __forceinline__ double min(double v)
  {
    return -max(-v);
  }

  static __device__ __forceinline__ double sum(double v)
  {
    v += dpp_f64<kDppQuadXor1>(v);
    v += dpp_f64<kDppQuadXor2>(v);
    v += dpp_f64<kDppRowHalfMirror>(v);
    v += dpp_f64<kDppRowMirror>(v);
    double a, b;
    rows_pair16(v, a, b);
    v = a + b;
    if(LG == 64)
    {
      halves_pair32(v, a, b);
      v = a + b;
    }
    return v;
  }

  // lowest group-relative lane index for which pred holds (LG if none), replicated in the group
  static __device__ __forceinline__ int first(bool pred)
  {
    const unsigned long long m = __ballot(pred);
    if(LG == 64) return m ? (int)__ffsll((long long)m) - 1 : 64;
    const unsigned mine = (threadIdx.x & 32) ? (unsigned)(m >> 32) : (unsigned)m;
    return mine ? __ffs((int)mine) - 1 : 32;
  }

  // pred as evaluated by lane `src` (group-relative, group uniform) of the caller's group
  static __device__ __forceinline__ bool bit(bool pred, int src)
  {
    const unsigned long long m = __ballot(pred);
    if(LG == 64) return (m >> src) & 1ull;
    const unsigned mine = (threadIdx.x & 32) ? (unsigned)(m >> 32) : (unsigned)m;
    return (mine >> src) & 1u;
  }

  static __device__ __forceinline__ bool any(bool pred)
  {
    const unsigned long long m = __ballot(pred);
    if(LG == 64) return m != 0ull;
    const unsigned mine = (threadIdx.x & 32) ? (unsigned)(m >> 32) : (unsigned)m;
    return mine != 0u;
  }

  // value held by lane `src` (group-relative) of the caller's group (LDS crossbar; use sparingly)
  static __device__ __forceinline__ double bcast(double v, int src)
  {
    return __shfl(v, src, LG);
  }
  static __device__ __forceinline__ int bcast(int v, int src)
  {
    return __shfl(v, src, LG);
  }
};
} // namespace ccc_amd
