// w64.h -- "wave vector" types for kernels written as ONE 64-lane wavefront in lock step.
//
// Device build (hipcc, gfx950): vf / vi / vb ARE double / int / bool -- one value per lane in a VGPR -- and the
// cross-lane primitives are the CDNA4 ones (v_readlane, DPP row operations inside the 16-lane rows,
// v_permlane16_swap / v_permlane32_swap across rows, ds_read / ds_write for LDS).  Zero cost: everything inlines.
//
// Host build (g++, tests/emu): vf / vi / vb are 64-element arrays with the same operators, the cross-lane primitives are
// loops over the lanes.  A kernel written against this header therefore runs unchanged on the host, 64 lanes in lock
// step, with the same IEEE operations in the same order (no FMA contraction on either side; fma only where the source
// says vfma) -- which is what lets the CPU test suite check a hand-written kernel bit for bit without a GPU.  The host
// build is a TEST AID (tests/emu); the product only ever runs the device build.
//
// Rules for kernel code: control flow only on wave-uniform values (plain C++ scalars, ballot masks); per-lane choices
// are sel(mask, a, b); per-lane memory access goes through ld / st / lds_ld / lds_st.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIP_DEVICE_COMPILE__)
#  define W64_DEVICE 1
#  include <hip/hip_runtime.h>
#  define W64_FN __device__ __forceinline__
#else
#  define W64_DEVICE 0
#  if defined(__HIPCC__)
#    define W64_FN __host__ __device__ inline // (hipcc's host pass only parses the kernels that use this header)
#  else
#    define W64_FN inline
#  endif
#endif

#if defined(__clang__)
#  pragma clang fp contract(off)
#endif

// loop unroll control that both compilers take
#define W64_PRAGMA_(x) _Pragma(#x)
#if defined(__clang__)
#  define W64_UNROLL(n) W64_PRAGMA_(unroll n)
#  define W64_UNROLL_T(n) W64_PRAGMA_(unroll n) // n: a template-dependent constant
#else
#  define W64_UNROLL(n) W64_PRAGMA_(GCC unroll n)
#  define W64_UNROLL_T(n) // (g++ 11 takes literals only; the host build is the emulation: unrolling does not matter)
#endif

namespace w64
{
#if W64_DEVICE
// =========================================================================================== device
using vf = double;
using vi = int;
using vb = bool;

W64_FN vi lane_id() { return static_cast<int>(threadIdx.x & 63); }
W64_FN vf splat(double x) { return x; }
W64_FN vi spl(int x) { return x; }
W64_FN vf sel(vb m, vf a, vf b) { return m ? a : b; }
W64_FN vi seli(vb m, vi a, vi b) { return m ? a : b; }
W64_FN vf vfma(vf a, vf b, vf c) { return __builtin_fma(a, b, c); }
W64_FN vf vsqrt(vf a) { return sqrt(a); }
W64_FN vf vabs(vf a) { return fabs(a); }
W64_FN vf vmin(vf a, vf b) { return fmin(a, b); }
W64_FN vf vmax(vf a, vf b) { return fmax(a, b); }
W64_FN vf vfloor(vf a) { return floor(a); }
W64_FN vi to_int(vf a) { return static_cast<int>(a); }
W64_FN unsigned long long ballot(vb m) { return __ballot(m); }

// value of lane k (wave-uniform k) as a scalar
W64_FN double read_lane(vf v, int k)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}
W64_FN int read_lane_i(vi v, int k) { return __builtin_amdgcn_readlane(v, k); }

template<int CTRL>
W64_FN vf dpp(vf v)
{
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// lane K of each 16-lane row to every lane of that row (DPP row_newbcast, gfx90a+)
template<int K>
W64_FN vf row_bcast(vf v)
{
  return dpp<0x150 + K>(v);
}
template<int K>
W64_FN vi row_bcast_i(vi v)
{
  return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xf, 0xf, true);
}
// lane (c + K) mod 16 of the same row (DPP row_ror: lane c reads lane c - K ... the hardware's rotate-right moves
// data towards higher lanes, so lane c receives the value of lane (c - K) mod 16)
template<int K>
W64_FN vf row_from_lower(vf v)
{
  return dpp<0x120 + K>(v);
}
W64_FN vf x1(vf v) { return dpp<0xB1>(v); }   // lane c ^ 1   quad_perm:[1,0,3,2]
W64_FN vf x2(vf v) { return dpp<0x4E>(v); }   // lane c ^ 2   quad_perm:[2,3,0,1]
W64_FN vf hm(vf v) { return dpp<0x141>(v); }  // row_half_mirror: c <-> 7 - c inside each 8 lanes
W64_FN vf rm(vf v) { return dpp<0x140>(v); }  // row_mirror:      c <-> 15 - c inside each row

// a = rows {0,0,2,2} of v, b = rows {1,1,3,3}
W64_FN void rows_pair(vf v, vf & a, vf & b)
{
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  a = __hiloint2double((int)rh[0], (int)rl[0]);
  b = __hiloint2double((int)rh[1], (int)rl[1]);
}
// a = lanes 0-31 of v in both halves, b = lanes 32-63
W64_FN void halves_pair(vf v, vf & a, vf & b)
{
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  a = __hiloint2double((int)rh[0], (int)rl[0]);
  b = __hiloint2double((int)rh[1], (int)rl[1]);
}

// per-lane memory access (global or LDS: plain pointers on the device)
// (base pointer + a 32-bit unsigned BYTE offset: the form the global-memory instructions take as "SGPR base + VGPR
//  offset" -- with a signed 64-bit index every lane carries a 64-bit address, and those addresses are what the register
//  allocator then spills.  Indices are non-negative and below 2^29 doubles.)
W64_FN vf ld(const double * p, vi idx)
{
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(p) + (static_cast<unsigned>(idx) << 3));
}
// masked load WITHOUT a branch: lanes outside the mask read element 0 (always valid) and drop it.  As `m ? p[idx] : other`
// every such load became a basic block of its own with an s_waitcnt vmcnt(0) behind it: a row of nine prefetches turned
// into nine serialised memory round trips.
W64_FN vf ld_if(const double * p, vi idx, vb m, double other = 0.0)
{
  const vf v = ld(p, m ? idx : 0);
  return m ? v : other;
}
// the same as a predicated load (exec-masked branch).  Fewer registers -- no address select, nothing in flight beside
// the load -- at the price of a serialised round trip per load: measured faster where the kernel is short of registers
// rather than of overlap (the 12-state DDP kernel at large batches: 105 k against 91 k solves/s), slower where latency
// binds (the 9-state kernel at one instance per wavefront slot: 36.3 k against 39.6 k).
W64_FN vf ld_if_branch(const double * p, vi idx, vb m, double other = 0.0) { return m ? p[idx] : other; }
W64_FN void st(double * p, vi idx, vf v, vb m)
{
  if(m) p[idx] = v;
}
W64_FN vi ldi(const int * p, vi idx) { return p[idx]; }
W64_FN void sth(unsigned short * p, vi idx, vi v, vb m)
{
  if(m) p[idx] = static_cast<unsigned short>(v);
}
// a value that is the same on every lane, as a scalar
W64_FN int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
W64_FN vi ldb(const unsigned char * p, vi idx) { return static_cast<int>(p[idx]); }
// the same value, but one the compiler cannot trace: what is computed from it is computed HERE, not hoisted out of the
// enclosing loops into registers that live (or spill) across them
W64_FN vi opaque(vi v)
{
  asm volatile("" : "+v"(v));
  return v;
}
// entry idx (0 .. 15) of a table of sixteen 4-bit values held in one 64-bit constant
W64_FN vi tbl4(unsigned long long t, vi idx) { return static_cast<int>((t >> (4 * idx)) & 15ull); }
// Orders this wavefront's LDS traffic: the LDS executes one wavefront's operations in order, so all that is needed is that
// the COMPILER keeps the accesses on their side of this point (a wavefront-scope fence + scheduling barrier: no
// s_barrier, and above all no s_waitcnt vmcnt(0) -- __syncthreads() would drain every global load in flight here).
// Kernels written against this header run ONE wavefront per workgroup.
W64_FN void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// Orders this wavefront's GLOBAL traffic as well (data written by some lanes and read by others through memory)
W64_FN void mem_sync() { __syncthreads(); }

#else
// =========================================================================================== host (64 lanes in lock step)
constexpr int kLanes = 64;
struct vf
{
  double v[kLanes];
};
struct vi
{
  int v[kLanes];
};
struct vb
{
  bool v[kLanes];
};
#  define W64_LOOP for(int l_ = 0; l_ < kLanes; ++l_)
inline vi lane_id()
{
  vi r;
  W64_LOOP r.v[l_] = l_;
  return r;
}
inline vf splat(double x)
{
  vf r;
  W64_LOOP r.v[l_] = x;
  return r;
}
inline vi spl(int x)
{
  vi r;
  W64_LOOP r.v[l_] = x;
  return r;
}
#  define W64_BIN_F(op)                                   \
    inline vf operator op(const vf & a, const vf & b)     \
    {                                                     \
      vf r;                                               \
      W64_LOOP r.v[l_] = a.v[l_] op b.v[l_];              \
      return r;                                           \
    }                                                     \
    inline vf operator op(const vf & a, double b)         \
    {                                                     \
      vf r;                                               \
      W64_LOOP r.v[l_] = a.v[l_] op b;                    \
      return r;                                           \
    }                                                     \
    inline vf operator op(double a, const vf & b)         \
    {                                                     \
      vf r;                                               \
      W64_LOOP r.v[l_] = a op b.v[l_];                    \
      return r;                                           \
    }
W64_BIN_F(+)
W64_BIN_F(-)
W64_BIN_F(*)
W64_BIN_F(/)
#  undef W64_BIN_F
inline vf operator-(const vf & a)
{
  vf r;
  W64_LOOP r.v[l_] = -a.v[l_];
  return r;
}
inline vf & operator+=(vf & a, const vf & b) { return a = a + b; }
inline vf & operator-=(vf & a, const vf & b) { return a = a - b; }
inline vf & operator*=(vf & a, const vf & b) { return a = a * b; }
inline vf & operator*=(vf & a, double b) { return a = a * b; }
#  define W64_CMP_F(op)                                   \
    inline vb operator op(const vf & a, const vf & b)     \
    {                                                     \
      vb r;                                               \
      W64_LOOP r.v[l_] = a.v[l_] op b.v[l_];              \
      return r;                                           \
    }                                                     \
    inline vb operator op(const vf & a, double b)         \
    {                                                     \
      vb r;                                               \
      W64_LOOP r.v[l_] = a.v[l_] op b;                    \
      return r;                                           \
    }
W64_CMP_F(<)
W64_CMP_F(>)
W64_CMP_F(<=)
W64_CMP_F(>=)
W64_CMP_F(==)
W64_CMP_F(!=)
#  undef W64_CMP_F
#  define W64_BIN_I(op)                                   \
    inline vi operator op(const vi & a, const vi & b)     \
    {                                                     \
      vi r;                                               \
      W64_LOOP r.v[l_] = a.v[l_] op b.v[l_];              \
      return r;                                           \
    }                                                     \
    inline vi operator op(const vi & a, int b)            \
    {                                                     \
      vi r;                                               \
      W64_LOOP r.v[l_] = a.v[l_] op b;                    \
      return r;                                           \
    }                                                     \
    inline vi operator op(int a, const vi & b)            \
    {                                                     \
      vi r;                                               \
      W64_LOOP r.v[l_] = a op b.v[l_];                    \
      return r;                                           \
    }
W64_BIN_I(+)
W64_BIN_I(-)
W64_BIN_I(*)
W64_BIN_I(&)
W64_BIN_I(>>)
W64_BIN_I(<<)
W64_BIN_I(|)
#  undef W64_BIN_I
#  define W64_CMP_I(op)                                   \
    inline vb operator op(const vi & a, const vi & b)     \
    {                                                     \
      vb r;                                               \
      W64_LOOP r.v[l_] = a.v[l_] op b.v[l_];              \
      return r;                                           \
    }                                                     \
    inline vb operator op(const vi & a, int b)            \
    {                                                     \
      vb r;                                               \
      W64_LOOP r.v[l_] = a.v[l_] op b;                    \
      return r;                                           \
    }
W64_CMP_I(<)
W64_CMP_I(>)
W64_CMP_I(<=)
W64_CMP_I(>=)
W64_CMP_I(==)
W64_CMP_I(!=)
#  undef W64_CMP_I
inline vb operator&&(const vb & a, const vb & b)
{
  vb r;
  W64_LOOP r.v[l_] = a.v[l_] && b.v[l_];
  return r;
}
inline vb operator||(const vb & a, const vb & b)
{
  vb r;
  W64_LOOP r.v[l_] = a.v[l_] || b.v[l_];
  return r;
}
inline vb operator!(const vb & a)
{
  vb r;
  W64_LOOP r.v[l_] = !a.v[l_];
  return r;
}
inline vb operator!=(const vb & a, const vb & b)
{
  vb r;
  W64_LOOP r.v[l_] = a.v[l_] != b.v[l_];
  return r;
}
inline vb operator&&(const vb & a, bool b)
{
  vb r;
  W64_LOOP r.v[l_] = a.v[l_] && b;
  return r;
}
inline vf sel(const vb & m, const vf & a, const vf & b)
{
  vf r;
  W64_LOOP r.v[l_] = m.v[l_] ? a.v[l_] : b.v[l_];
  return r;
}
inline vf sel(const vb & m, const vf & a, double b) { return sel(m, a, splat(b)); }
inline vf sel(const vb & m, double a, const vf & b) { return sel(m, splat(a), b); }
inline vf sel(const vb & m, double a, double b) { return sel(m, splat(a), splat(b)); }
inline vi seli(const vb & m, const vi & a, const vi & b)
{
  vi r;
  W64_LOOP r.v[l_] = m.v[l_] ? a.v[l_] : b.v[l_];
  return r;
}
inline vf vfma(const vf & a, const vf & b, const vf & c)
{
  vf r;
  W64_LOOP r.v[l_] = std::fma(a.v[l_], b.v[l_], c.v[l_]);
  return r;
}
inline vf vfma(double a, const vf & b, const vf & c) { return vfma(splat(a), b, c); }
inline vf vfma(const vf & a, double b, const vf & c) { return vfma(a, splat(b), c); }
inline vf vfma(const vf & a, const vf & b, double c) { return vfma(a, b, splat(c)); }
inline vf vfma(double a, const vf & b, double c) { return vfma(splat(a), b, splat(c)); }
inline vf vfma(const vf & a, double b, double c) { return vfma(a, splat(b), splat(c)); }
inline vf vfma(double a, double b, const vf & c) { return vfma(splat(a), splat(b), c); }
#  define W64_UN_F(name, expr)                 \
    inline vf name(const vf & a)               \
    {                                          \
      vf r;                                    \
      W64_LOOP r.v[l_] = expr(a.v[l_]);        \
      return r;                                \
    }
W64_UN_F(vsqrt, std::sqrt)
W64_UN_F(vabs, std::fabs)
W64_UN_F(vfloor, std::floor)
#  undef W64_UN_F
inline vf vmin(const vf & a, const vf & b)
{
  vf r;
  W64_LOOP r.v[l_] = std::fmin(a.v[l_], b.v[l_]);
  return r;
}
inline vf vmax(const vf & a, const vf & b)
{
  vf r;
  W64_LOOP r.v[l_] = std::fmax(a.v[l_], b.v[l_]);
  return r;
}
inline vf vmin(const vf & a, double b) { return vmin(a, splat(b)); }
inline vf vmax(const vf & a, double b) { return vmax(a, splat(b)); }
inline vi to_int(const vf & a)
{
  vi r;
  W64_LOOP r.v[l_] = static_cast<int>(a.v[l_]);
  return r;
}
inline unsigned long long ballot(const vb & m)
{
  unsigned long long r = 0;
  W64_LOOP if(m.v[l_]) r |= 1ull << l_;
  return r;
}
inline double read_lane(const vf & v, int k) { return v.v[k]; }
inline int read_lane_i(const vi & v, int k) { return v.v[k]; }
template<int K>
inline vf row_bcast(const vf & v)
{
  vf r;
  W64_LOOP r.v[l_] = v.v[(l_ & ~15) + K];
  return r;
}
template<int K>
inline vi row_bcast_i(const vi & v)
{
  vi r;
  W64_LOOP r.v[l_] = v.v[(l_ & ~15) + K];
  return r;
}
template<int K>
inline vf row_from_lower(const vf & v)
{
  vf r;
  W64_LOOP r.v[l_] = v.v[(l_ & ~15) + ((l_ - K) & 15)];
  return r;
}
inline vf x1(const vf & v)
{
  vf r;
  W64_LOOP r.v[l_] = v.v[l_ ^ 1];
  return r;
}
inline vf x2(const vf & v)
{
  vf r;
  W64_LOOP r.v[l_] = v.v[l_ ^ 2];
  return r;
}
inline vf hm(const vf & v)
{
  vf r;
  W64_LOOP r.v[l_] = v.v[(l_ & ~7) + (7 - (l_ & 7))];
  return r;
}
inline vf rm(const vf & v)
{
  vf r;
  W64_LOOP r.v[l_] = v.v[(l_ & ~15) + (15 - (l_ & 15))];
  return r;
}
inline void rows_pair(const vf & v, vf & a, vf & b)
{
  vf ra, rb;
  W64_LOOP
  {
    const int row = l_ >> 4, c = l_ & 15;
    ra.v[l_] = v.v[((row & ~1) << 4) + c];
    rb.v[l_] = v.v[((row | 1) << 4) + c];
  }
  a = ra;
  b = rb;
}
inline void halves_pair(const vf & v, vf & a, vf & b)
{
  vf ra, rb;
  W64_LOOP
  {
    ra.v[l_] = v.v[l_ & 31];
    rb.v[l_] = v.v[(l_ & 31) + 32];
  }
  a = ra;
  b = rb;
}
inline vf ld(const double * p, const vi & idx)
{
  vf r;
  W64_LOOP r.v[l_] = p[idx.v[l_]];
  return r;
}
inline vf ld_if(const double * p, const vi & idx, const vb & m, double other = 0.0)
{
  vf r;
  W64_LOOP r.v[l_] = m.v[l_] ? p[idx.v[l_]] : other;
  return r;
}
inline vf ld_if_branch(const double * p, const vi & idx, const vb & m, double other = 0.0) { return ld_if(p, idx, m, other); }
inline void st(double * p, const vi & idx, const vf & v, const vb & m)
{
  W64_LOOP if(m.v[l_]) p[idx.v[l_]] = v.v[l_];
}
inline vi ldi(const int * p, const vi & idx)
{
  vi r;
  W64_LOOP r.v[l_] = p[idx.v[l_]];
  return r;
}
inline vi ldb(const unsigned char * p, const vi & idx)
{
  vi r;
  W64_LOOP r.v[l_] = static_cast<int>(p[idx.v[l_]]);
  return r;
}
inline void sth(unsigned short * p, const vi & idx, const vi & v, const vb & m)
{
  W64_LOOP if(m.v[l_]) p[idx.v[l_]] = static_cast<unsigned short>(v.v[l_]);
}
inline int uniform_i(int v) { return v; }
inline vi opaque(const vi & v) { return v; }
inline vi tbl4(unsigned long long t, const vi & idx)
{
  vi r;
  W64_LOOP r.v[l_] = static_cast<int>((t >> (4 * idx.v[l_])) & 15ull);
  return r;
}
inline void wave_sync() {}
inline void mem_sync() {}
#endif

// =========================================================================================== common (both builds)
// Sum over the 16 lanes of each row, replicated in the row.  FIXED TREE (the oracle's tree16):
//   ((t0+t1)+(t2+t3)) + ((t4+t5)+(t6+t7))  +  ((t8+t9)+(t10+t11)) + ((t12+t13)+(t14+t15))
// every level adds a lane's value and its partner's: a + b == b + a bit for bit, so all lanes hold the same bits.
W64_FN vf sum16(vf v)
{
  v = v + x1(v);
  v = v + x2(v);
  v = v + hm(v);
  v = v + rm(v);
  return v;
}
W64_FN vf max16(vf v)
{
  v = vmax(v, x1(v));
  v = vmax(v, x2(v));
  v = vmax(v, hm(v));
  v = vmax(v, rm(v));
  return v;
}
// Sum over the four rows (lanes c, c+16, c+32, c+48), replicated: (row0 + row1) + (row2 + row3)
W64_FN vf sum_rows(vf v)
{
  vf a, b;
  rows_pair(v, a, b);
  v = a + b;
  halves_pair(v, a, b);
  return a + b;
}
} // namespace w64
