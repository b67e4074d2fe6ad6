// sym_tableau.h -- the symmetric sweep tableau of the range-QP kernels, packed in LDS.
//
// The tableau T of the dual active-set iteration (zmp.hip) is symmetric and stays symmetric under a pivot
// T'_ji = T_ji - (v_i / d) v_j.  The full-storage kernels keep all NR^2 entries in LDS and spend their time moving them
// (2 * 8 * NR^2 bytes through the 128 B/clk LDS port per pivot).  Here only the lower triangle is kept, cut in 4 x 4
// tiles (a, b), a >= b: NR/4 (NR/4 + 1) / 2 tiles of 128 bytes -- HALF the LDS traffic per pivot, HALF the footprint
// (128 rows: 68 KB, two workgroups per CU instead of one; 104 rows: 45 KB, three).
//
//   * thread t owns the tiles t and t + NT (row-major over a, then b <= a) and updates each in registers: 8 ds_read_b128,
//     16 FMAs, 8 ds_write_b128, plus the 4 + 4 pivot-column entries it needs (broadcast-friendly reads of the staging copy);
//   * layout: the sixteen doubles of a tile are eight 16-byte slots, slot s of tile t at double2 index s * NTP + t, so
//     that the b128 accesses of consecutive threads are contiguous (bank-conflict free);
//   * the diagonal tiles hold (and update) their upper half too; it is never read.
//
// Row k of T (= column k) is gathered entry by entry through entry(); that access is a few-way bank conflicted and
// happens twice per pivot.
#pragma once

#include <hip/hip_runtime.h>

namespace ccc_amd
{
template<int NR>
struct SymTab
{
  static_assert(NR % 4 == 0, "rows come in tiles of 4");
  static constexpr int NB = NR / 4;                // tile rows
  static constexpr int NTILE = NB * (NB + 1) / 2;  // tiles of the lower triangle
  static constexpr int TPT = NR > 112 ? 3 : 2;     // tiles per thread (thread t: tiles t, t + NT, ...)
  static constexpr int NT0 = ((NTILE + TPT - 1) / TPT + 63) / 64 * 64;
  static constexpr int NTR = (NR + 63) / 64 * 64;
  static constexpr int NT = NT0 > NTR ? NT0 : NTR; // threads per workgroup: at least one per row
  static constexpr int NTP = NTILE;                // tiles per slot plane
  static constexpr int kDoubles = 16 * NTP;        // LDS doubles

  // tile (a, b) of thread t: a (a + 1) / 2 + b = t
  static __device__ __forceinline__ void tile_of(int t, int & a, int & b)
  {
    int aa = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while(aa * (aa + 1) / 2 > t) --aa;
    while((aa + 1) * (aa + 2) / 2 <= t) ++aa;
    a = aa;
    b = t - aa * (aa + 1) / 2;
  }

  // index (in doubles) of entry (j, i), either order
  static __device__ __forceinline__ int entry(int j, int i)
  {
    const int hi = j > i ? j : i, lo = j > i ? i : j;
    const int a = hi >> 2, b = lo >> 2;
    const int tile = ((a * (a + 1)) >> 1) + b;
    const int e = ((hi & 3) << 2) | (lo & 3);
    return (((e >> 1) * NTP + tile) << 1) | (e & 1);
  }

  // tile t <- the 4 x 4 block (4a.., 4b..) of the row-major matrix G (row stride GS)
  static __device__ __forceinline__ void load_tile(double * S, const double * __restrict__ G, int GS, int t, int a, int b)
  {
    double2 * S2 = reinterpret_cast<double2 *>(S);
#pragma unroll
    for(int r = 0; r < 4; ++r)
    {
      const double * g = G + (size_t)(4 * a + r) * GS + 4 * b;
      S2[(2 * r) * NTP + t] = make_double2(g[0], g[1]);
      S2[(2 * r + 1) * NTP + t] = make_double2(g[2], g[3]);
    }
  }

  // rank-1 update of tile t = (a, b): T_rc -= (v[4b + c] rp) v[4a + r]   (v = staging copy of the pivot column)
  static __device__ __forceinline__ void update_tile(double * S, const double * v, double rp, int t, int a, int b)
  {
    double2 * S2 = reinterpret_cast<double2 *>(S);
    const double2 * v2 = reinterpret_cast<const double2 *>(v);
    double2 x[8];
#pragma unroll
    for(int s = 0; s < 8; ++s) x[s] = S2[s * NTP + t];
    const double2 r01 = v2[2 * a], r23 = v2[2 * a + 1], c01 = v2[2 * b], c23 = v2[2 * b + 1];
    const double vr[4] = {r01.x, r01.y, r23.x, r23.y};
    const double g[4] = {-c01.x * rp, -c01.y * rp, -c23.x * rp, -c23.y * rp};
#pragma unroll
    for(int r = 0; r < 4; ++r)
    {
      x[2 * r].x = fma(g[0], vr[r], x[2 * r].x);
      x[2 * r].y = fma(g[1], vr[r], x[2 * r].y);
      x[2 * r + 1].x = fma(g[2], vr[r], x[2 * r + 1].x);
      x[2 * r + 1].y = fma(g[3], vr[r], x[2 * r + 1].y);
    }
#pragma unroll
    for(int s = 0; s < 8; ++s) S2[s * NTP + t] = x[s];
  }

  // (T w)_i for row i: walks the tile row of i (entries (i, 4b + c)) and then the tile column of i (entries (4a + r, i))
  static __device__ __forceinline__ double matvec_row(const double * S, const double * w, int i)
  {
    const int ti = i >> 2, ri = i & 3;
    double acc = 0.0;
    const int rowbase = (ti * (ti + 1)) >> 1;
    for(int b = 0; b < ti; ++b) // tiles (ti, b), row ri
    {
      const int tile = rowbase + b;
#pragma unroll
      for(int c = 0; c < 4; ++c)
      {
        const int e = (ri << 2) | c;
        acc = fma(S[(((e >> 1) * NTP + tile) << 1) | (e & 1)], w[4 * b + c], acc);
      }
    }
    for(int a = ti; a < NB; ++a) // tiles (a, ti), column ri (the diagonal tile: lower half rows r >= ri, then its row)
    {
      const int tile = ((a * (a + 1)) >> 1) + ti;
#pragma unroll
      for(int r = 0; r < 4; ++r)
      {
        const bool lower = (a > ti) || (r >= ri);
        const int e = lower ? ((r << 2) | ri) : ((ri << 2) | r);
        acc = fma(S[(((e >> 1) * NTP + tile) << 1) | (e & 1)], w[4 * a + r], acc);
      }
    }
    return acc;
  }
};
} // namespace ccc_amd
