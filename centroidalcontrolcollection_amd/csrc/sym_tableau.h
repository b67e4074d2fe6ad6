// sym_tableau.h -- the symmetric sweep tableau of the range-QP kernels, packed in LDS.
//
// The tableau T of the dual active-set iteration (zmp.hip) is symmetric and stays symmetric under a pivot
// T'_ji = T_ji - (v_i / d) v_j.  The full-storage kernels keep all NR^2 entries in LDS and spend their time moving them
// (2 * 8 * NR^2 bytes through the 128 B/clk LDS port per pivot).  Here only the lower triangle is kept, cut in 4 x 4
// (or 2 x 2) tiles (a, b), a >= b: NR/4 (NR/4 + 1) / 2 tiles of 128 bytes -- HALF the LDS traffic per pivot, HALF the footprint
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
template<int NR, int TS = 4, int TPT_ = 2>
struct SymTab
{
  static_assert(TS == 2 || TS == 4, "tiles are 2 x 2 or 4 x 4");
  static_assert(NR % TS == 0, "rows come in whole tiles");
  static constexpr int TS_ = TS;
  static constexpr int NB = NR / TS;               // tile rows
  static constexpr int NTILE = NB * (NB + 1) / 2;  // tiles of the lower triangle
  static constexpr int SL = TS * TS / 2;           // 16-byte slots per tile
  static constexpr int TPT = TPT_;                 // tiles per thread (thread t: tiles t, t + NT, ...)
  static constexpr int NT0 = ((NTILE + TPT - 1) / TPT + 63) / 64 * 64;
  static constexpr int NTR = (NR + 63) / 64 * 64;
  static constexpr int NT = NT0 > NTR ? NT0 : NTR; // threads per workgroup: at least one per row
  static constexpr int NTP = NTILE;                // tiles per slot plane
  static constexpr int kDoubles = 2 * SL * NTP;    // LDS doubles
  // waves per SIMD the kernels ask the compiler for: two or three workgroups share a CU up to 320 threads
  // (40 rows: 129 VGPRs left alone, one over what four wavefronts allow; asked for, 127 without a spill and sixteen
  //  7 KB workgroups per CU instead of twelve: 29.7 -> 33.7 M solves/s at N = 40.  48 / 56 rows would spill: measured slower)
  static constexpr int kMinWaves = (NT <= 64 && NR <= 40) ? 4 : (NT <= 256 ? 3 : (NT <= 320 ? 4 : (NT <= 512 ? 2 : 4)));

  // tile (a, b) of thread t: a (a + 1) / 2 + b = t
  static __device__ __forceinline__ void tile_of(int t, int & a, int & b)
  {
    int aa = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while(aa * (aa + 1) / 2 > t) --aa;
    while((aa + 1) * (aa + 2) / 2 <= t) ++aa;
    a = aa;
    b = t - aa * (aa + 1) / 2;
  }

  // index (in doubles) of element e = r * TS + c of tile `tile`
  static __device__ __forceinline__ int slot_index(int tile, int e) { return (((e >> 1) * NTP + tile) << 1) | (e & 1); }

  // index (in doubles) of entry (j, i), either order
  static __device__ __forceinline__ int entry(int j, int i)
  {
    const int hi = j > i ? j : i, lo = j > i ? i : j;
    const int a = hi / TS, b = lo / TS;
    return slot_index(((a * (a + 1)) >> 1) + b, (hi % TS) * TS + (lo % TS));
  }

  // tile t <- the TS x TS block (TS a.., TS b..) of the row-major matrix G (row stride GS)
  static __device__ __forceinline__ void load_tile(double * S, const double * __restrict__ G, int GS, int t, int a, int b)
  {
    double2 * S2 = reinterpret_cast<double2 *>(S);
#pragma unroll
    for(int r = 0; r < TS; ++r)
    {
      const double * g = G + (size_t)(TS * a + r) * GS + TS * b;
#pragma unroll
      for(int h = 0; h < TS / 2; ++h) S2[(r * (TS / 2) + h) * NTP + t] = make_double2(g[2 * h], g[2 * h + 1]);
    }
  }

  // rank-1 update of tile t = (a, b): T_rc -= (v[TS b + c] rp) v[TS a + r]   (v = staging copy of the pivot column)
  static __device__ __forceinline__ void update_tile(double * S, const double * v, double rp, int t, int a, int b)
  {
    double2 * S2 = reinterpret_cast<double2 *>(S);
    const double2 * v2 = reinterpret_cast<const double2 *>(v);
    double2 x[SL];
#pragma unroll
    for(int s = 0; s < SL; ++s) x[s] = S2[s * NTP + t];
    double vr[TS], g[TS];
#pragma unroll
    for(int h = 0; h < TS / 2; ++h)
    {
      const double2 r2 = v2[(TS / 2) * a + h], c2 = v2[(TS / 2) * b + h];
      vr[2 * h] = r2.x;
      vr[2 * h + 1] = r2.y;
      g[2 * h] = -c2.x * rp;
      g[2 * h + 1] = -c2.y * rp;
    }
#pragma unroll
    for(int r = 0; r < TS; ++r)
#pragma unroll
      for(int h = 0; h < TS / 2; ++h)
      {
        x[r * (TS / 2) + h].x = fma(g[2 * h], vr[r], x[r * (TS / 2) + h].x);
        x[r * (TS / 2) + h].y = fma(g[2 * h + 1], vr[r], x[r * (TS / 2) + h].y);
      }
#pragma unroll
    for(int s = 0; s < SL; ++s) S2[s * NTP + t] = x[s];
  }

  // (T w)_i for row i: walks the tile row of i (entries (i, TS b + c)) and then the tile column of i (entries (TS a + r, i))
  static __device__ __forceinline__ double matvec_row(const double * S, const double * w, int i)
  {
    const int ti = i / TS, ri = i % TS;
    double acc = 0.0;
    const int rowbase = (ti * (ti + 1)) >> 1;
    for(int b = 0; b < ti; ++b) // tiles (ti, b), row ri
#pragma unroll
      for(int c = 0; c < TS; ++c) acc = fma(S[slot_index(rowbase + b, ri * TS + c)], w[TS * b + c], acc);
    for(int a = ti; a < NB; ++a) // tiles (a, ti), column ri (the diagonal tile: lower half rows r >= ri, then its row)
    {
      const int tile = ((a * (a + 1)) >> 1) + ti;
#pragma unroll
      for(int r = 0; r < TS; ++r)
      {
        const bool lower = (a > ti) || (r >= ri);
        acc = fma(S[slot_index(tile, lower ? r * TS + ri : ri * TS + r)], w[TS * a + r], acc);
      }
    }
    return acc;
  }
};
} // namespace ccc_amd
