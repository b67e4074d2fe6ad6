// zmp.hip -- batched CCC::LinearMpcZmp::planOnce() on MI355X (gfx950): kernels + C-ABI.
//
// Path replaced (reference file:line under /root/reference):
//   src/CommonModels.cpp:8-17, include/CCC/StateSpaceModel.h:164-216,
//   include/CCC/InvariantSequentialExtension.h:103-181        -> ZmpModel (host, closed form, once per handle)
//   src/LinearMpcZmp.cpp:46-81 (procOnce) incl. the external QP solve at :69
//   src/LinearMpcZmp.cpp:83-112 (planOnce: x axis then y axis)  -> zmp_plan_kernel (device, per instance)
//
// The QP of src/LinearMpcZmp.cpp:21-27,54-69 is   min 1/2 |u|^2   s.t.  lo <= B u <= hi,
//   B = B_seq (lower-triangular Toeplitz, invertible),  lo = zmin - A_seq x0,  hi = zmax - A_seq x0.
// Its KKT system in the multipliers mu (u = B' mu, G = B B' batch-constant SPD) reads
//   (G mu)_i = lo_i if mu_i > 0,  = hi_i if mu_i < 0,  in [lo_i, hi_i] if mu_i = 0.
// The kernel runs the Goldfarb-Idnani dual active-set iteration directly on that system: the working
// set W grows by the most violated row and shrinks by the dual ratio test.  Instead of a QR/Cholesky
// factorisation it keeps the symmetric sweep tableau T of G swept on W (T_WW = -G_WW^-1,
// T_iW = G_iW G_WW^-1, T_ij = Schur complement otherwise): ONE LANE PER ROW, the row in that lane's
// VGPRs.  Column p of T (search direction, Schur complement, ratio test) is row p by symmetry, so the
// lane that owns row p publishes it through a 300-byte LDS scratch and every lane of the group reads
// its own element plus the broadcast of the whole row; adding / dropping a row is then one rank-1
// update T -= g v' of register-resident data (N FMAs per lane, no data-dependent loop bounds).
//
// Mapping: N <= 32 (K1, zmp_plan_kernel): the two axes of one instance are the two 32-lane halves of ONE wavefront
// (one planOnce() per wavefront), the tableau in registers.  32 < N <= 200 (K2, zmp_plan_sym_kernel): one QP per
// workgroup, the tableau packed (lower triangle, sym_tableau.h) in LDS.  200 < N <= 256 (K3, zmp_plan_block_kernel):
// the full tableau in an HBM workspace; 256 < N <= 512: the same kernel at 512 rows (1024 threads per workgroup).
// Round 5: K1w (zmp_plan_kernel_w, 32 < N <= 64, one QP per wavefront) and K2r (zmp_k2r.inc, the packed tableau in register
// tiles, to 128 rows).  Round 6: KS (zmp_stage.inc), the same QP in its state-space form -- O(N) per iteration, one QP per
// lane -- on large batches of N >= 40, with the kernels above as the exact solver of what it does not certify.
#include "common.h"
#include "sym_tableau.h"
#include "wave_group.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ccc_amd
{
struct ZmpDev
{
  int N;            // horizon steps
  const double * G; // [NP][NP] B_seq B_seq', zero padded
  const double * A; // [NP][3]  A_seq rows, zero padded
  const double * b; // [NP]     first column of B_seq (B_seq[i][j] = b[i-j])
  double c2;        // C(0,2) = -com_height / g
  double dt;        // horizon_dt
  // K1 only (round 5): the pivot trips of every QP of this call -> hist (when given); order (when given) = the QPs sorted by
  // the trips the PREVIOUS call of the same size spent on them, longest first: slot s of the schedule solves QP order[s]
  const int * order;
  int * hist;
  int * diff; // (when given: |trips now - trips of the last call| per QP -> order_by_count's verdict on the history)
  // KS (zmp_stage.inc) and the exact kernels behind it: the QPs KS does not certify.  KS appends to fb_list / fb_count; a
  // block / sym / reg kernel launched with them set solves fb_list[0 .. *fb_count) instead of [0, nqp)
  int * fb_list;
  int * fb_count;
};

constexpr double kInf = __builtin_huge_val();

// scratch doubles per group: the broadcast pivot row [0, NP), then the reciprocal pivot
template<int NP>
struct ZmpScratch
{
  static constexpr int kRp = NP;
  static constexpr int kT = NP + 1;    // the full step of the entering row (K1_PIVOT_FAST)
  static constexpr int kSize = NP + 8; // keeps every group's base 16-byte aligned
};

typedef double v16d __attribute__((ext_vector_type(16)));

// Row li of the tableau in this lane's VGPRs, as 16-wide register tuples so that a wave-uniform column index
// is one s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off (no LDS, no scratch memory).
template<int NP>
struct RowRegs
{
  v16d t[NP / 16];

  // element `u` of the row, u uniform across the wavefront (held in an SGPR)
  __device__ __forceinline__ double at_uniform(int u) const
  {
    // (written out, not a loop over the tuples: a loop that is still rolled when the optimiser first looks keeps the whole
    //  struct in memory -- measured at three and four tuples: the row in scratch)
    double r = t[0][u & 15];
    if constexpr(NP / 16 > 1)
    {
      const double e = t[1][u & 15];
      r = ((u >> 4) == 1) ? e : r;
    }
    if constexpr(NP / 16 > 2)
    {
      const double e = t[2][u & 15];
      r = ((u >> 4) == 2) ? e : r;
    }
    if constexpr(NP / 16 > 3)
    {
      const double e = t[3][u & 15];
      r = ((u >> 4) == 3) ? e : r;
    }
    return r;
  }
};

// element idx of the row where idx is uniform inside each LG-lane group (lane i of a group gets T[i][idx])
template<int LG, int NP>
__device__ __forceinline__ double row_at_group_uniform(const RowRegs<NP> & T, int idx)
{
  const int u0 = __builtin_amdgcn_readlane(idx, 0);
  const double c0 = T.at_uniform(u0);
  if(LG == 64) return c0;
  const int u1 = __builtin_amdgcn_readlane(idx, 32);
  const double c1 = T.at_uniform(u1);
  return (threadIdx.x & 32) ? c1 : c0;
}

// Block layout of the row (round 6, csrc/zmp_k1.inc K1D_*): TS[k] = T[li][16 rblk + k], TO[k] = T[li][16 (1 - rblk) + k] with rblk
// the 16-lane DPP row of the lane inside its 32-lane group.  Element idx of the row, idx uniform inside each group:
__device__ __forceinline__ double row_at_group_uniform_blocks(const v16d & TS, const v16d & TO, int idx, int rblk)
{
  const int u0 = __builtin_amdgcn_readlane(idx, 0), u1 = __builtin_amdgcn_readlane(idx, 32);
  const double a0 = TS[u0 & 15], b0 = TO[u0 & 15];
  const double a1 = TS[u1 & 15], b1 = TO[u1 & 15];
  const double c0 = ((u0 >> 4) == rblk) ? a0 : b0;
  const double c1 = ((u1 >> 4) == rblk) ? a1 : b1;
  return (threadIdx.x & 32) ? c1 : c0;
}
#define K1D_TCOL(u) row_at_group_uniform_blocks(TS, TO, (u), rblk)

// the row accessors of csrc/zmp_k1.inc for the kernels that keep the row in a RowRegs struct `T`
#define K1_TGET(j) T.t[(j) / 16][(j) % 16]
#define K1_TSET(j, v) T.t[(j) / 16][(j) % 16] = (v)
#define K1_TCOL(u) row_at_group_uniform<LG, NP>(T, (u))

// K1, static pairing: QP (instance, axis) = (qp / 2, qp % 2), the two axes of an instance in the two halves of a
// wavefront.  The steps of the iteration are the sections of csrc/zmp_k1.inc, shared with zmp_plan_kernel_dyn.
// (the occupancy is asked for explicitly.  Rounds 2-5: three wavefronts per SIMD -- left alone hipcc allocates 169 VGPRs, one
//  more than three allow, and the kernel ran a quarter slower.  Round 6: FOUR, 128 VGPRs with 20 spilled dwords per lane:
//  the trip is bound by vector issue at 0.85, and a fourth wavefront fills more of the rest than the spills cost --
//  130.3 -> 136.0 M solves/s on rotating batches, 164.2 -> 170.5 M on a repeated one, two runs each)
#ifndef CCC_ZMP_K1_OCC
#  define CCC_ZMP_K1_OCC 4
#endif
template<int LG, int WAVES>
__global__ __launch_bounds__(WAVES * 64, CCC_ZMP_K1_OCC) void zmp_plan_kernel(ZmpDev P, long nqp, const double * __restrict__ x0,
                                                             const double * __restrict__ zlim, double control_dt,
                                                             double * __restrict__ zmp, double * __restrict__ jerk,
                                                             int * __restrict__ status)
{
  using Grp = WaveGroup<LG>;
  using Scr = ZmpScratch<LG>;
  constexpr int NP = LG;
  static_assert(LG == 32, "block layout: two 16-lane DPP rows per group");
  constexpr int QPW = 64 / LG; // QPs per wavefront
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double * Gs = smem;           // [NP][NP]
  double * bs = smem + NP * NP; // [NP]
  double * As = bs + NP;        // [NP][3]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & (LG - 1), grp = lane / LG, lic = li;
  const int rblk = (li >> 4) & 1; // the 16-lane DPP row of this lane inside its group
  double * scr = As + 3 * NP + (wave * QPW + grp) * Scr::kSize; // this group's scratch
  const double2 * scr2 = reinterpret_cast<const double2 *>(scr);

  for(int k = tid; k < NP * NP; k += WAVES * 64) Gs[k] = P.G[k];
  for(int k = tid; k < NP; k += WAVES * 64) bs[k] = P.b[k];
  for(int k = tid; k < 3 * NP; k += WAVES * 64) As[k] = P.A[k];
  __syncthreads();

  const int N = P.N;
  const int maxpass = 20 * N + 100;

  const long ntask = (nqp + QPW - 1) / QPW;
  for(long task = (long)blockIdx.x * WAVES + wave; task < ntask; task += (long)gridDim.x * WAVES)
  {
    // (instance, axis) = (qp / 2, qp % 2); without a history the two axes of an instance share a wavefront, with one the
    // two QPs of a wavefront are neighbours in the order of their last pivot counts -- the lock-step pair wastes little
    const long slot = task * QPW + grp;
    const bool valid = slot < nqp;
    const long qp = valid ? (P.order ? (long)P.order[slot] : slot) : 0;
    const bool row = valid && li < N;
    double lo, hi;
    // tableau T = G (W empty): lane li holds row li = column li of the symmetric G.  The diagonal lives in dg: the
    // in-row copy T[li][li] is never read by its owner (dgm mirrors its bits for the closing refinement).
    v16d TS, TO; // the row in block layout (K1D_LOAD)
    double dg, dgm;
    double z, mu;      // z = (G mu)_li, mu = multiplier of row li
    bool inW, side;    // while in W: side = sits on lo
    int p;             // entering row   (group uniform)
    double sig;        // its side +1/-1 (group uniform)
    int passes;
#define K1D_LOAD
#include "zmp_k1.inc"
#undef K1D_LOAD
    int st = CCC_STATUS_SOLVED;
    if(Grp::any(row && lo > hi)) st = CCC_STATUS_INFEASIBLE;
    bool done = !valid || st != CCC_STATUS_SOLVED;
    bool need_select = true;
#define K1_SELECT
#include "zmp_k1.inc"
#undef K1_SELECT

    for(int round = 0; round < 3; ++round)
    {
      select_entering();
      // one trip = one pivot (a row enters W or leaves it).  Written as an explicit guarded do-while: with a
      // top-tested loop hipcc keeps a second copy of the register-resident row and moves it back every trip
      if(__ballot(!done) != 0ull) do
        {
#define K1D_PIVOT
#include "zmp_k1.inc"
#undef K1D_PIVOT
        } while(__ballot(!done) != 0ull);

      const bool fin = true;
#define K1D_REFINE
#include "zmp_k1.inc"
#undef K1D_REFINE
      done = !reopen;
      need_select = true;
      __builtin_amdgcn_wave_barrier();
      if(__ballot(reopen) == 0ull) break;
    }
    const bool emit = true;
#define K1_WRITE
#include "zmp_k1.inc"
#undef K1_WRITE
  }
}

#undef K1_TGET
#undef K1_TSET
#undef K1_TCOL
// ... and for K1w below, which keeps it in SEPARATE sixteen-double tuples Tq0 .. Tq3: as members of one struct three or four
// tuples stay in memory whatever the code looks like (measured on a twenty-line kernel: 512 / 640 B of scratch; separate
// variables: none at three tuples)
#define K1_TGET(j) (((j) / 16 == 0) ? Tq0[(j) % 16] : (((j) / 16 == 1) ? Tq1[(j) % 16] : (((j) / 16 == 2) ? Tq2[(j) % 16] : Tq3[(j) % 16])))
#define K1_TSET(j, v)                                  \
  do                                                   \
  {                                                    \
    if((j) / 16 == 0) Tq0[(j) % 16] = (v);             \
    else if((j) / 16 == 1) Tq1[(j) % 16] = (v);        \
    else if((j) / 16 == 2) Tq2[(j) % 16] = (v);        \
    else Tq3[(j) % 16] = (v);                          \
  } while(0)
#define K1_TCOL(u) tq_col(Tq0, Tq1, Tq2, Tq3, __builtin_amdgcn_readfirstlane(u), NP)
__device__ __forceinline__ double tq_col(const v16d & a, const v16d & b, const v16d & c, const v16d & d, int u, int np)
{
  double r = a[u & 15];
  if(np > 16)
  {
    const double e = b[u & 15];
    r = ((u >> 4) == 1) ? e : r;
  }
  if(np > 32)
  {
    const double e = c[u & 15];
    r = ((u >> 4) == 2) ? e : r;
  }
  if(np > 48)
  {
    const double e = d[u & 15];
    r = ((u >> 4) == 3) ? e : r;
  }
  return r;
}

// K1w.  32 < N <= NPC <= 64: ONE QP per wavefront, lane li < NPC holds row li of an NPC-column tableau in NPC / 16
// indexable register tuples (the lanes beyond NPC idle along).  Round 5: the step from K1 to the
// LDS tableau cost a factor of three at N = 33 (105 M -> 38 M solves/s); K1's trip is per wavefront, so a wavefront that
// carries one 48-row QP instead of two 32-row ones should lose a factor of two and a bit, not three.  Same sections of
// csrc/zmp_k1.inc, same arithmetic; a QP per wavefront needs no pairing, so the static schedule is all there is.
// (Round 3 tried the 64-lane, 64-column instantiation under K1's three-wavefronts-per-SIMD register cap: the row went to
//  scratch, 6.7 M solves/s.  Here the cap follows the row: MINW = 3 at 48 columns, 2 at 64.)
template<int NPC, int WAVES, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void zmp_plan_kernel_w(ZmpDev P, int GS, long nqp, const double * __restrict__ x0,
                                                                     const double * __restrict__ zlim, double control_dt,
                                                                     double * __restrict__ zmp, double * __restrict__ jerk,
                                                                     int * __restrict__ status)
{
  constexpr int LG = 64, NP = NPC;
  static_assert(NP % 16 == 0 && NP <= 64, "the row lives in 16-wide register tuples");
  using Grp = WaveGroup<LG>;
  using Scr = ZmpScratch<LG>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double * Gs = smem;           // [NP][NP]
  double * bs = smem + NP * NP; // [NP]
  double * As = bs + NP;        // [NP][3]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane, lic = li < NP ? li : NP - 1;
  double * scr = As + 3 * NP + wave * Scr::kSize; // this wavefront's scratch
  const double2 * scr2 = reinterpret_cast<const double2 *>(scr);

  for(int k = tid; k < NP * NP; k += WAVES * 64) Gs[k] = P.G[(size_t)(k / NP) * GS + (k % NP)];
  for(int k = tid; k < NP; k += WAVES * 64) bs[k] = P.b[k];
  for(int k = tid; k < 3 * NP; k += WAVES * 64) As[k] = P.A[k];
  __syncthreads();

  const int N = P.N;
  const int maxpass = 20 * N + 100;
  for(long qp = (long)blockIdx.x * WAVES + wave; qp < nqp; qp += (long)gridDim.x * WAVES)
  {
    const bool valid = true;
    const bool row = li < N;
    double lo, hi;
    v16d Tq0, Tq1, Tq2, Tq3;
    Tq0 = Tq1 = Tq2 = Tq3 = 0.0;
    double dg, dgm;
    double z, mu;
    bool inW, side;
    int p;
    double sig;
    int passes;
#define K1_LOAD
#include "zmp_k1.inc"
#undef K1_LOAD
    if(li >= NP) dg = dgm = 1.0; // (idle lanes: a harmless curvature for the selection key)
    int st = CCC_STATUS_SOLVED;
    if(Grp::any(row && lo > hi)) st = CCC_STATUS_INFEASIBLE;
    bool done = st != CCC_STATUS_SOLVED;
    bool need_select = true;
#define K1_SELECT
#include "zmp_k1.inc"
#undef K1_SELECT
    for(int round = 0; round < 3; ++round)
    {
      select_entering();
      if(__ballot(!done) != 0ull) do
        {
#define K1_PIVOT_FAST
#include "zmp_k1.inc"
#undef K1_PIVOT_FAST
        } while(__ballot(!done) != 0ull);
      const bool fin = true;
#define K1_REFINE
#include "zmp_k1.inc"
#undef K1_REFINE
      done = !reopen;
      need_select = true;
      __builtin_amdgcn_wave_barrier();
      if(__ballot(reopen) == 0ull) break;
    }
    const bool emit = true;
#define K1_WRITE
#include "zmp_k1.inc"
#undef K1_WRITE
  }
}

#undef K1_TGET
#undef K1_TSET
#undef K1_TCOL
#define K1_TGET(j) T.t[(j) / 16][(j) % 16]
#define K1_TSET(j, v) T.t[(j) / 16][(j) % 16] = (v)
#define K1_TCOL(u) row_at_group_uniform<LG, NP>(T, (u))

// K1 with a work queue per 32-lane group ("dyn").  In zmp_plan_kernel the two axes of an instance run in lock-step: a
// wavefront spends max(pivots_x, pivots_y) trips on a pair (measured: 24.7 against a mean of 17.9 per QP).  Here every group
// takes its QPs from a global queue on its own: when one group finishes (refinement, outputs) it fetches and sets up the
// next QP while the other keeps pivoting -- the set-up / refinement code runs under divergence, once per QP, the pivot trip
// stays the same straight-line code.  The arithmetic of a QP is unchanged: the same sections of csrc/zmp_k1.inc.
constexpr int kQueues = 64;      // ticket counters of zmp_plan_kernel_dyn
constexpr int kQueueStride = 16; // in counters: one 128-byte line each

template<int LG, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void zmp_plan_kernel_dyn(ZmpDev P, long nqp, const double * __restrict__ x0,
                                                                 const double * __restrict__ zlim, double control_dt,
                                                                 double * __restrict__ zmp, double * __restrict__ jerk,
                                                                 int * __restrict__ status,
                                                                 unsigned long long * __restrict__ queue)
{
  using Grp = WaveGroup<LG>;
  using Scr = ZmpScratch<LG>;
  constexpr int NP = LG;
  static_assert(LG == 32, "block layout: two 16-lane DPP rows per group");
  constexpr int QPW = 64 / LG; // QPs per wavefront
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double * Gs = smem;           // [NP][NP]
  double * bs = smem + NP * NP; // [NP]
  double * As = bs + NP;        // [NP][3]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & (LG - 1), grp = lane / LG, lic = li;
  const int rblk = (li >> 4) & 1; // the 16-lane DPP row of this lane inside its group
  double * scr = As + 3 * NP + (wave * QPW + grp) * Scr::kSize; // this group's scratch
  const double2 * scr2 = reinterpret_cast<const double2 *>(scr);

  for(int k = tid; k < NP * NP; k += WAVES * 64) Gs[k] = P.G[k];
  for(int k = tid; k < NP; k += WAVES * 64) bs[k] = P.b[k];
  for(int k = tid; k < 3 * NP; k += WAVES * 64) As[k] = P.A[k];
  __syncthreads();

  const int N = P.N;
  const int maxpass = 20 * N + 100;

  // group state (uniform inside a group)
  enum { kNeed = 0, kActive = 1, kIdle = 2 };
  int phase = kNeed, left = 0;
  long qp = 0, next = 0;
  int myq = (int)((blockIdx.x * (WAVES * QPW) + wave * QPW + grp) % kQueues);
  bool valid = false, row = false;
  double lo = -kInf, hi = kInf;
  int st = CCC_STATUS_SOLVED;
  v16d TS, TO; // the row in block layout (K1D_LOAD)
#pragma unroll
  for(int j = 0; j < 16; ++j)
  {
    TS[j] = 0.0;
    TO[j] = 0.0;
  }
  double dg = 1.0, dgm = 1.0;
  double z = 0.0, mu = 0.0;
  bool inW = false, side = false;
  int p = 0;
  double sig = 0.0;
  bool done = true, need_select = false;
  int passes = 0, round = 0;
#define K1_SELECT
#include "zmp_k1.inc"
#undef K1_SELECT

  for(;;)
  {
    // ---- groups without a QP: take the next one from the queue and set it up
    if(__ballot(phase == kNeed) != 0ull)
    {
      // kQueues ticket counters (one atomic per QP on a SINGLE address serialises in L2: 131072 of them take as long as
      // the whole kernel), each serving a contiguous share of the QPs; a group starts at its own queue and moves on to
      // the next ones when that is exhausted
      if(__ballot(phase == kNeed && left == 0) != 0ull)
      {
        // (ticket t of queue k = slot t kQueues + k of the schedule: every queue holds the same mix of the slots, so that
        //  with an order by the last call's pivot counts -- P.order -- the long QPs are taken first by everybody)
        for(int tries = 0; tries < kQueues; ++tries) // (wave-uniform trip count; groups that found work idle along)
        {
          const bool want = phase == kNeed && left == 0;
          if(__ballot(want) == 0ull) break;
          long q = -1;
          if(want && li == 0)
          {
            const long t = (long)atomicAdd(queue + (size_t)myq * kQueueStride, 1ull);
            q = (t * kQueues + myq < nqp) ? t * kQueues + myq : -1;
          }
          q = __shfl(q, lane & ~(LG - 1)); // the group leader's ticket
          if(want)
          {
            if(q >= 0)
            {
              next = q;
              left = 1;
            }
            else
              myq = (myq + 1) % kQueues;
          }
        }
      }
      bool fresh = false;
      if(phase == kNeed)
      {
        valid = left > 0;
        qp = valid ? (P.order ? (long)P.order[next] : next) : 0;
        if(valid) --left;
        if(!valid)
        {
          phase = kIdle;
          row = false;
          done = true;
          need_select = false;
        }
        else
        {
          fresh = true;
          row = li < N;
#define K1D_LOAD
#include "zmp_k1.inc"
#undef K1D_LOAD
          round = 0;
          need_select = true;
          phase = kActive;
        }
      }
      // (group collectives outside the divergent region)
      const bool infeasible = Grp::any(fresh && row && lo > hi);
      if(fresh)
      {
        st = infeasible ? CCC_STATUS_INFEASIBLE : CCC_STATUS_SOLVED;
        done = infeasible;
      }
      select_entering();
    }
    if(__ballot(phase != kIdle) == 0ull) break;

    // ---- one pivot for every group that is iterating
    if(__ballot(phase == kActive && !done) != 0ull)
    {
#define K1D_PIVOT
#include "zmp_k1.inc"
#undef K1D_PIVOT
    }

    // ---- groups whose iteration stopped: closing refinement, then either re-open or emit and ask for the next QP
    if(__ballot(phase == kActive && done) != 0ull)
    {
      const bool fin = phase == kActive && done;
#define K1D_REFINE
#include "zmp_k1.inc"
#undef K1D_REFINE
      const bool again = fin && reopen && round + 1 < 3;
      if(fin)
      {
        ++round;
        done = !again;
        need_select = again;
      }
      __builtin_amdgcn_wave_barrier();
      if(__ballot(again) != 0ull) select_entering();
      const bool emit = fin && !again;
      if(__ballot(emit) != 0ull)
      {
#define K1_WRITE
#include "zmp_k1.inc"
#undef K1_WRITE
        if(emit) phase = kNeed;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K3.  Horizons beyond 200 steps do not fit the 160 KB of LDS even packed: one QP per 512-thread workgroup, thread
// (part, i) updates its share of column i of the FULL tableau, which lives in an HBM workspace ([j][i], i fastest:
// coalesced).  Slow (every pivot streams 2 x 512 KB through L2) -- there for completeness of the drop-in surface.
// Same dual active-set iteration and closing refinement as zmp_plan_kernel.
// ---------------------------------------------------------------------------------------------
constexpr int kBigNP = 256;  // 200 < N <= 256
constexpr int kHugeNP = 512; // 256 < N <= 512 (round 4: BASELINE config 1 as worded at a 5 ms step is 400 steps)

// -DCCC_ZMP_PROF (development builds only, scripts/zprof.py): the LDS-tableau kernels time the phases of a pivot with
// s_memtime (ZPROF(k) closes phase k: 0 selection, 1 ratio test, 2 pivot-column staging, 3 tile update, 4 row rewrite +
// bookkeeping, 5 refinement) and OVERWRITE the first entries of each QP's jerk output with the totals (K2 adds the
// wall-clock span and start time of the QP: that is how the idle tail of grid-stride scheduling was found).
struct BlockRed
{
  double val[8];
  int idx[8];
  double num;  // K2r: the full step's numerator of the entering row (zmp_k2r.inc)
  int flag[2]; // K2r: "a multiplier would change sign on this pivot", double-buffered over the pivots
};

struct SelRed
{
  double val[8], sig[8];
  int idx[8];
};

__device__ __forceinline__ double wave_lane_value(double v, int k) // k uniform: two v_readlane_b32, no LDS round trip
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}

// (min value over the NP-thread block, lowest thread index attaining it; index NP if every candidate is NaN)
template<int NP, bool GUARD = true>
__device__ __forceinline__ void block_argmin(double v, BlockRed * red, double & vmin, int & imin)
{
  const int tid = threadIdx.x, w = tid >> 6;
  const double wm = WaveGroup<64>::min(v);
  const int wi = WaveGroup<64>::first(v == wm);
  if(GUARD) __syncthreads(); // red may still be read by the previous reduction
  if((tid & 63) == 0 && w < (NP + 63) / 64) // rows live in the first wavefronts (part 0)
  {
    red->val[w] = wm;
    red->idx[w] = wi < 64 ? wi + 64 * w : NP;
  }
  __syncthreads();
  double best = red->val[0];
  int bi = red->idx[0];
#pragma unroll
  for(int k = 1; k < (NP + 63) / 64; ++k)
  {
    const double a = red->val[k];
    const int ia = red->idx[k];
    const bool take = (ia < NP) && (bi >= NP || a < best);
    best = take ? a : best;
    bi = take ? ia : bi;
  }
  vmin = best;
  imin = bi;
}

// PARTS threads per row: thread (part, i) updates its share of column i in the rank-1 update (more wavefronts in flight
// hide the LDS latency); thread (0, i) owns row i (bounds, multiplier, flags).
template<int NP, bool HBM, int PARTS, bool LIST>
__device__ __forceinline__ void zmp_plan_block_body(ZmpDev P, long nqp, const double * __restrict__ x0,
                                                    const double * __restrict__ zlim, double control_dt,
                                                    double * __restrict__ zmp, double * __restrict__ jerk,
                                                    int * __restrict__ status, double * __restrict__ ws)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int TS = HBM ? NP : NP + 1; // row stride of T; odd in LDS, so that column writes are bank-conflict free
  double * T = HBM ? ws + (size_t)blockIdx.x * NP * NP : smem; // [NP][TS]
  double * cb = HBM ? smem : smem + NP * TS;                   // [NP] staging of the pivot row / of mu / of rho
  BlockRed * red = reinterpret_cast<BlockRed *>(cb + NP);
  SelRed * sel = reinterpret_cast<SelRed *>(red + 1);
  const int i = threadIdx.x % NP, part = threadIdx.x / NP;
  const bool lead = part == 0;
  constexpr int JQ = NP / PARTS;
  const int N = P.N;
  const double a0 = P.A[i * 3 + 0], a1 = P.A[i * 3 + 1], a2 = P.A[i * 3 + 2];
  const double bi = P.b[i];
  const int maxpass = 20 * N + 100;

  const long ntodo = LIST ? (long)*P.fb_count : nqp;
  for(long todo = blockIdx.x; todo < ntodo; todo += gridDim.x)
  {
    const long qp = LIST ? (long)P.fb_list[todo] : todo;
    const bool row = lead && i < N;
    const double px = x0[qp * 3 + 0], vx = x0[qp * 3 + 1], ax = x0[qp * 3 + 2];
    double zl = 0, zh = 0;
    if(row)
    {
      zl = zlim[qp * 2 * N + i];
      zh = zlim[qp * 2 * N + N + i];
    }
    const double fr = a0 * px + a1 * vx + a2 * ax;
    const double lo = row ? zl - fr : -kInf;
    const double hi = row ? zh - fr : kInf;
    const double tl = row ? 1e-12 * (1.0 + fabs(lo)) : 0.0;
    const double th = row ? 1e-12 * (1.0 + fabs(hi)) : 0.0;
    int st = CCC_STATUS_SOLVED;
    if(__syncthreads_or(row && lo > hi)) st = CCC_STATUS_INFEASIBLE;

    for(int j = part * JQ; j < (part + 1) * JQ; ++j) T[j * TS + i] = P.G[j * NP + i];
    __syncthreads();

    double z = 0.0, mu = 0.0, dact = 0.0;
    bool inW = false;
    int p = 0;
    double psig = 0.0, pd = 0.0, sig = 0.0;
    bool done = st != CCC_STATUS_SOLVED; // block uniform
    bool need_select = true;
    int passes = 0;
#ifdef CCC_ZMP_PROF
    long long pc[6] = {0, 0, 0, 0, 0, 0}, pt = __builtin_readcyclecounter(), pn;
#undef ZPROF
#define ZPROF(k) pn = __builtin_readcyclecounter(); pc[k] += pn - pt; pt = pn;
#else
#undef ZPROF
#define ZPROF(k)
#endif

    // per wavefront: the most violated bound among the rows outside the working set, and its side
    auto post_select = [&]() {
      const double sl = (lo - z) - tl, sh = (z - hi) - th;
      const double key = (inW || !row) ? kInf : -fmax(sl, sh);
      const double wm = WaveGroup<64>::min(key);
      const int wi = WaveGroup<64>::first(key == wm);
      const double wsig = wave_lane_value((sl >= sh) ? 1.0 : -1.0, wi & 63);
      const int w = threadIdx.x >> 6;
      if((threadIdx.x & 63) == 0 && w < (NP + 63) / 64)
      {
        sel->val[w] = wm;
        sel->sig[w] = wsig;
        sel->idx[w] = wi < 64 ? wi + 64 * w : NP;
      }
    };
    for(int round = 0; round < 3 && !done; ++round)
    {
      post_select();
      __syncthreads();
      while(!done)
      {
        ZPROF(5)
        if(need_select) // the candidates were posted before the previous barrier (post_select)
        {
          double best = sel->val[0], sg = sel->sig[0];
          int cand = sel->idx[0];
#pragma unroll
          for(int k = 1; k < (NP + 63) / 64; ++k)
          {
            const double a = sel->val[k];
            const int ia = sel->idx[k];
            const bool take = (ia < NP) && (cand >= NP || a < best);
            best = take ? a : best;
            sg = take ? sel->sig[k] : sg;
            cand = take ? ia : cand;
          }
          if(!(-best > 0.0)) break;
          p = cand;
          sig = sg;
          if(lead && i == cand)
          {
            psig = sg;
            pd = (sg > 0.0) ? lo : hi;
          }
        }
        ZPROF(0)
        const double c = T[p * TS + i]; // column p = row p (symmetric)
        const double dm = -sig * c;
        const bool blocking = inW && ((mu > 0.0 && dm < 0.0) || (mu < 0.0 && dm > 0.0));
        const bool isp = lead && (i == p);
        const double num = isp ? psig * (pd - z) : -mu;
        const double den = isp ? c : dm;
        const double ratio = (isp || blocking) ? num / den : kInf;
        double t;
        int kk;
        block_argmin<NP, false>(ratio, red, t, kk); // (red was last read three barriers ago)
        if(kk >= NP)
        {
          st = CCC_STATUS_MAX_ITER;
          done = true;
          break;
        }
        ZPROF(1)
        const bool isadd = (kk == p);
        const double s = isadd ? 1.0 : -1.0;
        if(inW)
          mu = fma(t, dm, mu);
        else
          z = fma(sig * t, c, z);
        if(isp) mu += sig * t;
        // bookkeeping of the step, and the candidates of the next selection (they do not depend on the tableau
        // update: posting them here lets the selection ride on the barriers of the update)
        if(isadd)
        {
          if(isp)
          {
            inW = true;
            z = pd;
            dact = pd;
          }
          need_select = true;
        }
        else
        {
          if(lead && i == kk)
          {
            inW = false;
            mu = 0.0;
          }
          need_select = false;
        }
        if(need_select) post_select();
        // pivot on row/column kk
        const double v = T[kk * TS + i];
        if(lead) cb[i] = v;
        __syncthreads();
        const double rp = 1.0 / cb[kk];
        const double g = v * rp;
        ZPROF(2)
        // column i of the tableau, sixteen entries at a time: all loads of a chunk are issued before its stores (T and
        // cb may both be LDS, so the compiler must assume they alias and would otherwise serialise load - store - load)
        constexpr int CH = (JQ % 16 == 0) ? 16 : 8;
        static_assert(JQ % CH == 0, "a part's share of a column must be whole chunks");
        for(int j0 = part * JQ; j0 < (part + 1) * JQ; j0 += CH)
        {
          double tv[CH], cv[CH];
#pragma unroll
          for(int q = 0; q < CH; ++q) tv[q] = T[(j0 + q) * TS + i];
#pragma unroll
          for(int q = 0; q < CH; ++q) cv[q] = cb[j0 + q];
#pragma unroll
          for(int q = 0; q < CH; ++q) tv[q] = fma(-g, cv[q], tv[q]);
#pragma unroll
          for(int q = 0; q < CH; ++q) T[(j0 + q) * TS + i] = tv[q];
        }
        __syncthreads();
        ZPROF(3)
        // row and column kk (thread kk's column came out of the loop as rounding noise) and the pivot itself
        if(lead)
        {
          const double e = (i == kk) ? -rp : s * g;
          T[kk * TS + i] = e;
          T[i * TS + kk] = e;
        }
        __syncthreads();
        ZPROF(4)
        if(++passes > maxpass)
        {
          st = CCC_STATUS_MAX_ITER;
          done = true;
        }
      }
      if(st != CCC_STATUS_SOLVED) break;
      // closing refinement (see zmp_plan_kernel)
      __syncthreads();
      if(lead) cb[i] = inW ? mu : 0.0;
      __syncthreads();
      double acc = 0.0;
      for(int j = 0; j < NP; ++j) acc = fma(P.G[j * NP + i], cb[j], acc);
      const double rho = inW ? dact - acc : 0.0;
      __syncthreads();
      if(lead) cb[i] = rho;
      __syncthreads();
      double tr = 0.0;
      for(int j = 0; j < NP; ++j) tr = fma(T[j * TS + i], cb[j], tr);
      if(inW) mu -= tr;
      __syncthreads();
      if(lead) cb[i] = inW ? mu : 0.0;
      __syncthreads();
      acc = 0.0;
      for(int j = 0; j < NP; ++j) acc = fma(P.G[j * NP + i], cb[j], acc);
      z = inW ? dact : acc;
      const double sl = (lo - z) - tl, sh = (z - hi) - th;
      const int reopen = __syncthreads_or(row && !inW && fmax(sl, sh) > 0.0);
      need_select = true;
      if(!reopen) break;
    }

    ZPROF(5)
    // outputs
    __syncthreads();
    if(lead) cb[i] = row ? mu : 0.0;
    __syncthreads();
    if(lead && i == 0)
    {
      double u0 = 0.0;
      for(int r = 0; r < N; ++r) u0 = fma(P.b[r], cb[r], u0);
      const double cdt = control_dt < 0 ? P.dt : control_dt;
      const double com_acc = ax + cdt * u0;
      const double com_pos = px + cdt * vx + 0.5 * (cdt * cdt) * ax;
      double zv = com_pos + P.c2 * com_acc;
      zv = zv < zl ? zl : (zh < zv ? zh : zv);
      zmp[qp] = zv;
      if(status) status[qp] = (passes << 8) | st;
    }
    if(jerk && row)
    {
      double uj = 0.0;
      for(int r = i; r < N; ++r) uj = fma(P.b[r - i], cb[r], uj);
      jerk[qp * N + i] = uj;
    }
    __syncthreads();
#ifdef CCC_ZMP_PROF
    if(jerk && qp == 0 && threadIdx.x == 0)
      for(int q = 0; q < 6; ++q) jerk[q] = (double)pc[q];
#endif
    (void)bi;
  }
}

template<int NP, bool HBM, int PARTS>
__global__ __launch_bounds__(NP * PARTS) void zmp_plan_block_kernel(ZmpDev P, long nqp, const double * __restrict__ x0,
                                                            const double * __restrict__ zlim, double control_dt,
                                                            double * __restrict__ zmp, double * __restrict__ jerk,
                                                            int * __restrict__ status, double * __restrict__ ws)
{
  zmp_plan_block_body<NP, HBM, PARTS, false>(P, nqp, x0, zlim, control_dt, zmp, jerk, status, ws);
}

// (the QPs KS hands over, zmp_stage.inc: P.fb_list[0 .. *P.fb_count))
template<int NP, bool HBM, int PARTS>
__global__ __launch_bounds__(NP * PARTS) void zmp_plan_block_list_kernel(ZmpDev P, long nqp, const double * __restrict__ x0,
                                                                 const double * __restrict__ zlim, double control_dt,
                                                                 double * __restrict__ zmp, double * __restrict__ jerk,
                                                                 int * __restrict__ status, double * __restrict__ ws)
{
  zmp_plan_block_body<NP, HBM, PARTS, true>(P, nqp, x0, zlim, control_dt, zmp, jerk, status, ws);
}
// K2.  32 < N <= 200 (the reference test's 2 s @ 20 ms = 100 steps; BASELINE configs[0] as worded, 2 s @ 10 ms = 200):
// one QP per workgroup, the sweep tableau PACKED (lower triangle in 4 x 4 tiles -- 2 x 2 at 200 rows --, sym_tableau.h)
// in LDS; thread t updates the tiles t, t + NT, ..., thread i < NP owns row i (bounds, multiplier, flags).  One to
// twelve workgroups share a CU (7 KB at 40 rows ... 160 KB at 200), one workgroup is one to sixteen wavefronts.
// GS = row stride of P.G.
template<int NP, int TS, int TPT, bool LIST>
__device__ __forceinline__ void zmp_plan_sym_body(ZmpDev P, int GS, long nqp,
                                                                      const double * __restrict__ x0,
                                                                      const double * __restrict__ zlim, double control_dt,
                                                                      double * __restrict__ zmp, double * __restrict__ jerk,
                                                                      int * __restrict__ status)
{
  using ST = SymTab<NP, TS, TPT>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double * T = smem;                 // packed tableau
  double * cb = smem + ST::kDoubles; // [NP] staging of the pivot row / of mu / of rho
  BlockRed * red = reinterpret_cast<BlockRed *>(cb + NP);
  SelRed * sel = reinterpret_cast<SelRed *>(red + 1);
  const int i = threadIdx.x;
  const bool lead = i < NP;
  int ta[ST::TPT], tb[ST::TPT]; // this thread's tiles: i, i + NT, ...
#pragma unroll
  for(int k = 0; k < ST::TPT; ++k) ST::tile_of(i + k * ST::NT < ST::NTILE ? i + k * ST::NT : 0, ta[k], tb[k]);
  const int N = P.N;
  const double a0 = lead ? P.A[i * 3 + 0] : 0.0, a1 = lead ? P.A[i * 3 + 1] : 0.0, a2 = lead ? P.A[i * 3 + 2] : 0.0;
  const int maxpass = 20 * N + 100;

  const long ntodo = LIST ? (long)*P.fb_count : nqp;
  for(long todo = blockIdx.x; todo < ntodo; todo += gridDim.x)
  {
    const long qp = LIST ? (long)P.fb_list[todo] : todo;
#ifdef CCC_ZMP_PROF
    const long long rtq = wall_clock64();
#endif
    const bool row = lead && i < N;
    const double px = x0[qp * 3 + 0], vx = x0[qp * 3 + 1], ax = x0[qp * 3 + 2];
    double zl = 0, zh = 0;
    if(row)
    {
      zl = zlim[qp * 2 * N + i];
      zh = zlim[qp * 2 * N + N + i];
    }
    const double fr = a0 * px + a1 * vx + a2 * ax;
    const double lo = row ? zl - fr : -kInf;
    const double hi = row ? zh - fr : kInf;
    const double tl = row ? 1e-12 * (1.0 + fabs(lo)) : 0.0;
    const double th = row ? 1e-12 * (1.0 + fabs(hi)) : 0.0;
    int st = CCC_STATUS_SOLVED;
    if(__syncthreads_or(row && lo > hi)) st = CCC_STATUS_INFEASIBLE;

#pragma unroll
    for(int k = 0; k < ST::TPT; ++k)
      if(i + k * ST::NT < ST::NTILE) ST::load_tile(T, P.G, GS, i + k * ST::NT, ta[k], tb[k]);
    __syncthreads();

    double z = 0.0, mu = 0.0, dact = 0.0;
    bool inW = false;
    int p = 0;
    double psig = 0.0, pd = 0.0, sig = 0.0;
    bool done = st != CCC_STATUS_SOLVED; // block uniform
    bool need_select = true;
    int passes = 0;
#ifdef CCC_ZMP_PROF
    long long pc[6] = {0, 0, 0, 0, 0, 0}, pt = __builtin_readcyclecounter(), pn;
    const long long rt0 = wall_clock64(), ct0 = pt;
#undef ZPROF
#define ZPROF(k) pn = __builtin_readcyclecounter(); pc[k] += pn - pt; pt = pn;
#else
#undef ZPROF
#define ZPROF(k)
#endif

    // per wavefront: the most violated bound among the rows outside the working set, and its side
    auto post_select = [&]() {
      const double sl = (lo - z) - tl, sh = (z - hi) - th;
      const double key = (inW || !row) ? kInf : -fmax(sl, sh);
      const double wm = WaveGroup<64>::min(key);
      const int wi = WaveGroup<64>::first(key == wm);
      const double wsig = wave_lane_value((sl >= sh) ? 1.0 : -1.0, wi & 63);
      const int w = threadIdx.x >> 6;
      if((threadIdx.x & 63) == 0 && w < (NP + 63) / 64)
      {
        sel->val[w] = wm;
        sel->sig[w] = wsig;
        sel->idx[w] = wi < 64 ? wi + 64 * w : NP;
      }
    };
    for(int round = 0; round < 3 && !done; ++round)
    {
      post_select();
      __syncthreads();
      while(!done)
      {
        ZPROF(5)
        if(need_select) // the candidates were posted before the previous barrier (post_select)
        {
          double best = sel->val[0], sg = sel->sig[0];
          int cand = sel->idx[0];
#pragma unroll
          for(int k = 1; k < (NP + 63) / 64; ++k)
          {
            const double a = sel->val[k];
            const int ia = sel->idx[k];
            const bool take = (ia < NP) && (cand >= NP || a < best);
            best = take ? a : best;
            sg = take ? sel->sig[k] : sg;
            cand = take ? ia : cand;
          }
          if(!(-best > 0.0)) break;
          p = cand;
          sig = sg;
          if(lead && i == cand)
          {
            psig = sg;
            pd = (sg > 0.0) ? lo : hi;
          }
        }
        ZPROF(0)
        const double c = lead ? T[ST::entry(p, i)] : 0.0; // column p = row p (symmetric)
        if(lead) cb[i] = c; // staged as the pivot column already: most pivots add row p itself (published by the
                            // barrier of the ratio test below)
        const double dm = -sig * c;
        const bool blocking = inW && ((mu > 0.0 && dm < 0.0) || (mu < 0.0 && dm > 0.0));
        const bool isp = lead && (i == p);
        const double num = isp ? psig * (pd - z) : -mu;
        const double den = isp ? c : dm;
        const double ratio = (isp || blocking) ? num / den : kInf;
        double t;
        int kk;
        block_argmin<NP, false>(ratio, red, t, kk); // (red was last read three barriers ago)
        if(kk >= NP)
        {
          st = CCC_STATUS_MAX_ITER;
          done = true;
          break;
        }
        ZPROF(1)
        const bool isadd = (kk == p);
        const double s = isadd ? 1.0 : -1.0;
        if(inW)
          mu = fma(t, dm, mu);
        else
          z = fma(sig * t, c, z);
        if(isp) mu += sig * t;
        // bookkeeping of the step, and the candidates of the next selection (they do not depend on the tableau
        // update: posting them here lets the selection ride on the barriers of the update)
        if(isadd)
        {
          if(isp)
          {
            inW = true;
            z = pd;
            dact = pd;
          }
          need_select = true;
        }
        else
        {
          if(lead && i == kk)
          {
            inW = false;
            mu = 0.0;
          }
          need_select = false;
        }
        if(need_select) post_select();
        // pivot on row/column kk
        double v = c;
        if(!isadd) // a row leaves: stage its column instead
        {
          v = lead ? T[ST::entry(kk, i)] : 0.0;
          if(lead) cb[i] = v;
          __syncthreads();
        }
        const double rp = 1.0 / cb[kk];
        const double g = v * rp;
        ZPROF(2)
#pragma unroll
        for(int k = 0; k < ST::TPT; ++k)
          if(i + k * ST::NT < ST::NTILE) ST::update_tile(T, cb, rp, i + k * ST::NT, ta[k], tb[k]);
        __syncthreads();
        ZPROF(3)
        // row/column kk (the update left rounding noise there) and the pivot itself
        if(lead)
        {
          T[ST::entry(kk, i)] = (i == kk) ? -rp : s * g;
        }
        __syncthreads();
        ZPROF(4)
        if(++passes > maxpass)
        {
          st = CCC_STATUS_MAX_ITER;
          done = true;
        }
      }
      if(st != CCC_STATUS_SOLVED) break;
      // closing refinement (see zmp_plan_kernel)
      __syncthreads();
      if(lead) cb[i] = inW ? mu : 0.0;
      __syncthreads();
      double acc = 0.0;
      if(lead)
        for(int j = 0; j < NP; ++j) acc = fma(P.G[j * GS + i], cb[j], acc);
      const double rho = inW ? dact - acc : 0.0;
      __syncthreads();
      if(lead) cb[i] = rho;
      __syncthreads();
      const double tr = lead ? ST::matvec_row(T, cb, i) : 0.0;
      if(inW) mu -= tr;
      __syncthreads();
      if(lead) cb[i] = inW ? mu : 0.0;
      __syncthreads();
      acc = 0.0;
      if(lead)
        for(int j = 0; j < NP; ++j) acc = fma(P.G[j * GS + i], cb[j], acc);
      z = inW ? dact : acc;
      const double sl = (lo - z) - tl, sh = (z - hi) - th;
      const int reopen = __syncthreads_or(row && !inW && fmax(sl, sh) > 0.0);
      need_select = true;
      if(!reopen) break;
    }

    ZPROF(5)
    // outputs
    __syncthreads();
    if(lead) cb[i] = row ? mu : 0.0;
    __syncthreads();
    if(lead && i == 0)
    {
      double u0 = 0.0;
      for(int r = 0; r < N; ++r) u0 = fma(P.b[r], cb[r], u0);
      const double cdt = control_dt < 0 ? P.dt : control_dt;
      const double com_acc = ax + cdt * u0;
      const double com_pos = px + cdt * vx + 0.5 * (cdt * cdt) * ax;
      double zv = com_pos + P.c2 * com_acc;
      zv = zv < zl ? zl : (zh < zv ? zh : zv);
      zmp[qp] = zv;
      if(status) status[qp] = (passes << 8) | st;
    }
    if(jerk && row)
    {
      double uj = 0.0;
      for(int r = i; r < N; ++r) uj = fma(P.b[r - i], cb[r], uj);
      jerk[qp * N + i] = uj;
    }
    __syncthreads();
#ifdef CCC_ZMP_PROF
    if(jerk && threadIdx.x == 0)
    {
      for(int q = 0; q < 6; ++q) jerk[qp * N + q] = (double)pc[q];
      jerk[qp * N + 6] = (double)(wall_clock64() - rt0);
      jerk[qp * N + 7] = (double)(__builtin_readcyclecounter() - ct0);
      jerk[qp * N + 8] = (double)rtq;
      jerk[qp * N + 9] = (double)(rt0 - rtq);
    }
#endif
  }
}

template<int NP, int TS, int TPT>
__global__ __launch_bounds__((SymTab<NP, TS, TPT>::NT), (SymTab<NP, TS, TPT>::kMinWaves)) void zmp_plan_sym_kernel(ZmpDev P, int GS, long nqp,
                                                                      const double * __restrict__ x0,
                                                                      const double * __restrict__ zlim, double control_dt,
                                                                      double * __restrict__ zmp, double * __restrict__ jerk,
                                                                      int * __restrict__ status)
{
  zmp_plan_sym_body<NP, TS, TPT, false>(P, GS, nqp, x0, zlim, control_dt, zmp, jerk, status);
}

// (the QPs KS hands over, zmp_stage.inc)
template<int NP, int TS, int TPT>
__global__ __launch_bounds__((SymTab<NP, TS, TPT>::NT), (SymTab<NP, TS, TPT>::kMinWaves)) void zmp_plan_sym_list_kernel(ZmpDev P, int GS, long nqp,
                                                                      const double * __restrict__ x0,
                                                                      const double * __restrict__ zlim, double control_dt,
                                                                      double * __restrict__ zmp, double * __restrict__ jerk,
                                                                      int * __restrict__ status)
{
  zmp_plan_sym_body<NP, TS, TPT, true>(P, GS, nqp, x0, zlim, control_dt, zmp, jerk, status);
}

#include "zmp_k2r.inc"
#include "zmp_stage.inc"

} // namespace ccc_amd

// ---------------------------------------------------------------------------------------------
// The schedule of a call WITHOUT a usable history (round 6): a prediction of every QP's pivot trips, fed
// to the same counting sort the history feeds (csrc/common.hip order_by_count); the static kernel then runs the QPs longest
// first, like predictions paired in a wavefront.  What predicts: a QP's trips are the rows in its working set at the optimum
// (15.8 trips for 15.6 rows on the bench workload: rows hardly ever leave), and those are driven by the rows the
// UNCONSTRAINED optimum u = 0 violates, i.e. where the free response A_seq x0 leaves its limits -- the earlier in the
// horizon, the more: a violation a few LIPM time constants ahead is absorbed by a handful of active rows, one that is
// imminent drags its whole neighbourhood onto the bounds.  key = sum_i exp(-t_i / tau) [row i violated], tau = 2.35 sqrt(h / g)
// (0.75 s at h = 1 m), t_i = i dt: measured on the bench workload (numpy, counts of the kernel itself), the two QPs of a
// wavefront then spend 1.18 x their own trips in lock-step (the x and y axes of an instance: 1.39, the plain count of
// violated rows: 1.23, the last call's counts of a repeated batch: 1.00).  One pass over the inputs (HBM-bound, 71 MB at
// the headline), lane = row as in K1.  A schedule only: the answers do not depend on it.
// ---------------------------------------------------------------------------------------------
namespace ccc_amd
{
template<int WAVES>
__global__ __launch_bounds__(WAVES * 64) void zmp_predict_kernel(ZmpDev P, long nqp, const double * __restrict__ x0,
                                                                 const double * __restrict__ zlim, int * __restrict__ pred)
{
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31;
  const long qp = ((long)blockIdx.x * WAVES + (tid >> 6)) * 2 + (lane >> 5);
  const int N = P.N;
  float k = 0.0f;
  if(qp < nqp && li < N)
  {
    const double fr = P.A[li * 3 + 0] * x0[qp * 3 + 0] + P.A[li * 3 + 1] * x0[qp * 3 + 1] + P.A[li * 3 + 2] * x0[qp * 3 + 2];
    const double lo = zlim[qp * 2 * N + li] - fr, hi = zlim[qp * 2 * N + N + li] - fr;
    const float tau = 2.35f * sqrtf((float)-P.c2); // c2 = -h / g
    if(lo > 0.0 || hi < 0.0) k = 16.0f * __expf(-(float)li * (float)P.dt / tau);
  }
  for(int d = 16; d >= 1; d >>= 1) k += __shfl_xor(k, d, 32);
  if(li == 0 && qp < nqp) pred[qp] = (int)(k + 0.5f); // (<= 16 / (1 - exp(-dt / tau)): inside order_by_count's 256 buckets)
}
} // namespace ccc_amd

// =============================================================================================
// host side: model construction and the C-ABI
// =============================================================================================
using namespace ccc_amd;

struct ccc_zmp
{
  int device = 0;
  int N = 0;
  int NP = 0; // padded size = lanes per QP (32 or 64)
  double com_height = 0, horizon_duration = 0, horizon_dt = 0, c2 = 0;
  std::vector<double> A_seq, B_seq; // host copies, N x 3 and N x N
  double *dG = nullptr, *dA = nullptr, *db = nullptr;
  unsigned long long * queue = nullptr; // work-queue ticket counter of zmp_plan_kernel_dyn
  int *hist = nullptr, *order = nullptr; // K1: pivot trips per QP of the last call, and the schedule made from them
  int * order_scratch = nullptr;         // (order_by_count's table)
  int * diff = nullptr;                  // |trips - trips of the call before| per QP, and the verdict it leads to:
  int *trust_host = nullptr, *trust_dev = nullptr; // page-locked host memory, read without waiting (0: do not follow)
  int * pred = nullptr;                  // predicted trips per QP of THIS call (zmp_predict_kernel), when no history is followed
  std::vector<void *> retired;           // scheduling buffers that were outgrown: kept until the handle goes, because a
                                         // hipGraph captured at the smaller size still holds their addresses
  int64_t hist_cap = 0, hist_n = -1;     // (hist_n: the QPs of the call the counts belong to, -1 = none yet)
  int64_t diff_n = -1;                   // (... and of the call the differences belong to)
  unsigned watch = 0;
  bool inputs_in_host = false;           // (set by the host entry while the kernels read the caller's page-locked memory in
                                         //  place: a prediction pass would fetch the inputs over PCIe a second time)
  bool skip_history = false;             // (set by the host entry while it feeds CHUNKS of one batch: a chunk says nothing
                                         //  about the next)
  double * ws_big = nullptr; // HBM tableaus of the 128 < N <= 256 kernel
  // KS (zmp_stage.inc): per-wavefront stage records, and the list of the QPs it hands over to the exact kernel
  double * ws_stage = nullptr;
  int64_t ws_stage_blocks = 0;
  int *fb_list = nullptr, *fb_count = nullptr;
  int64_t fb_cap = 0;
  int num_cu = 0;
  // per-handle (= per-device) launch state: function attributes set, resident workgroups per CU of the queue kernel
  bool attr_set = false, attr_dyn = false;
  int per_cu = 0;
  // staging for the host-pointer entry point
  int64_t cap = 0, cap_dev = 0, cap_status = 0;
  double *h_in = nullptr, *h_out = nullptr; // pinned
  double *d_in = nullptr, *d_out = nullptr;
  int32_t *h_status = nullptr, *d_status = nullptr;
  hipStream_t stream = nullptr;
  // development switches, read ONCE in ccc_zmp_create (never per launch)
  int64_t env_queue_min = -1;   // CCC_ZMP_QUEUE_MIN: QPs from which the work-queue kernel runs (< 0: the measured default)
  bool env_static = false;      // CCC_ZMP_STATIC: never the work-queue kernel
  bool env_history = true;      // CCC_ZMP_HISTORY=0: never order a call by the last call's pivot counts
  bool env_predict = true;      // CCC_ZMP_PREDICT=0: never order a call by the predicted pivot counts (zmp_predict_kernel)
  int64_t env_predict_min = -1; // CCC_ZMP_PREDICT_MIN: QPs from which a call without a followed history is ordered by the
                                //                      prediction (< 0: the measured default)
  bool env_debug = false;       // CCC_ZMP_DEBUG: print the occupancy of the LDS-tableau kernels
  int env_kw = -1;              // CCC_ZMP_KW: 0 = never the one-QP-per-wavefront register kernel (K1w) for 32 < N <= 64 (the
                                //             default there: 39.2 / 34.1 / 28.9 / 19.2 / 17.5 M solves/s at N = 33 / 40 / 48 /
                                //             56 / 64 against 38.6 / 33.5 / 18.9 / 16.3 / 9.7 M for K2 / K2r, batch 65536);
                                //             2 = its 48-column build at two wavefronts per SIMD without spills (slower)
  int env_k2 = -1;              // CCC_ZMP_K2: 0 = the LDS tableau (K2) for every 32 < N <= 200, 1 = the register tiles (K2r)
                                //             wherever they are built (12 / 13: with two / three tiles per thread); < 0: the
                                //             measured default per size
  int64_t env_host_chunk = 0;   // CCC_ZMP_HOST_CHUNK: staging chunk of the host entry (0: the default)
  int env_stage = -1;           // CCC_ZMP_STAGE: 0 = never the stage-recursion kernel (KS), 1 = for every N > 32 and
                                //                batch; < 0: the measured default (N >= 40, from a batch size that
                                //                depends on the horizon: launch_block)
  int64_t env_stage_min = -1;   // CCC_ZMP_STAGE_MIN: QPs from which KS runs (< 0: the measured default)
  int env_stage_iters = 0;      // CCC_ZMP_STAGE_ITERS: KS's iteration limit before a QP is handed over (0: the default)
  int env_stage_waves = 0;      // CCC_ZMP_STAGE_WAVES: KS's wavefronts per SIMD at most (0: the default)
  int env_stage_pen = -1;       // CCC_ZMP_STAGE_PEN: KS's iterations on the penalised problem at most (< 0: the default)
  const char * last_kernel = "none"; // the kernel the last plan call launched (ccc_zmp_last_kernel)
  const char * last_order = "none";  // ... and what its schedule came from (CCC_ZMP_DEBUG prints it)
};

namespace
{
constexpr double kG = 9.80665; // include/CCC/Constants.h:10

// Closed form of src/CommonModels.cpp:8-17 + StateSpaceModel.h:164-216 + InvariantSequentialExtension.h:103-181
// for the jerk-input CoM-ZMP model: the continuous A is nilpotent, so exp([[A,B],[0,0]] dt) terminates:
//   Ad^n = [[1, n dt, (n dt)^2/2],[0, 1, n dt],[0, 0, 1]],  Bd = [dt^3/6, dt^2/2, dt]',  C = [1, 0, -h/g]
//   A_seq[i] = C Ad^(i+1),   B_seq[i][j] = C Ad^(i-j) Bd  (j <= i)
void build_model(ccc_zmp * h)
{
  const int N = h->N;
  const double dt = h->horizon_dt, c2 = -1 * h->com_height / kG;
  h->c2 = c2;
  h->A_seq.assign((size_t)N * 3, 0.0);
  h->B_seq.assign((size_t)N * N, 0.0);
  std::vector<double> b(N);
  const double Bd[3] = {dt * dt * dt / 6.0, dt * dt / 2.0, dt};
  for(int n = 0; n < N; n++)
  {
    const double nd = n * dt;
    // C Ad^n Bd = Bd0 + nd Bd1 + nd^2/2 Bd2 + c2 Bd2
    b[n] = Bd[0] + nd * Bd[1] + 0.5 * nd * nd * Bd[2] + c2 * Bd[2];
  }
  for(int i = 0; i < N; i++)
  {
    const double nd = (i + 1) * dt;
    h->A_seq[i * 3 + 0] = 1.0;
    h->A_seq[i * 3 + 1] = nd;
    h->A_seq[i * 3 + 2] = 0.5 * nd * nd + c2;
    for(int j = 0; j <= i; j++) h->B_seq[(size_t)i * N + j] = b[i - j];
  }
}

int upload_model(ccc_zmp * h)
{
  const int N = h->N, NP = h->NP;
  std::vector<double> G((size_t)NP * NP, 0.0), A((size_t)NP * 3, 0.0), b(NP, 0.0);
  for(int i = 0; i < N; i++)
  {
    for(int j = 0; j < N; j++)
    {
      long double s = 0;
      for(int k = 0; k < N; k++) s += (long double)h->B_seq[(size_t)i * N + k] * h->B_seq[(size_t)j * N + k];
      G[(size_t)i * NP + j] = (double)s;
    }
    for(int k = 0; k < 3; k++) A[i * 3 + k] = h->A_seq[i * 3 + k];
    b[i] = h->B_seq[(size_t)i * N + 0];
  }
  for(int i = N; i < NP; i++) G[(size_t)i * NP + i] = 1.0;
  CCC_HIP_CHECK(hipMalloc(&h->dG, G.size() * sizeof(double)));
  CCC_HIP_CHECK(hipMalloc(&h->dA, A.size() * sizeof(double)));
  CCC_HIP_CHECK(hipMalloc(&h->db, b.size() * sizeof(double)));
  CCC_HIP_CHECK(hipMemcpy(h->dG, G.data(), G.size() * sizeof(double), hipMemcpyHostToDevice));
  CCC_HIP_CHECK(hipMemcpy(h->dA, A.data(), A.size() * sizeof(double), hipMemcpyHostToDevice));
  CCC_HIP_CHECK(hipMemcpy(h->db, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice));
  return CCC_OK;
}

template<int LG, int WAVES>
int launch(ccc_zmp * h, int64_t n, const double * x0, const double * zlim, double control_dt, double * zmp,
           double * jerk, int32_t * status, hipStream_t stream)
{
  constexpr int QPW = 64 / LG;
  const size_t lds = ((size_t)LG * LG + 4 * LG + (size_t)WAVES * QPW * ZmpScratch<LG>::kSize) * sizeof(double);
  if(!h->attr_set)
  {
    CCC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&zmp_plan_kernel<LG, WAVES>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->attr_set = true;
  }
  const int64_t nqp = 2 * n;
  const int64_t ntask = (nqp + QPW - 1) / QPW;
  const int64_t want = (ntask + WAVES - 1) / WAVES;
  // a few blocks per resident slot: the hardware dispatcher then evens out the data-dependent pivot counts
  const int64_t resident = (int64_t)h->num_cu * (16 / WAVES);
  const int grid = (int)std::min<int64_t>(want, resident * 8);
  ZmpDev P{h->N, h->dG, h->dA, h->db, h->c2, h->horizon_dt};
  // round 5: a handle remembers the pivot trips of every QP of its last call; a call of the same size runs them longest
  // first (and, in the static kernel, pairs QPs of like counts in a wavefront).  Closed-loop callers repeat their batch
  // from cycle to cycle; for anybody else the order is as good as any other -- the answers never depend on it.
  // (inside a stream capture the buffers are not grown: the call runs unordered and keeps no counts)
  bool ordered = false, hist_verdict = false;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  // (below ~3000 instances a call is shorter than what ordering it costs -- the counting sort is two launches, 14 us:
  //  2048 instances 56.5 M solves/s unordered against 48.0 M in the order of a repeated batch's own counts; from 4096 the
  //  order wins, 73.9 -> 89.4 M)
  constexpr int64_t kSchedMinQp = 6144;
  const bool sched_ok = (h->env_history || h->env_predict) && !h->skip_history && nqp >= kSchedMinQp && nqp < (int64_t)1 << 30
                        && !(capturing && h->hist_cap < nqp);
  if(sched_ok && h->hist_cap < nqp)
  {
    // (the outgrown buffers are RETIRED, not freed: a hipGraph captured at the smaller size replays launches that read and
    //  write them -- ADVICE r5: freed, that was silent corruption; they go with the handle)
    for(int * q : {h->hist, h->order, h->diff, h->pred})
      if(q) h->retired.push_back(q);
    // (capacity grows by at least half, so that a caller whose batches creep upwards retires a geometric series of
    //  buffers -- at most twice the final size in all -- not one set per call)
    const int64_t cap = std::max<int64_t>(nqp, h->hist_cap + h->hist_cap / 2);
    h->hist = h->order = h->diff = h->pred = nullptr;
    h->hist_cap = 0;
    h->hist_n = -1;
    CCC_HIP_CHECK(hipMalloc(&h->hist, (size_t)cap * sizeof(int)));
    CCC_HIP_CHECK(hipMalloc(&h->order, (size_t)cap * sizeof(int)));
    CCC_HIP_CHECK(hipMalloc(&h->diff, (size_t)cap * sizeof(int)));
    CCC_HIP_CHECK(hipMalloc(&h->pred, (size_t)cap * sizeof(int)));
    if(!h->order_scratch) CCC_HIP_CHECK(hipMalloc(&h->order_scratch, (size_t)kOrderScratchInts * sizeof(int)));
    if(!h->trust_host)
    {
      CCC_HIP_CHECK(hipHostMalloc(&h->trust_host, 64, hipHostMallocMapped));
      *h->trust_host = 1;
      CCC_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->trust_dev), h->trust_host, 0));
    }
    h->hist_cap = cap;
  }
  if(sched_ok && h->env_history)
  {
    // (a history that does not predict -- unrelated batches of one size, call after call -- is worse than none: the two
    //  axes of an instance are better company for each other than two QPs picked by a wrong guess, 101 against 110 M
    //  solves/s at 65536.  A call that has counts to compare with keeps the differences; the sort of the next call, or a
    //  pass of its own when the history is not followed, turns them into a verdict that is read here a call or two late)
    const bool comparable = h->hist_n == nqp, verdict = comparable && h->diff_n == nqp;
    const bool trusted = *static_cast<volatile int *>(h->trust_host) != 0;
    ordered = comparable && trusted;
    // (round 6: while the history is NOT followed it is still watched, but only three calls in eight keep counts -- the
    //  first for the second to compare with, the third to sum the differences up: the twelve bytes per QP a recording call
    //  writes and the 7 us of the counting pass were 3 % of a call of rotating batches, 15 % at 4096 instances)
    const unsigned phase = trusted ? 7u : (h->watch++ & 7u);
    const bool record = trusted || phase >= 5u;
    P.hist = record ? h->hist : nullptr;
    P.order = ordered ? h->order : nullptr;
    P.diff = (record && comparable) ? h->diff : nullptr;
    if(verdict && !ordered && phase == 7u)
      if(int arc = order_by_count(h->hist, (int)nqp, nullptr, h->order_scratch, nullptr, 0, nullptr, stream, h->diff, h->trust_dev))
        return arc;
    hist_verdict = verdict;
    h->diff_n = (record && comparable) ? nqp : -1;
    if(!record) h->hist_n = -1; // (the counts in the buffer are not the previous call's any more)
  }
  // large batches: a work queue per 32-lane group (zmp_plan_kernel_dyn); below ~18 QPs per resident group the static
  // pairing (one instance per wavefront, more workgroups than fit: the hardware dispatcher balances) is faster -- measured
  // static / queue in M solves/s: 83.7 / 65.4 at 16384 instances, 90.3 / 83.3 at 32768, 95.1 / 93.2 at 49152,
  // 96.2 / 96.5 at 57344, 94.4 / 98.3 at 65536
  const int64_t queue_min = h->env_queue_min >= 0 ? h->env_queue_min : (int64_t)18 * h->num_cu * 12 * QPW;
  // (round 5: with an order from the last call the static kernel is the faster one at every size -- the wavefront's two QPs
  //  have like pivot counts, which is what the queue was for, without its divergent set-up: 127 against 96 M at 65536, 106
  //  against 78 M at 8192 instances, the batch repeated; the queue stays for calls without a history)
  // round 6: a call without a followed history is ordered by the PREDICTED trips (zmp_predict_kernel above) and runs on the
  // static kernel as well -- measured at 65536 instances, eight unrelated batches in turn: 114.5 (queue kernel) -> 130.2 M
  // solves/s on a default handle, 118.4 -> 133.0 M on one that keeps no history; 8192 instances: 96.7 M, 4096 (below the
  // threshold: the prediction's pass costs what it saves): 72 M
  const int64_t predict_min = h->env_predict_min >= 0 ? h->env_predict_min : 16384;
  const bool predicted = sched_ok && h->env_predict && !ordered && LG == 32 && nqp >= predict_min && !h->inputs_in_host;
  const bool use_queue = !h->env_static && nqp >= queue_min && !ordered && !predicted;
  h->last_kernel = use_queue ? (LG == 32 ? "zmp_plan_kernel_dyn<32,2>" : "zmp_plan_kernel_dyn")
                             : (LG == 32 ? "zmp_plan_kernel<32,2>" : "zmp_plan_kernel");
  h->last_order = ordered ? "last call's pivot counts" : (predicted ? "predicted pivot counts" : "none");
  if(use_queue)
  {
    if(!h->attr_dyn)
    {
      CCC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&zmp_plan_kernel_dyn<LG, WAVES>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      h->attr_dyn = true;
    }
    const size_t qbytes = (size_t)kQueues * kQueueStride * sizeof(unsigned long long);
    if(!h->queue)
    {
      CCC_NO_CAPTURE(stream, "ccc_zmp_plan_batch_device");
      CCC_HIP_CHECK(hipMalloc(&h->queue, qbytes));
    }
    if(int zrc = zero_words(h->queue, (int)(qbytes / 4), stream)) return zrc;
    int & per_cu = h->per_cu; // resident workgroups per CU (the kernel loops on the queue: one grid-full is all it needs)
    if(per_cu == 0)
    {
      int nb = 0;
      if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, zmp_plan_kernel_dyn<LG, WAVES>, WAVES * 64, lds) != hipSuccess || nb < 1)
        nb = 16 / WAVES;
      per_cu = nb;
    }
    const int gdyn = (int)std::min<int64_t>(want, (int64_t)h->num_cu * per_cu);
    hipLaunchKernelGGL((zmp_plan_kernel_dyn<LG, WAVES>), dim3(gdyn), dim3(WAVES * 64), lds, stream, P, (long)nqp, x0, zlim,
                       control_dt, zmp, jerk, status, h->queue);
    CCC_HIP_CHECK(hipGetLastError());
    if(P.hist) h->hist_n = nqp;
    return CCC_OK;
  }
  // the schedule of this call from a prediction of every QP's trips (no history followed) ...
  if(predicted)
  {
    constexpr int PW = 4;
    hipLaunchKernelGGL((zmp_predict_kernel<PW>), dim3((unsigned)((nqp + 2 * PW - 1) / (2 * PW))), dim3(PW * 64), 0, stream, P,
                       (long)nqp, x0, zlim, h->pred);
    if(int orc = order_by_count(h->pred, (int)nqp, h->order, h->order_scratch, nullptr, 0, nullptr, stream, nullptr, nullptr))
      return orc;
    P.order = h->order;
  }
  // ... or from the pivot counts of the last one: a counting sort, longest first (common.hip)
  if(ordered)
    if(int orc = order_by_count(h->hist, (int)nqp, h->order, h->order_scratch, nullptr, 0, nullptr, stream, hist_verdict ? h->diff : nullptr,
                                 h->trust_dev)) return orc;
  hipLaunchKernelGGL((zmp_plan_kernel<LG, WAVES>), dim3(grid), dim3(WAVES * 64), lds, stream, P, (long)nqp, x0, zlim,
                     control_dt, zmp, jerk, status);
  CCC_HIP_CHECK(hipGetLastError());
  if(P.hist) h->hist_n = nqp;
  return CCC_OK;
}
// the exact (dual active set) kernels of N > 32.  from_list: solve the QPs KS handed over (h->fb_list[0 .. *h->fb_count),
// both known on the device only) with a grid sized for a few per cent of the batch; blocks beyond the count return at once
int launch_exact(ccc_zmp * h, int64_t n, const double * x0, const double * zlim, double control_dt, double * zmp,
                 double * jerk, int32_t * status, hipStream_t stream, bool from_list)
{
  const int64_t nqp = 2 * n;
  ZmpDev P{h->N, h->dG, h->dA, h->db, h->c2, h->horizon_dt};
  if(from_list)
  {
    P.fb_list = h->fb_list;
    P.fb_count = h->fb_count;
  }
  const int64_t grid_qp = from_list ? std::min<int64_t>(nqp, (int64_t)h->num_cu * 12) : nqp; // (one workgroup per QP)
  const char * const stage_name = h->last_kernel;
  h->last_kernel = h->N > 200 ? "zmp_plan_block_kernel" : "zmp_plan_sym_kernel";
  if(h->N > 200) // beyond the LDS: the tableau in HBM
  {
    const int NPb = h->NP; // kBigNP or kHugeNP
    const int blocks = h->num_cu * 2;
    if(!h->ws_big)
    {
      CCC_NO_CAPTURE(stream, "ccc_zmp_plan_batch_device");
      CCC_HIP_CHECK(hipMalloc(&h->ws_big, (size_t)blocks * NPb * NPb * sizeof(double)));
    }
    const size_t lds = (size_t)NPb * sizeof(double) + sizeof(BlockRed) + sizeof(SelRed);
    const int grid = (int)std::min<int64_t>(grid_qp, blocks);
    if(NPb == kBigNP && !from_list)
      hipLaunchKernelGGL((zmp_plan_block_kernel<kBigNP, true, 2>), dim3(grid), dim3(kBigNP * 2), lds, stream, P, (long)nqp,
                         x0, zlim, control_dt, zmp, jerk, status, h->ws_big);
    else if(NPb == kBigNP)
      hipLaunchKernelGGL((zmp_plan_block_list_kernel<kBigNP, true, 2>), dim3(grid), dim3(kBigNP * 2), lds, stream, P,
                         (long)nqp, x0, zlim, control_dt, zmp, jerk, status, h->ws_big);
    else if(!from_list)
      hipLaunchKernelGGL((zmp_plan_block_kernel<kHugeNP, true, 2>), dim3(grid), dim3(kHugeNP * 2), lds, stream, P, (long)nqp,
                         x0, zlim, control_dt, zmp, jerk, status, h->ws_big);
    else
      hipLaunchKernelGGL((zmp_plan_block_list_kernel<kHugeNP, true, 2>), dim3(grid), dim3(kHugeNP * 2), lds, stream, P,
                         (long)nqp, x0, zlim, control_dt, zmp, jerk, status, h->ws_big);
    CCC_HIP_CHECK(hipGetLastError());
    if(from_list) h->last_kernel = stage_name;
    return CCC_OK;
  }
  // K1w: one QP per wavefront, the rows in register tuples (zmp_plan_kernel_w), 32 < N <= 64
  auto go_w = [&](auto kernel, int npc, int waves) -> int {
    const size_t lds = ((size_t)npc * npc + 4 * (size_t)npc + (size_t)waves * ZmpScratch<64>::kSize) * sizeof(double);
    CCC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    int nb = 0;
    if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, waves * 64, lds) != hipSuccess || nb < 1) nb = 2;
    // a few workgroups per resident slot: the hardware dispatcher evens out the data-dependent pivot counts
    const int64_t want = (nqp + waves - 1) / waves;
    const int grid = (int)std::min<int64_t>(want, (int64_t)h->num_cu * nb * 8);
    if(h->env_debug) std::fprintf(stderr, "zmp w kernel: %d columns, lds %zu B -> %d workgroups per CU (grid %d)\n", npc, lds, nb, grid);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(waves * 64), lds, stream, P, h->NP, (long)nqp, x0, zlim, control_dt, zmp, jerk,
                       status);
    h->last_kernel = "zmp_plan_kernel_w";
    CCC_HIP_CHECK(hipGetLastError());
    return CCC_OK;
  };
  if(h->N <= 64 && h->env_kw != 0 && !from_list)
  {
    if(h->N <= 48) return h->env_kw == 2 ? go_w(&zmp_plan_kernel_w<48, 4, 2>, 48, 4) : go_w(&zmp_plan_kernel_w<48, 4, 3>, 48, 4);
    return go_w(&zmp_plan_kernel_w<64, 4, 2>, 64, 4);
  }
  // K2r: the packed tableau in registers (csrc/zmp_k2r.inc), up to 128 rows
  auto go_reg = [&](auto kernel, auto rt) -> int {
    using RT = decltype(rt);
    const size_t lds = RT::lds_bytes();
    const int grid = (int)std::min<int64_t>(grid_qp, (int64_t)1 << 22);
    CCC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    if(h->env_debug)
    {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, RT::NT, lds);
      std::fprintf(stderr, "zmp reg kernel: rows %d threads %d tiles/thread %d lds %zu B -> %d workgroups per CU (grid %d)\n",
                   RT::NB * 4, RT::NT, RT::TPT, lds, nb, grid);
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(RT::NT), lds, stream, P, h->NP, (long)nqp, x0, zlim, control_dt, zmp, jerk,
                       status);
    h->last_kernel = from_list ? stage_name : "zmp_plan_reg_kernel";
    return CCC_OK;
  };
  // measured (round 5, batch 32768, one MI355X; solves/s K2 -> K2r with two | three tiles per thread): N = 40 29.9 -> 27.6 M
  // and 48 17.3 -> 16.9 M (the LDS tableau stays), 56 14.6 -> 15.7 | 13.7 M, 64 8.0 -> 7.9 | 9.6 M, 72 6.47 -> 6.55 | 4.75 M,
  // 80 4.85 -> 5.69 | 4.14 M, 96 3.13 -> 3.19 | 3.12 M, 100 2.27 -> 3.05 | 2.91 M (one tile per thread: 1.63 M), 112 2.01 ->
  // 2.15 | 1.37 M, 128 1.11 -> 0.67 | 1.15 M.  What a pivot costs at these sizes is the per-wavefront selection / staging /
  // ratio-test code around the update (~300 vector and ~280 scalar instructions per pivot and wavefront for 16 FMAs per
  // tile), so fewer, fatter wavefronts per QP win until the registers run out.
  const bool use_reg = h->env_k2 > 0 || (h->env_k2 < 0 && h->N > 48) || (from_list && h->env_k2 != 0);
  if(use_reg && h->N <= 128)
  {
    int rcr;
    const bool three = h->env_k2 == 13 || (h->env_k2 != 12 && ((h->N > 56 && h->N <= 64) || h->N > 112));
#define CCC_ZMP_REG(NR, TPT)                                                   \
  rcr = from_list ? go_reg(&zmp_plan_reg_list_kernel<NR, TPT>, RegTab<NR, TPT>{}) \
                  : go_reg(&zmp_plan_reg_kernel<NR, TPT>, RegTab<NR, TPT>{})
#define CCC_ZMP_REG23(NR) \
  do                      \
  {                       \
    if(three)             \
      CCC_ZMP_REG(NR, 3); \
    else                  \
      CCC_ZMP_REG(NR, 2); \
  } while(0)
    const int Nr = h->N;
    if(Nr <= 40) CCC_ZMP_REG(40, 1);
    else if(Nr <= 48) CCC_ZMP_REG(48, 2);
    else if(Nr <= 56) CCC_ZMP_REG23(56);
    else if(Nr <= 64) CCC_ZMP_REG23(64);
    else if(Nr <= 72) CCC_ZMP_REG23(72);
    else if(Nr <= 80) CCC_ZMP_REG23(80);
    else if(Nr <= 96) CCC_ZMP_REG23(96);
    else if(Nr <= 104) CCC_ZMP_REG23(104);
    else if(Nr <= 112) CCC_ZMP_REG23(112);
    else CCC_ZMP_REG23(128);
#undef CCC_ZMP_REG23
#undef CCC_ZMP_REG
    if(rcr != CCC_OK) return rcr;
    CCC_HIP_CHECK(hipGetLastError());
    return CCC_OK;
  }
  // packed LDS tableau sized to the horizon (rows rounded up to a tile boundary the instantiations cover)
  auto go = [&](auto kernel, auto st) -> int {
    using ST = decltype(st);
    const size_t lds = ((size_t)ST::kDoubles + ST::NB * ST::TS_) * sizeof(double) + sizeof(BlockRed) + sizeof(SelRed);
    // one workgroup per QP: the pivot count varies severalfold between QPs, so the balancing is left to the hardware
    // dispatcher (a QP takes ~100 us, the launch of a workgroup ~1 us)
    const int grid = (int)std::min<int64_t>(grid_qp, (int64_t)1 << 22);
    CCC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    if(h->env_debug)
    {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, ST::NT, lds);
      std::fprintf(stderr, "zmp sym kernel: rows %d threads %d lds %zu B -> %d workgroups per CU (grid %d)\n", ST::NB * ST::TS_, ST::NT,
                   lds, nb, grid);
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(ST::NT), lds, stream, P, h->NP, (long)nqp, x0, zlim, control_dt, zmp, jerk,
                       status);
    return CCC_OK;
  };
  int rc;
#define CCC_ZMP_SYM(NR, TS, TPT)                                                             \
  rc = from_list ? go(&zmp_plan_sym_list_kernel<NR, TS, TPT>, SymTab<NR, TS, TPT>{}) \
                 : go(&zmp_plan_sym_kernel<NR, TS, TPT>, SymTab<NR, TS, TPT>{})
  const int N = h->N;
  if(N <= 40) CCC_ZMP_SYM(40, 4, 2);
  else if(N <= 48) CCC_ZMP_SYM(48, 4, 2);
  else if(N <= 56) CCC_ZMP_SYM(56, 4, 2);
  else if(N <= 64) CCC_ZMP_SYM(64, 4, 2);
  else if(N <= 72) CCC_ZMP_SYM(72, 4, 2);
  else if(N <= 80) CCC_ZMP_SYM(80, 4, 2);
  else if(N <= 96) CCC_ZMP_SYM(96, 4, 2);
  else if(N <= 104) CCC_ZMP_SYM(104, 4, 2);
  else if(N <= 112) CCC_ZMP_SYM(112, 4, 2);
  else if(N <= 128) CCC_ZMP_SYM(128, 4, 3);
  else if(N <= 160) CCC_ZMP_SYM(160, 4, 2);
  else if(N <= 192) CCC_ZMP_SYM(192, 4, 3);
  else CCC_ZMP_SYM(200, 2, 5); // 2 x 2 tiles: 158 KB, the last size that fits the 160 KB of LDS
#undef CCC_ZMP_SYM
  if(rc != CCC_OK) return rc;
  CCC_HIP_CHECK(hipGetLastError());
  if(from_list) h->last_kernel = stage_name;
  return CCC_OK;
}

// N > 32: KS (zmp_stage.inc) on large batches of long horizons, the exact kernels on what it hands over; otherwise the
// exact kernels alone
int launch_block(ccc_zmp * h, int64_t n, const double * x0, const double * zlim, double control_dt, double * zmp,
                 double * jerk, int32_t * status, hipStream_t stream)
{
  const int64_t nqp = 2 * n;

  // KS's time is flat in the batch up to one QP per lane (65536 on 256 CUs: its iteration limit x 0.62 us x N, + 0.5 ms), the
  // exact kernels' is proportional to it: KS from where the two cross (measured, solves/s at 32768 instances KS | exact:
  // N = 48 48.7 | 31.8 M, 64 35.9 | 17.4 M, 72 21.8 | 7.0 M, 100 17.5 | 3.25 M, 128 12.5 | 1.2 M, 200 5.1 | 0.27 M;
  // crossovers at 8192 / 4500 / 2200 / 800 instances for N = 72 / 100 / 128 / 200)
  const int Nh = h->N;
  const int64_t stage_min = h->env_stage_min >= 0 ? h->env_stage_min
                            : Nh <= 48  ? 57344
                            : Nh <= 64  ? 36864
                            : Nh <= 80  ? 14336
                            : Nh <= 112 ? 10240
                            : Nh <= 128 ? 4608
                                        : 2048;
  const bool use_stage = nqp < ((int64_t)1 << 31) && h->N <= 512
                         && (h->env_stage > 0 || (h->env_stage < 0 && h->N >= 40 && nqp >= stage_min));
  if(!use_stage) return launch_exact(h, n, x0, zlim, control_dt, zmp, jerk, status, stream, false);
#ifndef CCC_ZMP_STAGE_CHUNK
#define CCC_ZMP_STAGE_CHUNK 8
#endif
  // stages per checkpoint.  (8 | 10 | 13: N = 100 1.93 | 1.94 | 1.94 ms, N = 64 1.03 | 1.11 | 1.10 ms per 32768 instances -- the
  //  kernel is bound by its fp64 operations, not by the accumulation registers it spills to (46 | 132 at 8 | 13) nor by the
  //  checkpoint traffic.  One wavefront per SIMD: built for two -- CCC_ZMP_STAGE_OCC=2, 256 VGPRs and 188 B of scratch -- 65536
  //  instances take 3.87 ms against 3.50 ms)
  constexpr int kChunk = CCC_ZMP_STAGE_CHUNK;
  const int waves = h->env_stage_waves > 0 ? h->env_stage_waves : 1; // per SIMD at most (the kernel is built for one)
  const int64_t blocks = std::min<int64_t>((nqp + 63) / 64, (int64_t)h->num_cu * 4 * waves);
  const size_t per_block = StageWs<kChunk>::doubles(h->N) * sizeof(double);
  if(blocks > h->ws_stage_blocks || nqp > h->fb_cap)
  {
    CCC_NO_CAPTURE(stream, "ccc_zmp_plan_batch_device");
    if(blocks > h->ws_stage_blocks)
    {
      const int64_t cap = std::min<int64_t>(std::max<int64_t>(blocks, h->ws_stage_blocks * 3 / 2), (int64_t)h->num_cu * 4 * waves);
      if(h->ws_stage) h->retired.push_back(h->ws_stage);
      h->ws_stage = nullptr;
      h->ws_stage_blocks = 0;
      CCC_HIP_CHECK(hipMalloc(&h->ws_stage, (size_t)cap * per_block));
      h->ws_stage_blocks = cap;
    }
    if(nqp > h->fb_cap)
    {
      if(h->fb_list) h->retired.push_back(h->fb_list);
      h->fb_list = nullptr;
      const int64_t cap = std::max<int64_t>(nqp, h->fb_cap * 3 / 2);
      CCC_HIP_CHECK(hipMalloc(&h->fb_list, (size_t)cap * sizeof(int)));
      h->fb_cap = cap;
    }
    if(!h->fb_count) CCC_HIP_CHECK(hipMalloc(&h->fb_count, 64));
  }
  ZmpDev P{h->N, h->dG, h->dA, h->db, h->c2, h->horizon_dt};
  P.fb_list = h->fb_list;
  P.fb_count = h->fb_count;
  // (one 64-byte block: [0] the hand-over count, [4] debug statistics, [8..9] KS's 64-bit ticket counter)
  CCC_HIP_CHECK(hipMemsetAsync(h->fb_count, 0, 64, stream));
  // iteration limits (numpy model, bench workload at N = 100: 11.5 iterations on average, 99 % within 21, 0.3 % cycle) and
  // the penalty of the first iterations, 30 w^6 (w^2 = g / h: jerk^2 against ZMP^2; flat between 10 and 100)
  // (the limit grows with the horizon -- what is left of the creeping is counted in stages -- and what it costs to hand a QP
  //  over grows faster; measured best, 32768 instances: 14 at N = 48 (41.6 M solves/s), 16 at 64 (29.8 M), 20 at 100, 24-26
  //  at 128 (11.3 M), 32 at 200 (5.1 M; 3.5 M with 24), 72 at N = 400 (207 k against 2.8 k for the HBM tableau alone))
  // (beyond four QPs per lane a slow QP holds up its own lane only -- the others draw their next -- and what is handed over is
  //  the exact kernel's at 6.5 M QPs/s: a fifth more sweeps, N = 100 at 262144 instances 10.9 -> 10.0 ms, 131072 5.90 -> 5.78)
  const int base_iters = 8 + (h->N <= 256 ? h->N / 8 : h->N / 6);
  const int iters = h->env_stage_iters > 0 ? h->env_stage_iters
                                           : (nqp >= 4 * (int64_t)h->num_cu * 256 ? base_iters + base_iters / 5 : base_iters);
  const int pen_iters = h->env_stage_pen >= 0 ? h->env_stage_pen : 12;
  const double w2 = -1.0 / h->c2, rho = 30.0 * w2 * w2 * w2;
  // the certificate's bound on the stationarity residual: planned ZMP = ... + c2 cdt u_0 and |du_0| <= |r|_2 <= sqrt(N) r_max
  const double cdt = control_dt < 0 ? h->horizon_dt : control_dt;
  const double cert_abs = 1e-10 / (std::fabs(h->c2) * std::max(cdt, 1e-4) * std::sqrt((double)h->N));
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  int * const stats = h->env_debug && !capturing ? h->fb_count + 4 : nullptr; // (fb_count is a 64-byte allocation)
  unsigned long long * const ticket = reinterpret_cast<unsigned long long *>(h->fb_count + 8);
  if(h->N <= 128)
    hipLaunchKernelGGL((zmp_plan_stage_kernel<kChunk, 2>), dim3((unsigned)blocks), dim3(64), 0, stream, P, (long)nqp, x0, zlim,
                       control_dt, zmp, jerk, status, h->ws_stage, iters, pen_iters, rho, 1e-10, cert_abs, ticket,
                       stats);
  else if(h->N <= 256)
    hipLaunchKernelGGL((zmp_plan_stage_kernel<kChunk, 4>), dim3((unsigned)blocks), dim3(64), 0, stream, P, (long)nqp, x0, zlim,
                       control_dt, zmp, jerk, status, h->ws_stage, iters, pen_iters, rho, 1e-10, cert_abs, ticket,
                       stats);
  else
    hipLaunchKernelGGL((zmp_plan_stage_kernel<kChunk, 8>), dim3((unsigned)blocks), dim3(64), 0, stream, P, (long)nqp, x0, zlim,
                       control_dt, zmp, jerk, status, h->ws_stage, iters, pen_iters, rho, 1e-10, cert_abs, ticket,
                       stats);
  CCC_HIP_CHECK(hipGetLastError());
  h->last_kernel = "zmp_plan_stage_kernel";
  if(stats)
  {
    int got[5] = {-1, 0, 0, 0, 0};
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(got, h->fb_count, sizeof(got), hipMemcpyDeviceToHost);
    std::fprintf(stderr, "zmp stage kernel: %d workgroups, %d iterations at most (%d penalised), %d of %lld QPs handed over, "
                         "sweeps of the slowest QP %d\n", (int)blocks, iters, pen_iters, got[0], (long long)nqp, got[4]);
  }
  return launch_exact(h, n, x0, zlim, control_dt, zmp, jerk, status, stream, true);
}
} // namespace

extern "C" const char * ccc_zmp_last_kernel(const ccc_zmp_t * h)
{
  return h ? h->last_kernel : "none";
}

extern "C" const char * ccc_zmp_last_schedule(const ccc_zmp_t * h)
{
  return h ? h->last_order : "none";
}

extern "C" int ccc_zmp_create(double com_height, double horizon_duration, double horizon_dt, int device,
                              ccc_zmp_t ** out)
{
  if(!out) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_create: out is NULL");
  *out = nullptr;
  if(!(com_height > 0) || !(horizon_duration > 0) || !(horizon_dt > 0))
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_create: com_height, horizon_duration, horizon_dt must be > 0");
  const int N = (int)std::ceil(horizon_duration / horizon_dt); // src/LinearMpcZmp.cpp:13
  if(N > kHugeNP)
    return fail(CCC_ERR_UNSUPPORTED, "ccc_zmp_create: horizon_steps %d > %d is not built into this library", N, kHugeNP);
  int rc = select_device(device);
  if(rc != CCC_OK) return rc;
  CCC_DEVICE_GUARD(device);
  ccc_zmp * h = new ccc_zmp();
  h->device = device;
  h->N = N;
  h->NP = N <= 32 ? 32 : (N <= 64 ? 64 : (N <= 96 ? 96 : (N <= 128 ? 128 : (N <= kBigNP ? kBigNP : kHugeNP))));
  h->com_height = com_height;
  h->horizon_duration = horizon_duration;
  h->horizon_dt = horizon_dt;
  if(const char * qm = std::getenv("CCC_ZMP_QUEUE_MIN")) h->env_queue_min = std::atoll(qm);
  h->env_static = std::getenv("CCC_ZMP_STATIC") != nullptr;
  if(const char * e = std::getenv("CCC_ZMP_HISTORY")) h->env_history = std::atoi(e) != 0;
  if(const char * e = std::getenv("CCC_ZMP_PREDICT")) h->env_predict = std::atoi(e) != 0;
  if(const char * e = std::getenv("CCC_ZMP_PREDICT_MIN")) h->env_predict_min = std::atoll(e);
  h->env_debug = std::getenv("CCC_ZMP_DEBUG") != nullptr;
  if(const char * k2 = std::getenv("CCC_ZMP_K2")) h->env_k2 = std::atoi(k2);
  if(const char * kw = std::getenv("CCC_ZMP_KW")) h->env_kw = std::atoi(kw);
  if(const char * ce = std::getenv("CCC_ZMP_HOST_CHUNK")) h->env_host_chunk = std::atoll(ce);
  if(const char * e = std::getenv("CCC_ZMP_STAGE")) h->env_stage = std::atoi(e);
  if(const char * e = std::getenv("CCC_ZMP_STAGE_MIN")) h->env_stage_min = std::atoll(e);
  if(const char * e = std::getenv("CCC_ZMP_STAGE_ITERS")) h->env_stage_iters = std::atoi(e);
  if(const char * e = std::getenv("CCC_ZMP_STAGE_WAVES")) h->env_stage_waves = std::atoi(e);
  if(const char * e = std::getenv("CCC_ZMP_STAGE_PEN")) h->env_stage_pen = std::atoi(e);
  build_model(h);
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if(e != hipSuccess)
  {
    delete h;
    return fail(CCC_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  }
  h->num_cu = prop.multiProcessorCount;
  rc = upload_model(h);
  if(rc != CCC_OK)
  {
    ccc_zmp_destroy(h);
    return rc;
  }
  *out = h;
  return CCC_OK;
}

extern "C" void ccc_zmp_destroy(ccc_zmp_t * h)
{
  if(!h) return;
  ccc_amd::DeviceGuard ccc_device_guard__(h->device);
  if(h->queue) (void)hipFree(h->queue);
  if(h->hist) (void)hipFree(h->hist);
  if(h->order) (void)hipFree(h->order);
  if(h->order_scratch) (void)hipFree(h->order_scratch);
  if(h->diff) (void)hipFree(h->diff);
  if(h->pred) (void)hipFree(h->pred);
  for(void * q : h->retired) (void)hipFree(q);
  if(h->trust_host) (void)hipHostFree(h->trust_host);
  if(h->dG) (void)hipFree(h->dG);
  if(h->dA) (void)hipFree(h->dA);
  if(h->db) (void)hipFree(h->db);
  if(h->ws_big) (void)hipFree(h->ws_big);
  if(h->ws_stage) (void)hipFree(h->ws_stage);
  if(h->fb_list) (void)hipFree(h->fb_list);
  if(h->fb_count) (void)hipFree(h->fb_count);
  if(h->d_in) (void)hipFree(h->d_in);
  if(h->d_out) (void)hipFree(h->d_out);
  if(h->d_status) (void)hipFree(h->d_status);
  if(h->h_in) (void)hipHostFree(h->h_in);
  if(h->h_out) (void)hipHostFree(h->h_out);
  if(h->h_status) (void)hipHostFree(h->h_status);
  if(h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int ccc_zmp_horizon_steps(const ccc_zmp_t * h)
{
  return h ? h->N : -1;
}

extern "C" int ccc_zmp_get_model(const ccc_zmp_t * h, double * com_height, double * horizon_dt)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_get_model: NULL handle");
  if(com_height) *com_height = h->com_height;
  if(horizon_dt) *horizon_dt = h->horizon_dt;
  return CCC_OK;
}

extern "C" int ccc_zmp_get_seq(const ccc_zmp_t * h, double * A_seq, double * B_seq)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_get_seq: NULL handle");
  if(A_seq) std::memcpy(A_seq, h->A_seq.data(), h->A_seq.size() * sizeof(double));
  if(B_seq) std::memcpy(B_seq, h->B_seq.data(), h->B_seq.size() * sizeof(double));
  return CCC_OK;
}

extern "C" int ccc_zmp_plan_batch_device(ccc_zmp_t * h, int64_t n, const double * x0, const double * zlim,
                                         double control_dt, double * zmp, double * jerk, int32_t * status,
                                         void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch_device: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch_device: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!x0 || !zlim || !zmp) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch_device: NULL x0/zlim/zmp");
  CCC_DEVICE_GUARD(h->device);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if(h->NP == 32) return launch<32, 2>(h, n, x0, zlim, control_dt, zmp, jerk, status, s);
  return launch_block(h, n, x0, zlim, control_dt, zmp, jerk, status, s);
}

// device-side status array for callers that pass none (the kernels always write one)
static int ensure_status(ccc_zmp * h, int64_t n)
{
  if(n <= h->cap_status) return CCC_OK;
  if(h->d_status) (void)hipFree(h->d_status);
  h->d_status = nullptr;
  h->cap_status = 0;
  CCC_HIP_CHECK(hipMalloc(&h->d_status, (size_t)n * 2 * sizeof(int32_t)));
  h->cap_status = n;
  return CCC_OK;
}

// pinned host staging for pageable caller buffers, plus (with_device: the N > 32 route) device-side copies
static int ensure_staging(ccc_zmp * h, int64_t n, bool with_device)
{
  const int N = h->N;
  const size_t in_elems = (size_t)n * (6 + 4 * N), out_elems = (size_t)n * (2 + 2 * N);
  if(!h->stream) CCC_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  if(n > h->cap)
  {
    if(h->h_in) (void)hipHostFree(h->h_in);
    if(h->h_out) (void)hipHostFree(h->h_out);
    if(h->h_status) (void)hipHostFree(h->h_status);
    h->h_in = h->h_out = nullptr;
    h->h_status = nullptr;
    h->cap = 0;
    CCC_HIP_CHECK(hipHostMalloc(&h->h_in, in_elems * sizeof(double), hipHostMallocDefault));
    CCC_HIP_CHECK(hipHostMalloc(&h->h_out, out_elems * sizeof(double), hipHostMallocDefault));
    CCC_HIP_CHECK(hipHostMalloc(&h->h_status, (size_t)n * 2 * sizeof(int32_t), hipHostMallocDefault));
    h->cap = n;
  }
  if(with_device && n > h->cap_dev)
  {
    if(h->d_in) (void)hipFree(h->d_in);
    if(h->d_out) (void)hipFree(h->d_out);
    h->d_in = h->d_out = nullptr;
    h->cap_dev = 0;
    CCC_HIP_CHECK(hipMalloc(&h->d_in, in_elems * sizeof(double)));
    CCC_HIP_CHECK(hipMalloc(&h->d_out, out_elems * sizeof(double)));
    h->cap_dev = n;
    return ensure_status(h, n);
  }
  return CCC_OK;
}

// The device's view of `p` when it is page-locked host memory (hipHostMalloc'ed or hipHostRegister'ed by the caller),
// nullptr for pageable memory.
template<class T>
static T * device_view(T * p)
{
  hipPointerAttribute_t attr;
  if(!p || hipPointerGetAttributes(&attr, p) != hipSuccess || attr.type != hipMemoryTypeHost)
  {
    (void)hipGetLastError(); // plain malloc'ed memory: not an error for us
    return nullptr;
  }
  void * d = nullptr;
  if(hipHostGetDevicePointer(&d, const_cast<void *>(static_cast<const void *>(p)), 0) != hipSuccess)
  {
    (void)hipGetLastError();
    return nullptr;
  }
  return static_cast<T *>(d);
}

// Host-pointer entry.  N <= 32: the kernel reads its 1 072 B per instance and writes its results ONCE, so it works on
// page-locked host memory in place -- no copy, no device staging, the PCIe transfer and the solve overlap inside one
// launch (measured: 1.39 ms per 65 536 instances against 1.27 ms for the bare 67 MB transfer; the copy-then-solve pipeline
// it replaces took 1.6-1.9 ms).  Pinned caller buffers ("inputs resident in pinned host memory") are used
// directly; pageable ones go through the handle's pinned staging, chunk by chunk, the host memcpy of chunk c + 1
// beside the kernel of chunk c.  N > 32: those kernels stream their operands more than once: one H2D, one launch, one D2H.
extern "C" int ccc_zmp_plan_batch(ccc_zmp_t * h, int64_t n, const double * x0, const double * zlim, double control_dt,
                                  double * zmp, double * jerk, int32_t * status)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!x0 || !zlim || !zmp) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch: NULL x0/zlim/zmp");
  CCC_DEVICE_GUARD(h->device);
  const size_t N = (size_t)h->N;
  const size_t nx = (size_t)n * 6, nz = (size_t)n * 2, nj = (size_t)n * 2 * N;
  const double * v_x0 = device_view(x0), * v_zl = device_view(zlim);
  double * v_z = device_view(zmp), * v_j = device_view(jerk);
  int32_t * v_st = device_view(status);
  const bool in_pinned = v_x0 && v_zl, out_pinned = v_z && (!jerk || v_j) && (!status || v_st);
  const bool in_place = h->NP == 32;
  int rc = CCC_OK;
  if(!(in_place && in_pinned && out_pinned))
  {
    rc = ensure_staging(h, n, !in_place);
    if(rc != CCC_OK) return rc;
  }
  else if(!h->stream)
    CCC_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  double * s_x0 = h->h_in, * s_zl = h->h_in + nx, * s_z = h->h_out, * s_j = h->h_out + nz;
  if(!in_place)
  {
    double * d_x0 = h->d_in, * d_zl = h->d_in + nx, * d_z = h->d_out, * d_j = h->d_out + nz;
    if(!in_pinned)
    {
      std::memcpy(s_x0, x0, nx * sizeof(double));
      std::memcpy(s_zl, zlim, nz * 2 * N * sizeof(double));
    }
    CCC_HIP_CHECK(hipMemcpyAsync(d_x0, in_pinned ? x0 : s_x0, nx * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CCC_HIP_CHECK(hipMemcpyAsync(d_zl, in_pinned ? zlim : s_zl, nz * 2 * N * sizeof(double), hipMemcpyHostToDevice,
                                 h->stream));
    rc = ccc_zmp_plan_batch_device(h, n, d_x0, d_zl, control_dt, d_z, jerk ? d_j : nullptr, h->d_status, h->stream);
    if(rc != CCC_OK) return rc;
    CCC_HIP_CHECK(hipMemcpyAsync(out_pinned ? zmp : s_z, d_z, nz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if(jerk)
      CCC_HIP_CHECK(hipMemcpyAsync(out_pinned ? jerk : s_j, d_j, nj * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if(status)
      CCC_HIP_CHECK(hipMemcpyAsync(out_pinned ? status : h->h_status, h->d_status, nz * sizeof(int32_t),
                                   hipMemcpyDeviceToHost, h->stream));
  }
  else
  {
    // results land in the caller's pinned buffers or in the pinned staging
    double * k_in = device_view(h->h_in), * k_out = device_view(h->h_out);
    double * o_z = out_pinned ? v_z : k_out, * o_j = !jerk ? nullptr : (out_pinned ? v_j : k_out + nz);
    int32_t * o_st = out_pinned ? v_st : device_view(h->h_status); // (the kernels always write a status array)
    if(out_pinned && !status)
    {
      rc = ensure_status(h, n);
      if(rc != CCC_OK) return rc;
      o_st = h->d_status;
    }
    int64_t chunk = in_pinned ? n : 8192;
    if(h->env_host_chunk > 0) chunk = std::max<int64_t>(256, std::min<int64_t>(h->env_host_chunk, n));
    h->skip_history = chunk < n;
    h->inputs_in_host = true;
    for(int64_t b = 0; b < n; b += chunk)
    {
      const size_t m = (size_t)std::min<int64_t>(chunk, n - b), o = (size_t)b;
      if(!in_pinned)
      {
        std::memcpy(s_x0 + o * 6, x0 + o * 6, m * 6 * sizeof(double));
        std::memcpy(s_zl + o * 4 * N, zlim + o * 4 * N, m * 4 * N * sizeof(double));
      }
      const double * k_x0 = (in_pinned ? v_x0 : k_in) + o * 6, * k_zl = (in_pinned ? v_zl : k_in + nx) + o * 4 * N;
      rc = ccc_zmp_plan_batch_device(h, (int64_t)m, k_x0, k_zl, control_dt, o_z + o * 2, o_j ? o_j + o * 2 * N : nullptr,
                                     o_st + o * 2, h->stream);
      if(rc != CCC_OK)
      {
        h->skip_history = false;
        h->inputs_in_host = false;
        return rc;
      }
    }
    h->skip_history = false;
    h->inputs_in_host = false;
  }
  CCC_HIP_CHECK(hipStreamSynchronize(h->stream));
  if(!out_pinned)
  {
    std::memcpy(zmp, s_z, nz * sizeof(double));
    if(jerk) std::memcpy(jerk, s_j, nj * sizeof(double));
    if(status) std::memcpy(status, h->h_status, nz * sizeof(int32_t));
  }
  return CCC_OK;
}
