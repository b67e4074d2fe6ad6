// ddp_group.h -- control-limited DDP / iLQR for CCC::DdpCentroidal (S = 9) and CCC::DdpSingleRigidBody (S = 12) with
// ONE PROBLEM INSTANCE PER 16-LANE GROUP: four instances per wavefront, lane l of a group = ridge l (= row l of the
// 16 x 16 input-space matrices, column l of the S x 16 ones, column l of the S x S ones for l < S).
//
// Replaces (reference file:line under /root/reference), as csrc/ddp_core.h does:
//   src/DdpCentroidal.cpp:32-64, :66-83, :85-121, :123-177          problem callbacks (S = 9)
//   src/DdpSingleRigidBody.cpp:26-38, :52-91, :93-112, :114-243     problem callbacks (S = 12)
//   src/DdpCentroidal.cpp:229,233 / src/DdpSingleRigidBody.cpp:299,303   the external nmpc_ddp::DDPSolver::solve
// and implements the specification frozen in oracle/ddp.c (reg_type 1) operation by operation: every product and sum of
// the oracle appears here with the same operands in the same order (terms that are structurally zero are skipped, which
// is exact), no FMA contraction, IEEE sqrt and division, the shared deterministic sin/cos -- the results are
// bit-identical to the oracle's (tests/test_ddp_gpu.py).
//
// Why this shape: the per-step matrices are 9..16 wide, so a wavefront per instance left 3/4 of
// the lanes idle and needed 20 KB of LDS and 852 B of scratch per lane.  Here
//   * everything a lane owns lives in its registers: row l of Quu / of the Cholesky factor, column l of Fu, T2 = Vxx Fu,
//     Qxu, T2' = K'Quu; column l of T1 = Vxx Fx, Qxx and of the new Vxx (l < S); right-hand side l of the gain solve;
//   * what every lane needs from every other one goes through a small per-instance LDS block (5.6 KB at S = 9, 6.9 KB
//     at S = 12) as ONE write + broadcast reads (all 16 lanes read the same address: conflict-free): Vxx, the rows of T2
//     that Fu's zero pattern keeps, the Cholesky factor, K, Qxu, T2' and 16-entry vectors;
//   * ordered reductions (the oracle's sequential sums over ridges) are computed redundantly by every lane from the
//     published terms, so scalars (box-QP value, line-search tests, costs, lambda) are uniform in a group without a
//     second round trip and the four instances of a wavefront diverge only through lane masks;
//   * the triangular solves (box-QP step, gains) run per lane on a broadcast factor -- no cross-lane dependency chain;
//     the gains take one right-hand side per lane.
// The four instances of a wavefront run in lock-step: loops run while ANY instance needs them, an instance that is
// finished (or whose backward pass failed and waits for its retry) keeps its state through masks.
#pragma once

#include <hip/hip_runtime.h>

#include "ddp_core.h"

#if defined(__clang__)
#  pragma clang fp contract(off)
#endif

// Section profiler (development aid, -DCCC_DDPG_PROF): cycles per section accumulated per lane-0 in registers, written
// over the first entries of the instance's `gm` workspace at the end (scripts/ddpg_sections.py reads them back through a
// debug copy).  Sections: 0 rollout, 1 backward total, 2 derivatives+products, 3 box-QP, 4 Cholesky, 5 gains, 6 value
// update; counters: 8 rollouts, 9 backward passes, 10 box-QP iterations, 11 factorisations, 12 line-search steps.
#if defined(CCC_DDPG_PROF)
#  define DPROF_T() ((long long)__builtin_readcyclecounter())
#  define DPROF_ADD(k, t0) prof[k] += (double)(DPROF_T() - (t0))
#  define DPROF_CNT(k) prof[k] += 1.0
#else
#  define DPROF_T() 0LL
#  define DPROF_ADD(k, t0) \
    do                     \
    {                      \
      (void)(t0);          \
    } while(0)
#  define DPROF_CNT(k) \
    do                 \
    {                  \
    } while(0)
#endif

namespace ccc_amd
{
namespace ddpg
{
constexpr int G = 16;          // lanes per instance = ridges per step the group kernel handles
constexpr int LS = 17;         // row stride of the 16 x 16 LDS matrices (odd: per-lane row reads spread over the banks)
using ddp::Params;

struct Batch
{
  const int * phase_dim;       // [n][P]
  const double * phase_vertex; // [n][P][16][3]
  const double * phase_ridge;  // [n][P][16][3]
  const int * step_phase;      // [n][N]
  const double * ref_pos;      // [n][N+1][3]
  const double * ref_ori;      // [n][N+1][3] (SRB)
  const double * inertia;      // [n][9]      (SRB)
  const double * x0;           // [n][S]
  const double * u_init;       // [n][N][16] or nullptr
  double * u_out;              // [n][N][16]
  double * x_out;              // [n][N+1][S] or nullptr
  int * iters;
  int * status;
  double * cost;
  // workspace, per instance: two trajectory buffers (current / line-search candidate, swapped on acceptance),
  // the gains, and the per-step terms of the gradient-norm test
  double * X;  // [n][2][N+1][S]
  double * U;  // [n][2][N][16]
  double * TF; // [n][2][N][3]   total contact force sum_r u_r rho_r of the step (the derivative's crossMat term)
  double * ks; // [n][N][16]
  double * Ks; // [n][N][16][S]
  double * gm; // [n][N]
};

// A stored number: kept as Real, computed with as double.  Real = double: a no-op (the oracle's arithmetic, bit for bit).
// Real = float (ccc_ddp_config_t::precision = 32, BASELINE configs[4]): the matrices of the backward pass -- Vxx, T2, Quu,
// the Cholesky factor, Qxu, K and the box-QP vectors -- are STORED in single precision (half the LDS and half the
// registers of the arrays), every product and sum is still formed in double.  A straight fp32 solver does not work on
// this problem: the reference's constants (box-QP gradient threshold 1e-8, cost_update_thre 1e-7, force weight 1e-6
// against curvatures of 1e-2) sit below single-precision resolution and the backward pass fails on every instance.
template<typename Real>
struct Stored
{
  Real v;
  __device__ __forceinline__ Stored() = default;
  __device__ __forceinline__ Stored & operator=(double x)
  {
    v = static_cast<Real>(x);
    return *this;
  }
  __device__ __forceinline__ operator double() const
  {
    return static_cast<double>(v);
  }
};

template<int S, typename Real>
struct Lds
{
  using St = Stored<Real>;
  static constexpr int KS = S + 1; // row stride of K (odd at S = 12)
  static constexpr int kA1Bytes = (G * LS * (int)sizeof(Real)) > (11 * G * 8) ? (G * LS * (int)sizeof(Real)) : (11 * G * 8);
  St Vxx[S * S];                   // [a][b]; parks the columns of Qxx during the box-QP; transposes T
  alignas(16) unsigned char A1raw[kA1Bytes]; // rows of T2 -> Cholesky factor -> Quu' -> T2'; the rollout's products (double)
  St Qxu[S * LS];                  // [a][r]
  St K[G * KS];                    // [r][a]
  St v[8][G];                      // published vectors (see the SL_* slots)
  __device__ __forceinline__ St * A1()
  {
    return reinterpret_cast<St *>(A1raw);
  }
  __device__ __forceinline__ double * roll()
  {
    return reinterpret_cast<double *>(A1raw);
  }
};

// vector slots of Lds::v
enum
{
  SL_T = 0, // terms of an ordered sum / scratch
  SL_U,     // nominal inputs u_i of the step (box limits lo = flo - u, hi = fhi - u)
  SL_X,     // box-QP iterate
  SL_G,     // box-QP gradient
  SL_R,     // right-hand side of the Newton step
  SL_D,     // diagonal of the matrix being factorised; later t4 = Quu k
  SL_RD,    // reciprocal diagonal of the Cholesky factor
  SL_VX     // Vx (S entries); Qu during the value update
};

// structural non-zeros of the discrete-time Fx = I + dt dF/dx (oracle/ddp_models.c mdl_state_eq_deriv)
template<int S>
__device__ constexpr bool fx_nz(int k, int b)
{
  if(k == b) return true;
  if(S == 9) return (k < 3 && b == k + 3) || (k >= 6 && b < 3 && (k - 6) != b);
  // S == 12
  if(k < 3) return b == k + 6;
  if(k < 6) return b == 3 || (b == 4 && k != 4) || b == 9 || b == 10 || (b == 11 && k == 3);
  if(k < 9) return false;
  return b < 3 || b >= 9;
}

template<int S, typename Real>
struct Group
{
  using St = Stored<Real>;
  static constexpr int R0 = (S == 9) ? 3 : 6; // first non-zero row of Fu (rows R0 .. R0+5)
  static constexpr int KS = Lds<S, Real>::KS;

  const Params & P;
  Lds<S, Real> & L;
  const int l;     // lane in the group
  const bool live; // the instance exists (dead groups of the last workgroup shadow the last instance, stores masked)
  const unsigned shift; // bit position of the group in a wavefront ballot

  // instance data
  const int * phase_dim;
  const double * phase_vertex;
  const double * phase_ridge;
  const int * step_phase;
  const double * ref_pos;
  const double * ref_ori;
  const double * inertia;
  const double * x0;
  const double * u_init;
  double *X, *U, *TF, *ks, *Ks, *gm;
  int N;

  // per-lane constants
  double wrun_own, wterm_own;
#if defined(CCC_DDPG_PROF)
  double prof[16];
#endif

  // Lanes of a group talk through LDS.  One wavefront per workgroup, and the LDS serves the requests of a wavefront in
  // order, so a write is visible to every later read of the same wavefront: a compiler-level barrier is all it takes
  // (no s_barrier, no s_waitcnt on the outstanding global loads).
  __device__ __forceinline__ static void sync()
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // lanes talking through GLOBAL memory (trajectories written by the owner lanes, read by all): a real barrier
  __device__ __forceinline__ static void gsync()
  {
    __syncthreads();
  }
  __device__ __forceinline__ unsigned gballot(bool p) const
  {
    return static_cast<unsigned>((__ballot(p) >> shift) & 0xffffull);
  }
  __device__ __forceinline__ static bool wany(bool p)
  {
    return __ballot(p) != 0ull;
  }

  __device__ __forceinline__ double ref_entry(int step, int a) const
  {
    if(a < 3) return ref_pos[static_cast<long>(step) * 3 + a];
    if(S == 12 && a < 6) return ref_ori[static_cast<long>(step) * 3 + a - 3];
    return 0.0;
  }
  __device__ __forceinline__ double * xbuf(int cb) const
  {
    return X + static_cast<long>(cb) * (N + 1) * S;
  }
  __device__ __forceinline__ double * ubuf(int cb) const
  {
    return U + static_cast<long>(cb) * N * G;
  }
  __device__ __forceinline__ double * tfbuf(int cb) const
  {
    return TF + static_cast<long>(cb) * N * 3;
  }

  // value of entry l of a vector every lane holds in full (v[k] uniform in the group)
  __device__ __forceinline__ double own(const St (&v)[G]) const
  {
    double r = v[0];
#pragma unroll
    for(int k = 1; k < G; k++) r = (l == k) ? v[k] : r;
    return r;
  }
  // sum_{k < m} t_k in increasing k from 0 (the oracle's sequential sums): the terms are published, every lane adds
  __device__ __forceinline__ double ordered_sum(double t, int m)
  {
    L.v[SL_T][l] = t;
    sync();
    double s = 0.0;
#pragma unroll
    for(int k = 0; k < G; k++)
      s += L.v[SL_T][k]; // entries >= m are zero
    sync();
    return s;
  }

  // ------------------------------------------------------------------------------------------ contact phase
  struct Contact
  {
    int ph = -1, m = 0;
    double V[3] = {0, 0, 0}, R[3] = {0, 0, 0};
  };
  __device__ __forceinline__ void load_contact(int step, Contact & c) const
  {
    int ph = step_phase[step]; // (clamped to the tables, as in csrc/ddp_core.h)
    ph = ph < 0 ? 0 : (ph >= P.P ? P.P - 1 : ph);
    if(ph != c.ph)
    {
      c.ph = ph;
      const int d = phase_dim[ph];
      c.m = d < 0 ? 0 : (d > G ? G : d);
      const long o = (static_cast<long>(ph) * G + l) * 3;
#pragma unroll
      for(int a = 0; a < 3; a++)
      {
        c.V[a] = phase_vertex[o + a];
        c.R[a] = phase_ridge[o + a];
      }
    }
  }

  // ------------------------------------------------------------------------------------------ rollout
  // alpha < 0: rollout of the initial inputs into buffer `cb`; otherwise the line-search candidate of `cb` into cb ^ 1.
  // Returns the trajectory cost (uniform in the group).  run: the group takes part (others execute masked, no stores).
  __device__ __forceinline__ double rollout(double alpha, int cb, bool run)
  {
    const long long t_roll = DPROF_T();
    DPROF_CNT(8);
    const bool initial = alpha < 0;
    const double * xs = xbuf(cb);
    const double * us = ubuf(cb);
    double * xo = initial ? xbuf(cb) : xbuf(cb ^ 1);
    double * uo = initial ? ubuf(cb) : ubuf(cb ^ 1);
    double * tfo = initial ? tfbuf(cb) : tfbuf(cb ^ 1);
    const bool wr = run && live;
    double x[S];
#pragma unroll
    for(int a = 0; a < S; a++) x[a] = initial ? x0[a] : xs[a];
    if(wr && l < S) xo[l] = initial ? x0[l] : xs[l];
    double cost = 0.0;
    Contact c;
    for(int i = 0; i < N; i++)
    {
      load_contact(i, c);
      const int m = c.m;
      const bool in = l < m;
      double u = 0.0;
      if(initial)
        u = (in && u_init) ? u_init[static_cast<long>(i) * G + l] : 0.0;
      else
      {
        double s = us[static_cast<long>(i) * G + l] + alpha * ks[static_cast<long>(i) * G + l];
        const double * Kr = Ks + (static_cast<long>(i) * G + l) * S;
        const double * xi = xs + static_cast<long>(i) * S;
#pragma unroll
        for(int a = 0; a < S; a++) s += Kr[a] * (x[a] - xi[a]);
        s = fmin(fmax(s, P.flo), P.fhi);
        u = in ? s : 0.0;
      }
      if(wr) uo[static_cast<long>(i) * G + l] = u;
      // per-ridge products; rows of A1: 0-2 u rho (total force), 3-5 u (p - c) x rho, 6 u^2, (SRB) 7-9 u rho / mass
      const double d[3] = {c.V[0] - x[0], c.V[1] - x[1], c.V[2] - x[2]};
      double cr[3];
      ddp::cross3(d, c.R, cr);
#pragma unroll
      for(int k = 0; k < 3; k++)
      {
        const double p = u * c.R[k];
        L.roll()[k * G + l] = p;
        L.roll()[(3 + k) * G + l] = u * cr[k];
        if(S == 12) L.roll()[(7 + k) * G + l] = p / P.mass;
      }
      L.roll()[6 * G + l] = u * u;
      sync();
      // ordered sums over the ridges, one row per lane (rows 0..9), then published
      {
        constexpr int NROW = (S == 9) ? 7 : 10;
        double acc = 0.0;
        if(S == 9)
        {
          if(l == 2) acc = -1 * P.mass * ddp::kGravity; // xd[5] starts from -m g (src/DdpCentroidal.cpp:43)
        }
        else
        {
          if(l == 9) acc = -1 * ddp::kGravity; // xd[8] starts from -g (src/DdpSingleRigidBody.cpp:74)
          if(l >= 3 && l < 6)
          {
            // wd starts from -w x (I w)
            const double * w = x + 9;
            double Iw[3], cw[3];
#pragma unroll
            for(int b = 0; b < 3; b++) Iw[b] = inertia[b * 3] * w[0] + inertia[b * 3 + 1] * w[1] + inertia[b * 3 + 2] * w[2];
            ddp::cross3(w, Iw, cw);
            acc = -1 * (l == 3 ? cw[0] : (l == 4 ? cw[1] : cw[2]));
          }
        }
        // Cen: lanes 0-2 dynamics force rows (0,1 double as tf_x, tf_y), 3-5 moment rows, 6 u^2, 7 tf_z (row 2 from 0)
        // SRB: lanes 0-2 tf rows, 3-5 moment rows (from -w x I w), 6 u^2, 7-9 force / mass rows
        const int row = (S == 9 && l == 7) ? 2 : (l < NROW ? l : 0);
        const double * rowp = L.roll() + row * G;
#pragma unroll
        for(int r = 0; r < G; r++)
          acc += rowp[r]; // absent ridges carry u = 0
        L.roll()[10 * G + l] = acc;
      }
      sync();
      double sums[10];
#pragma unroll
      for(int k = 0; k < ((S == 9) ? 8 : 10); k++) sums[k] = L.roll()[10 * G + k];
      sync();
      // running cost of (x_i, u_i) (src/DdpCentroidal.cpp:66-74): sequential over the state entries
      {
        double cs = 0.0;
#pragma unroll
        for(int a = 0; a < S; a++)
        {
          const double e = x[a] - ref_entry(i, a);
          cs += 0.5 * P.w_run[a] * e * e;
        }
        cost += cs + 0.5 * P.w_force * sums[6];
      }
      if(wr && l < 3) tfo[static_cast<long>(i) * 3 + l] = (S == 9 && l == 2) ? sums[7] : sums[l];
      // x_{i+1} = x_i + dt xdot
      double xd[S];
      if(S == 9)
      {
#pragma unroll
        for(int a = 0; a < 3; a++) xd[a] = x[3 + a] / P.mass;
#pragma unroll
        for(int a = 0; a < 6; a++) xd[3 + a] = sums[a];
      }
      else
      {
        double ca, sa, cb_, sb;
        ddp::det_sincos(x[3], &sa, &ca);
        ddp::det_sincos(x[4], &sb, &cb_);
        const double Km[9] = {(ca * sb) / cb_, (sb * sa) / cb_, 1.0, -1 * sa, ca, 0.0, ca / cb_, sa / cb_, 0.0};
        const double * w = x + 9;
#pragma unroll
        for(int a = 0; a < 3; a++)
        {
          xd[a] = x[6 + a];
          xd[3 + a] = Km[a * 3] * w[0] + Km[a * 3 + 1] * w[1] + Km[a * 3 + 2] * w[2];
          xd[6 + a] = sums[7 + a];
        }
        const double wd[3] = {sums[3], sums[4], sums[5]};
        double In[9], sol[3];
#pragma unroll
        for(int e = 0; e < 9; e++) In[e] = inertia[e];
        ddp::llt3_solve(In, wd, sol);
#pragma unroll
        for(int a = 0; a < 3; a++) xd[9 + a] = sol[a];
      }
#pragma unroll
      for(int a = 0; a < S; a++) x[a] = x[a] + P.dt * xd[a];
      if(wr && l < S)
      {
        double xv = x[0];
#pragma unroll
        for(int a = 1; a < S; a++) xv = (l == a) ? x[a] : xv;
        xo[static_cast<long>(i + 1) * S + l] = xv;
      }
    }
    {
      double cs = 0.0;
#pragma unroll
      for(int a = 0; a < S; a++)
      {
        const double e = x[a] - ref_entry(N, a);
        cs += 0.5 * P.w_term[a] * e * e;
      }
      cost += cs;
    }
    sync();
    DPROF_ADD(0, t_roll);
    return cost;
  }

  // ------------------------------------------------------------------------------------------ triangular solves
  // t <- (L L')^-1 t with the factor of the group in LDS (strict lower triangle in A1, reciprocal diagonal in SL_RD).
  // Every lane works on its own right-hand side (all equal in the box-QP, one column of Qxu' per lane in the gains).
  __device__ __forceinline__ void solve(St (&t)[G]) const
  {
#pragma unroll
    for(int a = 0; a < G; a++)
    {
      double s = t[a];
#pragma unroll
      for(int k = 0; k < a; k++) s -= L.A1()[a * LS + k] * t[k];
      t[a] = s * L.v[SL_RD][a];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for(int a = G - 1; a >= 0; a--)
    {
      double s = t[a];
#pragma unroll
      for(int k = G - 1; k > a; k--) s -= L.A1()[k * LS + a] * t[k];
      t[a] = s * L.v[SL_RD][a];
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // Cholesky of H~ (H with clamped / absent rows and columns replaced by identity: the factor of H_ff embedded) into
  // the LDS factor of the groups with `need`; returns false where a pivot is not positive (oracle: result -1).
  // Left-looking, lane = row: at pivot a every lane reads row a (written by lane a at the earlier pivots), forms the
  // pivot redundantly and its own entry of column a -- the oracle's sums in the oracle's order.
  __device__ __forceinline__ bool cholesky(const St (&Hreg)[G], double hdiag, unsigned clmask, int m, bool need)
  {
    const long long t_ch = DPROF_T();
    DPROF_CNT(11);
    const bool mine_free = l < m && !((clmask >> l) & 1u);
    L.v[SL_D][l] = mine_free ? hdiag : 1.0;
    sync();
    St Lrow[G];
    bool ok = true;
#pragma unroll
    for(int a = 0; a < G; a++)
    {
      const bool a_free = a < m && !((clmask >> a) & 1u);
      St La[G];
#pragma unroll
      for(int k = 0; k < a; k++) La[k] = L.A1()[a * LS + k];
      double d = L.v[SL_D][a];
#pragma unroll
      for(int k = 0; k < a; k++) d -= La[k] * La[k];
      ok = ok && (d > 0.0);
      const double sq = sqrt(d);
      const double rda = 1.0 / sq;
      double v = (mine_free && a_free) ? Hreg[a] : 0.0;
#pragma unroll
      for(int k = 0; k < a; k++) v -= Lrow[k] * La[k];
      v = v * rda;
      Lrow[a] = (l > a) ? v : 0.0;
      if(need && l > a) L.A1()[l * LS + a] = Lrow[a];
      if(need && l == a) L.v[SL_RD][a] = rda;
      sync();
    }
    DPROF_ADD(4, t_ch);
    return ok;
  }

  // ------------------------------------------------------------------------------------------ box-QP
  // min 1/2 k'Hk + g'k, lo <= k <= hi (oracle_box_qp): H row l in Hreg (regularised), g_l = gl, limits from the nominal
  // inputs in SL_U, warm start kw_l.  Out: the solution in SL_X (entries >= m zero), the clamped set, the boxQP.m
  // result.  The iterate, the gradient and the right-hand side live in LDS slots; a lane holds its own entries.
  __device__ __forceinline__ int box_qp(const St (&Hreg)[G], double hdiag, double gl, double kw, int m, bool run,
                                        unsigned & clmask_out)
  {
    const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
    const int max_iter = 500; // nmpc_ddp BoxQP::Configuration::max_iter
    const bool in = l < m;
    const unsigned inmask = m >= G ? 0xffffu : ((1u << m) - 1u);
    const double ul = L.v[SL_U][l];
    const double lo = P.flo - ul, hi = P.fhi - ul;
    double x = in ? fmin(fmax(kw, lo), hi) : 0.0;
    L.v[SL_X][l] = x;
    sync();
    x = L.v[SL_X][l];
    auto row_dot = [&](double s0, int slot) {
      double s = s0;
#pragma unroll
      for(int j = 0; j < G; j++)
        s += Hreg[j] * L.v[slot][j]; // row entries >= m are zero
      return s;
    };
    double value;
    {
      const double s = row_dot(0.0, SL_X);
      value = ordered_sum(x * gl + 0.5 * x * s, m);
    }
    double oldvalue = 0.0;
    bool cl = false;
    unsigned clmask = 0;
    int result = 0;
    bool fin = !run || m == 0;
    if(m == 0) result = 1; // nothing to solve (the oracle does not call the box-QP for a step without contact)
    int iter = 1;
    for(; iter <= max_iter; iter++)
    {
      if(!wany(!fin)) break;
      DPROF_CNT(10);
      // ---- top of the oracle's loop (state changes as selects: the four instances differ only through masks)
      fin = fin || (result != 0);
      {
        const bool small = !fin && iter > 1 && (oldvalue - value) < min_rel_improve * fabs(oldvalue);
        result = small ? 4 : result;
        fin = fin || small;
      }
      oldvalue = fin ? oldvalue : value;
      const double grad = row_dot(gl, SL_X);
      const bool ncl = in && ((x == lo && grad > 0) || (x == hi && grad < 0));
      const bool changed_l = !fin && (ncl != cl);
      cl = fin ? cl : ncl;
      const bool changed = (iter == 1) || (gballot(changed_l) != 0u);
      {
        const unsigned nm = gballot(cl) & inmask;
        clmask = fin ? clmask : nm;
      }
      {
        const bool all = !fin && clmask == inmask;
        result = all ? 6 : result;
        fin = fin || all;
      }
      const bool need = !fin && changed;
      if(wany(need))
      {
        const bool ok = cholesky(Hreg, hdiag, clmask, m, need);
        const bool bad = need && !ok;
        result = bad ? -1 : result;
        fin = fin || bad;
      }
      L.v[SL_G][l] = grad;
      sync();
      {
        double gn = 0.0;
#pragma unroll
        for(int k = 0; k < G; k++)
        {
          const double gk = L.v[SL_G][k];
          gn += ((clmask >> k) & 1u) ? 0.0 : gk * gk;
        }
        gn = sqrt(gn);
        const bool tiny = !fin && gn < min_grad;
        result = tiny ? 5 : result;
        fin = fin || tiny;
      }
      if(!wany(!fin)) break;
      // grad_clamped = g + H (x .* clamped) on the free rows, then the Newton step on the free set
      double gc = gl;
#pragma unroll
      for(int j = 0; j < G; j++)
        gc += ((clmask >> j) & 1u) ? Hreg[j] * L.v[SL_X][j] : 0.0;
      L.v[SL_R][l] = (in && !cl) ? gc : 0.0;
      sync();
      St srch[G];
#pragma unroll
      for(int k = 0; k < G; k++) srch[k] = L.v[SL_R][k];
      solve(srch);
#pragma unroll
      for(int k = 0; k < G; k++) srch[k] = (k < m && !((clmask >> k) & 1u)) ? -srch[k] - L.v[SL_X][k] : 0.0;
      double sdotg = 0.0;
#pragma unroll
      for(int k = 0; k < G; k++)
        sdotg += srch[k] * L.v[SL_G][k];
      {
        // no descent direction: in double the oracle's failure (result stays 0).  With single-precision storage the Newton
        // step of an iterate that is optimal TO THAT RESOLUTION is rounding noise (the solve loses 6e-8 |x| against a true
        // step of |grad| / |H|), so its sign against the gradient is a coin toss: there it means "converged" (result 5).
        const bool nodesc = !fin && sdotg >= 0;
        if(sizeof(Real) < 8) result = nodesc ? 5 : result;
        fin = fin || nodesc;
      }
      // ---- Armijo line search along the projected step
      const double srch_own = own(srch);
      double step = 1.0, step_used = 1.0, vc = value;
      bool ls = !fin;
      while(wany(ls))
      {
        DPROF_CNT(12);
        St cand[G];
#pragma unroll
        for(int k = 0; k < G; k++)
        {
          const double uk = L.v[SL_U][k];
          cand[k] = (k < m) ? fmin(fmax(L.v[SL_X][k] + step * srch[k], P.flo - uk), P.fhi - uk) : 0.0;
        }
        const double c_own = own(cand);
        double s = 0.0;
#pragma unroll
        for(int j = 0; j < G; j++) s += Hreg[j] * cand[j];
        const double v = ordered_sum(c_own * gl + 0.5 * c_own * s, m);
        vc = ls ? v : vc;
        step_used = ls ? step : step_used;
        const bool stop = !((v - oldvalue) / (step * sdotg) < armijo);
        const double nstep = step * step_dec;
        const bool under = nstep < min_step;
        result = (ls && !stop && under) ? 2 : result;
        step = (ls && !stop) ? nstep : step;
        ls = ls && !stop && !under;
      }
      {
        const double xn = in ? fmin(fmax(x + step_used * srch_own, lo), hi) : 0.0; // the accepted candidate (same expression)
        x = fin ? x : xn;
        value = fin ? value : vc;
      }
      L.v[SL_X][l] = x;
      sync();
      x = L.v[SL_X][l]; // the stored iterate is THE iterate (a no-op in double)
    }
    if(!fin && iter > max_iter && result == 0) result = 1;
    clmask_out = clmask;
    return result;
  }

  // ------------------------------------------------------------------------------------------ backward pass
  // One sweep over the horizon for the groups with `act`; returns false where a box-QP failed (oracle: retry with a
  // larger lambda).  dV0 / dV1: the expected-reduction terms; the per-step terms of the gradient norm go to gm.
  __device__ __forceinline__ bool backward_pass(int cb, double lambda, bool act, double & dV0_out, double & dV1_out)
  {
    const long long t_bw = DPROF_T();
    DPROF_CNT(9);
    const double * xs = xbuf(cb);
    const double * us = ubuf(cb);
    const double * tfs = tfbuf(cb);
    bool ok = true;
    double dV0 = 0.0, dV1 = 0.0;
    // terminal value (src/DdpCentroidal.cpp:156-177 at x_N): Vx in SL_VX, Vxx in LDS
    if(l < S)
    {
      L.v[SL_VX][l] = wterm_own * (xs[static_cast<long>(N) * S + l] - ref_entry(N, l));
#pragma unroll
      for(int a = 0; a < S; a++) L.Vxx[a * S + l] = (a == l) ? wterm_own : 0.0;
    }
    sync();
    Contact c;
    double kprev = 0.0;
    int m_next = -1;
    for(int i = N - 1; i >= 0; i--)
    {
      const long long t_s0 = DPROF_T();
      load_contact(i, c);
      const int m = c.m;
      const bool in = l < m;
      const bool run = act && ok;
      const double u = in ? us[static_cast<long>(i) * G + l] : 0.0;
      L.v[SL_U][l] = u;
      // ---- derivatives at (x_i, u_i): column l of Fu (rows R0..R0+5), the sparse Fx (uniform in the group)
      double fu[6];
      double FX[S][S];
      double x_own, Qx_own, Qu;
      St T1c[S], T2c[S];
      {
        double x[S];
#pragma unroll
        for(int a = 0; a < S; a++) x[a] = xs[static_cast<long>(i) * S + a];
        double tf[3];
#pragma unroll
        for(int a = 0; a < 3; a++) tf[a] = tfs[static_cast<long>(i) * 3 + a];
        const double d[3] = {c.V[0] - x[0], c.V[1] - x[1], c.V[2] - x[2]};
        double cr[3];
        ddp::cross3(d, c.R, cr);
        if(S == 9)
        {
#pragma unroll
          for(int a = 0; a < 3; a++)
          {
            fu[a] = in ? c.R[a] * P.dt : 0.0;
            fu[3 + a] = in ? cr[a] * P.dt : 0.0;
          }
          const double cm = (1 / P.mass) * P.dt;
#pragma unroll
          for(int a = 0; a < S; a++) FX[a][a] = 1.0;
          FX[0][3] = cm;
          FX[1][4] = cm;
          FX[2][5] = cm;
          FX[6][1] = (-tf[2]) * P.dt;
          FX[6][2] = tf[1] * P.dt;
          FX[7][0] = tf[2] * P.dt;
          FX[7][2] = (-tf[0]) * P.dt;
          FX[8][0] = (-tf[1]) * P.dt;
          FX[8][1] = tf[0] * P.dt;
        }
        else
        {
          double In[9];
#pragma unroll
          for(int e = 0; e < 9; e++) In[e] = inertia[e];
          double sol[3];
          ddp::llt3_solve(In, cr, sol);
#pragma unroll
          for(int a = 0; a < 3; a++)
          {
            fu[a] = in ? (c.R[a] / P.mass) * P.dt : 0.0;
            fu[3 + a] = in ? sol[a] * P.dt : 0.0;
          }
#pragma unroll
          for(int a = 0; a < S; a++) FX[a][a] = 1.0;
#pragma unroll
          for(int a = 0; a < 3; a++) FX[a][6 + a] = 1.0 * P.dt;
          const double w1 = x[9], w2 = x[10], w3 = x[11];
          double ca, sa, cb_, sb;
          ddp::det_sincos(x[3], &sa, &ca);
          ddp::det_sincos(x[4], &sb, &cb_);
          const double cb2 = cb_ * cb_, sb2 = sb * sb;
          const double Km[9] = {(ca * sb) / cb_, (sb * sa) / cb_, 1.0, -1 * sa, ca, 0.0, ca / cb_, sa / cb_, 0.0};
          FX[3][9] = Km[0] * P.dt;
          FX[3][10] = Km[1] * P.dt;
          FX[3][11] = Km[2] * P.dt;
          FX[4][9] = Km[3] * P.dt;
          FX[4][10] = Km[4] * P.dt;
          FX[5][9] = Km[6] * P.dt;
          FX[5][10] = Km[7] * P.dt;
          FX[3][3] = (-w1 * sa * sb / cb_ + w2 * sb * ca / cb_) * P.dt + 1.0;
          FX[4][3] = (-w1 * ca - w2 * sa) * P.dt;
          FX[5][3] = (-w1 * sa / cb_ + w2 * ca / cb_) * P.dt;
          FX[3][4] = (w1 * sb2 * ca / cb2 + w1 * ca + w2 * sa * sb2 / cb2 + w2 * sa) * P.dt;
          FX[5][4] = (w1 * sb * ca / cb2 + w2 * sa * sb / cb2) * P.dt;
          const double I11 = In[0], I12 = In[1], I13 = In[2], I22 = In[4], I23 = In[5], I33 = In[8];
          const double D[9] = {I12 * w3 - I13 * w2,
                               -I13 * w1 + I22 * w3 - 2 * I23 * w2 - I33 * w3,
                               I12 * w1 + I22 * w2 + 2 * I23 * w3 - I33 * w2,
                               -I11 * w3 + 2 * I13 * w1 + I23 * w2 + I33 * w3,
                               -I12 * w3 + I23 * w1,
                               -I11 * w1 - I12 * w2 - 2 * I13 * w3 + I33 * w1,
                               I11 * w2 - 2 * I12 * w1 - I22 * w2 - I23 * w3,
                               I11 * w1 + 2 * I12 * w2 + I13 * w3 - I22 * w1,
                               I13 * w2 - I23 * w1};
          const double CM[9] = {0, -tf[2], tf[1], tf[2], 0, -tf[0], -tf[1], tf[0], 0};
#pragma unroll
          for(int b = 0; b < 3; b++)
          {
            const double colD[3] = {D[b], D[3 + b], D[6 + b]}, colC[3] = {CM[b], CM[3 + b], CM[6 + b]};
            double sD[3], sC[3];
            ddp::llt3_solve(In, colD, sD);
            ddp::llt3_solve(In, colC, sC);
#pragma unroll
            for(int a = 0; a < 3; a++)
            {
              FX[9 + a][9 + b] = (a == b) ? sD[a] * P.dt + 1.0 : sD[a] * P.dt;
              FX[9 + a][b] = sC[a] * P.dt;
            }
          }
        }
        x_own = x[0];
#pragma unroll
        for(int a = 1; a < S; a++) x_own = (l == a) ? x[a] : x_own;
      }
      // column l of Fx (dense, zeros included) for the lanes that own a state entry
      St fxcol[S];
#pragma unroll
      for(int k = 0; k < S; k++)
      {
        double v = 0.0;
#pragma unroll
        for(int b = 0; b < S; b++)
          if(fx_nz<S>(k, b)) v = (l == b) ? FX[k][b] : v;
        fxcol[k] = v;
      }
      // ---- Qx (lane a), Qu (lane r)
      Qx_own = wrun_own * (x_own - ref_entry(i, l < S ? l : 0));
#pragma unroll
      for(int b = 0; b < S; b++) Qx_own += fxcol[b] * L.v[SL_VX][b];
      Qu = in ? P.w_force * u : 0.0;
#pragma unroll
      for(int j = 0; j < 6; j++) Qu += fu[j] * L.v[SL_VX][R0 + j];
      // ---- T1 = Vxx Fx (column l), T2 = Vxx Fu (column l)
#pragma unroll
      for(int a = 0; a < S; a++)
      {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for(int k = 0; k < S; k++)
        {
          const double v = L.Vxx[a * S + k];
          s1 += v * fxcol[k];
          if(k >= R0 && k < R0 + 6) s2 += v * fu[k - R0];
        }
        T1c[a] = s1;
        T2c[a] = s2;
        __builtin_amdgcn_sched_barrier(0);
      }
      sync(); // Vxx is consumed: its block now parks the columns of Qxx
      // ---- Qxx column l (parked in LDS), Qxu column l (to LDS: the gains read its rows)
#pragma unroll
      for(int a = 0; a < S; a++)
      {
        double s = (a == l) ? wrun_own : 0.0;
        double sx = 0.0;
#pragma unroll
        for(int k = 0; k < S; k++)
          if(fx_nz<S>(k, a))
          {
            s += FX[k][a] * T1c[k];
            sx += FX[k][a] * T2c[k];
          }
        if(l < S) L.Vxx[a * S + l] = s;
        L.Qxu[a * LS + l] = sx;
      }
      // ---- rows R0..R0+5 of T2 published, row l of Quu
#pragma unroll
      for(int j = 0; j < 6; j++) L.A1()[j * G + l] = T2c[R0 + j];
      sync();
      St Hreg[G];
      double quu_own = 0.0;
#pragma unroll
      for(int q = 0; q < G; q++)
      {
        double s = (l == q) ? P.w_force : 0.0;
#pragma unroll
        for(int j = 0; j < 6; j++) s += fu[j] * L.A1()[j * G + q];
        s = (in && q < m) ? s : 0.0;
        quu_own = (l == q) ? s : quu_own;
        Hreg[q] = (l == q) ? s + lambda : s; // reg_type 1: Quu_F = Quu + lambda I
        if((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      const double hdiag = quu_own + lambda;
      sync(); // the T2 rows are consumed: A1 becomes the Cholesky factor
      DPROF_ADD(2, t_s0);
      const long long t_s1 = DPROF_T();
      // ---- box-QP for the feed-forward term k (result in SL_X)
      unsigned clmask = 0;
      const bool warm = (i + 1 < N) && (m_next == m);
      int rc = 1;
      if(wany(run && m > 0))
        rc = box_qp(Hreg, hdiag, Qu, warm ? kprev : 0.0, m, run && m > 0, clmask);
      else
      {
        L.v[SL_X][l] = 0.0;
        sync();
      }
      if(m == 0) clmask = 0;
      if(run && m > 0 && rc < 1) ok = false;
      const double k_own = L.v[SL_X][l];
      DPROF_ADD(3, t_s1);
      const long long t_s2 = DPROF_T();
      // ---- gains: K_f = -Quu_F,ff^-1 Qxu_f', one right-hand side (state entry l) per lane
      St QxuRow[G], Kc[G];
      {
        const int a = l < S ? l : 0;
#pragma unroll
        for(int r = 0; r < G; r++) QxuRow[r] = L.Qxu[a * LS + r];
#pragma unroll
        for(int r = 0; r < G; r++) Kc[r] = (r < m && !((clmask >> r) & 1u)) ? QxuRow[r] : 0.0;
        if(wany(run && m > 0)) solve(Kc);
#pragma unroll
        for(int r = 0; r < G; r++) Kc[r] = (r < m && !((clmask >> r) & 1u)) ? -Kc[r] : 0.0;
        if(l < S)
        {
#pragma unroll
          for(int r = 0; r < G; r++) L.K[r * KS + l] = Kc[r];
        }
      }
      sync();
      const bool wr = act && ok && live;
      {
        double * Kr = Ks + (static_cast<long>(i) * G + l) * S;
#pragma unroll
        for(int a = 0; a < S; a++)
        {
          const double kv = L.K[l * KS + a];
          if(wr) Kr[a] = kv;
        }
        if(wr) ks[static_cast<long>(i) * G + l] = k_own;
      }
      DPROF_ADD(5, t_s2);
      const long long t_s3 = DPROF_T();
      // term of the gradient-norm test: max_r |k_r| / (|u_r| + 1)
      {
        double mx = 0.0;
#pragma unroll
        for(int r = 0; r < G; r++)
        {
          const double v = fabs(L.v[SL_X][r]) / (fabs(L.v[SL_U][r]) + 1.0); // zero for absent ridges
          mx = (v > mx) ? v : mx;
        }
        if(wr && l == 0) gm[i] = mx;
      }
      // ---- t4 = Quu k (SL_D), Qu (SL_VX: Vx is consumed), row l of Quu to LDS for the transposition
      {
        double t4 = 0.0;
#pragma unroll
        for(int q = 0; q < G; q++)
          t4 += ((l == q) ? quu_own : Hreg[q]) * L.v[SL_X][q];
        L.v[SL_D][l] = t4;
        L.v[SL_VX][l] = Qu;
#pragma unroll
        for(int q = 0; q < G; q++) L.A1()[l * LS + q] = (l == q) ? quu_own : Hreg[q];
      }
      sync();
      {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for(int r = 0; r < G; r++)
        {
          s0 += L.v[SL_X][r] * L.v[SL_VX][r];
          s1 += L.v[SL_X][r] * L.v[SL_D][r];
        }
        if(run)
        {
          dV0 += s0;
          dV1 += 0.5 * s1;
        }
      }
      double Vx_own = Qx_own;
#pragma unroll
      for(int r = 0; r < G; r++)
        Vx_own += Kc[r] * L.v[SL_D][r] + Kc[r] * L.v[SL_VX][r] + QxuRow[r] * L.v[SL_X][r];
      // ---- T2' = K'Quu (column l)
      St T2p[S];
      {
        St QC[G];
#pragma unroll
        for(int r = 0; r < G; r++) QC[r] = L.A1()[r * LS + l];
#pragma unroll
        for(int a = 0; a < S; a++)
        {
          double s = 0.0;
#pragma unroll
          for(int q = 0; q < G; q++)
            s += L.K[q * KS + a] * QC[q];
          T2p[a] = s;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      sync(); // every lane has its column of Quu: A1 takes T2'
#pragma unroll
      for(int a = 0; a < S; a++) L.A1()[a * G + l] = T2p[a];
      sync();
      // ---- Vxx = sym(Qxx + K'Quu K + K'Qxu' + Qxu K): column l of the sum, then the transposition through LDS
      St Tc[S];
#pragma unroll
      for(int a = 0; a < S; a++)
      {
        double s = (l < S) ? L.Vxx[a * S + l] : 0.0;
#pragma unroll
        for(int r = 0; r < G; r++)
          s += L.A1()[a * G + r] * Kc[r] + L.K[r * KS + a] * QxuRow[r] + L.Qxu[a * LS + r] * Kc[r];
        Tc[a] = s;
        __builtin_amdgcn_sched_barrier(0);
      }
      if(l < S)
      {
#pragma unroll
        for(int a = 0; a < S; a++) L.Vxx[a * S + l] = Tc[a];
      }
      sync();
      St Tr[S];
#pragma unroll
      for(int a = 0; a < S; a++) Tr[a] = (l < S) ? L.Vxx[l * S + a] : 0.0;
      sync();
      if(l < S)
      {
#pragma unroll
        for(int a = 0; a < S; a++) L.Vxx[a * S + l] = 0.5 * (Tc[a] + Tr[a]);
        L.v[SL_VX][l] = Vx_own;
      }
      kprev = k_own;
      m_next = m;
      sync();
      DPROF_ADD(6, t_s3);
    }
    dV0_out = dV0;
    dV1_out = dV1;
    DPROF_ADD(1, t_bw);
    return ok;
  }

  // ------------------------------------------------------------------------------------------ solve
  __device__ __forceinline__ void solve_instance(int * out_iters, int * out_status, double * out_cost, double * u_out,
                                                 double * x_out)
  {
    {
      double wr = P.w_run[0], wt = P.w_term[0];
#pragma unroll
      for(int a = 1; a < S; a++)
      {
        wr = (l == a) ? P.w_run[a] : wr;
        wt = (l == a) ? P.w_term[a] : wt;
      }
      wrun_own = wr;
      wterm_own = wt;
    }
#if defined(CCC_DDPG_PROF)
#pragma unroll
    for(int k = 0; k < 16; k++) prof[k] = 0.0;
    const long long t_all = DPROF_T();
#endif
    double lambda = P.lambda0, dlambda = P.dlambda0;
    int cb = 0;
    double cost = 0.0;
    int it = 1, status = 0;
    bool done = false;
    bool first = true; // the rollout of the initial inputs runs through the line-search slot of the first round
    auto increase = [&]() {
      dlambda = fmax(dlambda * P.lambda_factor, P.lambda_factor);
      lambda = fmax(lambda * dlambda, P.lambda_min);
    };
    auto decrease = [&]() {
      dlambda = fmin(dlambda / P.lambda_factor, 1.0 / P.lambda_factor);
      lambda = lambda * dlambda * (lambda > P.lambda_min ? 1.0 : 0.0);
    };
    while(first || wany(!done))
    {
      const bool act = !first && !done;
      double dV0 = 0.0, dV1 = 0.0;
      bool ok = true;
      bool searching = false;
      if(!first)
      {
        ok = backward_pass(cb, lambda, act, dV0, dV1);
        gsync();
        if(act)
        {
          if(!ok)
          {
            increase();
            if(lambda > P.lambda_max)
            {
              status = -1;
              done = true;
            }
          }
          else
          {
            double gsum = 0.0;
            for(int i = 0; i < N; i++) gsum += gm[i];
            gsum = gsum / N;
            if(gsum < P.k_rel_norm_thre && lambda < P.lambda_thre)
            {
              decrease();
              status = 1;
              done = true;
            }
            else
              searching = true;
          }
        }
      }
      // ---- line search over alpha_list (first round: the one rollout of the initial inputs)
      bool accepted = false;
      double actual = 0.0, costc = 0.0;
      for(int a = 0; a < 11; a++)
      {
        const bool go = first ? (a == 0) : (searching && !accepted);
        if(!wany(go)) break;
        const double alpha = first ? -1.0 : P.alpha[a];
        const double cc = rollout(alpha, cb, go);
        gsync();
        if(first)
          cost = cc;
        else if(go)
        {
          costc = cc;
          actual = cost - costc;
          const double expected = -alpha * (dV0 + alpha * dV1);
          const double ratio = expected > 0 ? actual / expected : (actual > 0 ? 1.0 : (actual < 0 ? -1.0 : 0.0));
          if(ratio > P.ratio_thre) accepted = true;
        }
      }
      if(first)
      {
        first = false;
        done = P.max_iter < 1;
        continue;
      }
      if(searching)
      {
        if(accepted)
        {
          decrease();
          cb ^= 1;
          cost = costc;
          if(actual < P.cost_thre)
          {
            status = 2;
            done = true;
          }
        }
        else
        {
          increase();
          if(lambda > P.lambda_max)
          {
            status = -1;
            done = true;
          }
        }
      }
      if(act && ok && !done)
      {
        it++;
        if(it > P.max_iter) done = true;
      }
    }
    if(it > P.max_iter) it = P.max_iter;
    // ---- outputs: the current trajectory
    if(live)
    {
      const double * us = ubuf(cb);
      for(int e = l; e < N * G; e += G) u_out[e] = us[e];
      if(x_out)
      {
        const double * xs = xbuf(cb);
        for(int e = l; e < (N + 1) * S; e += G) x_out[e] = xs[e];
      }
      if(l == 0)
      {
        if(out_iters) *out_iters = it;
        if(out_status) *out_status = status;
        if(out_cost) *out_cost = cost;
      }
#if defined(CCC_DDPG_PROF)
      prof[7] = (double)(DPROF_T() - t_all);
      if(l == 0)
#pragma unroll
        for(int k = 0; k < 16; k++) u_out[k] = prof[k]; // a profiling build returns timings INSTEAD of a plan
#endif
    }
  }
};

template<int S, typename Real>
__global__ __launch_bounds__(64, 1) void ddp_group_kernel(Params P, Batch B, long n)
{
  __shared__ Lds<S, Real> lds[4];
  const int lane = static_cast<int>(threadIdx.x), g = lane >> 4, l = lane & 15;
  const long inst = static_cast<long>(blockIdx.x) * 4 + g;
  const bool live = inst < n;
  const long b = live ? inst : n - 1;
  const int N = P.N;
  Group<S, Real> grp{P, lds[g], l, live, static_cast<unsigned>(g * 16)};
  grp.N = N;
  grp.phase_dim = B.phase_dim + b * P.P;
  grp.phase_vertex = B.phase_vertex + b * P.P * G * 3;
  grp.phase_ridge = B.phase_ridge + b * P.P * G * 3;
  grp.step_phase = B.step_phase + b * N;
  grp.ref_pos = B.ref_pos + b * (N + 1) * 3;
  grp.ref_ori = B.ref_ori ? B.ref_ori + b * (N + 1) * 3 : nullptr;
  grp.inertia = B.inertia ? B.inertia + b * 9 : nullptr;
  grp.x0 = B.x0 + b * S;
  grp.u_init = B.u_init ? B.u_init + b * N * G : nullptr;
  grp.X = B.X + b * 2 * (N + 1) * S;
  grp.U = B.U + b * 2 * N * G;
  grp.TF = B.TF + b * 2 * N * 3;
  grp.ks = B.ks + b * N * G;
  grp.Ks = B.Ks + b * N * G * S;
  grp.gm = B.gm + b * N;
  grp.solve_instance(B.iters ? B.iters + b : nullptr, B.status ? B.status + b : nullptr, B.cost ? B.cost + b : nullptr,
                     B.u_out + b * N * G, B.x_out ? B.x_out + b * (N + 1) * S : nullptr);
}
} // namespace ddpg
} // namespace ccc_amd
