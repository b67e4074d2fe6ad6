// ism.hip -- batched CCC::IntrinsicallyStableMpc::planOnce() on MI355X (gfx950): kernel + C-ABI.
// (the first widening of the hot path; same dual active-set machinery as LinearMpcZmp.)
//
// Path replaced (reference file:line under /root/reference):
//   src/IntrinsicallyStableMpc.cpp:8-45     IntrinsicallyStableMpc1d constructor: P (dt on and below the diagonal),
//                                           H = w_vel I + w_zmp P'P, the stability equality row a (eq. 14), ZMP rows +-P
//   src/IntrinsicallyStableMpc.cpp:63-104   procOnce: right-hand sides, the external QP solve (:93), ZMP update + clamp
//   src/IntrinsicallyStableMpc.cpp:106-139  planOnce: x axis, then y axis
//
// Per axis:  min 1/2 u'Hu + g'u,  g = w_zmp P'(z0 1 - zref),  s.t.  a'u = cp - z0,  zmin - z0 <= P u <= zmax - z0.
// With Ct = [P; a] ((N+1) x N) the constraint values are Ct u = Ct u* + G mu, G = Ct H^-1 Ct' batch-constant and
// u* = -H^-1 g the unconstrained minimiser; because g = -w_zmp P'r (r = zref - z0 1) the offsets are
// d = Ct u* = w_zmp G[:, :N] r -- G itself, no further matrix.  So this is LinearMpcZmp's range problem
// lo <= G mu <= hi (csrc/zmp.hip) with N+1 rows, the last one of zero width (an equality: it enters first and is never
// dropped), solved by the same sweep-tableau iteration: one QP per workgroup, the tableau packed in LDS (sym_tableau.h).  u0 = H^-1[0, :] Ct' (mu + w_zmp [r; 0]) needs one more constant row (hc).
#include "common.h"
#include "sym_tableau.h"
#include "wave_group.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ccc_amd
{
constexpr int kIsmNP = 192;   // rows per QP (N + 1 <= 192: the largest packed tableau of 4 x 4 tiles that fits the LDS)
constexpr int kIsmWR = kIsmNP / 64; // wavefronts that hold rows
constexpr double kIsmInf = __builtin_huge_val();

struct IsmDev
{
  int N;             // horizon steps
  const double * G;  // [NP][NP]  Ct H^-1 Ct', identity on the padding
  const double * Wc; // [N][NP]   H^-1 Ct'  (row 0 = hc)
  double w_zmp;
  double dt;
};

struct IsmRed
{
  double val[kIsmWR];
  int idx[kIsmWR];
};

struct IsmSel
{
  double val[kIsmWR], sig[kIsmWR];
  int idx[kIsmWR];
};

__device__ __forceinline__ double ism_lane_value(double v, int k) // k uniform: two v_readlane_b32
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}

// (min value over the rows -- the first WR wavefronts of the block --, lowest thread index attaining it; index kIsmNP
// if no finite candidate)
template<int WR>
__device__ __forceinline__ void ism_block_argmin(double v, IsmRed * red, double & vmin, int & imin)
{
  const int tid = threadIdx.x, w = tid >> 6;
  const double wm = WaveGroup<64>::min(v);
  const int wi = WaveGroup<64>::first(v == wm && v < kIsmInf);
  // (no barrier before the write: the only caller is the ratio test, whose previous read of red lies at least two
  // barriers back -- the update's and the row rewrite's)
  if((tid & 63) == 0 && w < WR)
  {
    red->val[w] = wm;
    red->idx[w] = wi < 64 ? wi + 64 * w : kIsmNP;
  }
  __syncthreads();
  vmin = red->val[0];
  imin = red->idx[0];
#pragma unroll
  for(int k = 1; k < WR; ++k) // (ties: the lowest row index)
  {
    const double b = red->val[k];
    const int ib = red->idx[k];
    const bool take = (ib < kIsmNP) && (imin >= kIsmNP || b < vmin);
    vmin = take ? b : vmin;
    imin = take ? ib : imin;
  }
}

// init [nqp][2] (capture_point, planned_zmp), ref [nqp][3][N] (ref zmp, zmin, zmax), zmp [nqp], vel [nqp][N] | null,
// status [nqp] | null.  A "qp" is one axis of one instance.
// One QP per workgroup, the sweep tableau packed (lower triangle in 4 x 4 tiles, sym_tableau.h) in LDS: thread t updates
// the tiles t, t + NT, ..., thread i < NR owns row i (bounds, multiplier, flags).  NR >= N + 1 rows.
template<int NR, int TPT>
__global__ __launch_bounds__((SymTab<NR, 4, TPT>::NT), (SymTab<NR, 4, TPT>::kMinWaves)) void ism_plan_kernel(
    IsmDev P, long nqp, const double * __restrict__ init, const double * __restrict__ ref, double control_dt,
    double * __restrict__ zmp, double * __restrict__ vel, int * __restrict__ status, const int * __restrict__ redo_list,
    const int * __restrict__ redo_count)
{
  using ST = SymTab<NR, 4, TPT>;
  constexpr int NP = kIsmNP; // row stride of P.G / P.Wc, and the "no candidate" index
  constexpr int WR = (NR + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double * T = smem;                 // packed tableau
  double * cb = smem + ST::kDoubles; // [NR] staging of the pivot row / of mu / of rho
  IsmRed * red = reinterpret_cast<IsmRed *>(cb + NR);
  IsmSel * sel = reinterpret_cast<IsmSel *>(red + 1);
  const int i = threadIdx.x;
  const bool lead = i < NR; // thread i owns row i: bounds, multiplier, flags
  int ta[TPT], tb[TPT];     // this thread's tiles: i, i + NT, ...
#pragma unroll
  for(int k = 0; k < TPT; ++k) ST::tile_of(i + k * ST::NT < ST::NTILE ? i + k * ST::NT : 0, ta[k], tb[k]);
  const int N = P.N;
  const int maxpass = 20 * (N + 1) + 100;

  // redo_list: the QPs the tridiagonal kernel below handed over (normally few); otherwise all of them
  const long nwork = redo_list ? (long)*redo_count : nqp;
  for(long wq = blockIdx.x; wq < nwork; wq += gridDim.x)
  {
    const long qp = redo_list ? (long)redo_list[wq] : wq;
    const bool rng = lead && i < N, iseq = lead && i == N, row = lead && i <= N;
    const double cp = init[qp * 2 + 0], z0 = init[qp * 2 + 1];
    double zr = 0, zl = 0, zh = 0;
    if(rng)
    {
      zr = ref[qp * 3 * N + i];
      zl = ref[qp * 3 * N + N + i];
      zh = ref[qp * 3 * N + 2 * N + i];
    }
    const double r = rng ? zr - z0 : 0.0;
    __syncthreads();
    if(lead) cb[i] = r;
    __syncthreads();
    // tableau <- G, then the offsets d = w_zmp G[:, :N] r off the LDS copy
#pragma unroll
    for(int k = 0; k < TPT; ++k)
      if(i + k * ST::NT < ST::NTILE) ST::load_tile(T, P.G, NP, i + k * ST::NT, ta[k], tb[k]);
    __syncthreads();
    const double d = lead ? ST::matvec_row(T, cb, i) * P.w_zmp : 0.0;
    const double lo = rng ? (zl - z0) - d : (iseq ? (cp - z0) - d : -kIsmInf);
    const double hi = rng ? (zh - z0) - d : (iseq ? (cp - z0) - d : kIsmInf);
    const double tl = row ? 1e-12 * (1.0 + fabs(lo)) : 0.0;
    const double th = row ? 1e-12 * (1.0 + fabs(hi)) : 0.0;
    int st = CCC_STATUS_SOLVED;
    if(__syncthreads_or(rng && lo > hi)) st = CCC_STATUS_INFEASIBLE;

    double z = 0.0, mu = 0.0, dact = 0.0;
    bool inW = false;
    int p = 0;
    double psig = 0.0, pd = 0.0, sig = 0.0;
    bool done = st != CCC_STATUS_SOLVED; // block uniform
    bool need_select = true;
    int passes = 0;

    // per wavefront: the most violated row outside the working set and its side (the equality row first)
    auto post_select = [&]() {
      const double sl = (lo - z) - tl, sh = (z - hi) - th;
      double score = (inW || !row) ? -kIsmInf : fmax(sl, sh);
      if(iseq && !inW) score = 1e300; // the equality row enters first (oracle/qp_gi.c) and stays
      const double key = score > 0.0 ? -score : kIsmInf;
      const double wm = WaveGroup<64>::min(key);
      const int wi = WaveGroup<64>::first(key == wm && key < kIsmInf);
      const double wsig = ism_lane_value((sl >= sh) ? 1.0 : -1.0, wi & 63);
      const int w = threadIdx.x >> 6;
      if((threadIdx.x & 63) == 0 && w < WR)
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
        if(need_select) // the candidates were posted before the previous barrier (post_select)
        {
          double best = sel->val[0], sg = sel->sig[0];
          int cand = sel->idx[0];
#pragma unroll
          for(int k = 1; k < WR; ++k)
          {
            const double a2 = sel->val[k];
            const int i2 = sel->idx[k];
            const bool take = (i2 < NP) && (cand >= NP || a2 < best);
            best = take ? a2 : best;
            sg = take ? sel->sig[k] : sg;
            cand = take ? i2 : cand;
          }
          if(cand >= NP) break;
          p = cand;
          sig = sg;
          if(lead && i == cand)
          {
            psig = sg;
            pd = (sg > 0.0) ? lo : hi;
          }
        }
        const double c = lead ? T[ST::entry(p, i)] : 0.0; // column p = row p (symmetric)
        if(lead) cb[i] = c; // staged as the pivot column already: most pivots add row p itself (published by the
                            // barriers of the ratio test below)
        const double dm = -sig * c;
        const bool blocking = inW && !iseq && ((mu > 0.0 && dm < 0.0) || (mu < 0.0 && dm > 0.0));
        const bool isp = lead && (i == p);
        const double num = isp ? psig * (pd - z) : -mu;
        const double den = isp ? c : dm;
        double ratio = (isp || blocking) ? num / den : kIsmInf;
        if(isp && !(c > 0.0)) ratio = kIsmInf; // no curvature left along row p: it cannot be satisfied
        double t;
        int kk;
        ism_block_argmin<WR>(ratio, red, t, kk);
        if(kk >= NP)
        {
          st = CCC_STATUS_INFEASIBLE;
          done = true;
          break;
        }
        const bool isadd = (kk == p);
        const double s = isadd ? 1.0 : -1.0;
        if(inW)
          mu = fma(t, dm, mu);
        else
          z = fma(sig * t, c, z);
        if(isp) mu += sig * t;
        // pivot on row/column kk
        // bookkeeping of the step, then the candidates of the next selection: they do not depend on the tableau update,
        // so posting them here lets the selection ride on the barriers of the update
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
        double v = c;
        if(!isadd) // a row leaves: stage its column instead
        {
          v = lead ? T[ST::entry(kk, i)] : 0.0;
          if(lead) cb[i] = v;
          __syncthreads();
        }
        const double rp = 1.0 / cb[kk];
        const double g = v * rp;
#pragma unroll
        for(int k = 0; k < TPT; ++k)
          if(i + k * ST::NT < ST::NTILE) ST::update_tile(T, cb, rp, i + k * ST::NT, ta[k], tb[k]);
        __syncthreads();
        if(lead) T[ST::entry(kk, i)] = (i == kk) ? -rp : s * g; // row/column kk (the update left noise there), the pivot
        __syncthreads();
        if(++passes > maxpass)
        {
          st = CCC_STATUS_MAX_ITER;
          done = true;
        }
      }
      if(st != CCC_STATUS_SOLVED) break;
      // closing refinement against the untouched G (see csrc/zmp.hip): rho = d_W - (G mu)_W, mu_W -= T_WW rho,
      // z = G mu recomputed; re-open if a row turns out violated
      __syncthreads();
      if(lead) cb[i] = inW ? mu : 0.0;
      __syncthreads();
      double acc = 0.0;
      if(lead)
        for(int j = 0; j < NR; ++j) acc = fma(P.G[j * NP + i], cb[j], acc);
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
        for(int j = 0; j < NR; ++j) acc = fma(P.G[j * NP + i], cb[j], acc);
      z = inW ? dact : acc;
      const double sl = (lo - z) - tl, sh = (z - hi) - th;
      const int reopen = __syncthreads_or(row && !inW && fmax(sl, sh) > 0.0);
      need_select = true;
      if(!reopen) break;
    }

    // certificate: every row of lo <= G mu <= hi holds at the returned multipliers (also catches NaN).  When the
    // capture point cannot be caught inside the ZMP limits the tableau runs out of curvature instead of hitting the
    // pass limit; such an instance must not be reported as solved.
    if(st == CCC_STATUS_SOLVED)
    {
      __syncthreads();
      if(lead) cb[i] = inW ? mu : 0.0;
      __syncthreads();
      double acc = 0.0;
      if(lead)
        for(int j = 0; j < NR; ++j) acc = fma(P.G[j * NP + i], cb[j], acc);
      const bool bad = row && !((lo - acc) <= 1e-9 * (1.0 + fabs(lo)) && (acc - hi) <= 1e-9 * (1.0 + fabs(hi)));
      if(__syncthreads_or(bad ? 1 : 0)) st = CCC_STATUS_INFEASIBLE;
    }
    // outputs: u = H^-1 Ct'(mu + w_zmp [r; 0]); zmp = clamp(z0 + control_dt u0, zmin0, zmax0)  (:93-101)
    __syncthreads();
    if(lead) cb[i] = row ? mu + P.w_zmp * r : 0.0;
    __syncthreads();
    if(lead && i == 0)
    {
      double u0 = 0.0;
      for(int k = 0; k <= N; ++k) u0 = fma(P.Wc[k], cb[k], u0);
      const double cdt = control_dt < 0 ? P.dt : control_dt;
      double zv = z0 + cdt * u0;
      zv = zv < zl ? zl : (zh < zv ? zh : zv);
      zmp[qp] = zv;
      if(status) status[qp] = (passes << 8) | st;
    }
    if(vel && rng)
    {
      double uj = 0.0;
      for(int k = 0; k <= N; ++k) uj = fma(P.Wc[(size_t)i * NP + k], cb[k], uj);
      vel[qp * N + i] = uj;
    }
    __syncthreads();
  }
}
// ---------------------------------------------------------------------------------------------------------------
// The same QP in the space of the ZMP positions, ONE QP PER WAVEFRONT (default path).
// With y_i = z_{i+1} = z0 + dt (u_0 + .. + u_i) the objective of src/IntrinsicallyStableMpc.cpp:29-32,66-70 is
//     w_vel/2 sum ((y_i - y_{i-1}) / dt)^2 + w_zmp/2 sum (y_i - zref_i)^2 ,      y_{-1} = z0,
// a TRIDIAGONAL, strictly diagonally dominant Hessian (condition number ~ 1 + 4 w_vel / (w_zmp dt^2) = 11 with the
// defaults), the ZMP limits are a box on y, and the stability row a'u = cp - z0 (:35-39,72) is one linear equality
// at'y = c.  So:
//   * the equality is dualised: for a multiplier nu the rest is a box QP  min 1/2 y'Hy + (q + nu at)'y, lo <= y <= hi,
//     and phi(nu) = at'y(nu) - c is monotone and piecewise linear -- a safeguarded Newton iteration on nu (slope
//     -at_F' H_FF^-1 at_F, a by-product of the inner solve) finds its root in 5-7 steps;
//   * the box QP is solved by projected Newton; the Newton step is a tridiagonal solve on the free rows (clamped rows
//     become identity rows), done by PARALLEL CYCLIC REDUCTION across the wavefront: 7 steps for 128 rows, two rows per
//     lane, both right-hand sides (the step and at) in one pass; strictly diagonally dominant => no pivoting needed;
//   * warm starts all the way (y and the clamped set carry over between values of nu): ~3 inner iterations per outer one.
// ~20 tridiagonal solves of ~500 instructions per QP instead of ~42 pivots of a 104 x 104 tableau.  QPs that do not
// converge within the iteration budget (an uncatchable capture point, i.e. an infeasible QP, drives nu to infinity) are
// handed to the tableau kernel above through a work list; it classifies them.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kPcrNP = 128;
constexpr int kPcrOuter = 40, kPcrInner = 30;

struct IsmPcrDev
{
  int N;
  const double * at; // [N]  (a_i - a_{i+1}) / dt, a_N = 0: the stability row in y
  double a0_dt;      // a_0 / dt
  double w_zmp, w_vel, dt;
};

__global__ __launch_bounds__(256) void ism_plan_pcr_kernel(IsmPcrDev P, long nqp, const double * __restrict__ init,
                                                           const double * __restrict__ ref, double control_dt,
                                                           double * __restrict__ zmp, double * __restrict__ vel,
                                                           int * __restrict__ status, int * __restrict__ redo_list,
                                                           int * __restrict__ redo_count, int max_outer)
{
  __shared__ double sh[4][6][kPcrNP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long qp = (long)blockIdx.x * 4 + wave;
  if(qp >= nqp) return; // (whole wavefront; the kernel uses no workgroup barrier)
  double * Sy = sh[wave][0];
  double * Sa = sh[wave][1];
  double * Sr = sh[wave][2];
  double * Sc = sh[wave][3];
  double * Sd = sh[wave][4];
  double * Se = sh[wave][5];
  const int N = P.N;
  const double kk = P.w_vel / (P.dt * P.dt), e = -kk;
  const double cp = init[qp * 2 + 0], z0 = init[qp * 2 + 1];
  const double c_eq = cp - z0 + P.a0_dt * z0;
  int idx[2] = {lane, lane + 64};
  bool in[2];
  double lo[2], hi[2], q[2], at[2], dgn[2], y[2];
  bool bad = false;
#pragma unroll
  for(int u = 0; u < 2; u++)
  {
    const int i = idx[u];
    in[u] = i < N;
    const double zr = in[u] ? ref[qp * 3 * N + i] : 0.0;
    lo[u] = in[u] ? ref[qp * 3 * N + N + i] : 0.0;
    hi[u] = in[u] ? ref[qp * 3 * N + 2 * N + i] : 0.0;
    at[u] = in[u] ? P.at[i] : 0.0;
    q[u] = -P.w_zmp * zr - (i == 0 ? kk * z0 : 0.0);
    dgn[u] = P.w_zmp + (i == N - 1 ? kk : 2.0 * kk);
    y[u] = fmin(fmax(z0, lo[u]), hi[u]);
    bad = bad || (in[u] && lo[u] > hi[u]);
  }
  auto wsum = [&](double a, double b) { return WaveGroup<64>::sum(a + b); };
  // (H v)_i for the rows of this lane, v staged in Sy
  auto hmul = [&](const double (&v)[2], double (&hv)[2]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for(int u = 0; u < 2; u++) Sy[idx[u]] = in[u] ? v[u] : 0.0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for(int u = 0; u < 2; u++)
    {
      const int i = idx[u];
      const double vm = (i > 0) ? Sy[i - 1] : 0.0, vp = (i + 1 < N) ? Sy[i + 1] : 0.0;
      hv[u] = in[u] ? dgn[u] * v[u] + e * (vm + vp) : 0.0;
    }
  };
  // parallel cyclic reduction of the tridiagonal system (a, b, c) with the right-hand sides d and f; r = 1 / b on entry
  // and on exit the solutions are d * r and f * r
  auto pcr = [&](double (&a_)[2], double (&b_)[2], double (&c_)[2], double (&d_)[2], double (&f_)[2], double (&r_)[2]) {
    for(int s = 1; s < N; s <<= 1)
    {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for(int u = 0; u < 2; u++)
      {
        Sa[idx[u]] = a_[u];
        Sr[idx[u]] = r_[u];
        Sc[idx[u]] = c_[u];
        Sd[idx[u]] = d_[u];
        Se[idx[u]] = f_[u];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for(int u = 0; u < 2; u++)
      {
        const int im = idx[u] - s, ip = idx[u] + s;
        const bool hm = im >= 0, hp = ip < kPcrNP;
        const double k1 = hm ? a_[u] * Sr[im] : 0.0, k2 = hp ? c_[u] * Sr[ip] : 0.0;
        const double am = hm ? Sa[im] : 0.0, cm = hm ? Sc[im] : 0.0, dm = hm ? Sd[im] : 0.0, fm = hm ? Se[im] : 0.0;
        const double ap = hp ? Sa[ip] : 0.0, cq = hp ? Sc[ip] : 0.0, dp = hp ? Sd[ip] : 0.0, fp = hp ? Se[ip] : 0.0;
        b_[u] = b_[u] - cm * k1 - ap * k2;
        d_[u] = d_[u] - dm * k1 - dp * k2;
        f_[u] = f_[u] - fm * k1 - fp * k2;
        a_[u] = -am * k1;
        c_[u] = -cq * k2;
        r_[u] = fast_rcp(b_[u]); // (v_rcp_f64 + two Newton steps: relative error ~1e-16, the pivots are >= w_zmp)
      }
    }
  };
  double nu = 0.0, nu_lo = -kIsmInf, nu_hi = kIsmInf;
  int solves = 0;
  {
    // first guess: the equality-constrained minimiser without the box, y = H^-1(-q) - nu H^-1 at with nu from at'y = c,
    // clamped -- the multiplier the outer iteration starts from is then usually within a step or two of the root
    double a_[2], b_[2], c_[2], d_[2], f_[2], r_[2];
#pragma unroll
    for(int u = 0; u < 2; u++)
    {
      a_[u] = (in[u] && idx[u] > 0) ? e : 0.0;
      c_[u] = (in[u] && idx[u] + 1 < N) ? e : 0.0;
      b_[u] = in[u] ? dgn[u] : 1.0;
      d_[u] = in[u] ? -q[u] : 0.0;
      f_[u] = in[u] ? at[u] : 0.0;
      r_[u] = fast_rcp(b_[u]);
    }
    pcr(a_, b_, c_, d_, f_, r_);
    ++solves;
    const double s0[2] = {d_[0] * r_[0], d_[1] * r_[1]}, s1[2] = {f_[0] * r_[0], f_[1] * r_[1]};
    const double num = wsum(at[0] * s0[0], at[1] * s0[1]) - c_eq, den = wsum(at[0] * s1[0], at[1] * s1[1]);
    if(den > 1e-300) nu = num / den;
#pragma unroll
    for(int u = 0; u < 2; u++) y[u] = in[u] ? fmin(fmax(s0[u] - nu * s1[u], lo[u]), hi[u]) : 0.0;
  }
  bool cl[2] = {false, false};
  double sdir[2] = {0.0, 0.0}; // H_FF^-1 at of the last solve
  double hv[2] = {0.0, 0.0};   // H y, carried from the line search into the next iteration
  bool have_hv = false;
  bool exact_move = false;     // the last change of y was the affine move along the multiplier, unclipped
  int st = CCC_STATUS_MAX_ITER;
  if(__any(bad)) st = CCC_STATUS_INFEASIBLE;
  for(int outer = 0; outer < max_outer && st == CCC_STATUS_MAX_ITER; outer++)
  {
    bool first = true, full = true, inner_ok = false;
    for(int inner = 0; inner < kPcrInner; inner++)
    {
      double qq[2], g[2];
      if(!have_hv) hmul(y, hv);
      have_hv = true;
      bool ncl[2];
      bool changed = false;
#pragma unroll
      for(int u = 0; u < 2; u++)
      {
        qq[u] = q[u] + nu * at[u];
        g[u] = hv[u] + qq[u];
        ncl[u] = in[u] && ((y[u] <= lo[u] && g[u] > 0.0) || (y[u] >= hi[u] && g[u] < 0.0));
        changed = changed || (ncl[u] != cl[u]);
      }
      if((!first || exact_move) && !__any(changed) && full)
      {
        inner_ok = true;
        break;
      }
      first = false;
      const double J0 = wsum(in[0] ? y[0] * (0.5 * hv[0] + qq[0]) : 0.0, in[1] ? y[1] * (0.5 * hv[1] + qq[1]) : 0.0);
      // ---- Newton step: tridiagonal system on the free rows, identity on the clamped / padding rows; PCR
      double a_[2], b_[2], c_[2], d_[2], f_[2], r_[2];
#pragma unroll
      for(int u = 0; u < 2; u++)
      {
        cl[u] = ncl[u];
        const bool fr = in[u] && !cl[u];
        a_[u] = (fr && idx[u] > 0) ? e : 0.0;
        c_[u] = (fr && idx[u] + 1 < N) ? e : 0.0;
        b_[u] = fr ? dgn[u] : 1.0;
        d_[u] = fr ? -qq[u] : (in[u] ? y[u] : 0.0);
        f_[u] = fr ? at[u] : 0.0;
        r_[u] = fast_rcp(b_[u]);
      }
      pcr(a_, b_, c_, d_, f_, r_);
      ++solves;
      double xn[2];
#pragma unroll
      for(int u = 0; u < 2; u++)
      {
        xn[u] = d_[u] * r_[u];
        sdir[u] = f_[u] * r_[u];
      }
      // ---- projected step, halved until the objective does not increase
      double alpha = 1.0;
      for(;;)
      {
        double yt[2], ht[2];
#pragma unroll
        for(int u = 0; u < 2; u++) yt[u] = in[u] ? fmin(fmax(y[u] + alpha * (xn[u] - y[u]), lo[u]), hi[u]) : 0.0;
        hmul(yt, ht);
        const double Jt = wsum(in[0] ? yt[0] * (0.5 * ht[0] + qq[0]) : 0.0, in[1] ? yt[1] * (0.5 * ht[1] + qq[1]) : 0.0);
        if(Jt <= J0 + 1e-14 * fabs(J0) || alpha < 1e-8)
        {
          y[0] = yt[0];
          y[1] = yt[1];
          hv[0] = ht[0]; // H y of the accepted point: the gradient of the next iteration
          hv[1] = ht[1];
          break;
        }
        alpha *= 0.5;
      }
      full = alpha == 1.0;
    }
    if(!inner_ok) break; // inner iteration out of budget: hand the QP over
    const double phi = wsum(at[0] * y[0], at[1] * y[1]) - c_eq;
    if(fabs(phi) <= 1e-13 * (1.0 + fabs(c_eq)))
    {
      st = CCC_STATUS_SOLVED;
      break;
    }
    if(phi > 0.0)
      nu_lo = nu;
    else
      nu_hi = nu;
    const double slope = wsum(at[0] * sdir[0], at[1] * sdir[1]); // > 0;  d phi / d nu = -slope
    double nn = slope > 1e-300 ? nu + phi / slope : (phi > 0.0 ? nu + 1.0 : nu - 1.0);
    if(!(nu_lo < nn && nn < nu_hi))
    {
      const bool both = nu_lo > -kIsmInf && nu_hi < kIsmInf;
      nn = both ? 0.5 * (nu_lo + nu_hi) : nu + (2.0 * fabs(nu) + 1.0) * (phi > 0.0 ? 1.0 : -1.0);
    }
    // on the current free set y is affine in the multiplier: y(nn) = y - (nn - nu) H_FF^-1 at.  If that move stays inside
    // the box, the next inner loop finds the clamped set unchanged and stops without a solve (and phi is then zero up
    // to rounding); otherwise it is the warm start
    {
      const double dn = nn - nu;
      bool clipped = false;
#pragma unroll
      for(int u = 0; u < 2; u++)
        if(in[u] && !cl[u])
        {
          const double t = y[u] - dn * sdir[u];
          const double tc = fmin(fmax(t, lo[u]), hi[u]);
          clipped = clipped || (tc != t);
          y[u] = tc;
        }
      exact_move = !__any(clipped);
      have_hv = false;
    }
    nu = nn;
  }
  if(st != CCC_STATUS_SOLVED)
  {
    if(lane == 0)
    {
      const int w = atomicAdd(redo_count, 1);
      redo_list[w] = (int)qp;
    }
    return; // the tableau kernel writes this QP's outputs
  }
  // ---- outputs: u_i = (y_i - y_{i-1}) / dt; zmp = clamp(z0 + control_dt u_0, zmin_0, zmax_0)  (:93-101)
  __builtin_amdgcn_wave_barrier();
  Sy[idx[0]] = y[0];
  Sy[idx[1]] = in[1] ? y[1] : 0.0;
  __builtin_amdgcn_wave_barrier();
  if(lane == 0)
  {
    const double u0 = (y[0] - z0) / P.dt;
    const double cdt = control_dt < 0 ? P.dt : control_dt;
    double zv = z0 + cdt * u0;
    zv = zv < lo[0] ? lo[0] : (hi[0] < zv ? hi[0] : zv);
    zmp[qp] = zv;
    if(status) status[qp] = (solves << 8) | st;
  }
  if(vel)
  {
#pragma unroll
    for(int u = 0; u < 2; u++)
      if(in[u]) vel[qp * N + idx[u]] = (y[u] - (idx[u] > 0 ? Sy[idx[u] - 1] : z0)) / P.dt;
  }
}
} // namespace ccc_amd

using namespace ccc_amd;

struct ccc_ism
{
  int device = 0;
  int N = 0;
  double com_height = 0, horizon_duration = 0, horizon_dt = 0, w_zmp = 1.0, w_zmp_vel = 1e-3;
  double *dG = nullptr, *dWc = nullptr;
  double * dAt = nullptr; // [N] stability row in ZMP-position variables (ism_plan_pcr_kernel)
  double a0_dt = 0;
  int * redo = nullptr;   // [1 + 2 n]: count, list of QPs handed to the tableau kernel
  int64_t redo_cap = 0;
  int num_cu = 0;
  // staging for the host-pointer entry point
  int64_t cap = 0;
  double *d_in = nullptr, *d_out = nullptr;
  int32_t * d_status = nullptr;
  hipStream_t stream = nullptr;
  // development switches, read ONCE in ccc_ism_create (never per launch)
  bool env_tableau = false;
  int env_pcr_outer = -1; // CCC_ISM_PCR_OUTER: the outer budget (small values exercise the list); < 0: the default
};

namespace
{
constexpr double kG = 9.80665; // include/CCC/Constants.h:10

// Batch constants in long double: H = w_vel I + w_zmp P'P (src/IntrinsicallyStableMpc.cpp:29-32), Ct = [P; a] with the
// stability row a (:35-39), Wc = H^-1 Ct' by Cholesky, G = Ct Wc.
int upload_model(ccc_ism * h)
{
  typedef long double ld;
  const int N = h->N, NP = kIsmNP, R = N + 1;
  const ld dt = h->horizon_dt;
  std::vector<ld> H((size_t)N * N), Ct((size_t)R * N, 0.0L), L((size_t)N * N, 0.0L), W((size_t)N * R);
  // P'P[i][j] = dt^2 * #{k >= max(i, j)} = dt^2 (N - max(i, j))
  for(int i = 0; i < N; i++)
    for(int j = 0; j < N; j++)
      H[(size_t)i * N + j] = (ld)h->w_zmp * dt * dt * (ld)(N - (i > j ? i : j)) + (i == j ? (ld)h->w_zmp_vel : 0.0L);
  for(int i = 0; i < N; i++)
    for(int j = 0; j <= i; j++) Ct[(size_t)i * N + j] = dt;
  const double omega = std::sqrt(kG / h->com_height), lambda = std::exp(-1 * omega * h->horizon_dt); // :15 (double, as there)
  {
    double a = (1 - lambda) / (omega * (1 - std::pow(lambda, N)));
    for(int j = 0; j < N; j++)
    {
      Ct[(size_t)N * N + j] = a;
      a = lambda * a;
    }
  }
  for(int j = 0; j < N; j++)
  {
    ld dgn = H[(size_t)j * N + j];
    for(int k = 0; k < j; k++) dgn -= L[(size_t)j * N + k] * L[(size_t)j * N + k];
    if(!(dgn > 0)) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ism_create: objective matrix is not positive definite");
    const ld ljj = sqrtl(dgn);
    L[(size_t)j * N + j] = ljj;
    for(int i = j + 1; i < N; i++)
    {
      ld s = H[(size_t)i * N + j];
      for(int k = 0; k < j; k++) s -= L[(size_t)i * N + k] * L[(size_t)j * N + k];
      L[(size_t)i * N + j] = s / ljj;
    }
  }
  for(int c = 0; c < R; c++)
  {
    std::vector<ld> y(N);
    for(int i = 0; i < N; i++)
    {
      ld s = Ct[(size_t)c * N + i];
      for(int k = 0; k < i; k++) s -= L[(size_t)i * N + k] * y[k];
      y[i] = s / L[(size_t)i * N + i];
    }
    for(int i = N - 1; i >= 0; i--)
    {
      ld s = y[i];
      for(int k = i + 1; k < N; k++) s -= L[(size_t)k * N + i] * y[k];
      y[i] = s / L[(size_t)i * N + i];
    }
    for(int i = 0; i < N; i++) W[(size_t)i * R + c] = y[i];
  }
  std::vector<double> G((size_t)NP * NP, 0.0), Wc((size_t)N * NP, 0.0);
  for(int a = 0; a < R; a++)
    for(int b = 0; b < R; b++)
    {
      ld s = 0;
      for(int k = 0; k < N; k++) s += Ct[(size_t)a * N + k] * W[(size_t)k * R + b];
      G[(size_t)a * NP + b] = (double)s;
    }
  for(int a = 0; a < R; a++) // exact symmetry (the kernel reads column i as row i)
    for(int b = 0; b < a; b++) G[(size_t)a * NP + b] = G[(size_t)b * NP + a] = 0.5 * (G[(size_t)a * NP + b] + G[(size_t)b * NP + a]);
  for(int a = R; a < NP; a++) G[(size_t)a * NP + a] = 1.0;
  for(int i = 0; i < N; i++)
    for(int c = 0; c < R; c++) Wc[(size_t)i * NP + c] = (double)W[(size_t)i * R + c];
  CCC_HIP_CHECK(hipMalloc(&h->dG, G.size() * sizeof(double)));
  CCC_HIP_CHECK(hipMalloc(&h->dWc, Wc.size() * sizeof(double)));
  CCC_HIP_CHECK(hipMemcpy(h->dG, G.data(), G.size() * sizeof(double), hipMemcpyHostToDevice));
  CCC_HIP_CHECK(hipMemcpy(h->dWc, Wc.data(), Wc.size() * sizeof(double), hipMemcpyHostToDevice));
  {
    // stability row in the ZMP-position variables y_i = z_{i+1}: a'u = sum a_i (y_i - y_{i-1}) / dt = at'y - a_0 z0 / dt
    std::vector<double> av(N + 1, 0.0), atv(N);
    double a = (1 - lambda) / (omega * (1 - std::pow(lambda, N)));
    for(int j = 0; j < N; j++)
    {
      av[j] = a;
      a = lambda * a;
    }
    for(int j = 0; j < N; j++) atv[j] = (av[j] - av[j + 1]) / h->horizon_dt;
    h->a0_dt = av[0] / h->horizon_dt;
    CCC_HIP_CHECK(hipMalloc(&h->dAt, atv.size() * sizeof(double)));
    CCC_HIP_CHECK(hipMemcpy(h->dAt, atv.data(), atv.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  return CCC_OK;
}
} // namespace

extern "C" int ccc_ism_create(double com_height, double horizon_duration, double horizon_dt, double w_zmp,
                              double w_zmp_vel, int device, ccc_ism_t ** out)
{
  if(!out) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ism_create: out is NULL");
  *out = nullptr;
  if(!(com_height > 0) || !(horizon_duration > 0) || !(horizon_dt > 0) || !(w_zmp >= 0) || !(w_zmp_vel > 0))
    return fail(CCC_ERR_INVALID_ARGUMENT,
                "ccc_ism_create: com_height, horizon_duration, horizon_dt, w_zmp_vel must be > 0 and w_zmp >= 0");
  const int N = (int)std::ceil(horizon_duration / horizon_dt); // src/IntrinsicallyStableMpc.cpp:14
  if(N + 1 > kIsmNP)
    return fail(CCC_ERR_UNSUPPORTED, "ccc_ism_create: horizon_steps %d > %d exceeds the LDS-resident tableau", N,
                kIsmNP - 1);
  int rc = select_device(device);
  if(rc != CCC_OK) return rc;
  CCC_DEVICE_GUARD(device);
  ccc_ism * h = new ccc_ism();
  h->device = device;
  h->env_tableau = std::getenv("CCC_ISM_TABLEAU") != nullptr;
  if(const char * mo = std::getenv("CCC_ISM_PCR_OUTER")) h->env_pcr_outer = std::atoi(mo);
  h->N = N;
  h->com_height = com_height;
  h->horizon_duration = horizon_duration;
  h->horizon_dt = horizon_dt;
  h->w_zmp = w_zmp;
  h->w_zmp_vel = w_zmp_vel;
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
    ccc_ism_destroy(h);
    return rc;
  }
  *out = h;
  return CCC_OK;
}

extern "C" void ccc_ism_destroy(ccc_ism_t * h)
{
  if(!h) return;
  ccc_amd::DeviceGuard ccc_device_guard__(h->device);
  if(h->dG) (void)hipFree(h->dG);
  if(h->dWc) (void)hipFree(h->dWc);
  if(h->dAt) (void)hipFree(h->dAt);
  if(h->redo) (void)hipFree(h->redo);
  if(h->d_in) (void)hipFree(h->d_in);
  if(h->d_out) (void)hipFree(h->d_out);
  if(h->d_status) (void)hipFree(h->d_status);
  if(h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int ccc_ism_horizon_steps(const ccc_ism_t * h)
{
  return h ? h->N : -1;
}

extern "C" int ccc_ism_plan_batch_device(ccc_ism_t * h, int64_t n, const double * init, const double * ref,
                                         double control_dt, double * zmp, double * vel, int32_t * status, void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ism_plan_batch_device: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ism_plan_batch_device: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!init || !ref || !zmp) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ism_plan_batch_device: NULL init/ref/zmp");
  CCC_DEVICE_GUARD(h->device);
  const int64_t nqp = 2 * n;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if(nqp > h->redo_cap) // (synchronous: not inside a captured stream)
  {
    CCC_NO_CAPTURE(stream, "ccc_ism_plan_batch_device");
    if(h->redo) CCC_HIP_CHECK(hipFree(h->redo));
    h->redo = nullptr;
    h->redo_cap = 0;
    CCC_HIP_CHECK(hipMalloc(&h->redo, (size_t)(nqp + 1) * sizeof(int)));
    h->redo_cap = nqp;
  }
  const bool tableau_only = h->env_tableau || h->N >= kPcrNP; // (the tridiagonal kernel: up to 127 steps)
  if(!tableau_only)
  {
    // default path: tridiagonal projected Newton, one QP per wavefront; what it cannot finish goes onto the list
    if(int zrc = zero_words(h->redo, 1, s)) return zrc;
    IsmPcrDev Q{h->N, h->dAt, h->a0_dt, h->w_zmp, h->w_zmp_vel, h->horizon_dt};
    hipLaunchKernelGGL(ism_plan_pcr_kernel, dim3((unsigned)((nqp + 3) / 4)), dim3(256), 0, s, Q, (long)nqp, init, ref,
                       control_dt, zmp, vel, status, h->redo + 1, h->redo, h->env_pcr_outer >= 0 ? h->env_pcr_outer : kPcrOuter);
    CCC_HIP_CHECK(hipGetLastError());
  }
  const int * rl = tableau_only ? nullptr : h->redo + 1;
  const int * rc_ = tableau_only ? nullptr : h->redo;
  // tableau kernel: one workgroup per QP (all of them, or the list); the hardware dispatcher evens out the pivot counts
  const int grid = tableau_only ? (int)std::min<int64_t>(nqp, (int64_t)1 << 22) : (int)std::min<int64_t>(nqp, (int64_t)h->num_cu * 6);
  IsmDev P{h->N, h->dG, h->dWc, h->w_zmp, h->horizon_dt};
  auto go = [&](auto kernel, auto st) -> int {
    using ST = decltype(st);
    const size_t lds = ((size_t)ST::kDoubles + ST::NB * 4) * sizeof(double) + sizeof(IsmRed) + sizeof(IsmSel);
    CCC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(ST::NT), lds, s, P, (long)nqp, init, ref, control_dt, zmp, vel, status, rl,
                       rc_);
    return CCC_OK;
  };
  const int R = h->N + 1; // rows: the ZMP limits of every step and the capture-point equality
  int rc;
  if(R <= 32)
    rc = go(&ism_plan_kernel<32, 2>, SymTab<32, 4, 2>{});
  else if(R <= 56)
    rc = go(&ism_plan_kernel<56, 2>, SymTab<56, 4, 2>{});
  else if(R <= 80)
    rc = go(&ism_plan_kernel<80, 2>, SymTab<80, 4, 2>{});
  else if(R <= 104)
    rc = go(&ism_plan_kernel<104, 2>, SymTab<104, 4, 2>{});
  else if(R <= 128)
    rc = go(&ism_plan_kernel<128, 3>, SymTab<128, 4, 3>{});
  else if(R <= 160)
    rc = go(&ism_plan_kernel<160, 2>, SymTab<160, 4, 2>{});
  else
    rc = go(&ism_plan_kernel<192, 3>, SymTab<192, 4, 3>{});
  if(rc != CCC_OK) return rc;
  CCC_HIP_CHECK(hipGetLastError());
  return CCC_OK;
}

extern "C" int ccc_ism_plan_batch(ccc_ism_t * h, int64_t n, const double * init, const double * ref, double control_dt,
                                  double * zmp, double * vel, int32_t * status)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ism_plan_batch: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ism_plan_batch: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!init || !ref || !zmp) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ism_plan_batch: NULL init/ref/zmp");
  CCC_DEVICE_GUARD(h->device);
  const size_t N = (size_t)h->N;
  const size_t ni = (size_t)n * 4, nr = (size_t)n * 6 * N, nz = (size_t)n * 2, nv = (size_t)n * 2 * N;
  if(n > h->cap)
  {
    if(h->d_in) (void)hipFree(h->d_in);
    if(h->d_out) (void)hipFree(h->d_out);
    if(h->d_status) (void)hipFree(h->d_status);
    h->d_in = h->d_out = nullptr;
    h->d_status = nullptr;
    h->cap = 0;
    CCC_HIP_CHECK(hipMalloc(&h->d_in, (ni + nr) * sizeof(double)));
    CCC_HIP_CHECK(hipMalloc(&h->d_out, (nz + nv) * sizeof(double)));
    CCC_HIP_CHECK(hipMalloc(&h->d_status, (size_t)n * 2 * sizeof(int32_t)));
    h->cap = n;
  }
  if(!h->stream) CCC_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CCC_HIP_CHECK(hipMemcpyAsync(h->d_in, init, ni * sizeof(double), hipMemcpyHostToDevice, h->stream));
  CCC_HIP_CHECK(hipMemcpyAsync(h->d_in + ni, ref, nr * sizeof(double), hipMemcpyHostToDevice, h->stream));
  int rc = ccc_ism_plan_batch_device(h, n, h->d_in, h->d_in + ni, control_dt, h->d_out, vel ? h->d_out + nz : nullptr,
                                     h->d_status, h->stream);
  if(rc != CCC_OK) return rc;
  CCC_HIP_CHECK(hipMemcpyAsync(zmp, h->d_out, nz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if(vel) CCC_HIP_CHECK(hipMemcpyAsync(vel, h->d_out + nz, nv * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if(status)
    CCC_HIP_CHECK(hipMemcpyAsync(status, h->d_status, (size_t)n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  CCC_HIP_CHECK(hipStreamSynchronize(h->stream));
  return CCC_OK;
}
