// z.hip -- batched CCC::LinearMpcZ::planOnce() on MI355X (gfx950): kernel + C-ABI.
// (the vertical companion of LinearMpcXY -- it plans the total_force_z that LinearMpcXY takes.)
//
// Path replaced (reference file:line under /root/reference):
//   src/LinearMpcZ.cpp:10-29     ModelContactPhase / ModelNoncontactPhase: x = [m z, m zdot], A = [[0,1],[0,0]],
//                                B = [0,1]' (contact only), C = [1/m, 0], E = [0, -m g]
//   include/CCC/StateSpaceModel.h:164-216   ZOH with offset -- closed form here (A^2 = 0):
//                                Ad = [[1,dt],[0,1]], Bd = [dt^2/2, dt]', Ed = -m g [dt^2/2, dt]'
//   include/CCC/VariantSequentialExtension.h:110-208   setup(extend_for_output = true) -- closed form here:
//                                output j (height after step j) responds to the force of contact step s <= j with
//                                c (j - s + 1/2), c = dt^2/m; free response z0 + t v0 - g t^2/2, t = (j+1) dt
//   src/LinearMpcZ.cpp:48-71     planOnce (zero force when there is no contact at current_time)
//   src/LinearMpcZ.cpp:73-94     procOnce: H = w_pos B'B + w_force I, g = -w_pos B'(ref - free response),
//                                bounds (10, 10 m g), the external QP solve (:93), [0]
//
// One instance per wavefront (64-thread workgroup), lane i = variable i (one per contact step, <= 64).  H is built in
// closed form, inverted by sweeping every variable (LDS tableau, 32 KB), and the bound-constrained QP is solved by
// the dual active-set / sweep-tableau iteration of LinearMpcZmp on G = H^-1; a closing primal refinement with the
// untouched H removes the drift, a final certificate checks the bounds.
#include "common.h"
#include "sym_tableau.h"
#include "wave_group.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ccc_amd
{
constexpr int kZNP = 64; // horizon steps / variables per instance
constexpr double kZG = 9.80665;
constexpr double kZInf = __builtin_huge_val();

struct ZParams
{
  int N;
  double mass, dt, w_pos, w_force, fmin, fmax;
};

struct ZBatch
{
  const int * contact;  // [n][N]
  const double * ref;   // [n][N]
  const double * x0;    // [n][2]
  double * force;       // [n]
  double * force_all;   // [n][N] per step, or null
  int * status;         // [n] or null
};

// (min over the wavefront, lowest lane attaining it; 64 if no finite candidate)
__device__ __forceinline__ void z_wave_argmin(double v, double & vmin, int & imin)
{
  vmin = WaveGroup<64>::min(v);
  imin = WaveGroup<64>::first(v == vmin && v < kZInf);
}

template<int NR, int TPT>
__global__ __launch_bounds__(kZNP) void z_plan_kernel(ZParams P, ZBatch B, long n, const int * __restrict__ redo_list,
                                                      const int * __restrict__ redo_count)
{
  constexpr int NP = kZNP;
  // sweep tableau: symmetric, packed in LDS as the 4 x 4 tiles of its lower triangle (sym_tableau.h), NR = N rounded up
  // to 8 rows: lane t updates the tiles t, t + 64, ...; at N = 40 the tableau takes 7 KB (full storage: 13 KB; padded
  // to 64 x 64: 32 KB).  Padding rows/columns hold the identity and stay so under the sweeps.
  using ST = SymTab<NR, 4, TPT>;
  static_assert(ST::NT == kZNP, "one wavefront per instance");
  __shared__ __attribute__((aligned(16))) double cb[NP];
  __shared__ double res[NP];
  __shared__ int svar[NP];
  __shared__ __attribute__((aligned(16))) double T[ST::kDoubles];
  const int i = threadIdx.x;
  const int N = P.N;
  const bool col = i < NR;
  int ta[TPT], tb[TPT]; // this lane's tiles: i, i + 64, ...
#pragma unroll
  for(int k = 0; k < TPT; ++k) ST::tile_of(i + k * 64 < ST::NTILE ? i + k * 64 : 0, ta[k], tb[k]);
  const double c = P.dt * P.dt / P.mass;

  // redo_list: the instances the streaming kernel below could not finish (normally none); otherwise the whole batch
  const long nwork = redo_list ? (long)*redo_count : n;
  for(long q = blockIdx.x; q < nwork; q += gridDim.x)
  {
    const long b = redo_list ? (long)redo_list[q] : q;
    __syncthreads();
    const bool ct = i < N && B.contact[b * N + i] != 0;
    const unsigned long long mask = __ballot(ct);
    const int nv = __popcll(mask);
    if(!(mask & 1ull)) // src/LinearMpcZ.cpp:54-57
    {
      if(i == 0)
      {
        B.force[b] = 0.0;
        if(B.status) B.status[b] = CCC_STATUS_SOLVED;
      }
      if(B.force_all && i < N) B.force_all[b * N + i] = 0.0;
      continue;
    }
    const int myvar = __popcll(mask & ((1ull << i) - 1ull)); // variable index of step i (if in contact)
    if(ct) svar[myvar] = i;
    // free response and residual of output i (height after step i)
    {
      const double z0 = B.x0[b * 2 + 0], v0 = B.x0[b * 2 + 1];
      const double t = (i + 1) * P.dt;
      res[i] = (i < N) ? B.ref[b * N + i] - (z0 + t * v0 - 0.5 * kZG * t * t) : 0.0;
    }
    __syncthreads();
    const bool row = i < nv;
    const int si = row ? svar[i] : 0;
    // H (closed form) and g
    double gi = 0.0;
    // H(l, i) in closed form: w_pos c^2 sum_{j=J}^{N-1} (j - a)(j - bb) + w_force delta, J = max(s_l, s_i), with
    // S1(m) = sum_{j<m} j, S2(m) = sum_{j<m} j^2.  Cheap enough that the closing refinement re-evaluates it instead of
    // keeping an untouched copy of H in LDS (32 KB: the difference between two and four workgroups per CU).
    auto h_steps = [&](int sa, int sb, bool diag) { // entry of H between the variables of steps sa and sb
      const double a = sa - 0.5, bb = sb - 0.5;
      const double J = sa > sb ? sa : sb;
      const double Nn = N;
      const double s1 = 0.5 * (Nn * (Nn - 1.0) - J * (J - 1.0));
      const double s2 = ((Nn - 1.0) * Nn * (2.0 * Nn - 1.0) - (J - 1.0) * J * (2.0 * J - 1.0)) / 6.0;
      return P.w_pos * (c * c) * (s2 - (a + bb) * s1 + a * bb * (Nn - J)) + (diag ? P.w_force : 0.0);
    };
    auto h_entry = [&](int l, int sl) { return h_steps(si, sl, l == i); };
    {
      const double a = si - 0.5;
      // tile (ta, tb) <- H[4 ta + r][4 tb + cc] (steps of the two variables from svar), identity on the padding
#pragma unroll
      for(int k = 0; k < TPT; ++k)
      {
        const int t = i + k * 64;
        if(t < ST::NTILE)
#pragma unroll
          for(int r = 0; r < 4; ++r)
#pragma unroll
            for(int cc = 0; cc < 4; ++cc)
            {
              const int vr = 4 * ta[k] + r, vc = 4 * tb[k] + cc;
              const bool in = vr < nv && vc < nv;
              T[ST::slot_index(t, r * 4 + cc)] =
                  in ? h_steps(svar[vr < nv ? vr : 0], svar[vc < nv ? vc : 0], vr == vc) : (vr == vc ? 1.0 : 0.0);
            }
      }
      if(row)
      {
        double s = 0.0;
        for(int j = si; j < N; j++) s = fma(c * (j - a), res[j], s);
        gi = -P.w_pos * s;
      }
    }
    __syncthreads();
    // T <- -H^-1 by sweeping every variable (only the tiles of the first ceil(nv / 4) tile rows take part)
    const int nbv = (nv + 3) >> 2, nta = (nbv * (nbv + 1)) >> 1;
    for(int kp = 0; kp < nv; kp++)
    {
      const double v = col ? T[ST::entry(kp, i)] : 0.0;
      cb[i] = v;
      __syncthreads();
      const double rp = 1.0 / cb[kp];
      const double g = v * rp;
#pragma unroll
      for(int k = 0; k < TPT; ++k)
        if(i + k * 64 < nta) ST::update_tile(T, cb, rp, i + k * 64, ta[k], tb[k]);
      __syncthreads();
      if(row) T[ST::entry(kp, i)] = (i == kp) ? -rp : g; // row/column kp (the update left noise there), the pivot
      __syncthreads();
    }
    // unconstrained minimiser lam0 = -H^-1 g, then G = H^-1 = -T
    cb[i] = row ? gi : 0.0;
    __syncthreads();
    const double lam0 = row ? ST::matvec_row(T, cb, i) : 0.0;
    __syncthreads();
    {
      double2 * T2 = reinterpret_cast<double2 *>(T);
#pragma unroll
      for(int k = 0; k < TPT; ++k)
        if(i + k * 64 < ST::NTILE)
#pragma unroll
          for(int sl = 0; sl < ST::SL; ++sl)
          {
            const double2 x = T2[sl * ST::NTP + i + k * 64];
            T2[sl * ST::NTP + i + k * 64] = make_double2(-x.x, -x.y);
          }
    }
    __syncthreads();
    const double lo = row ? P.fmin - lam0 : -kZInf;
    const double hi = row ? P.fmax - lam0 : kZInf;
    const double tl = row ? 1e-12 * (1.0 + fabs(lo)) : 0.0;
    const double th = row ? 1e-12 * (1.0 + fabs(hi)) : 0.0;
    int st = CCC_STATUS_SOLVED;

    double z = 0.0, mu = 0.0;
    bool inW = false;
    int p = 0;
    double psig = 0.0, pd = 0.0;
    bool done = false;
    bool need_select = true;
    int passes = 0;
    const int maxpass = 20 * nv + 100;

    for(int round = 0; round < 3 && !done; ++round)
    {
      while(!done)
      {
        if(need_select)
        {
          const double sl = (lo - z) - tl, sh = (z - hi) - th;
          const double score = (inW || !row) ? -kZInf : fmax(sl, sh);
          double m;
          int cand;
          z_wave_argmin(score > 0.0 ? -score : kZInf, m, cand);
          if(cand >= NP) break;
          p = cand;
          if(i == cand)
          {
            psig = (sl >= sh) ? 1.0 : -1.0;
            pd = (sl >= sh) ? lo : hi;
          }
        }
        const double sig = __shfl(psig, p);
        const double cc = col ? T[ST::entry(p, i)] : 0.0; // column p = row p (symmetric)
        const double dm = -sig * cc;
        const bool blocking = inW && ((mu > 0.0 && dm < 0.0) || (mu < 0.0 && dm > 0.0));
        const bool isp = (i == p);
        const double num = isp ? psig * (pd - z) : -mu;
        const double den = isp ? cc : dm;
        double ratio = (isp || blocking) ? num / den : kZInf;
        if(isp && !(cc > 0.0)) ratio = kZInf;
        double t;
        int kk;
        z_wave_argmin(ratio, t, kk);
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
          z = fma(sig * t, cc, z);
        if(isp) mu += sig * t;
        // pivot on row/column kk
        const double v = col ? T[ST::entry(kk, i)] : 0.0;
        __syncthreads();
        cb[i] = v;
        __syncthreads();
        const double rp = 1.0 / cb[kk];
        const double g = v * rp;
#pragma unroll
        for(int k = 0; k < TPT; ++k)
          if(i + k * 64 < nta) ST::update_tile(T, cb, rp, i + k * 64, ta[k], tb[k]);
        __syncthreads();
        if(row) T[ST::entry(kk, i)] = (i == kk) ? -rp : s * g;
        __syncthreads();
        if(isadd)
        {
          if(isp)
          {
            inW = true;
            z = pd;
          }
          need_select = true;
        }
        else
        {
          if(i == kk)
          {
            inW = false;
            mu = 0.0;
          }
          need_select = false;
        }
        if(++passes > maxpass)
        {
          st = CCC_STATUS_MAX_ITER;
          done = true;
        }
      }
      if(st != CCC_STATUS_SOLVED) break;
      // closing primal refinement with the untouched H: r = -(H lambda + g) on the free variables,
      // lambda_F += (H_FF)^-1 r -- on the tableau swept on W the free-free block IS (H_FF)^-1
      __syncthreads();
      cb[i] = row ? lam0 + z : 0.0;
      __syncthreads();
      double hl = gi;
      for(int j = 0; j < nv; ++j) hl = fma(h_entry(j, __builtin_amdgcn_readlane(si, j)), cb[j], hl);
      const double r = (row && !inW) ? -hl : 0.0;
      __syncthreads();
      cb[i] = r;
      __syncthreads();
      if(row && !inW) z += ST::matvec_row(T, cb, i);
      const double sl = (lo - z) - tl, sh = (z - hi) - th;
      const int reopen = __syncthreads_or(row && !inW && fmax(sl, sh) > 0.0);
      need_select = true;
      if(!reopen) break;
    }
    // certificate (also catches NaN): every force inside its bounds
    {
      const double lam = lam0 + z;
      const bool bad = row && !((P.fmin - lam) <= 1e-9 * (1.0 + fabs(P.fmin)) && (lam - P.fmax) <= 1e-9 * (1.0 + fabs(P.fmax)));
      if(__syncthreads_or(bad ? 1 : 0) && st == CCC_STATUS_SOLVED) st = CCC_STATUS_INFEASIBLE;
    }
    // outputs: variable 0 is the force of step 0 (src/LinearMpcZ.cpp:93)
    if(i == 0)
    {
      B.force[b] = lam0 + z;
      if(B.status) B.status[b] = (passes << 8) | st;
    }
    if(B.force_all)
    {
      if(i < N) B.force_all[b * N + i] = 0.0;
      __syncthreads();
      if(row) B.force_all[b * N + si] = lam0 + z;
    }
  }
}
// ---------------------------------------------------------------------------------------------------------------
// The same QP in O(N) per iteration, ONE INSTANCE PER LANE.  With the state x = [z, zdot] the problem is a box-constrained
// linear-quadratic tracking problem,
//     min  sum_j  w_pos/2 (z_{j+1} - ref_j)^2 + w_force/2 f_j^2 ,   x_{j+1} = A x_j + B f_j + e  (f_j = 0 without contact),
//     fmin <= f_j <= fmax                                             (H = w_pos B^'B^ + w_force I of src/LinearMpcZ.cpp:84-86)
// so a Newton step on the free variables is a Riccati sweep, not a dense factorisation.  Projected Newton (the scheme of
// the box-QP inside the DDP solver, with the Riccati recursion in place of the Cholesky factor):
//   backward sweep  -- costate -> gradient -> clamped set {f at a bound, gradient pushing outward}; Riccati with the
//                      clamped / contact-free steps as known inputs -> gains (K_j, k_j); the state trajectory is walked
//                      BACKWARDS with the inverse dynamics (A is unimodular), nothing of it is stored;
//   forward sweep   -- Newton candidate f+ = K x + k, projected step f(a) = clamp(f + a (f+ - f)) simulated alongside,
//                      accepted when the cost does not increase (a = 1 almost always);
//   converged       -- when a full step left the clamped set unchanged: f is then the exact minimiser on that set and
//                      the set is consistent with the multiplier signs.
// Two or three iterations on the bench workload (the tableau kernel above: N sweeps of an N x N tableau).  Per step the
// lane keeps f (two buffers), the gains and a flag in a workspace laid out [step][instance] (coalesced); the inputs are
// transposed into it once.  Instances that do not converge in kZMaxNewton iterations (none in any test) are appended to
// a list that the tableau kernel then works off.  (The budget is a latency bound, not a convergence problem: a wavefront
// lasts as long as its slowest lane, ~1 % of the bench instances need 13..40 sweeps because their projected full steps
// must be halved several times, and the tableau kernel solves those few in parallel, one wavefront each.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kZMaxSweeps = 12; // backward + forward sweeps a lane may spend before it hands the instance over

struct ZWork
{
  double * f;         // [wavefront][N][6][64]: f (two buffers), K0, K1, k, ref
  unsigned char * fl; // [wavefront][N][64]  1 = free
  int * redo_list;    // [n]
  int * redo_count;   // [1]
};

// emit_unfinished: horizons beyond the tableau kernel's 64 steps have no second kernel to hand over to: an instance that
// uses up the (then much larger) budget writes the iterate it has, flagged CCC_STATUS_MAX_ITER.
__global__ __launch_bounds__(256) void z_plan_stream_kernel(ZParams P, ZBatch B, ZWork W, long n, int max_newton,
                                                           int emit_unfinished)
{
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(b >= n) return;
  const int N = P.N;
  // workspace layout [wavefront][step][field][lane] (six doubles per step: f x 2, K0, K1, k, ref): what a wavefront
  // touches in a sweep is one linear run of memory
  const size_t blk = (size_t)(b >> 6), ln = (size_t)(b & 63);
  auto WZ = [&](int j, int f) -> double & { return W.f[((blk * N + j) * 6 + f) * 64 + ln]; };
  auto WF = [&](int j) -> unsigned char & { return W.fl[(blk * N + j) * 64 + ln]; };
  const double dt = P.dt, im = 1.0 / P.mass;
  const double B0 = 0.5 * dt * dt * im, B1 = dt * im;    // B
  const double e0 = -kZG * (0.5 * dt * dt), e1 = -kZG * dt; // e
  // (the contact flag of a step lives in bit 1 of its flag byte in the workspace, beside the free flag in bit 0: any
  //  horizon length)
  if(B.contact[b * N] == 0) // src/LinearMpcZ.cpp:54-57
  {
    B.force[b] = 0.0;
    if(B.status) B.status[b] = CCC_STATUS_SOLVED;
    if(B.force_all)
      for(int j = 0; j < N; j++) B.force_all[b * N + j] = 0.0;
    return;
  }
  const double z0 = B.x0[b * 2 + 0], v0 = B.x0[b * 2 + 1];
  int cur = 0;
  // start: the weight wherever there is contact; its cost and final state; the reference goes into the workspace layout
  const double fstart = fmin(fmax(P.mass * kZG, P.fmin), P.fmax);
  double J = 0, zN = z0, vN = v0;
  for(int jc = 0; jc < N; jc += 8)
  {
    double rr[8];
    int cc[8];
#pragma unroll
    for(int u = 0; u < 8; u++) // (eight strided loads of each in flight)
    {
      rr[u] = jc + u < N ? B.ref[b * N + jc + u] : 0.0;
      cc[u] = jc + u < N ? B.contact[b * N + jc + u] : 0;
    }
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      const int j = jc + u;
      if(j >= N) break;
      WZ(j, 5) = rr[u];
      WF(j) = cc[u] != 0 ? 2 : 0;
      const double f = cc[u] != 0 ? fstart : 0.0;
      WZ(j, cur) = f;
      const double zn = zN + dt * vN + B0 * f + e0;
      vN = vN + B1 * f + e1;
      zN = zn;
      const double r = zN - rr[u];
      J += 0.5 * P.w_pos * (r * r) + 0.5 * P.w_force * (f * f);
    }
  }
  int st = CCC_STATUS_MAX_ITER, it = 0, sweeps = 1;
  double alpha = 1.0;
  for(it = 0; sweeps < max_newton; it++)
  {
    ++sweeps;
    // ---- backward: costate, gradient, clamped set, Riccati; the state runs backwards from x_N
    double l0 = 0, l1 = 0;            // costate d cost / d x_{j+1}
    double P00 = 0, P01 = 0, P11 = 0; // value function 1/2 x'Px + p'x at step j + 1
    double p0 = 0, p1 = 0;
    double z = zN, v = vN;
    bool changed = false;
    for(int jc = N - 1; jc >= 0; jc -= 8) // eight steps at a time: their operands are loaded before the dependent chain
    {
      double rr[8], ff[8];
      unsigned char of[8];
#pragma unroll
      for(int u = 0; u < 8; u++)
      {
        const int j = jc - u;
        rr[u] = j >= 0 ? WZ(j, 5) : 0.0;
        ff[u] = j >= 0 ? WZ(j, cur) : 0.0;
        of[u] = j >= 0 ? WF(j) : 0;
      }
#pragma unroll
      for(int u = 0; u < 8; u++)
      {
        const int j = jc - u;
        if(j < 0) break;
        const double rj = rr[u];
        const bool ct = (of[u] & 2) != 0;
        const double f = ff[u];
      l0 += P.w_pos * (z - rj);
      const double T00 = P00 + P.w_pos, T01 = P01, T11 = P11; // P~ = P + w_pos e1 e1'
      const double t0 = p0 - P.w_pos * rj, t1 = p1;            // p~
      bool fr = false;
      if(ct)
      {
        const double grad = P.w_force * f + (B0 * l0 + B1 * l1);
        fr = !((f <= P.fmin && grad > 0.0) || (f >= P.fmax && grad < 0.0));
        if(it > 0 && ((of[u] & 1) != 0) != fr) changed = true;
        WF(j) = fr ? 3 : 2;
      }
      // P~ A, A'P~A  (A = [[1, dt], [0, 1]])
      const double M00 = T00, M01 = T00 * dt + T01, M11 = T01 * dt + T11; // P~ A (M10 = T01)
      const double N00 = M00, N01 = M01, N11 = dt * M01 + M11;                       // A'(P~ A), symmetric
      if(fr)
      {
        const double pb0 = T00 * B0 + T01 * B1, pb1 = T01 * B0 + T11 * B1; // P~ B
        const double quu = P.w_force + (B0 * pb0 + B1 * pb1);
        const double qx0 = pb0, qx1 = pb0 * dt + pb1;                      // B'P~A
        const double pe0 = T00 * e0 + T01 * e1 + t0, pe1 = T01 * e0 + T11 * e1 + t1; // P~ e + p~
        const double qu = B0 * pe0 + B1 * pe1;
        const double iq = fast_rcp(quu); // (v_rcp_f64 + two Newton steps: 9 dependent instructions instead of the 14 of a division)
        const double K0 = -qx0 * iq, K1 = -qx1 * iq, kk = -qu * iq;
        WZ(j, 2) = K0;
        WZ(j, 3) = K1;
        WZ(j, 4) = kk;
        P00 = N00 + qx0 * K0;
        P01 = N01 + qx0 * K1;
        P11 = N11 + qx1 * K1;
        p0 = pe0 + qx0 * kk;
        p1 = dt * pe0 + pe1 + qx1 * kk;
      }
      else
      {
        const double c0 = B0 * f + e0, c1 = B1 * f + e1; // known input (a bound, or no contact: f = 0)
        const double pe0 = T00 * c0 + T01 * c1 + t0, pe1 = T01 * c0 + T11 * c1 + t1;
        P00 = N00;
        P01 = N01;
        P11 = N11;
        p0 = pe0;
        p1 = dt * pe0 + pe1;
      }
      // costate and state one step back: lambda_j = A' lambda_{j+1};  x_j = A^-1 (x_{j+1} - B f - e)
      l1 = dt * l0 + l1;
      const double vp = v - (B1 * f + e1);
      z = z - (B0 * f + e0) - dt * vp;
      v = vp;
      }
    }
    if(it > 0 && !changed && alpha == 1.0)
    {
      st = CCC_STATUS_SOLVED;
      break;
    }
    // ---- forward: Newton candidate and projected step, simulated together
    alpha = 1.0;
    bool out_of_budget = false;
    for(;;)
    {
      if(sweeps >= max_newton)
      {
        out_of_budget = true;
        break;
      }
      ++sweeps;
      double zn_ = z0, vn_ = v0; // state under the Newton candidate
      double zc = z0, vc = v0;   // state under the projected step
      double Jc = 0;
      for(int jc = 0; jc < N; jc += 8)
      {
        double ff[8], rr[8], g0[8], g1[8], g2[8];
        unsigned char fl[8];
#pragma unroll
        for(int u = 0; u < 8; u++)
        {
          const int j = jc + u;
          const bool in = j < N;
          ff[u] = in ? WZ(j, cur) : 0.0;
          rr[u] = in ? WZ(j, 5) : 0.0;
          fl[u] = in ? WF(j) : 0;
          g0[u] = in ? WZ(j, 2) : 0.0;
          g1[u] = in ? WZ(j, 3) : 0.0;
          g2[u] = in ? WZ(j, 4) : 0.0;
        }
#pragma unroll
        for(int u = 0; u < 8; u++)
        {
          const int j = jc + u;
          if(j >= N) break;
          const double f = ff[u];
          double fn = f, fp = f;
          if(fl[u] & 2)
          {
            if(fl[u] & 1) fn = g0[u] * zn_ + g1[u] * vn_ + g2[u];
            fp = fmin(fmax(f + alpha * (fn - f), P.fmin), P.fmax);
          }
          WZ(j, cur ^ 1) = fp;
          const double zq = zn_ + dt * vn_ + B0 * fn + e0;
          vn_ = vn_ + B1 * fn + e1;
          zn_ = zq;
          const double zr = zc + dt * vc + B0 * fp + e0;
          vc = vc + B1 * fp + e1;
          zc = zr;
          const double r = zc - rr[u];
          Jc += 0.5 * P.w_pos * (r * r) + 0.5 * P.w_force * (fp * fp);
        }
      }
      if(Jc <= J + 1e-12 * fabs(J) || alpha < 1e-6)
      {
        J = Jc;
        zN = zc;
        vN = vc;
        cur ^= 1;
        break;
      }
      alpha *= 0.5;
    }
    if(out_of_budget) break;
  }
  if(st != CCC_STATUS_SOLVED && !emit_unfinished)
  {
    const int q = atomicAdd(W.redo_count, 1);
    W.redo_list[q] = (int)b;
    return; // the tableau kernel writes this instance's outputs
  }
  B.force[b] = WZ(0, cur); // step 0 (src/LinearMpcZ.cpp:93)
  if(B.status) B.status[b] = (it << 8) | st;
  if(B.force_all)
    for(int j = 0; j < N; j++) B.force_all[b * N + j] = WZ(j, cur);
}
} // namespace ccc_amd

using namespace ccc_amd;

struct ccc_z
{
  int device = 0;
  int N = 0;
  double mass = 0, dt = 0, w_pos = 1.0, w_force = 1e-7;
  int num_cu = 0;
  int64_t cap = 0;
  char * d_stage = nullptr;
  hipStream_t stream = nullptr;
  // workspace of the streaming kernel, grown to the largest batch seen
  char * ws = nullptr;
  int64_t ws_cap = 0;
  // development switches, read ONCE in ccc_z_create (never per launch)
  bool env_tableau = false, env_stream = false;
  int env_sweeps = -1; // CCC_Z_SWEEPS: the sweep budget (small values exercise the fallback); < 0: the default
};

extern "C" int ccc_z_create(double mass, double horizon_dt, int horizon_steps, double w_pos, double w_force, int device,
                            ccc_z_t ** out)
{
  if(!out) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_z_create: out is NULL");
  *out = nullptr;
  if(!(mass > 0) || !(horizon_dt > 0) || horizon_steps <= 0 || !(w_pos >= 0) || !(w_force > 0))
    return fail(CCC_ERR_INVALID_ARGUMENT,
                "ccc_z_create: mass, horizon_dt, horizon_steps, w_force must be > 0 and w_pos >= 0");
  if(horizon_steps > CCC_Z_MAX_STEPS_WIDE)
    return fail(CCC_ERR_UNSUPPORTED, "ccc_z_create: horizon_steps %d > %d is not built into this library",
                horizon_steps, CCC_Z_MAX_STEPS_WIDE);
  int rc = select_device(device);
  if(rc != CCC_OK) return rc;
  CCC_DEVICE_GUARD(device);
  ccc_z * h = new ccc_z();
  h->device = device;
  h->env_tableau = std::getenv("CCC_Z_TABLEAU") != nullptr;
  h->env_stream = std::getenv("CCC_Z_STREAM") != nullptr;
  if(const char * mi = std::getenv("CCC_Z_SWEEPS")) h->env_sweeps = std::atoi(mi);
  h->N = horizon_steps;
  h->mass = mass;
  h->dt = horizon_dt;
  h->w_pos = w_pos;
  h->w_force = w_force;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if(e != hipSuccess)
  {
    delete h;
    return fail(CCC_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  }
  h->num_cu = prop.multiProcessorCount;
  *out = h;
  return CCC_OK;
}

extern "C" void ccc_z_destroy(ccc_z_t * h)
{
  if(!h) return;
  ccc_amd::DeviceGuard ccc_device_guard__(h->device);
  if(h->d_stage) (void)hipFree(h->d_stage);
  if(h->ws) (void)hipFree(h->ws);
  if(h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int ccc_z_plan_batch_device(ccc_z_t * h, int64_t n, const int32_t * contact, const double * ref_pos,
                                       const double * x0, double * force, double * force_all, int32_t * status,
                                       void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_z_plan_batch_device: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_z_plan_batch_device: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!contact || !ref_pos || !x0 || !force)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_z_plan_batch_device: NULL contact/ref_pos/x0/force");
  CCC_DEVICE_GUARD(h->device);
  ZParams P{h->N, h->mass, h->dt, h->w_pos, h->w_force, 10.0, 10.0 * h->mass * kZG}; // src/LinearMpcZ.cpp:37
  ZBatch B{contact, ref_pos, x0, force, force_all, status};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t N = (size_t)h->N;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t n64 = ((size_t)n + 63) / 64 * 64; // whole wavefronts
  const size_t o_f = 0, o_fl = o_f + up(6 * N * n64 * 8), o_li = o_fl + up(N * n64), o_cn = o_li + up((size_t)n * 4),
               total = o_cn + 256;
  if(n > h->ws_cap) // (synchronous: not inside a captured stream)
  {
    CCC_NO_CAPTURE(stream, "ccc_z_plan_batch_device");
    if(h->ws) CCC_HIP_CHECK(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_cap = 0;
    CCC_HIP_CHECK(hipMalloc(&h->ws, total));
    h->ws_cap = n;
  }
  ZWork W{reinterpret_cast<double *>(h->ws + o_f), reinterpret_cast<unsigned char *>(h->ws + o_fl),
          reinterpret_cast<int *>(h->ws + o_li), reinterpret_cast<int *>(h->ws + o_cn)};
  // the streaming kernel has a latency floor (sweeps x steps x memory round trips: 0.3 ms at N = 40) and wants the device
  // full; below ~24 k instances at N = 40 the LDS-tableau kernel, one wavefront per instance, is faster (measured 0.05
  // against 0.31 ms at 512, 0.17 / 0.36 at 8192, 0.41 / 0.45 at 24576, 0.53 / 0.45 at 32768); its cost per instance grows
  // faster with N than the floor does, hence n N.  CCC_Z_TABLEAU / CCC_Z_STREAM force either path (development switches)
  // beyond the 64 steps of the tableau kernel: the streaming kernel alone, whatever the batch size, with a budget that
  // is a bound on the projected-Newton iteration (a descent method on a convex QP: it converges), not a hand-over point
  const bool wide = h->N > kZNP;
  const bool tableau_only = !wide && (h->env_tableau ||
                            (n * (int64_t)h->N < (int64_t)24576 * 40 && !h->env_stream && h->env_sweeps < 0));
  if(!tableau_only)
  {
    if(int zrc = zero_words(W.redo_count, 1, s)) return zrc;
    hipLaunchKernelGGL(z_plan_stream_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, P, B, W, (long)n,
                       wide ? 40 * kZMaxSweeps : (h->env_sweeps >= 0 ? h->env_sweeps : kZMaxSweeps), wide ? 1 : 0);
    CCC_HIP_CHECK(hipGetLastError());
    if(wide) return CCC_OK;
  }
  // the LDS-tableau kernel: works off the (normally empty) list of instances the streaming kernel gave up on
  const int grid = tableau_only ? (int)std::min<int64_t>(n, (int64_t)1 << 22) : (int)std::min<int64_t>(n, (int64_t)h->num_cu * 4);
  const int * rl = tableau_only ? nullptr : W.redo_list;
  const int * rc_ = tableau_only ? nullptr : W.redo_count;
  auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(kZNP), 0, s, P, B, (long)n, rl, rc_); };
  switch((h->N + 7) / 8)
  {
    case 1: case 2: go(z_plan_kernel<16, 1>); break;
    case 3: go(z_plan_kernel<24, 1>); break;
    case 4: go(z_plan_kernel<32, 1>); break;
    case 5: go(z_plan_kernel<40, 1>); break;
    case 6: go(z_plan_kernel<48, 2>); break;
    case 7: go(z_plan_kernel<56, 2>); break;
    default: go(z_plan_kernel<64, 3>); break;
  }
  CCC_HIP_CHECK(hipGetLastError());
  return CCC_OK;
}

extern "C" int ccc_z_plan_batch(ccc_z_t * h, int64_t n, const int32_t * contact, const double * ref_pos,
                                const double * x0, double * force, double * force_all, int32_t * status)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_z_plan_batch: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_z_plan_batch: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!contact || !ref_pos || !x0 || !force)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_z_plan_batch: NULL contact/ref_pos/x0/force");
  CCC_DEVICE_GUARD(h->device);
  const size_t N = (size_t)h->N;
  const size_t bc = (size_t)n * N * 4, br = (size_t)n * N * 8, bx = (size_t)n * 16, bf = (size_t)n * 8, bs = (size_t)n * 4;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t oc = 0, orf = oc + up(bc), ox = orf + up(br), of = ox + up(bx), ofa = of + up(bf), os = ofa + up(br),
               total = os + up(bs);
  if(n > h->cap)
  {
    if(h->d_stage) (void)hipFree(h->d_stage);
    h->d_stage = nullptr;
    h->cap = 0;
    CCC_HIP_CHECK(hipMalloc(&h->d_stage, total));
    h->cap = n;
  }
  if(!h->stream) CCC_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  char * d = h->d_stage;
  CCC_HIP_CHECK(hipMemcpyAsync(d + oc, contact, bc, hipMemcpyHostToDevice, h->stream));
  CCC_HIP_CHECK(hipMemcpyAsync(d + orf, ref_pos, br, hipMemcpyHostToDevice, h->stream));
  CCC_HIP_CHECK(hipMemcpyAsync(d + ox, x0, bx, hipMemcpyHostToDevice, h->stream));
  int rc = ccc_z_plan_batch_device(h, n, (const int32_t *)(d + oc), (const double *)(d + orf), (const double *)(d + ox),
                                   (double *)(d + of), force_all ? (double *)(d + ofa) : nullptr, (int32_t *)(d + os),
                                   h->stream);
  if(rc != CCC_OK) return rc;
  CCC_HIP_CHECK(hipMemcpyAsync(force, d + of, bf, hipMemcpyDeviceToHost, h->stream));
  if(force_all) CCC_HIP_CHECK(hipMemcpyAsync(force_all, d + ofa, br, hipMemcpyDeviceToHost, h->stream));
  if(status) CCC_HIP_CHECK(hipMemcpyAsync(status, d + os, bs, hipMemcpyDeviceToHost, h->stream));
  CCC_HIP_CHECK(hipStreamSynchronize(h->stream));
  return CCC_OK;
}
