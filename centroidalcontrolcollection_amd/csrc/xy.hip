// xy.hip -- batched CCC::LinearMpcXY::planOnce() on MI355X (gfx950): kernel + C-ABI.
//
// Path replaced (reference file:line under /root/reference):
//   src/LinearMpcXY.cpp:59-83      Model::Model (continuous A, B from the flattened contact ridges)
//   include/CCC/StateSpaceModel.h:170-180   ZOH discretisation -- closed form here: A^3 = 0, so
//                                  Ad = I + A dt + A^2 dt^2/2,  Bd = B dt + A B dt^2/2 + A^2 B dt^3/6
//   include/CCC/VariantSequentialExtension.h:110-208   A_seq x0 and B_seq -- never materialised (see below)
//   src/LinearMpcXY.cpp:116-182    procOnce: H = B'WB + w I, g = -B'W(ref - A x0), one equality row per contact step,
//                                  bounds [3, 3 m g], the external QP solve (:181), head(m0)
//
// Two kernels.  The default (large batches, and every problem beyond 20 steps x 16 ridges) is the STAGE-RECURSION kernel
// further down: a primal-dual active set, one instance per lane, 16 or 32 ridge slots per step, any horizon length, with
// single-change safeguard rounds.  First in this file, its fallback for <= 20 steps x 16 ridges and the path of small
// batches, the DUAL ACTIVE-SET kernel:
// One problem instance per 448-thread workgroup (7 wavefronts), thread i < 320 = variable (step i/16, ridge i%16); nothing
// but the inputs and outputs touches HBM.  The QP is solved by the Goldfarb-Idnani dual active-set iteration of the
// oracle (most violated bound enters, blocking multipliers leave), but in "stage space":
//   column i of B_seq is the response of the N six-dimensional states to the impulse b_i = Bd_s[:, r] applied at step
//   s, i.e. c_i = R_s b_i with R = sqrt(W) Phi shared by the 16 ridges of a step.  With bt_i = (b_i, rho_z,i) and F the
//   set of variables that are not clamped at a bound,
//       M_F  = w diag(I, 0) + sum_{i in F} ct_i ct_i'      (state-residual multipliers theta and equality multipliers),
//       Qt   = Rt' M_F^-1 Rt                              (7N x 7N, symmetric),
//   every quantity the iteration needs is a 7-term product with Qt:  D_ip = bt_i' Qt[s_i, s_p] bt_p gives the primal
//   direction z_i = (delta_ip - D_ip)/w of the free variables and the multiplier rates of the clamped ones, and a
//   variable changing sides is the rank-1 update Qt -+ pi pi'/(1 +- bt' pi_s), pi = Qt[:, s] bt.
// Qt lives in REGISTERS: thread t < N(N+1)/2 owns one 7x7 block (rb >= cb) of the lower triangle (98 VGPRs).
// Set-up: Qt for F = {} is V_s Phi(s, s')/w (backward Gramian recursion, sparse Ad), the 16 N variables are added by
// rank-1 updates, and the unit regularisation that keeps the equality rows non-singular meanwhile is removed again.
// The start point (equality-constrained minimiser) and the closing iterative refinement use EXACT residuals from the
// stage recursion x_{j+1} = Ad_j x_j + Bd_j lambda_j and its adjoint, with Qt as the approximate inverse.
#include "common.h"
#include "wave_group.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

// profiling switches (scripts/xy_variants.sh): number of active-set/refinement rounds, skip the set-up updates
#ifndef XY_ROUNDS
#define XY_ROUNDS 3
#endif

namespace ccc_amd
{
constexpr int kXyM = 16;
constexpr int kXyMaxN = CCC_XY_MAX_STEPS;
constexpr int kXyNV = kXyMaxN * kXyM;   // 320 variable slots: thread i < 320 = (step i/16, ridge i%16)
constexpr int kXyNT = 448;              // threads per workgroup (7 wavefronts): 2 x 210 half-blocks of Qt
constexpr int kXyWaves = kXyNT / 64;
constexpr int kXyNB = 7;                // stage block: 6 states + the step's equality row
constexpr int kXyQ = kXyMaxN * kXyNB;   // 140
constexpr double kXyG = 9.80665;
constexpr double kXyInf = __builtin_huge_val();

struct XyParams
{
  int N;
  double mass, dt;
  double w[6]; // output weight per state entry: [lmi.x, lm.x, lmi.y, lm.y, am.x, am.y]  (src/LinearMpcXY.cpp:45-57)
  double w_force;
  double flo, fhi; // force_range_
};

struct XyBatch
{
  const int * dim;
  const double *vertex, *ridge, *com_z, *total_force_z, *ref_out, *x0;
  double *u0, *lambda_all;
  int * status;
};

struct XyRed
{
  double val[kXyWaves];
  int idx[kXyWaves];
};

struct XyShared
{
  int ticket;                            // the list entry this workgroup took
  double bt[kXyNV][kXyNB];               // bt_i = (Bd_s[:, r], rho_z) of every variable
  double pi[2][kXyQ];                    // Qt[:, s] bt (double buffered: one barrier per rank-1 update)
  double part[kXyMaxN][kXyMaxN][kXyNB];  // partial products of the full Qt * gamma (refinement only)
  double V[kXyMaxN][6][6];               // Gramians V_s (set-up only)
  double gam[kXyQ];
  double c2[kXyMaxN], c3[kXyMaxN];       // sparse Ad_j: kappa dt, kappa dt^2/2 with kappa = f_z/m
  double fz[kXyMaxN];
  double ref[kXyMaxN][6];
  double beta[kXyMaxN][6];               // Bd_j lambda_j
  double adj[kXyMaxN][6];                // adjoint states
  double r2[kXyMaxN];                    // equality residuals
  double x0[6];
  double e6[kXyNB];
  double sg;
  double dsh;                            // bt_p' pi_s of the entering variable (written by its thread before the ratio test's barrier)
  int dims[kXyMaxN];
  XyRed red[2];
};

// y = Ad' x for the sparse Ad = I + {(0,1): c1, (2,3): c1, (4,2): -c2, (4,3): -c3, (5,0): c2, (5,1): c3}
// (A: src/LinearMpcXY.cpp:63-66, A^3 = 0).  The same operation right-multiplies a row by Ad.
__device__ __forceinline__ void xy_adT(double & x0, double & x1, double & x2, double & x3, double & x4, double & x5,
                                       double c1, double c2, double c3)
{
  x1 = fma(c1, x0, fma(c3, x5, x1));
  x3 = fma(c1, x2, fma(-c3, x4, x3));
  x0 = fma(c2, x5, x0);
  x2 = fma(-c2, x4, x2);
}
// y = Ad x
__device__ __forceinline__ void xy_ad(double & x0, double & x1, double & x2, double & x3, double & x4, double & x5,
                                      double c1, double c2, double c3)
{
  x4 = fma(-c2, x2, fma(-c3, x3, x4));
  x5 = fma(c2, x0, fma(c3, x1, x5));
  x0 = fma(c1, x1, x0);
  x2 = fma(c1, x3, x2);
}

__device__ __forceinline__ double xy_row16_sum(double v)
{
  v += dpp_f64<kDppQuadXor1>(v);
  v += dpp_f64<kDppQuadXor2>(v);
  v += dpp_f64<kDppRowHalfMirror>(v);
  v += dpp_f64<kDppRowMirror>(v);
  return v;
}

// (min over the block, lowest thread index attaining it; index kXyNT if every candidate is +inf/NaN).  One barrier:
// callers alternate the two reduction buffers.
__device__ __forceinline__ void xy_block_argmin(double v, XyRed * red, double & vmin, int & imin)
{
  const int tid = threadIdx.x, w = tid >> 6;
  if(tid < kXyNV) // wavefronts 5 and 6 hold blocks of Qt only, never a candidate
  {
    const double wm = WaveGroup<64>::min(v);
    const int wi = WaveGroup<64>::first(v == wm && v < kXyInf);
    if((tid & 63) == 0)
    {
      red->val[w] = wm;
      red->idx[w] = wi < 64 ? wi + 64 * w : kXyNT;
    }
  }
  else if((tid & 63) == 0)
  {
    red->val[w] = kXyInf;
    red->idx[w] = kXyNT;
  }
  __syncthreads();
  // the seven per-wavefront results, one per lane, reduced inside the first eight lanes (round 4: every lane used to
  // scan the seven entries itself -- 14 LDS reads and ~35 selects per argmin, two argmins per pivot, on all seven
  // wavefronts); same choice: the smallest value, the lowest wavefront on a tie, kXyNT when no wavefront has a candidate
  const int l = tid & 63;
  const int ia = l < kXyWaves ? red->idx[l < kXyWaves ? l : 0] : kXyNT;
  const double a = (l < kXyWaves && ia < kXyNT) ? red->val[l < kXyWaves ? l : 0] : kXyInf;
  double m = -a;
  m = max_raw(m, dpp_f64<kDppQuadXor1>(m));
  m = max_raw(m, dpp_f64<kDppQuadXor2>(m));
  m = max_raw(m, dpp_f64<kDppRowHalfMirror>(m));
  const unsigned long long hit = __ballot(ia < kXyNT && -a == m) & 0x7full;
  const int k = hit ? (int)__ffsll((long long)hit) - 1 : 0;
  vmin = -__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), 0), __builtin_amdgcn_readlane(__double2loint(m), 0));
  imin = hit ? __builtin_amdgcn_readlane(ia, k) : kXyNT;
}

// Half of one 7x7 block of the lower block triangle of Qt: rows a0 .. a0+3 (a0 = 0: rows 0-3, a0 = 4: rows 4-6 plus
// an unused one).  Lanes 2k and 2k+1 hold the two halves of block k.
struct XyHalf
{
  double q[4][kXyNB];
};

// pi = Qt[:, 7 s .. 7 s + 6] bv: the halves of the blocks in block-column s produce their rows, the two halves of a
// block in block-row s (left of the diagonal) add their partial column sums across the lane pair.
__device__ __forceinline__ void xy_col(const XyHalf & Q, bool owner, int rb, int cb, int a0, int s, const double * bv,
                                       double * pi)
{
  const bool colcase = owner && cb == s, rowcase = owner && rb == s && cb != s;
  if(colcase)
  {
    double b[kXyNB];
#pragma unroll
    for(int c = 0; c < kXyNB; c++) b[c] = bv[c];
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
      double acc = 0.0;
#pragma unroll
      for(int c = 0; c < kXyNB; c++) acc = fma(Q.q[r][c], b[c], acc);
      if(a0 + r < kXyNB) pi[kXyNB * rb + a0 + r] = acc;
    }
  }
  // (accumulated row by row: a different instruction stream from the branch above, so that the optimiser does not
  //  merge the two into one body that indexes Q dynamically and pushes it out of registers)
  if(__any(rowcase)) // wave-uniform: at most two wavefronts hold the blocks of block-row s
  {
    double acc[kXyNB];
    {
      double b[4];
#pragma unroll
      for(int r = 0; r < 4; r++) b[r] = (rowcase && a0 + r < kXyNB) ? bv[a0 + r] : 0.0;
#pragma unroll
      for(int c = 0; c < kXyNB; c++) acc[c] = Q.q[0][c] * b[0];
#pragma unroll
      for(int r = 1; r < 4; r++)
#pragma unroll
        for(int c = 0; c < kXyNB; c++) acc[c] = fma(Q.q[r][c], b[r], acc[c]);
    }
#pragma unroll
    for(int c = 0; c < kXyNB; c++) acc[c] += dpp_f64<kDppQuadXor1>(acc[c]);
    if(rowcase && a0 == 0)
    {
#pragma unroll
      for(int c = 0; c < kXyNB; c++) pi[kXyNB * cb + c] = acc[c];
    }
  }
}

// Qt -= sign * pi pi' / (1 + sign * bv' pi_s)   (sign = +1: the variable becomes free, -1: it is clamped).
// Every thread runs the update on its half block (threads without one keep a dummy; enable = false scales it to
// nothing): unconditional arithmetic keeps the entries in place in their registers.
// dsh: bv' pi_s when a thread has formed it already (the entering variable's own D of the pivot loop: one LDS read instead of
// fourteen and seven FMAs in every thread), else nullptr
__device__ __forceinline__ void xy_rank1(XyHalf & Q, bool enable, int rb, int cb, int a0, int s, const double * bv,
                                         const double * pi, double sign, const double * dsh = nullptr)
{
  double den = 1.0;
  if(dsh)
    den = fma(sign, *dsh, 1.0);
  else
  {
#pragma unroll
    for(int c = 0; c < kXyNB; c++) den = fma(sign * bv[c], pi[kXyNB * s + c], den);
  }
  const double coef = enable ? -sign * fast_rcp(den) : 0.0;
  double pr[4], pc[kXyNB];
#pragma unroll
  for(int r = 0; r < 4; r++) pr[r] = coef * pi[kXyNB * rb + (a0 + r < kXyNB ? a0 + r : 0)];
#pragma unroll
  for(int c = 0; c < kXyNB; c++) pc[c] = pi[kXyNB * cb + c];
#pragma unroll
  for(int r = 0; r < 4; r++)
#pragma unroll
    for(int c = 0; c < kXyNB; c++) Q.q[r][c] = fma(pr[r], pc[c], Q.q[r][c]);
}

__global__ __launch_bounds__(kXyNT, 4) void xy_plan_kernel(XyParams P, XyBatch B, long n, const int * __restrict__ redo_list,
                                                           const int * __restrict__ redo_count, int * ticket)
{
  constexpr int M = kXyM, NB = kXyNB;
  __shared__ XyShared sh;

  const int i = threadIdx.x;
  const int N = P.N;
  const int s_i = i >> 4, r_i = i & 15;
  // half-block ownership: threads 2k, 2k+1 hold block k = (rb, cb), cb <= rb, of the lower block triangle
  int rb = 0;
  {
    const int k = i >> 1;
    while((rb + 1) * (rb + 2) / 2 <= k) rb++;
  }
  int cb = (i >> 1) - rb * (rb + 1) / 2;
  const int a0 = (i & 1) * 4;
  const bool owner = rb < N;
  if(!owner) rb = cb = 0; // dummy block: same arithmetic, never stored
  const double c1 = P.dt;
  const double wf = P.w_force, iwf = 1.0 / P.w_force;
  if(i < NB) sh.e6[i] = (i == NB - 1) ? 1.0 : 0.0;

  // redo_list: the instances the stage-recursion kernel below handed over (normally few); otherwise the whole batch
  const long nwork = redo_list ? (long)*redo_count : n;
  // (a list is worked off through a ticket counter by a grid the size of what is resident: its entries need 150-250
  //  pivots each, and 65536 workgroups that find nothing cost 2 ms on their own)
  for(long wq = blockIdx.x;; wq += gridDim.x)
  {
    __syncthreads();
    if(ticket)
    {
      if(i == 0) sh.ticket = atomicAdd(ticket, 1);
      __syncthreads();
      wq = sh.ticket;
    }
    if(wq >= nwork) break;
    const long b = redo_list ? (long)redo_list[wq] : wq;
    // ---------------- per-step data and the variables' impulse vectors (src/LinearMpcXY.cpp:59-83, closed-form ZOH)
    if(i < N)
    {
      const double fz = B.total_force_z[b * N + i];
      const double kap = fz / P.mass;
      sh.fz[i] = fz;
      sh.c2[i] = kap * P.dt;
      sh.c3[i] = kap * P.dt * P.dt / 2;
      sh.dims[i] = min(max(B.dim[b * N + i], 0), kXyM); // (clamped to the slots there are)
    }
    for(int e = i; e < N * 6; e += kXyNT) sh.ref[e / 6][e % 6] = B.ref_out[(size_t)b * N * 6 + e];
    if(i < 6) sh.x0[i] = B.x0[b * 6 + i];
    const bool valid = s_i < N && r_i < min(max(B.dim[b * N + (s_i < N ? s_i : 0)], 0), kXyM);
    if(i < kXyNV)
    {
      double btv[NB];
#pragma unroll
      for(int c = 0; c < NB; c++) btv[c] = 0.0;
      if(valid)
      {
        const double * v = B.vertex + ((size_t)(b * N + s_i) * M + r_i) * 3;
        const double * rd = B.ridge + ((size_t)(b * N + s_i) * M + r_i) * 3;
        const double cz = B.com_z[b * N + s_i];
        const double fz = B.total_force_z[b * N + s_i] / P.mass;
        const double bc[6] = {0.0, rd[0], 0.0, rd[1], -1 * (v[2] - cz) * rd[1] + v[1] * rd[2],
                              (v[2] - cz) * rd[0] + -1 * v[0] * rd[2]};
        const double ab[6] = {bc[1], 0.0, bc[3], 0.0, -fz * bc[2], fz * bc[0]};
        const double aab[6] = {0.0, 0.0, 0.0, 0.0, -fz * bc[3], fz * bc[1]};
        const double dt = P.dt;
#pragma unroll
        for(int a = 0; a < 6; a++) btv[a] = bc[a] * dt + ab[a] * dt * dt / 2 + aab[a] * dt * dt * dt / 6;
        btv[6] = rd[2];
      }
#pragma unroll
      for(int c = 0; c < NB; c++) sh.bt[i][c] = btv[c];
    }
    __syncthreads();
    // ---------------- Gramians V_s = W + Ad_{s+1}' V_{s+1} Ad_{s+1} (thread s runs its own recursion)
    if(i < N)
    {
      double v[6][6];
#pragma unroll
      for(int a = 0; a < 6; a++)
#pragma unroll
        for(int c = 0; c < 6; c++) v[a][c] = (a == c) ? P.w[a] : 0.0;
      for(int j = N - 1; j > i; j--)
      {
        const double c2 = sh.c2[j], c3 = sh.c3[j];
#pragma unroll
        for(int a = 0; a < 6; a++) xy_adT(v[a][0], v[a][1], v[a][2], v[a][3], v[a][4], v[a][5], c1, c2, c3);
#pragma unroll
        for(int c = 0; c < 6; c++) xy_adT(v[0][c], v[1][c], v[2][c], v[3][c], v[4][c], v[5][c], c1, c2, c3);
#pragma unroll
        for(int a = 0; a < 6; a++) v[a][a] += P.w[a];
      }
#pragma unroll
      for(int a = 0; a < 6; a++)
#pragma unroll
        for(int c = 0; c < 6; c++) sh.V[i][a][c] = v[a][c];
    }
    __syncthreads();
    // ---------------- Qt for F = {}: block (s, s') = V_s Phi(s, s')/w (rows of V_s right-multiplied by Ad_s .. Ad_{s'+1});
    //                  the equality part is regularised by 1 (removed below)
    XyHalf Q;
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
      const int a = a0 + r;
#pragma unroll
      for(int c = 0; c < 6; c++) Q.q[r][c] = (a < 6) ? sh.V[rb][a < 6 ? a : 0][c] : 0.0;
      Q.q[r][6] = 0.0;
    }
    for(int j = rb; j > cb; j--)
    {
      const double c2 = sh.c2[j], c3 = sh.c3[j];
#pragma unroll
      for(int r = 0; r < 4; r++) xy_adT(Q.q[r][0], Q.q[r][1], Q.q[r][2], Q.q[r][3], Q.q[r][4], Q.q[r][5], c1, c2, c3);
    }
#pragma unroll
    for(int r = 0; r < 4; r++)
#pragma unroll
      for(int c = 0; c < 6; c++) Q.q[r][c] *= iwf;
    if(rb == cb && a0 == 4) Q.q[2][6] = 1.0;
    // ---------------- add every variable (F = all), then remove the regularisation of the contact steps' equality rows.
    //                  The m <= 16 variables of a step enter M_F only through the Gram matrix of their 7-vectors,
    //                  G_s = sum_i bt_i bt_i' (rank <= 6: the two position / momentum entries of an impulse are
    //                  proportional), so the step is added as the <= 6 columns l_j of its Cholesky factor, G_s = sum_j l_j l_j'
    //                  -- 120 rank-1 updates instead of 320, each on the CURRENT Qt like any pivot (a block update
    //                  Qt - Pi K Pi' with the whole step at once was tried: the late, nearly dependent vectors are then
    //                  formed by cancellation and multiplied by the initial 1 / w_force-sized block column -- Qt lost three
    //                  digits and the closing refinement no longer reached the KKT tolerances).
    int buf = 0;
#ifndef XY_PROF_NO_ADDS
    {
      double * const lv = &sh.part[0][0][0];             // [7][7]: the factor's columns
      int * const lflag = reinterpret_cast<int *>(lv + NB * NB); // [7]: column j is there (its pivot was not negligible)
      for(int sv = 0; sv < N; sv++)
      {
        const int m = sh.dims[sv];
        if(m == 0) continue;
        __syncthreads(); // (the previous step's columns have been read)
        if(i < 64)
        {
          // lane (a, c) of an 8 x 8 layout: G[a][c], then the outer-product Cholesky with the columns of negligible
          // pivot skipped (G is positive SEMI-definite)
          const int a = i >> 3, c = i & 7;
          const bool ok = a < NB && c < NB;
          const int aa = ok ? a : 0, cc = ok ? c : 0;
          double gg = 0.0;
          for(int v = 0; v < m; v++) gg = fma(sh.bt[sv * M + v][aa], sh.bt[sv * M + v][cc], gg);
          gg = ok ? gg : 0.0;
          double tr = (a == c) ? gg : 0.0; // trace -> the scale of the pivot test
          tr += dpp_f64<kDppQuadXor1>(tr);
          tr += dpp_f64<kDppQuadXor2>(tr);
          tr += dpp_f64<kDppRowHalfMirror>(tr);
          tr += __shfl_xor(tr, 8);
          tr += __shfl_xor(tr, 16);
          tr += __shfl_xor(tr, 32);
          const double tol = 1e-13 * tr;
#pragma unroll
          for(int j = 0; j < NB; j++)
          {
            const double piv = __shfl(gg, j * 9);
            const bool take = piv > tol;
            const double rs = take ? 1.0 / sqrt(piv) : 0.0;
            const double la = (a >= j) ? __shfl(gg, aa * 8 + j) * rs : 0.0; // l_j[a] = G[a][j] / sqrt(pivot)
            const double lc = (c >= j) ? __shfl(gg, cc * 8 + j) * rs : 0.0;
            if(ok && c == j) lv[j * NB + a] = la;
            if(i == 0) lflag[j] = take ? 1 : 0;
            gg = (a >= j && c >= j) ? fma(-la, lc, gg) : 0.0; // (rows / columns up to j are finished)
          }
        }
        __syncthreads();
        for(int j = 0; j < NB; j++)
        {
          if(!lflag[j]) continue;
          xy_col(Q, owner, rb, cb, a0, sv, lv + j * NB, sh.pi[buf]);
          __syncthreads();
          xy_rank1(Q, true, rb, cb, a0, sv, lv + j * NB, sh.pi[buf], 1.0);
          buf ^= 1;
        }
      }
      __syncthreads();
      for(int sv = 0; sv < N; sv++)
      {
        if(sh.dims[sv] == 0) continue;
        xy_col(Q, owner, rb, cb, a0, sv, sh.e6, sh.pi[buf]);
        __syncthreads();
        xy_rank1(Q, true, rb, cb, a0, sv, sh.e6, sh.pi[buf], -1.0);
        buf ^= 1;
      }
    }
#endif

    const double tl = 1e-12 * (1.0 + fabs(P.flo)), th = 1e-12 * (1.0 + fabs(P.fhi));
    double lam = 0.0, mu = 0.0;
    int stt = 0; // 0 free, -1 clamped at the lower bound, +1 at the upper bound
    int st = CCC_STATUS_SOLVED, passes = 0, rbuf = 0;
    const int maxpass = 20 * N * M + 100;
    const int ib = i < kXyNV ? i : 0; // row of sh.bt read by this thread (threads >= 320 hold no variable)

    // lambda += (approximate inverse)(exact residuals): x_{j+1} = Ad_j x_j + Bd_j lambda_j, e_j = W (x_{j+1} - ref_j),
    // adjoint p_j = e_j + Ad_{j+1}' p_{j+1}, gradient_i = w lambda_i + b_i' p_s, r2_s = f_z,s - sum rho_z lambda
    auto refine = [&]() __attribute__((always_inline)) {
      __syncthreads();
      {
        double t6[6];
#pragma unroll
        for(int a = 0; a < 6; a++) t6[a] = xy_row16_sum(sh.bt[ib][a] * lam);
        const double sz = xy_row16_sum(sh.bt[ib][6] * lam);
        if(r_i == 0 && s_i < N)
        {
#pragma unroll
          for(int a = 0; a < 6; a++) sh.beta[s_i][a] = t6[a];
          sh.r2[s_i] = sh.fz[s_i] - sz;
        }
      }
      __syncthreads();
      if(i == 0)
      {
        double x[6];
#pragma unroll
        for(int a = 0; a < 6; a++) x[a] = sh.x0[a];
        for(int j = 0; j < N; j++)
        {
          xy_ad(x[0], x[1], x[2], x[3], x[4], x[5], c1, sh.c2[j], sh.c3[j]);
#pragma unroll
          for(int a = 0; a < 6; a++)
          {
            x[a] += sh.beta[j][a];
            sh.adj[j][a] = P.w[a] * (x[a] - sh.ref[j][a]);
          }
        }
        double p[6] = {0, 0, 0, 0, 0, 0};
        for(int j = N - 1; j >= 0; j--)
        {
          if(j + 1 < N) xy_adT(p[0], p[1], p[2], p[3], p[4], p[5], c1, sh.c2[j + 1], sh.c3[j + 1]);
#pragma unroll
          for(int a = 0; a < 6; a++)
          {
            p[a] += sh.adj[j][a];
            sh.adj[j][a] = p[a];
          }
        }
      }
      __syncthreads();
      double r1 = 0.0;
      {
        double gr = wf * lam;
        if(s_i < N)
        {
#pragma unroll
          for(int a = 0; a < 6; a++) gr = fma(sh.bt[ib][a], sh.adj[s_i][a], gr);
        }
        // the gradient of a free variable is rho_z eta_s (equality multiplier) plus the residual: take the multiplier
        // part out (least squares over the step's free ridges) -- left in, it would have to cancel inside
        // r1 - bt' pi to 1e-12 relative, which the updated Qt cannot deliver
        const bool fr = valid && stt == 0;
        const double rz = sh.bt[ib][6];
        const double num = xy_row16_sum(fr ? rz * gr : 0.0), dsq = xy_row16_sum(fr ? rz * rz : 0.0);
        const double eta = dsq > 0.0 ? num / dsq : 0.0;
        r1 = fr ? -(gr - rz * eta) : 0.0;
        double t7[NB];
#pragma unroll
        for(int c = 0; c < NB; c++) t7[c] = xy_row16_sum(sh.bt[ib][c] * r1);
        if(r_i == 0 && s_i < N)
        {
          const bool has = sh.dims[s_i] > 0;
#pragma unroll
          for(int c = 0; c < NB; c++) sh.gam[NB * s_i + c] = has ? t7[c] : 0.0;
          if(has) sh.gam[NB * s_i + 6] -= wf * sh.r2[s_i];
        }
      }
      __syncthreads();
      {
        // full product Qt gamma from the half blocks: rows of the block times gamma_cb, and (left of the diagonal)
        // the transposed product, halves added across the lane pair
        double acc2[NB];
#pragma unroll
        for(int c = 0; c < NB; c++) acc2[c] = 0.0;
#pragma unroll
        for(int r = 0; r < 4; r++)
        {
          const int a = a0 + r;
          double acc = 0.0;
#pragma unroll
          for(int c = 0; c < NB; c++) acc = fma(Q.q[r][c], sh.gam[NB * cb + c], acc);
          if(owner && a < NB) sh.part[rb][cb][a] = acc;
          const double g2 = (a < NB) ? sh.gam[NB * rb + (a < NB ? a : 0)] : 0.0;
#pragma unroll
          for(int c = 0; c < NB; c++) acc2[c] = fma(Q.q[r][c], g2, acc2[c]);
        }
#pragma unroll
        for(int c = 0; c < NB; c++) acc2[c] += dpp_f64<kDppQuadXor1>(acc2[c]);
        if(owner && rb != cb && a0 == 0)
        {
#pragma unroll
          for(int c = 0; c < NB; c++) sh.part[cb][rb][c] = acc2[c];
        }
      }
      __syncthreads();
      if(i < N * NB)
      {
        double acc = 0.0;
        for(int q = 0; q < N; q++) acc += sh.part[i / NB][q][i % NB];
        sh.pi[buf][i] = acc;
      }
      __syncthreads();
      if(valid && stt == 0)
      {
        double d = 0.0;
#pragma unroll
        for(int c = 0; c < NB; c++) d = fma(sh.bt[ib][c], sh.pi[buf][NB * s_i + c], d);
        lam += (r1 - d) * iwf;
      }
      buf ^= 1;
    };

    // start point = equality-constrained minimiser (two refinement steps from 0), then rounds of
    // [active-set iteration, refinement]; a later round that changes nothing ends the solve
    for(int round = 0;; round++)
    {
      for(int rep = 0; rep < 2; rep++) refine();
      if(round >= XY_ROUNDS || st != CCC_STATUS_SOLVED) break;
      // ---------------- dual active-set iteration (oracle/qp_gi.c; the equality rows are always active).
      // One loop body = one column pi = Qt[:, s] bt of a "target" variable followed by at most one rank-1 update:
      //   target = entering variable p: directions, step length; a full step clamps p (update with this pi),
      //            a partial step only moves and makes the blocking variable kk the next target;
      //   target = blocking variable kk: it becomes free (update), then p is the target again.
      bool moved = false;
      int p = 0, target = -1;
      bool dropping = false; // uniform: the target is a blocking variable that leaves its bound
      double sg = 0.0, bound = 0.0;
      for(;;)
      {
        if(!dropping && target < 0)
        {
          double sl = 0.0, sh_ = 0.0, score = -kXyInf;
          if(i < kXyNV)
          {
            sl = (P.flo - lam) - tl;
            sh_ = (lam - P.fhi) - th;
            score = (valid && stt == 0) ? fmax(sl, sh_) : -kXyInf;
          }
          double m;
          xy_block_argmin(score > 0.0 ? -score : kXyInf, &sh.red[rbuf], m, p);
          rbuf ^= 1;
          if(p >= kXyNT) break;
          moved = true;
          sg = (sl >= sh_) ? 1.0 : -1.0;
          bound = (sl >= sh_) ? P.flo : P.fhi;
          if(i == p) sh.sg = sg;
          target = p;
        }
        const int stg = target >> 4;
        const double * btg = sh.bt[target];
        xy_col(Q, owner, rb, cb, a0, stg, btg, sh.pi[buf]);
        __syncthreads();
        double coef_sign = 1.0; // +1: target becomes free
        bool update = true;
        if(!dropping)
        {
          const bool isp = (i == p);
          double zdir = 0.0, dmu = 0.0, ratio = kXyInf;
          if(i < kXyNV) // (wave-uniform) wavefronts 5 and 6 hold no variables
          {
            const double sgp = sh.sg;
            double D = 0.0;
            if(s_i < N)
            {
#pragma unroll
              for(int c = 0; c < NB; c++) D = fma(sh.bt[ib][c], sh.pi[buf][NB * s_i + c], D);
            }
            // primal direction of the free variables, multiplier rates of the clamped ones
            if(isp) sh.dsh = D;
            zdir = isp ? sgp * (1.0 - D) * iwf : -sgp * D * iwf;
            dmu = (stt < 0) ? sgp * D : -sgp * D;
            if(isp)
            {
              const double curv = 1.0 - D;
              ratio = (curv > 1e-15) ? fabs(bound - lam) * wf * fast_rcp(curv) : kXyInf;
            }
            else if(valid && stt != 0 && dmu < 0.0)
              ratio = mu * fast_rcp(-dmu);
          }
          double t;
          int kk;
          xy_block_argmin(ratio, &sh.red[rbuf], t, kk);
          rbuf ^= 1;
          if(kk >= kXyNT || ++passes > maxpass)
          {
            st = CCC_STATUS_MAX_ITER;
            break;
          }
          const bool isadd = (kk == p);
          if(valid && stt == 0)
            lam = fma(t, zdir, lam);
          else if(valid)
            mu = fmax(fma(t, dmu, mu), 0.0);
          if(isp) mu += t;
          if(isadd)
          {
            if(isp)
            {
              stt = sg > 0.0 ? -1 : 1;
              lam = bound;
            }
            coef_sign = -1.0; // p is clamped: Qt += pi pi'/(1 - bt' pi_s)
            target = -1;      // select again
          }
          else
          {
            // the blocking variable kk leaves its bound (stays there, multiplier 0)
            if(i == kk)
            {
              stt = 0;
              mu = 0.0;
            }
            update = false;
            dropping = true;
            target = kk;
          }
        }
        else
        {
          dropping = false;
          target = p;
        }
        // (a full step clamps p: its D is in sh.dsh; a blocking variable that leaves its bound forms the denominator itself)
        if(coef_sign < 0.0)
          xy_rank1(Q, update, rb, cb, a0, stg, btg, sh.pi[buf], coef_sign, &sh.dsh);
        else
          xy_rank1(Q, update, rb, cb, a0, stg, btg, sh.pi[buf], coef_sign);
        buf ^= 1;
      }
      if(st != CCC_STATUS_SOLVED) break;
      if(round > 0 && !moved) break;
      // (next: iterative refinement with exact residuals, which removes the drift of the rank-1 updates; if it pushes
      //  a free variable across a bound, the next round picks that up)
    }

    // ---------------- certificate: bounds and equality rows hold at the returned point (also catches NaN).  An
    //                  infeasible instance (f_z that the bounded ridges cannot produce) drives M_F singular instead
    //                  of tripping the iteration limit; it must not be reported as solved.
    {
      const double sz = xy_row16_sum(valid ? sh.bt[ib][6] * lam : 0.0);
      bool bad = false;
      if(valid) bad = !(fmax(P.flo - lam, lam - P.fhi) <= 1e-7 * (1.0 + fabs(P.fhi)));
      if(s_i < N && r_i == 0 && sh.dims[s_i] > 0) bad = bad || !(fabs(sh.fz[s_i] - sz) <= 1e-7 * (1.0 + fabs(sh.fz[s_i])));
      if(__syncthreads_or(bad ? 1 : 0) && st == CCC_STATUS_SOLVED) st = CCC_STATUS_INFEASIBLE;
    }
    // ---------------- outputs (:181 head(m0)); slots beyond dim are zero
    if(s_i < N)
    {
      const double out = valid ? lam : 0.0;
      if(s_i == 0) B.u0[b * M + r_i] = out;
      if(B.lambda_all) B.lambda_all[((size_t)b * N + s_i) * M + r_i] = out;
    }
    if(i == 0 && B.status) B.status[b] = (passes << 8) | st;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same QP by a primal-dual active set whose equality-constrained solves are STAGE RECURSIONS, one instance per lane
// (default path; prototypes: tests/tools/xy_pdas_proto.py, xy_pdas_riccati_proto.py).
// In the states x_s the problem of src/LinearMpcXY.cpp:116-182 is an LQ tracking problem with box-bounded inputs (the
// force scales), input cost w_f I and one equality per contact step.  For a guess of the clamped set:
//   * stage s, next value 1/2 y'Py + p'y: with Pt = P + W, pt = p - W ref_s, S = sum_free b b', t = sum_free rho_z b,
//     alpha = sum_free rho_z^2, c = sum_clamped b lambda, d' = f_z - sum_clamped rho_z lambda, the inputs and the stage
//     multiplier eliminate in closed form (input cost w_f I => only 6 x 6 algebra):
//         lambda_F = -(B_F' pi + nu rho_F) / w_f,   nu = -(w_f d' + t'pi) / alpha,   pi = Pt y + pt,
//         (I + S'Pt / w_f) y = Ad x + c' - S'pt / w_f,   S' = S - t t'/alpha,  c' = c + t d'/alpha
//     => y = E x + f (6 x 6 solve with partial pivoting), P_s = Ad'Pt E, p_s = Ad'(Pt f + pt);
//   * forward pass: states, costates, stage multipliers, force scales of the free variables, bound multipliers
//     w_f lambda + b'pi + nu rho_z of the clamped ones -> new clamped set (out-of-bounds free variables are clamped, clamped
//     ones whose multiplier has the wrong sign are released; a contact step keeps at least one free variable);
//   * converged when the set repeats: then the iterate is the KKT point.  ~7 iterations on the bench data, where the dual
//     active set above needs 340 set-up updates + ~94 pivots of a 140 x 140 operator.
// E, f, Pt, pt, t, alpha, d' and the clamped set live in an HBM workspace laid out [stage][field][instance] (coalesced).
// Instances that do not settle within kXsMaxIt iterations (the iteration can cycle; none of the reference scenarios
// does) go onto a work list: for the dual active-set kernel above when the problem fits it (<= 20 steps of <= 16
// ridges), else for this kernel again in SINGLE-CHANGE mode -- per iteration only the most violated condition (bound
// violation of a free variable, or wrong-sign multiplier of a clamped one, on a common force scale) changes sides.  That
// is a principal pivoting method with the largest-violation rule on the strictly convex QP; it took 150-210 sweeps on
// the instances that cycle (prototype on the bench data: all converged, same answers) where the block update needs ~7.
// The kernel is a template on the ridge slots per step: 16 (one surface contact), 32 (two: double support) or 64 (up to
// four: feet and hands; src/LinearMpcXY.cpp:69-82 iterates the whole contact_list); the horizon length is a run-time value.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kXsMaxIt = 16;
// fields of a stage in the workspace: feedback E (36), f (6), P~ (upper triangle, 21), p~ (6); the clamped set's sums
// t (6), alpha, d', S (upper triangle, 21), c (6); the step's ridge count, f_z and reference (6); the clamped set
// (2 bits per ridge: 0 free, 1 at the lower bound, 2 at the upper bound)
constexpr int kXsE = 0, kXsF = 36, kXsPt = 42, kXsPv = 63, kXsT = 69, kXsAl = 75, kXsDp = 76, kXsDim = 77, kXsFz = 78,
              kXsRef = 79, kXsS = 85, kXsC = 106, kXsSt = 112, kXsSt2 = 113, // (ridges 0-15 / 16-31: each exact in a double)
              kXsPin = 114, // (6) the linear term of the value function ENTERING the stage, beside its matrix in kXsPt
              kXsFields = 120, // (120 + 16 x 6 fields x 512 B: stages start on 4 KB boundaries)
              kXsSt3 = 120, kXsSt4 = 121, // (64 ridge slots only: ridges 32-47 / 48-63)
              kXsFields64 = 128;
// fields of a stage before its ridge vectors, and 64-bit words of a saved clamped set, by ridge slots per step
constexpr int xs_fields(int M) { return M > 32 ? kXsFields64 : kXsFields; }
constexpr int kXsRidge = 6; // doubles of a ridge in the workspace (see RB in the kernel)
constexpr int xs_st_words(int M) { return M > 32 ? 2 : 1; }
// the clamped set of a stage: 2 bits per ridge
template<int M> struct XsBits { using type = unsigned long long; };
template<> struct XsBits<16> { using type = unsigned; };
template<> struct XsBits<64> { using type = unsigned __int128; };

struct XyWork
{
  double * ws;         // [N][kXsFields][n]
  double * rb;         // [N][16][7][n]: impulse vector (6) and rho_z of every ridge
  unsigned long long * st; // [N][words][n]: 2 bits per ridge (0 free, 1 at the lower bound, 2 at the upper bound)
  int * redo_list;     // [n]
  int * redo_count;    // [1]
  size_t ws_stride, rb_stride; // doubles from one wavefront's region to the next
  size_t st_stride;            // instances per stage in st
  // the kernel runs in rounds: the instances still changing their clamped set when a round's iterations are used up
  // are listed (their sets saved in st) and the next round takes them up again packed into whole wavefronts
  const int * in_list;
  const int * in_count;
  int * out_list;
  int * out_count;
  int single; // 1: single-change mode from the saved sets (the safeguard round)
  int wander; // > 0: hand an instance over when its last forward pass changed more ridges than this (see the kernel's end)
  // round 5: in_list of the FIRST round = the batch in the order of the sweeps the handle's last call spent on each instance
  // (longest first: the lanes of a wavefront stop together); resume = the listed instances bring a saved set (later rounds)
  int resume;
  int * hist; // sweeps of this call, per instance (255: handed to the dual kernel); nullptr: not kept
};

// index of (a, c), a <= c, in the row-wise packed upper triangle of a 6 x 6 matrix
__device__ __forceinline__ constexpr int xs_tri(int a, int c)
{
  return a * 6 - a * (a - 1) / 2 + (c - a);
}

// y = Ad x and v -> Ad'v for the closed-form ZOH of src/LinearMpcXY.cpp:59-83 (k2 = f_z/m dt, k3 = f_z/m dt^2/2)
__device__ __forceinline__ void xs_ad(double dt, double k2, double k3, const double (&x)[6], double (&y)[6])
{
  y[0] = x[0] + dt * x[1];
  y[1] = x[1];
  y[2] = x[2] + dt * x[3];
  y[3] = x[3];
  y[4] = x[4] - k2 * x[2] - k3 * x[3];
  y[5] = x[5] + k2 * x[0] + k3 * x[1];
}
__device__ __forceinline__ void xs_adT(double dt, double k2, double k3, const double (&v)[6], double (&o)[6])
{
  o[0] = v[0] + k2 * v[5];
  o[1] = dt * v[0] + v[1] + k3 * v[5];
  o[2] = v[2] - k2 * v[4];
  o[3] = dt * v[2] + v[3] - k3 * v[4];
  o[4] = v[4];
  o[5] = v[5];
}
// impulse vector of ridge r of a step (column of Bd) and its rho_z
__device__ __forceinline__ void xs_ridge(const XyParams & P, const double * __restrict__ v, const double * __restrict__ rd,
                                         double cz, double kap, double (&b)[6], double & az)
{
  const double bc4 = -1 * (v[2] - cz) * rd[1] + v[1] * rd[2], bc5 = (v[2] - cz) * rd[0] + -1 * v[0] * rd[2];
  const double dt = P.dt;
  b[0] = rd[0] * dt * dt / 2;
  b[1] = rd[0] * dt;
  b[2] = rd[1] * dt * dt / 2;
  b[3] = rd[1] * dt;
  b[4] = bc4 * dt + -kap * rd[1] * dt * dt * dt / 6;
  b[5] = bc5 * dt + kap * rd[0] * dt * dt * dt / 6;
  az = rd[2];
}

// one ridge's contribution to the sums over a clamped set: free (state 0) into S, t, alpha; clamped at `val` into c, d'
__device__ __forceinline__ void xs_accumulate(unsigned state, double val, const double (&bb)[6], double az, double (&S)[21],
                                              double (&t)[6], double (&c)[6], double & alpha, double & dprime)
{
  if(state == 0u)
  {
#pragma unroll
    for(int a = 0; a < 6; a++)
    {
      t[a] += bb[a] * az;
#pragma unroll
      for(int k = a; k < 6; k++) S[xs_tri(a, k)] += bb[a] * bb[k];
    }
    alpha += az * az;
  }
  else
  {
#pragma unroll
    for(int a = 0; a < 6; a++) c[a] += bb[a] * val;
    dprime -= az * val;
  }
}

constexpr int kXsLanes = 64; // instances per wavefront
constexpr int kXsRounds = 4;
constexpr const char * kXsRoundsDefault = "6,10";
// M: ridge slots per step; SINGLE: the single-change rounds (a separate instantiation: the block iteration, which nearly
// every instance finishes in, does not carry their registers)
// (Round 4, measured and dropped: half-filled wavefronts -- 32 instances per wavefront, 2048 wavefronts for config 4, two per
//  SIMD at 255 VGPRs -- run the first round in 6.16 ms against 6.00 ms: the block rounds of a full batch are bound by
//  the ~3.7 TB/s of mixed read / write workspace traffic they sustain, not by the latency of a sweep's dependent chain.)
template<int M, bool SINGLE>
__global__ __launch_bounds__(64) void xy_plan_stream_kernel(XyParams P, XyBatch B, XyWork W, long n, int it_begin, int max_it)
{
  static_assert(M == 16 || M == 32 || M == 64, "ridge slots per step");
  constexpr int kF = xs_fields(M), kStW = xs_st_words(M);
  constexpr int kXsStage = kF + M * kXsRidge; // fields of a stage + M ridges x (5 + 1 pad)
  using Bits = typename XsBits<M>::type; // 2 bits per ridge
  const long slot = (long)blockIdx.x * kXsLanes + threadIdx.x;
  if(slot >= (W.in_list ? (long)*W.in_count : n)) return;
  const long b = W.in_list ? (long)W.in_list[slot] : slot;
  const int N = P.N;
  const double wf = P.w_force, iwf = 1.0 / P.w_force, dt = P.dt;
  // workspace layout [wavefront][stage][field][lane]: everything a wavefront touches in a stage is one contiguous 47 KB
  // (+ 57 KB of ridge vectors) run of memory, stage after stage -- with [stage][field][instance] every 512-byte access
  // of a wavefront opened a DRAM page of its own
  const size_t blk = (size_t)blockIdx.x, ln = threadIdx.x;
  // (fields in pairs, [pair][lane][2]: two neighbouring fields of a lane are 16 contiguous bytes, one dwordx4 access --
  //  half the memory instructions and twice the bytes in flight per wavefront for the same vmcnt budget)
  auto FLD = [&](int s, int g) -> double & {
    return W.ws[blk * W.ws_stride + (size_t)((s * kXsStage + (g & ~1)) * kXsLanes) + ln * 2 + (g & 1)];
  };
  auto WS = [&](int s, int f) -> double & { return FLD(s, f); };
  // a ridge in the workspace: (b1, b3), (b4, b5), (rho_z, -) -- three 16-byte pairs.  The two position entries of its
  // impulse vector are not stored: b0 = rd0 dt dt / 2 = (b1 dt) / 2 and b2 = (b3 dt) / 2 are the very operations
  // xs_ridge forms them with (round 4: 6 instead of 8 doubles per ridge and sweep, a quarter of the ridge traffic)
  auto RB = [&](int s, int r, int f) -> double & { return FLD(s, kF + r * kXsRidge + f); };
  auto ridge_of = [&](const double (&v)[5], double (&bb)[6], double & az) {
    bb[1] = v[0];
    bb[3] = v[1];
    bb[0] = v[0] * dt / 2;
    bb[2] = v[1] * dt / 2;
    bb[4] = v[2];
    bb[5] = v[3];
    az = v[4];
  };
  auto load_ridge = [&](int s, int r, double (&bb)[6], double & az) {
    double v[5];
#pragma unroll
    for(int a = 0; a < 5; a++) v[a] = RB(s, r, a);
    ridge_of(v, bb, az);
  };
  // the clamped set of a stage in the workspace: 32 bits (16 ridges) per field, each exact in a double
  auto load_bits = [&](int s) -> Bits {
    Bits bits = (Bits)(unsigned)WS(s, kXsSt);
    if constexpr(M > 16) bits |= (Bits)(unsigned)WS(s, kXsSt2) << 32;
    if constexpr(M > 32) bits |= ((Bits)(unsigned)WS(s, kXsSt3) << 64) | ((Bits)(unsigned)WS(s, kXsSt4) << 96);
    return bits;
  };
  auto store_bits = [&](int s, Bits bits) {
    WS(s, kXsSt) = (double)(unsigned)bits;
    if constexpr(M > 16) WS(s, kXsSt2) = (double)(unsigned)(bits >> 32);
    if constexpr(M > 32)
    {
      WS(s, kXsSt3) = (double)(unsigned)(bits >> 64);
      WS(s, kXsSt4) = (double)(unsigned)(bits >> 96);
    }
  };
  // ... and in the list of saved sets (instances handed from round to round)
  auto saved_bits = [&](int s) -> Bits {
    Bits bits = (Bits)W.st[((size_t)s * kStW) * W.st_stride + b];
    if constexpr(M > 32) bits |= (Bits)W.st[((size_t)s * kStW + 1) * W.st_stride + b] << 64;
    return bits;
  };
  auto fold = [](Bits bits) -> unsigned long long {
    if constexpr(M > 32)
      return (unsigned long long)bits ^ ((unsigned long long)(bits >> 64) * 0x9e3779b97f4a7c15ull);
    else
      return (unsigned long long)bits;
  };
  // the impulse vectors of all ridges, once, into the coalesced layout (the instance-major inputs are read here only)
  for(int s = 0; s < N; s++)
  {
    const Bits bits0 = W.resume ? saved_bits(s) : Bits(0); // (a resumed instance: the set it had)
    store_bits(s, bits0);
    const int m = B.dim[b * N + s] < M ? (B.dim[b * N + s] > 0 ? B.dim[b * N + s] : 0) : M; // (0..M: the slots there are)
    const double fz0 = B.total_force_z[b * N + s];
    const double cz = B.com_z[b * N + s], kap = fz0 / P.mass;
    WS(s, kXsDim) = (double)m; // the step's scalars, so that the sweeps read nothing instance-major
    WS(s, kXsFz) = fz0;
#pragma unroll
    for(int a = 0; a < 6; a++) WS(s, kXsRef + a) = B.ref_out[((size_t)b * N + s) * 6 + a];
    // ... and the sums over the clamped set the backward recursion starts from (nothing clamped: S = sum b b', t =
    // sum b a_z, alpha = sum a_z^2, c = 0, d' = f_z); from then on the forward pass, which visits every ridge anyway,
    // leaves the sums of the set it chooses
    double S[21], t[6], cc[6], alpha = 0.0, dprime = fz0;
#pragma unroll
    for(int a = 0; a < 21; a++) S[a] = 0.0;
#pragma unroll
    for(int a = 0; a < 6; a++) t[a] = cc[a] = 0.0;
    for(int r = 0; r < m; r++)
    {
      double bb[6], az;
      xs_ridge(P, B.vertex + ((size_t)(b * N + s) * M + r) * 3, B.ridge + ((size_t)(b * N + s) * M + r) * 3, cz, kap, bb, az);
      RB(s, r, 0) = bb[1];
      RB(s, r, 1) = bb[3];
      RB(s, r, 2) = bb[4];
      RB(s, r, 3) = bb[5];
      RB(s, r, 4) = az;
      const unsigned st0 = (unsigned)(bits0 >> (2 * r)) & 3u;
      xs_accumulate(st0, st0 == 1u ? P.flo : P.fhi, bb, az, S, t, cc, alpha, dprime);
    }
#pragma unroll
    for(int a = 0; a < 21; a++) WS(s, kXsS + a) = S[a];
#pragma unroll
    for(int a = 0; a < 6; a++)
    {
      WS(s, kXsT + a) = t[a];
      WS(s, kXsC + a) = cc[a];
    }
    WS(s, kXsAl) = alpha;
    WS(s, kXsDp) = dprime;
  }
  double x0[6];
#pragma unroll
  for(int a = 0; a < 6; a++) x0[a] = B.x0[b * 6 + a];
  if(!SINGLE && !B.lambda_all) // (the slots of step 0 past its ridge count: written once, see the forward pass)
    for(int r = (int)WS(0, kXsDim); r < M; r++) B.u0[b * M + r] = 0.0;
  int it = 0;
  constexpr bool single = SINGLE;
  bool converged = false, cycling = false, gaveup = false;
  // single-change rounds: up to TWO changes PER STAGE and sweep (the two most violated conditions of each stage) until
  // the set returns to where it was two sweeps ago, from then on (strict) one change per sweep (the most violated of
  // all).  (Prototype, 600 instances of the bench data, 36 of them cycling, this fallback: one per stage 67 sweeps on
  // average / 91 at most, two per stage 45 / 59, three per stage no longer converges on all.)
  bool strict = false;
  // variable r (and r2, if >= 0) of stage s change sides (and, with them, `forced` is released: see below); the sums of
  // the stage are rebuilt
  auto apply_change = [&](int s, int r, unsigned ns, int r2, unsigned ns2, int forced) -> Bits {
    const int m = (int)WS(s, kXsDim);
    Bits bits = load_bits(s);
    bits = (bits & ~(Bits(3) << (2 * r))) | ((Bits)ns << (2 * r));
    if(r2 >= 0) bits = (bits & ~(Bits(3) << (2 * r2))) | ((Bits)ns2 << (2 * r2));
    if(forced >= 0) bits &= ~(Bits(3) << (2 * forced));
    double nS[21], nt[6], nc[6], nal = 0.0, ndp = WS(s, kXsFz);
#pragma unroll
    for(int a = 0; a < 21; a++) nS[a] = 0.0;
#pragma unroll
    for(int a = 0; a < 6; a++) nt[a] = nc[a] = 0.0;
    for(int q = 0; q < m; q++)
    {
      double bb[6], azq;
      load_ridge(s, q, bb, azq);
      const unsigned st = (unsigned)(bits >> (2 * q)) & 3u;
      xs_accumulate(st, st == 1u ? P.flo : P.fhi, bb, azq, nS, nt, nc, nal, ndp);
    }
#pragma unroll
    for(int a = 0; a < 21; a++) WS(s, kXsS + a) = nS[a];
#pragma unroll
    for(int a = 0; a < 6; a++)
    {
      WS(s, kXsT + a) = nt[a];
      WS(s, kXsC + a) = nc[a];
    }
    WS(s, kXsAl) = nal;
    WS(s, kXsDp) = ndp;
    store_bits(s, bits);
    return bits;
  };
  unsigned long long h1 = 0, h2 = 0, h3 = 0, h4 = 0; // hashes of the clamped sets of the last iterations
  int nchg = 0; // ridges that changed sides in the last forward pass of the block iteration
  // The value function of a stage depends on the sets of the LATER stages only: when the last forward pass changed
  // nothing beyond stage smax, the backward recursion resumes there, from the value function it stored on its way in
  // (kXsPt / kXsPin then hold it as it ENTERED the stage: the very numbers a full recursion would produce again).
  // Single-change rounds only (kResume): there the changes are few and a wavefront's lanes are few; in the block
  // iteration some lane of the 64 nearly always changes a late stage, the wavefront runs the whole loop anyway, and the
  // extra stores cost 4 % (measured).
  constexpr bool kResume = SINGLE;
  int smax = N - 1;
  for(it = it_begin; it < max_it && !converged && !cycling && !gaveup; it++)
  {
    // ---- backward recursion on the current clamped set
    double Pm[6][6], pv[6];
#pragma unroll
    for(int a = 0; a < 6; a++)
    {
      pv[a] = 0.0;
#pragma unroll
      for(int c = 0; c < 6; c++) Pm[a][c] = 0.0;
    }
    // (round 4) the fields a stage of the backward recursion reads are loaded one stage AHEAD: a sweep is a chain of
    // load -> 6 x 6 algebra -> store per stage, and the loads of stage s - 1 do not depend on what stage s computes
    // (8192 instances 4.8 -> 4.7 ms; the same for the 80 fields of a forward stage costs 232 B of scratch: 4.9 ms, not kept)
    struct BwIn
    {
      double dim, fz, ref[6], al, dp, t[6], c[6], S[21];
    };
    auto load_bw = [&](int s, BwIn & in) {
      in.dim = WS(s, kXsDim);
      in.fz = WS(s, kXsFz);
      in.al = WS(s, kXsAl);
      in.dp = WS(s, kXsDp);
#pragma unroll
      for(int a = 0; a < 6; a++)
      {
        in.ref[a] = WS(s, kXsRef + a);
        in.t[a] = WS(s, kXsT + a);
        in.c[a] = WS(s, kXsC + a);
      }
#pragma unroll
      for(int a = 0; a < 21; a++) in.S[a] = WS(s, kXsS + a);
    };
    BwIn nxt;
    if(!kResume) load_bw(N - 1, nxt);
    for(int s = N - 1; s >= 0; s--)
    {
      if(kResume && s > smax) continue;
      BwIn cur;
      if(kResume)
        load_bw(s, cur);
      else
      {
        cur = nxt;
        if(s > 0) load_bw(s - 1, nxt);
      }
      if(kResume && s == smax && s < N - 1)
      {
#pragma unroll
        for(int a = 0; a < 6; a++)
        {
          pv[a] = WS(s, kXsPin + a);
#pragma unroll
          for(int c = 0; c < 6; c++) Pm[a][c] = WS(s, kXsPt + (c >= a ? xs_tri(a, c) : xs_tri(c, a)));
        }
      }
      const int m = (int)cur.dim;
      const double fz = cur.fz;
      const double kap = fz / P.mass, k2 = kap * dt, k3 = kap * dt * dt / 2;
      double Pt[6][6], pt[6];
#pragma unroll
      for(int a = 0; a < 6; a++)
      {
        pt[a] = pv[a] - P.w[a] * cur.ref[a];
#pragma unroll
        for(int c = 0; c < 6; c++) Pt[a][c] = Pm[a][c] + (a == c ? P.w[a] : 0.0);
      }
      // the clamped set's sums, as the setup / the last forward pass left them
      double S[6][6], t[6], cc[6];
      const double alpha = cur.al, dprime = cur.dp;
#pragma unroll
      for(int a = 0; a < 6; a++)
      {
        t[a] = cur.t[a];
        cc[a] = cur.c[a];
#pragma unroll
        for(int c = 0; c < 6; c++) S[a][c] = cur.S[c >= a ? xs_tri(a, c) : xs_tri(c, a)];
      }
      if(m > 0 && alpha > 0.0)
      {
        const double ia = 1.0 / alpha;
#pragma unroll
        for(int a = 0; a < 6; a++)
        {
          cc[a] += t[a] * dprime * ia;
#pragma unroll
          for(int c = 0; c < 6; c++) S[a][c] -= t[a] * t[c] * ia;
        }
      }
      // augmented system [I + S Pt / w_f | Ad | c' - S pt / w_f]
      double A_[6][13];
#pragma unroll
      for(int a = 0; a < 6; a++)
      {
        double sp = 0.0;
#pragma unroll
        for(int c = 0; c < 6; c++)
        {
          double acc = 0.0;
#pragma unroll
          for(int k = 0; k < 6; k++) acc += S[a][k] * Pt[k][c];
          A_[a][c] = (a == c ? 1.0 : 0.0) + acc * iwf;
          sp += S[a][c] * pt[c];
        }
        A_[a][12] = cc[a] - sp * iwf;
      }
      {
        // Ad column by column: Ad e_c
#pragma unroll
        for(int c = 0; c < 6; c++)
        {
          double ec[6] = {0, 0, 0, 0, 0, 0}, col[6];
          ec[c] = 1.0;
          xs_ad(dt, k2, k3, ec, col);
#pragma unroll
          for(int a = 0; a < 6; a++) A_[a][6 + c] = col[a];
        }
      }
      // Gaussian elimination with partial pivoting; the row exchange is a chain of selects (no dynamic register index)
      double rdiag[6];
#pragma unroll
      for(int k = 0; k < 6; k++)
      {
        int pr = k;
        double best = fabs(A_[k][k]);
#pragma unroll
        for(int i = k + 1; i < 6; i++)
        {
          const double v = fabs(A_[i][k]);
          const bool tk = v > best;
          best = tk ? v : best;
          pr = tk ? i : pr;
        }
#pragma unroll
        for(int j = k; j < 13; j++)
        {
          const double ak = A_[k][j];
          double ap = ak;
#pragma unroll
          for(int i = k + 1; i < 6; i++) ap = (pr == i) ? A_[i][j] : ap;
#pragma unroll
          for(int i = k + 1; i < 6; i++) A_[i][j] = (pr == i) ? ak : A_[i][j];
          A_[k][j] = ap;
        }
        const double inv = 1.0 / A_[k][k];
        rdiag[k] = inv;
#pragma unroll
        for(int i = k + 1; i < 6; i++)
        {
          const double f = A_[i][k] * inv;
#pragma unroll
          for(int j = k + 1; j < 13; j++) A_[i][j] -= f * A_[k][j];
        }
      }
      // back substitution of the seven right-hand sides with the six reciprocals of the elimination (a division per
      // entry is ~14 VALU instructions in fp64: 42 of them were a quarter of the stage)
      double E[6][6], fv[6];
#pragma unroll
      for(int c = 0; c < 7; c++)
      {
        double xs[6];
#pragma unroll
        for(int i = 5; i >= 0; i--)
        {
          double acc = A_[i][6 + c];
#pragma unroll
          for(int j = i + 1; j < 6; j++) acc -= A_[i][j] * xs[j];
          xs[i] = acc * rdiag[i];
        }
#pragma unroll
        for(int i = 0; i < 6; i++)
        {
          if(c < 6)
            E[i][c] = xs[i];
          else
            fv[i] = xs[i];
        }
      }
      // store the stage, then the value function one step back: P = Ad'(Pt E) symmetrised, p = Ad'(Pt f + pt)
#pragma unroll
      for(int a = 0; a < 6; a++)
      {
#pragma unroll
        for(int c = 0; c < 6; c++)
        {
          WS(s, kXsE + a * 6 + c) = E[a][c];
          // (symmetric: the upper triangle.  kResume: the matrix as it entered the stage, P~ = this + diag(w))
          if(c >= a) WS(s, kXsPt + xs_tri(a, c)) = kResume ? Pm[a][c] : Pt[a][c];
        }
        WS(s, kXsF + a) = fv[a];
        WS(s, kXsPv + a) = pt[a];
        if(kResume) WS(s, kXsPin + a) = pv[a];
      }
      double Gm[6][6], gv[6];
#pragma unroll
      for(int a = 0; a < 6; a++)
      {
        double acc2 = pt[a];
#pragma unroll
        for(int k = 0; k < 6; k++) acc2 += Pt[a][k] * fv[k];
        gv[a] = acc2;
#pragma unroll
        for(int c = 0; c < 6; c++)
        {
          double acc = 0.0;
#pragma unroll
          for(int k = 0; k < 6; k++) acc += Pt[a][k] * E[k][c];
          Gm[a][c] = acc;
        }
      }
      double Pn[6][6];
#pragma unroll
      for(int c = 0; c < 6; c++)
      {
        double colv[6], o[6];
#pragma unroll
        for(int a = 0; a < 6; a++) colv[a] = Gm[a][c];
        xs_adT(dt, k2, k3, colv, o);
#pragma unroll
        for(int a = 0; a < 6; a++) Pn[a][c] = o[a];
      }
#pragma unroll
      for(int a = 0; a < 6; a++)
#pragma unroll
        for(int c = 0; c < 6; c++) Pm[a][c] = 0.5 * (Pn[a][c] + Pn[c][a]);
      xs_adT(dt, k2, k3, gv, pv);
    }
    // ---- forward pass: new clamped set (and, once it repeats, the outputs)
    //      single-change mode: the set is left as it is, the most violated condition is looked for and applied after
    //      the pass (cand_*), the sums of that one stage are rebuilt
    int smax_new = -1; // the last stage whose set this pass changes
    double cand_v = 0.0;
    int cand_s = -1, cand_r = 0, cand_forced = -1;
    unsigned cand_ns = 0u;
    const bool last_chance = single && it + 1 >= max_it; // out of budget: emit what there is, clipped, flagged
    for(int pass = 0; pass < 2; pass++)
    {
      const bool emit = pass == 1;
      if(emit && !converged && !gaveup) break;
      // (the block iteration has written the first-step force scales in its last forward pass already -- the pass that
      //  found the set unchanged computes exactly the values an extra pass would: that pass is only run for lambda_all)
      if(emit && !single && !B.lambda_all) break;
      bool changed = false;
      if(!emit && !single) nchg = 0;
      unsigned long long hh = 1469598103934665603ull;
      double x[6];
#pragma unroll
      for(int a = 0; a < 6; a++) x[a] = x0[a];
      for(int s = 0; s < N; s++)
      {
        const int m = (int)WS(s, kXsDim);
        double y[6], pi[6], tv[6];
#pragma unroll
        for(int a = 0; a < 6; a++)
        {
          double acc = WS(s, kXsF + a);
#pragma unroll
          for(int c = 0; c < 6; c++) acc += WS(s, kXsE + a * 6 + c) * x[c];
          y[a] = acc;
        }
        double tpi = 0.0;
#pragma unroll
        for(int a = 0; a < 6; a++)
        {
          double acc = WS(s, kXsPv + a);
#pragma unroll
          for(int c = 0; c < 6; c++)
            acc += (WS(s, kXsPt + (c >= a ? xs_tri(a, c) : xs_tri(c, a))) + ((kResume && a == c) ? P.w[a] : 0.0)) * y[c];
          pi[a] = acc;
          tv[a] = WS(s, kXsT + a);
          tpi += tv[a] * acc;
        }
        const double alpha = WS(s, kXsAl), dprime = WS(s, kXsDp);
        const double nu = alpha > 0.0 ? -(wf * dprime + tpi) / alpha : 0.0;
        const Bits bits = load_bits(s);
        Bits nb = bits;
        bool anyfree = false;
        const double fz = WS(s, kXsFz);
        double nS[21], nt[6], nc[6], nal = 0.0, ndp = fz; // the sums over the set this pass chooses
#pragma unroll
        for(int a = 0; a < 21; a++) nS[a] = 0.0;
#pragma unroll
        for(int a = 0; a < 6; a++) nt[a] = nc[a] = 0.0;
        double bestm = kXyInf;
        int besti = 0, nfree = 0;
        // (single-change mode) the clamped variables that most want to leave their bound: lowest multiplier at the lower
        // bound, highest at the upper
        double rel_lo_m = kXyInf, rel_hi_m = -kXyInf;
        int rel_lo_i = -1, rel_hi_i = -1;
        bool cand_here = false;
        double sc_v = 0.0, sc_v2 = 0.0; // the stage's own two most violated conditions
        int sc_r = 0, sc_r2 = -1;
        unsigned sc_ns = 0u, sc_ns2 = 0u;
        if(single) // (free variables of the stage: clamping the only one is a move only if another can be released)
          for(int r = 0; r < m; r++) nfree += ((bits >> (2 * r)) & 3ull) == 0ull ? 1 : 0;
        const bool may_clamp = nfree > 1 || m > nfree;
        for(int r0 = 0; r0 < m; r0 += 8) // eight ridges at a time: their 56 operands are in flight together
        {
          double rbv[8][5];
#pragma unroll
          for(int u = 0; u < 8; u++)
#pragma unroll
            for(int a = 0; a < 5; a++) rbv[u][a] = RB(s, r0 + u, a); // (all M slots exist; those past m are not used)
#pragma unroll
          for(int u = 0; u < 8; u++)
          {
            const int r = r0 + u;
            if(r >= m) break;
            double bb[6], az;
            ridge_of(rbv[u], bb, az);
            double bpi = nu * az;
#pragma unroll
            for(int a = 0; a < 6; a++) bpi += bb[a] * pi[a];
            const unsigned stt = (unsigned)(bits >> (2 * r)) & 3u;
            double lam, viol = 0.0;
            unsigned ns;
            if(stt == 0u)
            {
              lam = -bpi * iwf;
              ns = lam < P.flo ? 1u : (lam > P.fhi ? 2u : 0u);
              viol = !may_clamp ? 0.0 : (ns == 1u ? P.flo - lam : (ns == 2u ? lam - P.fhi : 0.0));
              if(!single)
              {
                nb = (nb & ~(Bits(3) << (2 * r))) | ((Bits)ns << (2 * r));
                anyfree = anyfree || ns == 0u;
              }
              if(emit && gaveup) lam = fmin(fmax(lam, P.flo), P.fhi);
            }
            else
            {
              lam = stt == 1u ? P.flo : P.fhi;
              const double mult = wf * lam + bpi;
              const bool release = (stt == 1u && mult < 0.0) || (stt == 2u && mult > 0.0);
              if(release && !single) nb &= ~(Bits(3) << (2 * r));
              ns = release ? 0u : stt;
              viol = release ? fabs(mult) * iwf : 0.0; // (a multiplier on the scale of a force: mult / w_f)
              anyfree = anyfree || release;
              if(fabs(mult) < bestm)
              {
                bestm = fabs(mult);
                besti = r;
              }
              if(stt == 1u && mult < rel_lo_m)
              {
                rel_lo_m = mult;
                rel_lo_i = r;
              }
              if(stt == 2u && mult > rel_hi_m)
              {
                rel_hi_m = mult;
                rel_hi_i = r;
              }
            }
            if(single && !emit && viol > sc_v2)
            {
              if(viol > sc_v)
              {
                sc_v2 = sc_v;
                sc_r2 = sc_v > 0.0 ? sc_r : -1;
                sc_ns2 = sc_ns;
                sc_v = viol;
                sc_r = r;
                sc_ns = ns;
              }
              else
              {
                sc_v2 = viol;
                sc_r2 = r;
                sc_ns2 = ns;
              }
            }
            if(single && !emit && viol > cand_v)
            {
              cand_v = viol;
              cand_s = s;
              cand_r = r;
              cand_ns = ns;
              cand_here = true;
            }
            if(emit)
            {
              if(s == 0) B.u0[b * M + r] = lam;
              if(B.lambda_all) B.lambda_all[((size_t)b * N + s) * M + r] = lam;
            }
            else if(!single)
            {
              if(s == 0 && !B.lambda_all) B.u0[b * M + r] = lam;
              xs_accumulate(ns, ns == 1u ? P.flo : P.fhi, bb, az, nS, nt, nc, nal, ndp);
            }
          }
        }
        if(emit)
          for(int r = m; r < M; r++)
          {
            if(s == 0) B.u0[b * M + r] = 0.0;
            if(B.lambda_all) B.lambda_all[((size_t)b * N + s) * M + r] = 0.0;
          }
        if(single && !emit)
        {
          // clamping the stage's only free variable: the stage equality needs one, so a clamped variable is released
          // with it -- one that can move the right way: the free variable ran into its UPPER bound, the stage needs more
          // force from a variable at its lower bound (the one whose multiplier asks for it most), and vice versa
          auto partner = [&](unsigned ns, int free_left) -> int {
            if(ns == 0u || free_left > 0) return -1;
            const int want = ns == 2u ? rel_lo_i : rel_hi_i, other = ns == 2u ? rel_hi_i : rel_lo_i;
            return want >= 0 ? want : other;
          };
          if(cand_here) cand_forced = partner(cand_ns, nfree - 1);
          Bits now = bits;
          if(!strict && sc_v > 0.0)
          {
            // free variables the stage keeps after the one or two moves (a clamp takes one away, a release adds one)
            int left = nfree + (sc_ns == 0u ? 1 : -1);
            if(sc_r2 >= 0) left += sc_ns2 == 0u ? 1 : -1;
            int forced_i = partner(sc_ns != 0u ? sc_ns : sc_ns2, left);
            if(forced_i == sc_r || forced_i == sc_r2) forced_i = -1; // (one of the moves is that release already)
            now = apply_change(s, sc_r, sc_ns, sc_r2, sc_ns2, forced_i);
            changed = true;
            smax_new = s;
          }
          hh = (hh ^ fold(now)) * 1099511628211ull;
        }
        const bool forced = !single && m > 0 && !anyfree;
        if(forced) nb &= ~(Bits(3) << (2 * besti)); // the stage equality needs a free variable
        if(!emit && !single)
        {
          if(forced) // (rare) the sums again, in ridge order, for the set with the forced release
          {
#pragma unroll
            for(int a = 0; a < 21; a++) nS[a] = 0.0;
#pragma unroll
            for(int a = 0; a < 6; a++) nt[a] = nc[a] = 0.0;
            nal = 0.0;
            ndp = fz;
            for(int r = 0; r < m; r++)
            {
              double bb[6], azr;
              load_ridge(s, r, bb, azr);
              const unsigned ns = (unsigned)(nb >> (2 * r)) & 3u;
              xs_accumulate(ns, ns == 1u ? P.flo : P.fhi, bb, azr, nS, nt, nc, nal, ndp);
            }
          }
          if(nb != bits) // (a stage whose set stays has its sums in the workspace already: the same values, not rewritten)
          {
#pragma unroll
            for(int a = 0; a < 21; a++) WS(s, kXsS + a) = nS[a];
#pragma unroll
            for(int a = 0; a < 6; a++)
            {
              WS(s, kXsT + a) = nt[a];
              WS(s, kXsC + a) = nc[a];
            }
            WS(s, kXsAl) = nal;
            WS(s, kXsDp) = ndp;
            store_bits(s, nb);
            changed = true;
            smax_new = s;
            // ridges of the stage that changed sides (2 bits each)
            const Bits df = nb ^ bits;
            const Bits any = (df | (df >> 1)) & (~Bits(0) / 3);
            if constexpr(M > 32)
              nchg += __popcll((unsigned long long)any) + __popcll((unsigned long long)(any >> 64));
            else
              nchg += __popcll((unsigned long long)any);
          }
          hh = (hh ^ fold(nb)) * 1099511628211ull;
        }
#pragma unroll
        for(int a = 0; a < 6; a++) x[a] = y[a];
      }
      if(!emit && !single)
      {
        converged = !changed;
        // back at the set of two iterations ago: a 2-cycle, hand the instance over (longer periods were looked for and
        // do not occur: the instances that never settle wander)
        cycling = !converged && hh == h2;
        h2 = h1;
        h1 = hh;
        smax = smax_new;
      }
      if(!emit && single)
      {
        converged = cand_s < 0;
        gaveup = !converged && last_chance;
        if(!converged && !gaveup && strict)
        {
          apply_change(cand_s, cand_r, cand_ns, -1, 0u, cand_forced);
          smax_new = cand_s;
        }
        smax = smax_new;
        if(!strict)
        {
          // the per-stage changes brought back a set of two, three or four sweeps ago, or have used up their share of
          // the budget (the prototype's worst case was 59 sweeps): one change per sweep from here on
          strict = !converged && (hh == h2 || hh == h3 || hh == h4 || it - it_begin >= 96);
          h4 = h3;
          h3 = h2;
          h2 = h1;
          h1 = hh;
        }
      }
    }
  }
  if(single)
  {
    // (converged or out of budget: the outputs are written either way)
    if(B.status) B.status[b] = ((it - 1) << 8) | (converged ? CCC_STATUS_SOLVED : CCC_STATUS_MAX_ITER);
    if(W.hist) W.hist[b] = 255;
    return;
  }
  if(!converged)
  {
    // the set goes with the instance: the next round, or the safeguard round, goes on from it (the dual kernel starts anew)
    for(int s = 0; s < N; s++)
    {
      const Bits bits = load_bits(s);
      W.st[((size_t)s * kStW) * W.st_stride + b] = (unsigned long long)bits;
      if constexpr(M > 32) W.st[((size_t)s * kStW + 1) * W.st_stride + b] = (unsigned long long)(bits >> 64);
    }
    // Round 4: an instance that still moves an eighth of its variables per iteration this late does not settle (prototype,
    // 300 bench instances: after ten iterations more than 30 of the 320 variables changed on 13 instances, all 13 among
    // the 16 that never settle, none among the 28 that do) -- it goes to the dual kernel now, beside the next round,
    // instead of after it (config 4: 15.3 -> 14.0 ms, 16384 instances: 7.1 -> 5.9 ms)
    const bool wandering = W.wander > 0 && nchg > W.wander;
    if(!cycling && !wandering && W.out_list)
    {
      const int q = atomicAdd(W.out_count, 1);
      W.out_list[q] = (int)b;
      return;
    }
    if(W.hist) W.hist[b] = 255;
    const int q = atomicAdd(W.redo_count, 1);
    W.redo_list[q] = (int)b;
    return;
  }
  if(B.status) B.status[b] = ((it - 1) << 8) | CCC_STATUS_SOLVED; // changes of the clamped set before it repeated
  if(W.hist) W.hist[b] = it;
}
} // namespace ccc_amd

using namespace ccc_amd;

struct ccc_xy
{
  int device = 0;
  ccc_xy_params_t prm{};
  int num_cu = 0, blocks = 0;
  int M = 16;        // ridge slots per step (params.max_ridges)
  bool wide = false; // beyond the dual active-set kernel's tables (more than 20 steps or 32 ridge slots)
  int64_t hcap = 0;
  void * d_stage = nullptr;
  hipStream_t stream = nullptr;
  // workspace of the stage-recursion kernel, grown to the largest batch seen
  char * ws = nullptr;
  int64_t ws_cap = 0;
  // the dual active-set kernel works off the first round's hand-overs beside the later rounds, on a stream of its own
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // development switches, read ONCE in ccc_xy_create (never per launch)
  bool env_safeguard = false, env_dual = false, env_stream = false;
  int env_pdas_iters = -1; // CCC_XY_PDAS_ITERS (< 0: the default cap)
  std::string env_rounds;  // CCC_XY_ROUNDS ("": the default round boundaries)
  int wander = 0;          // changed ridges per iteration beyond which a late instance counts as wandering (CCC_XY_WANDER; 0: off)
  // round 5: the sweeps every instance of the last call took, and the order of the next call of the same size made from
  // them (CCC_XY_HISTORY=0: never)
  int *hist = nullptr, *order = nullptr, *order_scratch = nullptr;
  std::vector<void *> retired; // outgrown hist / order buffers (see the launch code)
  int64_t hist_cap = 0, hist_n = -1;
  bool env_history = true;
};

extern "C" int ccc_xy_create(const ccc_xy_params_t * p, int device, ccc_xy_t ** out)
{
  if(!out || !p) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_create: NULL argument");
  *out = nullptr;
  if(!(p->mass > 0) || !(p->horizon_dt > 0) || p->horizon_steps <= 0)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_create: mass, horizon_dt, horizon_steps must be > 0");
  if(p->max_ridges != 0 && p->max_ridges != CCC_XY_MAX_RIDGES && p->max_ridges != CCC_XY_MAX_RIDGES_WIDE
     && p->max_ridges != CCC_XY_MAX_RIDGES_MULTI)
    return fail(CCC_ERR_UNSUPPORTED, "ccc_xy_create: max_ridges = %d, the kernels are built for %d, %d and %d", p->max_ridges,
                CCC_XY_MAX_RIDGES, CCC_XY_MAX_RIDGES_WIDE, CCC_XY_MAX_RIDGES_MULTI);
  if(p->horizon_steps > CCC_XY_MAX_STEPS_WIDE)
    return fail(CCC_ERR_UNSUPPORTED, "ccc_xy_create: horizon_steps %d > %d", p->horizon_steps, CCC_XY_MAX_STEPS_WIDE);
  int rc = select_device(device);
  if(rc != CCC_OK) return rc;
  CCC_DEVICE_GUARD(device);
  ccc_xy * h = new ccc_xy();
  h->device = device;
  h->prm = *p;
  h->M = p->max_ridges ? p->max_ridges : CCC_XY_MAX_RIDGES;
  h->prm.max_ridges = h->M;
  h->wide = h->M != kXyM || p->horizon_steps > kXyMaxN;
  h->env_safeguard = std::getenv("CCC_XY_SAFEGUARD") != nullptr;
  h->env_dual = std::getenv("CCC_XY_DUAL") != nullptr;
  h->env_stream = std::getenv("CCC_XY_STREAM") != nullptr;
  if(const char * mi = std::getenv("CCC_XY_PDAS_ITERS")) h->env_pdas_iters = std::atoi(mi);
  if(const char * rs = std::getenv("CCC_XY_ROUNDS")) h->env_rounds = rs;
  if(const char * e = std::getenv("CCC_XY_HISTORY")) h->env_history = std::atoi(e) != 0;
  h->wander = std::max(8, p->horizon_steps * h->M / 8);
  if(const char * wv = std::getenv("CCC_XY_WANDER")) h->wander = std::max(0, std::atoi(wv));
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if(e != hipSuccess)
  {
    delete h;
    return fail(CCC_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  }
  h->num_cu = prop.multiProcessorCount;
  e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
  if(e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
  if(e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
  if(e != hipSuccess)
  {
    ccc_xy_destroy(h); // frees whatever was created
    return fail(CCC_ERR_HIP, "ccc_xy_create: stream / event creation failed: %s", hipGetErrorString(e));
  }
  h->blocks = h->num_cu * 2; // two resident workgroups per CU (7 + 7 wavefronts of <= 128 VGPRs)
  *out = h;
  return CCC_OK;
}

extern "C" int ccc_xy_get_params(const ccc_xy_t * h, ccc_xy_params_t * params, int * device)
{
  if(!h || !params) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_get_params: NULL argument");
  *params = h->prm;
  if(device) *device = h->device;
  return CCC_OK;
}

extern "C" void ccc_xy_destroy(ccc_xy_t * h)
{
  if(!h) return;
  ccc_amd::DeviceGuard ccc_device_guard__(h->device);
  if(h->ws) (void)hipFree(h->ws);
  if(h->hist) (void)hipFree(h->hist);
  if(h->order) (void)hipFree(h->order);
  for(void * q : h->retired) (void)hipFree(q);
  if(h->order_scratch) (void)hipFree(h->order_scratch);
  if(h->d_stage) (void)hipFree(h->d_stage);
  if(h->stream) (void)hipStreamDestroy(h->stream);
  if(h->side) (void)hipStreamDestroy(h->side);
  if(h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if(h->ev_join) (void)hipEventDestroy(h->ev_join);
  delete h;
}

extern "C" int ccc_xy_plan_batch_device(ccc_xy_t * h, int64_t n, const int32_t * dim, const double * vertex,
                                        const double * ridge, const double * com_z, const double * total_force_z,
                                        const double * ref_out, const double * x0, double * u0, double * lambda_all,
                                        int32_t * status, void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_plan_batch_device: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_plan_batch_device: n < 0");
  if(n == 0) return CCC_OK;
  if(!dim || !vertex || !ridge || !com_z || !total_force_z || !ref_out || !x0 || !u0)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_plan_batch_device: NULL required array");
  CCC_DEVICE_GUARD(h->device);
  XyParams P;
  P.N = h->prm.horizon_steps;
  P.mass = h->prm.mass;
  P.dt = h->prm.horizon_dt;
  // src/LinearMpcXY.cpp:47-49
  P.w[0] = h->prm.w_lmi[0];
  P.w[1] = h->prm.w_lm[0];
  P.w[2] = h->prm.w_lmi[1];
  P.w[3] = h->prm.w_lm[1];
  P.w[4] = h->prm.w_am[0];
  P.w[5] = h->prm.w_am[1];
  P.w_force = h->prm.w_force;
  P.flo = 3.0; // src/LinearMpcXY.cpp:91
  P.fhi = 3.0 * h->prm.mass * kXyG;
  XyBatch B{dim, vertex, ridge, com_z, total_force_z, ref_out, x0, u0, lambda_all, status};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t N = (size_t)P.N;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t n64 = ((size_t)n + 63) / 64 * 64; // whole wavefronts
  // one region per wavefront: [stage][fields | 16 ridges x 7][lane] -- what a stage touches is one contiguous run
  const size_t kXsStage = (size_t)xs_fields(h->M) + (size_t)h->M * kXsRidge;
  // (a wavefront's region is N stages x kXsStage fields x its L instances: n64 instances take the same room whatever L)
  const size_t o_ws = 0, o_rb = 0, o_st = o_ws + up(n64 * N * kXsStage * 8),
               o_li = o_st + up(N * n64 * 8 * (size_t)xs_st_words(h->M)), o_l1 = o_li + up((size_t)n * 4), o_l2 = o_l1 + up((size_t)n * 4),
               o_l3 = o_l2 + up((size_t)n * 4), o_cn = o_l3 + (kXsRounds - 1) * up((size_t)n * 4), total = o_cn + 256;
  if(n > h->ws_cap) // (synchronous: not inside a captured stream)
  {
    CCC_NO_CAPTURE(stream, "ccc_xy_plan_batch_device");
    if(h->ws) CCC_HIP_CHECK(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_cap = 0;
    CCC_HIP_CHECK(hipMalloc(&h->ws, total));
    h->ws_cap = n;
  }
  XyWork W{reinterpret_cast<double *>(h->ws + o_ws), reinterpret_cast<double *>(h->ws + o_rb),
           reinterpret_cast<unsigned long long *>(h->ws + o_st),
           reinterpret_cast<int *>(h->ws + o_li), reinterpret_cast<int *>(h->ws + o_cn), 0, 0, n64,
           nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr};
  int * const round_list[2] = {reinterpret_cast<int *>(h->ws + o_l1), reinterpret_cast<int *>(h->ws + o_l2)};
  int * const round_count = reinterpret_cast<int *>(h->ws + o_cn) + 1; // [kXsRounds], after the redo count
  // hand-overs to the dual kernel: a list per round (the first round's is W.redo_list)
  auto redo_list_of = [&](int k) { return k ? reinterpret_cast<int *>(h->ws + o_l3 + (size_t)(k - 1) * up((size_t)n * 4)) : W.redo_list; };
  auto redo_count_of = [&](int k) { return k ? round_count + kXsRounds + (k - 1) : W.redo_count; };
  auto ticket_of = [&](int k) { return W.redo_count + 2 * kXsRounds + k; };
  const int lgrid = (int)std::min<int64_t>(n, (int64_t)h->num_cu * 4); // (two workgroups are resident per CU)
  // one instance per lane needs a batch to fill the device and has a latency floor of ~7 ms (16 dependent iterations, then
  // the dual kernel on what is left): below ~3300 instances the dual active-set kernel alone, one 448-thread workgroup
  // per instance, is faster (measured, end of round 4: 1.7 against 4.4 ms at 1024, 3.0 / 4.5 at 2048, 4.3 / 4.5 at 3072,
  // 5.6 / 4.6 at 4096); CCC_XY_DUAL / CCC_XY_STREAM force either path (development switches)
  const int grid = (int)std::min<int64_t>(n, (int64_t)1 << 22);
  // CCC_XY_SAFEGUARD (development switch): the single-change rounds instead of the dual kernel where both apply
  const bool safeguard = h->wide || h->env_safeguard;
  const bool dual_only = !safeguard && (h->env_dual || (n < 3328 && !h->env_stream && h->env_pdas_iters < 0));
  auto launch_stream = [&](const XyWork & Wk0, int it_begin, int it_end) {
    XyWork Wk = Wk0;
    Wk.ws_stride = Wk.rb_stride = N * kXsStage * (size_t)kXsLanes;
    const dim3 g((unsigned)((n + kXsLanes - 1) / kXsLanes)), b(kXsLanes);
    if(h->M == 16 && !Wk.single)
      hipLaunchKernelGGL((xy_plan_stream_kernel<16, false>), g, b, 0, s, P, B, Wk, (long)n, it_begin, it_end);
    else if(h->M == 16)
      hipLaunchKernelGGL((xy_plan_stream_kernel<16, true>), g, b, 0, s, P, B, Wk, (long)n, it_begin, it_end);
    else if(h->M == 32 && !Wk.single)
      hipLaunchKernelGGL((xy_plan_stream_kernel<32, false>), g, b, 0, s, P, B, Wk, (long)n, it_begin, it_end);
    else if(h->M == 32)
      hipLaunchKernelGGL((xy_plan_stream_kernel<32, true>), g, b, 0, s, P, B, Wk, (long)n, it_begin, it_end);
    else if(!Wk.single)
      hipLaunchKernelGGL((xy_plan_stream_kernel<64, false>), g, b, 0, s, P, B, Wk, (long)n, it_begin, it_end);
    else
      hipLaunchKernelGGL((xy_plan_stream_kernel<64, true>), g, b, 0, s, P, B, Wk, (long)n, it_begin, it_end);
  };
  // round 5: the lanes of a wavefront run in lock-step and a wavefront sweeps until its last lane's set repeats -- 2 to 7
  // sweeps on nineteen instances in twenty, so every wavefront of a batch in the caller's order runs the first round's seven.
  // A handle keeps the sweeps of its last call per instance; a call of the same size takes the instances in that order,
  // longest first: the wavefronts stop after the sweeps their instances need (4.5 on average at config 4).  A schedule
  // only: the arithmetic of an instance does not depend on its lane.
  bool ordered = false;
  int * n_word = W.redo_count + 3 * kXsRounds + 1; // (holds n for the first round's in_count when it runs from the order)
  if(!dual_only && h->env_history)
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = s && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if(!(capturing && h->hist_cap < n))
    {
      if(h->hist_cap < n)
      {
        // (outgrown buffers are RETIRED, not freed: a hipGraph captured at the smaller size still replays launches that
        //  read and write them -- ADVICE r5; they go with the handle)
        if(h->hist) h->retired.push_back(h->hist);
        if(h->order) h->retired.push_back(h->order);
        h->hist = h->order = nullptr;
        h->hist_cap = 0;
        h->hist_n = -1;
        CCC_HIP_CHECK(hipMalloc(&h->hist, (size_t)n * sizeof(int)));
        CCC_HIP_CHECK(hipMalloc(&h->order, (size_t)n * sizeof(int)));
        if(!h->order_scratch) CCC_HIP_CHECK(hipMalloc(&h->order_scratch, (size_t)kOrderScratchInts * sizeof(int)));
        h->hist_cap = n;
      }
      ordered = h->hist_n == n;
      W.hist = h->hist;
    }
  }
  if(!dual_only)
  {
    // iterations of the block iteration before what is left goes to the dual kernel (safeguard rounds: always 16).  A small
    // batch runs at the latency of its sweeps (0.33 ms each) while the dual kernel has room: fewer sweeps, more hand-overs
    // (measured, round 4: 8192 instances 5.7 -> 4.9 ms with 12, 16384 6.0 -> 5.6 ms with 14, 65536 14.0 ms with 16 against
    // 14.5 / 16.3 ms with 14 / 12).  CCC_XY_PDAS_ITERS overrides (small values exercise the work list)
    const bool small = !safeguard && n < 12288, medium = !safeguard && n < 32768;
    const int cap = h->env_pdas_iters >= 0 ? h->env_pdas_iters : (small ? 12 : (medium ? 14 : kXsMaxIt));
    // rounds of iterations (development switch CCC_XY_ROUNDS="a,b": the iteration counts at which the instances still
    // going are repacked; the instances of a wavefront need very different numbers of iterations, and a wavefront is as
    // slow as its slowest lane)
    int ends[kXsRounds], nr = 0;
    {
      std::string spec = h->env_rounds.empty() ? std::string(small && h->env_pdas_iters < 0 ? "6,9" : kXsRoundsDefault) : h->env_rounds;
      size_t pos = 0;
      while(pos < spec.size() && nr < kXsRounds - 1)
      {
        const int v = std::atoi(spec.c_str() + pos);
        if(v > (nr ? ends[nr - 1] : 0) && v < cap) ends[nr++] = v;
        pos = spec.find(',', pos);
        if(pos == std::string::npos) break;
        pos++;
      }
      ends[nr++] = cap;
    }
    if(ordered)
    {
      if(int orc = order_by_count(h->hist, (int)n, h->order, h->order_scratch, W.redo_count, 3 * kXsRounds, n_word, s)) return orc;
    }
    else if(int zrc = zero_words(W.redo_count, 3 * kXsRounds, s))
      return zrc;
    if(W.hist) h->hist_n = n;
    for(int k = 0; k < nr; k++)
    {
      XyWork Wk = W;
      Wk.in_list = k ? round_list[(k - 1) & 1] : (ordered ? h->order : nullptr);
      Wk.in_count = k ? round_count + (k - 1) : (ordered ? n_word : nullptr);
      Wk.resume = k > 0;
      Wk.out_list = k + 1 < nr ? round_list[k & 1] : nullptr;
      Wk.out_count = k + 1 < nr ? round_count + k : nullptr;
      // (safeguard: every round hands over to ONE list, worked off by the single-change round after the last)
      Wk.redo_list = redo_list_of(safeguard ? 0 : k);
      Wk.redo_count = redo_count_of(safeguard ? 0 : k);
      // (from the second round on, and only where the dual kernel takes the hand-overs)
      Wk.wander = (!safeguard && k >= 1 && k + 1 < nr) ? h->wander : 0;
      launch_stream(Wk, k ? ends[k - 1] : 0, ends[k]);
      if(k + 1 < nr && !safeguard)
      {
        // this round's hand-overs (the instances found cycling) start on the dual kernel now, beside the next round:
        // the later rounds run few wavefronts and leave most of the device idle
        CCC_HIP_CHECK(hipEventRecord(h->ev_fork, s));
        CCC_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
        hipLaunchKernelGGL(xy_plan_kernel, dim3(lgrid), dim3(kXyNT), 0, h->side, P, B, (long)n, Wk.redo_list, Wk.redo_count,
                           ticket_of(k));
      }
    }
    CCC_HIP_CHECK(hipGetLastError());
    if(safeguard)
    {
      // the instances whose block iteration cycles or wanders, from the sets they were handed over with: one change of
      // the clamped set per sweep (budget: the prototype's worst case was 213 sweeps at 20 x 16 variables; it scales
      // with the number of variables that end up at a bound)
      XyWork Ws = W;
      Ws.in_list = redo_list_of(0);
      Ws.in_count = redo_count_of(0);
      Ws.resume = 1;
      Ws.single = 1;
      const int budget = 200 + 4 * P.N * (h->M / 16) * 8;
      launch_stream(Ws, ends[nr - 1], ends[nr - 1] + budget);
      CCC_HIP_CHECK(hipGetLastError());
      return CCC_OK;
    }
    if(nr > 1)
    {
      CCC_HIP_CHECK(hipEventRecord(h->ev_join, h->side));
      hipLaunchKernelGGL(xy_plan_kernel, dim3(lgrid), dim3(kXyNT), 0, s, P, B, (long)n, redo_list_of(nr - 1), redo_count_of(nr - 1),
                         ticket_of(nr - 1));
      CCC_HIP_CHECK(hipStreamWaitEvent(s, h->ev_join, 0));
      CCC_HIP_CHECK(hipGetLastError());
      return CCC_OK;
    }
  }
  hipLaunchKernelGGL(xy_plan_kernel, dim3(dual_only ? grid : lgrid), dim3(kXyNT), 0, s, P, B, (long)n,
                     dual_only ? nullptr : W.redo_list, dual_only ? nullptr : W.redo_count, dual_only ? nullptr : ticket_of(0));
  CCC_HIP_CHECK(hipGetLastError());
  return CCC_OK;
}

extern "C" int ccc_xy_plan_batch(ccc_xy_t * h, int64_t n, const int32_t * dim, const double * vertex,
                                 const double * ridge, const double * com_z, const double * total_force_z,
                                 const double * ref_out, const double * x0, double * u0, double * lambda_all,
                                 int32_t * status)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_plan_batch: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_plan_batch: n < 0");
  if(n == 0) return CCC_OK;
  if(!dim || !vertex || !ridge || !com_z || !total_force_z || !ref_out || !x0 || !u0)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_plan_batch: NULL required array");
  CCC_DEVICE_GUARD(h->device);
  if(!h->stream) CCC_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t N = h->prm.horizon_steps, M = h->M;
  struct Seg
  {
    const void * src;
    void * dst;
    size_t bytes, off;
  };
  Seg seg[] = {{vertex, nullptr, n * N * M * 3 * 8, 0}, {ridge, nullptr, n * N * M * 3 * 8, 0},
               {com_z, nullptr, n * N * 8, 0},          {total_force_z, nullptr, n * N * 8, 0},
               {ref_out, nullptr, n * N * 6 * 8, 0},    {x0, nullptr, (size_t)n * 6 * 8, 0},
               {dim, nullptr, n * N * 4, 0},            {nullptr, u0, (size_t)n * M * 8, 0},
               {nullptr, lambda_all, n * N * M * 8, 0}, {nullptr, status, (size_t)n * 4, 0}};
  size_t total = 0;
  for(auto & s : seg)
  {
    s.off = total;
    total += (s.bytes + 255) / 256 * 256;
  }
  if((int64_t)total > h->hcap)
  {
    if(h->d_stage) (void)hipFree(h->d_stage);
    h->d_stage = nullptr;
    h->hcap = 0;
    CCC_HIP_CHECK(hipMalloc(&h->d_stage, total));
    h->hcap = (int64_t)total;
  }
  char * base = static_cast<char *>(h->d_stage);
  for(auto & s : seg)
    if(s.src) CCC_HIP_CHECK(hipMemcpyAsync(base + s.off, s.src, s.bytes, hipMemcpyHostToDevice, h->stream));
  int rc = ccc_xy_plan_batch_device(
      h, n, (const int32_t *)(base + seg[6].off), (const double *)(base + seg[0].off),
      (const double *)(base + seg[1].off), (const double *)(base + seg[2].off), (const double *)(base + seg[3].off),
      (const double *)(base + seg[4].off), (const double *)(base + seg[5].off), (double *)(base + seg[7].off),
      lambda_all ? (double *)(base + seg[8].off) : nullptr, status ? (int32_t *)(base + seg[9].off) : nullptr, h->stream);
  if(rc != CCC_OK) return rc;
  for(auto & s : seg)
    if(s.dst) CCC_HIP_CHECK(hipMemcpyAsync(s.dst, base + s.off, s.bytes, hipMemcpyDeviceToHost, h->stream));
  CCC_HIP_CHECK(hipStreamSynchronize(h->stream));
  return CCC_OK;
}
