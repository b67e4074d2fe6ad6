// ddpzmp.hip -- batched CCC::DdpZmp::planOnce() on MI355X (gfx950): kernel + C-ABI.
// ("DdpZmp (fixed 6/3 dims, unconstrained) comes for free from the DDP kernel" -- it does not reuse
// that kernel: with 6 states and 3 inputs a whole instance fits ONE LANE.)
//
// Path replaced (reference file:line under /root/reference):
//   src/DdpZmp.cpp:8-19,21-43      stateEq, runningCost, terminalCost
//   src/DdpZmp.cpp:45-72           calcStateEqDeriv
//   src/DdpZmp.cpp:86-110,126-146  calcRunningCostDeriv / calcTerminalCostDeriv (second order)
//   src/DdpZmp.cpp:156-174         planOnce, incl. the external ddp_solver_->solve (nmpc_ddp::DDPSolver<6, 3>, absent
//                                  dependency: its published algorithm as restated in oracle/ddp.c -- regularised
//                                  backward pass, Tassa's boxQP with infinite bounds for the gains, 11-step line search,
//                                  Levenberg-Marquardt schedule)
//
// Mapping: ONE INSTANCE PER LANE.  Value function (6 + 36 doubles), the step's derivatives and Q-function blocks live in
// the lane's registers as fully unrolled scalar code whose structural zeros (Fx has 13 non-zeros, Fu 5, Lxx 1) are
// compile-time masks.  The trajectories (x, u, candidate x / u), the gains (k, K) and the sampled RefData live in an HBM
// workspace laid out [wavefront][step][field][lane]: every load / store of a wavefront is 64 consecutive doubles (512 B,
// coalesced).  Unlike the wave-per-instance kernels of this library this one streams: per DDP iteration and horizon step
// a lane reads 13 + 34 and writes 21 + 9 doubles (backward + one forward pass) = 616 B against ~1.5 k flops, i.e. the
// kernel is HBM-bound by design.
//
// The arithmetic follows oracle/ddp.c + oracle/ddp_zmp.c operation by operation (same summation orders, no FMA
// contraction), except that products with structural zeros are not formed (they add +-0).
#include "common.h"

#include <cmath>
#include <cstring>

#pragma clang fp contract(off)

namespace ccc_amd
{
namespace dz
{
constexpr double kG = 9.80665;

struct Params
{
  int N;
  double mass, dt;
  double w_run_com_z, w_run_zmp, w_run_force_z, w_term_com_xy, w_term_com_z, w_term_com_vel;
  ccc_ddp_config_t cfg;
};

struct Batch
{
  const double * ref;    // [n][N+1][4]
  const double * x0;     // [n][6]
  const double * u_init; // [n][N][3] | null
  double * u_out;        // [n][N][3]
  double * x_out;        // [n][N+1][6] | null
  int * iters;           // [n] | null
  int * status;          // [n] | null
  double * cost;         // [n] | null
  double * ws;           // workspace, see Ws
};

// workspace [wavefront][step][field][lane], all fp64
struct Ws
{
  double * base;
  long n;
  int N;
  // one region per wavefront, [step][field][lane]: X[2] (6 each), U[2] (3 each), K1 (3), K2 (18), R (4) = 43 fields of
  // 512 B per step -- what a wavefront touches in a step is ONE contiguous 22 KB run, step after step (with the fields
  // laid out [field][instance] over the whole batch every 512-byte access opened a DRAM page of its own)
  static constexpr int kFields = 43, kX = 0, kU = 12, kK1 = 18, kK2 = 21, kR = 39;
  __host__ __device__ static size_t doubles_per_instance(int N)
  {
    return (size_t)(N + 1) * kFields;
  }
  __host__ __device__ static size_t instances_padded(long n) // (whole wavefronts)
  {
    return ((size_t)n + 63) / 64 * 64;
  }
  __device__ double * at(int step, int f, long inst) const
  {
    return base + (((size_t)(inst >> 6) * (N + 1) + step) * kFields + f) * 64 + (inst & 63);
  }
  __device__ double * X(int buf, int step, int f, long inst) const
  {
    return at(step, kX + buf * 6 + f, inst);
  }
  __device__ double * U(int buf, int step, int f, long inst) const
  {
    return at(step, kU + buf * 3 + f, inst);
  }
  __device__ double * K1(int step, int f, long inst) const
  {
    return at(step, kK1 + f, inst);
  }
  __device__ double * K2(int step, int f, long inst) const
  {
    return at(step, kK2 + f, inst);
  }
  __device__ double * R(int step, int f, long inst) const
  {
    return at(step, kR + f, inst);
  }
};

// structural non-zeros of Fx = I + dt D and of Fu (src/DdpZmp.cpp:53-71)
__device__ constexpr bool fx_nz(int r, int c)
{
  return r == c || (r == 0 && c == 1) || (r == 1 && (c == 0 || c == 4)) || (r == 2 && c == 3)
         || (r == 3 && (c == 2 || c == 4)) || (r == 4 && c == 5);
}
__device__ constexpr bool fu_nz(int r, int c)
{
  return (r == 1 && (c == 0 || c == 2)) || (r == 3 && (c == 1 || c == 2)) || (r == 5 && c == 2);
}

// src/DdpZmp.cpp:8-19
__device__ __forceinline__ void state_eq(const Params & P, const double * x, const double * u, const double * r,
                                         double * xn)
{
  const double den = P.mass * (x[4] - r[2]);
  double xd[6];
  xd[0] = x[1];
  xd[1] = (x[0] - u[0]) * u[2] / den;
  xd[2] = x[3];
  xd[3] = (x[2] - u[1]) * u[2] / den;
  xd[4] = x[5];
  xd[5] = u[2] / P.mass - kG;
#pragma unroll
  for(int a = 0; a < 6; a++) xn[a] = x[a] + P.dt * xd[a];
}

// src/DdpZmp.cpp:21-28
__device__ __forceinline__ double running_cost(const Params & P, const double * x, const double * u, const double * r)
{
  const double ez = x[4] - r[3], e0 = u[0] - r[0], e1 = u[1] - r[1], ef = u[2] - P.mass * kG;
  return P.w_run_com_z * 0.5 * (ez * ez) + P.w_run_zmp * 0.5 * (e0 * e0 + e1 * e1) + P.w_run_force_z * 0.5 * (ef * ef);
}

// src/DdpZmp.cpp:30-43
__device__ __forceinline__ double terminal_cost(const Params & P, const double * x, const double * r)
{
  const double e0 = x[0] - r[0], e1 = x[2] - r[1], ez = x[4] - r[3];
  return P.w_term_com_xy * 0.5 * (e0 * e0 + e1 * e1) + P.w_term_com_z * 0.5 * (ez * ez)
         + P.w_term_com_vel * 0.5 * ((x[1] * x[1] + x[3] * x[3]) + x[5] * x[5]);
}

// Tassa's boxQP with infinite bounds from x = 0 (oracle_box_qp with lo = -inf, hi = inf, n = 3): nothing is ever clamped,
// so it is a Newton step with an Armijo check, repeated until the gradient vanishes.  Returns the result code; Lf (lower
// Cholesky factor of H) and rd (reciprocals of its diagonal) are those of the first iteration.
__device__ __forceinline__ int box_qp3(const double (&H)[3][3], const double (&g)[3], double (&x)[3],
                                       double (&Lf)[3][3], double (&rd)[3])
{
  const int max_iter = 500; // nmpc_ddp BoxQP::Configuration::max_iter
  const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
  double grad[3], search[3], xc[3], tmp[3];
  int result = 0, iter = 0;
#pragma unroll
  for(int i = 0; i < 3; i++) x[i] = 0.0;
  double value = 0;
#pragma unroll
  for(int i = 0; i < 3; i++)
  {
    double s = 0;
#pragma unroll
    for(int j = 0; j < 3; j++) s += H[i][j] * x[j];
    value += x[i] * g[i] + 0.5 * x[i] * s;
  }
  double oldvalue = 0;
  for(iter = 1; iter <= max_iter; iter++)
  {
    if(result != 0) break;
    if(iter > 1 && (oldvalue - value) < min_rel_improve * fabs(oldvalue))
    {
      result = 4;
      break;
    }
    oldvalue = value;
#pragma unroll
    for(int i = 0; i < 3; i++)
    {
      double s = g[i];
#pragma unroll
      for(int j = 0; j < 3; j++) s += H[i][j] * x[j];
      grad[i] = s;
    }
    if(iter == 1)
    {
      bool ok = true;
#pragma unroll
      for(int a = 0; a < 3; a++)
      {
        double s = H[a][a];
#pragma unroll
        for(int k = 0; k < a; k++) s -= Lf[a][k] * Lf[a][k];
        if(ok && !(s > 0.0)) ok = false;
        Lf[a][a] = sqrt(s);
        rd[a] = 1.0 / Lf[a][a];
#pragma unroll
        for(int b = a + 1; b < 3; b++)
        {
          double t = H[b][a];
#pragma unroll
          for(int k = 0; k < a; k++) t -= Lf[b][k] * Lf[a][k];
          Lf[b][a] = t * rd[a];
        }
      }
      if(!ok)
      {
        result = -1;
        break;
      }
    }
    double gnorm = 0;
#pragma unroll
    for(int a = 0; a < 3; a++) gnorm += grad[a] * grad[a];
    gnorm = sqrt(gnorm);
    if(gnorm < min_grad)
    {
      result = 5;
      break;
    }
#pragma unroll
    for(int a = 0; a < 3; a++) tmp[a] = g[a];
#pragma unroll
    for(int a = 0; a < 3; a++)
    {
      double s = tmp[a];
#pragma unroll
      for(int k = 0; k < a; k++) s -= Lf[a][k] * tmp[k];
      tmp[a] = s * rd[a];
    }
#pragma unroll
    for(int a = 2; a >= 0; a--)
    {
      double s = tmp[a];
#pragma unroll
      for(int k = 2; k > a; k--) s -= Lf[k][a] * tmp[k];
      tmp[a] = s * rd[a];
    }
#pragma unroll
    for(int a = 0; a < 3; a++) search[a] = -tmp[a] - x[a];
    double sdotg = 0;
#pragma unroll
    for(int i = 0; i < 3; i++) sdotg += search[i] * grad[i];
    if(sdotg >= 0) break; // no descent direction: result stays 0
    double step = 1.0, vc = 0;
    for(;;)
    {
#pragma unroll
      for(int i = 0; i < 3; i++) xc[i] = x[i] + step * search[i]; // (clamping to +-inf is the identity)
      vc = 0;
#pragma unroll
      for(int i = 0; i < 3; i++)
      {
        double s = 0;
#pragma unroll
        for(int j = 0; j < 3; j++) s += H[i][j] * xc[j];
        vc += xc[i] * g[i] + 0.5 * xc[i] * s;
      }
      if(!((vc - oldvalue) / (step * sdotg) < armijo)) break;
      step *= step_dec;
      if(step < min_step)
      {
        result = 2;
        break;
      }
    }
#pragma unroll
    for(int i = 0; i < 3; i++) x[i] = xc[i];
    value = vc;
  }
  if(iter > max_iter && result == 0) result = 1;
  return result;
}

struct Solver
{
  const Params & P;
  const Ws & W;
  const long inst;
  int cur = 0; // buffer holding the accepted trajectory
  double lambda, dlambda;
  double dV0 = 0, dV1 = 0;
  double cost = 0, costc = 0;

  __device__ Solver(const Params & p, const Ws & w, long i) : P(p), W(w), inst(i)
  {
    lambda = p.cfg.initial_lambda;
    dlambda = p.cfg.initial_dlambda;
  }

  __device__ void increase_lambda()
  {
    dlambda = fmax(dlambda * P.cfg.lambda_factor, P.cfg.lambda_factor);
    lambda = fmax(lambda * dlambda, P.cfg.lambda_min);
  }
  __device__ void decrease_lambda()
  {
    dlambda = fmin(dlambda / P.cfg.lambda_factor, 1.0 / P.cfg.lambda_factor);
    lambda = lambda * dlambda * (lambda > P.cfg.lambda_min ? 1.0 : 0.0);
  }

  // oracle/ddp.c backward_pass with the derivatives of oracle/ddp_zmp.c evaluated in place; false: Quu + regularisation
  // not positive definite at some step
  __device__ bool backward()
  {
    const int N = P.N;
    double Vx[6], Vxx[6][6];
    {
      double x[6], r[4];
#pragma unroll
      for(int a = 0; a < 6; a++) x[a] = *W.X(cur, N, a, inst);
#pragma unroll
      for(int a = 0; a < 4; a++) r[a] = *W.R(N, a, inst);
      // src/DdpZmp.cpp:126-146
      Vx[0] = P.w_term_com_xy * (x[0] - r[0]);
      Vx[1] = P.w_term_com_vel * x[1];
      Vx[2] = P.w_term_com_xy * (x[2] - r[1]);
      Vx[3] = P.w_term_com_vel * x[3];
      Vx[4] = P.w_term_com_z * (x[4] - r[3]);
      Vx[5] = P.w_term_com_vel * x[5];
#pragma unroll
      for(int a = 0; a < 6; a++)
#pragma unroll
        for(int b = 0; b < 6; b++) Vxx[a][b] = 0.0;
      Vxx[0][0] = P.w_term_com_xy;
      Vxx[1][1] = P.w_term_com_vel;
      Vxx[2][2] = P.w_term_com_xy;
      Vxx[3][3] = P.w_term_com_vel;
      Vxx[4][4] = P.w_term_com_z;
      Vxx[5][5] = P.w_term_com_vel;
    }
    dV0 = dV1 = 0;
    bool ok = true;
    // operands of the step, loaded one step ahead
    double xs[6], us[3], rs[4];
    auto fetch = [&](int i) {
#pragma unroll
      for(int a = 0; a < 6; a++) xs[a] = *W.X(cur, i, a, inst);
#pragma unroll
      for(int a = 0; a < 3; a++) us[a] = *W.U(cur, i, a, inst);
#pragma unroll
      for(int a = 0; a < 4; a++) rs[a] = *W.R(i, a, inst);
    };
    fetch(N - 1);
    for(int i = N - 1; i >= 0; i--)
    {
      double x[6], u[3], r[4];
#pragma unroll
      for(int a = 0; a < 6; a++) x[a] = xs[a];
#pragma unroll
      for(int a = 0; a < 3; a++) u[a] = us[a];
#pragma unroll
      for(int a = 0; a < 4; a++) r[a] = rs[a];
      if(i > 0) fetch(i - 1);
      // ---- derivatives (src/DdpZmp.cpp:45-72, 86-110)
      double Fx[6][6], Fu[6][3];
      {
        const double d = x[4] - r[2];
        const double den = P.mass * d, den2 = P.mass * (d * d);
#pragma unroll
        for(int a = 0; a < 6; a++)
#pragma unroll
          for(int b = 0; b < 6; b++) Fx[a][b] = 0.0;
        Fx[0][1] = 1;
        Fx[1][0] = u[2] / den;
        Fx[1][4] = -1 * (x[0] - u[0]) * u[2] / den2;
        Fx[2][3] = 1;
        Fx[3][2] = u[2] / den;
        Fx[3][4] = -1 * (x[2] - u[1]) * u[2] / den2;
        Fx[4][5] = 1;
#pragma unroll
        for(int a = 0; a < 6; a++)
#pragma unroll
          for(int b = 0; b < 6; b++)
            if(fx_nz(a, b)) Fx[a][b] *= P.dt;
#pragma unroll
        for(int a = 0; a < 6; a++) Fx[a][a] += 1.0;
#pragma unroll
        for(int a = 0; a < 6; a++)
#pragma unroll
          for(int b = 0; b < 3; b++) Fu[a][b] = 0.0;
        Fu[1][0] = -1 * u[2] / den;
        Fu[1][2] = (x[0] - u[0]) / den;
        Fu[3][1] = -1 * u[2] / den;
        Fu[3][2] = (x[2] - u[1]) / den;
        Fu[5][2] = 1 / P.mass;
#pragma unroll
        for(int a = 0; a < 6; a++)
#pragma unroll
          for(int b = 0; b < 3; b++)
            if(fu_nz(a, b)) Fu[a][b] *= P.dt;
      }
      const double Lx4 = P.w_run_com_z * (x[4] - r[3]);
      const double Lu[3] = {P.w_run_zmp * (u[0] - r[0]), P.w_run_zmp * (u[1] - r[1]),
                            P.w_run_force_z * (u[2] - P.mass * kG)};
      const double Luu[3] = {P.w_run_zmp, P.w_run_zmp, P.w_run_force_z}; // diagonal
      // ---- Qx = Lx + Fx' Vx ; Qu = Lu + Fu' Vx
      double Qx[6], Qu[3];
#pragma unroll
      for(int a = 0; a < 6; a++)
      {
        double s = (a == 4) ? Lx4 : 0.0;
#pragma unroll
        for(int b = 0; b < 6; b++)
          if(fx_nz(b, a)) s += Fx[b][a] * Vx[b];
        Qx[a] = s;
      }
#pragma unroll
      for(int q = 0; q < 3; q++)
      {
        double s = Lu[q];
#pragma unroll
        for(int b = 0; b < 6; b++)
          if(fu_nz(b, q)) s += Fu[b][q] * Vx[b];
        Qu[q] = s;
      }
      // ---- Qxx = Lxx + Fx' (Vxx Fx), column by column (T1 column b)
      double Qxx[6][6];
#pragma unroll
      for(int b = 0; b < 6; b++)
      {
        double t1[6];
#pragma unroll
        for(int a = 0; a < 6; a++)
        {
          double s = 0;
#pragma unroll
          for(int k = 0; k < 6; k++)
            if(fx_nz(k, b)) s += Vxx[a][k] * Fx[k][b];
          t1[a] = s;
        }
#pragma unroll
        for(int a = 0; a < 6; a++)
        {
          double s = (a == 4 && b == 4) ? P.w_run_com_z : 0.0;
#pragma unroll
          for(int k = 0; k < 6; k++)
            if(fx_nz(k, a)) s += Fx[k][a] * t1[k];
          Qxx[a][b] = s;
        }
      }
      // ---- unregularised Qxu, Quu and regularised Qxur, QuuF (reg_type 1: Quu_F + lambda I, 2: Vxx + lambda I)
      const double lambda_v = P.cfg.reg_type == 2 ? lambda : 0.0, lambda_q = P.cfg.reg_type == 2 ? 0.0 : lambda;
      double Qxu[6][3], Quu[3][3], Qxur[6][3], QuuF[3][3];
#pragma unroll
      for(int pass = 0; pass < 2; pass++)
      {
#pragma unroll
        for(int q = 0; q < 3; q++)
        {
          double t2[6]; // column q of (Vxx [+ lambda I]) Fu
#pragma unroll
          for(int a = 0; a < 6; a++)
          {
            double s = 0;
#pragma unroll
            for(int k = 0; k < 6; k++)
              if(fu_nz(k, q)) s += ((pass == 1 && a == k) ? Vxx[a][k] + lambda_v : Vxx[a][k]) * Fu[k][q];
            t2[a] = s;
          }
#pragma unroll
          for(int a = 0; a < 6; a++)
          {
            double s = 0.0; // Lxu = 0
#pragma unroll
            for(int k = 0; k < 6; k++)
              if(fx_nz(k, a)) s += Fx[k][a] * t2[k];
            if(pass == 0)
              Qxu[a][q] = s;
            else
              Qxur[a][q] = s;
          }
#pragma unroll
          for(int p = 0; p < 3; p++)
          {
            double s = (p == q) ? Luu[p] : 0.0;
#pragma unroll
            for(int k = 0; k < 6; k++)
              if(fu_nz(k, p)) s += Fu[k][p] * t2[k];
            if(pass == 0)
              Quu[p][q] = s;
            else
              QuuF[p][q] = (p == q) ? s + lambda_q : s;
          }
        }
      }
      // ---- gains
      double kq[3], Lf[3][3], rd[3];
#pragma unroll
      for(int a = 0; a < 3; a++)
#pragma unroll
        for(int b = 0; b < 3; b++) Lf[a][b] = 0.0;
      const int rc = box_qp3(QuuF, Qu, kq, Lf, rd);
      if(rc < 1)
      {
        ok = false;
        break;
      }
      double Km[3][6]; // K_f = -QuuF^-1 Qxur'
#pragma unroll
      for(int a = 0; a < 6; a++)
      {
        double t3[3];
#pragma unroll
        for(int f = 0; f < 3; f++)
        {
          double s = Qxur[a][f];
#pragma unroll
          for(int k = 0; k < f; k++) s -= Lf[f][k] * t3[k];
          t3[f] = s * rd[f];
        }
#pragma unroll
        for(int f = 2; f >= 0; f--)
        {
          double s = t3[f];
#pragma unroll
          for(int k = 2; k > f; k--) s -= Lf[k][f] * t3[k];
          t3[f] = s * rd[f];
        }
#pragma unroll
        for(int f = 0; f < 3; f++) Km[f][a] = -t3[f];
      }
#pragma unroll
      for(int q = 0; q < 3; q++) *W.K1(i, q, inst) = kq[q];
#pragma unroll
      for(int q = 0; q < 3; q++)
#pragma unroll
        for(int a = 0; a < 6; a++) *W.K2(i, q * 6 + a, inst) = Km[q][a];
      // ---- dV, Vx, Vxx
      {
        double t4[3];
        double s0 = 0, s1 = 0;
#pragma unroll
        for(int q = 0; q < 3; q++)
        {
          double s = 0;
#pragma unroll
          for(int p = 0; p < 3; p++) s += Quu[q][p] * kq[p];
          t4[q] = s;
          s0 += kq[q] * Qu[q];
          s1 += kq[q] * s;
        }
        dV0 += s0;
        dV1 += 0.5 * s1;
        double vxn[6];
#pragma unroll
        for(int a = 0; a < 6; a++)
        {
          double s = Qx[a];
#pragma unroll
          for(int q = 0; q < 3; q++) s += Km[q][a] * t4[q] + Km[q][a] * Qu[q] + Qxu[a][q] * kq[q];
          vxn[a] = s;
        }
        double T2[6][3]; // K' Quu
#pragma unroll
        for(int a = 0; a < 6; a++)
#pragma unroll
          for(int q = 0; q < 3; q++)
          {
            double s = 0;
#pragma unroll
            for(int p = 0; p < 3; p++) s += Km[p][a] * Quu[p][q];
            T2[a][q] = s;
          }
        double T1[6][6];
#pragma unroll
        for(int a = 0; a < 6; a++)
#pragma unroll
          for(int b = 0; b < 6; b++)
          {
            double s = Qxx[a][b];
#pragma unroll
            for(int q = 0; q < 3; q++) s += T2[a][q] * Km[q][b] + Km[q][a] * Qxu[b][q] + Qxu[a][q] * Km[q][b];
            T1[a][b] = s;
          }
#pragma unroll
        for(int a = 0; a < 6; a++)
        {
          Vx[a] = vxn[a];
#pragma unroll
          for(int b = 0; b < 6; b++) Vxx[a][b] = 0.5 * (T1[a][b] + T1[b][a]);
        }
      }
    }
    return ok;
  }

  // mean over the steps of max_r |k_r| / (|u_r| + 1)
  __device__ double gain_norm()
  {
    double g = 0;
    for(int i = 0; i < P.N; i++)
    {
      double mx = 0;
#pragma unroll
      for(int q = 0; q < 3; q++)
      {
        const double v = fabs(*W.K1(i, q, inst)) / (fabs(*W.U(cur, i, q, inst)) + 1.0);
        if(v > mx) mx = v;
      }
      g += mx;
    }
    return g / P.N;
  }

  // oracle/ddp.c forward_pass + rollout_cost: candidate trajectory into the other buffer
  __device__ void forward(double alpha)
  {
    const int N = P.N, oth = cur ^ 1;
    double xn[6];
#pragma unroll
    for(int a = 0; a < 6; a++)
    {
      xn[a] = *W.X(cur, 0, a, inst);
      *W.X(oth, 0, a, inst) = xn[a];
    }
    double c = 0;
    for(int i = 0; i < N; i++)
    {
      double xi[6], ui[3], ki[3], Ki[3][6], r[4];
#pragma unroll
      for(int a = 0; a < 6; a++) xi[a] = *W.X(cur, i, a, inst);
#pragma unroll
      for(int q = 0; q < 3; q++) ui[q] = *W.U(cur, i, q, inst);
#pragma unroll
      for(int q = 0; q < 3; q++) ki[q] = *W.K1(i, q, inst);
#pragma unroll
      for(int q = 0; q < 3; q++)
#pragma unroll
        for(int a = 0; a < 6; a++) Ki[q][a] = *W.K2(i, q * 6 + a, inst);
#pragma unroll
      for(int a = 0; a < 4; a++) r[a] = *W.R(i, a, inst);
      double un[3];
#pragma unroll
      for(int q = 0; q < 3; q++)
      {
        double s = ui[q] + alpha * ki[q];
#pragma unroll
        for(int a = 0; a < 6; a++) s += Ki[q][a] * (xn[a] - xi[a]);
        un[q] = s;
        *W.U(oth, i, q, inst) = s;
      }
      c += running_cost(P, xn, un, r);
      double xx[6];
      state_eq(P, xn, un, r, xx);
#pragma unroll
      for(int a = 0; a < 6; a++)
      {
        xn[a] = xx[a];
        *W.X(oth, i + 1, a, inst) = xx[a];
      }
    }
    double r[4];
#pragma unroll
    for(int a = 0; a < 4; a++) r[a] = *W.R(N, a, inst);
    c += terminal_cost(P, xn, r);
    costc = c;
  }
};

// oracle_ddp_solve on the workspace: RefData in W.R, the initial input sequence in W.U(0, ...); leaves the planned
// trajectory in buffer sv.cur.  Returns the iterations executed, status in `status`.
__device__ __forceinline__ int solve(const Params & P, const Ws & W, long inst, const double (&x0)[6], Solver & sv,
                                     int & status)
{
  const int N = P.N;
  // ---- initial rollout and its cost
  sv.cur = 0;
  sv.lambda = P.cfg.initial_lambda;
  sv.dlambda = P.cfg.initial_dlambda;
  {
    double x[6];
#pragma unroll
    for(int a = 0; a < 6; a++)
    {
      x[a] = x0[a];
      *W.X(0, 0, a, inst) = x[a];
    }
    double c = 0;
    for(int i = 0; i < N; i++)
    {
      double u[3], r[4], xn[6];
#pragma unroll
      for(int q = 0; q < 3; q++) u[q] = *W.U(0, i, q, inst);
#pragma unroll
      for(int a = 0; a < 4; a++) r[a] = *W.R(i, a, inst);
      c += running_cost(P, x, u, r);
      state_eq(P, x, u, r, xn);
#pragma unroll
      for(int a = 0; a < 6; a++)
      {
        x[a] = xn[a];
        *W.X(0, i + 1, a, inst) = xn[a];
      }
    }
    double r[4];
#pragma unroll
    for(int a = 0; a < 4; a++) r[a] = *W.R(N, a, inst);
    c += terminal_cost(P, x, r);
    sv.cost = c;
  }
  // ---- iterations
  const ccc_ddp_config_t & C = P.cfg;
  int iter = 0;
  status = 0;
  for(iter = 1; iter <= C.max_iter; iter++)
  {
    bool bp_ok = false;
    for(;;)
    {
      if(sv.backward())
      {
        bp_ok = true;
        break;
      }
      sv.increase_lambda();
      if(sv.lambda > C.lambda_max) break;
    }
    if(!bp_ok)
    {
      status = -1;
      break;
    }
    const double g = sv.gain_norm();
    if(g < C.k_rel_norm_thre && sv.lambda < C.lambda_thre)
    {
      sv.decrease_lambda();
      status = 1;
      break;
    }
    bool accepted = false;
    double actual = 0;
    for(int a = 0; a < 11; a++)
    {
      const double alpha = C.alpha_list[a];
      sv.forward(alpha);
      actual = sv.cost - sv.costc;
      const double expected = -alpha * (sv.dV0 + alpha * sv.dV1);
      const double ratio = expected > 0 ? actual / expected : (actual > 0 ? 1.0 : (actual < 0 ? -1.0 : 0.0));
      if(ratio > C.cost_update_ratio_thre)
      {
        accepted = true;
        break;
      }
    }
    if(accepted)
    {
      sv.decrease_lambda();
      sv.cur ^= 1;
      sv.cost = sv.costc;
      if(actual < C.cost_update_thre)
      {
        status = 2;
        break;
      }
    }
    else
    {
      sv.increase_lambda();
      if(sv.lambda > C.lambda_max)
      {
        status = -1;
        break;
      }
    }
  }
  if(iter > C.max_iter) iter = C.max_iter;
  return iter;
}

__global__ __launch_bounds__(64) void ddpzmp_plan_kernel(Params P, Batch B, long n)
{
  const long inst = (long)blockIdx.x * 64 + threadIdx.x;
  if(inst >= n) return;
  const int N = P.N;
  Ws W{B.ws, n, N};
  // ---- inputs into the workspace layout
  for(int i = 0; i <= N; i++)
#pragma unroll
    for(int a = 0; a < 4; a++) *W.R(i, a, inst) = B.ref[((size_t)inst * (N + 1) + i) * 4 + a];
  for(int i = 0; i < N; i++)
#pragma unroll
    for(int q = 0; q < 3; q++) *W.U(0, i, q, inst) = B.u_init ? B.u_init[((size_t)inst * N + i) * 3 + q] : 0.0;
  double x0[6];
#pragma unroll
  for(int a = 0; a < 6; a++) x0[a] = B.x0[inst * 6 + a];
  Solver sv(P, W, inst);
  int status = 0;
  const int iter = solve(P, W, inst, x0, sv, status);
  // ---- outputs
  for(int i = 0; i < N; i++)
#pragma unroll
    for(int q = 0; q < 3; q++) B.u_out[((size_t)inst * N + i) * 3 + q] = *W.U(sv.cur, i, q, inst);
  if(B.x_out)
    for(int i = 0; i <= N; i++)
#pragma unroll
      for(int a = 0; a < 6; a++) B.x_out[((size_t)inst * (N + 1) + i) * 6 + a] = *W.X(sv.cur, i, a, inst);
  if(B.iters) B.iters[inst] = iter;
  if(B.status) B.status[inst] = status;
  if(B.cost) B.cost[inst] = sv.cost;
}

// ---------------------------------------------------------------------------------------------------------------
// The control loop of tests/src/TestDdpZmp.cpp:70-125 for n instances, entirely on the device (SURVEY.md 8(f) rank 4:
// plan -> simulate -> plan ...): per cycle the RefData of the horizon is sampled from the instance's reference-ZMP
// polyline (FootstepManager::refZmp, tests/src/FootstepManager.h:228-237: linear interpolation between the knots of
// ref_zmp_list_, evaluated at t + 1e-6), the planner runs warm-started with its previous input sequence (:88-91; first
// cycle (CoM xy, m g), :84-86), ComZmpSim3d advances by sim_dt (tests/src/SimModels.h:194-201: the horizontal model is
// rebuilt with the current CoM height every cycle) and the kicks of :118-125 are added.
// ---------------------------------------------------------------------------------------------------------------
struct Loop
{
  int K;                  // knots per instance
  const double * knot_t;  // [K][n]
  const double * knot_z;  // [K][2][n]
  double com_height;      // RefData::com_z (and zmp z = 0)
  double * state;         // [6][n] in/out: [cx, vx, cy, vy, cz, vz]
  double t0, sim_dt;
  int cycles, n_disturb;
  double disturb_t[8], disturb_v; // impulse per mass, added to BOTH horizontal velocities (SimModels.h:206-210)
  double * stats;         // [4][n]: max |planned zmp - ref zmp|, max |cz - com_height|, final |planned zmp - ref zmp|,
                          //         DDP iterations in total
  double * log;           // [cycles][3][n] planned (zmp x, zmp y, f_z) per cycle, or null
};

__global__ __launch_bounds__(64) void ddpzmp_closed_loop_kernel(Params P, Loop L, double * ws, long n)
{
  const long inst = (long)blockIdx.x * 64 + threadIdx.x;
  if(inst >= n) return;
  const int N = P.N, K = L.K;
  Ws W{ws, n, N};
  double st[6];
#pragma unroll
  for(int a = 0; a < 6; a++) st[a] = L.state[(size_t)a * n + inst];
  Solver sv(P, W, inst);
  auto knot_time = [&](int k) { return L.knot_t[(size_t)k * n + inst]; };
  // reference ZMP at time t (FootstepManager::refZmp): segment search forward from `k` (t never decreases)
  auto ref_zmp = [&](double t, int & k, double & zx, double & zy) {
    t = t + 1e-6;
    while(k + 1 < K && knot_time(k + 1) <= t) ++k; // k = last knot with time <= t (0 if t lies before the first)
    const double ta = knot_time(k);
    if(k + 1 >= K || t < ta)
    {
      zx = L.knot_z[((size_t)k * 2 + 0) * n + inst];
      zy = L.knot_z[((size_t)k * 2 + 1) * n + inst];
      return;
    }
    const double tb = knot_time(k + 1);
    const double ratio = (t - ta) / (tb - ta);
    zx = (1 - ratio) * L.knot_z[((size_t)k * 2 + 0) * n + inst] + ratio * L.knot_z[((size_t)(k + 1) * 2 + 0) * n + inst];
    zy = (1 - ratio) * L.knot_z[((size_t)k * 2 + 1) * n + inst] + ratio * L.knot_z[((size_t)(k + 1) * 2 + 1) * n + inst];
  };
  double t = L.t0, worst_zmp = 0, worst_z = 0, last_zmp = 0, iters_total = 0;
  int k0 = 0;
  for(int c = 0; c < L.cycles; c++)
  {
    // ---- RefData of the horizon (TestDdpZmp.cpp:45-51)
    double rz0x = 0, rz0y = 0;
    {
      int k = k0;
      for(int i = 0; i <= N; i++)
      {
        double zx, zy;
        ref_zmp(t + i * P.dt, k, zx, zy);
        if(i == 0)
        {
          k0 = k;
          rz0x = zx;
          rz0y = zy;
        }
        *W.R(i, 0, inst) = zx;
        *W.R(i, 1, inst) = zy;
        *W.R(i, 2, inst) = 0.0;
        *W.R(i, 3, inst) = L.com_height;
      }
    }
    // ---- warm start: the previous plan as it is (no shift), first cycle (CoM xy, m g)
    const int prev = sv.cur;
    for(int i = 0; i < N; i++)
#pragma unroll
      for(int q = 0; q < 3; q++)
      {
        const double v = (c == 0) ? (q == 0 ? st[0] : (q == 1 ? st[2] : P.mass * kG)) : *W.U(prev, i, q, inst);
        if(c == 0 || prev != 0) *W.U(0, i, q, inst) = v;
      }
    int status = 0;
    iters_total += solve(P, W, inst, st, sv, status);
    const double zx = *W.U(sv.cur, 0, 0, inst), zy = *W.U(sv.cur, 0, 1, inst), fz = *W.U(sv.cur, 0, 2, inst);
    if(L.log)
    {
      L.log[((size_t)c * 3 + 0) * n + inst] = zx;
      L.log[((size_t)c * 3 + 1) * n + inst] = zy;
      L.log[((size_t)c * 3 + 2) * n + inst] = fz;
    }
    {
      const double ex = zx - rz0x, ey = zy - rz0y;
      last_zmp = sqrt(ex * ex + ey * ey);
      worst_zmp = fmax(worst_zmp, last_zmp);
      worst_z = fmax(worst_z, fabs(st[4] - L.com_height));
    }
    // ---- simulate (SimModels.h:11-41,44-73,194-201): exact ZOH of x'' = w^2 (x - zmp) at the current height, and of
    //      the vertical double integrator under gravity
    t += L.sim_dt;
    {
      const double w = sqrt(kG / st[4]);
      const double ch = cosh(w * L.sim_dt), sh = sinh(w * L.sim_dt);
      const double x = st[0], vx = st[1], y = st[2], vy = st[3];
      st[0] = ch * x + sh / w * vx + (1 - ch) * zx;
      st[1] = w * sh * x + ch * vx + -w * sh * zx;
      st[2] = ch * y + sh / w * vy + (1 - ch) * zy;
      st[3] = w * sh * y + ch * vy + -w * sh * zy;
      const double z = st[4], vz = st[5], h = L.sim_dt;
      st[4] = z + h * vz + 0.5 * h * h / P.mass * fz + -kG * (0.5 * h * h);
      st[5] = vz + h / P.mass * fz + -kG * h;
    }
    for(int d = 0; d < L.n_disturb; d++)
      if(L.disturb_t[d] <= t && t < L.disturb_t[d] + L.sim_dt)
      {
        st[1] += L.disturb_v;
        st[3] += L.disturb_v;
        break;
      }
  }
#pragma unroll
  for(int a = 0; a < 6; a++) L.state[(size_t)a * n + inst] = st[a];
  if(L.stats)
  {
    L.stats[(size_t)0 * n + inst] = worst_zmp;
    L.stats[(size_t)1 * n + inst] = worst_z;
    L.stats[(size_t)2 * n + inst] = last_zmp;
    L.stats[(size_t)3 * n + inst] = iters_total;
  }
}
} // namespace dz
} // namespace ccc_amd

using namespace ccc_amd;

struct ccc_ddpzmp
{
  int device = 0;
  dz::Params P{};
  double * ws = nullptr;
  int64_t ws_cap = 0; // instances the workspace holds
  // staging for the host-pointer entry point
  char * d_stage = nullptr;
  size_t stage_bytes = 0;
  hipStream_t stream = nullptr;
};

extern "C" void ccc_ddpzmp_default_config(ccc_ddp_config_t * c)
{
  if(!c) return;
  c->max_iter = 500;
  c->initial_lambda = 1e-4;
  c->initial_dlambda = 1.0;
  c->lambda_factor = 1.6;
  c->lambda_min = 1e-6;
  c->lambda_max = 1e10;
  c->k_rel_norm_thre = 1e-4;
  c->lambda_thre = 1e-5;
  c->cost_update_ratio_thre = 0.0;
  c->cost_update_thre = 1e-7;
  for(int i = 0; i < 11; i++) c->alpha_list[i] = std::pow(10.0, -3.0 * i / 10.0);
  c->reg_type = 1;
  c->precision = 64;
  c->warm_start_guard = 0; // (ignored by this class)
}

extern "C" int ccc_ddpzmp_create(double mass, double horizon_dt, int horizon_steps, const double * weights, int device,
                                 ccc_ddpzmp_t ** out)
{
  if(!out) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_create: out is NULL");
  *out = nullptr;
  if(!(mass > 0) || !(horizon_dt > 0) || horizon_steps <= 0)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_create: mass, horizon_dt, horizon_steps must be > 0");
  if(horizon_steps > 4096)
    return fail(CCC_ERR_UNSUPPORTED, "ccc_ddpzmp_create: horizon_steps %d > 4096", horizon_steps);
  int rc = select_device(device);
  if(rc != CCC_OK) return rc;
  CCC_DEVICE_GUARD(device);
  ccc_ddpzmp * h = new ccc_ddpzmp();
  h->device = device;
  h->P.N = horizon_steps;
  h->P.mass = mass;
  h->P.dt = horizon_dt;
  static const double kDefault[6] = {1e2, 1e-1, 1e-4, 1.0, 1e2, 1.0}; // include/CCC/DdpZmp.h:72-77
  const double * w = weights ? weights : kDefault;
  h->P.w_run_com_z = w[0];
  h->P.w_run_zmp = w[1];
  h->P.w_run_force_z = w[2];
  h->P.w_term_com_xy = w[3];
  h->P.w_term_com_z = w[4];
  h->P.w_term_com_vel = w[5];
  ccc_ddpzmp_default_config(&h->P.cfg);
  *out = h;
  return CCC_OK;
}

extern "C" void ccc_ddpzmp_destroy(ccc_ddpzmp_t * h)
{
  if(!h) return;
  ccc_amd::DeviceGuard ccc_device_guard__(h->device);
  if(h->ws) (void)hipFree(h->ws);
  if(h->d_stage) (void)hipFree(h->d_stage);
  if(h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int ccc_ddpzmp_set_config(ccc_ddpzmp_t * h, const ccc_ddp_config_t * cfg)
{
  if(!h || !cfg) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_set_config: NULL argument");
  if(cfg->max_iter < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_set_config: max_iter < 0");
  if(cfg->reg_type != 1 && cfg->reg_type != 2) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_set_config: reg_type must be 1 or 2");
  if(cfg->precision != 64)
    return fail(CCC_ERR_UNSUPPORTED, "ccc_ddpzmp_set_config: precision = %d, the DdpZmp kernel is fp64 only", cfg->precision);
  h->P.cfg = *cfg;
  return CCC_OK;
}

extern "C" int64_t ccc_ddpzmp_workspace_bytes(const ccc_ddpzmp_t * h, int64_t n)
{
  if(!h || n < 0) return 0;
  return (int64_t)(dz::Ws::doubles_per_instance(h->P.N) * sizeof(double) * dz::Ws::instances_padded(n));
}

extern "C" int ccc_ddpzmp_plan_batch_device(ccc_ddpzmp_t * h, int64_t n, const double * ref, const double * x0,
                                            const double * u_init, double * u_out, double * x_out, int32_t * iters,
                                            int32_t * status, double * cost, void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_plan_batch_device: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_plan_batch_device: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!ref || !x0 || !u_out) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_plan_batch_device: NULL ref/x0/u_out");
  CCC_DEVICE_GUARD(h->device);
  if(n > h->ws_cap) // the workspace grows to the largest batch seen (synchronously: not inside a captured stream)
  {
    CCC_NO_CAPTURE(stream, "ccc_ddpzmp device entry");
    if(h->ws) CCC_HIP_CHECK(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_cap = 0;
    CCC_HIP_CHECK(hipMalloc(&h->ws, dz::Ws::doubles_per_instance(h->P.N) * sizeof(double) * dz::Ws::instances_padded(n)));
    h->ws_cap = n;
  }
  dz::Batch B{ref, x0, u_init, u_out, x_out, iters, status, cost, h->ws};
  const int grid = (int)((n + 63) / 64);
  hipLaunchKernelGGL(dz::ddpzmp_plan_kernel, dim3(grid), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), h->P, B,
                     (long)n);
  CCC_HIP_CHECK(hipGetLastError());
  return CCC_OK;
}

extern "C" int ccc_ddpzmp_plan_batch(ccc_ddpzmp_t * h, int64_t n, const double * ref, const double * x0,
                                     const double * u_init, double * u_out, double * x_out, int32_t * iters,
                                     int32_t * status, double * cost)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_plan_batch: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_plan_batch: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!ref || !x0 || !u_out) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_plan_batch: NULL ref/x0/u_out");
  CCC_DEVICE_GUARD(h->device);
  const size_t N = (size_t)h->P.N;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t b_ref = (size_t)n * (N + 1) * 4 * 8, b_x0 = (size_t)n * 6 * 8, b_u = (size_t)n * N * 3 * 8,
               b_x = (size_t)n * (N + 1) * 6 * 8, b_i = (size_t)n * 4, b_c = (size_t)n * 8;
  const size_t o_ref = 0, o_x0 = o_ref + up(b_ref), o_ui = o_x0 + up(b_x0), o_uo = o_ui + up(b_u), o_xo = o_uo + up(b_u),
               o_it = o_xo + up(b_x), o_st = o_it + up(b_i), o_co = o_st + up(b_i), total = o_co + up(b_c);
  if(total > h->stage_bytes)
  {
    if(h->d_stage) CCC_HIP_CHECK(hipFree(h->d_stage));
    h->d_stage = nullptr;
    h->stage_bytes = 0;
    CCC_HIP_CHECK(hipMalloc(&h->d_stage, total));
    h->stage_bytes = total;
  }
  if(!h->stream) CCC_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  char * d = h->d_stage;
  CCC_HIP_CHECK(hipMemcpyAsync(d + o_ref, ref, b_ref, hipMemcpyHostToDevice, h->stream));
  CCC_HIP_CHECK(hipMemcpyAsync(d + o_x0, x0, b_x0, hipMemcpyHostToDevice, h->stream));
  if(u_init) CCC_HIP_CHECK(hipMemcpyAsync(d + o_ui, u_init, b_u, hipMemcpyHostToDevice, h->stream));
  int rc = ccc_ddpzmp_plan_batch_device(
      h, n, reinterpret_cast<const double *>(d + o_ref), reinterpret_cast<const double *>(d + o_x0),
      u_init ? reinterpret_cast<const double *>(d + o_ui) : nullptr, reinterpret_cast<double *>(d + o_uo),
      x_out ? reinterpret_cast<double *>(d + o_xo) : nullptr, iters ? reinterpret_cast<int32_t *>(d + o_it) : nullptr,
      status ? reinterpret_cast<int32_t *>(d + o_st) : nullptr, cost ? reinterpret_cast<double *>(d + o_co) : nullptr,
      h->stream);
  if(rc != CCC_OK) return rc;
  CCC_HIP_CHECK(hipMemcpyAsync(u_out, d + o_uo, b_u, hipMemcpyDeviceToHost, h->stream));
  if(x_out) CCC_HIP_CHECK(hipMemcpyAsync(x_out, d + o_xo, b_x, hipMemcpyDeviceToHost, h->stream));
  if(iters) CCC_HIP_CHECK(hipMemcpyAsync(iters, d + o_it, b_i, hipMemcpyDeviceToHost, h->stream));
  if(status) CCC_HIP_CHECK(hipMemcpyAsync(status, d + o_st, b_i, hipMemcpyDeviceToHost, h->stream));
  if(cost) CCC_HIP_CHECK(hipMemcpyAsync(cost, d + o_co, b_c, hipMemcpyDeviceToHost, h->stream));
  CCC_HIP_CHECK(hipStreamSynchronize(h->stream));
  return CCC_OK;
}

extern "C" int ccc_ddpzmp_closed_loop_device(ccc_ddpzmp_t * h, int64_t n, int K, const double * knot_t,
                                             const double * knot_zmp, double com_height, double * state, double t0,
                                             double sim_dt, int cycles, int n_disturb, const double * disturb_times,
                                             double disturb_impulse, double * stats, double * log, void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_closed_loop_device: NULL handle");
  if(n < 0 || K < 1 || cycles < 0 || !(sim_dt > 0))
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_closed_loop_device: need n >= 0, K >= 1, cycles >= 0, sim_dt > 0");
  if(n == 0 || cycles == 0) return CCC_OK;
  if(!knot_t || !knot_zmp || !state) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_closed_loop_device: NULL argument");
  if(n_disturb < 0 || n_disturb > 8 || (n_disturb > 0 && !disturb_times))
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddpzmp_closed_loop_device: 0 <= n_disturb <= 8 (HOST array of times)");
  CCC_DEVICE_GUARD(h->device);
  if(n > h->ws_cap)
  {
    CCC_NO_CAPTURE(stream, "ccc_ddpzmp device entry");
    if(h->ws) CCC_HIP_CHECK(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_cap = 0;
    CCC_HIP_CHECK(hipMalloc(&h->ws, dz::Ws::doubles_per_instance(h->P.N) * sizeof(double) * dz::Ws::instances_padded(n)));
    h->ws_cap = n;
  }
  dz::Loop L{};
  L.K = K;
  L.knot_t = knot_t;
  L.knot_z = knot_zmp;
  L.com_height = com_height;
  L.state = state;
  L.t0 = t0;
  L.sim_dt = sim_dt;
  L.cycles = cycles;
  L.n_disturb = n_disturb;
  for(int d = 0; d < n_disturb; d++) L.disturb_t[d] = disturb_times[d];
  L.disturb_v = disturb_impulse;
  L.stats = stats;
  L.log = log;
  const int grid = (int)((n + 63) / 64);
  hipLaunchKernelGGL(dz::ddpzmp_closed_loop_kernel, dim3(grid), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), h->P, L,
                     h->ws, (long)n);
  CCC_HIP_CHECK(hipGetLastError());
  return CCC_OK;
}
