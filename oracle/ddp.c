/* ddp.c -- control-limited DDP / iLQR solver of the CPU oracle (TEST INFRASTRUCTURE ONLY, see ccc_oracle.h).
 *
 * Stands in for nmpc_ddp::DDPSolver<S, Eigen::Dynamic> (isri-aist/NMPC, NOT vendored in the reference and not
 * version-pinned: README.md:53, .github/workflows/ci-colcon.yaml:80-86), the external call at
 *   /root/reference/src/DdpCentroidal.cpp:229,233      ddp_solver_->solve(current_time, x0, u_list)
 *   /root/reference/src/DdpSingleRigidBody.cpp:299,303
 * PARITY UNPINNED at this boundary: the iterate after k iterations depends on solver internals that are not in
 * the reference tree.  This file freezes the algorithm of SURVEY.md Appendix B.2 -- Tassa, Mansard, Todorov,
 * "Control-limited differential dynamic programming" (ICRA 2014) with the published iLQG/boxQP reference
 * parameters, which nmpc_ddp follows -- and the HIP kernels implement exactly this specification:
 *
 *   solve:    lambda = initial_lambda, dlambda = initial_dlambda; rollout of the initial inputs; then max_iter times:
 *     (1) derivatives along the current trajectory (first-order dynamics: iLQR/Gauss-Newton, the reference throws on
 *         the second-order overload, include/CCC/DdpCentroidal.h:217-228)
 *     (2) backward pass; on failure (box-QP/Cholesky) dlambda = max(dlambda*f, f), lambda = max(lambda*dlambda,
 *         lambda_min), stop if lambda > lambda_max, else retry
 *     (3) g = mean_i max_j |k_ij| / (|u_ij| + 1); if g < k_rel_norm_thre and lambda < lambda_thre: decrease lambda, stop
 *     (4) line search over alpha_list: u' = clamp(u + alpha k + K (x' - x)); accept the first alpha with
 *         (cost - cost') / (-alpha (dV0 + alpha dV1)) > cost_update_ratio_thre  (sign(cost - cost') if the
 *         expectation is not positive)
 *     (5) accepted: dlambda = min(dlambda/f, 1/f), lambda = lambda*dlambda*(lambda > lambda_min); stop if the cost
 *         decrease < cost_update_thre;  rejected: increase lambda as in (2), stop if lambda > lambda_max
 *   backward pass: reg_type 1 (default; iLQG.m "regType 1: q_uu + lambda*eye()", which nmpc_ddp's reg_type follows):
 *       Quu_F = Luu + Fu' Vxx Fu + lambda I, Qxu_reg = Qxu;   reg_type 2: Vxx_reg = Vxx + lambda I in both.
 *     Evidence for 1 (round 2, tests/test_oracle_ddp.py): on the reference's own SRB scenario the cold solve from
 *     u = 0 converges in 29 of 32 runs (x0 perturbed by 1e-10) with reg_type 1 and in 3 of 32 with reg_type 2
 *     (27 end with lambda > lambda_max); and the CCC constructors lower initial_lambda to 1e-6 = running_force
 *     (src/DdpCentroidal.cpp:199), which only matters when lambda is added to Quu.
 *     k = boxQP(Quu_F, Qu, lo - u, hi - u, warm start k_{i+1}); K_free = -Quu_F,ff^-1 Qxu_reg,f'; clamped rows of K = 0;
 *     dV += [k'Qu, 1/2 k'Quu k]; Vx = Qx + K'Quu k + K'Qu + Qxu k; Vxx = sym(Qxx + K'Quu K + K'Qxu' + Qxu K)
 *   warm-start guard (config warm_start_guard, NOT nmpc_ddp -- this repository's addition, round 4): a warm start whose
 *     open-loop rollout costs more than the rollout of zero inputs, or is not finite, is replaced by zero inputs.
 *   boxQP: projected Newton (Tassa's boxQP.m) with nmpc_ddp's BoxQP::Configuration: max_iter 500, grad_thre 1e-8,
 *     rel_improve_thre 1e-8, step_factor 0.6, min_step 1e-22, armijo_param 0.1.
 */
#include "ccc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void oracle_ddp_default_config(oracle_ddp_config_t * c)
{
  c->with_input_constraint = 1;
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
  c->reg_type = 1;
  c->arith = 0;
  c->warm_start_guard = 1; /* = ccc_ddp_default_config (csrc/ddp.hip); 0 = the recalled nmpc_ddp behaviour */
  for(int i = 0; i < 11; i++) c->alpha_list[i] = pow(10.0, -3.0 * i / 10.0);
}

/* ------------------------------------------------------------------------------------------- box QP */

/* min 1/2 x'Hx + g'x, lo <= x <= hi; x enters as the warm start.  H: n x n row-major.
 * Outputs: x, is_free[n], Lf (nf x nf lower Cholesky factor of H_ff, row-major with stride n), rd (nf reciprocals
 * of its diagonal: the triangular solves multiply by rd instead of dividing; the backward substitution subtracts
 * in DEcreasing column order -- both choices are part of the frozen specification the HIP kernel shares).
 * Returns the boxQP.m result code (>= 1 success; -1 not positive definite; 0 no descent direction). */
int oracle_box_qp(int n, const double * H, const double * g, const double * lo, const double * hi, double * x,
                  int * is_free, double * Lf, double * rd, int * n_free_out, int * iters_out)
{
  const int max_iter = 500; /* nmpc_ddp BoxQP::Configuration::max_iter (boxQP.m has 100) */
  const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
  double * grad = (double *)malloc(sizeof(double) * n * 4);
  double * search = grad + n, * xc = grad + 2 * n, * tmp = grad + 3 * n;
  int * clamped = (int *)calloc(n, sizeof(int));
  int * old_clamped = (int *)calloc(n, sizeof(int));
  int * fidx = (int *)calloc(n, sizeof(int));
  int result = 0, nf = 0, iter = 0;

  for(int i = 0; i < n; i++) x[i] = fmin(fmax(x[i], lo[i]), hi[i]);
  double value = 0;
  for(int i = 0; i < n; i++)
  {
    double s = 0;
    for(int j = 0; j < n; j++) s += H[i * n + j] * x[j];
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
    for(int i = 0; i < n; i++)
    {
      double s = g[i];
      for(int j = 0; j < n; j++) s += H[i * n + j] * x[j];
      grad[i] = s;
    }
    int changed = (iter == 1), all_clamped = 1;
    for(int i = 0; i < n; i++)
    {
      old_clamped[i] = clamped[i];
      clamped[i] = ((x[i] == lo[i] && grad[i] > 0) || (x[i] == hi[i] && grad[i] < 0)) ? 1 : 0;
      if(clamped[i] != old_clamped[i]) changed = 1;
      if(!clamped[i]) all_clamped = 0;
    }
    if(all_clamped)
    {
      result = 6;
      break;
    }
    if(changed)
    {
      nf = 0;
      for(int i = 0; i < n; i++)
        if(!clamped[i]) fidx[nf++] = i;
      /* Cholesky of H_ff */
      int ok = 1;
      for(int a = 0; a < nf && ok; a++)
      {
        double s = H[fidx[a] * n + fidx[a]];
        for(int k = 0; k < a; k++) s -= Lf[a * n + k] * Lf[a * n + k];
        if(!(s > 0.0))
        {
          ok = 0;
          break;
        }
        Lf[a * n + a] = sqrt(s);
        rd[a] = 1.0 / Lf[a * n + a];
        for(int b = a + 1; b < nf; b++)
        {
          double t = H[fidx[b] * n + fidx[a]];
          for(int k = 0; k < a; k++) t -= Lf[b * n + k] * Lf[a * n + k];
          Lf[b * n + a] = t * rd[a];
        }
      }
      if(!ok)
      {
        result = -1;
        break;
      }
    }
    double gnorm = 0;
    for(int a = 0; a < nf; a++) gnorm += grad[fidx[a]] * grad[fidx[a]];
    gnorm = sqrt(gnorm);
    if(gnorm < min_grad)
    {
      result = 5;
      break;
    }
    /* grad_clamped = g + H (x .* clamped); search_f = -H_ff^-1 grad_clamped_f - x_f */
    for(int a = 0; a < nf; a++)
    {
      double s = g[fidx[a]];
      for(int j = 0; j < n; j++)
        if(clamped[j]) s += H[fidx[a] * n + j] * x[j];
      tmp[a] = s;
    }
    for(int a = 0; a < nf; a++)
    {
      double s = tmp[a];
      for(int k = 0; k < a; k++) s -= Lf[a * n + k] * tmp[k];
      tmp[a] = s * rd[a];
    }
    for(int a = nf - 1; a >= 0; a--)
    {
      double s = tmp[a];
      for(int k = nf - 1; k > a; k--) s -= Lf[k * n + a] * tmp[k];
      tmp[a] = s * rd[a];
    }
    for(int i = 0; i < n; i++) search[i] = 0;
    for(int a = 0; a < nf; a++) search[fidx[a]] = -tmp[a] - x[fidx[a]];
    double sdotg = 0;
    for(int i = 0; i < n; i++) sdotg += search[i] * grad[i];
    if(sdotg >= 0) break; /* no descent direction: result stays 0 */
    double step = 1.0, vc = 0;
    for(;;)
    {
      for(int i = 0; i < n; i++) xc[i] = fmin(fmax(x[i] + step * search[i], lo[i]), hi[i]);
      vc = 0;
      for(int i = 0; i < n; i++)
      {
        double s = 0;
        for(int j = 0; j < n; j++) s += H[i * n + j] * xc[j];
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
    for(int i = 0; i < n; i++) x[i] = xc[i];
    value = vc;
  }
  if(iter > max_iter && result == 0) result = 1;
  for(int i = 0; i < n; i++) is_free[i] = !clamped[i];
  nf = 0;
  for(int i = 0; i < n; i++) nf += is_free[i];
  if(n_free_out) *n_free_out = nf;
  if(iters_out) *iters_out = iter;
  free(grad);
  free(clamped);
  free(old_clamped);
  free(fidx);
  return result;
}

/* ------------------------------------------------------------------------------------------- DDP */

#include <stdio.h>
/* development aid: when non-zero, oracle_ddp_solve prints one line per line-search candidate to stderr */
int oracle_ddp_trace = 0;

typedef struct
{
  const oracle_ddp_problem_t * p;
  const oracle_ddp_config_t * c;
  int S, N, M; /* M = stride of the per-step input arrays */
  double *x, *u, *xc, *uc;       /* trajectories and line-search candidates */
  double *Fx, *Fu, *Lx, *Lu, *Lxx, *Luu, *Lxu; /* per-step derivatives */
  double *Vx_T, *Vxx_T;          /* terminal */
  double *k, *K;                 /* gains: N x M, N x M x S */
  int * m;                       /* per-step input dim */
  double cost, costc;
  double dV[2];
  double lambda, dlambda;
} ddp_t;

static double rollout_cost(ddp_t * d, const double * x, const double * u)
{
  double c = 0;
  for(int i = 0; i < d->N; i++) c += d->p->running_cost(d->p->user, i, x + (size_t)i * d->S, u + (size_t)i * d->M);
  c += d->p->terminal_cost(d->p->user, x + (size_t)d->N * d->S);
  return c;
}

static void increase_lambda(ddp_t * d)
{
  d->dlambda = fmax(d->dlambda * d->c->lambda_factor, d->c->lambda_factor);
  d->lambda = fmax(d->lambda * d->dlambda, d->c->lambda_min);
}

static void decrease_lambda(ddp_t * d)
{
  d->dlambda = fmin(d->dlambda / d->c->lambda_factor, 1.0 / d->c->lambda_factor);
  d->lambda = d->lambda * d->dlambda * (d->lambda > d->c->lambda_min ? 1.0 : 0.0);
}

static int backward_pass(ddp_t * d)
{
  const int S = d->S, N = d->N, M = d->M;
  double * Vx = (double *)malloc(sizeof(double) * (S + S * S * 4 + S * M * 3 + M * M * 3 + M * 7 + S * 2));
  double * Vxx = Vx + S;
  double * Vxxr = Vxx + S * S;
  double * Qxx = Vxxr + S * S;
  double * T1 = Qxx + S * S;  /* S x S scratch */
  double * Qxu = T1 + S * S;  /* S x m */
  double * Qxur = Qxu + S * M;
  double * T2 = Qxur + S * M; /* S x m scratch: Vxx Fu */
  double * Quu = T2 + S * M;  /* m x m */
  double * QuuF = Quu + M * M;
  double * Lf = QuuF + M * M;
  double * Qu = Lf + M * M;
  double * lo = Qu + M, * hi = lo + M, * kq = hi + M, * t3 = kq + M, * t4 = t3 + M;
  double * rd = t4 + M;
  double * Qx = rd + M, * vxn = Qx + S;
  int * is_free = (int *)malloc(sizeof(int) * M);
  int ok = 1;

  memcpy(Vx, d->Vx_T, sizeof(double) * S);
  memcpy(Vxx, d->Vxx_T, sizeof(double) * S * S);
  d->dV[0] = d->dV[1] = 0;
  for(int i = N - 1; i >= 0 && ok; i--)
  {
    const int m = d->m[i];
    const double * Fx = d->Fx + (size_t)i * S * S;
    const double * Fu = d->Fu + (size_t)i * S * M; /* S x m, row stride M */
    const double * Lx = d->Lx + (size_t)i * S, * Lu = d->Lu + (size_t)i * M;
    const double * Lxx = d->Lxx + (size_t)i * S * S, * Luu = d->Luu + (size_t)i * M * M;
    const double * Lxu = d->Lxu + (size_t)i * S * M;
    double * ki = d->k + (size_t)i * M, * Ki = d->K + (size_t)i * M * S;
    /* Qx = Lx + Fx' Vx ; Qu = Lu + Fu' Vx */
    for(int a = 0; a < S; a++)
    {
      double s = Lx[a];
      for(int b = 0; b < S; b++) s += Fx[b * S + a] * Vx[b];
      Qx[a] = s;
    }
    for(int r = 0; r < m; r++)
    {
      double s = Lu[r];
      for(int b = 0; b < S; b++) s += Fu[b * M + r] * Vx[b];
      Qu[r] = s;
    }
    for(int a = 0; a < S * S; a++) Vxxr[a] = Vxx[a];
    /* reg_type 1 (nmpc_ddp / iLQG default): Quu_F + lambda I;  reg_type 2: Vxx + lambda I */
    const double lambda_v = d->c->reg_type == 2 ? d->lambda : 0.0, lambda_q = d->c->reg_type == 2 ? 0.0 : d->lambda;
    for(int a = 0; a < S; a++) Vxxr[a * S + a] += lambda_v;
    /* Qxx = Lxx + Fx' Vxx Fx */
    for(int a = 0; a < S; a++)
      for(int b = 0; b < S; b++)
      {
        double s = 0;
        for(int k = 0; k < S; k++) s += Vxx[a * S + k] * Fx[k * S + b];
        T1[a * S + b] = s;
      }
    for(int a = 0; a < S; a++)
      for(int b = 0; b < S; b++)
      {
        double s = Lxx[a * S + b];
        for(int k = 0; k < S; k++) s += Fx[k * S + a] * T1[k * S + b];
        Qxx[a * S + b] = s;
      }
    /* unregularised: Qxu = Lxu + Fx' Vxx Fu ; Quu = Luu + Fu' Vxx Fu */
    for(int a = 0; a < S; a++)
      for(int r = 0; r < m; r++)
      {
        double s = 0;
        for(int k = 0; k < S; k++) s += Vxx[a * S + k] * Fu[k * M + r];
        T2[a * M + r] = s;
      }
    for(int a = 0; a < S; a++)
      for(int r = 0; r < m; r++)
      {
        double s = Lxu[a * M + r];
        for(int k = 0; k < S; k++) s += Fx[k * S + a] * T2[k * M + r];
        Qxu[a * M + r] = s;
      }
    for(int r = 0; r < m; r++)
      for(int q = 0; q < m; q++)
      {
        double s = Luu[r * M + q];
        for(int k = 0; k < S; k++) s += Fu[k * M + r] * T2[k * M + q];
        Quu[r * m + q] = s;
      }
    /* regularised */
    for(int a = 0; a < S; a++)
      for(int r = 0; r < m; r++)
      {
        double s = 0;
        for(int k = 0; k < S; k++) s += Vxxr[a * S + k] * Fu[k * M + r];
        T2[a * M + r] = s;
      }
    for(int a = 0; a < S; a++)
      for(int r = 0; r < m; r++)
      {
        double s = Lxu[a * M + r];
        for(int k = 0; k < S; k++) s += Fx[k * S + a] * T2[k * M + r];
        Qxur[a * M + r] = s;
      }
    for(int r = 0; r < m; r++)
      for(int q = 0; q < m; q++)
      {
        double s = Luu[r * M + q];
        for(int k = 0; k < S; k++) s += Fu[k * M + r] * T2[k * M + q];
        QuuF[r * m + q] = s;
      }
    for(int r = 0; r < m; r++) QuuF[r * m + r] += lambda_q;
    for(int r = 0; r < M; r++) ki[r] = 0;
    for(int a = 0; a < M * S; a++) Ki[a] = 0;
    int nf = 0;
    if(m > 0)
    {
      if(d->c->with_input_constraint)
      {
        d->p->input_limits(d->p->user, i, lo, hi);
        const double * ui = d->u + (size_t)i * M;
        for(int r = 0; r < m; r++)
        {
          lo[r] -= ui[r];
          hi[r] -= ui[r];
        }
        /* warm start: the gain of step i+1 of this pass (zeros for the last step or on a dimension change) */
        if(i + 1 < N && d->m[i + 1] == m)
          memcpy(kq, d->k + (size_t)(i + 1) * M, sizeof(double) * m);
        else
          memset(kq, 0, sizeof(double) * m);
        int rc = oracle_box_qp(m, QuuF, Qu, lo, hi, kq, is_free, Lf, rd, &nf, NULL);
        if(rc < 1)
        {
          ok = 0;
          break;
        }
      }
      else
      {
        for(int r = 0; r < m; r++)
        {
          lo[r] = -INFINITY;
          hi[r] = INFINITY;
          kq[r] = 0;
        }
        int rc = oracle_box_qp(m, QuuF, Qu, lo, hi, kq, is_free, Lf, rd, &nf, NULL);
        if(rc < 1)
        {
          ok = 0;
          break;
        }
      }
      memcpy(ki, kq, sizeof(double) * m);
      /* K_f = -QuuF_ff^-1 Qxur_f'  (row r of K is the feedback of input r) */
      int fidx[256];
      int cnt = 0;
      for(int r = 0; r < m; r++)
        if(is_free[r]) fidx[cnt++] = r;
      for(int a = 0; a < S; a++)
      {
        for(int f = 0; f < cnt; f++)
        {
          double s = Qxur[a * M + fidx[f]];
          for(int k = 0; k < f; k++) s -= Lf[f * m + k] * t3[k];
          t3[f] = s * rd[f];
        }
        for(int f = cnt - 1; f >= 0; f--)
        {
          double s = t3[f];
          for(int k = cnt - 1; k > f; k--) s -= Lf[k * m + f] * t3[k];
          t3[f] = s * rd[f];
        }
        for(int f = 0; f < cnt; f++) Ki[fidx[f] * S + a] = -t3[f];
      }
    }
    /* dV, Vx, Vxx */
    {
      double s0 = 0, s1 = 0;
      for(int r = 0; r < m; r++)
      {
        double s = 0;
        for(int q = 0; q < m; q++) s += Quu[r * m + q] * ki[q];
        t4[r] = s; /* Quu k */
        s0 += ki[r] * Qu[r];
        s1 += ki[r] * s;
      }
      d->dV[0] += s0;
      d->dV[1] += 0.5 * s1;
      for(int a = 0; a < S; a++)
      {
        double s = Qx[a];
        for(int r = 0; r < m; r++) s += Ki[r * S + a] * t4[r] + Ki[r * S + a] * Qu[r] + Qxu[a * M + r] * ki[r];
        vxn[a] = s;
      }
      /* T2 (S x m) = K' Quu */
      for(int a = 0; a < S; a++)
        for(int r = 0; r < m; r++)
        {
          double s = 0;
          for(int q = 0; q < m; q++) s += Ki[q * S + a] * Quu[q * m + r];
          T2[a * M + r] = s;
        }
      for(int a = 0; a < S; a++)
        for(int b = 0; b < S; b++)
        {
          double s = Qxx[a * S + b];
          for(int r = 0; r < m; r++)
            s += T2[a * M + r] * Ki[r * S + b] + Ki[r * S + a] * Qxu[b * M + r] + Qxu[a * M + r] * Ki[r * S + b];
          T1[a * S + b] = s;
        }
      for(int a = 0; a < S; a++)
      {
        Vx[a] = vxn[a];
        for(int b = 0; b < S; b++) Vxx[a * S + b] = 0.5 * (T1[a * S + b] + T1[b * S + a]);
      }
    }
  }
  free(Vx);
  free(is_free);
  return ok;
}

static void forward_pass(ddp_t * d, double alpha)
{
  const int S = d->S, N = d->N, M = d->M;
  double lo[256], hi[256];
  memcpy(d->xc, d->x, sizeof(double) * S);
  for(int i = 0; i < N; i++)
  {
    const int m = d->m[i];
    const double * xi = d->x + (size_t)i * S, * ui = d->u + (size_t)i * M;
    double * xn = d->xc + (size_t)i * S, * un = d->uc + (size_t)i * M;
    const double * ki = d->k + (size_t)i * M, * Ki = d->K + (size_t)i * M * S;
    if(m > 0 && d->c->with_input_constraint) d->p->input_limits(d->p->user, i, lo, hi);
    for(int r = 0; r < M; r++) un[r] = 0;
    for(int r = 0; r < m; r++)
    {
      double s = ui[r] + alpha * ki[r];
      for(int a = 0; a < S; a++) s += Ki[r * S + a] * (xn[a] - xi[a]);
      if(d->c->with_input_constraint) s = fmin(fmax(s, lo[r]), hi[r]);
      un[r] = s;
    }
    d->p->state_eq(d->p->user, i, xn, un, d->xc + (size_t)(i + 1) * S);
  }
  d->costc = rollout_cost(d, d->xc, d->uc);
}

static void derivatives(ddp_t * d)
{
  const int S = d->S, N = d->N, M = d->M;
  for(int i = 0; i < N; i++)
  {
    d->p->state_eq_deriv(d->p->user, i, d->x + (size_t)i * S, d->u + (size_t)i * M, d->Fx + (size_t)i * S * S,
                         d->Fu + (size_t)i * S * M);
    d->p->running_cost_deriv(d->p->user, i, d->x + (size_t)i * S, d->u + (size_t)i * M, d->Lx + (size_t)i * S,
                             d->Lu + (size_t)i * M, d->Lxx + (size_t)i * S * S, d->Luu + (size_t)i * M * M,
                             d->Lxu + (size_t)i * S * M);
  }
  d->p->terminal_cost_deriv(d->p->user, d->x + (size_t)N * S, d->Vx_T, d->Vxx_T);
}

int oracle_ddp_solve(const oracle_ddp_problem_t * p, const oracle_ddp_config_t * c, const double * x0,
                     const double * u_init, double * x_out, double * u_out, oracle_ddp_result_t * res)
{
  ddp_t d;
  memset(&d, 0, sizeof(d));
  const int S = p->S, N = p->N, M = p->M;
  d.p = p;
  d.c = c;
  d.S = S;
  d.N = N;
  d.M = M;
  size_t nx = (size_t)(N + 1) * S, nu = (size_t)N * M;
  d.x = (double *)calloc(nx * 2 + nu * 2, sizeof(double));
  d.xc = d.x + nx;
  d.u = d.xc + nx;
  d.uc = d.u + nu;
  d.Fx = (double *)calloc((size_t)N * (S * S * 2 + S * M * 2 + S + M + M * M) + S + S * S, sizeof(double));
  d.Fu = d.Fx + (size_t)N * S * S;
  d.Lx = d.Fu + (size_t)N * S * M;
  d.Lu = d.Lx + (size_t)N * S;
  d.Lxx = d.Lu + (size_t)N * M;
  d.Luu = d.Lxx + (size_t)N * S * S;
  d.Lxu = d.Luu + (size_t)N * M * M;
  d.Vx_T = d.Lxu + (size_t)N * S * M;
  d.Vxx_T = d.Vx_T + S;
  d.k = (double *)calloc(nu + nu * S, sizeof(double));
  d.K = d.k + nu;
  d.m = (int *)calloc(N, sizeof(int));
  for(int i = 0; i < N; i++) d.m[i] = p->input_dim(p->user, i);

  d.lambda = c->initial_lambda;
  d.dlambda = c->initial_dlambda;
  memcpy(d.x, x0, sizeof(double) * S);
  for(int i = 0; i < N; i++)
  {
    for(int r = 0; r < d.m[i]; r++) d.u[(size_t)i * M + r] = u_init ? u_init[(size_t)i * M + r] : 0.0;
    p->state_eq(p->user, i, d.x + (size_t)i * S, d.u + (size_t)i * M, d.x + (size_t)(i + 1) * S);
  }
  d.cost = rollout_cost(&d, d.x, d.u);
  int warm_replaced = 0;
  if(c->warm_start_guard && u_init)
  {
    /* warm-start guard (not nmpc_ddp; ccc_oracle.h): keep the warm start only if its rollout is no worse than the
     * rollout of zero inputs, the cold start of src/DdpCentroidal.cpp:221-229 */
    memcpy(d.xc, x0, sizeof(double) * S);
    memset(d.uc, 0, sizeof(double) * nu);
    for(int i = 0; i < N; i++)
      p->state_eq(p->user, i, d.xc + (size_t)i * S, d.uc + (size_t)i * M, d.xc + (size_t)(i + 1) * S);
    const double cold = rollout_cost(&d, d.xc, d.uc);
    if(!(d.cost <= cold))
    {
      memcpy(d.x, d.xc, sizeof(double) * nx);
      memcpy(d.u, d.uc, sizeof(double) * nu);
      d.cost = cold;
      warm_replaced = 1;
    }
  }
  double initial_cost = d.cost;

  int iter = 0, status = 0, need_deriv = 1, n_accept = 0;
  /* status: 0 max_iter reached, 1 gradient small, 2 cost change small, -1 lambda > lambda_max */
  for(iter = 1; iter <= c->max_iter; iter++)
  {
    if(need_deriv)
    {
      derivatives(&d);
      need_deriv = 0;
    }
    int bp_ok = 0;
    for(;;)
    {
      if(backward_pass(&d))
      {
        bp_ok = 1;
        break;
      }
      increase_lambda(&d);
      if(d.lambda > c->lambda_max) break;
    }
    if(!bp_ok)
    {
      status = -1;
      break;
    }
    double g = 0;
    for(int i = 0; i < N; i++)
    {
      double mx = 0;
      for(int r = 0; r < d.m[i]; r++)
      {
        double v = fabs(d.k[(size_t)i * M + r]) / (fabs(d.u[(size_t)i * M + r]) + 1.0);
        if(v > mx) mx = v;
      }
      g += mx;
    }
    g /= N;
    if(g < c->k_rel_norm_thre && d.lambda < c->lambda_thre)
    {
      decrease_lambda(&d);
      status = 1;
      break;
    }
    int accepted = 0;
    double actual = 0;
    for(int a = 0; a < 11; a++)
    {
      double alpha = c->alpha_list[a];
      forward_pass(&d, alpha);
      actual = d.cost - d.costc;
      double expected = -alpha * (d.dV[0] + alpha * d.dV[1]);
      double ratio = expected > 0 ? actual / expected : (actual > 0 ? 1.0 : (actual < 0 ? -1.0 : 0.0));
      if(oracle_ddp_trace)
        fprintf(stderr, "  iter %d alpha %.4f cost %.6g -> %.6g expected %.4g ratio %.4g lambda %.3g\n", iter, alpha,
                d.cost, d.costc, expected, ratio, d.lambda);
      if(ratio > c->cost_update_ratio_thre)
      {
        accepted = 1;
        break;
      }
    }
    if(accepted)
    {
      decrease_lambda(&d);
      memcpy(d.x, d.xc, sizeof(double) * nx);
      memcpy(d.u, d.uc, sizeof(double) * nu);
      d.cost = d.costc;
      need_deriv = 1;
      n_accept++;
      if(actual < c->cost_update_thre)
      {
        status = 2;
        break;
      }
    }
    else
    {
      increase_lambda(&d);
      if(d.lambda > c->lambda_max)
      {
        status = -1;
        break;
      }
    }
  }
  if(iter > c->max_iter) iter = c->max_iter;
  if(x_out) memcpy(x_out, d.x, sizeof(double) * nx);
  if(u_out) memcpy(u_out, d.u, sizeof(double) * nu);
  if(res)
  {
    res->iters = iter;
    res->status = status;
    res->cost = d.cost;
    res->initial_cost = initial_cost;
    res->lambda = d.lambda;
    res->accepted = n_accept;
    res->warm_replaced = warm_replaced;
  }
  free(d.x);
  free(d.Fx);
  free(d.k);
  free(d.m);
  return status;
}
