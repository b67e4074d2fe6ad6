/* ddp_models.c -- DDP problem definitions of CCC::DdpCentroidal and CCC::DdpSingleRigidBody, restated in C
 * (TEST INFRASTRUCTURE ONLY, see ccc_oracle.h).  Follows, function by function:
 *   /root/reference/src/DdpCentroidal.cpp:21-30     inputDim            :32-64   stateEq
 *                                      :66-83     runningCost/terminalCost   :85-121  calcStateEqDeriv
 *                                      :123-177   cost derivatives           :193-237 constructor / planOnce
 *   /root/reference/src/DdpSingleRigidBody.cpp:26-38 matAngularVelToEulerDot  :52-91   stateEq
 *                                      :93-112    costs                      :114-185 calcStateEqDeriv
 *                                      :187-243   cost derivatives           :260-307 constructor / planOnce
 * The contact list arrives flattened in the contact -> vertex -> ridge order of DdpCentroidal.cpp:49-60.
 */
#include "ccc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_G 9.80665

static void cross3(const double * a, const double * b, double * c)
{
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* 3x3 SPD solve by Cholesky (Eigen::LLT<Matrix3d>::solve, DdpSingleRigidBody.cpp:88,122-123) */
static void llt3_solve(const double * I, const double * b, double * x)
{
  double L[9] = {0};
  L[0] = sqrt(I[0]);
  L[3] = I[3] / L[0];
  L[6] = I[6] / L[0];
  L[4] = sqrt(I[4] - L[3] * L[3]);
  L[7] = (I[7] - L[6] * L[3]) / L[4];
  L[8] = sqrt(I[8] - L[6] * L[6] - L[7] * L[7]);
  double y0 = b[0] / L[0];
  double y1 = (b[1] - L[3] * y0) / L[4];
  double y2 = (b[2] - L[6] * y0 - L[7] * y1) / L[8];
  x[2] = y2 / L[8];
  x[1] = (y1 - L[7] * x[2]) / L[4];
  x[0] = (y0 - L[3] * x[1] - L[6] * x[2]) / L[0];
}

static int mdl_input_dim(void * user, int step)
{
  const oracle_ddp_model_t * m = (const oracle_ddp_model_t *)user;
  return m->phase_dim[m->step_phase[step]];
}

static void mdl_limits(void * user, int step, double * lo, double * hi)
{
  /* setInputLimitsFunc lambda, src/DdpCentroidal.cpp:202-210 */
  const oracle_ddp_model_t * m = (const oracle_ddp_model_t *)user;
  int dim = mdl_input_dim(user, step);
  for(int r = 0; r < dim; r++)
  {
    lo[r] = m->force_lo;
    hi[r] = m->force_hi;
  }
}


/* Deterministic sin/cos (Cody-Waite reduction by pi/2 + the classic fdlibm minimax kernels), accurate to ~1 ulp for
 * |x| < 1e3.  The reference calls std::sin / std::cos (src/DdpSingleRigidBody.cpp:30-33,128-131); glibc and the GPU's
 * device libm differ from each other in the last ulp, and DDP's discrete decisions can amplify one ulp into a different
 * iterate.  Oracle and HIP kernel therefore both evaluate THIS restatement (same operations, no FMA contraction), which
 * keeps the single-rigid-body model bit-reproducible across the two; tests check it against libm to 2 ulp. */
static void det_sincos(double x, double * s, double * c)
{
  const double fn = floor(x * 6.36619772367581382433e-01 + 0.5);
  const int n = (int)fn;
  double r = x - fn * 1.57079632673412561417e+00;
  r = r - fn * 6.07710050650619224932e-11;
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01
                    + z * (8.33333333332248946124e-03
                           + z * (-1.98412698298579493134e-04
                                  + z * (2.75573137070700676789e-06
                                         + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double pc = 4.16666666666666019037e-02
                    + z * (-1.38888888888741095749e-03
                           + z * (2.48015872894767294178e-05
                                  + z * (-2.75573143513906633035e-07
                                         + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double sn = r + r * z * ps;
  const double cs = 1.0 - (0.5 * z - z * z * pc);
  switch(n & 3)
  {
    case 0:
      *s = sn;
      *c = cs;
      break;
    case 1:
      *s = cs;
      *c = -sn;
      break;
    case 2:
      *s = -sn;
      *c = -cs;
      break;
    default:
      *s = -cs;
      *c = sn;
      break;
  }
}

/* matAngularVelToEulerDot, src/DdpSingleRigidBody.cpp:26-38 (row-major 3x3) */
static void euler_trans(const double * ori, double * K)
{
  double ca, sa, cb, sb;
  det_sincos(ori[0], &sa, &ca);
  det_sincos(ori[1], &sb, &cb);
  K[0] = (ca * sb) / cb;
  K[1] = (sb * sa) / cb;
  K[2] = 1.0;
  K[3] = -1 * sa;
  K[4] = ca;
  K[5] = 0.0;
  K[6] = ca / cb;
  K[7] = sa / cb;
  K[8] = 0.0;
}

static void mdl_state_eq(void * user, int step, const double * x, const double * u, double * xn)
{
  const oracle_ddp_model_t * m = (const oracle_ddp_model_t *)user;
  const int ph = m->step_phase[step], dim = m->phase_dim[ph];
  const double * V = m->phase_vertex + (size_t)ph * m->M * 3;
  const double * R = m->phase_ridge + (size_t)ph * m->M * 3;
  if(m->model == 0)
  {
    /* src/DdpCentroidal.cpp:32-64 */
    double xd[9];
    for(int a = 0; a < 3; a++) xd[a] = x[3 + a] / m->mass;
    xd[3] = 0;
    xd[4] = 0;
    xd[5] = -1 * m->mass * ORACLE_G;
    xd[6] = xd[7] = xd[8] = 0;
    for(int r = 0; r < dim; r++)
    {
      double d[3] = {V[r * 3] - x[0], V[r * 3 + 1] - x[1], V[r * 3 + 2] - x[2]}, c[3];
      cross3(d, R + r * 3, c);
      for(int a = 0; a < 3; a++)
      {
        xd[3 + a] += u[r] * R[r * 3 + a];
        xd[6 + a] += u[r] * c[a];
      }
    }
    for(int a = 0; a < 9; a++) xn[a] = x[a] + m->dt * xd[a];
  }
  else
  {
    /* src/DdpSingleRigidBody.cpp:52-91; :56-57: motion_param_func_(t).inertia_mat is the STEP's */
    const double * I = m->inertia + (m->inertia_per_phase ? (size_t)ph * 9 : 0);
    double xd[12], K[9];
    const double * w = x + 9;
    for(int a = 0; a < 3; a++) xd[a] = x[6 + a];
    euler_trans(x + 3, K);
    for(int a = 0; a < 3; a++) xd[3 + a] = K[a * 3] * w[0] + K[a * 3 + 1] * w[1] + K[a * 3 + 2] * w[2];
    xd[6] = 0;
    xd[7] = 0;
    xd[8] = -1 * ORACLE_G;
    double Iw[3], wd[3];
    for(int a = 0; a < 3; a++) Iw[a] = I[a * 3] * w[0] + I[a * 3 + 1] * w[1] + I[a * 3 + 2] * w[2];
    cross3(w, Iw, wd);
    for(int a = 0; a < 3; a++) wd[a] = -1 * wd[a];
    for(int r = 0; r < dim; r++)
    {
      double d[3] = {V[r * 3] - x[0], V[r * 3 + 1] - x[1], V[r * 3 + 2] - x[2]}, c[3];
      cross3(d, R + r * 3, c);
      for(int a = 0; a < 3; a++)
      {
        xd[6 + a] += u[r] * R[r * 3 + a] / m->mass;
        wd[a] += u[r] * c[a];
      }
    }
    llt3_solve(I, wd, xd + 9);
    for(int a = 0; a < 12; a++) xn[a] = x[a] + m->dt * xd[a];
  }
}

static const double * ref_of(const oracle_ddp_model_t * m, int step, double * buf)
{
  /* stacked reference of the weighted state entries: Cen [pos, 0, 0]; SRB [pos, ori, 0, 0] */
  int S = m->model == 0 ? 9 : 12;
  for(int a = 0; a < S; a++) buf[a] = 0;
  for(int a = 0; a < 3; a++) buf[a] = m->ref_pos[(size_t)step * 3 + a];
  if(m->model == 1)
    for(int a = 0; a < 3; a++) buf[3 + a] = m->ref_ori[(size_t)step * 3 + a];
  return buf;
}

static double mdl_running_cost(void * user, int step, const double * x, const double * u)
{
  /* src/DdpCentroidal.cpp:66-74, src/DdpSingleRigidBody.cpp:93-103 */
  const oracle_ddp_model_t * m = (const oracle_ddp_model_t *)user;
  int S = m->model == 0 ? 9 : 12, dim = mdl_input_dim(user, step);
  double rb[12], c = 0, un = 0;
  const double * ref = ref_of(m, step, rb);
  for(int a = 0; a < S; a++) c += 0.5 * m->w_run[a] * (x[a] - ref[a]) * (x[a] - ref[a]);
  for(int r = 0; r < dim; r++) un += u[r] * u[r];
  return c + 0.5 * m->w_force * un;
}

static double mdl_terminal_cost(void * user, const double * x)
{
  /* src/DdpCentroidal.cpp:76-83, src/DdpSingleRigidBody.cpp:105-112; sampled at step N */
  const oracle_ddp_model_t * m = (const oracle_ddp_model_t *)user;
  int S = m->model == 0 ? 9 : 12;
  double rb[12], c = 0;
  const double * ref = ref_of(m, m->N, rb);
  for(int a = 0; a < S; a++) c += 0.5 * m->w_term[a] * (x[a] - ref[a]) * (x[a] - ref[a]);
  return c;
}

static void mdl_state_eq_deriv(void * user, int step, const double * x, const double * u, double * Fx, double * Fu)
{
  const oracle_ddp_model_t * m = (const oracle_ddp_model_t *)user;
  const int ph = m->step_phase[step], dim = m->phase_dim[ph], M = m->M;
  const double * V = m->phase_vertex + (size_t)ph * M * 3;
  const double * R = m->phase_ridge + (size_t)ph * M * 3;
  const int S = m->model == 0 ? 9 : 12;
  memset(Fx, 0, sizeof(double) * S * S);
  memset(Fu, 0, sizeof(double) * S * M);
  double tf[3] = {0, 0, 0};
  if(m->model == 0)
  {
    /* src/DdpCentroidal.cpp:85-121 */
    for(int a = 0; a < 3; a++) Fx[a * S + 3 + a] = 1 / m->mass;
    for(int r = 0; r < dim; r++)
    {
      double d[3] = {V[r * 3] - x[0], V[r * 3 + 1] - x[1], V[r * 3 + 2] - x[2]}, c[3];
      cross3(d, R + r * 3, c);
      for(int a = 0; a < 3; a++)
      {
        tf[a] += u[r] * R[r * 3 + a];
        Fu[(3 + a) * M + r] = R[r * 3 + a];
        Fu[(6 + a) * M + r] = c[a];
      }
    }
    /* crossMat(totalForce) into block (6, 0) */
    Fx[6 * S + 1] = -tf[2];
    Fx[6 * S + 2] = tf[1];
    Fx[7 * S + 0] = tf[2];
    Fx[7 * S + 2] = -tf[0];
    Fx[8 * S + 0] = -tf[1];
    Fx[8 * S + 1] = tf[0];
  }
  else
  {
    /* src/DdpSingleRigidBody.cpp:114-185; :120-123: the step's inertia matrix and its factor */
    const double * I = m->inertia + (m->inertia_per_phase ? (size_t)ph * 9 : 0);
    const double * ori = x + 3;
    for(int a = 0; a < 3; a++) Fx[a * S + 6 + a] = 1.0;
    double K[9];
    euler_trans(ori, K);
    for(int a = 0; a < 3; a++)
      for(int b = 0; b < 3; b++) Fx[(3 + a) * S + 9 + b] = K[a * 3 + b];
    double w1 = x[9], w2 = x[10], w3 = x[11];
    double ca, sa, cb, sb;
    det_sincos(ori[0], &sa, &ca);
    det_sincos(ori[1], &sb, &cb);
    double cb2 = cb * cb, sb2 = sb * sb;
    double I11 = I[0], I12 = I[1], I13 = I[2], I22 = I[4], I23 = I[5], I33 = I[8];
    /* block (3,3), column 0 and column 1 (column 2 stays zero) */
    Fx[3 * S + 3] = -w1 * sa * sb / cb + w2 * sb * ca / cb;
    Fx[4 * S + 3] = -w1 * ca - w2 * sa;
    Fx[5 * S + 3] = -w1 * sa / cb + w2 * ca / cb;
    Fx[3 * S + 4] = w1 * sb2 * ca / cb2 + w1 * ca + w2 * sa * sb2 / cb2 + w2 * sa;
    Fx[4 * S + 4] = 0.0;
    Fx[5 * S + 4] = w1 * sb * ca / cb2 + w2 * sa * sb / cb2;
    /* block (9,9) = I^-1 d(-w x I w)/dw  (Eigen operator<< fills row by row) */
    double D[9] = {I12 * w3 - I13 * w2,
                   -I13 * w1 + I22 * w3 - 2 * I23 * w2 - I33 * w3,
                   I12 * w1 + I22 * w2 + 2 * I23 * w3 - I33 * w2,
                   -I11 * w3 + 2 * I13 * w1 + I23 * w2 + I33 * w3,
                   -I12 * w3 + I23 * w1,
                   -I11 * w1 - I12 * w2 - 2 * I13 * w3 + I33 * w1,
                   I11 * w2 - 2 * I12 * w1 - I22 * w2 - I23 * w3,
                   I11 * w1 + 2 * I12 * w2 + I13 * w3 - I22 * w1,
                   I13 * w2 - I23 * w1};
    for(int b = 0; b < 3; b++)
    {
      double col[3] = {D[b], D[3 + b], D[6 + b]}, sol[3];
      llt3_solve(I, col, sol);
      for(int a = 0; a < 3; a++) Fx[(9 + a) * S + 9 + b] = sol[a];
    }
    for(int r = 0; r < dim; r++)
    {
      double d[3] = {V[r * 3] - x[0], V[r * 3 + 1] - x[1], V[r * 3 + 2] - x[2]}, c[3], sol[3];
      cross3(d, R + r * 3, c);
      llt3_solve(I, c, sol);
      for(int a = 0; a < 3; a++)
      {
        tf[a] += u[r] * R[r * 3 + a];
        Fu[(6 + a) * M + r] = R[r * 3 + a] / m->mass;
        Fu[(9 + a) * M + r] = sol[a];
      }
    }
    double CM[9] = {0, -tf[2], tf[1], tf[2], 0, -tf[0], -tf[1], tf[0], 0};
    for(int b = 0; b < 3; b++)
    {
      double col[3] = {CM[b], CM[3 + b], CM[6 + b]}, sol[3];
      llt3_solve(I, col, sol);
      for(int a = 0; a < 3; a++) Fx[(9 + a) * S + b] = sol[a];
    }
  }
  for(int a = 0; a < S * S; a++) Fx[a] *= m->dt;
  for(int a = 0; a < S; a++) Fx[a * S + a] += 1.0;
  for(int a = 0; a < S; a++)
    for(int r = 0; r < dim; r++) Fu[a * M + r] *= m->dt;
}

static void mdl_running_cost_deriv(void * user, int step, const double * x, const double * u, double * Lx,
                                   double * Lu, double * Lxx, double * Luu, double * Lxu)
{
  /* src/DdpCentroidal.cpp:123-154, src/DdpSingleRigidBody.cpp:187-220 */
  const oracle_ddp_model_t * m = (const oracle_ddp_model_t *)user;
  int S = m->model == 0 ? 9 : 12, dim = mdl_input_dim(user, step), M = m->M;
  double rb[12];
  const double * ref = ref_of(m, step, rb);
  memset(Lxx, 0, sizeof(double) * S * S);
  memset(Luu, 0, sizeof(double) * M * M);
  memset(Lxu, 0, sizeof(double) * S * M);
  for(int a = 0; a < S; a++)
  {
    Lx[a] = m->w_run[a] * (x[a] - ref[a]);
    Lxx[a * S + a] = m->w_run[a];
  }
  for(int r = 0; r < M; r++) Lu[r] = 0;
  for(int r = 0; r < dim; r++)
  {
    Lu[r] = m->w_force * u[r];
    Luu[r * M + r] = m->w_force;
  }
}

static void mdl_terminal_cost_deriv(void * user, const double * x, double * Vx, double * Vxx)
{
  /* src/DdpCentroidal.cpp:156-177, src/DdpSingleRigidBody.cpp:222-243 */
  const oracle_ddp_model_t * m = (const oracle_ddp_model_t *)user;
  int S = m->model == 0 ? 9 : 12;
  double rb[12];
  const double * ref = ref_of(m, m->N, rb);
  memset(Vxx, 0, sizeof(double) * S * S);
  for(int a = 0; a < S; a++)
  {
    Vx[a] = m->w_term[a] * (x[a] - ref[a]);
    Vxx[a * S + a] = m->w_term[a];
  }
}

void oracle_ddp_model_problem(const oracle_ddp_model_t * mdl, oracle_ddp_problem_t * prob)
{
  prob->S = mdl->model == 0 ? 9 : 12;
  prob->N = mdl->N;
  prob->M = mdl->M;
  prob->user = (void *)mdl;
  prob->input_dim = mdl_input_dim;
  prob->state_eq = mdl_state_eq;
  prob->running_cost = mdl_running_cost;
  prob->terminal_cost = mdl_terminal_cost;
  prob->state_eq_deriv = mdl_state_eq_deriv;
  prob->running_cost_deriv = mdl_running_cost_deriv;
  prob->terminal_cost_deriv = mdl_terminal_cost_deriv;
  prob->input_limits = mdl_limits;
}

int oracle_ddp_plan_batch(const oracle_ddp_model_t * shared, const oracle_ddp_config_t * cfg, long n,
                          const int * phase_dim, const double * phase_vertex, const double * phase_ridge,
                          const int * step_phase, const double * ref_pos, const double * ref_ori,
                          const double * inertia, const double * x0, const double * u_init, double * u_out,
                          double * x_out, int * iters, int * status, double * cost, int nthreads)
{
  const int S = shared->model == 0 ? 9 : 12, N = shared->N, P = shared->P, M = shared->M;
  int worst = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 1 ? nthreads : 1) reduction(min : worst)
#endif
  for(long b = 0; b < n; b++)
  {
    oracle_ddp_model_t mdl = *shared;
    mdl.phase_dim = phase_dim + (size_t)b * P;
    mdl.phase_vertex = phase_vertex + (size_t)b * P * M * 3;
    mdl.phase_ridge = phase_ridge + (size_t)b * P * M * 3;
    mdl.step_phase = step_phase + (size_t)b * N;
    mdl.ref_pos = ref_pos + (size_t)b * (N + 1) * 3;
    mdl.ref_ori = ref_ori ? ref_ori + (size_t)b * (N + 1) * 3 : NULL;
    mdl.inertia = inertia ? inertia + (size_t)b * 9 * (shared->inertia_per_phase ? P : 1) : NULL;
    oracle_ddp_problem_t prob;
    oracle_ddp_model_problem(&mdl, &prob);
    oracle_ddp_result_t res;
    int st;
    if(cfg->arith == 1)
      st = oracle_ddp_solve_tile(&mdl, cfg, x0 + (size_t)b * S, u_init ? u_init + (size_t)b * N * M : NULL,
                                 x_out ? x_out + (size_t)b * (N + 1) * S : NULL, u_out + (size_t)b * N * M, &res);
    else
      st = oracle_ddp_solve(&prob, cfg, x0 + (size_t)b * S, u_init ? u_init + (size_t)b * N * M : NULL,
                            x_out ? x_out + (size_t)b * (N + 1) * S : NULL, u_out + (size_t)b * N * M, &res);
    if(iters) iters[b] = res.iters;
    if(status) status[b] = res.warm_replaced ? (ORACLE_DDP_STATUS_WARM_REPLACED | (st & 0xff)) : st;
    if(cost) cost[b] = res.cost;
    if(st < worst) worst = st;
  }
  (void)nthreads;
  return worst;
}

/* test access to the problem callbacks (finite-difference derivative checks of the reference tests,
 * tests/src/TestDdpCentroidal.cpp:176-284, tests/src/TestDdpSingleRigidBody.cpp:197-308) */
void oracle_ddp_model_eval(const oracle_ddp_model_t * mdl, int step, const double * x, const double * u,
                           double * x_next, double * Fx, double * Fu, double * run_cost, double * term_cost,
                           double * Lx, double * Lu, double * Vx)
{
  const int S = mdl->model == 0 ? 9 : 12, M = mdl->M;
  if(x_next) mdl_state_eq((void *)mdl, step, x, u, x_next);
  if(Fx && Fu) mdl_state_eq_deriv((void *)mdl, step, x, u, Fx, Fu);
  if(run_cost) *run_cost = mdl_running_cost((void *)mdl, step, x, u);
  if(term_cost) *term_cost = mdl_terminal_cost((void *)mdl, x);
  if(Lx && Lu)
  {
    double * Lxx = (double *)malloc(sizeof(double) * (S * S + M * M + S * M));
    mdl_running_cost_deriv((void *)mdl, step, x, u, Lx, Lu, Lxx, Lxx + S * S, Lxx + S * S + M * M);
    free(Lxx);
  }
  if(Vx)
  {
    double * Vxx = (double *)malloc(sizeof(double) * S * S);
    mdl_terminal_cost_deriv((void *)mdl, x, Vx, Vxx);
    free(Vxx);
  }
}

void oracle_det_sincos(double x, double * s, double * c)
{
  det_sincos(x, s, c);
}
