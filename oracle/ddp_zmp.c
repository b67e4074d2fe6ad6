/* ddp_zmp.c -- CPU restatement of CCC::DdpZmp (6-state / 3-input CoM-ZMP DDP, no input constraint).
 *
 * TEST INFRASTRUCTURE ONLY (see ccc_oracle.h): the parity checker and CPU baseline of the HIP path; PARITY UNPINNED like
 * the rest of the oracle (the reference holds no golden vectors; nmpc_ddp is an absent dependency whose published
 * algorithm ddp.c restates).
 *
 * Follows (file:line under /root/reference):
 *   src/DdpZmp.cpp:8-19      stateEq        x+ = x + dt [vx, (cx - zx) fz / (m (cz - zz)), vy, (cy - zy) fz / (m (cz - zz)),
 *                                                         vz, fz / m - g]
 *   src/DdpZmp.cpp:21-28     runningCost    w_cz/2 (cz - cz_ref)^2 + w_zmp/2 |u_xy - zmp_ref_xy|^2 + w_fz/2 (fz - m g)^2
 *   src/DdpZmp.cpp:30-43     terminalCost
 *   src/DdpZmp.cpp:45-72     calcStateEqDeriv
 *   src/DdpZmp.cpp:86-110    calcRunningCostDeriv (second order)
 *   src/DdpZmp.cpp:126-146   calcTerminalCostDeriv (second order)
 *   src/DdpZmp.cpp:156-174   planOnce: solve from InitialParam::toState() with u_list (zeros when empty); zmp = u0[0:2],
 *                            force_z = u0[2]
 *   include/CCC/DdpZmp.h:277-282  constructor: nmpc_ddp defaults (no input constraint), horizon_steps
 * The callbacks' time argument is the horizon step: RefData is sampled at current_time + i dt, i = 0..N.
 * State x = [cx, vx, cy, vy, cz, vz], input u = [zmp_x, zmp_y, f_z].  The arithmetic (operation order included) is
 * the specification the HIP kernel csrc/ddpzmp.hip shares.
 */
#include "ccc_oracle.h"

#include <math.h>
#include <string.h>

#define ZG 9.80665 /* CCC::constants::g */

typedef struct
{
  const oracle_ddpzmp_params_t * prm;
  const double * ref; /* [N+1][4]: zmp x, y, z, com_z */
} zmodel_t;

static int z_input_dim(void * user, int step)
{
  (void)user;
  (void)step;
  return 3;
}

static void z_state_eq(void * user, int step, const double * x, const double * u, double * xn)
{
  const zmodel_t * m = (const zmodel_t *)user;
  const double * r = m->ref + (size_t)step * 4;
  const double mass = m->prm->mass, dt = m->prm->dt;
  const double den = mass * (x[4] - r[2]);
  double xd[6];
  xd[0] = x[1];
  xd[1] = (x[0] - u[0]) * u[2] / den;
  xd[2] = x[3];
  xd[3] = (x[2] - u[1]) * u[2] / den;
  xd[4] = x[5];
  xd[5] = u[2] / mass - ZG;
  for(int a = 0; a < 6; a++) xn[a] = x[a] + dt * xd[a];
}

static double z_running_cost(void * user, int step, const double * x, const double * u)
{
  const zmodel_t * m = (const zmodel_t *)user;
  const double * r = m->ref + (size_t)step * 4;
  const oracle_ddpzmp_params_t * p = m->prm;
  const double ez = x[4] - r[3], e0 = u[0] - r[0], e1 = u[1] - r[1], ef = u[2] - p->mass * ZG;
  return p->w_run_com_z * 0.5 * (ez * ez) + p->w_run_zmp * 0.5 * (e0 * e0 + e1 * e1) + p->w_run_force_z * 0.5 * (ef * ef);
}

static double z_terminal_cost_at(const zmodel_t * m, int step, const double * x)
{
  const double * r = m->ref + (size_t)step * 4;
  const oracle_ddpzmp_params_t * p = m->prm;
  const double e0 = x[0] - r[0], e1 = x[2] - r[1], ez = x[4] - r[3];
  return p->w_term_com_xy * 0.5 * (e0 * e0 + e1 * e1) + p->w_term_com_z * 0.5 * (ez * ez)
         + p->w_term_com_vel * 0.5 * ((x[1] * x[1] + x[3] * x[3]) + x[5] * x[5]);
}

static double z_terminal_cost(void * user, const double * x)
{
  const zmodel_t * m = (const zmodel_t *)user;
  return z_terminal_cost_at(m, m->prm->N, x);
}

static void z_state_eq_deriv(void * user, int step, const double * x, const double * u, double * Fx, double * Fu)
{
  const zmodel_t * m = (const zmodel_t *)user;
  const double * r = m->ref + (size_t)step * 4;
  const double mass = m->prm->mass, dt = m->prm->dt;
  const double d = x[4] - r[2];
  const double den = mass * d, den2 = mass * (d * d);
  memset(Fx, 0, sizeof(double) * 36);
  Fx[0 * 6 + 1] = 1;
  Fx[1 * 6 + 0] = u[2] / den;
  Fx[1 * 6 + 4] = -1 * (x[0] - u[0]) * u[2] / den2;
  Fx[2 * 6 + 3] = 1;
  Fx[3 * 6 + 2] = u[2] / den;
  Fx[3 * 6 + 4] = -1 * (x[2] - u[1]) * u[2] / den2;
  Fx[4 * 6 + 5] = 1;
  for(int a = 0; a < 36; a++) Fx[a] *= dt;
  for(int a = 0; a < 6; a++) Fx[a * 6 + a] += 1.0;
  memset(Fu, 0, sizeof(double) * 18); /* 6 x 3, row stride M = 3 */
  Fu[1 * 3 + 0] = -1 * u[2] / den;
  Fu[1 * 3 + 2] = (x[0] - u[0]) / den;
  Fu[3 * 3 + 1] = -1 * u[2] / den;
  Fu[3 * 3 + 2] = (x[2] - u[1]) / den;
  Fu[5 * 3 + 2] = 1 / mass;
  for(int a = 0; a < 18; a++) Fu[a] *= dt;
}

static void z_running_cost_deriv(void * user, int step, const double * x, const double * u, double * Lx, double * Lu,
                                 double * Lxx, double * Luu, double * Lxu)
{
  const zmodel_t * m = (const zmodel_t *)user;
  const double * r = m->ref + (size_t)step * 4;
  const oracle_ddpzmp_params_t * p = m->prm;
  memset(Lx, 0, sizeof(double) * 6);
  Lx[4] = p->w_run_com_z * (x[4] - r[3]);
  Lu[0] = p->w_run_zmp * (u[0] - r[0]);
  Lu[1] = p->w_run_zmp * (u[1] - r[1]);
  Lu[2] = p->w_run_force_z * (u[2] - p->mass * ZG);
  memset(Lxx, 0, sizeof(double) * 36);
  Lxx[4 * 6 + 4] = p->w_run_com_z;
  memset(Luu, 0, sizeof(double) * 9);
  Luu[0] = p->w_run_zmp;
  Luu[4] = p->w_run_zmp;
  Luu[8] = p->w_run_force_z;
  memset(Lxu, 0, sizeof(double) * 18);
}

static void z_terminal_cost_deriv(void * user, const double * x, double * Vx, double * Vxx)
{
  const zmodel_t * m = (const zmodel_t *)user;
  const oracle_ddpzmp_params_t * p = m->prm;
  const double * r = m->ref + (size_t)p->N * 4;
  Vx[0] = p->w_term_com_xy * (x[0] - r[0]);
  Vx[1] = p->w_term_com_vel * x[1];
  Vx[2] = p->w_term_com_xy * (x[2] - r[1]);
  Vx[3] = p->w_term_com_vel * x[3];
  Vx[4] = p->w_term_com_z * (x[4] - r[3]);
  Vx[5] = p->w_term_com_vel * x[5];
  memset(Vxx, 0, sizeof(double) * 36);
  Vxx[0] = p->w_term_com_xy;
  Vxx[7] = p->w_term_com_vel;
  Vxx[14] = p->w_term_com_xy;
  Vxx[21] = p->w_term_com_vel;
  Vxx[28] = p->w_term_com_z;
  Vxx[35] = p->w_term_com_vel;
}

static void z_problem(const zmodel_t * m, oracle_ddp_problem_t * prob)
{
  memset(prob, 0, sizeof(*prob));
  prob->S = 6;
  prob->N = m->prm->N;
  prob->M = 3;
  prob->user = (void *)m;
  prob->input_dim = z_input_dim;
  prob->state_eq = z_state_eq;
  prob->running_cost = z_running_cost;
  prob->terminal_cost = z_terminal_cost;
  prob->state_eq_deriv = z_state_eq_deriv;
  prob->running_cost_deriv = z_running_cost_deriv;
  prob->terminal_cost_deriv = z_terminal_cost_deriv;
  prob->input_limits = NULL;
}

void oracle_ddpzmp_default_config(oracle_ddp_config_t * c)
{
  oracle_ddp_default_config(c);
  c->with_input_constraint = 0; /* nmpc_ddp default; CCC::DdpZmp does not enable it */
  c->warm_start_guard = 0;      /* the guard belongs to the force-scale planners; the product's DdpZmp ignores it */
}

int oracle_ddpzmp_plan_batch(const oracle_ddpzmp_params_t * prm, const oracle_ddp_config_t * cfg, long n,
                             const double * ref, const double * x0, const double * u_init, double * u_out,
                             double * x_out, int * iters, int * status, double * cost, int nthreads)
{
  const int N = prm->N;
  int worst = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 1 ? nthreads : 1) reduction(min : worst)
#endif
  for(long b = 0; b < n; b++)
  {
    zmodel_t m = {prm, ref + (size_t)b * (N + 1) * 4};
    oracle_ddp_problem_t prob;
    z_problem(&m, &prob);
    oracle_ddp_result_t res;
    int st = oracle_ddp_solve(&prob, cfg, x0 + (size_t)b * 6, u_init ? u_init + (size_t)b * N * 3 : NULL,
                              x_out ? x_out + (size_t)b * (N + 1) * 6 : NULL, u_out + (size_t)b * N * 3, &res);
    if(iters) iters[b] = res.iters;
    if(status) status[b] = st;
    if(cost) cost[b] = res.cost;
    if(st < worst) worst = st;
  }
  (void)nthreads;
  return worst;
}

/* test access to the problem callbacks (finite-difference checks of tests/src/TestDdpZmp.cpp:139-250) */
void oracle_ddpzmp_eval(const oracle_ddpzmp_params_t * prm, const double * ref, int step, const double * x,
                        const double * u, double * x_next, double * Fx, double * Fu, double * run_cost,
                        double * term_cost, double * Lx, double * Lu, double * Vx)
{
  zmodel_t m = {prm, ref};
  double Lxx[36], Luu[9], Lxu[18], Vxx[36], lx[6], lu[3], vx[6];
  if(x_next) z_state_eq(&m, step, x, u, x_next);
  if(Fx && Fu) z_state_eq_deriv(&m, step, x, u, Fx, Fu);
  if(run_cost) *run_cost = z_running_cost(&m, step, x, u);
  if(term_cost) *term_cost = z_terminal_cost_at(&m, step, x);
  if(Lx || Lu)
  {
    z_running_cost_deriv(&m, step, x, u, lx, lu, Lxx, Luu, Lxu);
    if(Lx) memcpy(Lx, lx, sizeof(lx));
    if(Lu) memcpy(Lu, lu, sizeof(lu));
  }
  if(Vx)
  {
    /* terminal derivative evaluated with the RefData of `step` (the test probes arbitrary times) */
    oracle_ddpzmp_params_t p2 = *prm;
    p2.N = step;
    zmodel_t m2 = {&p2, ref};
    z_terminal_cost_deriv(&m2, x, vx, Vxx);
    memcpy(Vx, vx, sizeof(vx));
  }
}
