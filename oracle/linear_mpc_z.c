/* linear_mpc_z.c -- CPU restatement of CCC::LinearMpcZ (TEST INFRASTRUCTURE ONLY, see ccc_oracle.h).
 *
 * Follows, step by step:
 *   /root/reference/src/LinearMpcZ.cpp:10-29       ModelContactPhase / ModelNoncontactPhase
 *   /root/reference/src/LinearMpcZ.cpp:31-46       constructor (force_range_ = (10, 10 m g), both models discretised)
 *   /root/reference/include/CCC/StateSpaceModel.h:164-216          calcDiscMatrix with an offset vector E
 *   /root/reference/include/CCC/VariantSequentialExtension.h:110-208   setup(extend_for_output = true)
 *   /root/reference/src/LinearMpcZ.cpp:48-71       planOnce (zero force without contact at current_time)
 *   /root/reference/src/LinearMpcZ.cpp:73-94       procOnce (QP coefficients, bounds, solve, [0])
 * The QP solve (:93, external QpSolverCollection) is oracle_qp_solve (qp_gi.c).
 */
#include "ccc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_G 9.80665
#define Z_S 2

struct oracle_z
{
  int N;
  double mass, dt, w_pos, w_force;
  double Ad_c[4], Bd_c[2], Ed_c[2]; /* contact phase */
  double Ad_n[4], Ed_n[2];          /* non-contact phase (no input) */
  double C[2];
  double fmin, fmax;
};

oracle_z_t * oracle_z_create(double mass, double horizon_dt, int horizon_steps, double w_pos, double w_force)
{
  oracle_z_t * o = (oracle_z_t *)calloc(1, sizeof(*o));
  o->N = horizon_steps;
  o->mass = mass;
  o->dt = horizon_dt;
  o->w_pos = w_pos;
  o->w_force = w_force;
  /* :10-29 */
  const double A[4] = {0, 1, 0, 0}, B[2] = {0, 1}, E[2] = {0, -1 * mass * ORACLE_G};
  o->C[0] = 1 / mass;
  o->C[1] = 0;
  /* :41-42 */
  oracle_calc_disc_matrix(Z_S, 1, A, B, E, horizon_dt, o->Ad_c, o->Bd_c, o->Ed_c);
  oracle_calc_disc_matrix(Z_S, 0, A, NULL, E, horizon_dt, o->Ad_n, NULL, o->Ed_n);
  /* :37 */
  o->fmin = 10.0;
  o->fmax = 10.0 * mass * ORACLE_G;
  return o;
}

void oracle_z_destroy(oracle_z_t * o)
{
  free(o);
}

/* planOnce (:48-71) + procOnce (:73-94).  contact[N] (0/1), ref_pos[N] sampled at current_time + i dt,
 * x0 = InitialParam = [pos, vel].  Outputs the planned force (0 without contact at step 0) and optionally every
 * QP variable (force_all, one per contact step, compact). */
int oracle_z_plan_once(const oracle_z_t * o, const int * contact, const double * ref_pos, const double * x0,
                       double * force, double * force_all, int * iters)
{
  const int N = o->N, S = Z_S;
  if(iters) *iters = 0;
  if(!contact[0]) /* :54-57 */
  {
    *force = 0.0;
    return 0;
  }
  int n = 0;
  for(int i = 0; i < N; i++) n += contact[i] ? 1 : 0;
  const double cx[2] = {o->mass * x0[0], o->mass * x0[1]}; /* :70 */
  /* VariantSequentialExtension::setup (:110-208) */
  double * A_seq = (double *)calloc((size_t)N * S * S, sizeof(double));
  double * B_seq = (double *)calloc((size_t)N * S * n, sizeof(double));
  double * E_seq = (double *)calloc((size_t)N * S, sizeof(double));
  int accum = 0;
  for(int i = 0; i < N; i++)
  {
    const double * Ad_i = contact[i] ? o->Ad_c : o->Ad_n;
    const double * Ed_i = contact[i] ? o->Ed_c : o->Ed_n;
    const int m = contact[i] ? 1 : 0;
    if(i == 0)
      memcpy(A_seq, Ad_i, sizeof(double) * S * S);
    else
      for(int a = 0; a < S; a++)
        for(int b = 0; b < S; b++)
        {
          double s = 0;
          for(int k = 0; k < S; k++) s += Ad_i[a * S + k] * A_seq[((size_t)(i - 1) * S + k) * S + b];
          A_seq[((size_t)i * S + a) * S + b] = s;
        }
    if(m)
      for(int j = i; j < N; j++)
      {
        const double * Ad_j = contact[j] ? o->Ad_c : o->Ad_n;
        for(int a = 0; a < S; a++)
        {
          if(j == i)
            B_seq[((size_t)j * S + a) * n + accum] = o->Bd_c[a];
          else
          {
            double s = 0;
            for(int k = 0; k < S; k++) s += Ad_j[a * S + k] * B_seq[((size_t)(j - 1) * S + k) * n + accum];
            B_seq[((size_t)j * S + a) * n + accum] = s;
          }
        }
      }
    for(int a = 0; a < S; a++)
    {
      double s = Ed_i[a];
      if(i > 0)
        for(int k = 0; k < S; k++) s += Ad_i[a * S + k] * E_seq[(size_t)(i - 1) * S + k];
      E_seq[(size_t)i * S + a] = s;
    }
    accum += m;
  }
  /* extend for output: C_seq = blockdiag(C) (:187-206) -> one output row per step */
  double * Ao = (double *)calloc((size_t)N * S, sizeof(double));
  double * Bo = (double *)calloc((size_t)N * n, sizeof(double));
  double * Eo = (double *)calloc(N, sizeof(double));
  for(int i = 0; i < N; i++)
  {
    for(int b = 0; b < S; b++)
    {
      double s = 0;
      for(int k = 0; k < S; k++) s += o->C[k] * A_seq[((size_t)i * S + k) * S + b];
      Ao[(size_t)i * S + b] = s;
    }
    for(int c = 0; c < n; c++)
    {
      double s = 0;
      for(int k = 0; k < S; k++) s += o->C[k] * B_seq[((size_t)i * S + k) * n + c];
      Bo[(size_t)i * n + c] = s;
    }
    double s = 0;
    for(int k = 0; k < S; k++) s += o->C[k] * E_seq[(size_t)i * S + k];
    Eo[i] = s;
  }
  /* procOnce (:84-90) */
  double * H = (double *)calloc((size_t)n * n, sizeof(double));
  double * g = (double *)calloc(n, sizeof(double));
  double * res = (double *)calloc(N, sizeof(double));
  double * xl = (double *)malloc(sizeof(double) * n);
  double * xu = (double *)malloc(sizeof(double) * n);
  double * sol = (double *)calloc(n, sizeof(double));
  for(int p = 0; p < n; p++)
    for(int q = 0; q < n; q++)
    {
      double s = 0;
      for(int k = 0; k < N; k++) s += Bo[(size_t)k * n + p] * Bo[(size_t)k * n + q];
      H[(size_t)p * n + q] = o->w_pos * s + (p == q ? o->w_force : 0.0);
    }
  for(int k = 0; k < N; k++) res[k] = ref_pos[k] - (Ao[(size_t)k * S + 0] * cx[0] + Ao[(size_t)k * S + 1] * cx[1]) - Eo[k];
  for(int p = 0; p < n; p++)
  {
    double s = 0;
    for(int k = 0; k < N; k++) s += Bo[(size_t)k * n + p] * res[k];
    g[p] = -1 * o->w_pos * s;
    xl[p] = o->fmin;
    xu[p] = o->fmax;
  }
  int rc = oracle_qp_solve(n, 0, 0, H, g, NULL, NULL, NULL, NULL, xl, xu, sol, iters, NULL);
  *force = sol[0]; /* :93 */
  if(force_all) memcpy(force_all, sol, sizeof(double) * n);
  free(A_seq);
  free(B_seq);
  free(E_seq);
  free(Ao);
  free(Bo);
  free(Eo);
  free(H);
  free(g);
  free(res);
  free(xl);
  free(xu);
  free(sol);
  return rc;
}

/* batch: contact [n][N] i32, ref_pos [n][N], x0 [n][2]; force [n], force_all [n][N] (compact per instance, rest 0)
 * or NULL, status [n] or NULL, iters [n] or NULL */
int oracle_z_plan_batch(const oracle_z_t * o, long n, const int * contact, const double * ref_pos, const double * x0,
                        double * force, double * force_all, int * status, int * iters, int nthreads)
{
  const int N = o->N;
  int worst = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 1 ? nthreads : 1) reduction(max : worst)
#endif
  for(long k = 0; k < n; k++)
  {
    int it = 0;
    if(force_all) memset(force_all + (size_t)k * N, 0, sizeof(double) * N);
    int rc = oracle_z_plan_once(o, contact + (size_t)k * N, ref_pos + (size_t)k * N, x0 + k * 2, force + k,
                                force_all ? force_all + (size_t)k * N : NULL, &it);
    if(iters) iters[k] = it;
    if(status) status[k] = rc;
    if(rc > worst) worst = rc;
  }
  (void)nthreads;
  return worst;
}
