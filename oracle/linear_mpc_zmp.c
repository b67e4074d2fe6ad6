/* linear_mpc_zmp.c -- CPU restatement of CCC::LinearMpcZmp (TEST INFRASTRUCTURE ONLY, see ccc_oracle.h).
 *
 * Follows, step by step:
 *   /root/reference/src/CommonModels.cpp:8-17                        ComZmpModelJerkInput
 *   /root/reference/include/CCC/StateSpaceModel.h:164-216            calcDiscMatrix (matrix exponential)
 *   /root/reference/include/CCC/InvariantSequentialExtension.h:103-181  setup(extend_for_output = true)
 *   /root/reference/src/LinearMpcZmp.cpp:9-28                        LinearMpcZmp1d constructor (QP constants)
 *   /root/reference/src/LinearMpcZmp.cpp:46-81                       LinearMpcZmp1d::procOnce
 *   /root/reference/src/LinearMpcZmp.cpp:83-112                      LinearMpcZmp::planOnce (x then y)
 */
#include "ccc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_G 9.80665 /* include/CCC/Constants.h:10 */

struct oracle_zmp
{
  int N;
  double dt;
  double com_height;
  double C[3];    /* output row */
  double * A_seq; /* N x 3 */
  double * B_seq; /* N x N */
  /* QP constants, src/LinearMpcZmp.cpp:21-27 */
  double * obj_mat;  /* N x N identity */
  double * obj_vec;  /* N zeros */
  double * ineq_mat; /* 2N x N = [-B_seq; B_seq] */
  double * x_min;    /* -1e10 */
  double * x_max;    /* +1e10 */
};

oracle_zmp_t * oracle_zmp_create(double com_height, double horizon_duration, double horizon_dt)
{
  oracle_zmp_t * o = (oracle_zmp_t *)calloc(1, sizeof(*o));
  /* src/LinearMpcZmp.cpp:13 */
  int N = (int)ceil(horizon_duration / horizon_dt);
  o->N = N;
  o->dt = horizon_dt;
  o->com_height = com_height;

  /* src/CommonModels.cpp:8-17 */
  double A[9] = {0}, B[3] = {0};
  A[0 * 3 + 1] = 1;
  A[1 * 3 + 2] = 1;
  B[2] = 1;
  o->C[0] = 1;
  o->C[1] = 0;
  o->C[2] = -1 * com_height / ORACLE_G;

  /* src/LinearMpcZmp.cpp:17 -> StateSpaceModel.h:195-203 (E == 0, fixed-size branch) */
  double Ad[9], Bd[3], Ed[3];
  oracle_calc_disc_matrix(3, 1, A, B, NULL, horizon_dt, Ad, Bd, Ed);

  /* InvariantSequentialExtension.h:115-160 with StateDim = 3, InputDim = 1 */
  int S = 3;
  double * As = (double *)calloc((size_t)N * S * S, sizeof(double));
  double * Bs = (double *)calloc((size_t)N * S * N, sizeof(double));
  for(int i = 0; i < N; i++)
  {
    if(i == 0)
      memcpy(As, Ad, sizeof(Ad));
    else
      for(int r = 0; r < S; r++)
        for(int c = 0; c < S; c++)
        {
          double s = 0;
          for(int k = 0; k < S; k++) s += Ad[r * S + k] * As[((i - 1) * S + k) * S + c];
          As[(i * S + r) * S + c] = s;
        }
    for(int j = 0; j < N - i; j++)
    {
      if(j == 0)
      {
        if(i == 0)
          for(int r = 0; r < S; r++) Bs[(size_t)(r)*N + 0] = Bd[r];
        else
          for(int r = 0; r < S; r++)
          {
            double s = 0;
            for(int k = 0; k < S; k++) s += Ad[r * S + k] * Bs[(size_t)((i - 1) * S + k) * N + 0];
            Bs[(size_t)(i * S + r) * N + 0] = s;
          }
      }
      else
        for(int r = 0; r < S; r++) Bs[(size_t)((i + j) * S + r) * N + j] = Bs[(size_t)(i * S + r) * N + 0];
    }
  }
  /* InvariantSequentialExtension.h:163-180: apply the block-diagonal C_seq */
  o->A_seq = (double *)calloc((size_t)N * 3, sizeof(double));
  o->B_seq = (double *)calloc((size_t)N * N, sizeof(double));
  for(int i = 0; i < N; i++)
  {
    for(int c = 0; c < 3; c++)
    {
      double s = 0;
      for(int k = 0; k < S; k++) s += o->C[k] * As[(i * S + k) * S + c];
      o->A_seq[i * 3 + c] = s;
    }
    for(int j = 0; j < N; j++)
    {
      double s = 0;
      for(int k = 0; k < S; k++) s += o->C[k] * Bs[(size_t)(i * S + k) * N + j];
      o->B_seq[(size_t)i * N + j] = s;
    }
  }
  free(As);
  free(Bs);

  /* src/LinearMpcZmp.cpp:21-27 */
  o->obj_mat = (double *)calloc((size_t)N * N, sizeof(double));
  o->obj_vec = (double *)calloc(N, sizeof(double));
  o->ineq_mat = (double *)calloc((size_t)2 * N * N, sizeof(double));
  o->x_min = (double *)calloc(N, sizeof(double));
  o->x_max = (double *)calloc(N, sizeof(double));
  for(int i = 0; i < N; i++)
  {
    o->obj_mat[(size_t)i * N + i] = 1.0;
    o->x_min[i] = -1e10;
    o->x_max[i] = 1e10;
    for(int j = 0; j < N; j++)
    {
      o->ineq_mat[(size_t)i * N + j] = -1 * o->B_seq[(size_t)i * N + j];
      o->ineq_mat[(size_t)(N + i) * N + j] = o->B_seq[(size_t)i * N + j];
    }
  }
  return o;
}

void oracle_zmp_destroy(oracle_zmp_t * o)
{
  if(!o) return;
  free(o->A_seq);
  free(o->B_seq);
  free(o->obj_mat);
  free(o->obj_vec);
  free(o->ineq_mat);
  free(o->x_min);
  free(o->x_max);
  free(o);
}

int oracle_zmp_horizon_steps(const oracle_zmp_t * o)
{
  return o->N;
}

void oracle_zmp_get_seq(const oracle_zmp_t * o, double * A_seq, double * B_seq)
{
  if(A_seq) memcpy(A_seq, o->A_seq, (size_t)o->N * 3 * sizeof(double));
  if(B_seq) memcpy(B_seq, o->B_seq, (size_t)o->N * o->N * sizeof(double));
}

static double clampd(double v, double lo, double hi)
{
  /* std::clamp(v, lo, hi), src/LinearMpcZmp.cpp:78 */
  return v < lo ? lo : (hi < v ? hi : v);
}

/* src/LinearMpcZmp.cpp:46-81 */
int oracle_zmp_proc_once(oracle_zmp_t * o, const double * zmin, const double * zmax, const double * x0,
                         double control_dt, double * zmp, double * jerk_seq, int * iters)
{
  int N = o->N;
  /* (per-thread scratch: no allocation per solve) */
  static _Thread_local double * scratch = NULL;
  static _Thread_local int scratch_n = 0;
  if(N > scratch_n)
  {
    free(scratch);
    scratch = (double *)malloc((size_t)3 * N * sizeof(double));
    scratch_n = scratch ? N : 0;
    if(!scratch) return 3;
  }
  double * ineq_vec = scratch;
  double * sol = scratch + 2 * N;
  /* :54-55 */
  for(int i = 0; i < N; i++)
  {
    double s = 0;
    for(int k = 0; k < 3; k++) s += o->A_seq[i * 3 + k] * x0[k];
    ineq_vec[i] = s;
    ineq_vec[N + i] = -1 * s;
  }
  /* :56-66 */
  for(int i = 0; i < N; i++)
  {
    ineq_vec[i] -= zmin[i];
    ineq_vec[N + i] += zmax[i];
  }
  /* :69 */
  int rc = oracle_qp_solve(N, 0, 2 * N, o->obj_mat, o->obj_vec, NULL, NULL, o->ineq_mat, ineq_vec, o->x_min,
                           o->x_max, sol, iters, NULL);
  double com_jerk = sol[0];
  /* :72-78 */
  if(control_dt < 0) control_dt = o->dt;
  double com_acc = x0[2] + control_dt * com_jerk;
  double com_pos = x0[0] + control_dt * x0[1] + 0.5 * pow(control_dt, 2) * x0[2];
  *zmp = clampd(com_pos + o->C[2] * com_acc, zmin[0], zmax[0]);
  if(jerk_seq) memcpy(jerk_seq, sol, (size_t)N * sizeof(double));
  return rc;
}

/* src/LinearMpcZmp.cpp:83-112 for a batch of already-sampled instances */
int oracle_zmp_plan_batch(oracle_zmp_t * o, long n, const double * x0, const double * zlim,
                          double control_dt, double * zmp, double * jerk, int * status, int * iters,
                          int nthreads)
{
  int N = o->N;
  int worst = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 1 ? nthreads : 1) reduction(max : worst)
#endif
  for(long b = 0; b < n; b++)
  {
    int st = 0;
    for(int ax = 0; ax < 2; ax++)
    {
      const double * lim = zlim + ((size_t)b * 2 + ax) * 2 * N;
      int it = 0;
      int rc = oracle_zmp_proc_once(o, lim, lim + N, x0 + ((size_t)b * 2 + ax) * 3, control_dt,
                                    zmp + (size_t)b * 2 + ax, jerk ? jerk + ((size_t)b * 2 + ax) * N : NULL, &it);
      if(iters) iters[(size_t)b * 2 + ax] = it;
      if(rc > st) st = rc;
    }
    if(status) status[b] = st;
    if(st > worst) worst = st;
  }
  (void)nthreads;
  return worst;
}
