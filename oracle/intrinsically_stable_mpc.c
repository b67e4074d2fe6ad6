/* intrinsically_stable_mpc.c -- CPU restatement of CCC::IntrinsicallyStableMpc (TEST INFRASTRUCTURE ONLY, see
 * ccc_oracle.h).
 *
 * Follows, step by step:
 *   /root/reference/src/IntrinsicallyStableMpc.cpp:8-45     IntrinsicallyStableMpc1d constructor (P, QP constants,
 *                                                            stability equality row of eq. (14), ZMP rows of eq. (8))
 *   /root/reference/src/IntrinsicallyStableMpc.cpp:63-104   IntrinsicallyStableMpc1d::procOnce
 *   /root/reference/src/IntrinsicallyStableMpc.cpp:106-139  IntrinsicallyStableMpc::planOnce (x then y)
 * The QP solve (:93, external QpSolverCollection) is oracle_qp_solve (qp_gi.c).
 */
#include "ccc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_G 9.80665 /* include/CCC/Constants.h:10 */

struct oracle_ism
{
  int N;
  double dt, omega, lambda;
  double w_zmp, w_zmp_vel;
  double * P;        /* N x N : horizon_dt on and below the diagonal (:18-25) */
  double * obj_mat;  /* N x N */
  double * eq_mat;   /* 1 x N */
  double * ineq_mat; /* 2N x N = [-P; P] */
  double * x_min;
  double * x_max;
};

oracle_ism_t * oracle_ism_create(double com_height, double horizon_duration, double horizon_dt, double w_zmp,
                                 double w_zmp_vel)
{
  oracle_ism_t * o = (oracle_ism_t *)calloc(1, sizeof(*o));
  const int N = (int)ceil(horizon_duration / horizon_dt); /* :14 */
  o->N = N;
  o->dt = horizon_dt;
  o->omega = sqrt(ORACLE_G / com_height);            /* :15 */
  o->lambda = exp(-1 * o->omega * horizon_dt);       /* :15 */
  o->w_zmp = w_zmp;
  o->w_zmp_vel = w_zmp_vel;
  o->P = (double *)calloc((size_t)N * N, sizeof(double));
  for(int i = 0; i < N; i++)
    for(int j = 0; j < i + 1; j++) o->P[(size_t)i * N + j] = horizon_dt;
  /* obj_mat = zmp_vel I + zmp P'P (:29-32) */
  o->obj_mat = (double *)calloc((size_t)N * N, sizeof(double));
  for(int i = 0; i < N; i++)
    for(int j = 0; j < N; j++)
    {
      double s = 0;
      for(int k = 0; k < N; k++) s += o->P[(size_t)k * N + i] * o->P[(size_t)k * N + j];
      o->obj_mat[(size_t)i * N + j] = w_zmp * s + (i == j ? w_zmp_vel : 0.0);
    }
  /* eq_mat (:35-39) */
  o->eq_mat = (double *)calloc(N, sizeof(double));
  o->eq_mat[0] = (1 - o->lambda) / (o->omega * (1 - pow(o->lambda, N)));
  for(int i = 1; i < N; i++) o->eq_mat[i] = o->lambda * o->eq_mat[i - 1];
  /* ineq_mat = [-P; P] (:41), bounds (:42-43) */
  o->ineq_mat = (double *)calloc((size_t)2 * N * N, sizeof(double));
  for(int i = 0; i < N; i++)
    for(int j = 0; j < N; j++)
    {
      o->ineq_mat[(size_t)i * N + j] = -1 * o->P[(size_t)i * N + j];
      o->ineq_mat[(size_t)(N + i) * N + j] = o->P[(size_t)i * N + j];
    }
  o->x_min = (double *)malloc(sizeof(double) * N);
  o->x_max = (double *)malloc(sizeof(double) * N);
  for(int i = 0; i < N; i++)
  {
    o->x_min[i] = -1e10;
    o->x_max[i] = 1e10;
  }
  return o;
}

void oracle_ism_destroy(oracle_ism_t * o)
{
  if(!o) return;
  free(o->P);
  free(o->obj_mat);
  free(o->eq_mat);
  free(o->ineq_mat);
  free(o->x_min);
  free(o->x_max);
  free(o);
}

int oracle_ism_horizon_steps(const oracle_ism_t * o)
{
  return o->N;
}

/* procOnce (:63-104): ref_zmp / zmin / zmax are the N sampled values of one axis */
int oracle_ism_proc_once(const oracle_ism_t * o, const double * ref_zmp, const double * zmin, const double * zmax,
                         double capture_point, double planned_zmp, double control_dt, double * zmp,
                         double * zmp_vel_seq, int * iters)
{
  const int N = o->N;
  double * ineq_vec = (double *)malloc(sizeof(double) * 2 * N);
  double * obj_vec = (double *)malloc(sizeof(double) * N);
  double * diff = (double *)malloc(sizeof(double) * N);
  double * x = (double *)malloc(sizeof(double) * N);
  double eq_vec = capture_point - planned_zmp; /* :72 */
  for(int i = 0; i < N; i++)
  {
    ineq_vec[i] = -1 * zmin[i];     /* :79 */
    ineq_vec[i + N] = zmax[i];      /* :80 */
    diff[i] = planned_zmp - ref_zmp[i];
  }
  /* obj_vec = zmp * P' (planned_zmp 1 - ref) (:87-88) */
  for(int j = 0; j < N; j++)
  {
    double s = 0;
    for(int k = 0; k < N; k++) s += o->P[(size_t)k * N + j] * diff[k];
    obj_vec[j] = o->w_zmp * s;
  }
  for(int i = 0; i < N; i++)
  {
    ineq_vec[i] += planned_zmp;     /* :89 */
    ineq_vec[i + N] -= planned_zmp; /* :90 */
  }
  int rc = oracle_qp_solve(N, 1, 2 * N, o->obj_mat, obj_vec, o->eq_mat, &eq_vec, o->ineq_mat, ineq_vec, o->x_min,
                           o->x_max, x, iters, NULL);
  const double zmp_vel = x[0]; /* :93 */
  if(control_dt < 0) control_dt = o->dt; /* :96-99 */
  double z = planned_zmp + control_dt * zmp_vel;
  z = z < zmin[0] ? zmin[0] : (zmax[0] < z ? zmax[0] : z); /* std::clamp, :100-101 */
  *zmp = z;
  if(zmp_vel_seq) memcpy(zmp_vel_seq, x, sizeof(double) * N);
  free(ineq_vec);
  free(obj_vec);
  free(diff);
  free(x);
  return rc;
}

/* planOnce for a batch (:106-139), callbacks already sampled.  Layouts (shared with include/ccc_amd.h):
 *   init [n][2 axes][2] (capture_point, planned_zmp), ref [n][2 axes][3][N] (ref zmp, zmin, zmax rows),
 *   zmp [n][2], vel [n][2][N] or NULL, status [n] or NULL, iters [n][2] or NULL */
int oracle_ism_plan_batch(const oracle_ism_t * o, long n, const double * init, const double * ref, double control_dt,
                          double * zmp, double * vel, int * status, int * iters, int nthreads)
{
  const int N = o->N;
  int worst = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 1 ? nthreads : 1) reduction(max : worst)
#endif
  for(long k = 0; k < n; k++)
  {
    int st = 0;
    for(int ax = 0; ax < 2; ax++)
    {
      const double * r = ref + ((size_t)k * 2 + ax) * 3 * N;
      int it = 0;
      int rc = oracle_ism_proc_once(o, r, r + N, r + 2 * N, init[(k * 2 + ax) * 2 + 0], init[(k * 2 + ax) * 2 + 1],
                                    control_dt, zmp + k * 2 + ax, vel ? vel + ((size_t)k * 2 + ax) * N : NULL, &it);
      if(iters) iters[k * 2 + ax] = it;
      if(rc > st) st = rc;
    }
    if(status) status[k] = st;
    if(st > worst) worst = st;
  }
  (void)nthreads;
  return worst;
}
