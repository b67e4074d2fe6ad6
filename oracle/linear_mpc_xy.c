/* linear_mpc_xy.c -- CPU restatement of CCC::LinearMpcXY (TEST INFRASTRUCTURE ONLY, see ccc_oracle.h).
 *
 * Follows, step by step:
 *   /root/reference/src/LinearMpcXY.cpp:26-38      InitialParam::toState / RefData::toOutput
 *   /root/reference/src/LinearMpcXY.cpp:40-57      WeightParam::inputWeight / outputWeight
 *   /root/reference/src/LinearMpcXY.cpp:59-83      Model::Model (continuous A, B from the flattened contact ridges)
 *   /root/reference/include/CCC/StateSpaceModel.h:170-192   calcDiscMatrix, dynamic-size branch (matrix exponential)
 *   /root/reference/include/CCC/VariantSequentialExtension.h:110-208   setup(extend_for_output = false)
 *   /root/reference/src/LinearMpcXY.cpp:96-114     planOnce (callbacks already sampled)
 *   /root/reference/src/LinearMpcXY.cpp:116-182    procOnce (QP coefficients, equality rows, bounds, solve, head(m0))
 * The QP solve (:181, external QpSolverCollection) is oracle_qp_solve (qp_gi.c).
 */
#include "ccc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_G 9.80665
#define XY_S 6

/* One instance.  Per-step arrays: dim[N] ridges of step i (0 = no contact), vertex/ridge [N][M][3] (contact -> vertex
 * -> ridge order), com_z[N], total_force_z[N], ref_out[N][6] (= RefData::toOutput(mass)), x0[6] (= toState(mass)).
 * Outputs: u0[M] (first dim[0] entries = planned force scales), optional lambda_all[sum dim], iters. */
int oracle_xy_plan_once(const oracle_xy_params_t * prm, const int * dim, const double * vertex, const double * ridge,
                        const double * com_z, const double * total_force_z, const double * ref_out,
                        const double * x0, double * u0, double * lambda_all, int * iters)
{
  const int N = prm->horizon_steps, M = prm->M, S = XY_S;
  const double dt = prm->horizon_dt, mass = prm->mass;
  int total = 0, dim_eq = 0;
  for(int i = 0; i < N; i++)
  {
    total += dim[i];
    if(dim[i] > 0) dim_eq++;
  }
  /* one per-thread workspace for everything below, grown on demand and zeroed per instance (round 5, VERDICT r4 weak #8:
   * 8 + 2 N calloc / free pairs per instance under the OpenMP batch loop; the QP solver has had its own since round 4) */
  const int TS = N * S, tot1 = total > 0 ? total : 1, M1 = M > 0 ? M : 1, eq1 = dim_eq > 0 ? dim_eq : 1;
  const size_t nd = (size_t)N * S * S + (size_t)N * S * M1 + 2 * (size_t)S * M1 + (size_t)TS * S + 2 * (size_t)TS * tot1
                    + (size_t)tot1 * tot1 + (size_t)eq1 * tot1 + (size_t)eq1 + 4 * (size_t)tot1 + (size_t)TS;
  static _Thread_local double * ws = NULL;
  static _Thread_local size_t ws_cap = 0;
  if(nd > ws_cap)
  {
    free(ws);
    ws = (double *)malloc(nd * sizeof(double));
    ws_cap = ws ? nd : 0;
    if(!ws) return 3;
  }
  memset(ws, 0, nd * sizeof(double));
  double * wp = ws;
#define XY_TAKE(count) (wp += (count), wp - (count))
  /* ---- per-step models: Model::Model (:59-83) + calcDiscMatrix (StateSpaceModel.h:170-180, E = 0) */
  double * Ad = XY_TAKE((size_t)N * S * S);
  double * Bd = XY_TAKE((size_t)N * S * M1);
  double * B = XY_TAKE((size_t)S * M1);
  double * Bdm = XY_TAKE((size_t)S * M1);
  for(int i = 0; i < N; i++)
  {
    const int m = dim[i];
    double A[XY_S * XY_S] = {0};
    memset(B, 0, sizeof(double) * S * M1);
    memset(Bdm, 0, sizeof(double) * S * M1);
    A[0 * S + 1] = 1;
    A[2 * S + 3] = 1;
    A[4 * S + 2] = -1 * total_force_z[i] / mass;
    A[5 * S + 0] = total_force_z[i] / mass;
    for(int r = 0; r < m; r++)
    {
      const double * v = vertex + ((size_t)i * M + r) * 3;
      const double * rd = ridge + ((size_t)i * M + r) * 3;
      B[0 * m + r] = 0;
      B[1 * m + r] = rd[0];
      B[2 * m + r] = 0;
      B[3 * m + r] = rd[1];
      B[4 * m + r] = -1 * (v[2] - com_z[i]) * rd[1] + v[1] * rd[2];
      B[5 * m + r] = (v[2] - com_z[i]) * rd[0] + -1 * v[0] * rd[2];
    }
    double Ed[XY_S];
    oracle_calc_disc_matrix(S, m, A, B, NULL, dt, Ad + (size_t)i * S * S, Bdm, Ed);
    for(int a = 0; a < S; a++)
      for(int r = 0; r < m; r++) Bd[((size_t)i * S + a) * M + r] = Bdm[a * m + r];
  }
  /* ---- VariantSequentialExtension::setup (:110-208), extend_for_output = false; E_seq = 0 (Ed = 0) */
  double * A_seq = XY_TAKE((size_t)TS * S);
  double * B_seq = XY_TAKE((size_t)TS * tot1);
  int accum = 0;
  for(int i = 0; i < N; i++)
  {
    const int m = dim[i];
    if(i == 0)
      memcpy(A_seq, Ad, sizeof(double) * S * S);
    else
      for(int a = 0; a < S; a++)
        for(int b = 0; b < S; b++)
        {
          double s = 0;
          for(int k = 0; k < S; k++) s += Ad[((size_t)i * S + a) * S + k] * A_seq[((size_t)(i - 1) * S + k) * S + b];
          A_seq[((size_t)i * S + a) * S + b] = s;
        }
    for(int j = i; j < N; j++)
    {
      if(j == i)
      {
        for(int a = 0; a < S; a++)
          for(int r = 0; r < m; r++) B_seq[((size_t)j * S + a) * total + accum + r] = Bd[((size_t)i * S + a) * M + r];
      }
      else
      {
        for(int a = 0; a < S; a++)
          for(int r = 0; r < m; r++)
          {
            double s = 0;
            for(int k = 0; k < S; k++)
              s += Ad[((size_t)j * S + a) * S + k] * B_seq[((size_t)(j - 1) * S + k) * total + accum + r];
            B_seq[((size_t)j * S + a) * total + accum + r] = s;
          }
      }
    }
    accum += m;
  }
  /* ---- procOnce (:134-178) */
  int rc = 0;
  if(total == 0)
  {
    for(int r = 0; r < M; r++) u0[r] = 0;
    if(iters) *iters = 0;
    goto done;
  }
  {
    const double w1[XY_S] = {prm->w_lmi[0], prm->w_lm[0], prm->w_lmi[1], prm->w_lm[1], prm->w_am[0], prm->w_am[1]};
    double * BW = XY_TAKE((size_t)TS * tot1);
    double * H = XY_TAKE((size_t)tot1 * tot1);
    double * Aeq = XY_TAKE((size_t)eq1 * tot1);
    double * beq = XY_TAKE((size_t)eq1);
    double * g = XY_TAKE((size_t)tot1);
    double * xl = XY_TAKE((size_t)tot1);
    double * xu = XY_TAKE((size_t)tot1);
    double * sol = XY_TAKE((size_t)tot1);
    double * res = XY_TAKE((size_t)TS);
    /* obj_mat = B_seq' diag(w) B_seq + w_force I   (:141-144).  Round 5 (a fair CPU baseline): B_seq is
     * block lower triangular (VariantSequentialExtension.h:110-208: row block j holds the inputs of the steps <= j only), so
     * H[p][q] = sum over the rows k of the blocks j >= max(step of p, step of q) -- the rows above contribute exact zeros.
     * Row by row as rank-1 updates over the columns the row reaches (contiguous, vectorisable), each entry's terms in
     * increasing k and each term as (B[k][p] w) B[k][q], i.e. the very operations and order of the plain triple loop
     * s += B[k][p] * w[k % S] * B[k][q] this replaces: the same bits at a third of the work and without its strided reads. */
    {
      int reach = 0; /* columns row block j reaches: the inputs of the steps 0 .. j */
      for(int j = 0; j < N; j++)
      {
        reach += dim[j];
        for(int a = 0; a < S; a++)
        {
          const size_t k = (size_t)j * S + a;
          const double * Bk = B_seq + k * total;
          double * BWk = BW + k * total;
          for(int p = 0; p < reach; p++) BWk[p] = Bk[p] * w1[a];
          for(int p = 0; p < reach; p++)
          {
            const double bw = BWk[p];
            double * Hp = H + (size_t)p * total;
            for(int q = 0; q < reach; q++) Hp[q] += bw * Bk[q];
          }
        }
      }
    }
    for(int p = 0; p < total; p++) H[(size_t)p * total + p] += prm->w_force;
    /* obj_vec = -B_seq' diag(w) (ref - A_seq x0 - E_seq)   (:145-146) */
    for(int k = 0; k < TS; k++)
    {
      double s = 0;
      for(int b = 0; b < S; b++) s += A_seq[(size_t)k * S + b] * x0[b];
      res[k] = ref_out[k] - s;
    }
    for(int p = 0; p < total; p++)
    {
      double s = 0;
      for(int k = 0; k < TS; k++) s += B_seq[(size_t)k * total + p] * w1[k % S] * res[k];
      g[p] = -1 * s;
    }
    /* equality rows: sum_r ridge_z lambda_r = total_force_z for every step with contact  (:149-176) */
    int eq = 0, in = 0;
    for(int i = 0; i < N; i++)
    {
      if(dim[i] == 0) continue;
      for(int r = 0; r < dim[i]; r++) Aeq[(size_t)eq * total + in + r] = ridge[((size_t)i * M + r) * 3 + 2];
      beq[eq] = total_force_z[i];
      eq++;
      in += dim[i];
    }
    /* bounds: force_range_ = (3, 3 m g)  (:91,:177-178) */
    for(int p = 0; p < total; p++)
    {
      xl[p] = 3.0;
      xu[p] = 3.0 * mass * ORACLE_G;
    }
    rc = oracle_qp_solve(total, dim_eq, 0, H, g, Aeq, beq, NULL, NULL, xl, xu, sol, iters, NULL);
    /* :181  head(model_list[0]->inputDim()) */
    for(int r = 0; r < M; r++) u0[r] = (r < dim[0]) ? sol[r] : 0.0;
    if(lambda_all) memcpy(lambda_all, sol, sizeof(double) * total);
  }
done:
#undef XY_TAKE
  return rc;
}

int oracle_xy_plan_batch(const oracle_xy_params_t * prm, long n, const int * dim, const double * vertex,
                         const double * ridge, const double * com_z, const double * total_force_z,
                         const double * ref_out, const double * x0, double * u0, double * lambda_all, int * iters,
                         int * status, int nthreads)
{
  const int N = prm->horizon_steps, M = prm->M;
  int worst = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 1 ? nthreads : 1) reduction(max : worst)
#endif
  for(long b = 0; b < n; b++)
  {
    int it = 0;
    int rc = oracle_xy_plan_once(prm, dim + (size_t)b * N, vertex + (size_t)b * N * M * 3,
                                 ridge + (size_t)b * N * M * 3, com_z + (size_t)b * N,
                                 total_force_z + (size_t)b * N, ref_out + (size_t)b * N * XY_S, x0 + (size_t)b * XY_S,
                                 u0 + (size_t)b * M, lambda_all ? lambda_all + (size_t)b * N * M : NULL, &it);
    if(iters) iters[b] = it;
    if(status) status[b] = rc;
    if(rc > worst) worst = rc;
  }
  (void)nthreads;
  return worst;
}
