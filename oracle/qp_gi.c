/* qp_gi.c -- dense strictly-convex QP by the Goldfarb-Idnani dual active-set method.
 * TEST INFRASTRUCTURE ONLY (see ccc_oracle.h).
 *
 * Stands in for QpSolverCollection::QpSolver::solve(QpCoeff&), the external call at
 * /root/reference/src/LinearMpcZmp.cpp:69 and /root/reference/src/LinearMpcXY.cpp:181
 * (QpSolverCollection is not vendored in the reference; CI resolves QpSolverType::Any to QLD,
 * which implements this same dual method).  Restated from the published algorithm:
 *   D. Goldfarb, A. Idnani, "A numerically stable dual method for solving strictly convex
 *   quadratic programs", Mathematical Programming 27 (1983).
 *
 * Internally every constraint is a' x >= b:
 *   equality   j : a =  Aeq[j], b =  beq[j]    (multiplier free)
 *   inequality j : a = -Cin[j], b = -din[j]
 *   lower bound k: a =  e_k,    b =  xl[k]
 *   upper bound k: a = -e_k,    b = -xu[k]
 */
#include "ccc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
  int n, me, mi;
  const double *Aeq, *beq, *Cin, *din, *xl, *xu;
} cons_t;

static int cons_count(const cons_t * c)
{
  return c->me + c->mi + 2 * c->n;
}

/* a'x - b for constraint id */
static double cons_slack(const cons_t * c, int id, const double * x)
{
  int n = c->n;
  if(id < c->me)
  {
    const double * a = c->Aeq + (size_t)id * n;
    double s = 0;
    for(int i = 0; i < n; i++) s += a[i] * x[i];
    return s - c->beq[id];
  }
  id -= c->me;
  if(id < c->mi)
  {
    const double * a = c->Cin + (size_t)id * n;
    double s = 0;
    for(int i = 0; i < n; i++) s += a[i] * x[i];
    return c->din[id] - s;
  }
  id -= c->mi;
  if(id < n) return x[id] - c->xl[id];
  id -= n;
  return c->xu[id] - x[id];
}

static double cons_rhs_mag(const cons_t * c, int id)
{
  if(id < c->me) return fabs(c->beq[id]);
  id -= c->me;
  if(id < c->mi) return fabs(c->din[id]);
  id -= c->mi;
  if(id < c->n) return fabs(c->xl[id]);
  return fabs(c->xu[id - c->n]);
}

static int cons_is_absent(const cons_t * c, int id)
{
  id -= c->me + c->mi;
  if(id < 0) return 0;
  if(id < c->n) return isinf(c->xl[id]) ? 1 : 0;
  return isinf(c->xu[id - c->n]) ? 1 : 0;
}

/* normal vector a of constraint id into out[n] */
static void cons_normal(const cons_t * c, int id, double * out)
{
  int n = c->n;
  if(id < c->me)
  {
    memcpy(out, c->Aeq + (size_t)id * n, n * sizeof(double));
    return;
  }
  id -= c->me;
  if(id < c->mi)
  {
    const double * a = c->Cin + (size_t)id * n;
    for(int i = 0; i < n; i++) out[i] = -a[i];
    return;
  }
  id -= c->mi;
  memset(out, 0, n * sizeof(double));
  if(id < n)
    out[id] = 1.0;
  else
    out[id - n] = -1.0;
}

typedef struct
{
  int n, q;
  double * J; /* n x n, J = L^-T Q */
  double * R; /* n x n, leading q x q upper triangular */
  double * d; /* n */
  double * z; /* n */
  double * r; /* n */
} fact_t;

static void compute_d(fact_t * f, const double * np)
{
  int n = f->n;
  for(int j = 0; j < n; j++)
  {
    double s = 0;
    for(int i = 0; i < n; i++) s += f->J[i * n + j] * np[i];
    f->d[j] = s;
  }
}

static void compute_z_r(fact_t * f)
{
  int n = f->n, q = f->q;
  for(int i = 0; i < n; i++)
  {
    double s = 0;
    for(int j = q; j < n; j++) s += f->J[i * n + j] * f->d[j];
    f->z[i] = s;
  }
  for(int i = q - 1; i >= 0; i--)
  {
    double s = f->d[i];
    for(int j = i + 1; j < q; j++) s -= f->R[i * n + j] * f->r[j];
    f->r[i] = s / f->R[i * n + i];
  }
}

/* append the constraint whose J'n is in f->d; returns 0 if (numerically) dependent */
static int fact_add(fact_t * f, double * rnorm)
{
  int n = f->n, q = f->q;
  for(int j = n - 1; j > q; j--)
  {
    double a = f->d[j - 1], b = f->d[j];
    if(b == 0.0) continue;
    double h = hypot(a, b);
    double cs = a / h, sn = b / h;
    f->d[j - 1] = h;
    f->d[j] = 0.0;
    for(int k = 0; k < n; k++)
    {
      double t1 = f->J[k * n + j - 1], t2 = f->J[k * n + j];
      f->J[k * n + j - 1] = cs * t1 + sn * t2;
      f->J[k * n + j] = -sn * t1 + cs * t2;
    }
  }
  if(fabs(f->d[q]) <= DBL_EPSILON * (*rnorm) * 16.0) return 0;
  for(int i = 0; i <= q; i++) f->R[i * n + q] = f->d[i];
  if(fabs(f->d[q]) > *rnorm) *rnorm = fabs(f->d[q]);
  f->q = q + 1;
  return 1;
}

/* remove the constraint at position l of the active list */
static void fact_del(fact_t * f, int l)
{
  int n = f->n, q = f->q;
  for(int j = l; j < q - 1; j++)
    for(int i = 0; i <= j + 1; i++) f->R[i * n + j] = f->R[i * n + j + 1];
  for(int i = 0; i < q; i++) f->R[i * n + q - 1] = 0.0;
  f->q = q - 1;
  for(int j = l; j < f->q; j++)
  {
    double a = f->R[j * n + j], b = f->R[(j + 1) * n + j];
    if(b == 0.0) continue;
    double h = hypot(a, b);
    double cs = a / h, sn = b / h;
    f->R[j * n + j] = h;
    f->R[(j + 1) * n + j] = 0.0;
    for(int k = j + 1; k < f->q; k++)
    {
      double t1 = f->R[j * n + k], t2 = f->R[(j + 1) * n + k];
      f->R[j * n + k] = cs * t1 + sn * t2;
      f->R[(j + 1) * n + k] = -sn * t1 + cs * t2;
    }
    for(int k = 0; k < n; k++)
    {
      double t1 = f->J[k * n + j], t2 = f->J[k * n + j + 1];
      f->J[k * n + j] = cs * t1 + sn * t2;
      f->J[k * n + j + 1] = -sn * t1 + cs * t2;
    }
  }
}

int oracle_qp_solve(int n, int me, int mi, const double * H, const double * g, const double * Aeq,
                    const double * beq, const double * Cin, const double * din, const double * xl,
                    const double * xu, double * x, int * iters, double * lam_in)
{
  cons_t cs = {n, me, mi, Aeq, beq, Cin, din, xl, xu};
  int m = cons_count(&cs);
  size_t nn = (size_t)n * n;
  int rc = 0, it = 0;

  /* one per-thread workspace, grown on demand and zeroed per solve (round 4: 14 calloc / free pairs per QP made the
   * OpenMP batch loops of the CPU baseline scale 6x on 256 threads) */
  const size_t nd = 4 * nn + 4 * (size_t)n + (size_t)(n + 1);
  const size_t bytes = nd * sizeof(double) + (size_t)(n + 1) * sizeof(int) + 2 * (size_t)m;
  static _Thread_local char * ws = NULL;
  static _Thread_local size_t ws_cap = 0;
  if(bytes > ws_cap)
  {
    free(ws);
    ws = (char *)malloc(bytes);
    ws_cap = ws ? bytes : 0;
    if(!ws) return 3;
  }
  memset(ws, 0, bytes);
  double * L = (double *)ws;
  double * Li = L + nn;
  fact_t f;
  f.n = n;
  f.q = 0;
  f.J = Li + nn;
  f.R = f.J + nn;
  f.d = f.R + nn;
  f.z = f.d + n;
  f.r = f.z + n;
  double * np = f.r + n;
  double * u = np + n;
  int * act = (int *)(u + (n + 1));
  char * is_act = (char *)(act + (n + 1));
  char * excl = is_act + m;
  if(iters) *iters = 0;
  if(lam_in)
    for(int i = 0; i < mi; i++) lam_in[i] = 0.0;

  /* Cholesky H = L L' */
  for(int j = 0; j < n; j++)
  {
    double s = H[j * n + j];
    for(int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if(!(s > 0.0))
    {
      rc = 3;
      goto done;
    }
    L[j * n + j] = sqrt(s);
    for(int i = j + 1; i < n; i++)
    {
      double t = H[i * n + j];
      for(int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / L[j * n + j];
    }
  }
  /* Li = L^-1 (lower); J = Li' */
  for(int c = 0; c < n; c++)
    for(int i = c; i < n; i++)
    {
      double s = (i == c) ? 1.0 : 0.0;
      for(int k = c; k < i; k++) s -= L[i * n + k] * Li[k * n + c];
      Li[i * n + c] = s / L[i * n + i];
    }
  for(int i = 0; i < n; i++)
    for(int j = 0; j < n; j++) f.J[i * n + j] = Li[j * n + i];
  /* unconstrained minimiser x = -H^-1 g = -Li' Li g */
  for(int i = 0; i < n; i++)
  {
    double s = 0;
    for(int k = 0; k <= i; k++) s += Li[i * n + k] * g[k];
    np[i] = s;
  }
  for(int i = 0; i < n; i++)
  {
    double s = 0;
    for(int k = i; k < n; k++) s += Li[k * n + i] * np[k];
    x[i] = -s;
  }
  double rnorm = 1.0;

  /* equality constraints: always full steps of either sign */
  for(int e = 0; e < me; e++)
  {
    cons_normal(&cs, e, np);
    compute_d(&f, np);
    compute_z_r(&f);
    double zn = 0;
    for(int i = 0; i < n; i++) zn += f.z[i] * np[i];
    double s = cons_slack(&cs, e, x);
    if(!(fabs(zn) > DBL_EPSILON))
    {
      rc = 1; /* dependent equality rows */
      goto done;
    }
    double t = -s / zn;
    for(int i = 0; i < n; i++) x[i] += t * f.z[i];
    for(int k = 0; k < f.q; k++) u[k] -= t * f.r[k];
    u[f.q] = t;
    act[f.q] = e;
    is_act[e] = 1;
    if(!fact_add(&f, &rnorm))
    {
      rc = 1;
      goto done;
    }
  }

  const int max_iter = 50 * (n + m) + 100;
  for(;;)
  {
    /* most violated inactive constraint */
    int ip = -1;
    double worst = 0.0;
    for(int c = me; c < m; c++)
    {
      if(is_act[c] || excl[c] || cons_is_absent(&cs, c)) continue;
      double s = cons_slack(&cs, c, x);
      double tol = 1e-12 * (1.0 + cons_rhs_mag(&cs, c));
      if(s < -tol && s < worst)
      {
        worst = s;
        ip = c;
      }
    }
    if(ip < 0) break;

    cons_normal(&cs, ip, np);
    u[f.q] = 0.0;
    act[f.q] = ip;

    for(;;)
    {
      if(++it > max_iter)
      {
        rc = 2;
        goto done;
      }
      compute_d(&f, np);
      compute_z_r(&f);
      /* partial (dual) step length */
      double t1 = INFINITY;
      int l = -1;
      for(int k = me; k < f.q; k++)
        if(f.r[k] > 0.0)
        {
          double t = u[k] / f.r[k];
          if(t < t1)
          {
            t1 = t;
            l = k;
          }
        }
      /* full (primal) step length */
      double zz = 0, zn = 0;
      for(int i = 0; i < n; i++)
      {
        zz += f.z[i] * f.z[i];
        zn += f.z[i] * np[i];
      }
      double s = cons_slack(&cs, ip, x);
      double t2 = (zz > DBL_EPSILON * DBL_EPSILON && zn > 0.0) ? -s / zn : INFINITY;
      double t = t1 < t2 ? t1 : t2;
      if(isinf(t))
      {
        rc = 1; /* dual unbounded: primal infeasible */
        goto done;
      }
      if(isinf(t2))
      {
        /* step in dual space only, drop l */
        for(int k = 0; k < f.q; k++) u[k] -= t * f.r[k];
        u[f.q] += t;
        is_act[act[l]] = 0;
        fact_del(&f, l);
        for(int k = l; k < f.q + 1; k++)
        {
          u[k] = u[k + 1];
          act[k] = act[k + 1];
        }
        continue;
      }
      for(int i = 0; i < n; i++) x[i] += t * f.z[i];
      for(int k = 0; k < f.q; k++) u[k] -= t * f.r[k];
      u[f.q] += t;
      if(t == t2)
      {
        /* full step: constraint becomes active */
        if(!fact_add(&f, &rnorm))
        {
          /* numerically dependent on the active set: skip it for this round */
          excl[ip] = 1;
          break;
        }
        is_act[ip] = 1;
        memset(excl, 0, m);
        break;
      }
      /* partial step: drop l, keep working on ip */
      is_act[act[l]] = 0;
      fact_del(&f, l);
      for(int k = l; k < f.q + 1; k++)
      {
        u[k] = u[k + 1];
        act[k] = act[k + 1];
      }
    }
  }

done:
  if(iters) *iters = it;
  if(lam_in && rc == 0)
    for(int k = me; k < f.q; k++)
    {
      int c = act[k] - me;
      if(c >= 0 && c < mi) lam_in[c] = u[k];
    }
  return rc;
}
