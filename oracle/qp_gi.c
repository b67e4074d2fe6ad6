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
  double * Jt; /* n x n, J = L^-T Q stored TRANSPOSED: Jt[j * n + i] = J[i][j] -- every loop of the iteration (the column sums
               * J'n, the Givens rotations of two columns of J, z = J2 d2) then runs over contiguous memory (round 5: the
               * row-major J made the CPU baseline of LinearMpcXY, n = 320, run at the speed of its cache misses; same
               * operations in the same order, same bits) */
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
    const double * Jj = f->Jt + (size_t)j * n;
    for(int i = 0; i < n; i++) s += Jj[i] * np[i];
    f->d[j] = s;
  }
}

/* the same for a bound constraint, whose normal is sign e_k: every other term of the sum is an exact zero, so J'n is a
 * scaled row of Jt (QLD, too, keeps the bounds out of its dense constraint rows) */
static void compute_d_bound(fact_t * f, int k, double sign)
{
  int n = f->n;
  for(int j = 0; j < n; j++) f->d[j] = 0.0 + f->Jt[(size_t)j * n + k] * sign;
}

static void compute_z_r(fact_t * f)
{
  int n = f->n, q = f->q;
  for(int i = 0; i < n; i++) f->z[i] = 0;
  for(int j = q; j < n; j++) /* (per entry of z the terms still arrive in increasing j) */
  {
    const double * Jj = f->Jt + (size_t)j * n;
    const double dj = f->d[j];
    for(int i = 0; i < n; i++) f->z[i] += Jj[i] * dj;
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
    double * Ja = f->Jt + (size_t)(j - 1) * n, * Jb = f->Jt + (size_t)j * n;
    for(int k = 0; k < n; k++)
    {
      double t1 = Ja[k], t2 = Jb[k];
      Ja[k] = cs * t1 + sn * t2;
      Jb[k] = -sn * t1 + cs * t2;
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
    double * Ja = f->Jt + (size_t)j * n, * Jb = f->Jt + (size_t)(j + 1) * n;
    for(int k = 0; k < n; k++)
    {
      double t1 = Ja[k], t2 = Jb[k];
      Ja[k] = cs * t1 + sn * t2;
      Jb[k] = -sn * t1 + cs * t2;
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
  f.Jt = Li + nn;
  f.R = f.Jt + nn;
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

  /* Cholesky H = L L', right-looking: column j is finished, then its rank-1 update goes into the trailing rows -- every
   * entry (i, c) still receives its subtractions l_i0 l_c0, l_i1 l_c1, ... one after the other in increasing k, i.e. the
   * operations and order of the dot-product form  t = H[i][c]; for k < c: t -= L[i][k] L[c][k]  this replaces (same bits),
   * but the inner loop now runs over independent entries instead of one dependent chain (round 5: n = 320 for LinearMpcXY).
   * LT = L' is filled on the way: column j of L contiguous, for this update and the substitution below. */
  double * LT = f.R; /* (R is not in use before the first constraint is added; zeroed again below) */
  for(int i = 0; i < n; i++)
    for(int c = 0; c <= i; c++) L[i * n + c] = H[i * n + c];
  for(int j = 0; j < n; j++)
  {
    const double s = L[j * n + j];
    if(!(s > 0.0))
    {
      rc = 3;
      memset(LT, 0, nn * sizeof(double));
      goto done;
    }
    const double ljj = sqrt(s);
    L[j * n + j] = ljj;
    double * col = LT + (size_t)j * n; /* col[i] = L[i][j] */
    col[j] = ljj;
    for(int i = j + 1; i < n; i++)
    {
      L[i * n + j] = L[i * n + j] / ljj;
      col[i] = L[i * n + j];
    }
    for(int i = j + 1; i < n; i++)
    {
      const double lij = col[i];
      double * Li_ = L + (size_t)i * n;
      for(int c = j + 1; c <= i; c++) Li_[c] -= lij * col[c];
    }
  }
  /* Li = L^-1 (lower), held transposed (LiT[c * n + k] = Li[k][c]: the substitution reads it along k); J = Li', so the
   * transposed J of the iteration is Li itself */
  double * LiT = Li;
  for(int c = 0; c < n; c++)
  {
    /* column c of L^-1 by forward substitution, column-oriented: s_i = delta_ic, and as soon as entry k is final every
     * later s_i loses L[i][k] Li[k][c] -- per entry the subtractions of  s -= L[i][k] Li[k][c], k = c .. i - 1  in that order */
    double * sc = LiT + (size_t)c * n;
    for(int i = c; i < n; i++) sc[i] = (i == c) ? 1.0 : 0.0;
    for(int k = c; k < n; k++)
    {
      sc[k] = sc[k] / L[k * n + k];
      const double v = sc[k];
      const double * colk = LT + (size_t)k * n; /* colk[i] = L[i][k] */
      for(int i = k + 1; i < n; i++) sc[i] -= colk[i] * v;
    }
  }
  memset(LT, 0, nn * sizeof(double)); /* (R again) */
  for(int i = 0; i < n; i++)
    for(int j = 0; j < n; j++) f.Jt[i * n + j] = LiT[j * n + i]; /* = Li[i][j] */
  /* unconstrained minimiser x = -H^-1 g = -Li' Li g */
  for(int i = 0; i < n; i++)
  {
    double s = 0;
    for(int k = 0; k <= i; k++) s += f.Jt[i * n + k] * g[k];
    np[i] = s;
  }
  for(int i = 0; i < n; i++)
  {
    double s = 0;
    for(int k = i; k < n; k++) s += LiT[i * n + k] * np[k];
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
      if(ip >= me + mi)
        compute_d_bound(&f, ip - me - mi < n ? ip - me - mi : ip - me - mi - n, ip - me - mi < n ? 1.0 : -1.0);
      else
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
