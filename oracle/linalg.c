/* linalg.c -- dense helpers of the CPU oracle (TEST INFRASTRUCTURE ONLY, see ccc_oracle.h). */
#include "ccc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static void matmul(int n, const double * a, const double * b, double * c)
{
  for(int i = 0; i < n; i++)
    for(int j = 0; j < n; j++)
    {
      double s = 0;
      for(int k = 0; k < n; k++) s += a[i * n + k] * b[k * n + j];
      c[i * n + j] = s;
    }
}

/* exp(M): scale M by 2^-s so that its 1-norm is <= 1/4, sum the Taylor series until the terms vanish
 * in double precision, square s times.  For the nilpotent matrices of the hot path (SURVEY.md A.1,
 * A.3) the series terminates after <= 4 terms and the result is exact up to rounding of the
 * individual products. */
void oracle_expm(int n, const double * M, double * out)
{
  double norm1 = 0;
  for(int j = 0; j < n; j++)
  {
    double c = 0;
    for(int i = 0; i < n; i++) c += fabs(M[i * n + j]);
    if(c > norm1) norm1 = c;
  }
  int s = 0;
  double scale = 1.0;
  while(norm1 * scale > 0.25)
  {
    scale *= 0.5;
    s++;
  }
  size_t nn = (size_t)n * n;
  double * a = (double *)malloc(nn * sizeof(double));
  double * term = (double *)malloc(nn * sizeof(double));
  double * tmp = (double *)malloc(nn * sizeof(double));
  for(size_t i = 0; i < nn; i++) a[i] = M[i] * scale;
  memset(out, 0, nn * sizeof(double));
  memset(term, 0, nn * sizeof(double));
  for(int i = 0; i < n; i++)
  {
    out[i * n + i] = 1.0;
    term[i * n + i] = 1.0;
  }
  for(int k = 1; k <= 40; k++)
  {
    matmul(n, term, a, tmp);
    double tn = 0;
    for(size_t i = 0; i < nn; i++)
    {
      term[i] = tmp[i] / (double)k;
      out[i] += term[i];
      if(fabs(term[i]) > tn) tn = fabs(term[i]);
    }
    if(tn == 0.0 || tn < 1e-300) break;
    if(k > 24 && tn < 1e-40) break;
  }
  for(int i = 0; i < s; i++)
  {
    matmul(n, out, out, tmp);
    memcpy(out, tmp, nn * sizeof(double));
  }
  free(a);
  free(term);
  free(tmp);
}

/* include/CCC/StateSpaceModel.h:164-216: exp of dt*[[A, B],[0, 0]] when E == 0 (:173-180, :195-203),
 * exp of dt*[[A, B, E],[0, 0, 0]] otherwise (:182-191, :205-214). */
void oracle_calc_disc_matrix(int ns, int ni, const double * A, const double * B, const double * E, double dt,
                             double * Ad, double * Bd, double * Ed)
{
  double enorm = 0;
  if(E)
    for(int i = 0; i < ns; i++) enorm += E[i] * E[i];
  int aug = ns + ni + (enorm > 0 ? 1 : 0);
  size_t nn = (size_t)aug * aug;
  double * Mx = (double *)calloc(nn, sizeof(double));
  double * Ex = (double *)calloc(nn, sizeof(double));
  for(int i = 0; i < ns; i++)
  {
    for(int j = 0; j < ns; j++) Mx[i * aug + j] = dt * A[i * ns + j];
    for(int j = 0; j < ni; j++) Mx[i * aug + ns + j] = dt * B[i * ni + j];
    if(enorm > 0) Mx[i * aug + ns + ni] = dt * E[i];
  }
  oracle_expm(aug, Mx, Ex);
  for(int i = 0; i < ns; i++)
  {
    for(int j = 0; j < ns; j++) Ad[i * ns + j] = Ex[i * aug + j];
    for(int j = 0; j < ni; j++) Bd[i * ni + j] = Ex[i * aug + ns + j];
    if(Ed) Ed[i] = (enorm > 0) ? Ex[i * aug + ns + ni] : 0.0;
  }
  free(Mx);
  free(Ex);
}
