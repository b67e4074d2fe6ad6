/* ddp_tile.c -- the DDP solver of oracle/ddp.c + the problems of oracle/ddp_models.c in the "TILE" ARITHMETIC
 * (oracle_ddp_config_t::arith = 1; TEST INFRASTRUCTURE ONLY, see ccc_oracle.h).
 *
 * Same algorithm, same reference lines (nmpc_ddp::DDPSolver::solve at /root/reference/src/DdpCentroidal.cpp:229,233 and
 * src/DdpSingleRigidBody.cpp:299,303; the problem callbacks src/DdpCentroidal.cpp:32-177 and
 * src/DdpSingleRigidBody.cpp:26-243): every quantity below has its counterpart in ddp.c / ddp_models.c.  What changes
 * is the ORDER -- and, since round 4, the FORM -- in which the sums are taken: nmpc_ddp forms them with Eigen, whose
 * order is not pinned either, so no order is more faithful than another (parity unpinned at this
 * boundary).  ddp.c keeps plain left-to-right sums and dense matrices (arith = 0: the independent cross-check); this
 * file freezes what a 64-lane wavefront computes without serialising:
 *   tree16   sums over 16 terms:  ((t0+t1)+(t2+t3)) + ((t4+t5)+(t6+t7)) + the same of t8..t15
 *   treeM    sums over the M = 16 B ridges of a step (B = 1, 2, 4 blocks of 16: one, two, up to four surface contacts):
 *            w_c = t_c | t_c + t_{c+16} | (t_c + t_{c+16}) + (t_{c+32} + t_{c+48}), then tree16(w)
 *   chains   products over the state dimension and over the six force / moment rows: one fma chain in increasing index
 *   the backward step in STRUCTURED form (round 4): Fu has six non-zero rows G, so Quu = w_force I + G' V6 G is the
 *            identity plus a matrix of rank 6 and Qux = G' W; the box-QP, the gains and the value update run on 6 x 6 and
 *            6 x S objects and per-ridge 6-vectors -- no M x M matrix is ever formed (see backward_pass_struct)
 * Inputs beyond a step's dimension are exact zeros and take part in the sums (x + 0 = x).  Ridge strides M = 16, 32, 64,
 * reg_type 1 and 2.  All elementwise formulas (cross products, Euler-angle kinematics, the 3x3 inertia solves, cost
 * terms) are those of ddp_models.c, unfused; fma() appears exactly where written.
 * The two arithmetics agree to rounding (tests/test_ddp_tile_emu.py) and the HIP kernel csrc/ddp_tile.h reproduces
 * this file bit for bit.
 */
#include "ccc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define G_ 9.80665
#define MMAX 64 /* largest ridge stride */

static double tree16(const double * t)
{
  double a[8], b[4];
  for(int j = 0; j < 8; j++) a[j] = t[2 * j] + t[2 * j + 1];
  for(int j = 0; j < 4; j++) b[j] = a[2 * j] + a[2 * j + 1];
  const double c0 = b[0] + b[1], c1 = b[2] + b[3];
  return c0 + c1;
}

/* sum over the M = 16 B ridges: the blocks added entry by entry, then the tree */
static double treeM(const double * t, int M_)
{
  double w[16];
  for(int c = 0; c < 16; c++)
    w[c] = M_ == 16 ? t[c] : (M_ == 32 ? t[c] + t[c + 16] : (t[c] + t[c + 16]) + (t[c + 32] + t[c + 48]));
  return tree16(w);
}

static void cross3(const double * a, const double * b, double * c)
{
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* Eigen::LLT<Matrix3d>::solve, src/DdpSingleRigidBody.cpp:88,122-123: the factor as ddp_models.c forms it; the six
 * divisions of the two substitutions are MULTIPLICATIONS by the reciprocals of the factor's diagonal (round 5: the
 * single-rigid-body derivatives solve with this factor nine times per backward step, on the device 57 IEEE divisions of
 * ~28 instructions each -- a fifth of a median solve's time; the inertia matrix is constant over a contact phase, so
 * the kernel forms the three reciprocals once per phase change).  SPEC: r_kk = 1 / l_kk; y0 = b0 r00; y1 = (b1 - l10 y0) r11; y2 = (b2 - l20 y0 - l21
 * y1) r22; x2 = y2 r22; x1 = (y1 - l21 x2) r11; x0 = (y0 - l10 x1 - l20 x2) r00. */
static void llt3_solve(const double * I, const double * b, double * x)
{
  const double l00 = sqrt(I[0]);
  const double l10 = I[3] / l00, l20 = I[6] / l00;
  const double l11 = sqrt(I[4] - l10 * l10);
  const double l21 = (I[7] - l20 * l10) / l11;
  const double l22 = sqrt(I[8] - l20 * l20 - l21 * l21);
  const double r00 = 1.0 / l00, r11 = 1.0 / l11, r22 = 1.0 / l22;
  const double y0 = b[0] * r00;
  const double y1 = (b[1] - l10 * y0) * r11;
  const double y2 = (b[2] - l20 * y0 - l21 * y1) * r22;
  x[2] = y2 * r22;
  x[1] = (y1 - l21 * x[2]) * r11;
  x[0] = (y0 - l10 * x[1] - l20 * x[2]) * r00;
}

typedef struct
{
  const oracle_ddp_model_t * m;
  const oracle_ddp_config_t * c;
  int S, N;
  double *x, *u;          /* current trajectory */
  double *xc[4], *uc[4];  /* line-search candidates */
  double *k, *K;          /* gains N x 16, N x 16 x S */
  double cost, dV[2], lambda, dlambda;
} tile_t;

static int dim_of(const oracle_ddp_model_t * m, int step)
{
  int p = m->step_phase[step];
  p = p < 0 ? 0 : (p >= m->P ? m->P - 1 : p);
  int d = m->phase_dim[p];
  return d < 0 ? 0 : (d > m->M ? m->M : d);
}
static int phase_of(const oracle_ddp_model_t * m, int step)
{
  int p = m->step_phase[step];
  return p < 0 ? 0 : (p >= m->P ? m->P - 1 : p);
}

static void ref_of(const oracle_ddp_model_t * m, int step, double * buf)
{
  const int S = m->model == 0 ? 9 : 12;
  for(int a = 0; a < S; a++) buf[a] = 0;
  for(int a = 0; a < 3; a++) buf[a] = m->ref_pos[(size_t)step * 3 + a];
  if(m->model == 1)
    for(int a = 0; a < 3; a++) buf[3 + a] = m->ref_ori[(size_t)step * 3 + a];
}

/* what stateEq and its derivatives share: per ridge (zero beyond the dimension) vertex, ridge, arm x ridge, and the
 * ridge sums as trees */
typedef struct
{
  double V[MMAX][3], R[MMAX][3], cr[MMAX][3];
  double force[3], moment[3], accel[3];
  const double * inertia; /* MotionParam::inertia_mat of the STEP (src/DdpSingleRigidBody.cpp:56-57,120-123): the instance's
                           * one matrix, or its phase's with oracle_ddp_model_t::inertia_per_phase */
} terms_t;

static void terms_of(const oracle_ddp_model_t * m, int step, const double * x, const double * u, terms_t * T)
{
  const int dim = dim_of(m, step), ph = phase_of(m, step), M_ = m->M;
  const double * V = m->phase_vertex + (size_t)ph * M_ * 3;
  const double * R = m->phase_ridge + (size_t)ph * M_ * 3;
  T->inertia = m->inertia ? m->inertia + (m->inertia_per_phase ? (size_t)ph * 9 : 0) : NULL;
  for(int r = 0; r < M_; r++)
  {
    for(int k = 0; k < 3; k++)
    {
      T->V[r][k] = r < dim ? V[r * 3 + k] : 0.0;
      T->R[r][k] = r < dim ? R[r * 3 + k] : 0.0;
    }
    const double d[3] = {T->V[r][0] - x[0], T->V[r][1] - x[1], T->V[r][2] - x[2]};
    cross3(d, T->R[r], T->cr[r]);
  }
  const double inv_mass = 1.0 / m->mass;
  for(int k = 0; k < 3; k++)
  {
    double t[MMAX];
    for(int r = 0; r < M_; r++) t[r] = u[r] * T->R[r][k];
    T->force[k] = treeM(t, M_);
    for(int r = 0; r < M_; r++) t[r] = u[r] * T->cr[r][k];
    T->moment[k] = treeM(t, M_);
    for(int r = 0; r < M_; r++) t[r] = (u[r] * T->R[r][k]) * inv_mass; /* (round 5: x (1 / mass), not / mass) */
    T->accel[k] = treeM(t, M_);
  }
}

/* src/DdpCentroidal.cpp:32-64 / src/DdpSingleRigidBody.cpp:52-91 */
static void state_eq(const oracle_ddp_model_t * m, const terms_t * T, const double * x, double * xn)
{
  if(m->model == 0)
  {
    double xd[9];
    const double inv_mass = 1.0 / m->mass;
    for(int a = 0; a < 3; a++) xd[a] = x[3 + a] * inv_mass;
    xd[3] = T->force[0];
    xd[4] = T->force[1];
    xd[5] = -1 * m->mass * G_ + T->force[2];
    for(int a = 0; a < 3; a++) xd[6 + a] = T->moment[a];
    for(int a = 0; a < 9; a++) xn[a] = x[a] + m->dt * xd[a];
  }
  else
  {
    const double * I = T->inertia;
    const double * w = x + 9;
    double xd[12], sa, ca, sb, cb;
    for(int a = 0; a < 3; a++) xd[a] = x[6 + a];
    oracle_det_sincos(x[3], &sa, &ca);
    oracle_det_sincos(x[4], &sb, &cb);
    /* matAngularVelToEulerDot(ori) * angular_vel, src/DdpSingleRigidBody.cpp:26-38,72; the divisions by cos(beta) as
     * multiplications by ONE reciprocal (round 5) */
    const double rcb = 1.0 / cb;
    xd[3] = ((ca * sb) * rcb) * w[0] + ((sb * sa) * rcb) * w[1] + 1.0 * w[2];
    xd[4] = (-1 * sa) * w[0] + ca * w[1] + 0.0 * w[2];
    xd[5] = (ca * rcb) * w[0] + (sa * rcb) * w[1] + 0.0 * w[2];
    xd[6] = T->accel[0];
    xd[7] = T->accel[1];
    xd[8] = -1 * G_ + T->accel[2];
    double Iw[3], cw[3], wd[3];
    for(int a = 0; a < 3; a++) Iw[a] = I[a * 3] * w[0] + I[a * 3 + 1] * w[1] + I[a * 3 + 2] * w[2];
    cross3(w, Iw, cw);
    for(int a = 0; a < 3; a++) wd[a] = -1 * cw[a] + T->moment[a];
    llt3_solve(I, wd, xd + 9);
    for(int a = 0; a < 12; a++) xn[a] = x[a] + m->dt * xd[a];
  }
}

/* src/DdpCentroidal.cpp:66-83 */
static double running_cost(const oracle_ddp_model_t * m, int step, const double * x, const double * u)
{
  const int S = m->model == 0 ? 9 : 12, M_ = m->M;
  double ref[12], t[MMAX];
  ref_of(m, step, ref);
  for(int a = 0; a < 16; a++)
  {
    const double e = a < S ? x[a] - ref[a] : 0.0;
    t[a] = a < S ? 0.5 * m->w_run[a] * e * e : 0.0;
  }
  const double cx = tree16(t);
  for(int r = 0; r < M_; r++) t[r] = u[r] * u[r];
  const double un = treeM(t, M_);
  return cx + 0.5 * m->w_force * un;
}
static double terminal_cost(const oracle_ddp_model_t * m, const double * x)
{
  const int S = m->model == 0 ? 9 : 12;
  double ref[12], t[MMAX];
  ref_of(m, m->N, ref);
  for(int a = 0; a < 16; a++)
  {
    const double e = a < S ? x[a] - ref[a] : 0.0;
    t[a] = a < S ? 0.5 * m->w_term[a] * e * e : 0.0;
  }
  return tree16(t);
}

/* Fx (S x S, dense) and the six non-zero rows of Fu (rows FU0 .. FU0+5, 16 columns):
 * src/DdpCentroidal.cpp:85-121 / src/DdpSingleRigidBody.cpp:114-185 with totalForce as a tree */
static void state_eq_deriv(const oracle_ddp_model_t * m, const terms_t * T, const double * x, double * Fx,
                           double (*Fu)[MMAX])
{
  const int S = m->model == 0 ? 9 : 12, M_ = m->M;
  const double dt = m->dt;
  memset(Fx, 0, sizeof(double) * S * S);
  const double * tf = T->force;
  if(m->model == 0)
  {
    for(int r = 0; r < M_; r++)
      for(int k = 0; k < 3; k++)
      {
        Fu[k][r] = T->R[r][k] * dt;
        Fu[3 + k][r] = T->cr[r][k] * dt;
      }
    for(int a = 0; a < 3; a++) Fx[a * S + 3 + a] = (1 / m->mass) * dt;
    Fx[6 * S + 1] = (-tf[2]) * dt;
    Fx[6 * S + 2] = tf[1] * dt;
    Fx[7 * S + 0] = tf[2] * dt;
    Fx[7 * S + 2] = (-tf[0]) * dt;
    Fx[8 * S + 0] = (-tf[1]) * dt;
    Fx[8 * S + 1] = tf[0] * dt;
  }
  else
  {
    const double * I = T->inertia;
    const double inv_mass = 1.0 / m->mass;
    for(int r = 0; r < M_; r++)
    {
      double sol[3];
      llt3_solve(I, T->cr[r], sol);
      for(int k = 0; k < 3; k++)
      {
        Fu[k][r] = (T->R[r][k] * inv_mass) * dt;
        Fu[3 + k][r] = sol[k] * dt;
      }
    }
    const double w1 = x[9], w2 = x[10], w3 = x[11];
    const double I11 = I[0], I12 = I[1], I13 = I[2], I22 = I[4], I23 = I[5], I33 = I[8];
    const double D[9] = {I12 * w3 - I13 * w2,
                         -I13 * w1 + I22 * w3 - 2 * I23 * w2 - I33 * w3,
                         I12 * w1 + I22 * w2 + 2 * I23 * w3 - I33 * w2,
                         -I11 * w3 + 2 * I13 * w1 + I23 * w2 + I33 * w3,
                         -I12 * w3 + I23 * w1,
                         -I11 * w1 - I12 * w2 - 2 * I13 * w3 + I33 * w1,
                         I11 * w2 - 2 * I12 * w1 - I22 * w2 - I23 * w3,
                         I11 * w1 + 2 * I12 * w2 + I13 * w3 - I22 * w1,
                         I13 * w2 - I23 * w1};
    const double CM[9] = {0.0, -tf[2], tf[1], tf[2], 0.0, -tf[0], -tf[1], tf[0], 0.0};
    for(int b = 0; b < 3; b++)
    {
      const double colD[3] = {D[b], D[3 + b], D[6 + b]}, colC[3] = {CM[b], CM[3 + b], CM[6 + b]};
      double sD[3], sC[3];
      llt3_solve(I, colD, sD);
      llt3_solve(I, colC, sC);
      for(int a = 0; a < 3; a++)
      {
        Fx[(9 + a) * S + 9 + b] = sD[a] * dt;
        Fx[(9 + a) * S + b] = sC[a] * dt;
      }
    }
    double sa, ca, sb, cb;
    oracle_det_sincos(x[3], &sa, &ca);
    oracle_det_sincos(x[4], &sb, &cb);
    /* (round 5: rcb = 1 / cos(beta) once, rcb2 = rcb rcb; every "/ cb" and "/ cb2" of src/DdpSingleRigidBody.cpp:140-185 a
     *  multiplication, the products otherwise left to right as written there) */
    const double sb2 = sb * sb, rcb = 1.0 / cb, rcb2 = rcb * rcb;
    for(int a = 0; a < 3; a++) Fx[a * S + 6 + a] = 1.0 * dt;
    const double K[9] = {(ca * sb) * rcb, (sb * sa) * rcb, 1.0, -1 * sa, ca, 0.0, ca * rcb, sa * rcb, 0.0};
    for(int a = 0; a < 3; a++)
      for(int b = 0; b < 3; b++) Fx[(3 + a) * S + 9 + b] = K[a * 3 + b] * dt;
    Fx[3 * S + 3] = (-w1 * sa * sb * rcb + w2 * sb * ca * rcb) * dt;
    Fx[4 * S + 3] = (-w1 * ca - w2 * sa) * dt;
    Fx[5 * S + 3] = (-w1 * sa * rcb + w2 * ca * rcb) * dt;
    Fx[3 * S + 4] = (w1 * sb2 * ca * rcb2 + w1 * ca + w2 * sa * sb2 * rcb2 + w2 * sa) * dt;
    Fx[5 * S + 4] = (w1 * sb * ca * rcb2 + w2 * sa * sb * rcb2) * dt;
  }
  for(int a = 0; a < S; a++) Fx[a * S + a] = Fx[a * S + a] + 1.0;
}

typedef unsigned long long mask_t;

/* ------------------------------------------------------------------------------------------- DDP */
static void increase_lambda(tile_t * d)
{
  d->dlambda = fmax(d->dlambda * d->c->lambda_factor, d->c->lambda_factor);
  d->lambda = fmax(d->lambda * d->dlambda, d->c->lambda_min);
}
static void decrease_lambda(tile_t * d)
{
  d->dlambda = fmin(d->dlambda / d->c->lambda_factor, 1.0 / d->c->lambda_factor);
  d->lambda = d->lambda * d->dlambda * (d->lambda > d->c->lambda_min ? 1.0 : 0.0);
}

/* ------------------------------------------------------------------------------------------- STRUCTURED backward pass
 * (round 4; replaces the dense backward pass of round 3).  Same algorithm, same quantities as backward_pass() above, formed WITHOUT
 * any M x M object.  With G = the six non-zero rows of Fu (6 x M), rows6 = FU0 .. FU0+5, V6 = Vxx[rows6, rows6],
 * W = (Vxx Fx)[rows6, :] (6 x S), lq / lv = lambda on Quu (reg_type 1) / on Vxx (reg_type 2), alpha = w_force + lq,
 * V6r = V6 + lv I, Wr = W + lv Fx[rows6, :]:
 *     Quu   = w_force I + G' V6 G           Quu_F   = alpha I + G' V6r G          (identity plus rank 6)
 *     Qux   = G' W                          Qux_reg = G' Wr
 * so for a free set f, with C_f = sum_{r in f} g_r g_r' (6 x 6) and M_f = alpha I + V6r C_f (6 x 6):
 *     Quu_F,ff^-1 b = (b - G_f' M_f^-1 V6r G_f b) / alpha                          (Woodbury; see s_direction)
 *     K_f = -Quu_F,ff^-1 Qux_reg,f = -G_f' Y,   Y = M_f^-1 Wr   (6 x S; exact: I - M_f^-1 V6r C_f = alpha M_f^-1)
 *     K'Z = D'E,  D = C_f Y,  E = w_force Y - 2 W + V6 D                            (Z = Quu K + 2 Qux)
 * M_f^-1 by Gauss-Jordan elimination with partial pivoting on [M_f | I].  Order of every sum: as written below.
 * The box-QP is Tassa's projected Newton, statement by statement as box_qp_tile(), with H y = alpha y + G'(V6r (G y)).  A failed factorisation = an inverse with an entry that is not finite (zero pivot, NaN, overflow); an indefinite Quu_F (Vxx is positive
 * semi-definite in exact arithmetic, so this takes NaN / overflow) shows up as "no descent direction" as before. */
typedef struct
{
  int M_, m;
  double alpha, inv_alpha, ratio; /* ratio = w_force / alpha */
  double G[6][MMAX];
  double V6r[6][6], Vx6[6];
  const double * u;               /* the step's nominal inputs (Qu = w_force u + G' Vx6) */
  double Cf[6][6], Minv[6][6];
} sqp_t;

/* out = A v for a 6 x 6 A.  SPEC: out_j = A[j][0] v_0; fma(A[j][l], v_l, .), l = 1 .. 5 */
static void apply6(const double (*A)[6], const double * v, double * out)
{
  for(int j = 0; j < 6; j++)
  {
    double s = A[j][0] * v[0];
    for(int l = 1; l < 6; l++) s = fma(A[j][l], v[l], s);
    out[j] = s;
  }
}

/* out_j = treeM(G[j][r] y_r) */
static void six_sums(const sqp_t * q, const double * y, double * out)
{
  double t[MMAX];
  for(int j = 0; j < 6; j++)
  {
    for(int r = 0; r < q->M_; r++) t[r] = q->G[j][r] * y[r];
    out[j] = treeM(t, q->M_);
  }
}

/* hy = Quu_F y = alpha y + G' (V6r (G y)) */
static void s_matvec(const sqp_t * q, const double * y, double * hy)
{
  double gy[6], vy[6];
  six_sums(q, y, gy);
  apply6(q->V6r, gy, vy);
  for(int r = 0; r < q->M_; r++)
  {
    double s = q->alpha * y[r];
    for(int j = 0; j < 6; j++) s = fma(q->G[j][r], vy[j], s);
    hy[r] = s;
  }
}

static double s_value(const sqp_t * q, const double * lin, const double * y, double * hy)
{
  double t[MMAX];
  s_matvec(q, y, hy);
  for(int r = 0; r < q->M_; r++) t[r] = fma(0.5 * y[r], hy[r], y[r] * lin[r]);
  return treeM(t, q->M_);
}

/* C_f, M_f = alpha I + V6r C_f, Minv = M_f^-1 (Gauss-Jordan, partial pivoting); 0 when the inverse is not finite */
static int s_factor(sqp_t * q, const int * fr)
{
  for(int j = 0; j < 6; j++)
    for(int l = 0; l < 6; l++)
    {
      double s = 0.0;
      for(int r = 0; r < q->M_; r++)
        if(fr[r]) s = fma(q->G[j][r], q->G[l][r], s);
      q->Cf[j][l] = s;
    }
  double A[6][12];
  for(int j = 0; j < 6; j++)
    for(int l = 0; l < 6; l++)
    {
      double s = j == l ? q->alpha : 0.0;
      for(int t = 0; t < 6; t++) s = fma(q->V6r[j][t], q->Cf[t][l], s);
      A[j][l] = s;
      A[j][6 + l] = j == l ? 1.0 : 0.0;
    }
  for(int k = 0; k < 6; k++)
  {
    int p = k;
    double best = fabs(A[k][k]);
    for(int i = k + 1; i < 6; i++)
      if(fabs(A[i][k]) > best)
      {
        best = fabs(A[i][k]);
        p = i;
      }
    if(p != k)
      for(int j = 0; j < 12; j++)
      {
        const double tmp = A[k][j];
        A[k][j] = A[p][j];
        A[p][j] = tmp;
      }
    const double rp = 1.0 / A[k][k];
    double mult[6];
    for(int i = 0; i < 6; i++) mult[i] = A[i][k] * rp;
    for(int j = 0; j < 12; j++)
    {
      const double akj = A[k][j];
      for(int i = 0; i < 6; i++)
        if(i != k) A[i][j] = fma(-mult[i], akj, A[i][j]);
      A[k][j] = akj * rp;
    }
  }
  /* a zero pivot (1 / 0) or a pivot that is not finite leaves inf / NaN behind: failure = an entry of the inverse that is
   * not finite */
  int ok = 1;
  for(int j = 0; j < 6; j++)
    for(int t = 0; t < 6; t++)
    {
      q->Minv[j][t] = A[j][6 + t];
      if(!(fabs(A[j][6 + t]) <= 1.7976931348623157e308)) ok = 0;
    }
  return ok;
}

/* Round 5: the free set changes by ONE ridge r.  C_f' = C_f + sigma g g', M_f' = M_f + sigma p g' with g = g_r, p = V6r g
 * (sigma = +1: r becomes free, -1: r becomes clamped), so by Sherman-Morrison
 *     Minv' = Minv - (sigma / den) (Minv p)(g' Minv),   den = 1 + sigma g' Minv p
 * -- a dozen 6-term sums and one division instead of the 16 B-term sums of C_f and a 6 x 6 Gauss-Jordan inverse.  Between two
 * consecutive projected-Newton iterations the clamped set changes by one to four ridges on nine refactorisations in ten
 * (measured on BASELINE's configs 3 and 5), and the instances that set a batch's time refactor five times per backward step.
 * In exact arithmetic g' Minv p = g' (alpha V6r^-1 + C_f)^-1 g >= 0, so den >= 1 when a ridge is freed; when one is
 * clamped den = 1 / (1 + tau) > 0 can be tiny (the ridge was the only one covering its direction: tau ~ g' V6r g / alpha),
 * and the update would cancel: below S_UPDATE_DEN_MIN -- or when anything is not finite -- the caller factorises afresh.
 * SPEC: p = apply6(V6r, g); a = apply6(Minv, p); b_l = g_0 Minv[0][l], fma(g_t, Minv[t][l], .), t = 1 .. 5;
 * gam = g_0 a_0, fma(g_t, a_t, .); den = fma(sigma, gam, 1); scale = sigma / den;
 * Minv[j][l] = fma(-(a_j scale), b_l, Minv[j][l]); Cf[j][l] = fma(sigma g_j, g_l, Cf[j][l]).  Returns 0 (nothing usable) when
 * den is below the threshold or not finite, or an updated entry of Minv is not finite. */
#define S_UPDATE_KMAX 4      /* changed ridges one refactorisation absorbs by rank-one updates (in increasing ridge index) */
#define S_UPDATE_UMAX 12     /* rank-one updates in a row before the next fresh factorisation (bounds the drift) */
#define S_UPDATE_DEN_MIN 0.01
static int s_update(sqp_t * q, int r, double sigma)
{
  double g[6], p[6], a[6], b[6];
  for(int j = 0; j < 6; j++) g[j] = q->G[j][r];
  apply6(q->V6r, g, p);
  apply6(q->Minv, p, a);
  for(int l = 0; l < 6; l++)
  {
    double s = g[0] * q->Minv[0][l];
    for(int t = 1; t < 6; t++) s = fma(g[t], q->Minv[t][l], s);
    b[l] = s;
  }
  double gam = g[0] * a[0];
  for(int t = 1; t < 6; t++) gam = fma(g[t], a[t], gam);
  const double den = fma(sigma, gam, 1.0);
  if(!(fabs(den) >= S_UPDATE_DEN_MIN) || !(fabs(den) <= 1.7976931348623157e308)) return 0;
  const double scale = sigma / den;
  int ok = 1;
  for(int j = 0; j < 6; j++)
  {
    const double as = a[j] * scale, sg = sigma * g[j];
    for(int l = 0; l < 6; l++)
    {
      q->Minv[j][l] = fma(-as, b[l], q->Minv[j][l]);
      q->Cf[j][l] = fma(sg, g[l], q->Cf[j][l]);
      if(!(fabs(q->Minv[j][l]) <= 1.7976931348623157e308)) ok = 0;
    }
  }
  return ok;
}

/* sol = Quu_F,ff^-1 (q + Quu_F xcl)_f on the free rows, WITHOUT the cancellation of the plain Woodbury formula:
 * with q = w_force u + G' Vx6 the right-hand side is w_force u_f + G_f' beta, beta = Vx6 + V6r (G xcl), and
 *   Quu_F,ff^-1 G_f' beta = G_f' M_f^-1 beta              (push-through identity: exact, nothing subtracted)
 *   Quu_F,ff^-1 u_f       = (u_f - G_f' M_f^-1 V6r G_f u_f) / alpha
 * so  sol_r = ratio u_r + g_r' gamma,  gamma = M_f^-1 (beta - ratio V6r (G_f u_f)),  ratio = w_force / alpha:
 * the error is at rounding level relative to |u|, whatever mu / alpha (round 4; with lambda on Vxx -- reg_type 2 -- or a
 * large Vxx the subtracting form loses mu / alpha digits). */
static void s_direction(const sqp_t * q, const int * fr, const double * xcl, double * sol)
{
  double gxc[6], gfu[6], uf[MMAX], beta[6], tv[6], delta[6], gamma[6];
  six_sums(q, xcl, gxc);
  for(int r = 0; r < q->M_; r++) uf[r] = fr[r] ? q->u[r] : 0.0;
  six_sums(q, uf, gfu);
  for(int j = 0; j < 6; j++)
  {
    double b = q->Vx6[j];
    for(int l = 0; l < 6; l++) b = fma(q->V6r[j][l], gxc[l], b);
    beta[j] = b;
  }
  apply6(q->V6r, gfu, tv);
  for(int j = 0; j < 6; j++) delta[j] = fma(-q->ratio, tv[j], beta[j]);
  apply6(q->Minv, delta, gamma);
  for(int r = 0; r < q->M_; r++)
  {
    double s = q->ratio * q->u[r];
    for(int j = 0; j < 6; j++) s = fma(q->G[j][r], gamma[j], s);
    sol[r] = fr[r] ? s : 0.0;
  }
}

/* box_qp_tile() on the structured Hessian; fr_out: the free set the factor in q belongs to (empty: nothing is free) */
static int box_qp_struct(sqp_t * q, const double * lin, const double * lo, const double * hi, double * x, int * fr_out)
{
  const int M_ = q->M_, m = q->m, max_iter = 500;
  const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
  int cl[MMAX] = {0}, oldc[MMAX], fr[MMAX] = {0};
  int frf[MMAX] = {0}, nupd = 0; /* the free set the factor (Cf, Minv) belongs to; rank-one updates since it was formed afresh */
  double hy[MMAX];
  for(int c = 0; c < M_; c++) x[c] = c < m ? fmin(fmax(x[c], lo[c]), hi[c]) : 0.0;
  double value = s_value(q, lin, x, hy), oldvalue = 0.0;
  int result = 0, iter;
  for(iter = 1; iter <= max_iter; iter++)
  {
    if(result != 0) break;
    if(iter > 1 && (oldvalue - value) < min_rel_improve * fabs(oldvalue))
    {
      result = 4;
      break;
    }
    oldvalue = value;
    double grad[MMAX];
    for(int c = 0; c < M_; c++) grad[c] = lin[c] + hy[c]; /* hy = H x of the last value taken at x */
    int changed = (iter == 1), nclamped = 0;
    for(int c = 0; c < M_; c++)
    {
      oldc[c] = cl[c];
      cl[c] = (c < m && ((x[c] == lo[c] && grad[c] > 0) || (x[c] == hi[c] && grad[c] < 0))) ? 1 : 0;
      if(c < m && cl[c] != oldc[c]) changed = 1;
      nclamped += cl[c];
    }
    if(nclamped == m)
    {
      result = 6;
      break;
    }
    if(changed)
    {
      for(int c = 0; c < M_; c++) fr[c] = c < m && !cl[c];
      /* round 5: a set that differs from the factorised one by a few ridges is reached by rank-one updates (s_update),
       * ridge by ridge in increasing index; afresh in the first iteration (new V6r, G), beyond S_UPDATE_KMAX changes, after
       * S_UPDATE_UMAX updates in a row, and whenever an update declines */
      int nch = 0;
      for(int c = 0; c < m; c++) nch += (fr[c] != frf[c]);
      int fresh = (iter == 1) || nch > S_UPDATE_KMAX || nupd + nch > S_UPDATE_UMAX;
      for(int c = 0; c < m && !fresh; c++)
        if(fr[c] != frf[c])
        {
          if(s_update(q, c, fr[c] ? 1.0 : -1.0))
            nupd++;
          else
            fresh = 1;
        }
      if(fresh)
      {
        if(!s_factor(q, fr))
        {
          result = -1;
          break;
        }
        nupd = 0;
      }
      for(int c = 0; c < M_; c++) frf[c] = fr[c];
    }
    double t[MMAX];
    for(int c = 0; c < M_; c++) t[c] = fr[c] ? grad[c] * grad[c] : 0.0;
    const double gn = sqrt(treeM(t, M_));
    if(gn < min_grad)
    {
      result = 5;
      break;
    }
    double xcl[MMAX], rhs[MMAX], srch[MMAX];
    for(int c = 0; c < M_; c++) xcl[c] = cl[c] ? x[c] : 0.0;
    s_direction(q, fr, xcl, rhs);
    for(int c = 0; c < M_; c++) srch[c] = fr[c] ? -rhs[c] - x[c] : 0.0;
    for(int c = 0; c < M_; c++) t[c] = srch[c] * grad[c];
    const double sdotg = treeM(t, M_);
    if(sdotg >= 0) break; /* no descent direction: result stays 0 */
    double step = 1.0, vc = 0, xc[MMAX];
    for(;;)
    {
      for(int c = 0; c < M_; c++) xc[c] = c < m ? fmin(fmax(x[c] + step * srch[c], lo[c]), hi[c]) : 0.0;
      vc = s_value(q, lin, xc, hy);
      if(!((vc - oldvalue) / (step * sdotg) < armijo)) break;
      step *= step_dec;
      if(step < min_step)
      {
        result = 2;
        break;
      }
    }
    for(int c = 0; c < M_; c++) x[c] = xc[c];
    value = vc;
  }
  if(iter > max_iter && result == 0) result = 1;
  for(int c = 0; c < M_; c++) fr_out[c] = (result == 6) ? 0 : (c < m && !cl[c]);
  return result;
}

/* Round 5: the products with Fx = df/dx use its STRUCTURE.  Fx = I + dt A, and A of either model has a handful of entries
 * per column (src/DdpCentroidal.cpp:85-121: d pos / d momentum and d moment / d pos; src/DdpSingleRigidBody.cpp:114-185: d pos /
 * d vel, the Euler-rate block, the angular-acceleration blocks).  Column c of Fx can be non-zero only in the leading rows of
 * FX_NZ[c]; every list is filled to one length per model (3 / 6) with rows whose entry in that column is an exact zero, so
 * that each sum has the same number of terms on every lane of the kernel.  T1 = Vxx Fx, Qxx = Lxx + Fx' T1 and Qx = Lx + Fx' Vx
 * add the terms of the listed rows in the listed order: what the dense chains add, minus exact zeros -- the same values
 * wherever Vxx is finite, a third (9 states) or half (12 states) of the operations.
 * SPEC (L = 3 / 6, b_k = FX_NZ[c][k]):
 *   T1[a][c] = Vxx[a][b_0] Fx[b_0][c]; then fma(Vxx[a][b_k], Fx[b_k][c], .), k = 1 .. L-1
 *   Qxx[c][r] = (c == r) w_run[c]; then fma(Fx[b_k][c], T1[b_k][r], .), k = 0 .. L-1
 *   Qx[c] = Lx[c]; then fma(Fx[b_k][c], Vx[b_k], .), k = 0 .. L-1 */
#define FX_NZ_MAX 6
static const signed char FX_NZ9[9][FX_NZ_MAX] = {{0, 7, 8}, {1, 6, 8}, {2, 6, 7}, {0, 3, 6}, {1, 4, 6}, {2, 5, 6}, {6, 0, 1}, {7, 0, 1}, {8, 0, 1}};
static const signed char FX_NZ12[12][FX_NZ_MAX] = {{0, 9, 10, 11, 3, 4},  {1, 9, 10, 11, 3, 4},  {2, 9, 10, 11, 3, 4},  {3, 4, 5, 0, 1, 2},
                                                   {3, 4, 5, 0, 1, 2},    {5, 0, 1, 2, 3, 4},    {0, 6, 3, 4, 5, 9},    {1, 7, 3, 4, 5, 9},
                                                   {2, 8, 3, 4, 5, 9},    {3, 4, 5, 9, 10, 11},  {3, 4, 5, 9, 10, 11},  {3, 4, 5, 9, 10, 11}};

static int backward_pass_struct(tile_t * d, double * gsum)
{
  const oracle_ddp_model_t * m = d->m;
  const int S = d->S, N = d->N, FU0 = S == 9 ? 3 : 6, M_ = m->M;
  const double lq = d->c->reg_type == 2 ? 0.0 : d->lambda, lv = d->c->reg_type == 2 ? d->lambda : 0.0;
  double Vxx[144], Vx[12], ref[12];
  memset(Vxx, 0, sizeof(Vxx));
  ref_of(m, N, ref);
  for(int a = 0; a < S; a++)
  {
    Vxx[a * S + a] = m->w_term[a];
    Vx[a] = m->w_term[a] * (d->x[(size_t)N * S + a] - ref[a]);
  }
  d->dV[0] = d->dV[1] = 0;
  *gsum = 0;
  double kprev[MMAX] = {0};
  int mprev = -1;
  sqp_t * q = (sqp_t *)malloc(sizeof(sqp_t));
  for(int i = N - 1; i >= 0; i--)
  {
    const int dim = dim_of(m, i);
    const double * x = d->x + (size_t)i * S;
    double u[MMAX];
    for(int r = 0; r < M_; r++) u[r] = r < dim ? d->u[(size_t)i * M_ + r] : 0.0;
    terms_t T;
    terms_of(m, i, x, u, &T);
    double Fx[144], Fu[6][MMAX];
    state_eq_deriv(m, &T, x, Fx, Fu);
    /* Qx, Qu as in backward_pass() */
    double Qx[12], Qu[MMAX];
    ref_of(m, i, ref);
    const signed char (*NZ)[FX_NZ_MAX] = S == 9 ? FX_NZ9 : FX_NZ12;
    const int NL = S == 9 ? 3 : 6;
    for(int a = 0; a < S; a++)
    {
      double s = m->w_run[a] * (x[a] - ref[a]);
      for(int k = 0; k < NL; k++) s = fma(Fx[NZ[a][k] * S + a], Vx[NZ[a][k]], s);
      Qx[a] = s;
    }
    for(int r = 0; r < M_; r++)
    {
      double s = m->w_force * u[r];
      for(int b = 0; b < 6; b++) s = fma(Fu[b][r], Vx[FU0 + b], s);
      Qu[r] = r < dim ? s : 0.0;
    }
    /* T1 = Vxx Fx ; Qxx = Lxx + Fx' T1, over the rows Fx's structure leaves (FX_NZ above) */
    double T1[144], Qxx[144];
    for(int a = 0; a < S; a++)
      for(int b2 = 0; b2 < S; b2++)
      {
        double s = Vxx[a * S + NZ[b2][0]] * Fx[NZ[b2][0] * S + b2];
        for(int k = 1; k < NL; k++) s = fma(Vxx[a * S + NZ[b2][k]], Fx[NZ[b2][k] * S + b2], s);
        T1[a * S + b2] = s;
      }
    for(int a = 0; a < S; a++)
      for(int b2 = 0; b2 < S; b2++)
      {
        double s = a == b2 ? m->w_run[a] : 0.0;
        for(int k = 0; k < NL; k++) s = fma(Fx[NZ[a][k] * S + a], T1[NZ[a][k] * S + b2], s);
        Qxx[a * S + b2] = s;
      }
    /* the six-dimensional pieces */
    double V6[6][6], W[6][12], Wr[6][12];
    q->M_ = M_;
    q->m = dim;
    q->alpha = m->w_force + lq;
    q->inv_alpha = 1.0 / q->alpha;
    q->ratio = m->w_force * q->inv_alpha;
    q->u = u;
    for(int j = 0; j < 6; j++)
    {
      q->Vx6[j] = Vx[FU0 + j];
      for(int l = 0; l < 6; l++)
      {
        V6[j][l] = Vxx[(FU0 + j) * S + FU0 + l];
        q->V6r[j][l] = j == l ? V6[j][l] + lv : V6[j][l];
      }
      for(int a = 0; a < S; a++)
      {
        W[j][a] = T1[(FU0 + j) * S + a];
        Wr[j][a] = fma(lv, Fx[(FU0 + j) * S + a], W[j][a]);
      }
      for(int r = 0; r < M_; r++) q->G[j][r] = r < dim ? Fu[j][r] : 0.0;
    }
    /* box-QP and gains */
    double k[MMAX] = {0}, K[MMAX][12], Y[6][12];
    int fr[MMAX] = {0};
    memset(K, 0, sizeof(K));
    memset(Y, 0, sizeof(Y));
    memset(q->Cf, 0, sizeof(q->Cf));
    if(dim > 0)
    {
      double lo[MMAX], hi[MMAX];
      for(int r = 0; r < M_; r++)
      {
        lo[r] = r < dim ? m->force_lo - u[r] : 0.0;
        hi[r] = r < dim ? m->force_hi - u[r] : 0.0;
        k[r] = (mprev == dim) ? kprev[r] : 0.0;
      }
      const int rc = box_qp_struct(q, Qu, lo, hi, k, fr);
      if(rc < 1)
      {
        free(q);
        return 0;
      }
      int nfree = 0;
      for(int r = 0; r < M_; r++) nfree += fr[r];
      if(nfree == 0)
        memset(q->Cf, 0, sizeof(q->Cf)); /* everything clamped: no feedback (the factor in q is of an older set) */
      else
        for(int j = 0; j < 6; j++)
          for(int a = 0; a < S; a++)
          {
            double s = q->Minv[j][0] * Wr[0][a];
            for(int t = 1; t < 6; t++) s = fma(q->Minv[j][t], Wr[t][a], s);
            Y[j][a] = s;
          }
      for(int r = 0; r < M_; r++)
        for(int a = 0; a < S; a++)
        {
          double s = q->G[0][r] * Y[0][a];
          for(int j = 1; j < 6; j++) s = fma(q->G[j][r], Y[j][a], s);
          K[r][a] = fr[r] ? -s : 0.0;
        }
    }
    for(int r = 0; r < M_; r++)
    {
      d->k[(size_t)i * M_ + r] = k[r];
      for(int a = 0; a < S; a++) d->K[((size_t)i * M_ + r) * S + a] = K[r][a];
    }
    {
      double mx = 0.0;
      for(int r = 0; r < dim; r++) mx = fmax(mx, fabs(k[r]) / (fabs(u[r]) + 1.0));
      *gsum += mx;
    }
    /* dV, Vx, Vxx */
    double gk[6], vk[6], gfv[6], t4[MMAX], t[MMAX];
    six_sums(q, k, gk);
    apply6(V6, gk, vk);
    for(int r = 0; r < M_; r++)
    {
      double s = m->w_force * k[r];
      for(int j = 0; j < 6; j++) s = fma(q->G[j][r], vk[j], s);
      t4[r] = s; /* (Quu k)_r = w_force k_r + g_r' V6 (G k) */
    }
    for(int r = 0; r < M_; r++) t[r] = k[r] * Qu[r];
    d->dV[0] += treeM(t, M_);
    for(int r = 0; r < M_; r++) t[r] = k[r] * t4[r];
    d->dV[1] += 0.5 * treeM(t, M_);
    for(int r = 0; r < M_; r++) t[r] = fr[r] ? t4[r] + Qu[r] : 0.0;
    six_sums(q, t, gfv);
    double vxn[12];
    for(int a = 0; a < S; a++)
    {
      double s = Qx[a];
      for(int j = 0; j < 6; j++) s = fma(W[j][a], gk[j], s);
      for(int j = 0; j < 6; j++) s = fma(-Y[j][a], gfv[j], s);
      vxn[a] = s;
    }
    double D[6][12], E[6][12];
    for(int j = 0; j < 6; j++)
      for(int a = 0; a < S; a++)
      {
        double s = q->Cf[j][0] * Y[0][a];
        for(int t2 = 1; t2 < 6; t2++) s = fma(q->Cf[j][t2], Y[t2][a], s);
        D[j][a] = s;
      }
    for(int j = 0; j < 6; j++)
      for(int a = 0; a < S; a++)
      {
        double s = m->w_force * Y[j][a] - 2.0 * W[j][a];
        for(int t2 = 0; t2 < 6; t2++) s = fma(V6[j][t2], D[t2][a], s);
        E[j][a] = s;
      }
    for(int a = 0; a < S; a++)
      for(int b = a; b < S; b++)
      {
        double tab = D[0][a] * E[0][b], tba = D[0][b] * E[0][a];
        for(int j = 1; j < 6; j++)
        {
          tab = fma(D[j][a], E[j][b], tab);
          tba = fma(D[j][b], E[j][a], tba);
        }
        const double v = 0.5 * ((Qxx[a * S + b] + Qxx[b * S + a]) + (tab + tba));
        Vxx[a * S + b] = v;
        Vxx[b * S + a] = v;
      }
    for(int a = 0; a < S; a++) Vx[a] = vxn[a];
    memcpy(kprev, k, sizeof(k));
    mprev = dim;
  }
  free(q);
  return 1;
}

/* forward pass for one step size into candidate slot q; returns its cost */
static double forward_pass(tile_t * d, double alpha, int q)
{
  const oracle_ddp_model_t * m = d->m;
  const int S = d->S, N = d->N, M_ = m->M;
  double * xc = d->xc[q], * uc = d->uc[q];
  memcpy(xc, d->x, sizeof(double) * S);
  double cost = 0;
  for(int i = 0; i < N; i++)
  {
    const int dim = dim_of(m, i);
    const double * xi = d->x + (size_t)i * S, * ui = d->u + (size_t)i * M_;
    double * xn = xc + (size_t)i * S, * un = uc + (size_t)i * M_;
    const double * ki = d->k + (size_t)i * M_, * Ki = d->K + (size_t)i * M_ * S;
    for(int r = 0; r < M_; r++)
    {
      double s = ui[r] + alpha * ki[r];
      for(int a = 0; a < S; a++) s = fma(Ki[r * S + a], xn[a] - xi[a], s);
      un[r] = r < dim ? fmin(fmax(s, m->force_lo), m->force_hi) : 0.0;
    }
    cost = cost + running_cost(m, i, xn, un);
    terms_t T;
    terms_of(m, i, xn, un, &T);
    state_eq(m, &T, xn, xc + (size_t)(i + 1) * S);
  }
  return cost + terminal_cost(m, xc + (size_t)N * S);
}

int oracle_ddp_solve_tile(const oracle_ddp_model_t * m, const oracle_ddp_config_t * c, const double * x0,
                          const double * u_init, double * x_out, double * u_out, oracle_ddp_result_t * res)
{
  tile_t d;
  memset(&d, 0, sizeof(d));
  const int S = m->model == 0 ? 9 : 12, N = m->N, M_ = m->M;
  if((M_ != 16 && M_ != 32 && M_ != 64) || !c->with_input_constraint) return -100; /* not in this arithmetic */
  d.m = m;
  d.c = c;
  d.S = S;
  d.N = N;
  const size_t nx = (size_t)(N + 1) * S, nu = (size_t)N * M_;
  double * buf = (double *)calloc(5 * (nx + nu) + nu + nu * S, sizeof(double));
  d.x = buf;
  d.u = buf + nx;
  for(int q = 0; q < 4; q++)
  {
    d.xc[q] = buf + (size_t)(q + 1) * (nx + nu);
    d.uc[q] = d.xc[q] + nx;
  }
  d.k = buf + 5 * (nx + nu);
  d.K = d.k + nu;
  d.lambda = c->initial_lambda;
  d.dlambda = c->initial_dlambda;
  memcpy(d.x, x0, sizeof(double) * S);
  d.cost = 0;
  for(int i = 0; i < N; i++)
  {
    const int dim = dim_of(m, i);
    double * ui = d.u + (size_t)i * M_;
    for(int r = 0; r < M_; r++) ui[r] = (r < dim && u_init) ? u_init[(size_t)i * M_ + r] : 0.0;
    d.cost = d.cost + running_cost(m, i, d.x + (size_t)i * S, ui);
    terms_t T;
    terms_of(m, i, d.x + (size_t)i * S, ui, &T);
    state_eq(m, &T, d.x + (size_t)i * S, d.x + (size_t)(i + 1) * S);
  }
  d.cost = d.cost + terminal_cost(m, d.x + (size_t)N * S);
  int warm_replaced = 0;
  if(c->warm_start_guard && u_init)
  {
    /* warm-start guard (ccc_oracle.h): the rollout of zero inputs, same arithmetic, into the first candidate slot */
    double * xz = d.xc[0], * uz = d.uc[0];
    memcpy(xz, x0, sizeof(double) * S);
    double cold = 0;
    for(int i = 0; i < N; i++)
    {
      double * ui = uz + (size_t)i * M_;
      for(int r = 0; r < M_; r++) ui[r] = 0.0;
      cold = cold + running_cost(m, i, xz + (size_t)i * S, ui);
      terms_t T;
      terms_of(m, i, xz + (size_t)i * S, ui, &T);
      state_eq(m, &T, xz + (size_t)i * S, xz + (size_t)(i + 1) * S);
    }
    cold = cold + terminal_cost(m, xz + (size_t)N * S);
    if(!(d.cost <= cold))
    {
      memcpy(d.x, xz, sizeof(double) * nx);
      memcpy(d.u, uz, sizeof(double) * nu);
      d.cost = cold;
      warm_replaced = 1;
    }
  }
  const double initial_cost = d.cost;

  int iter = 0, status = 0, n_accept = 0;
  for(iter = 1; iter <= c->max_iter; iter++)
  {
    int bp_ok = 0;
    double gsum = 0;
    for(;;)
    {
      if(backward_pass_struct(&d, &gsum))
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
    const double g = gsum / N;
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
      const double alpha = c->alpha_list[a];
      const double costc = forward_pass(&d, alpha, 0);
      actual = d.cost - costc;
      const double expected = -alpha * (d.dV[0] + alpha * d.dV[1]);
      const double ratio = expected > 0 ? actual / expected : (actual > 0 ? 1.0 : (actual < 0 ? -1.0 : 0.0));
      if(ratio > c->cost_update_ratio_thre)
      {
        accepted = 1;
        memcpy(d.x, d.xc[0], sizeof(double) * nx);
        memcpy(d.u, d.uc[0], sizeof(double) * nu);
        d.cost = costc;
        break;
      }
    }
    if(accepted)
    {
      decrease_lambda(&d);
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
    res->warm_replaced = warm_replaced;
    res->lambda = d.lambda;
    res->accepted = n_accept;
  }
  free(buf);
  return status;
}
