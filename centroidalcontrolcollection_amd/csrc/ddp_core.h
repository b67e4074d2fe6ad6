// ddp_core.h -- one wavefront = one control-limited DDP / iLQR problem instance (CCC::DdpCentroidal,
// CCC::DdpSingleRigidBody), written as a sequence of PHASES.
//
// Replaces (reference file:line under /root/reference):
//   src/DdpCentroidal.cpp:32-64, :66-83, :85-121, :123-177          problem callbacks (S = 9)
//   src/DdpSingleRigidBody.cpp:26-38, :52-91, :93-112, :114-243     problem callbacks (S = 12)
//   src/DdpCentroidal.cpp:229,233 / src/DdpSingleRigidBody.cpp:299,303   the external nmpc_ddp::DDPSolver::solve
// The solver algorithm is the one frozen in SURVEY.md App. B.2 and spelled out in oracle/ddp.c (Tassa et al.,
// control-limited DDP, box-QP by projected Newton); this file and the oracle implement the same specification
// independently.
//
// Execution model: in a phase every lane runs the same body on its own elements of the small per-instance
// matrices (all staged in LDS, struct Mem); phases are separated by a wavefront-wide barrier.  Per-lane
// registers never carry state across phases -- uniform scalars are re-read from LDS -- which is also what
// lets tests/emu compile this very file for the host (lanes run one after the other) and check the phase
// logic on a machine without a GPU.  The host build is a TEST AID; the product only ever runs the HIP build.
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#  include <hip/hip_runtime.h>
#  define CCC_DDP_FN __device__ __forceinline__
#else
#  define CCC_DDP_FN inline
#endif

// Since round 3 the DEFAULT kernel of the DDP planners is csrc/ddp_tile.h (matrices distributed over the wavefront, the
// tile arithmetic, 16 / 32 / 64 ridges per step).  This file is the ROW-PER-LANE solver in the left-to-right arithmetic
// of oracle/ddp.c, kept for what the tile kernel does not take: reg_type 2, precision 32, and CCC_DDP_LEGACY (the
// reference's SRB closed loop as written).  Two device builds, same algorithm and same register-resident code (a 16-lane
// group holds a row of the 16 x 16 input-space matrices per lane):
//   - the FULL build (csrc/ddp.hip): contact phases of up to 16 ridges, at most kMaxPhases of them and kMaxSteps horizon
//     steps, all staged in LDS (20 KB per wavefront, eight wavefronts per CU);
//   - the LEAN32 build (csrc/ddp_lean32.hip, namespace ddp_lean32): the same tables, compiled for reg_type 1 only (see
//     CCC_DDP_REG1_ONLY below) with single-precision storage of the backward pass.
// The host build (tests/emu) compiles the plain PHASE versions (the "#else" branches of "#if CCC_DDP_FAST"), lanes one
// after the other.
#if defined(CCC_DDP_LEAN) && defined(CCC_DDP_STORE_FLOAT)
#  define CCC_DDP_NS ddp_lean32
#elif defined(CCC_DDP_LEAN)
#  define CCC_DDP_NS ddp_lean
#else
#  define CCC_DDP_NS ddp
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#  define CCC_DDP_FAST 1
#else
#  define CCC_DDP_FAST 0
#endif
// The lean device build is compiled for reg_type 1 only (lambda on Quu: the default): Quu_F = Quu + lambda I and
// Qxu_r = Qxu are then not stored -- 3.7 KB less LDS for the single-rigid-body model (eight wavefronts per CU instead of six).
#if CCC_DDP_FAST && defined(CCC_DDP_LEAN)
#  define CCC_DDP_REG1_ONLY 1
#else
#  define CCC_DDP_REG1_ONLY 0
#endif

// No FMA contraction in this translation unit: every product and sum rounds separately, exactly as in the
// oracle (oracle/ddp.c, gcc -std=c11 => -ffp-contract=off).  With IEEE sqrt and division on both sides the
// centroidal model then reproduces the oracle's iterates bit for bit, so that the discrete decisions of the
// algorithm (line-search acceptance, box-QP clamping, termination tests) cannot flip between the two.  The
// kernel is bound by phase latency, not by VALU issue, so the extra v_mul/v_add pairs are not what limits it.
#if defined(__clang__)
#  pragma clang fp contract(off)
#endif

// Section profiler (development aid, off unless the library is built with -DCCC_DDP_PROF, see scripts/ddp_sections.py):
// lane 0 accumulates shader-clock cycles per section in LDS; solve() then overwrites the first 16 planned inputs with
// the totals, so a profiling build returns timings INSTEAD of a plan.
#if defined(CCC_DDP_PROF) && defined(__HIP_DEVICE_COMPILE__)
#  define CCC_PROF_START() long long prof_t_ = (long long)__builtin_readcyclecounter()
#  define CCC_PROF_RESTART() prof_t_ = (long long)__builtin_readcyclecounter()
#  define CCC_PROF_ADD(k)                                                                   \
    do                                                                                      \
    {                                                                                       \
      const long long prof_n_ = (long long)__builtin_readcyclecounter();                    \
      if((threadIdx.x & 63) == 0) mem.prof[k] += (double)(prof_n_ - prof_t_);               \
      prof_t_ = prof_n_;                                                                    \
    } while(0)
#else
#  define CCC_PROF_START() \
    do                     \
    {                      \
    } while(0)
#  define CCC_PROF_RESTART() \
    do                       \
    {                        \
    } while(0)
#  define CCC_PROF_ADD(k) \
    do                    \
    {                     \
    } while(0)
#endif

namespace ccc_amd
{
// plain data shared by the two device builds (and by the launch code of csrc/ddp.hip)
namespace ddp_common
{
// Batch-constant parameters (by value to the kernel)
struct Params
{
  int model; // 0 = DdpCentroidal (S = 9), 1 = DdpSingleRigidBody (S = 12)
  int N, P;  // horizon steps, contact phases per instance
  double mass, dt;
  double w_run[12], w_term[12], w_force; // WeightParam
  double flo, fhi;                       // force_scale_limits_
  // nmpc_ddp configuration (SURVEY.md App. B.2 + overrides of src/DdpCentroidal.cpp:197-201)
  int max_iter;
  double lambda0, dlambda0, lambda_factor, lambda_min, lambda_max;
  double k_rel_norm_thre, lambda_thre, ratio_thre, cost_thre;
  double alpha[11];
  int reg_type; // 1: Quu_F + lambda I, 2: Vxx + lambda I (oracle/ddp.c)
  int warm_guard; // ccc_ddp_config_t::warm_start_guard
};

// Per-instance problem data and workspace (global memory)
struct Instance
{
  const int * phase_dim;       // [P]
  const double * phase_vertex; // [P][M][3]
  const double * phase_ridge;  // [P][M][3]
  const int * step_phase;      // [N]
  const double * ref_pos;      // [N+1][3]
  const double * ref_ori;      // [N+1][3]  (SRB)
  const double * inertia;      // [9]       (SRB)
  const double * x0;           // [S]
  const double * u_init;       // [N][M] or nullptr
  double *xs, *us;             // current trajectory   [(N+1)][S], [N][M]   (us is the u_out of the C-ABI)
  double *xc, *uc;             // line-search candidate
  double *ks, *Ks;             // gains [N][M], [N][M][S]
  int * out_iters;
  int * out_status;
  double * out_cost;
};
} // namespace ddp_common

namespace CCC_DDP_NS
{
using ddp_common::Instance;
using ddp_common::Params;

// Storage type of the backward pass's matrices in LDS.  double everywhere, except in the build with
// CCC_DDP_STORE_FLOAT (csrc/ddp_lean32.hip: ccc_ddp_config_t::precision = 32, BASELINE configs[4] "fp32 with fp64 tolerance
// check"): single-precision STORAGE, double arithmetic -- every read widens, every write rounds once.  (The host pass of
// that translation unit, which only parses the kernel, keeps double.)
#if defined(CCC_DDP_STORE_FLOAT) && defined(__HIP_DEVICE_COMPILE__)
struct St
{
  float v;
  CCC_DDP_FN St & operator=(double x)
  {
    v = static_cast<float>(x);
    return *this;
  }
  CCC_DDP_FN operator double() const
  {
    return static_cast<double>(v);
  }
  CCC_DDP_FN St & operator*=(double x)
  {
    v = static_cast<float>(static_cast<double>(v) * x);
    return *this;
  }
};
#else
using St = double;
#endif

// profiler sections
enum
{
  PR_DERIV = 0,
  PR_PRODUCTS,
  PR_BOXQP_VALUE,
  PR_BOXQP_GRAD,
  PR_BOXQP_CHOL,
  PR_BOXQP_SOLVE,
  PR_BOXQP_LINESEARCH,
  PR_GAINS,
  PR_VALUE_UPDATE,
  PR_ROLLOUT,
  PR_OTHER,
  PR_BOXQP_CALLS,
  PR_BOXQP_ITERS,
  PR_BOXQP_CHOLS,
  PR_ROLLOUTS,
  PR_TOTAL
};
constexpr int kWave = 64;
constexpr int kMaxSteps = 128; // horizon steps the per-instance LDS tables are sized for
constexpr int kMaxPhases = 4;  // contact phases per instance
constexpr double kGravity = 9.80665; // include/CCC/Constants.h:10

template<class F>
CCC_DDP_FN void phase(F && f)
{
#if defined(__HIP_DEVICE_COMPILE__)
  f(static_cast<int>(threadIdx.x & 63));
  __syncthreads(); // one wavefront per workgroup: an s_barrier that only orders this wave's LDS traffic
#else
  for(int lane = 0; lane < kWave; ++lane) f(lane);
#endif
}

// Row stride of the M x M matrices Quu, QuuF, Lf in LDS: M + 1.  Lane = row accesses (the factor's rows into registers,
// the rows of Quu_F for the box-QP) at a stride of M doubles fall into one bank group -- 43 % of the LDS-active cycles of
// the round-1 kernel were bank conflicts.  The padding costs 3 x 16 x 8 B at M = 16; the device build pays for it with
// the box-QP work vectors only the phase versions use (below), so eight wavefronts per CU still fit.
template<int M> constexpr int row_stride() { return M + 1; }

template<int S, int M>
struct Mem
{
  static constexpr int LQ = row_stride<M>();
  St Vxx[S * S], Vx[S], Fx[S * S], Fu[S * M];
  St Qx[S], Qu[M], Qxx[S * S], Qxu[S * M], Quu[M * LQ];
#if !CCC_DDP_REG1_ONLY
  St Qxur[S * M], QuuF[M * LQ];
#endif
  St T1[S * S], T2[S * M], Lf[M * LQ], K[M * S];
#if defined(CCC_DDP_STORE_FLOAT)
  double stage[8 * M]; // staging of the one-phase centroidal rollout (doubles; the other builds borrow Qxx ..)
#endif
  double k[M], kq[M], lo[M], hi[M];
  alignas(16) double t4[M]; // (also the column buffer of the device factorisation: read back as 128-bit broadcasts)
#if !CCC_DDP_FAST
  double grad[M], srch[M], xcand[M], tmp[M]; // (box-QP state of the phase versions; the device build keeps it in registers)
#endif
  double x[S], xn[S], xd[S], u[M], un[M], ref[S], tf[4], wd[4];
  double rd[M];  // reciprocal diagonal of the box-QP Cholesky factor
  double sc[16]; // uniform scalars
#if defined(CCC_DDP_PROF)
  double prof[16];
#endif
  int clamped[M];
#if !CCC_DDP_FAST
  int oldc[M];
#endif
  int ic[8]; // uniform ints
  // per-instance problem tables, staged once per solve (every model evaluation reads them)
  // in global memory (any horizon length, up to one contact phase per horizon step)
  double pV[kMaxPhases * M * 3], pR[kMaxPhases * M * 3];
  int pdim[kMaxPhases];
  unsigned char sphase[kMaxSteps];
};

// indices into Mem::sc / Mem::ic
enum
{
  SC_VALUE = 0,
  SC_OLDVALUE,
  SC_SDOTG,
  SC_STEP,
  SC_VC,
  SC_COST,
  SC_COSTC,
  SC_DV0,
  SC_DV1,
  SC_LAMBDA,
  SC_DLAMBDA,
  SC_G,
  SC_GNORM
};
enum
{
  IC_RESULT = 0,
  IC_CHANGED,
  IC_ALLCL,
  IC_OK,
  IC_FLAG
};

CCC_DDP_FN void cross3(const double * a, const double * b, double * c)
{
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}


// Deterministic sin/cos (Cody-Waite reduction by pi/2 + the classic fdlibm minimax kernels), accurate to ~1 ulp for
// |x| < 1e3.  The reference calls std::sin / std::cos (src/DdpSingleRigidBody.cpp:30-33,128-131); glibc and the GPU's
// device libm differ from each other in the last ulp, and DDP's discrete decisions can amplify one ulp into a different
// iterate.  Oracle and HIP kernel therefore both evaluate THIS restatement (same operations, no FMA contraction), which
// keeps the single-rigid-body model bit-reproducible across the two; tests check it against libm to 2 ulp.
CCC_DDP_FN void det_sincos(double x, double * s, double * c)
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

// Eigen::LLT<Matrix3d>::solve (src/DdpSingleRigidBody.cpp:88,122-123)
CCC_DDP_FN void llt3_solve(const double * I, const double * b, double * x)
{
  const double l00 = sqrt(I[0]);
  const double l10 = I[3] / l00, l20 = I[6] / l00;
  const double l11 = sqrt(I[4] - l10 * l10);
  const double l21 = (I[7] - l20 * l10) / l11;
  const double l22 = sqrt(I[8] - l20 * l20 - l21 * l21);
  const double y0 = b[0] / l00;
  const double y1 = (b[1] - l10 * y0) / l11;
  const double y2 = (b[2] - l20 * y0 - l21 * y1) / l22;
  x[2] = y2 / l22;
  x[1] = (y1 - l21 * x[2]) / l11;
  x[0] = (y0 - l10 * x[1] - l20 * x[2]) / l00;
}

#if defined(__HIP_DEVICE_COMPILE__)
// value held by lane k of the wavefront, k uniform: two v_readlane_b32 into an SGPR pair (no LDS crossbar round trip,
// which is what __shfl costs even for a constant lane)
CCC_DDP_FN double lane_value(double v, int k)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}
#endif

template<int S, int M>
struct Solver
{
  static constexpr int LQ = Mem<S, M>::LQ;
  static_assert(M == 16, "a row of the 16 x 16 input-space matrices per lane of a 16-lane group");
  static constexpr unsigned long long kRowMask = (1ull << M) - 1ull; // the lanes of the first M-lane group

  // entry (i, k) of the regularised Quu_F the box-QP works on
  CCC_DDP_FN double quuF(int i, int k) const
  {
#if CCC_DDP_REG1_ONLY
    const double q = mem.Quu[i * LQ + k]; // reg_type 1: the same sums, lambda added to the diagonal last
    return i == k ? q + mem.sc[SC_LAMBDA] : q;
#else
    return mem.QuuF[i * LQ + k];
#endif
  }

  const Params & P;
  const Instance & I;
  Mem<S, M> & mem;

  CCC_DDP_FN Solver(const Params & p, const Instance & i, Mem<S, M> & m) : P(p), I(i), mem(m) {}

  CCC_DDP_FN int dim_of(int step) const
  {
    return mem.pdim[mem.sphase[step]];
  }
  CCC_DDP_FN const double * vert_of(int step) const
  {
    return mem.pV + static_cast<int>(mem.sphase[step]) * M * 3;
  }
  CCC_DDP_FN const double * ridge_of(int step) const
  {
    return mem.pR + static_cast<int>(mem.sphase[step]) * M * 3;
  }
  // stage the contact tables of this instance in LDS
  CCC_DDP_FN void stage_problem()
  {
    phase([&](int lane) {
      for(int e = lane; e < P.P * M * 3; e += kWave)
      {
        mem.pV[e] = I.phase_vertex[e];
        mem.pR[e] = I.phase_ridge[e];
      }
      // (phase indices and ridge counts are clamped to the tables: caller data cannot make the kernel read outside them)
      if(lane < P.P)
      {
        const int d = I.phase_dim[lane];
        mem.pdim[lane] = d < 0 ? 0 : (d > M ? M : d);
      }
      for(int e = lane; e < P.N; e += kWave)
      {
        const int p = I.step_phase[e];
        mem.sphase[e] = static_cast<unsigned char>(p < 0 ? 0 : (p >= P.P ? P.P - 1 : p));
      }
    });
  }

  // reference of the weighted state entries at a step (Cen: [pos, 0, 0]; SRB: [pos, ori, 0, 0])
  CCC_DDP_FN double ref_entry(int step, int a) const
  {
    if(a < 3) return I.ref_pos[static_cast<long>(step) * 3 + a];
    if(S == 12 && a < 6) return I.ref_ori[static_cast<long>(step) * 3 + a - 3];
    return 0.0;
  }

  // ---- x_next = stateEq(step, x, u): src/DdpCentroidal.cpp:32-64 / src/DdpSingleRigidBody.cpp:52-91.
  //      x, u, out are LDS arrays of mem (out != x).
  CCC_DDP_FN void state_eq(int step, const double * x, const double * u, double * out)
  {
    const int dim = dim_of(step);
    const double * V = vert_of(step);
    const double * R = ridge_of(step);
    if constexpr(S == 9)
    {
      phase([&](int lane) {
        if(lane < 9)
        {
          const int a = lane;
          double xd;
          if(a < 3)
            xd = x[3 + a] / P.mass;
          else if(a < 6)
          {
            xd = (a == 5) ? -1 * P.mass * kGravity : 0.0;
            for(int r = 0; r < dim; r++) xd += u[r] * R[r * 3 + a - 3];
          }
          else
          {
            xd = 0.0;
            for(int r = 0; r < dim; r++)
            {
              const double d[3] = {V[r * 3] - x[0], V[r * 3 + 1] - x[1], V[r * 3 + 2] - x[2]};
              double c[3];
              cross3(d, R + r * 3, c);
              xd += u[r] * c[a - 6];
            }
          }
          out[a] = x[a] + P.dt * xd;
        }
      });
    }
    else
    {
      phase([&](int lane) {
        if(lane < 9)
        {
          const int a = lane;
          const double * w = x + 9;
          double xd;
          if(a < 3)
            xd = x[6 + a];
          else if(a < 6)
          {
            // matAngularVelToEulerDot(ori) * angular_vel, src/DdpSingleRigidBody.cpp:26-38,72
            double ca, sa, cb, sb;
            det_sincos(x[3], &sa, &ca);
            det_sincos(x[4], &sb, &cb);
            const double K[9] = {(ca * sb) / cb, (sb * sa) / cb, 1.0, -1 * sa, ca, 0.0, ca / cb, sa / cb, 0.0};
            const int b = a - 3;
            xd = K[b * 3] * w[0] + K[b * 3 + 1] * w[1] + K[b * 3 + 2] * w[2];
          }
          else
          {
            xd = (a == 8) ? -1 * kGravity : 0.0;
            for(int r = 0; r < dim; r++) xd += u[r] * R[r * 3 + a - 6] / P.mass;
          }
          mem.xd[a] = xd;
        }
        else if(lane < 12)
        {
          // -w x (I w) + sum_r u_r (p_r - c) x rho_r   (before the inertia solve)
          const int a = lane - 9;
          const double * w = x + 9;
          const double * In = I.inertia;
          double Iw[3], cw[3];
          for(int b = 0; b < 3; b++) Iw[b] = In[b * 3] * w[0] + In[b * 3 + 1] * w[1] + In[b * 3 + 2] * w[2];
          cross3(w, Iw, cw);
          double wd = -1 * cw[a];
          for(int r = 0; r < dim; r++)
          {
            const double d[3] = {V[r * 3] - x[0], V[r * 3 + 1] - x[1], V[r * 3 + 2] - x[2]};
            double c[3];
            cross3(d, R + r * 3, c);
            wd += u[r] * c[a];
          }
          mem.wd[a] = wd;
        }
      });
      phase([&](int lane) {
        if(lane == 0)
        {
          double sol[3];
          llt3_solve(I.inertia, mem.wd, sol);
          for(int a = 0; a < 3; a++) mem.xd[9 + a] = sol[a];
        }
      });
      phase([&](int lane) {
        if(lane < 12) out[lane] = x[lane] + P.dt * mem.xd[lane];
      });
    }
  }

  // running cost of (step, x, u) -- evaluated by ONE lane inside a phase (sequential sums like the oracle)
  CCC_DDP_FN double running_cost(int step, const double * x, const double * u) const
  {
    const int dim = dim_of(step);
    double c = 0, un = 0;
    for(int a = 0; a < S; a++)
    {
      const double e = x[a] - ref_entry(step, a);
      c += 0.5 * P.w_run[a] * e * e;
    }
    for(int r = 0; r < dim; r++) un += u[r] * u[r];
    return c + 0.5 * P.w_force * un;
  }
  CCC_DDP_FN double terminal_cost(const double * x) const
  {
    double c = 0;
    for(int a = 0; a < S; a++)
    {
      const double e = x[a] - ref_entry(P.N, a);
      c += 0.5 * P.w_term[a] * e * e;
    }
    return c;
  }

  // ---- Fx, Fu at (step, mem.x, mem.u): src/DdpCentroidal.cpp:85-121 / src/DdpSingleRigidBody.cpp:114-185
  CCC_DDP_FN void state_eq_deriv(int step)
  {
    const int dim = dim_of(step);
    const double * V = vert_of(step);
    const double * R = ridge_of(step);
    const double * x = mem.x;
    const double * u = mem.u;
    phase([&](int lane) {
      for(int e = lane; e < S * S; e += kWave) mem.Fx[e] = 0.0;
      for(int e = lane; e < S * M; e += kWave) mem.Fu[e] = 0.0;
      if(lane < 3)
      {
        double tf = 0.0;
        for(int r = 0; r < dim; r++) tf += u[r] * R[r * 3 + lane];
        mem.tf[lane] = tf;
      }
    });
    if constexpr(S == 9)
    {
      phase([&](int lane) {
        if(lane < dim)
        {
          const int r = lane;
          const double d[3] = {V[r * 3] - x[0], V[r * 3 + 1] - x[1], V[r * 3 + 2] - x[2]};
          double c[3];
          cross3(d, R + r * 3, c);
          for(int a = 0; a < 3; a++)
          {
            mem.Fu[(3 + a) * M + r] = R[r * 3 + a];
            mem.Fu[(6 + a) * M + r] = c[a];
          }
        }
        if(lane == 32)
        {
          const double * tf = mem.tf;
          for(int a = 0; a < 3; a++) mem.Fx[a * S + 3 + a] = 1 / P.mass;
          mem.Fx[6 * S + 1] = -tf[2];
          mem.Fx[6 * S + 2] = tf[1];
          mem.Fx[7 * S + 0] = tf[2];
          mem.Fx[7 * S + 2] = -tf[0];
          mem.Fx[8 * S + 0] = -tf[1];
          mem.Fx[8 * S + 1] = tf[0];
        }
      });
    }
    else
    {
      phase([&](int lane) {
        const double * In = I.inertia;
        if(lane < dim)
        {
          const int r = lane;
          const double d[3] = {V[r * 3] - x[0], V[r * 3 + 1] - x[1], V[r * 3 + 2] - x[2]};
          double c[3], sol[3];
          cross3(d, R + r * 3, c);
          llt3_solve(In, c, sol);
          for(int a = 0; a < 3; a++)
          {
            mem.Fu[(6 + a) * M + r] = R[r * 3 + a] / P.mass;
            mem.Fu[(9 + a) * M + r] = sol[a];
          }
        }
        if(lane >= 32 && lane < 35)
        {
          // column b of I^-1 d(-w x I w)/dw  -> block (9, 9)
          const int b = lane - 32;
          const double w1 = x[9], w2 = x[10], w3 = x[11];
          const double I11 = In[0], I12 = In[1], I13 = In[2], I22 = In[4], I23 = In[5], I33 = In[8];
          const double D[9] = {I12 * w3 - I13 * w2,
                               -I13 * w1 + I22 * w3 - 2 * I23 * w2 - I33 * w3,
                               I12 * w1 + I22 * w2 + 2 * I23 * w3 - I33 * w2,
                               -I11 * w3 + 2 * I13 * w1 + I23 * w2 + I33 * w3,
                               -I12 * w3 + I23 * w1,
                               -I11 * w1 - I12 * w2 - 2 * I13 * w3 + I33 * w1,
                               I11 * w2 - 2 * I12 * w1 - I22 * w2 - I23 * w3,
                               I11 * w1 + 2 * I12 * w2 + I13 * w3 - I22 * w1,
                               I13 * w2 - I23 * w1};
          const double col[3] = {D[b], D[3 + b], D[6 + b]};
          double sol[3];
          llt3_solve(In, col, sol);
          for(int a = 0; a < 3; a++) mem.Fx[(9 + a) * S + 9 + b] = sol[a];
        }
        if(lane >= 35 && lane < 38)
        {
          // column b of I^-1 crossMat(totalForce) -> block (9, 0)
          const int b = lane - 35;
          const double * tf = mem.tf;
          const double CM[9] = {0, -tf[2], tf[1], tf[2], 0, -tf[0], -tf[1], tf[0], 0};
          const double col[3] = {CM[b], CM[3 + b], CM[6 + b]};
          double sol[3];
          llt3_solve(In, col, sol);
          for(int a = 0; a < 3; a++) mem.Fx[(9 + a) * S + b] = sol[a];
        }
        if(lane == 38)
        {
          const double w1 = x[9], w2 = x[10];
          double ca, sa, cb, sb;
          det_sincos(x[3], &sa, &ca);
          det_sincos(x[4], &sb, &cb);
          const double cb2 = cb * cb, sb2 = sb * sb;
          for(int a = 0; a < 3; a++) mem.Fx[a * S + 6 + a] = 1.0;
          const double K[9] = {(ca * sb) / cb, (sb * sa) / cb, 1.0, -1 * sa, ca, 0.0, ca / cb, sa / cb, 0.0};
          for(int a = 0; a < 3; a++)
            for(int b = 0; b < 3; b++) mem.Fx[(3 + a) * S + 9 + b] = K[a * 3 + b];
          mem.Fx[3 * S + 3] = -w1 * sa * sb / cb + w2 * sb * ca / cb;
          mem.Fx[4 * S + 3] = -w1 * ca - w2 * sa;
          mem.Fx[5 * S + 3] = -w1 * sa / cb + w2 * ca / cb;
          mem.Fx[3 * S + 4] = w1 * sb2 * ca / cb2 + w1 * ca + w2 * sa * sb2 / cb2 + w2 * sa;
          mem.Fx[4 * S + 4] = 0.0;
          mem.Fx[5 * S + 4] = w1 * sb * ca / cb2 + w2 * sa * sb / cb2;
        }
      });
    }
    phase([&](int lane) {
      for(int e = lane; e < S * S; e += kWave)
      {
        double v = mem.Fx[e] * P.dt;
        if(e / S == e % S) v += 1.0;
        mem.Fx[e] = v;
      }
      for(int e = lane; e < S * M; e += kWave) mem.Fu[e] *= P.dt;
    });
  }

  // ---- box-QP: min 1/2 k'Hk + g'k, lo <= k <= hi with H = mem.QuuF (m x m, stride m), g = mem.Qu,
  //      warm start in mem.kq; result in mem.kq / mem.clamped / mem.Lf (Cholesky of H with the clamped rows and
  //      columns replaced by identity = the factor of H_ff embedded).  Tassa's boxQP.m; returns result >= 1 on success.
#if CCC_DDP_FAST
  // Device box-QP: the whole projected-Newton iteration as ONE phase.  Lane i (mod 16) keeps row i of H, its
  // entries of g / lo / hi / x / gradient and the clamped flag in registers; every vector entry another lane needs
  // arrives by v_readlane, every reduction is the oracle's sequential sum run as a readlane chain (same order,
  // same roundings), and all decisions are taken on values that are identical in every lane, so control flow
  // stays uniform.  Only the Cholesky factor goes through LDS (cholesky_phase / load_factor_lane, when the clamped
  // set changes).  Statement by statement the arithmetic is that of the phase version below, which the host
  // emulation (tests/emu) keeps using.
  CCC_DDP_FN int box_qp(int m)
  {
    // (compile-time sizes: the full M, and at M = 32 also 16 -- a single-support step of a walk: size tests fold away)
    return m == M ? box_qp_dev<M>(m) : box_qp_dev<0>(m);
  }

  template<int MM>
  CCC_DDP_FN int box_qp_dev(int m_rt)
  {
    const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
    const int max_iter = 500; // nmpc_ddp BoxQP::Configuration::max_iter
    const int m = MM ? MM : __builtin_amdgcn_readfirstlane(m_rt);
    const int lane = static_cast<int>(threadIdx.x & 63), i = lane & (M - 1);
    const bool in = i < m;
    CCC_PROF_START();
#if defined(CCC_DDP_PROF)
    if(lane == 0) mem.prof[PR_BOXQP_CALLS] += 1.0;
#endif
    double Hr[M];
#  pragma unroll
    for(int k = 0; k < M; ++k) Hr[k] = (in && k < m) ? quuF(i, k) : 0.0;
    const double gi = in ? mem.Qu[i] : 0.0;
    const double lo = in ? mem.lo[i] : 0.0, hi = in ? mem.hi[i] : 0.0;
    double x = in ? fmin(fmax(mem.kq[i], lo), hi) : 0.0;
    bool cl = false;
    // A vector held one entry per lane reaches all lanes either by a readlane pair per entry or -- at M = 32, where the
    // 64 readlanes of every sum were what the iteration spent its time on -- through LDS: every lane publishes its
    // entry (mem.t4, free until the value update), all read the vector back as 128-bit broadcasts.  The sums themselves
    // run in the same order either way.
    constexpr bool kVectorViaLds = false; // (an LDS broadcast instead of v_readlane: was the 32-ridge build's path)
    double * const vbuf = mem.t4;
    auto everywhere = [&](double v, double (&out)[M]) {
      if constexpr(kVectorViaLds)
      {
        if(lane < M) vbuf[i] = v;
        __builtin_amdgcn_wave_barrier();
#  pragma unroll
        for(int k = 0; k < M; ++k) out[k] = vbuf[k];
        __builtin_amdgcn_wave_barrier();
      }
      else
      {
#  pragma unroll
        for(int k = 0; k < M; ++k) out[k] = lane_value(v, k);
      }
    };
    // sum_{k<m} t_k in increasing k, starting from 0 (the oracle's loops)
    auto seq_sum = [&](double t) {
      double tv[M];
      everywhere(t, tv);
      double v = 0;
#  pragma unroll
      for(int k = 0; k < M; ++k)
        if(k < m) v += tv[k];
      return v;
    };
    // s0 + sum_{j<m} H[i][j] y_j in increasing j
    auto row_dot = [&](double s0, double y) {
      double yv[M];
      everywhere(y, yv);
      double s = s0;
#  pragma unroll
      for(int j = 0; j < M; ++j)
        if(j < m) s += Hr[j] * yv[j];
      return s;
    };
    auto value_of = [&](double y) {
      const double s = row_dot(0.0, y);
      return seq_sum(y * gi + 0.5 * y * s);
    };
    double value = value_of(x), oldvalue = 0.0;
    CCC_PROF_ADD(PR_BOXQP_VALUE);
    int result = 0, iter;
    for(iter = 1; iter <= max_iter; iter++)
    {
      if(result != 0) break;
#if defined(CCC_DDP_PROF)
      if(lane == 0) mem.prof[PR_BOXQP_ITERS] += 1.0;
#endif
      if(iter > 1 && (oldvalue - value) < min_rel_improve * fabs(oldvalue))
      {
        result = 4;
        break;
      }
      oldvalue = value;
      const double grad = row_dot(gi, x);
      const bool oldc = cl;
      cl = in && ((x == lo && grad > 0) || (x == hi && grad < 0));
      const unsigned long long inmask = (m >= M) ? kRowMask : ((1ull << m) - 1ull);
      const unsigned long long clmask = __ballot(cl) & inmask;
      const bool changed = (iter == 1) || ((__ballot(cl != oldc) & inmask) != 0ull);
      CCC_PROF_ADD(PR_BOXQP_GRAD);
      if(clmask == inmask)
      {
        result = 6;
        break;
      }
      if(changed)
      {
#if defined(CCC_DDP_PROF)
        if(lane == 0) mem.prof[PR_BOXQP_CHOLS] += 1.0;
#endif
        __syncthreads();
        if(lane < m) mem.clamped[lane] = cl ? 1 : 0;
        __syncthreads();
        cholesky_phase<MM>(m, clmask);
        if(mem.ic[IC_OK] == 0)
        {
          result = -1;
          break;
        }
        CCC_PROF_ADD(PR_BOXQP_CHOL);
      }
      double gn = 0;
      {
        const double g2 = grad * grad; // (sums over a subset of the rows: readlanes for those only, also at M = 32)
#  pragma unroll
        for(int k = 0; k < M; ++k)
          if(k < m && !((clmask >> k) & 1ull)) gn += lane_value(g2, k);
      }
      gn = sqrt(gn);
      if(gn < min_grad) result = 5;
      // grad_clamped = g + H (x .* clamped) on the free rows
      double gc = gi;
#  pragma unroll
      for(int j = 0; j < M; ++j)
        if(j < m && ((clmask >> j) & 1ull)) gc += Hr[j] * lane_value(x, j);
      CCC_PROF_ADD(PR_BOXQP_GRAD);
      if(result != 0) break;
      double sol;
      {
        // the factor is re-read from LDS for every solve: keeping its 32 entries live across the factorisation would
        // not fit the register budget of two waves per SIMD
        double lr[M], lc[M], rdi;
        load_factor_lane<MM>(m, i, lr, lc, rdi);
        sol = solve_lane<64, MM>(m, i, (in && !cl) ? gc : 0.0, lr, lc, rdi, clmask);
      }
      const double srch = (in && !cl) ? -sol - x : 0.0;
      const double sdotg = seq_sum(srch * grad);
      CCC_PROF_ADD(PR_BOXQP_SOLVE);
      double step = 1.0;
#if defined(CCC_DDP_STORE_FLOAT)
      // With single-precision storage the Newton step of an iterate that is optimal TO THAT RESOLUTION is rounding noise,
      // its sign against the gradient a coin toss: "no descent direction" then means converged (DESIGN.md 7c has
      // the measurements behind the rule)
      if(sdotg >= 0)
      {
        result = 5;
        break;
      }
#else
      if(sdotg >= 0) break; // no descent direction: result stays 0
#endif
      double xc, vc;
      for(;;)
      {
        xc = in ? fmin(fmax(x + step * srch, lo), hi) : 0.0;
        vc = value_of(xc);
        if(!((vc - oldvalue) / (step * sdotg) < armijo)) break;
        step *= step_dec;
        if(step < min_step)
        {
          result = 2;
          break;
        }
      }
      x = xc;
      value = vc;
      CCC_PROF_ADD(PR_BOXQP_LINESEARCH);
    }
    __syncthreads();
    if(lane < m)
    {
      mem.kq[lane] = x;
      mem.clamped[lane] = cl ? 1 : 0;
    }
    __syncthreads();
    if(iter > max_iter && result == 0) result = 1;
    return result;
  }
#else
  CCC_DDP_FN int box_qp(int m)
  {
    const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
    const int max_iter = 500; // nmpc_ddp BoxQP::Configuration::max_iter
    const St * H = mem.QuuF;
    const St * g = mem.Qu;
    phase([&](int lane) {
      if(lane < m)
      {
        mem.kq[lane] = fmin(fmax(mem.kq[lane], mem.lo[lane]), mem.hi[lane]);
        mem.clamped[lane] = 0;
      }
      if(lane == 0) mem.ic[IC_RESULT] = 0;
    });
    auto value_of = [&](const double * x, int slot) {
      // value = x'g + 1/2 x'Hx, rows in parallel then a sequential sum (same order as the oracle)
      phase([&](int lane) {
        if(lane < m)
        {
          double s = 0;
          for(int j = 0; j < m; j++) s += H[lane * LQ + j] * x[j];
          mem.tmp[lane] = x[lane] * g[lane] + 0.5 * x[lane] * s;
        }
      });
      phase([&](int lane) {
        if(lane == 0)
        {
          double v = 0;
          for(int i = 0; i < m; i++) v += mem.tmp[i];
          mem.sc[slot] = v;
        }
      });
    };
    CCC_PROF_START();
    value_of(mem.kq, SC_VALUE);
    CCC_PROF_ADD(PR_BOXQP_VALUE);
#if defined(CCC_DDP_PROF) && defined(__HIP_DEVICE_COMPILE__)
    if((threadIdx.x & 63) == 0) mem.prof[PR_BOXQP_CALLS] += 1.0;
#endif
    int iter;
    for(iter = 1; iter <= max_iter; iter++)
    {
      if(mem.ic[IC_RESULT] != 0) break;
#if defined(CCC_DDP_PROF) && defined(__HIP_DEVICE_COMPILE__)
      if((threadIdx.x & 63) == 0) mem.prof[PR_BOXQP_ITERS] += 1.0;
#endif
      if(iter > 1 && (mem.sc[SC_OLDVALUE] - mem.sc[SC_VALUE]) < min_rel_improve * fabs(mem.sc[SC_OLDVALUE]))
      {
        phase([&](int lane) {
          if(lane == 0) mem.ic[IC_RESULT] = 4;
        });
        break;
      }
      phase([&](int lane) {
        if(lane == 0) mem.sc[SC_OLDVALUE] = mem.sc[SC_VALUE];
        if(lane < m)
        {
          double s = g[lane];
          for(int j = 0; j < m; j++) s += H[lane * LQ + j] * mem.kq[j];
          mem.grad[lane] = s;
          mem.oldc[lane] = mem.clamped[lane];
          mem.clamped[lane] = ((mem.kq[lane] == mem.lo[lane] && s > 0) || (mem.kq[lane] == mem.hi[lane] && s < 0)) ? 1 : 0;
        }
      });
      phase([&](int lane) {
        if(lane == 0)
        {
          int changed = (iter == 1), all = 1;
          for(int i = 0; i < m; i++)
          {
            if(mem.clamped[i] != mem.oldc[i]) changed = 1;
            if(!mem.clamped[i]) all = 0;
          }
          mem.ic[IC_CHANGED] = changed;
          mem.ic[IC_ALLCL] = all;
          if(all) mem.ic[IC_RESULT] = 6;
        }
      });
      CCC_PROF_ADD(PR_BOXQP_GRAD);
      if(mem.ic[IC_ALLCL]) break;
      if(mem.ic[IC_CHANGED])
      {
#if defined(CCC_DDP_PROF) && defined(__HIP_DEVICE_COMPILE__)
        if((threadIdx.x & 63) == 0) mem.prof[PR_BOXQP_CHOLS] += 1.0;
#endif
        if(!cholesky_free(m))
        {
          phase([&](int lane) {
            if(lane == 0) mem.ic[IC_RESULT] = -1;
          });
          break;
        }
      }
      CCC_PROF_ADD(PR_BOXQP_CHOL);
      phase([&](int lane) {
        if(lane == 0)
        {
          double gn = 0;
          for(int i = 0; i < m; i++)
            if(!mem.clamped[i]) gn += mem.grad[i] * mem.grad[i];
          gn = sqrt(gn);
          mem.sc[SC_GNORM] = gn;
          if(gn < min_grad) mem.ic[IC_RESULT] = 5;
        }
        // grad_clamped = g + H (x .* clamped) on the free rows
        if(lane < m)
        {
          double s = g[lane];
          for(int j = 0; j < m; j++)
            if(mem.clamped[j]) s += H[lane * LQ + j] * mem.kq[j];
          mem.tmp[lane] = mem.clamped[lane] ? 0.0 : s;
        }
      });
      CCC_PROF_ADD(PR_BOXQP_GRAD);
      if(mem.ic[IC_RESULT] != 0) break;
      solve_free(m, mem.tmp); // tmp <- H_ff^-1 tmp on the free rows
      CCC_PROF_ADD(PR_BOXQP_SOLVE);
      phase([&](int lane) {
        if(lane < m) mem.srch[lane] = mem.clamped[lane] ? 0.0 : -mem.tmp[lane] - mem.kq[lane];
      });
      phase([&](int lane) {
        if(lane == 0)
        {
          double s = 0;
          for(int i = 0; i < m; i++) s += mem.srch[i] * mem.grad[i];
          mem.sc[SC_SDOTG] = s;
          mem.sc[SC_STEP] = 1.0;
        }
      });
      if(mem.sc[SC_SDOTG] >= 0) break; // no descent direction: result stays 0
      for(;;)
      {
        phase([&](int lane) {
          if(lane < m)
            mem.xcand[lane] = fmin(fmax(mem.kq[lane] + mem.sc[SC_STEP] * mem.srch[lane], mem.lo[lane]), mem.hi[lane]);
        });
        value_of(mem.xcand, SC_VC);
        if(!((mem.sc[SC_VC] - mem.sc[SC_OLDVALUE]) / (mem.sc[SC_STEP] * mem.sc[SC_SDOTG]) < armijo)) break;
        bool stop = false;
        phase([&](int lane) {
          if(lane == 0)
          {
            mem.sc[SC_STEP] *= step_dec;
            if(mem.sc[SC_STEP] < min_step) mem.ic[IC_RESULT] = 2;
          }
        });
        stop = mem.ic[IC_RESULT] == 2;
        if(stop) break;
      }
      phase([&](int lane) {
        if(lane < m) mem.kq[lane] = mem.xcand[lane];
        if(lane == 0) mem.sc[SC_VALUE] = mem.sc[SC_VC];
      });
      CCC_PROF_ADD(PR_BOXQP_LINESEARCH);
    }
    int result = mem.ic[IC_RESULT];
    if(iter > max_iter && result == 0) result = 1;
    return result;
  }
#endif

  // Cholesky of H~ (H = mem.QuuF with the clamped rows/columns replaced by identity, i.e. the factor of H_ff
  // embedded) into mem.Lf (lower, stride m) + reciprocal diagonal mem.rd.  Arithmetic per entry exactly as the
  // oracle: subtract the products in increasing k, diagonal = sqrt, sub-diagonal = value * (1 / diagonal).
  //
  // Device: ONE phase, lane i (mod 16) keeps row i in registers; pivots and the scaled column travel by
  // v_readlane (constant lane numbers after unrolling) -- no LDS round trips, no barriers inside.
  CCC_DDP_FN bool cholesky_free(int m)
  {
#if CCC_DDP_FAST
    const unsigned long long clm = clamped_mask(m);
    if(m == M)
      cholesky_phase<M>(m, clm);
    else
      cholesky_phase<0>(m, clm);
    return mem.ic[IC_OK] != 0;
#else
    // Phase version, right-looking, lane = row: per column j one phase scales the column below the pivot, the next
    // writes the pivot and subtracts the column's outer product from row i (entries k = j+1 .. i).  Every entry
    // (i, k) thus has its products subtracted in increasing j, as the oracle does.
    const St * H = mem.QuuF;
    phase([&](int lane) {
      if(lane < m)
        for(int k = 0; k <= lane; k++)
        {
          const bool cl = mem.clamped[lane] || mem.clamped[k];
          mem.Lf[lane * LQ + k] = cl ? (lane == k ? 1.0 : 0.0) : H[lane * LQ + k];
        }
      if(lane == 0) mem.ic[IC_OK] = 1;
    });
    for(int j = 0; j < m; j++)
    {
      phase([&](int lane) {
        if(lane > j && lane < m)
        {
          const double r = 1.0 / sqrt(mem.Lf[j * LQ + j]);
          mem.Lf[lane * LQ + j] = mem.Lf[lane * LQ + j] * r;
        }
      });
      phase([&](int lane) {
        if(lane == j)
        {
          const double d = mem.Lf[j * LQ + j];
          if(!(d > 0.0)) mem.ic[IC_OK] = 0;
          const double sq = sqrt(d);
          mem.Lf[j * LQ + j] = sq;
          mem.rd[j] = 1.0 / sq;
        }
        else if(lane > j && lane < m)
        {
          const double lij = mem.Lf[lane * LQ + j];
          for(int k = j + 1; k <= lane; k++) mem.Lf[lane * LQ + k] -= lij * mem.Lf[k * LQ + j];
        }
      });
    }
    return mem.ic[IC_OK] != 0;
#endif
  }

#if CCC_DDP_FAST
  // MM = 16: every step has the full 16 ridges (the usual case) -- all size tests fold away and the factorisation is
  // straight-line code; MM = 0: size m at run time.
  // the clamped rows as a wavefront-uniform bit mask (bit i = row i): their columns of the factor are identity columns,
  // and every step below that would only multiply by their zeros is skipped -- exactly: x - 0 * y = x
  CCC_DDP_FN unsigned long long clamped_mask(int m) const
  {
    const int lane = static_cast<int>(threadIdx.x & 63);
    return __ballot(lane < m && mem.clamped[lane] != 0) & kRowMask;
  }

  template<int MM>
  CCC_DDP_FN void cholesky_phase(int m_rt, unsigned long long clmask)
  {
    const int m = MM ? MM : m_rt;
    phase([&](int lane) {
      const int i = lane & (M - 1);
      double a[M];
#  pragma unroll
      for(int k = 0; k < M; ++k)
      {
        const bool in = (i < m) && (k < m);
        const bool cl = in && (mem.clamped[i] || mem.clamped[k]);
        a[k] = in ? (cl ? (i == k ? 1.0 : 0.0) : quuF(i, k)) : (i == k ? 1.0 : 0.0);
      }
      bool ok = true;
      double rdi = 1.0;
      // The scaled column j reaches the other rows either by a readlane pair per (j, k), or through LDS: every lane
      // publishes its entry, all read the column back as broadcasts (constant addresses: 128-bit reads; mem.t4 is free
      // until the value update; LDS operations of a wavefront execute in order, so the reads of column j see its writes
      // and are done before column j + 1 is written).  Measured: LDS +1.6 % at M = 32 and +3 % for the 12-state model,
      // -3 % for the 9-state model at M = 16.
      constexpr bool kColumnViaLds = (S == 12);
      double * const col = mem.t4;
#  pragma unroll
      for(int j = 0; j < M; ++j)
      {
        if(j < m && !((clmask >> j) & 1ull)) // (a clamped column is e_j already: pivot 1, nothing below, nothing to subtract)
        {
          const double d = lane_value(a[j], j);
          ok = ok && (d > 0.0);
          const double sq = sqrt(d);
          const double r = 1.0 / sq;
          // (entries above the diagonal -- a[k] of a lane i < k -- are never read: they take part in the arithmetic
          //  unpredicated, which saves two selects per multiply-subtract)
          if(i == j)
          {
            a[j] = sq;
            rdi = r;
          }
          else
            a[j] = a[j] * r;
          if constexpr(kColumnViaLds)
          {
            if(lane < M) col[i] = a[j];
            __builtin_amdgcn_wave_barrier();
          }
#  pragma unroll
          for(int k = j + 1; k < M; ++k)
          {
            if(k < m)
            {
              const double lkj = kColumnViaLds ? col[k] : lane_value(a[j], k);
              a[k] -= a[j] * lkj;
            }
          }
          if constexpr(kColumnViaLds) __builtin_amdgcn_wave_barrier();
        }
      }
      if(lane < m)
      {
#  pragma unroll
        for(int k = 0; k < M; ++k)
          if(k <= i) mem.Lf[i * LQ + k] = a[k];
        mem.rd[i] = rdi;
      }
      if(lane == 0) mem.ic[IC_OK] = ok ? 1 : 0;
    });
  }
#endif

  // registers of lane i for the triangular solves: row i of L left of the diagonal, column i below it
#if CCC_DDP_FAST
  template<int MM>
  CCC_DDP_FN void load_factor_lane(int m_rt, int i, double (&lr)[M], double (&lc)[M], double & rdi) const
  {
    const int m = MM ? MM : m_rt;
#  pragma unroll
    for(int k = 0; k < M; ++k)
    {
      lr[k] = (i < m && k < i) ? mem.Lf[i * LQ + k] : 0.0;
      lc[k] = (i < m && k < m && k > i) ? mem.Lf[k * LQ + i] : 0.0;
    }
    rdi = (i < m) ? mem.rd[i] : 1.0;
  }
  // acc <- (L L')^-1 acc inside a 16-lane group (WIDTH = 16) or with every group redundant (WIDTH = 64)
  template<int WIDTH, int MM>
  static CCC_DDP_FN double solve_lane(int m_rt, int i, double acc, const double (&lr)[M], const double (&lc)[M], double rdi,
                                      unsigned long long clmask)
  {
    const int m = MM ? MM : m_rt;
#  pragma unroll
    for(int k = 0; k < M; ++k)
    {
      if(k < m && !((clmask >> k) & 1ull)) // (clamped: right-hand side and solution entry are zero)
      {
        const double yk = (WIDTH == 64) ? lane_value(acc * rdi, k) : __shfl(acc * rdi, k, WIDTH);
        if(i == k)
          acc = yk;
        else if(i > k)
          acc -= lr[k] * yk;
      }
    }
#  pragma unroll
    for(int k = M - 1; k >= 0; --k)
    {
      if(k < m && !((clmask >> k) & 1ull))
      {
        const double zk = (WIDTH == 64) ? lane_value(acc * rdi, k) : __shfl(acc * rdi, k, WIDTH);
        if(i == k)
          acc = zk;
        else if(i < k)
          acc -= lc[k] * zk;
      }
    }
    return acc;
  }
#endif

  // v <- H~^-1 v using mem.Lf / mem.rd: forward substitution in increasing, backward in DEcreasing column order,
  // multiplying by the reciprocal diagonal (the oracle's order).  v is an LDS vector of length m.
  CCC_DDP_FN void solve_free(int m, double * v)
  {
#if CCC_DDP_FAST
    if(m == M)
      solve_free_phase<M>(m, v);
    else
      solve_free_phase<0>(m, v);
#else
    phase([&](int lane) {
      if(lane == 0) solve_free_seq(m, v);
    });
#endif
  }

  // the two substitutions by ONE lane (phase versions: box-QP direction by lane 0, the S gain columns by S lanes)
  CCC_DDP_FN void solve_free_seq(int m, double * v) const
  {
    for(int a = 0; a < m; a++)
    {
      double s = v[a];
      for(int k = 0; k < a; k++) s -= mem.Lf[a * LQ + k] * v[k];
      v[a] = s * mem.rd[a];
    }
    for(int a = m - 1; a >= 0; a--)
    {
      double s = v[a];
      for(int k = m - 1; k > a; k--) s -= mem.Lf[k * LQ + a] * v[k];
      v[a] = s * mem.rd[a];
    }
  }

#if CCC_DDP_FAST
  template<int MM>
  CCC_DDP_FN void solve_free_phase(int m_rt, double * v)
  {
    const int m = MM ? MM : m_rt;
    const unsigned long long clm = clamped_mask(m);
    phase([&](int lane) {
      const int i = lane & (M - 1);
      double lr[M], lc[M], rdi;
      load_factor_lane<MM>(m, i, lr, lc, rdi);
      const double acc = solve_lane<64, MM>(m, i, (i < m) ? v[i] : 0.0, lr, lc, rdi, clm);
      if(lane < m) v[i] = acc;
    });
  }

  // Device gains: lane a (< S) runs the two triangular solves of ITS right-hand side (state index a) start to finish
  // with the factor read from LDS (every lane reads the same entry: a broadcast) -- the host statement order, no
  // cross-lane traffic and therefore no LDS-crossbar latency inside the 2 x 16-step recurrences.
  template<int MM>
  CCC_DDP_FN void gains_phase(int m_rt)
  {
    const int m = MM ? MM : m_rt;
    const unsigned long long clm = clamped_mask(m);
    phase([&](int lane) {
      if(lane < S)
      {
        const int a = lane;
        double t3[M];
#if CCC_DDP_REG1_ONLY
        const St * const Qxur = mem.Qxu;
#else
        const St * const Qxur = P.reg_type == 2 ? mem.Qxur : mem.Qxu; // (reg_type 1: the same matrix)
#endif
#  pragma unroll
        for(int f = 0; f < M; f++) t3[f] = (f < m && !mem.clamped[f]) ? Qxur[a * M + f] : 0.0;
#  pragma unroll
        for(int r = 0; r < M; r++)
        {
          if(r < m && !((clm >> r) & 1ull)) // (a clamped row: t3[r] is zero and stays zero)
          {
            double sum = t3[r];
#  pragma unroll
            for(int k = 0; k < M; k++)
              if(k < r) sum -= mem.Lf[r * LQ + k] * t3[k];
            t3[r] = sum * mem.rd[r];
          }
        }
#  pragma unroll
        for(int r = M - 1; r >= 0; r--)
        {
          if(r < m && !((clm >> r) & 1ull))
          {
            double sum = t3[r];
#  pragma unroll
            for(int k = M - 1; k >= 0; k--)
              if(k > r && k < m) sum -= mem.Lf[k * LQ + r] * t3[k];
            t3[r] = sum * mem.rd[r];
          }
        }
#  pragma unroll
        for(int f = 0; f < M; f++)
          if(f < m) mem.K[f * S + a] = mem.clamped[f] ? 0.0 : -t3[f];
      }
      if(lane < m) mem.k[lane] = mem.kq[lane];
    });
  }
#endif

  // K_f = -H_ff^-1 Qxur_f' (clamped rows of K stay zero), k <- kq.  Device: the four 16-lane groups of the
  // wavefront each solve one right-hand side (state index) at a time.
  CCC_DDP_FN void gains(int m)
  {
#if CCC_DDP_FAST
    if(m == M)
      gains_phase<M>(m);
    else
      gains_phase<0>(m);
#else
    // lane a < S solves for state index a; its right-hand side lives in row a of mem.T2 (free between the products
    // and the value update)
    phase([&](int lane) {
      if(lane < S)
      {
        const int a = lane;
        St * t3 = mem.T2 + a * M;
        for(int f = 0; f < m; f++) t3[f] = mem.clamped[f] ? 0.0 : mem.Qxur[a * M + f];
        solve_free_seq(m, t3);
        for(int f = 0; f < m; f++) mem.K[f * S + a] = mem.clamped[f] ? 0.0 : -t3[f];
      }
      else if(lane >= 32 && lane - 32 < m)
        mem.k[lane - 32] = mem.kq[lane - 32];
    });
#endif
  }

#if CCC_DDP_FAST
  // Device matrix product with one OUTPUT COLUMN per lane: lane (g, c) = (lane / 16, lane % 16) keeps column c of the
  // right factor in registers and produces C(a, c) for the rows a = g, g + 4, ..; the left factor's entries are read
  // from LDS as broadcasts (one address per 16-lane group).  The element-per-lane loops of the phase version read two
  // LDS operands per multiply-add and are bound by LDS bandwidth (8 wavefronts per CU); this form reads half as many
  // words, almost all of them broadcasts.  Sums run over k = 0 .. S-1 in increasing order from the same start value:
  // identical results.   C(a,c) = [DIAG] + sum_k Aop(a,k) B(k,c),  Aop = A or A' (TRANS), optionally A + lam I (LAM);
  //   DIAG 0: none, 1: w_run[a] on a == c, 2: w_force on a == c, 3: as 2 and lam added to the diagonal after the sum.
  //   C2 (optional): a second copy of C with lam added to the diagonal after the sum (Quu and Quu + lambda I at once).
  template<bool TRANS, int DIAG, bool LAM>
  CCC_DDP_FN void colprod(int lane, int rows, int ncols, const St * A, int lda, const St * B, int ldb, double lam,
                          St * C, int ldc, St * C2 = nullptr) const
  {
    const int c = lane & (M - 1), g = lane / M;
    const bool act = c < ncols;
    double Bc[S];
#  pragma unroll
    for(int k = 0; k < S; k++) Bc[k] = act ? static_cast<double>(B[k * ldb + c]) : 0.0;
    for(int a = g; a < rows; a += kWave / M)
    {
      double sum = 0.0;
      if(DIAG == 1) sum = (a == c) ? P.w_run[a] : 0.0;
      if(DIAG == 2 || DIAG == 3) sum = (a == c) ? P.w_force : 0.0;
#  pragma unroll
      for(int k = 0; k < S; k++)
      {
        double av = TRANS ? static_cast<double>(A[k * lda + a]) : static_cast<double>(A[a * lda + k]);
        if(LAM) av = av + (a == k ? lam : 0.0);
        sum += av * Bc[k];
      }
      if(DIAG == 3) sum = (a == c) ? sum + lam : sum;
      if(act) C[a * ldc + c] = sum;
      if(C2 && act) C2[a * ldc + c] = (a == c) ? sum + lam : sum;
    }
  }
#endif

  // ---- backward pass (oracle/ddp.c backward_pass); returns false when a box-QP / Cholesky fails
  CCC_DDP_FN bool backward_pass()
  {
    const int N = P.N;
    phase([&](int lane) {
      // terminal value: src/DdpCentroidal.cpp:156-177 at x_N
      for(int e = lane; e < S * S; e += kWave) mem.Vxx[e] = (e / S == e % S) ? P.w_term[e / S] : 0.0;
      if(lane < S) mem.Vx[lane] = P.w_term[lane] * (I.xs[static_cast<long>(N) * S + lane] - ref_entry(N, lane));
      if(lane == 0)
      {
        mem.sc[SC_DV0] = 0.0;
        mem.sc[SC_DV1] = 0.0;
      }
    });
    StepRegs sr;
#if CCC_DDP_FAST
    fetch_step(N - 1, static_cast<int>(threadIdx.x & 63), sr);
#endif
    for(int i = N - 1; i >= 0; i--)
    {
      const int m = dim_of(i);
      // the usual case of a full 16-ridge contact gets its own instantiation: index arithmetic by constants, loops the
      // compiler can unroll (same statements, same results)
      const bool ok = (m == M) ? backward_step<M>(i, m, sr) : backward_step<0>(i, m, sr);
      if(!ok) return false;
    }
    return true;
  }

  // Device: the per-step operands that live in HBM (nominal x_i, u_i, the reference entry and the feed-forward term
  // of step i+1 used as box-QP warm start) are carried in registers -- x/u/ref of step i-1 are requested while step i
  // computes, k of step i+1 is simply kept -- so that no backward step starts with an HBM round trip.
  struct StepRegs
  {
    double x = 0, u = 0, ref = 0, kprev = 0;
  };
  CCC_DDP_FN void fetch_step(int i, int lane, StepRegs & r) const
  {
    r.x = (lane < S) ? I.xs[static_cast<long>(i) * S + lane] : 0.0;
    r.u = (lane < M) ? I.us[static_cast<long>(i) * M + lane] : 0.0;
    r.ref = (lane < S) ? ref_entry(i, lane) : 0.0;
  }

  template<int MM>
  CCC_DDP_FN bool backward_step(int i, int m_rt, StepRegs & sr)
  {
    const int N = P.N;
    {
      const int m = MM ? MM : m_rt;
      const double lambda_v = P.reg_type == 2 ? mem.sc[SC_LAMBDA] : 0.0;
      const double lambda_q = P.reg_type == 2 ? 0.0 : mem.sc[SC_LAMBDA];
#if CCC_DDP_FAST
      const StepRegs cur = sr;
      {
        const int lane = static_cast<int>(threadIdx.x & 63);
        if(i > 0) fetch_step(i - 1, lane, sr); // sr.kprev is set further down
      }
#endif
      phase([&](int lane) {
#if CCC_DDP_FAST
        if(lane < S) mem.x[lane] = cur.x;
        if(lane < M) mem.u[lane] = (lane < m) ? cur.u : 0.0;
#else
        if(lane < S) mem.x[lane] = I.xs[static_cast<long>(i) * S + lane];
        if(lane < M) mem.u[lane] = (lane < m) ? I.us[static_cast<long>(i) * M + lane] : 0.0;
#endif
      });
      CCC_PROF_START();
      state_eq_deriv(i);
      CCC_PROF_ADD(PR_DERIV);
      phase([&](int lane) {
        // Qx = Lx + Fx'Vx ; Qu = Lu + Fu'Vx
        if(lane < S)
        {
#if CCC_DDP_FAST
          double s = P.w_run[lane] * (mem.x[lane] - cur.ref);
#else
          double s = P.w_run[lane] * (mem.x[lane] - ref_entry(i, lane));
#endif
          for(int b = 0; b < S; b++) s += mem.Fx[b * S + lane] * mem.Vx[b];
          mem.Qx[lane] = s;
        }
        else if(lane >= 32 && lane - 32 < m)
        {
          const int r = lane - 32;
          double s = P.w_force * mem.u[r];
          for(int b = 0; b < S; b++) s += mem.Fu[b * M + r] * mem.Vx[b];
          mem.Qu[r] = s;
        }
        // T1 = Vxx Fx ; T2 = Vxx Fu ; regularised T2r = (Vxx + lambda I) Fu, parked in mem.Lf (free until the box-QP
        // factorises) so that everything built from T1 / T2 / T2r fits one more phase
        St * const T2r = mem.Lf;
#if CCC_DDP_FAST
        colprod<false, 0, false>(lane, S, S, mem.Vxx, S, mem.Fx, S, 0.0, mem.T1, S);
        colprod<false, 0, false>(lane, S, m, mem.Vxx, S, mem.Fu, M, 0.0, mem.T2, M);
        // (reg_type 1 regularises Quu, not Vxx: T2r = T2, Qxur = Qxu and Quu_F = Quu + lambda I -- the same sums, so the
        //  three products are not repeated)
        if(P.reg_type == 2) colprod<false, 0, true>(lane, S, m, mem.Vxx, S, mem.Fu, M, lambda_v, T2r, M);
#else
        for(int e = lane; e < S * m; e += kWave)
        {
          const int a = e / m, r = e % m;
          double s = 0;
          for(int k = 0; k < S; k++) s += (mem.Vxx[a * S + k] + (a == k ? lambda_v : 0.0)) * mem.Fu[k * M + r];
          T2r[a * M + r] = s;
        }
        for(int e = lane; e < S * S; e += kWave)
        {
          const int a = e / S, b = e % S;
          double s = 0;
          for(int k = 0; k < S; k++) s += mem.Vxx[a * S + k] * mem.Fx[k * S + b];
          mem.T1[e] = s;
        }
        for(int e = lane; e < S * m; e += kWave)
        {
          const int a = e / m, r = e % m;
          double s = 0;
          for(int k = 0; k < S; k++) s += mem.Vxx[a * S + k] * mem.Fu[k * M + r];
          mem.T2[a * M + r] = s;
        }
#endif
      });
      phase([&](int lane) {
        // Qxx = Lxx + Fx'T1 ; Qxu = Fx'T2 ; Quu = Luu + Fu'T2   (Lxu = 0, Lxx = diag(w_run), Luu = w_force I)
#if CCC_DDP_FAST
        colprod<true, 1, false>(lane, S, S, mem.Fx, S, mem.T1, S, 0.0, mem.Qxx, S);
        colprod<true, 0, false>(lane, S, m, mem.Fx, S, mem.T2, M, 0.0, mem.Qxu, M);
#if CCC_DDP_REG1_ONLY
        colprod<true, 2, false>(lane, m, m, mem.Fu, M, mem.T2, M, lambda_q, mem.Quu, LQ);
#else
        colprod<true, 2, false>(lane, m, m, mem.Fu, M, mem.T2, M, lambda_q, mem.Quu, LQ, P.reg_type == 2 ? nullptr : mem.QuuF);
#endif
#else
        for(int e = lane; e < S * S; e += kWave)
        {
          const int a = e / S, b = e % S;
          double s = (a == b) ? P.w_run[a] : 0.0;
          for(int k = 0; k < S; k++) s += mem.Fx[k * S + a] * mem.T1[k * S + b];
          mem.Qxx[e] = s;
        }
        for(int e = lane; e < S * m; e += kWave)
        {
          const int a = e / m, r = e % m;
          double s = 0.0;
          for(int k = 0; k < S; k++) s += mem.Fx[k * S + a] * mem.T2[k * M + r];
          mem.Qxu[a * M + r] = s;
        }
        for(int e = lane; e < m * m; e += kWave)
        {
          const int r = e / m, q = e % m;
          double s = (r == q) ? P.w_force : 0.0;
          for(int k = 0; k < S; k++) s += mem.Fu[k * M + r] * mem.T2[k * M + q];
          mem.Quu[r * LQ + q] = s;
        }
#endif
        // regularised versions from T2r
        const St * const T2r = mem.Lf;
#if CCC_DDP_REG1_ONLY
        (void)T2r;
#elif CCC_DDP_FAST
        if(P.reg_type == 2)
        {
          colprod<true, 0, false>(lane, S, m, mem.Fx, S, T2r, M, 0.0, mem.Qxur, M);
          colprod<true, 3, false>(lane, m, m, mem.Fu, M, T2r, M, lambda_q, mem.QuuF, LQ);
        }
#else
        for(int e = lane; e < S * m; e += kWave)
        {
          const int a = e / m, r = e % m;
          double s = 0.0;
          for(int k = 0; k < S; k++) s += mem.Fx[k * S + a] * T2r[k * M + r];
          mem.Qxur[a * M + r] = s;
        }
        for(int e = lane; e < m * m; e += kWave)
        {
          const int r = e / m, q = e % m;
          double s = (r == q) ? P.w_force : 0.0;
          for(int k = 0; k < S; k++) s += mem.Fu[k * M + r] * T2r[k * M + q];
          if(r == q) s = s + lambda_q;
          mem.QuuF[r * LQ + q] = s;
        }
#endif
        // box limits on the input CHANGE and the warm start (gain of step i+1 of this pass, zeros on a dim change)
        if(lane < M)
        {
          mem.lo[lane] = P.flo - mem.u[lane];
          mem.hi[lane] = P.fhi - mem.u[lane];
          const bool warm = (i + 1 < N) && (dim_of(i + 1) == m);
#if CCC_DDP_FAST
          mem.kq[lane] = (warm && lane < m) ? cur.kprev : 0.0;
#else
          mem.kq[lane] = (warm && lane < m) ? I.ks[static_cast<long>(i + 1) * M + lane] : 0.0;
#endif
          mem.k[lane] = 0.0;
        }
        for(int e = lane; e < M * S; e += kWave) mem.K[e] = 0.0;
      });
      CCC_PROF_ADD(PR_PRODUCTS);
      if(m > 0)
      {
        const int rc = box_qp(m);
        if(rc < 1) return false;
        CCC_PROF_RESTART();
        gains(m);
        CCC_PROF_ADD(PR_GAINS);
      }
      CCC_PROF_RESTART();
      phase([&](int lane) {
        // t4 = Quu k ; gains to global memory
        if(lane < m)
        {
          double s = 0;
          for(int q = 0; q < m; q++) s += mem.Quu[lane * LQ + q] * mem.k[q];
          mem.t4[lane] = s;
        }
        if(lane < M)
        {
          I.ks[static_cast<long>(i) * M + lane] = mem.k[lane];
#if CCC_DDP_FAST
          sr.kprev = mem.k[lane];
#endif
        }
        for(int e = lane; e < M * S; e += kWave) I.Ks[static_cast<long>(i) * M * S + e] = mem.K[e];
        // T2 (S x m) = K' Quu
        for(int e = lane; e < S * m; e += kWave)
        {
          const int a = e / m, r = e % m;
          double s = 0;
          for(int q = 0; q < m; q++) s += mem.K[q * S + a] * mem.Quu[q * LQ + r];
          mem.T2[a * M + r] = s;
        }
      });
      phase([&](int lane) {
        if(lane == 63)
        {
          double s0 = 0, s1 = 0;
          for(int r = 0; r < m; r++)
          {
            s0 += mem.k[r] * mem.Qu[r];
            s1 += mem.k[r] * mem.t4[r];
          }
          mem.sc[SC_DV0] += s0;
          mem.sc[SC_DV1] += 0.5 * s1;
        }
        if(lane < S)
        {
          const int a = lane;
          double s = mem.Qx[a];
          for(int r = 0; r < m; r++)
            s += mem.K[r * S + a] * mem.t4[r] + mem.K[r * S + a] * mem.Qu[r] + mem.Qxu[a * M + r] * mem.k[r];
          mem.Vx[a] = s;
        }
        for(int e = lane; e < S * S; e += kWave)
        {
          const int a = e / S, b = e % S;
          double s = mem.Qxx[e];
          for(int r = 0; r < m; r++)
            s += mem.T2[a * M + r] * mem.K[r * S + b] + mem.K[r * S + a] * mem.Qxu[b * M + r]
                 + mem.Qxu[a * M + r] * mem.K[r * S + b];
          mem.T1[e] = s;
        }
      });
      phase([&](int lane) {
        for(int e = lane; e < S * S; e += kWave)
        {
          const int a = e / S, b = e % S;
          mem.Vxx[e] = 0.5 * (mem.T1[a * S + b] + mem.T1[b * S + a]);
        }
      });
      CCC_PROF_ADD(PR_VALUE_UPDATE);
    }
    return true;
  }

  // ---- rollout of the initial inputs / line-search candidate.  alpha < 0: plain rollout of I.us into I.xs.
#if CCC_DDP_FAST
  // Device rollout for the centroidal model (S = 9): ONE phase for the whole trajectory.  Lane a keeps state entry a,
  // lane r the input of ridge r; the feedback, the dynamics and the cost reach across lanes with v_readlane in the
  // oracle's summation order (no LDS staging, no barriers per step), and the gains / nominal trajectory of the next
  // step are fetched from HBM while the current step computes.  Statement by statement the arithmetic is that of the
  // phase version below (state_eq, running_cost, terminal_cost), which the host emulation keeps using.
  CCC_DDP_FN void rollout_centroidal(double alpha)
  {
    const int N = P.N;
    const bool initial = alpha < 0;
    const bool cold = alpha < -1.5; // the warm-start guard's rollout of zero inputs from x0 into the candidate buffers
    double * xo = (initial && !cold) ? I.xs : I.xc;
    double * uo = (initial && !cold) ? I.us : I.uc;
    const int lane = static_cast<int>(threadIdx.x & 63);
    const bool st = lane < S;
    double x = st ? (initial ? I.x0[lane] : I.xs[lane]) : 0.0;
    if(st) xo[lane] = x;
    const double wr = st ? P.w_run[lane < S ? lane : 0] : 0.0, wt = st ? P.w_term[lane < S ? lane : 0] : 0.0;
    double cost = 0.0; // accumulated by lane kCostLane only
    constexpr int kCostLane = 9, kTermLane = 10;
    // per-step staging of the products whose ordered sums the dynamics and the cost need: rows 0-2 u_r rho_r,
    // 3-5 u_r (p_r - c) x rho_r, 6 u_r^2 (columns = ridges), row 7 the state-cost terms.  The block of backward-pass
    // matrices Qxx .. of Mem is idle during a rollout.
#if defined(CCC_DDP_STORE_FLOAT)
    double * const pr = mem.stage; // (the backward-pass block is single precision in this build)
#else
    double * const pr = mem.Qxx;
#endif
    static_assert(8 * M <= 2 * S * S + 3 * S * M, "staging area too small");
    // operands of step 0 (the next step's are fetched from HBM while the current one computes)
    double us_c = 0.0, ks_c = 0.0, Kr_c[S], xi_c[S], ref_c = 0.0;
    auto fetch = [&](int i, double & us_v, double & ks_v, double (&Kr_v)[S], double (&xi_v)[S], double & ref_v) {
      const int ln = lane < M ? lane : 0;
      if(initial)
        us_v = (I.u_init && !cold) ? I.u_init[static_cast<long>(i) * M + ln] : 0.0;
      else
      {
        us_v = I.us[static_cast<long>(i) * M + ln];
        ks_v = I.ks[static_cast<long>(i) * M + ln];
        const double * Kr = I.Ks + (static_cast<long>(i) * M + ln) * S;
        const double * xi = I.xs + static_cast<long>(i) * S;
#  pragma unroll
        for(int a = 0; a < S; a++)
        {
          Kr_v[a] = Kr[a];
          xi_v[a] = xi[a];
        }
      }
      ref_v = (lane < 3) ? I.ref_pos[static_cast<long>(i) * 3 + lane] : 0.0;
    };
#  pragma unroll
    for(int a = 0; a < S; a++) Kr_c[a] = xi_c[a] = 0.0;
    fetch(0, us_c, ks_c, Kr_c, xi_c, ref_c);
    // which ordered sum this lane owns: lanes 3-8 the six force / moment sums, kCostLane sum u^2, kTermLane the state cost
    const int myrow = (lane >= 3 && lane < 9) ? lane - 3 : (lane == kCostLane ? 6 : 7);
    const bool sums = (lane >= 3 && lane <= kTermLane);
    for(int i = 0; i < N; i++)
    {
      double us_n = 0.0, ks_n = 0.0, Kr_n[S], xi_n[S], ref_n = 0.0;
#  pragma unroll
      for(int a = 0; a < S; a++) Kr_n[a] = xi_n[a] = 0.0;
      if(i + 1 < N)
        fetch(i + 1, us_n, ks_n, Kr_n, xi_n, ref_n);
      else
        ref_n = (lane < 3) ? I.ref_pos[static_cast<long>(N) * 3 + lane] : 0.0; // terminal reference
      const int m = __builtin_amdgcn_readfirstlane(dim_of(i));
      const double * V = vert_of(i);
      const double * R = ridge_of(i);
      double xa[S];
#  pragma unroll
      for(int a = 0; a < S; a++) xa[a] = lane_value(x, a);
      // input of ridge `lane`
      double u = 0.0;
      if(lane < m)
      {
        if(initial)
          u = us_c;
        else
        {
          double s = us_c + alpha * ks_c;
#  pragma unroll
          for(int a = 0; a < S; a++) s += Kr_c[a] * (xa[a] - xi_c[a]);
          u = fmin(fmax(s, P.flo), P.fhi);
        }
      }
      if(lane < M) uo[static_cast<long>(i) * M + lane] = u;
      // dynamics, src/DdpCentroidal.cpp:32-64: lane r forms the products of its own ridge, lane a its state-cost term;
      // after one LDS round trip lanes 3-10 each add up one row in increasing index -- the oracle's order
      __syncthreads();
      if(lane < m)
      {
        const double rr[3] = {R[lane * 3], R[lane * 3 + 1], R[lane * 3 + 2]};
        const double d[3] = {V[lane * 3] - xa[0], V[lane * 3 + 1] - xa[1], V[lane * 3 + 2] - xa[2]};
        double c[3];
        cross3(d, rr, c);
#  pragma unroll
        for(int k = 0; k < 3; k++)
        {
          pr[k * M + lane] = u * rr[k];
          pr[(3 + k) * M + lane] = u * c[k];
        }
        pr[6 * M + lane] = u * u;
      }
      if(st)
      {
        const double e = x - ref_c;
        pr[7 * M + lane] = 0.5 * wr * e * e;
      }
      __syncthreads();
      double acc = (lane == 5) ? -1 * P.mass * kGravity : 0.0;
      if(sums)
      {
        const int cnt = (lane == kTermLane) ? S : m;
        const double * row = pr + myrow * M;
        for(int r = 0; r < cnt; r++) acc += row[r];
      }
      const double cterm = lane_value(acc, kTermLane);
      if(lane == kCostLane) cost += cterm + 0.5 * P.w_force * acc;
      double xd = 0.0;
      if(lane < 3)
        xd = (lane == 0 ? xa[3] : (lane == 1 ? xa[4] : xa[5])) / P.mass;
      else if(lane < 9)
        xd = acc;
      x = x + P.dt * xd;
      if(st) xo[static_cast<long>(i + 1) * S + lane] = x;
      us_c = us_n;
      ks_c = ks_n;
      ref_c = ref_n;
#  pragma unroll
      for(int a = 0; a < S; a++)
      {
        Kr_c[a] = Kr_n[a];
        xi_c[a] = xi_n[a];
      }
    }
    {
      const double e = x - ref_c;
      const double term = 0.5 * wt * e * e;
      double c = 0;
#  pragma unroll
      for(int a = 0; a < S; a++) c += lane_value(term, a);
      if(lane == kCostLane) cost += c;
    }
    __syncthreads();
    if(st) mem.x[lane] = x;
    if(lane == kCostLane) mem.sc[(initial && !cold) ? SC_COST : SC_COSTC] = cost;
    __syncthreads();
  }
#endif

  CCC_DDP_FN void rollout(double alpha)
  {
#if CCC_DDP_FAST
    if constexpr(S == 9)
    {
      rollout_centroidal(alpha);
      return;
    }
#endif
    const int N = P.N;
    const bool initial = alpha < 0;
    const bool cold = alpha < -1.5; // the warm-start guard's rollout of zero inputs from x0 into the candidate buffers
    const int cslot = (initial && !cold) ? SC_COST : SC_COSTC;
    double * xo = (initial && !cold) ? I.xs : I.xc;
    double * uo = (initial && !cold) ? I.us : I.uc;
    phase([&](int lane) {
      if(lane < S)
      {
        const double v = initial ? I.x0[lane] : I.xs[lane];
        mem.x[lane] = v;
        xo[lane] = v;
      }
      if(lane == 0) mem.sc[cslot] = 0.0;
    });
    for(int i = 0; i < N; i++)
    {
      const int m = dim_of(i);
      phase([&](int lane) {
        if(lane < M)
        {
          double s = 0.0;
          if(lane < m)
          {
            if(initial)
              s = (I.u_init && !cold) ? I.u_init[static_cast<long>(i) * M + lane] : 0.0;
            else
            {
              s = I.us[static_cast<long>(i) * M + lane] + alpha * I.ks[static_cast<long>(i) * M + lane];
              const double * Kr = I.Ks + (static_cast<long>(i) * M + lane) * S;
              const double * xi = I.xs + static_cast<long>(i) * S;
              for(int a = 0; a < S; a++) s += Kr[a] * (mem.x[a] - xi[a]);
              s = fmin(fmax(s, P.flo), P.fhi);
            }
          }
          mem.un[lane] = s;
          uo[static_cast<long>(i) * M + lane] = s;
        }
      });
      state_eq(i, mem.x, mem.un, mem.xn);
      phase([&](int lane) {
        if(lane == 32) mem.sc[cslot] += running_cost(i, mem.x, mem.un);
        if(lane < S) xo[static_cast<long>(i + 1) * S + lane] = mem.xn[lane];
      });
      phase([&](int lane) {
        if(lane < S) mem.x[lane] = mem.xn[lane];
      });
    }
    phase([&](int lane) {
      if(lane == 0) mem.sc[cslot] += terminal_cost(mem.x);
    });
  }

  CCC_DDP_FN void increase_lambda()
  {
    phase([&](int lane) {
      if(lane == 0)
      {
        mem.sc[SC_DLAMBDA] = fmax(mem.sc[SC_DLAMBDA] * P.lambda_factor, P.lambda_factor);
        mem.sc[SC_LAMBDA] = fmax(mem.sc[SC_LAMBDA] * mem.sc[SC_DLAMBDA], P.lambda_min);
      }
    });
  }
  CCC_DDP_FN void decrease_lambda()
  {
    phase([&](int lane) {
      if(lane == 0)
      {
        mem.sc[SC_DLAMBDA] = fmin(mem.sc[SC_DLAMBDA] / P.lambda_factor, 1.0 / P.lambda_factor);
        mem.sc[SC_LAMBDA] = mem.sc[SC_LAMBDA] * mem.sc[SC_DLAMBDA] * (mem.sc[SC_LAMBDA] > P.lambda_min ? 1.0 : 0.0);
      }
    });
  }

  // ---- the whole solve (oracle/ddp.c oracle_ddp_solve)
  CCC_DDP_FN void solve()
  {
    const int N = P.N;
    stage_problem();
    phase([&](int lane) {
      if(lane == 0)
      {
        mem.sc[SC_LAMBDA] = P.lambda0;
        mem.sc[SC_DLAMBDA] = P.dlambda0;
      }
#if defined(CCC_DDP_PROF)
      if(lane < 16) mem.prof[lane] = 0.0;
#endif
    });
#if defined(CCC_DDP_PROF) && defined(__HIP_DEVICE_COMPILE__)
    const long long prof_begin_ = (long long)__builtin_readcyclecounter();
#endif
    rollout(-1.0);
    if(P.warm_guard && I.u_init)
    {
      // warm-start guard (oracle/ddp.c): keep the warm start only if it rolls out no worse than zero inputs
      rollout(-2.0);
      if(!(mem.sc[SC_COST] <= mem.sc[SC_COSTC]))
        phase([&](int lane) {
          for(int e = lane; e < (N + 1) * S; e += kWave) I.xs[e] = I.xc[e];
          for(int e = lane; e < N * M; e += kWave) I.us[e] = I.uc[e];
          if(lane == 0) mem.sc[SC_COST] = mem.sc[SC_COSTC];
        });
    }
    int iter = 0, status = 0;
    for(iter = 1; iter <= P.max_iter; iter++)
    {
      bool bp_ok = false;
      for(;;)
      {
        if(backward_pass())
        {
          bp_ok = true;
          break;
        }
        increase_lambda();
        if(mem.sc[SC_LAMBDA] > P.lambda_max) break;
      }
      if(!bp_ok)
      {
        status = -1;
        break;
      }
      phase([&](int lane) {
        if(lane == 0)
        {
          double g = 0;
          for(int i = 0; i < N; i++)
          {
            const int m = dim_of(i);
            double mx = 0;
            for(int r = 0; r < m; r++)
            {
              const double v =
                  fabs(I.ks[static_cast<long>(i) * M + r]) / (fabs(I.us[static_cast<long>(i) * M + r]) + 1.0);
              if(v > mx) mx = v;
            }
            g += mx;
          }
          mem.sc[SC_G] = g / N;
        }
      });
      if(mem.sc[SC_G] < P.k_rel_norm_thre && mem.sc[SC_LAMBDA] < P.lambda_thre)
      {
        decrease_lambda();
        status = 1;
        break;
      }
      bool accepted = false;
      double actual = 0;
      for(int a = 0; a < 11; a++)
      {
        const double alpha = P.alpha[a];
        {
          CCC_PROF_START();
          rollout(alpha);
          CCC_PROF_ADD(PR_ROLLOUT);
#if defined(CCC_DDP_PROF) && defined(__HIP_DEVICE_COMPILE__)
          if((threadIdx.x & 63) == 0) mem.prof[PR_ROLLOUTS] += 1.0;
#endif
        }
        actual = mem.sc[SC_COST] - mem.sc[SC_COSTC];
        const double expected = -alpha * (mem.sc[SC_DV0] + alpha * mem.sc[SC_DV1]);
        const double ratio = expected > 0 ? actual / expected : (actual > 0 ? 1.0 : (actual < 0 ? -1.0 : 0.0));
        if(ratio > P.ratio_thre)
        {
          accepted = true;
          break;
        }
      }
      if(accepted)
      {
        decrease_lambda();
        phase([&](int lane) {
          for(int e = lane; e < (N + 1) * S; e += kWave) I.xs[e] = I.xc[e];
          for(int e = lane; e < N * M; e += kWave) I.us[e] = I.uc[e];
          if(lane == 0) mem.sc[SC_COST] = mem.sc[SC_COSTC];
        });
        if(actual < P.cost_thre)
        {
          status = 2;
          break;
        }
      }
      else
      {
        increase_lambda();
        if(mem.sc[SC_LAMBDA] > P.lambda_max)
        {
          status = -1;
          break;
        }
      }
    }
    if(iter > P.max_iter) iter = P.max_iter;
#if defined(CCC_DDP_PROF) && defined(__HIP_DEVICE_COMPILE__)
    phase([&](int lane) {
      if(lane == 0) mem.prof[PR_TOTAL] = (double)((long long)__builtin_readcyclecounter() - prof_begin_);
    });
    phase([&](int lane) {
      if(lane < 16) I.us[lane] = mem.prof[lane]; // a profiling build returns timings instead of a plan
    });
#endif
    phase([&](int lane) {
      if(lane == 0)
      {
        if(I.out_iters) *I.out_iters = iter;
        if(I.out_status) *I.out_status = status;
        if(I.out_cost) *I.out_cost = mem.sc[SC_COST];
      }
    });
  }
};
} // namespace CCC_DDP_NS
} // namespace ccc_amd
