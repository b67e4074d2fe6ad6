// ddp_tile.h -- control-limited DDP for CCC::DdpCentroidal (S = 9) and CCC::DdpSingleRigidBody (S = 12), M = 16 B ridges
// per step (B = 1, 2, 4 blocks of 16: one, two, up to four surface contacts): one instance per wavefront, every matrix
// DISTRIBUTED over the 64 lanes (round 3), the backward step in STRUCTURED form (round 4: no M x M object).
//
// Replaces (reference file:line under /root/reference):
//   src/DdpCentroidal.cpp:32-64, :66-83, :85-121, :123-177          problem callbacks (S = 9)
//   src/DdpSingleRigidBody.cpp:26-38, :52-91, :93-112, :114-243     problem callbacks (S = 12)
//   src/DdpCentroidal.cpp:229,233 / src/DdpSingleRigidBody.cpp:299,303   the external nmpc_ddp::DDPSolver::solve
// Algorithm: the one frozen in oracle/ddp.c (Tassa's control-limited DDP, reg_type 1 and 2, the warm-start guard).
// ARITHMETIC: the "tile" specification of oracle/ddp_tile.c -- every long sum a fixed tree or fma chain that maps onto the
// CDNA4 cross-lane paths, the backward step in structured form.  Kernel and oracle implement that specification
// independently and agree bit for bit.
//
// Lane = 16 g + c (g = row of the wavefront, c = lane in the row).  Layouts (b = 0 .. B-1 indexes the blocks of 16 ridges):
//   vectors over the M ridges        lane (g, c) holds v[c + 16 b] in v[b]        (replicated in the four rows)
//   state vectors                    lane (g, a), a < S, holds x[a]               (replicated)
//   per-ridge 6-vectors g_r          lane (g, c) holds the six non-zero entries of column c + 16 b of Fu
//   S x M matrices (K')              lane (g, c) holds rows a = g, g+4, g+8 of the columns c + 16 b
//   S x S and 6 x S, 6 x 6 matrices  LDS (Vxx, Fx, T1/Qxx; W, Y, D, E; C_f, M_f, M_f^-1)
// The backward step never forms an M x M object (round 4): Fu has six non-zero rows G, so Quu = w_force I + G' V6 G is
// the identity plus rank 6; the box-QP's products are two packed 16-lane DPP trees (row g of the wavefront reduces
// components g and 4 + (g & 1) of G y) and a 6 x 6 product; its "factorisation" is a 6 x 6 Gauss-Jordan inverse, one
// column per lane, pivot choices passed on by DPP row broadcasts; gains and value update are 6 x S algebra.  LDS is
// ~7.5 / 9 KB per wavefront (S = 9 / 12) at EVERY ridge stride.
// The line search runs FOUR step sizes at once, one per row (nmpc_ddp tries them in order and takes the first that is
// accepted; evaluating four side by side and taking the first accepted gives the same answer).
//
// Written against csrc/w64.h: the same source runs on the host with 64 lanes in lock step (tests/emu), which is how the
// CPU suite checks this kernel bit for bit against the oracle without a GPU.
#pragma once

#include "ddp_batch.h" // ddp_common::Params
#include "w64.h"

#include <type_traits>

#if defined(__clang__)
#  pragma clang fp contract(off)
#endif

namespace ccc_amd
{
namespace ddp_tile
{
// -DCCC_TILE_PROBE: the large pieces as separate functions, so that -Rpass-analysis=kernel-resource-usage reports
// their registers one by one (development aid)
#if defined(CCC_TILE_PROBE) && defined(__HIP_DEVICE_COMPILE__)
#  define CCC_TILE_PIECE __device__ __attribute__((noinline))
#else
#  define CCC_TILE_PIECE W64_FN
#endif
using namespace w64;
using ddp_common::Params;

// unroll factors of the loops whose full unrolling costs more registers than four wavefronts per SIMD leave (measured)
// (the Z loop: per model, see Solver::kUnrollZ)
#ifndef CCC_TILE_U_CF
#  define CCC_TILE_U_CF 8
#endif
#ifndef CCC_TILE_U_PAIR
#  define CCC_TILE_U_PAIR 4
#endif

// Section profiler (development aid: -DCCC_TILE_PROF, scripts/ddp_tile_sections.py): shader-clock cycles per section
// accumulated in LDS; solve_instance() then overwrites the first planned inputs with the totals, so a profiling build
// returns timings INSTEAD of a plan.
enum
{
  TP_DERIV = 0,
  TP_PRODUCTS,
  TP_QP_VALUE,
  TP_QP_GRAD,
  TP_QP_FACTOR,
  TP_QP_SOLVE,
  TP_QP_SEARCH,
  TP_GAINS,
  TP_VALUE,
  TP_FORWARD,
  TP_OTHER,
  TP_QP_CALLS,
  TP_QP_ITERS,
  TP_QP_FACTORS,
  TP_FORWARDS,
  TP_F_CF,
  TP_F_MF,
  TP_F_GJ,
  TP_F_UPD,   // cycles in rank-one updates (qp_update), and their number
  TP_F_UPDS,
  TP_FW_FETCH, // forward step (development aid): issue of the fetches, feedback + input store, running cost, terms, state
  TP_FW_FEED,
  TP_FW_COST,
  TP_FW_TERMS,
  TP_FW_STATE,
  TP_N
};
#if defined(CCC_TILE_PROF) && defined(__HIP_DEVICE_COMPILE__)
#  define TILE_PROF_START() long long prof_t_ = (long long)__builtin_readcyclecounter()
#  define TILE_PROF_ADD(k)                                                        \
    do                                                                            \
    {                                                                             \
      const long long prof_n_ = (long long)__builtin_readcyclecounter();          \
      if((threadIdx.x & 63) == 0) mem.prof[k] += (double)(prof_n_ - prof_t_);     \
      prof_t_ = prof_n_;                                                          \
    } while(0)
#  define TILE_PROF_COUNT(k)                              \
    do                                                    \
    {                                                     \
      if((threadIdx.x & 63) == 0) mem.prof[k] += 1.0;     \
    } while(0)
#else
#  define TILE_PROF_START() do {} while(0)
#  define TILE_PROF_ADD(k) do {} while(0)
#  define TILE_PROF_COUNT(k) do {} while(0)
#endif

constexpr int kSlots = 5;   // trajectory buffers: the current one + four line-search candidates
constexpr double kGravity = 9.80665; // include/CCC/Constants.h:10

// Per-instance problem data and workspace (global memory); M = 16 B is the ridge stride of every array
struct Instance
{
  const int * phase_dim;       // [P]
  const double * phase_vertex; // [P][M][3]
  const double * phase_ridge;  // [P][M][3]
  const int * step_phase;      // [N]
  const double * ref_pos;      // [N+1][3]
  const double * ref_ori;      // [N+1][3]  (SRB)
  const double * inertia;      // [9] or, in the IPP builds (Params::inertia_per_phase), [P][9]   (SRB)
  const double * x0;           // [S]
  const double * u_init;       // [N][M] or nullptr
  double * xbuf;               // [kSlots][N+1][S]
  double * ubuf;               // [kSlots][N][M]
  double * ks;                 // [N][M]
  double * Ks;                 // [N][M][S]
  double * u_out;              // [N][M]
  double * x_out;              // [N+1][S] or nullptr
  int * out_iters;
  int * out_status;
  double * out_cost;
};

template<int S, int B>
struct alignas(16) Mem
{
  static constexpr int M = 16 * B;
  static constexpr int LS = 16;        // row stride of the 6 x S matrices
  alignas(16) double Vxx[S * S];
  alignas(16) double Fx[S * S];        // [b][c]
  alignas(16) double T1[S * S];        // Vxx Fx, then Qxx
  alignas(16) double W[6 * LS];        // (Vxx Fx)[rows6, :]            [j * LS + a]
  alignas(16) double Y[6 * LS];        // M_f^-1 Wr
  alignas(16) double D[6 * LS];        // C_f Y
  alignas(16) double E[6 * LS];        // w_force Y - 2 W + V6 D
  static constexpr int LG = M + 1;     // row stride of the g_r table: odd, so that its six rows start in different banks
  alignas(16) double Gl[6 * LG];       // the six non-zero rows of Fu    [j * LG + r]
  alignas(16) double Cf[36], Mf[36], Minv[36];
  alignas(16) double u6[3][8];         // p, a, b of a rank-one update of Minv (qp_update)
  alignas(16) double cvr[6 * M];       // vertex (rows 0 .. 2) and ridge (3 .. 5) of the ridges of the cached contact phase
  alignas(16) double Vx[16], Qx[16];
  double wrun[16], wterm[16];
  double alpha[12];
  double inertia[9];
  double llt[6];                       // Cholesky factor of the inertia matrix (vllt3_factor)
  double rll[4];                       // 1 / l00, 1 / l11, 1 / l22 of that factor; 1 / mass (round 5: the solves multiply)
  static constexpr int kStepTable = 256;
  unsigned short stepinfo[kStepTable]; // (phase << 7) | ridges of the horizon's steps (round 5: see Solver::step_info)
  unsigned char pair[80];              // (a, b), a <= b, of the entries of Vxx's upper triangle: a | b << 4
  int warm_replaced;                   // 1: the warm-start guard replaced u_init (kept here, not in a register: it is
                                       //    written once per solve and read at its end)
#if defined(CCC_TILE_PROF)
  double prof[TP_N];
#endif
};

// Deterministic sin / cos on every lane (the restatement the oracle shares: Cody-Waite reduction by
// pi/2 + the fdlibm minimax kernels; <= 1 ulp for |x| < 1e3)
W64_FN void vsincos(vf x, vf & s, vf & c)
{
  const vf fn = vfloor(x * 6.36619772367581382433e-01 + 0.5);
  const vi n = to_int(fn);
  vf r = x - fn * 1.57079632673412561417e+00;
  r = r - fn * 6.07710050650619224932e-11;
  const vf z = r * r;
  const vf ps = -1.66666666666666324348e-01
                + z * (8.33333333332248946124e-03
                       + z * (-1.98412698298579493134e-04
                              + z * (2.75573137070700676789e-06
                                     + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const vf pc = 4.16666666666666019037e-02
                + z * (-1.38888888888741095749e-03
                       + z * (2.48015872894767294178e-05
                              + z * (-2.75573143513906633035e-07
                                     + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const vf sn = r + r * z * ps;
  const vf cs = 1.0 - (0.5 * z - z * z * pc);
  const vi q = n & 3;
  s = sel(q == 0, sn, sel(q == 1, cs, sel(q == 2, -sn, -cs)));
  c = sel(q == 0, cs, sel(q == 1, -sn, sel(q == 2, -cs, sn)));
}

// Eigen::LLT<Matrix3d>::solve on every lane (src/DdpSingleRigidBody.cpp:88,122-123): the statements of oracle/ddp_models.c.
// The factor of the inertia matrix is the same for every solve of an instance: vllt3_factor() forms it once (init()) with
// the statements the oracle forms it with at every call -- the same six numbers, so every solve is bit for bit the same
W64_FN void vllt3_factor(const double * I, double * L)
{
  const double l00 = std::sqrt(I[0]);
  const double l10 = I[3] / l00, l20 = I[6] / l00;
  const double l11 = std::sqrt(I[4] - l10 * l10);
  const double l21 = (I[7] - l20 * l10) / l11;
  const double l22 = std::sqrt(I[8] - l20 * l20 - l21 * l21);
  L[0] = l00;
  L[1] = l10;
  L[2] = l20;
  L[3] = l11;
  L[4] = l21;
  L[5] = l22;
}
// (round 5: R = the reciprocals of the factor's diagonal, formed once per instance -- the substitutions multiply; SPEC:
//  oracle/ddp_tile.c llt3_solve)
W64_FN void vllt3(const double * L, const double * R, const vf (&b)[3], vf (&x)[3])
{
  const double l10 = L[1], l20 = L[2], l21 = L[4];
  const double r00 = R[0], r11 = R[1], r22 = R[2];
  const vf y0 = b[0] * r00;
  const vf y1 = (b[1] - l10 * y0) * r11;
  const vf y2 = (b[2] - l20 * y0 - l21 * y1) * r22;
  x[2] = y2 * r22;
  x[1] = (y1 - l21 * x[2]) * r11;
  x[0] = (y0 - l10 * x[1] - l20 * x[2]) * r00;
}

// IPP (single-rigid-body model only): MotionParam::inertia_mat per contact phase (Params::inertia_per_phase) -- a build of
// its own, so that the register allocation of the one-matrix-per-instance kernels (config 5) is what it was
template<int S, int B, bool IPP = false>
struct Solver
{
  static_assert(!IPP || S == 12, "the centroidal model has no inertia matrix");
  static_assert(B == 1 || B == 2 || B == 4, "16, 32 or 64 ridges per step");
  static constexpr int M = 16 * B;
  static constexpr int LS = Mem<S, B>::LS;
  static constexpr int LG = Mem<S, B>::LG;
  static constexpr int FU0 = (S == 9) ? 3 : 6; // first non-zero row of Fu (its six non-zero rows are FU0 .. FU0+5)
  static constexpr int NP = S * (S + 1) / 2;   // entries of the upper triangle of Vxx
  static constexpr int NPASS = (NP + 63) / 64;
  // unroll factor of the Z = Quu K + 2 Qux loop (measured per model, round 3: S = 9: 41.0 k solves/s at 4, 34.7 k at 1;
  // S = 12: 133 k at 2, 126 k at 4)
#ifdef CCC_TILE_U_Z
  static constexpr int kUnrollZ = CCC_TILE_U_Z;
#else
  static constexpr int kUnrollZ = (S == 9) ? 4 : 2;
#endif
  // oracle/ddp_tile.c FX_NZ9 / FX_NZ12: the rows of Fx's column c that enter the products, kNzLen per column; entry k of
  // every column as sixteen 4-bit values (column c in bits 4c .. 4c+3)
  static constexpr int kNzLen = S == 9 ? 3 : 6;
  static W64_FN constexpr unsigned long long fx_nz_table(int k)
  {
    constexpr unsigned char t9[9][3] = {{0, 7, 8}, {1, 6, 8}, {2, 6, 7}, {0, 3, 6}, {1, 4, 6}, {2, 5, 6}, {6, 0, 1}, {7, 0, 1}, {8, 0, 1}};
    constexpr unsigned char t12[12][6] = {{0, 9, 10, 11, 3, 4}, {1, 9, 10, 11, 3, 4}, {2, 9, 10, 11, 3, 4}, {3, 4, 5, 0, 1, 2},
                                          {3, 4, 5, 0, 1, 2},   {5, 0, 1, 2, 3, 4},   {0, 6, 3, 4, 5, 9},   {1, 7, 3, 4, 5, 9},
                                          {2, 8, 3, 4, 5, 9},   {3, 4, 5, 9, 10, 11}, {3, 4, 5, 9, 10, 11}, {3, 4, 5, 9, 10, 11}};
    unsigned long long r = 0;
    for(int cc = 0; cc < S; cc++) r |= static_cast<unsigned long long>(S == 9 ? t9[cc][k % 3] : t12[cc % 12][k]) << (4 * cc);
    return r;
  }
  // unroll factor of the S-term products T1 = Vxx Fx, Qxx = Fx' T1 (measured, round 4, 16 ridges at 256 VGPRs: fully
  // unrolled + 1.5 % for 100-200 B of scratch -- not taken; fully unrolled at 128 VGPRs: - 25 %)
#ifdef CCC_TILE_U_PROD
  static constexpr int kUnrollProd = CCC_TILE_U_PROD;
#else
  static constexpr int kUnrollProd = 3;
#endif
  // one bit per ridge: clamped-or-unused sets
  using mask_t = std::conditional_t<(B <= 2), unsigned, unsigned long long>;
  static constexpr mask_t kAll = (B == 1) ? static_cast<mask_t>(0xffffu) : static_cast<mask_t>(~static_cast<mask_t>(0));

  const Params & P;
  const Instance & I;
  Mem<S, B> & mem;

  // masked loads: branch-free for the 9-state model, predicated for the 12-state one (measured, see w64::ld_if_branch)
  static W64_FN vf ldm(const double * p, vi idx, vb m) { return S == 12 ? ld_if_branch(p, idx, m) : ld_if(p, idx, m); }

  // lane coordinates
  vi lane, c, g;
  vi j6;             // min(c, 5): the row of the 6 x 6 matrices this lane works on (the results are taken from lanes 0 .. 5)
  vb inS;            // c < S
  vi arow[3];        // rows of the S x M matrices this lane holds: g, g + 4, g + 8 (clamped to S - 1 when not valid)
  vb aval[3];

  // solver state (wave-uniform scalars)
  double lambda, dlambda, cost, dV0, dV1;
  int cur;           // slot of the current trajectory
  // vertex and ridge of the ridges c + 16 b in the contact phase `ph_cached` (zero beyond its dimension): reloaded only
  // when a step is in another phase than the one before -- a horizon has a handful of phases
  int ph_cached;
  // (round 5: in LDS, mem.cvr, not in registers -- twelve VGPRs per block of ridges that were live through the whole solve)
  W64_FN vf Vc(int b, int k) const { return ld(mem.cvr, k * M + c + 16 * b); }
  W64_FN vf Rc(int b, int k) const { return ld(mem.cvr, (3 + k) * M + c + 16 * b); }

  W64_FN Solver(const Params & p, const Instance & i, Mem<S, B> & m) : P(p), I(i), mem(m) {}

  // ------------------------------------------------------------------------------------------------ ridge vectors
  // SPEC (sums over the M ridges): w_c = v_c                              (B = 1)
  //                                w_c = v_c + v_{c+16}                   (B = 2)
  //                                w_c = (v_c + v_{c+16}) + (v_{c+32} + v_{c+48})   (B = 4), then tree16(w)
  template<int AB>
  static W64_FN vf bsum(const vf (&v)[B])
  {
    if constexpr(AB == 1)
      return v[0];
    else if constexpr(AB == 2)
      return v[0] + v[1];
    else
      return (v[0] + v[1]) + (v[2] + v[3]);
  }
  template<int AB>
  static W64_FN vf sumM(const vf (&v)[B]) { return sum16(bsum<AB>(v)); }
  // the 16 bits of block b of a set
  static W64_FN unsigned piece(mask_t m, int b) { return static_cast<unsigned>(m >> (16 * b)) & 0xffffu; }
  // lane (g, c): is ridge c + 16 b / column 16 b + 4 g + s in the set ?
  W64_FN vb row_in(mask_t m, int b) const { return ((spl(static_cast<int>(piece(m, b))) >> c) & 1) != 0; }
  W64_FN vb col_in(mask_t m, int b, int s) const { return ((spl(static_cast<int>(piece(m, b))) >> (g * 4 + s)) & 1) != 0; }
  template<int AB>
  W64_FN mask_t ballotM(const vb (&p)[B]) const
  {
    mask_t r = 0;
    for(int b = 0; b < AB; b++) r |= static_cast<mask_t>(ballot(p[b]) & 0xffffull) << (16 * b);
    return r;
  }

  // ------------------------------------------------------------------------------------------------ set-up
  W64_FN void init()
  {
    lane = lane_id();
    c = lane & 15;
    g = lane >> 4;
    j6 = seli(c < 6, c, spl(5));
    inS = c < S;
    for(int t = 0; t < 3; t++)
    {
      const vi a = g + 4 * t;
      aval[t] = a < S;
      arow[t] = seli(aval[t], a, spl(S - 1));
    }
    // pair index -> (a, b), row-major over the upper triangle: (0,0) (0,1) .. (0,S-1) (1,1) ..
    {
      int q = 0;
      for(int a = 0; a < S; a++)
        for(int b = a; b < S; b++) mem.pair[q++] = static_cast<unsigned char>(a | (b << 4));
      for(; q < 80; q++) mem.pair[q] = 0;
    }
    // small tables -> LDS (per-lane indexed reads of kernel arguments would go through scratch)
    for(int e = 0; e < 16; e++)
    {
      mem.wrun[e] = e < S ? P.w_run[e] : 0.0;
      mem.wterm[e] = e < S ? P.w_term[e] : 0.0;
    }
    for(int e = 0; e < 12; e++) mem.alpha[e] = e < 11 ? P.alpha[e] : 0.0;
    // (one inertia matrix per instance: here; one per contact phase -- IPP -- in contact_of())
    for(int e = 0; e < 9; e++) mem.inertia[e] = (S == 12 && !IPP) ? I.inertia[e] : 0.0;
    for(int e = 0; e < 3; e++) mem.rll[e] = 0.0;
    if(S == 12 && !IPP)
    {
      double lf[6];
      vllt3_factor(I.inertia, lf);
      for(int e = 0; e < 6; e++) mem.llt[e] = lf[e];
      mem.rll[0] = 1.0 / lf[0];
      mem.rll[1] = 1.0 / lf[3];
      mem.rll[2] = 1.0 / lf[5];
    }
    mem.rll[3] = 1.0 / P.mass;
    // the steps' phases and ridge counts (step_info): lane l looks up the steps l, l + 64, ...
    for(int e = 0; e < P.N && e < Mem<S, B>::kStepTable; e += 64)
    {
      const vb ok = (lane + e < P.N) && (lane + e < Mem<S, B>::kStepTable);
      const vi st_ = seli(ok, lane + e, spl(0));
      vi ph = ldi(I.step_phase, st_);
      ph = seli(ph < 0, spl(0), seli(ph >= P.P, spl(P.P - 1), ph));
      vi dm = ldi(I.phase_dim, ph);
      dm = seli(dm < 0, spl(0), seli(dm > M, spl(M), dm));
      sth(mem.stepinfo, st_, (ph << 7) | dm, ok);
    }
#if defined(CCC_TILE_PROF)
    for(int e = 0; e < TP_N; e++) mem.prof[e] = 0.0;
#endif
    ph_cached = -1;
    wave_sync();
  }

  // Phase and ridge count of a step.  Round 5: from an LDS table filled once per solve (init()) -- the two look-ups are
  // DEPENDENT scalar loads from global memory (step_phase[step], then phase_dim[phase]), and the operand fetches of the
  // forward and backward passes cannot be issued before both have returned: measured, 1.4-1.9 k of the 3.3-4.0 k cycles of a
  // forward step on a lone wavefront were the issue of its prefetch.  Horizons beyond the table take the loads.
  W64_FN void step_info(int step, int & ph, int & m) const
  {
    if(step < Mem<S, B>::kStepTable && P.P <= 512)
    {
      const int pm = uniform_i(static_cast<int>(mem.stepinfo[step]));
      ph = pm >> 7;
      m = pm & 0x7f;
    }
    else
    {
      ph = phase_of(step);
      m = dim_of_phase(ph);
    }
  }
  W64_FN int phase_of(int step) const
  {
    const int p = I.step_phase[step];
    return p < 0 ? 0 : (p >= P.P ? P.P - 1 : p);
  }
  W64_FN int dim_of_phase(int ph) const
  {
    const int d = I.phase_dim[ph];
    return d < 0 ? 0 : (d > M ? M : d);
  }
  // MotionParam::inertia_mat number k of the instance, its Cholesky factor and the reciprocals of the factor's diagonal ->
  // LDS.  The reference reads motion_param_func_(t).inertia_mat at every step (src/DdpSingleRigidBody.cpp:56-57 in
  // stateEq, :120-123 in calcStateEqDeriv, where it also factorises it): with IPP a contact phase IS a distinct MotionParam
  // -- contact list and inertia matrix -- and the matrix is cached with the phase's contact vectors.  The factor's
  // statements are vllt3_factor's whichever way the matrix is indexed, so a phase's six numbers are the ones the oracle
  // forms at each of its steps.
  W64_FN void inertia_of(int k)
  {
    const double * In = I.inertia + static_cast<long>(k) * 9;
    double lf[6];
    vllt3_factor(In, lf);
    for(int e = 0; e < 9; e++) mem.inertia[e] = In[e];
    for(int e = 0; e < 6; e++) mem.llt[e] = lf[e];
    mem.rll[0] = 1.0 / lf[0];
    mem.rll[1] = 1.0 / lf[3];
    mem.rll[2] = 1.0 / lf[5];
  }
  // the step's contact phase into Vc / Rc
  template<int AB>
  W64_FN void contact_of(int ph, int dim)
  {
    if(ph == ph_cached) return;
    ph_cached = ph;
    if(IPP) inertia_of(ph);
    const long base = static_cast<long>(ph) * M * 3;
    for(int b = 0; b < AB; b++)
    {
      const vi r = c + 16 * b;
      const vb in = r < dim;
      for(int k = 0; k < 3; k++)
      {
        st(mem.cvr, k * M + r, ldm(I.phase_vertex + base, r * 3 + k, in), g == 0);
        st(mem.cvr, (3 + k) * M + r, ldm(I.phase_ridge + base, r * 3 + k, in), g == 0);
      }
    }
    wave_sync();
  }
  // reference of the weighted state entries (Cen: [pos, 0, 0]; SRB: [pos, ori, 0, 0]) on the lanes a < S
  W64_FN vf ref_of(int step) const
  {
    vf r = ldm(I.ref_pos + static_cast<long>(step) * 3, c, c < 3);
    if(S == 12) r = sel(c >= 3 && c < 6, ldm(I.ref_ori + static_cast<long>(step) * 3, c - 3, c >= 3 && c < 6), r);
    return r;
  }

  // ------------------------------------------------------------------------------------------------ the model
  // Everything a step of the model needs from (x, u) that is shared between stateEq and its derivatives.
  struct Terms
  {
    vf cr[B][3];          // (vertex - pos) x ridge of the ridges c + 16 b
    vf force[3];          // sum_r u_r ridge_r                          (sumM)
    vf moment[3];         // sum_r u_r (vertex_r - pos) x ridge_r       (sumM)
    vf accel[3];          // sum_r (u_r ridge_r) / m                    (sumM; single-rigid-body model)
  };
  // x: state on the lanes a < S of every row (rows may differ: the line search), u: the force scales of the ridges c + 16 b
  template<int AB>
  CCC_TILE_PIECE void terms_of(int ph, int dim, vf x, const vf (&u)[B], Terms & T)
  {
    contact_of<AB>(ph, dim);
    const vf p0 = row_bcast<0>(x), p1 = row_bcast<1>(x), p2 = row_bcast<2>(x);
    for(int b = 0; b < AB; b++)
    {
      const vf d0 = Vc(b, 0) - p0, d1 = Vc(b, 1) - p1, d2 = Vc(b, 2) - p2;
      T.cr[b][0] = d1 * Rc(b, 2) - d2 * Rc(b, 1);
      T.cr[b][1] = d2 * Rc(b, 0) - d0 * Rc(b, 2);
      T.cr[b][2] = d0 * Rc(b, 1) - d1 * Rc(b, 0);
    }
    for(int k = 0; k < 3; k++)
    {
      vf t[B];
      for(int b = 0; b < AB; b++) t[b] = u[b] * Rc(b, k);
      T.force[k] = sumM<AB>(t);
      for(int b = 0; b < AB; b++) t[b] = u[b] * T.cr[b][k];
      T.moment[k] = sumM<AB>(t);
      if(S == 12)
      {
        for(int b = 0; b < AB; b++) t[b] = (u[b] * Rc(b, k)) * mem.rll[3];
        T.accel[k] = sumM<AB>(t);
      }
    }
  }
  // x_next = stateEq(step, x, u): src/DdpCentroidal.cpp:32-64 / src/DdpSingleRigidBody.cpp:52-91
  CCC_TILE_PIECE vf state_eq(const Terms & T, vf x) const
  {
    vf xd;
    if(S == 9)
    {
      // pos' = P / m, P' = -m g e_z + force, L' = moment
      const vf shifted = sel(c == 0, row_bcast<3>(x), sel(c == 1, row_bcast<4>(x), row_bcast<5>(x)));
      const vf fz = -1 * P.mass * kGravity + T.force[2];
      xd = sel(c < 3, shifted * mem.rll[3],
               sel(c == 3, T.force[0], sel(c == 4, T.force[1], sel(c == 5, fz,
               sel(c == 6, T.moment[0], sel(c == 7, T.moment[1], T.moment[2]))))));
    }
    else
    {
      const vf w0 = row_bcast<9>(x), w1 = row_bcast<10>(x), w2 = row_bcast<11>(x);
      vf sa, ca, sb, cb;
      vsincos(row_bcast<3>(x), sa, ca);
      vsincos(row_bcast<4>(x), sb, cb);
      // matAngularVelToEulerDot(ori) * angular_vel, src/DdpSingleRigidBody.cpp:26-38,72
      const vf rcb = 1.0 / cb; // (round 5: one reciprocal, the divisions by cos(beta) multiply; SPEC: oracle/ddp_tile.c state_eq)
      const vf e0 = ((ca * sb) * rcb) * w0 + ((sb * sa) * rcb) * w1 + 1.0 * w2;
      const vf e1 = (-1 * sa) * w0 + ca * w1 + 0.0 * w2;
      const vf e2 = (ca * rcb) * w0 + (sa * rcb) * w1 + 0.0 * w2;
      const double * In = mem.inertia;
      const vf Iw0 = In[0] * w0 + In[1] * w1 + In[2] * w2;
      const vf Iw1 = In[3] * w0 + In[4] * w1 + In[5] * w2;
      const vf Iw2 = In[6] * w0 + In[7] * w1 + In[8] * w2;
      const vf cw0 = w1 * Iw2 - w2 * Iw1, cw1 = w2 * Iw0 - w0 * Iw2, cw2 = w0 * Iw1 - w1 * Iw0;
      const vf wd[3] = {-1 * cw0 + T.moment[0], -1 * cw1 + T.moment[1], -1 * cw2 + T.moment[2]};
      vf sol[3];
      vllt3(mem.llt, mem.rll, wd, sol);
      const vf az = -1 * kGravity + T.accel[2];
      const vf vshift = sel(c == 0, row_bcast<6>(x), sel(c == 1, row_bcast<7>(x), row_bcast<8>(x)));
      xd = sel(c < 3, vshift,
               sel(c == 3, e0, sel(c == 4, e1, sel(c == 5, e2,
               sel(c == 6, T.accel[0], sel(c == 7, T.accel[1], sel(c == 8, az,
               sel(c == 9, sol[0], sel(c == 10, sol[1], sol[2])))))))));
    }
    return sel(inS, x + P.dt * xd, 0.0);
  }
  // running / terminal cost of (x, u) per row: src/DdpCentroidal.cpp:66-83
  // (ref = ref_of(step): fetched ahead by the callers with the step's other operands -- round 5: asked for here, the load was
  //  consumed on the spot, a trip to L2 exposed in every step of every rollout)
  template<int AB>
  W64_FN vf running_cost(vf ref, vf x, const vf (&u)[B]) const
  {
    const vf e = x - ref;
    const vf cx = sum16(sel(inS, 0.5 * ld(mem.wrun, c) * e * e, 0.0));
    vf t[B];
    for(int b = 0; b < AB; b++) t[b] = u[b] * u[b];
    const vf un = sumM<AB>(t);
    return cx + 0.5 * P.w_force * un;
  }
  W64_FN vf terminal_cost(vf x) const
  {
    const vf e = x - ref_of(P.N);
    return sum16(sel(inS, 0.5 * ld(mem.wterm, c) * e * e, 0.0));
  }

  // Fx -> mem.Fx ([b][c], dense), Fu: the six non-zero rows of the columns c + 16 b in registers
  // (src/DdpCentroidal.cpp:85-121 / src/DdpSingleRigidBody.cpp:114-185), at (x, u) with the step's Terms
  template<int AB>
  CCC_TILE_PIECE void state_eq_deriv(const Terms & T, vf x, vf (&Fu)[B][6])
  {
    const vb first = lane == 0;
    for(int e = 0; e < S * S; e += 64) st(mem.Fx, lane + e, splat(0.0), lane + e < S * S);
    wave_sync();
    const double dt = P.dt;
    if(S == 9)
    {
      for(int b = 0; b < AB; b++)
        for(int k = 0; k < 3; k++)
        {
          Fu[b][k] = Rc(b, k) * dt;
          Fu[b][3 + k] = T.cr[b][k] * dt;
        }
      // (the scalars are the same on every lane: lane 0 stores them)
      const vf tf0 = T.force[0], tf1 = T.force[1], tf2 = T.force[2];
      for(int a = 0; a < 3; a++) st(mem.Fx, spl(a * S + 3 + a), splat((1 / P.mass) * dt), first);
      st(mem.Fx, spl(6 * S + 1), (-tf2) * dt, first);
      st(mem.Fx, spl(6 * S + 2), tf1 * dt, first);
      st(mem.Fx, spl(7 * S + 0), tf2 * dt, first);
      st(mem.Fx, spl(7 * S + 2), (-tf0) * dt, first);
      st(mem.Fx, spl(8 * S + 0), (-tf1) * dt, first);
      st(mem.Fx, spl(8 * S + 1), tf0 * dt, first);
    }
    else
    {
      const double * In = mem.inertia;
      for(int b = 0; b < AB; b++)
      {
        vf sol[3];
        vllt3(mem.llt, mem.rll, T.cr[b], sol);
        for(int k = 0; k < 3; k++)
        {
          Fu[b][k] = (Rc(b, k) * mem.rll[3]) * dt;
          Fu[b][3 + k] = sol[k] * dt;
        }
      }
      const vf w1 = row_bcast<9>(x), w2 = row_bcast<10>(x), w3 = row_bcast<11>(x);
      const double I11 = In[0], I12 = In[1], I13 = In[2], I22 = In[4], I23 = In[5], I33 = In[8];
      const vf D[9] = {I12 * w3 - I13 * w2,
                       -I13 * w1 + I22 * w3 - 2 * I23 * w2 - I33 * w3,
                       I12 * w1 + I22 * w2 + 2 * I23 * w3 - I33 * w2,
                       -I11 * w3 + 2 * I13 * w1 + I23 * w2 + I33 * w3,
                       -I12 * w3 + I23 * w1,
                       -I11 * w1 - I12 * w2 - 2 * I13 * w3 + I33 * w1,
                       I11 * w2 - 2 * I12 * w1 - I22 * w2 - I23 * w3,
                       I11 * w1 + 2 * I12 * w2 + I13 * w3 - I22 * w1,
                       I13 * w2 - I23 * w1};
      const vf tf0 = T.force[0], tf1 = T.force[1], tf2 = T.force[2];
      const vf zero = splat(0.0);
      const vf CM[9] = {zero, -tf2, tf1, tf2, zero, -tf0, -tf1, tf0, zero};
      for(int b = 0; b < 3; b++)
      {
        // column b of I^-1 d(-w x I w)/dw -> block (9, 9); column b of I^-1 crossMat(totalForce) -> block (9, 0)
        const vf colD[3] = {D[b], D[3 + b], D[6 + b]}, colC[3] = {CM[b], CM[3 + b], CM[6 + b]};
        vf sD[3], sC[3];
        vllt3(mem.llt, mem.rll, colD, sD);
        vllt3(mem.llt, mem.rll, colC, sC);
        for(int a = 0; a < 3; a++)
        {
          st(mem.Fx, spl((9 + a) * S + 9 + b), sD[a] * dt, first);
          st(mem.Fx, spl((9 + a) * S + b), sC[a] * dt, first);
        }
      }
      vf sa, ca, sb, cb;
      vsincos(row_bcast<3>(x), sa, ca);
      vsincos(row_bcast<4>(x), sb, cb);
      const vf sb2 = sb * sb, rcb = 1.0 / cb, rcb2 = rcb * rcb; // (round 5: SPEC oracle/ddp_tile.c state_eq_deriv)
      for(int a = 0; a < 3; a++) st(mem.Fx, spl(a * S + 6 + a), splat(1.0 * dt), first);
      const vf K[9] = {(ca * sb) * rcb, (sb * sa) * rcb, splat(1.0), -1 * sa, ca, zero, ca * rcb, sa * rcb, zero};
      for(int a = 0; a < 3; a++)
        for(int b = 0; b < 3; b++) st(mem.Fx, spl((3 + a) * S + 9 + b), K[a * 3 + b] * dt, first);
      st(mem.Fx, spl(3 * S + 3), (-w1 * sa * sb * rcb + w2 * sb * ca * rcb) * dt, first);
      st(mem.Fx, spl(4 * S + 3), (-w1 * ca - w2 * sa) * dt, first);
      st(mem.Fx, spl(5 * S + 3), (-w1 * sa * rcb + w2 * ca * rcb) * dt, first);
      st(mem.Fx, spl(3 * S + 4), (w1 * sb2 * ca * rcb2 + w1 * ca + w2 * sa * sb2 * rcb2 + w2 * sa) * dt, first);
      st(mem.Fx, spl(5 * S + 4), (w1 * sb * ca * rcb2 + w2 * sa * sb * rcb2) * dt, first);
    }
    wave_sync();
    // diagonal: entry * dt + 1
    st(mem.Fx, c * S + c, ld(mem.Fx, seli(inS, c * S + c, spl(0))) + 1.0, inS && (g == 0));
    wave_sync();
  }

  // ------------------------------------------------------------------------------------------------ structured box-QP
  // oracle/ddp_tile.c (round 4): no M x M object is formed.  With G = the six non-zero rows of Fu, V6r = Vxx[rows6,
  // rows6] (+ lambda I for reg_type 2), P = V6r G and alpha = w_force (+ lambda for reg_type 1):
  //   Quu_F = alpha I + G' V6r G = alpha I + P' G,   Quu_F,ff^-1 b = (b - G_f' M_f^-1 (P_f b)) / alpha,
  //   M_f = alpha I + V6r C_f,  C_f = sum_{r free} g_r g_r'  (6 x 6).
  // Per lane (ridge c + 16 b): g_r and p_r in registers (six doubles each); the sums over the ridges of the six
  // components run as TWO trees: row g of the wavefront reduces component g (tree A) and component 4 + (g & 1) (tree B).
  struct Qp
  {
    int m;
    double alpha, inv_alpha, ratio, lv; // ratio = w_force / alpha
    vf G[B][6];
    vf Ga[B], Gb[B];                    // the rows' own components of g_r: [g] and [4 + (g & 1)]
    vf u[B];                            // the step's nominal inputs (Qu = w_force u + G' Vx6)
    vf v6r[6];                          // lane j < 6: row j of V6r = Vxx[rows6, rows6] (+ lv on the diagonal)
    vb in[B];
  };
  static W64_FN vf pick4(vi g, const vf (&v)[6]) { return sel(g == 0, v[0], sel(g == 1, v[1], sel(g == 2, v[2], v[3]))); }
  // the six ridge sums sum_r G[j][r] y_r.  SPEC: t_r = G[j][r] y_r; treeM
  template<int AB>
  W64_FN void six_sums(const Qp & Q, const vf (&y)[B], double (&out)[6]) const
  {
    vf t[B];
    for(int b = 0; b < AB; b++) t[b] = Q.Ga[b] * y[b];
    const vf sa = sumM<AB>(t);
    for(int b = 0; b < AB; b++) t[b] = Q.Gb[b] * y[b];
    const vf sb = sumM<AB>(t);
    out[0] = read_lane(sa, 0);
    out[1] = read_lane(sa, 16);
    out[2] = read_lane(sa, 32);
    out[3] = read_lane(sa, 48);
    out[4] = read_lane(sb, 0);
    out[5] = read_lane(sb, 16);
  }
  // row j (on lane j < 6) of a 6 x 6 matrix in LDS: elem(l) = A[j][l].  SPEC (apply6): out_j = A[j][0] v_0;
  // fma(A[j][l], v_l, .), l = 1 .. 5 -- every lane computes "its" row, the six results come back as scalars
  template<class E>
  W64_FN void apply6(E elem, const double (&v)[6], double (&out)[6]) const
  {
    vf s = elem(0) * v[0];
    for(int l = 1; l < 6; l++) s = vfma(elem(l), splat(v[l]), s);
    for(int j = 0; j < 6; j++) out[j] = read_lane(s, j);
  }
  // V6r[j6][l] = Vxx[FU0 + j6][FU0 + l] (+ lv on the diagonal)
  W64_FN vf v6r_elem(int l, double lv) const
  {
    const vf v = ld(mem.Vxx, (j6 + FU0) * S + FU0 + l);
    return sel(j6 == l, v + lv, v);
  }
  // hy = Quu_F y.  SPEC: gy = G y (six sums); vy = V6r gy (apply6); hy_r = alpha y_r; hy_r = fma(G[j][r], vy_j, hy_r)
  template<int AB>
  W64_FN void qp_matvec(const Qp & Q, const vf (&y)[B], vf (&hy)[B]) const
  {
    double gy[6], vy[6];
    six_sums<AB>(Q, y, gy);
    apply6([&](int l) { return Q.v6r[l]; }, gy, vy);
    for(int b = 0; b < AB; b++)
    {
      vf s = Q.alpha * y[b];
      for(int j = 0; j < 6; j++) s = vfma(Q.G[b][j], splat(vy[j]), s);
      hy[b] = s;
    }
  }
  // C_f, M_f = alpha I + V6r C_f, Minv = M_f^-1 by Gauss-Jordan elimination with partial pivoting -> mem.Cf, mem.Minv;
  // false when the inverse is not finite.  SPEC: oracle/ddp_tile.c s_factor
  template<int AB>
  CCC_TILE_PIECE bool qp_factor(const Qp & Q, mask_t freemask)
  {
    // entry n = 6 j + l of the 6 x 6 matrices on the lanes n < 36
    const vb live = lane < 36;
    const vi n = seli(live, lane, spl(0));
    const vi j = (n * 43) >> 8, l = n - 6 * j;
    TILE_PROF_START();
    {
      // (branch-free: the LDS reads of all ridges are in flight together; a ridge that is not free leaves acc alone)
      vf acc = splat(0.0);
      W64_UNROLL(CCC_TILE_U_CF)
      for(int r = 0; r < 16 * AB; r++)
      {
        const vf nx = vfma(ld(mem.Gl, j * LG + r), ld(mem.Gl, l * LG + r), acc);
        acc = ((freemask >> r) & 1u) ? nx : acc;
      }
      st(mem.Cf, n, acc, live);
    }
    wave_sync();
    TILE_PROF_ADD(TP_F_CF);
    {
      vf acc = sel(j == l, splat(Q.alpha), 0.0);
      for(int t = 0; t < 6; t++)
      {
        const vf v6 = ld(mem.Vxx, (j + FU0) * S + FU0 + t);
        const vf v6r = sel(j == t, v6 + Q.lv, v6);
        acc = vfma(v6r, ld(mem.Cf, l + 6 * t), acc);
      }
      st(mem.Mf, n, acc, live);
    }
    wave_sync();
    TILE_PROF_ADD(TP_F_MF);
    // column `lane` of [M_f | I] in registers (lanes 0 .. 11; the others carry zeros along)
    vf a[6];
    {
      const vb mcol = lane < 6, icol = lane >= 6 && lane < 12;
      const vi col = seli(mcol, lane, spl(0));
      for(int i = 0; i < 6; i++) a[i] = sel(mcol, ld(mem.Mf, col + 6 * i), sel(icol && (lane == 6 + i), 1.0, 0.0));
    }
    gj_step<0>(a);
    // failure = an entry of the inverse that is not finite (a zero pivot leaves inf / NaN behind)
    vb bad = lane < 0;
    for(int i = 0; i < 6; i++)
    {
      st(mem.Minv, seli(lane >= 6 && lane < 12, lane - 6 + 6 * i, spl(0)), a[i], lane >= 6 && lane < 12);
      bad = bad || !(vabs(a[i]) <= 1.7976931348623157e308);
    }
    const bool ok = (ballot(bad) & 0xfc0ull) == 0ull;
    wave_sync();
    TILE_PROF_ADD(TP_F_GJ);
    return ok;
  }
  // Every lane works on its own column; the pivot column's choices (pivot row, reciprocal, multipliers) reach the
  // others by DPP row broadcasts -- the twelve live columns sit in the first row of the wavefront -- so a pivot is a
  // straight run of vector instructions: no scalar registers, no branches.
  template<int K>
  W64_FN void gj_step(vf (&a)[6])
  {
    if constexpr(K < 6)
    {
      vi pl = spl(K);
      vf best = vabs(a[K]);
      for(int i = K + 1; i < 6; i++)
      {
        const vf ai = vabs(a[i]);
        const vb gt = ai > best;
        best = sel(gt, ai, best);
        pl = seli(gt, spl(i), pl);
      }
      const vi p = row_bcast_i<K>(pl);
      // rows p and K change places
      {
        vf ak = a[K];
        for(int i = K + 1; i < 6; i++)
        {
          const vb is = p == i;
          const vf ai = a[i];
          a[i] = sel(is, a[K], ai);
          ak = sel(is, ai, ak);
        }
        a[K] = ak;
      }
      const vf rpl = 1.0 / a[K];
      vf mb[6];
      for(int i = 0; i < 6; i++)
        if(i != K) mb[i] = row_bcast<K>(a[i] * rpl);
      const vf rp = row_bcast<K>(rpl);
      const vf akj = a[K];
      for(int i = 0; i < 6; i++)
        if(i != K) a[i] = vfma(-mb[i], akj, a[i]);
      a[K] = akj * rp;
      gj_step<K + 1>(a);
    }
  }
  // Round 5 -- the free set changes by ONE ridge r (sigma = +1: it becomes free, -1: clamped): C_f += sigma g g',
  // M_f += sigma p g', p = V6r g, so Minv -= (sigma / den) (Minv p)(g' Minv), den = 1 + sigma g' Minv p (Sherman-Morrison):
  // a dozen 6-term sums and one division instead of C_f's 16 B-term sums and a Gauss-Jordan inverse.  false = declined
  // (den small or not finite, or an updated entry not finite): the caller factorises afresh.  SPEC: oracle/ddp_tile.c
  // s_update, statement by statement: p and a by apply6 (row j on lane j), b with column l on lane l, the three vectors
  // handed to the 36 entry lanes through LDS.
  static constexpr int kUpdateKmax = 4;   // S_UPDATE_KMAX (the default of Params::update_kmax)
  static constexpr int kUpdateUmax = 12;  // S_UPDATE_UMAX
  W64_FN bool qp_update(const Qp & Q, int r, double sigma)
  {
    const vb row6 = lane < 6;
    // (g_t is re-read where it is used -- broadcast LDS reads -- rather than held: the box-QP is where the kernel's
    //  register pressure peaks)
    auto gt = [&](int t) { return ld(mem.Gl, spl(t * LG + r)); };
    {
      vf ps = Q.v6r[0] * gt(0);
      for(int l = 1; l < 6; l++) ps = vfma(Q.v6r[l], gt(l), ps);
      st(mem.u6[0], j6, ps, row6);
    }
    wave_sync();
    {
      vf as = ld(mem.Minv, j6 * 6) * ld(mem.u6[0], spl(0));
      for(int l = 1; l < 6; l++) as = vfma(ld(mem.Minv, j6 * 6 + l), ld(mem.u6[0], spl(l)), as);
      st(mem.u6[1], j6, as, row6);
      vf bs = gt(0) * ld(mem.Minv, j6);
      for(int t = 1; t < 6; t++) bs = vfma(gt(t), ld(mem.Minv, t * 6 + j6), bs);
      st(mem.u6[2], j6, bs, row6);
    }
    wave_sync();
    vf gam = gt(0) * ld(mem.u6[1], spl(0));
    for(int t = 1; t < 6; t++) gam = vfma(gt(t), ld(mem.u6[1], spl(t)), gam);
    const vf den = vfma(splat(sigma), gam, splat(1.0));
    const double d0 = read_lane(den, 0);
    if(!(std::fabs(d0) >= 0.01) || !(std::fabs(d0) <= 1.7976931348623157e308)) return false; // S_UPDATE_DEN_MIN
    const vf scale = splat(sigma) / den;
    const vb live = lane < 36;
    const vi n = seli(live, lane, spl(0));
    const vi j = (n * 43) >> 8, l = n - 6 * j;
    const vf as = ld(mem.u6[1], j) * scale, sg = sigma * ld(mem.Gl, j * LG + r);
    const vf mi = vfma(-as, ld(mem.u6[2], l), ld(mem.Minv, n));
    const vf cf = vfma(sg, ld(mem.Gl, l * LG + r), ld(mem.Cf, n));
    const bool ok = (ballot(!(vabs(mi) <= 1.7976931348623157e308)) & 0xfffffffffull) == 0ull;
    st(mem.Minv, n, mi, live);
    st(mem.Cf, n, cf, live);
    wave_sync();
    return ok;
  }
  // sol = Quu_F,ff^-1 (q + Quu_F xcl) on the free rows in the cancellation-free form of oracle/ddp_tile.c s_direction:
  //   beta = Vx6 + V6r (G xcl); delta = beta - ratio V6r (G_f u_f); gamma = Minv delta; sol_r = ratio u_r + g_r' gamma
  template<int AB>
  W64_FN void qp_direction(const Qp & Q, const vb (&fr)[B], const vf (&xcl)[B], vf (&sol)[B]) const
  {
    double gxc[6], gfu[6], beta[6], tv[6], delta[6], gamma[6];
    six_sums<AB>(Q, xcl, gxc);
    {
      vf uf[B];
      for(int b = 0; b < AB; b++) uf[b] = sel(fr[b], Q.u[b], 0.0);
      six_sums<AB>(Q, uf, gfu);
    }
    // SPEC: beta_j = Vx6_j; fma(V6r[j][l], gxc_l, .), l = 0 .. 5
    {
      vf sb = ld(mem.Vx, j6 + FU0);
      for(int l = 0; l < 6; l++) sb = vfma(Q.v6r[l], splat(gxc[l]), sb);
      for(int j = 0; j < 6; j++) beta[j] = read_lane(sb, j);
    }
    apply6([&](int l) { return Q.v6r[l]; }, gfu, tv);
    for(int j = 0; j < 6; j++) delta[j] = std::fma(-Q.ratio, tv[j], beta[j]);
    apply6([&](int l) { return ld(mem.Minv, j6 * 6 + l); }, delta, gamma);
    for(int b = 0; b < AB; b++)
    {
      vf s = Q.ratio * Q.u[b];
      for(int j = 0; j < 6; j++) s = vfma(Q.G[b][j], splat(gamma[j]), s);
      sol[b] = sel(fr[b], s, 0.0);
    }
  }

  // Box-QP (Tassa's boxQP.m, nmpc_ddp's parameters): min 1/2 x'Hx + q'x, lo <= x <= hi over the first m ridges,
  // H = Quu_F in the structured form above.  x enters as the warm start.  On success (result >= 1) x is the minimiser,
  // freemask the free ridges (empty when everything is clamped) and mem.Cf / mem.Minv belong to that set.
  template<int AB>
  CCC_TILE_PIECE int box_qp(const Qp & Q, const vf (&q)[B], vf (&x)[B], mask_t & freemask)
  {
    // (the bounds lo = flo - u, hi = fhi - u of the ridges in range are formed where they are used, from Q.u, instead of
    //  being held through the iteration -- two or four registers per block of ridges at the point where the kernel's register
    //  pressure peaks; every use is masked by Q.in, beyond the step's dimension the values do not matter)
    auto lo_of = [&](int b) { return P.flo - Q.u[b]; };
    auto hi_of = [&](int b) { return P.fhi - Q.u[b]; };
    const double min_grad = 1e-8, min_rel_improve = 1e-8, step_dec = 0.6, min_step = 1e-22, armijo = 0.1;
    const int max_iter = 500; // nmpc_ddp BoxQP::Configuration::max_iter
    const int m = Q.m;
    vb cl[B], fr[B];
    for(int b = 0; b < AB; b++)
    {
      cl[b] = lane < 0; // all false
      fr[b] = lane < 0;
      x[b] = sel(Q.in[b], vmin(vmax(x[b], lo_of(b)), hi_of(b)), 0.0);
    }
    const mask_t inmask = (m >= M) ? kAll : static_cast<mask_t>((static_cast<mask_t>(1) << m) - 1u);
    // value(y) = sum_c y_c q_c + 1/2 y_c (H y)_c.  SPEC: gy = G y (six sums, treeM); vy = V6r gy (apply6); (H y)_r =
    // alpha y_r, then fma(G[j][r], vy_j, .); t_c = fma(0.5 y_c, (H y)_c, y_c q_c); sumM.
    // Evaluated for FOUR vectors at once, one per row of the wavefront (round 4: the Armijo search of Tassa's box-QP tries
    // step, 0.6 step, 0.36 step, ... in order; the instances that set a batch's makespan backtrack eight to ten times per
    // search, one dependent evaluation after the other): every row forms all six ridge sums of ITS vector with the same
    // trees, the 6 x 6 product on its lanes c < 6, and hands the six results round by DPP row broadcasts -- the same
    // operations in the same order as the one-vector form, so each row's value is that form's value bit for bit, with no
    // scalar round trip.  Returns the values (replicated in each row) and H y per row.
    // (H y is kept: the gradient of the next iteration is q + H x at the x the last value was taken at)
    vf hy[B];
    auto value_of4 = [&](const vf (&y)[B], vf (&hy4)[B]) {
      vf gy[6];
      for(int j = 0; j < 6; j++)
      {
        vf t[B];
        for(int b = 0; b < AB; b++) t[b] = Q.G[b][j] * y[b];
        gy[j] = sumM<AB>(t);
      }
      vf sv = Q.v6r[0] * gy[0];
      for(int l = 1; l < 6; l++) sv = vfma(Q.v6r[l], gy[l], sv);
      const vf vy[6] = {row_bcast<0>(sv), row_bcast<1>(sv), row_bcast<2>(sv), row_bcast<3>(sv), row_bcast<4>(sv), row_bcast<5>(sv)};
      vf t[B];
      for(int b = 0; b < AB; b++)
      {
        vf h = Q.alpha * y[b];
        for(int j = 0; j < 6; j++) h = vfma(Q.G[b][j], vy[j], h);
        hy4[b] = h;
        t[b] = vfma(0.5 * y[b], h, y[b] * q[b]);
      }
      return sumM<AB>(t);
    };
    // row w (wave-uniform) of v on every row
    auto pick_row = [&](vf v, int w) {
      vf a, b2, lo2, hi2;
      rows_pair(v, a, b2);
      halves_pair((w & 1) ? b2 : a, lo2, hi2);
      return (w & 2) ? hi2 : lo2;
    };
    auto value_of = [&](const vf (&y)[B]) { return read_lane(value_of4(y, hy), 0); }; // (the same vector on every row)
    TILE_PROF_START();
    TILE_PROF_COUNT(TP_QP_CALLS);
    double value = value_of(x), oldvalue = 0.0;
    TILE_PROF_ADD(TP_QP_VALUE);
    freemask = 0;
    mask_t factmask = 0; // the free set the factor in mem.Cf / mem.Minv belongs to
    int nupd = 0;        // rank-one updates since it was formed afresh
    int result = 0, iter;
    for(iter = 1; iter <= max_iter; iter++)
    {
      if(result != 0) break;
      TILE_PROF_COUNT(TP_QP_ITERS);
      if(iter > 1 && (oldvalue - value) < min_rel_improve * std::fabs(oldvalue))
      {
        result = 4;
        break;
      }
      oldvalue = value;
      vf grad[B];
      vb diff[B];
      for(int b = 0; b < AB; b++)
      {
        grad[b] = q[b] + hy[b];
        const vb oldc = cl[b];
        cl[b] = Q.in[b] && (((x[b] == lo_of(b)) && (grad[b] > 0.0)) || ((x[b] == hi_of(b)) && (grad[b] < 0.0)));
        diff[b] = cl[b] != oldc;
      }
      const mask_t clmask = ballotM<AB>(cl) & inmask;
      const bool changed = (iter == 1) || ((ballotM<AB>(diff) & inmask) != 0u);
      TILE_PROF_ADD(TP_QP_GRAD);
      if(clmask == inmask)
      {
        result = 6;
        break;
      }
      if(changed)
      {
        TILE_PROF_COUNT(TP_QP_FACTORS);
        freemask = inmask & ~clmask;
        for(int b = 0; b < AB; b++) fr[b] = Q.in[b] && !cl[b];
        // round 5: a set a few ridges from the factorised one is reached by rank-one updates, ridge by ridge in increasing
        // index (oracle/ddp_tile.c box_qp_struct); afresh in the first iteration, beyond kUpdateKmax changes, after
        // kUpdateUmax updates in a row, and whenever an update declines
        unsigned long long dm = static_cast<unsigned long long>((freemask ^ factmask) & inmask);
        const int nch = __builtin_popcountll(dm);
        bool fresh = (iter == 1) || nch > P.update_kmax || nupd + nch > kUpdateUmax;
        while(!fresh && dm != 0ull)
        {
          const int r = __builtin_ctzll(dm);
          dm &= dm - 1ull;
          TILE_PROF_COUNT(TP_F_UPDS);
          if(qp_update(Q, r, ((freemask >> r) & 1u) ? 1.0 : -1.0))
            nupd++;
          else
            fresh = true;
        }
#if defined(CCC_TILE_PROF) && defined(__HIP_DEVICE_COMPILE__)
        if(!fresh && (threadIdx.x & 63) == 0) mem.prof[TP_F_UPD] += (double)((long long)__builtin_readcyclecounter() - prof_t_);
#endif
        if(fresh)
        {
          if(!qp_factor<AB>(Q, freemask))
          {
            result = -1;
            break;
          }
          nupd = 0;
        }
        factmask = freemask;
      }
      TILE_PROF_ADD(TP_QP_FACTOR);
      vf t[B];
      for(int b = 0; b < AB; b++) t[b] = sel(fr[b], grad[b] * grad[b], 0.0);
      // |grad| on the free rows.  SPEC: sqrt(sumM(free ? grad^2 : 0))
      const double gn = std::sqrt(read_lane(sumM<AB>(t), 0));
      if(gn < min_grad)
      {
        result = 5;
        break;
      }
      // search = -H_ff^-1 (q + H (x .* clamped))_f - x_f
      vf xcl[B], rhs[B], srch[B];
      for(int b = 0; b < AB; b++) xcl[b] = sel(cl[b], x[b], 0.0);
      qp_direction<AB>(Q, fr, xcl, rhs);
      for(int b = 0; b < AB; b++)
      {
        srch[b] = sel(fr[b], -rhs[b] - x[b], 0.0);
        t[b] = srch[b] * grad[b];
      }
      const double sdotg = read_lane(sumM<AB>(t), 0);
      TILE_PROF_ADD(TP_QP_SOLVE);
      if(sdotg >= 0) break; // no descent direction: result stays 0
      double step = 1.0, vc = 0.0;
      vf xc[B];
      for(;;)
      {
        // the next four step sizes of the list, one per row; candidate k is the last one tried when it passes the Armijo
        // test or when the step after it would be below min_step (result 2) -- the first such row is where the
        // one-at-a-time loop stops
        const double s1 = step * step_dec, s2 = s1 * step_dec, s3 = s2 * step_dec;
        const vf stepv = sel(g == 0, splat(step), sel(g == 1, splat(s1), sel(g == 2, splat(s2), splat(s3))));
        vf xc4[B], hy4[B];
        for(int b = 0; b < AB; b++) xc4[b] = sel(Q.in[b], vmin(vmax(x[b] + stepv * srch[b], lo_of(b)), hi_of(b)), 0.0);
        const vf vc4 = value_of4(xc4, hy4);
        const vb pass = !(((vc4 - oldvalue) / (stepv * sdotg)) < armijo);
        const vb stop = pass || ((stepv * step_dec) < min_step);
        const unsigned long long sm = ballot(stop);
        if(sm == 0ull)
        {
          step = s3 * step_dec;
          continue;
        }
        const int w = (sm & 0xffffull) ? 0 : ((sm & 0xffff0000ull) ? 1 : ((sm & 0xffff00000000ull) ? 2 : 3));
        if(((ballot(pass) >> (16 * w)) & 1ull) == 0ull) result = 2;
        vc = read_lane(vc4, 16 * w);
        for(int b = 0; b < AB; b++)
        {
          xc[b] = pick_row(xc4[b], w);
          hy[b] = pick_row(hy4[b], w);
        }
        break;
      }
      for(int b = 0; b < AB; b++) x[b] = xc[b];
      value = vc;
      TILE_PROF_ADD(TP_QP_SEARCH);
    }
    if(iter > max_iter && result == 0) result = 1;
    freemask = (result == 6) ? static_cast<mask_t>(0) : (inmask & ~(ballotM<AB>(cl) & inmask));
    return result;
  }

  // ------------------------------------------------------------------------------------------------ backward pass
  W64_FN const double * xcur() const { return I.xbuf + static_cast<long>(cur) * (P.N + 1) * S; }
  W64_FN const double * ucur() const { return I.ubuf + static_cast<long>(cur) * P.N * M; }

  // oracle/ddp_tile.c backward_pass_struct: one step with the first AB <= B blocks of 16 ridges live (m <= 16 AB: the
  // further blocks are exact zeros in every sum of the specification and are left out); false when the box-QP fails.
  // gsum: sum_i max_c |k_c| / (|u_c| + 1)
  template<int AB>
  CCC_TILE_PIECE bool backward_step(int i, int m, int ph, vf x, const vf (&u)[B], vf (&kprev)[B], int mprev, double & gsum)
  {
    Qp Q;
    Q.m = m;
    const double lq = P.reg_type == 2 ? 0.0 : lambda, lv = P.reg_type == 2 ? lambda : 0.0;
    Q.alpha = P.w_force + lq;
    Q.inv_alpha = 1.0 / Q.alpha;
    Q.ratio = P.w_force * Q.inv_alpha;
    Q.lv = lv;
    for(int b = 0; b < AB; b++) Q.in[b] = c + 16 * b < m;
    TILE_PROF_START();
    const vf ref = ref_of(i); // (asked for here, used after the derivatives: the trip to memory runs beside them)
    Terms T;
    terms_of<AB>(ph, m, x, u, T);
    vf Fu[B][6];
    state_eq_deriv<AB>(T, x, Fu);
    TILE_PROF_ADD(TP_DERIV);
    const vi col = seli(inS, c, spl(0));
    // Qx = Lx + Fx' Vx (lanes a < S).  SPEC: s = Lx_a; s = fma(Fx[b][a], Vx[b], s), b = 0 .. S-1
    // (round 5: over the rows Fx's structure leaves -- oracle/ddp_tile.c FX_NZ; nz[k] = row b_k of this lane's column,
    //  fk[k] = Fx[b_k][c], shared by the three products)
    vi nz[kNzLen];
    const vi colo = opaque(col); // (the row numbers are a few operations: made here, every step, not kept across the solver)
    for(int k = 0; k < kNzLen; k++) nz[k] = tbl4(fx_nz_table(k), colo);
    {
      vf s = sel(inS, ld(mem.wrun, c) * (x - ref), 0.0);
      for(int k = 0; k < kNzLen; k++) s = vfma(ld(mem.Fx, nz[k] * S + col), ld(mem.Vx, nz[k]), s);
      st(mem.Qx, c, s, inS && (g == 0));
    }
    // T1 = Vxx Fx (lanes c < S).  SPEC: s = Vxx[a][b_0] Fx[b_0][c]; s = fma(Vxx[a][b_k], Fx[b_k][c], s), k = 1 .. L-1
    {
      vf s3[3];
      const vf f0 = ld(mem.Fx, nz[0] * S + col);
      for(int t = 0; t < 3; t++) s3[t] = ld(mem.Vxx, arow[t] * S + nz[0]) * f0;
      for(int k = 1; k < kNzLen; k++)
      {
        const vf fk = ld(mem.Fx, nz[k] * S + col);
        for(int t = 0; t < 3; t++) s3[t] = vfma(ld(mem.Vxx, arow[t] * S + nz[k]), fk, s3[t]);
      }
      for(int t = 0; t < 3; t++) st(mem.T1, arow[t] * S + col, s3[t], aval[t] && inS);
    }
    wave_sync();
    // W = T1[rows6, :] -> mem.W (T1's place takes Qxx below)
    for(int j = 0; j < 6; j++) st(mem.W, j * LS + c, ld(mem.T1, (FU0 + j) * S + col), inS && (g == 0));
    // Qxx = Lxx + Fx' T1; this lane: Qxx[c][r], r = its rows.  SPEC: s = (c == r) w_run[c]; s = fma(Fx[b_k][c], T1[b_k][r], s)
    vf Qxx[3];
    {
      for(int t = 0; t < 3; t++) Qxx[t] = sel(arow[t] == c, ld(mem.wrun, arow[t]), 0.0);
      for(int k = 0; k < kNzLen; k++)
      {
        const vf fk = ld(mem.Fx, nz[k] * S + col);
        for(int t = 0; t < 3; t++) Qxx[t] = vfma(fk, ld(mem.T1, nz[k] * S + arow[t]), Qxx[t]);
      }
    }
    wave_sync();
    for(int t = 0; t < 3; t++) st(mem.T1, col * S + arow[t], Qxx[t], aval[t] && inS); // T1 <- Qxx
    wave_sync();
    // Qu = Lu + Fu' Vx.  SPEC: s = w_force u_c; s = fma(Fu[b][c], Vx[b], s), b = FU0 .. FU0+5
    vf Qu[B];
    for(int b = 0; b < AB; b++)
    {
      vf s = P.w_force * u[b];
      for(int bb = 0; bb < 6; bb++) s = vfma(Fu[b][bb], splat(mem.Vx[FU0 + bb]), s);
      Qu[b] = sel(Q.in[b], s, 0.0);
    }
    // the ridge vectors g_r (zero beyond the step's dimension), also in LDS for the C_f sums
    for(int b = 0; b < AB; b++)
    {
      for(int j = 0; j < 6; j++)
      {
        Q.G[b][j] = sel(Q.in[b], Fu[b][j], 0.0);
        st(mem.Gl, j * LG + c + 16 * b, Q.G[b][j], g == 0);
      }
      Q.Ga[b] = pick4(g, Q.G[b]);
      Q.Gb[b] = sel((g & 1) == 0, Q.G[b][4], Q.G[b][5]);
      Q.u[b] = u[b];
    }
    for(int l = 0; l < 6; l++) Q.v6r[l] = v6r_elem(l, lv);
    TILE_PROF_ADD(TP_PRODUCTS);
    // box-QP and gains
    vf k[B], K[B][3];
    vb fr[B];
    for(int b = 0; b < AB; b++)
    {
      k[b] = splat(0.0);
      fr[b] = lane < 0;
      for(int t = 0; t < 3; t++) K[b][t] = splat(0.0);
    }
    mask_t freemask = 0;
    if(m > 0)
    {
      // warm start: the feed-forward of step i + 1 of this pass (zeros for the last step or on a dimension change); the
      // bounds force_lo - u, force_hi - u are formed inside from Q.u
      for(int b = 0; b < AB; b++) k[b] = (mprev == m) ? kprev[b] : splat(0.0);
      const int rc = box_qp<AB>(Q, Qu, k, freemask);
      if(rc < 1) return false;
      TILE_PROF_ADD(TP_OTHER); // (the box-QP accounts for itself: this slice is its entry and exit)
      for(int b = 0; b < AB; b++) fr[b] = row_in(freemask, b);
    }
    // Y = M_f^-1 Wr: lane (g, a), a < S, forms the rows j = g and j = 4 + g (g < 2).  SPEC: s = Minv[j][0] Wr[0][a];
    // fma(Minv[j][t], Wr[t][a], .), t = 1 .. 5; Wr[t][a] = fma(lv, Fx[FU0 + t][a], W[t][a]).  Nothing free (or no
    // contact): Y = 0, C_f = 0
    const vi jA = g, jB = seli(g < 2, g + 4, spl(5));
    const vb hasB = g < 2;
    {
      vf wr[6];
      for(int t = 0; t < 6; t++) wr[t] = vfma(splat(lv), ld(mem.Fx, (FU0 + t) * S + col), ld(mem.W, t * LS + col));
      vf sA = ld(mem.Minv, jA * 6) * wr[0], sB = ld(mem.Minv, jB * 6) * wr[0];
      for(int t = 1; t < 6; t++)
      {
        sA = vfma(ld(mem.Minv, jA * 6 + t), wr[t], sA);
        sB = vfma(ld(mem.Minv, jB * 6 + t), wr[t], sB);
      }
      if(freemask == 0)
      {
        sA = splat(0.0);
        sB = splat(0.0);
        st(mem.Cf, seli(lane < 36, lane, spl(0)), splat(0.0), lane < 36);
      }
      st(mem.Y, jA * LS + c, sA, inS);
      st(mem.Y, jB * LS + c, sB, inS && hasB);
    }
    wave_sync();
    // K = -G_f' Y (rows a_t of the columns c + 16 b), clamped rows = 0.
    // SPEC: s = G[0][r] Y[0][a]; fma(G[j][r], Y[j][a], .), j = 1 .. 5; K[r][a] = free ? -s : 0
    for(int t = 0; t < 3; t++)
    {
      vf ya[6];
      for(int j = 0; j < 6; j++) ya[j] = ld(mem.Y, j * LS + arow[t]);
      for(int b = 0; b < AB; b++)
      {
        vf s = Q.G[b][0] * ya[0];
        for(int j = 1; j < 6; j++) s = vfma(Q.G[b][j], ya[j], s);
        K[b][t] = sel(fr[b] && aval[t], -s, 0.0);
      }
    }
    // gains -> global memory (the forward passes read them)
    for(int b = 0; b < AB; b++)
    {
      st(I.ks + static_cast<long>(i) * M, c + 16 * b, k[b], g == 0);
      for(int t = 0; t < 3; t++) st(I.Ks + static_cast<long>(i) * M * S, (c + 16 * b) * S + arow[t], K[b][t], aval[t]);
    }
    TILE_PROF_ADD(TP_GAINS);
    // termination measure: max_c |k_c| / (|u_c| + 1)
    {
      vf mx = sel(Q.in[0], vabs(k[0]) / (vabs(u[0]) + 1.0), 0.0);
      for(int b = 1; b < AB; b++) mx = vmax(mx, sel(Q.in[b], vabs(k[b]) / (vabs(u[b]) + 1.0), 0.0));
      gsum += read_lane(max16(mx), 0);
    }
    // gk = G k; vk = V6 gk (unregularised); t4 = Quu k.  SPEC: t4_r = w_force k_r; fma(G[j][r], vk_j, .), j = 0 .. 5
    double gk[6], vk[6], gfv[6];
    six_sums<AB>(Q, k, gk);
    apply6([&](int l) { return ld(mem.Vxx, (j6 + FU0) * S + FU0 + l); }, gk, vk);
    vf t4[B];
    for(int b = 0; b < AB; b++)
    {
      vf s = P.w_force * k[b];
      for(int j = 0; j < 6; j++) s = vfma(Q.G[b][j], splat(vk[j]), s);
      t4[b] = s;
    }
    // dV += [k'Qu, 1/2 k'Quu k].  SPEC: sumM(k_c Qu_c), 0.5 sumM(k_c (Quu k)_c)
    {
      vf t[B];
      for(int b = 0; b < AB; b++) t[b] = k[b] * Qu[b];
      dV0 += read_lane(sumM<AB>(t), 0);
      for(int b = 0; b < AB; b++) t[b] = k[b] * t4[b];
      dV1 += 0.5 * read_lane(sumM<AB>(t), 0);
    }
    // gfv = G_f (Quu k + Qu).  SPEC: treeM(free ? G[j][r] (t4_r + Qu_r) : 0)
    {
      vf v[B];
      for(int b = 0; b < AB; b++) v[b] = sel(fr[b], t4[b] + Qu[b], 0.0);
      six_sums<AB>(Q, v, gfv);
    }
    // Vx = Qx + W' gk - Y' gfv (lanes a < S); D = C_f Y and E = w_force Y - 2 W + V6 D: lane (g, a) the rows g and 4 + g
    {
      vf y6[6];
      for(int j = 0; j < 6; j++) y6[j] = ld(mem.Y, j * LS + col);
      vf s = ld(mem.Qx, col);
      for(int j = 0; j < 6; j++) s = vfma(ld(mem.W, j * LS + col), splat(gk[j]), s);
      for(int j = 0; j < 6; j++) s = vfma(-y6[j], splat(gfv[j]), s);
      st(mem.Vx, c, s, inS && (g == 0));
      // SPEC: D[j][a] = Cf[j][0] Y[0][a]; fma(Cf[j][t], Y[t][a], .), t = 1 .. 5
      vf dA = ld(mem.Cf, jA * 6) * y6[0], dB = ld(mem.Cf, jB * 6) * y6[0];
      for(int t = 1; t < 6; t++)
      {
        dA = vfma(ld(mem.Cf, jA * 6 + t), y6[t], dA);
        dB = vfma(ld(mem.Cf, jB * 6 + t), y6[t], dB);
      }
      st(mem.D, jA * LS + c, dA, inS);
      st(mem.D, jB * LS + c, dB, inS && hasB);
      wave_sync();
      // SPEC: E[j][a] = w_force Y[j][a] - 2 W[j][a]; fma(V6[j][t], D[t][a], .), t = 0 .. 5
      vf eA = P.w_force * ld(mem.Y, jA * LS + col) - 2.0 * ld(mem.W, jA * LS + col);
      vf eB = P.w_force * ld(mem.Y, jB * LS + col) - 2.0 * ld(mem.W, jB * LS + col);
      for(int t = 0; t < 6; t++)
      {
        const vf dt_ = ld(mem.D, t * LS + col);
        eA = vfma(ld(mem.Vxx, (jA + FU0) * S + FU0 + t), dt_, eA);
        eB = vfma(ld(mem.Vxx, (jB + FU0) * S + FU0 + t), dt_, eB);
      }
      st(mem.E, jA * LS + c, eA, inS);
      st(mem.E, jB * LS + c, eB, inS && hasB);
    }
    wave_sync();
    // Vxx(a, b) = Vxx(b, a) = 1/2 ((Qxx(a,b) + Qxx(b,a)) + (T(a,b) + T(b,a))), T = D'E
    // SPEC: tab = D[0][a] E[0][b]; fma(D[j][a], E[j][b], .), j = 1 .. 5; tba likewise
    for(int q = 0; q < NPASS; q++)
    {
      const vi pidx = lane + 64 * q;
      const vb pv = pidx < NP;
      const vi ab = ldb(mem.pair, seli(pv, pidx, spl(0)));
      const vi pa = ab & 15, pb = ab >> 4;
      vf tab = ld(mem.D, pa) * ld(mem.E, pb), tba = ld(mem.D, pb) * ld(mem.E, pa);
      for(int j = 1; j < 6; j++)
      {
        tab = vfma(ld(mem.D, j * LS + pa), ld(mem.E, j * LS + pb), tab);
        tba = vfma(ld(mem.D, j * LS + pb), ld(mem.E, j * LS + pa), tba);
      }
      const vf v = 0.5 * ((ld(mem.T1, pa * S + pb) + ld(mem.T1, pb * S + pa)) + (tab + tba));
      st(mem.Vxx, pa * S + pb, v, pv);
      st(mem.Vxx, pb * S + pa, v, pv);
    }
    wave_sync();
    TILE_PROF_ADD(TP_VALUE);
    for(int b = 0; b < B; b++) kprev[b] = b < AB ? k[b] : splat(0.0);
    return true;
  }

  CCC_TILE_PIECE bool backward_pass(double & gsum)
  {
    const int N = P.N;
    const double * xs = xcur();
    const double * us = ucur();
    mem_sync(); // (the trajectory was written by other lanes of this wavefront)
    // terminal value: src/DdpCentroidal.cpp:156-177 at x_N
    for(int e = 0; e < S * S; e += 64) st(mem.Vxx, lane + e, splat(0.0), lane + e < S * S);
    wave_sync();
    st(mem.Vxx, seli(inS, c * S + c, spl(0)), ld(mem.wterm, c), inS && (g == 0));
    {
      const vf xN = ldm(xs + static_cast<long>(N) * S, c, inS);
      st(mem.Vx, c, sel(inS, ld(mem.wterm, c) * (xN - ref_of(N)), 0.0), g == 0);
    }
    wave_sync();
    dV0 = 0.0;
    dV1 = 0.0;
    gsum = 0.0;
    vf kprev[B];
    for(int b = 0; b < B; b++) kprev[b] = splat(0.0);
    int mprev = -1;
    // the operands of step i - 1 are fetched while step i computes
    int ph_n, m_n;
    step_info(N - 1, ph_n, m_n);
    vf x_n = ldm(xs + static_cast<long>(N - 1) * S, c, inS);
    vf u_n[B];
    for(int b = 0; b < B; b++) u_n[b] = ldm(us + static_cast<long>(N - 1) * M, c + 16 * b, c + 16 * b < m_n);
    for(int i = N - 1; i >= 0; i--)
    {
      const int m = m_n, ph = ph_n;
      vf u[B];
      for(int b = 0; b < B; b++) u[b] = u_n[b];
      const vf x = x_n;
      if(i > 0)
      {
        step_info(i - 1, ph_n, m_n);
        x_n = ldm(xs + static_cast<long>(i - 1) * S, c, inS);
        for(int b = 0; b < B; b++) u_n[b] = ldm(us + static_cast<long>(i - 1) * M, c + 16 * b, c + 16 * b < m_n);
      }
      bool ok;
      if constexpr(B == 1)
        ok = backward_step<1>(i, m, ph, x, u, kprev, mprev, gsum);
      else if constexpr(B == 2)
        ok = m <= 16 ? backward_step<1>(i, m, ph, x, u, kprev, mprev, gsum) : backward_step<2>(i, m, ph, x, u, kprev, mprev, gsum);
      else
        ok = m <= 16 ? backward_step<1>(i, m, ph, x, u, kprev, mprev, gsum)
                     : (m <= 32 ? backward_step<2>(i, m, ph, x, u, kprev, mprev, gsum) : backward_step<4>(i, m, ph, x, u, kprev, mprev, gsum));
      if(!ok) return false;
      mprev = m;
    }
    return true;
  }

  // ------------------------------------------------------------------------------------------------ forward passes
  // Four candidates at once: row g rolls out alpha[first + g] into slot cand[g].  Returns the costs per row.
#ifdef CCC_TILE_NO_PREFETCH_K
  static constexpr bool kPrefetchK = false;
#else
  static constexpr bool kPrefetchK = (B == 1); // the gain rows are fetched ahead with the step's other operands (at 32 and
                                                // 64 ridges per step they are loaded in the step: fetched ahead, 36 / 72
                                                // more registers cost more than the wait -- measured, 64 ridges 46.9 -> 49.8 ms)
#endif
  // steps the forward pass fetches ahead (see forward_pass)
#ifdef CCC_TILE_FWD_DEPTH
  static constexpr int kFwdDepth = CCC_TILE_FWD_DEPTH;
#else
  static constexpr int kFwdDepth = (B == 1 && S == 9) ? 2 : 1; // (S = 12: a second buffer spills 42 dwords for no gain, measured)
#endif
  struct FwdOps
  {
    int ph, m;
    vf xi, ref, ui[B], ki[B];
    vf Kr[kPrefetchK ? B : 1][kPrefetchK ? S : 1];
  };
  // one step of the rollouts with the first AB <= B blocks of 16 ridges live (the further ones are written as zeros)
  template<int AB>
  W64_FN void forward_step(int i, int m, int ph, vf alpha, vi xoff, vi uoff, vf xi, vf ref, const vf (&ui)[B], const vf (&ki)[B],
                           const vf (&Krp)[kPrefetchK ? B : 1][kPrefetchK ? S : 1], vf & x, vf & costc)
  {
    TILE_PROF_START();
    const vf dx = x - xi;
    // SPEC: s = u_c + alpha k_c; s = fma(K[c][a], dx_a, s), a = 0 .. S-1; clamp
    vf un[B];
    for(int b = 0; b < AB; b++)
    {
      const vb in = c + 16 * b < m;
      vf Kr[S];
      for(int a = 0; a < S; a++)
      {
        if constexpr(kPrefetchK)
          Kr[a] = Krp[b][a];
        else
          Kr[a] = ldm(I.Ks + static_cast<long>(i) * M * S, (c + 16 * b) * S + a, in);
      }
      vf s = ui[b] + alpha * ki[b];
      s = feedback<0>(s, dx, Kr);
      un[b] = sel(in, vmin(vmax(s, P.flo), P.fhi), 0.0);
      st(I.ubuf, uoff + i * M + c + 16 * b, un[b], c < 16);
    }
    for(int b = AB; b < B; b++) st(I.ubuf, uoff + i * M + c + 16 * b, splat(0.0), c < 16);
    TILE_PROF_ADD(TP_FW_FEED);
    costc = costc + running_cost<AB>(ref, x, un);
    TILE_PROF_ADD(TP_FW_COST);
    Terms T;
    terms_of<AB>(ph, m, x, un, T);
    TILE_PROF_ADD(TP_FW_TERMS);
    x = state_eq(T, x);
    st(I.xbuf, xoff + (i + 1) * S + c, x, inS);
    TILE_PROF_ADD(TP_FW_STATE);
  }

  CCC_TILE_PIECE vf forward_pass(int first, const int (&cand)[4])
  {
    const int N = P.N;
    const double * xs = xcur();
    const double * us = ucur();
    const vf alpha = ld(mem.alpha, seli(g + first < 12, g + first, spl(11)));
    const vi slot = seli(g == 0, spl(cand[0]), seli(g == 1, spl(cand[1]), seli(g == 2, spl(cand[2]), spl(cand[3]))));
    const vi xoff = slot * ((N + 1) * S), uoff = slot * (N * M);
    mem_sync(); // (the gains were written by other lanes of this wavefront)
    TILE_PROF_START();
    TILE_PROF_COUNT(TP_FORWARDS);
    vf x = ldm(I.x0, c, inS);
    vf costc = splat(0.0);
    st(I.xbuf, xoff + c, x, inS);
    // The operands of step i + kFwdDepth are fetched while step i computes: the gains of an instance (115 KB at 100 steps
    // of 16 ridges) were written by the backward pass and come back from HBM.  Round 5: two steps ahead instead of one at 16
    // ridges (config 3 52.2 -> 51.2 ms; three steps ahead: 54.7 ms, the third buffer spills).  Explicit buffers o0 .. o2
    // taken in turn by a loop unrolled by the depth: indexed by a constant, registers.  Same loads, same arithmetic, same
    // order: the same bits.
    FwdOps o0, o1, o2;
    auto fetch = [&](int i, FwdOps & o) {
      step_info(i, o.ph, o.m);
      o.xi = ldm(xs + static_cast<long>(i) * S, c, inS);
      o.ref = ref_of(i);
      for(int b = 0; b < B; b++)
      {
        const vb inn = c + 16 * b < o.m;
        o.ui[b] = ldm(us + static_cast<long>(i) * M, c + 16 * b, inn);
        o.ki[b] = ldm(I.ks + static_cast<long>(i) * M, c + 16 * b, inn);
        if constexpr(kPrefetchK)
          for(int a = 0; a < S; a++) o.Kr[b][a] = ldm(I.Ks + static_cast<long>(i) * M * S, (c + 16 * b) * S + a, inn);
      }
    };
    auto run = [&](int i, FwdOps & o) {
      // (the step's operands by value, then the buffer is free for the fetch of step i + kFwdDepth)
      const FwdOps cur = o;
#if defined(CCC_TILE_PROF) && defined(__HIP_DEVICE_COMPILE__)
      const long long fw_t0 = (long long)__builtin_readcyclecounter();
#endif
      if(i + kFwdDepth < N) fetch(i + kFwdDepth, o);
#if defined(CCC_TILE_PROF) && defined(__HIP_DEVICE_COMPILE__)
      if((threadIdx.x & 63) == 0) mem.prof[TP_FW_FETCH] += (double)((long long)__builtin_readcyclecounter() - fw_t0);
#endif
      const int m = cur.m, ph = cur.ph;
      if constexpr(B == 1)
        forward_step<1>(i, m, ph, alpha, xoff, uoff, cur.xi, cur.ref, cur.ui, cur.ki, cur.Kr, x, costc);
      else if constexpr(B == 2)
      {
        if(m <= 16)
          forward_step<1>(i, m, ph, alpha, xoff, uoff, cur.xi, cur.ref, cur.ui, cur.ki, cur.Kr, x, costc);
        else
          forward_step<2>(i, m, ph, alpha, xoff, uoff, cur.xi, cur.ref, cur.ui, cur.ki, cur.Kr, x, costc);
      }
      else
      {
        if(m <= 16)
          forward_step<1>(i, m, ph, alpha, xoff, uoff, cur.xi, cur.ref, cur.ui, cur.ki, cur.Kr, x, costc);
        else if(m <= 32)
          forward_step<2>(i, m, ph, alpha, xoff, uoff, cur.xi, cur.ref, cur.ui, cur.ki, cur.Kr, x, costc);
        else
          forward_step<4>(i, m, ph, alpha, xoff, uoff, cur.xi, cur.ref, cur.ui, cur.ki, cur.Kr, x, costc);
      }
    };
    fetch(0, o0);
    if(kFwdDepth > 1 && 1 < N) fetch(1, o1);
    if(kFwdDepth > 2 && 2 < N) fetch(2, o2);
    for(int i = 0; i < N; i += kFwdDepth)
    {
      run(i, o0);
      if(kFwdDepth > 1 && i + 1 < N) run(i + 1, o1);
      if(kFwdDepth > 2 && i + 2 < N) run(i + 2, o2);
    }
    const vf total = costc + terminal_cost(x);
    TILE_PROF_ADD(TP_FORWARD);
    return total;
  }
  template<int A>
  W64_FN vf feedback(vf s, vf dx, const vf (&Kr)[S]) const
  {
    if constexpr(A < S)
    {
      s = vfma(Kr[A], row_bcast<A>(dx), s);
      return feedback<A + 1>(s, dx, Kr);
    }
    else
      return s;
  }

  // the trajectory of the initial inputs (u_init, or zeros) into `slot`; returns its cost
  CCC_TILE_PIECE double initial_rollout(int slot, const double * u_init)
  {
    const int N = P.N;
    const long xo = static_cast<long>(slot) * (N + 1) * S, uo = static_cast<long>(slot) * N * M;
    vf x = ldm(I.x0, c, inS);
    vf cc = splat(0.0);
    st(I.xbuf + xo, c, x, inS && (g == 0));
    vf ref_n = ref_of(0); // (the reference a step ahead, as in the forward passes)
    for(int i = 0; i < N; i++)
    {
      int ph, m;
      step_info(i, ph, m);
      const vf ref = ref_n;
      if(i + 1 < N) ref_n = ref_of(i + 1);
      vf u[B];
      for(int b = 0; b < B; b++)
      {
        const vb in = c + 16 * b < m;
        u[b] = u_init ? ldm(u_init + static_cast<long>(i) * M, c + 16 * b, in) : splat(0.0);
        st(I.ubuf + uo, i * M + c + 16 * b, u[b], g == 0);
      }
      cc = cc + running_cost<B>(ref, x, u);
      Terms T;
      terms_of<B>(ph, m, x, u, T);
      x = state_eq(T, x);
      st(I.xbuf + xo, (i + 1) * S + c, x, inS && (g == 0));
    }
    return read_lane(cc + terminal_cost(x), 0);
  }

  // ------------------------------------------------------------------------------------------------ the solve
  W64_FN void increase_lambda()
  {
    dlambda = std::fmax(dlambda * P.lambda_factor, P.lambda_factor);
    lambda = std::fmax(lambda * dlambda, P.lambda_min);
  }
  W64_FN void decrease_lambda()
  {
    dlambda = std::fmin(dlambda / P.lambda_factor, 1.0 / P.lambda_factor);
    lambda = lambda * dlambda * (lambda > P.lambda_min ? 1.0 : 0.0);
  }

  // oracle/ddp.c oracle_ddp_solve, in three pieces so that the launch code can run an instance in SLICES of iterations
  // (csrc/ddp_tile.hip: a batch larger than one resident set is scheduled longest-first from the time its first
  // iterations took): begin() = start-up, iterate(budget) = up to `budget` iterations (false: the solve goes on),
  // finish() = outputs; suspend() / resume() carry the complete state of the iteration across slices -- the current
  // trajectory, its cost, the regularisation and the iteration count -- so a sliced solve is the unsliced one bit for bit.
  int iters_done, exit_status; // (whether the warm-start guard fired lives in mem.warm_replaced)
  W64_FN void solve_instance()
  {
    begin();
    (void)iterate(-1);
    finish();
  }
  W64_FN void begin()
  {
    init();
    lambda = P.lambda0;
    dlambda = P.dlambda0;
    cur = 0;
    mem.warm_replaced = 0;
    cost = initial_rollout(0, I.u_init);
    if(P.warm_guard && I.u_init)
    {
      // warm-start guard (oracle/ddp_tile.c; not nmpc_ddp): a warm start that rolls out worse than zero inputs -- the
      // start of src/DdpCentroidal.cpp:221-229 -- or not finite is dropped
      const double cold = initial_rollout(1, nullptr);
      if(!(cost <= cold))
      {
        cur = 1;
        cost = cold;
        mem.warm_replaced = 1;
      }
    }
    iters_done = 0;
    exit_status = 0;
  }
  // the state of a suspended solve: inputs -> I.u_out (overwritten by the result in the end), states -> sx [(N+1) S],
  // scalars -> ss [0..3] and ss [6] (the guard's flag; ss [4..5] belong to the timing aid of the launch code)
  W64_FN void suspend(double * sx, double * ss)
  {
    mem_sync();
    const int N = P.N;
    const double * xs = xcur();
    const double * us = ucur();
    for(int e = 0; e < N * M; e += 64) st(I.u_out, lane + e, ldm(us, lane + e, lane + e < N * M), lane + e < N * M);
    for(int e = 0; e < (N + 1) * S; e += 64) st(sx, lane + e, ldm(xs, lane + e, lane + e < (N + 1) * S), lane + e < (N + 1) * S);
    const vf sc = sel(lane == 0, splat(cost), sel(lane == 1, splat(lambda), sel(lane == 2, splat(dlambda),
                  sel(lane == 3, splat((double)iters_done), splat((double)mem.warm_replaced)))));
    st(ss, lane, sc, (lane < 4) || (lane == 6));
    mem_sync();
  }
  W64_FN void resume(const double * sx, const double * ss)
  {
    init();
    const int N = P.N;
    cur = 0;
    for(int e = 0; e < N * M; e += 64) st(I.ubuf, lane + e, ldm(I.u_out, lane + e, lane + e < N * M), lane + e < N * M);
    for(int e = 0; e < (N + 1) * S; e += 64) st(I.xbuf, lane + e, ldm(sx, lane + e, lane + e < (N + 1) * S), lane + e < (N + 1) * S);
    cost = ss[0];
    lambda = ss[1];
    dlambda = ss[2];
    iters_done = (int)ss[3];
    mem.warm_replaced = (int)ss[6];
    exit_status = 0;
    mem_sync();
  }
#if defined(CCC_TILE_TIMING)
  double timing_busy, timing_first; // (development aid, set by the launch code: 100 MHz ticks)
  long long timing_slice0;
#endif
  W64_FN bool iterate(int budget)
  {
    int iter = 0, status = 0, used = 0;
    for(iter = iters_done + 1; iter <= P.max_iter; iter++)
    {
      if(budget >= 0 && used == budget)
      {
        iters_done = iter - 1; // (iterations completed; none of them ended the solve)
        return false;
      }
      used++;
      bool bp_ok = false;
      double gsum = 0.0;
      for(;;)
      {
        if(backward_pass(gsum))
        {
          bp_ok = true;
          break;
        }
        increase_lambda();
        if(lambda > P.lambda_max) break;
      }
      if(!bp_ok)
      {
        status = -1;
        break;
      }
      const double gn = gsum / P.N;
      if(gn < P.k_rel_norm_thre && lambda < P.lambda_thre)
      {
        decrease_lambda();
        status = 1;
        break;
      }
      bool accepted = false;
      double actual = 0.0;
      int win = 0;
      // the four slots that do not hold the current trajectory
      int cand[4];
      for(int q = 0, s = 0; s < kSlots; s++)
        if(s != cur) cand[q++] = s;
      for(int first = 0; first < 11 && !accepted; first += 4)
      {
        const vf costc = forward_pass(first, cand);
        const vf alpha = ld(mem.alpha, seli(g + first < 12, g + first, spl(11)));
        const vf act = cost - costc;
        const vf expected = -alpha * (dV0 + alpha * dV1);
        const vf ratio = sel(expected > 0.0, act / expected, sel(act > 0.0, 1.0, sel(act < 0.0, -1.0, 0.0)));
        const vb acc = (ratio > P.ratio_thre) && (g + first < 11);
        const unsigned long long am = ballot(acc);
        if(am != 0ull)
        {
          // the first accepted step size in the list's order = the lowest row
          win = (am & 0xffffull) ? 0 : ((am & 0xffff0000ull) ? 1 : ((am & 0xffff00000000ull) ? 2 : 3));
          accepted = true;
          actual = read_lane(act, 16 * win);
          cost = read_lane(costc, 16 * win);
        }
      }
      if(accepted)
      {
        decrease_lambda();
        cur = cand[win];
        if(actual < P.cost_thre)
        {
          status = 2;
          break;
        }
      }
      else
      {
        increase_lambda();
        if(lambda > P.lambda_max)
        {
          status = -1;
          break;
        }
      }
    }
    if(iter > P.max_iter) iter = P.max_iter;
    iters_done = iter;
    exit_status = status;
    return true;
  }
  W64_FN void finish()
  {
    const int iter = iters_done, status = exit_status;
    mem_sync();
    // results
    {
      const int N = P.N;
      const double * xs = xcur();
      const double * us = ucur();
      for(int e = 0; e < N * M; e += 64) st(I.u_out, lane + e, ldm(us, lane + e, lane + e < N * M), lane + e < N * M);
      if(I.x_out)
        for(int e = 0; e < (N + 1) * S; e += 64)
          st(I.x_out, lane + e, ldm(xs, lane + e, lane + e < (N + 1) * S), lane + e < (N + 1) * S);
#if defined(CCC_TILE_PROF) && defined(__HIP_DEVICE_COMPILE__)
      mem_sync();
      if((threadIdx.x & 63) == 0)
        for(int e = 0; e < TP_N; e++) I.u_out[e] = mem.prof[e];
#endif
      if(I.out_iters) I.out_iters[0] = iter;
      // (the exit code as it always was unless the guard fired: ccc_amd.h CCC_DDP_STATUS_WARM_REPLACED)
      if(I.out_status) I.out_status[0] = mem.warm_replaced ? (0x100 | (status & 0xff)) : status;
      if(I.out_cost) I.out_cost[0] = cost;
#if defined(CCC_TILE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
      // (development aid, scripts/ddp_sched_probe.py: the cost output carries the busy ticks, the first planned input the
      //  first start tick and the second the finish tick)
      const long long timing_now = (long long)wall_clock64();
      if(I.out_cost) I.out_cost[0] = timing_busy + (double)(timing_now - timing_slice0);
      if((threadIdx.x & 63) == 0)
      {
        I.u_out[0] = timing_first;
        I.u_out[1] = (double)timing_now;
      }
#endif
    }
  }
};
} // namespace ddp_tile
} // namespace ccc_amd
