// zmp.hip -- batched CCC::LinearMpcZmp::planOnce() on MI355X (gfx950): kernels + C-ABI.
//
// Path replaced (reference file:line under /root/reference):
//   src/CommonModels.cpp:8-17, include/CCC/StateSpaceModel.h:164-216,
//   include/CCC/InvariantSequentialExtension.h:103-181        -> ZmpModel (host, closed form, once per handle)
//   src/LinearMpcZmp.cpp:46-81 (procOnce) incl. the external QP solve at :69
//   src/LinearMpcZmp.cpp:83-112 (planOnce: x axis then y axis)  -> zmp_plan_kernel (device, per instance)
//
// The QP of src/LinearMpcZmp.cpp:21-27,54-69 is   min 1/2 |u|^2   s.t.  lo <= B u <= hi,
//   B = B_seq (lower-triangular Toeplitz, invertible),  lo = zmin - A_seq x0,  hi = zmax - A_seq x0.
// Its KKT system in the multipliers mu (u = B' mu, G = B B' batch-constant SPD) reads
//   (G mu)_i = lo_i if mu_i > 0,  = hi_i if mu_i < 0,  in [lo_i, hi_i] if mu_i = 0.
// The kernel runs the Goldfarb-Idnani dual active-set iteration directly on that system: the working
// set W grows by the most violated row and shrinks by the dual ratio test.  Instead of a QR/Cholesky
// factorisation it keeps the symmetric sweep tableau T of G swept on W (T_WW = -G_WW^-1,
// T_iW = G_iW G_WW^-1, T_ij = Schur complement otherwise): ONE LANE PER ROW, the row in that lane's
// VGPRs.  Column p of T (search direction, Schur complement, ratio test) is row p by symmetry, so the
// lane that owns row p publishes it through a 300-byte LDS scratch and every lane of the group reads
// its own element plus the broadcast of the whole row; adding / dropping a row is then one rank-1
// update T -= g v' of register-resident data (N FMAs per lane, no data-dependent loop bounds).
//
// Mapping: N <= 32: the two axes of one instance are the two 32-lane halves of ONE wavefront (one
// planOnce() per wavefront); 32 < N <= 64: one axis per wavefront.
#include "common.h"
#include "wave_group.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace ccc_amd
{
struct ZmpDev
{
  int N;            // horizon steps
  const double * G; // [NP][NP] B_seq B_seq', zero padded
  const double * A; // [NP][3]  A_seq rows, zero padded
  const double * b; // [NP]     first column of B_seq (B_seq[i][j] = b[i-j])
  double c2;        // C(0,2) = -com_height / g
  double dt;        // horizon_dt
};

constexpr double kInf = __builtin_huge_val();

// scratch doubles per group: the published row [0, NP), then sign / reciprocal pivot
template<int NP>
struct ZmpScratch
{
  static constexpr int kSig = NP;
  static constexpr int kRp = NP + 1;
  static constexpr int kSize = NP + 8; // keeps every group's base 16-byte aligned
};

template<int LG, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void zmp_plan_kernel(ZmpDev P, long nqp, const double * __restrict__ x0,
                                                             const double * __restrict__ zlim, double control_dt,
                                                             double * __restrict__ zmp, double * __restrict__ jerk,
                                                             int * __restrict__ status)
{
  using Grp = WaveGroup<LG>;
  using Scr = ZmpScratch<LG>;
  constexpr int NP = LG;
  constexpr int QPW = 64 / LG; // QPs per wavefront
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double * Gs = smem;           // [NP][NP]
  double * bs = smem + NP * NP; // [NP]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & (LG - 1), grp = lane / LG;
  double * scr = smem + NP * NP + NP + (wave * QPW + grp) * Scr::kSize; // this group's scratch
  const double2 * scr2 = reinterpret_cast<const double2 *>(scr);

  for(int k = tid; k < NP * NP; k += WAVES * 64) Gs[k] = P.G[k];
  for(int k = tid; k < NP; k += WAVES * 64) bs[k] = P.b[k];
  __syncthreads();

  const int N = P.N;
  const double a0 = P.A[li * 3 + 0], a1 = P.A[li * 3 + 1], a2 = P.A[li * 3 + 2];
  const double bi = bs[li];
  const int maxpass = 20 * N + 100;

  const long ntask = (nqp + QPW - 1) / QPW;
  for(long task = (long)blockIdx.x * WAVES + wave; task < ntask; task += (long)gridDim.x * WAVES)
  {
    const long qp = task * QPW + grp; // (instance, axis) = (qp / 2, qp % 2)
    const bool valid = qp < nqp;
    const bool row = valid && li < N;

    // ---- src/LinearMpcZmp.cpp:54-66: lo/hi of the box on B u
    double px = 0, vx = 0, ax = 0, zl = 0, zh = 0;
    if(valid)
    {
      px = x0[qp * 3 + 0];
      vx = x0[qp * 3 + 1];
      ax = x0[qp * 3 + 2];
    }
    if(row)
    {
      zl = zlim[qp * 2 * N + li];
      zh = zlim[qp * 2 * N + N + li];
    }
    const double fr = a0 * px + a1 * vx + a2 * ax; // (A_seq x0)_i
    const double lo = row ? zl - fr : -kInf;
    const double hi = row ? zh - fr : kInf;
    const double tl = row ? 1e-12 * (1.0 + fabs(lo)) : 0.0;
    const double th = row ? 1e-12 * (1.0 + fabs(hi)) : 0.0;

    int st = CCC_STATUS_SOLVED;
    if(Grp::any(row && lo > hi)) st = CCC_STATUS_INFEASIBLE;

    // ---- tableau T = G (W empty): lane li holds row li = column li of the symmetric G; the diagonal
    //      lives in dg (the in-row copy T[li] is never read by its owner and is patched on publication)
    double T[NP];
#pragma unroll
    for(int j = 0; j < NP; ++j) T[j] = Gs[j * NP + li];
    double dg = Gs[li * NP + li];
    double dgm = dg; // bitwise mirror of the (unreadable) in-row register T[li], see the refinement below

    double z = 0.0, mu = 0.0; // z = (G mu)_li, mu = multiplier of row li
    double dact = 0.0;         // bound row li sits on while it is in W
    bool inW = false;
    int p = 0;                  // entering row (group uniform)
    double psig = 0.0, pd = 0.0; // meaningful in lane p only: side (+1 lower, -1 upper) and bound it moves to
    bool done = !valid || st != CCC_STATUS_SOLVED;
    bool need_select = true;
    int passes = 0;

    for(int round = 0; round < 3; ++round)
    {
    for(;;)
    {
      // -- Goldfarb-Idnani step 1: the most violated row enters
      if(__ballot(need_select && !done) != 0ull)
      {
        const double sl = (lo - z) - tl, sh = (z - hi) - th;
        const double score = (inW || !row) ? -kInf : fmax(sl, sh);
        const double m = Grp::max(score);
        const int cand = Grp::first(score == m);
        if(need_select && !done)
        {
          if(m > 0.0)
          {
            p = cand;
            if(li == cand)
            {
              psig = (sl >= sh) ? 1.0 : -1.0;
              pd = (sl >= sh) ? lo : hi;
            }
          }
          else
            done = true;
        }
      }
      if(__ballot(!done) == 0ull) break;

      // -- lane p publishes row p (= column p of T): search direction for W, Schur complement elsewhere
      if(li == p && !done)
      {
#pragma unroll
        for(int j = 0; j < NP; j += 2) *reinterpret_cast<double2 *>(scr + j) = make_double2(T[j], T[j + 1]);
        scr[p] = dg;
        scr[Scr::kSig] = psig;
      }
      __builtin_amdgcn_wave_barrier();
      const double c = scr[li];
      const double sig = scr[Scr::kSig];

      // -- step length: full step (row p reaches its bound) vs dual ratio test over W, in ONE min
      const double dm = -sig * c;
      const bool blocking = inW && ((mu > 0.0 && dm < 0.0) || (mu < 0.0 && dm > 0.0));
      const bool isp = (li == p);
      const double num = isp ? psig * (pd - z) : -mu;
      const double den = isp ? dg : dm;
      const double ratio = (!done && (isp || blocking)) ? num / den : kInf;
      const double t = Grp::min(ratio);
      int kk = Grp::first(ratio == t);
      if(!done && kk >= LG)
      { // NaN step: numerical breakdown, report instead of spinning
        done = true;
        st = CCC_STATUS_MAX_ITER;
      }
      kk = kk >= LG ? 0 : kk;
      const bool isadd = (kk == p);
      const double s = isadd ? 1.0 : -1.0;
      if(!done)
      {
        if(inW)
          mu = fma(t, dm, mu);
        else
          z = fma(sig * t, c, z);
        if(isp) mu += sig * t;
      }

      // -- pivot row kk (add: p itself, already published; drop: the blocking row): patch the diagonal so that
      //    the generic update writes column kk, and publish 1/pivot
      if(li == kk && !done)
      {
        if(!isadd)
        {
#pragma unroll
          for(int j = 0; j < NP; j += 2) *reinterpret_cast<double2 *>(scr + j) = make_double2(T[j], T[j + 1]);
        }
        scr[kk] = dg - s;
        scr[Scr::kRp] = 1.0 / dg;
      }
      __builtin_amdgcn_wave_barrier();
      const double v = scr[li];
      const double rp = scr[Scr::kRp];
      // T'_ij = T_ij - (v_i rp) v_j ; column kk: v_i - (v_i rp)(v_kk - s) = s v_i rp ; row kk: g = 1 - s rp
      double g = (li == kk) ? (1.0 - s * rp) : v * rp;
      g = done ? 0.0 : g;
#pragma unroll
      for(int j0 = 0; j0 < NP; j0 += 8)
      { // 8 broadcast values in flight at a time: keeps the row + state inside 128 VGPRs (4 waves/SIMD)
        const double2 v0 = scr2[j0 / 2 + 0], v1 = scr2[j0 / 2 + 1], v2 = scr2[j0 / 2 + 2], v3 = scr2[j0 / 2 + 3];
        T[j0 + 0] = fma(-g, v0.x, T[j0 + 0]);
        T[j0 + 1] = fma(-g, v0.y, T[j0 + 1]);
        T[j0 + 2] = fma(-g, v1.x, T[j0 + 2]);
        T[j0 + 3] = fma(-g, v1.y, T[j0 + 3]);
        T[j0 + 4] = fma(-g, v2.x, T[j0 + 4]);
        T[j0 + 5] = fma(-g, v2.y, T[j0 + 5]);
        T[j0 + 6] = fma(-g, v3.x, T[j0 + 6]);
        T[j0 + 7] = fma(-g, v3.y, T[j0 + 7]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if(!done)
      {
        dg = (li == kk) ? -rp : fma(-g, v, dg);
        dgm = fma(-g, v, dgm);
        if(isadd)
        {
          if(isp)
          {
            inW = true;
            z = pd;
            dact = pd;
          }
          need_select = true;
        }
        else
        {
          if(li == kk)
          {
            inW = false;
            mu = 0.0;
          }
          need_select = false;
        }
        if(++passes > maxpass)
        {
          done = true;
          st = CCC_STATUS_MAX_ITER;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }

    // ---- closing iterative refinement against the untouched G (removes the drift of the rank-1 updates):
    //      rho = d_W - (G mu)_W ;  mu_W += G_WW^-1 rho = -T_WW rho ;  z = G mu recomputed; a row the fresh z
    //      shows violated re-opens the iteration (rare: rows that sat within the drift of their bound)
    {
      const bool ok = valid && st == CCC_STATUS_SOLVED;
      const bool act = inW && ok;
      scr[li] = act ? mu : 0.0;
      __builtin_amdgcn_wave_barrier();
      double acc = 0.0;
#pragma unroll
      for(int j = 0; j < NP; j += 2)
      {
        const double2 mb = scr2[j / 2];
        acc = fma(Gs[j * NP + li], mb.x, acc);
        acc = fma(Gs[(j + 1) * NP + li], mb.y, acc);
      }
      const double rho = act ? dact - acc : 0.0;
      __builtin_amdgcn_wave_barrier();
      scr[li] = rho;
      __builtin_amdgcn_wave_barrier();
      double tr = 0.0;
#pragma unroll
      for(int j = 0; j < NP; j += 2)
      {
        const double2 rb = scr2[j / 2];
        tr = fma(T[j], rb.x, tr);
        tr = fma(T[j + 1], rb.y, tr);
      }
      tr = fma(dg - dgm, rho, tr); // replace the stale in-row diagonal by the true one
      if(act) mu -= tr;
      __builtin_amdgcn_wave_barrier();
      scr[li] = act ? mu : 0.0;
      __builtin_amdgcn_wave_barrier();
      acc = 0.0;
#pragma unroll
      for(int j = 0; j < NP; j += 2)
      {
        const double2 mb = scr2[j / 2];
        acc = fma(Gs[j * NP + li], mb.x, acc);
        acc = fma(Gs[(j + 1) * NP + li], mb.y, acc);
      }
      if(ok) z = inW ? dact : acc;
      const double sl = (lo - z) - tl, sh = (z - hi) - th;
      const bool reopen = Grp::any(ok && row && !inW && fmax(sl, sh) > 0.0);
      done = !reopen;
      need_select = true;
      __builtin_amdgcn_wave_barrier();
      if(__ballot(reopen) == 0ull) break;
    }
    }

    // ---- outputs: jerk[0] = (B' mu)_0, then src/LinearMpcZmp.cpp:72-78
    const double u0 = Grp::sum(row ? bi * mu : 0.0);
    if(valid && li == 0)
    {
      const double cdt = control_dt < 0 ? P.dt : control_dt;
      const double com_acc = ax + cdt * u0;
      const double com_pos = px + cdt * vx + 0.5 * (cdt * cdt) * ax;
      double zv = com_pos + P.c2 * com_acc;
      zv = zv < zl ? zl : (zh < zv ? zh : zv);
      zmp[qp] = zv;
      if(status) status[qp] = (passes << 8) | st;
    }
    if(jerk)
    {
      // u_j = sum_{i >= j} b[i - j] mu_i   (mu is zero outside W)
      scr[li] = row ? mu : 0.0;
      __builtin_amdgcn_wave_barrier();
      double uj = 0.0;
#pragma unroll 4
      for(int i = 0; i < NP; ++i)
      {
        const double m = scr[i];
        const int dlt = i - li;
        const double bv = bs[dlt >= 0 ? dlt : 0];
        uj = (dlt >= 0) ? fma(bv, m, uj) : uj;
      }
      if(row) jerk[qp * N + li] = uj;
      __builtin_amdgcn_wave_barrier();
    }
  }
}
} // namespace ccc_amd

// =============================================================================================
// host side: model construction and the C-ABI
// =============================================================================================
using namespace ccc_amd;

struct ccc_zmp
{
  int device = 0;
  int N = 0;
  int NP = 0; // padded size = lanes per QP (32 or 64)
  double com_height = 0, horizon_duration = 0, horizon_dt = 0, c2 = 0;
  std::vector<double> A_seq, B_seq; // host copies, N x 3 and N x N
  double *dG = nullptr, *dA = nullptr, *db = nullptr;
  int num_cu = 0;
  // staging for the host-pointer entry point
  int64_t cap = 0;
  double *h_in = nullptr, *h_out = nullptr; // pinned
  double *d_in = nullptr, *d_out = nullptr;
  int32_t *h_status = nullptr, *d_status = nullptr;
  hipStream_t stream = nullptr;
};

namespace
{
constexpr double kG = 9.80665; // include/CCC/Constants.h:10

// Closed form of src/CommonModels.cpp:8-17 + StateSpaceModel.h:164-216 + InvariantSequentialExtension.h:103-181
// for the jerk-input CoM-ZMP model: the continuous A is nilpotent, so exp([[A,B],[0,0]] dt) terminates:
//   Ad^n = [[1, n dt, (n dt)^2/2],[0, 1, n dt],[0, 0, 1]],  Bd = [dt^3/6, dt^2/2, dt]',  C = [1, 0, -h/g]
//   A_seq[i] = C Ad^(i+1),   B_seq[i][j] = C Ad^(i-j) Bd  (j <= i)
void build_model(ccc_zmp * h)
{
  const int N = h->N;
  const double dt = h->horizon_dt, c2 = -1 * h->com_height / kG;
  h->c2 = c2;
  h->A_seq.assign((size_t)N * 3, 0.0);
  h->B_seq.assign((size_t)N * N, 0.0);
  std::vector<double> b(N);
  const double Bd[3] = {dt * dt * dt / 6.0, dt * dt / 2.0, dt};
  for(int n = 0; n < N; n++)
  {
    const double nd = n * dt;
    // C Ad^n Bd = Bd0 + nd Bd1 + nd^2/2 Bd2 + c2 Bd2
    b[n] = Bd[0] + nd * Bd[1] + 0.5 * nd * nd * Bd[2] + c2 * Bd[2];
  }
  for(int i = 0; i < N; i++)
  {
    const double nd = (i + 1) * dt;
    h->A_seq[i * 3 + 0] = 1.0;
    h->A_seq[i * 3 + 1] = nd;
    h->A_seq[i * 3 + 2] = 0.5 * nd * nd + c2;
    for(int j = 0; j <= i; j++) h->B_seq[(size_t)i * N + j] = b[i - j];
  }
}

int upload_model(ccc_zmp * h)
{
  const int N = h->N, NP = h->NP;
  std::vector<double> G((size_t)NP * NP, 0.0), A((size_t)NP * 3, 0.0), b(NP, 0.0);
  for(int i = 0; i < N; i++)
  {
    for(int j = 0; j < N; j++)
    {
      long double s = 0;
      for(int k = 0; k < N; k++) s += (long double)h->B_seq[(size_t)i * N + k] * h->B_seq[(size_t)j * N + k];
      G[(size_t)i * NP + j] = (double)s;
    }
    for(int k = 0; k < 3; k++) A[i * 3 + k] = h->A_seq[i * 3 + k];
    b[i] = h->B_seq[(size_t)i * N + 0];
  }
  for(int i = N; i < NP; i++) G[(size_t)i * NP + i] = 1.0;
  CCC_HIP_CHECK(hipMalloc(&h->dG, G.size() * sizeof(double)));
  CCC_HIP_CHECK(hipMalloc(&h->dA, A.size() * sizeof(double)));
  CCC_HIP_CHECK(hipMalloc(&h->db, b.size() * sizeof(double)));
  CCC_HIP_CHECK(hipMemcpy(h->dG, G.data(), G.size() * sizeof(double), hipMemcpyHostToDevice));
  CCC_HIP_CHECK(hipMemcpy(h->dA, A.data(), A.size() * sizeof(double), hipMemcpyHostToDevice));
  CCC_HIP_CHECK(hipMemcpy(h->db, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice));
  return CCC_OK;
}

template<int LG, int WAVES>
int launch(ccc_zmp * h, int64_t n, const double * x0, const double * zlim, double control_dt, double * zmp,
           double * jerk, int32_t * status, hipStream_t stream)
{
  constexpr int QPW = 64 / LG;
  const size_t lds = ((size_t)LG * LG + LG + (size_t)WAVES * QPW * ZmpScratch<LG>::kSize) * sizeof(double);
  static bool attr_set = false;
  if(!attr_set)
  {
    CCC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&zmp_plan_kernel<LG, WAVES>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int64_t nqp = 2 * n;
  const int64_t ntask = (nqp + QPW - 1) / QPW;
  const int64_t want = (ntask + WAVES - 1) / WAVES;
  // a few blocks per resident slot: the hardware dispatcher then evens out the data-dependent pivot counts
  const int64_t resident = (int64_t)h->num_cu * (16 / WAVES);
  const int grid = (int)std::min<int64_t>(want, resident * 8);
  ZmpDev P{h->N, h->dG, h->dA, h->db, h->c2, h->horizon_dt};
  hipLaunchKernelGGL((zmp_plan_kernel<LG, WAVES>), dim3(grid), dim3(WAVES * 64), lds, stream, P, (long)nqp, x0, zlim,
                     control_dt, zmp, jerk, status);
  CCC_HIP_CHECK(hipGetLastError());
  return CCC_OK;
}
} // namespace

extern "C" int ccc_zmp_create(double com_height, double horizon_duration, double horizon_dt, int device,
                              ccc_zmp_t ** out)
{
  if(!out) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_create: out is NULL");
  *out = nullptr;
  if(!(com_height > 0) || !(horizon_duration > 0) || !(horizon_dt > 0))
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_create: com_height, horizon_duration, horizon_dt must be > 0");
  const int N = (int)std::ceil(horizon_duration / horizon_dt); // src/LinearMpcZmp.cpp:13
  if(N > 64)
    return fail(CCC_ERR_UNSUPPORTED, "ccc_zmp_create: horizon_steps %d > 64 is not built into this library yet", N);
  int rc = select_device(device);
  if(rc != CCC_OK) return rc;
  ccc_zmp * h = new ccc_zmp();
  h->device = device;
  h->N = N;
  h->NP = N <= 32 ? 32 : 64;
  h->com_height = com_height;
  h->horizon_duration = horizon_duration;
  h->horizon_dt = horizon_dt;
  build_model(h);
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if(e != hipSuccess)
  {
    delete h;
    return fail(CCC_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  }
  h->num_cu = prop.multiProcessorCount;
  rc = upload_model(h);
  if(rc != CCC_OK)
  {
    ccc_zmp_destroy(h);
    return rc;
  }
  *out = h;
  return CCC_OK;
}

extern "C" void ccc_zmp_destroy(ccc_zmp_t * h)
{
  if(!h) return;
  (void)hipSetDevice(h->device);
  if(h->dG) (void)hipFree(h->dG);
  if(h->dA) (void)hipFree(h->dA);
  if(h->db) (void)hipFree(h->db);
  if(h->d_in) (void)hipFree(h->d_in);
  if(h->d_out) (void)hipFree(h->d_out);
  if(h->d_status) (void)hipFree(h->d_status);
  if(h->h_in) (void)hipHostFree(h->h_in);
  if(h->h_out) (void)hipHostFree(h->h_out);
  if(h->h_status) (void)hipHostFree(h->h_status);
  if(h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int ccc_zmp_horizon_steps(const ccc_zmp_t * h)
{
  return h ? h->N : -1;
}

extern "C" int ccc_zmp_get_seq(const ccc_zmp_t * h, double * A_seq, double * B_seq)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_get_seq: NULL handle");
  if(A_seq) std::memcpy(A_seq, h->A_seq.data(), h->A_seq.size() * sizeof(double));
  if(B_seq) std::memcpy(B_seq, h->B_seq.data(), h->B_seq.size() * sizeof(double));
  return CCC_OK;
}

extern "C" int ccc_zmp_plan_batch_device(ccc_zmp_t * h, int64_t n, const double * x0, const double * zlim,
                                         double control_dt, double * zmp, double * jerk, int32_t * status,
                                         void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch_device: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch_device: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!x0 || !zlim || !zmp) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch_device: NULL x0/zlim/zmp");
  CCC_HIP_CHECK(hipSetDevice(h->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if(h->NP == 32) return launch<32, 4>(h, n, x0, zlim, control_dt, zmp, jerk, status, s);
  return launch<64, 2>(h, n, x0, zlim, control_dt, zmp, jerk, status, s);
}

static int ensure_staging(ccc_zmp * h, int64_t n)
{
  if(n <= h->cap) return CCC_OK;
  const int N = h->N;
  if(h->d_in) (void)hipFree(h->d_in);
  if(h->d_out) (void)hipFree(h->d_out);
  if(h->d_status) (void)hipFree(h->d_status);
  if(h->h_in) (void)hipHostFree(h->h_in);
  if(h->h_out) (void)hipHostFree(h->h_out);
  if(h->h_status) (void)hipHostFree(h->h_status);
  h->d_in = h->d_out = h->h_in = h->h_out = nullptr;
  h->d_status = h->h_status = nullptr;
  h->cap = 0;
  const size_t in_elems = (size_t)n * (6 + 4 * N), out_elems = (size_t)n * (2 + 2 * N);
  CCC_HIP_CHECK(hipMalloc(&h->d_in, in_elems * sizeof(double)));
  CCC_HIP_CHECK(hipMalloc(&h->d_out, out_elems * sizeof(double)));
  CCC_HIP_CHECK(hipMalloc(&h->d_status, (size_t)n * 2 * sizeof(int32_t)));
  CCC_HIP_CHECK(hipHostMalloc(&h->h_in, in_elems * sizeof(double), hipHostMallocDefault));
  CCC_HIP_CHECK(hipHostMalloc(&h->h_out, out_elems * sizeof(double), hipHostMallocDefault));
  CCC_HIP_CHECK(hipHostMalloc(&h->h_status, (size_t)n * 2 * sizeof(int32_t), hipHostMallocDefault));
  if(!h->stream) CCC_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  h->cap = n;
  return CCC_OK;
}

extern "C" int ccc_zmp_plan_batch(ccc_zmp_t * h, int64_t n, const double * x0, const double * zlim, double control_dt,
                                  double * zmp, double * jerk, int32_t * status)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!x0 || !zlim || !zmp) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_plan_batch: NULL x0/zlim/zmp");
  CCC_HIP_CHECK(hipSetDevice(h->device));
  int rc = ensure_staging(h, n);
  if(rc != CCC_OK) return rc;
  const int N = h->N;
  const size_t nx = (size_t)n * 6, nl = (size_t)n * 4 * N, nz = (size_t)n * 2, nj = (size_t)n * 2 * N;
  std::memcpy(h->h_in, x0, nx * sizeof(double));
  std::memcpy(h->h_in + nx, zlim, nl * sizeof(double));
  CCC_HIP_CHECK(hipMemcpyAsync(h->d_in, h->h_in, (nx + nl) * sizeof(double), hipMemcpyHostToDevice, h->stream));
  rc = ccc_zmp_plan_batch_device(h, n, h->d_in, h->d_in + nx, control_dt, h->d_out, jerk ? h->d_out + nz : nullptr,
                                 h->d_status, h->stream);
  if(rc != CCC_OK) return rc;
  CCC_HIP_CHECK(hipMemcpyAsync(h->h_out, h->d_out, (nz + (jerk ? nj : 0)) * sizeof(double), hipMemcpyDeviceToHost,
                               h->stream));
  CCC_HIP_CHECK(hipMemcpyAsync(h->h_status, h->d_status, (size_t)n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost,
                               h->stream));
  CCC_HIP_CHECK(hipStreamSynchronize(h->stream));
  std::memcpy(zmp, h->h_out, nz * sizeof(double));
  if(jerk) std::memcpy(jerk, h->h_out + nz, nj * sizeof(double));
  if(status) std::memcpy(status, h->h_status, (size_t)n * 2 * sizeof(int32_t));
  return CCC_OK;
}
