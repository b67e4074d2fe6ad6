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
// factorisation it keeps the columns {T[:,w] : w in W} of the symmetric sweep tableau of G swept on W
// (T_WW = -G_WW^-1, T_iW = G_iW G_WW^-1) -- one lane per row, columns in LDS -- from which column p of
// the tableau (search direction + Schur complement) costs |W| FMAs per lane, and adding/dropping a
// row is one rank-1 update of |W| columns.  A closing iterative-refinement step against the
// untouched G removes the drift of the incremental updates.
//
// Mapping: one lane per horizon step; N <= 32: the two axes of one instance are the two 32-lane
// halves of ONE wavefront (one planOnce() per wavefront); 32 < N <= 64: one axis per wavefront.
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

template<int LG, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void zmp_plan_kernel(ZmpDev P, long nqp, const double * __restrict__ x0,
                                                             const double * __restrict__ zlim, double control_dt,
                                                             double * __restrict__ zmp, double * __restrict__ jerk,
                                                             int * __restrict__ status)
{
  using Grp = WaveGroup<LG>;
  constexpr int NP = LG;
  constexpr int QPW = 64 / LG; // QPs per wavefront
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double * Gs = smem;           // [NP][NP]
  double * bs = smem + NP * NP; // [NP]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & (LG - 1), grp = lane / LG;
  double * Yw = smem + NP * NP + NP + wave * (LG * 64); // this wavefront's tableau columns: Y[slot][lane]

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

    // ---- dual active-set state (per lane: row li of the KKT system)
    double z = 0.0, mu = 0.0, dact = 0.0;
    bool inW = false;
    int slot_idx = 0; // lane s: row stored in slot s
    int myslot = 0;   // lane i in W: slot holding column i
    int q = 0;        // |W| (group uniform)
    int p = 0;
    double sig = 0.0, dsel = 0.0;
    bool done = !valid || st != CCC_STATUS_SOLVED;
    bool need_select = true;
    int passes = 0;

    for(int round = 0; round < 3; ++round)
    {
      for(;;)
      {
        // -- select the most violated row (Goldfarb-Idnani step 1)
        {
          const double sl = (lo - z) - tl, sh = (z - hi) - th;
          double score = (inW || !row) ? -kInf : fmax(sl, sh);
          int cand = li;
          Grp::argmax(score, cand);
          const double my_sig = (sl >= sh) ? 1.0 : -1.0;
          const double my_d = (sl >= sh) ? lo : hi;
          const double csig = Grp::bcast(my_sig, cand);
          const double cd = Grp::bcast(my_d, cand);
          if(need_select && !done)
          {
            if(score > 0.0)
            {
              p = cand;
              sig = csig;
              dsel = cd;
            }
            else
              done = true;
          }
        }
        if(__ballot(!done) == 0ull) break;

        const int qmax = Grp::wave_max_of_group_uniform(done ? 0 : q);

        // -- column p of the swept tableau: c_W = G_WW^-1 G_Wp, c_i = Schur complement column otherwise
        double c = (inW || !row) ? 0.0 : Gs[p * NP + li];
        {
          const double gws = (li < q) ? Gs[slot_idx * NP + p] : 0.0;
          for(int s = 0; s < qmax; ++s)
          {
            const double gw = Grp::bcast(gws, s);
            const double y = Yw[s * 64 + lane];
            c = (s < q) ? fma(-y, gw, c) : c;
          }
        }
        const double delta = Grp::bcast(c, p);
        const double zp = Grp::bcast(z, p);
        // -- step lengths: full step t2 (row p reaches its bound), dual ratio test t1
        const double t2 = sig * (dsel - zp) / delta;
        const double dm = -sig * c;
        const bool blocking = inW && ((mu > 0.0 && dm < 0.0) || (mu < 0.0 && dm > 0.0));
        double t1 = blocking ? -mu / dm : kInf;
        int k = li;
        Grp::argmin(t1, k);
        const bool drop = t1 < t2;
        const double t = drop ? t1 : t2;
        if(!done)
        {
          if(inW)
            mu = fma(t, dm, mu);
          else
            z = fma(sig * t, c, z);
          if(li == p) mu += sig * t;
        }
        // -- pivot: sweep p in (add) or sweep k out (drop); both are one rank-1 update of the stored columns
        const int pi = drop ? k : p;
        int sk = Grp::bcast(myslot, k);
        sk = drop ? sk : 0;
        const double v = drop ? Yw[sk * 64 + lane] : c;
        const double rp = 1.0 / Grp::bcast(v, pi);
        const double vw = __shfl(v, slot_idx, LG); // lane s: v at the row stored in slot s
        for(int s = 0; s < qmax; ++s)
        {
          const double f = Grp::bcast(vw, s) * rp;
          const double y = Yw[s * 64 + lane];
          const double yn = (li == pi) ? (drop ? -f : f) : fma(-v, f, y);
          if(!done && s < q && !(drop && s == sk)) Yw[s * 64 + lane] = yn;
        }
        const int last = q > 0 ? q - 1 : 0;
        const int wl = Grp::bcast(slot_idx, last);
        if(!done)
        {
          if(!drop)
          {
            Yw[q * 64 + lane] = (li == p) ? -rp : c * rp;
            if(li == q) slot_idx = p;
            if(li == p)
            {
              myslot = q;
              inW = true;
              z = dsel;
              dact = dsel;
            }
            q += 1;
            need_select = true;
          }
          else
          {
            if(sk != last)
            {
              Yw[sk * 64 + lane] = Yw[last * 64 + lane];
              if(li == sk) slot_idx = wl;
              if(li == wl) myslot = sk;
            }
            if(li == k)
            {
              inW = false;
              mu = 0.0;
            }
            q -= 1;
            need_select = false;
          }
          if(++passes > maxpass)
          {
            done = true;
            st = CCC_STATUS_MAX_ITER;
          }
        }
      }

      // -- closing refinement against the original G: rho = d_W - (G mu)_W, mu_W += G_WW^-1 rho,
      //    then z = G mu recomputed from scratch for the final optimality check
      const bool ok = valid && st == CCC_STATUS_SOLVED;
      const int qmax = Grp::wave_max_of_group_uniform(ok ? q : 0);
      double acc = 0.0;
      {
        const double muw = __shfl(mu, slot_idx, LG);
        for(int s = 0; s < qmax; ++s)
        {
          const int w = Grp::bcast(slot_idx, s);
          const double m = Grp::bcast(muw, s);
          const double gv = Gs[w * NP + li];
          acc = (s < q) ? fma(gv, m, acc) : acc;
        }
      }
      const double rho = (inW && ok) ? dact - acc : 0.0;
      {
        const double rhow = __shfl(rho, slot_idx, LG);
        double dmu = 0.0;
        for(int s = 0; s < qmax; ++s)
        {
          const double r = Grp::bcast(rhow, s);
          const double y = Yw[s * 64 + lane];
          dmu = (s < q) ? fma(-y, r, dmu) : dmu;
        }
        if(inW && ok) mu += dmu;
      }
      acc = 0.0;
      {
        const double muw = __shfl(mu, slot_idx, LG);
        for(int s = 0; s < qmax; ++s)
        {
          const int w = Grp::bcast(slot_idx, s);
          const double m = Grp::bcast(muw, s);
          const double gv = Gs[w * NP + li];
          acc = (s < q) ? fma(gv, m, acc) : acc;
        }
      }
      if(ok) z = inW ? dact : acc;
      // a row that the refreshed z shows violated re-opens the iteration (rare: only rows that sat
      // within the drift of the incremental updates of their bound)
      const double sl = (lo - z) - tl, sh = (z - hi) - th;
      const bool viol = ok && row && !inW && fmax(sl, sh) > 0.0;
      const bool reopen = Grp::any(viol);
      done = !reopen;
      need_select = true;
      if(__ballot(reopen) == 0ull) break;
    }

    // ---- outputs: jerk[0] = (B' mu)_0, src/LinearMpcZmp.cpp:69-78
    const double u0 = Grp::sum(row ? bi * mu : 0.0);
    const double zmin0 = Grp::bcast(zl, 0), zmax0 = Grp::bcast(zh, 0);
    if(valid && li == 0)
    {
      const double cdt = control_dt < 0 ? P.dt : control_dt;
      const double com_acc = ax + cdt * u0;
      const double com_pos = px + cdt * vx + 0.5 * (cdt * cdt) * ax;
      double zv = com_pos + P.c2 * com_acc;
      zv = zv < zmin0 ? zmin0 : (zmax0 < zv ? zmax0 : zv);
      zmp[qp] = zv;
      if(status) status[qp] = (passes << 8) | st;
    }
    if(jerk)
    {
      // u_j = sum_{w in W, w >= j} b[w - j] mu_w
      const int qmax = Grp::wave_max_of_group_uniform(valid ? q : 0);
      const double muw = __shfl(mu, slot_idx, LG);
      double uj = 0.0;
      for(int s = 0; s < qmax; ++s)
      {
        const int w = Grp::bcast(slot_idx, s);
        const double m = Grp::bcast(muw, s);
        const int dlt = w - li;
        const double bv = bs[dlt >= 0 ? dlt : 0];
        uj = (s < q && dlt >= 0) ? fma(bv, m, uj) : uj;
      }
      if(row) jerk[qp * N + li] = uj;
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
  const size_t lds = ((size_t)LG * LG + LG + (size_t)WAVES * LG * 64) * sizeof(double);
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
  // resident blocks: LDS-bound (160 KiB per CU)
  const int per_cu = (int)std::max<size_t>(1, (160 * 1024) / lds);
  const int64_t resident = (int64_t)h->num_cu * per_cu;
  const int grid = (int)std::min<int64_t>(want, resident * 4);
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
