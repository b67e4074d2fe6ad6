// centroidal_loop.hip -- the reference's closed-loop tests of the force-scale planners, for n instances on the device
// (plan -> ForceColl::calcTotalWrench -> CentroidalSim::update -> plan ...):
//   tests/src/TestDdpCentroidal.cpp:96-150      DdpCentroidal      (warm start, max_iter = 1 after the first cycle)
//   tests/src/TestDdpSingleRigidBody.cpp:103-170 DdpSingleRigidBody (+ the ZYX / XYZ reversal of the orientation)
//   tests/src/TestLinearMpcXY.cpp:98-132        LinearMpcXY
//   tests/src/SimModels.h:233-340               CentroidalSim (exact ZOH of the 18-state integrator chain, addDisturb)
// The callbacks of the tests (motion_param_func / ref_data_func: piecewise constant in time) become a per-instance
// CONTACT TIMELINE sampled on the device; the planners are the kernels of csrc/ddp.hip and csrc/xy.hip, called through
// the C-ABI of this library, the wrench is that of csrc/wrench.hip.  Everything stays in HBM between the cycles.
#include "common.h"

#include <cstring>
#include <vector>

#if defined(__clang__)
#  pragma clang fp contract(off) // the simulator matches the numpy fixture (fixtures_ddp.CentroidalSim) operation by operation
#endif

namespace ccc_amd
{
constexpr double kLoopG = 9.80665;

struct Timeline
{
  int K, C, M; // M: ridge slots per contact entry = the planner handle's max_ridges
  const double * seg_end;
  const int * seg_contact;
  const double * seg_ref;
  const int * contact_dim;
  const double * contact_vertex;
  const double * contact_ridge;
  double eps;
};

__device__ __forceinline__ int segment_at(const Timeline & T, long k, double t)
{
  const double * e = T.seg_end + k * T.K;
  int s = 0;
  while(s < T.K - 1 && !(t < e[s])) s++; // the first segment whose end lies beyond t (the last one never ends)
  return s;
}

// DDP planners: step_phase [n][N], ref_pos / ref_ori [n][N+1][3] at t + i dt (+ eps); the phase tables are the contact
// tables themselves.  One thread per (instance, sample).
__global__ void sample_ddp_kernel(Timeline T, long n, int N, double t, double dt, int * step_phase, double * ref_pos,
                                  double * ref_ori)
{
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(id >= n * (N + 1)) return;
  const long k = id / (N + 1);
  const int i = (int)(id % (N + 1));
  const int s = segment_at(T, k, t + i * dt + T.eps);
  const double * r = T.seg_ref + (k * T.K + s) * 6;
  for(int a = 0; a < 3; a++)
  {
    ref_pos[(k * (N + 1) + i) * 3 + a] = r[a];
    if(ref_ori) ref_ori[(k * (N + 1) + i) * 3 + a] = r[3 + a];
  }
  if(i < N) step_phase[k * N + i] = min(max(T.seg_contact[k * T.K + s], 0), T.C - 1); // (clamped to the contact table)
}

// LinearMpcXY: the per-step arrays of ccc_xy_plan_batch_device
__global__ void sample_xy_kernel(Timeline T, long n, int N, double t, double dt, double mass, double com_z,
                                 double total_force_z, int * dim, double * vertex, double * ridge, double * cz,
                                 double * fz, double * ref_out)
{
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(id >= n * N) return;
  const long k = id / N;
  const int s = segment_at(T, k, t + (int)(id % N) * dt + T.eps);
  const int c = min(max(T.seg_contact[k * T.K + s], 0), T.C - 1); // (clamped to the contact table)
  const int m = T.contact_dim[k * T.C + c];
  dim[id] = m;
  for(int e = 0; e < T.M * 3; e++)
  {
    vertex[id * T.M * 3 + e] = T.contact_vertex[(k * T.C + c) * T.M * 3 + e];
    ridge[id * T.M * 3 + e] = T.contact_ridge[(k * T.C + c) * T.M * 3 + e];
  }
  cz[id] = com_z;
  fz[id] = total_force_z;
  const double * r = T.seg_ref + (k * T.K + s) * 6;
  // RefData::toOutput(mass), src/LinearMpcXY.cpp:33-38 with vel = angular momentum = 0
  ref_out[id * 6 + 0] = mass * r[0];
  ref_out[id * 6 + 1] = 0.0;
  ref_out[id * 6 + 2] = mass * r[1];
  ref_out[id * 6 + 3] = 0.0;
  ref_out[id * 6 + 4] = 0.0;
  ref_out[id * 6 + 5] = 0.0;
}

// planner state from the simulator state [pos 3, ori 3 (x, y, z), vel 3, ang_vel 3, lin_mom 3, ang_mom 3]
//   model 0 (DdpCentroidal::InitialParam::toState, src/DdpCentroidal.cpp:186-191): [pos, mass vel, ang_mom]
//   model 1 (DdpSingleRigidBody, TestDdpSingleRigidBody.cpp:110-115): [pos, ori reversed (Z, Y, X), vel, ang_vel]
//   model 2 (LinearMpcXY::InitialParam::toState, src/LinearMpcXY.cpp:26-31): [m px, m vx, m py, m vy, Lx, Ly]
__global__ void planner_state_kernel(long n, int model, double mass, const double * sim, double * x0)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const double * s = sim + k * 18;
  if(model == 0)
  {
    double * x = x0 + k * 9;
    for(int a = 0; a < 3; a++)
    {
      x[a] = s[a];
      x[3 + a] = mass * s[6 + a];
      x[6 + a] = s[15 + a];
    }
  }
  else if(model == 1)
  {
    double * x = x0 + k * 12;
    for(int a = 0; a < 3; a++)
    {
      x[a] = s[a];
      x[3 + a] = s[3 + 2 - a];
      x[6 + a] = s[6 + a];
      x[9 + a] = s[9 + a];
    }
  }
  else
  {
    double * x = x0 + k * 6;
    x[0] = mass * s[0];
    x[1] = mass * s[6];
    x[2] = mass * s[1];
    x[3] = mass * s[7];
    x[4] = s[15];
    x[5] = s[16];
  }
}

// warm start of the next cycle (TestDdpCentroidal.cpp:102-114): the previous input sequence UNSHIFTED, zeroed where the
// input dimension of the step changed; dims_prev is updated.  One thread per (instance, step).
__global__ void warm_start_kernel(long n, int N, int P, int M, const int * phase_dim, const int * step_phase,
                                  int * dims_prev, double * u)
{
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(id >= n * N) return;
  const long k = id / N;
  const int m = phase_dim[k * P + step_phase[id]];
  if(dims_prev[id] != m)
    for(int r = 0; r < M; r++) u[id * M + r] = 0.0;
  dims_prev[id] = m;
}

// total wrench of the planned force scales of the CURRENT contact list about the CoM, then CentroidalSim::update, the
// disturbance, and the statistics of the reference tests' assertions.  One thread per instance.
struct SimArgs
{
  long n;
  int model;        // 0 / 1: contact of step 0 from the phase tables; 2: from the per-step XY arrays
  int N, P, M; // M: ridge stride of the contact arrays
  const int * phase_dim;
  const double * phase_vertex;
  const double * phase_ridge;
  const int * step_phase; // DDP: [n][N];  XY: dim [n][N]
  const double * scales;
  long scale_stride;
  double mass, dt;
  const double * inertia; // [n][3] diagonal moment of inertia of the simulator
  int kick;               // add the disturbance after this update
  double kick_lin[3];
  const double * ref_now; // [n][6] reference (pos, ori ZYX) at the cycle's time, for the statistics
  double * sim;
  double * stats;         // [n][8]: max over the cycles of |pos - ref|, |ori - ref_ori (unreversed, as the test)|, |vel|,
                          //          |ang_vel|, |ang_mom|; [5] cycles whose warm start the guard replaced; [6..7] free
  const int * plan_status; // optional [n]: the DDP status words of this cycle's plan (CCC_DDP_STATUS_WARM_REPLACED)
  double * log;           // optional [n][9] of this cycle: pos, force, moment
};

__global__ void sim_step_kernel(SimArgs A)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= A.n) return;
  double * s = A.sim + k * 18;
  int m;
  const double *V, *R;
  if(A.model == 2)
  {
    m = A.step_phase[k * A.N];
    V = A.phase_vertex + k * A.N * A.M * 3;
    R = A.phase_ridge + k * A.N * A.M * 3;
  }
  else
  {
    const int ph = A.step_phase[k * A.N];
    m = A.phase_dim[k * A.P + ph];
    V = A.phase_vertex + (k * A.P + ph) * A.M * 3;
    R = A.phase_ridge + (k * A.P + ph) * A.M * 3;
  }
  // ForceColl::calcTotalWrench(contact_list, scales, sim.state_.pos.linear())
  double f[3] = {0, 0, 0}, tq[3] = {0, 0, 0};
  for(int r = 0; r < m; r++)
  {
    const double sc = A.scales[k * A.scale_stride + r];
    const double px = V[r * 3] - s[0], py = V[r * 3 + 1] - s[1], pz = V[r * 3 + 2] - s[2];
    const double * d = R + r * 3;
    f[0] += sc * d[0];
    f[1] += sc * d[1];
    f[2] += sc * d[2];
    tq[0] += sc * (py * d[2] - pz * d[1]);
    tq[1] += sc * (pz * d[0] - px * d[2]);
    tq[2] += sc * (px * d[1] - py * d[0]);
  }
  // statistics BEFORE the update (the test checks the state it planned from, TestDdpCentroidal.cpp:133-135)
  if(A.stats)
  {
    const double * rf = A.ref_now + k * 6;
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
    for(int a = 0; a < 3; a++)
    {
      e0 += (s[a] - rf[a]) * (s[a] - rf[a]);
      e1 += (s[3 + a] - rf[3 + a]) * (s[3 + a] - rf[3 + a]);
      e2 += s[6 + a] * s[6 + a];
      e3 += s[9 + a] * s[9 + a];
      e4 += s[15 + a] * s[15 + a];
    }
    double * st = A.stats + k * 8;
    st[0] = fmax(st[0], sqrt(e0));
    st[1] = fmax(st[1], sqrt(e1));
    st[2] = fmax(st[2], sqrt(e2));
    st[3] = fmax(st[3], sqrt(e3));
    st[4] = fmax(st[4], sqrt(e4));
    if(A.plan_status)
    {
      const int w = A.plan_status[k];
      if(CCC_DDP_STATUS_WARM_REPLACED(w)) st[5] += 1.0;
    }
  }
  if(A.log)
  {
    double * lg = A.log + k * 9;
    for(int a = 0; a < 3; a++)
    {
      lg[a] = s[a];
      lg[3 + a] = f[a];
      lg[6 + a] = tq[a];
    }
  }
  // CentroidalSim::update (SimModels.h:296-340): exact ZOH of the integrator chain, the operations of
  // fixtures_ddp.CentroidalSim.update in the same order
  const double dt = A.dt;
  const double * I = A.inertia + k * 3;
  for(int a = 0; a < 3; a++)
  {
    const double g = a == 2 ? -kLoopG : 0.0;
    const double acc = f[a] / A.mass + g;
    const double al = tq[a] / I[a];
    s[a] = s[a] + s[6 + a] * dt + 0.5 * acc * dt * dt;
    s[3 + a] = s[3 + a] + s[9 + a] * dt + 0.5 * al * dt * dt;
    s[6 + a] = s[6 + a] + acc * dt;
    s[9 + a] = s[9 + a] + al * dt;
    const double gm = a == 2 ? -A.mass * kLoopG : 0.0;
    s[12 + a] = s[12 + a] + (f[a] + gm) * dt;
    s[15 + a] = s[15 + a] + tq[a] * dt;
  }
  if(A.kick)
    for(int a = 0; a < 3; a++) s[6 + a] = s[6 + a] + A.kick_lin[a]; // addDisturb: the linear velocity (SimModels.h:326-330)
}

// reference (pos, ori) of the instance's timeline at time t, for the statistics
__global__ void ref_now_kernel(Timeline T, long n, double t, double * ref_now)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const int s = segment_at(T, k, t + T.eps);
  for(int a = 0; a < 6; a++) ref_now[k * 6 + a] = T.seg_ref[(k * T.K + s) * 6 + a];
}
} // namespace ccc_amd

using namespace ccc_amd;

namespace
{
struct DevBuf
{
  std::vector<void *> ptrs;
  template<class T>
  int get(T ** p, size_t count)
  {
    void * q = nullptr;
    if(hipMalloc(&q, count * sizeof(T)) != hipSuccess) return fail(CCC_ERR_HIP, "closed loop: hipMalloc of %zu bytes failed", count * sizeof(T));
    ptrs.push_back(q);
    *p = static_cast<T *>(q);
    return CCC_OK;
  }
  ~DevBuf()
  {
    for(void * q : ptrs) (void)hipFree(q);
  }
};

int check_timeline(const ccc_contact_timeline_t * tl, const char * who)
{
  if(!tl || tl->K <= 0 || tl->C <= 0 || !tl->seg_end || !tl->seg_contact || !tl->seg_ref || !tl->contact_dim
     || !tl->contact_vertex || !tl->contact_ridge)
    return fail(CCC_ERR_INVALID_ARGUMENT, "%s: incomplete contact timeline", who);
  return CCC_OK;
}

Timeline to_dev(const ccc_contact_timeline_t * tl, int M)
{
  return Timeline{tl->K, tl->C, M, tl->seg_end, tl->seg_contact, tl->seg_ref, tl->contact_dim, tl->contact_vertex,
                  tl->contact_ridge, tl->time_eps};
}

inline unsigned blocks(long items)
{
  return (unsigned)((items + 255) / 256);
}
} // namespace

extern "C" int ccc_ddp_closed_loop_device(ccc_ddp_t * h, int64_t n, const ccc_contact_timeline_t * tl,
                                          const double * inertia_diag, double * sim_state, double t0, double sim_dt,
                                          int cycles, int first_max_iter, int warm_max_iter, int n_disturb,
                                          const double * disturb_times, const double * disturb_lin, double * stats,
                                          double * log, double * t_end, void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_closed_loop_device: NULL handle");
  if(n <= 0 || cycles < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_closed_loop_device: n <= 0 or cycles < 0");
  if(int rc = check_timeline(tl, "ccc_ddp_closed_loop_device")) return rc;
  if(!inertia_diag || !sim_state) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_closed_loop_device: NULL inertia / state");
  if(n_disturb < 0 || (n_disturb > 0 && (!disturb_times || !disturb_lin)))
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_closed_loop_device: bad disturbance list");
  ccc_ddp_params_t prm;
  ccc_ddp_config_t cfg0;
  if(int rc = ccc_ddp_get_params(h, &prm)) return rc;
  if(int rc = ccc_ddp_get_config(h, &cfg0)) return rc;
  if(tl->C != prm.max_phases)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_closed_loop_device: the timeline has %d contact entries, the handle %d phases",
                tl->C, prm.max_phases);
  const int kLoopM = prm.max_ridges; // ridge slots per contact entry of the timeline = the handle's ridge stride
  const int N = prm.horizon_steps, S = ccc_ddp_state_dim(h), model = prm.model == CCC_DDP_SINGLE_RIGID_BODY ? 1 : 0;
  int device = 0;
  if(int rc = ccc_ddp_get_device(h, &device)) return rc;
  CCC_DEVICE_GUARD(device);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DevBuf buf;
  int *step_phase, *dims_prev, *plan_status;
  double *ref_pos, *ref_ori = nullptr, *x0, *u, *ref_now, *inertia9 = nullptr;
  if(int rc = buf.get(&step_phase, (size_t)n * N)) return rc;
  if(int rc = buf.get(&plan_status, (size_t)n)) return rc;
  if(int rc = buf.get(&dims_prev, (size_t)n * N)) return rc;
  if(int rc = buf.get(&ref_pos, (size_t)n * (N + 1) * 3)) return rc;
  if(model == 1)
  {
    if(int rc = buf.get(&ref_ori, (size_t)n * (N + 1) * 3)) return rc;
    if(int rc = buf.get(&inertia9, (size_t)n * 9)) return rc;
    // MotionParam::inertia_mat = diag(moment_of_inertia) (TestDdpSingleRigidBody.cpp:56)
    std::vector<double> hI((size_t)n * 3), h9((size_t)n * 9, 0.0);
    CCC_HIP_CHECK(hipMemcpyAsync(hI.data(), inertia_diag, hI.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    CCC_HIP_CHECK(hipStreamSynchronize(s));
    for(int64_t k = 0; k < n; k++)
      for(int a = 0; a < 3; a++) h9[k * 9 + a * 4] = hI[k * 3 + a];
    CCC_HIP_CHECK(hipMemcpyAsync(inertia9, h9.data(), h9.size() * sizeof(double), hipMemcpyHostToDevice, s));
    CCC_HIP_CHECK(hipStreamSynchronize(s));
  }
  if(int rc = buf.get(&x0, (size_t)n * S)) return rc;
  if(int rc = buf.get(&u, (size_t)n * N * kLoopM)) return rc;
  if(int rc = buf.get(&ref_now, (size_t)n * 6)) return rc;
  CCC_HIP_CHECK(hipMemsetAsync(dims_prev, 0xff, (size_t)n * N * sizeof(int), s)); // -1: "no previous plan"
  CCC_HIP_CHECK(hipMemsetAsync(u, 0, (size_t)n * N * kLoopM * sizeof(double), s));
  if(stats) CCC_HIP_CHECK(hipMemsetAsync(stats, 0, (size_t)n * 8 * sizeof(double), s));
  const Timeline T = to_dev(tl, kLoopM);
  double t = t0;
  int rc = CCC_OK;
  // (the loop's simulator has ONE inertia per instance, diag(inertia_diag): the planner takes it in the one-matrix layout
  //  whatever layout the handle was created for)
  if(prm.inertia_per_phase) (void)ccc_ddp_set_inertia_per_phase(h, 0);
  for(int c = 0; c < cycles && rc == CCC_OK; c++)
  {
    hipLaunchKernelGGL(sample_ddp_kernel, dim3(blocks(n * (N + 1))), dim3(256), 0, s, T, (long)n, N, t, prm.horizon_dt,
                       step_phase, ref_pos, ref_ori);
    hipLaunchKernelGGL(ref_now_kernel, dim3(blocks(n)), dim3(256), 0, s, T, (long)n, t, ref_now);
    hipLaunchKernelGGL(planner_state_kernel, dim3(blocks(n)), dim3(256), 0, s, (long)n, model, prm.mass, sim_state, x0);
    // first cycle: cold start (zeros) with the full budget; afterwards the unshifted warm start with max_iter = warm
    hipLaunchKernelGGL(warm_start_kernel, dim3(blocks(n * N)), dim3(256), 0, s, (long)n, N, tl->C, kLoopM, tl->contact_dim,
                       step_phase, dims_prev, u);
    ccc_ddp_config_t cfg = cfg0;
    cfg.max_iter = c == 0 ? first_max_iter : warm_max_iter;
    rc = ccc_ddp_set_config(h, &cfg);
    if(rc == CCC_OK)
      rc = ccc_ddp_plan_batch_device(h, n, tl->contact_dim, tl->contact_vertex, tl->contact_ridge, step_phase, ref_pos,
                                     ref_ori, inertia9, x0, u, u, nullptr, nullptr, plan_status, nullptr, s);
    if(rc != CCC_OK) break;
    t += sim_dt; // t += sim_dt BEFORE the update and the disturbance test, as the reference loop
    SimArgs A{};
    A.n = n;
    A.model = model;
    A.N = N;
    A.P = tl->C;
    A.M = kLoopM;
    A.phase_dim = tl->contact_dim;
    A.phase_vertex = tl->contact_vertex;
    A.phase_ridge = tl->contact_ridge;
    A.step_phase = step_phase;
    A.scales = u;
    A.scale_stride = (long)N * kLoopM;
    A.mass = prm.mass;
    A.dt = sim_dt;
    A.inertia = inertia_diag;
    A.kick = 0;
    for(int d = 0; d < n_disturb; d++)
      if(disturb_times[d] <= t && t < disturb_times[d] + sim_dt)
      {
        A.kick = 1;
        for(int a = 0; a < 3; a++) A.kick_lin[a] = disturb_lin[a];
        break;
      }
    A.ref_now = ref_now;
    A.sim = sim_state;
    A.stats = stats;
    A.plan_status = plan_status;
    A.log = log ? log + (size_t)c * n * 9 : nullptr;
    hipLaunchKernelGGL(sim_step_kernel, dim3(blocks(n)), dim3(256), 0, s, A);
  }
  (void)ccc_ddp_set_config(h, &cfg0);
  if(prm.inertia_per_phase) (void)ccc_ddp_set_inertia_per_phase(h, 1);
  if(rc != CCC_OK) return rc;
  CCC_HIP_CHECK(hipGetLastError());
  CCC_HIP_CHECK(hipStreamSynchronize(s)); // the workspaces die with this call
  if(t_end) *t_end = t;
  return CCC_OK;
}

extern "C" int ccc_xy_closed_loop_device(ccc_xy_t * h, int64_t n, const ccc_contact_timeline_t * tl, double com_z,
                                         const double * inertia_diag, double * sim_state, double t0, double sim_dt,
                                         int cycles, double * stats, double * log, double * t_end, void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_closed_loop_device: NULL handle");
  if(n <= 0 || cycles < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_closed_loop_device: n <= 0 or cycles < 0");
  if(int rc = check_timeline(tl, "ccc_xy_closed_loop_device")) return rc;
  if(!inertia_diag || !sim_state) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_closed_loop_device: NULL inertia / state");
  ccc_xy_params_t prm;
  int device = 0;
  if(int rc = ccc_xy_get_params(h, &prm, &device)) return rc;
  const int kLoopM = prm.max_ridges; // ridge slots per contact entry of the timeline = the handle's ridge stride
  const int N = prm.horizon_steps;
  CCC_DEVICE_GUARD(device);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DevBuf buf;
  int * dim;
  double *vertex, *ridge, *cz, *fz, *ref_out, *x0, *u0, *ref_now;
  if(int rc = buf.get(&dim, (size_t)n * N)) return rc;
  if(int rc = buf.get(&vertex, (size_t)n * N * kLoopM * 3)) return rc;
  if(int rc = buf.get(&ridge, (size_t)n * N * kLoopM * 3)) return rc;
  if(int rc = buf.get(&cz, (size_t)n * N)) return rc;
  if(int rc = buf.get(&fz, (size_t)n * N)) return rc;
  if(int rc = buf.get(&ref_out, (size_t)n * N * 6)) return rc;
  if(int rc = buf.get(&x0, (size_t)n * 6)) return rc;
  if(int rc = buf.get(&u0, (size_t)n * kLoopM)) return rc;
  if(int rc = buf.get(&ref_now, (size_t)n * 6)) return rc;
  if(stats) CCC_HIP_CHECK(hipMemsetAsync(stats, 0, (size_t)n * 8 * sizeof(double), s));
  const Timeline T = to_dev(tl, kLoopM);
  const double total_force_z = prm.mass * kLoopG; // MotionParam::total_force_z of the test (TestLinearMpcXY.cpp:33)
  double t = t0;
  int rc = CCC_OK;
  for(int c = 0; c < cycles && rc == CCC_OK; c++)
  {
    hipLaunchKernelGGL(sample_xy_kernel, dim3(blocks(n * N)), dim3(256), 0, s, T, (long)n, N, t, prm.horizon_dt, prm.mass,
                       com_z, total_force_z, dim, vertex, ridge, cz, fz, ref_out);
    hipLaunchKernelGGL(ref_now_kernel, dim3(blocks(n)), dim3(256), 0, s, T, (long)n, t, ref_now);
    hipLaunchKernelGGL(planner_state_kernel, dim3(blocks(n)), dim3(256), 0, s, (long)n, 2, prm.mass, sim_state, x0);
    rc = ccc_xy_plan_batch_device(h, n, dim, vertex, ridge, cz, fz, ref_out, x0, u0, nullptr, nullptr, s);
    if(rc != CCC_OK) break;
    t += sim_dt;
    SimArgs A{};
    A.n = n;
    A.model = 2;
    A.N = N;
    A.P = 0;
    A.M = kLoopM;
    A.phase_vertex = vertex;
    A.phase_ridge = ridge;
    A.step_phase = dim;
    A.scales = u0;
    A.scale_stride = kLoopM;
    A.mass = prm.mass;
    A.dt = sim_dt;
    A.inertia = inertia_diag;
    A.kick = 0;
    A.ref_now = ref_now;
    A.sim = sim_state;
    A.stats = stats;
    A.log = log ? log + (size_t)c * n * 9 : nullptr;
    hipLaunchKernelGGL(sim_step_kernel, dim3(blocks(n)), dim3(256), 0, s, A);
  }
  if(rc != CCC_OK) return rc;
  CCC_HIP_CHECK(hipGetLastError());
  CCC_HIP_CHECK(hipStreamSynchronize(s));
  if(t_end) *t_end = t;
  return CCC_OK;
}
