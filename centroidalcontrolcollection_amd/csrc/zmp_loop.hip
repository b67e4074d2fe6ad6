// zmp_loop.hip -- the steps either side of LinearMpcZmp::planOnce on the device:
//   * reference sampling: footstep timelines -> ZMP-limit sequences (what the N std::function calls of
//     src/LinearMpcZmp.cpp:86-98 into FootstepManager::makeLinearMpcZmpRefData produce on the host),
//   * the simulation step of the reference's closed-loop test and the whole control loop
//     (tests/src/TestLinearMpcZmp.cpp:55-102: plan -> ComZmpSim2d::update -> disturbance), batched, so that
//     Monte-Carlo robustness sweeps never leave the GPU.
//
// Restates (reference file:line under /root/reference):
//   tests/src/FootstepManager.h:147-206,242-254   ref_footstance_list_ / zmpLimits: a foot is off the ground during
//                                                 [swing_start, swing_end) and sits at its new position from swing_end on;
//                                                 limits = support region of the feet on the ground +- half a foot size
//   tests/src/FootstepManager.h:356-365           makeLinearMpcZmpRefData (+1e-6 twice)
//   tests/src/SimModels.h:11-41,76-137            ComZmpSimModel1d / ComZmpSim2d (exact ZOH of the LIPM), addDisturb
//   tests/src/TestLinearMpcZmp.cpp:62-71          InitialParam from the simulated state: acc = g/h (pos - planned_zmp)
#include "common.h"

#include <cmath>
#include <vector>

// every product and sum of the sampler rounds separately (the host fixture it is compared with bit for bit does)
#pragma clang fp contract(off)

namespace ccc_amd
{
struct Timeline
{
  int K;                      // footsteps per instance
  const double * foot0;       // [n][2 (L, R)][2]
  const double * foot_pos;    // [n][K][2]
  const int * foot_id;        // [n][K]  0 = left, 1 = right
  const double * swing_start; // [n][K]
  const double * swing_end;   // [n][K]
  double half_x, half_y;      // half the foot size (FootstepManager.h:464)
};

// limits of instance k at time t (already including the two epsilons): FootstepManager.h:179-196,242-254
__device__ __forceinline__ void timeline_limits(const Timeline & L, long k, double t, double & lox, double & loy,
                                                double & hix, double & hiy)
{
  double px[2] = {L.foot0[k * 4 + 0], L.foot0[k * 4 + 2]};
  double py[2] = {L.foot0[k * 4 + 1], L.foot0[k * 4 + 3]};
  bool ground[2] = {true, true};
  for(int j = 0; j < L.K; j++)
  {
    const int f = L.foot_id[k * L.K + j] != 0 ? 1 : 0;
    const bool landed = t >= L.swing_end[k * L.K + j];
    const bool swinging = t >= L.swing_start[k * L.K + j] && !landed;
    if(landed)
    {
      px[f] = L.foot_pos[(k * L.K + j) * 2 + 0];
      py[f] = L.foot_pos[(k * L.K + j) * 2 + 1];
    }
    if(swinging) ground[f] = false;
  }
  const double big = 1e30;
  lox = fmin(ground[0] ? px[0] : big, ground[1] ? px[1] : big) - L.half_x;
  loy = fmin(ground[0] ? py[0] : big, ground[1] ? py[1] : big) - L.half_y;
  hix = fmax(ground[0] ? px[0] : -big, ground[1] ? px[1] : -big) + L.half_x;
  hiy = fmax(ground[0] ? py[0] : -big, ground[1] ? py[1] : -big) + L.half_y;
}

// zlim [n][2 axes][2 (min, max)][N]; t_eval per instance (array) or common (scalar when t_arr == nullptr)
__global__ void zmp_sample_limits_kernel(Timeline L, long n, int N, double dt, const double * __restrict__ t_arr,
                                         double t_common, double * __restrict__ zlim)
{
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= n * N) return;
  const long k = e / N;
  const int i = (int)(e % N);
  const double t0 = t_arr ? t_arr[k] : t_common;
  const double t = (t0 + dt * i) + 2e-6; // LinearMpcZmp.cpp:88, then FootstepManager.h:360 and :245
  double lox, loy, hix, hiy;
  timeline_limits(L, k, t, lox, loy, hix, hiy);
  double * z = zlim + k * 4 * N;
  z[0 * N + i] = lox;
  z[1 * N + i] = hix;
  z[2 * N + i] = loy;
  z[3 * N + i] = hiy;
}

struct SimStep
{
  double a00, a01, a10, a11, b0, b1; // exact ZOH of x'' = w^2 (x - zmp) over sim_dt (SimModels.h:11-41)
  double g_over_h;
  double imp;                        // disturbance added to both axes' velocity this cycle (0: none; SimModels.h:125-129)
};

// one control cycle after the plan: record, simulate, disturb, next InitialParam.
//   com [n][2 axes][2] (pos, vel) in/out, zmp [n][2] planned this cycle, x0 [n][2][3] out for the next plan,
//   viol [n] counts cycles whose planned ZMP left the limits at the CURRENT time (TestLinearMpcZmp.cpp:86-87)
__global__ void zmp_sim_step_kernel(SimStep S, Timeline L, long n, double t_now, const double * __restrict__ zmp,
                                    double * __restrict__ com, double * __restrict__ x0, int * __restrict__ viol,
                                    double * __restrict__ traj_com, double * __restrict__ traj_zmp)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const double zx = zmp[k * 2 + 0], zy = zmp[k * 2 + 1];
  if(viol)
  {
    double lox, loy, hix, hiy;
    timeline_limits(L, k, t_now + 1e-6, lox, loy, hix, hiy); // zmpLimits(t) of the check, FootstepManager.h:245
    if(!(zx - lox >= 0 && zy - loy >= 0 && hix - zx >= 0 && hiy - zy >= 0)) viol[k] += 1;
  }
  double px = com[k * 4 + 0], vx = com[k * 4 + 1], py = com[k * 4 + 2], vy = com[k * 4 + 3];
  if(traj_com)
  {
    traj_com[k * 2 + 0] = px;
    traj_com[k * 2 + 1] = py;
  }
  if(traj_zmp)
  {
    traj_zmp[k * 2 + 0] = zx;
    traj_zmp[k * 2 + 1] = zy;
  }
  const double npx = (S.a00 * px + S.a01 * vx) + S.b0 * zx, nvx = (S.a10 * px + S.a11 * vx) + S.b1 * zx;
  const double npy = (S.a00 * py + S.a01 * vy) + S.b0 * zy, nvy = (S.a10 * py + S.a11 * vy) + S.b1 * zy;
  px = npx;
  py = npy;
  vx = nvx + S.imp;
  vy = nvy + S.imp;
  com[k * 4 + 0] = px;
  com[k * 4 + 1] = vx;
  com[k * 4 + 2] = py;
  com[k * 4 + 3] = vy;
  // TestLinearMpcZmp.cpp:66-69
  x0[k * 6 + 0] = px;
  x0[k * 6 + 1] = vx;
  x0[k * 6 + 2] = S.g_over_h * (px - zx);
  x0[k * 6 + 3] = py;
  x0[k * 6 + 4] = vy;
  x0[k * 6 + 5] = S.g_over_h * (py - zy);
}
} // namespace ccc_amd

using namespace ccc_amd;

namespace
{
constexpr double kG = 9.80665;

int check_timeline(const char * who, int64_t n, int K, const double * foot0, const double * foot_pos,
                   const int32_t * foot_id, const double * swing_start, const double * swing_end)
{
  if(n < 0 || K < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "%s: n or K < 0", who);
  if(!foot0 || (K > 0 && (!foot_pos || !foot_id || !swing_start || !swing_end)))
    return fail(CCC_ERR_INVALID_ARGUMENT, "%s: NULL footstep timeline array", who);
  return CCC_OK;
}
} // namespace

extern "C" int ccc_zmp_sample_limits_device(ccc_zmp_t * h, int64_t n, int K, const double * foot0,
                                            const double * foot_pos, const int32_t * foot_id,
                                            const double * swing_start, const double * swing_end,
                                            const double * foot_size, const double * t_eval, double t_common,
                                            double * zlim, void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sample_limits_device: NULL handle");
  int rc = check_timeline("ccc_zmp_sample_limits_device", n, K, foot0, foot_pos, foot_id, swing_start, swing_end);
  if(rc != CCC_OK) return rc;
  if(n == 0) return CCC_OK;
  if(!zlim) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sample_limits_device: NULL zlim");
  double com_height = 0, dt = 0;
  rc = ccc_zmp_get_model(h, &com_height, &dt);
  if(rc != CCC_OK) return rc;
  const int N = ccc_zmp_horizon_steps(h);
  Timeline L{K, foot0, foot_pos, foot_id, swing_start, swing_end, 0.5 * (foot_size ? foot_size[0] : 0.1),
             0.5 * (foot_size ? foot_size[1] : 0.05)};
  const long total = (long)n * N;
  hipLaunchKernelGGL(zmp_sample_limits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), L, (long)n, N, dt, t_eval, t_common, zlim);
  CCC_HIP_CHECK(hipGetLastError());
  return CCC_OK;
}

extern "C" int ccc_zmp_closed_loop_device(ccc_zmp_t * h, int64_t n, int K, const double * foot0,
                                          const double * foot_pos, const int32_t * foot_id,
                                          const double * swing_start, const double * swing_end,
                                          const double * foot_size, double * com_state, double * planned_zmp,
                                          double t0, double sim_dt, int cycles, int n_disturb,
                                          const double * disturb_times, double disturb_impulse, double * work_x0,
                                          double * work_zlim, int32_t * violations, double * traj_com,
                                          double * traj_zmp, double * t_end, void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_closed_loop_device: NULL handle");
  int rc = check_timeline("ccc_zmp_closed_loop_device", n, K, foot0, foot_pos, foot_id, swing_start, swing_end);
  if(rc != CCC_OK) return rc;
  if(cycles < 0 || !(sim_dt > 0)) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_closed_loop_device: cycles < 0 or sim_dt <= 0");
  if(n_disturb > 0 && !disturb_times) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_closed_loop_device: NULL disturb_times");
  if(t_end) *t_end = t0;
  if(n == 0 || cycles == 0) return CCC_OK;
  if(!com_state || !planned_zmp || !work_x0 || !work_zlim)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_closed_loop_device: NULL state / workspace array");
  double com_height = 0, dt = 0;
  rc = ccc_zmp_get_model(h, &com_height, &dt);
  if(rc != CCC_OK) return rc;
  const int N = ccc_zmp_horizon_steps(h);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Timeline L{K, foot0, foot_pos, foot_id, swing_start, swing_end, 0.5 * (foot_size ? foot_size[0] : 0.1),
             0.5 * (foot_size ? foot_size[1] : 0.05)};
  const double w = std::sqrt(kG / com_height), ch = std::cosh(w * sim_dt), sh = std::sinh(w * sim_dt);
  SimStep S{ch, sh / w, w * sh, ch, 1 - ch, -w * sh, kG / com_height, 0.0};
  // the first InitialParam: acc = g/h (pos - planned_zmp) from the state handed in (TestLinearMpcZmp.cpp:52,66-69)
  {
    SimStep I{1, 0, 0, 1, 0, 0, S.g_over_h, 0.0}; // identity step: only forms x0
    hipLaunchKernelGGL(zmp_sim_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, I, L, (long)n, t0,
                       planned_zmp, com_state, work_x0, (int *)nullptr, (double *)nullptr, (double *)nullptr);
  }
  double t = t0;
  const long total = (long)n * N;
  for(int c = 0; c < cycles; c++)
  {
    hipLaunchKernelGGL(zmp_sample_limits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L, (long)n, N,
                       dt, (const double *)nullptr, t, work_zlim);
    rc = ccc_zmp_plan_batch_device(h, n, work_x0, work_zlim, sim_dt, planned_zmp, nullptr, nullptr, stream);
    if(rc != CCC_OK) return rc;
    const double t_next = t + sim_dt; // TestLinearMpcZmp.cpp:90
    SimStep C = S;
    for(int d = 0; d < n_disturb; d++)
      if(disturb_times[d] <= t_next && t_next < disturb_times[d] + sim_dt) // :94-101 (first match only)
      {
        C.imp = disturb_impulse;
        break;
      }
    hipLaunchKernelGGL(zmp_sim_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, C, L, (long)n, t,
                       planned_zmp, com_state, work_x0, violations, traj_com ? traj_com + (size_t)c * n * 2 : nullptr,
                       traj_zmp ? traj_zmp + (size_t)c * n * 2 : nullptr);
    t = t_next;
  }
  CCC_HIP_CHECK(hipGetLastError());
  if(t_end) *t_end = t;
  return CCC_OK;
}
