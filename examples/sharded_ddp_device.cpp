// sharded_ddp_device.cpp -- a C++ host that keeps its DdpSingleRigidBody problems RESIDENT on every GPU of the node and
// reaches them through the C-ABI alone (include/ccc_amd.h "One node, several GPUs": ccc_ddp_sharded_*): device-resident
// shards, one planner handle + stream per device, one grouped in-place RCCL all-gather of the planned first-step force
// scales -- BASELINE config 5's "8 x MI355X" for a host that is one process.  Checked against the host-array entry of a
// single device.  (The only HIP calls are the host's own allocation and copies.)
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/sharded_ddp_device.cpp
//       -Lcentroidalcontrolcollection_amd/lib -lccc_amd -L/opt/rocm/lib -lamdhip64
#include <ccc_amd.h>
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x)                                                                        \
  do                                                                                     \
  {                                                                                      \
    hipError_t e_ = (x);                                                                 \
    if(e_ != hipSuccess)                                                                 \
    {                                                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      return 1;                                                                          \
    }                                                                                    \
  } while(0)

template<class T>
static T * to_device(int dev, const T * host, size_t count)
{
  T * d = nullptr;
  if(hipSetDevice(dev) != hipSuccess || hipMalloc(&d, count * sizeof(T)) != hipSuccess) return nullptr;
  if(host && hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main(int argc, char ** argv)
{
  const int64_t m = argc > 1 ? std::atoll(argv[1]) : 512; // instances per device
  const int N = 50, M = 16, P = 4, S = 12;
  const int D = ccc_device_count();
  if(D <= 0)
  {
    std::fprintf(stderr, "no gfx950 device: %s\n", ccc_last_error_string());
    return 2;
  }
  const int64_t n = m * D;
  std::vector<int> devices(D);
  for(int d = 0; d < D; d++) devices[d] = d;
  // problems: a rect contact (4 vertices x 4 friction-pyramid ridges) under the body, standing reference
  std::vector<int32_t> phase_dim(n * P, 0), step_phase(n * N, 0);
  std::vector<double> pv(n * P * M * 3, 0.0), pr(n * P * M * 3, 0.0), ref_pos(n * (N + 1) * 3), ref_ori(n * (N + 1) * 3, 0.0),
      inertia(n * 9, 0.0), x0(n * S, 0.0);
  unsigned s = 2468u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0) - 0.5; };
  const double mu = 0.5, nrm = std::sqrt(mu * mu + 1.0);
  const double vx[4] = {-0.1, -0.1, 0.1, 0.1}, vy[4] = {-0.5, 0.5, 0.5, -0.5};
  const double rx[4] = {mu, 0, -mu, 0}, ry[4] = {0, mu, 0, -mu};
  for(int64_t k = 0; k < n; k++)
  {
    phase_dim[k * P] = M;
    for(int v = 0; v < 4; v++)
      for(int r = 0; r < 4; r++)
      {
        double * V = &pv[((k * P) * M + v * 4 + r) * 3], * R = &pr[((k * P) * M + v * 4 + r) * 3];
        V[0] = vx[v], V[1] = vy[v], V[2] = 0.0;
        R[0] = rx[r] / nrm, R[1] = ry[r] / nrm, R[2] = 1.0 / nrm;
      }
    for(int i = 0; i <= N; i++) ref_pos[(k * (N + 1) + i) * 3 + 2] = 1.0;
    inertia[k * 9 + 0] = 40.0, inertia[k * 9 + 4] = 20.0, inertia[k * 9 + 8] = 10.0;
    x0[k * S + 0] = 0.05 * rnd(), x0[k * S + 1] = 0.05 * rnd(), x0[k * S + 2] = 1.0 + 0.05 * rnd();
    for(int a = 6; a < 9; a++) x0[k * S + a] = 0.2 * rnd();
  }
  ccc_ddp_params_t prm{};
  prm.model = CCC_DDP_SINGLE_RIGID_BODY;
  prm.mass = 100.0, prm.horizon_dt = 0.03, prm.horizon_steps = N, prm.max_phases = P, prm.max_ridges = M;
  const double wr[12] = {1, 1, 10, 0.5, 0.5, 0.5, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01};
  for(int a = 0; a < 12; a++) prm.w_run[a] = prm.w_term[a] = wr[a];
  prm.w_force = 1e-6, prm.force_scale_limits[0] = 0.0, prm.force_scale_limits[1] = 1e6;
  ccc_ddp_config_t cfg;
  ccc_ddp_default_config(&cfg);
  cfg.max_iter = 10;

  // one device: host arrays
  std::vector<double> u1(n * N * M);
  ccc_ddp_t * one = nullptr;
  if(ccc_ddp_create(&prm, 0, &one) != CCC_OK || ccc_ddp_set_config(one, &cfg) != CCC_OK
     || ccc_ddp_plan_batch(one, n, phase_dim.data(), pv.data(), pr.data(), step_phase.data(), ref_pos.data(), ref_ori.data(),
                           inertia.data(), x0.data(), nullptr, u1.data(), nullptr, nullptr, nullptr, nullptr) != CCC_OK)
  {
    std::fprintf(stderr, "single: %s\n", ccc_last_error_string());
    return 1;
  }
  // every device: its shard resident
  std::vector<const int32_t *> d_pd(D), d_sp(D);
  std::vector<const double *> d_pv(D), d_pr(D), d_rp(D), d_ro(D), d_in(D), d_x0(D);
  std::vector<double *> d_u(D), d_u0(D);
  for(int r = 0; r < D; r++)
  {
    const int64_t b = r * m;
    d_pd[r] = to_device(r, &phase_dim[b * P], m * P);
    d_sp[r] = to_device(r, &step_phase[b * N], m * N);
    d_pv[r] = to_device(r, &pv[b * P * M * 3], m * P * M * 3);
    d_pr[r] = to_device(r, &pr[b * P * M * 3], m * P * M * 3);
    d_rp[r] = to_device(r, &ref_pos[b * (N + 1) * 3], m * (N + 1) * 3);
    d_ro[r] = to_device(r, &ref_ori[b * (N + 1) * 3], m * (N + 1) * 3);
    d_in[r] = to_device(r, &inertia[b * 9], m * 9);
    d_x0[r] = to_device(r, &x0[b * S], m * S);
    d_u[r] = to_device<double>(r, nullptr, m * N * M);
    d_u0[r] = to_device<double>(r, nullptr, n * M);
    if(!d_pd[r] || !d_sp[r] || !d_pv[r] || !d_pr[r] || !d_rp[r] || !d_ro[r] || !d_in[r] || !d_x0[r] || !d_u[r] || !d_u0[r])
    {
      std::fprintf(stderr, "device %d: allocation / copy failed\n", r);
      return 1;
    }
    HIP_OK(hipDeviceSynchronize());
  }
  ccc_ddp_sharded_t * sh = nullptr;
  if(ccc_ddp_sharded_create(&prm, &cfg, devices.data(), D, &sh) != CCC_OK
     || ccc_ddp_sharded_plan_batch_device(sh, m, d_pd.data(), d_pv.data(), d_pr.data(), d_sp.data(), d_rp.data(), d_ro.data(),
                                          d_in.data(), d_x0.data(), nullptr, d_u.data(), d_u0.data(), nullptr, nullptr,
                                          nullptr, nullptr) != CCC_OK)
  {
    std::fprintf(stderr, "sharded: %s\n", ccc_last_error_string());
    return 1;
  }
  // every device now holds the first-step force scales of EVERY shard
  int64_t differ = 0;
  std::vector<double> u0(n * M);
  for(int r = 0; r < D; r++)
  {
    HIP_OK(hipSetDevice(r));
    HIP_OK(hipMemcpy(u0.data(), d_u0[r], n * M * sizeof(double), hipMemcpyDeviceToHost));
    for(int64_t k = 0; k < n; k++)
      for(int c = 0; c < M; c++) differ += u0[k * M + c] != u1[k * N * M + c];
  }
  std::printf("devices=%d instances=%lld differ=%lld\n", D, (long long)n, (long long)differ);
  ccc_ddp_sharded_destroy(sh);
  ccc_ddp_destroy(one);
  return differ == 0 ? 0 : 1;
}
