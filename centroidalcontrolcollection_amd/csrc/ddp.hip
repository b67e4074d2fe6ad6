// ddp.hip -- batched CCC::DdpCentroidal / CCC::DdpSingleRigidBody planOnce() on MI355X: the C-ABI around the tile kernel.
//
// One problem instance per wavefront (one 64-thread workgroup), the whole solve -- forward rollouts, analytic
// derivatives, backward Riccati sweep with box-QP, line search -- on the device (csrc/ddp_tile.h, launched by
// csrc/ddp_tile.hip): every ridge stride (16, 32, 64), both regularisations, fp64; arithmetic = the tile specification
// (oracle/ddp_tile.c, ccc_ddp_arithmetic = 1).  Trajectories, the four line-search candidates and the feedback gains
// live in an HBM workspace owned by the handle, one slot per RESIDENT workgroup (a work queue hands out the instances).
// Removed in round 4: the row-per-lane kernels of rounds 1-2 (left-to-right arithmetic: CCC_DDP_LEGACY, reg_type 2) and
// the fp32-storage build (precision 32) -- the tile kernel takes reg_type 2 itself and is three times as fast as the
// fp32-storage build was.
#include "common.h"
#include "ddp_batch.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace ccc_amd;

struct ccc_ddp
{
  int device = 0;
  ccc_ddp_params_t prm{};
  ccc_ddp_config_t cfg{};
  int S = 9;
  int M = CCC_DDP_MAX_RIDGES; // ridge stride of the per-phase / per-step arrays (params.max_ridges)
  int64_t tcap = 0;           // workspace of the tile kernel (csrc/ddp_tile.hip): one slot per resident workgroup
  double * ws_t = nullptr;
  void * sched = nullptr;      // its scheduling state (csrc/ddp_batch.h DdpSched), sized for scap instances
  int64_t scap = -1;           // (-1: not allocated; 0: counters only)
  int slice = 2, slice_next = 4; // iterations of an instance's first / later slices (CCC_DDP_SLICE="a,b", development switch; 0: plain queue)
  bool slice_env = false;      // (CCC_DDP_SLICE given: no choice by batch size)
  int slots = 0;               // CCC_DDP_SLOTS (development switch): resident workgroups to launch, 0 = what fits
  int update_kmax = 4;         // CCC_DDP_UPDATE_KMAX (development switch; the specification's S_UPDATE_KMAX is 4)
  int64_t hist_n = -1;         // batch size of the last sliced launch: its per-instance busy times (DdpSched::prev) order the
                               // next launch of the same size, longest first (closed-loop callers repeat the batch)
  int hist_runs = 0;           // consecutive sliced launches of that size so far
  int history = 1;             // CCC_DDP_HISTORY=0 (development switch) turns the history off
  int num_cu = 0;
  int per_cu[2] = {0, 0};      // resident workgroups per CU of this handle's kernel, one matrix per instance / per phase
                               // (ddp_tile_blocks_per_cu: launch bounds capped by the runtime's occupancy query)
  int * abort_host = nullptr;  // page-locked word a launch's waits raise when they give up (DdpSched::abort_host)
  long long spin_budget_ms = 10000; // budget of a scheduler wait without progress (CCC_DDP_SPIN_BUDGET_MS, development
                               // switch; 0 = wait for ever); handed to the kernel as a number of looks of ~4 us
  int test_drop = -1;          // CCC_DDP_TEST_DROP (tests only): this instance of every sliced launch is lost
  // staging for the host entry
  int64_t hcap = 0;
  void * d_stage = nullptr;
  hipStream_t stream = nullptr;
};

extern "C" void ccc_ddp_default_config(ccc_ddp_config_t * c)
{
  if(!c) return;
  c->max_iter = 500;
  c->initial_lambda = 1e-6; // src/DdpCentroidal.cpp:199
  c->initial_dlambda = 1.0;
  c->lambda_factor = 1.6;
  c->lambda_min = 1e-8; // :200
  c->lambda_max = 1e10;
  c->k_rel_norm_thre = 1e-4;
  c->lambda_thre = 1e-7; // :201
  c->cost_update_ratio_thre = 0.0;
  c->cost_update_thre = 1e-7;
  for(int i = 0; i < 11; i++) c->alpha_list[i] = std::pow(10.0, -3.0 * i / 10.0);
  c->reg_type = 1;
  c->precision = 64;
  c->warm_start_guard = 1;
}

extern "C" int ccc_ddp_create(const ccc_ddp_params_t * p, int device, ccc_ddp_t ** out)
{
  if(!out || !p) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_create: NULL argument");
  *out = nullptr;
  if(p->model != CCC_DDP_CENTROIDAL && p->model != CCC_DDP_SINGLE_RIGID_BODY)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_create: unknown model %d", p->model);
  if(!(p->mass > 0) || !(p->horizon_dt > 0) || p->horizon_steps <= 0 || p->max_phases <= 0)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_create: mass, horizon_dt, horizon_steps, max_phases must be > 0");
  if(p->max_ridges != 0 && p->max_ridges != CCC_DDP_MAX_RIDGES && p->max_ridges != CCC_DDP_MAX_RIDGES_WIDE
     && p->max_ridges != CCC_DDP_MAX_RIDGES_MULTI)
    return fail(CCC_ERR_UNSUPPORTED, "ccc_ddp_create: max_ridges = %d, the kernels are built for %d, %d and %d",
                p->max_ridges, CCC_DDP_MAX_RIDGES, CCC_DDP_MAX_RIDGES_WIDE, CCC_DDP_MAX_RIDGES_MULTI);
  int rc = select_device(device);
  if(rc != CCC_OK) return rc;
  CCC_DEVICE_GUARD(device);
  ccc_ddp * h = new ccc_ddp();
  h->device = device;
  h->prm = *p;
  h->M = p->max_ridges ? p->max_ridges : CCC_DDP_MAX_RIDGES;
  h->prm.max_ridges = h->M;
  h->S = p->model == CCC_DDP_CENTROIDAL ? 9 : 12;
  ccc_ddp_default_config(&h->cfg);
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if(e != hipSuccess)
  {
    delete h;
    return fail(CCC_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  }
  h->num_cu = prop.multiProcessorCount;
  // development switches, read once here
  if(const char * v = std::getenv("CCC_DDP_SLICE"))
  {
    h->slice_env = true;
    h->slice = std::max(0, std::atoi(v));
    if(const char * c2 = std::strchr(v, ',')) h->slice_next = std::max(1, std::atoi(c2 + 1));
  }
  if(const char * v = std::getenv("CCC_DDP_SLOTS")) h->slots = std::max(0, std::atoi(v));
  if(const char * v = std::getenv("CCC_DDP_UPDATE_KMAX")) h->update_kmax = std::max(0, std::atoi(v));
  if(const char * v = std::getenv("CCC_DDP_HISTORY")) h->history = std::atoi(v);
  if(const char * v = std::getenv("CCC_DDP_SPIN_BUDGET_MS")) h->spin_budget_ms = std::max(0LL, std::atoll(v));
  if(const char * v = std::getenv("CCC_DDP_TEST_DROP")) h->test_drop = std::atoi(v);
  // the resident set the work queue is launched as: what the launch bounds ask for, checked against the runtime's occupancy
  // figure for this kernel on this device (no assumption about co-residency is left unchecked)
  for(int ipp = 0; ipp < (h->S == 12 ? 2 : 1); ipp++)
  {
    h->per_cu[ipp] = ddp_tile_blocks_per_cu(h->S, h->M, ipp != 0);
    if(h->per_cu[ipp] <= 0)
    {
      delete h;
      return fail(CCC_ERR_HIP, "ccc_ddp_create: the occupancy query for the DDP kernel (S = %d, M = %d) failed: %s", p->model == 0 ? 9 : 12,
                  p->max_ridges ? p->max_ridges : CCC_DDP_MAX_RIDGES, hipGetErrorString(hipGetLastError()));
    }
  }
  {
    void * w = nullptr;
    e = hipHostMalloc(&w, 64, hipHostMallocMapped);
    if(e != hipSuccess)
    {
      delete h;
      return fail(CCC_ERR_HIP, "ccc_ddp_create: hipHostMalloc: %s", hipGetErrorString(e));
    }
    h->abort_host = static_cast<int *>(w);
    *h->abort_host = 0;
  }
  *out = h;
  return CCC_OK;
}

extern "C" void ccc_ddp_destroy(ccc_ddp_t * h)
{
  if(!h) return;
  ccc_amd::DeviceGuard ccc_device_guard__(h->device);
  if(h->ws_t) (void)hipFree(h->ws_t);
  if(h->sched) (void)hipFree(h->sched);
  if(h->d_stage) (void)hipFree(h->d_stage);
  if(h->abort_host) (void)hipHostFree(h->abort_host);
  if(h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static void fill_params(const ccc_ddp * h, ddp_common::Params & P)
{
  std::memset(&P, 0, sizeof(P));
  P.model = h->prm.model;
  P.N = h->prm.horizon_steps;
  P.P = h->prm.max_phases;
  P.mass = h->prm.mass;
  P.dt = h->prm.horizon_dt;
  for(int a = 0; a < 12; a++)
  {
    P.w_run[a] = h->prm.w_run[a];
    P.w_term[a] = h->prm.w_term[a];
  }
  P.w_force = h->prm.w_force;
  P.flo = h->prm.force_scale_limits[0];
  P.fhi = h->prm.force_scale_limits[1];
  P.max_iter = h->cfg.max_iter;
  P.lambda0 = h->cfg.initial_lambda;
  P.dlambda0 = h->cfg.initial_dlambda;
  P.lambda_factor = h->cfg.lambda_factor;
  P.lambda_min = h->cfg.lambda_min;
  P.lambda_max = h->cfg.lambda_max;
  P.k_rel_norm_thre = h->cfg.k_rel_norm_thre;
  P.lambda_thre = h->cfg.lambda_thre;
  P.ratio_thre = h->cfg.cost_update_ratio_thre;
  P.cost_thre = h->cfg.cost_update_thre;
  for(int i = 0; i < 11; i++) P.alpha[i] = h->cfg.alpha_list[i];
  P.reg_type = h->cfg.reg_type;
  P.warm_guard = h->cfg.warm_start_guard ? 1 : 0;
  P.inertia_per_phase = (h->prm.model == CCC_DDP_SINGLE_RIGID_BODY && h->prm.inertia_per_phase) ? 1 : 0;
  P.update_kmax = h->update_kmax;
}

extern "C" int ccc_ddp_arithmetic(const ccc_ddp_t * h)
{
  return h ? 1 : -1; // (one kernel, one arithmetic: oracle/ddp_tile.c)
}

extern "C" int ccc_ddp_effective_precision(const ccc_ddp_t * h)
{
  return h ? 64 : -1; // (one kernel, fp64; cfg.precision = 32 is a request it over-fulfils: ccc_amd.h)
}

extern "C" int ccc_ddp_set_config(ccc_ddp_t * h, const ccc_ddp_config_t * cfg)
{
  if(!h || !cfg) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_set_config: NULL argument");
  if(cfg->max_iter < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_set_config: max_iter < 0");
  if(cfg->reg_type != 1 && cfg->reg_type != 2) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_set_config: reg_type must be 1 or 2");
  if(cfg->precision != 64 && cfg->precision != 32)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_set_config: precision must be 64 or 32");
  h->cfg = *cfg;
  return CCC_OK;
}

// force_scale_limits_ is a live public member of the reference classes, read at every solve
// (src/DdpCentroidal.cpp:202-210, src/DdpSingleRigidBody.cpp:272-280)
extern "C" int ccc_ddp_set_limits(ccc_ddp_t * h, double lo, double hi)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_set_limits: NULL handle");
  if(std::isnan(lo) || std::isnan(hi) || lo > hi)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_set_limits: need lo <= hi, got [%g, %g]", lo, hi);
  h->prm.force_scale_limits[0] = lo;
  h->prm.force_scale_limits[1] = hi;
  return CCC_OK;
}

extern "C" int ccc_ddp_set_inertia_per_phase(ccc_ddp_t * h, int per_phase)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_set_inertia_per_phase: NULL handle");
  h->prm.inertia_per_phase = per_phase ? 1 : 0;
  return CCC_OK;
}

extern "C" int ccc_ddp_state_dim(const ccc_ddp_t * h)
{
  return h ? h->S : -1;
}

extern "C" int ccc_ddp_get_params(const ccc_ddp_t * h, ccc_ddp_params_t * params)
{
  if(!h || !params) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_get_params: NULL argument");
  *params = h->prm;
  return CCC_OK;
}

extern "C" int ccc_ddp_get_config(const ccc_ddp_t * h, ccc_ddp_config_t * cfg)
{
  if(!h || !cfg) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_get_config: NULL argument");
  *cfg = h->cfg;
  return CCC_OK;
}

extern "C" int ccc_ddp_get_device(const ccc_ddp_t * h, int * device)
{
  if(!h || !device) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_get_device: NULL argument");
  *device = h->device;
  return CCC_OK;
}

extern "C" int ccc_ddp_plan_batch_device(ccc_ddp_t * h, int64_t n, const int32_t * phase_dim,
                                         const double * phase_vertex, const double * phase_ridge,
                                         const int32_t * step_phase, const double * ref_pos, const double * ref_ori,
                                         const double * inertia, const double * x0, const double * u_init,
                                         double * u_out, double * x_out, int32_t * iters, int32_t * status,
                                         double * cost, void * stream)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_plan_batch_device: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_plan_batch_device: n < 0");
  if(n == 0) return CCC_OK;
  if(!phase_dim || !phase_vertex || !phase_ridge || !step_phase || !ref_pos || !x0 || !u_out)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_plan_batch_device: NULL required array");
  if(h->prm.model == CCC_DDP_SINGLE_RIGID_BODY && (!ref_ori || !inertia))
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_plan_batch_device: the single-rigid-body model needs ref_ori and inertia");
  CCC_DEVICE_GUARD(h->device);
  // (precision = 32 runs this same fp64 kernel: ccc_amd.h)
  const int per_cu = h->per_cu[(h->S == 12 && h->prm.inertia_per_phase) ? 1 : 0];
  int grid = ddp_tile_grid((long)n, per_cu, h->num_cu);
  if(h->slots > 0 && h->slots < grid) grid = h->slots;
  if(grid > h->tcap)
  {
    CCC_NO_CAPTURE(stream, "ccc_ddp_plan_batch_device");
    if(h->ws_t) (void)hipFree(h->ws_t);
    h->ws_t = nullptr;
    h->tcap = 0;
    // (sized for a full resident set at once: later, larger batches do not allocate again)
    const int full = ddp_tile_grid(1L << 40, std::max(h->per_cu[0], h->per_cu[1]), h->num_cu);
    CCC_HIP_CHECK(hipMalloc(&h->ws_t, (size_t)full * ddp_tile_ws_doubles(h->prm.horizon_steps, h->S, h->M) * sizeof(double)));
    h->tcap = full;
  }
  // longest-first scheduling (csrc/ddp_batch.h) when the batch does not fit one resident set; its lists are per instance
  const bool sliced = h->slice > 0 && n > grid;
  const int64_t need = sliced ? n : 0;
  if(need > h->scap)
  {
    CCC_NO_CAPTURE(stream, "ccc_ddp_plan_batch_device");
    if(h->sched) (void)hipFree(h->sched);
    h->sched = nullptr;
    h->scap = -1;
    CCC_HIP_CHECK(hipMalloc(&h->sched, ddp_sched_bytes((long)need, h->prm.horizon_steps, h->S)));
    h->scap = need;
    h->hist_n = -1; // (the busy times went with the old allocation)
    h->hist_runs = 0;
  }
  DdpSched sched = ddp_sched_carve(h->sched, (long)h->scap, h->prm.horizon_steps, h->S);
  sched.slice = sliced ? h->slice : 0;
  sched.slice_next = h->slice_next;
  if(sliced && !h->slice_env)
  {
    // (a suspension and the resumption after it write and read an instance's trajectories and rebuild its LDS tables -- more of
    //  them since round 5: with many instances per slot, each suspended several times, short slices cost more than they
    //  balance.  Measured without a history, SRB N = 50, slices 2,4 / 3,6 / 5,10: 32768 instances 197 / 141 / 133 ms,
    //  16384 105 / 89 / 91 ms, 8192 72.5 / 72.2 / 74.5 ms; DdpCentroidal N = 100, 4096 instances 54.5 / - / 57.5 ms)
    const int64_t per_slot = n / grid;
    if(per_slot >= 16)
    {
      sched.slice = 5;
      sched.slice_next = 10;
    }
    else if(per_slot >= 6)
    {
      sched.slice = 3;
      sched.slice_next = 6;
    }
  }
  if(sliced)
  {
    h->hist_runs = (h->hist_n == n) ? h->hist_runs + 1 : 0; // launches of this size before this one
    h->hist_n = n;                                          // (this launch leaves its busy times for the next one)
  }
  sched.use_history = (sliced && h->history && h->hist_runs >= 1) ? (h->hist_runs >= 2 ? 2 : 1) : 0;
  sched.abort_host = h->abort_host;
  sched.spin_limit = (unsigned)std::min(4000000000LL, h->spin_budget_ms * 250LL); // (a look of the long wait: ~4 us)
  sched.test_drop = sliced ? h->test_drop : -1;
  __atomic_store_n(h->abort_host, 0, __ATOMIC_RELAXED); // (raised again by this launch's waits if they give up)
  ddp_common::Params P;
  fill_params(h, P);
  DdpBatch B{phase_dim, phase_vertex, phase_ridge, step_phase, ref_pos, ref_ori, inertia, x0, u_init, u_out, x_out, iters, status, cost};
  CCC_HIP_CHECK(launch_ddp_tile(P, B, h->ws_t, sched, grid, (long)n, h->S, h->M, reinterpret_cast<hipStream_t>(stream)));
  return CCC_OK;
}

extern "C" int ccc_ddp_plan_batch(ccc_ddp_t * h, int64_t n, const int32_t * phase_dim, const double * phase_vertex,
                                  const double * phase_ridge, const int32_t * step_phase, const double * ref_pos,
                                  const double * ref_ori, const double * inertia, const double * x0,
                                  const double * u_init, double * u_out, double * x_out, int32_t * iters,
                                  int32_t * status, double * cost)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_plan_batch: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_plan_batch: n < 0");
  if(n == 0) return CCC_OK;
  if(!phase_dim || !phase_vertex || !phase_ridge || !step_phase || !ref_pos || !x0 || !u_out)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_plan_batch: NULL required array");
  CCC_DEVICE_GUARD(h->device);
  const size_t N = h->prm.horizon_steps, S = h->S, M = h->M, Pn = h->prm.max_phases;
  if(!h->stream) CCC_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  // one staging allocation, carved into the arrays (all sizes are multiples of 4 bytes; doubles first)
  struct Seg
  {
    const void * src;
    void * dst_host;
    size_t bytes;
    size_t off;
  };
  std::vector<Seg> in = {{phase_vertex, nullptr, n * Pn * M * 3 * 8, 0}, {phase_ridge, nullptr, n * Pn * M * 3 * 8, 0},
                         {ref_pos, nullptr, n * (N + 1) * 3 * 8, 0},     {ref_ori, nullptr, n * (N + 1) * 3 * 8, 0},
                         {inertia, nullptr, (size_t)n * 9 * 8 * (h->prm.inertia_per_phase ? Pn : 1), 0}, {x0, nullptr, n * S * 8, 0},
                         {u_init, nullptr, n * N * M * 8, 0},             {phase_dim, nullptr, n * Pn * 4, 0},
                         {step_phase, nullptr, n * N * 4, 0}};
  std::vector<Seg> outv = {{nullptr, u_out, n * N * M * 8, 0},
                           {nullptr, x_out, n * (N + 1) * S * 8, 0},
                           {nullptr, cost, (size_t)n * 8, 0},
                           {nullptr, iters, (size_t)n * 4, 0},
                           {nullptr, status, (size_t)n * 4, 0}};
  size_t total = 0;
  for(auto & s : in)
  {
    s.off = total;
    total += (s.bytes + 255) / 256 * 256;
  }
  for(auto & s : outv)
  {
    s.off = total;
    total += (s.bytes + 255) / 256 * 256;
  }
  if((int64_t)total > h->hcap)
  {
    if(h->d_stage) (void)hipFree(h->d_stage);
    h->d_stage = nullptr;
    h->hcap = 0;
    CCC_HIP_CHECK(hipMalloc(&h->d_stage, total));
    h->hcap = (int64_t)total;
  }
  char * base = static_cast<char *>(h->d_stage);
  for(auto & s : in)
    if(s.src) CCC_HIP_CHECK(hipMemcpyAsync(base + s.off, s.src, s.bytes, hipMemcpyHostToDevice, h->stream));
  auto dptr = [&](const Seg & s, const void * host) -> void * { return host ? base + s.off : nullptr; };
  int rc = ccc_ddp_plan_batch_device(
      h, n, (const int32_t *)dptr(in[7], phase_dim), (const double *)dptr(in[0], phase_vertex),
      (const double *)dptr(in[1], phase_ridge), (const int32_t *)dptr(in[8], step_phase),
      (const double *)dptr(in[2], ref_pos), (const double *)dptr(in[3], ref_ori), (const double *)dptr(in[4], inertia),
      (const double *)dptr(in[5], x0), (const double *)dptr(in[6], u_init), (double *)(base + outv[0].off),
      (double *)dptr(outv[1], x_out), (int32_t *)dptr(outv[3], iters), (int32_t *)dptr(outv[4], status),
      (double *)dptr(outv[2], cost), h->stream);
  if(rc != CCC_OK) return rc;
  for(auto & s : outv)
    if(s.dst_host) CCC_HIP_CHECK(hipMemcpyAsync(s.dst_host, base + s.off, s.bytes, hipMemcpyDeviceToHost, h->stream));
  CCC_HIP_CHECK(hipStreamSynchronize(h->stream));
  if(ccc_ddp_last_call_aborted(h) == 1)
    return fail(CCC_ERR_HIP, "ccc_ddp_plan_batch: the scheduler of the DDP kernel gave up a wait that saw no progress for about %.1f s; the "
                             "instances it did not complete carry status CCC_DDP_STATUS_ABORTED", (double)h->spin_budget_ms * 1e-3);
  return CCC_OK;
}

// 1: a wait of the most recent launch's scheduler gave up (DdpSched: bounded waits); valid once that launch has completed
extern "C" int ccc_ddp_last_call_aborted(const ccc_ddp_t * h)
{
  if(!h) return -1;
  return h->abort_host && __atomic_load_n(h->abort_host, __ATOMIC_RELAXED) != 0 ? 1 : 0;
}
