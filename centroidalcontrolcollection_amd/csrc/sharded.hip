// sharded.hip -- one node, several GPUs behind the C-ABI: contiguous batch shards, one planner handle per device, an RCCL
// all-gather of the planned outputs for device-resident callers (include/ccc_amd.h "One node, several
// GPUs").  A class-agnostic SHARD GROUP (device list, one stream and one RCCL communicator per device, ordering against
// the caller's streams, the grouped in-place all-gather) carries LinearMpcZmp, LinearMpcXY and the DDP planners -- the
// classes BASELINE's configs put on eight GPUs.  Host-side only apart from one gather kernel: the planning kernels are
// those of csrc/zmp.hip, csrc/xy.hip, csrc/ddp*.hip.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <mutex>
#include <thread>
#include <vector>

using namespace ccc_amd;

namespace
{
// RCCL entry points, resolved at first use (the library is not a link-time dependency of libccc_amd.so: a process that
// already carries an RCCL -- e.g. through torch -- keeps using that one)
struct Rccl
{
  void * lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char * (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl & rccl()
{
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for(const char * name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"})
    {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if(r.lib) break;
    }
    if(!r.lib) return;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString;
  });
  return r;
}
} // namespace

// ------------------------------------------------------------------------------------------------ the shard group
namespace
{
struct ShardGroup
{
  std::vector<int> devices;
  std::vector<hipStream_t> streams; // one per device: the plan kernels and the all-gather of that device run on it
  std::vector<hipEvent_t> events;   // one per device: orders the group's stream behind the caller's
  std::vector<ncclComm_t> comms;    // created by the first collective

  int create(const int * devs, int num, const char * who)
  {
    if(!devs || num <= 0) return fail(CCC_ERR_INVALID_ARGUMENT, "%s: empty device list", who);
    for(int a = 0; a < num; a++)
      for(int b = a + 1; b < num; b++)
        if(devs[a] == devs[b]) return fail(CCC_ERR_INVALID_ARGUMENT, "%s: device %d listed twice", who, devs[a]);
    devices.assign(devs, devs + num);
    streams.assign(num, nullptr);
    events.assign(num, nullptr);
    for(int r = 0; r < num; r++)
    {
      int rc = select_device(devices[r]);
      if(rc != CCC_OK) return rc;
      DeviceGuard g(devices[r]);
      if(!g.ok || hipStreamCreateWithFlags(&streams[r], hipStreamNonBlocking) != hipSuccess
         || hipEventCreateWithFlags(&events[r], hipEventDisableTiming) != hipSuccess)
        return fail(CCC_ERR_HIP, "%s: cannot create a stream on device %d", who, devices[r]);
    }
    return CCC_OK;
  }
  void destroy()
  {
    for(ncclComm_t c : comms)
      if(c) (void)rccl().CommDestroy(c);
    comms.clear();
    for(size_t r = 0; r < devices.size(); r++)
    {
      DeviceGuard g(devices[r]);
      if(r < events.size() && events[r]) (void)hipEventDestroy(events[r]);
      if(r < streams.size() && streams[r]) (void)hipStreamDestroy(streams[r]);
    }
  }
  int size() const { return static_cast<int>(devices.size()); }
  // The group's stream of device r waits for what the caller has enqueued so far on ITS stream of that device
  // (caller_streams[r]; a NULL list or entry means the legacy default stream, which is what e.g. torch computes on unless
  // told otherwise) -- inputs produced by asynchronous kernels are complete before the plan kernels read them, and
  // buffers the caller is still filling are not overwritten early.
  int order_after_caller(void * const * caller_streams)
  {
    for(int r = 0; r < size(); r++)
    {
      DeviceGuard g(devices[r]);
      hipStream_t cs = caller_streams ? reinterpret_cast<hipStream_t>(caller_streams[r]) : nullptr;
      CCC_HIP_CHECK(hipEventRecord(events[r], cs));
      CCC_HIP_CHECK(hipStreamWaitEvent(streams[r], events[r], 0));
    }
    return CCC_OK;
  }
  // in-place all-gather of `count` doubles per device: slot r of buf[r] (count doubles at offset r * count) is device
  // r's contribution; afterwards every buf[r] holds all D slots.  Grouped, one call per device on its stream.
  int all_gather(double * const * buf, size_t count, const char * who)
  {
    Rccl & R = rccl();
    if(!R.ok) return fail(CCC_ERR_UNSUPPORTED, "%s: librccl.so could not be loaded", who);
    const int D = size();
    if(comms.empty())
    {
      comms.assign(D, nullptr);
      ncclResult_t e = R.CommInitAll(comms.data(), D, devices.data());
      if(e != ncclSuccess)
      {
        comms.clear();
        return fail(CCC_ERR_HIP, "ncclCommInitAll over %d device(s) failed: %s", D, R.GetErrorString(e));
      }
    }
    ncclResult_t e = R.GroupStart();
    for(int r = 0; r < D && e == ncclSuccess; r++)
    {
      DeviceGuard g(devices[r]);
      e = R.AllGather(buf[r] + r * count, buf[r], count, ncclDouble, comms[r], streams[r]);
    }
    ncclResult_t e2 = R.GroupEnd();
    if(e == ncclSuccess) e = e2;
    if(e != ncclSuccess) return fail(CCC_ERR_HIP, "%s: ncclAllGather failed: %s", who, R.GetErrorString(e));
    return CCC_OK;
  }
  // every stream of the group, whatever the others report (error paths)
  void drain()
  {
    for(int r = 0; r < size(); r++)
    {
      DeviceGuard g(devices[r]);
      (void)hipStreamSynchronize(streams[r]);
    }
  }
  int synchronize()
  {
    for(int r = 0; r < size(); r++)
    {
      DeviceGuard g(devices[r]);
      CCC_HIP_CHECK(hipStreamSynchronize(streams[r]));
    }
    return CCC_OK;
  }
};

// first horizon step of a planned input sequence [n][N][M] -> [n][M] (what the DDP planners' all-gather carries)
__global__ void first_step_kernel(long n, int N, int M, const double * u, double * u0)
{
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(id >= n * M) return;
  u0[id] = u[(id / M) * N * M + id % M];
}
} // namespace

// An entry documented as synchronous must not return while kernels it enqueued on other devices still write the
// caller's buffers: on any error after the first launch, drain the group's streams, keep the error.
static int fail_after_launch(ShardGroup & grp, int rc)
{
  const std::string keep = last_error();
  grp.drain();
  last_error() = keep;
  return rc;
}

struct ccc_zmp_sharded
{
  ShardGroup grp;
  std::vector<ccc_zmp_t *> handles;
};
struct ccc_xy_sharded
{
  ShardGroup grp;
  std::vector<ccc_xy_t *> handles;
};
struct ccc_ddp_sharded
{
  ShardGroup grp;
  std::vector<ccc_ddp_t *> handles;
  int N = 0, M = 0;
};

extern "C" int ccc_shard_bounds(int64_t n, int num_shards, int shard, int64_t * begin, int64_t * end)
{
  if(n < 0 || num_shards <= 0 || shard < 0 || shard >= num_shards || !begin || !end)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_shard_bounds: bad arguments");
  const int64_t base = n / num_shards, extra = n % num_shards;
  *begin = shard * base + std::min<int64_t>(shard, extra);
  *end = *begin + base + (shard < extra ? 1 : 0);
  return CCC_OK;
}

// ------------------------------------------------------------------------------------------------ LinearMpcZmp
extern "C" void ccc_zmp_sharded_destroy(ccc_zmp_sharded_t * h)
{
  if(!h) return;
  h->grp.destroy();
  for(ccc_zmp_t * z : h->handles)
    if(z) ccc_zmp_destroy(z);
  delete h;
}

extern "C" int ccc_zmp_sharded_create(double com_height, double horizon_duration, double horizon_dt, const int * devices,
                                      int num_devices, ccc_zmp_sharded_t ** out)
{
  if(!out) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_create: out is NULL");
  *out = nullptr;
  ccc_zmp_sharded * h = new ccc_zmp_sharded();
  int rc = h->grp.create(devices, num_devices, "ccc_zmp_sharded_create");
  h->handles.assign(rc == CCC_OK ? num_devices : 0, nullptr);
  for(int r = 0; rc == CCC_OK && r < num_devices; r++)
    rc = ccc_zmp_create(com_height, horizon_duration, horizon_dt, devices[r], &h->handles[r]);
  if(rc != CCC_OK)
  {
    const std::string keep = last_error();
    ccc_zmp_sharded_destroy(h);
    last_error() = keep;
    return rc;
  }
  *out = h;
  return CCC_OK;
}

extern "C" int ccc_zmp_sharded_num_devices(const ccc_zmp_sharded_t * h)
{
  return h ? h->grp.size() : -1;
}

extern "C" int ccc_zmp_sharded_plan_batch(ccc_zmp_sharded_t * h, int64_t n, const double * x0, const double * zlim,
                                          double control_dt, double * zmp, int32_t * status)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_plan_batch: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_plan_batch: n < 0");
  if(n == 0) return CCC_OK;
  if(!x0 || !zlim || !zmp) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_plan_batch: NULL x0/zlim/zmp");
  const int D = h->grp.size();
  const int64_t N = ccc_zmp_horizon_steps(h->handles[0]);
  std::vector<int> rcs(D, CCC_OK);
  std::vector<std::string> errs(D);
  std::vector<std::thread> workers;
  for(int r = 0; r < D; r++)
  {
    workers.emplace_back([&, r] {
      int64_t b = 0, e = 0;
      (void)ccc_shard_bounds(n, D, r, &b, &e);
      if(e == b) return;
      rcs[r] = ccc_zmp_plan_batch(h->handles[r], e - b, x0 + b * 6, zlim + b * 4 * N, control_dt, zmp + b * 2, nullptr,
                                  status ? status + b * 2 : nullptr);
      if(rcs[r] != CCC_OK) errs[r] = last_error(); // the error text is per thread
    });
  }
  for(auto & w : workers) w.join();
  for(int r = 0; r < D; r++)
    if(rcs[r] != CCC_OK)
      return fail(rcs[r], "ccc_zmp_sharded_plan_batch: device %d: %s", h->grp.devices[r], errs[r].c_str());
  return CCC_OK;
}

extern "C" int ccc_zmp_sharded_plan_batch_device_ordered(ccc_zmp_sharded_t * h, int64_t n_per_device,
                                                         const double * const * x0, const double * const * zlim,
                                                         double control_dt, double * const * zmp_all,
                                                         int32_t * const * status, void * const * caller_streams)
{
  const char * who = "ccc_zmp_sharded_plan_batch_device";
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "%s: NULL handle", who);
  if(n_per_device <= 0) return fail(CCC_ERR_INVALID_ARGUMENT, "%s: n_per_device <= 0", who);
  if(!x0 || !zlim || !zmp_all) return fail(CCC_ERR_INVALID_ARGUMENT, "%s: NULL array list", who);
  const int D = h->grp.size();
  for(int r = 0; r < D; r++)
    if(!x0[r] || !zlim[r] || !zmp_all[r])
      return fail(CCC_ERR_INVALID_ARGUMENT, "%s: NULL array for device %d", who, h->grp.devices[r]);
  int rc = h->grp.order_after_caller(caller_streams);
  if(rc != CCC_OK) return rc;
  const size_t cnt = static_cast<size_t>(n_per_device) * 2;
  // every device plans its shard into its own slot of its zmp_all
  for(int r = 0; r < D; r++)
  {
    rc = ccc_zmp_plan_batch_device(h->handles[r], n_per_device, x0[r], zlim[r], control_dt, zmp_all[r] + r * cnt, nullptr,
                                   status ? status[r] : nullptr, h->grp.streams[r]);
    if(rc != CCC_OK) return fail_after_launch(h->grp, rc);
  }
  rc = h->grp.all_gather(zmp_all, cnt, who);
  if(rc != CCC_OK) return fail_after_launch(h->grp, rc);
  return h->grp.synchronize();
}

extern "C" int ccc_zmp_sharded_plan_batch_device(ccc_zmp_sharded_t * h, int64_t n_per_device, const double * const * x0,
                                                 const double * const * zlim, double control_dt,
                                                 double * const * zmp_all, int32_t * const * status)
{
  return ccc_zmp_sharded_plan_batch_device_ordered(h, n_per_device, x0, zlim, control_dt, zmp_all, status, nullptr);
}

// ------------------------------------------------------------------------------------------------ LinearMpcXY
extern "C" void ccc_xy_sharded_destroy(ccc_xy_sharded_t * h)
{
  if(!h) return;
  h->grp.destroy();
  for(ccc_xy_t * z : h->handles)
    if(z) ccc_xy_destroy(z);
  delete h;
}

extern "C" int ccc_xy_sharded_create(const ccc_xy_params_t * params, const int * devices, int num_devices,
                                     ccc_xy_sharded_t ** out)
{
  if(!out || !params) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_xy_sharded_create: NULL argument");
  *out = nullptr;
  ccc_xy_sharded * h = new ccc_xy_sharded();
  int rc = h->grp.create(devices, num_devices, "ccc_xy_sharded_create");
  h->handles.assign(rc == CCC_OK ? num_devices : 0, nullptr);
  for(int r = 0; rc == CCC_OK && r < num_devices; r++) rc = ccc_xy_create(params, devices[r], &h->handles[r]);
  if(rc != CCC_OK)
  {
    const std::string keep = last_error();
    ccc_xy_sharded_destroy(h);
    last_error() = keep;
    return rc;
  }
  *out = h;
  return CCC_OK;
}

extern "C" int ccc_xy_sharded_num_devices(const ccc_xy_sharded_t * h)
{
  return h ? h->grp.size() : -1;
}

extern "C" int ccc_xy_sharded_plan_batch_device(ccc_xy_sharded_t * h, int64_t n_per_device, const int32_t * const * dim,
                                                const double * const * vertex, const double * const * ridge,
                                                const double * const * com_z, const double * const * total_force_z,
                                                const double * const * ref_out, const double * const * x0,
                                                double * const * u0_all, int32_t * const * status,
                                                void * const * caller_streams)
{
  const char * who = "ccc_xy_sharded_plan_batch_device";
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "%s: NULL handle", who);
  if(n_per_device <= 0) return fail(CCC_ERR_INVALID_ARGUMENT, "%s: n_per_device <= 0", who);
  if(!dim || !vertex || !ridge || !com_z || !total_force_z || !ref_out || !x0 || !u0_all)
    return fail(CCC_ERR_INVALID_ARGUMENT, "%s: NULL array list", who);
  const int D = h->grp.size();
  ccc_xy_params_t prm;
  int rc = ccc_xy_get_params(h->handles[0], &prm, nullptr);
  if(rc != CCC_OK) return rc;
  const size_t cnt = static_cast<size_t>(n_per_device) * (prm.max_ridges ? prm.max_ridges : CCC_XY_MAX_RIDGES);
  // (every per-device pointer is checked BEFORE anything is enqueued)
  for(int r = 0; r < D; r++)
    if(!dim[r] || !vertex[r] || !ridge[r] || !com_z[r] || !total_force_z[r] || !ref_out[r] || !x0[r] || !u0_all[r])
      return fail(CCC_ERR_INVALID_ARGUMENT, "%s: NULL array for device %d", who, h->grp.devices[r]);
  rc = h->grp.order_after_caller(caller_streams);
  if(rc != CCC_OK) return rc;
  for(int r = 0; r < D; r++)
  {
    rc = ccc_xy_plan_batch_device(h->handles[r], n_per_device, dim[r], vertex[r], ridge[r], com_z[r], total_force_z[r],
                                  ref_out[r], x0[r], u0_all[r] + r * cnt, nullptr, status ? status[r] : nullptr,
                                  h->grp.streams[r]);
    if(rc != CCC_OK) return fail_after_launch(h->grp, rc);
  }
  rc = h->grp.all_gather(u0_all, cnt, who);
  if(rc != CCC_OK) return fail_after_launch(h->grp, rc);
  return h->grp.synchronize();
}

// ------------------------------------------------------------------------------------------------ DDP planners
extern "C" void ccc_ddp_sharded_destroy(ccc_ddp_sharded_t * h)
{
  if(!h) return;
  h->grp.destroy();
  for(ccc_ddp_t * z : h->handles)
    if(z) ccc_ddp_destroy(z);
  delete h;
}

extern "C" int ccc_ddp_sharded_create(const ccc_ddp_params_t * params, const ccc_ddp_config_t * config,
                                      const int * devices, int num_devices, ccc_ddp_sharded_t ** out)
{
  if(!out || !params) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_sharded_create: NULL argument");
  *out = nullptr;
  ccc_ddp_sharded * h = new ccc_ddp_sharded();
  int rc = h->grp.create(devices, num_devices, "ccc_ddp_sharded_create");
  h->handles.assign(rc == CCC_OK ? num_devices : 0, nullptr);
  for(int r = 0; rc == CCC_OK && r < num_devices; r++)
  {
    rc = ccc_ddp_create(params, devices[r], &h->handles[r]);
    if(rc == CCC_OK && config) rc = ccc_ddp_set_config(h->handles[r], config);
  }
  if(rc == CCC_OK)
  {
    ccc_ddp_params_t prm;
    rc = ccc_ddp_get_params(h->handles[0], &prm);
    h->N = prm.horizon_steps;
    h->M = prm.max_ridges;
  }
  if(rc != CCC_OK)
  {
    const std::string keep = last_error();
    ccc_ddp_sharded_destroy(h);
    last_error() = keep;
    return rc;
  }
  *out = h;
  return CCC_OK;
}

extern "C" int ccc_ddp_sharded_num_devices(const ccc_ddp_sharded_t * h)
{
  return h ? h->grp.size() : -1;
}

extern "C" int ccc_ddp_sharded_set_config(ccc_ddp_sharded_t * h, const ccc_ddp_config_t * config)
{
  if(!h || !config) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_sharded_set_config: NULL argument");
  for(ccc_ddp_t * d : h->handles)
  {
    int rc = ccc_ddp_set_config(d, config);
    if(rc != CCC_OK) return rc;
  }
  return CCC_OK;
}

// force_scale_limits_ of every shard's planner (ccc_ddp_set_limits: the reference reads the member at every solve)
extern "C" int ccc_ddp_sharded_set_limits(ccc_ddp_sharded_t * h, double lo, double hi)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_ddp_sharded_set_limits: NULL handle");
  for(ccc_ddp_t * d : h->handles)
  {
    int rc = ccc_ddp_set_limits(d, lo, hi);
    if(rc != CCC_OK) return rc;
  }
  return CCC_OK;
}

extern "C" int ccc_ddp_sharded_plan_batch_device(ccc_ddp_sharded_t * h, int64_t n_per_device,
                                                 const int32_t * const * phase_dim, const double * const * phase_vertex,
                                                 const double * const * phase_ridge, const int32_t * const * step_phase,
                                                 const double * const * ref_pos, const double * const * ref_ori,
                                                 const double * const * inertia, const double * const * x0,
                                                 const double * const * u_init, double * const * u_out,
                                                 double * const * u0_all, int32_t * const * iters,
                                                 int32_t * const * status, double * const * cost,
                                                 void * const * caller_streams)
{
  const char * who = "ccc_ddp_sharded_plan_batch_device";
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "%s: NULL handle", who);
  if(n_per_device <= 0) return fail(CCC_ERR_INVALID_ARGUMENT, "%s: n_per_device <= 0", who);
  if(!phase_dim || !phase_vertex || !phase_ridge || !step_phase || !ref_pos || !x0 || !u_out || !u0_all)
    return fail(CCC_ERR_INVALID_ARGUMENT, "%s: NULL array list", who);
  const int D = h->grp.size();
  const size_t cnt = static_cast<size_t>(n_per_device) * h->M;
  for(int r = 0; r < D; r++)
    if(!phase_dim[r] || !phase_vertex[r] || !phase_ridge[r] || !step_phase[r] || !ref_pos[r] || !x0[r] || !u_out[r]
       || !u0_all[r])
      return fail(CCC_ERR_INVALID_ARGUMENT, "%s: NULL array for device %d", who, h->grp.devices[r]);
  int rc = h->grp.order_after_caller(caller_streams);
  if(rc != CCC_OK) return rc;
  for(int r = 0; r < D; r++)
  {
    rc = ccc_ddp_plan_batch_device(h->handles[r], n_per_device, phase_dim[r], phase_vertex[r], phase_ridge[r],
                                   step_phase[r], ref_pos[r], ref_ori ? ref_ori[r] : nullptr,
                                   inertia ? inertia[r] : nullptr, x0[r], u_init ? u_init[r] : nullptr, u_out[r], nullptr,
                                   iters ? iters[r] : nullptr, status ? status[r] : nullptr, cost ? cost[r] : nullptr,
                                   h->grp.streams[r]);
    if(rc != CCC_OK) return fail_after_launch(h->grp, rc);
    DeviceGuard g(h->grp.devices[r]);
    const long total = static_cast<long>(cnt);
    hipLaunchKernelGGL(first_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->grp.streams[r],
                       (long)n_per_device, h->N, h->M, u_out[r], u0_all[r] + r * cnt);
    const hipError_t le = hipGetLastError();
    if(le != hipSuccess)
      return fail_after_launch(h->grp, fail(CCC_ERR_HIP, "%s: first_step_kernel: %s", who, hipGetErrorString(le)));
  }
  rc = h->grp.all_gather(u0_all, cnt, who);
  if(rc != CCC_OK) return fail_after_launch(h->grp, rc);
  return h->grp.synchronize();
}
