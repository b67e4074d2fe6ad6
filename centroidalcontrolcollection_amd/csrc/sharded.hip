// sharded.hip -- one node, several GPUs behind the C-ABI: contiguous batch shards, one ccc_zmp_t per device, an RCCL
// all-gather of the planned ZMPs for device-resident callers (include/ccc_amd.h "One node, several
// GPUs").  Host-side only: the kernels are those of csrc/zmp.hip.
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

struct ccc_zmp_sharded
{
  std::vector<int> devices;
  std::vector<ccc_zmp_t *> handles;
  std::vector<hipStream_t> streams; // device entry: one stream per device
  std::vector<ncclComm_t> comms;    // created by the first device-resident call
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

extern "C" void ccc_zmp_sharded_destroy(ccc_zmp_sharded_t * h)
{
  if(!h) return;
  for(ncclComm_t c : h->comms)
    if(c) (void)rccl().CommDestroy(c);
  for(size_t r = 0; r < h->handles.size(); r++)
  {
    if(r < h->streams.size() && h->streams[r])
    {
      DeviceGuard g(h->devices[r]);
      (void)hipStreamDestroy(h->streams[r]);
    }
    if(h->handles[r]) ccc_zmp_destroy(h->handles[r]);
  }
  delete h;
}

extern "C" int ccc_zmp_sharded_create(double com_height, double horizon_duration, double horizon_dt, const int * devices,
                                      int num_devices, ccc_zmp_sharded_t ** out)
{
  if(!out) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_create: out is NULL");
  *out = nullptr;
  if(!devices || num_devices <= 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_create: empty device list");
  for(int a = 0; a < num_devices; a++)
    for(int b = a + 1; b < num_devices; b++)
      if(devices[a] == devices[b])
        return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_create: device %d listed twice", devices[a]);
  ccc_zmp_sharded * h = new ccc_zmp_sharded();
  h->devices.assign(devices, devices + num_devices);
  h->handles.assign(num_devices, nullptr);
  h->streams.assign(num_devices, nullptr);
  for(int r = 0; r < num_devices; r++)
  {
    int rc = ccc_zmp_create(com_height, horizon_duration, horizon_dt, devices[r], &h->handles[r]);
    if(rc == CCC_OK)
    {
      DeviceGuard g(devices[r]);
      if(!g.ok || hipStreamCreateWithFlags(&h->streams[r], hipStreamNonBlocking) != hipSuccess)
        rc = fail(CCC_ERR_HIP, "ccc_zmp_sharded_create: cannot create a stream on device %d", devices[r]);
    }
    if(rc != CCC_OK)
    {
      const std::string keep = last_error();
      ccc_zmp_sharded_destroy(h);
      last_error() = keep;
      return rc;
    }
  }
  *out = h;
  return CCC_OK;
}

extern "C" int ccc_zmp_sharded_num_devices(const ccc_zmp_sharded_t * h)
{
  return h ? static_cast<int>(h->devices.size()) : -1;
}

extern "C" int ccc_zmp_sharded_plan_batch(ccc_zmp_sharded_t * h, int64_t n, const double * x0, const double * zlim,
                                          double control_dt, double * zmp, int32_t * status)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_plan_batch: NULL handle");
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_plan_batch: n < 0");
  if(n == 0) return CCC_OK;
  if(!x0 || !zlim || !zmp) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_plan_batch: NULL x0/zlim/zmp");
  const int D = static_cast<int>(h->devices.size());
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
    if(rcs[r] != CCC_OK) return fail(rcs[r], "ccc_zmp_sharded_plan_batch: device %d: %s", h->devices[r], errs[r].c_str());
  return CCC_OK;
}

extern "C" int ccc_zmp_sharded_plan_batch_device(ccc_zmp_sharded_t * h, int64_t n_per_device, const double * const * x0,
                                                 const double * const * zlim, double control_dt,
                                                 double * const * zmp_all, int32_t * const * status)
{
  if(!h) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_plan_batch_device: NULL handle");
  if(n_per_device <= 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_plan_batch_device: n_per_device <= 0");
  if(!x0 || !zlim || !zmp_all) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_plan_batch_device: NULL array list");
  const int D = static_cast<int>(h->devices.size());
  Rccl & R = rccl();
  if(!R.ok) return fail(CCC_ERR_UNSUPPORTED, "ccc_zmp_sharded_plan_batch_device: librccl.so could not be loaded");
  if(h->comms.empty())
  {
    h->comms.assign(D, nullptr);
    ncclResult_t e = R.CommInitAll(h->comms.data(), D, h->devices.data());
    if(e != ncclSuccess)
    {
      h->comms.clear();
      return fail(CCC_ERR_HIP, "ncclCommInitAll over %d device(s) failed: %s", D, R.GetErrorString(e));
    }
  }
  const size_t cnt = static_cast<size_t>(n_per_device) * 2;
  // every device plans its shard into its own slot of its zmp_all
  for(int r = 0; r < D; r++)
  {
    if(!x0[r] || !zlim[r] || !zmp_all[r])
      return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_zmp_sharded_plan_batch_device: NULL array for device %d", h->devices[r]);
    int rc = ccc_zmp_plan_batch_device(h->handles[r], n_per_device, x0[r], zlim[r], control_dt, zmp_all[r] + r * cnt,
                                       nullptr, status ? status[r] : nullptr, h->streams[r]);
    if(rc != CCC_OK) return rc;
  }
  // in-place all-gather: the send buffer of rank r is its slot of the receive buffer
  ncclResult_t e = R.GroupStart();
  for(int r = 0; r < D && e == ncclSuccess; r++)
  {
    DeviceGuard g(h->devices[r]);
    e = R.AllGather(zmp_all[r] + r * cnt, zmp_all[r], cnt, ncclDouble, h->comms[r], h->streams[r]);
  }
  ncclResult_t e2 = R.GroupEnd();
  if(e == ncclSuccess) e = e2;
  if(e != ncclSuccess) return fail(CCC_ERR_HIP, "ncclAllGather failed: %s", R.GetErrorString(e));
  for(int r = 0; r < D; r++)
  {
    DeviceGuard g(h->devices[r]);
    CCC_HIP_CHECK(hipStreamSynchronize(h->streams[r]));
  }
  return CCC_OK;
}
