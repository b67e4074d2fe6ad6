// common.hip -- error reporting and device selection of libccc_amd.
#include "common.h"

#include <cstring>

namespace ccc_amd
{
std::string & last_error()
{
  static thread_local std::string err;
  return err;
}

__global__ void zero_words_kernel(unsigned * p, int words)
{
  for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) p[i] = 0u;
}

int zero_words(void * p, int words, void * stream)
{
  if(words <= 0) return CCC_OK;
  const int blocks = words < 256 * 64 ? (words + 255) / 256 : 64;
  hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     static_cast<unsigned *>(p), words);
  CCC_HIP_CHECK(hipGetLastError());
  return CCC_OK;
}

int refuse_growth_in_capture(void * stream, const char * who)
{
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if(stream && hipStreamIsCapturing(reinterpret_cast<hipStream_t>(stream), &st) == hipSuccess &&
     st != hipStreamCaptureStatusNone)
    return fail(CCC_ERR_INVALID_ARGUMENT,
                "%s: the workspace has to grow for this batch size, which cannot happen inside a stream capture; "
                "call once eagerly with the largest batch before capturing",
                who);
  return CCC_OK;
}

int select_device(int device)
{
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if(e != hipSuccess || count <= 0)
    return fail(CCC_ERR_NO_DEVICE, "no HIP device visible (%s): libccc_amd has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if(device < 0 || device >= count)
    return fail(CCC_ERR_INVALID_ARGUMENT, "device %d out of range [0, %d)", device, count);
  hipDeviceProp_t prop;
  CCC_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if(std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(CCC_ERR_NO_DEVICE, "device %d is %s; libccc_amd is built for gfx950 (MI355X) only", device,
                prop.gcnArchName);
  return CCC_OK;
}
} // namespace ccc_amd

extern "C" const char * ccc_last_error_string(void)
{
  return ccc_amd::last_error().c_str();
}

extern "C" int ccc_abi_version(void)
{
  // 2: ccc_ddp_config_t::reg_type, sharded entry points, ccc_device_count; 3: max_ridges in ccc_ddp_params_t /
  // ccc_xy_params_t; 4: ccc_ddp_config_t::warm_start_guard (the struct grew), CCC_DDP_STATUS_WARM_REPLACED_BIT
  return CCC_ABI_VERSION;
}

extern "C" int ccc_device_count(void)
{
  int count = 0;
  if(hipGetDeviceCount(&count) != hipSuccess) return 0;
  int ok = 0;
  for(int d = 0; d < count; d++)
  {
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}
