// wrench.hip -- the step AFTER LinearMpcXY / DdpCentroidal / DdpSingleRigidBody::planOnce: the
// planned force scales of the first horizon step turned into the total wrench the simulator (or the robot's wrench
// distribution) consumes,
//   ForceColl::calcTotalWrench(motion_param.contact_list, planned_force_scales, moment_origin)
//   (external dependency; call sites tests/src/TestLinearMpcXY.cpp:119-120, TestDdpCentroidal.cpp:125-126,
//    TestDdpSingleRigidBody.cpp:141-142):  force = sum_r s_r ridge_r,  moment = sum_r s_r (vertex_r - origin) x ridge_r,
// with the contact list flattened contact -> vertex -> ridge exactly as the planners take it (include/ccc_amd.h).
// One instance per lane; inputs are read once (24 + 24 + 8 bytes per ridge): a pure HBM-streaming epilogue that keeps the
// planned scales on the device between the planner and a device-side simulator.
#include "common.h"

namespace ccc_amd
{
__global__ __launch_bounds__(256) void total_wrench_kernel(long n, int M, const int * __restrict__ dim,
                                                           const double * __restrict__ vertex,
                                                           const double * __restrict__ ridge,
                                                           const double * __restrict__ scales, int scale_stride,
                                                           const double * __restrict__ origin, double * __restrict__ wrench)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const int m = dim[k];
  const double ox = origin[k * 3 + 0], oy = origin[k * 3 + 1], oz = origin[k * 3 + 2];
  double f[3] = {0, 0, 0}, t[3] = {0, 0, 0};
  for(int r = 0; r < m; r++)
  {
    const double s = scales[(size_t)k * scale_stride + r];
    const double * v = vertex + ((size_t)k * M + r) * 3;
    const double * d = ridge + ((size_t)k * M + r) * 3;
    const double px = v[0] - ox, py = v[1] - oy, pz = v[2] - oz;
    f[0] += s * d[0];
    f[1] += s * d[1];
    f[2] += s * d[2];
    t[0] += s * (py * d[2] - pz * d[1]);
    t[1] += s * (pz * d[0] - px * d[2]);
    t[2] += s * (px * d[1] - py * d[0]);
  }
  // sva::ForceVecd::vector() order: [couple; force]
  for(int a = 0; a < 3; a++)
  {
    wrench[k * 6 + a] = t[a];
    wrench[k * 6 + 3 + a] = f[a];
  }
}
} // namespace ccc_amd

using namespace ccc_amd;

extern "C" int ccc_total_wrench_device(int64_t n, int max_ridges, const int32_t * dim, const double * vertex,
                                       const double * ridge, const double * scales, int scale_stride,
                                       const double * origin, double * wrench, void * stream)
{
  if(n < 0) return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_total_wrench_device: n = %lld < 0", (long long)n);
  if(n == 0) return CCC_OK;
  if(!dim || !vertex || !ridge || !scales || !origin || !wrench)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_total_wrench_device: NULL argument");
  if(max_ridges <= 0 || scale_stride <= 0)
    return fail(CCC_ERR_INVALID_ARGUMENT, "ccc_total_wrench_device: max_ridges and scale_stride must be > 0");
  const int grid = (int)((n + 255) / 256);
  hipLaunchKernelGGL(total_wrench_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (long)n,
                     max_ridges, dim, vertex, ridge, scales, scale_stride, origin, wrench);
  CCC_HIP_CHECK(hipGetLastError());
  return CCC_OK;
}
