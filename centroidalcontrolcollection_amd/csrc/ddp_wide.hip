// ddp_wide.hip -- the WIDE build of the wavefront DDP solver (csrc/ddp_core.h with -DCCC_DDP_WIDE semantics): contact
// lists the fast kernel of csrc/ddp.hip is not built for.
//
// Replaces the same reference code as csrc/ddp.hip (src/DdpCentroidal.cpp:213-237, src/DdpSingleRigidBody.cpp:283-307
// with the external nmpc_ddp solve), for the inputs where
//   - a horizon step carries more than 16 ridges: src/DdpCentroidal.cpp:49-60 and src/DdpSingleRigidBody.cpp:74-85
//     iterate an arbitrary contact_list; two surface contacts (double support) are 2 x 4 vertices x 4 ridges = 32;
//   - more than four distinct contact lists fall into one horizon (up to one per step), or the horizon is longer than
//     the 128 steps the fast kernel's LDS tables hold.
// One instance per wavefront.  Compiled for the default regularisation only (reg_type 1: Quu_F = Quu + lambda I and
// Qxu_r = Qxu are formed on the fly, not stored), the per-step matrices take 31.8 KB (centroidal) / 37.0 KB (single rigid
// body) of LDS at M = 32 -- Quu and the Cholesky factor are 32 x 33 doubles each -- so five / four wavefronts share a CU;
// the contact tables and the step -> phase map are read from global memory (L2-resident: 26 KB per instance at 10
// phases).  The code is csrc/ddp_core.h's register-resident device path with M lanes per row; its plain phase versions,
// which the CPU test-suite runs through tests/emu for this table layout as well, and the device path both reproduce
// the oracle bit for bit.
#define CCC_DDP_WIDE 1
#include "ddp_core.h"

#include "ddp_batch.h"

namespace ccc_amd
{
template<int S, int M>
__global__ __launch_bounds__(64, (M == 16 ? 2 : 1)) void ddp_wide_kernel(ddp_common::Params P, DdpBatch B, long n)
{
  __shared__ ddp_wide::Mem<S, M> mem;
  const long N = P.N;
  for(long b = blockIdx.x; b < n; b += gridDim.x)
  {
    ddp_common::Instance I;
    I.phase_dim = B.phase_dim + b * P.P;
    I.phase_vertex = B.phase_vertex + b * P.P * M * 3;
    I.phase_ridge = B.phase_ridge + b * P.P * M * 3;
    I.step_phase = B.step_phase + b * N;
    I.ref_pos = B.ref_pos + b * (N + 1) * 3;
    I.ref_ori = B.ref_ori ? B.ref_ori + b * (N + 1) * 3 : nullptr;
    I.inertia = B.inertia ? B.inertia + b * 9 : nullptr;
    I.x0 = B.x0 + b * S;
    I.u_init = B.u_init ? B.u_init + b * N * M : nullptr;
    I.xs = B.x_out + b * (N + 1) * S;
    I.us = B.u_out + b * N * M;
    I.xc = B.xc + b * (N + 1) * S;
    I.uc = B.uc + b * N * M;
    I.ks = B.ks + b * N * M;
    I.Ks = B.Ks + b * N * M * S;
    I.out_iters = B.iters ? B.iters + b : nullptr;
    I.out_status = B.status ? B.status + b : nullptr;
    I.out_cost = B.cost ? B.cost + b : nullptr;
    ddp_wide::Solver<S, M> solver(P, I, mem);
    solver.solve();
    __syncthreads();
  }
}

hipError_t launch_ddp_wide(const ddp_common::Params & P, const DdpBatch & B, long n, int S, int M, hipStream_t stream)
{
  const int grid = (int)(n < (1L << 22) ? n : (1L << 22));
  if(S == 9 && M == 16)
    hipLaunchKernelGGL((ddp_wide_kernel<9, 16>), dim3(grid), dim3(64), 0, stream, P, B, n);
  else if(S == 9)
    hipLaunchKernelGGL((ddp_wide_kernel<9, 32>), dim3(grid), dim3(64), 0, stream, P, B, n);
  else if(M == 16)
    hipLaunchKernelGGL((ddp_wide_kernel<12, 16>), dim3(grid), dim3(64), 0, stream, P, B, n);
  else
    hipLaunchKernelGGL((ddp_wide_kernel<12, 32>), dim3(grid), dim3(64), 0, stream, P, B, n);
  return hipGetLastError();
}
} // namespace ccc_amd
