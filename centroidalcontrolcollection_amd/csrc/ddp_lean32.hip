// ddp_lean32.hip -- ccc_ddp_config_t::precision = 32 (BASELINE configs[4]: "DdpSingleRigidBody 12-state SRB ... fp32 with
// fp64 tolerance check"; the centroidal model takes the mode as well): the lean build of csrc/ddp_lean.hip with the backward pass's matrices STORED in single
// precision in LDS (value function, derivatives, Q blocks, Cholesky factor, gains) and every operation on them in
// double -- reads widen, writes round once.  Trajectories, rollouts, costs, the box-QP iterate and every line-search /
// termination decision stay in double.  12.6 KB of LDS per wavefront instead of 20.3.  Not a nmpc_ddp option; results are compared with
// the fp64 oracle through a tolerance (tests/test_ddp_gpu.py::test_srb_fp32_storage_against_fp64_oracle_config5).
#define CCC_DDP_LEAN 1
#define CCC_DDP_STORE_FLOAT 1
#include "ddp_core.h"

#include "ddp_batch.h"

namespace ccc_amd
{
// Three wavefronts per SIMD for the single-rigid-body model (168 VGPRs, 816 B of scratch; 13.6 KB of LDS: eleven per CU):
// 50.0 -> 51.6 k solves/s at config 5.  The centroidal model keeps two: at config 3's batch (4096 = two turns of the
// 2048 resident wavefronts) a third resident wavefront only slows the first turn down (15.0 -> 14.2 k).
template<int S, int M>
__global__ __launch_bounds__(64, (S == 12 ? 3 : 2)) void ddp_lean32_kernel(ddp_common::Params P, DdpBatch B, long n)
{
  __shared__ ddp_lean32::Mem<S, M> mem;
  const int N = P.N;
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
    ddp_lean32::Solver<S, M> solver(P, I, mem);
    solver.solve();
    __syncthreads();
  }
}

hipError_t launch_ddp_lean32(const ddp_common::Params & P, const DdpBatch & B, long n, int S, hipStream_t stream)
{
  const int grid = (int)(n < (1L << 22) ? n : (1L << 22)); // one workgroup per instance: the dispatcher balances
  if(S == 9)
    hipLaunchKernelGGL((ddp_lean32_kernel<9, 16>), dim3(grid), dim3(64), 0, stream, P, B, n);
  else
    hipLaunchKernelGGL((ddp_lean32_kernel<12, 16>), dim3(grid), dim3(64), 0, stream, P, B, n);
  return hipGetLastError();
}
} // namespace ccc_amd
