// ddp_lean.hip -- the LEAN build of the wavefront DDP solver (csrc/ddp_core.h): the fast build's problem sizes (<= 16
// ridges per step, <= 4 contact phases, <= 128 steps, tables in LDS) compiled for reg_type 1 only, which is the default
// (lambda regularises Quu: oracle/ddp.c).  Quu_F = Quu + lambda I and Qxu_r = Qxu are then not
// stored: 16.2 KB of LDS per wavefront for the centroidal model, 20.3 KB for the single-rigid-body model -- eight resident
// wavefronts per CU for both (the full build of csrc/ddp.hip, kept for reg_type 2, holds six of the latter).
// Replaces the same reference code as csrc/ddp.hip; bit-identical results (tests/test_ddp_gpu.py runs on this build).
#define CCC_DDP_LEAN 1
#include "ddp_core.h"

#include "ddp_batch.h"

namespace ccc_amd
{
template<int S, int M>
__global__ __launch_bounds__(64, 2) void ddp_lean_kernel(ddp_common::Params P, DdpBatch B, long n)
{
  __shared__ ddp_lean::Mem<S, M> mem;
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
    ddp_lean::Solver<S, M> solver(P, I, mem);
    solver.solve();
    __syncthreads();
  }
}

hipError_t launch_ddp_lean(const ddp_common::Params & P, const DdpBatch & B, long n, int S, hipStream_t stream)
{
  const int grid = (int)(n < (1L << 22) ? n : (1L << 22)); // one workgroup per instance: the dispatcher balances
  if(S == 9)
    hipLaunchKernelGGL((ddp_lean_kernel<9, 16>), dim3(grid), dim3(64), 0, stream, P, B, n);
  else
    hipLaunchKernelGGL((ddp_lean_kernel<12, 16>), dim3(grid), dim3(64), 0, stream, P, B, n);
  return hipGetLastError();
}
} // namespace ccc_amd
