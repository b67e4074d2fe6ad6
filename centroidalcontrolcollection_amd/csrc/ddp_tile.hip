// ddp_tile.hip -- the TILE build of the DDP planners (csrc/ddp_tile.h): CCC::DdpCentroidal / CCC::DdpSingleRigidBody with
// the default regularisation, one instance per wavefront, every matrix distributed over the 64 lanes.
//   M = 16 ridges per step (one surface contact):  <= 128 VGPRs and <= 10 KB of LDS per wavefront -> four wavefronts per
//                                                  SIMD, sixteen instances per CU
//   M = 32 (two surface contacts, double support): 16-20 KB of LDS -> two wavefronts per SIMD
//   M = 64 (up to four surface contacts):          ~50 KB of LDS -> three wavefronts per CU
// What ccc_ddp_plan_batch_device runs by default (csrc/ddp.hip); its arithmetic is the tile specification of
// oracle/ddp_tile.c, reproduced bit for bit (tests/test_ddp_gpu.py, tests/test_ddp_tile_emu.py).
// Replaces the same reference code as csrc/ddp.hip.
#include "ddp_tile.h"

namespace ccc_amd
{
// doubles of workspace per instance: trajectories [kSlots][N+1][S] and [kSlots][N][M], gains [N][M] and [N][M][S]
size_t ddp_tile_ws_doubles(int N, int S, int M)
{
  return (size_t)ddp_tile::kSlots * ((size_t)(N + 1) * S + (size_t)N * M) + (size_t)N * M + (size_t)N * M * S;
}

// wavefronts per SIMD the register budget is set for (measured, round 4: the kernel is bound by instruction issue, not by
// latency -- at 16 ridges two wavefronts per SIMD without spills (256 VGPRs) beat three (168, 240-490 B of scratch) and
// four (128, 430-660 B): config 3 50.9 k / 50.4 k / 46.6 k solves/s, config 5 shape 152 k / 139 k / 128 k)
#ifndef CCC_TILE_WAVES
#  define CCC_TILE_WAVES 2
#endif
#ifndef CCC_TILE_WAVES2
#  define CCC_TILE_WAVES2 2
#endif
#ifndef CCC_TILE_WAVES4
#  define CCC_TILE_WAVES4 1
#endif
template<int S, int NB>
__global__ __launch_bounds__(64, (NB == 1 ? CCC_TILE_WAVES : (NB == 2 ? CCC_TILE_WAVES2 : CCC_TILE_WAVES4))) void ddp_tile_kernel(
    ddp_common::Params P, DdpBatch B, double * ws, size_t ws_stride, long n, unsigned * ticket)
{
  __shared__ ddp_tile::Mem<S, NB> mem;
  __shared__ long next_b;
  constexpr int M = 16 * NB;
  const int N = P.N;
  // Work queue (round 4): the grid is one resident set of workgroups; each takes the next instance from a ticket counter
  // when it finishes one (DDP solves differ severalfold in length: the hardware dispatcher used to be the queue, at the
  // price of one workspace per INSTANCE -- now one per resident workgroup).
  for(;;)
  {
    if(threadIdx.x == 0) next_b = (long)atomicAdd(ticket, 1u);
    __syncthreads();
    const long b = next_b;
    __syncthreads();
    if(b >= n) break;
    ddp_tile::Instance I;
    I.phase_dim = B.phase_dim + b * P.P;
    I.phase_vertex = B.phase_vertex + b * P.P * M * 3;
    I.phase_ridge = B.phase_ridge + b * P.P * M * 3;
    I.step_phase = B.step_phase + b * N;
    I.ref_pos = B.ref_pos + b * (N + 1) * 3;
    I.ref_ori = B.ref_ori ? B.ref_ori + b * (N + 1) * 3 : nullptr;
    I.inertia = B.inertia ? B.inertia + b * 9 : nullptr;
    I.x0 = B.x0 + b * S;
    I.u_init = B.u_init ? B.u_init + b * N * M : nullptr;
    double * w = ws + (size_t)blockIdx.x * ws_stride;
    I.xbuf = w;
    w += (size_t)ddp_tile::kSlots * (N + 1) * S;
    I.ubuf = w;
    w += (size_t)ddp_tile::kSlots * N * M;
    I.ks = w;
    w += (size_t)N * M;
    I.Ks = w;
    I.u_out = B.u_out + b * N * M;
    I.x_out = B.x_out ? B.x_out + b * (N + 1) * S : nullptr;
    I.out_iters = B.iters ? B.iters + b : nullptr;
    I.out_status = B.status ? B.status + b : nullptr;
    I.out_cost = B.cost ? B.cost + b : nullptr;
    ddp_tile::Solver<S, NB> solver(P, I, mem);
    solver.solve_instance();
    __syncthreads();
  }
}

// workgroups of one resident set: wavefronts per SIMD (launch bounds above) x 4 SIMDs x CUs
int ddp_tile_grid(long n, int M, int num_cu)
{
  const int per_cu = 4 * (M == 16 ? CCC_TILE_WAVES : (M == 32 ? CCC_TILE_WAVES2 : CCC_TILE_WAVES4));
  const long resident = (long)per_cu * (num_cu > 0 ? num_cu : 256);
  return (int)(n < resident ? n : resident);
}

hipError_t launch_ddp_tile(const ddp_common::Params & P, const DdpBatch & B, double * ws, unsigned * ticket, int grid, long n,
                           int S, int M, hipStream_t stream)
{
  const size_t stride = ddp_tile_ws_doubles(P.N, S, M);
  hipError_t e = hipMemsetAsync(ticket, 0, sizeof(unsigned), stream);
  if(e != hipSuccess) return e;
#define CCC_TILE_LAUNCH(S_, NB_) hipLaunchKernelGGL((ddp_tile_kernel<S_, NB_>), dim3(grid), dim3(64), 0, stream, P, B, ws, stride, n, ticket)
  if(S == 9 && M == 16) CCC_TILE_LAUNCH(9, 1);
  else if(S == 12 && M == 16) CCC_TILE_LAUNCH(12, 1);
  else if(S == 9 && M == 32) CCC_TILE_LAUNCH(9, 2);
  else if(S == 12 && M == 32) CCC_TILE_LAUNCH(12, 2);
  else if(S == 9 && M == 64) CCC_TILE_LAUNCH(9, 4);
  else if(S == 12 && M == 64) CCC_TILE_LAUNCH(12, 4);
  else return hipErrorInvalidValue;
#undef CCC_TILE_LAUNCH
  return hipGetLastError();
}
} // namespace ccc_amd
