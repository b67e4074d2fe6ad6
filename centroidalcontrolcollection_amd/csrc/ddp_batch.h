// ddp_batch.h -- what csrc/ddp.hip hands to the DDP kernel launch (csrc/ddp_tile.hip): the batch-constant parameters,
// the batch's arrays (C-ABI layouts of ccc_amd.h) and the launch interface.
#pragma once

#if defined(__HIPCC__)
#  include <hip/hip_runtime.h>
#endif
#include <cstddef>

namespace ccc_amd
{
namespace ddp_common
{
// Batch-constant parameters (by value to the kernel)
struct Params
{
  int model; // 0 = DdpCentroidal (S = 9), 1 = DdpSingleRigidBody (S = 12)
  int N, P;  // horizon steps, contact phases per instance
  double mass, dt;
  double w_run[12], w_term[12], w_force; // WeightParam
  double flo, fhi;                       // force_scale_limits_
  // nmpc_ddp configuration (SURVEY.md App. B.2 + overrides of src/DdpCentroidal.cpp:197-201)
  int max_iter;
  double lambda0, dlambda0, lambda_factor, lambda_min, lambda_max;
  double k_rel_norm_thre, lambda_thre, ratio_thre, cost_thre;
  double alpha[11];
  int reg_type;   // 1: Quu_F + lambda I, 2: Vxx + lambda I (oracle/ddp.c)
  int warm_guard; // ccc_ddp_config_t::warm_start_guard
};
} // namespace ddp_common

struct DdpBatch
{
  const int * phase_dim;
  const double * phase_vertex;
  const double * phase_ridge;
  const int * step_phase;
  const double * ref_pos;
  const double * ref_ori;
  const double * inertia;
  const double * x0;
  const double * u_init;
  double * u_out;
  double * x_out; // or nullptr
  int * iters;
  int * status;
  double * cost;
};

#if defined(__HIPCC__)
// csrc/ddp_tile.hip: the tile build (csrc/ddp_tile.h), S in {9, 12}, M in {16, 32, 64} (ridge stride of the arrays = the
// handle's max_ridges), any number of contact phases and horizon steps, reg_type 1 and 2.  One resident set of workgroups
// (ddp_tile_grid) pulls instances from a ticket counter; ws = grid x ddp_tile_ws_doubles(N, S, M) doubles of workspace,
// ticket = one unsigned in device memory (reset by the launch)
size_t ddp_tile_ws_doubles(int N, int S, int M);
int ddp_tile_grid(long n, int M, int num_cu);
hipError_t launch_ddp_tile(const ddp_common::Params & P, const DdpBatch & B, double * ws, unsigned * ticket, int grid, long n,
                           int S, int M, hipStream_t stream);
#endif
} // namespace ccc_amd
