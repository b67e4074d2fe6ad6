// ddp_batch.h -- what csrc/ddp.hip hands to a DDP kernel launch: the batch's arrays (C-ABI layouts of ccc_amd.h) plus the
// handle's workspaces.  Shared by the builds (csrc/ddp.hip, csrc/ddp_tile.hip, csrc/ddp_lean32.hip).
#pragma once

#include <hip/hip_runtime.h>

namespace ccc_amd
{
namespace ddp_common
{
struct Params;
}

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
  double * x_out; // may alias the workspace
  double *xc, *uc, *ks, *Ks;
  int * iters;
  int * status;
  double * cost;
};

// csrc/ddp_lean32.hip: the lean build with single-precision storage of the backward pass (precision = 32).
hipError_t launch_ddp_lean32(const ddp_common::Params & P, const DdpBatch & B, long n, int S, hipStream_t stream);

// csrc/ddp_tile.hip: the tile build (csrc/ddp_tile.h), S in {9, 12}, M in {16, 32, 64} (ridge stride of the arrays = the
// handle's max_ridges), any number of contact phases and horizon steps, reg_type 1 and 2.  One resident set of workgroups
// (ddp_tile_grid) pulls instances from a ticket counter; ws = grid x ddp_tile_ws_doubles(N, S, M) doubles of workspace,
// ticket = one unsigned in device memory (reset by the launch)
size_t ddp_tile_ws_doubles(int N, int S, int M);
int ddp_tile_grid(long n, int M, int num_cu);
hipError_t launch_ddp_tile(const ddp_common::Params & P, const DdpBatch & B, double * ws, unsigned * ticket, int grid, long n,
                           int S, int M, hipStream_t stream);
} // namespace ccc_amd
