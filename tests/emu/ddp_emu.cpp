// ddp_emu.cpp -- TEST AID: compiles the product's wavefront DDP code (csrc/ddp_core.h) for the host, where the 64
// lanes of a phase run one after the other.  Lets the CPU test-suite check the phase logic of the HIP kernel against
// the oracle without a GPU.  Never linked into, imported by or shipped with the product.
// (The row-per-lane solver in the left-to-right arithmetic; the default kernel's emulation is ddp_tile_emu.cpp.)
#include "../../centroidalcontrolcollection_amd/csrc/ddp_core.h"

#include <cstdint>
#include <vector>

using namespace ccc_amd::CCC_DDP_NS;

extern "C" int ccc_ddp_emu_plan_batch(const Params * P, long n, int M, const int * phase_dim, const double * phase_vertex,
                                       const double * phase_ridge, const int * step_phase, const double * ref_pos,
                                       const double * ref_ori, const double * inertia, const double * x0,
                                       const double * u_init, double * u_out, double * x_out, int * iters, int * status,
                                       double * cost)
{
  if(M != 16) return 1;
  const int S = P->model == 0 ? 9 : 12, N = P->N, Pn = P->P;
  std::vector<double> xc((size_t)(N + 1) * S), uc((size_t)N * M), ks((size_t)N * M), Ks((size_t)N * M * S),
      xs((size_t)(N + 1) * S);
  for(long b = 0; b < n; b++)
  {
    Instance I;
    I.phase_dim = phase_dim + b * Pn;
    I.phase_vertex = phase_vertex + (size_t)b * Pn * M * 3;
    I.phase_ridge = phase_ridge + (size_t)b * Pn * M * 3;
    I.step_phase = step_phase + (size_t)b * N;
    I.ref_pos = ref_pos + (size_t)b * (N + 1) * 3;
    I.ref_ori = ref_ori ? ref_ori + (size_t)b * (N + 1) * 3 : nullptr;
    I.inertia = inertia ? inertia + (size_t)b * 9 : nullptr;
    I.x0 = x0 + (size_t)b * S;
    I.u_init = u_init ? u_init + (size_t)b * N * M : nullptr;
    I.xs = x_out ? x_out + (size_t)b * (N + 1) * S : xs.data();
    I.us = u_out + (size_t)b * N * M;
    I.xc = xc.data();
    I.uc = uc.data();
    I.ks = ks.data();
    I.Ks = Ks.data();
    I.out_iters = iters ? iters + b : nullptr;
    I.out_status = status ? status + b : nullptr;
    I.out_cost = cost ? cost + b : nullptr;
    std::fill(ks.begin(), ks.end(), 0.0);
    std::fill(Ks.begin(), Ks.end(), 0.0);
    if(P->model == 0)
    {
      static Mem<9, 16> mem;
      Solver<9, 16>(*P, I, mem).solve();
    }
    else
    {
      static Mem<12, 16> mem;
      Solver<12, 16>(*P, I, mem).solve();
    }
  }
  return 0;
}
