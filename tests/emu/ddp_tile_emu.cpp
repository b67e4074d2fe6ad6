// ddp_tile_emu.cpp -- TEST AID: compiles the product's tile-layout DDP kernel (csrc/ddp_tile.h, written against
// csrc/w64.h) for the host, where the 64 lanes of the wavefront run in lock step as 64-element arrays.  Lets the CPU
// test-suite check the kernel's arithmetic bit for bit against the oracle (oracle/ddp_tile.c) without a GPU.
// Never linked into, imported by or shipped with the product.
#include "../../centroidalcontrolcollection_amd/csrc/ddp_tile.h"

#include <cstdint>
#include <cstring>
#include <vector>

using namespace ccc_amd;
using ccc_amd::ddp_common::Params;

template<int S, int B, bool IPP = false>
static void run_one(const Params & P, const ddp_tile::Instance & I)
{
  static ddp_tile::Mem<S, B> mem;
  ddp_tile::Solver<S, B, IPP> solver(P, I, mem);
  solver.solve_instance();
}

// the same solve in slices of `slice` iterations: between two slices the state goes through suspend() / resume() and
// everything else a wavefront owns (LDS, trajectory slots, gains) is overwritten, as if another instance had used it
template<int S, int B, bool IPP = false>
static void run_sliced(const Params & P, const ddp_tile::Instance & I, int slice, std::vector<double> & xbuf, std::vector<double> & ubuf,
                       std::vector<double> & ks, std::vector<double> & Ks)
{
  static ddp_tile::Mem<S, B> mem;
  std::vector<double> sx((size_t)(P.N + 1) * S), ss(8);
  bool fresh = true;
  for(;;)
  {
    ddp_tile::Solver<S, B, IPP> solver(P, I, mem);
    if(fresh)
      solver.begin();
    else
      solver.resume(sx.data(), ss.data());
    fresh = false;
    if(solver.iterate(slice))
    {
      solver.finish();
      return;
    }
    solver.suspend(sx.data(), ss.data());
    std::memset(static_cast<void *>(&mem), 0xa5, sizeof(mem));
    for(auto * v : {&xbuf, &ubuf, &ks, &Ks})
      for(double & e : *v) e = -1.2345e300;
  }
}

extern "C" int ccc_ddp_tile_emu_lds_bytes(int S, int M)
{
  if(M == 16) return S == 9 ? (int)sizeof(ddp_tile::Mem<9, 1>) : (int)sizeof(ddp_tile::Mem<12, 1>);
  if(M == 32) return S == 9 ? (int)sizeof(ddp_tile::Mem<9, 2>) : (int)sizeof(ddp_tile::Mem<12, 2>);
  if(M == 64) return S == 9 ? (int)sizeof(ddp_tile::Mem<9, 4>) : (int)sizeof(ddp_tile::Mem<12, 4>);
  return -1;
}

extern "C" int ccc_ddp_tile_emu_plan_batch(const Params * P, int M, long n, const int * phase_dim, const double * phase_vertex,
                                            const double * phase_ridge, const int * step_phase, const double * ref_pos,
                                            const double * ref_ori, const double * inertia, const double * x0,
                                            const double * u_init, double * u_out, double * x_out, int * iters,
                                            int * status, double * cost, int slice)
{
  const int S = P->model == 0 ? 9 : 12, N = P->N, Pn = P->P;
  if(M != 16 && M != 32 && M != 64) return -1;
  std::vector<double> xbuf((size_t)ddp_tile::kSlots * (N + 1) * S), ubuf((size_t)ddp_tile::kSlots * N * M),
      ks((size_t)N * M), Ks((size_t)N * M * S);
  for(long b = 0; b < n; b++)
  {
    ddp_tile::Instance I;
    I.phase_dim = phase_dim + b * Pn;
    I.phase_vertex = phase_vertex + (size_t)b * Pn * M * 3;
    I.phase_ridge = phase_ridge + (size_t)b * Pn * M * 3;
    I.step_phase = step_phase + (size_t)b * N;
    I.ref_pos = ref_pos + (size_t)b * (N + 1) * 3;
    I.ref_ori = ref_ori ? ref_ori + (size_t)b * (N + 1) * 3 : nullptr;
    I.inertia = inertia ? inertia + (size_t)b * 9 * (P->inertia_per_phase ? Pn : 1) : nullptr;
    I.x0 = x0 + (size_t)b * S;
    I.u_init = u_init ? u_init + (size_t)b * N * M : nullptr;
    I.xbuf = xbuf.data();
    I.ubuf = ubuf.data();
    I.ks = ks.data();
    I.Ks = Ks.data();
    I.u_out = u_out + (size_t)b * N * M;
    I.x_out = x_out ? x_out + (size_t)b * (N + 1) * S : nullptr;
    I.out_iters = iters ? iters + b : nullptr;
    I.out_status = status ? status + b : nullptr;
    I.out_cost = cost ? cost + b : nullptr;
    if(S == 12 && P->inertia_per_phase)
    {
      // the builds with one inertia matrix per contact phase (ccc_ddp_params_t::inertia_per_phase)
      if(slice > 0)
      {
        if(M == 16) run_sliced<12, 1, true>(*P, I, slice, xbuf, ubuf, ks, Ks);
        else if(M == 32) run_sliced<12, 2, true>(*P, I, slice, xbuf, ubuf, ks, Ks);
        else run_sliced<12, 4, true>(*P, I, slice, xbuf, ubuf, ks, Ks);
      }
      else if(M == 16) run_one<12, 1, true>(*P, I);
      else if(M == 32) run_one<12, 2, true>(*P, I);
      else run_one<12, 4, true>(*P, I);
    }
    else if(slice > 0)
    {
      if(S == 9 && M == 16) run_sliced<9, 1>(*P, I, slice, xbuf, ubuf, ks, Ks);
      else if(S == 12 && M == 16) run_sliced<12, 1>(*P, I, slice, xbuf, ubuf, ks, Ks);
      else if(S == 9 && M == 32) run_sliced<9, 2>(*P, I, slice, xbuf, ubuf, ks, Ks);
      else if(S == 12 && M == 32) run_sliced<12, 2>(*P, I, slice, xbuf, ubuf, ks, Ks);
      else if(S == 9) run_sliced<9, 4>(*P, I, slice, xbuf, ubuf, ks, Ks);
      else run_sliced<12, 4>(*P, I, slice, xbuf, ubuf, ks, Ks);
    }
    else if(S == 9 && M == 16) run_one<9, 1>(*P, I);
    else if(S == 12 && M == 16) run_one<12, 1>(*P, I);
    else if(S == 9 && M == 32) run_one<9, 2>(*P, I);
    else if(S == 12 && M == 32) run_one<12, 2>(*P, I);
    else if(S == 9) run_one<9, 4>(*P, I);
    else run_one<12, 4>(*P, I);
  }
  return 0;
}
