/* DdpShimBase.h -- shared plumbing of the DdpCentroidal / DdpSingleRigidBody header shims: sampling the reference's
 * std::function callbacks into the flattened per-instance arrays of include/ccc_amd.h (contact lists -> contact
 * phases in contact -> vertex -> ridge order, /root/reference/src/DdpCentroidal.cpp:49-60) and the stand-ins for the
 * nmpc_ddp::DDPSolver members the reference's tests touch (config().max_iter, controlData().u_list,
 * traceDataList().back().iter: /root/reference/tests/src/TestDdpCentroidal.cpp:102-129).
 */
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../ccc_amd.h"
#include "EigenLite.h"

namespace CCC
{
namespace ddp_shim
{
struct TraceData
{
  int iter = 0;
  int status = 0; //!< 0 max_iter reached, 1 gradient small, 2 cost change small, -1 lambda exceeded lambda_max (new)
  /** (new) the warm-start guard (config().warm_start_guard, on by default, not a nmpc_ddp option) replaced
      InitialParam::u_list by zero inputs in this planOnce(): on exactly these calls the plan is not what the reference
      computes from the same u_list (/root/reference/src/DdpSingleRigidBody.cpp:299-303).  Never true with
      config().warm_start_guard = 0. */
  bool warm_start_replaced = false;
};

/** nmpc_ddp::DDPSolver::Configuration as the reference's callers see it: the solver parameters of ccc_ddp_config_t plus
    the two fields the CCC constructors set / the tests read (src/DdpCentroidal.cpp:197-198,
    tests/src/TestDdpCentroidal.cpp:105).  with_input_constraint is always true for these planners. */
struct Configuration : public ccc_ddp_config_t
{
  int horizon_steps = 0;
  bool with_input_constraint = true;
};

struct ControlData
{
  std::vector<VectorXd> u_list;
  std::vector<std::vector<double>> x_list;
};

/** Stand-in for nmpc_ddp::DDPSolver<S, Dynamic>: the members the reference's callers use. */
class Solver
{
public:
  Solver()
  {
    ccc_ddp_default_config(&config_);
  }
  Configuration & config()
  {
    return config_;
  }
  const ControlData & controlData() const
  {
    return control_data_;
  }
  const std::vector<TraceData> & traceDataList() const
  {
    return trace_data_list_;
  }

  Configuration config_;
  ControlData control_data_;
  std::vector<TraceData> trace_data_list_;
};

/** Flattened problem of ONE instance (the n = 1 case of the C-ABI arrays).  Contact phases are collected without a
    limit; pack() lays them out for the handle that can take them. */
struct Flat
{
  int N = 0, P = 0, M = 0; // P, M: layout of the packed arrays (set by pack())
  std::vector<int32_t> phase_dim, step_phase;
  std::vector<double> phase_vertex, phase_ridge, ref_pos, ref_ori, inertia; // inertia: [P][3][3] (inertia_per_phase)
  std::vector<int> dims; // input dimension per step

  void init(int n_steps)
  {
    N = n_steps;
    step_phase.assign(static_cast<size_t>(N), 0);
    ref_pos.assign(static_cast<size_t>(N + 1) * 3, 0.0);
    ref_ori.assign(static_cast<size_t>(N + 1) * 3, 0.0);
    inertia.clear();
    dims.assign(static_cast<size_t>(N), 0);
    phases_V_.clear();
    phases_R_.clear();
    phases_I_.clear();
    max_dim_ = 0;
  }

  /** Register the MotionParam of step i (find or append its phase): the contact list in src/DdpCentroidal.cpp:49-60 order
      and, for the single-rigid-body model, MotionParam::inertia_mat (row-major 3 x 3; nullptr for DdpCentroidal).  The
      reference reads motion_param_func_(t).inertia_mat at every step (src/DdpSingleRigidBody.cpp:56-57,120-123), so two
      steps share a phase only when contact list AND inertia matrix agree. */
  void setStepContacts(int i, const std::vector<std::shared_ptr<Contact>> & contact_list, const double * inertia9 = nullptr)
  {
    std::vector<double> V, R;
    for(const auto & contact : contact_list)
      for(const auto & vr : contact->vertexWithRidgeList_)
        for(const auto & ridge : vr.ridgeList)
          for(int a = 0; a < 3; a++)
          {
            V.push_back(vr.vertex[a]);
            R.push_back(ridge[a]);
          }
    const int m = static_cast<int>(V.size() / 3);
    if(m > CCC_DDP_MAX_RIDGES_MULTI)
      throw std::runtime_error("[DDP shim] " + std::to_string(m) + " ridges in a contact list, the kernels are built for "
                               + std::to_string(CCC_DDP_MAX_RIDGES_MULTI) + " (four 4-vertex surface contacts)");
    dims[static_cast<size_t>(i)] = m;
    max_dim_ = std::max(max_dim_, m);
    std::vector<double> In;
    if(inertia9) In.assign(inertia9, inertia9 + 9);
    for(size_t k = 0; k < phases_V_.size(); k++)
      if(phases_V_[k] == V && phases_R_[k] == R && phases_I_[k] == In)
      {
        step_phase[static_cast<size_t>(i)] = static_cast<int32_t>(k);
        return;
      }
    phases_V_.push_back(V);
    phases_R_.push_back(R);
    phases_I_.push_back(In);
    step_phase[static_cast<size_t>(i)] = static_cast<int32_t>(phases_V_.size() - 1);
  }

  int numPhases() const
  {
    return static_cast<int>(phases_V_.size());
  }
  int maxDim() const
  {
    return max_dim_;
  }

  /** Lay the phase tables out as [P][M][3] (the C-ABI layout of a handle with max_phases = P, max_ridges = M). */
  void pack(int n_phases, int ridge_stride)
  {
    P = n_phases;
    M = ridge_stride;
    phase_dim.assign(static_cast<size_t>(P), 0);
    phase_vertex.assign(static_cast<size_t>(P) * M * 3, 0.0);
    phase_ridge.assign(static_cast<size_t>(P) * M * 3, 0.0);
    inertia.assign(static_cast<size_t>(P) * 9, 0.0);
    for(int k = 0; k < P; k++) inertia[static_cast<size_t>(k) * 9] = inertia[static_cast<size_t>(k) * 9 + 4] = inertia[static_cast<size_t>(k) * 9 + 8] = 1.0;
    for(size_t k = 0; k < phases_V_.size(); k++)
    {
      if(!phases_I_[k].empty()) std::copy(phases_I_[k].begin(), phases_I_[k].end(), inertia.begin() + static_cast<long>(k) * 9);
      phase_dim[k] = static_cast<int32_t>(phases_V_[k].size() / 3);
      std::copy(phases_V_[k].begin(), phases_V_[k].end(), phase_vertex.begin() + static_cast<long>(k) * M * 3);
      std::copy(phases_R_[k].begin(), phases_R_[k].end(), phase_ridge.begin() + static_cast<long>(k) * M * 3);
    }
  }

private:
  std::vector<std::vector<double>> phases_V_, phases_R_, phases_I_;
  int max_dim_ = 0;
};

inline void check(int rc, const char * who)
{
  if(rc != CCC_OK) throw std::runtime_error(std::string("[") + who + "] " + ccc_last_error_string());
}

/** The library handles behind one shim object: the FAST one (<= 16 ridges per step, <= max_phases contact phases; made
    by the constructor) and, created on first need, TWINS with the smallest ridge stride that holds the sampled contact
    lists (16, 32 or 64 ridges; one phase per horizon step if need be) -- the reference takes any contact_list
    (src/DdpCentroidal.cpp:49-60), so must planOnce(). */
struct Handles
{
  ccc_ddp_params_t params{};
  int device = 0;
  std::shared_ptr<ccc_ddp_t> fast, twin[3];

  void create(const ccc_ddp_params_t & p, int dev, const char * who)
  {
    params = p;
    device = dev;
    ccc_ddp_t * h = nullptr;
    check(ccc_ddp_create(&params, device, &h), who);
    fast.reset(h, ccc_ddp_destroy);
  }

  /** Pack f for the handle that takes it and return that handle. */
  ccc_ddp_t * select(Flat & f, const char * who)
  {
    if(f.maxDim() <= CCC_DDP_MAX_RIDGES && f.numPhases() <= params.max_phases)
    {
      f.pack(params.max_phases, CCC_DDP_MAX_RIDGES);
      return fast.get();
    }
    const int k = f.maxDim() <= CCC_DDP_MAX_RIDGES ? 0 : (f.maxDim() <= CCC_DDP_MAX_RIDGES_WIDE ? 1 : 2);
    const int stride = k == 0 ? CCC_DDP_MAX_RIDGES : (k == 1 ? CCC_DDP_MAX_RIDGES_WIDE : CCC_DDP_MAX_RIDGES_MULTI);
    if(!twin[k])
    {
      ccc_ddp_params_t p = params;
      p.max_phases = params.horizon_steps;
      p.max_ridges = stride;
      ccc_ddp_t * h = nullptr;
      check(ccc_ddp_create(&p, device, &h), who);
      twin[k].reset(h, ccc_ddp_destroy);
    }
    f.pack(params.horizon_steps, stride);
    return twin[k].get();
  }
};

/** Run one instance through ccc_ddp_plan_batch and fill the solver stand-in; returns u_list[0]. */
inline VectorXd solveOne(ccc_ddp_t * h, Solver & solver, const Flat & f, bool srb, const std::vector<double> & x0,
                         const std::vector<VectorXd> & u_init_list, const std::array<double, 2> & force_scale_limits,
                         const char * who)
{
  const int N = f.N, M = f.M, S = static_cast<int>(x0.size());
  std::vector<double> u_init, u(static_cast<size_t>(N) * M, 0.0), x(static_cast<size_t>(N + 1) * S, 0.0);
  if(!u_init_list.empty())
  {
    if(static_cast<int>(u_init_list.size()) != N) throw std::runtime_error(std::string("[") + who + "] u_list length");
    u_init.assign(static_cast<size_t>(N) * M, 0.0);
    for(int i = 0; i < N; i++)
    {
      if(u_init_list[static_cast<size_t>(i)].size() != f.dims[static_cast<size_t>(i)])
        throw std::runtime_error(std::string("[") + who + "] u_list[i] size differs from inputDim");
      for(int r = 0; r < f.dims[static_cast<size_t>(i)]; r++)
        u_init[static_cast<size_t>(i) * M + r] = u_init_list[static_cast<size_t>(i)][r];
    }
  }
  int32_t iters = 0, status = 0;
  double cost = 0;
  check(ccc_ddp_set_config(h, &solver.config_), who);
  // force_scale_limits_ is read by the reference at every solve (the lambda of src/DdpCentroidal.cpp:202-210): its value NOW
  check(ccc_ddp_set_limits(h, force_scale_limits[0], force_scale_limits[1]), who);
  check(ccc_ddp_plan_batch(h, 1, f.phase_dim.data(), f.phase_vertex.data(), f.phase_ridge.data(), f.step_phase.data(),
                           f.ref_pos.data(), srb ? f.ref_ori.data() : nullptr, srb ? f.inertia.data() : nullptr,
                           x0.data(), u_init.empty() ? nullptr : u_init.data(), u.data(), x.data(), &iters, &status,
                           &cost),
        who);
  solver.control_data_.u_list.assign(static_cast<size_t>(N), VectorXd());
  for(int i = 0; i < N; i++)
  {
    VectorXd ui(f.dims[static_cast<size_t>(i)]);
    for(int r = 0; r < f.dims[static_cast<size_t>(i)]; r++) ui[r] = u[static_cast<size_t>(i) * M + r];
    solver.control_data_.u_list[static_cast<size_t>(i)] = ui;
  }
  solver.control_data_.x_list.assign(static_cast<size_t>(N + 1), std::vector<double>(static_cast<size_t>(S)));
  for(int i = 0; i <= N; i++)
    for(int a = 0; a < S; a++) solver.control_data_.x_list[static_cast<size_t>(i)][static_cast<size_t>(a)] = x[static_cast<size_t>(i) * S + a];
  TraceData td;
  td.iter = iters;
  td.status = CCC_DDP_STATUS_EXIT(status);
  td.warm_start_replaced = CCC_DDP_STATUS_WARM_REPLACED(status);
  solver.trace_data_list_.assign(1, td);
  if(td.status < 0)
  {
    // nmpc_ddp prints "[DDP] Failure: lambda is too large" and keeps the last accepted sequence; same here
    std::fprintf(stderr, "[%s] DDP did not converge: regularisation exceeded lambda_max after %d iteration(s); the "
                         "returned plan is the last accepted input sequence.\n",
                 who, iters);
  }
  return solver.control_data_.u_list[0];
}
} // namespace ddp_shim
} // namespace CCC
