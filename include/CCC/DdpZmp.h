/* DdpZmp.h -- drop-in header shim: the class surface of /root/reference/include/CCC/DdpZmp.h over the MI355X C-ABI
 * (include/ccc_amd.h).  Same namespace, class, RefData / PlannedData / WeightParam / InitialParam, constructor and
 * planOnce() signature; ddp_solver_->config() and ddp_solver_->controlData().u_list are kept as members of a small
 * stand-in (the solver itself runs on the device).  planOnceBatch() is new: n independent planOnce() problems in one
 * launch.
 */
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../ccc_amd.h"

#include "EigenLite.h"

namespace CCC
{
/** \brief DDP-based CoM-ZMP planner, batched on MI355X.  Mirrors CCC::DdpZmp, /root/reference/include/CCC/DdpZmp.h. */
class DdpZmp
{
public:
  /** \brief Reference data (DdpZmp.h:19-28). */
  struct RefData
  {
    Vector3d zmp = Vector3d::Zero(); //!< ZMP [m]
    double com_z = 0;                //!< CoM z position [m]
  };

  /** \brief Planned data (DdpZmp.h:31-40). */
  struct PlannedData
  {
    Vector2d zmp;       //!< ZMP [m]
    double force_z = 0; //!< Z force [N]
  };

  /** \brief Weight parameter (DdpZmp.h:43-88, same defaults). */
  struct WeightParam
  {
    double running_com_pos_z, running_zmp, running_force_z, terminal_com_pos_xy, terminal_com_pos_z, terminal_com_vel;
    WeightParam(double _running_com_pos_z = 1e2,
                double _running_zmp = 1e-1,
                double _running_force_z = 1e-4,
                double _terminal_com_pos_xy = 1.0,
                double _terminal_com_pos_z = 1e2,
                double _terminal_com_vel = 1.0)
    : running_com_pos_z(_running_com_pos_z), running_zmp(_running_zmp), running_force_z(_running_force_z),
      terminal_com_pos_xy(_terminal_com_pos_xy), terminal_com_pos_z(_terminal_com_pos_z),
      terminal_com_vel(_terminal_com_vel)
    {
    }
  };

  /** \brief Input of the DDP problem: [zmp_x, zmp_y, force_z] (DdpProblem::InputDimVector). */
  using InputDimVector = Vector3d;

  /** \brief Initial parameter (DdpZmp.h:247-266). */
  struct InitialParam
  {
    Vector3d pos = Vector3d::Zero();         //!< CoM position [m]
    Vector3d vel = Vector3d::Zero();         //!< CoM velocity [m/s]
    std::vector<InputDimVector> u_list = {}; //!< Initial guess of the input sequence (empty: zeros)
  };

  /** \brief Stand-in for nmpc_ddp::DDPSolver<6, 3>: the configuration and the planned input sequence. */
  struct Solver
  {
    struct ControlData
    {
      std::vector<InputDimVector> u_list;
    };
    struct TraceData
    {
      int iter = 0;
    };
    ccc_ddp_config_t & config()
    {
      return config_;
    }
    const ControlData & controlData() const
    {
      return control_;
    }
    const std::vector<TraceData> & traceDataList() const
    {
      return trace_;
    }
    ccc_ddp_config_t config_;
    ControlData control_;
    std::vector<TraceData> trace_ = std::vector<TraceData>(1);
  };

public:
  /** \brief Constructor (DdpZmp.h:277-282).
      \param mass robot mass [kg]
      \param horizon_dt discretization timestep in horizon [sec]
      \param horizon_steps number of steps in horizon
      \param weight_param objective weight parameter
      \param device HIP device ordinal (new) */
  DdpZmp(double mass, double horizon_dt, int horizon_steps, const WeightParam & weight_param = WeightParam(), int device = 0)
  : ddp_solver_(std::make_shared<Solver>()), mass_(mass), horizon_dt_(horizon_dt), horizon_steps_(horizon_steps)
  {
    const double w[6] = {weight_param.running_com_pos_z,   weight_param.running_zmp,        weight_param.running_force_z,
                         weight_param.terminal_com_pos_xy, weight_param.terminal_com_pos_z, weight_param.terminal_com_vel};
    ccc_ddpzmp_t * h = nullptr;
    check(ccc_ddpzmp_create(mass, horizon_dt, horizon_steps, w, device, &h));
    handle_.reset(h, ccc_ddpzmp_destroy);
    ccc_ddpzmp_default_config(&ddp_solver_->config_);
  }

  /** \brief Plan one step (DdpZmp.h:290-292, src/DdpZmp.cpp:156-174). */
  PlannedData planOnce(const std::function<RefData(double)> & ref_data_func,
                       const InitialParam & initial_param,
                       double current_time)
  {
    return planOnceBatch({ref_data_func}, {initial_param}, {current_time})[0];
  }

  /** \brief Plan n independent instances in one launch (new).  controlData().u_list / traceDataList() hold those of the
      LAST instance, as after n sequential planOnce() calls. */
  std::vector<PlannedData> planOnceBatch(const std::vector<std::function<RefData(double)>> & ref_data_funcs,
                                         const std::vector<InitialParam> & initial_params,
                                         const std::vector<double> & current_times)
  {
    const size_t n = ref_data_funcs.size(), N = static_cast<size_t>(horizon_steps_);
    if(initial_params.size() != n || current_times.size() != n)
    {
      throw std::runtime_error("[DdpZmp::planOnceBatch] argument sizes differ");
    }
    std::vector<double> ref(n * (N + 1) * 4), x0(n * 6), u_init(n * N * 3, 0.0), u_out(n * N * 3);
    std::vector<int32_t> iters(n);
    for(size_t k = 0; k < n; k++)
    {
      for(size_t i = 0; i <= N; i++) // RefData at current_time + i dt (the solver's time argument, src/DdpZmp.cpp:12,23,32)
      {
        const RefData rd = ref_data_funcs[k](current_times[k] + static_cast<double>(i) * horizon_dt_);
        double * r = ref.data() + (k * (N + 1) + i) * 4;
        r[0] = rd.zmp[0];
        r[1] = rd.zmp[1];
        r[2] = rd.zmp[2];
        r[3] = rd.com_z;
      }
      const InitialParam & ip = initial_params[k]; // toState(), src/DdpZmp.cpp:149-154
      const double s[6] = {ip.pos[0], ip.vel[0], ip.pos[1], ip.vel[1], ip.pos[2], ip.vel[2]};
      for(int a = 0; a < 6; a++) x0[k * 6 + a] = s[a];
      if(!ip.u_list.empty())
      {
        if(ip.u_list.size() != N) throw std::runtime_error("[DdpZmp::planOnce] u_list length must be horizon_steps");
        for(size_t i = 0; i < N; i++)
          for(int q = 0; q < 3; q++) u_init[(k * N + i) * 3 + q] = ip.u_list[i][q];
      }
    }
    check(ccc_ddpzmp_set_config(handle_.get(), &ddp_solver_->config_));
    check(ccc_ddpzmp_plan_batch(handle_.get(), static_cast<int64_t>(n), ref.data(), x0.data(), u_init.data(), u_out.data(),
                                nullptr, iters.data(), nullptr, nullptr));
    std::vector<PlannedData> out(n);
    for(size_t k = 0; k < n; k++)
    {
      out[k].zmp = Vector2d(u_out[k * N * 3 + 0], u_out[k * N * 3 + 1]); // src/DdpZmp.cpp:171-172
      out[k].force_z = u_out[k * N * 3 + 2];
    }
    if(n > 0)
    {
      auto & ul = ddp_solver_->control_.u_list;
      ul.resize(N);
      for(size_t i = 0; i < N; i++)
        ul[i] = InputDimVector(u_out[((n - 1) * N + i) * 3], u_out[((n - 1) * N + i) * 3 + 1], u_out[((n - 1) * N + i) * 3 + 2]);
      ddp_solver_->trace_.back().iter = iters[n - 1];
    }
    return out;
  }

  /** \brief The C-ABI handle, for the flat-array entry points of ccc_amd.h. */
  ccc_ddpzmp_t * handle() const
  {
    return handle_.get();
  }

public:
  std::shared_ptr<Solver> ddp_solver_; //!< DDP solver stand-in (config, controlData, traceDataList)
  double mass_ = 0;                    //!< Robot mass [kg]
  double horizon_dt_ = 0;              //!< Discretization timestep in horizon [sec]
  int horizon_steps_ = 0;              //!< Number of steps in horizon

protected:
  static void check(int rc)
  {
    if(rc != CCC_OK)
    {
      throw std::runtime_error(std::string("[CCC::DdpZmp] ") + ccc_last_error_string());
    }
  }

  std::shared_ptr<ccc_ddpzmp_t> handle_;
};
} // namespace CCC
