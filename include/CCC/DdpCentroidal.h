/* DdpCentroidal.h -- drop-in header shim: the class surface of /root/reference/include/CCC/DdpCentroidal.h:13-366 over
 * the MI355X C-ABI (include/ccc_amd.h).  Same namespace, class, nested MotionParam / RefData / WeightParam /
 * InitialParam, constructor and planOnce() signature; ddp_solver_ exposes config().max_iter, controlData().u_list and
 * traceDataList().back().iter like the nmpc_ddp solver the reference holds; ddp_problem_ exposes dt() / inputDim(t).
 * ForceColl::Contact / Eigen are used when installed, otherwise the stand-ins of EigenLite.h.
 */
#pragma once

#include "DdpShimBase.h"

namespace CCC
{
/** \brief Differential dynamic programming (DDP) for centroidal model, batched on MI355X. */
class DdpCentroidal
{
public:
  /** \brief Motion parameter (DdpCentroidal.h:19-25). */
  struct MotionParam
  {
    std::vector<std::shared_ptr<Contact>> contact_list;
  };

  /** \brief Reference data (DdpCentroidal.h:28-34). */
  struct RefData
  {
    Vector3d pos = Vector3d::Zero(); //!< CoM position [m]
  };

  /** \brief Weight parameter (DdpCentroidal.h:37-81, same defaults). */
  struct WeightParam
  {
    Vector3d running_pos, running_linear_momentum, running_angular_momentum;
    double running_force;
    Vector3d terminal_pos, terminal_linear_momentum, terminal_angular_momentum;

    WeightParam(const Vector3d & _running_pos = Vector3d::Constant(1.0),
                const Vector3d & _running_linear_momentum = Vector3d::Constant(0.0),
                const Vector3d & _running_angular_momentum = Vector3d::Constant(1.0),
                double _running_force = 1e-6,
                const Vector3d & _terminal_pos = Vector3d::Constant(1.0),
                const Vector3d & _terminal_linear_momentum = Vector3d::Constant(0.0),
                const Vector3d & _terminal_angular_momentum = Vector3d::Constant(1.0))
    : running_pos(_running_pos), running_linear_momentum(_running_linear_momentum),
      running_angular_momentum(_running_angular_momentum), running_force(_running_force), terminal_pos(_terminal_pos),
      terminal_linear_momentum(_terminal_linear_momentum), terminal_angular_momentum(_terminal_angular_momentum)
    {
    }
  };

  /** \brief The part of DdpCentroidal::DdpProblem callers use (dt(), inputDim(t), mass_). */
  class DdpProblem
  {
  public:
    DdpProblem(double horizon_dt, double mass) : mass_(mass), dt_(horizon_dt) {}
    double dt() const
    {
      return dt_;
    }
    int stateDim() const
    {
      return 9;
    }
    /** src/DdpCentroidal.cpp:21-30 (valid after planOnce stored the callback) */
    int inputDim(double t) const
    {
      int input_dim = 0;
      for(const auto & contact : motion_param_func_(t).contact_list) input_dim += contact->ridgeNum();
      return input_dim;
    }
    double mass_ = 0;
    std::function<MotionParam(double)> motion_param_func_;
    std::function<RefData(double)> ref_data_func_;

  protected:
    double dt_ = 0;
  };

  /** \brief Initial parameter (DdpCentroidal.h:295-330). */
  struct InitialParam
  {
    Vector3d pos = Vector3d::Zero();              //!< CoM position [m]
    Vector3d vel = Vector3d::Zero();              //!< CoM velocity [m/s]
    Vector3d angular_momentum = Vector3d::Zero(); //!< Angular momentum [kg m^2/s]
    std::vector<VectorXd> u_list = {};            //!< initial guess of the input sequence (empty: zeros)

    /** src/DdpCentroidal.cpp:186-191 */
    std::vector<double> toState(double mass) const
    {
      return {pos[0], pos[1], pos[2], mass * vel[0], mass * vel[1], mass * vel[2],
              angular_momentum[0], angular_momentum[1], angular_momentum[2]};
    }
  };

public:
  /** \brief Constructor (DdpCentroidal.h:342, src/DdpCentroidal.cpp:193-211). */
  DdpCentroidal(double mass, double horizon_dt, int horizon_steps, const WeightParam & weight_param = WeightParam(),
                int device = 0, int max_phases = 4)
  : ddp_problem_(std::make_shared<DdpProblem>(horizon_dt, mass)), ddp_solver_(std::make_shared<ddp_shim::Solver>()),
    horizon_steps_(horizon_steps), max_phases_(max_phases)
  {
    ccc_ddp_params_t p{};
    p.model = CCC_DDP_CENTROIDAL;
    p.mass = mass;
    p.horizon_dt = horizon_dt;
    p.horizon_steps = horizon_steps;
    for(int a = 0; a < 3; a++)
    {
      p.w_run[a] = weight_param.running_pos[a];
      p.w_run[3 + a] = weight_param.running_linear_momentum[a];
      p.w_run[6 + a] = weight_param.running_angular_momentum[a];
      p.w_term[a] = weight_param.terminal_pos[a];
      p.w_term[3 + a] = weight_param.terminal_linear_momentum[a];
      p.w_term[6 + a] = weight_param.terminal_angular_momentum[a];
    }
    p.w_force = weight_param.running_force;
    p.force_scale_limits[0] = force_scale_limits_[0];
    p.force_scale_limits[1] = force_scale_limits_[1];
    p.max_phases = max_phases;
    handles_.create(p, device, "DdpCentroidal");
    ddp_solver_->config().horizon_steps = horizon_steps; // src/DdpCentroidal.cpp:198
    ddp_solver_->config().max_iter = 500; // nmpc_ddp default
  }

  /** \brief Plan one step (DdpCentroidal.h:351-354, src/DdpCentroidal.cpp:213-237).
      \returns planned force scales */
  VectorXd planOnce(const std::function<MotionParam(double)> & motion_param_func,
                    const std::function<RefData(double)> & ref_data_func,
                    const InitialParam & initial_param,
                    double current_time)
  {
    ddp_problem_->motion_param_func_ = motion_param_func;
    ddp_problem_->ref_data_func_ = ref_data_func;
    ddp_shim::Flat f;
    f.init(horizon_steps_);
    for(int i = 0; i <= horizon_steps_; i++)
    {
      const double t = current_time + i * ddp_problem_->dt();
      const RefData ref = ref_data_func(t);
      for(int a = 0; a < 3; a++) f.ref_pos[static_cast<size_t>(i) * 3 + a] = ref.pos[a];
      if(i < horizon_steps_) f.setStepContacts(i, motion_param_func(t).contact_list);
    }
    return ddp_shim::solveOne(handles_.select(f, "DdpCentroidal"), *ddp_solver_, f, false, initial_param.toState(ddp_problem_->mass_),
                              initial_param.u_list, force_scale_limits_, "DdpCentroidal");
  }

  /** \brief The C-ABI handle (<= 16 ridges per step, <= max_phases phases), for the flat-array batch entry points of
      ccc_amd.h.  planOnce() itself takes any contact list up to 64 ridges per step and any number of phases: what the
      constructor's tables do not hold goes to a further handle with the ridge stride it needs (16, 32 or 64), created on first
      need (ddp_shim::Handles). */
  ccc_ddp_t * handle() const
  {
    return handles_.fast.get();
  }

public:
  //! DDP problem
  std::shared_ptr<DdpProblem> ddp_problem_;

  //! DDP solver
  std::shared_ptr<ddp_shim::Solver> ddp_solver_;

  //! Force scale limits (DdpCentroidal.h:364).  Live, as in the reference, whose input-limits lambda reads the member at
  //! every solve (src/DdpCentroidal.cpp:202-210): planOnce() hands its current value to the handle (ccc_ddp_set_limits)
  std::array<double, 2> force_scale_limits_ = {0.0, 1e6};

protected:
  ddp_shim::Handles handles_;
  int horizon_steps_ = 0;
  int max_phases_ = 4;
};
} // namespace CCC
