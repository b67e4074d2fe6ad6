/* DdpSingleRigidBody.h -- drop-in header shim: the class surface of
 * /root/reference/include/CCC/DdpSingleRigidBody.h over the MI355X C-ABI (include/ccc_amd.h).
 * See DdpCentroidal.h in this directory for the conventions.
 */
#pragma once

#include "DdpShimBase.h"

namespace CCC
{
/** \brief Differential dynamic programming (DDP) for single rigid-body model, batched on MI355X. */
class DdpSingleRigidBody
{
public:
  /** \brief Motion parameter (DdpSingleRigidBody.h:25-34). */
  struct MotionParam
  {
    std::vector<std::shared_ptr<Contact>> contact_list;
    Matrix3d inertia_mat = Matrix3d::Identity(); //!< inertia matrix in world frame [kg m^2]
  };

  /** \brief Reference data (DdpSingleRigidBody.h:37-46). */
  struct RefData
  {
    Vector3d pos = Vector3d::Zero(); //!< CoM position [m]
    Vector3d ori = Vector3d::Zero(); //!< base link orientation, ZYX Euler angles [rad]
  };

  /** \brief Weight parameter (DdpSingleRigidBody.h:49-110, same defaults). */
  struct WeightParam
  {
    Vector3d running_pos, running_ori, running_linear_vel, running_angular_vel;
    double running_force;
    Vector3d terminal_pos, terminal_ori, terminal_linear_vel, terminal_angular_vel;

    WeightParam(const Vector3d & _running_pos = Vector3d::Constant(1.0),
                const Vector3d & _running_ori = Vector3d::Constant(1.0),
                const Vector3d & _running_linear_vel = Vector3d::Constant(0.01),
                const Vector3d & _running_angular_vel = Vector3d::Constant(0.01),
                double _running_force = 1e-6,
                const Vector3d & _terminal_pos = Vector3d::Constant(1.0),
                const Vector3d & _terminal_ori = Vector3d::Constant(1.0),
                const Vector3d & _terminal_linear_vel = Vector3d::Constant(0.01),
                const Vector3d & _terminal_angular_vel = Vector3d::Constant(0.01))
    : running_pos(_running_pos), running_ori(_running_ori), running_linear_vel(_running_linear_vel),
      running_angular_vel(_running_angular_vel), running_force(_running_force), terminal_pos(_terminal_pos),
      terminal_ori(_terminal_ori), terminal_linear_vel(_terminal_linear_vel),
      terminal_angular_vel(_terminal_angular_vel)
    {
    }
  };

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
      return 12;
    }
    /** src/DdpSingleRigidBody.cpp:40-50 */
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

  /** \brief Initial parameter. */
  struct InitialParam
  {
    Vector3d pos = Vector3d::Zero();
    Vector3d ori = Vector3d::Zero();
    Vector3d linear_vel = Vector3d::Zero();
    Vector3d angular_vel = Vector3d::Zero();
    std::vector<VectorXd> u_list = {};

    /** src/DdpSingleRigidBody.cpp:253-258 */
    std::vector<double> toState() const
    {
      return {pos[0], pos[1], pos[2], ori[0], ori[1], ori[2], linear_vel[0], linear_vel[1], linear_vel[2],
              angular_vel[0], angular_vel[1], angular_vel[2]};
    }
  };

public:
  /** \brief Constructor (src/DdpSingleRigidBody.cpp:260-281). */
  DdpSingleRigidBody(double mass, double horizon_dt, int horizon_steps,
                     const WeightParam & weight_param = WeightParam(), int device = 0, int max_phases = 4)
  : ddp_problem_(std::make_shared<DdpProblem>(horizon_dt, mass)), ddp_solver_(std::make_shared<ddp_shim::Solver>()),
    horizon_steps_(horizon_steps), max_phases_(max_phases)
  {
    ccc_ddp_params_t p{};
    p.model = CCC_DDP_SINGLE_RIGID_BODY;
    p.mass = mass;
    p.horizon_dt = horizon_dt;
    p.horizon_steps = horizon_steps;
    for(int a = 0; a < 3; a++)
    {
      p.w_run[a] = weight_param.running_pos[a];
      p.w_run[3 + a] = weight_param.running_ori[a];
      p.w_run[6 + a] = weight_param.running_linear_vel[a];
      p.w_run[9 + a] = weight_param.running_angular_vel[a];
      p.w_term[a] = weight_param.terminal_pos[a];
      p.w_term[3 + a] = weight_param.terminal_ori[a];
      p.w_term[6 + a] = weight_param.terminal_linear_vel[a];
      p.w_term[9 + a] = weight_param.terminal_angular_vel[a];
    }
    p.w_force = weight_param.running_force;
    p.force_scale_limits[0] = force_scale_limits_[0];
    p.force_scale_limits[1] = force_scale_limits_[1];
    p.max_phases = max_phases;
    p.inertia_per_phase = 1; // a phase = a distinct MotionParam: contact list and inertia_mat (ddp_shim::Flat)
    handles_.create(p, device, "DdpSingleRigidBody");
    ddp_solver_->config().horizon_steps = horizon_steps; // src/DdpCentroidal.cpp:198
    ddp_solver_->config().max_iter = 500;
  }

  /** \brief Plan one step (src/DdpSingleRigidBody.cpp:283-307).
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
      for(int a = 0; a < 3; a++)
      {
        f.ref_pos[static_cast<size_t>(i) * 3 + a] = ref.pos[a];
        f.ref_ori[static_cast<size_t>(i) * 3 + a] = ref.ori[a];
      }
      if(i < horizon_steps_)
      {
        // MotionParam::inertia_mat of EVERY step, as stateEq / calcStateEqDeriv read it
        // (src/DdpSingleRigidBody.cpp:56-57,120-123)
        const MotionParam mp = motion_param_func(t);
        double inertia9[9];
        for(int r = 0; r < 3; r++)
          for(int c = 0; c < 3; c++) inertia9[r * 3 + c] = mp.inertia_mat(r, c);
        f.setStepContacts(i, mp.contact_list, inertia9);
      }
    }
    return ddp_shim::solveOne(handles_.select(f, "DdpSingleRigidBody"), *ddp_solver_, f, true, initial_param.toState(), initial_param.u_list,
                              force_scale_limits_, "DdpSingleRigidBody");
  }

  /** \brief The C-ABI handle of the fast kernel (see DdpCentroidal::handle()); created with inertia_per_phase = 1: its
      `inertia` argument is [n][max_phases][3][3] (ccc_ddp_set_inertia_per_phase(handle(), 0) for one matrix per instance). */
  ccc_ddp_t * handle() const
  {
    return handles_.fast.get();
  }

public:
  std::shared_ptr<DdpProblem> ddp_problem_;
  std::shared_ptr<ddp_shim::Solver> ddp_solver_;
  //! Force scale limits: live, read at every planOnce() like the reference's (src/DdpSingleRigidBody.cpp:272-280)
  std::array<double, 2> force_scale_limits_ = {0.0, 1e6};

protected:
  ddp_shim::Handles handles_;
  int horizon_steps_ = 0;
  int max_phases_ = 4;
};
} // namespace CCC
