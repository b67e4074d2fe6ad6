/* LinearMpcXY.h -- drop-in header shim: the class surface of /root/reference/include/CCC/LinearMpcXY.h:27-257 over
 * the MI355X C-ABI (include/ccc_amd.h).  Same namespace, class, nested MotionParam / InitialParam / RefData /
 * WeightParam, constructor and planOnce() signature; the trailing QpSolverCollection::QpSolverType argument is
 * accepted as an int and ignored (the QP is solved by this library's own exact active-set kernel).
 * ForceColl::Contact / Eigen are used when installed, otherwise the stand-ins of EigenLite.h.
 * planOnceBatch() is new: n independent planOnce() problems in one launch.
 */
#pragma once

#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../ccc_amd.h"

#include "EigenLite.h"
#include "ShimCommon.h"

namespace CCC
{
/** \brief QP-based linear MPC for the horizontal centroidal motion (Audren 2014 / Nagasaka 2012), batched on MI355X.
    Mirrors CCC::LinearMpcXY, /root/reference/include/CCC/LinearMpcXY.h:27-257. */
class LinearMpcXY
{
public:
  /** \brief State dimension (LinearMpcXY.h:31). */
  static constexpr int state_dim_ = 6;

  /** \brief Motion parameter (LinearMpcXY.h:38-52). */
  struct MotionParam
  {
    double com_z = 0;         //!< CoM z position [m]
    double total_force_z = 0; //!< Total z force [N]
    std::vector<std::shared_ptr<Contact>> contact_list;
  };

  /** \brief Initial parameter (LinearMpcXY.h:55-73). */
  struct InitialParam
  {
    Vector2d pos = Vector2d::Zero();              //!< CoM position [m]
    Vector2d vel = Vector2d::Zero();              //!< CoM linear velocity [m/s]
    Vector2d angular_momentum = Vector2d::Zero(); //!< Angular momentum [kg m^2/s]

    /** src/LinearMpcXY.cpp:26-31 */
    std::array<double, 6> toState(double mass) const
    {
      return {mass * pos.x(), mass * vel.x(), mass * pos.y(), mass * vel.y(), angular_momentum.x(),
              angular_momentum.y()};
    }
  };

  /** \brief Reference data (LinearMpcXY.h:76-101). */
  struct RefData
  {
    Vector2d pos = Vector2d::Zero();
    Vector2d vel = Vector2d::Zero();
    Vector2d angular_momentum = Vector2d::Zero();

    static constexpr int outputDim()
    {
      return 6;
    }

    /** src/LinearMpcXY.cpp:33-38 */
    std::array<double, 6> toOutput(double mass) const
    {
      return {mass * pos.x(), mass * vel.x(), mass * pos.y(), mass * vel.y(), angular_momentum.x(),
              angular_momentum.y()};
    }
  };

  /** \brief Weight parameter (LinearMpcXY.h:104-142, same defaults). */
  struct WeightParam
  {
    Vector2d linear_momentum_integral, linear_momentum, angular_momentum;
    double force;

    WeightParam(const Vector2d & _linear_momentum_integral = Vector2d(1.0, 1.0),
                const Vector2d & _linear_momentum = Vector2d(0.0, 0.0),
                const Vector2d & _angular_momentum = Vector2d(1.0, 1.0),
                double _force = 1e-5)
    : linear_momentum_integral(_linear_momentum_integral), linear_momentum(_linear_momentum),
      angular_momentum(_angular_momentum), force(_force)
    {
    }
  };

public:
  /** \brief Constructor (LinearMpcXY.h:211-215, src/LinearMpcXY.cpp:85-94).
      \param mass robot mass [kg]
      \param horizon_dt discretization timestep in horizon [sec]
      \param horizon_steps number of steps in horizon (<= CCC_XY_MAX_STEPS_WIDE)
      \param weight_param objective weight parameter
      \param qp_solver_type ignored (kept for source compatibility)
      \param device HIP device ordinal (new) */
  LinearMpcXY(double mass,
              double horizon_dt,
              int horizon_steps,
              const WeightParam & weight_param = WeightParam(),
              QpSolverCollection::QpSolverType qp_solver_type = QpSolverCollection::QpSolverType::Any,
              int device = 0)
  : mass_(mass), horizon_dt_(horizon_dt), horizon_steps_(horizon_steps), weight_param_(weight_param)
  {
    (void)qp_solver_type;
    ccc_xy_params_t p{};
    p.mass = mass;
    p.horizon_dt = horizon_dt;
    p.horizon_steps = horizon_steps;
    for(int a = 0; a < 2; a++)
    {
      p.w_lmi[a] = weight_param.linear_momentum_integral[a];
      p.w_lm[a] = weight_param.linear_momentum[a];
      p.w_am[a] = weight_param.angular_momentum[a];
    }
    p.w_force = weight_param.force;
    params_ = p;
    device_ = device;
    ccc_xy_t * h = nullptr;
    check(ccc_xy_create(&p, device, &h));
    handle_.reset(h, ccc_xy_destroy);
    force_range_[1] = 3.0 * mass * 9.80665;
  }

  /** \brief Plan one step (LinearMpcXY.h:224-227, src/LinearMpcXY.cpp:96-114).
      \param motion_param_func function of motion parameter
      \param ref_data_func function of reference data
      \param initial_param initial parameter
      \param current_time current time (i.e., start time of horizon) [sec]
      \returns planned force scales of the first horizon step */
  VectorXd planOnce(const std::function<MotionParam(double)> & motion_param_func,
                    const std::function<RefData(double)> & ref_data_func,
                    const InitialParam & initial_param,
                    double current_time)
  {
    Flat f(1, horizon_steps_);
    const int m0 = sample(f, 0, motion_param_func, ref_data_func, initial_param, current_time);
    ccc_xy_t * const h = select(f);
    std::vector<double> u0(static_cast<size_t>(f.M));
    last_status_.assign(1, 0);
    check(ccc_xy_plan_batch(h, 1, f.dim.data(), f.vertex.data(), f.ridge.data(), f.com_z.data(),
                            f.total_force_z.data(), f.ref_out.data(), f.x0.data(), u0.data(), nullptr,
                            last_status_.data()));
    shim::reportStatus("LinearMpcXY", last_status_);
    VectorXd out(m0);
    for(int r = 0; r < m0; r++) out[r] = u0[static_cast<size_t>(r)];
    return out;
  }

  /** \brief Plan n independent instances in one launch (new).
      \returns planned force scales of the first horizon step of every instance */
  std::vector<VectorXd> planOnceBatch(const std::vector<std::function<MotionParam(double)>> & motion_param_funcs,
                                      const std::vector<std::function<RefData(double)>> & ref_data_funcs,
                                      const std::vector<InitialParam> & initial_params,
                                      const std::vector<double> & current_times)
  {
    const size_t n = motion_param_funcs.size();
    if(ref_data_funcs.size() != n || initial_params.size() != n || current_times.size() != n)
    {
      throw std::runtime_error("[LinearMpcXY::planOnceBatch] argument sizes differ");
    }
    Flat f(n, horizon_steps_);
    std::vector<int> m0(n);
    for(size_t k = 0; k < n; k++)
      m0[k] = sample(f, k, motion_param_funcs[k], ref_data_funcs[k], initial_params[k], current_times[k]);
    ccc_xy_t * const h = select(f);
    const size_t M = static_cast<size_t>(f.M);
    std::vector<double> u0(n * M);
    last_status_.assign(n, 0);
    check(ccc_xy_plan_batch(h, static_cast<int64_t>(n), f.dim.data(), f.vertex.data(), f.ridge.data(),
                            f.com_z.data(), f.total_force_z.data(), f.ref_out.data(), f.x0.data(), u0.data(), nullptr,
                            last_status_.data()));
    shim::reportStatus("LinearMpcXY", last_status_);
    std::vector<VectorXd> out(n);
    for(size_t k = 0; k < n; k++)
    {
      out[k] = VectorXd(m0[k]);
      for(int r = 0; r < m0[k]; r++) out[k][r] = u0[k * M + static_cast<size_t>(r)];
    }
    return out;
  }

  /** \brief The C-ABI handle (16 ridge slots per step), for the flat-array entry points of ccc_amd.h.  planOnce() and
      planOnceBatch() take any contact list up to 64 ridges per step: what 16 slots do not hold goes to a further handle
      with max_ridges = 32 or 64, created on first need. */
  ccc_xy_t * handle() const
  {
    return handle_.get();
  }

  /** \brief Solver status of the last call, one per instance: (pivots << 8) | CCC_STATUS_* (new). */
  const std::vector<int32_t> & lastStatuses() const
  {
    return last_status_;
  }

public:
  //! Robot mass [kg]
  const double mass_ = 0;

  //! Discretization timestep in horizon [sec]
  const double horizon_dt_ = 0;

  //! Number of steps in horizon
  const int horizon_steps_ = 0;

  //! Min/max scale of ridge force (src/LinearMpcXY.cpp:91)
  std::array<double, 2> force_range_ = {3.0, 0.0}; // upper = 3 m g, set in the constructor

protected:
  std::vector<int32_t> last_status_;

protected:
  /** Flat arrays of ccc_xy_plan_batch for n instances; sampled with 64 ridge slots per step, see select(). */
  struct Flat
  {
    Flat(size_t n, int N)
    : dim(n * N, 0), vertex(n * N * CCC_XY_MAX_RIDGES_MULTI * 3, 0.0), ridge(n * N * CCC_XY_MAX_RIDGES_MULTI * 3, 0.0),
      com_z(n * N, 0.0), total_force_z(n * N, 0.0), ref_out(n * N * 6, 0.0), x0(n * 6, 0.0)
    {
    }
    std::vector<int32_t> dim;
    std::vector<double> vertex, ridge, com_z, total_force_z, ref_out, x0;
    int M = CCC_XY_MAX_RIDGES_MULTI; //!< ridge slots per step of vertex / ridge
  };

  /** The handle that takes the sampled problems: the smallest ridge stride (16, 32 or 64 slots per step) that holds
      every sampled contact list -- the arrays are compacted to that stride; handles beyond the constructor's 16 are
      created on first need.  The reference takes any contact_list (src/LinearMpcXY.cpp:69-82). */
  ccc_xy_t * select(Flat & f)
  {
    int32_t mx = 0;
    for(int32_t d : f.dim) mx = d > mx ? d : mx;
    const int k = mx <= CCC_XY_MAX_RIDGES ? 0 : (mx <= CCC_XY_MAX_RIDGES_WIDE ? 1 : 2);
    const size_t Mn = k == 0 ? CCC_XY_MAX_RIDGES : (k == 1 ? CCC_XY_MAX_RIDGES_WIDE : CCC_XY_MAX_RIDGES_MULTI);
    const size_t steps = f.dim.size(), Mw = static_cast<size_t>(f.M);
    if(Mn < Mw)
    {
      for(size_t e = 0; e < steps; e++)
        for(size_t j = 0; j < Mn * 3; j++)
        {
          f.vertex[e * Mn * 3 + j] = f.vertex[e * Mw * 3 + j];
          f.ridge[e * Mn * 3 + j] = f.ridge[e * Mw * 3 + j];
        }
      f.M = static_cast<int>(Mn);
    }
    if(k == 0) return handle_.get();
    std::shared_ptr<ccc_xy_t> & twin = k == 1 ? wide_handle_ : multi_handle_;
    if(!twin)
    {
      ccc_xy_params_t p = params_;
      p.max_ridges = static_cast<int>(Mn);
      ccc_xy_t * h = nullptr;
      check(ccc_xy_create(&p, device_, &h));
      twin.reset(h, ccc_xy_destroy);
    }
    return twin.get();
  }

  /** src/LinearMpcXY.cpp:102-110 (sampling) and :69-82 (contact -> vertex -> ridge order); returns dim of step 0 */
  int sample(Flat & f,
             size_t k,
             const std::function<MotionParam(double)> & motion_param_func,
             const std::function<RefData(double)> & ref_data_func,
             const InitialParam & initial_param,
             double current_time) const
  {
    const size_t N = static_cast<size_t>(horizon_steps_), M = CCC_XY_MAX_RIDGES_MULTI;
    for(size_t i = 0; i < N; i++)
    {
      const double t = current_time + static_cast<double>(i) * horizon_dt_;
      const MotionParam mp = motion_param_func(t);
      size_t r = 0;
      for(const auto & contact : mp.contact_list)
      {
        for(const auto & vr : contact->vertexWithRidgeList_)
        {
          for(const auto & rd : vr.ridgeList)
          {
            if(r >= M)
              throw std::runtime_error("[LinearMpcXY] more than 64 ridges in one contact list (the kernels are built for four "
                                       "4-vertex surface contacts)");
            for(int a = 0; a < 3; a++)
            {
              f.vertex[((k * N + i) * M + r) * 3 + static_cast<size_t>(a)] = vr.vertex[a];
              f.ridge[((k * N + i) * M + r) * 3 + static_cast<size_t>(a)] = rd[a];
            }
            r++;
          }
        }
      }
      f.dim[k * N + i] = static_cast<int32_t>(r);
      f.com_z[k * N + i] = mp.com_z;
      f.total_force_z[k * N + i] = mp.total_force_z;
      const auto ref = ref_data_func(t).toOutput(mass_);
      for(size_t a = 0; a < 6; a++) f.ref_out[(k * N + i) * 6 + a] = ref[a];
    }
    const auto x = initial_param.toState(mass_);
    for(size_t a = 0; a < 6; a++) f.x0[k * 6 + a] = x[a];
    return f.dim[k * N];
  }

  static void check(int rc)
  {
    if(rc != CCC_OK)
    {
      throw std::runtime_error(std::string("[LinearMpcXY] ") + ccc_last_error_string());
    }
  }

protected:
  WeightParam weight_param_;
  std::shared_ptr<ccc_xy_t> handle_, wide_handle_, multi_handle_;
  ccc_xy_params_t params_{};
  int device_ = 0;
};
} // namespace CCC
