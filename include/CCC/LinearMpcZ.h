/* LinearMpcZ.h -- drop-in header shim: the class surface of /root/reference/include/CCC/LinearMpcZ.h:14-179 over the
 * MI355X C-ABI (include/ccc_amd.h).  Same namespace, class, InitialParam (= Vector2d: position, velocity), WeightParam,
 * constructor and planOnce() signature; the QpSolverCollection::QpSolverType argument is accepted as an int and
 * ignored.  planOnceBatch() is new: n independent planOnce() problems in one launch.
 */
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../ccc_amd.h"

#include "EigenLite.h"
#include "ShimCommon.h"

namespace CCC
{
/** \brief QP-based linear MPC for the vertical CoM motion, batched on MI355X.
    Mirrors CCC::LinearMpcZ, /root/reference/include/CCC/LinearMpcZ.h:14-179. */
class LinearMpcZ
{
public:
  /** \brief State dimension (LinearMpcZ.h:18). */
  static constexpr int state_dim_ = 2;

  /** \brief Initial parameter: position and velocity (LinearMpcZ.h:31). */
  using InitialParam = Vector2d;

  /** \brief Weight parameter (LinearMpcZ.h:34-47, same defaults). */
  struct WeightParam
  {
    double pos;   //!< Position weight
    double force; //!< Force weight
    WeightParam(double _pos = 1.0, double _force = 1e-7) : pos(_pos), force(_force) {}
  };

public:
  /** \brief Constructor (LinearMpcZ.h:127-131, src/LinearMpcZ.cpp:31-46).
      \param mass robot mass [kg]
      \param horizon_dt discretization timestep in horizon [sec]
      \param horizon_steps number of steps in horizon (<= CCC_Z_MAX_STEPS_WIDE)
      \param weight_param objective weight parameter
      \param qp_solver_type ignored (kept for source compatibility)
      \param device HIP device ordinal (new) */
  LinearMpcZ(double mass,
             double horizon_dt,
             int horizon_steps,
             const WeightParam & weight_param = WeightParam(),
             QpSolverCollection::QpSolverType qp_solver_type = QpSolverCollection::QpSolverType::Any,
             int device = 0)
  : mass_(mass), horizon_dt_(horizon_dt), horizon_steps_(horizon_steps), weight_param_(weight_param),
    force_range_(10.0, 10.0 * mass * 9.80665)
  {
    (void)qp_solver_type;
    ccc_z_t * h = nullptr;
    check(ccc_z_create(mass, horizon_dt, horizon_steps, weight_param.pos, weight_param.force, device, &h));
    handle_.reset(h, ccc_z_destroy);
  }

  /** \brief Plan one step (LinearMpcZ.h:140-143, src/LinearMpcZ.cpp:48-71).
      \param contact_func function to return whether it is in contact phase
      \param ref_pos_func function of reference position [m]
      \param initial_param initial parameter (position, velocity)
      \param current_time current time (i.e., start time of horizon) [sec]
      \returns planned force */
  double planOnce(const std::function<bool(double)> & contact_func,
                  const std::function<double(double)> & ref_pos_func,
                  const InitialParam & initial_param,
                  double current_time)
  {
    const size_t N = static_cast<size_t>(horizon_steps_);
    std::vector<int32_t> contact(N);
    std::vector<double> ref(N);
    sample(contact_func, ref_pos_func, current_time, contact.data(), ref.data());
    const double x0[2] = {initial_param[0], initial_param[1]};
    double force = 0;
    last_status_.assign(1, 0);
    check(ccc_z_plan_batch(handle_.get(), 1, contact.data(), ref.data(), x0, &force, nullptr, last_status_.data()));
    shim::reportStatus("LinearMpcZ", last_status_);
    return force;
  }

  /** \brief Plan n independent instances in one launch (new). */
  std::vector<double> planOnceBatch(const std::vector<std::function<bool(double)>> & contact_funcs,
                                    const std::vector<std::function<double(double)>> & ref_pos_funcs,
                                    const std::vector<InitialParam> & initial_params,
                                    const std::vector<double> & current_times)
  {
    const size_t n = contact_funcs.size(), N = static_cast<size_t>(horizon_steps_);
    if(ref_pos_funcs.size() != n || initial_params.size() != n || current_times.size() != n)
    {
      throw std::runtime_error("[LinearMpcZ::planOnceBatch] argument sizes differ");
    }
    std::vector<int32_t> contact(n * N);
    std::vector<double> ref(n * N), x0(2 * n), force(n);
    for(size_t k = 0; k < n; k++)
    {
      sample(contact_funcs[k], ref_pos_funcs[k], current_times[k], contact.data() + k * N, ref.data() + k * N);
      x0[2 * k] = initial_params[k][0];
      x0[2 * k + 1] = initial_params[k][1];
    }
    last_status_.assign(n, 0);
    check(ccc_z_plan_batch(handle_.get(), static_cast<int64_t>(n), contact.data(), ref.data(), x0.data(), force.data(),
                           nullptr, last_status_.data()));
    shim::reportStatus("LinearMpcZ", last_status_);
    return force;
  }

  /** \brief The C-ABI handle, for the flat-array entry points of ccc_amd.h. */
  ccc_z_t * handle() const
  {
    return handle_.get();
  }

  /** \brief Solver status of the last call, one per instance: (pivots << 8) | CCC_STATUS_* (new). */
  const std::vector<int32_t> & lastStatuses() const
  {
    return last_status_;
  }

public:
  double mass_ = 0;                        //!< Robot mass [kg]
  double horizon_dt_ = 0;                  //!< Discretization timestep in horizon [sec]
  int horizon_steps_ = 0;                  //!< Number of steps in horizon
  WeightParam weight_param_;               //!< Weight parameter
  std::pair<double, double> force_range_;  //!< Min/max force (src/LinearMpcZ.cpp:37)

protected:
  void sample(const std::function<bool(double)> & contact_func,
              const std::function<double(double)> & ref_pos_func,
              double current_time,
              int32_t * contact,
              double * ref) const
  {
    // src/LinearMpcZ.cpp:59-67
    for(int i = 0; i < horizon_steps_; i++)
    {
      const double t = current_time + i * horizon_dt_;
      contact[i] = contact_func(t) ? 1 : 0;
      ref[i] = ref_pos_func(t);
    }
  }

  static void check(int rc)
  {
    if(rc != CCC_OK)
    {
      throw std::runtime_error(std::string("[LinearMpcZ] ") + ccc_last_error_string());
    }
  }

protected:
  std::shared_ptr<ccc_z_t> handle_;
  std::vector<int32_t> last_status_;
};
} // namespace CCC
