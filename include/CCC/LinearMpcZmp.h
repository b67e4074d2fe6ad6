/* LinearMpcZmp.h -- drop-in header shim: the class surface of the reference's
 * /root/reference/include/CCC/LinearMpcZmp.h:97-165 over the MI355X C-ABI (include/ccc_amd.h).
 *
 * Same namespace, class, nested types, constructor and planOnce() signature as the reference, so code
 * written against CCC::LinearMpcZmp recompiles against this header and links libccc_amd.so instead of
 * libCCC.so.  Differences, all forced by the missing third-party stack:
 *   - with Eigen available (__has_include(<Eigen/Core>)) the vector types ARE Eigen::Vector2d; without it a
 *     20-line stand-in with x()/y()/operator[] is used;
 *   - the trailing QpSolverCollection::QpSolverType argument is accepted (the real enum when QpSolverCollection is
 *     installed, a stand-in otherwise) and ignored: the QP is solved on the GPU by this library's own exact
 *     active-set kernel; a QP that is not solved is reported on stderr and through lastStatuses() (ShimCommon.h);
 *   - LinearMpcZmp1d (LinearMpcZmp.h:20-90) is provided over the same kernel;
 *   - planOnceBatch() is new: n independent planOnce() problems in one launch.
 * Errors surface as std::runtime_error, like the reference's argument checks.
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
/** \brief QP-based linear MPC for CoM-ZMP model (Wieber 2006), batched on MI355X.
    Mirrors CCC::LinearMpcZmp, /root/reference/include/CCC/LinearMpcZmp.h:97-165. */
class LinearMpcZmp
{
public:
  /** \brief Reference data (LinearMpcZmp.h:105-111). */
  struct RefData
  {
    //! Min/max limits of ZMP [m]
    std::array<Vector2d, 2> zmp_limits;
  };

  /** \brief Initial parameter (LinearMpcZmp.h:113-125). */
  struct InitialParam
  {
    Vector2d pos = Vector2d::Zero(); //!< CoM position [m]
    Vector2d vel = Vector2d::Zero(); //!< CoM velocity [m/s]
    Vector2d acc = Vector2d::Zero(); //!< CoM acceleration [m/s^2]
  };

public:
  /** \brief Constructor (LinearMpcZmp.h:134-142).
      \param com_height height of robot CoM [m]
      \param horizon_duration horizon duration [sec]
      \param horizon_dt discretization timestep in horizon [sec]
      \param qp_solver_type ignored (kept for source compatibility)
      \param device HIP device ordinal (new) */
  LinearMpcZmp(double com_height,
               double horizon_duration,
               double horizon_dt,
               QpSolverCollection::QpSolverType qp_solver_type = QpSolverCollection::QpSolverType::Any,
               int device = 0)
  {
    (void)qp_solver_type;
    ccc_zmp_t * h = nullptr;
    check(ccc_zmp_create(com_height, horizon_duration, horizon_dt, device, &h));
    handle_.reset(h, ccc_zmp_destroy);
    horizon_dt_ = horizon_dt;
    horizon_steps_ = ccc_zmp_horizon_steps(h);
  }

  /** \brief Plan one step (LinearMpcZmp.h:151-154, src/LinearMpcZmp.cpp:83-112).
      \param ref_data_func function of reference data
      \param initial_param initial parameter
      \param current_time current time (i.e., start time of horizon) [sec]
      \param control_dt control timestep used to calculate ZMP (if omitted, horizon_dt is used)
      \returns planned ZMP */
  Vector2d planOnce(const std::function<RefData(double)> & ref_data_func,
                    const InitialParam & initial_param,
                    double current_time,
                    double control_dt = -1)
  {
    std::vector<double> x0(6), zlim(4 * static_cast<size_t>(horizon_steps_));
    pack(initial_param, x0.data());
    sample(ref_data_func, current_time, zlim.data());
    double zmp[2];
    last_status_.assign(2, 0);
    check(ccc_zmp_plan_batch(handle_.get(), 1, x0.data(), zlim.data(), control_dt, zmp, nullptr, last_status_.data()));
    shim::reportStatus("LinearMpcZmp", last_status_);
    return Vector2d(zmp[0], zmp[1]);
  }

  /** \brief Plan n independent instances in one launch (new).
      \param ref_data_funcs one callback per instance, sampled at current_times[k] + i * horizon_dt
      \returns planned ZMP of every instance */
  std::vector<Vector2d> planOnceBatch(const std::vector<std::function<RefData(double)>> & ref_data_funcs,
                                      const std::vector<InitialParam> & initial_params,
                                      const std::vector<double> & current_times,
                                      double control_dt = -1)
  {
    const size_t n = ref_data_funcs.size();
    if(initial_params.size() != n || current_times.size() != n)
    {
      throw std::runtime_error("[LinearMpcZmp::planOnceBatch] argument sizes differ");
    }
    const size_t N = static_cast<size_t>(horizon_steps_);
    std::vector<double> x0(6 * n), zlim(4 * N * n), zmp(2 * n);
    for(size_t k = 0; k < n; k++)
    {
      pack(initial_params[k], x0.data() + 6 * k);
      sample(ref_data_funcs[k], current_times[k], zlim.data() + 4 * N * k);
    }
    last_status_.assign(2 * n, 0);
    check(ccc_zmp_plan_batch(handle_.get(), static_cast<int64_t>(n), x0.data(), zlim.data(), control_dt, zmp.data(),
                             nullptr, last_status_.data()));
    shim::reportStatus("LinearMpcZmp", last_status_);
    std::vector<Vector2d> out(n);
    for(size_t k = 0; k < n; k++) out[k] = Vector2d(zmp[2 * k], zmp[2 * k + 1]);
    return out;
  }

  /** \brief Number of steps in horizon (LinearMpcZmp1d::horizon_steps_). */
  int horizonSteps() const
  {
    return horizon_steps_;
  }

  /** \brief The C-ABI handle, for the flat-array entry points of ccc_amd.h. */
  ccc_zmp_t * handle() const
  {
    return handle_.get();
  }

  /** \brief Solver status of the last call, [instance][axis]: (pivots << 8) | CCC_STATUS_* (new). */
  const std::vector<int32_t> & lastStatuses() const
  {
    return last_status_;
  }

protected:
  static void check(int rc)
  {
    if(rc != CCC_OK)
    {
      throw std::runtime_error(std::string("[LinearMpcZmp] ") + ccc_last_error_string());
    }
  }

  static void pack(const InitialParam & ip, double * x0)
  {
    // src/LinearMpcZmp.cpp:103-108: per axis (pos, vel, acc)
    x0[0] = ip.pos.x();
    x0[1] = ip.vel.x();
    x0[2] = ip.acc.x();
    x0[3] = ip.pos.y();
    x0[4] = ip.vel.y();
    x0[5] = ip.acc.y();
  }

  void sample(const std::function<RefData(double)> & ref_data_func, double current_time, double * zlim) const
  {
    // src/LinearMpcZmp.cpp:86-98 -> layout [axis][min/max][N] of ccc_amd.h
    const int N = horizon_steps_;
    for(int i = 0; i < N; i++)
    {
      double t = current_time + i * horizon_dt_;
      const RefData ref_data = ref_data_func(t);
      for(int j = 0; j < 2; j++)
      {
        zlim[(0 * 2 + j) * N + i] = ref_data.zmp_limits[j].x();
        zlim[(1 * 2 + j) * N + i] = ref_data.zmp_limits[j].y();
      }
    }
  }

protected:
  std::shared_ptr<ccc_zmp_t> handle_;
  double horizon_dt_ = 0;
  int horizon_steps_ = -1;
  std::vector<int32_t> last_status_;
};

/** \brief QP-based linear MPC for the one-dimensional CoM-ZMP model.
    Mirrors CCC::LinearMpcZmp1d, /root/reference/include/CCC/LinearMpcZmp.h:20-90 (planOnce = src/LinearMpcZmp.cpp:30-44
    + procOnce :46-81).  The C-ABI plans both axes of an instance in one wavefront; the 1-d problem rides on the x axis
    and the y axis is given the same (feasible) problem. */
class LinearMpcZmp1d
{
public:
  /** \brief Reference data (LinearMpcZmp.h:26-30). */
  struct RefData
  {
    //! Min/max limits of ZMP [m]
    std::array<double, 2> zmp_limits;
  };

  /** \brief Initial parameter (LinearMpcZmp.h:32-36): CoM position, velocity, acceleration. */
  using InitialParam = Vector3d;

public:
  /** \brief Constructor (LinearMpcZmp.h:45-48). */
  LinearMpcZmp1d(double com_height,
                 double horizon_duration,
                 double horizon_dt,
                 QpSolverCollection::QpSolverType qp_solver_type = QpSolverCollection::QpSolverType::Any,
                 int device = 0)
  {
    (void)qp_solver_type;
    ccc_zmp_t * h = nullptr;
    if(ccc_zmp_create(com_height, horizon_duration, horizon_dt, device, &h) != CCC_OK)
    {
      throw std::runtime_error(std::string("[LinearMpcZmp1d] ") + ccc_last_error_string());
    }
    handle_.reset(h, ccc_zmp_destroy);
    horizon_dt_ = horizon_dt;
    horizon_steps_ = ccc_zmp_horizon_steps(h);
  }

  /** \brief Plan one step (LinearMpcZmp.h:57-60).
      \returns planned ZMP */
  double planOnce(const std::function<RefData(double)> & ref_data_func,
                  const InitialParam & initial_param,
                  double current_time,
                  double control_dt = -1)
  {
    const size_t N = static_cast<size_t>(horizon_steps_);
    std::vector<double> x0(6), zlim(4 * N);
    for(int a = 0; a < 2; a++)
    {
      for(int k = 0; k < 3; k++) x0[3 * a + k] = initial_param[k];
    }
    for(size_t i = 0; i < N; i++)
    {
      const RefData ref_data = ref_data_func(current_time + static_cast<double>(i) * horizon_dt_);
      for(int a = 0; a < 2; a++)
      {
        for(int j = 0; j < 2; j++) zlim[(static_cast<size_t>(a) * 2 + j) * N + i] = ref_data.zmp_limits[j];
      }
    }
    double zmp[2];
    last_status_.assign(2, 0);
    if(ccc_zmp_plan_batch(handle_.get(), 1, x0.data(), zlim.data(), control_dt, zmp, nullptr, last_status_.data())
       != CCC_OK)
    {
      throw std::runtime_error(std::string("[LinearMpcZmp1d] ") + ccc_last_error_string());
    }
    last_status_.resize(1);
    shim::reportStatus("LinearMpcZmp1d", last_status_);
    return zmp[0];
  }

  /** \brief Solver status of the last call: (pivots << 8) | CCC_STATUS_* (new). */
  int32_t lastStatus() const
  {
    return last_status_.empty() ? 0 : last_status_[0];
  }

protected:
  std::shared_ptr<ccc_zmp_t> handle_;
  //! Discretization timestep in horizon [sec]
  double horizon_dt_ = 0;
  //! Number of steps in horizon
  int horizon_steps_ = -1;
  std::vector<int32_t> last_status_;
};
} // namespace CCC
